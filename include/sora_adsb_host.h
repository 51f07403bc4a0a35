/*
 * sora_adsb_host.h -- the reference's own C entry points for the ADS-B path,
 * kept by name, argument meaning and error behaviour, implemented on top of
 * the gfx950 library (sora_adsb_gpu.h).  libsora_adsb_host.so.
 *
 *   B1  src/ads-b/plane-iq.h:10-13        demodulator iterator
 *   B2  src/ads-b/planes-main-loop.h:6-9  mode entry
 *   B3  src/ads-b/plane.h:12-38           validator, plane state, printer
 *       (with src/util/iterator.h:4-14 for plane_set_iterate)
 *   B4  src/radio/radio.h:11-25, radio-file.h:6-14   file / stdin frontend
 *
 * A reference build links these in place of plane-iq.c / plane.c /
 * planes-main-loop.c / radio-file.c / radio.c; the reference's
 * own, unmodified planes-main-loop.c runs on top of B1 + B3 as well
 * (oracle/Makefile: _ref/ref_loop_on_gpu, tests/test_host.py).
 */
#ifndef SORA_ADSB_HOST_H_
#define SORA_ADSB_HOST_H_

#include <complex.h>
#include <stdint.h>
#include <stdio.h>
#include <sys/time.h>
#include <sys/types.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- B1: reference src/ads-b/plane-iq.h:10-13 ---- */
struct plane_iq_state;

/* uc8 from `f`, like the reference (plane-iq.c:88, :121-139).  NULL (with a
 * message on stderr) when no gfx950 device can be used: there is no CPU path. */
struct plane_iq_state *plane_iq_state_new(FILE *);
void plane_iq_state_delete(struct plane_iq_state *);
/* Next candidate as 112 '0'/'1' characters, valid until the next call; NULL at
 * the end of the stream.  The reference's semantics are the default: EVERY
 * candidate that passes the preamble test is yielded, in position order,
 * whether plane_set_parse_message() will accept it or not (plane-iq.c:141-194,
 * :187-192).  NULL is also returned when the device fails; plane_iq_error()
 * tells the two apart (the reference's iterator cannot fail). */
const char *plane_iq_get_next(struct plane_iq_state *, struct timeval *, double *);

/* extension: other encodings of radio-file.h:6-12 */
struct plane_iq_state *plane_iq_state_new_encoded(FILE *, int encoding);
/* extension: only the frames plane_set_parse_message() accepts -- the device
 * applies the acceptance rule (plane.c:473-484) and the rejected 99 % never
 * cross PCIe.  What this library's planes_main_loop() uses: it drops the
 * rejected ones without side effects anyway (planes-main-loop.c:75-77).
 * SORA_ADSB_EMIT_ALL=1 / =0 in the environment overrides the choice made by
 * the constructor, for all three. */
struct plane_iq_state *plane_iq_state_new_accepted(FILE *, int encoding);
/* position (plane-iq.c:47) of the frame last returned */
unsigned long long plane_iq_last_position(const struct plane_iq_state *);
/* 0, or the SORA_ERR_* code that ended the stream early */
int plane_iq_error(const struct plane_iq_state *);

/* ---- B2: reference src/ads-b/planes-main-loop.h:6-9 ---- */
void planes_main_loop(FILE *, int flags);
#define PLANES_INPUT_FROM_RAW		0x01
#define PLANES_OUTPUT_AS_DECODED	0x10
#define PLANES_OUTPUT_AS_BITSTRING	0x20
/* extension: raw input in another encoding; returns 0, or the SORA_ERR_* code
 * of a device failure (which must not pass for the end of the input) */
int planes_main_loop_encoded(FILE *, int flags, int encoding);
/* status of the last planes_main_loop() / _encoded() call (the reference's
 * entry point is void) */
int planes_main_loop_status(void);

/* ---- B3: reference src/ads-b/plane.h:12-38 ---- */
struct plane;
struct plane_set;

struct plane_set *plane_set_new(void);
void plane_set_delete(struct plane_set *);
void plane_set_free(struct plane_set *);
struct plane *plane_set_lookup_create(struct plane_set *, uint32_t address);
struct plane *plane_set_parse_message(struct plane_set *, const char *,
	struct timeval, double);
/* src/util/iterator.h:4-14; plane.h:18.  Yields every struct plane * of the set
 * (next), or address and plane (next_pair: the key is the 24-bit address cast
 * to a pointer, plane.c:101); order unspecified, as the reference's hash order
 * is.  The set must not change while an iterator is live (hash.c:221). */
struct iterator {
    int (*is_at_end)(struct iterator *);
    void *(*next)(struct iterator *);
    void (*next_pair)(struct iterator *, const void **, void **);
    void (*free)(struct iterator *);
};
#define iterator_is_at_end(it) ((it)->is_at_end((it)))
#define iterator_next(it) ((it)->next((it)))
#define iterator_next_pair(it, a, b) ((it)->next_pair((it), (a), (b)))
#define iterator_delete(it) ((it)->free((it)))
struct iterator *plane_set_iterate(struct plane_set *);

struct plane *plane_clone(struct plane *);
void plane_delete(struct plane *);
void plane_print(struct plane *);
const char *plane_get_category(struct plane *);
const char *plane_get_squawk_id(struct plane *);
const char *plane_get_flight_id(struct plane *);
int plane_is_ground_velocity_available(struct plane *);
int plane_is_heading_available(struct plane *);
double plane_get_ground_velocity(struct plane *);
double plane_get_heading(struct plane *);
int plane_is_altitude_available(struct plane *);
long plane_get_altitude(struct plane *);
int plane_is_gnss_vs_baro_available(struct plane *);
int plane_get_gnss_vs_baro(struct plane *);
int plane_is_vert_rate_available(struct plane *);
int plane_get_vert_rate(struct plane *);
int plane_is_lat_long_available(struct plane *);
double plane_get_latitude(struct plane *);
double plane_get_longitude(struct plane *);
/* extension: plane_print into a buffer (no newline); returns the length */
int plane_snprint(struct plane *, char *buf, size_t len);

/* ---- B4: reference src/radio/radio.h:11-25, radio-file.h:6-14 ---- */
typedef uint64_t t_frequency;			/* src/common/frequency.h */
struct sample { double complex v; };		/* src/signal/sample.h:6-8 */

struct radio_methods;
struct radio {
    struct radio_methods *m;
};
/* radio.h:15: installs a vtable, filling empty slots with dummies that return
 * -1 (0 for get_file_position; close does nothing), radio.c:18-86 */
void radio_init(struct radio *, struct radio_methods *);
struct radio_methods {
    int (*set_frequency)(struct radio *, t_frequency);
    int (*get_frequency)(struct radio *, t_frequency *);
    int (*set_sample_rate)(struct radio *, unsigned long);
    int (*get_sample_rate)(struct radio *, unsigned long *);
    ssize_t (*read)(struct radio *, struct sample *, size_t);
    off_t (*get_file_position)(struct radio *);
    void (*close)(struct radio *);
};

#define RADIO_FILE_ENCODING_UC8		1
#define RADIO_FILE_ENCODING_SC16	2
#define RADIO_FILE_ENCODING_SC8		3
#define RADIO_FILE_ENCODING_U8		32
#define RADIO_FILE_ENCODING_S16		33

/* "-" is stdin (radio-file.c:40-41); NULL after perror() when it cannot open */
struct radio *radio_file_open(const char *, int encoding);
/* extension: what the ADS-B branch needs of a file radio */
FILE *radio_file_stream(struct radio *);
int radio_file_encoding(struct radio *);

#ifdef __cplusplus
}
#endif

#endif /* SORA_ADSB_HOST_H_ */

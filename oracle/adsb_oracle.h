/*
 * TEST INFRASTRUCTURE ONLY.
 *
 * CPU restatement of Sora's ADS-B receive chain, used as the parity oracle
 * for the MI355X path.  Only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may load this library; the product (sora_amd/) never does.
 *
 * Every function cites the reference lines it restates
 * (paths relative to /root/reference).  The restatement treats the chain as a
 * pure function of the sample array instead of the reference's ring buffer +
 * in-flight list; tests/test_oracle_vs_reference.py pins it against the
 * compiled reference (oracle/_ref) and tests/golden/ holds the vectors that
 * the compiled reference produced.
 *
 * Parity status: uc8 path PINNED against the compiled reference.
 * sc8 / sc16 routing into ADS-B does not exist in the reference
 * (SURVEY.md D4): for those encodings this file is the definition, anchored
 * on the reference's scaling constants (src/radio/radio-file.c:136-141) --
 * "parity unpinned" by the reference.
 */
#ifndef ADSB_ORACLE_H_
#define ADSB_ORACLE_H_

#include <stddef.h>
#include <stdint.h>
#include <stdio.h>

/* encodings: same numbers as reference src/radio/radio-file.h:6-12 */
#define ORACLE_ENC_UC8	1
#define ORACLE_ENC_SC16	2
#define ORACLE_ENC_SC8	3

#define ORACLE_MSG_BITS	112
/* a candidate at p reads samples p .. p+239 */
#define ORACLE_WINDOW	240

struct oracle_frame {
    uint64_t position;			/* in_flight_msg.position, plane-iq.c:47 */
    char bits[ORACLE_MSG_BITS + 1];	/* '0'/'1' + NUL, plane-iq.c:75 */
    double amplitude_56;		/* plane-iq.c:72-73 */
    double amplitude;			/* amplitude_56 / 56, plane-iq.c:189 */
    int32_t df;				/* bits[0,5), plane.c:461 */
    int32_t long_message;		/* plane.c:462 */
    uint32_t syndrome;			/* crc ^ ap, plane.c:472-475, :480 */
    uint32_t address;			/* plane.c:474 or :480 */
    int32_t accepted;			/* parse_message returned non-NULL */
    int32_t corrected_bit;		/* -1, or the bit oracle_set_correction(1)
					 * made the oracle flip (extension D3) */
};

/* a1: plane-iq.c:121-139 (uc8); sc8/sc16 by radio-file.c:136-141 scaling.
 * Returns the number of samples written (an odd trailing byte / partial
 * sample is dropped, plane-iq.c:131, :138). */
size_t oracle_magnitude(const void *iq, size_t nbytes, int encoding, float *m);

/* a3: plane-iq.c:173-175.  Requires p + 9 < n. */
int oracle_preamble(const float *m, size_t p);

/* a5/a6: plane-iq.c:58-86, :187-189.  Requires p + 240 <= n. */
void oracle_slice(const float *m, size_t p, char *bits113,
	double *amplitude_56);

/* a8: plane.c:373-385 */
uint32_t oracle_crc24(const char *bits, int size);

/* plane set: plane.c:57-105; never expires */
struct oracle_plane_set;
struct oracle_plane_set *oracle_plane_set_new(void);
void oracle_plane_set_delete(struct oracle_plane_set *);
size_t oracle_plane_set_count(const struct oracle_plane_set *);

/*
 * The demodulator as a pure function: every p in increasing order with
 * oracle_preamble(p) and p + 240 <= n (plane-iq.c:141-194; SURVEY.md
 * section 4 facts), validated in order by the acceptance rule of
 * plane.c:457-484 against `pset` (which it updates; field decode included).
 * Writes up to `cap` frames, returns the number of frames that WOULD have
 * been written with unlimited cap.  accepted_only != 0 keeps only frames the
 * validator accepted.  base_position is added to reported positions and used
 * for stream time.
 */
/*
 * Extension D3, OFF by default: the reference never
 * corrects.  When on, a DF11/17/18 candidate whose syndrome equals that of a
 * single flipped bit b, 5 <= b < 56 (DF11) or 112 (DF17/18), has that bit
 * flipped back and is parsed again; oracle_frame.syndrome keeps the syndrome
 * before the flip.  A definition of this build ("parity unpinned").
 */
void oracle_set_correction(int on);
/* the flipped bit a syndrome points at in an nbits-long message, or -1 */
int oracle_correctable_bit(uint32_t syndrome, int nbits);

size_t oracle_scan(const float *m, size_t n, uint64_t base_position,
	struct oracle_plane_set *pset, int accepted_only,
	struct oracle_frame *out, size_t cap);

/* number of preamble hits only (for rate facts and the timing leg) */
size_t oracle_count_candidates(const float *m, size_t n);

/*
 * planes_main_loop, planes-main-loop.c:49-99, FROM_RAW mode, with stream
 * time t = t0 + position / rate (see oracle/ref_shim_plane_iq.c).
 * flags: 0x10 decoded, 0x20 bit string (planes-main-loop.h:7-9).
 * Returns the number of accepted frames.
 */
size_t oracle_main_loop(const float *m, size_t n, int flags, long long t0,
	unsigned long long rate, FILE *out);

/* a9-a11 on one frame: plane.c:457-596.  Returns 1 if accepted. */
int oracle_parse_message(struct oracle_plane_set *pset, const char *bits,
	long long tv_sec, long long tv_usec, double amplitude,
	struct oracle_frame *info);

/* a13: plane.c:598-632 into buf (no trailing newline); returns length.
 * Prints the plane last touched by oracle_parse_message(). */
int oracle_print_last(struct oracle_plane_set *pset, char *buf, size_t len);

#endif /* ADSB_ORACLE_H_ */

/*
 * planes_main_loop.c -- the ADS-B mode entry of reference
 * src/ads-b/planes-main-loop.h:6-9: pull frames, validate, print.
 * Raw input goes through the GPU demodulator (plane_iq_gpu.c).  The ASCII
 * input mode of the reference (planes-main-loop.c:15-47) is host-only text
 * parsing outside the hot path; it is kept so that `sora --adsb-decode` without
 * --adsb-from-raw still reads what --adsb-to-bitstring wrote.
 *
 * The reference's loop cannot fail: its demodulator is a few lines of C.  Here
 * the device can (no GPU, out of memory), and a failure must not look like the
 * end of the input: planes_main_loop_encoded() returns the SORA_ERR_* code,
 * planes_main_loop() (void in the reference) keeps it for
 * planes_main_loop_status(), and `sora` exits non-zero.
 */
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include "sora_adsb_gpu.h"
#include "sora_adsb_host.h"

#define MESSAGE_BITS	112
#define LINE_MAX_CHARS	(MESSAGE_BITS + 100)

static int last_status;

/*
 * One line of the --adsb-to-bitstring format, "<bits> @ sec,usec,0,amplitude"
 * (planes-main-loop.c:79-82), split at the '@' and the commas: the message is
 * what stands before " @" (at most 112 characters), the time is field 0 in
 * whole seconds, the amplitude field 3 when there are that many.  Empty lines
 * and UHD driver chatter are skipped; a line without '@' ends the input, as in
 * planes-main-loop.c:28-30.  Returns 0 at the end of the input.
 */
static int
read_bitstring_line(FILE *f, char *line, struct timeval *tv, double *amplitude) {
    char *fields[4] = { NULL, NULL, NULL, NULL };
    char *at, *p;
    size_t len;
    int n = 0;

    for (;;) {
	if (fgets(line, LINE_MAX_CHARS, f) == NULL)
	    return 0;
	if (line[0] != '\0' && line[0] != '\n' && strstr(line, "UHD_") == NULL)
	    break;
    }
    if ((at = strchr(line, '@')) == NULL)
	return 0;
    for (p = at + 1; n < 4; p = strchr(p, ',') + 1) {
	fields[n++] = p;
	if (strchr(p, ',') == NULL)
	    break;
    }
    /* the reference takes the second field's start from strtoll's end pointer:
     * more than one field only if the seconds are followed directly by ',' */
    tv->tv_sec = (time_t) strtoll(fields[0], &p, 10);
    tv->tv_usec = 0;
    if (*p == ',' && n == 4)
	*amplitude = strtod(fields[3], NULL);
    len = (size_t) (at - line) - 1;
    line[len < MESSAGE_BITS? len : MESSAGE_BITS] = '\0';
    return 1;
}

int
planes_main_loop_encoded(FILE *f, int flags, int encoding) {
    struct plane_set *planes = plane_set_new();
    struct plane_iq_state *iq = NULL;
    char text[LINE_MAX_CHARS];
    int shown_day = -1, status = SORA_OK;
    double amplitude = 0;
    struct timeval tv;

    if (flags & PLANES_INPUT_FROM_RAW) {
	iq = plane_iq_state_new_accepted(f, encoding);
	if (iq == NULL)			/* reported on stderr; nothing can be decoded */
	    return last_status = SORA_ERR_NO_DEVICE;
    }

    for (;;) {
	const char *msg;
	struct plane *pl;

	if (iq != NULL) {
	    msg = plane_iq_get_next(iq, &tv, &amplitude);
	    if (msg == NULL)
		break;
	} else {
	    if (!read_bitstring_line(f, text, &tv, &amplitude))
		break;
	    msg = text;
	}

	pl = plane_set_parse_message(planes, msg, tv, amplitude);
	if (pl == NULL)
	    continue;				/* planes-main-loop.c:76-77 */

	if (flags & PLANES_OUTPUT_AS_BITSTRING)	/* planes-main-loop.c:79-82 */
	    printf("%s @ %lld,%lld,%llu,%g\n", msg, (long long) tv.tv_sec,
		    (long long) tv.tv_usec, 0ull, amplitude);

	if (flags & PLANES_OUTPUT_AS_DECODED) {	/* planes-main-loop.c:84-94 */
	    struct tm tm;
	    char line[256];

	    localtime_r(&tv.tv_sec, &tm);
	    if (tm.tm_yday != shown_day) {
		printf("--- %d-%02d-%02d ---\n", tm.tm_year + 1900, tm.tm_mon + 1,
			tm.tm_mday);
		shown_day = tm.tm_yday;
	    }
	    plane_snprint(pl, line, sizeof line);
	    printf("%s\n", line);
	}
    }

    if (iq != NULL) {
	status = plane_iq_error(iq);		/* 0 unless the device ended the stream */
	plane_iq_state_delete(iq);
    }
    /* the reference never frees the plane set on this path either */
    return last_status = status;
}

void
planes_main_loop(FILE *f, int flags) {
    (void) planes_main_loop_encoded(f, flags, SORA_ENC_UC8);
}

int
planes_main_loop_status(void) {
    return last_status;
}

/*
 * planes_main_loop.c -- the ADS-B mode entry of reference
 * src/ads-b/planes-main-loop.h:6-9: pull frames, validate, print.
 * Raw input goes through the GPU demodulator (plane_iq_gpu.c); the ASCII
 * input mode of the reference (planes-main-loop.c:15-47) is host-only text
 * parsing and is kept as it is specified there.
 */
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include "sora_adsb_gpu.h"
#include "sora_adsb_host.h"

#define MESSAGE_BITS	112
#define LINE_MAX_CHARS	(MESSAGE_BITS + 100)

/* one "<bits> @ sec,usec,0,amplitude" line, as --adsb-to-bitstring writes
 * them (planes-main-loop.c:79-82); 0 at end of input */
static int
read_bitstring_line(FILE *f, char *line, struct timeval *tv, double *amplitude) {
    char *at, *p;
    size_t len;

    do {
	if (fgets(line, LINE_MAX_CHARS, f) == NULL)
	    return 0;
    } while (line[0] == '\0' || line[0] == '\n' || strstr(line, "UHD_") != NULL);

    at = strchr(line, '@');
    if (at == NULL)
	return 0;
    len = (size_t) (at - line) - 1;
    if (len > MESSAGE_BITS)
	len = MESSAGE_BITS;
    line[len] = '\0';
    tv->tv_sec = (time_t) strtoll(at + 1, &p, 10);
    tv->tv_usec = 0;
    if (*p == ',' && (p = strchr(p + 1, ',')) != NULL && (p = strchr(p + 1, ',')) != NULL)
	*amplitude = strtod(p + 1, NULL);
    return 1;
}

void
planes_main_loop_encoded(FILE *f, int flags, int encoding) {
    struct plane_set *planes = plane_set_new();
    struct plane_iq_state *iq = NULL;
    char text[LINE_MAX_CHARS];
    int shown_day = -1;
    double amplitude = 0;
    struct timeval tv;

    if (flags & PLANES_INPUT_FROM_RAW) {
	iq = plane_iq_state_new_encoded(f, encoding);
	if (iq == NULL)
	    return;		/* reported on stderr; nothing can be decoded */
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

    plane_iq_state_delete(iq);
    /* the reference never frees the plane set on this path either */
}

void
planes_main_loop(FILE *f, int flags) {
    planes_main_loop_encoded(f, flags, SORA_ENC_UC8);
}

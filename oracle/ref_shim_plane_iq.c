/*
 * TEST INFRASTRUCTURE ONLY -- the reference demodulator with stream time.
 *
 * plane_iq_ifm_new() stamps every candidate with gettimeofday()
 * (reference src/ads-b/plane-iq.c:46), which makes decoded text
 * (CPR pairing, plane.c:397-401; HH:MM:SS, plane.c:628-629) depend on the
 * wall clock.  This translation unit compiles the UNMODIFIED reference file
 * with gettimeofday replaced, at its single call site, by a clock derived
 * from the sample position:
 *     t = T0 + position / RATE
 * `state` is in scope at that call site.  T0 and RATE come from the
 * environment (SORA_T0 seconds, default 1700000000; SORA_RATE samples/s,
 * default 2000000), the same variables the sora_amd host library reads.
 */
#include <sys/time.h>
#include <stdlib.h>

struct plane_iq_state;
static int ref_shim_time(unsigned long long position, struct timeval *tv);

#define gettimeofday(tvp, tzp) ref_shim_time(state->position, (tvp))
#include "ads-b/plane-iq.c"
#undef gettimeofday

static int
ref_shim_time(unsigned long long position, struct timeval *tv) {
    static long long t0 = -1;
    static unsigned long long rate;

    if (t0 < 0) {
	const char *s = getenv("SORA_T0");
	const char *r = getenv("SORA_RATE");
	t0 = s != NULL? atoll(s) : 1700000000ll;
	rate = r != NULL? strtoull(r, NULL, 10) : 2000000ull;
	if (rate == 0)
	    rate = 2000000ull;
    }
    tv->tv_sec = t0 + (long long) (position / rate);
    tv->tv_usec = (long long) ((position % rate) * 1000000ull / rate);

    return 0;
}

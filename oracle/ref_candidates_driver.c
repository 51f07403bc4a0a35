/*
 * TEST INFRASTRUCTURE ONLY -- candidate dump from the compiled reference.
 *
 * The reference never prints the sample position of a candidate
 * (the printf at src/ads-b/plane-iq.c:190 is commented out).  This driver
 * includes the UNMODIFIED reference file textually so that
 * `struct in_flight_msg` is visible, recovers the node from the string
 * plane_iq_get_next() returns, and prints one line per candidate:
 *
 *   <position> <112 bits> <amplitude_56 as %a> <amplitude as %a> <accepted>
 *
 * `accepted` is 1 when the reference validator
 * plane_set_parse_message() (src/ads-b/plane.c:457) returns a plane, i.e.
 * when planes_main_loop() would have printed the frame
 * (src/ads-b/planes-main-loop.c:75-77).
 *
 * usage: ref_candidates [--accepted-only] < capture.uc8
 */
#include <stddef.h>
#include <string.h>

#include "ads-b/plane-iq.c"

#include <ads-b/plane.h>
#include <util/hash.h>

int
main(int argc, char *argv[]) {
    struct plane_iq_state *st;
    struct plane_set *pset;
    const char *msg;
    struct timeval tv;
    double amplitude;
    int accepted_only = argc > 1 && strcmp(argv[1], "--accepted-only") == 0;

    hash_init();
    pset = plane_set_new();
    st = plane_iq_state_new(stdin);

    while ((msg = plane_iq_get_next(st, &tv, &amplitude)) != NULL) {
	const struct in_flight_msg *m = (const struct in_flight_msg *)
		(msg - offsetof(struct in_flight_msg, msg));
	int accepted =
		plane_set_parse_message(pset, msg, tv, amplitude) != NULL;

	if (accepted_only && !accepted)
	    continue;
	printf("%llu %s %a %a %d\n", m->position, msg, m->amplitude_56,
		amplitude, accepted);
    }
    fprintf(stderr, "final_position %llu\n", st->position);
    plane_iq_state_delete(st);

    return 0;
}

/*
 * TEST INFRASTRUCTURE ONLY -- driver for the compiled reference (oracle/_ref).
 *
 * Runs the reference ADS-B chain exactly as src/main.c does for
 * `sora --adsb-decode --adsb-from-raw [--adsb-to-bitstring]`:
 *   hash_init()                       (reference src/main.c:172)
 *   planes_main_loop(stdin, flags)    (reference src/main.c:265-275)
 *
 * The reference sources are compiled where they lie under /root/reference
 * (see oracle/Makefile); nothing from them is copied here.  main.c itself
 * cannot be built (it needs GTK2/FFTW3), so this file
 * restates only its ADS-B branch.
 *
 * usage: ref_sora_adsb [--adsb-from-raw] [--adsb-to-bitstring] < capture.uc8
 *        (without --adsb-from-raw the input is the ASCII bit-string format)
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <ads-b/planes-main-loop.h>
#include <util/hash.h>

int
main(int argc, char *argv[]) {
    int flags = 0;
    int to_bitstring = 0;
    int i;

    for (i = 1; i < argc; i++) {
	if (strcmp(argv[i], "--adsb-to-bitstring") == 0)
	    to_bitstring = 1;
	else if (strcmp(argv[i], "--adsb-from-raw") == 0)
	    flags |= PLANES_INPUT_FROM_RAW;
	else if (strcmp(argv[i], "--adsb-decode") == 0 || strcmp(argv[i], "-d") == 0)
	    ;
	else {
	    fprintf(stderr, "unknown argument %s\n", argv[i]);
	    return EXIT_FAILURE;
	}
    }
    flags |= to_bitstring? PLANES_OUTPUT_AS_BITSTRING : PLANES_OUTPUT_AS_DECODED;

    hash_init();
    planes_main_loop(stdin, flags);

    return EXIT_SUCCESS;
}

/*
 * main.c -- `sora` for the ADS-B path on MI355X: the option table of reference
 * src/main.c:73-109 restricted to what that path uses, plus the routing
 * north_star asks for: with --file / --file-encoding the ADS-B
 * branch reads that file in that encoding instead of uc8 from stdin
 * (reference: src/main.c:265-275 returns before --file is opened).
 *
 *   sora --adsb-decode --adsb-from-raw [--adsb-to-bitstring] < capture.uc8
 *   sora -d --adsb-from-raw --file=capture.sc16 --file-encoding=sc16
 *
 * Scan, GUI and the hardware frontends are outside this build.
 */
#include <getopt.h>
#include <stdlib.h>
#include <string.h>

#include "sora_adsb_gpu.h"
#include "sora_adsb_host.h"

enum { OPT_FILE = 256, OPT_FILE_ENCODING };

static int adsb_from_raw, adsb_to_bitstring;

static const struct option long_options[] = {
    { "adsb-decode", no_argument, NULL, 'd' },
    { "adsb-from-raw", no_argument, &adsb_from_raw, 1 },
    { "adsb-to-bitstring", no_argument, &adsb_to_bitstring, 1 },
    { "file", required_argument, NULL, OPT_FILE },
    { "file-encoding", required_argument, NULL, OPT_FILE_ENCODING },
    { "quiet", no_argument, NULL, 'q' },
    { "verbose", no_argument, NULL, 'v' },
    { NULL, 0, NULL, 0 }
};

static void
usage(const char *argv0) {
    fprintf(stderr,
	    "usage: %s --adsb-decode [--adsb-from-raw] [--adsb-to-bitstring]\n"
	    "          [--file=NAME|-] [--file-encoding=uc8|sc8|sc16]\n"
	    "%s\n", argv0, sora_adsb_version());
    exit(EXIT_FAILURE);
}

int
main(int argc, char *argv[]) {
    const char *file_name = NULL;
    int encoding = 0, adsb_decode = 0, c, flags = 0, status;
    struct radio *radio = NULL;
    FILE *in = stdin;

    while ((c = getopt_long(argc, argv, "dqv", long_options, NULL)) != -1) {
	switch (c) {
	case 'd':
	    adsb_decode = 1;
	    break;
	case 'q':
	case 'v':
	case 0:
	    break;
	case OPT_FILE:
	    file_name = optarg;
	    break;
	case OPT_FILE_ENCODING:		/* src/main.c:215-231 */
	    if (strcmp(optarg, "uc8") == 0)
		encoding = RADIO_FILE_ENCODING_UC8;
	    else if (strcmp(optarg, "sc8") == 0)
		encoding = RADIO_FILE_ENCODING_SC8;
	    else if (strcmp(optarg, "sc16") == 0)
		encoding = RADIO_FILE_ENCODING_SC16;
	    else if (strcmp(optarg, "u8") == 0)
		encoding = RADIO_FILE_ENCODING_U8;
	    else if (strcmp(optarg, "s16") == 0)
		encoding = RADIO_FILE_ENCODING_S16;
	    else
		fprintf(stderr, "Unknown encoding %s\n", optarg);
	    break;
	default:
	    usage(argv[0]);
	}
    }
    if (!adsb_decode) {
	fprintf(stderr, "%s: only --adsb-decode is built here\n", argv[0]);
	usage(argv[0]);
    }

    flags |= adsb_to_bitstring? PLANES_OUTPUT_AS_BITSTRING : PLANES_OUTPUT_AS_DECODED;
    if (adsb_from_raw)
	flags |= PLANES_INPUT_FROM_RAW;		/* src/main.c:266-272 */

    if (file_name != NULL) {
	radio = radio_file_open(file_name, encoding != 0? encoding : RADIO_FILE_ENCODING_UC8);
	if (radio == NULL) {
	    fprintf(stderr, "can't open file\n");
	    return EXIT_FAILURE;
	}
	in = radio_file_stream(radio);
    }
    if (encoding == 0)
	encoding = RADIO_FILE_ENCODING_UC8;	/* what plane-iq.c assumes */
    if (adsb_from_raw && encoding != RADIO_FILE_ENCODING_UC8 &&
	    encoding != RADIO_FILE_ENCODING_SC8 && encoding != RADIO_FILE_ENCODING_SC16) {
	fprintf(stderr, "%s: ADS-B needs complex samples (uc8, sc8 or sc16)\n", argv[0]);
	return EXIT_FAILURE;
    }

    status = planes_main_loop_encoded(in, flags, encoding);
    fflush(stdout);
    if (radio != NULL)
	radio->m->close(radio);
    if (status != SORA_OK) {
	/* the device failed: what was printed is not the whole capture */
	fprintf(stderr, "%s: ADS-B decoding failed: %s\n", argv[0], sora_adsb_strerror(status));
	return EXIT_FAILURE;
    }
    return EXIT_SUCCESS;
}

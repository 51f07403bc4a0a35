/*
 * TEST INFRASTRUCTURE ONLY -- drives the REFERENCE's file frontend
 * (src/radio/radio-file.c, compiled where it lies by oracle/Makefile) and
 * prints what it reads: one "%a %a" line (re, im) per sample.
 *
 *     ref_radio_file FILE ENCODING [SAMPLES_PER_READ]
 *
 * ENCODING is the number of radio-file.h:6-12; the samples go through
 * radio->m->read() exactly as src/main.c:293-340 does, in reads of
 * SAMPLES_PER_READ (default 300) so that partial reads are exercised.
 */
#include <radio/radio-file.h>

#include <complex.h>
#include <stdio.h>
#include <stdlib.h>

int
main(int argc, char **argv) {
    struct radio *radio;
    struct sample *buf;
    size_t chunk = 300;
    ssize_t got;

    if (argc < 3) {
	fprintf(stderr, "usage: %s FILE ENCODING [SAMPLES_PER_READ]\n", argv[0]);
	return 2;
    }
    if (argc > 3)
	chunk = (size_t) atol(argv[3]);
    radio = radio_file_open(argv[1], atoi(argv[2]));
    if (radio == NULL)
	return 1;
    buf = malloc(chunk * sizeof *buf);
    while ((got = radio->m->read(radio, buf, chunk)) > 0) {
	ssize_t i;
	for (i = 0; i < got; i++)
	    printf("%a %a\n", creal(buf[i].v), cimag(buf[i].v));
    }
    fprintf(stderr, "last_read %ld\n", (long) got);
    radio->m->close(radio);
    free(buf);
    return 0;
}

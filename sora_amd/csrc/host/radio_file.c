/*
 * radio_file.c -- the file / stdin frontend of reference
 * src/radio/radio-file.h:6-14 behind the radio vtable of src/radio/radio.h:11-25.
 * The ADS-B branch only needs the open stream and its encoding
 * (radio_file_stream / radio_file_encoding): the samples themselves go to the
 * device as raw bytes, not through read().  read() is kept for callers of the
 * vtable and widens to `double complex` with the scalings of
 * radio-file.c:131-148.
 */
#include <stdlib.h>
#include <string.h>

#include "sora_adsb_host.h"

struct file_radio {
    struct radio radio;		/* first: a struct radio * is a struct file_radio * */
    FILE *file;
    t_frequency frequency;
    unsigned long rate;
    int encoding;
};

static struct file_radio *
self(struct radio *r) {
    return (struct file_radio *) r;
}

static int set_frequency(struct radio *r, t_frequency f) { self(r)->frequency = f; return 0; }
static int get_frequency(struct radio *r, t_frequency *f) { *f = self(r)->frequency; return 0; }
static int set_rate(struct radio *r, unsigned long s) { self(r)->rate = s; return 0; }
static int get_rate(struct radio *r, unsigned long *s) { *s = self(r)->rate; return 0; }

static int
item_bytes(int encoding) {
    switch (encoding) {
    case RADIO_FILE_ENCODING_UC8:
    case RADIO_FILE_ENCODING_SC8:
    case RADIO_FILE_ENCODING_S16:
	return 2;
    case RADIO_FILE_ENCODING_SC16:
	return 4;
    case RADIO_FILE_ENCODING_U8:
	return 1;
    default:
	return 0;
    }
}

static ssize_t
file_read(struct radio *r, struct sample *out, size_t len) {
    struct file_radio *fr = self(r);
    const int size = item_bytes(fr->encoding);
    const unsigned char *u8 = (const unsigned char *) out;
    const signed char *s8 = (const signed char *) out;
    const int16_t *s16 = (const int16_t *) out;
    size_t got;
    long i;

    if (size == 0) {
	fprintf(stderr, "unsupported encoding\n");	/* radio-file.c:117-119 */
	return 0;
    }
    /* whole items only (radio-file.c:122); the raw items sit at the start of
     * the output buffer and are widened from the back so none is overwritten */
    got = fread(out, (size_t) size, len, fr->file);
    if (got < 1)
	return 0;
    for (i = (long) got - 1; i >= 0; i--) {
	double re, im = 0;

	switch (fr->encoding) {
	case RADIO_FILE_ENCODING_UC8:
	    re = (u8[2 * i] - 128.0) / 128.0;
	    im = (u8[2 * i + 1] - 128.0) / 128.0;
	    break;
	case RADIO_FILE_ENCODING_SC8:
	    re = s8[2 * i] / 128.0;
	    im = s8[2 * i + 1] / 128.0;
	    break;
	case RADIO_FILE_ENCODING_SC16:
	    re = s16[2 * i] / 32768.0;
	    im = s16[2 * i + 1] / 32768.0;
	    break;
	case RADIO_FILE_ENCODING_U8:
	    re = (u8[i] - 128.0) / 128.0;
	    break;
	default:				/* S16 */
	    re = s16[i] / 32768.0;
	    break;
	}
	out[i].v = re + im * I;
    }
    return (ssize_t) got;
}

static off_t file_position(struct radio *r) { return ftello(self(r)->file); }

static void
file_close(struct radio *r) {
    fclose(self(r)->file);		/* stdin included, radio-file.c:161-167 */
    free(r);
}

static struct radio_methods file_methods = {
    set_frequency, get_frequency, set_rate, get_rate, file_read, file_position, file_close
};

/* the vtable core of radio.c:18-86: a frontend leaves the slots it does not
 * implement NULL and radio_init() fills them in */
static int no_set_frequency(struct radio *r, t_frequency f) { (void) r; (void) f; return -1; }
static int no_get_frequency(struct radio *r, t_frequency *f) { (void) r; (void) f; return -1; }
static int no_set_rate(struct radio *r, unsigned long s) { (void) r; (void) s; return -1; }
static int no_get_rate(struct radio *r, unsigned long *s) { (void) r; (void) s; return -1; }
static ssize_t no_read(struct radio *r, struct sample *b, size_t n)
{ (void) r; (void) b; (void) n; return -1; }
static off_t no_position(struct radio *r) { (void) r; return 0; }
static void no_close(struct radio *r) { (void) r; }

void
radio_init(struct radio *r, struct radio_methods *m) {
    if (m->set_frequency == NULL) m->set_frequency = no_set_frequency;
    if (m->get_frequency == NULL) m->get_frequency = no_get_frequency;
    if (m->set_sample_rate == NULL) m->set_sample_rate = no_set_rate;
    if (m->get_sample_rate == NULL) m->get_sample_rate = no_get_rate;
    if (m->read == NULL) m->read = no_read;
    if (m->get_file_position == NULL) m->get_file_position = no_position;
    if (m->close == NULL) m->close = no_close;
    r->m = m;
}

struct radio *
radio_file_open(const char *name, int encoding) {
    struct file_radio *fr = calloc(1, sizeof *fr);

    if (fr == NULL)
	abort();
    if (strcmp(name, "-") == 0)
	fr->file = stdin;			/* radio-file.c:40-41 */
    else if ((fr->file = fopen(name, "rb")) == NULL) {
	perror(name);
	free(fr);
	return NULL;
    }
    fr->rate = 1;				/* radio-file.c:52 */
    fr->encoding = encoding;
    radio_init(&fr->radio, &file_methods);	/* radio-file.c:55 */
    return &fr->radio;
}

FILE *radio_file_stream(struct radio *r) { return self(r)->file; }
int radio_file_encoding(struct radio *r) { return self(r)->encoding; }

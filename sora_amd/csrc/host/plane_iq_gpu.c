/*
 * plane_iq_gpu.c -- the demodulator iterator of reference
 * src/ads-b/plane-iq.h:10-13 on top of the gfx950 library.
 *
 * The reference pulls one sample at a time through a 4096-float ring
 * (plane-iq.c:141-194).  Here the library's streaming ingest
 * (sora_adsb_stream_*) reads the FILE in blocks of tens of millions of
 * samples into pinned memory, double buffered against the kernels, every
 * block together with the last 239 samples of the previous one (a candidate
 * at p reads p .. p+239); the frames the device returns are yielded one per
 * call, in position order -- the order in which the reference completes them
 * (plane-iq.c:76-83).
 *
 * Which frames: the reference's iterator yields EVERY candidate, accepted or
 * not (plane-iq.c:187-192), and so do plane_iq_state_new() and
 * plane_iq_state_new_encoded(): a caller written against the reference (its own
 * planes-main-loop.c linked against this library)
 * sees what it saw before.  This library's planes_main_loop() drops rejected
 * candidates without side effects (planes-main-loop.c:75-77), so it opens the
 * state with plane_iq_state_new_accepted(): the device applies the acceptance
 * rule and only the accepted frames cross PCIe.  SORA_ADSB_EMIT_ALL=1 / =0
 * forces either behaviour for both.
 *
 * Time: the reference stamps a candidate with gettimeofday() when it meets
 * the preamble (plane-iq.c:46), i.e. with the time at which it happened to
 * process that sample.  Here the stamp is stream time,
 *     t = T0 + position / RATE,
 * T0 = gettimeofday() when the state is created unless SORA_T0 is set,
 * RATE = SORA_RATE or 2 MS/s: the same for every run over the same capture.
 */
#include <stdlib.h>
#include <string.h>

#include "sora_adsb_gpu.h"
#include "sora_adsb_host.h"

#define DEFAULT_BLOCK_SAMPLES	(32u << 20)	/* 64 MB of uc8 per submit */

struct plane_iq_state {
    FILE *f;
    int encoding;
    int bytes_per_sample;
    sora_adsb_ctx *ctx;

    size_t block_samples;
    int eof;

    const struct sora_adsb_frame *frames;	/* of the current block: library memory */
    size_t n_frames, next_frame;

    long long t0;
    unsigned long long rate;
    unsigned long long last_position;
    int error;
    char msg[SORA_ADSB_MSG_BITS + 1];
};

static int
encoding_bytes(int encoding) {
    switch (encoding) {
    case SORA_ENC_UC8:
    case SORA_ENC_SC8:
	return 2;
    case SORA_ENC_SC16:
	return 4;
    default:
	return 0;
    }
}

/* sora_adsb_read_fn over a FILE: fread semantics */
static size_t
read_file(void *user, void *dst, size_t nbytes) {
    return fread(dst, 1, nbytes, (FILE *) user);
}

static struct plane_iq_state *
state_new(FILE *f, int encoding, int emit_all) {
    struct plane_iq_state *st;
    struct sora_adsb_config cfg;
    const char *s;
    int rc;

    if (encoding_bytes(encoding) == 0) {
	fprintf(stderr, "sora: unsupported encoding %d for ADS-B input\n", encoding);
	return NULL;
    }
    st = calloc(1, sizeof *st);
    if (st == NULL)
	return NULL;
    st->f = f;
    st->encoding = encoding;
    st->bytes_per_sample = encoding_bytes(encoding);
    st->block_samples = DEFAULT_BLOCK_SAMPLES;
    if ((s = getenv("SORA_ADSB_BLOCK_SAMPLES")) != NULL && atoll(s) >= 1024)
	st->block_samples = (size_t) atoll(s);

    if ((s = getenv("SORA_T0")) != NULL)
	st->t0 = atoll(s);
    else {
	struct timeval now;
	(void) gettimeofday(&now, NULL);
	st->t0 = now.tv_sec;
    }
    st->rate = 2000000ull;
    if ((s = getenv("SORA_RATE")) != NULL && strtoull(s, NULL, 10) > 0)
	st->rate = strtoull(s, NULL, 10);

    memset(&cfg, 0, sizeof cfg);
    cfg.max_chunk_samples = st->block_samples + SORA_ADSB_HALO;
    if ((s = getenv("SORA_ADSB_EMIT_ALL")) != NULL && *s != '\0')
	emit_all = atoi(s) != 0;
    cfg.emit_mode = emit_all? SORA_EMIT_CANDIDATES : SORA_EMIT_ACCEPTED;
    /* opt-in extension: off, as in the reference */
    if ((s = getenv("SORA_ADSB_CORRECT")) != NULL && atoi(s) != 0)
	cfg.options |= SORA_OPT_CORRECT_SINGLE_BIT;
    /* Candidate records: the bound outright -- p and p + 1 are never both
     * candidates (plane-iq.c:173: m[p+1] < m[p], then m[p+2] > m[p+1]), so a
     * block holds at most one per two positions: 8 bytes of device memory per
     * slot, 128 MB for the default block.  Squitter and frame lists start at
     * what busy airspace needs; the library grows a list that overflows and
     * runs the block again (sora_adsb_gpu.h, SORA_OPT_FIXED_LISTS), so no
     * input ends the stream early.  SORA_ADSB_MAX_CANDIDATES: tests of that. */
    cfg.max_candidates = cfg.max_chunk_samples / 2 + 1024;
    if ((s = getenv("SORA_ADSB_MAX_CANDIDATES")) != NULL && atoll(s) > 0)
	cfg.max_candidates = (uint64_t) atoll(s);
    cfg.max_squitters = cfg.max_chunk_samples / 64 + 65536;
    cfg.max_frames = emit_all? cfg.max_chunk_samples / 16 + 65536 : cfg.max_squitters;
    if (cfg.max_frames > cfg.max_candidates && emit_all)
	cfg.max_frames = cfg.max_candidates;
    s = getenv("SORA_ADSB_DEVICE");
    rc = sora_adsb_create(&st->ctx, s != NULL? atoi(s) : 0, &cfg);
    if (rc != SORA_OK) {
	fprintf(stderr, "sora: cannot use the GPU ADS-B path: %s (%s)\n",
		sora_adsb_strerror(rc), sora_adsb_last_error(st->ctx));
	sora_adsb_destroy(st->ctx);
	free(st);
	return NULL;
    }

    rc = sora_adsb_stream_begin(st->ctx, encoding, st->block_samples, read_file, f);
    if (rc != SORA_OK) {
	fprintf(stderr, "sora: cannot open the GPU ADS-B stream: %s (%s)\n",
		sora_adsb_strerror(rc), sora_adsb_last_error(st->ctx));
	plane_iq_state_delete(st);
	return NULL;
    }
    return st;
}

struct plane_iq_state *
plane_iq_state_new_encoded(FILE *f, int encoding) {
    return state_new(f, encoding, 1);		/* every candidate, plane-iq.c:187-192 */
}

struct plane_iq_state *
plane_iq_state_new_accepted(FILE *f, int encoding) {
    return state_new(f, encoding, 0);
}

struct plane_iq_state *
plane_iq_state_new(FILE *f) {
    return state_new(f, SORA_ENC_UC8, 1);	/* uc8: plane-iq.c:121-139 */
}

void
plane_iq_state_delete(struct plane_iq_state *st) {
    if (st == NULL)
	return;
    (void) sora_adsb_stream_end(st->ctx);
    sora_adsb_destroy(st->ctx);
    free(st);			/* the FILE stays open, like plane-iq.c:114-119 */
}

/* the frames of the next block; 0 when the stream is over */
static int
next_block(struct plane_iq_state *st) {
    int rc;

    if (st->eof)
	return 0;
    rc = sora_adsb_stream_next(st->ctx, &st->frames, &st->n_frames);
    st->next_frame = 0;
    if (rc == SORA_END) {
	st->eof = 1;
	st->n_frames = 0;
	return 0;
    }
    if (rc != SORA_OK) {
	fprintf(stderr, "sora: GPU ADS-B path failed: %s (%s)\n",
		sora_adsb_strerror(rc), sora_adsb_last_error(st->ctx));
	st->error = rc;
	st->eof = 1;
	st->n_frames = 0;
	return 0;
    }
    return 1;
}

const char *
plane_iq_get_next(struct plane_iq_state *st, struct timeval *tvp, double *amplitude) {
    const struct sora_adsb_frame *fr;
    int k;

    if (st == NULL)
	return NULL;
    while (st->next_frame >= st->n_frames) {
	st->n_frames = st->next_frame = 0;
	if (!next_block(st))
	    return NULL;
    }
    fr = &st->frames[st->next_frame++];

    for (k = 0; k < SORA_ADSB_MSG_BITS; k++)
	st->msg[k] = (char) ('0' + ((fr->bits[k >> 5] >> (31 - (k & 31))) & 1u));
    st->msg[SORA_ADSB_MSG_BITS] = '\0';
    st->last_position = fr->position;
    tvp->tv_sec = (time_t) (st->t0 + (long long) (fr->position / st->rate));
    tvp->tv_usec = (suseconds_t) ((fr->position % st->rate) * 1000000ull / st->rate);
    *amplitude = fr->amplitude_56 / 56;		/* plane-iq.c:189 */
    return st->msg;
}

unsigned long long
plane_iq_last_position(const struct plane_iq_state *st) {
    return st->last_position;
}

int
plane_iq_error(const struct plane_iq_state *st) {
    return st != NULL? st->error : SORA_ERR_INVALID;
}

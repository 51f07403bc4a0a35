/*
 * sora_synth.h -- deterministic synthetic IQ captures with injected Mode S
 * frames.  Bench and test tooling: the reference has
 * no generator (it has no tests or benchmarks at all).
 *
 * The capture is defined sample by sample by integer arithmetic only, so the
 * CPU fill (small test captures) and the gfx950 fill (multi-GB bench
 * captures, generated straight into HBM) produce identical bytes:
 *
 *   noise(i)  : a counter-based hash of (seed, i), binomial for the 8-bit
 *               encodings (mean 127.5 / -0.5, sigma ~3), Irwin-Hall for sc16
 *               (sigma ~400)
 *   frame f   : adds (amp_i, amp_q) to every "high" half-microsecond chip:
 *               preamble chips 0, 2, 7, 9, then for data bit k (MSB first)
 *               chip 16+2k when the bit is 1, chip 17+2k when it is 0
 *               (Mode S PPM at 2 MS/s; this is the pattern the reference
 *               predicate src/ads-b/plane-iq.c:173-175 and slicer :65 expect)
 *   sample    : clip(noise + sum over covering frames)
 */
#ifndef SORA_SYNTH_H_
#define SORA_SYNTH_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#if defined(__HIPCC__)
#define SORA_SYNTH_HD __host__ __device__ static inline
#else
#define SORA_SYNTH_HD static inline
#endif

/* encodings: same numbers as reference src/radio/radio-file.h:6-12 */
#define SORA_ENC_UC8	1
#define SORA_ENC_SC16	2
#define SORA_ENC_SC8	3

/* 32 bytes; sorted by position in a frame table */
struct sora_synth_frame {
    uint64_t position;	/* sample index of the first preamble chip */
    uint32_t bits[4];	/* message, MSB first: bit k is bits[k/32] >> (31-k%32) */
    int16_t amp_i;	/* added to I on a high chip, in encoding counts */
    int16_t amp_q;	/* added to Q on a high chip */
    uint16_t nbits;	/* 56 or 112 */
    uint16_t kind;	/* SORA_SYNTH_KIND_*: what the generator meant it to be */
};

#define SORA_SYNTH_KIND_DF17		1
#define SORA_SYNTH_KIND_DF11		2
#define SORA_SYNTH_KIND_REPLY_KNOWN	3	/* DF4/5/20/21 from a squittering ICAO */
#define SORA_SYNTH_KIND_REPLY_UNKNOWN	4	/* same, from an ICAO that never squitters */
#define SORA_SYNTH_KIND_FLIPPED		0x100	/* or-ed in: one bit was flipped */

struct sora_synth_params {
    uint64_t seed;
    uint64_t n_samples;
    int32_t encoding;		/* SORA_ENC_* */
    uint32_t n_icao;		/* size of the squittering address pool */
    uint32_t mean_gap;		/* mean samples between frame starts; gaps are
				 * uniform in [1, 2*mean_gap-1] */
    uint32_t df11_permille;	/* share of DF11 among frames */
    uint32_t reply_permille;	/* share of DF4/5/20/21 replies */
    uint32_t unknown_permille;	/* of the replies: from never-squittering ICAOs */
    uint32_t flip_permille;	/* frames that get one flipped bit */
    int32_t amp_min, amp_max;	/* pulse amplitude, encoding counts */
    int32_t amp_log_uniform;	/* 0: uniform, 1: log-uniform */
    uint32_t first_position;	/* first frame starts at or after this */
};

/* SURVEY.md section 8d table rows; config in 1..5 (5 uses encoding arg) */
int sora_synth_default_params(int config, struct sora_synth_params *p);

/*
 * Builds the frame table for a capture (host, deterministic in p->seed).
 * Returns the number of frames the capture holds; writes at most cap.
 */
size_t sora_synth_make_frames(const struct sora_synth_params *p,
	struct sora_synth_frame *frames, size_t cap);

/* Fill bytes for samples [first, first+count) into host memory. */
int sora_synth_fill_host(void *dst, uint64_t first, uint64_t count,
	int encoding, uint64_t seed,
	const struct sora_synth_frame *frames, size_t nframes);

/*
 * Same bytes into DEVICE memory (gfx950 kernel).  dst is a device pointer,
 * frames a HOST table (copied to the device for the call).  stream is a
 * hipStream_t or NULL.  Returns 0 or a SORA_ERR_* code (sora_adsb_gpu.h).
 */
int sora_synth_fill_device(void *dst, uint64_t first, uint64_t count,
	int encoding, uint64_t seed,
	const struct sora_synth_frame *frames, size_t nframes, void *stream);

/* Mode S CRC-24 over the first nbits-24 bits of an MSB-first word array */
uint32_t sora_modes_crc24(const uint32_t bits[4], int nbits_covered);

/* Frame builders used by the generator and by tests (bits out, MSB first) */
void sora_modes_build_df17(uint32_t bits[4], uint32_t icao, uint32_t ca,
	uint64_t me56);
void sora_modes_build_df11(uint32_t bits[4], uint32_t icao, uint32_t ca);
void sora_modes_build_short_reply(uint32_t bits[4], uint32_t df,
	uint32_t payload27, uint32_t icao);
void sora_modes_build_long_reply(uint32_t bits[4], uint32_t df,
	uint32_t payload27, uint64_t mb56, uint32_t icao);

/* ---- the per-sample definition, shared by host and device code ---- */

SORA_SYNTH_HD uint64_t
sora_synth_mix(uint64_t x) {
    x += 0x9e3779b97f4a7c15ull;
    x = (x ^ (x >> 30)) * 0xbf58476d1ce4e5b9ull;
    x = (x ^ (x >> 27)) * 0x94d049bb133111ebull;
    return x ^ (x >> 31);
}

SORA_SYNTH_HD int
sora_synth_popcount36(uint64_t h) {
    h &= 0xfffffffffull;
    h = h - ((h >> 1) & 0x5555555555555555ull);
    h = (h & 0x3333333333333333ull) + ((h >> 2) & 0x3333333333333333ull);
    h = (h + (h >> 4)) & 0x0f0f0f0f0f0f0f0full;
    return (int) ((h * 0x0101010101010101ull) >> 56);
}

/* noise around zero for component c (0 = I, 1 = Q) of sample i */
SORA_SYNTH_HD int
sora_synth_noise(uint64_t seed, uint64_t i, int c, int encoding) {
    uint64_t h = sora_synth_mix(seed ^ sora_synth_mix(2 * i + (uint64_t) c));

    if (encoding == SORA_ENC_SC16) {
	int s = (int) (h & 1023) + (int) ((h >> 10) & 1023) +
		(int) ((h >> 20) & 1023) + (int) ((h >> 30) & 1023);
	return ((s * 693) >> 10) - 1385;
    }
    /* binomial(36, 1/2) - 18 - (coin): mean -0.5, sigma ~3.04 */
    return sora_synth_popcount36(h) - 18 - (int) ((h >> 36) & 1);
}

/* 1 when chip d (0..239) of the frame carries a pulse */
SORA_SYNTH_HD int
sora_synth_chip_high(const struct sora_synth_frame *f, unsigned d) {
    if (d < 16)
	return d == 0 || d == 2 || d == 7 || d == 9;
    d -= 16;
    if (d >= 2u * f->nbits)
	return 0;
    {
	unsigned k = d >> 1;
	unsigned bit = (f->bits[k >> 5] >> (31 - (k & 31))) & 1u;
	return (int) (bit ^ (d & 1u));
    }
}

SORA_SYNTH_HD int
sora_synth_clip(int v, int encoding) {
    if (encoding == SORA_ENC_SC16)
	return v < -32768? -32768 : v > 32767? 32767 : v;
    if (encoding == SORA_ENC_SC8)
	return v < -128? -128 : v > 127? 127 : v;
    return v < 0? 0 : v > 255? 255 : v;		/* uc8, after the +128 */
}

/*
 * I and Q of sample i given the frames that may cover it:
 * frames[lo .. hi) must hold every frame with position in (i-240, i].
 */
SORA_SYNTH_HD void
sora_synth_sample(uint64_t seed, uint64_t i, int encoding,
	const struct sora_synth_frame *frames, size_t lo, size_t hi,
	int *out_i, int *out_q) {
    int vi = sora_synth_noise(seed, i, 0, encoding);
    int vq = sora_synth_noise(seed, i, 1, encoding);
    size_t f;

    for (f = lo; f < hi; f++) {
	uint64_t pos = frames[f].position;
	if (pos <= i && i - pos < 240 &&
		sora_synth_chip_high(&frames[f], (unsigned) (i - pos))) {
	    vi += frames[f].amp_i;
	    vq += frames[f].amp_q;
	}
    }
    if (encoding == SORA_ENC_UC8) {
	vi += 128;
	vq += 128;
    }
    *out_i = sora_synth_clip(vi, encoding);
    *out_q = sora_synth_clip(vq, encoding);
}

#ifdef __cplusplus
}
#endif

#endif /* SORA_SYNTH_H_ */

/*
 * synth.hip -- synthetic capture generator (include/sora_synth.h): Mode S
 * frame builders, the deterministic frame table, and the host / gfx950 fills
 * that both evaluate the per-sample definition of sora_synth.h.
 *
 * The reference ships no generator, fixture or test;
 * this is bench and test tooling that lives beside the product library so
 * that multi-GB captures are produced directly in HBM.
 */
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <string.h>

#include "../../include/sora_adsb_gpu.h"
#include "../../include/sora_synth.h"

/* ------------------------------------------------------ Mode S builders -- */

static inline void
put_bits(uint32_t bits[4], int off, int n, uint64_t v) {
    for (int i = 0; i < n; i++) {
	int k = off + i;
	uint32_t b = (uint32_t) ((v >> (n - 1 - i)) & 1);
	bits[k >> 5] = (bits[k >> 5] & ~(1u << (31 - (k & 31)))) |
		(b << (31 - (k & 31)));
    }
}

extern "C" uint32_t
sora_modes_crc24(const uint32_t bits[4], int nbits_covered) {
    /* generator 0x1FFF409, zero init, MSB first (reference plane.c:373-385) */
    uint32_t r = 0;
    for (int k = 0; k < nbits_covered; k++) {
	uint32_t b = (bits[k >> 5] >> (31 - (k & 31))) & 1u;
	r ^= b << 23;
	r <<= 1;
	if (r & 0x1000000u)
	    r ^= 0x1fff409u;
    }
    return r & 0xffffffu;
}

extern "C" void
sora_modes_build_df17(uint32_t bits[4], uint32_t icao, uint32_t ca, uint64_t me56) {
    memset(bits, 0, 16);
    put_bits(bits, 0, 5, 17);
    put_bits(bits, 5, 3, ca);
    put_bits(bits, 8, 24, icao);
    put_bits(bits, 32, 56, me56);
    put_bits(bits, 88, 24, sora_modes_crc24(bits, 88));
}

extern "C" void
sora_modes_build_df11(uint32_t bits[4], uint32_t icao, uint32_t ca) {
    memset(bits, 0, 16);
    put_bits(bits, 0, 5, 11);
    put_bits(bits, 5, 3, ca);
    put_bits(bits, 8, 24, icao);
    put_bits(bits, 32, 24, sora_modes_crc24(bits, 32));	/* IID 0 */
}

extern "C" void
sora_modes_build_short_reply(uint32_t bits[4], uint32_t df, uint32_t payload27,
	uint32_t icao) {
    memset(bits, 0, 16);
    put_bits(bits, 0, 5, df);
    put_bits(bits, 5, 27, payload27);
    put_bits(bits, 32, 24, sora_modes_crc24(bits, 32) ^ (icao & 0xffffffu));
}

extern "C" void
sora_modes_build_long_reply(uint32_t bits[4], uint32_t df, uint32_t payload27,
	uint64_t mb56, uint32_t icao) {
    memset(bits, 0, 16);
    put_bits(bits, 0, 5, df);
    put_bits(bits, 5, 27, payload27);
    put_bits(bits, 32, 56, mb56);
    put_bits(bits, 88, 24, sora_modes_crc24(bits, 88) ^ (icao & 0xffffffu));
}

/* ----------------------------------------------------------- frame table -- */

namespace {

struct rng {
    uint64_t s;
    uint64_t next() { s += 0x9e3779b97f4a7c15ull; return sora_synth_mix(s); }
    uint32_t below(uint32_t n) { return (uint32_t) (next() % n); }
};

uint32_t
pool_icao(uint64_t seed, uint32_t idx, int unknown) {
    /* squittering pool: top nibble 4; never-squittering pool: top nibble A */
    uint32_t v = (uint32_t) (sora_synth_mix(seed ^ (0x1ca0000ull + idx)) & 0xfffffu);
    return (unknown? 0xa00000u : 0x400000u) | v;
}

/* CPR-encode (airborne, 17 bits) */
void
cpr_encode(double lat, double lon, int odd, uint32_t *yz, uint32_t *xz) {
    const double dlat = 360.0 / (odd? 59 : 60);
    double m = fmod(lat, dlat);
    if (m < 0)
	m += dlat;
    uint32_t y = (uint32_t) floor(131072.0 * m / dlat + 0.5);
    double rlat = dlat * ((double) y / 131072.0 + floor(lat / dlat));
    /* NL(rlat) */
    int nl;
    double a = fabs(rlat);
    if (a < 1e-9)
	nl = 59;
    else if (a >= 87.0)
	nl = a > 87.0? 1 : 2;
    else {
	double c = cos(M_PI / 180.0 * a);
	nl = (int) floor(2 * M_PI / acos(1 - (1 - cos(M_PI / 30)) / (c * c)));
    }
    int n = nl - odd;
    if (n < 1)
	n = 1;
    const double dlon = 360.0 / n;
    double ml = fmod(lon, dlon);
    if (ml < 0)
	ml += dlon;
    uint32_t x = (uint32_t) floor(131072.0 * ml / dlon + 0.5);
    *yz = y & 0x1ffff;
    *xz = x & 0x1ffff;
}

uint64_t
me_identification(rng &r) {
    /* TC 4, category 3, 8 callsign characters */
    static const char set[] = "ABCDEFGHIJKLMNOPQRSTUVWXYZ0123456789 ";
    uint64_t me = ((uint64_t) 4 << 51) | ((uint64_t) 3 << 48);
    for (int i = 0; i < 8; i++) {
	char c = set[r.below(sizeof set - 1)];
	uint64_t v = c == ' '? 32 : (c >= 'A' && c <= 'Z')? (uint64_t) (c - 'A' + 1)
		: (uint64_t) (c - '0' + 48);
	me |= v << (42 - 6 * i);
    }
    return me;
}

uint64_t
me_position(rng &r, uint32_t icao_idx, int odd) {
    /* TC 11, altitude with Q bit, CPR of a per-aircraft fixed point */
    double lat = 30.0 + (double) (icao_idx % 97) * 0.31;
    double lon = -10.0 + (double) (icao_idx % 89) * 0.47;
    uint32_t yz, xz;
    cpr_encode(lat, lon, odd, &yz, &xz);
    uint32_t n = 400 + r.below(1200);		/* (n*25 - 1000) ft */
    uint32_t alt12 = ((n >> 4) << 5) | 0x10 | (n & 0xf);
    uint64_t me = (uint64_t) 11 << 51;
    me |= (uint64_t) (alt12 & 0xfff) << 36;
    me |= (uint64_t) (odd & 1) << 34;
    me |= (uint64_t) yz << 17;
    me |= (uint64_t) xz;
    return me;
}

uint64_t
me_velocity(rng &r) {
    /* TC 19 subtype 1: signed E-W / N-S ground speed */
    uint64_t me = ((uint64_t) 19 << 51) | ((uint64_t) 1 << 48);
    me |= (uint64_t) r.below(2) << 42;			/* E-W sign */
    me |= (uint64_t) (1 + r.below(600)) << 32;		/* E-W velocity */
    me |= (uint64_t) r.below(2) << 31;			/* N-S sign */
    me |= (uint64_t) (1 + r.below(600)) << 21;		/* N-S velocity */
    me |= (uint64_t) r.below(1 << 21);
    return me;
}

/* 64 unit vectors * 1024 for the carrier phase */
void
phase_vector(uint32_t idx, int amp, int16_t *ai, int16_t *aq) {
    double ph = 2 * M_PI * (double) (idx & 63) / 64.0;
    *ai = (int16_t) lrint(amp * cos(ph));
    *aq = (int16_t) lrint(amp * sin(ph));
}

} /* namespace */

extern "C" int
sora_synth_default_params(int config, struct sora_synth_params *p) {
    memset(p, 0, sizeof *p);
    p->seed = 0x50a4ad5bull + (uint64_t) config;
    p->encoding = SORA_ENC_UC8;
    p->first_position = 1000;
    switch (config) {
    case 1:	/* 60 s, DF17 ~200/s from 50 ICAOs, amplitude 40-120 */
	p->n_samples = 120000000ull;
	p->n_icao = 50;
	p->mean_gap = 10000;
	p->reply_permille = 20;
	p->unknown_permille = 500;
	p->amp_min = 40; p->amp_max = 120;
	return 0;
    case 2:	/* 1 h, DF17 ~400/s + DF11 ~200/s from 300 ICAOs, mixed SNR */
    case 3:
	p->n_samples = 7200000000ull;
	p->n_icao = 300;
	p->mean_gap = 3333;
	p->df11_permille = 330;
	p->reply_permille = 10;
	p->unknown_permille = 500;
	p->amp_min = 8; p->amp_max = 120;
	p->amp_log_uniform = 1;
	return 0;
    case 4:	/* dense stress, sc16, ~10k frames/s, 20 % single-bit errors */
	p->n_samples = 120000000ull;
	p->encoding = SORA_ENC_SC16;
	p->n_icao = 300;
	p->mean_gap = 200;
	p->df11_permille = 330;
	p->reply_permille = 10;
	p->unknown_permille = 500;
	p->flip_permille = 200;
	p->amp_min = 1200; p->amp_max = 16000;
	p->amp_log_uniform = 1;
	return 0;
    case 5:	/* one archive recording; the caller varies seed / encoding */
	p->n_samples = 120000000ull;
	p->n_icao = 50;
	p->mean_gap = 10000;
	p->reply_permille = 20;
	p->unknown_permille = 500;
	p->amp_min = 40; p->amp_max = 120;
	return 0;
    default:
	return SORA_ERR_INVALID;
    }
}

extern "C" size_t
sora_synth_make_frames(const struct sora_synth_params *p,
	struct sora_synth_frame *frames, size_t cap) {
    rng r = { p->seed ^ 0xf4a3e5ull };
    uint64_t pos = p->first_position;
    size_t n = 0;
    const uint32_t gap_span = p->mean_gap > 0? 2 * p->mean_gap - 1 : 1;

    if (p->mean_gap == 0 || p->n_icao == 0)
	return 0;
    for (;;) {
	pos += 1 + r.below(gap_span);
	if (pos + 240 > p->n_samples)
	    break;
	struct sora_synth_frame f;
	memset(&f, 0, sizeof f);
	f.position = pos;

	uint32_t sel = r.below(1000);
	uint32_t idx = r.below(p->n_icao);
	if (sel < p->reply_permille) {
	    int unknown = r.below(1000) < p->unknown_permille;
	    uint32_t icao = pool_icao(p->seed, idx, unknown);
	    uint32_t which = r.below(4);
	    uint32_t payload = (uint32_t) r.next() & 0x7ffffffu;
	    f.kind = unknown? SORA_SYNTH_KIND_REPLY_UNKNOWN : SORA_SYNTH_KIND_REPLY_KNOWN;
	    if (which == 0) {		/* DF4 altitude: AC is the low 13 bits */
		uint32_t ac = payload & 0x1fffu & ~0x40u;	/* M = 0 */
		if (r.below(4) != 0)
		    ac |= 0x10u;	/* Q = 1: 25 ft code; else Gillham */
		payload = (payload & ~0x1fffu) | ac;
		sora_modes_build_short_reply(f.bits, 4, payload, icao);
		f.nbits = 56;
	    } else if (which == 1) {	/* DF5 identity */
		sora_modes_build_short_reply(f.bits, 5, payload, icao);
		f.nbits = 56;
	    } else {			/* DF20 / DF21 Comm-B */
		sora_modes_build_long_reply(f.bits, which == 2? 20 : 21, payload,
			r.next() & 0xffffffffffffffull, icao);
		f.nbits = 112;
	    }
	} else if (sel < p->reply_permille + p->df11_permille) {
	    sora_modes_build_df11(f.bits, pool_icao(p->seed, idx, 0), 5);
	    f.nbits = 56;
	    f.kind = SORA_SYNTH_KIND_DF11;
	} else {
	    uint32_t t = r.below(8);
	    uint64_t me = t == 0? me_identification(r) : t <= 2? me_velocity(r)
		    : me_position(r, idx, (int) r.below(2));
	    sora_modes_build_df17(f.bits, pool_icao(p->seed, idx, 0), 5, me);
	    f.nbits = 112;
	    f.kind = SORA_SYNTH_KIND_DF17;
	}

	if (r.below(1000) < p->flip_permille) {
	    uint32_t k = r.below(f.nbits);
	    f.bits[k >> 5] ^= 1u << (31 - (k & 31));
	    f.kind |= SORA_SYNTH_KIND_FLIPPED;
	}

	int amp;
	if (p->amp_log_uniform) {
	    double u = (double) (r.next() >> 11) / 9007199254740992.0;
	    amp = (int) lrint(p->amp_min *
		    pow((double) p->amp_max / (double) p->amp_min, u));
	} else {
	    amp = p->amp_min + (int) r.below((uint32_t) (p->amp_max - p->amp_min + 1));
	}
	phase_vector(r.below(64), amp, &f.amp_i, &f.amp_q);

	if (n < cap)
	    frames[n] = f;
	n++;
    }
    return n;
}

/* ------------------------------------------------------------- host fill -- */

extern "C" int
sora_synth_fill_host(void *dst, uint64_t first, uint64_t count, int encoding,
	uint64_t seed, const struct sora_synth_frame *frames, size_t nframes) {
    size_t lo = 0;

    if (encoding != SORA_ENC_UC8 && encoding != SORA_ENC_SC8 &&
	    encoding != SORA_ENC_SC16)
	return SORA_ERR_INVALID;
    for (uint64_t k = 0; k < count; k++) {
	const uint64_t i = first + k;
	int vi, vq;

	while (lo < nframes && frames[lo].position + 240 <= i)
	    lo++;
	size_t hi = lo;
	while (hi < nframes && frames[hi].position <= i)
	    hi++;
	sora_synth_sample(seed, i, encoding, frames, lo, hi, &vi, &vq);
	if (encoding == SORA_ENC_SC16) {
	    int16_t *o = (int16_t *) dst + 2 * k;
	    o[0] = (int16_t) vi;
	    o[1] = (int16_t) vq;
	} else {
	    uint8_t *o = (uint8_t *) dst + 2 * k;
	    o[0] = (uint8_t) vi;
	    o[1] = (uint8_t) vq;
	}
    }
    return SORA_OK;
}

/* ----------------------------------------------------------- device fill -- */

namespace {

constexpr int kFillThreads = 256;
constexpr int kFillPerThread = 8;
constexpr int kFillBlock = kFillThreads * kFillPerThread;	/* samples per block pass */

__global__ void __launch_bounds__(kFillThreads)
synth_fill_kernel(uint8_t *dst, uint64_t first, uint64_t count, int encoding,
	uint64_t seed, const sora_synth_frame *frames, uint64_t nframes) {
    __shared__ uint64_t s_lo, s_hi;
    const uint64_t nblocks = (count + kFillBlock - 1) / kFillBlock;

    for (uint64_t blk = blockIdx.x; blk < nblocks; blk += gridDim.x) {
	const uint64_t k0 = blk * kFillBlock;
	const uint64_t i0 = first + k0;

	__syncthreads();
	if (threadIdx.x == 0) {
	    /* frames with position in (i0 - 240, i0 + kFillBlock) */
	    uint64_t a = 0, b = nframes;
	    while (a < b) {		/* first frame with position + 240 > i0 */
		uint64_t m = (a + b) / 2;
		if (frames[m].position + 240 <= i0)
		    a = m + 1;
		else
		    b = m;
	    }
	    s_lo = a;
	    b = nframes;
	    while (a < b) {		/* first frame with position >= i0 + block */
		uint64_t m = (a + b) / 2;
		if (frames[m].position < i0 + kFillBlock)
		    a = m + 1;
		else
		    b = m;
	    }
	    s_hi = a;
	}
	__syncthreads();
	const uint64_t lo = s_lo, hi = s_hi;

	/* 8 consecutive samples per thread: one 16-byte (or two) store */
	const uint64_t k = k0 + (uint64_t) threadIdx.x * kFillPerThread;
	if (k >= count)
	    continue;
	int vi[kFillPerThread], vq[kFillPerThread];
#pragma unroll
	for (int j = 0; j < kFillPerThread; j++)
	    sora_synth_sample(seed, first + k + j, encoding, frames, lo, hi,
		    &vi[j], &vq[j]);
	const bool full = k + kFillPerThread <= count;
	if (encoding == SORA_ENC_SC16) {
	    int16_t *o = reinterpret_cast<int16_t *>(dst) + 2 * k;
	    for (int j = 0; j < kFillPerThread && (full || k + j < count); j++) {
		o[2 * j] = (int16_t) vi[j];
		o[2 * j + 1] = (int16_t) vq[j];
	    }
	} else if (full && ((reinterpret_cast<uintptr_t>(dst) + 2 * k) & 15) == 0) {
	    uint32_t w[4];
#pragma unroll
	    for (int j = 0; j < 4; j++)
		w[j] = (uint32_t) (vi[2 * j] & 0xff) |
			((uint32_t) (vq[2 * j] & 0xff) << 8) |
			((uint32_t) (vi[2 * j + 1] & 0xff) << 16) |
			((uint32_t) (vq[2 * j + 1] & 0xff) << 24);
	    *reinterpret_cast<uint4 *>(dst + 2 * k) = make_uint4(w[0], w[1], w[2], w[3]);
	} else {
	    uint8_t *o = dst + 2 * k;
	    for (int j = 0; j < kFillPerThread && k + j < count; j++) {
		o[2 * j] = (uint8_t) vi[j];
		o[2 * j + 1] = (uint8_t) vq[j];
	    }
	}
    }
}

} /* namespace */

extern "C" int
sora_synth_fill_device(void *dst, uint64_t first, uint64_t count, int encoding,
	uint64_t seed, const struct sora_synth_frame *frames, size_t nframes,
	void *stream) {
    hipStream_t st = (hipStream_t) stream;
    sora_synth_frame *d_frames = nullptr;
    hipError_t e;

    if (encoding != SORA_ENC_UC8 && encoding != SORA_ENC_SC8 &&
	    encoding != SORA_ENC_SC16)
	return SORA_ERR_INVALID;
    if (count == 0)
	return SORA_OK;
    e = hipMalloc(&d_frames, (nframes + 1) * sizeof *d_frames);
    if (e != hipSuccess)
	return SORA_ERR_NOMEM;
    if (nframes > 0) {
	e = hipMemcpyAsync(d_frames, frames, nframes * sizeof *d_frames,
		hipMemcpyHostToDevice, st);
	if (e != hipSuccess) {
	    (void) hipFree(d_frames);
	    return SORA_ERR_HIP;
	}
    }
    uint64_t nblocks = (count + kFillBlock - 1) / kFillBlock;
    uint32_t grid = nblocks > 16384? 16384u : (uint32_t) nblocks;
    (void) hipGetLastError();
    hipLaunchKernelGGL(synth_fill_kernel, dim3(grid), dim3(kFillThreads), 0, st,
	    (uint8_t *) dst, first, count, encoding, seed, d_frames,
	    (uint64_t) nframes);
    e = hipGetLastError();
    if (e == hipSuccess)
	e = hipStreamSynchronize(st);
    (void) hipFree(d_frames);
    return e == hipSuccess? SORA_OK : SORA_ERR_HIP;
}

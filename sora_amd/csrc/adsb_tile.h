/*
 * adsb_tile.h -- the per-lane arithmetic of the gfx950 scan kernel, written as
 * host+device inline functions so that the same code runs inside
 * adsb_kernels.hip and inside the CPU lane emulator of the test-suite
 * (tests/emul/).  Nothing here touches memory spaces or wave intrinsics.
 *
 * Vocabulary (reference src/ads-b/plane-iq.c):
 *   M[i]      integer squared magnitude of sample i, I^2+Q^2 in counts.  For
 *             uc8/sc8 the reference's float m[i] is exactly M[i] * 2^-14
 *             (plane-iq.c:132-135), so integer compares equal float compares.
 *   M'[i]     what the 8-bit kernels hold in registers: M[i] - 32768 as a
 *             signed 16-bit half (M <= 32768, so M' is in [-32768, 0]): the
 *             order is the same, and differences saturate instead of wrapping
 *   pair j    samples 2j (low half of a dword) and 2j+1 (high half)
 *   NU[i]     "not up": M[i+1] <= M[i].  This IS the PPM bit the slicer
 *             assigns (plane-iq.c:65); bit k of a candidate at p is
 *             NU[p+16+2k], i.e. 112 consecutive elements of the NU array of
 *             p's parity.
 *   D[i]      "down": M[i+1] < M[i]
 *   C[i]      M[i] > max(M[i+4], M[i+5], M[i+6])
 *   predicate plane-iq.c:173-175:
 *             D[p] & !NU[p+1] & D[p+2] & C[p] & D[p+7] & !NU[p+8]
 *
 * Bit arrays are split by parity (E: even positions, O: odd) and packed
 * MSB first: element j of an array is bit 31-(j%32) of word j/32, so that a
 * message read out of them has bit 0 (the first DF bit) on top.  A lane
 * produces 16 elements of each array (one 16-bit half word).
 */
#ifndef SORA_ADSB_TILE_H_
#define SORA_ADSB_TILE_H_

#include <stdint.h>

#if defined(__HIPCC__)
#define SORA_HD __host__ __device__ static inline
#else
#define SORA_HD static inline
#endif

/* ---- tile geometry for the 8-bit encodings ---- */
#ifndef SORA_TILE_THREADS
#define SORA_TILE_THREADS	512
#endif
#define SORA_LANE_PAIRS		16		/* pairs (dwords of M) per lane */
#define SORA_LANE_POS		32		/* positions per lane */
#define SORA_OWN_LANES		(SORA_TILE_THREADS - 8)	/* lanes whose positions a tile owns */
#define SORA_TILE_POS		(SORA_OWN_LANES * SORA_LANE_POS)	/* 16128 */
#define SORA_TILE_SAMPLES	(SORA_TILE_THREADS * SORA_LANE_POS)	/* 16384 */
/* M in LDS: lane g's 16 dwords are the 64 bytes at dword 16*g, their four
 * 16-byte chunks permuted by an XOR swizzle: chunk i of lane g sits at chunk
 * i ^ ((g >> 2) & 3).  No padding (32 KB for a tile), and both the 16-byte
 * stores of the load phase (8 consecutive lanes of a store fill two whole
 * lanes' runs) and the 16-byte reads of the compare phase stay bank-conflict
 * free: ds_read_b128 serves lanes {0-3,12-15,20-27}, {4-11,16-19,28-31}, ...
 * together, and within such a group the lanes with equal g mod 4 (the ones
 * that share a 64-bank column) have four different (g >> 2) & 3. */
#define SORA_LANE_STRIDE	16
#define SORA_M_DWORDS		(SORA_TILE_THREADS * SORA_LANE_STRIDE + 16)
/* bit arrays: one 16-bit half word per lane, +1 word of lookahead; as 32-bit
 * words, MSB first: lane 2w is the top half of word w */
#define SORA_WORDS		(SORA_TILE_THREADS / 2 + 1)

/* ---- tile geometry for sc16 (32-bit magnitudes): 256 lanes of 32 positions,
 * lane g's 32 dwords of M start at dword 36*g (36 = 4*9, 9 odd: the 16-byte
 * stores and reads stay bank-conflict free) ---- */
#define SORA16_TILE_THREADS	256
#define SORA16_OWN_LANES	248
#define SORA16_TILE_POS		(SORA16_OWN_LANES * SORA_LANE_POS)	/* 7936 */
#define SORA16_TILE_SAMPLES	(SORA16_TILE_THREADS * SORA_LANE_POS)	/* 8192 */
#define SORA16_LANE_STRIDE	36
#define SORA16_M_DWORDS		(SORA16_TILE_THREADS * SORA16_LANE_STRIDE + 8)
#define SORA16_WORDS		(SORA16_TILE_THREADS / 2 + 1)
#define SORA16_LOOKAHEAD	8		/* dwords of the next lane a lane reads */

#define SORA_WINDOW		240

/* full record (squitter lists, accepted lists): kind[63:60] | chunk-local
 * position[59:24] | address or syndrome[23:0].  The scan's candidate records
 * are the 5-byte key + tail form of sora_scan_args (adsb_kernels.h). */
#define SORA_REC_KIND_REPLY	0ull	/* DF other than 11/17/18 */
#define SORA_REC_KIND_SQUITTER	1ull	/* DF 11/17/18, syndrome 0 */
#define SORA_REC_KIND_DIRTY	2ull	/* DF 11/17/18, syndrome != 0 */
#define SORA_REC_KIND_CORRECTED	3ull	/* DF 11/17/18, single-bit syndrome, option on */
#define SORA_REC_ACCEPTED	4ull	/* or-ed into kind by the filter */
#define SORA_REC_POS_BITS	36

SORA_HD uint64_t
sora_rec_make(uint64_t kind, uint64_t pos, uint32_t addr) {
    return (kind << 60) | (pos << 24) | (uint64_t) (addr & 0xffffffu);
}
SORA_HD uint64_t sora_rec_pos(uint64_t r) { return (r >> 24) & 0xfffffffffull; }
SORA_HD uint32_t sora_rec_addr(uint64_t r) { return (uint32_t) r & 0xffffffu; }
SORA_HD uint32_t sora_rec_kind(uint64_t r) { return (uint32_t) (r >> 60); }

/* dword index of chunk i (0..3) of lane g's run in the swizzled M layout */
SORA_HD uint32_t
sora_m_chunk(uint32_t g, uint32_t i) {
    return SORA_LANE_STRIDE * g + 4 * (i ^ ((g >> 2) & 3));
}

/* ---- a1: magnitude of the two samples held in one dword of uc8/sc8 IQ ---- */

SORA_HD uint32_t
sora_mag2_pack8(uint32_t iq, uint32_t flip) {
    /* flip = 0x80808080 turns offset-binary uc8 into two's complement
     * (b - 128, plane-iq.c:132-133); 0 for sc8 */
    uint32_t x = iq ^ flip;
#if defined(__HIP_DEVICE_COMPILE__)
    /* the builtin, not inline asm of the VOP3P form with an inline-0
     * accumulator: that variant returned wrong sums on gfx950 hardware */
    int lo = (int) (x & 0xffffu), hi = (int) (x >> 16);
    uint32_t m0 = (uint32_t) __builtin_amdgcn_sdot4(lo, lo, 0, false);
    uint32_t m1 = (uint32_t) __builtin_amdgcn_sdot4(hi, hi, 0, false);
#else
    int i0 = (int8_t) (x & 0xff), q0 = (int8_t) ((x >> 8) & 0xff);
    int i1 = (int8_t) ((x >> 16) & 0xff), q1 = (int8_t) (x >> 24);
    uint32_t m0 = (uint32_t) (i0 * i0 + q0 * q0);
    uint32_t m1 = (uint32_t) (i1 * i1 + q1 * q1);
#endif
    /* each <= 32768; biased to signed halves */
    return ((m0 - 0x8000u) & 0xffffu) | ((m1 - 0x8000u) << 16);
}

/*
 * Four dwords (eight samples) at once, the form the scan kernels use.  Per
 * dword x (after the flip): M0 - 32768 = dot4(x with its high half cleared,
 * the same) - 32768 and M1 - 32768 likewise from the high half -- the bias is
 * the dot product's accumulator operand -- and one v_perm_b32 puts the two low
 * halves side by side: two v_bitop3 ((v ^ flip) & mask), two dots and the
 * permute, five VALU instructions per dword.  The dot products are written as
 * the VOP3P v_dot4_i32_i8 in one asm block.  gfx90a+ does not interlock a dot
 * result against the next VALU instructions (LLVM inserts 3 wait states for
 * its own dots, DotWriteDifferentVALURead); inside this block every result is
 * read at least 3 instructions after the dot that wrote it.
 */
SORA_HD void
sora_mag2_pack8x4(const uint32_t v[4], uint32_t flip, uint32_t m[4]) {
#if defined(__HIP_DEVICE_COMPILE__)
    const uint32_t l0 = (v[0] ^ flip) & 0xffffu, l1 = (v[1] ^ flip) & 0xffffu,
	    l2 = (v[2] ^ flip) & 0xffffu, l3 = (v[3] ^ flip) & 0xffffu;
    const uint32_t h0 = (v[0] ^ flip) & 0xffff0000u, h1 = (v[1] ^ flip) & 0xffff0000u,
	    h2 = (v[2] ^ flip) & 0xffff0000u, h3 = (v[3] ^ flip) & 0xffff0000u;
    uint32_t t0, t1, t2, t3, g0, g1, g2, g3;
    asm("v_dot4_i32_i8 %0, %8, %8, %16\n\t"
	"v_dot4_i32_i8 %4, %12, %12, %16\n\t"
	"v_dot4_i32_i8 %1, %9, %9, %16\n\t"
	"v_dot4_i32_i8 %5, %13, %13, %16\n\t"
	"v_dot4_i32_i8 %2, %10, %10, %16\n\t"
	"v_dot4_i32_i8 %6, %14, %14, %16\n\t"
	"v_dot4_i32_i8 %3, %11, %11, %16\n\t"
	"v_dot4_i32_i8 %7, %15, %15, %16\n\t"
	"v_perm_b32 %0, %4, %0, %17\n\t"
	"v_perm_b32 %1, %5, %1, %17\n\t"
	"v_perm_b32 %2, %6, %2, %17\n\t"
	"v_perm_b32 %3, %7, %3, %17"
	: "=&v"(t0), "=&v"(t1), "=&v"(t2), "=&v"(t3),
	  "=&v"(g0), "=&v"(g1), "=&v"(g2), "=&v"(g3)
	: "v"(l0), "v"(l1), "v"(l2), "v"(l3), "v"(h0), "v"(h1), "v"(h2), "v"(h3),
	  "s"(0xffff8000u), "s"(0x05040100u));
    m[0] = t0; m[1] = t1; m[2] = t2; m[3] = t3;
#else
    for (int i = 0; i < 4; i++)
	m[i] = sora_mag2_pack8(v[i], flip);
#endif
}

/*
 * sc16 (I, Q little-endian int16 in one dword): the magnitude is DEFINED as
 * m = (float) (uint32) (I^2 + Q^2) * 2^-30, one correctly rounded conversion
 * of the exact integer (SURVEY.md section 8a, sc16 note; the reference has no
 * sc16 path into ADS-B).  Returned: the bit pattern of the unscaled float
 * (float) (I^2 + Q^2).  Non-negative floats order like their bit patterns and
 * the power-of-two scale changes no comparison, so the kernel compares these
 * words as integers; they are below 2^31, so a difference keeps its sign.
 */
SORA_HD uint32_t
sora_float_bits(float f) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __float_as_uint(f);
#else
    union { float f; uint32_t u; } c;
    c.f = f;
    return c.u;
#endif
}

SORA_HD uint32_t
sora_mag_sc16(uint32_t iq) {
    const int32_t i = (int16_t) (iq & 0xffffu), q = (int16_t) (iq >> 16);
    const uint32_t sum = (uint32_t) (i * i) + (uint32_t) (q * q);	/* <= 2^31 */

    return sora_float_bits((float) sum);
}

/* four samples at once: v_dot2_i32_i16 with an inline 0 accumulator, then the
 * conversions; the same wait-state distances as sora_mag2_pack8x4 */
SORA_HD void
sora_mag_sc16x4(const uint32_t v[4], uint32_t m[4]) {
#if defined(__HIP_DEVICE_COMPILE__)
    uint32_t t0, t1, t2, t3;
    asm("v_dot2_i32_i16 %0, %4, %4, 0\n\t"
	"v_dot2_i32_i16 %1, %5, %5, 0\n\t"
	"v_dot2_i32_i16 %2, %6, %6, 0\n\t"
	"v_dot2_i32_i16 %3, %7, %7, 0\n\t"
	"v_cvt_f32_u32_e32 %0, %0\n\t"
	"v_cvt_f32_u32_e32 %1, %1\n\t"
	"v_cvt_f32_u32_e32 %2, %2\n\t"
	"v_cvt_f32_u32_e32 %3, %3"
	: "=&v"(t0), "=&v"(t1), "=&v"(t2), "=&v"(t3)
	: "v"(v[0]), "v"(v[1]), "v"(v[2]), "v"(v[3]));
    m[0] = t0; m[1] = t1; m[2] = t2; m[3] = t3;
#else
    for (int i = 0; i < 4; i++)
	m[i] = sora_mag_sc16(v[i]);
#endif
}

/* ---- packed 16-bit arithmetic: one VALU instruction per pair on gfx950 ---- */

/* per half, signed: a - b, saturating at -32768 / 32767 (v_pk_sub_i16 ... clamp;
 * the saturation is on the exact difference, tools/probe/pk_i16_probe.cpp) */
SORA_HD uint32_t
sora_pk_sub_i16_sat(uint32_t a, uint32_t b) {
#if defined(__HIP_DEVICE_COMPILE__)
    uint32_t r;
    asm("v_pk_sub_i16 %0, %1, %2 clamp" : "=v"(r) : "v"(a), "v"(b));
    return r;
#else
    uint32_t r = 0;
    for (int h = 0; h < 2; h++) {
	int32_t d = (int32_t) (int16_t) (a >> (16 * h)) - (int32_t) (int16_t) (b >> (16 * h));
	d = d > 32767? 32767 : d < -32768? -32768 : d;
	r |= ((uint32_t) d & 0xffffu) << (16 * h);
    }
    return r;
#endif
}

/* per half, signed maximum (v_pk_max_i16) */
SORA_HD uint32_t
sora_pk_max_i16(uint32_t a, uint32_t b) {
#if defined(__HIP_DEVICE_COMPILE__)
    uint32_t r;
    asm("v_pk_max_i16 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
#else
    uint32_t r = 0;
    for (int h = 0; h < 2; h++) {
	const int16_t x = (int16_t) (a >> (16 * h)), y = (int16_t) (b >> (16 * h));
	r |= ((uint32_t) (uint16_t) (x > y? x : y)) << (16 * h);
    }
    return r;
#endif
}

/* acc | (x & mask): one v_and_or_b32 (kept as asm so that the compiler does not
 * re-associate a chain of them) */
SORA_HD uint32_t
sora_and_or(uint32_t x, uint32_t mask, uint32_t acc) {
#if defined(__HIP_DEVICE_COMPILE__)
    uint32_t r;
    asm("v_and_or_b32 %0, %1, %2, %3" : "=v"(r) : "v"(x), "s"(mask), "v"(acc));
    return r;
#else
    return acc | (x & mask);
#endif
}

/* [M[2j+1], M[2j+2]] from pairs j and j+1 */
SORA_HD uint32_t
sora_pair_shift(uint32_t p, uint32_t pn) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_alignbit(pn, p, 16);
#else
    return (p >> 16) | (pn << 16);
#endif
}

/*
 * The signs of four halves as four bytes: byte 0/1 = 0xff iff the low/high
 * half of a is negative, byte 2/3 the same for b.  One v_perm_b32: its byte
 * selectors 8..11 stand for "bit 7 of byte 1, 3, 5, 7 of {S0:S1}, replicated".
 */
SORA_HD uint32_t
sora_sign_bytes(uint32_t a, uint32_t b) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_perm(b, a, 0x0b0a0908u);
#else
    return ((a & 0x8000u)? 0xffu : 0u) | ((a & 0x80000000u)? 0xff00u : 0u) |
	    ((b & 0x8000u)? 0xff0000u : 0u) | ((b & 0x80000000u)? 0xff000000u : 0u);
#endif
}

/* bytes (b3, b2, b1, b0) -> halves (b1:b3, b0:b2): see sora_lane_compare */
SORA_HD uint32_t
sora_flag_bytes_to_halves(uint32_t v) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_perm(v, v, 0x01030002u);
#else
    return ((v >> 16) & 0xffu) | ((v & 0xffu) << 8) | ((v >> 24) << 16) |
	    (((v >> 8) & 0xffu) << 24);
#endif
}

/* ---- a3/a5 on one lane's run of 16 pairs ---- */

/*
 * P[0..19]: the lane's 16 dwords of packed M' followed by the next lane's
 * first 4.  Each result holds the EVEN positions in its low half and the ODD
 * positions in its high half; within a half, pair s of the lane is bit 15-s.
 *   up   M[i+1] >  M[i]		(NU = ~up is the slicer bit, plane-iq.c:65)
 *   down M[i+1] <  M[i]
 *   c    M[i] > max(M[i+4], M[i+5], M[i+6])
 * Each is the sign of a saturating difference (v_pk_sub_i16 clamp: the exact
 * difference of two M' can be +32768).  The signs are collected four at a time:
 * sora_sign_bytes turns the differences of pairs m and m + 8 into four bytes of
 * 0x00 / 0xff, v_and_or_b32 with 0x01010101 << (7 - m) keeps bit 7 - m of each,
 * so that after m = 0..7 byte 0 holds the even positions of pairs 0..7, MSB
 * first, byte 1 their odd positions, bytes 2 and 3 the same for pairs 8..15; one
 * more v_perm_b32 puts the bytes in half-word order.  Per lane run: 18 shifted
 * pairs, 32 maxima, 48 differences, 24 + 24 + 3 to collect them: 149 VALU
 * instructions, 9.3 per pair (11 until round 2, when every difference went
 * through a sign pattern, v_pk_mad_i16 clamp, and every pair through a
 * v_and_or_b32 of its own per bit set).
 */
SORA_HD void
sora_lane_compare(const uint32_t *P, uint32_t *up, uint32_t *down, uint32_t *c) {
    uint32_t S[SORA_LANE_PAIRS + 2];
    uint32_t W[SORA_LANE_PAIRS + 2];
    uint32_t acc_u = 0, acc_d = 0, acc_c = 0;

#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
    for (int s = 0; s < SORA_LANE_PAIRS + 2; s++)
	S[s] = sora_pair_shift(P[s], P[s + 1]);
    /* W[s] = [max M'[2s..2s+2], max M'[2s+1..2s+3]] */
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
    for (int s = 2; s < SORA_LANE_PAIRS + 2; s++)
	W[s] = sora_pk_max_i16(sora_pk_max_i16(P[s], S[s]), P[s + 1]);
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
    for (int m = 0; m < SORA_LANE_PAIRS / 2; m++) {
	const int n = m + SORA_LANE_PAIRS / 2;
	const uint32_t keep = 0x01010101u << (7 - m);

	acc_d = sora_and_or(sora_sign_bytes(sora_pk_sub_i16_sat(S[m], P[m]),
		sora_pk_sub_i16_sat(S[n], P[n])), keep, acc_d);
	acc_u = sora_and_or(sora_sign_bytes(sora_pk_sub_i16_sat(P[m], S[m]),
		sora_pk_sub_i16_sat(P[n], S[n])), keep, acc_u);
	acc_c = sora_and_or(sora_sign_bytes(sora_pk_sub_i16_sat(W[m + 2], P[m]),
		sora_pk_sub_i16_sat(W[n + 2], P[n])), keep, acc_c);
    }
    *up = sora_flag_bytes_to_halves(acc_u);
    *down = sora_flag_bytes_to_halves(acc_d);
    *c = sora_flag_bytes_to_halves(acc_c);
}

/* (acc << 1) | (the sign of d): one v_alignbit_b32 */
SORA_HD uint32_t
sora_shift_in_sign(uint32_t acc, uint32_t d) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_alignbit(acc, d, 31);
#else
    return (acc << 1) | (d >> 31);
#endif
}

SORA_HD uint32_t
sora_max3(uint32_t a, uint32_t b, uint32_t c) {
    uint32_t m = a > b? a : b;
    return m > c? m : c;		/* v_max3_u32 */
}

/*
 * The same three bit sets from 32-bit magnitudes (sc16): M[0..31] are the
 * lane's positions, M[32..38] the next lane's first seven.  Same result layout
 * as sora_lane_compare: even positions in the low half, odd in the high half,
 * pair s at bit 15-s.  Seven VALU instructions per position.
 */
SORA_HD void
sora_lane_compare32(const uint32_t *M, uint32_t *up, uint32_t *down, uint32_t *c) {
    uint32_t u[2] = {0, 0}, d[2] = {0, 0}, k[2] = {0, 0};

#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
    for (int i = 0; i < SORA_LANE_POS; i++) {
	const uint32_t a = M[i], b = M[i + 1];
	const uint32_t w = sora_max3(M[i + 4], M[i + 5], M[i + 6]);
	u[i & 1] = sora_shift_in_sign(u[i & 1], a - b);	/* b > a */
	d[i & 1] = sora_shift_in_sign(d[i & 1], b - a);	/* b < a */
	k[i & 1] = sora_shift_in_sign(k[i & 1], w - a);	/* a > w */
    }
    *up = u[0] | (u[1] << 16);
    *down = d[0] | (d[1] << 16);
    *c = k[0] | (k[1] << 16);
}

/*
 * Candidate masks of one lane.  Arguments are 32-bit windows of the MSB-first
 * bit arrays: the lane's own 16 elements in the top half, the next lane's 16
 * below them; results are valid in the top half.
 * Even p = 2j:   D[2j] & U[2j+1] & D[2j+2] & D[2j+7] & U[2j+8] & C[2j]
 *              = DE[j] & !NUO[j] & DE[j+1] & DO[j+3] & !NUE[j+4] & CE[j]
 * Odd p = 2j+1:  D[2j+1] & U[2j+2] & D[2j+3] & D[2j+8] & U[2j+9] & C[2j+1]
 *              = DO[j] & !NUE[j+1] & DO[j+1] & DE[j+4] & !NUO[j+4] & CO[j]
 */
SORA_HD void
sora_lane_predicate(uint32_t nue, uint32_t nuo, uint32_t de, uint32_t dod,
	uint32_t ce, uint32_t co, uint32_t *pred_e, uint32_t *pred_o) {
    *pred_e = de & ~nuo & (de << 1) & (dod << 3) & ~(nue << 4) & ce & 0xffff0000u;
    *pred_o = dod & ~(nue << 1) & (dod << 1) & (de << 4) & ~(nuo << 4) & co &
	    0xffff0000u;
}

/* mask with the first `count` elements set (MSB first), count clamped to [0,16] */
SORA_HD uint32_t
sora_first_elements(int64_t count) {
    if (count <= 0)
	return 0;
    if (count >= 16)
	return 0xffff0000u;
    return 0xffffffffu << (32 - (int) count);
}

/*
 * Owned-position masks of lane g: tile-local positions 32g+2b (even) and
 * 32g+2b+1 (odd) are kept iff below `limit` (the number of positions of this
 * tile that are to be scanned) and the lane is not a halo lane.
 */
SORA_HD void
sora_lane_valid_n(int g, int64_t limit, int own_lanes, uint32_t *valid_e,
	uint32_t *valid_o) {
    int64_t left = limit - (int64_t) g * SORA_LANE_POS;

    if (g >= own_lanes)
	left = 0;
    *valid_e = sora_first_elements((left + 1) >> 1);
    *valid_o = sora_first_elements(left >> 1);
}

SORA_HD void
sora_lane_valid(int g, int64_t limit, uint32_t *valid_e, uint32_t *valid_o) {
    sora_lane_valid_n(g, limit, SORA_OWN_LANES, valid_e, valid_o);
}

/* ---- a7..a9: one candidate out of the NU bit arrays ---- */

/* CRC-24, generator 0x1FFF409 (plane.c:373-385), MSB first: byte * x^24 mod G */
SORA_HD constexpr uint32_t
sora_crc24_table_entry(uint32_t byte) {
    uint32_t r = byte << 16;
    for (int i = 0; i < 8; i++) {
	r <<= 1;
	if (r & 0x1000000u)
	    r ^= 0x1fff409u;
    }
    return r & 0xffffffu;
}

/*
 * Sliced tables: the CRC over the first 88 bits (11 bytes b0..b10) of a
 * message is the XOR over k of T[k][b_k], T[k][v] = v * x^(8*(10-k)+24) mod G
 * (the CRC is linear and starts from 0).  The CRC over the first 32 bits is
 * the XOR of T[7+k][b_k], k = 0..3.  Eleven independent lookups instead of a
 * chain of eleven dependent ones.
 */
#define SORA_CRC_SLICES		11
#define SORA_CRC_TABLE_WORDS	(SORA_CRC_SLICES * 256)

SORA_HD constexpr uint32_t
sora_crc24_slice_entry(int k, uint32_t byte) {
    uint32_t r = sora_crc24_table_entry(byte);
    for (int i = 0; i < 8 * (10 - k); i++) {
	r <<= 1;
	if (r & 0x1000000u)
	    r ^= 0x1fff409u;
    }
    return r & 0xffffffu;
}

/*
 * Single-bit correction (option).  Flipping bit b of
 * an n-bit message adds x^(n-1-b) mod G to the syndrome, so the correctable
 * syndromes are POW[k] = x^k mod G, k = 0..111 (k <= n - 6: the DF bits are
 * left alone).  Lookup: a 512-entry table addressed by a multiplicative hash
 * that is perfect on those 112 values (multiplier found by search, checked at
 * compile time); an entry holds POW[k] | k << 24, empty entries hold 0xff << 24.
 */
#define SORA_FIX_TABLE_WORDS	512
#define SORA_FIX_MULT		0xc60d7f03u

SORA_HD constexpr uint32_t
sora_crc_pow(int k) {
    uint32_t v = 1;
    for (int i = 0; i < k; i++) {
	v <<= 1;
	if (v & 0x1000000u)
	    v ^= 0x1fff409u;
    }
    return v;
}

SORA_HD constexpr uint32_t
sora_fix_hash(uint32_t syndrome) {
    return (syndrome * SORA_FIX_MULT) >> 23;
}

SORA_HD constexpr bool
sora_fix_hash_is_perfect() {
    bool used[SORA_FIX_TABLE_WORDS] = {};
    for (int k = 0; k < 112; k++) {
	const uint32_t h = sora_fix_hash(sora_crc_pow(k));
	if (used[h])
	    return false;
	used[h] = true;
    }
    return true;
}
static_assert(sora_fix_hash_is_perfect(), "SORA_FIX_MULT collides");

/* bit to flip in an nbits-long message with this syndrome, or -1.
 * table: SORA_FIX_TABLE_WORDS words as described above */
SORA_HD int
sora_fix_bit(uint32_t syndrome, int nbits, const uint32_t *table) {
    const uint32_t e = table[sora_fix_hash(syndrome)];
    const int k = (int) (e >> 24);

    if ((e & 0xffffffu) != syndrome || k > nbits - 6)
	return -1;
    return nbits - 1 - k;
}

struct sora_candidate {
    uint32_t bits[4];	/* MSB first; bits[3] low half cleared */
    uint32_t df;
    uint32_t is_long;
    uint32_t squitter;	/* DF 11 / 17 / 18 */
    uint32_t syndrome;	/* crc ^ ap */
    uint32_t address;	/* bits[8,32) for squitters, syndrome otherwise */
};

/* (hi:lo) >> s, low word; s in 0..31: one v_alignbit_b32 */
SORA_HD uint32_t
sora_funnel(uint32_t hi, uint32_t lo, uint32_t s) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_alignbit(hi, lo, s);
#else
    return s == 0? lo : (hi << (32 - s)) | (lo >> s);
#endif
}

/*
 * 112 consecutive elements starting at element e >= 1 of an MSB-first bit
 * array.  With w = ceil(e / 32) and s = -e mod 32, window k is the funnel
 * (A[w-1+k] : A[w+k]) >> s for every e, a word-aligned start (s = 0) included:
 * five words, four v_alignbit_b32, no special case.
 */
SORA_HD void
sora_extract_bits(const uint32_t *words, uint32_t e, uint32_t bits[4]) {
    const uint32_t w = (e + 31) >> 5, s = (0u - e) & 31;
    const uint32_t a0 = words[w - 1], a1 = words[w], a2 = words[w + 1],
	    a3 = words[w + 2], a4 = words[w + 3];

    bits[0] = sora_funnel(a0, a1, s);
    bits[1] = sora_funnel(a1, a2, s);
    bits[2] = sora_funnel(a2, a3, s);
    bits[3] = sora_funnel(a3, a4, s) & 0xffff0000u;
}

/* plane.c:461-484 up to, but excluding, the plane-set lookup.
 * tables: SORA_CRC_TABLE_WORDS sliced CRC words (sora_crc24_slice_entry) */
SORA_HD void
sora_classify(const uint32_t bits[4], const uint32_t *tables,
	struct sora_candidate *c) {
    const uint32_t df = bits[0] >> 27;
    const uint32_t is_long = df >= 16 && df <= 24;	/* plane.c:462 */
    const uint32_t lmask = is_long? 0xffffffffu : 0u;
    /* bytes 0..3 use slices 0..3 when long, 7..10 when short */
    const uint32_t *t03 = tables + (is_long? 0 : 7 * 256);
    uint32_t crc, ap;

    /* plane.c:472: CRC over the first 32 bits, or 88 when long */
    crc = t03[bits[0] >> 24] ^ t03[256 + ((bits[0] >> 16) & 0xff)] ^
	    t03[512 + ((bits[0] >> 8) & 0xff)] ^ t03[768 + (bits[0] & 0xff)];
    crc ^= (tables[4 * 256 + (bits[1] >> 24)] ^
	    tables[5 * 256 + ((bits[1] >> 16) & 0xff)] ^
	    tables[6 * 256 + ((bits[1] >> 8) & 0xff)] ^
	    tables[7 * 256 + (bits[1] & 0xff)] ^
	    tables[8 * 256 + (bits[2] >> 24)] ^
	    tables[9 * 256 + ((bits[2] >> 16) & 0xff)] ^
	    tables[10 * 256 + ((bits[2] >> 8) & 0xff)]) & lmask;
    ap = is_long? (((bits[2] & 0xffu) << 16) | (bits[3] >> 16))	/* bits[88,112) */
	    : (bits[1] >> 8);					/* bits[32,56) */

    c->bits[0] = bits[0]; c->bits[1] = bits[1];
    c->bits[2] = bits[2]; c->bits[3] = bits[3];
    c->df = df;
    c->is_long = is_long;
    c->squitter = df == 11 || df == 17 || df == 18;	/* plane.c:473 */
    c->syndrome = crc ^ ap;
    c->address = c->squitter? (bits[0] & 0xffffffu) : c->syndrome;
}

#endif /* SORA_ADSB_TILE_H_ */

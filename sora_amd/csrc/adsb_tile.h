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
 * message read out of them has bit 0 (the first DF bit) on top.
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
#define SORA_TILE_THREADS	256
#define SORA_LANE_PAIRS		32		/* pairs (dwords of M) per lane */
#define SORA_LANE_POS		64		/* positions per lane */
#define SORA_OWN_LANES		252		/* lanes whose positions a tile owns */
#define SORA_TILE_POS		(SORA_OWN_LANES * SORA_LANE_POS)	/* 16128 */
#define SORA_TILE_SAMPLES	(SORA_TILE_THREADS * SORA_LANE_POS)	/* 16384 */
/* M in LDS: lane g's 32 dwords start at dword 36*g (4 dwords of padding keep
 * both the 16-byte stores of the load phase and the 16-byte reads of the
 * compare phase bank-conflict free) */
#define SORA_LANE_STRIDE	36
#define SORA_M_DWORDS		(SORA_TILE_THREADS * SORA_LANE_STRIDE + 4)
#define SORA_WORDS		(SORA_TILE_THREADS + 1)	/* bit-array words + 1 */

#define SORA_WINDOW		240

/* record: kind[63:60] | chunk-local position[59:24] | address or syndrome[23:0] */
#define SORA_REC_KIND_REPLY	0ull	/* DF other than 11/17/18 */
#define SORA_REC_KIND_SQUITTER	1ull	/* DF 11/17/18, syndrome 0 */
#define SORA_REC_KIND_DIRTY	2ull	/* DF 11/17/18, syndrome != 0 */
#define SORA_REC_ACCEPTED	4ull	/* or-ed into kind by the filter */
#define SORA_REC_POS_BITS	36

SORA_HD uint64_t
sora_rec_make(uint64_t kind, uint64_t pos, uint32_t addr) {
    return (kind << 60) | (pos << 24) | (uint64_t) (addr & 0xffffffu);
}
SORA_HD uint64_t sora_rec_pos(uint64_t r) { return (r >> 24) & 0xfffffffffull; }
SORA_HD uint32_t sora_rec_addr(uint64_t r) { return (uint32_t) r & 0xffffffu; }
SORA_HD uint32_t sora_rec_kind(uint64_t r) { return (uint32_t) (r >> 60); }

/* ---- a1: magnitude of the two samples held in one dword of uc8/sc8 IQ ---- */

SORA_HD uint32_t
sora_mag2_pack8(uint32_t iq, uint32_t flip) {
    /* flip = 0x80808080 turns offset-binary uc8 into two's complement
     * (b - 128, plane-iq.c:132-133); 0 for sc8 */
    uint32_t x = iq ^ flip;
#if defined(__HIP_DEVICE_COMPILE__)
    int lo = (int) (x & 0xffffu), hi = (int) (x >> 16);
    uint32_t m0 = (uint32_t) __builtin_amdgcn_sdot4(lo, lo, 0, false);
    uint32_t m1 = (uint32_t) __builtin_amdgcn_sdot4(hi, hi, 0, false);
#else
    int i0 = (int8_t) (x & 0xff), q0 = (int8_t) ((x >> 8) & 0xff);
    int i1 = (int8_t) ((x >> 16) & 0xff), q1 = (int8_t) (x >> 24);
    uint32_t m0 = (uint32_t) (i0 * i0 + q0 * q0);
    uint32_t m1 = (uint32_t) (i1 * i1 + q1 * q1);
#endif
    return m0 | (m1 << 16);		/* each <= 32768 */
}

/* ---- a3/a5 on one lane's run of 32 pairs ---- */


/*
 * P[0..35]: the lane's 32 dwords of packed M followed by the next lane's
 * first 4 (only 3 are used).  Element b of each result word (bit 31-b) is
 * pair b of the lane.
 */
SORA_HD void
sora_lane_compare(const uint32_t *P, uint32_t *nue, uint32_t *nuo,
	uint32_t *de, uint32_t *dod, uint32_t *ce, uint32_t *co) {
    uint32_t w_nue = 0, w_nuo = 0, w_de = 0, w_do = 0, w_ce = 0, w_co = 0;

#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
    for (int s = 0; s < SORA_LANE_PAIRS; s++) {
	uint32_t a_lo = P[s] & 0xffffu, a_hi = P[s] >> 16;
	uint32_t b_lo = P[s + 1] & 0xffffu;
	uint32_t c_lo = P[s + 2] & 0xffffu, c_hi = P[s + 2] >> 16;
	uint32_t d_lo = P[s + 3] & 0xffffu, d_hi = P[s + 3] >> 16;
	uint32_t mid = c_hi > d_lo? c_hi : d_lo;
	uint32_t mx_e = c_lo > mid? c_lo : mid;	/* max M[2j+4..2j+6] */
	uint32_t mx_o = d_hi > mid? d_hi : mid;	/* max M[2j+5..2j+7] */

	w_nue = (w_nue << 1) | (uint32_t) (a_hi <= a_lo);
	w_de  = (w_de  << 1) | (uint32_t) (a_hi <  a_lo);
	w_nuo = (w_nuo << 1) | (uint32_t) (b_lo <= a_hi);
	w_do  = (w_do  << 1) | (uint32_t) (b_lo <  a_hi);
	w_ce  = (w_ce  << 1) | (uint32_t) (a_lo > mx_e);
	w_co  = (w_co  << 1) | (uint32_t) (a_hi > mx_o);
    }
    *nue = w_nue; *nuo = w_nuo; *de = w_de; *dod = w_do; *ce = w_ce; *co = w_co;
}

/* element j+k of an MSB-first bit array, aligned onto element j;
 * x: this word, xn: the next word; 1 <= k <= 31 */
SORA_HD uint32_t
sora_ahead(uint32_t x, uint32_t xn, int k) {
    return (x << k) | (xn >> (32 - k));
}

/*
 * Candidate masks of one lane from its own words and the next lane's.
 * Even p = 2j:   D[2j] & U[2j+1] & D[2j+2] & D[2j+7] & U[2j+8] & C[2j]
 *              = DE[j] & !NUO[j] & DE[j+1] & DO[j+3] & !NUE[j+4] & CE[j]
 * Odd p = 2j+1:  D[2j+1] & U[2j+2] & D[2j+3] & D[2j+8] & U[2j+9] & C[2j+1]
 *              = DO[j] & !NUE[j+1] & DO[j+1] & DE[j+4] & !NUO[j+4] & CO[j]
 */
SORA_HD void
sora_lane_predicate(uint32_t nue, uint32_t nuo, uint32_t de, uint32_t dod,
	uint32_t ce, uint32_t co,
	uint32_t nue_n, uint32_t nuo_n, uint32_t de_n, uint32_t dod_n,
	uint32_t *pred_e, uint32_t *pred_o) {
    *pred_e = de & ~nuo & sora_ahead(de, de_n, 1) & sora_ahead(dod, dod_n, 3) &
	    ~sora_ahead(nue, nue_n, 4) & ce;
    *pred_o = dod & ~sora_ahead(nue, nue_n, 1) & sora_ahead(dod, dod_n, 1) &
	    sora_ahead(de, de_n, 4) & ~sora_ahead(nuo, nuo_n, 4) & co;
}

/* mask with the first `count` elements set (MSB first), count clamped to [0,32] */
SORA_HD uint32_t
sora_first_elements(int64_t count) {
    if (count <= 0)
	return 0;
    if (count >= 32)
	return 0xffffffffu;
    return 0xffffffffu << (32 - (int) count);
}

/*
 * Owned-position masks of lane g: tile-local positions 64g+2b (even) and
 * 64g+2b+1 (odd) are kept iff below `limit` (the number of positions of this
 * tile that are to be scanned) and the lane is not a halo lane.
 */
SORA_HD void
sora_lane_valid(int g, int64_t limit, uint32_t *valid_e, uint32_t *valid_o) {
    int64_t left = limit - (int64_t) g * SORA_LANE_POS;

    if (g >= SORA_OWN_LANES)
	left = 0;
    *valid_e = sora_first_elements((left + 1) >> 1);
    *valid_o = sora_first_elements(left >> 1);
}

/* ---- a7..a9: one candidate out of the NU bit arrays ---- */

/* CRC-24 byte table, generator 0x1FFF409 (plane.c:373-385), MSB first */
SORA_HD uint32_t
sora_crc24_table_entry(uint32_t byte) {
    uint32_t r = byte << 16;
    for (int i = 0; i < 8; i++) {
	r <<= 1;
	if (r & 0x1000000u)
	    r ^= 0x1fff409u;
    }
    return r & 0xffffffu;
}

struct sora_candidate {
    uint32_t bits[4];	/* MSB first; bits[3] low half cleared */
    uint32_t df;
    uint32_t is_long;
    uint32_t squitter;	/* DF 11 / 17 / 18 */
    uint32_t syndrome;	/* crc ^ ap */
    uint32_t address;	/* bits[8,32) for squitters, syndrome otherwise */
};

/* 112 consecutive elements starting at element e of an MSB-first bit array */
SORA_HD void
sora_extract_bits(const uint32_t *words, uint32_t e, uint32_t bits[4]) {
    uint32_t w0 = e >> 5, sh = e & 31;
    uint32_t a0 = words[w0], a1 = words[w0 + 1], a2 = words[w0 + 2],
	    a3 = words[w0 + 3], a4 = words[w0 + 4];

    if (sh == 0) {
	bits[0] = a0; bits[1] = a1; bits[2] = a2; bits[3] = a3;
    } else {
	bits[0] = (a0 << sh) | (a1 >> (32 - sh));
	bits[1] = (a1 << sh) | (a2 >> (32 - sh));
	bits[2] = (a2 << sh) | (a3 >> (32 - sh));
	bits[3] = (a3 << sh) | (a4 >> (32 - sh));
    }
    bits[3] &= 0xffff0000u;
}

/* plane.c:461-484 up to, but excluding, the plane-set lookup */
SORA_HD void
sora_classify(const uint32_t bits[4], const uint32_t *crc_table,
	struct sora_candidate *c) {
    uint32_t df = bits[0] >> 27;
    uint32_t is_long = df >= 16 && df <= 24;		/* plane.c:462 */
    uint32_t crc = 0, ap;

    /* plane.c:472: CRC over the first 32 bits, or 88 when long */
    for (int i = 3; i >= 0; i--)
	crc = ((crc << 8) ^ crc_table[((crc >> 16) ^ (bits[0] >> (8 * i))) & 0xff])
		& 0xffffffu;
    if (is_long) {
	for (int i = 3; i >= 0; i--)
	    crc = ((crc << 8) ^
		    crc_table[((crc >> 16) ^ (bits[1] >> (8 * i))) & 0xff]) & 0xffffffu;
	for (int i = 3; i >= 1; i--)
	    crc = ((crc << 8) ^
		    crc_table[((crc >> 16) ^ (bits[2] >> (8 * i))) & 0xff]) & 0xffffffu;
	ap = ((bits[2] & 0xffu) << 16) | (bits[3] >> 16);	/* bits[88,112) */
    } else {
	ap = bits[1] >> 8;					/* bits[32,56) */
    }

    c->bits[0] = bits[0]; c->bits[1] = bits[1];
    c->bits[2] = bits[2]; c->bits[3] = bits[3];
    c->df = df;
    c->is_long = is_long;
    c->squitter = df == 11 || df == 17 || df == 18;	/* plane.c:473 */
    c->syndrome = crc ^ ap;
    c->address = c->squitter? (bits[0] & 0xffffffu) : c->syndrome;
}

#endif /* SORA_ADSB_TILE_H_ */

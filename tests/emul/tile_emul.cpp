/*
 * TEST INFRASTRUCTURE -- CPU lane emulator of adsb_scan_kernel<false> and <true>.
 *
 * Runs the per-lane functions of sora_amd/csrc/adsb_tile.h (the very code the
 * gfx950 kernel compiles) for every lane of every tile, sequentially, with
 * plain arrays in place of LDS and phase boundaries in place of barriers.
 * It lets the CPU-only test tier check the kernel's index arithmetic, bit
 * packing, predicate algebra and CRC against the oracle without a GPU.  It is
 * not a product path: nothing outside tests/ builds or loads it.
 */
#include <stdint.h>
#include <string.h>
#include <vector>

#include "../../sora_amd/csrc/adsb_tile.h"

extern "C" size_t
emul_scan8(const uint8_t *iq, uint64_t n_samples, uint64_t n_scan, int encoding,
	int emit_all, uint64_t *records, size_t cap) {
    const uint32_t flip = encoding == 1? 0x80808080u : 0u;
    const uint64_t n_tiles = (n_scan + SORA_TILE_POS - 1) / SORA_TILE_POS;
    std::vector<uint32_t> s_m(SORA_M_DWORDS);
    /* 16-bit halves, one per lane, read back as 32-bit words (little endian) */
    std::vector<uint16_t> s_nue(2 * SORA_WORDS), s_nuo(2 * SORA_WORDS),
	    s_de(2 * SORA_WORDS), s_do(2 * SORA_WORDS);
    std::vector<uint32_t> l_c(SORA_TILE_THREADS);
    std::vector<uint32_t> crc_table(SORA_CRC_TABLE_WORDS);
    size_t n_out = 0;
    const int loads = SORA_TILE_SAMPLES * 2 / 16 / SORA_TILE_THREADS;

    for (int k = 0; k < SORA_CRC_SLICES; k++)
	for (uint32_t i = 0; i < 256; i++)
	    crc_table[k * 256 + i] = sora_crc24_slice_entry(k, i);

    for (uint64_t tile = 0; tile < n_tiles; tile++) {
	const uint64_t tile_start = tile * SORA_TILE_POS;

	/* phase 1 */
	for (int r = 0; r < loads; r++)
	    for (uint32_t tid = 0; tid < SORA_TILE_THREADS; tid++) {
		uint32_t pair = (r * SORA_TILE_THREADS + tid) * 4;
		for (int k = 0; k < 4; k++) {
		    uint64_t byte = (tile_start + 2ull * (pair + k)) * 2;
		    uint32_t v = 0;
		    /* buffer loads return 0 for out-of-range dwords */
		    uint64_t limit = (n_samples * 2 + 3) & ~3ull;
		    if (byte + 4 <= limit) {
			uint8_t b[4] = {0, 0, 0, 0};
			for (int c = 0; c < 4; c++)
			    if (byte + c < n_samples * 2)
				b[c] = iq[byte + c];
			v = b[0] | (b[1] << 8) | (b[2] << 16) | ((uint32_t) b[3] << 24);
		    }
		    /* pair = 16 * lane + 4 * chunk: the kernel's swizzled store */
		    s_m[sora_m_chunk(pair >> 4, (pair >> 2) & 3) + k] =
			    sora_mag2_pack8(v, flip);
		}
	    }
	for (int k = 0; k < 16; k++)
	    s_m[SORA_TILE_THREADS * SORA_LANE_STRIDE + k] = 0;

	/* phase 2 */
	for (uint32_t tid = 0; tid < SORA_TILE_THREADS; tid++) {
	    uint32_t P[SORA_LANE_PAIRS + 4], up, down, cc;
	    for (int i = 0; i < SORA_LANE_PAIRS; i++)
		P[i] = s_m[sora_m_chunk(tid, i >> 2) + (i & 3)];
	    for (int i = 0; i < 4; i++)
		P[SORA_LANE_PAIRS + i] = s_m[sora_m_chunk(tid + 1, 0) + i];
	    sora_lane_compare(P, &up, &down, &cc);
	    s_nue[tid ^ 1] = (uint16_t) (~up & 0xffffu);
	    s_nuo[tid ^ 1] = (uint16_t) ((~up) >> 16);
	    s_de[tid ^ 1] = (uint16_t) (down & 0xffffu);
	    s_do[tid ^ 1] = (uint16_t) (down >> 16);
	    l_c[tid] = cc;
	}
	for (int k = 0; k < 2; k++) {
	    s_nue[SORA_TILE_THREADS + k] = s_nuo[SORA_TILE_THREADS + k] = 0;
	    s_de[SORA_TILE_THREADS + k] = s_do[SORA_TILE_THREADS + k] = 0;
	}

	/* phase 3 */
	std::vector<uint16_t> queue;
	uint64_t left = n_scan - tile_start;
	int64_t limit = left < (uint64_t) SORA_TILE_POS? (int64_t) left
		: (int64_t) SORA_TILE_POS;
	for (uint32_t tid = 0; tid < SORA_TILE_THREADS; tid++) {
	    uint32_t pe, po, ve, vo;
	    const uint32_t me = tid ^ 1, nx = (tid + 1) ^ 1;
	    sora_lane_predicate(((uint32_t) s_nue[me] << 16) | s_nue[nx],
		    ((uint32_t) s_nuo[me] << 16) | s_nuo[nx],
		    ((uint32_t) s_de[me] << 16) | s_de[nx],
		    ((uint32_t) s_do[me] << 16) | s_do[nx],
		    l_c[tid] << 16, l_c[tid] & 0xffff0000u, &pe, &po);
	    sora_lane_valid((int) tid, limit, &ve, &vo);
	    pe &= ve;
	    po &= vo;
	    uint32_t both = pe | po;
	    if (pe & po)
		return (size_t) -1;		/* claimed impossible */
	    while (both != 0) {
		uint32_t b = (uint32_t) __builtin_clz(both);
		uint32_t bit = 0x80000000u >> b;
		queue.push_back((uint16_t) (tid * SORA_LANE_POS + 2 * b +
			((po & bit) != 0)));
		both &= ~bit;
	    }
	}
	if (queue.size() > SORA_TILE_POS / 2)
	    return (size_t) -2;

	/* phase 4 */
	for (size_t i = 0; i < queue.size(); i++) {
	    uint32_t q = queue[i];
	    const uint32_t *words = reinterpret_cast<const uint32_t *>(
		    (q & 1)? s_nuo.data() : s_nue.data());
	    uint32_t bits[4];
	    sora_candidate c;

	    sora_extract_bits(words, (q >> 1) + 8, bits);
	    sora_classify(bits, crc_table.data(), &c);
	    uint64_t kind = !c.squitter? SORA_REC_KIND_REPLY :
		    c.syndrome == 0? SORA_REC_KIND_SQUITTER : SORA_REC_KIND_DIRTY;
	    if (kind == SORA_REC_KIND_DIRTY && !emit_all)
		continue;
	    uint64_t rec = sora_rec_make(kind, tile_start + q,
		    kind == SORA_REC_KIND_DIRTY? c.syndrome : c.address);
	    if (n_out < cap)
		records[n_out] = rec;
	    n_out++;
	}
    }
    return n_out;
}

/*
 * The sc16 variant (adsb_scan_kernel<true>): 256 lanes of 32 positions,
 * 32-bit magnitudes, sora_lane_compare32; phases 3 and 4 are the same code.
 */
extern "C" size_t
emul_scan16(const uint8_t *iq, uint64_t n_samples, uint64_t n_scan, int emit_all,
	uint64_t *records, size_t cap) {
    const uint64_t n_tiles = (n_scan + SORA16_TILE_POS - 1) / SORA16_TILE_POS;
    std::vector<uint32_t> s_m(SORA16_M_DWORDS);
    std::vector<uint16_t> s_nue(2 * SORA16_WORDS), s_nuo(2 * SORA16_WORDS),
	    s_de(2 * SORA16_WORDS), s_do(2 * SORA16_WORDS);
    std::vector<uint32_t> l_c(SORA16_TILE_THREADS);
    std::vector<uint32_t> crc_table(SORA_CRC_TABLE_WORDS);
    size_t n_out = 0;
    const int loads = SORA16_TILE_SAMPLES * 4 / 16 / SORA16_TILE_THREADS;

    for (int k = 0; k < SORA_CRC_SLICES; k++)
	for (uint32_t i = 0; i < 256; i++)
	    crc_table[k * 256 + i] = sora_crc24_slice_entry(k, i);

    for (uint64_t tile = 0; tile < n_tiles; tile++) {
	const uint64_t tile_start = tile * SORA16_TILE_POS;

	/* phase 1 */
	for (int r = 0; r < loads; r++)
	    for (uint32_t tid = 0; tid < SORA16_TILE_THREADS; tid++) {
		uint32_t first = (r * SORA16_TILE_THREADS + tid) * 4;
		uint32_t raw[4], mm[4];
		for (int k = 0; k < 4; k++) {
		    uint64_t sample = tile_start + first + k;
		    raw[k] = 0;		/* buffer loads return 0 out of range */
		    if (sample < n_samples)
			memcpy(&raw[k], iq + 4 * sample, 4);
		}
		sora_mag_sc16x4(raw, mm);
		for (int k = 0; k < 4; k++)
		    s_m[first + 4 * (first >> 5) + k] = mm[k];
	    }
	for (int k = 0; k < SORA16_LOOKAHEAD; k++)
	    s_m[SORA16_TILE_THREADS * SORA16_LANE_STRIDE + k] = 0;

	/* phase 2 */
	for (uint32_t tid = 0; tid < SORA16_TILE_THREADS; tid++) {
	    uint32_t M[SORA_LANE_POS + SORA16_LOOKAHEAD], up, down, cc;
	    for (int i = 0; i < SORA_LANE_POS; i++)
		M[i] = s_m[tid * SORA16_LANE_STRIDE + i];
	    for (int i = 0; i < SORA16_LOOKAHEAD; i++)
		M[SORA_LANE_POS + i] = s_m[(tid + 1) * SORA16_LANE_STRIDE + i];
	    sora_lane_compare32(M, &up, &down, &cc);
	    s_nue[tid ^ 1] = (uint16_t) (~up & 0xffffu);
	    s_nuo[tid ^ 1] = (uint16_t) ((~up) >> 16);
	    s_de[tid ^ 1] = (uint16_t) (down & 0xffffu);
	    s_do[tid ^ 1] = (uint16_t) (down >> 16);
	    l_c[tid] = cc;
	}
	for (int k = 0; k < 2; k++) {
	    s_nue[SORA16_TILE_THREADS + k] = s_nuo[SORA16_TILE_THREADS + k] = 0;
	    s_de[SORA16_TILE_THREADS + k] = s_do[SORA16_TILE_THREADS + k] = 0;
	}

	/* phase 3 */
	std::vector<uint16_t> queue;
	uint64_t left = n_scan - tile_start;
	int64_t limit = left < (uint64_t) SORA16_TILE_POS? (int64_t) left
		: (int64_t) SORA16_TILE_POS;
	for (uint32_t tid = 0; tid < SORA16_TILE_THREADS; tid++) {
	    uint32_t pe, po, ve, vo;
	    const uint32_t me = tid ^ 1, nx = (tid + 1) ^ 1;
	    sora_lane_predicate(((uint32_t) s_nue[me] << 16) | s_nue[nx],
		    ((uint32_t) s_nuo[me] << 16) | s_nuo[nx],
		    ((uint32_t) s_de[me] << 16) | s_de[nx],
		    ((uint32_t) s_do[me] << 16) | s_do[nx],
		    l_c[tid] << 16, l_c[tid] & 0xffff0000u, &pe, &po);
	    sora_lane_valid_n((int) tid, limit, SORA16_OWN_LANES, &ve, &vo);
	    pe &= ve;
	    po &= vo;
	    uint32_t both = pe | po;
	    if (pe & po)
		return (size_t) -1;
	    while (both != 0) {
		uint32_t b = (uint32_t) __builtin_clz(both);
		uint32_t bit = 0x80000000u >> b;
		queue.push_back((uint16_t) (tid * SORA_LANE_POS + 2 * b +
			((po & bit) != 0)));
		both &= ~bit;
	    }
	}
	if (queue.size() > SORA16_TILE_POS / 2)
	    return (size_t) -2;

	/* phase 4 */
	for (size_t i = 0; i < queue.size(); i++) {
	    uint32_t q = queue[i];
	    const uint32_t *words = reinterpret_cast<const uint32_t *>(
		    (q & 1)? s_nuo.data() : s_nue.data());
	    uint32_t bits[4];
	    sora_candidate c;

	    sora_extract_bits(words, (q >> 1) + 8, bits);
	    sora_classify(bits, crc_table.data(), &c);
	    uint64_t kind = !c.squitter? SORA_REC_KIND_REPLY :
		    c.syndrome == 0? SORA_REC_KIND_SQUITTER : SORA_REC_KIND_DIRTY;
	    if (kind == SORA_REC_KIND_DIRTY && !emit_all)
		continue;
	    uint64_t rec = sora_rec_make(kind, tile_start + q,
		    kind == SORA_REC_KIND_DIRTY? c.syndrome : c.address);
	    if (n_out < cap)
		records[n_out] = rec;
	    n_out++;
	}
    }
    return n_out;
}

/* bits of one candidate the way phase 4 extracts them, for a direct check */
extern "C" int
emul_tile_geometry(int *tile_pos, int *tile_samples, int *m_dwords) {
    *tile_pos = SORA_TILE_POS;
    *tile_samples = SORA_TILE_SAMPLES;
    *m_dwords = SORA_M_DWORDS;
    return 0;
}

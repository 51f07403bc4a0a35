/*
 * TEST INFRASTRUCTURE -- CPU lane emulator of adsb_scan8_kernel.
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
    std::vector<uint32_t> s_nue(SORA_WORDS), s_nuo(SORA_WORDS), s_de(SORA_WORDS),
	    s_do(SORA_WORDS), w_ce(SORA_WORDS), w_co(SORA_WORDS);
    uint32_t crc_table[256];
    size_t n_out = 0;

    for (uint32_t i = 0; i < 256; i++)
	crc_table[i] = sora_crc24_table_entry(i);

    for (uint64_t tile = 0; tile < n_tiles; tile++) {
	const uint64_t tile_start = tile * SORA_TILE_POS;

	/* phase 1 */
	for (int r = 0; r < 8; r++)
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
		    s_m[pair + k + 4 * ((pair + k) >> 5)] = sora_mag2_pack8(v, flip);
		}
	    }
	for (int k = 0; k < 4; k++)
	    s_m[SORA_TILE_THREADS * SORA_LANE_STRIDE + k] = 0;

	/* phase 2 */
	for (uint32_t tid = 0; tid < SORA_TILE_THREADS; tid++) {
	    uint32_t P[SORA_LANE_STRIDE];
	    for (int i = 0; i < 32; i++)
		P[i] = s_m[tid * SORA_LANE_STRIDE + i];
	    for (int i = 0; i < 4; i++)
		P[32 + i] = s_m[(tid + 1) * SORA_LANE_STRIDE + i];
	    sora_lane_compare(P, &s_nue[tid], &s_nuo[tid], &s_de[tid], &s_do[tid],
		    &w_ce[tid], &w_co[tid]);
	}
	s_nue[SORA_TILE_THREADS] = s_nuo[SORA_TILE_THREADS] = 0;
	s_de[SORA_TILE_THREADS] = s_do[SORA_TILE_THREADS] = 0;

	/* phase 3 */
	std::vector<uint16_t> queue;
	uint64_t left = n_scan - tile_start;
	int64_t limit = left < (uint64_t) SORA_TILE_POS? (int64_t) left
		: (int64_t) SORA_TILE_POS;
	for (uint32_t tid = 0; tid < SORA_TILE_THREADS; tid++) {
	    uint32_t pe, po, ve, vo;
	    sora_lane_predicate(s_nue[tid], s_nuo[tid], s_de[tid], s_do[tid],
		    w_ce[tid], w_co[tid], s_nue[tid + 1], s_nuo[tid + 1],
		    s_de[tid + 1], s_do[tid + 1], &pe, &po);
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
	    const uint32_t *words = (q & 1)? s_nuo.data() : s_nue.data();
	    uint32_t bits[4];
	    sora_candidate c;

	    sora_extract_bits(words, (q >> 1) + 8, bits);
	    sora_classify(bits, crc_table, &c);
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

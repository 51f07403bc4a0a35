/*
 * adsb_kernels.hip -- gfx950 (MI355X) kernels of the ADS-B receive chain.
 *
 * Replaces the per-sample loop of reference src/ads-b/plane-iq.c:141-194 and
 * the validation head of src/ads-b/plane.c:457-484.  Design notes and
 * measurements live in DESIGN.md; the per-lane arithmetic is in adsb_tile.h.
 *
 * Scan kernels (one launch per chunk; a workgroup walks a contiguous range of
 * tiles and writes its candidate records, in position order, into a region of
 * its own, so the concatenation of the regions is sorted by construction and
 * there is no global atomic, compaction or sort):
 *   adsb_scan_direct_kernel<256>   uc8 / sc8, the default.  A lane loads its own
 *       64 bytes (32 positions), keeps the 16-bit magnitudes in registers,
 *       takes the next lane's first pairs over DPP; only the NU bit arrays, the
 *       hit prefix and the hit queue go through LDS: two barriers per tile.
 *       The next tile's samples are in flight while a tile is worked on; every
 *       pass of the tile loop issues the same vector memory operations (four
 *       loads, two stores) so that the wait for the samples is a fixed
 *       `all but the N youngest' that leaves the record stores out.  The
 *       records collect in an LDS ring and leave 1024 at a time.
 *   adsb_scan_kernel<true>         sc16: 32-bit magnitudes staged through LDS.
 *   adsb_scan_kernel<false>        uc8 / sc8 with the magnitudes staged through
 *       LDS (SORA_ADSB_SCAN_VARIANT=0): the independent second implementation
 *       the variant tests compare the default against.
 *   phases of every one of them:
 *     1  16-byte coalesced buffer loads of raw IQ, I^2+Q^2 by v_dot4_i32_i8
 *        (v_dot2_i32_i16 + v_cvt for sc16)
 *     2  compare bits of a lane's 32 positions with packed 16-bit VALU: NU (the
 *        PPM slicer bit of EVERY position), D, C, split by parity
 *     3  predicate = a few word-wide ANDs of shifted words; hits are counted,
 *        prefix-summed (DPP scan + one LDS atomic per wave) and queued in
 *        position order
 *     4  one lane per candidate: 112 message bits are 112 consecutive elements
 *        of an NU array (funnel shifts), CRC-24 by eleven sliced table lookups,
 *        DF classification, a 5-byte record at the slot of its queue index
 * Resolve kernels: first_seen / add_known (known addresses), filter (the one
 * stateful rule as first_seen[A] < P), scan_counts + flatten (offsets, flat
 * list), finalize (re-slices accepted frames from the samples: bits, CRC,
 * amplitude).
 */
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "adsb_kernels.h"
#include "adsb_tile.h"

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));

namespace {

constexpr int kSmall = 256;		/* block size of the list kernels */

/*
 * Tile geometry of the two scan kernels.  The queue (u16 tile-local positions)
 * and the CRC tables reuse the M buffer once phase 2 has consumed it.
 */
template <bool kWide> struct tile_geom;
template <> struct tile_geom<false> {		/* uc8, sc8: 16-bit magnitudes */
    static constexpr int threads = SORA_TILE_THREADS;
    static constexpr int own_lanes = SORA_OWN_LANES;
    static constexpr int tile_pos = SORA_TILE_POS;
    static constexpr int tile_bytes = SORA_TILE_SAMPLES * 2;
    static constexpr int bytes_per_sample = 2;
    static constexpr int lane_stride = SORA_LANE_STRIDE;
    static constexpr int m_dwords = SORA_M_DWORDS;
    static constexpr int m_tail = 16;		/* zeroed dwords behind the last lane */
    static constexpr int words = SORA_WORDS;
#ifdef SORA_WAVES_PER_SIMD
    static constexpr int waves_per_simd = SORA_WAVES_PER_SIMD;
#else
    static constexpr int waves_per_simd = 6;	/* 24 waves per CU */
#endif
};
template <> struct tile_geom<true> {		/* sc16: 32-bit magnitudes */
    static constexpr int threads = SORA16_TILE_THREADS;
    static constexpr int own_lanes = SORA16_OWN_LANES;
    static constexpr int tile_pos = SORA16_TILE_POS;
    static constexpr int tile_bytes = SORA16_TILE_SAMPLES * 4;
    static constexpr int bytes_per_sample = 4;
    static constexpr int lane_stride = SORA16_LANE_STRIDE;
    static constexpr int m_dwords = SORA16_M_DWORDS;
    static constexpr int m_tail = SORA16_LOOKAHEAD;
    static constexpr int words = SORA16_WORDS;
    static constexpr int waves_per_simd = 4;	/* 4 workgroups of 4 waves per CU */
};
constexpr int kCrcCopies = SORA_CRC_TABLE_WORDS / 4;	/* 16-byte pieces of the tables */
constexpr int kFixCopies = SORA_FIX_TABLE_WORDS / 4;	/* ... of the correction lookup */

/* the sliced CRC tables (adsb_tile.h), built at compile time; every tile
 * copies them from L2 into the LDS that M has vacated */
struct crc_slices {
    uint32_t w[SORA_CRC_TABLE_WORDS + SORA_FIX_TABLE_WORDS];
    constexpr crc_slices() : w{} {
	for (int k = 0; k < SORA_CRC_SLICES; k++)
	    for (uint32_t v = 0; v < 256; v++)
		w[k * 256 + v] = sora_crc24_slice_entry(k, v);
	/* behind them, the single-bit correction lookup (copied only when
	 * the option is on) */
	for (int i = 0; i < SORA_FIX_TABLE_WORDS; i++)
	    w[SORA_CRC_TABLE_WORDS + i] = 0xff000000u;
	for (int k = 0; k < 112; k++) {
	    const uint32_t pw = sora_crc_pow(k);
	    w[SORA_CRC_TABLE_WORDS + sora_fix_hash(pw)] = pw | ((uint32_t) k << 24);
	}
    }
};
__device__ const crc_slices g_crc_slices = crc_slices();

__device__ __forceinline__ uint32_t
lane_id() {
    return __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0));
}

/* number of set bits of a wave mask below this lane */
__device__ __forceinline__ uint32_t
rank_in_mask(unsigned long long mask) {
    return __builtin_amdgcn_mbcnt_hi((uint32_t) (mask >> 32),
	    __builtin_amdgcn_mbcnt_lo((uint32_t) mask, 0));
}

/* inclusive prefix sum over the 64 lanes of a wave: six v_add_u32_dpp
 * (row_shr 1/2/4/8 within rows of 16, then row_bcast 15 and 31); lanes a DPP
 * source does not reach read 0 (bound_ctrl) or keep their value (row_mask).
 * The two s_nop cover the VALU-write -> DPP-read hazard of gfx9. */
__device__ __forceinline__ uint32_t
wave_inclusive_scan(uint32_t v) {
    asm("s_nop 1\n\t"
	"v_add_u32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\t"
	"s_nop 1\n\t"
	"v_add_u32_dpp %0, %0, %0 row_shr:2 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\t"
	"s_nop 1\n\t"
	"v_add_u32_dpp %0, %0, %0 row_shr:4 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\t"
	"s_nop 1\n\t"
	"v_add_u32_dpp %0, %0, %0 row_shr:8 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\t"
	"s_nop 1\n\t"
	"v_add_u32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
	"s_nop 1\n\t"
	"v_add_u32_dpp %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf\n\t"
	"s_nop 1"
	: "+v"(v));
    return v;
}

/*
 * Barrier of the scan kernel's tile loop.  Everything the workgroup shares is
 * in LDS, so only LDS traffic has to have landed: __syncthreads() also waits
 * for vmcnt(0), i.e. for the record stores of phase 4 and for loads issued
 * ahead, a memory round trip per tile that nothing needs.
 */
__device__ __forceinline__ void
lds_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

/* ------------------------------------------------------------------ scan -- */

/*
 * Diagnostic build only (-DSORA_SCAN_STAMPS, libsora_adsb_gpu_stamps.so): per
 * phase shader-clock sums of wave 0, stored behind the statistics words.  The
 * shipped library holds no stamp.
 */
#ifdef SORA_SCAN_STAMPS
#define STAMP_DECL uint64_t st_t[8] = {0, 0, 0, 0, 0, 0, 0, 0}; uint64_t st_last = 0
#define STAMP_BEGIN() do { __builtin_amdgcn_sched_barrier(0); \
	st_last = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_sched_barrier(0); } while (0)
#define STAMP(i) do { __builtin_amdgcn_sched_barrier(0); \
	uint64_t now_ = __builtin_amdgcn_s_memtime(); \
	st_t[i] += now_ - st_last; st_last = now_; \
	__builtin_amdgcn_sched_barrier(0); } while (0)
#define STAMP_WAIT_VM() asm volatile("s_waitcnt vmcnt(0)" ::: "memory")
#define STAMP_WAIT_LDS() asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory")
#else
#define STAMP_DECL
#define STAMP_BEGIN()
#define STAMP(i)
#define STAMP_WAIT_VM()
#define STAMP_WAIT_LDS()
#endif

__device__ __forceinline__ void note_known(uint64_t *first_seen, uint32_t *known_bits,
	uint32_t addr, uint64_t position);

/*
 * first_seen_kernel's work for one workgroup's squitter list (about sixty
 * entries of the 1-hour capture), done by the scan kernel as the workgroup
 * ends: four dependent round trips that the other workgroups of the CU cover,
 * instead of a kernel of its own that does nothing else (0.08 ms per capture).
 * Call behind a barrier at which every wave has waited for its list stores;
 * the list is read past the L1.
 */
__device__ __forceinline__ void
merge_squitter_list(const sora_scan_args &a, const uint64_t *sq_out, uint32_t n,
	uint32_t tid, uint32_t threads) {
    if (a.first_seen == nullptr)
	return;
    if (n > a.sq_cap)
	n = a.sq_cap;
    for (uint32_t i = tid; i < n; i += threads) {
	const uint64_t r = __hip_atomic_load(&sq_out[i], __ATOMIC_RELAXED,
		__HIP_MEMORY_SCOPE_AGENT);
	note_known(a.first_seen, a.known_bits, sora_rec_addr(r),
		a.seen_tag | (a.base_position + sora_rec_pos(r)));
    }
}

/* a region that did not fit its share of a list says so (sora_resolve_args::
 * summary); nothing is written when everything fits */
__device__ __forceinline__ void
note_overflow(uint64_t *summary, int word, uint32_t count, uint32_t cap) {
    if (summary != nullptr && count > cap)
	atomicMax(reinterpret_cast<unsigned long long *>(&summary[word]),
		(unsigned long long) count);
}

/* v of the lane `n` places up in this lane's row of 16, wrapping (row_ror) */
template <int n> __device__ __forceinline__ uint32_t
row_rotate(uint32_t v) {
    return (uint32_t) __builtin_amdgcn_update_dpp(0, (int) v, 0x120 + n, 0xf, 0xf, false);
}

__device__ __forceinline__ uint32_t
row_xor(uint32_t v) {
    v ^= row_rotate<8>(v);
    v ^= row_rotate<4>(v);
    v ^= row_rotate<2>(v);
    v ^= row_rotate<1>(v);
    return v;
}

__device__ __forceinline__ uint32_t
row_sum(uint32_t v) {
    v += row_rotate<8>(v);
    v += row_rotate<4>(v);
    v += row_rotate<2>(v);
    v += row_rotate<1>(v);
    return v;
}

__device__ __forceinline__ double
row_sum(double v) {
#define SORA_ROW_STEP(n) do { \
	const uint32_t lo = row_rotate<n>((uint32_t) __double_as_longlong(v)); \
	const uint32_t hi = row_rotate<n>((uint32_t) (__double_as_longlong(v) >> 32)); \
	v += __longlong_as_double((long long) (((uint64_t) hi << 32) | lo)); \
    } while (0)
    SORA_ROW_STEP(8);
    SORA_ROW_STEP(4);
    SORA_ROW_STEP(2);
    SORA_ROW_STEP(1);
#undef SORA_ROW_STEP
    return v;
}

typedef uint32_t u32x4_a2 __attribute__((ext_vector_type(4), aligned(2)));
typedef uint32_t u32x4_a4 __attribute__((ext_vector_type(4), aligned(4)));

/*
 * This lane's share of a squitter's amplitude sum in sc16 (plane-iq.c:72-73 over
 * the 112 samples behind the preamble): eight samples, a dword each; pairs
 * (0,1) (2,3) ... are the two halves of a bit: the float sum of a pair,
 * accumulated in double -- every term is an integer, so the order of the
 * additions does not show.  row_sum() of it over the fourteen lanes that hold
 * samples (the other two hold zeros) is the sum the frame carries.
 */
__device__ __forceinline__ double
squitter_sum_wide(const u32x4 &s0, const u32x4 &s1) {
    const uint32_t d[8] = {s0[0], s0[1], s0[2], s0[3], s1[0], s1[1], s1[2], s1[3]};
    double dsum = 0;
#pragma unroll
    for (int i = 0; i < 8; i += 2) {
	const float s = __fadd_rn(__uint_as_float(sora_mag_sc16(d[i])),
		__uint_as_float(sora_mag_sc16(d[i + 1])));
	dsum += (double) s;
    }
    return dsum;
}

template <bool kWide>
__global__ void __launch_bounds__(tile_geom<kWide>::threads, tile_geom<kWide>::waves_per_simd)
adsb_scan_kernel(sora_scan_args a) {
    using G = tile_geom<kWide>;
    constexpr int kThreads = G::threads;
    constexpr int kWaves = kThreads / 64;
    constexpr int kLoads = G::tile_bytes / 16 / kThreads;	/* 16-byte loads per lane */
    constexpr int kTilePos = G::tile_pos;
    constexpr int kQueueMax = kTilePos / 2;		/* p and p+1 never both hit */
    constexpr int kCrcTableDword = (kQueueMax / 2 + 63) & ~63;	/* past the queue */
    /* The buffer holds M in phases 1-2, then the queue.  The sliced CRC tables
     * (and the correction lookup) follow it there in the sc16 kernel, copied in
     * again every tile; the 8-bit kernel, whose swizzled M needs no padding,
     * has room to keep them in LDS for good (a copy per tile was 5 % of its
     * time). */
    constexpr bool kTablesStay = !kWide;
    constexpr int kTabWords = SORA_CRC_TABLE_WORDS + SORA_FIX_TABLE_WORDS;
    constexpr int kAfterM = kCrcTableDword + (kTablesStay? 0 : kTabWords);
    constexpr int kBufDwords = G::m_dwords > kAfterM? G::m_dwords : kAfterM;

    __shared__ __attribute__((aligned(16))) uint32_t s_m[kBufDwords];
    /* bit arrays, written as 16-bit halves (one per lane), read as 32-bit words */
    __shared__ __attribute__((aligned(4))) uint16_t s_bits[4][2 * G::words];
    uint16_t *const s_nue = s_bits[0], *const s_nuo = s_bits[1];
    uint16_t *const s_de = s_bits[2], *const s_do = s_bits[3];
    __shared__ __attribute__((aligned(16))) uint32_t s_tab[kTablesStay? kTabWords : 4];
    /* the hits of a tile, in position order.  They normally fit this small
     * queue, and then nothing of phase 4 lives in the M buffer: a wave that is
     * done with its candidates goes straight on to the next tile's loads and
     * magnitudes.  A tile with more hits queues them in the M buffer instead
     * and pays a barrier before that buffer is filled again. */
    constexpr int kSmallQueue = kTablesStay? 1536 : 1;
    __shared__ uint16_t s_small_queue[kSmallQueue];
    /* s_cum[w]: hits of the waves before wave w; s_cum[kWaves]: of the tile.
     * Zeroed in phase 1, added to by every wave (one LDS atomic instruction
     * whose lanes are the later waves), read behind the barrier. */
    __shared__ uint32_t s_cum[kWaves + 1];
    __shared__ uint32_t s_sq_total;	/* clean squitters so far, all tiles */
    /* sc16: where in the tile the squitters it added to the list lie (0xffff: a
     * corrected one), for their amplitude sums (sora_scan_args::sq_sum) */
    __shared__ uint16_t s_amp_q[kWide? SORA_AMP_SLOTS : 1];

    const uint32_t tid = threadIdx.x;
    const uint32_t lane = tid & 63, wave = tid >> 6;
    const uint32_t group = blockIdx.x, groups = gridDim.x;
    const uint32_t flip = a.encoding == SORA_ENC_UC8? 0x80808080u : 0u;

    /* contiguous tile range of this workgroup */
    const uint64_t tile_begin = a.n_tiles * group / groups;
    const uint64_t tile_end = a.n_tiles * (group + 1) / groups;

    uint8_t *rec_out = a.rec + group * SORA_REC_REGION_BYTES(a.rec_cap);
    uint64_t *sq_out = a.sq + (uint64_t) group * a.sq_cap;
    u32x4 *sq_bits_out = reinterpret_cast<u32x4 *>(a.sq_bits) + (uint64_t) group * a.sq_cap;
    uint32_t n_rec = 0;			/* workgroup-uniform */
    if (tid == 0)
	s_sq_total = 0;			/* visible behind the first tile's barriers */

    /*
     * sc16: the amplitude sums of a tile's clean squitters, one tile late.  The
     * squitters phase 4 appended to the list while it worked on tile t (entries
     * amp_k0 .. amp_k0 + amp_n of the list, positions amp_start + s_amp_q[]) are
     * summed while tile t + 1 is loaded: sixteen lanes a squitter, fourteen of
     * them 32 bytes each, issued together with the next tile's own loads, so
     * that no round trip is added to the tile.  finalize then has nothing to read
     * for these frames but their list entries.  The bytes are not saved -- by then
     * a tile has left the L2 (128 workgroups an XCD, 32 KB each: the 4 MB of it),
     * and FETCH_SIZE of this kernel on dense traffic is 17.0 GB for 14.4 -- they
     * are fetched where 4 waves a SIMD have other work meanwhile, instead of by
     * a kernel of dependent round trips: scan + 0.17 ms, finalize - 0.67 ms.
     */
    const bool amp_on = kWide && a.sq_sum != nullptr;
    double *const sq_sum_out = amp_on? a.sq_sum + (uint64_t) group * a.sq_cap : nullptr;
    uint32_t amp_k0 = 0, amp_n = 0, sq_seen = 0;	/* workgroup-uniform */
    uint64_t amp_start = 0;
    auto amp_issue = [&](u32x4 (&smp)[2][2], uint32_t (&entry)[2]) {
#pragma unroll
	for (int ps = 0; ps < 2; ps++) {
	    const uint32_t slot = 16 * ps + (tid >> 4), sub = tid & 15;
	    smp[ps][0] = smp[ps][1] = u32x4{0, 0, 0, 0};
	    entry[ps] = ~0u;
	    if (slot < amp_n) {
		const uint32_t q = s_amp_q[slot];
		if (q != 0xffffu && amp_k0 + slot < a.sq_cap) {
		    entry[ps] = amp_k0 + slot;
		    if (sub < 14) {
			const uint8_t *src = a.iq + 4 * (amp_start + q + 16) + 32 * sub;
			smp[ps][0] = *reinterpret_cast<const u32x4_a4 *>(src);
			smp[ps][1] = *reinterpret_cast<const u32x4_a4 *>(src + 16);
		    }
		}
	    }
	}
    };
    auto amp_finish = [&](const u32x4 (&smp)[2][2], const uint32_t (&entry)[2]) {
#pragma unroll
	for (int ps = 0; ps < 2; ps++) {
	    if (16u * ps >= amp_n)
		break;				/* workgroup-uniform */
	    const double total = row_sum(squitter_sum_wide(smp[ps][0], smp[ps][1]));
	    if (entry[ps] != ~0u && (tid & 15) == 0)
		sq_sum_out[entry[ps]] = total;
	}
    };
    uint32_t st_cand = 0, st_sq = 0, st_dirty = 0, st_fixed = 0;

    uint16_t *const s_big_queue = reinterpret_cast<uint16_t *>(s_m);
    uint32_t *s_crc = kTablesStay? s_tab : s_m + kCrcTableDword;
    if constexpr (kTablesStay) {
	const u32x4 *src = reinterpret_cast<const u32x4 *>(g_crc_slices.w);
	for (uint32_t c = tid; c < (uint32_t) kTabWords / 4; c += kThreads)
	    *reinterpret_cast<u32x4 *>(&s_tab[4 * c]) = src[c];
	/* visible to every wave behind the first tile's barriers */
    }
    STAMP_DECL;
    const uint32_t *w_nue = reinterpret_cast<const uint32_t *>(s_nue);
    const uint32_t *w_nuo = reinterpret_cast<const uint32_t *>(s_nuo);

    /* the raw samples of a tile, 16 bytes per lane and load.  (Issuing them a
     * tile ahead, under phases 3 and 4, was measured 20 % SLOWER: the 16 extra
     * live registers spill, and every spill reload waits for the loads.) */
    auto load_tile = [&](u32x4 *v, uint64_t tile_no, uint32_t lane16) {
	const uint64_t start = tile_no * kTilePos;
	uint64_t left = (a.n_samples - start) * G::bytes_per_sample;
	left = (left + 3) & ~3ull;		/* see sora_adsb_scan(): an odd
						 * sample count reads its last dword whole */
	const uint32_t span = left < (uint64_t) G::tile_bytes?
		(uint32_t) left : (uint32_t) G::tile_bytes;
	auto rsrc = __builtin_amdgcn_make_buffer_rsrc(
		const_cast<uint8_t *>(a.iq) + start * G::bytes_per_sample, 0, span,
		0x00020000);
#pragma unroll
	for (int r = 0; r < kLoads; r++)
	    v[r] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, lane16,
		    r * kThreads * 16, 0);
    };

    for (uint64_t tile = tile_begin; tile < tile_end; tile++) {
	const uint64_t tile_start = tile * kTilePos;
	/* the thread index, made opaque once per tile: LDS addresses derived from
	 * it are then formed inside the loop as one register plus the immediate
	 * offset field of the ds instructions; hoisted out of the loop they are a
	 * dozen VGPRs of their own, which at 80 VGPRs means spills */
	uint32_t t = tid;
	asm volatile("" : "+v"(t));
	STAMP_BEGIN();

	/* ---- phase 1: load, magnitude, LDS ---- */
	{
	    u32x4 amp_smp[2][2];
	    uint32_t amp_entry[2];
	    if constexpr (kWide) {
		if (amp_n != 0)
		    amp_issue(amp_smp, amp_entry);
	    }
	    u32x4 v[kLoads];
	    load_tile(v, tile, t * 16);
	    if constexpr (kWide) {
		if (amp_n != 0)
		    amp_finish(amp_smp, amp_entry);
	    }
	    STAMP_WAIT_VM();
	    STAMP(0);			/* HBM latency of the tile's loads */
#pragma unroll
	    for (int r = 0; r < kLoads; r++) {
		/* this load's four M dwords in lane-run order: M dword
		 * (r * kThreads + t) * 4 and on; one load instruction fills
		 * whole lane runs, so r contributes a constant */
		const uint32_t raw[4] = {v[r][0], v[r][1], v[r][2], v[r][3]};
		uint32_t mm[4];
		uint32_t at;
		if constexpr (kWide) {
		    sora_mag_sc16x4(raw, mm);		/* one dword per sample */
		    /* 32 + 4 dwords per lane */
		    at = r * (kThreads * 4 / 32) * G::lane_stride + 4 * t + 4 * (t >> 3);
		} else {
		    sora_mag2_pack8x4(raw, flip, mm);	/* one dword per pair */
		    /* chunk t & 3 of lane r * (kThreads / 4) + (t >> 2), swizzled
		     * (kThreads / 4 is a multiple of 16: r adds a constant) */
		    at = r * (kThreads * 4 / 16) * G::lane_stride +
			    sora_m_chunk(t >> 2, t & 3);
		}
		const u32x4 m = {mm[0], mm[1], mm[2], mm[3]};
		*reinterpret_cast<u32x4 *>(&s_m[at]) = m;
	    }
	    if (tid < G::m_tail)
		s_m[kThreads * G::lane_stride + tid] = 0;
	    if (tid <= kWaves)
		s_cum[tid] = 0;
	}
	STAMP_WAIT_LDS();
	STAMP(1);			/* magnitudes + LDS stores */
	lds_barrier();
	STAMP(2);			/* barrier 1 */

	/* ---- phase 2: compare bits of this lane's 32 positions ---- */
	uint32_t up, down, cc;
	if constexpr (kWide) {
	    uint32_t M[SORA_LANE_POS + SORA16_LOOKAHEAD];
#pragma unroll
	    for (int i = 0; i < (SORA_LANE_POS + SORA16_LOOKAHEAD) / 4; i++) {
		/* the next lane's run starts 36 dwords on: dwords 32..35 are padding */
		const uint32_t at = t * G::lane_stride + (i < 8? 4 * i : 4 * i + 4);
		u32x4 q4 = *reinterpret_cast<const u32x4 *>(&s_m[at]);
		M[4 * i] = q4[0]; M[4 * i + 1] = q4[1];
		M[4 * i + 2] = q4[2]; M[4 * i + 3] = q4[3];
	    }
	    sora_lane_compare32(M, &up, &down, &cc);
	} else {
	    uint32_t P[SORA_LANE_PAIRS + 4];
#pragma unroll
	    for (int i = 0; i < 4; i++) {
		u32x4 q4 = *reinterpret_cast<const u32x4 *>(&s_m[sora_m_chunk(t, i)]);
		P[4 * i] = q4[0]; P[4 * i + 1] = q4[1];
		P[4 * i + 2] = q4[2]; P[4 * i + 3] = q4[3];
	    }
	    {
		u32x4 q4 = *reinterpret_cast<const u32x4 *>(&s_m[sora_m_chunk(t + 1, 0)]);
		P[16] = q4[0]; P[17] = q4[1]; P[18] = q4[2]; P[19] = q4[3];
	    }
	    sora_lane_compare(P, &up, &down, &cc);
	}
	STAMP(3);			/* LDS reads + compare words */
	/* lane g is half g of the arrays; MSB first puts even g in the high half
	 * of a little-endian word, i.e. at 16-bit index g ^ 1 */
	const uint32_t nue16 = ~up & 0xffffu, nuo16 = (~up) >> 16;
	const uint32_t de16 = down & 0xffffu, do16 = down >> 16;
	s_nue[t ^ 1] = (uint16_t) nue16;
	s_nuo[t ^ 1] = (uint16_t) nuo16;
	s_de[t ^ 1] = (uint16_t) de16;
	s_do[t ^ 1] = (uint16_t) do16;
	if (tid < 2) {
	    s_nue[kThreads + tid] = 0;
	    s_nuo[kThreads + tid] = 0;
	    s_de[kThreads + tid] = 0;
	    s_do[kThreads + tid] = 0;
	}
	lds_barrier();	/* M is dead from here on: s_m becomes queue + CRC table */
	STAMP(4);			/* word stores + barrier 2 */

	/* ---- phase 3: predicate words, ordered queue of hits ---- */
	/* the CRC tables for phase 4: fetched now, stored once the scan is done */
	constexpr int kTabs = kTablesStay? 0 : (kCrcCopies + kFixCopies + kThreads - 1) / kThreads;
	/* pieces every lane copies need no test */
#define SORA_ALWAYS(c) (((c) + 1) * kThreads <= kCrcCopies)
	const uint32_t copies = a.correct? kCrcCopies + kFixCopies : kCrcCopies;
	u32x4 tab[kTabs > 0? kTabs : 1];
	if constexpr (!kTablesStay) {
	    const u32x4 *src = reinterpret_cast<const u32x4 *>(g_crc_slices.w);
#pragma unroll
	    for (int c = 0; c < kTabs; c++)
		if (SORA_ALWAYS(c) || c * kThreads + tid < copies)
		    tab[c] = src[c * kThreads + tid];
	}
	uint32_t pe, po;
	{
	    const uint64_t left = a.n_scan - tile_start;
	    const uint32_t nx = (t + 1) ^ 1;	/* the next lane's half */

	    sora_lane_predicate((nue16 << 16) | s_nue[nx], (nuo16 << 16) | s_nuo[nx],
		    (de16 << 16) | s_de[nx], (do16 << 16) | s_do[nx],
		    cc << 16, cc & 0xffff0000u, &pe, &po);
	    if (left >= (uint64_t) kTilePos) {	/* all but the last tile */
		if (tid >= G::own_lanes)
		    pe = po = 0;
	    } else {
		uint32_t ve, vo;
		sora_lane_valid_n((int) tid, (int64_t) left, G::own_lanes, &ve, &vo);
		pe &= ve;
		po &= vo;
	    }
	}

	const uint32_t cnt = __popc(pe | po);	/* a pair never hits on both
						 * parities: D[2j] & U[2j+1] vs D[2j+1] */
	const uint32_t incl = wave_inclusive_scan(cnt);
	{
	    const uint32_t wave_hits = (uint32_t) __builtin_amdgcn_readlane((int) incl, 63);
	    if (lane > wave && lane <= (uint32_t) kWaves)
		atomicAdd(&s_cum[lane], wave_hits);
	}
	if constexpr (!kTablesStay) {
#pragma unroll
	    for (int c = 0; c < kTabs; c++)
		if (SORA_ALWAYS(c) || c * kThreads + tid < copies)
		    *reinterpret_cast<u32x4 *>(&s_crc[4 * c * kThreads + 4 * t]) = tab[c];
	}
	lds_barrier();
	const uint32_t total = s_cum[kWaves];
	uint32_t off = s_cum[wave] + incl - cnt;
	const bool big = !kTablesStay || total > (uint32_t) kSmallQueue;	/* uniform */
	uint16_t *const s_queue = big? s_big_queue : s_small_queue;
	{
	    uint32_t both = pe | po;
	    while (both != 0) {
		/* element b of the lane: position 2b (even array) or 2b + 1 (odd
		 * array; never both, the predicates exclude each other) */
		const uint32_t b = __clz(both);
		const uint32_t twice = __builtin_amdgcn_alignbit(b, po << b, 31);
		s_queue[off++] = (uint16_t) (tid * SORA_LANE_POS | twice);
		both &= ~(0x80000000u >> b);
	    }
	}
	lds_barrier();
	STAMP(5);			/* predicate, block scan, queue (2 barriers) */

	/* ---- phase 4: one lane per candidate ----
	 * Every candidate gets the record slot of its queue index, so the
	 * record list is in position order with no compaction at all (dirty
	 * squitters ride along; the filter drops them).  Clean squitters, rare,
	 * are appended to their own list through an LDS counter; that list
	 * only feeds the first-seen table and needs no order. */
	st_cand += total;
	for (uint32_t base = 0; base < total; base += kThreads) {
	    const uint32_t i = base + tid;

	    if (i < total) {
		uint32_t q = s_queue[i];
		const uint32_t *words = (q & 1)? w_nuo : w_nue;
		uint32_t bits[4];
		sora_candidate c;

		sora_extract_bits(words, (q >> 1) + 8, bits);
		sora_classify(bits, s_crc, &c);
		uint64_t kind = !c.squitter? SORA_REC_KIND_REPLY :
			c.syndrome == 0? SORA_REC_KIND_SQUITTER : SORA_REC_KIND_DIRTY;
		uint32_t payload = kind == SORA_REC_KIND_DIRTY? c.syndrome : c.address;
		if (a.correct && kind == SORA_REC_KIND_DIRTY) {
		    /* option: a syndrome that one flipped bit explains */
		    const int b = sora_fix_bit(c.syndrome, c.is_long? 112 : 56,
			    s_crc + SORA_CRC_TABLE_WORDS);
		    if (b >= 0) {
			kind = SORA_REC_KIND_CORRECTED;
			payload = c.address;		/* bits[8,32) */
			if (b >= 8 && b < 32)
			    payload ^= 1u << (31 - b);
		    }
		}
		/* the full record, and the bits, only for the squitter list: its
		 * candidate record names the list entry instead of the address */
		uint32_t word = payload;
		if (kind == SORA_REC_KIND_SQUITTER || kind == SORA_REC_KIND_CORRECTED) {
		    const uint32_t k = atomicAdd(&s_sq_total, 1u);
		    if (k < a.sq_cap) {
			sq_out[k] = sora_rec_make(kind, tile_start + q, payload);
			sq_bits_out[k] = u32x4{c.bits[0], c.bits[1], c.bits[2], c.bits[3]};
		    }
		    if constexpr (kWide) {
			if (amp_on) {
			    if (k - sq_seen < (uint32_t) SORA_AMP_SLOTS)
				s_amp_q[k - sq_seen] = kind == SORA_REC_KIND_SQUITTER?
					(uint16_t) q : (uint16_t) 0xffffu;
			    else if (k < a.sq_cap)
				sq_sum_out[k] = -1.0;	/* finalize sums this one */
			}
		    }
		    word = k & 0xffffffu;
		}
		/* key and tail (sora_scan_args) */
		const uint32_t idx = n_rec + i;
		if (idx < a.rec_cap) {
		    *reinterpret_cast<uint32_t *>(rec_out + SORA_REC_KEY_AT(idx)) =
			    word | ((uint32_t) kind << 24) | ((q >> 8) << 26);
		    rec_out[SORA_REC_TAIL_AT(idx)] = (uint8_t) q;
		}
		/* wave-uniform counts: ballots land in SGPRs, off the VALU */
		st_sq += (uint32_t) __popcll(__ballot(kind == SORA_REC_KIND_SQUITTER));
		st_dirty += (uint32_t) __popcll(__ballot(kind == SORA_REC_KIND_DIRTY));
		if (a.correct)
		    st_fixed += (uint32_t) __popcll(__ballot(kind == SORA_REC_KIND_CORRECTED));
	    }
	}
	n_rec += total;
	if (tid == 0)
	    a.tile_end[tile] = n_rec;	/* one thin store per tile beside the records' */
	/* The next tile refills only the M buffer before its first barrier, and
	 * a wave passes that barrier after its last read of the queue, the
	 * tables and the NU arrays: no barrier here unless the queue is in M. */
	if (big)
	    lds_barrier();
	if constexpr (kWide) {
	    /* (big: behind the barrier) what this tile added to the squitter list */
	    const uint32_t now = s_sq_total;
	    amp_k0 = sq_seen;
	    amp_n = now - sq_seen < (uint32_t) SORA_AMP_SLOTS? now - sq_seen : SORA_AMP_SLOTS;
	    amp_start = tile_start;
	    sq_seen = now;
	    if (!amp_on)
		amp_n = 0;
	}
	STAMP(6);			/* candidates: extract, CRC, records */
    }

    /* per-workgroup results */
    lds_barrier();			/* the last tile's squitter appends */
    if constexpr (kWide) {
	if (amp_n != 0) {		/* ... and their sums */
	    u32x4 amp_smp[2][2];
	    uint32_t amp_entry[2];
	    amp_issue(amp_smp, amp_entry);
	    amp_finish(amp_smp, amp_entry);
	}
    }
    if (tid == 0) {
	a.rec_count[group] = n_rec;
	a.sq_count[group] = s_sq_total;
	note_overflow(a.summary, SORA_SUM_REC_OVER, n_rec, a.rec_cap);
	note_overflow(a.summary, SORA_SUM_SQ_OVER, s_sq_total, a.sq_cap);
    }
    /* statistics: st_cand is workgroup-uniform, the others wave-uniform */
    {
	__shared__ uint32_t s_stat[kWaves][3];
	if (lane == 0) {
	    s_stat[wave][0] = st_sq;
	    s_stat[wave][1] = st_dirty;
	    s_stat[wave][2] = st_fixed;
	}
	__syncthreads();
	if (tid == 0) {
	    uint64_t t1 = 0, t2 = 0, t3 = 0;
	    for (int w = 0; w < kWaves; w++) {
		t1 += s_stat[w][0];
		t2 += s_stat[w][1];
		t3 += s_stat[w][2];
	    }
	    uint64_t *st = a.stats + (uint64_t) group * SORA_STAT_WORDS;
	    st[SORA_STAT_CANDIDATES] = st_cand;
	    st[SORA_STAT_SQUITTERS] = t1;
	    st[SORA_STAT_DIRTY] = t2;
	    st[SORA_STAT_REPLIES] = st_cand - t1 - t2 - t3;
	    st[SORA_STAT_CORRECTED] = t3;
#ifdef SORA_SCAN_STAMPS
	    for (int i = 0; i < 8; i++)
		st[SORA_STAT_WORDS_BASE + i] = st_t[i];
#endif
	}
    }
    merge_squitter_list(a, sq_out, s_sq_total, tid, kThreads);
}

/*
 * The 8-bit scan with the magnitudes kept in registers.  Lane l of wave w
 * loads ITS OWN 64 bytes (lane run 63 w + l of the tile: four 16-byte loads)
 * and never stores M: the next lane's first four pairs come over DPP
 * (wave_shl:1), and lane 63 of a wave is a halo lane -- the run it holds is
 * computed again, completely, by lane 0 of the next wave -- so phases 1-3
 * need nothing from another wave and no barrier.  What is shared through LDS
 * is the NU bit arrays (phase 4 reads 240 positions ahead), double-buffered
 * by tile parity, the hit prefix, the queue and the ring the records leave
 * through (below): two barriers per tile instead of four, 31 KB of LDS instead
 * of 53.
 */
template <int kThreads> struct direct_geom {
    static constexpr int waves = kThreads / 64;
    static constexpr int lanes = 63 * waves;		/* lane runs computed */
    static constexpr int own = lanes - 8;		/* ... whose positions are emitted */
    static constexpr int tile_pos = own * SORA_LANE_POS;
    static constexpr int tile_bytes = (lanes + 1) * 64;	/* with the last wave's halo run */
};

__device__ __forceinline__ uint32_t
next_lane(uint32_t v) {
    /* lane l reads lane l + 1; lane 63 reads 0 */
    return (uint32_t) __builtin_amdgcn_update_dpp(0, (int) v, 0x130, 0xf, 0xf, true);
}

/*
 * Waves per SIMD (= workgroups per CU) of the direct kernel: 89 VGPRs (with the
 * next tile's samples held while a tile is worked on) and 31 KB of LDS both
 * allow five.  Three to seven ran the 1-hour capture within 2 % of each other
 * before the record ring; with a ring twice as long there would be four.
 */
#ifndef SORA_DIRECT_WAVES
#define SORA_DIRECT_WAVES 5
#endif
/*
 * Candidate records leave the workgroup 1024 at a time.  Measured on this GPU
 * (tools/probe/stream_mix_probe.cpp): beside a streaming read at full rate, a
 * thin stream of stores costs far more than its bytes -- two 6-byte records per
 * tile and workgroup (11 MB in all) slow the 14.4 GB read by 0.44 ms, the 169 a
 * tile has on average by 0.87 ms, wherever they go (own regions, interleaved
 * blocks, lines that stay in L2), whatever their width or cache policy, and
 * although nothing waits for them; the same bytes written every sixth tile cost
 * 0.68 ms, every sixteenth 0.59.  So phase 4 puts a record into a ring in LDS
 * (slot modulo 2048), and once a tile has completed a run of 1024 slots the
 * whole workgroup writes that run with one 16-byte store of keys and one
 * 4-byte store of tails per thread.
 */
constexpr uint32_t kStageUnit = 1024;		/* records per flush: four per thread */
constexpr uint32_t kStageRing = 2 * kStageUnit;	/* a unit being filled behind one being written */
constexpr uint32_t kStagePass = 512;		/* records staged between two flushes, at most */
constexpr int kSqLds = 64;		/* clean squitters a workgroup keeps in LDS (record + bits) */

template <int kThreads>
__global__ void __launch_bounds__(kThreads, SORA_DIRECT_WAVES)
adsb_scan_direct_kernel(sora_scan_args a) {
    using D = direct_geom<kThreads>;
    constexpr int kWaves = D::waves;
    constexpr int kTilePos = D::tile_pos;
    /* the hit queue holds kQueueCap candidates; a tile with more (p and p + 1
     * never both hit: up to kTilePos / 2) is worked off in windows of that many */
    constexpr uint32_t kQueueCap = 2 * kStagePass;
    constexpr int kWords = (D::lanes + 2) / 2 + 1;
    constexpr int kTabWords = SORA_CRC_TABLE_WORDS + SORA_FIX_TABLE_WORDS;

    /* [tile parity][even, odd positions]: 16-bit halves (one per lane run),
     * read as 32-bit words */
    __shared__ __attribute__((aligned(4))) uint16_t s_bits[2][2][2 * kWords];
    __shared__ __attribute__((aligned(16))) uint32_t s_tab[kTabWords];
    __shared__ uint16_t s_queue[kQueueCap];
    __shared__ uint32_t s_cum[2][kWaves + 1];	/* by tile parity, see below */
    __shared__ uint32_t s_sq_total;
    /* the workgroup's clean squitters (about forty per range of the 1-hour
     * capture) and their message bits: kept here and written out as the
     * workgroup ends, so that the tile loop issues no store that only some tiles
     * have (see the tile loop); those beyond kSqLds go straight to the global
     * lists */
    __shared__ uint64_t s_sq[kSqLds];
    __shared__ __attribute__((aligned(16))) u32x4 s_sq_bits[kSqLds];
    /* the records on their way out: slot n of the region at n mod kStageRing */
    __shared__ __attribute__((aligned(16))) uint32_t s_stage_key[kStageRing];
    __shared__ __attribute__((aligned(16))) uint8_t s_stage_tail[kStageRing];
    /* records of the range up to and including each of its tiles (the filter
     * finds a record's tile with it); written out as the workgroup ends */
    __shared__ uint32_t s_tile_end[SORA_RANGE_TILES_MAX];

    const uint32_t tid = threadIdx.x;
    const uint32_t lane = tid & 63, wave = tid >> 6;
    const uint32_t group = blockIdx.x, groups = gridDim.x;
    const uint32_t flip = a.encoding == SORA_ENC_UC8? 0x80808080u : 0u;
    const uint64_t tile_begin = a.n_tiles * group / groups;
    const uint64_t tile_end = a.n_tiles * (group + 1) / groups;

    uint64_t *sq_out = a.sq + (uint64_t) group * a.sq_cap;
    u32x4 *sq_bits_out = reinterpret_cast<u32x4 *>(a.sq_bits) + (uint64_t) group * a.sq_cap;
    uint32_t n_rec = 0;		/* records so far (counted on past the region's end) */
    uint32_t n_flushed = 0;	/* ... of which written out: a multiple of kStageUnit */
    uint32_t st_cand = 0, st_sq = 0, st_dirty = 0, st_fixed = 0;

    /* The record region of this workgroup as two buffers: a store beyond the
     * region (a full region, or a lane that has no record: offset ~0) is
     * dropped by the bounds check -- no branch around the stores. */
    const uint64_t rec_bytes = SORA_REC_REGION_BYTES(a.rec_cap);
    uint8_t *const rec_out = a.rec + group * rec_bytes;
    auto rsrc_rec = __builtin_amdgcn_make_buffer_rsrc(rec_out, 0,
	    rec_bytes > 0xfffffffcull? 0xfffffffcu : (uint32_t) rec_bytes, 0x00020000);

    {
	const u32x4 *src = reinterpret_cast<const u32x4 *>(g_crc_slices.w);
	for (uint32_t c = tid; c < (uint32_t) kTabWords / 4; c += kThreads)
	    *reinterpret_cast<u32x4 *>(&s_tab[4 * c]) = src[c];
	if (tid <= (uint32_t) kWaves)
	    s_cum[0][tid] = s_cum[1][tid] = 0;
	if (tid == 0)
	    s_sq_total = 0;
    }
    __syncthreads();

    /* the 64 bytes of lane run g of a tile: four 16-byte loads.  `live` false:
     * the same four instructions on an empty buffer (they fetch nothing and
     * return zeros) -- see the tile loop for why they are still issued */
    auto load_run = [&](u32x4 *v, uint64_t tile_no, uint32_t g, bool live) {
	const uint64_t start = live? tile_no * kTilePos : 0;
	uint64_t left = live? (a.n_samples - start) * 2 : 0;
	left = (left + 3) & ~3ull;
	const uint32_t span = left < (uint64_t) D::tile_bytes?
		(uint32_t) left : (uint32_t) D::tile_bytes;
	auto rsrc = __builtin_amdgcn_make_buffer_rsrc(
		const_cast<uint8_t *>(a.iq) + start * 2, 0, span, 0x00020000);
#pragma unroll
	for (int r = 0; r < 4; r++)
	    v[r] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, g * 64, r * 16, 0);
    };
    /* a record's key and tail for slot idx of the region, into the ring (what
     * does not fit the region is counted, not kept) */
    auto stage_record = [&](uint32_t idx, uint32_t key, uint32_t tail) {
	if (idx < a.rec_cap) {
	    s_stage_key[idx & (kStageRing - 1)] = key;
	    s_stage_tail[idx & (kStageRing - 1)] = (uint8_t) tail;
	}
    };
    /* Slots [n_flushed, n_flushed + kStageUnit) out of the ring into the region:
     * thread t takes four, a 16-byte store of keys and a 4-byte store of tails
     * (kStageUnit = 4 * kThreads; the region's capacity is a multiple of 8, so
     * the bounds check drops whole stores).  `live` false: the same two stores
     * with an offset out of range -- every pass of the tile loop issues them. */
    static_assert(kStageUnit == 4 * kThreads, "one flush = four records per thread");
    auto flush_unit = [&](uint32_t t, bool live) {
	u32x4 k4 = {0, 0, 0, 0};
	uint32_t t4 = 0;
	const uint32_t first = n_flushed + 4 * t;
	if (live) {
	    k4 = *reinterpret_cast<const u32x4 *>(&s_stage_key[first & (kStageRing - 1)]);
	    t4 = *reinterpret_cast<const uint32_t *>(&s_stage_tail[first & (kStageRing - 1)]);
	}
	/* four records of one unit (the unit is a multiple of four): their keys
	 * and their tails lie in the same 1280-byte unit, a flush is one piece */
	__builtin_amdgcn_raw_buffer_store_b128(k4, rsrc_rec,
		live? SORA_REC_KEY_AT(first) : 0xffffffffu, 0, 0);
	__builtin_amdgcn_raw_buffer_store_b32(t4, rsrc_rec,
		live? SORA_REC_TAIL_AT(first) : 0xffffffffu, 0, 0);
    };

    /*
     * The tile loop is software-pipelined over the memory system: the samples
     * of tile t + 1 are requested as soon as the registers that held tile t's
     * are free (behind phase 1) and arrive while phases 2-4 of tile t run.
     * (Without this a wave waits for HBM at the top of every tile and is
     * parked for half of its life, SQ_WAIT_ANY 53 % of SQ_WAVE_CYCLES; with it
     * the wait is 0.4 % of the tile.  Seven such waves per SIMD and five of
     * these take the same time on the 1-hour capture -- the kernel is bound
     * by VALU issue either way, DESIGN.md 5.1 -- but a short shard no longer
     * pays the latency.)  Vector memory operations retire in issue order, and the wait
     * at the top of the loop is `all but the N youngest', N fixed at compile
     * time: for that wait not to include the record stores of phase 4 -- which
     * come AFTER the loads -- every pass through the loop must issue the SAME
     * operations: four loads and two stores, whatever the tile holds.  Hence
     * loads on an empty buffer behind the last tile, the record flush with an
     * out-of-range offset on the passes that have no full unit to write, the
     * squitter list in LDS, and two dropped stores in front of the loop so that
     * the first pass finds what every other pass finds.  (A tile with more than
     * kStagePass candidates flushes between its parts: more operations behind
     * the loads only make the wait conservative.)
     */
    u32x4 v[4];
    load_run(v, tile_begin, 63 * wave + lane, tile_begin < tile_end);
    flush_unit(tid, false);
    STAMP_DECL;

    for (uint64_t tile = tile_begin; tile < tile_end; tile++) {
	const uint64_t tile_start = tile * kTilePos;
	const uint32_t par = (uint32_t) (tile - tile_begin) & 1;
	/* The thread index, made opaque once per tile (see adsb_scan_kernel): the
	 * LDS addresses derived from it are then formed inside the loop, next to
	 * their use; hoisted out of it they are a dozen VGPRs of loop invariants,
	 * and at this kernel's register budget a spilled invariant is reloaded
	 * behind an s_waitcnt vmcnt(0) that also waits for the next tile's loads.
	 * t: thread, l: lane, w: wave (scalar), g: lane run of this lane. */
	uint32_t t = tid;
	asm volatile("" : "+v"(t));
	const uint32_t l = t & 63;
	const uint32_t w = (uint32_t) __builtin_amdgcn_readfirstlane((int) (t >> 6));
	const uint32_t g = 63 * w + l;

	/* ---- phase 1: this lane's 64 bytes, magnitudes in registers ---- */
	uint32_t P[SORA_LANE_PAIRS + 4];
	STAMP_BEGIN();
#ifdef SORA_SCAN_STAMPS
	asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
#endif
	STAMP(0);			/* what is left of the HBM latency of the tile's loads */
	{
#pragma unroll
	    for (int r = 0; r < 4; r++) {
		const uint32_t raw[4] = {v[r][0], v[r][1], v[r][2], v[r][3]};
		sora_mag2_pack8x4(raw, flip, &P[4 * r]);
	    }
	    load_run(v, tile + 1, g, tile + 1 < tile_end);	/* in flight through phases 2-4 */
#pragma unroll
	    for (int i = 0; i < 4; i++)
		P[16 + i] = next_lane(P[i]);
	}

	STAMP(1);			/* magnitudes, next tile's loads issued */
	/* ---- phase 2: compare bits of this lane's 32 positions ---- */
	uint32_t up, down, cc;
	sora_lane_compare(P, &up, &down, &cc);
	const uint32_t nu = ~up;
	STAMP(2);			/* compares */
	uint16_t *const s_nue = s_bits[par][0], *const s_nuo = s_bits[par][1];
	if (l != 63) {		/* the halo lane's run belongs to the next wave */
	    s_nue[g ^ 1] = (uint16_t) (nu & 0xffffu);
	    s_nuo[g ^ 1] = (uint16_t) (nu >> 16);
	}

	/* ---- phase 3: predicate words, ordered queue of hits ---- */
	uint32_t pe, po;
	{
	    const uint64_t left = a.n_scan - tile_start;
	    const uint32_t nu_next = next_lane(nu), down_next = next_lane(down);
	    /* own 16 elements above the next lane's 16 (the halo lane's first
	     * elements are complete: they need none of ITS lookahead) */
	    sora_lane_predicate(__builtin_amdgcn_perm(nu, nu_next, 0x05040100u),
		    __builtin_amdgcn_perm(nu, nu_next, 0x07060302u),
		    __builtin_amdgcn_perm(down, down_next, 0x05040100u),
		    __builtin_amdgcn_perm(down, down_next, 0x07060302u),
		    cc << 16, cc & 0xffff0000u, &pe, &po);
	    if (left >= (uint64_t) kTilePos) {	/* all but the last tile */
		if (g >= (uint32_t) D::own)
		    pe = po = 0;
	    } else {
		uint32_t ve, vo;
		sora_lane_valid_n((int) g, (int64_t) left, D::own, &ve, &vo);
		pe &= ve;
		po &= vo;
	    }
	    if (l == 63)
		pe = po = 0;
	}
	const uint32_t cnt = __popc(pe | po);
	const uint32_t incl = wave_inclusive_scan(cnt);
	{
	    const uint32_t wave_hits = (uint32_t) __builtin_amdgcn_readlane((int) incl, 63);
	    if (l > w && l <= (uint32_t) kWaves)
		atomicAdd(&s_cum[par][l], wave_hits);
	}
	STAMP(3);			/* predicate, hit count, wave prefix */
	lds_barrier();			/* A: every wave is done with the previous tile */
	STAMP(4);			/* barrier A */
	const uint32_t total = (uint32_t) __builtin_amdgcn_readfirstlane((int) s_cum[par][kWaves]);
	uint32_t off = s_cum[par][w] + incl - cnt;
	/* the records of the tiles before this one are in the ring (every wave has
	 * left their phase 4): a full unit among them goes out now */
	{
	    const bool ready = n_rec - n_flushed >= kStageUnit;
	    flush_unit(t, ready);
	    if (ready)
		n_flushed += kStageUnit;
	}
	/* the other parity's prefix is used next by the following tile's atomics,
	 * which no wave reaches before barrier B */
	if (t <= (uint32_t) kWaves)
	    s_cum[par ^ 1][t] = 0;
	/* ---- phase 4: one lane per candidate (as adsb_scan_kernel) ---- */
	const uint32_t *w_nue = reinterpret_cast<const uint32_t *>(s_nue);
	const uint32_t *w_nuo = reinterpret_cast<const uint32_t *>(s_nuo);
	st_cand += total;
	/* queue entry j: its record (key and tail) and statistics */
	auto candidate = [&](uint32_t j, uint32_t *key, uint32_t *tail) {
	    uint32_t q = s_queue[j];
	    const uint32_t *words = (q & 1)? w_nuo : w_nue;
	    uint32_t bits[4];
	    sora_candidate c;

	    sora_extract_bits(words, (q >> 1) + 8, bits);
	    sora_classify(bits, s_tab, &c);
	    uint32_t kind = !c.squitter? (uint32_t) SORA_REC_KIND_REPLY :
		    c.syndrome == 0? (uint32_t) SORA_REC_KIND_SQUITTER :
		    (uint32_t) SORA_REC_KIND_DIRTY;
	    uint32_t payload = kind == SORA_REC_KIND_DIRTY? c.syndrome : c.address;
	    if (a.correct && kind == SORA_REC_KIND_DIRTY) {
		const int b = sora_fix_bit(c.syndrome, c.is_long? 112 : 56,
			s_tab + SORA_CRC_TABLE_WORDS);
		if (b >= 0) {
		    kind = (uint32_t) SORA_REC_KIND_CORRECTED;
		    payload = c.address;
		    if (b >= 8 && b < 32)
			payload ^= 1u << (31 - b);
		}
	    }
	    uint32_t word = payload;
	    if (kind == SORA_REC_KIND_SQUITTER || kind == SORA_REC_KIND_CORRECTED) {
		/* into the squitter list with its bits; the candidate record
		 * names the list entry instead of the address */
		const uint32_t k = atomicAdd(&s_sq_total, 1u);
		const uint64_t r = sora_rec_make(kind, tile_start + q, payload);
		const u32x4 b4 = {c.bits[0], c.bits[1], c.bits[2], c.bits[3]};
		if (k < (uint32_t) kSqLds) {
		    s_sq[k] = r;
		    s_sq_bits[k] = b4;
		} else if (k < a.sq_cap) {
		    sq_out[k] = r;
		    sq_bits_out[k] = b4;
		}
		word = k & 0xffffffu;
	    }
	    *key = word | (kind << 24) | ((q >> 8) << 26);
	    *tail = q;
	    st_sq += (uint32_t) __popcll(__ballot(kind == SORA_REC_KIND_SQUITTER));
	    st_dirty += (uint32_t) __popcll(__ballot(kind == SORA_REC_KIND_DIRTY));
	    if (a.correct)
		st_fixed += (uint32_t) __popcll(__ballot(kind == SORA_REC_KIND_CORRECTED));
	};
	/* Candidates [win, win + kQueueCap) of the tile at a time: nearly always
	 * all of them (a tile has about 170), so this loop is one pass -- queue,
	 * barrier B, candidates -- and the barriers at its end belong to dense
	 * tiles alone.  (Measured and left out: rotating, from tile to tile, which
	 * wave takes the first 64 candidates, since waves 0 and 1 always work
	 * through phase 4 and wave 3 never does: no change.) */
	for (uint32_t win = 0; ; ) {
	    {
		uint32_t both = pe | po, at = off - win;	/* wraps below the window */
		while (both != 0) {
		    const uint32_t b = __clz(both);
		    const uint32_t twice = __builtin_amdgcn_alignbit(b, po << b, 31);
		    if (at < kQueueCap)
			s_queue[at] = (uint16_t) (g * SORA_LANE_POS | twice);
		    at++;
		    both &= ~(0x80000000u >> b);
		}
	    }
	    STAMP(5);			/* queue */
	    lds_barrier();		/* B */
	    STAMP(6);			/* barrier B */
	    const uint32_t stop = total - win < kQueueCap? total : win + kQueueCap;
	    if (stop - win <= kStagePass) {
		/* the common case (and up to three times it): no barrier, no store */
		for (uint32_t i = win + t; i < stop; i += kThreads) {
		    uint32_t key, tail;
		    candidate(i - win, &key, &tail);
		    stage_record(n_rec + i, key, tail);
		}
	    } else {
		/* kStagePass candidates at a time, and a unit out of the ring
		 * whenever one is full (the ring holds two) */
		for (uint32_t first = win; first < stop; first += kStagePass) {
		    const uint32_t last = stop - first < kStagePass? stop : first + kStagePass;
		    for (uint32_t i = first + t; i < last; i += kThreads) {
			uint32_t key, tail;
			candidate(i - win, &key, &tail);
			stage_record(n_rec + i, key, tail);
		    }
		    lds_barrier();
		    const bool ready = n_rec + last - n_flushed >= kStageUnit;
		    flush_unit(t, ready);
		    if (ready)
			n_flushed += kStageUnit;
		}
	    }
	    win += kQueueCap;
	    if (win >= total)
		break;
	    lds_barrier();		/* the queue is read; the next window fills it */
	}
	n_rec += total;
	if (t == 0 && tile - tile_begin < (uint64_t) SORA_RANGE_TILES_MAX)
	    s_tile_end[tile - tile_begin] = n_rec;
	STAMP(7);			/* candidates */
	/* the next tile writes the other parity's NU arrays and prefix; its queue
	 * writes come behind its barrier A */
    }

    lds_barrier();			/* the last tile's records and squitter appends */
    {
	/* what is left in the ring: fewer than kStageUnit + kStagePass records */

	const uint32_t end = n_rec < a.rec_cap? n_rec : a.rec_cap;
	for (uint32_t i = n_flushed + tid; i < end; i += kThreads) {
	    *reinterpret_cast<uint32_t *>(rec_out + SORA_REC_KEY_AT(i)) =
		    s_stage_key[i & (kStageRing - 1)];
	    rec_out[SORA_REC_TAIL_AT(i)] = s_stage_tail[i & (kStageRing - 1)];
	}
	const uint64_t mine = tile_end - tile_begin;
	for (uint64_t i = tid; i < mine && i < (uint64_t) SORA_RANGE_TILES_MAX; i += kThreads)
	    a.tile_end[tile_begin + i] = s_tile_end[i];
    }
    const uint32_t n_sq = s_sq_total;
    if (tid == 0) {
	a.rec_count[group] = n_rec;
	a.sq_count[group] = n_sq;
	note_overflow(a.summary, SORA_SUM_REC_OVER, n_rec, a.rec_cap);
	note_overflow(a.summary, SORA_SUM_SQ_OVER, n_sq, a.sq_cap);
    }
    {
	__shared__ uint32_t s_stat[kWaves][3];
	if (lane == 0) {
	    s_stat[wave][0] = st_sq;
	    s_stat[wave][1] = st_dirty;
	    s_stat[wave][2] = st_fixed;
	}
	__syncthreads();
	if (tid == 0) {
	    uint64_t t1 = 0, t2 = 0, t3 = 0;
	    for (int w = 0; w < kWaves; w++) {
		t1 += s_stat[w][0];
		t2 += s_stat[w][1];
		t3 += s_stat[w][2];
	    }
	    uint64_t *st = a.stats + (uint64_t) group * SORA_STAT_WORDS;
	    st[SORA_STAT_CANDIDATES] = st_cand;
	    st[SORA_STAT_SQUITTERS] = t1;
	    st[SORA_STAT_DIRTY] = t2;
	    st[SORA_STAT_REPLIES] = st_cand - t1 - t2 - t3;
	    st[SORA_STAT_CORRECTED] = t3;
#ifdef SORA_SCAN_STAMPS
	    for (int i = 0; i < 8; i++)
		st[SORA_STAT_WORDS_BASE + i] = st_t[i];
#endif
	}
    }
    /* the squitter list: out of LDS into the workgroup's global list (export,
     * first_seen_kernel), and -- first_seen_kernel's work, hidden behind the
     * other workgroups of the CU -- into the first-seen table */
    {
	const uint32_t n = n_sq < a.sq_cap? n_sq : a.sq_cap;
	for (uint32_t i = tid; i < n; i += kThreads) {
	    uint64_t r;
	    if (i < (uint32_t) kSqLds) {
		r = s_sq[i];
		sq_out[i] = r;
		sq_bits_out[i] = s_sq_bits[i];
	    } else {
		r = __hip_atomic_load(&sq_out[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
	    }
	    if (a.first_seen != nullptr)
		note_known(a.first_seen, a.known_bits, sora_rec_addr(r),
			a.seen_tag | (a.base_position + sora_rec_pos(r)));
	}
    }
}

/* ------------------------------------------------------- small kernels -- */

__global__ void
fill_u64_kernel(uint64_t *dst, uint64_t value, uint64_t n) {
    uint64_t i = (uint64_t) blockIdx.x * blockDim.x + threadIdx.x;
    const uint64_t stride = (uint64_t) gridDim.x * blockDim.x;

    for (; i < n; i += stride)
	dst[i] = value;
}

/*
 * Known addresses: first_seen[A] = smallest global position of an accepted
 * DF11/17/18 frame from A (64-bit, 128 MiB, touched sparsely) and a 2 MiB
 * bitmap of the same set that stays in L2 for the filter's random lookups.
 * The plain pre-check is only a shortcut: first_seen values never grow, so a
 * stale read can cost an atomic, never skip a needed one.
 */
__device__ __forceinline__ void
note_known(uint64_t *first_seen, uint32_t *known_bits, uint32_t addr,
	uint64_t position) {
    if (first_seen[addr] > position)
	atomicMin(reinterpret_cast<unsigned long long *>(&first_seen[addr]),
		(unsigned long long) position);
    const uint32_t bit = 1u << (addr & 31);
    if ((known_bits[addr >> 5] & bit) == 0 &&
	    (atomicOr(&known_bits[addr >> 5], bit) & bit) == 0) {
	/* the thread that set the bit also sets it in the fold: same bit of
	 * word (addr mod 2^17) / 32 */
	uint32_t *fold = known_bits + SORA_KNOWN_BITMAP_WORDS;
	atomicOr(&fold[(addr & ((1u << SORA_KNOWN_FOLD_BITS) - 1)) >> 5], bit);
    }
}

/*
 * The same set in the hashed form (sora_resolve_args::seen_keys): linear
 * probing, a slot is claimed by one atomicCAS on its key (keys only ever go
 * from empty to an address), the position is an atomicMin like above.  A full
 * table drops the address and says so in *seen_fail.
 */
__device__ __forceinline__ uint32_t
seen_slot(uint32_t addr, uint32_t mask) {
    return ((addr * 0x9e3779b1u) >> 8) & mask;
}

__device__ __forceinline__ void
note_known_hashed(const sora_resolve_args &a, uint32_t addr, uint64_t position) {
    uint32_t slot = seen_slot(addr, a.seen_mask);
    bool placed = false;
    for (uint32_t probe = 0; probe <= a.seen_mask; probe++) {
	uint32_t k = __hip_atomic_load(&a.seen_keys[slot], __ATOMIC_RELAXED,
		__HIP_MEMORY_SCOPE_AGENT);
	if (k == SORA_SEEN_EMPTY)
	    k = atomicCAS(&a.seen_keys[slot], SORA_SEEN_EMPTY, addr);
	if (k == SORA_SEEN_EMPTY || k == addr) {
	    atomicMin(reinterpret_cast<unsigned long long *>(&a.first_seen[slot]),
		    (unsigned long long) position);
	    placed = true;
	    break;
	}
	slot = (slot + 1) & a.seen_mask;
    }
    if (!placed) {
	atomicOr(a.seen_fail, 1u);
	return;			/* not in the bitmap either: its replies are dropped */
    }
    const uint32_t bit = 1u << (addr & 31);
    if ((atomicOr(&a.known_bits[addr >> 5], bit) & bit) == 0) {
	uint32_t *fold = a.known_bits + SORA_KNOWN_BITMAP_WORDS;
	atomicOr(&fold[(addr & ((1u << SORA_KNOWN_FOLD_BITS) - 1)) >> 5], bit);
    }
}

/* first-seen position of a known address (its bit is set in the bitmap) */
__device__ __forceinline__ uint64_t
seen_lookup_hashed(const sora_resolve_args &a, uint32_t addr) {
    uint32_t slot = seen_slot(addr, a.seen_mask);
    for (uint32_t probe = 0; probe <= a.seen_mask; probe++) {
	const uint32_t k = a.seen_keys[slot];
	if (k == addr)
	    return a.first_seen[slot];
	if (k == SORA_SEEN_EMPTY)
	    break;
	slot = (slot + 1) & a.seen_mask;
    }
    return ~0ull;
}

/* one block per workgroup region of the scan (one wave per region, several
 * regions per wave, measured slower: 0.082 -> 0.107 ms, the kernel waits for the
 * few hundred hot lines of first_seen, not for block starts) */
template <bool kHashed>
__global__ void
first_seen_kernel(sora_resolve_args a) {
    const uint32_t group = blockIdx.x;
    uint32_t n = a.sq_count[group];
    const uint64_t *sq = a.sq + (uint64_t) group * a.sq_cap;

    if (n > a.sq_cap)
	n = a.sq_cap;
    for (uint32_t i = threadIdx.x; i < n; i += blockDim.x) {
	uint64_t r = sq[i];
	const uint64_t position = a.seen_tag | (a.base_position + sora_rec_pos(r));
	if constexpr (kHashed)
	    note_known_hashed(a, sora_rec_addr(r), position);
	else
	    note_known(a.first_seen, a.known_bits, sora_rec_addr(r), position);
    }
}

__global__ void
add_known_kernel(uint64_t *first_seen, uint32_t *known_bits, uint64_t seen_tag,
	const uint64_t *records, uint64_t n) {
    uint64_t i = (uint64_t) blockIdx.x * blockDim.x + threadIdx.x;
    const uint64_t stride = (uint64_t) gridDim.x * blockDim.x;

    for (; i < n; i += stride) {
	uint64_t r = records[i];
	note_known(first_seen, known_bits, (uint32_t) r & 0xffffffu, seen_tag | (r >> 24));
    }
}

/*
 * The exchange blocks of several shards, as an all-gather leaves them: block b
 * is `stride` words, word 0 the number of records the shard exported, then the
 * records.  The count is read here, on the device: no host round trip between
 * the collective and the resolve.  Block `skip` (this shard's own) is left out.
 */
__global__ void
add_known_blocks_kernel(uint64_t *first_seen, uint32_t *known_bits, uint64_t seen_tag,
	const uint64_t *blocks, uint32_t stride, uint32_t skip) {
    const uint32_t b = blockIdx.y;
    if (b == skip)
	return;
    const uint64_t *blk = blocks + (uint64_t) b * stride;
    uint64_t n = blk[0];
    if (n > stride - 1)
	n = stride - 1;			/* the host reports the overflow */
    for (uint64_t i = (uint64_t) blockIdx.x * blockDim.x + threadIdx.x; i < n;
	    i += (uint64_t) gridDim.x * blockDim.x) {
	const uint64_t r = blk[1 + i];
	note_known(first_seen, known_bits, (uint32_t) r & 0xffffffu, seen_tag | (r >> 24));
    }
}

/* offsets[i] = sum_{j<i} min(counts[j], cap); offsets[n] = total.  One block
 * of 16 waves: a wave takes a contiguous sixteenth of the counts in chunks of
 * 64 (coalesced loads and stores), sums it, the block adds up the waves, and
 * the wave goes over its chunks again (they are in L2) with a running base and
 * one wave scan per chunk.  kNarrow: 64 * cap < 2^32, a chunk's prefix sums fit 32 bits
 * and take the DPP scan (a dependent chain of 64-bit shuffles per chunk was
 * what the kernel spent its time on). */
template <bool kNarrow>
__global__ void __launch_bounds__(1024)
scan_counts_kernel(const uint32_t *counts, uint32_t cap, uint64_t *offsets,
	uint32_t n, uint64_t *total) {
    __shared__ uint64_t s_wave[16];
    const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint32_t per = ((n + 1023) / 1024) * 64;	/* counts per wave */
    const uint32_t begin = wave * per < n? wave * per : n;
    const uint32_t end = begin + per < n? begin + per : n;

    constexpr int kUnroll = 8;		/* chunk loads in flight */
    uint64_t sum = 0;
    for (uint32_t first = begin; first < end; first += 64 * kUnroll) {
	uint32_t c[kUnroll];
#pragma unroll
	for (int u = 0; u < kUnroll; u++) {
	    const uint32_t i = first + u * 64 + lane;
	    c[u] = i < end? counts[i] : 0;
	}
#pragma unroll
	for (int u = 0; u < kUnroll; u++)
	    sum += c[u] < cap? c[u] : cap;
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1)
	sum += __shfl_xor(sum, d, 64);
    if (lane == 0)
	s_wave[wave] = sum;
    __syncthreads();
    uint64_t base = 0;
    for (uint32_t w = 0; w < wave; w++)
	base += s_wave[w];
    for (uint32_t first = begin; first < end; first += 64 * kUnroll) {
	uint32_t c[kUnroll];
#pragma unroll
	for (int u = 0; u < kUnroll; u++) {
	    const uint32_t i = first + u * 64 + lane;
	    c[u] = i < end? counts[i] : 0;
	    c[u] = c[u] < cap? c[u] : cap;
	}
#pragma unroll
	for (int u = 0; u < kUnroll; u++) {
	    const uint32_t i = first + u * 64 + lane;
	    uint64_t incl;
	    if constexpr (kNarrow) {
		incl = wave_inclusive_scan(c[u]);
	    } else {
		incl = c[u];
#pragma unroll
		for (int d = 1; d < 64; d <<= 1) {
		    const uint64_t o = __shfl_up(incl, d, 64);
		    if (lane >= (uint32_t) d)
			incl += o;
		}
	    }
	    if (i < end)
		offsets[i] = base + incl - c[u];
	    base += kNarrow? (uint64_t) (uint32_t) __builtin_amdgcn_readlane((int) (uint32_t) incl, 63) :
		    __shfl(incl, 63, 64);
	}
    }
    if (tid == 1023) {
	offsets[n] = base;	/* the last wave's base has passed every count */
	if (total != nullptr)
	    *total = base;
    }
}

/*
 * What another time shard has to learn from this one: per address, the
 * EARLIEST clean squitter of this scan -- the record whose position is the
 * address's entry in first_seen (first_seen_kernel ran before).  A few hundred
 * records per hour instead of every squitter.  Flat records
 * (global position << 24 | address), unordered, counted by one atomic each.
 */
__global__ void
export_first_kernel(sora_resolve_args a, uint64_t *dst, uint64_t dst_cap,
	unsigned long long *counter) {
    const uint32_t group = blockIdx.x;
    uint32_t n = a.sq_count[group];
    const uint64_t *sq = a.sq + (uint64_t) group * a.sq_cap;

    if (n > a.sq_cap)
	n = a.sq_cap;
    for (uint32_t i = threadIdx.x; i < n; i += blockDim.x) {
	const uint64_t r = sq[i];
	const uint64_t position = a.base_position + sora_rec_pos(r);
	const uint32_t addr = sora_rec_addr(r);
	if (a.first_seen[addr] == (a.seen_tag | position)) {
	    const unsigned long long k = atomicAdd(counter, 1ull);
	    if (k < dst_cap)
		dst[k] = (position << 24) | addr;
	}
    }
}

/*
 * The stateful rule of plane.c:479-483 as a pure function: a reply at global
 * position P with address A is accepted iff first_seen[A] < P.  Each wave
 * takes a contiguous part of one scan region (a.filter_parts parts per region)
 * and compacts it, in order, into an accepted region of its own
 * (region = filter_parts * group + part): no barriers behind the LDS fill.
 *
 * An iteration is 512 records, eight consecutive ones per lane (two 16-byte
 * loads of keys, one 8-byte load of tails), in two stages:
 *   1  per record, without a branch: kind, address, its bit in the LDS copy of
 *      the fold -> `maybe`, the lane's records that need a closer look (clean and
 *      corrected squitters, replies whose fold bit is set: about one record in a
 *      hundred; everything in SORA_EMIT_CANDIDATES mode)
 *   2  those records, compacted through a 64-entry queue of the wave in LDS (a
 *      wave prefix sum of the lanes' counts gives the slots, so the queue is in
 *      record order), ONE per lane: the record's tile from the region's
 *      tile_end[] slice, its position, the exact bitmap and first_seen in L2,
 *      and the kept ones out behind a ballot.
 * (Until round 3 stage 2 went through the lane's eight records one after the
 * other under nested branches, with every lane of the wave waiting on the one
 * or two that had something: 700 instructions per iteration, a third of them
 * scalar mask bookkeeping; the filter was bound by that, not by its 0.77 GB.)
 */
/* floor(num / den) for num < 2^53 and den < 2^32, by one division in double: an
 * integer quotient comes out exactly, and any other lies at least 1 / den from
 * an integer, further than the division's rounding (num / den * 2^-53) can move
 * it.  The scan kernel divides the same numbers as integers. */
__device__ __forceinline__ uint64_t
exact_quotient(uint64_t num, uint32_t den) {
    return (uint64_t) ((double) num / (double) den);
}

constexpr int kFilterWaves = kSmall / 64;
#ifndef SORA_FILTER_UNROLL
#define SORA_FILTER_UNROLL 8
#endif
constexpr int kFilterUnroll = SORA_FILTER_UNROLL;	/* records per lane and iteration (16, and
							 * the next iteration's loads issued ahead:
							 * both measured, no change) */
constexpr uint32_t kFilterQueue = 64;	/* survivors of stage 1 looked at per round */
/* a wave's loop is one round trip to HBM per iteration, so the kernel is as fast
 * as there are waves to overlap them: regions are split until there are this
 * many parts (24 waves per CU of a 256-CU part) */
constexpr uint32_t kFilterTargetParts = 6144;

template <bool kHashed>
__global__ void __launch_bounds__(kSmall)
filter_kernel(sora_resolve_args a, uint32_t range_tiles) {
    /* the 2^17-bit fold of the known-address bitmap: with a few hundred planes
     * known it passes a fraction of a percent of the noise replies, and only
     * those go on to the exact bitmap and the first-seen table in L2 */
    __shared__ uint32_t s_fold[SORA_KNOWN_FOLD_WORDS];
    /* per wave: key in the low word; tail | index within the iteration << 8 above */
    __shared__ uint64_t s_queue[kFilterWaves][kFilterQueue];
    /* per wave: the record counts up to the end of each tile of its scan region
     * (sora_scan_args::tile_end); a record's position counts from its tile.
     * Dynamic, as long as the longest range of this scan (`range_tiles`): the
     * kernel is as fast as the CU holds waves */
    extern __shared__ uint32_t s_tile_end[];
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t region = blockIdx.x * kFilterWaves + wave;
    const bool live = region < a.regions;		/* wave-uniform */
    const uint32_t group = live? region / a.filter_parts : 0;
    const uint32_t part = live? region % a.filter_parts : 0;
    /* the tiles of the scan region, as the scan kernel computes them (quotients
     * by way of double, see exact_quotient: a 64-bit integer division is a
     * hundred instructions, and a wave's whole part is a few thousand) */
    const uint64_t tile_first = exact_quotient(a.n_tiles * group, a.groups);
    uint32_t n_region_tiles = (uint32_t) (exact_quotient(a.n_tiles * (group + 1), a.groups) -
	    tile_first);
    if (n_region_tiles > range_tiles)
	n_region_tiles = range_tiles;			/* not reached: the launcher computes it */
    uint32_t *const te = s_tile_end + wave * range_tiles;
    uint64_t *const queue = s_queue[wave];
    {
	const u32x4 *src = reinterpret_cast<const u32x4 *>(a.known_bits +
		SORA_KNOWN_BITMAP_WORDS);
	for (uint32_t i = threadIdx.x; i < SORA_KNOWN_FOLD_WORDS / 4; i += kSmall)
	    *reinterpret_cast<u32x4 *>(&s_fold[4 * i]) = src[i];
	if (live)
	    for (uint32_t j = lane; j < n_region_tiles; j += 64)
		te[j] = a.tile_end[tile_first + j];
    }
    __syncthreads();
    if (!live)				/* no barrier below */
	return;
    const uint8_t *rec = a.rec + group * SORA_REC_REGION_BYTES(a.rec_cap);
    uint64_t *acc = a.acc + (uint64_t) region * a.acc_cap;
    uint32_t n = a.rec_count[group];
    uint32_t n_acc = 0;

    if (n > a.rec_cap)
	n = a.rec_cap;
    /* where a record's tile is looked for first: tiles hold about the same number */
    const float tiles_per_record = n > 0? (float) n_region_tiles / (float) n : 0.0f;
    /* parts begin at multiples of eight records: with rec_cap a multiple of
     * eight (sora_adsb_scan) a lane's eight keys are 32 aligned bytes and its
     * eight tails 8 */
    constexpr uint32_t kAlign = ~(uint32_t) (kFilterUnroll - 1);
    const uint32_t begin = (uint32_t) exact_quotient((uint64_t) n * part, a.filter_parts) & kAlign;
    const uint32_t end = part + 1 == a.filter_parts? n :
	    (uint32_t) exact_quotient((uint64_t) n * (part + 1), a.filter_parts) & kAlign;
    const uint32_t all = a.emit_all != 0? 1u : 0u;

    for (uint32_t base = begin; base < end; base += 64 * kFilterUnroll) {
	const uint32_t i0 = base + lane * kFilterUnroll;
	uint32_t k[kFilterUnroll], t[kFilterUnroll];
	{
	    u32x4 kq[kFilterUnroll / 4];
	    uint32_t tq[kFilterUnroll / 4];
#pragma unroll
	    for (int q = 0; q < kFilterUnroll / 4; q++) {
		kq[q] = u32x4{0, 0, 0, 0};
		tq[q] = 0;
	    }
	    if (i0 < end) {		/* the loads may pass `end`, never the list */
#pragma unroll
		for (int q = 0; q < kFilterUnroll / 4; q++)
		    kq[q] = *reinterpret_cast<const u32x4 *>(rec + SORA_REC_KEY_AT(i0) + 16 * q);
		if constexpr (kFilterUnroll == 16) {
		    const u32x4 tt = *reinterpret_cast<const u32x4 *>(rec + SORA_REC_TAIL_AT(i0));
		    tq[0] = tt[0]; tq[1] = tt[1]; tq[2] = tt[2]; tq[3] = tt[3];
		} else {
		    const u32x2 tt = *reinterpret_cast<const u32x2 *>(rec + SORA_REC_TAIL_AT(i0));
		    tq[0] = tt[0]; tq[1] = tt[1];
		}
	    }
#pragma unroll
	    for (int u = 0; u < kFilterUnroll; u++) {
		k[u] = kq[u / 4][u % 4];
		t[u] = tq[u / 4];	/* the tail is byte u % 4 of it: cut out for survivors only */
	    }
	}
	/* ---- stage 1: which records need a closer look (no branches) ---- */
	uint32_t maybe = 0;
#pragma unroll
	for (int u = 0; u < kFilterUnroll; u++) {
	    const uint32_t kind = (k[u] >> 24) & 3u;
	    const uint32_t addr = k[u] & 0xffffffu;
	    /* a squitter's payload is its list index, not an address: its lookup is
	     * idle work, cheaper than a branch around it */
	    const uint32_t bit = (s_fold[(addr & ((1u << SORA_KNOWN_FOLD_BITS) - 1)) >> 5] >>
		    (addr & 31)) & 1u;
	    /* kind: 0 reply, 1 squitter, 2 dirty, 3 corrected */
	    const uint32_t open = (kind == SORA_REC_KIND_REPLY? bit : kind) & 1u;
	    maybe |= open << u;
	}
	{
	    /* the records of this lane that exist; everything in emit-all mode */
	    constexpr uint32_t kAll = (1u << kFilterUnroll) - 1u;
	    const uint32_t left = i0 < end? end - i0 : 0u;
	    const uint32_t valid = left >= (uint32_t) kFilterUnroll? kAll : (1u << left) - 1u;
	    maybe = (all != 0? kAll : maybe) & valid;
	}
	const uint32_t cnt = __popc(maybe);
	if (__ballot(cnt != 0) == 0)
	    continue;				/* wave-uniform */
	/* ---- stage 2: those records, one per lane, in record order ---- */
	const uint32_t incl = wave_inclusive_scan(cnt);
	const uint32_t total = (uint32_t) __builtin_amdgcn_readlane((int) incl, 63);
	for (uint32_t round = 0; round < total; round += kFilterQueue) {
	    {
		uint32_t slot = incl - cnt - round;	/* wraps below the round's window */
#pragma unroll
		for (int u = 0; u < kFilterUnroll; u++)
		    if (maybe & (1u << u)) {
			if (slot < kFilterQueue)
			    queue[slot] = (uint64_t) k[u] |
				    ((uint64_t) (((t[u] >> (8 * (u % 4))) & 0xffu) |
				    ((lane * kFilterUnroll + u) << 8)) << 32);
			slot++;
		    }
	    }
	    /* the queue is this wave's alone and a wave's LDS operations complete
	     * in order: no barrier, only no reordering by the compiler */
	    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
	    __builtin_amdgcn_wave_barrier();
	    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
	    const uint32_t here = total - round < kFilterQueue? total - round : kFilterQueue;
	    bool keep = false;
	    uint64_t rec = 0;
	    if (lane < here) {
		const uint64_t e = queue[lane];
		const uint32_t kk = (uint32_t) e, hi = (uint32_t) (e >> 32);
		const uint32_t i = base + (hi >> 8);		/* the record's slot in the region */
		const uint32_t kind = (kk >> 24) & 3u, addr = kk & 0xffffffu;
		/* its tile: from a proportional guess, then along tile_end[] */
		uint32_t tl = (uint32_t) ((float) i * tiles_per_record);
		if (tl >= n_region_tiles)
		    tl = n_region_tiles - 1;
		while (tl > 0 && te[tl - 1] > i)
		    tl--;
		while (tl + 1 < n_region_tiles && te[tl] <= i)
		    tl++;
		const uint64_t pos = (tile_first + tl) * a.tile_pos + (((kk >> 26) << 8) | (hi & 0xffu));
		bool accepted = (kind & 1u) != 0;		/* clean or corrected squitter */
		if (kind == SORA_REC_KIND_REPLY)		/* is its address known, and since when? */
		    accepted = ((a.known_bits[addr >> 5] >> (addr & 31)) & 1) &&
			    (kHashed? seen_lookup_hashed(a, addr) : a.first_seen[addr]) <
			    (a.seen_tag | (a.base_position + pos));
		rec = sora_rec_make(kind, pos, addr);
		if (accepted)
		    rec |= SORA_REC_ACCEPTED << 60;
		keep = accepted || all != 0;
	    }
	    const unsigned long long kept = __ballot(keep);
	    if (keep) {
		const uint32_t idx = n_acc + rank_in_mask(kept);
		if (idx < a.acc_cap)
		    acc[idx] = rec;
	    }
	    n_acc += (uint32_t) __popcll(kept);
	    __builtin_amdgcn_wave_barrier();	/* the next round refills the queue */
	}
    }
    if (lane == 0) {
	a.acc_count[region] = n_acc;
	note_overflow(a.summary, SORA_SUM_ACC_OVER, n_acc, a.acc_cap);
    }
}

/* accepted regions -> one flat, position-sorted list */
__global__ void __launch_bounds__(kSmall)
flatten_kernel(sora_resolve_args a) {
    const uint32_t region = blockIdx.x;
    const uint64_t *acc = a.acc + (uint64_t) region * a.acc_cap;
    uint32_t n = a.acc_count[region];
    const uint64_t off = a.acc_offset[region];

    if (n > a.acc_cap)
	n = a.acc_cap;
    for (uint32_t i = threadIdx.x; i < n; i += kSmall)
	if (off + i < a.frames_cap)
	    a.acc_flat[off + i] = acc[i];
}

/* x^n mod G for n = 0..111, G = 0x1FFF409: the CRC contribution of one bit */
struct crc_powers {
    uint32_t r[112];
    constexpr crc_powers() : r{} {
	uint32_t v = 1;
	for (int n = 0; n < 112; n++) {
	    r[n] = v;
	    v <<= 1;
	    if (v & 0x1000000u)
		v ^= 0x1fff409u;
	}
    }
};
__constant__ crc_powers c_crc_pow = crc_powers();

__device__ __forceinline__ float
magnitude_f32(const uint8_t *iq, uint64_t i, int encoding) {
    /* row a1; float result is exact for the 8-bit encodings */
    if (encoding == SORA_ENC_SC16) {
	/* the definition of sora_mag_sc16 (adsb_tile.h), scaled by 2^-30 */
	const uint32_t w = reinterpret_cast<const uint32_t *>(iq)[i];
	return __uint_as_float(sora_mag_sc16(w)) * (1.0f / 1073741824.0f);
    }
    /* one 16-bit load per sample (I in the low byte) */
    const uint32_t w = reinterpret_cast<const uint16_t *>(iq)[i];
    int vi, vq;
    if (encoding == SORA_ENC_UC8) {
	vi = (int) (w & 0xff) - 128;
	vq = (int) (w >> 8) - 128;
    } else {
	vi = (int8_t) (w & 0xff);
	vq = (int8_t) (w >> 8);
    }
    return (float) (vi * vi + vq * vq) * (1.0f / 16384.0f);
}

/*
 * finalize: re-slices every accepted position from the raw samples in HBM: the
 * independent second implementation of the slicer (plane-iq.c:58-86) and of
 * the CRC (plane.c:373-385), and the only place amplitudes are computed (only
 * accepted frames need them).
 *
 * Sixteen lanes per frame, four frames per wave and round.  Lane `sub` of a
 * group holds the bit pairs k = 16 j + sub, j = 0..6: one load per pair (the
 * four bytes of two 8-bit samples, 2-byte aligned; eight bytes for sc16), so a
 * group reads 64 contiguous bytes per load.  A compare's wave mask holds 16
 * message bits of each of the four frames; the CRC is the XOR over set bits
 * of x^n mod G, reduced over the 16 lanes of a DPP row.
 */
typedef uint32_t u32_a2 __attribute__((aligned(2)));

constexpr int kFramePairs = 7;		/* bit pairs per lane: 7 * 16 = 112 */

/* x^n mod G of the bits a lane holds */
struct group_powers {
    uint32_t pl[6];	/* x^(111 - k): bit k of the 88-bit span of a long message */
    uint32_t ps[2];	/* x^(55 - k): bit k of the 32-bit span of a short one */
};

__device__ __forceinline__ group_powers
load_group_powers(uint32_t sub) {
    group_powers w;
#pragma unroll
    for (int j = 0; j < 6; j++) {
	const int k = 16 * j + (int) sub;
	w.pl[j] = k < 88? c_crc_pow.r[111 - k] : 0;
    }
#pragma unroll
    for (int j = 0; j < 2; j++)
	w.ps[j] = c_crc_pow.r[55 - (16 * j + (int) sub)];
    return w;
}

template <bool kWide> struct frame_part {
    uint32_t w[kWide? 2 * kFramePairs : kFramePairs];
};

/* `given`: the frame's bits are known (a squitter: the scan kept them), only the
 * amplitude is wanted: the 112 samples of bits 0..55, pairs j < 4 */
template <bool kWide> __device__ __forceinline__ frame_part<kWide>
load_frame_part(const uint8_t *iq, uint64_t p, uint32_t sub, bool given) {
    frame_part<kWide> f;
    if constexpr (kWide) {
	const uint32_t *src = reinterpret_cast<const uint32_t *>(iq) + (p + 16) + 2 * sub;
#pragma unroll
	for (int j = 0; j < kFramePairs; j++) {
	    f.w[2 * j] = f.w[2 * j + 1] = 0;
	    if (j < 4 || !given) {
		f.w[2 * j] = src[32 * j];
		f.w[2 * j + 1] = src[32 * j + 1];
	    }
	}
    } else {
	const u32_a2 *src = reinterpret_cast<const u32_a2 *>(iq + 2 * (p + 16)) + sub;
#pragma unroll
	for (int j = 0; j < kFramePairs; j++) {
	    f.w[j] = 0;
	    if (j < 4 || !given)
		f.w[j] = src[16 * j];
	}
    }
    return f;
}

/*
 * The bits the scan kept for a clean or corrected squitter (sora_scan_args::
 * sq_bits), as this lane's seven: bit j of the result is message bit 16 j + sub.
 * The record names the entry within its scan region's list; the region is the
 * one whose tile range holds the position (the quotients are exact in double:
 * positions stay below 2^36, tiles * groups below 2^41, and a quotient that is
 * not an integer is further from one than a double rounds).
 */
__device__ __forceinline__ uint64_t
squitter_entry(const sora_resolve_args &a, uint64_t r) {
    const uint64_t tile = (uint64_t) ((double) sora_rec_pos(r) / (double) a.tile_pos);
    const uint64_t group = (uint64_t) ((double) ((tile + 1) * a.groups - 1) / (double) a.n_tiles);
    const uint32_t k = sora_rec_addr(r);
    if (group >= a.groups || k >= a.sq_cap)
	return 0;				/* not reached: such a scan is run again */
    return group * a.sq_cap + k;
}

__device__ __forceinline__ uint32_t
load_given_bits(const sora_resolve_args &a, uint64_t r, uint32_t sub) {
    const uint32_t *w = a.sq_bits + 4 * squitter_entry(a, r);
    uint32_t mine = 0;
#pragma unroll
    for (int j = 0; j < kFramePairs; j++)
	mine |= ((w[j >> 1] >> (31 - 16 * (j & 1) - sub)) & 1u) << j;
    return mine;
}

/* frame f of the list, in the form the context hands out (plane.c:461-480 for
 * the fields derived from the bits) */
template <bool kWire> __device__ __forceinline__ void
store_frame(const sora_resolve_args &a, uint64_t f, uint64_t r, const uint32_t bits[4],
	uint32_t syndrome, int corrected, uint32_t sum56, double a56, uint64_t amp) {
    const uint32_t df = bits[0] >> 27;
    const bool is_long = df >= 16 && df <= 24;
    const bool squitter = df == 11 || df == 17 || df == 18;
    const uint64_t pos = a.base_position + sora_rec_pos(r);
    const uint32_t flags = (is_long? SORA_FRAME_LONG : 0) |
	    (squitter? SORA_FRAME_SQUITTER : 0) |
	    (corrected >= 0? SORA_FRAME_CORRECTED : 0) |
	    ((sora_rec_kind(r) & SORA_REC_ACCEPTED)? SORA_FRAME_ACCEPTED : 0);
    if constexpr (kWire) {
	sora_adsb_wire_frame w;
	w.bits[0] = bits[0]; w.bits[1] = bits[1]; w.bits[2] = bits[2];
	w.bits3_cb_flags = (bits[3] & 0xffff0000u) | (((uint32_t) corrected & 0xffu) << 8) | flags;
	w.position_lo = (uint32_t) pos;
	w.pos_hi_amp_hi = ((uint32_t) (pos >> 32) << 24) | ((uint32_t) (amp >> 32) & 0xffffffu);
	w.amp_lo = (uint32_t) amp;
	reinterpret_cast<sora_adsb_wire_frame *>(a.frames)[f] = w;
    } else {
	sora_adsb_frame fr;
	fr.position = pos;
	fr.bits[0] = bits[0]; fr.bits[1] = bits[1];
	fr.bits[2] = bits[2]; fr.bits[3] = bits[3];
	fr.amplitude_56 = a56;
	fr.syndrome = syndrome;
	fr.address = squitter? (bits[0] & 0xffffffu) : syndrome;
	fr.sum56 = sum56;
	fr.df = (uint8_t) df;
	fr.corrected_bit = (int8_t) corrected;
	fr.flags = (uint8_t) flags;
	fr.reserved = 0;
	a.frames[f] = fr;
    }
}

/*
 * Four frames, one per row of the wave: frame index f and record r are this
 * row's.  Every sum below is exact in any order: the 8-bit terms are integers;
 * the sc16 terms are integer-valued floats below 2^33 (times 2^-31), 56 of
 * them, so the reference's in-order double accumulation (plane-iq.c:72-73)
 * never rounds either.
 */
template <bool kWide, bool kWire> __device__ __forceinline__ void
finalize_four(const sora_resolve_args &a, uint64_t f, bool valid, uint64_t r,
	const frame_part<kWide> &part, bool given, uint32_t given_bits, uint32_t lane,
	const group_powers &pw, uint32_t flip, bool any_corrected) {
    const uint32_t sub = lane & 15, row_shift = lane & 48;
    uint32_t chunk[kFramePairs];
    uint32_t x_long = 0, x_short = 0, sum56 = 0;
    double dsum = 0;

#pragma unroll
    for (int j = 0; j < kFramePairs; j++) {
	bool nu;			/* m[p+17+2k] <= m[p+16+2k] */
	const bool counts = j < 3 || (j == 3 && sub < 8);	/* k < 56 */
	if constexpr (kWide) {
	    const uint32_t m0 = sora_mag_sc16(part.w[2 * j]);
	    const uint32_t m1 = sora_mag_sc16(part.w[2 * j + 1]);
	    nu = m1 <= m0;		/* non-negative floats order as their bits */
	    if (j < 4) {
		const float s = __fadd_rn(__uint_as_float(m0), __uint_as_float(m1));
		dsum += counts? (double) s : 0.0;
	    }
	} else {
	    const uint32_t x = part.w[j] ^ flip;
	    const int both = __builtin_amdgcn_sdot4((int) x, (int) x, 0, false);
	    const int second = __builtin_amdgcn_sdot4((int) x, (int) (x & 0xffff0000u), 0, false);
	    nu = second <= both - second;
	    if (j < 4)
		sum56 += counts? (uint32_t) both : 0u;
	}
	if (given)			/* this row's frame came with its bits */
	    nu = (given_bits >> j) & 1u;
	const unsigned long long mask = __ballot(nu);
	chunk[j] = (uint32_t) (mask >> row_shift) & 0xffffu;
	if (j < 6)
	    x_long ^= nu? pw.pl[j] : 0u;
	if (j < 2)
	    x_short ^= nu? pw.ps[j] : 0u;
    }
    uint32_t bits[4];
    bits[0] = __brev(chunk[0] | (chunk[1] << 16));
    bits[1] = __brev(chunk[2] | (chunk[3] << 16));
    bits[2] = __brev(chunk[4] | (chunk[5] << 16));
    bits[3] = __brev(chunk[6]);

    /* plane.c:461-475 */
    const uint32_t df = bits[0] >> 27;
    const bool is_long = df >= 16 && df <= 24;
    const uint32_t crc = row_xor(is_long? x_long : x_short);
    const uint32_t ap = is_long? (((bits[2] & 0xffu) << 16) | (bits[3] >> 16))
	    : (bits[1] >> 8);
    const uint32_t syndrome = crc ^ ap;

    /* option: the scan marked this squitter as one flipped bit away from a
     * clean one; find the bit (syndrome == x^k mod G, b = n - 1 - k) and flip
     * it back */
    int corrected = -1;
    if (any_corrected) {				/* wave-uniform, rare */
	uint32_t found = 0;
#pragma unroll
	for (int j = 0; j < kFramePairs; j++) {
	    const uint32_t k = 16 * j + sub;
	    if (c_crc_pow.r[k] == syndrome)
		found = k + 1;
	}
	found = row_sum(found);				/* at most one lane has it */
	const int nbits = is_long? 112 : 56;
	if ((sora_rec_kind(r) & 3) == SORA_REC_KIND_CORRECTED && found != 0 &&
		(int) found - 1 <= nbits - 6) {
	    corrected = nbits - (int) found;
#pragma unroll
	    for (int i = 0; i < 4; i++)
		if ((corrected >> 5) == i)
		    bits[i] ^= 0x80000000u >> (corrected & 31);
	}
    }

    double a56;
    uint64_t amp;		/* the integer the amplitude is a multiple of */
    if constexpr (kWide) {
	const double total = row_sum(dsum);		/* an integer below 2^40 */
	a56 = total * (1.0 / 2147483648.0);		/* float (s0+s1)/2, s = m * 2^-30 */
	amp = (uint64_t) total;
	sum56 = 0;
    } else {
	sum56 = row_sum(sum56);
	a56 = (double) sum56 * (1.0 / 32768.0);		/* exact: see sora_adsb_frame */
	amp = sum56;
    }
    if (valid && sub == 0)
	store_frame<kWire>(a, f, r, bits, syndrome, corrected, sum56, a56, amp);
}

/* blocks per CU of the finalize kernel: 82 VGPRs hold 5 at a time; measured
 * resolve 0.518 ms with 8, 0.507 with 5, 0.54 with 6, 0.505 with 12, 0.498 with
 * 24 (short-lived blocks even out the tail) */
#ifndef SORA_FINALIZE_BLOCKS
#define SORA_FINALIZE_BLOCKS 24
#endif
constexpr int kFinalizeDepth = 2;	/* rounds (of four frames) whose sample fetches are in flight */

/* lane i takes the value of lane i - n of its row of 16, lanes below n take 0 */
template <int n> __device__ __forceinline__ uint32_t
row_shift_right(uint32_t v) {
    return (uint32_t) __builtin_amdgcn_update_dpp(0, (int) v, 0x110 + n, 0xf, 0xf, true);
}

/*
 * The clean squitters of a batch -- nearly every frame of a capture -- need no
 * slicing and no CRC here: the scan kept their bits (sora_scan_args::sq_bits)
 * and their syndrome is 0.  What is left is the amplitude (plane-iq.c:72-73), the
 * sum over the 112 samples behind the preamble: 224 contiguous bytes (448 for
 * sc16), 32 per lane, seven lanes of eight per frame (fourteen of sixteen for
 * sc16), eight (four) frames per round of the wave, and a handful of
 * instructions per frame where the slicing path (finalize_four) spends seventy.
 * `fast`: bit i set = frame i of the batch is one of them.
 */
template <bool kWide, bool kWire> __device__ __forceinline__ void
finalize_squitters(const sora_resolve_args &a, uint64_t batch, uint32_t cnt, uint64_t mine,
	unsigned long long fast, uint32_t lane, uint32_t flip) {
    constexpr uint32_t kLanes = kWide? 16 : 8;		/* per frame */
    constexpr uint32_t kFrames = 64 / kLanes;		/* per round */
    constexpr uint32_t kActive = kWide? 14 : 7;		/* lanes of a frame that load */
    const uint32_t sub = lane % kLanes, grp = lane / kLanes;

    if constexpr (kWide) {
	/* the scan kernel has summed most of them already (sora_scan_args::sq_sum):
	 * a lane a frame, its list entry and nothing else */
	if (a.sq_sum != nullptr) {
	    const bool cand = lane < cnt && ((fast >> lane) & 1) != 0;
	    double total = -1.0;
	    uint64_t entry = 0;
	    if (cand) {
		entry = squitter_entry(a, mine);
		total = a.sq_sum[entry];
	    }
	    const bool ready = cand && total >= 0.0;
	    if (ready) {
		const u32x4 w = *reinterpret_cast<const u32x4 *>(a.sq_bits + 4 * entry);
		const uint32_t bits[4] = {w[0], w[1], w[2], w[3]};
		store_frame<kWire>(a, batch + lane, mine, bits, 0u, -1, 0u,
			total * (1.0 / 2147483648.0), (uint64_t) total);
	    }
	    fast &= ~__ballot(ready);
	}
    }
    for (uint32_t j = 0; j < cnt; j += kFrames) {
	if (((fast >> j) & ((1ull << kFrames) - 1)) == 0)
	    continue;					/* wave-uniform */
	const uint32_t idx = j + grp;
	const bool is = idx < cnt && ((fast >> idx) & 1) != 0;
	const uint32_t k = idx < cnt? idx : cnt - 1;
	const uint64_t r = (uint64_t) (uint32_t) __shfl((int) (uint32_t) mine, (int) k, 64) |
		((uint64_t) (uint32_t) __shfl((int) (uint32_t) (mine >> 32), (int) k, 64) << 32);
	const uint64_t p = sora_rec_pos(r);
	u32x4 s0 = {0, 0, 0, 0}, s1 = {0, 0, 0, 0}, w = {0, 0, 0, 0};
	if (is) {
	    w = *reinterpret_cast<const u32x4 *>(a.sq_bits + 4 * squitter_entry(a, r));
	    if (sub < kActive) {
		if constexpr (kWide) {
		    const uint8_t *src = a.iq + 4 * (p + 16) + 32 * sub;
		    s0 = *reinterpret_cast<const u32x4_a4 *>(src);
		    s1 = *reinterpret_cast<const u32x4_a4 *>(src + 16);
		} else {
		    const uint8_t *src = a.iq + 2 * (p + 16) + 32 * sub;
		    s0 = *reinterpret_cast<const u32x4_a2 *>(src);
		    s1 = *reinterpret_cast<const u32x4_a2 *>(src + 16);
		}
	    }
	}
	const uint32_t d[8] = {s0[0], s0[1], s0[2], s0[3], s1[0], s1[1], s1[2], s1[3]};
	uint32_t sum56 = 0;
	double a56;
	uint64_t amp;
	bool writer;
	if constexpr (kWide) {
	    const double total = row_sum(squitter_sum_wide(s0, s1));
	    a56 = total * (1.0 / 2147483648.0);
	    amp = (uint64_t) total;
	    writer = sub == 0;
	} else {
	    /* a dword is a pair: I0^2 + Q0^2 + I1^2 + Q1^2 in one dot product */
	    int acc = 0;
#pragma unroll
	    for (int i = 0; i < 8; i++) {
		const uint32_t x = d[i] ^ flip;
		acc = __builtin_amdgcn_sdot4((int) x, (int) x, acc, false);
	    }
	    /* the lanes that did not load hold zeros ^ flip: they are left out */
	    uint32_t v = sub < kActive? (uint32_t) acc : 0u;
	    v += row_shift_right<4>(v);
	    v += row_shift_right<2>(v);
	    v += row_shift_right<1>(v);			/* lane 7 of the eight: their sum */
	    sum56 = v;
	    a56 = (double) sum56 * (1.0 / 32768.0);
	    amp = sum56;
	    writer = sub == kLanes - 1;
	}
	if (is && writer) {
	    const uint32_t bits[4] = {w[0], w[1], w[2], w[3]};
	    store_frame<kWire>(a, batch + idx, r, bits, 0u, -1, sum56, a56, amp);
	}
    }
}

template <bool kWide, bool kWire>
__global__ void __launch_bounds__(kSmall)
finalize_kernel(sora_resolve_args a) {
    const uint32_t flip = a.encoding == SORA_ENC_UC8? 0x80808080u : 0u;
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t sub = lane & 15, row = lane >> 4;
    const group_powers pw = load_group_powers(sub);
    const uint64_t n_waves = (uint64_t) gridDim.x * (kSmall / 64);
    const uint64_t gw = (uint64_t) blockIdx.x * (kSmall / 64) + wave;
    uint64_t total = a.acc_offset[a.regions];

    if (total > a.frames_cap)
	total = a.frames_cap;
    if (a.n_samples < SORA_WINDOW)
	total = 0;
    /* a wave takes up to 64 consecutive frames at a time: one coalesced load
     * brings their records (no dependent record fetch per frame); row g of round
     * i works on frame 4 i + g.  A short list goes out in batches of 32 or 16 so
     * that every wave of the grid has one: the rounds of a batch are dependent
     * trips to memory, and a 900 M-sample shard's 0.2 M frames in batches of 64
     * keep one wave in seven busy for eight rounds each (33 -> 30 us). */
    const uint64_t per = total >= 64 * n_waves? 64 : total >= 32 * n_waves? 32 : 16;
    for (uint64_t batch = gw * per; batch < total; batch += n_waves * per) {
	const uint32_t cnt = total - batch < per? (uint32_t) (total - batch) : (uint32_t) per;
	const uint64_t mine = lane < cnt? a.acc_flat[batch + lane] : 0;
	const bool any_corrected =
		__ballot(lane < cnt && (sora_rec_kind(mine) & 3) == SORA_REC_KIND_CORRECTED) != 0;
	/* clean squitters whose bits the scan kept: the short way */
	const unsigned long long fast = __ballot(lane < cnt && a.sq_bits != nullptr &&
		(sora_rec_kind(mine) & 3) == SORA_REC_KIND_SQUITTER &&
		sora_rec_pos(mine) + SORA_WINDOW <= a.n_samples);
	if (fast != 0)
	    finalize_squitters<kWide, kWire>(a, batch, cnt, mine, fast, lane, flip);

	/* everything else is sliced from the samples, four frames a round */
	for (uint32_t j = 0; j < cnt; j += 4 * kFinalizeDepth) {
	    if (((~fast >> j) & ((1ull << (4 * kFinalizeDepth)) - 1) &
		    (cnt - j >= 64? ~0ull : (1ull << (cnt - j)) - 1)) == 0)
		continue;				/* wave-uniform: nothing left here */
	    uint64_t r[kFinalizeDepth];
	    uint32_t idx[kFinalizeDepth], given_bits[kFinalizeDepth];
	    bool sane[kFinalizeDepth], given[kFinalizeDepth];
	    frame_part<kWide> part[kFinalizeDepth];
#pragma unroll
	    for (int u = 0; u < kFinalizeDepth; u++) {
		idx[u] = j + 4 * u + row;
		const uint32_t k = idx[u] < cnt? idx[u] : cnt - 1;
		r[u] = (uint64_t) (uint32_t) __shfl((int) (uint32_t) mine, (int) k, 64) |
			((uint64_t) (uint32_t) __shfl((int) (uint32_t) (mine >> 32), (int) k, 64) << 32);
		/* the scan only records p with p + 240 <= n_samples; a record that
		 * says otherwise is corrupt, and following it would read outside
		 * the capture: such a frame is left out */
		sane[u] = sora_rec_pos(r[u]) + SORA_WINDOW <= a.n_samples;
		/* a squitter's bits are in the scan's list: only the 112 samples
		 * of its amplitude are read; a reply is sliced here */
		const bool done = idx[u] < cnt && ((fast >> idx[u]) & 1) != 0;
		sane[u] = sane[u] && !done;
		given[u] = a.sq_bits != nullptr && sane[u] && (sora_rec_kind(r[u]) & 1u) != 0;
		given_bits[u] = given[u]? load_given_bits(a, r[u], sub) : 0u;
		part[u] = load_frame_part<kWide>(a.iq, sane[u]? sora_rec_pos(r[u]) : 0, sub,
			given[u]);
	    }
#pragma unroll
	    for (int u = 0; u < kFinalizeDepth; u++)
		if (j + 4 * u < cnt)				/* wave-uniform */
			finalize_four<kWide, kWire>(a, batch + idx[u], idx[u] < cnt && sane[u], r[u],
				part[u], given[u], given_bits[u], lane, pw, flip, any_corrected);
	}
    }
}

/*
 * sora_adsb_decode_noalloc(): what a context finds out on the host from the
 * per-region counters (fetch_counts in adsb_gpu.hip), found out on the device
 * instead, since that entry point never waits: status[0] = frames available,
 * status[1] = SORA_NOALLOC_* flags of what did not fit.
 */
__global__ void __launch_bounds__(kSmall)
noalloc_status_kernel(sora_resolve_args a, uint64_t cap_frames, uint64_t *status) {
    __shared__ uint32_t s_flags;
    if (threadIdx.x == 0)
	s_flags = 0;
    __syncthreads();
    uint32_t flags = 0;
    for (uint32_t i = threadIdx.x; i < a.groups; i += kSmall) {
	if (a.rec_count[i] > a.rec_cap)
	    flags |= SORA_NOALLOC_RECORDS_FULL;
	if (a.sq_count[i] > a.sq_cap)
	    flags |= SORA_NOALLOC_SQUITTERS_FULL;
    }
    for (uint32_t i = threadIdx.x; i < a.regions; i += kSmall)
	if (a.acc_count[i] > a.acc_cap)
	    flags |= SORA_NOALLOC_FRAMES_FULL;
    if (flags != 0)
	atomicOr(&s_flags, flags);
    __syncthreads();
    if (threadIdx.x == 0) {
	const uint64_t total = a.acc_offset[a.regions];
	flags = s_flags;
	if (total > a.frames_cap || total > cap_frames)
	    flags |= SORA_NOALLOC_FRAMES_FULL;
	if (a.seen_fail != nullptr && *a.seen_fail != 0)
	    flags |= SORA_NOALLOC_ADDRESSES_FULL;
	status[0] = total;
	status[1] = flags;
    }
}

__global__ void
magnitude_kernel(const uint8_t *iq, uint64_t n, int encoding, float *m) {
    uint64_t i = (uint64_t) blockIdx.x * blockDim.x + threadIdx.x;
    const uint64_t stride = (uint64_t) gridDim.x * blockDim.x;

    for (; i < n; i += stride)
	m[i] = magnitude_f32(iq, i, encoding);
}

uint32_t
grid_for(uint64_t n, uint32_t block) {
    uint64_t g = (n + block - 1) / block;
    return g < 1? 1 : g > 8192? 8192 : (uint32_t) g;
}

} /* namespace */

/* ---------------------------------------------------------- launchers -- */

int
sora_scan_merges(int encoding, int variant) {
    (void) encoding;
    (void) variant;
    return 1;		/* every scan kernel does */
}

uint32_t
sora_scan_tile_pos(int encoding, int variant) {
    if (encoding != SORA_ENC_SC16 && variant == 1)
	return direct_geom<256>::tile_pos;
    return encoding == SORA_ENC_SC16? SORA16_TILE_POS : SORA_TILE_POS;
}

uint64_t
sora_scan_tiles(uint64_t n_scan, int encoding, int variant) {
    const uint64_t pos = sora_scan_tile_pos(encoding, variant);
    return (n_scan + pos - 1) / pos;
}

uint32_t
sora_scan_groups_per_cu(int encoding, int variant) {
    if (encoding != SORA_ENC_SC16 && variant == 1)
	return SORA_DIRECT_WAVES;	/* 256 threads: as many workgroups as waves per SIMD */
    /* workgroups of one CU: bounded by LDS (53 KB and 39 KB per workgroup) */
#ifdef SORA_GROUPS_PER_CU
    return encoding == SORA_ENC_SC16? 4 : SORA_GROUPS_PER_CU;
#else
    return encoding == SORA_ENC_SC16? 4 : 3 * 512 / SORA_TILE_THREADS;
#endif
}

hipError_t
sora_launch_scan(const sora_scan_args *a, uint32_t groups, int variant, hipStream_t stream) {
    if (a->encoding != SORA_ENC_SC16 && variant == 1) {
	(void) hipGetLastError();
	hipLaunchKernelGGL(adsb_scan_direct_kernel<256>, dim3(groups), dim3(256), 0,
		stream, *a);
	return hipGetLastError();
    }
    switch (a->encoding) {
    case SORA_ENC_UC8:
    case SORA_ENC_SC8:
	(void) hipGetLastError();
	hipLaunchKernelGGL(adsb_scan_kernel<false>, dim3(groups),
		dim3(tile_geom<false>::threads), 0, stream, *a);
	break;
    case SORA_ENC_SC16:
	(void) hipGetLastError();
	hipLaunchKernelGGL(adsb_scan_kernel<true>, dim3(groups),
		dim3(tile_geom<true>::threads), 0, stream, *a);
	break;
    default:
	return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

hipError_t
sora_launch_fill_u64(uint64_t *dst, uint64_t value, uint64_t n,
	hipStream_t stream) {
    (void) hipGetLastError();
    hipLaunchKernelGGL(fill_u64_kernel, dim3(grid_for(n, 1024)), dim3(256), 0,
	    stream, dst, value, n);
    return hipGetLastError();
}

hipError_t
sora_launch_first_seen(const sora_resolve_args *a, hipStream_t stream) {
    (void) hipGetLastError();
    if (a->seen_keys != nullptr)
	hipLaunchKernelGGL(first_seen_kernel<true>, dim3(a->groups), dim3(kSmall), 0,
		stream, *a);
    else
	hipLaunchKernelGGL(first_seen_kernel<false>, dim3(a->groups), dim3(kSmall), 0,
		stream, *a);
    return hipGetLastError();
}

hipError_t
sora_launch_add_known(uint64_t *first_seen, uint32_t *known_bits, uint64_t seen_tag,
	const uint64_t *records, uint64_t n, hipStream_t stream) {
    if (n == 0)
	return hipSuccess;
    (void) hipGetLastError();
    hipLaunchKernelGGL(add_known_kernel, dim3(grid_for(n, 256)), dim3(256), 0,
	    stream, first_seen, known_bits, seen_tag, records, n);
    return hipGetLastError();
}

hipError_t
sora_launch_add_known_blocks(uint64_t *first_seen, uint32_t *known_bits, uint64_t seen_tag,
	const uint64_t *blocks, uint32_t n_blocks, uint32_t stride, uint32_t skip,
	hipStream_t stream) {
    if (n_blocks == 0 || stride < 2)
	return hipSuccess;
    (void) hipGetLastError();
    hipLaunchKernelGGL(add_known_blocks_kernel, dim3(grid_for(stride - 1, 256) > 16?
	    16 : grid_for(stride - 1, 256), n_blocks), dim3(256), 0, stream,
	    first_seen, known_bits, seen_tag, blocks, stride, skip);
    return hipGetLastError();
}

hipError_t
sora_launch_scan_counts(const uint32_t *counts, uint32_t cap,
	uint64_t *offsets, uint32_t n, uint64_t *total, hipStream_t stream) {
    (void) hipGetLastError();
    if (cap < (1u << 26))
	hipLaunchKernelGGL(scan_counts_kernel<true>, dim3(1), dim3(1024), 0, stream,
		counts, cap, offsets, n, total);
    else
	hipLaunchKernelGGL(scan_counts_kernel<false>, dim3(1), dim3(1024), 0, stream,
		counts, cap, offsets, n, total);
    return hipGetLastError();
}

hipError_t
sora_launch_export_first(const sora_resolve_args *a, uint64_t *dst, uint64_t dst_cap,
	uint64_t *counter, hipStream_t stream) {
    hipError_t e = hipMemsetAsync(counter, 0, sizeof(uint64_t), stream);
    if (e != hipSuccess)
	return e;
    (void) hipGetLastError();
    hipLaunchKernelGGL(export_first_kernel, dim3(a->groups), dim3(256), 0, stream,
	    *a, dst, dst_cap, reinterpret_cast<unsigned long long *>(counter));
    return hipGetLastError();
}

hipError_t
sora_launch_filter(const sora_resolve_args *a, hipStream_t stream) {
    (void) hipGetLastError();
    const dim3 grid((a->regions + kFilterWaves - 1) / kFilterWaves);
    /* tiles of the longest scan region: n_tiles * (g + 1) / groups - n_tiles * g / groups */
    const uint64_t range = a->n_tiles / (a->groups > 0? a->groups : 1) + 1;
    if (range > SORA_RANGE_TILES_MAX)
	return hipErrorInvalidValue;		/* sora_adsb_scan refuses such ranges */
    const uint32_t range_tiles = (uint32_t) range;
    const size_t lds = (size_t) kFilterWaves * range_tiles * sizeof(uint32_t);
    if (a->seen_keys != nullptr)
	hipLaunchKernelGGL(filter_kernel<true>, grid, dim3(kSmall), lds, stream, *a, range_tiles);
    else
	hipLaunchKernelGGL(filter_kernel<false>, grid, dim3(kSmall), lds, stream, *a, range_tiles);
    return hipGetLastError();
}

hipError_t
sora_launch_noalloc_status(const sora_resolve_args *a, uint64_t cap_frames, uint64_t *status,
	hipStream_t stream) {
    (void) hipGetLastError();
    hipLaunchKernelGGL(noalloc_status_kernel, dim3(1), dim3(kSmall), 0, stream, *a,
	    cap_frames, status);
    return hipGetLastError();
}

uint32_t
sora_filter_parts(uint32_t groups) {
    /* at most 8 parts per region: the accepted list's capacity is shared out
     * over the regions, and a short capture has few scan regions */
    const uint32_t parts = groups == 0? 1 : (kFilterTargetParts + groups - 1) / groups;
    return parts > 8? 8 : parts;
}

uint32_t
sora_filter_max_regions(uint32_t max_groups) {
    /* groups * ceil(target / groups) < target + groups */
    return kFilterTargetParts + max_groups;
}

hipError_t
sora_launch_finalize(const sora_resolve_args *a, uint32_t cus, hipStream_t stream) {
    (void) hipGetLastError();
    hipLaunchKernelGGL(flatten_kernel, dim3(a->regions), dim3(kSmall), 0, stream,
	    *a);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess)
	return e;
    /* SORA_FINALIZE_BLOCKS blocks of 4 waves per CU: every wave slot busy
     * hiding the two dependent fetches (record, then samples) of a frame */
    (void) hipGetLastError();
    const dim3 grid(cus * SORA_FINALIZE_BLOCKS), block(kSmall);
    if (a->encoding == SORA_ENC_SC16 && a->wire)
	hipLaunchKernelGGL((finalize_kernel<true, true>), grid, block, 0, stream, *a);
    else if (a->encoding == SORA_ENC_SC16)
	hipLaunchKernelGGL((finalize_kernel<true, false>), grid, block, 0, stream, *a);
    else if (a->wire)
	hipLaunchKernelGGL((finalize_kernel<false, true>), grid, block, 0, stream, *a);
    else
	hipLaunchKernelGGL((finalize_kernel<false, false>), grid, block, 0, stream, *a);
    return hipGetLastError();
}

hipError_t
sora_launch_magnitude(const uint8_t *iq, uint64_t n_samples, int encoding,
	float *m, hipStream_t stream) {
    if (n_samples == 0)
	return hipSuccess;
    (void) hipGetLastError();
    hipLaunchKernelGGL(magnitude_kernel, dim3(grid_for(n_samples, 256)),
	    dim3(256), 0, stream, iq, n_samples, encoding, m);
    return hipGetLastError();
}

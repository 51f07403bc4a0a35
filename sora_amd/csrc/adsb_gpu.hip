/*
 * adsb_gpu.hip -- the C ABI of include/sora_adsb_gpu.h: contexts, device
 * buffers and the launch sequence around the kernels of adsb_kernels.hip.
 * Host code only; every error becomes an integer code (no exceptions, no
 * abort: SURVEY.md section 8b).
 */
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "../../include/sora_adsb_gpu.h"
#include "adsb_kernels.h"
#include "adsb_tile.h"

#define SORA_ADDRESSES	(1u << 24)

struct sora_adsb_ctx {
    int device;
    hipStream_t stream;
    bool own_stream;
    /* Where everything behind the scan runs -- exchange, resolve, the copies to
     * the host: the same stream unless sora_adsb_set_resolve_stream() named
     * another.  The operations of ONE context stay in the order they were
     * called in (an event each way, recorded only when the other stream has had
     * work since); what the second stream buys is that another context's scan,
     * queued on the scan stream, need not wait for this context's resolve. */
    hipStream_t rstream;
    hipEvent_t ev_to_resolve, ev_to_scan;
    bool resolve_behind;	/* the scan stream has had work the resolve stream has not waited for */
    bool scan_behind;		/* ... and the other way round */
    bool summary_clean;		/* the summary words are zero (a reset has just cleared them) */
    hipEvent_t ev_scan_begin, ev_scan_end;
    bool scan_timed;

    sora_adsb_config cfg;
    uint32_t max_groups;	/* workgroups the scan kernel may use */
    uint32_t scan_variant;	/* 8-bit scan kernel: 1 magnitudes in registers (default), 0 through LDS */
    uint32_t scan_rounds, scan_min_tiles;	/* see sora_adsb_create() */
    uint32_t scan_sums;		/* 0: no scan kernel sums amplitudes (SORA_ADSB_SCAN_SUMS) */
    uint32_t max_regions;	/* accepted regions (sora_filter_max_regions) */
    uint32_t cus;
    uint64_t rec_total, sq_total, acc_total, frames_cap;

    /* device */
    uint64_t *d_rec, *d_sq, *d_acc, *d_acc_flat, *d_first_seen, *d_stats;
    uint32_t *d_known_bits;
    uint32_t *d_sq_bits;	/* sora_scan_args::sq_bits: four words per squitter slot */
    double *d_sq_sum;		/* sora_scan_args::sq_sum */
    uint64_t *d_acc_offset, *d_sq_offset;
    uint32_t *d_rec_count, *d_sq_count, *d_acc_count;
    uint32_t *d_tile_end;	/* sora_scan_args::tile_end: one word per tile of a chunk */
    uint64_t tile_slots;
    sora_adsb_frame *d_frames;
    uint8_t *d_stage;
    size_t stage_bytes;

    /* what a collect reads first (sora_resolve_args::summary) and its pinned
     * mirror: one 32-byte copy tells whether every list held and how many frames
     * there are; the per-region counters below are fetched only when one did not,
     * and for sora_adsb_get_counts */
    uint64_t *d_summary, *h_summary;
    /* epoch of the first-seen table (sora_resolve_args::seen_tag): a reset is a
     * new tag, not 128 MB of stores */
    uint32_t epoch;

    /* host mirrors */
    uint32_t *h_counts;		/* 2 * max_groups + max_regions */
    uint64_t *h_stats;		/* max_groups * SORA_STAT_WORDS */
    uint64_t h_total[2];

    /* the chunk in flight */
    bool scanned, resolved;
    bool merged;		/* its squitters are in first_seen */
    bool exported;		/* d_export holds its first-seen records */
    bool exchanged;		/* its squitter export has left the context (another
				 * shard may have merged it): a squitter-list overflow
				 * can then no longer be repaired by a local re-run */
    /* arguments of the last scan: a list overflow re-runs it with larger lists */
    struct {
	const void *d_iq;
	size_t nbytes;
	int encoding;
	uint64_t base_position, n_owned;
    } last;
    uint32_t regrows;		/* re-runs so far (sora_adsb_regrow_count) */
    size_t last_collected;	/* frames the last collect copied to the host */
    uint64_t *d_export;		/* one record per address first seen in this scan */
    uint64_t export_cap;
    sora_scan_args scan;
    sora_resolve_args res;
    uint32_t groups;

    /* streaming ingest (sora_adsb_stream_*) */
    struct {
	bool active, done;
	int encoding, bps;
	uint64_t block;
	sora_adsb_read_fn read;
	void *user;
	uint8_t *h_in[2], *d_in[2];
	sora_adsb_frame *h_out[2];
	uint64_t out_cap[2];	/* frames h_out[slot] holds */
	hipStream_t copy;
	hipEvent_t ev_in[2];	/* block is on the device */
	hipEvent_t ev_free[2];	/* the kernels are done with d_in[slot] */
	uint64_t n_samples[2], base[2], n_owned[2];
	bool filled[2], last[2];
	uint64_t consumed;	/* samples taken from the source */
	uint64_t carry;		/* samples the next block starts with */
	int cur;
    } st;

    char err[256];
};

namespace {

int
fail(sora_adsb_ctx *ctx, int code, const char *fmt, ...) {
    if (ctx != nullptr) {
	va_list ap;
	va_start(ap, fmt);
	vsnprintf(ctx->err, sizeof ctx->err, fmt, ap);
	va_end(ap);
    }
    return code;
}

void stream_release(sora_adsb_ctx *ctx);

/* Before work is queued on the scan stream / on the resolve stream: whatever the
 * context queued on the other one comes first. */
hipError_t
use_scan_stream(sora_adsb_ctx *ctx) {
    if (ctx->rstream == ctx->stream)
	return hipSuccess;
    if (ctx->scan_behind) {
	hipError_t e = hipEventRecord(ctx->ev_to_scan, ctx->rstream);
	if (e == hipSuccess)
	    e = hipStreamWaitEvent(ctx->stream, ctx->ev_to_scan, 0);
	if (e != hipSuccess)
	    return e;
	ctx->scan_behind = false;
    }
    ctx->resolve_behind = true;
    return hipSuccess;
}

hipError_t
use_resolve_stream(sora_adsb_ctx *ctx) {
    if (ctx->rstream == ctx->stream)
	return hipSuccess;
    if (ctx->resolve_behind) {
	hipError_t e = hipEventRecord(ctx->ev_to_resolve, ctx->stream);
	if (e == hipSuccess)
	    e = hipStreamWaitEvent(ctx->rstream, ctx->ev_to_resolve, 0);
	if (e != hipSuccess)
	    return e;
	ctx->resolve_behind = false;
    }
    ctx->scan_behind = true;
    return hipSuccess;
}

#define HIP_TRY(ctx, call) do { \
	hipError_t e_ = (call); \
	if (e_ != hipSuccess) \
	    return fail((ctx), SORA_ERR_HIP, "%s: %s", #call, hipGetErrorString(e_)); \
    } while (0)

constexpr uint32_t kScanRounds = 32;	/* rounds of scan workgroups per CU */
constexpr uint64_t kSummaryAlloc = 32;	/* words behind d_summary: zeroed 256 bytes at a time */
constexpr uint32_t kMinTiles = 8;	/* tiles per workgroup, at least */
constexpr size_t kAheadBytes = 2u << 20;	/* collect: frame lists up to this are sent ahead of their count */

uint32_t
env_number(const char *name, uint32_t fallback, long lo, long hi) {
    const char *e = getenv(name);
    if (e == nullptr || *e == '\0')
	return fallback;
    char *end = nullptr;
    const long v = strtol(e, &end, 10);
    return *end == '\0' && v >= lo && v <= hi? (uint32_t) v : fallback;
}

int
bytes_per_sample(int encoding) {
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

template <typename T> hipError_t
dev_alloc(T **p, uint64_t n) {
    return hipMalloc(reinterpret_cast<void **>(p), (n > 0? n : 1) * sizeof(T));
}

/*
 * The lists whose sizes depend on the traffic: candidate records (8 bytes per
 * slot for the 5 used: room for the alignment of the tails and for the
 * filter's 16-byte loads past a short region), clean squitters, accepted
 * records (per region and flat) and frames.  Allocated by sora_adsb_create()
 * from the configuration and again, larger, when a scan overflowed them.  The
 * scan's lists (records, squitters) and the resolve's (accepted, flat, frames)
 * are two sets: when only the resolve's were too small the scan's keep their
 * contents and only the resolve runs again.
 */
void
free_resolve_lists(sora_adsb_ctx *ctx) {
    (void) hipFree(ctx->d_acc);
    (void) hipFree(ctx->d_acc_flat);
    (void) hipFree(ctx->d_frames);
    ctx->d_acc = ctx->d_acc_flat = nullptr;
    ctx->d_frames = nullptr;
}

void
free_lists(sora_adsb_ctx *ctx) {
    (void) hipFree(ctx->d_rec);
    (void) hipFree(ctx->d_sq);
    (void) hipFree(ctx->d_sq_bits);
    (void) hipFree(ctx->d_sq_sum);
    ctx->d_rec = ctx->d_sq = nullptr;
    ctx->d_sq_bits = nullptr;
    ctx->d_sq_sum = nullptr;
    free_resolve_lists(ctx);
}

/* shares the lists out over the workgroup regions of the scan at hand
 * (ctx->groups, ctx->res.regions): pointers and per-region capacities */
void
bind_lists(sora_adsb_ctx *ctx) {
    sora_scan_args &a = ctx->scan;
    sora_resolve_args &r = ctx->res;
    const uint32_t groups = ctx->groups;
    const auto share = [](uint64_t total, uint32_t parts) {
	const uint64_t each = total / (parts > 0? parts : 1);
	return (uint32_t) (each > 0x7fffffffull? 0x7fffffffull : each);
    };

    /* the record buffer: units of SORA_REC_UNIT records, keys then tails */
    a.rec = reinterpret_cast<uint8_t *>(ctx->d_rec);
    a.tile_end = ctx->d_tile_end;
    a.sq = ctx->d_sq;
    a.sq_bits = ctx->d_sq_bits;
    /* the sc16 scan kernel sums its squitters' amplitudes itself */
    a.sq_sum = a.encoding == SORA_ENC_SC16 && ctx->scan_sums != 0? ctx->d_sq_sum : nullptr;
    a.rec_cap = share(ctx->rec_total, groups) & ~(SORA_REC_UNIT - 1);	/* >= a unit: sora_adsb_scan */
    a.sq_cap = share(ctx->sq_total, groups);
    r.rec = a.rec;
    r.tile_end = a.tile_end;
    r.rec_cap = a.rec_cap;
    r.sq = ctx->d_sq;
    r.sq_bits = ctx->d_sq_bits;
    r.sq_sum = a.sq_sum;
    r.sq_cap = a.sq_cap;
    r.acc = ctx->d_acc;
    r.acc_cap = share(ctx->acc_total, r.regions);
    r.acc_flat = ctx->d_acc_flat;
    r.frames = ctx->d_frames;
    r.frames_cap = ctx->frames_cap;
}

/* -DSORA_POISON_LISTS (test builds): a list fresh from hipMalloc holds a
 * pattern no kernel writes, so that a kernel that reads what nothing wrote
 * shows in the frames instead of passing on what the allocator left there */
template <typename T> hipError_t
list_alloc(T **p, uint64_t n) {
    hipError_t e = dev_alloc(p, n);
#ifdef SORA_POISON_LISTS
    if (e == hipSuccess)
	e = hipMemset(*p, 0xa5, (n > 0? n : 1) * sizeof(T));
#endif
    return e;
}

hipError_t
alloc_resolve_lists(sora_adsb_ctx *ctx) {
    hipError_t e;
    if ((e = list_alloc(&ctx->d_acc, ctx->acc_total)) != hipSuccess ||
	    (e = list_alloc(&ctx->d_acc_flat, ctx->frames_cap)) != hipSuccess ||
	    (e = list_alloc(&ctx->d_frames, ctx->frames_cap)) != hipSuccess)
	return e;
    return hipSuccess;
}

hipError_t
alloc_lists(sora_adsb_ctx *ctx) {
    hipError_t e;
    if ((e = list_alloc(&ctx->d_rec, ctx->rec_total + 8)) != hipSuccess ||
	    (e = list_alloc(&ctx->d_sq, ctx->sq_total)) != hipSuccess ||
	    (e = list_alloc(&ctx->d_sq_bits, 4 * ctx->sq_total)) != hipSuccess ||
	    (e = list_alloc(&ctx->d_sq_sum, ctx->sq_total)) != hipSuccess)
	return e;
    return alloc_resolve_lists(ctx);
}

} /* namespace */

extern "C" const char *
sora_adsb_version(void) {
    return "sora_adsb_gpu 0.1 (gfx950)";
}

extern "C" const char *
sora_adsb_strerror(int code) {
    switch (code) {
    case SORA_OK: return "ok";
    case SORA_ERR_INVALID: return "invalid argument";
    case SORA_ERR_NO_DEVICE: return "no usable HIP device";
    case SORA_ERR_HIP: return "HIP call failed";
    case SORA_ERR_OVERFLOW: return "device list overflow";
    case SORA_ERR_NOMEM: return "out of memory";
    case SORA_ERR_ALIGN: return "device pointer not 16-byte aligned";
    case SORA_ERR_STATE: return "call out of order";
    default: return "unknown error";
    }
}

extern "C" int
sora_adsb_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess)
	return 0;
    return n;
}

extern "C" const char *
sora_adsb_last_error(const sora_adsb_ctx *ctx) {
    return ctx != nullptr? ctx->err : "";
}

extern "C" void
sora_adsb_destroy(sora_adsb_ctx *ctx) {
    if (ctx == nullptr)
	return;
    (void) hipSetDevice(ctx->device);
    if (ctx->stream != nullptr)
	(void) hipStreamSynchronize(ctx->stream);
    if (ctx->rstream != nullptr && ctx->rstream != ctx->stream)
	(void) hipStreamSynchronize(ctx->rstream);
    if (ctx->st.active) {
	(void) hipStreamSynchronize(ctx->st.copy);
	stream_release(ctx);
    }
    free_lists(ctx);
    (void) hipFree(ctx->d_known_bits);
    (void) hipFree(ctx->d_first_seen);
    (void) hipFree(ctx->d_stats);
    (void) hipFree(ctx->d_acc_offset);
    (void) hipFree(ctx->d_sq_offset);
    (void) hipFree(ctx->d_export);
    (void) hipFree(ctx->d_rec_count);
    (void) hipFree(ctx->d_sq_count);
    (void) hipFree(ctx->d_acc_count);
    (void) hipFree(ctx->d_tile_end);
    (void) hipFree(ctx->d_stage);
    if (ctx->h_summary != nullptr)
	(void) hipHostFree(ctx->h_summary);
    if (ctx->h_counts != nullptr)
	(void) hipHostFree(ctx->h_counts);
    if (ctx->h_stats != nullptr)
	(void) hipHostFree(ctx->h_stats);
    if (ctx->ev_scan_begin != nullptr)
	(void) hipEventDestroy(ctx->ev_scan_begin);
    if (ctx->ev_scan_end != nullptr)
	(void) hipEventDestroy(ctx->ev_scan_end);
    if (ctx->ev_to_resolve != nullptr)
	(void) hipEventDestroy(ctx->ev_to_resolve);
    if (ctx->ev_to_scan != nullptr)
	(void) hipEventDestroy(ctx->ev_to_scan);
    if (ctx->own_stream && ctx->stream != nullptr)
	(void) hipStreamDestroy(ctx->stream);
    free(ctx);
}

namespace {

constexpr uint32_t kLastEpoch = 0xfffffeu;

uint64_t
seen_tag(const sora_adsb_ctx *ctx) {
    return (uint64_t) (0xffffffu - ctx->epoch) << 40;
}

} /* namespace */

extern "C" int
sora_adsb_reset(sora_adsb_ctx *ctx) {
    if (ctx == nullptr)
	return SORA_ERR_INVALID;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, use_scan_stream(ctx));
    /* forgetting every address = the next epoch: what the table holds is then
     * larger than anything stored or asked from now on (sora_resolve_args::
     * seen_tag).  Only the first use and a wrap of the 24-bit epoch fill it. */
    if (ctx->epoch == 0 || ctx->epoch >= kLastEpoch) {
	HIP_TRY(ctx, sora_launch_fill_u64(ctx->d_first_seen, ~0ull, SORA_ADDRESSES,
		ctx->stream));
	ctx->epoch = 1;
    } else {
	ctx->epoch++;
    }
    /* the summary words lie behind the bitmap: one fill clears both (a new
     * capture's reset + scan is then one memset, not two) */
    HIP_TRY(ctx, hipMemsetAsync(ctx->d_known_bits, 0, SORA_KNOWN_WORDS * sizeof(uint32_t) +
	    kSummaryAlloc * sizeof(uint64_t), ctx->stream));
    ctx->summary_clean = true;
    ctx->scanned = ctx->resolved = false;
    return SORA_OK;
}

extern "C" int
sora_adsb_create(sora_adsb_ctx **out, int device, const sora_adsb_config *cfg) {
    int ndev = 0;
    hipDeviceProp_t prop;

    if (out == nullptr || cfg == nullptr || cfg->max_chunk_samples == 0)
	return SORA_ERR_INVALID;
    *out = nullptr;
    if (cfg->emit_mode != SORA_EMIT_ACCEPTED && cfg->emit_mode != SORA_EMIT_CANDIDATES)
	return SORA_ERR_INVALID;
    if (cfg->max_chunk_samples >= (1ull << SORA_REC_POS_BITS))
	return SORA_ERR_INVALID;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
	return SORA_ERR_NO_DEVICE;
    if (device < 0 || device >= ndev)
	return SORA_ERR_NO_DEVICE;

    sora_adsb_ctx *ctx = static_cast<sora_adsb_ctx *>(calloc(1, sizeof *ctx));
    if (ctx == nullptr)
	return SORA_ERR_NOMEM;
    ctx->device = device;
    ctx->cfg = *cfg;
    *out = ctx;			/* so the caller can read last_error and destroy */

    HIP_TRY(ctx, hipSetDevice(device));
    HIP_TRY(ctx, hipGetDeviceProperties(&prop, device));
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
	return fail(ctx, SORA_ERR_NO_DEVICE,
		"device %d is %s: this library holds gfx950 code only", device,
		prop.gcnArchName);

    if (cfg->stream != nullptr) {
	ctx->stream = static_cast<hipStream_t>(cfg->stream);
    } else {
	HIP_TRY(ctx, hipStreamCreateWithFlags(&ctx->stream, hipStreamDefault));
	ctx->own_stream = true;
    }
    ctx->rstream = ctx->stream;
    HIP_TRY(ctx, hipEventCreate(&ctx->ev_scan_begin));
    HIP_TRY(ctx, hipEventCreate(&ctx->ev_scan_end));
    HIP_TRY(ctx, hipEventCreateWithFlags(&ctx->ev_to_resolve, hipEventDisableTiming));
    HIP_TRY(ctx, hipEventCreateWithFlags(&ctx->ev_to_scan, hipEventDisableTiming));

    /* the scan kernels run sora_scan_groups_per_cu() workgroups on every CU
     * (what their LDS lets a CU hold): at most 4 */
    ctx->cus = prop.multiProcessorCount > 0? (uint32_t) prop.multiProcessorCount : 1;
    /* ... in kScanRounds rounds: a workgroup takes a short range of tiles, so
     * CUs that run ahead (clock, HBM channel luck) take more ranges instead of
     * waiting for the slowest: measured 3.67 -> 3.39 ms on the 1-hour capture
     * from 1 round to 16 in round 1, and with the LDS record ring of round 2
     * 3.80 / 3.73 / 3.68 / 3.66 / 3.66 ms per step at 6 / 12 / 16 / 32 / 48
     * rounds.  A range is at least kMinTiles tiles (the CRC tables are loaded
     * into LDS once per range).  SORA_ADSB_SCAN_ROUNDS / SORA_ADSB_SCAN_MIN_TILES
     * override both (tuning and tests). */
    ctx->scan_variant = env_number("SORA_ADSB_SCAN_VARIANT", 1, 0, 1);
    ctx->scan_rounds = env_number("SORA_ADSB_SCAN_ROUNDS", kScanRounds, 1, 64);
    ctx->scan_min_tiles = env_number("SORA_ADSB_SCAN_MIN_TILES", kMinTiles, 1, 1 << 20);
    ctx->scan_sums = env_number("SORA_ADSB_SCAN_SUMS", 1, 0, 1);
    ctx->max_groups = ctx->cus * 8 * ctx->scan_rounds;
    ctx->max_regions = sora_filter_max_regions(ctx->max_groups);

    const uint64_t n = cfg->max_chunk_samples;
    ctx->rec_total = cfg->max_candidates != 0? cfg->max_candidates : n / 16 + 65536;
    ctx->rec_total = (ctx->rec_total + SORA_REC_UNIT - 1) & ~(uint64_t) (SORA_REC_UNIT - 1);	/* whole units */
    ctx->sq_total = cfg->max_squitters != 0? cfg->max_squitters : n / 256 + 65536;
    ctx->acc_total = cfg->max_frames != 0? cfg->max_frames :
	    cfg->emit_mode == SORA_EMIT_CANDIDATES? ctx->rec_total : ctx->sq_total;
    ctx->frames_cap = ctx->acc_total;
    /* at most one export record per address, and never more than squitters */
    ctx->export_cap = ctx->sq_total < (1ull << 22)? ctx->sq_total : (1ull << 22);

    const uint32_t g = ctx->max_groups;
    const uint32_t nr = ctx->max_regions;
    /* no scan kernel's tile has fewer than 7808 positions */
    ctx->tile_slots = n / 7808 + 16;
    if (alloc_lists(ctx) != hipSuccess ||
	    dev_alloc(&ctx->d_known_bits, SORA_KNOWN_WORDS + 2 * kSummaryAlloc) != hipSuccess ||
	    dev_alloc(&ctx->d_first_seen, SORA_ADDRESSES) != hipSuccess ||
	    dev_alloc(&ctx->d_stats, (uint64_t) g * SORA_STAT_WORDS) != hipSuccess ||
	    dev_alloc(&ctx->d_acc_offset, nr + 1) != hipSuccess ||
	    dev_alloc(&ctx->d_sq_offset, g + 1) != hipSuccess ||
	    dev_alloc(&ctx->d_export, ctx->export_cap) != hipSuccess ||
	    dev_alloc(&ctx->d_rec_count, g) != hipSuccess ||
	    dev_alloc(&ctx->d_sq_count, g) != hipSuccess ||
	    dev_alloc(&ctx->d_acc_count, nr) != hipSuccess ||
	    dev_alloc(&ctx->d_tile_end, ctx->tile_slots) != hipSuccess)
	return fail(ctx, SORA_ERR_NOMEM, "hipMalloc of the device lists failed");
    /* SORA_KNOWN_WORDS * 4 bytes is a multiple of 256: the words are aligned */
    ctx->d_summary = reinterpret_cast<uint64_t *>(ctx->d_known_bits + SORA_KNOWN_WORDS);
    HIP_TRY(ctx, hipHostMalloc(reinterpret_cast<void **>(&ctx->h_summary),
	    SORA_SUM_WORDS * sizeof(uint64_t), hipHostMallocDefault));
    HIP_TRY(ctx, hipHostMalloc(reinterpret_cast<void **>(&ctx->h_counts),
	    (2 * (size_t) g + nr) * sizeof(uint32_t), hipHostMallocDefault));
    HIP_TRY(ctx, hipHostMalloc(reinterpret_cast<void **>(&ctx->h_stats),
	    (size_t) g * SORA_STAT_WORDS * sizeof(uint64_t), hipHostMallocDefault));

    int rc = sora_adsb_reset(ctx);
    if (rc != SORA_OK)
	return rc;
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return SORA_OK;
}

extern "C" int
sora_adsb_scan(sora_adsb_ctx *ctx, const void *d_iq, size_t nbytes, int encoding,
	uint64_t base_position, uint64_t n_owned) {
    if (ctx == nullptr)
	return SORA_ERR_INVALID;
    const int bps = bytes_per_sample(encoding);
    if (bps == 0)
	return fail(ctx, SORA_ERR_INVALID, "unknown encoding %d", encoding);
    if (d_iq == nullptr && nbytes != 0)
	return fail(ctx, SORA_ERR_INVALID, "null IQ pointer");
    if ((reinterpret_cast<uintptr_t>(d_iq) & 15) != 0)
	return fail(ctx, SORA_ERR_ALIGN, "IQ pointer %p is not 16-byte aligned", d_iq);
    const uint64_t n_samples = nbytes / (size_t) bps;
    if (n_samples > ctx->cfg.max_chunk_samples)
	return fail(ctx, SORA_ERR_INVALID,
		"chunk of %llu samples exceeds max_chunk_samples %llu",
		(unsigned long long) n_samples,
		(unsigned long long) ctx->cfg.max_chunk_samples);
    if (base_position + n_samples >= (1ull << 40))
	return fail(ctx, SORA_ERR_INVALID, "positions beyond 2^40 are not supported");

    HIP_TRY(ctx, hipSetDevice(ctx->device));

    /* positions p < n_owned with p + 240 <= n_samples (plane-iq.c:148-166) */
    uint64_t n_scan = n_samples >= SORA_ADSB_WINDOW? n_samples - SORA_ADSB_HALO : 0;
    if (n_owned < n_scan)
	n_scan = n_owned;
    const uint64_t n_tiles = sora_scan_tiles(n_scan, encoding, (int) ctx->scan_variant);
    uint32_t groups = ctx->cus * sora_scan_groups_per_cu(encoding, (int) ctx->scan_variant) * ctx->scan_rounds;
    if (n_tiles / ctx->scan_min_tiles < groups)
	groups = n_tiles >= ctx->scan_min_tiles? (uint32_t) (n_tiles / ctx->scan_min_tiles) : 1;

    if (n_tiles > ctx->tile_slots)
	return fail(ctx, SORA_ERR_INVALID, "%llu tiles for %llu slots", (unsigned long long) n_tiles,
		(unsigned long long) ctx->tile_slots);

    /* a region is whole units of SORA_REC_UNIT records (sora_scan_args::rec) and
     * holds at least one (few, long regions when the record list is tiny) */
    if (ctx->rec_total / groups < SORA_REC_UNIT)
	groups = ctx->rec_total / SORA_REC_UNIT > 0? (uint32_t) (ctx->rec_total / SORA_REC_UNIT) : 1;

    /* the tiles of a workgroup's range have a word each in the LDS of the scan
     * kernel and of the filter: SORA_RANGE_TILES_MAX (adsb_kernels.h) */
    if ((n_tiles / groups + 1) * sora_scan_tile_pos(encoding, (int) ctx->scan_variant) >=
	    (1ull << SORA_REGION_POS_BITS))
	return fail(ctx, SORA_ERR_INVALID, "chunk of %llu samples is too long for %u scan regions",
		(unsigned long long) n_samples, groups);

    sora_scan_args &a = ctx->scan;
    memset(&a, 0, sizeof a);
    a.iq = static_cast<const uint8_t *>(d_iq);
    a.n_samples = n_samples;
    a.n_scan = n_scan;
    a.n_tiles = n_tiles;
    a.encoding = encoding;
    a.emit_all = ctx->cfg.emit_mode == SORA_EMIT_CANDIDATES;
    a.correct = (ctx->cfg.options & SORA_OPT_CORRECT_SINGLE_BIT) != 0;
    a.rec_count = ctx->d_rec_count;
    a.sq_count = ctx->d_sq_count;
    a.stats = ctx->d_stats;
    const bool merges = sora_scan_merges(encoding, (int) ctx->scan_variant) != 0;
    if (merges) {
	a.first_seen = ctx->d_first_seen;
	a.known_bits = ctx->d_known_bits;
	a.base_position = base_position;
	a.seen_tag = seen_tag(ctx);
    }
    a.summary = ctx->d_summary;

    sora_resolve_args &r = ctx->res;
    memset(&r, 0, sizeof r);
    r.iq = a.iq;
    r.n_samples = n_samples;
    r.base_position = base_position;
    r.encoding = encoding;
    r.emit_all = a.emit_all;
    r.groups = groups;
    r.n_tiles = n_tiles;
    r.tile_pos = sora_scan_tile_pos(encoding, (int) ctx->scan_variant);
    r.rec_count = ctx->d_rec_count;
    r.sq_count = ctx->d_sq_count;
    r.first_seen = ctx->d_first_seen;
    r.known_bits = ctx->d_known_bits;
    r.seen_tag = seen_tag(ctx);
    r.summary = ctx->d_summary;
    r.filter_parts = sora_filter_parts(groups);
    r.regions = groups * r.filter_parts;
    r.acc_count = ctx->d_acc_count;
    r.acc_offset = ctx->d_acc_offset;
    r.wire = (ctx->cfg.options & SORA_OPT_WIRE_FRAMES) != 0;

    ctx->groups = groups;
    bind_lists(ctx);
    ctx->scanned = false;
    ctx->resolved = false;
    ctx->merged = merges;		/* else first_seen_kernel, before the first use */
    ctx->exported = false;
    ctx->exchanged = false;
    ctx->last.d_iq = d_iq;
    ctx->last.nbytes = nbytes;
    ctx->last.encoding = encoding;
    ctx->last.base_position = base_position;
    ctx->last.n_owned = n_owned;

    HIP_TRY(ctx, use_scan_stream(ctx));
    if (!ctx->summary_clean)
	HIP_TRY(ctx, hipMemsetAsync(ctx->d_summary, 0, kSummaryAlloc * sizeof(uint64_t),
		ctx->stream));
    ctx->summary_clean = false;
    HIP_TRY(ctx, hipEventRecord(ctx->ev_scan_begin, ctx->stream));
    HIP_TRY(ctx, sora_launch_scan(&a, groups, (int) ctx->scan_variant, ctx->stream));
    HIP_TRY(ctx, hipEventRecord(ctx->ev_scan_end, ctx->stream));
    ctx->scan_timed = true;
    ctx->scanned = true;
    return SORA_OK;
}

namespace {

/* first-seen merge of this scan's squitters + the export list (once per scan) */
int
export_first(sora_adsb_ctx *ctx) {
    if (ctx->exported)
	return SORA_OK;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, use_resolve_stream(ctx));
    if (!ctx->merged) {
	HIP_TRY(ctx, sora_launch_first_seen(&ctx->res, ctx->rstream));
	ctx->merged = true;
    }
    HIP_TRY(ctx, sora_launch_export_first(&ctx->res, ctx->d_export, ctx->export_cap,
	    ctx->d_sq_offset, ctx->rstream));
    HIP_TRY(ctx, hipMemcpyAsync(&ctx->h_total[0], ctx->d_sq_offset, sizeof(uint64_t),
	    hipMemcpyDeviceToHost, ctx->rstream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->rstream));
    if (ctx->h_total[0] > ctx->export_cap)
	return fail(ctx, SORA_ERR_OVERFLOW, "%llu addresses first seen in one scan "
		"exceed the export list (%llu)", (unsigned long long) ctx->h_total[0],
		(unsigned long long) ctx->export_cap);
    ctx->exported = true;
    return SORA_OK;
}

} /* namespace */

extern "C" int
sora_adsb_squitter_count(sora_adsb_ctx *ctx, uint64_t *n) {
    if (ctx == nullptr || n == nullptr)
	return SORA_ERR_INVALID;
    if (!ctx->scanned)
	return fail(ctx, SORA_ERR_STATE, "no scan to count squitters of");
    int rc = export_first(ctx);
    if (rc != SORA_OK)
	return rc;
    *n = ctx->h_total[0];
    return SORA_OK;
}

extern "C" int
sora_adsb_copy_squitters(sora_adsb_ctx *ctx, void *d_dst, size_t cap_records) {
    if (ctx == nullptr || (d_dst == nullptr && cap_records != 0))
	return SORA_ERR_INVALID;
    if (!ctx->scanned)
	return fail(ctx, SORA_ERR_STATE, "no scan to export squitters of");
    int rc = export_first(ctx);
    if (rc != SORA_OK)
	return rc;
    const size_t n = ctx->h_total[0] < cap_records? (size_t) ctx->h_total[0] : cap_records;
    ctx->exchanged = true;
    if (n > 0)
	HIP_TRY(ctx, hipMemcpyAsync(d_dst, ctx->d_export, n * sizeof(uint64_t),
		hipMemcpyDeviceToDevice, ctx->rstream));
    return SORA_OK;
}

extern "C" int
sora_adsb_add_known(sora_adsb_ctx *ctx, const void *d_records, size_t n_records) {
    if (ctx == nullptr || (d_records == nullptr && n_records != 0))
	return SORA_ERR_INVALID;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, use_resolve_stream(ctx));
    HIP_TRY(ctx, sora_launch_add_known(ctx->d_first_seen, ctx->d_known_bits, seen_tag(ctx),
	    static_cast<const uint64_t *>(d_records), n_records, ctx->rstream));
    return SORA_OK;
}

extern "C" int
sora_adsb_export_block(sora_adsb_ctx *ctx, void *d_block, size_t block_records) {
    if (ctx == nullptr || d_block == nullptr || block_records < 2 ||
	    block_records > 0xffffffffull)
	return SORA_ERR_INVALID;
    if (!ctx->scanned)
	return fail(ctx, SORA_ERR_STATE, "no scan to export squitters of");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, use_resolve_stream(ctx));
    if (!ctx->merged) {
	HIP_TRY(ctx, sora_launch_first_seen(&ctx->res, ctx->rstream));
	ctx->merged = true;
    }
    uint64_t *blk = static_cast<uint64_t *>(d_block);
    HIP_TRY(ctx, sora_launch_export_first(&ctx->res, blk + 1, block_records - 1, blk,
	    ctx->rstream));
    ctx->exchanged = true;
    return SORA_OK;
}

extern "C" int
sora_adsb_add_known_blocks(sora_adsb_ctx *ctx, const void *d_blocks, size_t n_blocks,
	size_t block_records, int skip_block) {
    if (ctx == nullptr || (d_blocks == nullptr && n_blocks != 0) || block_records < 2 ||
	    block_records > 0xffffffffull || n_blocks > 65535)
	return SORA_ERR_INVALID;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, use_resolve_stream(ctx));
    HIP_TRY(ctx, sora_launch_add_known_blocks(ctx->d_first_seen, ctx->d_known_bits,
	    seen_tag(ctx), static_cast<const uint64_t *>(d_blocks), (uint32_t) n_blocks,
	    (uint32_t) block_records, skip_block < 0? 0xffffffffu : (uint32_t) skip_block,
	    ctx->rstream));
    return SORA_OK;
}

extern "C" int
sora_adsb_resolve(sora_adsb_ctx *ctx) {
    if (ctx == nullptr)
	return SORA_ERR_INVALID;
    if (!ctx->scanned)
	return fail(ctx, SORA_ERR_STATE, "resolve without a scan");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, use_resolve_stream(ctx));
    if (!ctx->merged) {
	HIP_TRY(ctx, sora_launch_first_seen(&ctx->res, ctx->rstream));
	ctx->merged = true;
    }
    HIP_TRY(ctx, sora_launch_filter(&ctx->res, ctx->rstream));
    HIP_TRY(ctx, sora_launch_scan_counts(ctx->d_acc_count, ctx->res.acc_cap,
	    ctx->d_acc_offset, ctx->res.regions, ctx->d_summary + SORA_SUM_FRAMES, ctx->rstream));
    HIP_TRY(ctx, sora_launch_finalize(&ctx->res, ctx->cus, ctx->rstream));
    ctx->resolved = true;
    return SORA_OK;
}

extern "C" int
sora_adsb_submit_device(sora_adsb_ctx *ctx, const void *d_iq, size_t nbytes,
	int encoding, uint64_t base_position, uint64_t n_owned) {
    int rc = sora_adsb_scan(ctx, d_iq, nbytes, encoding, base_position, n_owned);
    if (rc != SORA_OK)
	return rc;
    return sora_adsb_resolve(ctx);
}

extern "C" int
sora_adsb_submit_host(sora_adsb_ctx *ctx, const void *h_iq, size_t nbytes,
	int encoding, uint64_t base_position, uint64_t n_owned) {
    if (ctx == nullptr || (h_iq == nullptr && nbytes != 0))
	return SORA_ERR_INVALID;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    /* the last resolve may still read the staging buffer */
    HIP_TRY(ctx, use_scan_stream(ctx));
    const size_t need = ((nbytes + 15) & ~(size_t) 15) + 16;
    if (ctx->stage_bytes < need) {
	HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
	(void) hipFree(ctx->d_stage);
	ctx->d_stage = nullptr;
	ctx->stage_bytes = 0;
	if (hipMalloc(reinterpret_cast<void **>(&ctx->d_stage), need) != hipSuccess)
	    return fail(ctx, SORA_ERR_NOMEM, "staging buffer of %zu bytes", need);
	ctx->stage_bytes = need;
    }
    if (nbytes > 0)
	HIP_TRY(ctx, hipMemcpyAsync(ctx->d_stage, h_iq, nbytes,
		hipMemcpyHostToDevice, ctx->stream));
    return sora_adsb_submit_device(ctx, ctx->d_stage, nbytes, encoding,
	    base_position, n_owned);
}

namespace {

/* copies the per-group counters to the host and checks them; syncs */
int
fetch_counts(sora_adsb_ctx *ctx, bool *overflow) {
    const uint32_t g = ctx->groups;

    HIP_TRY(ctx, use_resolve_stream(ctx));
    HIP_TRY(ctx, hipMemcpyAsync(ctx->h_counts, ctx->d_rec_count,
	    g * sizeof(uint32_t), hipMemcpyDeviceToHost, ctx->rstream));
    HIP_TRY(ctx, hipMemcpyAsync(ctx->h_counts + g, ctx->d_sq_count,
	    g * sizeof(uint32_t), hipMemcpyDeviceToHost, ctx->rstream));
    const uint32_t nr = ctx->res.regions;
    if (ctx->resolved) {
	HIP_TRY(ctx, hipMemcpyAsync(ctx->h_counts + 2 * g, ctx->d_acc_count,
		nr * sizeof(uint32_t), hipMemcpyDeviceToHost, ctx->rstream));
	HIP_TRY(ctx, hipMemcpyAsync(&ctx->h_total[1], ctx->d_acc_offset + nr,
		sizeof(uint64_t), hipMemcpyDeviceToHost, ctx->rstream));
    }
    HIP_TRY(ctx, hipMemcpyAsync(ctx->h_stats, ctx->d_stats,
	    (size_t) g * SORA_STAT_WORDS * sizeof(uint64_t),
	    hipMemcpyDeviceToHost, ctx->rstream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->rstream));

    *overflow = false;
    for (uint32_t i = 0; i < g; i++) {
	if (ctx->h_counts[i] > ctx->res.rec_cap ||
		ctx->h_counts[g + i] > ctx->res.sq_cap)
	    *overflow = true;
    }
    for (uint32_t i = 0; ctx->resolved && i < nr; i++) {
	if (ctx->h_counts[2 * g + i] > ctx->res.acc_cap)
	    *overflow = true;
    }
    if (ctx->resolved && ctx->h_total[1] > ctx->frames_cap)
	*overflow = true;
    return SORA_OK;
}

/*
 * Counts on the host and, if a list overflowed, the chunk once more with
 * lists of the sizes the counts ask for: the kernels count every record they
 * would have written, so one re-run of the scan (records, squitters) and then
 * one of the resolve (accepted records, frames) is enough.  The reference never
 * drops a candidate (plane-iq.c:141-194); neither does a burst of candidates
 * here.  Not repaired, *overflow stays set: SORA_OPT_FIXED_LISTS; a squitter
 * list that overflowed after its (then truncated) export went to other shards.
 * Needs the IQ buffer of the scan still in place (it is until the collect:
 * the resolve reads it too).
 */
int
ensure_complete(sora_adsb_ctx *ctx, bool *overflow) {
    for (int attempt = 0; ; attempt++) {
	int rc = fetch_counts(ctx, overflow);
	if (rc != SORA_OK || !*overflow)
	    return rc;

	const uint32_t g = ctx->groups, nr = ctx->res.regions;
	uint64_t need_rec = 0, need_sq = 0, need_acc = 0, need_frames = 0;
	for (uint32_t i = 0; i < g; i++) {
	    if (ctx->h_counts[i] > need_rec)
		need_rec = ctx->h_counts[i];
	    if (ctx->h_counts[g + i] > need_sq)
		need_sq = ctx->h_counts[g + i];
	}
	for (uint32_t i = 0; ctx->resolved && i < nr; i++) {
	    if (ctx->h_counts[2 * g + i] > need_acc)
		need_acc = ctx->h_counts[2 * g + i];
	    need_frames += ctx->h_counts[2 * g + i];
	}
	/* the three ways out with *overflow still set, each with what the caller
	 * can do about it (sora_adsb_last_error) */
	const auto give_up = [&](const char *why) {
	    (void) fail(ctx, SORA_ERR_OVERFLOW, "%s: the fullest of %u scan regions needs %llu "
		    "record and %llu squitter slots (it has %u and %u), the fullest of %u "
		    "accepted regions %llu (it has %u), %llu frames in all (room for %llu)", why,
		    g, (unsigned long long) need_rec, (unsigned long long) need_sq,
		    ctx->res.rec_cap, ctx->res.sq_cap, nr, (unsigned long long) need_acc,
		    ctx->res.acc_cap, (unsigned long long) need_frames,
		    (unsigned long long) ctx->frames_cap);
	    return SORA_OK;
	};
	if ((ctx->cfg.options & SORA_OPT_FIXED_LISTS) != 0)
	    return give_up("a device list overflowed and SORA_OPT_FIXED_LISTS keeps them as "
		    "they are; raise max_candidates / max_squitters / max_frames");
	if (attempt >= 5)
	    return give_up("a device list still overflowed after five re-runs with larger lists");
	const bool scan_over = need_rec > ctx->res.rec_cap || need_sq > ctx->res.sq_cap;
	if (need_sq > ctx->res.sq_cap && ctx->exchanged)
	    return give_up("the squitter list overflowed and its truncated export has already "
		    "left this context: raise max_squitters and run every shard again");

	/* a quarter more than was needed, shared out evenly like the lists are */
	const auto grown = [](uint64_t per_region, uint32_t regions) {
	    return ((per_region + per_region / 4 + 2 * SORA_REC_UNIT - 1) & ~(uint64_t) (SORA_REC_UNIT - 1)) * regions;
	};
	uint64_t rec_total = ctx->rec_total, sq_total = ctx->sq_total;
	uint64_t acc_total = ctx->acc_total, frames_cap = ctx->frames_cap;
	if (need_rec > ctx->res.rec_cap)
	    rec_total = grown(need_rec, g);
	if (need_sq > ctx->res.sq_cap)
	    sq_total = grown(need_sq, g);
	if (need_acc > ctx->res.acc_cap)
	    acc_total = grown(need_acc, nr);
	if (need_frames > frames_cap)
	    frames_cap = need_frames + need_frames / 4 + 64;
	if (scan_over && !ctx->cfg.emit_mode && acc_total < sq_total)
	    acc_total = sq_total;	/* accepted frames are at least the squitters */
	if (scan_over && ctx->cfg.emit_mode == SORA_EMIT_CANDIDATES && acc_total < rec_total)
	    acc_total = rec_total;
	if (frames_cap < acc_total && scan_over)
	    frames_cap = acc_total;

	/* Every kernel that uses the old lists is done: fetch_counts waited.  A
	 * larger export list first, while nothing has been freed: a failure there
	 * leaves the context as it was. */
	const bool was_resolved = ctx->resolved;
	const uint64_t export_cap = sq_total < (1ull << 22)? sq_total : (1ull << 22);
	if (export_cap > ctx->export_cap) {
	    uint64_t *d_export = nullptr;
	    if (dev_alloc(&d_export, export_cap) != hipSuccess)
		return fail(ctx, SORA_ERR_NOMEM, "export list of %llu records",
			(unsigned long long) export_cap);
	    (void) hipFree(ctx->d_export);
	    ctx->d_export = d_export;
	    ctx->export_cap = export_cap;
	    ctx->exported = false;
	}
	/* The scan's lists (records, squitters) are replaced only when one of THEM
	 * was too small, and then the scan runs again.  When only the accepted or
	 * the frame list was, the records stay where they are -- the filter is
	 * about to read them again -- and only the resolve's lists are replaced. */
	hipError_t e;
	if (scan_over) {
	    free_lists(ctx);
	    ctx->rec_total = rec_total;
	    ctx->sq_total = sq_total;
	    ctx->acc_total = acc_total;
	    ctx->frames_cap = frames_cap;
	    e = alloc_lists(ctx);
	} else {
	    free_resolve_lists(ctx);
	    ctx->acc_total = acc_total;
	    ctx->frames_cap = frames_cap;
	    e = alloc_resolve_lists(ctx);
	}
	if (e != hipSuccess) {
	    ctx->scanned = ctx->resolved = false;
	    return fail(ctx, SORA_ERR_NOMEM, "lists of %llu records, %llu squitters, %llu "
		    "frames after an overflow", (unsigned long long) rec_total,
		    (unsigned long long) sq_total, (unsigned long long) frames_cap);
	}
	ctx->regrows++;
	if (scan_over) {
	    const bool exchanged = ctx->exchanged;
	    rc = sora_adsb_scan(ctx, ctx->last.d_iq, ctx->last.nbytes, ctx->last.encoding,
		    ctx->last.base_position, ctx->last.n_owned);
	    ctx->exchanged = exchanged;
	    if (rc != SORA_OK)
		return rc;
	} else {
	    bind_lists(ctx);
	    HIP_TRY(ctx, hipMemsetAsync(ctx->d_summary + SORA_SUM_ACC_OVER, 0, sizeof(uint64_t),
		    ctx->rstream));
	}
	if (was_resolved && (rc = sora_adsb_resolve(ctx)) != SORA_OK)
	    return rc;
    }
}

int
collect_common(sora_adsb_ctx *ctx, void *dst, size_t cap, size_t *n,
	hipMemcpyKind kind, bool wire) {
    bool overflow = false;

    if (ctx == nullptr || n == nullptr || (dst == nullptr && cap != 0))
	return SORA_ERR_INVALID;
    *n = 0;
    if (wire != ((ctx->cfg.options & SORA_OPT_WIRE_FRAMES) != 0))
	return fail(ctx, SORA_ERR_STATE, wire? "this context was created without "
		"SORA_OPT_WIRE_FRAMES" : "this context hands out wire frames: "
		"sora_adsb_collect_wire");
    if (!ctx->resolved)
	return fail(ctx, SORA_ERR_STATE, "collect without a resolve");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, use_resolve_stream(ctx));
    const size_t frame_bytes = wire? sizeof(sora_adsb_wire_frame) : sizeof(sora_adsb_frame);
    /* the common case in one small copy: every list held */
    HIP_TRY(ctx, hipMemcpyAsync(ctx->h_summary, ctx->d_summary,
	    SORA_SUM_WORDS * sizeof(uint64_t), hipMemcpyDeviceToHost, ctx->rstream));
    /* ... and, to the host, in one wait: as many frames as the last collect of
     * this context found and a thirty-second more are on their way before the
     * count is known (captures that follow one another resemble one another; a
     * wait costs the host 10 us, what 0.5 MB on the link cost: short lists
     * only, a long one is bound by the link and pays for every byte too many);
     * what is then missing follows below */
    size_t ahead = 0;
    if (kind == hipMemcpyDeviceToHost && ctx->last_collected > 0 &&
	    ctx->last_collected * frame_bytes <= kAheadBytes) {
	ahead = ctx->last_collected + ctx->last_collected / 32 + 64;
	if (ahead > cap)
	    ahead = cap;
	if (ahead > ctx->frames_cap)
	    ahead = (size_t) ctx->frames_cap;
	if (ahead > 0)
	    HIP_TRY(ctx, hipMemcpyAsync(dst, ctx->d_frames, ahead * frame_bytes, kind,
		    ctx->rstream));
    }
    HIP_TRY(ctx, hipStreamSynchronize(ctx->rstream));
    if (ctx->h_summary[SORA_SUM_REC_OVER] != 0 || ctx->h_summary[SORA_SUM_SQ_OVER] != 0 ||
	    ctx->h_summary[SORA_SUM_ACC_OVER] != 0 ||
	    ctx->h_summary[SORA_SUM_FRAMES] > ctx->frames_cap) {
	/* one did not: every region's counters, larger lists, the chunk again */
	int rc = ensure_complete(ctx, &overflow);
	if (rc != SORA_OK)
	    return rc;
	if (overflow)
	    return SORA_ERR_OVERFLOW;	/* ensure_complete has said which list and why */
	ahead = 0;			/* the frames are new ones, in a new list */
    } else {
	ctx->h_total[1] = ctx->h_summary[SORA_SUM_FRAMES];
    }
    const uint64_t total = ctx->h_total[1];
    *n = (size_t) total;
    const size_t take = total < cap? (size_t) total : cap;
    if (kind == hipMemcpyDeviceToHost && cap > 0)	/* (cap 0 only asks for the count) */
	ctx->last_collected = take;
    if (take > ahead) {
	HIP_TRY(ctx, hipMemcpyAsync(static_cast<uint8_t *>(dst) + ahead * frame_bytes,
		reinterpret_cast<const uint8_t *>(ctx->d_frames) + ahead * frame_bytes,
		(take - ahead) * frame_bytes, kind, ctx->rstream));
	HIP_TRY(ctx, hipStreamSynchronize(ctx->rstream));
    }
    return SORA_OK;
}

} /* namespace */

extern "C" int
sora_adsb_collect(sora_adsb_ctx *ctx, sora_adsb_frame *out, size_t cap, size_t *n) {
    return collect_common(ctx, out, cap, n, hipMemcpyDeviceToHost, false);
}

extern "C" int
sora_adsb_collect_device(sora_adsb_ctx *ctx, void *d_out, size_t cap, size_t *n) {
    return collect_common(ctx, d_out, cap, n, hipMemcpyDeviceToDevice, false);
}

extern "C" int
sora_adsb_collect_wire(sora_adsb_ctx *ctx, sora_adsb_wire_frame *out, size_t cap, size_t *n) {
    return collect_common(ctx, out, cap, n, hipMemcpyDeviceToHost, true);
}

extern "C" int
sora_adsb_collect_wire_device(sora_adsb_ctx *ctx, void *d_out, size_t cap, size_t *n) {
    return collect_common(ctx, d_out, cap, n, hipMemcpyDeviceToDevice, true);
}

/*
 * Wire frames -> frames, on the host.  Everything a frame holds beyond the
 * wire frame is a function of the bits: DF (plane.c:461), long (:462), the
 * CRC-24 over the first 88 or 32 bits (:373-385) and with it syndrome and
 * address (:472-480).  A corrected frame carries the corrected bits; the
 * syndrome the device found before the flip is that of its one flipped bit,
 * x^(n - 1 - b) mod G (adsb_tile.h).
 */
extern "C" int
sora_adsb_unpack_frames(const sora_adsb_wire_frame *in, size_t n, int encoding,
	sora_adsb_frame *out) {
    const int bps = bytes_per_sample(encoding);
    if (bps == 0 || ((in == nullptr || out == nullptr) && n != 0))
	return SORA_ERR_INVALID;
    for (size_t i = 0; i < n; i++) {
	const sora_adsb_wire_frame &w = in[i];
	sora_adsb_frame f;
	memset(&f, 0, sizeof f);
	f.bits[0] = w.bits[0];
	f.bits[1] = w.bits[1];
	f.bits[2] = w.bits[2];
	f.bits[3] = w.bits3_cb_flags & 0xffff0000u;
	f.flags = (uint8_t) (w.bits3_cb_flags & 0xffu);
	f.corrected_bit = (int8_t) (uint8_t) ((w.bits3_cb_flags >> 8) & 0xffu);
	f.position = ((uint64_t) (w.pos_hi_amp_hi >> 24) << 32) | w.position_lo;
	const uint64_t amp = ((uint64_t) (w.pos_hi_amp_hi & 0xffffffu) << 32) | w.amp_lo;
	f.amplitude_56 = (double) amp * (encoding == SORA_ENC_SC16? 1.0 / 2147483648.0 :
		1.0 / 32768.0);
	f.sum56 = encoding == SORA_ENC_SC16? 0 : (uint32_t) amp;
	f.df = (uint8_t) (f.bits[0] >> 27);
	const bool is_long = f.df >= 16 && f.df <= 24;
	const int nbits = is_long? 112 : 56;
	uint32_t crc = 0;
	for (int k = 0; k < nbits - 24; k++) {
	    crc = (crc << 1) ^ (((f.bits[k >> 5] >> (31 - (k & 31))) & 1u) << 24);
	    if (crc & 0x1000000u)
		crc ^= 0x1fff409u;
	}
	const uint32_t ap = is_long? (((f.bits[2] & 0xffu) << 16) | (f.bits[3] >> 16)) :
		(f.bits[1] >> 8);
	f.syndrome = (crc & 0xffffffu) ^ ap;
	if (f.corrected_bit >= 0)
	    f.syndrome ^= sora_crc_pow(nbits - 1 - f.corrected_bit);
	const bool squitter = f.df == 11 || f.df == 17 || f.df == 18;
	f.address = squitter? (f.bits[0] & 0xffffffu) : f.syndrome;
	out[i] = f;
    }
    return SORA_OK;
}

extern "C" int
sora_adsb_get_counts(sora_adsb_ctx *ctx, struct sora_adsb_counts *out) {
    bool overflow = false;

    if (ctx == nullptr || out == nullptr)
	return SORA_ERR_INVALID;
    memset(out, 0, sizeof *out);
    if (!ctx->scanned)
	return fail(ctx, SORA_ERR_STATE, "no scan to report on");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    int rc = ensure_complete(ctx, &overflow);
    if (rc != SORA_OK)
	return rc;
    out->positions = ctx->scan.n_scan;
    for (uint32_t i = 0; i < ctx->groups; i++) {
	const uint64_t *st = ctx->h_stats + (size_t) i * SORA_STAT_WORDS;
	out->candidates += st[SORA_STAT_CANDIDATES];
	out->squitters += st[SORA_STAT_SQUITTERS];
	out->dirty_squitters += st[SORA_STAT_DIRTY];
	out->replies += st[SORA_STAT_REPLIES];
	out->corrected += st[SORA_STAT_CORRECTED];
    }
    for (uint32_t i = 0; ctx->resolved && i < ctx->res.regions; i++)
	out->frames += ctx->h_counts[2 * ctx->groups + i];
    return overflow? SORA_ERR_OVERFLOW : SORA_OK;
}

/* ---- streaming ingest ---- */

namespace {

void
stream_release(sora_adsb_ctx *ctx) {
    auto &s = ctx->st;
    for (int i = 0; i < 2; i++) {
	if (s.h_in[i] != nullptr)
	    (void) hipHostFree(s.h_in[i]);
	if (s.h_out[i] != nullptr)
	    (void) hipHostFree(s.h_out[i]);
	if (s.d_in[i] != nullptr)
	    (void) hipFree(s.d_in[i]);
	if (s.ev_in[i] != nullptr)
	    (void) hipEventDestroy(s.ev_in[i]);
	if (s.ev_free[i] != nullptr)
	    (void) hipEventDestroy(s.ev_free[i]);
    }
    if (s.copy != nullptr)
	(void) hipStreamDestroy(s.copy);
    memset(&s, 0, sizeof s);
}

/* reads the next block into slot (behind the carried tail of the other slot)
 * and starts its copy to the device */
int
stream_fill(sora_adsb_ctx *ctx, int slot) {
    auto &s = ctx->st;
    const size_t bps = (size_t) s.bps;
    const int prev = slot ^ 1;

    /* the pinned buffer may still be the source of its last copy */
    HIP_TRY(ctx, hipEventSynchronize(s.ev_in[slot]));
    if (s.carry > 0)
	memcpy(s.h_in[slot], s.h_in[prev] + (s.n_samples[prev] - s.carry) * bps,
		s.carry * bps);
    const size_t want = (size_t) s.block * bps;
    uint8_t *dst = s.h_in[slot] + s.carry * bps;
    size_t got = 0;
    while (got < want) {
	const size_t k = s.read(s.user, dst + got, want - got);
	if (k == 0 || k > want - got)
	    break;
	got += k;
    }
    /* whole samples only: a trailing partial sample is dropped
     * (plane-iq.c:131, :138; radio-file.c:122) */
    const uint64_t fresh = got / bps;
    s.last[slot] = got < want;
    s.n_samples[slot] = s.carry + fresh;
    s.base[slot] = s.consumed - s.carry;
    s.n_owned[slot] = s.last[slot]? s.n_samples[slot] :
	    s.n_samples[slot] - SORA_ADSB_HALO;
    s.consumed += fresh;
    s.carry = s.last[slot]? 0 : SORA_ADSB_HALO;
    s.filled[slot] = true;

    HIP_TRY(ctx, hipStreamWaitEvent(s.copy, s.ev_free[slot], 0));
    if (s.n_samples[slot] > 0)
	HIP_TRY(ctx, hipMemcpyAsync(s.d_in[slot], s.h_in[slot],
		(size_t) s.n_samples[slot] * bps, hipMemcpyHostToDevice, s.copy));
    HIP_TRY(ctx, hipEventRecord(s.ev_in[slot], s.copy));
    return SORA_OK;
}

} /* namespace */

extern "C" int
sora_adsb_stream_begin(sora_adsb_ctx *ctx, int encoding, uint64_t block_samples,
	sora_adsb_read_fn read, void *user) {
    if (ctx == nullptr)
	return SORA_ERR_INVALID;
    const int bps = bytes_per_sample(encoding);
    if (bps == 0 || read == nullptr)
	return fail(ctx, SORA_ERR_INVALID, "bad stream arguments");
    if (block_samples < 1024 ||
	    block_samples + SORA_ADSB_HALO > ctx->cfg.max_chunk_samples)
	return fail(ctx, SORA_ERR_INVALID, "block of %llu samples: need >= 1024 and "
		"block + 239 <= max_chunk_samples (%llu)",
		(unsigned long long) block_samples,
		(unsigned long long) ctx->cfg.max_chunk_samples);
    if (ctx->st.active)
	return fail(ctx, SORA_ERR_STATE, "a stream is already open");
    if ((ctx->cfg.options & SORA_OPT_WIRE_FRAMES) != 0)
	return fail(ctx, SORA_ERR_STATE, "the stream calls hand out struct sora_adsb_frame: "
		"create the context without SORA_OPT_WIRE_FRAMES");
    HIP_TRY(ctx, hipSetDevice(ctx->device));

    auto &s = ctx->st;
    memset(&s, 0, sizeof s);
    s.encoding = encoding;
    s.bps = bps;
    s.block = block_samples;
    s.read = read;
    s.user = user;
    const size_t in_bytes = (size_t) (block_samples + SORA_ADSB_HALO) * (size_t) bps + 16;
    const size_t out_bytes = (size_t) (ctx->frames_cap > 0? ctx->frames_cap : 1) *
	    sizeof(sora_adsb_frame);
    bool ok = hipStreamCreateWithFlags(&s.copy, hipStreamNonBlocking) == hipSuccess;
    for (int i = 0; ok && i < 2; i++) {
	ok = hipHostMalloc(reinterpret_cast<void **>(&s.h_in[i]), in_bytes,
			hipHostMallocDefault) == hipSuccess &&
		hipHostMalloc(reinterpret_cast<void **>(&s.h_out[i]), out_bytes,
			hipHostMallocDefault) == hipSuccess &&
		hipMalloc(reinterpret_cast<void **>(&s.d_in[i]), in_bytes) == hipSuccess &&
		hipEventCreateWithFlags(&s.ev_in[i], hipEventDisableTiming) == hipSuccess &&
		hipEventCreateWithFlags(&s.ev_free[i], hipEventDisableTiming) == hipSuccess;
    }
    if (!ok) {
	stream_release(ctx);
	return fail(ctx, SORA_ERR_NOMEM, "stream buffers (2 x %zu bytes pinned and "
		"device, 2 x %zu bytes of frames)", in_bytes, out_bytes);
    }
    s.out_cap[0] = s.out_cap[1] = ctx->frames_cap > 0? ctx->frames_cap : 1;
    s.active = true;
    return SORA_OK;
}

extern "C" int
sora_adsb_stream_next(sora_adsb_ctx *ctx, const sora_adsb_frame **frames, size_t *n) {
    if (ctx == nullptr || frames == nullptr || n == nullptr)
	return SORA_ERR_INVALID;
    *frames = nullptr;
    *n = 0;
    auto &s = ctx->st;
    if (!s.active)
	return fail(ctx, SORA_ERR_STATE, "no stream is open");
    if (s.done)
	return SORA_END;
    HIP_TRY(ctx, hipSetDevice(ctx->device));

    const int cur = s.cur;
    int rc;
    if (!s.filled[cur] && (rc = stream_fill(ctx, cur)) != SORA_OK)
	return rc;
    /* this block: scan + resolve behind its copy */
    HIP_TRY(ctx, hipStreamWaitEvent(ctx->stream, s.ev_in[cur], 0));
    rc = sora_adsb_submit_device(ctx, s.d_in[cur],
	    (size_t) s.n_samples[cur] * (size_t) s.bps, s.encoding, s.base[cur],
	    s.n_owned[cur]);
    if (rc != SORA_OK)
	return rc;
    HIP_TRY(ctx, hipEventRecord(s.ev_free[cur], ctx->rstream));
    /* the next block: read and copied while the kernels run */
    if (!s.last[cur] && (rc = stream_fill(ctx, cur ^ 1)) != SORA_OK)
	return rc;
    /* waits for the block; a list that overflowed is grown and the block run
     * again (inside the collect), and the pinned frame buffer follows */
    size_t got = 0;
    rc = sora_adsb_collect(ctx, s.h_out[cur], (size_t) s.out_cap[cur], &got);
    if (rc == SORA_OK && got > s.out_cap[cur]) {
	(void) hipHostFree(s.h_out[cur]);
	s.h_out[cur] = nullptr;
	s.out_cap[cur] = ctx->frames_cap > got? ctx->frames_cap : got;
	if (hipHostMalloc(reinterpret_cast<void **>(&s.h_out[cur]),
		(size_t) s.out_cap[cur] * sizeof(sora_adsb_frame), hipHostMallocDefault) != hipSuccess)
	    return fail(ctx, SORA_ERR_NOMEM, "pinned buffer for %llu frames",
		    (unsigned long long) s.out_cap[cur]);
	rc = sora_adsb_collect(ctx, s.h_out[cur], (size_t) s.out_cap[cur], &got);
    }
    if (rc != SORA_OK)
	return rc;
    *frames = s.h_out[cur];
    *n = got;
    s.filled[cur] = false;
    if (s.last[cur])
	s.done = true;
    s.cur = cur ^ 1;
    return SORA_OK;
}

extern "C" int
sora_adsb_stream_end(sora_adsb_ctx *ctx) {
    if (ctx == nullptr)
	return SORA_ERR_INVALID;
    if (!ctx->st.active)
	return SORA_OK;
    (void) hipSetDevice(ctx->device);
    (void) hipStreamSynchronize(ctx->st.copy);
    (void) hipStreamSynchronize(ctx->stream);
    if (ctx->rstream != ctx->stream)
	(void) hipStreamSynchronize(ctx->rstream);
    stream_release(ctx);
    return SORA_OK;
}

#ifdef SORA_SCAN_STAMPS
/* diagnostic build only: sum over workgroups of the 8 phase clocks of wave 0 */
extern "C" int
sora_adsb_debug_stamps(sora_adsb_ctx *ctx, uint64_t out[8]) {
    bool overflow = false;
    int rc = fetch_counts(ctx, &overflow);
    if (rc != SORA_OK)
	return rc;
    for (int k = 0; k < 8; k++)
	out[k] = 0;
    for (uint32_t i = 0; i < ctx->groups; i++)
	for (int k = 0; k < 8; k++)
	    out[k] += ctx->h_stats[(size_t) i * SORA_STAT_WORDS + SORA_STAT_WORDS_BASE + k];
    return SORA_OK;
}
#endif

extern "C" int
sora_adsb_list_regrows(const sora_adsb_ctx *ctx) {
    return ctx != nullptr? (int) ctx->regrows : 0;
}

extern "C" int
sora_adsb_synchronize(sora_adsb_ctx *ctx) {
    if (ctx == nullptr)
	return SORA_ERR_INVALID;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    if (ctx->rstream != ctx->stream)
	HIP_TRY(ctx, hipStreamSynchronize(ctx->rstream));
    return SORA_OK;
}

/*
 * Everything behind the scan -- squitter export, add_known, resolve, the copies
 * of collect -- on a stream of its own (NULL: back on the scan's).  The calls
 * of this context keep their order; what changes is that work ANOTHER context
 * queues on the scan stream (the next capture's scan) no longer waits behind
 * this context's resolve, and this context's frame copy no longer waits behind
 * that scan.
 */
extern "C" int
sora_adsb_set_resolve_stream(sora_adsb_ctx *ctx, void *stream) {
    if (ctx == nullptr)
	return SORA_ERR_INVALID;
    if (ctx->st.active)
	return fail(ctx, SORA_ERR_STATE, "a stream of blocks is open");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    if (ctx->rstream != ctx->stream)
	HIP_TRY(ctx, hipStreamSynchronize(ctx->rstream));
    ctx->rstream = stream != nullptr? static_cast<hipStream_t>(stream) : ctx->stream;
    ctx->resolve_behind = ctx->scan_behind = false;
    return SORA_OK;
}

extern "C" int
sora_adsb_last_scan_ms(sora_adsb_ctx *ctx, float *ms) {
    if (ctx == nullptr || ms == nullptr)
	return SORA_ERR_INVALID;
    if (!ctx->scan_timed)
	return fail(ctx, SORA_ERR_STATE, "no scan has been timed");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, hipEventSynchronize(ctx->ev_scan_end));
    HIP_TRY(ctx, hipEventElapsedTime(ms, ctx->ev_scan_begin, ctx->ev_scan_end));
    return SORA_OK;
}

extern "C" const char *
sora_adsb_scan_kernel_name(const sora_adsb_ctx *ctx, int encoding) {
    if (ctx == nullptr || bytes_per_sample(encoding) == 0)
	return "";
    if (encoding == SORA_ENC_SC16)
	return "adsb_scan_kernel<true>";
    return ctx->scan_variant == 1? "adsb_scan_direct_kernel<256>" : "adsb_scan_kernel<false>";
}

extern "C" int
sora_adsb_magnitude(sora_adsb_ctx *ctx, const void *d_iq, size_t nbytes,
	int encoding, float *d_m) {
    if (ctx == nullptr)
	return SORA_ERR_INVALID;
    const int bps = bytes_per_sample(encoding);
    if (bps == 0 || (d_iq == nullptr && nbytes != 0) || (d_m == nullptr && nbytes != 0))
	return fail(ctx, SORA_ERR_INVALID, "bad magnitude arguments");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, use_scan_stream(ctx));
    HIP_TRY(ctx, sora_launch_magnitude(static_cast<const uint8_t *>(d_iq),
	    nbytes / (size_t) bps, encoding, d_m, ctx->stream));
    return SORA_OK;
}

/* ---- device memory helpers ---- */

extern "C" int
sora_adsb_device_alloc(int device, size_t nbytes, void **d_ptr) {
    if (d_ptr == nullptr)
	return SORA_ERR_INVALID;
    *d_ptr = nullptr;
    if (hipSetDevice(device) != hipSuccess)
	return SORA_ERR_NO_DEVICE;
    if (hipMalloc(d_ptr, nbytes > 0? nbytes : 1) != hipSuccess)
	return SORA_ERR_NOMEM;
    return SORA_OK;
}

extern "C" int
sora_adsb_device_free(int device, void *d_ptr) {
    if (hipSetDevice(device) != hipSuccess)
	return SORA_ERR_NO_DEVICE;
    return hipFree(d_ptr) == hipSuccess? SORA_OK : SORA_ERR_HIP;
}

extern "C" int
sora_adsb_copy_to_device(int device, void *d_dst, const void *h_src, size_t nbytes) {
    if (hipSetDevice(device) != hipSuccess)
	return SORA_ERR_NO_DEVICE;
    if (nbytes == 0)
	return SORA_OK;
    return hipMemcpy(d_dst, h_src, nbytes, hipMemcpyHostToDevice) == hipSuccess?
	    SORA_OK : SORA_ERR_HIP;
}

extern "C" int
sora_adsb_copy_to_host(int device, void *h_dst, const void *d_src, size_t nbytes) {
    if (hipSetDevice(device) != hipSuccess)
	return SORA_ERR_NO_DEVICE;
    if (nbytes == 0)
	return SORA_OK;
    return hipMemcpy(h_dst, d_src, nbytes, hipMemcpyDeviceToHost) == hipSuccess?
	    SORA_OK : SORA_ERR_HIP;
}

/*
 * Host-to-host copy that does not read the destination's cache lines first:
 * 16-byte non-temporal stores (movntdq), for moving a frame list from a pinned
 * staging buffer into memory shared between processes.  With several ranks of
 * a node copying at once the memory bus is what is shared, and a plain memcpy
 * below glibc's non-temporal threshold moves every line three times (read
 * source, read destination for ownership, write back) instead of twice.
 */
extern "C" int
sora_adsb_host_copy(void *dst, const void *src, size_t nbytes) {
    if ((dst == nullptr || src == nullptr) && nbytes != 0)
	return SORA_ERR_INVALID;
    typedef long long v2di __attribute__((vector_size(16)));
    typedef long long v2di_u __attribute__((vector_size(16), aligned(1)));
    unsigned char *d = static_cast<unsigned char *>(dst);
    const unsigned char *s = static_cast<const unsigned char *>(src);
    size_t head = (16 - (reinterpret_cast<uintptr_t>(d) & 15)) & 15;
    if (head > nbytes)
	head = nbytes;
    memcpy(d, s, head);
    d += head; s += head; nbytes -= head;
    size_t body = nbytes & ~(size_t) 63;
    for (size_t i = 0; i < body; i += 64) {
	const v2di a0 = *reinterpret_cast<const v2di_u *>(s + i);
	const v2di a1 = *reinterpret_cast<const v2di_u *>(s + i + 16);
	const v2di a2 = *reinterpret_cast<const v2di_u *>(s + i + 32);
	const v2di a3 = *reinterpret_cast<const v2di_u *>(s + i + 48);
	__builtin_nontemporal_store(a0, reinterpret_cast<v2di *>(d + i));
	__builtin_nontemporal_store(a1, reinterpret_cast<v2di *>(d + i + 16));
	__builtin_nontemporal_store(a2, reinterpret_cast<v2di *>(d + i + 32));
	__builtin_nontemporal_store(a3, reinterpret_cast<v2di *>(d + i + 48));
    }
    memcpy(d + body, s + body, nbytes - body);
    __atomic_thread_fence(__ATOMIC_SEQ_CST);	/* mfence: orders the streaming stores */
    return SORA_OK;
}

extern "C" int
sora_adsb_host_register(int device, void *h_ptr, size_t nbytes) {
    if (h_ptr == nullptr || nbytes == 0)
	return SORA_ERR_INVALID;
    if (hipSetDevice(device) != hipSuccess)
	return SORA_ERR_NO_DEVICE;
    if (hipHostRegister(h_ptr, nbytes, hipHostRegisterDefault) == hipSuccess)
	return SORA_OK;
    (void) hipGetLastError();	/* not sticky: the caller may go on without */
    return SORA_ERR_HIP;
}

extern "C" int
sora_adsb_host_unregister(int device, void *h_ptr) {
    if (h_ptr == nullptr)
	return SORA_ERR_INVALID;
    if (hipSetDevice(device) != hipSuccess)
	return SORA_ERR_NO_DEVICE;
    return hipHostUnregister(h_ptr) == hipSuccess? SORA_OK : SORA_ERR_HIP;
}

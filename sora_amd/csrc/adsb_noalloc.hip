/*
 * adsb_noalloc.hip -- sora_adsb_decode_noalloc(): the scan + resolve sequence
 * of a context (adsb_gpu.hip) with every list in one static device array, for
 * callers that cannot allocate (include/sora_adsb_gpu.h).  Host code only; the
 * kernels are the ones a context launches (adsb_kernels.hip), with the
 * first-seen table in its hashed form because 2^24 entries do not fit a modest
 * array.
 */
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string.h>

#include <mutex>

#include "../../include/sora_adsb_gpu.h"
#include "adsb_kernels.h"
#include "adsb_tile.h"

namespace {

/* ---- layout of the array: fixed offsets, every part 256-byte aligned ---- */
constexpr uint64_t kMaxSamples = SORA_NOALLOC_MAX_SAMPLES;
constexpr uint32_t kMaxGroups = 256;			/* scan workgroups (regions) */
constexpr uint32_t kMaxRegions = kMaxGroups * 8;	/* accepted regions: <= 8 parts each */
/* record slots: one candidate per two positions at most, per region rounded up
 * to whole tiles (sora_adsb_decode_noalloc picks the regions so that it fits) */
constexpr uint64_t kRecSlots = kMaxSamples / 2 + (kMaxSamples >> 3);	/* 1.25 Mi */
constexpr uint64_t kSqSlots = 1u << 18;
constexpr uint64_t kAccSlots = kRecSlots + 16 * kMaxRegions;
constexpr uint64_t kFlatSlots = 1u << 20;
constexpr uint32_t kSeenSlots = SORA_NOALLOC_MAX_ADDRESSES;

constexpr uint64_t
aligned(uint64_t n) {
    return (n + 255) & ~255ull;
}

constexpr uint64_t kOffRec = 0;						/* 8 B per slot */
constexpr uint64_t kOffSq = kOffRec + aligned((kRecSlots + 8) * 8);
constexpr uint64_t kOffSqBits = kOffSq + aligned(kSqSlots * 8);		/* 16 B per slot */
constexpr uint64_t kOffAcc = kOffSqBits + aligned(kSqSlots * 16);
constexpr uint64_t kOffFlat = kOffAcc + aligned(kAccSlots * 8);
constexpr uint64_t kOffKnown = kOffFlat + aligned(kFlatSlots * 8);	/* bitmap + fold */
constexpr uint64_t kOffSeenVal = kOffKnown + aligned(SORA_KNOWN_WORDS * 4ull);
constexpr uint64_t kOffSeenKey = kOffSeenVal + aligned(kSeenSlots * 8ull);	/* behind the
								 * values: one memset of 0xff clears both */
constexpr uint64_t kOffFail = kOffSeenKey + aligned(kSeenSlots * 4ull);
constexpr uint64_t kOffRecCount = kOffFail + 256;
constexpr uint64_t kOffSqCount = kOffRecCount + aligned(kMaxGroups * 4);
constexpr uint64_t kOffAccCount = kOffSqCount + aligned(kMaxGroups * 4);
constexpr uint64_t kOffAccOffset = kOffAccCount + aligned(kMaxRegions * 4);
constexpr uint64_t kOffStats = kOffAccOffset + aligned((kMaxRegions + 1) * 8);
constexpr uint64_t kOffTileEnd = kOffStats + aligned((uint64_t) kMaxGroups * SORA_STAT_WORDS * 8);
constexpr uint64_t kTileSlots = kMaxSamples / 7808 + 16;	/* no scan kernel's tile is shorter */
constexpr uint64_t kScratchUsed = kOffTileEnd + aligned(kTileSlots * 4);
static_assert(kScratchUsed <= SORA_NOALLOC_SCRATCH_BYTES, "the lists do not fit the array");
static_assert(kRecSlots % SORA_REC_UNIT == 0, "record regions are whole units");

} /* namespace */

/*
 * The array.  In host code its name is only a handle for hipGetSymbolAddress():
 * every GPU holds its own copy at its own address, looked up once per device
 * below, and that pointer -- never the name -- goes to kernels and memsets.
 */
static __device__ __align__(256) unsigned char scratch[SORA_NOALLOC_SCRATCH_BYTES];

namespace {

constexpr int kMaxDevices = 64;

struct device_slot {
    std::mutex lock;
    bool ready = false;		/* set once the look-ups below have all succeeded */
    unsigned char *ptr = nullptr;	/* this device's copy of `scratch` */
    uint32_t cus = 0;
    bool gfx950 = false;
} g_dev[kMaxDevices];

/* Per device, with that device current: what kind it is and where its copy of
 * the array lies.  A look-up that fails is not remembered: the next call tries
 * again (a transient failure must not disable the device for the process). */
hipError_t
device_setup(int device) {
    device_slot &d = g_dev[device];
    std::lock_guard<std::mutex> hold(d.lock);
    if (d.ready)
	return hipSuccess;
    hipDeviceProp_t prop;
    void *p = nullptr;

    hipError_t e = hipGetDeviceProperties(&prop, device);
    if (e != hipSuccess)
	return e;
    d.gfx950 = strncmp(prop.gcnArchName, "gfx950", 6) == 0;
    d.cus = prop.multiProcessorCount > 0? (uint32_t) prop.multiProcessorCount : 1;
    if (d.gfx950) {
	e = hipGetSymbolAddress(&p, HIP_SYMBOL(scratch));
	if (e != hipSuccess)
	    return e;
	d.ptr = static_cast<unsigned char *>(p);
    }
    d.ready = true;
    return hipSuccess;
}

/* the caller's current device, put back when the entry point returns */
struct device_guard {
    int before = -1;
    device_guard() {
	if (hipGetDevice(&before) != hipSuccess)
	    before = -1;
    }
    ~device_guard() {
	if (before >= 0)
	    (void) hipSetDevice(before);
    }
};

int
bytes_per_sample(int encoding) {
    return encoding == SORA_ENC_UC8 || encoding == SORA_ENC_SC8? 2 :
	    encoding == SORA_ENC_SC16? 4 : 0;
}

constexpr int kVariant = 1;		/* the 8-bit scan kernel contexts use by default */

} /* namespace */

extern "C" int
sora_adsb_decode_noalloc(int device, void *stream_, const void *d_iq, size_t nbytes,
	int encoding, uint64_t base_position, uint64_t n_owned, int emit_mode, int options,
	int keep_known, sora_adsb_frame *d_frames, size_t cap_frames, uint64_t *d_status) {
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    const int bps = bytes_per_sample(encoding);

    if (bps == 0 || d_status == nullptr || (d_iq == nullptr && nbytes != 0) ||
	    (d_frames == nullptr && cap_frames != 0) ||
	    (emit_mode != SORA_EMIT_ACCEPTED && emit_mode != SORA_EMIT_CANDIDATES))
	return SORA_ERR_INVALID;
    if ((reinterpret_cast<uintptr_t>(d_iq) & 15) != 0)
	return SORA_ERR_ALIGN;
    const uint64_t n_samples = nbytes / (size_t) bps;
    if (n_samples > kMaxSamples)
	return SORA_ERR_INVALID;	/* more than SORA_NOALLOC_MAX_SAMPLES: use a context */
    if (base_position + n_samples >= (1ull << 40))
	return SORA_ERR_INVALID;
    if (device < 0 || device >= kMaxDevices)
	return SORA_ERR_NO_DEVICE;
    device_guard restore;
    if (hipSetDevice(device) != hipSuccess)
	return SORA_ERR_NO_DEVICE;
    if (device_setup(device) != hipSuccess)
	return SORA_ERR_HIP;
    device_slot &dev = g_dev[device];
    if (!dev.gfx950 || dev.ptr == nullptr)
	return SORA_ERR_NO_DEVICE;
    unsigned char *const base = dev.ptr;

    /* the scan's geometry, as sora_adsb_scan() derives it */
    uint64_t n_scan = n_samples >= SORA_ADSB_WINDOW? n_samples - SORA_ADSB_HALO : 0;
    if (n_owned < n_scan)
	n_scan = n_owned;
    const uint32_t tile_pos = sora_scan_tile_pos(encoding, kVariant);
    const uint64_t n_tiles = sora_scan_tiles(n_scan, encoding, kVariant);
    uint32_t groups = n_tiles / 2 > kMaxGroups? kMaxGroups :
	    n_tiles >= 2? (uint32_t) (n_tiles / 2) : 1;
    /* a region holds what its tiles can produce at most: one candidate per two
     * positions, in whole 8-record groups; fewer, longer regions until the
     * slots suffice (one region always does) */
    const auto region_slots = [&](uint32_t g) {
	const uint64_t tiles = (n_tiles + g - 1) / g;
	return (uint32_t) ((tiles * tile_pos / 2 + 8 + SORA_REC_UNIT - 1) & ~(uint64_t) (SORA_REC_UNIT - 1));
    };
    while (groups > 1 && (uint64_t) groups * region_slots(groups) > kRecSlots)
	groups--;
    if ((uint64_t) groups * region_slots(groups) > kRecSlots)
	return SORA_ERR_INVALID;	/* not reached for n_samples <= kMaxSamples */

    sora_scan_args a;
    memset(&a, 0, sizeof a);
    a.iq = static_cast<const uint8_t *>(d_iq);
    a.n_samples = n_samples;
    a.n_scan = n_scan;
    a.n_tiles = n_tiles;
    a.encoding = encoding;
    a.emit_all = emit_mode == SORA_EMIT_CANDIDATES;
    a.correct = (options & SORA_OPT_CORRECT_SINGLE_BIT) != 0;
    a.rec = base + kOffRec;
    a.tile_end = reinterpret_cast<uint32_t *>(base + kOffTileEnd);
    a.sq = reinterpret_cast<uint64_t *>(base + kOffSq);
    a.sq_bits = reinterpret_cast<uint32_t *>(base + kOffSqBits);
    a.rec_cap = region_slots(groups);
    a.sq_cap = (uint32_t) (kSqSlots / groups);
    a.rec_count = reinterpret_cast<uint32_t *>(base + kOffRecCount);
    a.sq_count = reinterpret_cast<uint32_t *>(base + kOffSqCount);
    a.stats = reinterpret_cast<uint64_t *>(base + kOffStats);
    /* first_seen stays NULL: the scan kernel merges into the direct table only;
     * first_seen_kernel<hashed> follows */

    sora_resolve_args r;
    memset(&r, 0, sizeof r);
    r.iq = a.iq;
    r.n_samples = n_samples;
    r.base_position = base_position;
    r.encoding = encoding;
    r.emit_all = a.emit_all;
    r.groups = groups;
    r.rec = a.rec;
    r.tile_end = a.tile_end;
    r.n_tiles = n_tiles;
    r.tile_pos = tile_pos;
    r.rec_count = a.rec_count;
    r.rec_cap = a.rec_cap;
    r.sq = a.sq;
    r.sq_bits = a.sq_bits;
    r.sq_count = a.sq_count;
    r.sq_cap = a.sq_cap;
    r.first_seen = reinterpret_cast<uint64_t *>(base + kOffSeenVal);
    r.known_bits = reinterpret_cast<uint32_t *>(base + kOffKnown);
    r.seen_keys = reinterpret_cast<uint32_t *>(base + kOffSeenKey);
    r.seen_mask = kSeenSlots - 1;
    r.seen_fail = reinterpret_cast<uint32_t *>(base + kOffFail);
    r.filter_parts = sora_filter_parts(groups);
    r.regions = groups * r.filter_parts;
    r.acc = reinterpret_cast<uint64_t *>(base + kOffAcc);
    r.acc_count = reinterpret_cast<uint32_t *>(base + kOffAccCount);
    r.acc_cap = (uint32_t) (kAccSlots / r.regions);
    r.acc_offset = reinterpret_cast<uint64_t *>(base + kOffAccOffset);
    r.acc_flat = reinterpret_cast<uint64_t *>(base + kOffFlat);
    r.frames = d_frames;
    r.frames_cap = cap_frames < kFlatSlots? cap_frames : kFlatSlots;
    if (r.regions > kMaxRegions)
	return SORA_ERR_INVALID;	/* not reached: parts <= 8 */

#define NOALLOC_TRY(call) do { if ((call) != hipSuccess) return SORA_ERR_HIP; } while (0)
    /* what must start clean: the array keeps the last call's contents */
    if (!keep_known) {
	NOALLOC_TRY(hipMemsetAsync(base + kOffKnown, 0, SORA_KNOWN_WORDS * 4ull, stream));
	NOALLOC_TRY(hipMemsetAsync(base + kOffSeenVal, 0xff,
		kOffSeenKey + kSeenSlots * 4ull - kOffSeenVal, stream));
    }
    NOALLOC_TRY(hipMemsetAsync(base + kOffFail, 0, sizeof(uint32_t), stream));
    /* (the per-region counters are written by every region the kernels run) */
    NOALLOC_TRY(sora_launch_scan(&a, groups, kVariant, stream));
    NOALLOC_TRY(sora_launch_first_seen(&r, stream));
    NOALLOC_TRY(sora_launch_filter(&r, stream));
    NOALLOC_TRY(sora_launch_scan_counts(r.acc_count, r.acc_cap, r.acc_offset, r.regions,
	    nullptr, stream));
    NOALLOC_TRY(sora_launch_finalize(&r, dev.cus, stream));
    NOALLOC_TRY(sora_launch_noalloc_status(&r, cap_frames, d_status, stream));
#undef NOALLOC_TRY
    return SORA_OK;
}

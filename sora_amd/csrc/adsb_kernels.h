/*
 * adsb_kernels.h -- launch interface between the C-ABI context
 * (adsb_gpu.hip) and the gfx950 kernels (adsb_kernels.hip).  Internal.
 */
#ifndef SORA_ADSB_KERNELS_H_
#define SORA_ADSB_KERNELS_H_

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/sora_adsb_gpu.h"

/* per-workgroup statistics written by the scan kernel */
#define SORA_STAT_CANDIDATES	0
#define SORA_STAT_SQUITTERS	1
#define SORA_STAT_DIRTY		2
#define SORA_STAT_REPLIES	3
#define SORA_STAT_CORRECTED	4
#define SORA_STAT_WORDS_BASE	5
#ifdef SORA_SCAN_STAMPS
#define SORA_STAT_WORDS		13	/* + 8 phase clocks (diagnostic build) */
#else
#define SORA_STAT_WORDS		5
#endif

#define SORA_KNOWN_BITMAP_WORDS	(1u << 19)	/* 2^24 addresses */
#define SORA_KNOWN_FOLD_BITS	17
#define SORA_KNOWN_FOLD_WORDS	(1u << (SORA_KNOWN_FOLD_BITS - 5))	/* 16 KB */
#define SORA_KNOWN_WORDS	(SORA_KNOWN_BITMAP_WORDS + SORA_KNOWN_FOLD_WORDS)

#define SORA_AMP_SLOTS	32	/* see sora_scan_args::sq_sum */

struct sora_scan_args {
    const uint8_t *iq;		/* device, 16-byte aligned */
    uint64_t n_samples;		/* samples in the chunk */
    uint64_t n_scan;		/* positions [0, n_scan) are scanned */
    uint64_t n_tiles;
    int32_t encoding;
    int32_t emit_all;		/* also record dirty squitters */
    int32_t correct;		/* SORA_OPT_CORRECT_SINGLE_BIT */
    int32_t pad_;
    /* candidate records, one per candidate in position order, in two arrays
     * (5 bytes instead of the 8 of a full record: the scan writes and the
     * filter reads them all): key = payload[23:0] | kind[25:24] |
     * tile position[13:8] << 26, tail = tile position[7:0]; the tile position
     * counts from the first position of the candidate's tile (< 2^14 in every
     * scan kernel).  Which tile: tile_end[t], for every tile t of the chunk, is
     * the number of records of t's workgroup range up to and including t */
    /* Both in ONE array, in units of SORA_REC_UNIT records: the unit's keys
     * (4 bytes each), then its tails -- 5 * SORA_REC_UNIT bytes per unit, a region
     * is rec_cap / SORA_REC_UNIT units (rec_cap is a multiple of the unit).  A
     * flush of the scan kernel (1024 records) is then one contiguous 5 KB piece,
     * and a filter iteration reads its keys and tails out of the same piece
     * (until round 3 two arrays: two thin store streams beside the reads). */
    uint8_t *rec;		/* [groups][rec_cap / SORA_REC_UNIT] units */
    uint32_t *tile_end;		/* [n_tiles] */
    uint64_t *sq;		/* [groups][sq_cap] clean squitters */
    /* [groups][sq_cap][4]: the 112 message bits of each of them as phase 4
     * sliced them (MSB first, sora_candidate::bits), so that finalize need not
     * slice them again from 224 samples: the candidate record of a clean (or
     * corrected) squitter carries its index in this list where a reply's carries
     * its address */
    uint32_t *sq_bits;
    /* [groups][sq_cap], or null: the amplitude sum of a clean squitter (the sum
     * over its 112 samples that finalize hands out, plane-iq.c:72-73) where the
     * scan kernel computes it itself, under its own loads and arithmetic -- the
     * sc16 kernel does, for up to SORA_AMP_SLOTS squitters per tile; an
     * entry it did not get to is negative, and finalize sums that one's samples */
    double *sq_sum;
    uint32_t rec_cap, sq_cap;	/* per-group capacities */
    uint32_t *rec_count;	/* [groups] records produced (may exceed cap) */
    uint32_t *sq_count;		/* [groups] */
    uint64_t *stats;		/* [groups][SORA_STAT_WORDS] */
    /* the scan kernels merge a workgroup's clean squitters into the first-seen
     * table as the workgroup ends (first_seen_kernel's work, hidden behind
     * the other workgroups of the CU); NULL: not merged */
    uint64_t *first_seen;
    uint32_t *known_bits;
    uint64_t base_position;
    uint64_t seen_tag;		/* see sora_resolve_args */
    uint64_t *summary;		/* see sora_resolve_args; NULL: none */
};

/* does this scan kernel merge its squitters itself? */
int sora_scan_merges(int encoding, int variant);

struct sora_resolve_args {
    const uint8_t *iq;
    uint64_t n_samples;
    uint64_t base_position;
    int32_t encoding;
    int32_t emit_all;
    uint32_t groups;
    const uint8_t *rec;		/* scan output, see sora_scan_args */
    const uint32_t *tile_end;
    uint64_t n_tiles;		/* of the scan: region g starts at tile n_tiles * g / groups */
    uint32_t tile_pos;		/* positions per tile of the scan kernel that ran */
    const uint32_t *rec_count;
    uint32_t rec_cap;
    const uint64_t *sq;
    const uint32_t *sq_bits;	/* see sora_scan_args */
    const double *sq_sum;	/* see sora_scan_args; null: none were computed */
    const uint32_t *sq_count;
    uint32_t sq_cap;
    uint64_t *first_seen;	/* [1 << 24] smallest global position, or ~0
				 * (hashed form: [seen_mask + 1] values, below) */
    uint32_t *known_bits;	/* [SORA_KNOWN_WORDS] bitmap of the addresses in
				 * first_seen, then its 2^17-bit fold
				 * (bit addr & 0x1ffff) for the filter's LDS */
    /* Hashed form of the first-seen table, for sora_adsb_decode_noalloc() whose
     * scratch cannot hold 2^24 entries: open addressing over seen_mask + 1 slots,
     * seen_keys[slot] = address or SORA_SEEN_EMPTY, first_seen[slot] its position.
     * NULL in every context: they use the direct table. */
    uint32_t *seen_keys;
    uint32_t seen_mask;
    uint32_t *seen_fail;	/* set when the table is full (an address was dropped) */
    /* Epoch of the first-seen table, in bits 63..40 of every entry (positions
     * stay below 2^40): a reset takes the next, SMALLER tag instead of refilling
     * 128 MB -- entries of earlier epochs are larger than anything this epoch
     * stores or compares with, so atomicMin replaces them and `entry < tag | P'
     * is false for them, as for the all-ones entry of an address never seen. */
    uint64_t seen_tag;
    /* What the host needs to know before it copies the frames, in four words
     * (SORA_SUM_*), zeroed when the scan starts: the largest count of a region
     * that did not fit its share of a list (0: every one did) and the number of
     * frames.  One 32-byte copy instead of every region's counters. */
    uint64_t *summary;
    uint32_t regions;		/* accepted regions: groups * filter_parts */
    uint32_t filter_parts;	/* sora_filter_parts(groups) */
    uint64_t *acc;		/* [regions][acc_cap] accepted records */
    uint32_t *acc_count;	/* [regions] */
    uint32_t acc_cap;
    uint64_t *acc_offset;	/* [regions + 1] exclusive scan of clamped counts */
    uint64_t *acc_flat;		/* [frames_cap] the accepted records, flat */
    struct sora_adsb_frame *frames;	/* wire != 0: struct sora_adsb_wire_frame */
    uint64_t frames_cap;
    int32_t wire;		/* SORA_OPT_WIRE_FRAMES */
    int32_t pad_;
};

#define SORA_SEEN_EMPTY		0xffffffffu

/* candidate records in memory (sora_scan_args::rec) */
#define SORA_REC_UNIT		256u			/* records per unit */
#define SORA_REC_UNIT_BYTES	(5u * SORA_REC_UNIT)	/* keys, then tails */
/* byte offset, from the start of a region, of record i's key and of its tail */
#define SORA_REC_KEY_AT(i)	(((i) / SORA_REC_UNIT) * SORA_REC_UNIT_BYTES + 4u * ((i) % SORA_REC_UNIT))
#define SORA_REC_TAIL_AT(i)	(((i) / SORA_REC_UNIT) * SORA_REC_UNIT_BYTES + 4u * SORA_REC_UNIT + (i) % SORA_REC_UNIT)
/* bytes of a region of rec_cap records */
#define SORA_REC_REGION_BYTES(rec_cap)	((uint64_t) ((rec_cap) / SORA_REC_UNIT) * SORA_REC_UNIT_BYTES)

#define SORA_SUM_REC_OVER	0	/* max records of a scan region beyond rec_cap */
#define SORA_SUM_SQ_OVER	1	/* ... squitters beyond sq_cap */
#define SORA_SUM_ACC_OVER	2	/* ... accepted records of a region beyond acc_cap */
#define SORA_SUM_FRAMES		3	/* acc_offset[regions]: frames available */
#define SORA_SUM_WORDS		4

/* The tiles of one workgroup's range: the scan kernel and the filter keep a
 * word per tile of a range in LDS (sora_scan_args::tile_end).  sora_adsb_scan
 * checks it as positions: fewer than 2^SORA_REGION_POS_BITS per range, which at
 * 7808 positions per tile or more is fewer than SORA_RANGE_TILES_MAX tiles. */
#define SORA_REGION_POS_BITS	22
#define SORA_RANGE_TILES_MAX	544

/* positions per tile of the scan kernel for this encoding and variant */
uint32_t sora_scan_tile_pos(int encoding, int variant);
/* tiles (workgroup passes) a chunk of n_scan positions needs */
uint64_t sora_scan_tiles(uint64_t n_scan, int encoding, int variant);
/* LDS bytes / thread count of the scan kernel, for documentation and tests */
uint32_t sora_scan_groups_per_cu(int encoding, int variant);

hipError_t sora_launch_scan(const struct sora_scan_args *a, uint32_t groups, int variant,
	hipStream_t stream);
hipError_t sora_launch_fill_u64(uint64_t *dst, uint64_t value, uint64_t n,
	hipStream_t stream);
/* first_seen[addr] = min(first_seen[addr], base + pos) over the squitter lists */
hipError_t sora_launch_first_seen(const struct sora_resolve_args *a,
	hipStream_t stream);
/* same from flat global records (position << 24 | address) */
hipError_t sora_launch_add_known(uint64_t *first_seen, uint32_t *known_bits,
	uint64_t seen_tag, const uint64_t *records, uint64_t n, hipStream_t stream);
/* same from n_blocks exchange blocks of `stride` words (word 0: record count,
 * read on the device), leaving out block `skip` */
hipError_t sora_launch_add_known_blocks(uint64_t *first_seen, uint32_t *known_bits,
	uint64_t seen_tag, const uint64_t *blocks, uint32_t n_blocks, uint32_t stride,
	uint32_t skip, hipStream_t stream);
/* per address the earliest squitter of this scan (needs first_seen) -> flat
 * global records, unordered; *counter = how many there are (may exceed cap) */
hipError_t sora_launch_export_first(const struct sora_resolve_args *a, uint64_t *dst,
	uint64_t dst_cap, uint64_t *counter, hipStream_t stream);
hipError_t sora_launch_filter(const struct sora_resolve_args *a,
	hipStream_t stream);
/* sora_adsb_decode_noalloc(): frames available and overflow flags of the run
 * into status[0..1], on the device (SORA_NOALLOC_* in sora_adsb_gpu.h) */
hipError_t sora_launch_noalloc_status(const struct sora_resolve_args *a,
	uint64_t cap_frames, uint64_t *status, hipStream_t stream);
/* offsets[i] = sum of the clamped counts before i; offsets[n] and, if not NULL,
 * *total = their sum */
hipError_t sora_launch_scan_counts(const uint32_t *counts, uint32_t cap,
	uint64_t *offsets, uint32_t n, uint64_t *total, hipStream_t stream);
uint32_t sora_filter_parts(uint32_t groups);		/* accepted regions per scan region */
uint32_t sora_filter_max_regions(uint32_t max_groups);
/* flattens the accepted regions (needs acc_offset) and re-slices the frames */
hipError_t sora_launch_finalize(const struct sora_resolve_args *a, uint32_t cus,
	hipStream_t stream);
hipError_t sora_launch_magnitude(const uint8_t *iq, uint64_t n_samples,
	int encoding, float *m, hipStream_t stream);

#endif /* SORA_ADSB_KERNELS_H_ */

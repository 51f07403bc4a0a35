/*
 * sora_adsb_gpu.h -- C ABI of libsora_adsb_gpu.so, the MI355X (gfx950)
 * implementation of Sora's ADS-B receive chain.
 *
 * The reference has no plugin / FFI layer: its seams
 * are the plain C functions of src/ads-b/plane-iq.h:10-13 (demodulator
 * iterator), src/ads-b/plane.h:16-17 (validator) and
 * src/ads-b/planes-main-loop.h:6-9 (mode entry).  Those keep their names and
 * signatures in include/sora_adsb_host.h; this header is the thin layer
 * underneath them that hands large IQ buffers to the HIP kernels.  Plain
 * pointers, sizes and integer return codes only: no exceptions and no abort()
 * cross this boundary (the reference aborts on allocation failure,
 * src/util/exception.c:58-59; this library returns SORA_ERR_NOMEM instead).
 *
 * One context per device and per stream of samples; a context is used from
 * one host thread at a time.
 *
 * What a context computes, per position p of a chunk (all of it a pure
 * function of samples p .. p+239, reference src/ads-b/plane-iq.c:121-194):
 *   magnitude m = I^2 + Q^2                      plane-iq.c:131-136
 *   candidate iff the 8-comparison predicate     plane-iq.c:173-175
 *   112 PPM bits, bit_k = m[p+17+2k] <= m[p+16+2k]   plane-iq.c:58-86
 *   amplitude_56 = sum_{k<56} (m[p+16+2k]+m[p+17+2k])/2   plane-iq.c:62-73
 *   DF, CRC-24 syndrome, address                 plane.c:457-484
 * and then the one stateful rule: a DF11/17/18 frame
 * is accepted iff its syndrome is 0; any other DF iff an accepted DF11/17/18
 * frame with the same address starts at a smaller position.
 */
#ifndef SORA_ADSB_GPU_H_
#define SORA_ADSB_GPU_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SORA_OK			0
#define SORA_END		1	/* sora_adsb_stream_next: the stream is over */
#define SORA_ERR_INVALID	(-1)	/* bad argument */
#define SORA_ERR_NO_DEVICE	(-2)	/* no usable gfx950 device */
#define SORA_ERR_HIP		(-3)	/* a HIP call failed: sora_adsb_last_error() */
#define SORA_ERR_OVERFLOW	(-4)	/* a device list was too small and could not be
					 * grown; see SORA_OPT_FIXED_LISTS */
#define SORA_ERR_NOMEM		(-5)
#define SORA_ERR_ALIGN		(-6)	/* device pointer not 16-byte aligned */
#define SORA_ERR_STATE		(-7)	/* call out of order */

/* encodings: the numbers of reference src/radio/radio-file.h:6-12 */
#define SORA_ENC_UC8	1
#define SORA_ENC_SC16	2
#define SORA_ENC_SC8	3

#define SORA_ADSB_WINDOW	240	/* samples a candidate reads */
#define SORA_ADSB_HALO		239	/* samples past the last owned position */
#define SORA_ADSB_MSG_BITS	112

/* emit modes */
#define SORA_EMIT_ACCEPTED	0	/* frames plane_set_parse_message() accepts
					 * (what planes_main_loop prints) */
#define SORA_EMIT_CANDIDATES	1	/* every candidate plane_iq_get_next()
					 * yields, accepted or not (parity mode) */

/* frame flags */
#define SORA_FRAME_LONG		0x01	/* 16 <= DF <= 24, plane.c:462 */
#define SORA_FRAME_SQUITTER	0x02	/* DF 11, 17, 18, plane.c:473 */
#define SORA_FRAME_ACCEPTED	0x04
#define SORA_FRAME_CORRECTED	0x08	/* one bit was flipped back (extension, below) */

/*
 * Options.  SORA_OPT_CORRECT_SINGLE_BIT is SURVEY.md's extension D3 / section
 * 8 f.3 and is OFF by default: the reference never corrects (plane.c:476).
 * With it, a DF11/17/18 candidate whose syndrome is that of ONE flipped bit b,
 * 5 <= b < 56 (DF11) or < 112 (DF17/18), is accepted with that bit flipped
 * back: bits, address and df are those of the corrected message, syndrome is
 * the one found before the flip, corrected_bit = b, and its address becomes
 * known from its position on like any squitter's.
 */
#define SORA_OPT_CORRECT_SINGLE_BIT	0x01
/*
 * The device lists (candidate records, clean squitters, accepted frames) are
 * sized by sora_adsb_config and shared out evenly over the scan's workgroup
 * regions, so a burst of candidates can overflow one region while the lists are
 * nearly empty overall.  The reference never drops a candidate
 * (plane-iq.c:141-194), and by default neither does this library: the kernels
 * count what they could not store, and sora_adsb_collect / _get_counts /
 * _stream_next then allocate lists of the sizes that were needed and run the
 * chunk again before they return (the IQ buffer of the scan must still be in
 * place, as it must be for the resolve anyway).  With SORA_OPT_FIXED_LISTS the
 * lists never change and an overflow is reported as SORA_ERR_OVERFLOW (for
 * callers that must not allocate after create, and for tests of that report).
 * One case is reported even without it: a squitter list that overflowed AFTER
 * its export went to other shards (sora_adsb_export_block, _copy_squitters) --
 * they have merged a truncated list, a local re-run cannot repair that.
 */
#define SORA_OPT_FIXED_LISTS		0x02
/*
 * The frame list leaves the device as 28-byte struct sora_adsb_wire_frame
 * instead of 48-byte struct sora_adsb_frame: with dense traffic (BASELINE.json
 * configs[3]: 6 M frames per 14.4 GB) the copy of the frame list over PCIe is
 * what a step waits for.  Nothing is lost: DF, syndrome and address are
 * functions of the bits, the amplitude travels as the exact integer it is a
 * multiple of; sora_adsb_unpack_frames() rebuilds the 48-byte frames on the
 * host, whenever and for whichever frames the caller wants them.  A context
 * with this option hands out frames through sora_adsb_collect_wire[_device]
 * only (sora_adsb_collect, _collect_device and the sora_adsb_stream_* calls
 * return SORA_ERR_STATE).
 */
#define SORA_OPT_WIRE_FRAMES		0x04

/* 48 bytes, what the demodulator + validator know about one frame */
struct sora_adsb_frame {
    uint64_t position;	   /* in_flight_msg.position, plane-iq.c:47 (global) */
    uint32_t bits[4];	   /* 112 bits MSB first (bit k = bits[k/32]>>(31-k%32));
			    * all 112 are sliced even for short frames
			    * (SURVEY.md D2); low 16 bits of bits[3] are 0 */
    double amplitude_56;   /* plane-iq.c:72-73; amplitude = this / 56 (:189) */
    uint32_t syndrome;	   /* crc ^ ap, plane.c:472-475 / :480 */
    uint32_t address;	   /* bits[8,32) for squitters, crc ^ ap otherwise */
    uint32_t sum56;	   /* 8-bit encodings: integer sum over k<56 of
			    * M[p+16+2k]+M[p+17+2k], M = I^2+Q^2 in counts;
			    * amplitude_56 == sum56 * 2^-15 exactly.  0 for sc16 */
    uint8_t df;		   /* bits[0,5) */
    int8_t corrected_bit;  /* -1: none (the reference never corrects, D3);
			    * else see SORA_OPT_CORRECT_SINGLE_BIT */
    uint8_t flags;	   /* SORA_FRAME_* */
    uint8_t reserved;
};

/* 28 bytes: a frame as it crosses PCIe under SORA_OPT_WIRE_FRAMES */
struct sora_adsb_wire_frame {
    uint32_t bits[3];		/* message bits 0..95, MSB first */
    uint32_t bits3_cb_flags;	/* bits 96..111 << 16 | corrected_bit (0xff: none)
				 * << 8 | SORA_FRAME_* flags */
    uint32_t position_lo;	/* position bits 31..0 */
    uint32_t pos_hi_amp_hi;	/* position bits 39..32 << 24 | amplitude bits 55..32 */
    uint32_t amp_lo;		/* amplitude bits 31..0: the integer A with
				 * amplitude_56 = A * 2^-15 (uc8, sc8: A = sum56) or
				 * A * 2^-31 (sc16), both exact */
};

struct sora_adsb_config {
    uint64_t max_chunk_samples;	/* largest chunk a scan will be given */
    uint64_t max_candidates;	/* initial device list sizes (they grow on
				 * overflow, see SORA_OPT_FIXED_LISTS);
				 * 0 = max_chunk_samples/16 + 65536 */
    uint64_t max_squitters;	/* 0 = max_chunk_samples/256 + 65536 */
    uint64_t max_frames;	/* 0 = max_squitters (EMIT_ACCEPTED) or
				 * max_candidates (EMIT_CANDIDATES) */
    int32_t emit_mode;		/* SORA_EMIT_* */
    int32_t options;		/* SORA_OPT_*, 0 = the reference's behaviour */
    void *stream;		/* hipStream_t to run on; NULL = own stream */
};

struct sora_adsb_counts {	/* of the last scan / resolve */
    uint64_t positions;		/* positions scanned */
    uint64_t candidates;	/* predicate hits (plane_iq_get_next yields) */
    uint64_t squitters;		/* clean DF11/17/18 */
    uint64_t dirty_squitters;	/* DF11/17/18 with a non-zero syndrome */
    uint64_t replies;		/* candidates of every other DF */
    uint64_t frames;		/* frames sora_adsb_collect will return */
    uint64_t corrected;		/* dirty squitters taken back by
				 * SORA_OPT_CORRECT_SINGLE_BIT (not in dirty_squitters) */
};

typedef struct sora_adsb_ctx sora_adsb_ctx;

/* Library / device queries.  No GPU needed for the first two. */
const char *sora_adsb_version(void);
const char *sora_adsb_strerror(int code);
int sora_adsb_device_count(void);

int sora_adsb_create(sora_adsb_ctx **out, int device,
	const struct sora_adsb_config *cfg);
void sora_adsb_destroy(sora_adsb_ctx *ctx);
const char *sora_adsb_last_error(const sora_adsb_ctx *ctx);

/* Forget every known address (a new, unrelated capture). */
int sora_adsb_reset(sora_adsb_ctx *ctx);

/*
 * Stage 1 -- the fused scan kernel, asynchronous on the context's stream.
 * d_iq: DEVICE pointer (16-byte aligned) to `nbytes` bytes of interleaved IQ;
 * a trailing partial sample is dropped (plane-iq.c:131, :138).  Positions
 * 0 .. n_owned-1 of the chunk are scanned, and of those only p with
 * p + 240 <= n_samples (plane-iq.c:148-166): a middle shard
 * passes its owned count and carries 239 halo samples behind it, the last
 * shard passes n_owned = n_samples.  base_position is the global index of
 * the chunk's first sample.  An odd 8-bit sample count reads the last dword of
 * the buffer whole (two bytes past nbytes, inside the 16-byte alignment slack
 * every device allocation has); those bytes only reach positions the
 * p + 240 <= n_samples rule drops.  SORA_ERR_INVALID also when a workgroup's range of
 * tiles would span 2^22 positions or more: the scan kernel and the filter keep
 * one word per tile of a range in LDS (where the range's records of each tile
 * end), 544 of them.  That is chunks beyond about 10^11 samples, or less with
 * SORA_ADSB_SCAN_ROUNDS / SORA_ADSB_SCAN_MIN_TILES set to few, long ranges.
 */
int sora_adsb_scan(sora_adsb_ctx *ctx, const void *d_iq, size_t nbytes,
	int encoding, uint64_t base_position, uint64_t n_owned);

/*
 * Exchange step for time-sharded captures.  What the
 * other shards need from this one is, per address, the earliest clean squitter
 * of the last scan: 8-byte records (position << 24 | address), one per address
 * this scan saw first (a few hundred per hour, not every squitter).
 * squitter_count merges the scan's squitters into the first-seen table, builds
 * that list and returns its length (synchronizes); copy_squitters copies it to
 * a DEVICE buffer the caller owns; add_known merges records of other shards
 * into this context's table.  The copies are asynchronous.
 */
int sora_adsb_squitter_count(sora_adsb_ctx *ctx, uint64_t *n); /* syncs */
int sora_adsb_copy_squitters(sora_adsb_ctx *ctx, void *d_dst, size_t cap_records);
int sora_adsb_add_known(sora_adsb_ctx *ctx, const void *d_records,
	size_t n_records);

/*
 * The same exchange with no host synchronisation, for a collective of fixed
 * size queued on the context's stream (an all-gather of one block per shard):
 * export_block writes an exchange block into DEVICE memory of the caller --
 * word 0 the number of records this scan exports, words 1 .. block_records-1
 * the records (the first block_records-1 of them if there are more);
 * add_known_blocks merges n_blocks such blocks, laid end to end, into this
 * context's table, reading every count on the device and leaving out block
 * skip_block (this shard's own; -1: none).  Both are asynchronous.  A count
 * above block_records-1 means records were dropped: the caller reads the
 * counts back when it collects the frames and must then treat the result as
 * overflowed.  (No counterpart in the reference: plane.c:479-483 is one
 * sequential validator.)
 */
int sora_adsb_export_block(sora_adsb_ctx *ctx, void *d_block, size_t block_records);
int sora_adsb_add_known_blocks(sora_adsb_ctx *ctx, const void *d_blocks,
	size_t n_blocks, size_t block_records, int skip_block);

/*
 * Stage 2 -- merges this scan's squitters into the first-seen table, filters
 * the replies against it, sorts the accepted positions and re-slices them
 * into frames (bits, amplitude, syndrome).  Asynchronous.
 */
int sora_adsb_resolve(sora_adsb_ctx *ctx);

/* scan + resolve on a device buffer */
int sora_adsb_submit_device(sora_adsb_ctx *ctx, const void *d_iq, size_t nbytes,
	int encoding, uint64_t base_position, uint64_t n_owned);
/* same from host memory: staged through the context's device buffer */
int sora_adsb_submit_host(sora_adsb_ctx *ctx, const void *h_iq, size_t nbytes,
	int encoding, uint64_t base_position, uint64_t n_owned);

/*
 * Streaming ingest: captures larger than HBM, files,
 * live stdin.  No counterpart in the reference, whose iterator pulls single
 * samples through a 4096-float ring (plane-iq.c:141-166).  The library reads
 * blocks of block_samples through `read`, carries the last 239 samples of a
 * block in front of the next (a candidate at p reads p .. p+239), and double
 * buffers: while block k is scanned and resolved on the context's stream,
 * block k+1 is read into pinned memory and copied to the device on a second
 * stream.  Known addresses persist across blocks (plane.c:90-105).
 *
 * read(user, dst, nbytes) stores up to nbytes bytes and returns the count,
 * fewer than asked only at the end of the stream (fread semantics).
 * stream_next hands out the frames of one block, in position order, in memory
 * that stays valid until the next stream_next / stream_end; it returns SORA_END
 * (with *n = 0) once the last block has been handed out.
 * Needs block_samples + 239 <= max_chunk_samples and block_samples >= 1024.
 */
typedef size_t (*sora_adsb_read_fn)(void *user, void *dst, size_t nbytes);
int sora_adsb_stream_begin(sora_adsb_ctx *ctx, int encoding, uint64_t block_samples,
	sora_adsb_read_fn read, void *user);
int sora_adsb_stream_next(sora_adsb_ctx *ctx, const struct sora_adsb_frame **frames,
	size_t *n);
int sora_adsb_stream_end(sora_adsb_ctx *ctx);

/*
 * Waits for the stream, then copies the frames of the last resolve, in
 * increasing position, to host memory (sora_adsb_collect) or to a device
 * buffer (sora_adsb_collect_device).  *n is the number available; at most
 * cap are written.  A device list that overflowed is grown and the chunk run
 * again first (SORA_OPT_FIXED_LISTS: SORA_ERR_OVERFLOW instead).
 * What `out` holds behind the first min(*n, cap) frames is unspecified: to the
 * host a short list travels before its count is known, sized by the context's
 * last collect, and may have been longer than the list turned out to be.
 */
int sora_adsb_collect(sora_adsb_ctx *ctx, struct sora_adsb_frame *out,
	size_t cap, size_t *n);
int sora_adsb_collect_device(sora_adsb_ctx *ctx, void *d_out, size_t cap,
	size_t *n);

/* The same for a context created with SORA_OPT_WIRE_FRAMES. */
int sora_adsb_collect_wire(sora_adsb_ctx *ctx, struct sora_adsb_wire_frame *out,
	size_t cap, size_t *n);
int sora_adsb_collect_wire_device(sora_adsb_ctx *ctx, void *d_out, size_t cap,
	size_t *n);
/*
 * Wire frames back into struct sora_adsb_frame, field for field what the
 * context would have written without the option (host code, no device).
 * `encoding`: the one the frames were scanned with (it scales the amplitude).
 */
int sora_adsb_unpack_frames(const struct sora_adsb_wire_frame *in, size_t n,
	int encoding, struct sora_adsb_frame *out);

int sora_adsb_get_counts(sora_adsb_ctx *ctx, struct sora_adsb_counts *out); /* syncs */
/* How often this context has re-run a chunk with larger lists so far. */
int sora_adsb_list_regrows(const sora_adsb_ctx *ctx);
int sora_adsb_synchronize(sora_adsb_ctx *ctx);

/*
 * Two contexts that take turns on one GPU (two captures in flight): give both
 * the same stream in sora_adsb_config and each a second stream here.  The
 * scans then follow one another on the first stream with no wait between
 * them, while everything behind a scan -- squitter export, add_known*, the
 * resolve, the copies of sora_adsb_collect* -- runs on `stream` (a hipStream_t;
 * NULL: back on the context's own), ordered against this context's scans by
 * events.  The calls of one context take effect in the order they were made
 * either way.  Waits for the context's pending work.  (No counterpart in the
 * reference: its loop is one thread, plane-iq.c:141-194.)
 */
int sora_adsb_set_resolve_stream(sora_adsb_ctx *ctx, void *stream);

/* Duration of the last scan kernel alone, HIP events on the context's
 * stream, in milliseconds (syncs). */
int sora_adsb_last_scan_ms(sora_adsb_ctx *ctx, float *ms);
/* Name of the scan kernel this context launches for `encoding`, as profilers
 * print it (8-bit encodings: see SORA_ADSB_SCAN_VARIANT in INTEGRATION.md);
 * "" for a bad argument.  For benchmarks that report per-kernel figures. */
const char *sora_adsb_scan_kernel_name(const sora_adsb_ctx *ctx, int encoding);

/*
 * scan + resolve + frames to a device buffer -- sora_adsb_submit_device()
 * followed by sora_adsb_collect_device() -- for callers that CANNOT ALLOCATE:
 * code under HIP graph capture, tight serving loops, harnesses that forbid
 * hipMalloc.  It has no counterpart in the reference (whose demodulator holds a
 * 4096-float ring and a malloc'ed list of in-flight messages,
 * plane-iq.c:22-56) and none among the reference's callers; it exists for those
 * callers alone.  It needs no context: every list the kernels use lives in one
 * static device array of the library, SORA_NOALLOC_SCRATCH_BYTES (42 MiB) per
 * GPU the entry point has run on, held until the process ends.  The call
 * allocates nothing, frees nothing and never waits: it queues its kernels on
 * `stream` (a hipStream_t; NULL = the null stream) and returns.
 *
 * At most SORA_NOALLOC_MAX_SAMPLES samples per call (one second at 2 MS/s);
 * more is SORA_ERR_INVALID.  Arguments as sora_adsb_submit_device; emit_mode
 * and options as in sora_adsb_config (SORA_OPT_FIXED_LISTS is implied).
 * keep_known = 0 starts from an empty set of known addresses, like a new
 * context; != 0 keeps the set the previous call on this GPU left, so that
 * consecutive chunks of one stream (base_position advancing, 239-sample halo
 * behind every chunk but the last) decode like one capture.
 * d_frames: DEVICE memory for cap_frames frames, in increasing position.
 * d_status: DEVICE memory for two uint64_t: [0] the number of frames available
 * (at most cap_frames are written), [1] zero, or SORA_NOALLOC_* flags of what
 * did not fit -- the frame list is then incomplete.  The record list is sized
 * for the physical bound (one candidate per two positions) and cannot overflow.
 *
 * All calls on one GPU share that GPU's array: calls that overlap on one GPU
 * (two streams, two threads) race; serialize them.  The first call on a GPU
 * looks the array up (hipGetSymbolAddress) and reads the device properties:
 * make it outside a graph capture (nbytes = 0 will do).
 */
#define SORA_NOALLOC_MAX_SAMPLES	(1u << 21)
#define SORA_NOALLOC_SCRATCH_BYTES	(42u << 20)
#define SORA_NOALLOC_MAX_ADDRESSES	65536	/* distinct known addresses */
#define SORA_NOALLOC_RECORDS_FULL	0x01	/* cannot happen; checked all the same */
#define SORA_NOALLOC_SQUITTERS_FULL	0x02	/* > 2^18 clean squitters in the chunk */
#define SORA_NOALLOC_FRAMES_FULL	0x04	/* more frames than cap_frames (or 2^20) */
#define SORA_NOALLOC_ADDRESSES_FULL	0x08	/* > SORA_NOALLOC_MAX_ADDRESSES known */
int sora_adsb_decode_noalloc(int device, void *stream, const void *d_iq, size_t nbytes,
	int encoding, uint64_t base_position, uint64_t n_owned, int emit_mode,
	int options, int keep_known, struct sora_adsb_frame *d_frames, size_t cap_frames,
	uint64_t *d_status);

/*
 * Magnitudes only (row a1), for the 1-ulp check north_star asks for:
 * m[i] as float for every sample, DEVICE in, DEVICE out.
 */
int sora_adsb_magnitude(sora_adsb_ctx *ctx, const void *d_iq, size_t nbytes,
	int encoding, float *d_m);

/* Device memory helpers so that C callers need no HIP headers. */
int sora_adsb_device_alloc(int device, size_t nbytes, void **d_ptr);
int sora_adsb_device_free(int device, void *d_ptr);
int sora_adsb_copy_to_device(int device, void *d_dst, const void *h_src,
	size_t nbytes);
int sora_adsb_copy_to_host(int device, void *h_dst, const void *d_src,
	size_t nbytes);
/*
 * Host memory to host memory with non-temporal stores: for moving a frame
 * list from a pinned staging buffer into memory shared between the ranks of a
 * node (where the segment cannot be page-locked, see below) without reading
 * the destination first.  Touches no device.  (No counterpart in the
 * reference.)
 */
int sora_adsb_host_copy(void *dst, const void *src, size_t nbytes);

/*
 * Page-locks host memory the caller owns (for example a shared-memory segment
 * that several per-GPU processes write their frame lists into) so that
 * sora_adsb_collect() copies into it by DMA at full PCIe rate.  No counterpart
 * in the reference.
 */
int sora_adsb_host_register(int device, void *h_ptr, size_t nbytes);
int sora_adsb_host_unregister(int device, void *h_ptr);

#ifdef __cplusplus
}
#endif

#endif /* SORA_ADSB_GPU_H_ */

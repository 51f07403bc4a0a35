// stream_mix_probe.cpp -- how a streaming read, VALU work, workgroup barriers and a
// thin stream of record stores share an MI355X, in the scan kernel's shape:
// 256-thread workgroups, five per CU, 16 rounds, a 63-run wave stride, the next
// tile's four 16-byte loads issued one tile ahead.
//   hipcc -O3 --offload-arch=gfx950 -o stream_mix_probe stream_mix_probe.cpp
// Per tile every wave executes `valu` packed-16-bit VALU instructions on eight
// register chains, the workgroup passes `barriers` LDS barriers, and the first
// `recs` threads store a 4-byte key and a 2-byte tail (the scan writes about
// 169 per tile).  Reported: time for 14.4 GB and the read rate.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { \
    printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
constexpr uint32_t kTile = 244 * 64;	// bytes owned per tile (252 runs read)

template <int kValu, int kBarriers, int kRecs, int kBurst = 1, int kNt = 0, int kCoal = 0>
__global__ void __launch_bounds__(256, 5)
mix(const uint8_t *src, uint64_t n_tiles, uint32_t *key, uint16_t *tail, uint32_t rec_cap,
	uint32_t *out) {
    __shared__ uint32_t s_x[257];
    const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint64_t begin = n_tiles * blockIdx.x / gridDim.x;
    const uint64_t end = n_tiles * (blockIdx.x + 1) / gridDim.x;
    uint32_t r[8], n_rec = 0;
    for (int i = 0; i < 8; i++)
	r[i] = tid * (i + 3);
    uint32_t *my_key = key + (uint64_t) blockIdx.x * rec_cap;
    uint16_t *my_tail = tail + (uint64_t) blockIdx.x * rec_cap;
    auto load = [&](u32x4 *v, uint64_t t) {
	auto rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t *>(src) + t * kTile, 0,
		16384 + 64, 0x00020000);
#pragma unroll
	for (int q = 0; q < 4; q++)
	    v[q] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, (kCoal & 1)? wave * 4032 + q * 1008 + lane * 16 :
		    (63 * wave + lane) * 64 + q * 16, 0, ((kCoal & 2)? 2 : 0) | ((kCoal & 4)? 16 : 0));
    };
    auto rsrc_key = __builtin_amdgcn_make_buffer_rsrc(my_key, 0, rec_cap * 4, 0x00020000);
    auto rsrc_tail = __builtin_amdgcn_make_buffer_rsrc(my_tail, 0, rec_cap * 2, 0x00020000);
    auto rsrc_ctr = __builtin_amdgcn_make_buffer_rsrc(out + 2048, 0, 4, 0x00020000);
    auto rsrc_logk = __builtin_amdgcn_make_buffer_rsrc(key, 0, 900000u * 256u * 4u, 0x00020000);
    auto rsrc_logt = __builtin_amdgcn_make_buffer_rsrc(tail, 0, 900000u * 256u * 2u, 0x00020000);
    auto rsrc_allk = __builtin_amdgcn_make_buffer_rsrc(key, 0, 20480u * 16384u * 4u, 0x00020000);
    auto rsrc_allt = __builtin_amdgcn_make_buffer_rsrc(tail, 0, 20480u * 16384u * 2u, 0x00020000);
    uint32_t chunk = 0, pending = 0;
    uint64_t last_wall = __builtin_amdgcn_s_memrealtime() >> (kBurst <= -10? -kBurst : 0);
    u32x4 v[4];
    if (begin < end)
	load(v, begin);
    if (kRecs > 0 && kBurst == -1) {
	chunk = (uint32_t) __builtin_amdgcn_raw_ptr_buffer_atomic_add_i32(1, rsrc_ctr,
		tid == 0? 0u : 0xfffffff0u, 0, 0);
	__builtin_amdgcn_raw_buffer_store_b32(0u, rsrc_logk, 0xfffffff0u, 0, 0);
	__builtin_amdgcn_raw_buffer_store_b16((uint16_t) 0, rsrc_logt, 0xfffffff0u, 0, 0);
    }
    if (kRecs > 0 && (kBurst == 1 || kBurst <= -2 || kBurst >= 100)) {	// the first pass finds what every pass finds
	__builtin_amdgcn_raw_buffer_store_b32(0u, rsrc_key, 0xfffffff0u, 0, 0);
	if (kNt != 11)
	    __builtin_amdgcn_raw_buffer_store_b16((uint16_t) 0, rsrc_tail, 0xfffffff0u, 0, 0);
    }
    for (uint64_t t = begin; t < end; t++) {
	uint32_t x = 0;
#pragma unroll
	for (int q = 0; q < 4; q++)
	    x ^= v[q][0] ^ v[q][1] ^ v[q][2] ^ v[q][3];
	load(v, t + 1 < end? t + 1 : t);
#pragma unroll
	for (int k = 0; k < kValu / 8; k++) {
#pragma unroll
	    for (int i = 0; i < 8; i++)
		asm volatile("v_pk_sub_i16 %0, %0, %1 clamp" : "+v"(r[i]) : "v"(x));
	}
	if (kBarriers > 0) {
	    s_x[tid] = x;
	    for (int b = 0; b < kBarriers; b++) {
		asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
		x += s_x[(tid + 64 * (b + 1)) & 255];
	    }
	}
	if (kRecs > 0 && kBurst >= 100) {
	    // staged (as if in LDS) and flushed every kBurst - 100 tiles by ONE pair of
	    // wide stores per thread (16 bytes of keys, 8 of tails), issued on every
	    // pass and out of range on the others: the fixed operation count again
	    constexpr uint32_t kEvery = kBurst - 100;
	    const bool flush = (t - begin) % kEvery == kEvery - 1;
	    const uint32_t quads = (kRecs * kEvery + 3) / 4;
	    const uint32_t q = flush && tid < quads? n_rec / 4 + tid : 0x0ffffff0u;
	    typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
	    u32x4 k4 = {x, x, x, x};
	    u32x2 t2 = {x, r[0]};
	    __builtin_amdgcn_raw_buffer_store_b128(k4, rsrc_key, q * 16, 0, 0);
	    __builtin_amdgcn_raw_buffer_store_b64(t2, rsrc_tail, q * 8, 0, 0);
	    if (flush)
		n_rec += quads * 4;
	} else if (kRecs > 0 && kBurst <= -10) {
	    // records held back (as if staged in LDS) and written by every workgroup
	    // in the same short window of real time: when the 100 MHz clock crosses a
	    // multiple of 2^(-kBurst) ticks
	    const uint64_t now = __builtin_readcyclecounter() >> (-kBurst);
	    const uint64_t wall = __builtin_amdgcn_s_memrealtime() >> (-kBurst);
	    (void) now;
	    pending += kRecs;
	    if (wall != last_wall || t + 1 == end) {
		last_wall = wall;
		for (uint32_t i = tid; i < pending; i += 256) {
		    if (n_rec + i < rec_cap) {
			my_key[n_rec + i] = x;
			my_tail[n_rec + i] = (uint16_t) r[0];
		    }
		}
		n_rec += pending;
		pending = 0;
	    }
	} else if (kRecs > 0 && kBurst <= -2) {
	    // the workgroup's own dense record sequence, stored block-interleaved:
	    // its j-th block of kBlock records sits at block j * gridDim.x + blockIdx.x,
	    // so that workgroups that run together write next to each other
	    constexpr uint32_t kBlock = kBurst == -2? 256 : kBurst == -3? 64 : 1024;
	    const uint32_t slot = n_rec + tid;
	    const uint32_t at = ((slot / kBlock) * gridDim.x + blockIdx.x) * kBlock + slot % kBlock;
	    const uint32_t idx = tid < (uint32_t) kRecs? at : 0x3ffffff0u;
	    __builtin_amdgcn_raw_buffer_store_b32(x, rsrc_allk, idx * 4, 0, 0);
	    __builtin_amdgcn_raw_buffer_store_b16((uint16_t) r[0], rsrc_allt, idx * 2, 0, 0);
	    n_rec += kRecs;
	} else if (kRecs > 0 && kBurst == -1) {
	    // sequential log, the chunk taken one tile ahead: thread 0's atomic of
	    // the previous pass has long returned; it is handed round through LDS
	    // at the pass's own barriers, and the next one is requested now
	    if (tid == 0)
		s_x[256] = chunk;
	    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
	    const uint64_t c = s_x[256] % 900000u;
	    chunk = (uint32_t) __builtin_amdgcn_raw_ptr_buffer_atomic_add_i32(1, rsrc_ctr,
		    tid == 0? 0u : 0xfffffff0u, 0, 0);
	    const uint32_t idx = tid < (uint32_t) kRecs? (uint32_t) c * 256u + tid : 0xfffffff0u;
	    __builtin_amdgcn_raw_buffer_store_b32(x, rsrc_logk, idx * 4, 0, 0);
	    __builtin_amdgcn_raw_buffer_store_b16((uint16_t) r[0], rsrc_logt, idx * 2, 0, 0);
	} else if (kRecs > 0 && kBurst == 0) {
	    // one sequential log for the whole grid: the next free chunk of 192 records
	    __shared__ uint32_t s_chunk;
	    if (tid == 0)
		s_chunk = atomicAdd(out + 1024, 1u);
	    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
	    const uint64_t c = s_chunk % 900000u;
	    if (tid < (uint32_t) kRecs) {
		key[c * 192 + tid] = x;
		tail[c * 192 + tid] = (uint16_t) r[0];
	    }
	} else if (kRecs > 0 && kNt >= 20) {
	    // the same bytes in wider stores: 20: a lane writes four keys (16 bytes)
	    // and four tails (8 bytes); 21..23: dword keys with cache policy bits
	    // sc0 / sc1 / sc0 sc1 on the stores
	    if (kNt == 20) {
		const uint32_t quads = (kRecs + 3) / 4;
		const uint32_t q = tid < quads? n_rec / 4 + tid : 0x0ffffff0u;
		u32x4 k4 = {x, x, x, x};
		__builtin_amdgcn_raw_buffer_store_b128(k4, rsrc_key, q * 16, 0, 0);
		typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
		u32x2 t2 = {x, r[0]};
		__builtin_amdgcn_raw_buffer_store_b64(t2, rsrc_tail, q * 8, 0, 0);
		n_rec += quads * 4;
	    } else {
		constexpr int aux = kNt == 21? 1 : kNt == 22? 16 : kNt == 23? 17 : kNt == 24? 18 : kNt == 25? 19 : 3;
		const uint32_t idx = tid < (uint32_t) kRecs? n_rec + tid : 0x3ffffff0u;
		__builtin_amdgcn_raw_buffer_store_b32(x, rsrc_key, idx * 4, 0, aux);
		__builtin_amdgcn_raw_buffer_store_b16((uint16_t) r[0], rsrc_tail, idx * 2, 0, aux);
		n_rec += kRecs;
	    }
	} else if (kRecs > 0 && kNt >= 10) {
	    // what part of the cost is the instructions, what part the memory:
	    // 10: both stores, every lane out of range (nothing is written)
	    // 11: the key store alone      12: both, always into the region's first
	    // 256 slots (they stay in L2)
	    const uint32_t idx = kNt == 10? 0x3ffffff0u : tid < (uint32_t) kRecs? (kNt == 12? tid :
		    n_rec + tid) : 0x3ffffff0u;
	    __builtin_amdgcn_raw_buffer_store_b32(x, rsrc_key, idx * 4, 0, 0);
	    if (kNt != 11)
		__builtin_amdgcn_raw_buffer_store_b16((uint16_t) r[0], rsrc_tail, idx * 2, 0, 0);
	    n_rec += kRecs;
	} else if (kRecs > 0 && kBurst == 1) {
	    // as the scan kernel does it: every thread issues both stores on every
	    // pass, threads without a record with an offset the buffer check drops, so
	    // that the wait for the next tile's loads (issued before) never covers them
	    const uint32_t idx = tid < (uint32_t) kRecs? n_rec + tid : 0x3fffffffu;
	    __builtin_amdgcn_raw_buffer_store_b32(x, rsrc_key, idx * 4, 0, kNt? 2 : 0);
	    __builtin_amdgcn_raw_buffer_store_b16((uint16_t) r[0], rsrc_tail, idx * 2, 0, kNt? 2 : 0);
	    n_rec += kRecs;
	} else if (kRecs > 0 && (t - begin) % (kBurst > 0? kBurst : 1) == (kBurst > 0? kBurst : 1) - 1) {
	    // the records of kBurst tiles at once (as if staged in LDS in between)
	    for (uint32_t i = tid; i < (uint32_t) (kRecs * kBurst); i += 256) {
		if (n_rec + i < rec_cap) {
		    my_key[n_rec + i] = x;
		    my_tail[n_rec + i] = (uint16_t) r[0];
		}
	    }
	    n_rec += kRecs * kBurst;
	}
	r[1] ^= x;
    }
    uint32_t acc = 0;
    for (int i = 0; i < 8; i++)
	acc ^= r[i];
    if (acc == 0x12345678u)
	out[blockIdx.x] = acc;
}

template <int kValu, int kBarriers, int kRecs, int kBurst = 1, int kNt = 0, int kCoal = 0> int
run(const uint8_t *d, uint64_t bytes, uint32_t *key, uint16_t *tail, uint32_t rec_cap,
	uint32_t *d_out) {
    const uint64_t n_tiles = (bytes - 32768) / kTile;
    const int blocks = 256 * 5 * 16;
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    float best = 1e9f;
    for (int i = 0; i < 5; i++) {
	CHECK(hipEventRecord(e0));
	hipLaunchKernelGGL((mix<kValu, kBarriers, kRecs, kBurst, kNt, kCoal>), dim3(blocks), dim3(256), 0, 0, d,
		n_tiles, key, tail, rec_cap, d_out);
	CHECK(hipEventRecord(e1));
	CHECK(hipEventSynchronize(e1));
	float ms;
	CHECK(hipEventElapsedTime(&ms, e0, e1));
	if (i > 0 && ms < best)
	    best = ms;
    }
    printf("%s valu/tile %3d  barriers %d  records/tile %3d burst %2d nt %d:  %.3f ms  %.2f TB/s read  (VALU alone at "
	    "580 G/s: %.2f ms)\n", kCoal == 0? "lane-run " : kCoal == 1? "coalesced" : kCoal == 3? "coalesced nt" : kCoal == 2? "lane-run nt" : kCoal == 5? "coalesced sc1" : "other", kValu, kBarriers, kRecs, kBurst, kNt, best, n_tiles * (double) kTile / best / 1e9,
	    n_tiles * 4.0 * kValu / 580e9 * 1e3);
    return 0;
}

int
main(int argc, char **argv) {
    const uint64_t bytes = 14400000000ull;
    const uint32_t rec_cap = 16384;
    // argv[1]: how the record arrays are allocated: 0 hipMalloc, 1 uncached, 2 fine-grained
    const int kind = argc > 1? atoi(argv[1]) : 0;
    uint8_t *d;
    uint32_t *d_out, *key;
    uint16_t *tail;
    CHECK(hipMalloc(&d, bytes));
    CHECK(hipMalloc(&d_out, 1 << 20));
    const size_t kb = (size_t) 256 * 5 * 16 * rec_cap * 4;
    if (kind == 0) {
	CHECK(hipMalloc(&key, kb));
	CHECK(hipMalloc(&tail, kb / 2));
    } else {
	const unsigned flags = kind == 1? hipDeviceMallocUncached : hipDeviceMallocFinegrained;
	CHECK(hipExtMallocWithFlags((void **) &key, kb, flags));
	CHECK(hipExtMallocWithFlags((void **) &tail, kb / 2, flags));
	printf("record arrays: %s\n", kind == 1? "uncached" : "fine-grained");
    }
    CHECK(hipMemset(d, 1, bytes));
    run<320, 2, 0>(d, bytes, key, tail, rec_cap, d_out);
    run<320, 2, 2>(d, bytes, key, tail, rec_cap, d_out);
    run<320, 2, 169>(d, bytes, key, tail, rec_cap, d_out);
    run<400, 2, 169>(d, bytes, key, tail, rec_cap, d_out);
    run<320, 2, 169, 106, 0>(d, bytes, key, tail, rec_cap, d_out);
    if (kind != 0)
	return 0;
    run<0, 0, 0>(d, bytes, key, tail, rec_cap, d_out);
    run<200, 0, 0>(d, bytes, key, tail, rec_cap, d_out);
    run<320, 0, 0>(d, bytes, key, tail, rec_cap, d_out);
    run<400, 0, 0>(d, bytes, key, tail, rec_cap, d_out);
    run<480, 0, 0>(d, bytes, key, tail, rec_cap, d_out);
    run<0, 2, 0>(d, bytes, key, tail, rec_cap, d_out);
    run<320, 2, 0>(d, bytes, key, tail, rec_cap, d_out);
    run<400, 2, 0>(d, bytes, key, tail, rec_cap, d_out);
    run<0, 0, 169>(d, bytes, key, tail, rec_cap, d_out);
    run<320, 0, 169>(d, bytes, key, tail, rec_cap, d_out);
    run<400, 0, 169>(d, bytes, key, tail, rec_cap, d_out);
    run<320, 2, 169>(d, bytes, key, tail, rec_cap, d_out);
    run<400, 2, 169>(d, bytes, key, tail, rec_cap, d_out);
    run<480, 2, 169>(d, bytes, key, tail, rec_cap, d_out);
    run<400, 2, 256>(d, bytes, key, tail, rec_cap, d_out);
    run<320, 2, 169, 1, 1>(d, bytes, key, tail, rec_cap, d_out);
    run<320, 2, 169, 4, 0>(d, bytes, key, tail, rec_cap, d_out);
    run<320, 2, 169, 16, 0>(d, bytes, key, tail, rec_cap, d_out);
    run<320, 2, 169, 16, 1>(d, bytes, key, tail, rec_cap, d_out);
    run<320, 2, 169, 45, 0>(d, bytes, key, tail, rec_cap, d_out);
    run<320, 2, 64, 1, 0>(d, bytes, key, tail, rec_cap, d_out);
    run<320, 2, 32, 1, 0>(d, bytes, key, tail, rec_cap, d_out);
    run<320, 2, 2, 1, 0>(d, bytes, key, tail, rec_cap, d_out);
    CHECK(hipMemset(d_out, 0, 1 << 20));
    run<320, 2, 0, 1, 0, 3>(d, bytes, key, tail, rec_cap, d_out);
    run<320, 2, 169, 1, 0, 3>(d, bytes, key, tail, rec_cap, d_out);
    run<320, 2, 0, 1, 0, 2>(d, bytes, key, tail, rec_cap, d_out);
    run<320, 2, 169, 1, 0, 2>(d, bytes, key, tail, rec_cap, d_out);
    run<320, 2, 0, 1, 0, 5>(d, bytes, key, tail, rec_cap, d_out);
    run<320, 2, 169, 1, 0, 5>(d, bytes, key, tail, rec_cap, d_out);
    run<320, 2, 0, 1, 0, 1>(d, bytes, key, tail, rec_cap, d_out);
    run<320, 2, 2, 1, 0, 1>(d, bytes, key, tail, rec_cap, d_out);
    run<320, 2, 169, 1, 0, 1>(d, bytes, key, tail, rec_cap, d_out);
    run<320, 2, 169, 1, 20>(d, bytes, key, tail, rec_cap, d_out);
    run<320, 2, 169, 1, 21>(d, bytes, key, tail, rec_cap, d_out);
    run<320, 2, 169, 1, 22>(d, bytes, key, tail, rec_cap, d_out);
    run<320, 2, 169, 1, 23>(d, bytes, key, tail, rec_cap, d_out);
    run<320, 2, 169, 1, 24>(d, bytes, key, tail, rec_cap, d_out);
    run<320, 2, 169, 1, 25>(d, bytes, key, tail, rec_cap, d_out);
    run<320, 2, 169, 1, 26>(d, bytes, key, tail, rec_cap, d_out);
    run<320, 2, 2, 1, 12>(d, bytes, key, tail, rec_cap, d_out);
    run<320, 2, 169, 1, 10>(d, bytes, key, tail, rec_cap, d_out);
    run<320, 2, 169, 1, 11>(d, bytes, key, tail, rec_cap, d_out);
    run<320, 2, 169, 1, 12>(d, bytes, key, tail, rec_cap, d_out);
    run<320, 2, 169, 101, 0>(d, bytes, key, tail, rec_cap, d_out);
    run<320, 2, 169, 102, 0>(d, bytes, key, tail, rec_cap, d_out);
    run<320, 2, 169, 104, 0>(d, bytes, key, tail, rec_cap, d_out);
    run<320, 2, 169, 106, 0>(d, bytes, key, tail, rec_cap, d_out);
    run<400, 2, 169, 106, 0>(d, bytes, key, tail, rec_cap, d_out);
    run<320, 2, 169, -10, 0>(d, bytes, key, tail, rec_cap, d_out);
    run<320, 2, 169, -11, 0>(d, bytes, key, tail, rec_cap, d_out);
    run<320, 2, 169, -12, 0>(d, bytes, key, tail, rec_cap, d_out);
    run<320, 2, 169, -13, 0>(d, bytes, key, tail, rec_cap, d_out);
    run<320, 2, 169, -14, 0>(d, bytes, key, tail, rec_cap, d_out);
    run<320, 2, 169, -2, 0>(d, bytes, key, tail, rec_cap, d_out);
    run<400, 2, 169, -2, 0>(d, bytes, key, tail, rec_cap, d_out);
    run<400, 2, 169, -3, 0>(d, bytes, key, tail, rec_cap, d_out);
    run<400, 2, 169, -4, 0>(d, bytes, key, tail, rec_cap, d_out);
    run<400, 2, 256, -2, 0>(d, bytes, key, tail, rec_cap, d_out);
    return 0;
}

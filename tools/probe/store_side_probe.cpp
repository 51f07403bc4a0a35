// store_side_probe.cpp -- is the cost of a thin record stream beside a streaming
// read (stream_mix_probe.cpp) paid in the CU that stores or on the memory side?
// The same 14.4 GB read + 320 VALU per tile by 5 x 256 x 16 workgroups, and the
// same 0.93 GB of records written meanwhile by `writers` EXTRA workgroups that do
// nothing else (they loop until the readers are done, pacing themselves so that
// the bytes come out over the whole read).
//   hipcc -O3 --offload-arch=gfx950 -o store_side_probe store_side_probe.cpp
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { \
    printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
constexpr uint32_t kTile = 244 * 64;

__global__ void __launch_bounds__(256, 5)
reader(const uint8_t *src, uint64_t n_tiles, uint32_t *out) {
    __shared__ uint32_t s_x[256];
    const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint64_t begin = n_tiles * blockIdx.x / gridDim.x;
    const uint64_t end = n_tiles * (blockIdx.x + 1) / gridDim.x;
    uint32_t r[8];
    for (int i = 0; i < 8; i++)
	r[i] = tid * (i + 3);
    auto load = [&](u32x4 *v, uint64_t t) {
	auto rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t *>(src) + t * kTile, 0,
		16384 + 64, 0x00020000);
#pragma unroll
	for (int q = 0; q < 4; q++)
	    v[q] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, (63 * wave + lane) * 64 + q * 16, 0, 0);
    };
    u32x4 v[4];
    if (begin < end)
	load(v, begin);
    for (uint64_t t = begin; t < end; t++) {
	uint32_t x = 0;
#pragma unroll
	for (int q = 0; q < 4; q++)
	    x ^= v[q][0] ^ v[q][1] ^ v[q][2] ^ v[q][3];
	load(v, t + 1 < end? t + 1 : t);
#pragma unroll
	for (int k = 0; k < 40; k++) {
#pragma unroll
	    for (int i = 0; i < 8; i++)
		asm volatile("v_pk_sub_i16 %0, %0, %1 clamp" : "+v"(r[i]) : "v"(x));
	}
	s_x[tid] = x;
	for (int b = 0; b < 2; b++) {
	    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
	    x += s_x[(tid + 64 * (b + 1)) & 255];
	}
	r[1] ^= x;
    }
    uint32_t acc = 0;
    for (int i = 0; i < 8; i++)
	acc ^= r[i];
    if (acc == 0x12345678u)
	out[blockIdx.x] = acc;
}

// writes `bytes_each` bytes of keys (4 B per lane and store) over about `us` microseconds
__global__ void __launch_bounds__(256)
writer(uint32_t *dst, uint64_t words_each, uint32_t us) {
    uint32_t *mine = dst + (uint64_t) blockIdx.x * words_each;
    const uint64_t t0 = __builtin_amdgcn_s_memrealtime();	// 100 MHz
    const uint64_t steps = words_each / 256;
    for (uint64_t s = 0; s < steps; s++) {
	mine[s * 256 + threadIdx.x] = (uint32_t) s;
	// pace: step s is due at t0 + us * s / steps
	const uint64_t due = t0 + (uint64_t) us * 100 * s / steps;
	while (__builtin_amdgcn_s_memrealtime() < due)
	    __builtin_amdgcn_s_sleep(8);
    }
}

int
main() {
    const uint64_t bytes = 14400000000ull;
    const uint64_t n_tiles = (bytes - 32768) / kTile;
    uint8_t *d;
    uint32_t *d_out, *d_rec;
    CHECK(hipMalloc(&d, bytes));
    CHECK(hipMalloc(&d_out, 1 << 20));
    CHECK(hipMalloc(&d_rec, 1ull << 30));
    CHECK(hipMemset(d, 1, bytes));
    hipStream_t s_read, s_write;
    CHECK(hipStreamCreateWithFlags(&s_read, hipStreamNonBlocking));
    CHECK(hipStreamCreateWithFlags(&s_write, hipStreamNonBlocking));
    hipEvent_t e0, e1, w0, w1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    CHECK(hipEventCreate(&w0));
    CHECK(hipEventCreate(&w1));
    for (int writers : {0, 4, 8, 16, 32, 64, 128, 512}) {
	float best = 1e9f, wms = 0;
	for (int i = 0; i < 4; i++) {
	    CHECK(hipDeviceSynchronize());
	    if (writers > 0) {
		// 0.93 GB in all, spread over ~2.3 ms; launched first so that its few
		// workgroups are resident when the readers fill the rest of the chip
		const uint64_t words_each = 930000000ull / 4 / writers / 256 * 256;
		CHECK(hipEventRecord(w0, s_write));
		hipLaunchKernelGGL(writer, dim3(writers), dim3(256), 0, s_write, d_rec, words_each, 2300u);
		CHECK(hipEventRecord(w1, s_write));
	    }
	    CHECK(hipEventRecord(e0, s_read));
	    hipLaunchKernelGGL(reader, dim3(256 * 5 * 16), dim3(256), 0, s_read, d, n_tiles, d_out);
	    CHECK(hipEventRecord(e1, s_read));
	    CHECK(hipEventSynchronize(e1));
	    float ms;
	    CHECK(hipEventElapsedTime(&ms, e0, e1));
	    if (i > 0 && ms < best)
		best = ms;
	    if (writers > 0) {
		CHECK(hipEventSynchronize(w1));
		CHECK(hipEventElapsedTime(&wms, w0, w1));
	    }
	}
	printf("reader alone on its CUs, %3d writer workgroups beside it: %.3f ms  %.2f TB/s read "
		"(the writers took %.3f ms)\n",
		writers, best, n_tiles * (double) kTile / best / 1e9, wms);
    }
    return 0;
}

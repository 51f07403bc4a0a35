// read_pattern_probe.cpp -- what a streaming read of 14.4 GB reaches on this GPU
// with the scan kernel's load pattern and with others, ALU work aside.
//   hipcc -O3 --offload-arch=gfx950 -o read_pattern_probe read_pattern_probe.cpp
// Patterns (a 256-thread workgroup reads `tile` bytes per pass, four 16-byte
// loads per lane):
//   0 lane-run:   lane l reads bytes 64 l + 16 r, r = 0..3 (its own 64 bytes);
//                 every 128-byte line is touched by four load instructions
//   1 coalesced:  load r of a wave reads 1 KB contiguous (lane l: 1024 r + 16 l)
//   2 lane-run with the scan kernel's 63-run wave stride (halo lanes)
// Each with the loads consumed at once, or issued one pass ahead (prefetch).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { \
    printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

template <int kPattern, bool kPrefetch, int kWavesPerSimd>
__global__ void __launch_bounds__(256, kWavesPerSimd)
reader(const uint8_t *src, uint64_t n_tiles, uint32_t *out) {
    const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint64_t begin = n_tiles * blockIdx.x / gridDim.x;
    const uint64_t end = n_tiles * (blockIdx.x + 1) / gridDim.x;
    constexpr uint32_t kTile = kPattern == 2? 252 * 64 - 8 * 64 : 16384;   // bytes owned per pass
    uint32_t acc = 0;
    auto load = [&](u32x4 *v, uint64_t t) {
	auto rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t *>(src) + t * kTile, 0,
		16384 + 64, 0x00020000);
#pragma unroll
	for (int r = 0; r < 4; r++) {
	    uint32_t off;
	    if (kPattern == 0)
		off = tid * 64 + r * 16;
	    else if (kPattern == 1)
		off = wave * 4096 + r * 1024 + lane * 16;
	    else
		off = (63 * wave + lane) * 64 + r * 16;
	    v[r] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, off, 0, 0);
	}
    };
    auto use = [&](const u32x4 *v) {
#pragma unroll
	for (int r = 0; r < 4; r++)
	    acc += v[r][0] ^ v[r][1] ^ v[r][2] ^ v[r][3];
    };
    u32x4 v[4];
    if (kPrefetch) {
	if (begin < end)
	    load(v, begin);
	for (uint64_t t = begin; t < end; t++) {
	    u32x4 w[4] = {v[0], v[1], v[2], v[3]};
	    asm volatile("" : "+v"(w[0]), "+v"(w[1]), "+v"(w[2]), "+v"(w[3]));
	    if (t + 1 < end)
		load(v, t + 1);
	    use(w);
	}
    } else {
	for (uint64_t t = begin; t < end; t++) {
	    load(v, t);
	    use(v);
	}
    }
    if (acc == 0x12345678u)
	out[blockIdx.x] = acc;
}

template <int kPattern, bool kPrefetch, int kW> int
run(const char *name, const uint8_t *d, uint64_t bytes, uint32_t *d_out, int rounds) {
    constexpr uint32_t kTile = kPattern == 2? 252 * 64 - 8 * 64 : 16384;
    const uint64_t n_tiles = (bytes - 32768) / kTile;
    const int blocks = 256 * kW * rounds;
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    float best = 1e9f;
    for (int i = 0; i < 6; i++) {
	CHECK(hipEventRecord(e0));
	hipLaunchKernelGGL((reader<kPattern, kPrefetch, kW>), dim3(blocks), dim3(256), 0, 0, d,
		n_tiles, d_out);
	CHECK(hipEventRecord(e1));
	CHECK(hipEventSynchronize(e1));
	float ms;
	CHECK(hipEventElapsedTime(&ms, e0, e1));
	if (i > 0 && ms < best)
	    best = ms;
    }
    printf("%-28s w=%d rounds=%2d  %.3f ms  %.2f TB/s\n", name, kW, rounds, best,
	    n_tiles * (double) kTile / best / 1e9);
    return 0;
}

int
main() {
    const uint64_t bytes = 14400000000ull;
    uint8_t *d;
    uint32_t *d_out;
    CHECK(hipMalloc(&d, bytes));
    CHECK(hipMalloc(&d_out, 1 << 20));
    CHECK(hipMemset(d, 1, bytes));
    for (int rounds : {1, 16}) {
	run<0, false, 7>("lane-run", d, bytes, d_out, rounds);
	run<0, false, 8>("lane-run", d, bytes, d_out, rounds);
	run<0, true, 5>("lane-run prefetch", d, bytes, d_out, rounds);
	run<0, true, 8>("lane-run prefetch", d, bytes, d_out, rounds);
	run<1, false, 7>("coalesced", d, bytes, d_out, rounds);
	run<1, false, 8>("coalesced", d, bytes, d_out, rounds);
	run<1, true, 5>("coalesced prefetch", d, bytes, d_out, rounds);
	run<1, true, 8>("coalesced prefetch", d, bytes, d_out, rounds);
	run<2, false, 7>("lane-run 63-stride", d, bytes, d_out, rounds);
	run<2, true, 5>("lane-run 63-stride prefetch", d, bytes, d_out, rounds);
    }
    return 0;
}

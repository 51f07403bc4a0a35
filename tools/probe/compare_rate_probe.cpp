// compare_rate_probe.cpp -- the lane compare of adsb_tile.h alone, on registers,
// at the scan kernel's occupancy: is the scan bound by these instructions?
//   hipcc -O3 --offload-arch=gfx950 -I sora_amd/csrc -o compare_rate_probe compare_rate_probe.cpp
//   (-DOLD_TILE_H=\"path\" times another version of the header instead)
#include <hip/hip_runtime.h>
#include <cstdio>
#ifdef OLD_TILE_H
#include OLD_TILE_H
#else
#include "adsb_tile.h"
#endif

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { \
    printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

template <int kWaves>
__global__ void __launch_bounds__(256, kWaves)
probe(uint32_t *out, int iters, uint32_t seed) {
    uint32_t P[SORA_LANE_PAIRS + 4];
    for (int i = 0; i < SORA_LANE_PAIRS + 4; i++)
	P[i] = sora_mag2_pack8((seed + threadIdx.x) * (2654435761u + 40503u * i) + blockIdx.x, 0x80808080u);
    uint32_t acc = 0;
    for (int it = 0; it < iters; it++) {
	uint32_t up, down, cc;
	sora_lane_compare(P, &up, &down, &cc);
	acc ^= up + 3 * down + 5 * cc;
	// new data for the next round, one cheap operation per register
#pragma unroll
	for (int i = 0; i < SORA_LANE_PAIRS + 4; i++)
	    P[i] ^= (acc >> (i & 7)) & 0x00030003u;
    }
    out[blockIdx.x * 256 + threadIdx.x] = acc;
}

int
main() {
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount, iters = 4000;
    uint32_t *d_out;
    CHECK(hipMalloc(&d_out, (size_t) cus * 8 * 256 * 4));
    for (int w : {1, 3, 5, 7}) {
	const int blocks = cus * w;
	for (int rep = 0; rep < 2; rep++) {
	    hipEvent_t e0, e1;
	    CHECK(hipEventCreate(&e0));
	    CHECK(hipEventCreate(&e1));
	    CHECK(hipEventRecord(e0));
	    if (w == 1) hipLaunchKernelGGL(probe<1>, dim3(blocks), dim3(256), 0, 0, d_out, iters, 7u);
	    if (w == 3) hipLaunchKernelGGL(probe<3>, dim3(blocks), dim3(256), 0, 0, d_out, iters, 7u);
	    if (w == 5) hipLaunchKernelGGL(probe<5>, dim3(blocks), dim3(256), 0, 0, d_out, iters, 7u);
	    if (w == 7) hipLaunchKernelGGL(probe<7>, dim3(blocks), dim3(256), 0, 0, d_out, iters, 7u);
	    CHECK(hipEventRecord(e1));
	    CHECK(hipDeviceSynchronize());
	    float ms;
	    CHECK(hipEventElapsedTime(&ms, e0, e1));
	    if (rep == 1) {
		const double runs = (double) blocks * 256 * iters;
		printf("waves/SIMD %d: %.3f ms, %.1f G lane runs/s, %.2f ns per lane run and SIMD; "
			"the 1-hour capture has 0.229 G lane runs\n", w, ms, runs / ms / 1e6,
			ms * 1e6 / (runs / 64 / (cus * 4)) );
	    }
	}
    }
    return 0;
}

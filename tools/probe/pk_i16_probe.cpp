/* Probe: do v_pk_sub_i16 / v_pk_mad_i16 with clamp saturate the EXACT result on
 * gfx950?  (The biased 16-bit compare of adsb_tile.h relies on it.)
 *   hipcc --offload-arch=gfx950 -o pk_i16_probe pk_i16_probe.cpp && ./pk_i16_probe */
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>

__global__ void probe(uint32_t *out_mad, uint32_t *out_sub, const uint32_t *a, const uint32_t *b, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n)
	return;
    uint32_t w = a[i], x, d;
    asm volatile("v_pk_mad_i16 %0, %1, %2, 0 clamp" : "=v"(x) : "v"(w), "s"(0x80008000u));
    asm volatile("v_pk_sub_i16 %0, %1, %2 clamp" : "=v"(d) : "v"(a[i]), "v"(b[i]));
    out_mad[i] = x;
    out_sub[i] = d;
}

static int16_t sat16(int64_t v) { return v > 32767? 32767 : v < -32768? -32768 : (int16_t) v; }

int main() {
    const int n = 65536;
    uint32_t *ha = new uint32_t[n], *hb = new uint32_t[n], *hm = new uint32_t[n], *hs = new uint32_t[n];
    for (int i = 0; i < n; i++) {
	ha[i] = (uint32_t) i | ((uint32_t) ((i * 40503u) & 0xffff) << 16);
	hb[i] = (uint32_t) ((i * 25173u + 13849u) & 0xffff) | ((uint32_t) ((i * 7u) & 0xffff) << 16);
    }
    uint32_t *da, *db, *dm, *ds;
    hipMalloc(&da, n * 4); hipMalloc(&db, n * 4); hipMalloc(&dm, n * 4); hipMalloc(&ds, n * 4);
    hipMemcpy(da, ha, n * 4, hipMemcpyHostToDevice);
    hipMemcpy(db, hb, n * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(probe, dim3(n / 256), dim3(256), 0, 0, dm, ds, da, db, n);
    if (hipDeviceSynchronize() != hipSuccess) { printf("kernel failed\n"); return 2; }
    hipMemcpy(hm, dm, n * 4, hipMemcpyDeviceToHost);
    hipMemcpy(hs, ds, n * 4, hipMemcpyDeviceToHost);
    int bad_mad = 0, bad_sub = 0;
    for (int i = 0; i < n; i++)
	for (int h = 0; h < 2; h++) {
	    int16_t a = (int16_t) (ha[i] >> (16 * h)), b = (int16_t) (hb[i] >> (16 * h));
	    int16_t want_m = sat16((int64_t) a * -32768), want_s = sat16((int64_t) a - b);
	    int16_t got_m = (int16_t) (hm[i] >> (16 * h)), got_s = (int16_t) (hs[i] >> (16 * h));
	    if (got_m != want_m && bad_mad++ < 5)
		printf("mad: w=%d got %d want %d\n", a, got_m, want_m);
	    if (got_s != want_s && bad_sub++ < 5)
		printf("sub: %d - %d got %d want %d\n", a, b, got_s, want_s);
	}
    printf("pk_mad_i16 clamp: %d mismatches; pk_sub_i16 clamp: %d mismatches (of %d)\n", bad_mad, bad_sub, 2 * n);
    return bad_mad || bad_sub;
}

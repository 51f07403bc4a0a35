// valu_rate_probe.cpp -- cycles per wave64 VALU instruction on one gfx950 SIMD,
// for the instructions the ADS-B scan kernel is made of, at 1..8 waves per SIMD.
//   hipcc -O2 --offload-arch=gfx950 -o valu_rate_probe valu_rate_probe.cpp && ./valu_rate_probe
// Each wave runs kIter x 64 instructions of one kind on 8 independent register
// chains and stamps s_memtime around them; reported: the median over waves of
// (cycles of the wave) / (instructions of ALL waves of its SIMD), i.e. the
// SIMD's cycles per instruction when `w` waves share it.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { \
    printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

constexpr int kIter = 2000;

#define REP8(op) op(0) op(1) op(2) op(3) op(4) op(5) op(6) op(7)
#define BODY(op) REP8(op) REP8(op) REP8(op) REP8(op) REP8(op) REP8(op) REP8(op) REP8(op)

template <int kKind>
__global__ void __launch_bounds__(256)
probe(uint32_t *out, uint64_t *cycles, uint32_t seed) {
    uint32_t r[8], a = seed + threadIdx.x, b = seed * 3 + 1, c = 0x00010001u << (threadIdx.x & 15);
    for (int i = 0; i < 8; i++)
	r[i] = a * (i + 1);
    asm volatile("s_mov_b64 s[40:41], 0x5555\n\ts_mov_b64 vcc, 0x3333" ::: "s40", "s41", "vcc");
    __builtin_amdgcn_s_barrier();
    const uint64_t t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < kIter; it++) {
#define OP_ADD(i)   asm volatile("v_add_u32 %0, %0, %1" : "+v"(r[i]) : "v"(a));
#define OP_PKSUB(i) asm volatile("v_pk_sub_i16 %0, %0, %1 clamp" : "+v"(r[i]) : "v"(a));
#define OP_PKMAD(i) asm volatile("v_pk_mad_i16 %0, %0, %1, 0 clamp" : "+v"(r[i]) : "s"(0x80008000u));
#define OP_PKMAX(i) asm volatile("v_pk_max_i16 %0, %0, %1" : "+v"(r[i]) : "v"(a));
#define OP_ANDOR(i) asm volatile("v_and_or_b32 %0, %1, %2, %0" : "+v"(r[i]) : "v"(a), "s"(0x00100010u));
#define OP_ALIGN(i) asm volatile("v_alignbit_b32 %0, %0, %1, 16" : "+v"(r[i]) : "v"(a));
#define OP_PERM(i)  asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(r[i]) : "v"(a), "s"(0x05040100u));
#define OP_DOT4(i)  asm volatile("v_dot4_i32_i8 %0, %1, %1, %0" : "+v"(r[i]) : "v"(a));
#define OP_BITOP(i) asm volatile("v_bitop3_b32 %0, %0, %1, %2 bitop3:0x28" : "+v"(r[i]) : "v"(a), "v"(b));
#define OP_XOR(i)   asm volatile("v_xor_b32 %0, %0, %1" : "+v"(r[i]) : "v"(a));
#define OP_DPP(i)   asm volatile("v_mov_b32_dpp %0, %1 wave_shl:1 row_mask:0xf bank_mask:0xf" : "+v"(r[i]) : "v"(a));
#define OP_LSHLOR(i) asm volatile("v_lshl_or_b32 %0, %0, 1, %1" : "+v"(r[i]) : "v"(a));
#define OP_BFE(i)   asm volatile("v_bfe_u32 %0, %0, 3, 8" : "+v"(r[i]));
#define OP_ADDDPP(i) asm volatile("v_add_u32_dpp %0, %1, %0 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0" : "+v"(r[i]) : "v"(a));
#define OP_CMP(i)   asm volatile("v_cmp_lt_u32 vcc, %0, %1" :: "v"(r[i]), "v"(a) : "vcc");
#define OP_CMPMOV(i) asm volatile("v_cmp_lt_u32 vcc, %0, %1\n\ts_mov_b64 s[40:41], vcc" :: "v"(r[i]), "v"(a) : "vcc", "s40", "s41");
#define OP_CMPAND(i) asm volatile("v_cmp_lt_u32 vcc, %0, %1\n\ts_and_b64 s[40:41], s[40:41], vcc" :: "v"(r[i]), "v"(a) : "vcc", "s40", "s41", "scc");
#define OP_CMP64(i) asm volatile("v_cmp_lt_u32_e64 s[40:41], %0, %1" :: "v"(r[i]), "v"(a) : "s40", "s41");
#define OP_CMPADDC(i) asm volatile("v_cmp_lt_u32 vcc, %1, %2\n\tv_addc_co_u32 %0, vcc, %0, %0, vcc" : "+v"(r[i]) : "v"(b), "v"(a) : "vcc");
#define OP_ADDC(i)  asm volatile("v_addc_co_u32 %0, vcc, %0, %0, vcc" : "+v"(r[i]) :: "vcc");
#define OP_MAX3(i)  asm volatile("v_max3_u32 %0, %0, %1, %2" : "+v"(r[i]) : "v"(a), "v"(b));
#define OP_MAX(i)   asm volatile("v_max_u32 %0, %0, %1" : "+v"(r[i]) : "v"(a));
#define OP_ANDS(i)  asm volatile("v_and_b32 %0, %1, %0" : "+v"(r[i]) : "s"(0xffff0000u));
#define OP_NOT(i)   asm volatile("v_not_b32 %0, %0" : "+v"(r[i]));
#define OP_CNDMASK(i) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(r[i]) : "v"(a) : "vcc");
#define OP_DOT4U(i) asm volatile("v_dot4_u32_u8 %0, %1, %1, %0" : "+v"(r[i]) : "v"(a));
#define OP_SUB(i)   asm volatile("v_sub_u32 %0, %0, %1" : "+v"(r[i]) : "v"(a));
#define OP_LSHR(i)  asm volatile("v_lshrrev_b32 %0, 1, %0" : "+v"(r[i]));
#define OP_CMPSALU4(i) asm volatile("v_cmp_lt_u32 vcc, %0, %1\n\ts_and_b64 s[40:41], s[42:43], vcc\n\ts_andn2_b64 s[44:45], s[40:41], s[46:47]\n\ts_lshr_b64 s[42:43], s[44:45], 1\n\ts_mov_b64 s[46:47], vcc" :: "v"(r[i]), "v"(a) : "vcc", "s40", "s41", "s42", "s43", "s44", "s45", "s46", "s47", "scc");
#define OP_CND64(i) asm volatile("v_cndmask_b32_e64 %0, %0, %1, s[40:41]" : "+v"(r[i]) : "v"(a) : "s40", "s41");
#define OP_CMPCND(i) asm volatile("v_cmp_lt_u32 vcc, %0, %1\n\tv_cndmask_b32 %0, %0, %1, vcc" : "+v"(r[i]) : "v"(a) : "vcc");
#define OP_CNDMIX(i) asm volatile("v_cndmask_b32 %0, %0, %1, vcc\n\tv_xor_b32 %0, %0, %1\n\tv_add_u32 %0, %0, %1\n\tv_xor_b32 %0, %0, %1" : "+v"(r[i]) : "v"(a) : "vcc");
#define OP_CNDK(i) asm volatile("v_cndmask_b32 %0, 0, %1, vcc" : "+v"(r[i]) : "v"(a) : "vcc");
	if constexpr (kKind == 0) { BODY(OP_ADD) }
	if constexpr (kKind == 40) { BODY(OP_CND64) }
	if constexpr (kKind == 41) { BODY(OP_CMPCND) }
	if constexpr (kKind == 42) { BODY(OP_CNDMIX) }
	if constexpr (kKind == 43) { BODY(OP_CNDK) }
	if constexpr (kKind == 20) { BODY(OP_CMP) }
	if constexpr (kKind == 21) { BODY(OP_CMPMOV) }
	if constexpr (kKind == 22) { BODY(OP_CMPAND) }
	if constexpr (kKind == 23) { BODY(OP_CMP64) }
	if constexpr (kKind == 24) { BODY(OP_CMPADDC) }
	if constexpr (kKind == 25) { BODY(OP_ADDC) }
	if constexpr (kKind == 26) { BODY(OP_MAX3) }
	if constexpr (kKind == 27) { BODY(OP_MAX) }
	if constexpr (kKind == 28) { BODY(OP_ANDS) }
	if constexpr (kKind == 29) { BODY(OP_NOT) }
	if constexpr (kKind == 30) { BODY(OP_CNDMASK) }
	if constexpr (kKind == 31) { BODY(OP_DOT4U) }
	if constexpr (kKind == 32) { BODY(OP_SUB) }
	if constexpr (kKind == 33) { BODY(OP_LSHR) }
	if constexpr (kKind == 34) { BODY(OP_CMPSALU4) }
	if constexpr (kKind == 1) { BODY(OP_PKSUB) }
	if constexpr (kKind == 2) { BODY(OP_PKMAD) }
	if constexpr (kKind == 3) { BODY(OP_PKMAX) }
	if constexpr (kKind == 4) { BODY(OP_ANDOR) }
	if constexpr (kKind == 5) { BODY(OP_ALIGN) }
	if constexpr (kKind == 6) { BODY(OP_PERM) }
	if constexpr (kKind == 7) { BODY(OP_DOT4) }
	if constexpr (kKind == 8) { BODY(OP_BITOP) }
	if constexpr (kKind == 9) { BODY(OP_XOR) }
	if constexpr (kKind == 10) { BODY(OP_DPP) }
	if constexpr (kKind == 11) { BODY(OP_LSHLOR) }
	if constexpr (kKind == 12) { BODY(OP_BFE) }
	if constexpr (kKind == 13) { BODY(OP_ADDDPP) }
    }
    const uint64_t t1 = __builtin_amdgcn_s_memtime();
    uint32_t s = c;
    for (int i = 0; i < 8; i++)
	s ^= r[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0)
	cycles[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
}

template <int kKind> int
run(const char *name, uint32_t *d_out, uint64_t *d_cyc, int cus) {
    printf("%-16s", name);
    for (int w : {1, 2, 3, 4, 5, 6, 7, 8}) {
	// w blocks of 4 waves per CU: one wave of each block on every SIMD
	const int blocks = cus * w;
	hipLaunchKernelGGL(probe<kKind>, dim3(blocks), dim3(256), 0, 0, d_out, d_cyc, 12345u);
	CHECK(hipDeviceSynchronize());
	hipEvent_t e0, e1;
	CHECK(hipEventCreate(&e0));
	CHECK(hipEventCreate(&e1));
	CHECK(hipEventRecord(e0));
	hipLaunchKernelGGL(probe<kKind>, dim3(blocks), dim3(256), 0, 0, d_out, d_cyc, 12345u);
	CHECK(hipEventRecord(e1));
	CHECK(hipDeviceSynchronize());
	float ms;
	CHECK(hipEventElapsedTime(&ms, e0, e1));
	std::vector<uint64_t> c(blocks * 4);
	CHECK(hipMemcpy(c.data(), d_cyc, c.size() * 8, hipMemcpyDeviceToHost));
	std::sort(c.begin(), c.end());
	const double per = (double) c[c.size() / 2] / ((double) kIter * 64 * w);
	// wall clock: wave-instructions per ns over the chip (1024 SIMDs)
	printf("  w=%d %5.2f (%4.0f G/s)", w, per, (double) kIter * 64 * 4 * blocks / (ms * 1e6));
    }
    printf("\n");
    return 0;
}

int
main() {
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    uint32_t *d_out;
    uint64_t *d_cyc;
    CHECK(hipMalloc(&d_out, (size_t) cus * 8 * 256 * 4));
    CHECK(hipMalloc(&d_cyc, (size_t) cus * 8 * 4 * 8));
    printf("%s, %d CUs; SIMD cycles per wave64 instruction with w waves per SIMD\n",
	    prop.gcnArchName, cus);
    run<0>("v_add_u32", d_out, d_cyc, cus);
    run<9>("v_xor_b32", d_out, d_cyc, cus);
    run<1>("v_pk_sub_i16 cl", d_out, d_cyc, cus);
    run<2>("v_pk_mad_i16 cl", d_out, d_cyc, cus);
    run<3>("v_pk_max_i16", d_out, d_cyc, cus);
    run<4>("v_and_or_b32", d_out, d_cyc, cus);
    run<5>("v_alignbit_b32", d_out, d_cyc, cus);
    run<6>("v_perm_b32", d_out, d_cyc, cus);
    run<7>("v_dot4_i32_i8", d_out, d_cyc, cus);
    run<8>("v_bitop3_b32", d_out, d_cyc, cus);
    run<10>("v_mov_b32_dpp", d_out, d_cyc, cus);
    run<11>("v_lshl_or_b32", d_out, d_cyc, cus);
    run<12>("v_bfe_u32", d_out, d_cyc, cus);
    run<13>("v_add_u32_dpp", d_out, d_cyc, cus);
    printf("-- per OP below: the instruction group named (2 = VALU + VALU, +s = with SALU)\n");
    run<20>("v_cmp e32", d_out, d_cyc, cus);
    run<21>("v_cmp +s_mov", d_out, d_cyc, cus);
    run<22>("v_cmp +s_and", d_out, d_cyc, cus);
    run<34>("v_cmp +4 salu", d_out, d_cyc, cus);
    run<23>("v_cmp e64 sgpr", d_out, d_cyc, cus);
    run<24>("v_cmp,v_addc 2", d_out, d_cyc, cus);
    run<25>("v_addc e32", d_out, d_cyc, cus);
    run<26>("v_max3_u32", d_out, d_cyc, cus);
    run<27>("v_max_u32", d_out, d_cyc, cus);
    run<28>("v_and_b32 s,v", d_out, d_cyc, cus);
    run<29>("v_not_b32", d_out, d_cyc, cus);
    run<30>("v_cndmask e32", d_out, d_cyc, cus);
    run<40>("v_cndmask e64 s", d_out, d_cyc, cus);
    run<41>("v_cmp,v_cndmask2", d_out, d_cyc, cus);
    run<42>("cndmask+3 fast 4", d_out, d_cyc, cus);
    run<43>("v_cndmask 0,v", d_out, d_cyc, cus);
    run<31>("v_dot4_u32_u8", d_out, d_cyc, cus);
    run<32>("v_sub_u32", d_out, d_cyc, cus);
    run<33>("v_lshrrev_b32", d_out, d_cyc, cus);
    return 0;
}

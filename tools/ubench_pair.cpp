// tools/ubench_pair.cpp — do two VALU instruction classes overlap on a gfx950 SIMD?  Every test interleaves 4 instructions of kind A
// with 4 of kind B (all 8 on different destination registers, no dependence between them), 8 waves per SIMD, and prints the time
// per PAIR next to the times of 8 x A and 8 x B alone: pair ~ A + B means one issue port, pair ~ max(A, B) means two pipes.
// Build: hipcc --offload-arch=gfx950 -O2 tools/ubench_pair.cpp -o tools/ubench_pair.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

#define REP4(x) x x x x
#define REP16(x) REP4(REP4(x))
#define OPS : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7) : "v"(b0), "v"(b1), "v"(i2), "v"(i3)
// %8 = b0 (float), %9 = b1 (float), %10 = i2 (int / packed bytes), %11 = i3
#define A_FMA(d)   "v_fma_f32 " #d ", %8, %9, " #d "\n"
#define A_MUL(d)   "v_mul_f32 " #d ", %8, " #d "\n"
#define A_ADDU(d)  "v_add_u32 " #d ", %10, " #d "\n"
#define B_MIX(d)   "v_fma_mix_f32 " #d ", %10, %8, %9 op_sel_hi:[1,0,0]\n"
#define B_CVT(d)   "v_cvt_f32_ubyte1 " #d ", %10\n"
#define B_MAD(d)   "v_mad_i32_i24 " #d ", %10, %11, %10\n"
#define B_MED(d)   "v_med3_i32 " #d ", %10, 0, %11\n"
#define B_PK(d)    "v_cvt_pk_u8_f32 " #d ", %8, 1, %10\n"
#define B_PERM(d)  "v_perm_b32 " #d ", %10, %11, %11\n"
#define B_SDWA(d)  "v_cvt_f32_u32_sdwa " #d ", %10 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1\n"
#define B_ADDS(d)  "v_add_u32_sdwa " #d ", %10, %11 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_1\n"
#define B_ASHR(d)  "v_ashr_pk_u8_i32 " #d ", %10, %11, 16\n"
#define B_MUL24(d) "v_mul_i32_i24 " #d ", %10, %11\n"
#define ALT(A, B) asm volatile(A(%0) B(%1) A(%2) B(%3) A(%4) B(%5) A(%6) B(%7) OPS);
#define ALL(A)    asm volatile(A(%0) A(%1) A(%2) A(%3) A(%4) A(%5) A(%6) A(%7) OPS);
// 2 x A per B (the row loops' ratio is about 4 F : 5 S)
#define ALT21(A, B) asm volatile(A(%0) A(%1) B(%2) A(%3) A(%4) B(%5) A(%6) B(%7) OPS);

template <int T>
__global__ __launch_bounds__(256) void bench(uint32_t *out, int iters, float seed) {
    float b0 = seed * 0.999f, b1 = seed * 1e-3f;
    uint32_t i2 = 0x00550033u + threadIdx.x, i3 = 0x00FFFFFFu;
    uint32_t r0 = 1, r1 = 2, r2 = 3, r3 = 4, r4 = 5, r5 = 6, r6 = 7, r7 = 8;
    for (int it = 0; it < iters; it++) {
        if (T == 0) { REP16(ALL(A_FMA)) }
        if (T == 1) { REP16(ALL(B_MIX)) }   if (T == 2) { REP16(ALT(A_FMA, B_MIX)) }
        if (T == 3) { REP16(ALL(B_CVT)) }   if (T == 4) { REP16(ALT(A_FMA, B_CVT)) }
        if (T == 5) { REP16(ALL(B_MAD)) }   if (T == 6) { REP16(ALT(A_FMA, B_MAD)) }
        if (T == 7) { REP16(ALL(B_MED)) }   if (T == 8) { REP16(ALT(A_FMA, B_MED)) }
        if (T == 9) { REP16(ALL(B_PK)) }    if (T == 10) { REP16(ALT(A_FMA, B_PK)) }
        if (T == 11) { REP16(ALL(B_PERM)) } if (T == 12) { REP16(ALT(A_FMA, B_PERM)) }
        if (T == 13) { REP16(ALL(B_SDWA)) } if (T == 14) { REP16(ALT(A_FMA, B_SDWA)) }
        if (T == 15) { REP16(ALL(B_ADDS)) } if (T == 16) { REP16(ALT(A_MUL, B_ADDS)) }
        if (T == 17) { REP16(ALT(B_MIX, B_CVT)) }
        if (T == 18) { REP16(ALT(B_MAD, B_CVT)) }
        if (T == 19) { REP16(ALT(B_MIX, B_MAD)) }
        if (T == 20) { REP16(ALT21(A_FMA, B_MIX)) }
        if (T == 21) { REP16(ALT21(A_FMA, B_MAD)) }
        if (T == 22) { REP16(ALL(B_ASHR)) } if (T == 23) { REP16(ALT(A_FMA, B_ASHR)) }
        if (T == 24) { REP16(ALL(A_MUL)) }  if (T == 25) { REP16(ALT(A_MUL, B_MIX)) }
        if (T == 26) { REP16(ALL(A_ADDU)) } if (T == 27) { REP16(ALT(A_ADDU, B_MIX)) }
        if (T == 28) { REP16(ALL(B_MUL24)) } if (T == 29) { REP16(ALT(A_FMA, B_MUL24)) }
    }
    if ((r0 ^ r1 ^ r2 ^ r3 ^ r4 ^ r5 ^ r6 ^ r7) == 0x12345678u) out[0] = 1;
}

template <int T>
double run(uint32_t *d_out, int waves_per_simd = 8) {
    const int iters = 100;
    dim3 block(256), grid(256 * waves_per_simd);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(bench<T>, grid, block, 0, 0, d_out, iters, 1.5f);
    (void)hipDeviceSynchronize();
    float best = 1e30f;
    for (int rep = 0; rep < 3; rep++) {
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL(bench<T>, grid, block, 0, 0, d_out, iters, 1.5f);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        best = ms < best ? ms : best;
    }
    return best * 1e6 / ((double)iters * 16.0 * 8 * waves_per_simd);      // ns per wave-instruction per SIMD
}
#define PAIR(TA, TB, TP, NAME) { double a = run<TA>(d_out), b = run<TB>(d_out), p = run<TP>(d_out); \
    printf("%-34s A %.3f  B %.3f  ns/instr alone;  interleaved %.3f ns/instr  = pair %.2f ns vs A + B %.2f, max %.2f\n", NAME, a, b, p, 2 * p, a + b, a > b ? a : b); fflush(stdout); }
int main() {
    uint32_t *d_out; (void)hipMalloc(&d_out, 1024);
    PAIR(0, 1, 2, "v_fma_f32 | v_fma_mix_f32")
    PAIR(0, 3, 4, "v_fma_f32 | v_cvt_f32_ubyte1")
    PAIR(0, 5, 6, "v_fma_f32 | v_mad_i32_i24")
    PAIR(0, 28, 29, "v_fma_f32 | v_mul_i32_i24")
    PAIR(0, 7, 8, "v_fma_f32 | v_med3_i32")
    PAIR(0, 9, 10, "v_fma_f32 | v_cvt_pk_u8_f32")
    PAIR(0, 11, 12, "v_fma_f32 | v_perm_b32")
    PAIR(0, 13, 14, "v_fma_f32 | v_cvt_f32_u32_sdwa")
    PAIR(24, 15, 16, "v_mul_f32 | v_add_u32_sdwa")
    PAIR(24, 1, 25, "v_mul_f32 | v_fma_mix_f32")
    PAIR(26, 1, 27, "v_add_u32 | v_fma_mix_f32")
    PAIR(0, 22, 23, "v_fma_f32 | v_ashr_pk_u8_i32")
    PAIR(1, 3, 17, "v_fma_mix_f32 | v_cvt_f32_ubyte1")
    PAIR(5, 3, 18, "v_mad_i32_i24 | v_cvt_f32_ubyte1")
    PAIR(1, 5, 19, "v_fma_mix_f32 | v_mad_i32_i24")
    { double p = run<20>(d_out); printf("2 x v_fma_f32 per v_fma_mix_f32: %.3f ns/instr = triple %.2f ns\n", p, 3 * p); }
    { double p = run<21>(d_out); printf("2 x v_fma_f32 per v_mad_i32_i24: %.3f ns/instr = triple %.2f ns\n", p, 3 * p); }
    return 0;
}

// tools/ubench_power.cpp — SUSTAINED rate, shader clock and socket power of one vector instruction class at a time (seconds, not microseconds:
// the governor settles): what an instruction costs when the 1400 W cap, not the issue port, is the limit (profiles/r05_notes.md section 10).
// Usage: ubench_power.bin <class index> <seconds>; prints ns per wave-instruction per SIMD; sample rocm-smi beside it (tools/power_classes.sh).
// The kernel and the instruction macros are those of tools/ubench_tput.cpp:
// THROUGHPUT (not dependent-chain latency) of the VALU instructions the pixel kernels lean on, gfx950:
// 8 independent destination registers per wave, 1 / 2 / 4 / 8 waves per SIMD, every CU busy.  ubench_valu.cpp has two dependent
// chains per wave and under-states what the row loops of the wave kernels sustain (profiles/r02_notes.md, end of section 7).
// Build: hipcc --offload-arch=gfx950 -O2 tools/ubench_power.cpp -o tools/ubench_power.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

#define REP4(x) x x x x
#define REP16(x) REP4(REP4(x))

// one "group" = 8 instructions on 8 different destination registers
#define G8(INS) asm volatile(INS(%0) "\n" INS(%1) "\n" INS(%2) "\n" INS(%3) "\n" INS(%4) "\n" INS(%5) "\n" INS(%6) "\n" INS(%7) \
    : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7) : "v"(b0), "v"(b1), "v"(i2));

#define STR(x) #x
#define I_FMA(d)      "v_fma_f32 " #d ", " #d ", %8, %9"
#define I_MUL(d)      "v_mul_f32 " #d ", " #d ", %8"
#define I_ADD(d)      "v_add_f32 " #d ", " #d ", %8"
#define I_CVTUB(d)    "v_cvt_f32_ubyte1 " #d ", %10"
#define I_CVTUBS(d)   "v_cvt_f32_ubyte2 " #d ", " #d
#define I_FMAMIX(d)   "v_fma_mix_f32 " #d ", %10, %8, " #d " op_sel_hi:[1,0,0]"
#define I_FMAMIXH(d)  "v_fma_mix_f32 " #d ", %10, %8, " #d " op_sel:[1,0,0] op_sel_hi:[1,0,0]"
#define I_MULSDWA(d)  "v_mul_f32_sdwa " #d ", %10, %8 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:DWORD"
#define I_ADDSDWAD(d) "v_add_f32_sdwa " #d ", %8, %9 dst_sel:BYTE_1 dst_unused:UNUSED_PRESERVE src0_sel:DWORD src1_sel:DWORD"
#define I_MAD24(d)    "v_mad_i32_i24 " #d ", " #d ", %10, %10"
#define I_MUL24(d)    "v_mul_i32_i24 " #d ", " #d ", %10"
#define I_CVTPK(d)    "v_cvt_pk_u8_f32 " #d ", %8, 1, " #d
#define I_PERM(d)     "v_perm_b32 " #d ", " #d ", %10, %10"
#define I_ASHRPK(d)   "v_ashr_pk_u8_i32 " #d ", " #d ", %10, 16"
#define I_PKFMA(d)    "v_pk_fma_f16 " #d ", " #d ", %8, %9"
#define I_AND(d)      "v_and_b32 " #d ", " #d ", %10"
#define I_ANDOR(d)    "v_and_or_b32 " #d ", " #d ", %10, %10"
#define I_BFE(d)      "v_bfe_u32 " #d ", " #d ", 8, 8"
#define I_ADDU(d)     "v_add_u32 " #d ", " #d ", %10"
#define I_ADD3(d)     "v_add3_u32 " #d ", " #d ", %10, %10"
#define I_LSHLADD(d)  "v_lshl_add_u32 " #d ", " #d ", 2, %10"
#define I_SHIFT(d)    "v_lshrrev_b32 " #d ", 8, " #d
#define I_MOV(d)      "v_mov_b32 " #d ", %10"
#define I_CNDMASK(d)  "v_cndmask_b32 " #d ", " #d ", %10, vcc"
#define I_DOT4(d)     "v_dot4_i32_i8 " #d ", %10, %10, " #d
#define I_DOT2F(d)    "v_dot2_f32_f16 " #d ", %10, %10, " #d
#define I_CVTF16(d)   "v_cvt_f32_f16 " #d ", %10"
#define I_MED3(d)     "v_med3_f32 " #d ", " #d ", %8, %9"
#define I_MAXF(d)     "v_max_f32 " #d ", " #d ", %8"
#define I_SUBREV(d)   "v_sub_f32 " #d ", %8, " #d
#define I_RNDNE(d)    "v_rndne_f32 " #d ", " #d
#define I_FMAK(d)     "v_fmac_f32 " #d ", %8, %9"
#define I_MADU24(d)   "v_mad_u32_u24 " #d ", " #d ", %10, %10"
#define I_MULLO(d)    "v_mul_lo_u32 " #d ", " #d ", %10"
#define I_SAD(d)      "v_sad_u8 " #d ", " #d ", %10, %10"
#define I_LERP(d)     "v_lerp_u8 " #d ", " #d ", %10, %10"
#define I_CVTU32(d)   "v_cvt_u32_f32 " #d ", " #d
#define I_XOR(d)      "v_xor_b32 " #d ", " #d ", %10"
#define I_LSHLOR(d)   "v_lshl_or_b32 " #d ", " #d ", 8, %10"
#define I_PKMULF32(d) "v_mul_f32 " #d ", " #d ", %8"
// mixes: 4 fast + 4 slow alternating on different registers
#define MIX_FMA_CVT asm volatile("v_fma_f32 %0, %0, %8, %9\n v_cvt_f32_ubyte1 %1, %10\n v_fma_f32 %2, %2, %8, %9\n v_cvt_f32_ubyte2 %3, %10\n" \
    "v_fma_f32 %4, %4, %8, %9\n v_cvt_f32_ubyte3 %5, %10\n v_fma_f32 %6, %6, %8, %9\n v_cvt_f32_ubyte0 %7, %10" \
    : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7) : "v"(b0), "v"(b1), "v"(i2));
#define MIX_FMA_MAD asm volatile("v_fma_f32 %0, %0, %8, %9\n v_mad_i32_i24 %1, %1, %10, %10\n v_fma_f32 %2, %2, %8, %9\n v_mad_i32_i24 %3, %3, %10, %10\n" \
    "v_fma_f32 %4, %4, %8, %9\n v_mad_i32_i24 %5, %5, %10, %10\n v_fma_f32 %6, %6, %8, %9\n v_mad_i32_i24 %7, %7, %10, %10" \
    : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7) : "v"(b0), "v"(b1), "v"(i2));
// salu next to valu: does a scalar instruction between vector ones cost vector issue slots?
#define MIX_FMA_SALU asm volatile("v_fma_f32 %0, %0, %8, %9\n s_add_u32 s20, s20, 1\n v_fma_f32 %1, %1, %8, %9\n s_add_u32 s21, s21, 1\n v_fma_f32 %2, %2, %8, %9\n s_add_u32 s20, s20, 1\n v_fma_f32 %3, %3, %8, %9\n s_add_u32 s21, s21, 1\n" \
    "v_fma_f32 %4, %4, %8, %9\n s_add_u32 s20, s20, 1\n v_fma_f32 %5, %5, %8, %9\n s_add_u32 s21, s21, 1\n v_fma_f32 %6, %6, %8, %9\n s_add_u32 s20, s20, 1\n v_fma_f32 %7, %7, %8, %9\n s_add_u32 s21, s21, 1" \
    : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7) : "v"(b0), "v"(b1), "v"(i2) : "s20", "s21");

template <int OP>
__global__ __launch_bounds__(256) void bench(uint32_t *out, int iters, float seed) {
    float b0 = seed * 0.999f, b1 = seed * 1e-3f;
    uint32_t i2 = 0x00550033u + threadIdx.x;
    uint32_t r0 = threadIdx.x + 1, r1 = r0 * 3, r2 = r0 * 5, r3 = r0 * 7, r4 = r0 * 11, r5 = r0 * 13, r6 = r0 * 17, r7 = r0 * 19;
    for (int it = 0; it < iters; it++) {
        if (OP == 0) { REP16(G8(I_FMA)) }
        if (OP == 1) { REP16(G8(I_MUL)) }
        if (OP == 2) { REP16(G8(I_ADD)) }
        if (OP == 3) { REP16(G8(I_CVTUB)) }
        if (OP == 4) { REP16(G8(I_FMAMIX)) }
        if (OP == 5) { REP16(G8(I_FMAMIXH)) }
        if (OP == 6) { REP16(G8(I_MULSDWA)) }
        if (OP == 7) { REP16(G8(I_ADDSDWAD)) }
        if (OP == 8) { REP16(G8(I_MAD24)) }
        if (OP == 9) { REP16(G8(I_MUL24)) }
        if (OP == 10) { REP16(G8(I_CVTPK)) }
        if (OP == 11) { REP16(G8(I_PERM)) }
        if (OP == 12) { REP16(G8(I_ASHRPK)) }
        if (OP == 13) { REP16(G8(I_PKFMA)) }
        if (OP == 14) { REP16(G8(I_AND)) }
        if (OP == 15) { REP16(G8(I_ANDOR)) }
        if (OP == 16) { REP16(G8(I_BFE)) }
        if (OP == 17) { REP16(G8(I_ADDU)) }
        if (OP == 18) { REP16(G8(I_ADD3)) }
        if (OP == 19) { REP16(G8(I_LSHLADD)) }
        if (OP == 20) { REP16(G8(I_SHIFT)) }
        if (OP == 21) { REP16(G8(I_MOV)) }
        if (OP == 22) { REP16(G8(I_CNDMASK)) }
        if (OP == 23) { REP16(G8(I_DOT4)) }
        if (OP == 24) { REP16(G8(I_DOT2F)) }
        if (OP == 25) { REP16(G8(I_CVTF16)) }
        if (OP == 26) { REP16(G8(I_MED3)) }
        if (OP == 27) { REP16(G8(I_MAXF)) }
        if (OP == 28) { REP16(G8(I_RNDNE)) }
        if (OP == 29) { REP16(G8(I_FMAK)) }
        if (OP == 30) { REP16(G8(I_MADU24)) }
        if (OP == 31) { REP16(G8(I_MULLO)) }
        if (OP == 32) { REP16(G8(I_SAD)) }
        if (OP == 33) { REP16(G8(I_LERP)) }
        if (OP == 34) { REP16(G8(I_CVTU32)) }
        if (OP == 35) { REP16(G8(I_XOR)) }
        if (OP == 36) { REP16(G8(I_LSHLOR)) }
        if (OP == 37) { REP16(G8(I_CVTUBS)) }
        if (OP == 40) { REP16(MIX_FMA_CVT) }
        if (OP == 41) { REP16(MIX_FMA_MAD) }
        if (OP == 42) { REP16(MIX_FMA_SALU) }
    }
    if ((r0 ^ r1 ^ r2 ^ r3 ^ r4 ^ r5 ^ r6 ^ r7) == 0x12345678u) out[0] = 1;  // keep results live
}

#include <chrono>
template <int OP>
void sustained(const char *name, uint32_t *d_out, double seconds, int per_group = 8) {
    const int iters = 2000, waves_per_simd = 8;
    dim3 block(256), grid(256 * waves_per_simd);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(bench<OP>, grid, block, 0, 0, d_out, iters, 1.5f);
    hipDeviceSynchronize();
    const auto t0 = std::chrono::steady_clock::now();
    double last_ms = 0; int launches = 0;
    while (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() < seconds) {
        hipEventRecord(e0);
        for (int k = 0; k < 8; k++) hipLaunchKernelGGL(bench<OP>, grid, block, 0, 0, d_out, iters, 1.5f);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        last_ms = ms / 8; launches += 8;
    }
    const double n_inst = (double)iters * 16.0 * per_group;
    printf("%-22s sustained %.3f ns per wave-instruction per SIMD (last of %d launches)\n", name, last_ms * 1e6 / (n_inst * waves_per_simd), launches);
}
#define CASE(OP, NAME) case OP: sustained<OP>(NAME, d_out, seconds); break;
int main(int argc, char **argv) {
    uint32_t *d_out; hipMalloc(&d_out, 1024);
    const int op = argc > 1 ? atoi(argv[1]) : 0;
    const double seconds = argc > 2 ? atof(argv[2]) : 3.0;
    switch (op) {
        CASE(0, "v_fma_f32") CASE(1, "v_mul_f32") CASE(2, "v_add_f32") CASE(3, "v_cvt_f32_ubyte1") CASE(4, "v_fma_mix_f32") CASE(8, "v_mad_i32_i24")
        CASE(10, "v_cvt_pk_u8_f32") CASE(11, "v_perm_b32") CASE(12, "v_ashr_pk_u8_i32") CASE(14, "v_and_b32") CASE(21, "v_mov_b32") CASE(23, "v_dot4_i32_i8")
        CASE(26, "v_med3_f32") CASE(13, "v_pk_fma_f16") CASE(31, "v_mul_lo_u32") CASE(40, "mix fma+cvt_ubyte")
        default: printf("unknown class %d\n", op);
    }
    return 0;
}

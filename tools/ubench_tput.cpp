// tools/ubench_tput.cpp — THROUGHPUT (not dependent-chain latency) of the VALU instructions the pixel kernels lean on, gfx950:
// 8 independent destination registers per wave, 1 / 2 / 4 / 8 waves per SIMD, every CU busy.  ubench_valu.cpp has two dependent
// chains per wave and under-states what the row loops of the wave kernels sustain (profiles/r02_notes.md, end of section 7).
// Build: hipcc --offload-arch=gfx950 -O2 tools/ubench_tput.cpp -o tools/ubench_tput.bin
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
#define I_SATPK(d)    "v_sat_pk_u8_i16 " #d ", " #d
#define I_MED3I(d)    "v_med3_i32 " #d ", " #d ", 0, %10"
#define I_CVTFI(d)    "v_cvt_f32_i32 " #d ", " #d
#define I_ALIGNBIT(d) "v_alignbit_b32 " #d ", " #d ", %10, 16"
// whole packs of three 16.16 sums (registers r0..r2 / r4..r6) into a BGRA word, two pixels per group of (5 | 5 | 3) x 2 instructions
#define PACK_MED3 asm volatile("v_med3_i32 %0, %0, 0, %10\n v_med3_i32 %1, %1, 0, %10\n v_med3_i32 %2, %2, 0, %10\n v_perm_b32 %3, %1, %0, %10\n v_perm_b32 %3, %2, %3, %10\n" \
    "v_med3_i32 %4, %4, 0, %10\n v_med3_i32 %5, %5, 0, %10\n v_med3_i32 %6, %6, 0, %10\n v_perm_b32 %7, %5, %4, %10\n v_perm_b32 %7, %6, %7, %10" \
    : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7) : "v"(b0), "v"(b1), "v"(i2));
#define PACK_SATPK asm volatile("v_perm_b32 %3, %1, %0, %10\n v_sat_pk_u8_i16 %3, %3\n v_perm_b32 %0, %10, %2, %10\n v_sat_pk_u8_i16 %0, %0\n v_lshl_or_b32 %3, %0, 16, %3\n" \
    "v_perm_b32 %7, %5, %4, %10\n v_sat_pk_u8_i16 %7, %7\n v_perm_b32 %4, %10, %6, %10\n v_sat_pk_u8_i16 %4, %4\n v_lshl_or_b32 %7, %4, 16, %7" \
    : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7) : "v"(b0), "v"(b1), "v"(i2));
#define PACK_ASHR asm volatile("v_ashr_pk_u8_i32 %3, %0, %1, 16\n v_ashr_pk_u8_i32 %0, %2, %10, 16\n v_perm_b32 %3, %0, %3, %10\n" \
    "v_ashr_pk_u8_i32 %7, %4, %5, 16\n v_ashr_pk_u8_i32 %4, %6, %10, 16\n v_perm_b32 %7, %4, %7, %10" \
    : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7) : "v"(b0), "v"(b1), "v"(i2));
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
        if (OP == 50) { REP16(G8(I_SATPK)) }
        if (OP == 51) { REP16(G8(I_MED3I)) }
        if (OP == 52) { REP16(G8(I_CVTFI)) }
        if (OP == 53) { REP16(G8(I_ALIGNBIT)) }
        if (OP == 54) { REP16(PACK_MED3) }
        if (OP == 55) { REP16(PACK_SATPK) }
        if (OP == 56) { REP16(PACK_ASHR) }
        if (OP == 40) { REP16(MIX_FMA_CVT) }
        if (OP == 41) { REP16(MIX_FMA_MAD) }
        if (OP == 42) { REP16(MIX_FMA_SALU) }
    }
    if ((r0 ^ r1 ^ r2 ^ r3 ^ r4 ^ r5 ^ r6 ^ r7) == 0x12345678u) out[0] = 1;  // keep results live
}

template <int OP>
void run(const char *name, uint32_t *d_out, int waves_per_simd, int per_group = 8) {
    const int iters = 100;
    // one block = 4 waves = one wave per SIMD of a CU; waves_per_simd blocks per CU, 256 CUs
    dim3 block(256), grid(256 * waves_per_simd);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(bench<OP>, grid, block, 0, 0, d_out, iters, 1.5f);
    hipDeviceSynchronize();
    float best = 1e30f;
    for (int rep = 0; rep < 3; rep++) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(bench<OP>, grid, block, 0, 0, d_out, iters, 1.5f);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        best = ms < best ? ms : best;
    }
    const double n_inst = (double)iters * 16.0 * per_group;       // vector instructions per wave
    const double ns_per_inst = best * 1e6 / (n_inst * waves_per_simd);
    printf("%-22s waves/SIMD=%d  %.3f ns per wave-instruction per SIMD  (%.2f clk @2.4GHz)\n", name, waves_per_simd, ns_per_inst, ns_per_inst * 2.4);
    fflush(stdout);
}

#define RUN(OP, NAME) for (int w : {1, 2, 4, 8}) run<OP>(NAME, d_out, w);
int main(int argc, char **argv) {
    uint32_t *d_out; hipMalloc(&d_out, 1024);
    if (argc > 1 && argv[1][0] == 'p') {      // packing three 16.16 sums into a BGRA word: single instructions, then whole packs (ns per PIXEL = ns x instructions per pack)
        RUN(50, "v_sat_pk_u8_i16") RUN(51, "v_med3_i32") RUN(52, "v_cvt_f32_i32") RUN(53, "v_alignbit_b32") RUN(11, "v_perm_b32") RUN(12, "v_ashr_pk_u8_i32")
        for (int w : {1, 2, 4, 8}) run<54>("pack: 3 med3 + 2 perm (5 per pixel)", d_out, w, 10);
        for (int w : {1, 2, 4, 8}) run<55>("pack: 2 perm + 2 sat_pk + lshl_or (5)", d_out, w, 10);
        for (int w : {1, 2, 4, 8}) run<56>("pack: 2 ashr_pk + perm (3)", d_out, w, 6);
        return 0;
    }
    if (argc > 1) {      // second part only (the first call of the round ran out of its time limit behind v_lshl_add_u32)
        RUN(20, "v_lshrrev_b32") RUN(21, "v_mov_b32") RUN(26, "v_med3_f32") RUN(27, "v_max_f32") RUN(28, "v_rndne_f32")
        RUN(40, "mix fma+cvt_ubyte") RUN(41, "mix fma+mad24")
        for (int w : {1, 2, 4, 8}) run<42>("mix fma+s_add (valu only)", d_out, w, 8);
        RUN(22, "v_cndmask_b32")
        return 0;
    }
    RUN(0, "v_fma_f32") RUN(1, "v_mul_f32") RUN(2, "v_add_f32") RUN(29, "v_fmac_f32")
    RUN(3, "v_cvt_f32_ubyte1") RUN(37, "v_cvt_f32_ubyte2 (dep)") RUN(25, "v_cvt_f32_f16")
    RUN(4, "v_fma_mix_f32 lo") RUN(5, "v_fma_mix_f32 hi")
    RUN(6, "v_mul_f32_sdwa BYTE_1") RUN(7, "v_add_f32_sdwa dst:BYTE_1")
    RUN(8, "v_mad_i32_i24") RUN(9, "v_mul_i32_i24") RUN(30, "v_mad_u32_u24") RUN(31, "v_mul_lo_u32")
    RUN(10, "v_cvt_pk_u8_f32") RUN(11, "v_perm_b32") RUN(12, "v_ashr_pk_u8_i32") RUN(34, "v_cvt_u32_f32")
    RUN(13, "v_pk_fma_f16") RUN(23, "v_dot4_i32_i8") RUN(24, "v_dot2_f32_f16") RUN(32, "v_sad_u8") RUN(33, "v_lerp_u8")
    RUN(14, "v_and_b32") RUN(15, "v_and_or_b32") RUN(35, "v_xor_b32") RUN(36, "v_lshl_or_b32") RUN(16, "v_bfe_u32")
    RUN(17, "v_add_u32") RUN(18, "v_add3_u32") RUN(19, "v_lshl_add_u32") RUN(20, "v_lshrrev_b32") RUN(21, "v_mov_b32") RUN(22, "v_cndmask_b32")
    RUN(26, "v_med3_f32") RUN(27, "v_max_f32") RUN(28, "v_rndne_f32")
    RUN(40, "mix fma+cvt_ubyte") RUN(41, "mix fma+mad24")
    for (int w : {1, 2, 4, 8}) run<42>("mix fma+s_add (valu only)", d_out, w, 8);
    return 0;
}

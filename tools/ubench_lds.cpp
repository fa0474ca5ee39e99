// tools/ubench_lds.cpp — LDS instruction throughput on gfx950 (clk per wave64 instruction per CU), the forms the pixel kernels use
// for their taps: ds_read_u8 / u16 / b32 / b64 / b128 (per-lane addresses, and one broadcast address), ds_read2_b32, d16 loads,
// ds_write_b128; plus v_readlane / v_alignbyte / v_mov_dpp on the VALU side.  Every CU runs W waves per SIMD.
// Build: hipcc --offload-arch=gfx950 -O2 tools/ubench_lds.cpp -o tools/ubench_lds.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>

#define REP4(x) x x x x
#define REP16(x) REP4(REP4(x))
#define L8(INS) asm volatile(INS(%0) "\n" INS(%1) "\n" INS(%2) "\n" INS(%3) "\n" INS(%4) "\n" INS(%5) "\n" INS(%6) "\n" INS(%7) "\n s_waitcnt lgkmcnt(0)" \
    : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7) : "v"(addr), "v"(seed));
#define L8W(INS) asm volatile(INS(%0) "\n" INS(%1) "\n" INS(%2) "\n" INS(%3) "\n" INS(%4) "\n" INS(%5) "\n" INS(%6) "\n" INS(%7) "\n s_waitcnt lgkmcnt(0)" \
    : "+v"(q0), "+v"(q1), "+v"(q2), "+v"(q3), "+v"(q4), "+v"(q5), "+v"(q6), "+v"(q7) : "v"(addr), "v"(seed));
#define I_U8(d)    "ds_read_u8 " #d ", %8 offset:3"
#define I_U16(d)   "ds_read_u16 " #d ", %8 offset:2"
#define I_B32(d)   "ds_read_b32 " #d ", %8 offset:4"
#define I_D16(d)   "ds_read_u8_d16 " #d ", %8 offset:1"
#define I_D16H(d)  "ds_read_u8_d16_hi " #d ", %8 offset:5"
#define I_B64(d)   "ds_read_b64 " #d ", %8 offset:8"
#define I_R2(d)    "ds_read2_b32 " #d ", %8 offset0:1 offset1:2"
#define I_B128(d)  "ds_read_b128 " #d ", %8 offset:16"
#define I_W128(d)  "ds_write_b128 %8, " #d " offset:16"
#define I_RL(d)    "v_readlane_b32 s20, " #d ", 3\n v_readlane_b32 s21, " #d ", 7"
#define I_ALIGN(d) "v_alignbyte_b32 " #d ", " #d ", %9, %8"
#define I_DPP(d)   "v_mov_b32_dpp " #d ", %9 quad_perm:[0,0,2,2] row_mask:0xf bank_mask:0xf"
#define I_BPERM(d) "ds_bpermute_b32 " #d ", %8, %9"
#define I_SWZ(d)   "ds_swizzle_b32 " #d ", %9 offset:swizzle(QUAD_PERM, 0, 0, 2, 2)"

typedef uint32_t u2 __attribute__((ext_vector_type(2)));
typedef uint32_t u4 __attribute__((ext_vector_type(4)));

template <int OP, int MODE>
__global__ __launch_bounds__(256) void bench(uint32_t *out, int iters, uint32_t seed) {
    extern __shared__ uint8_t smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < 4096; i += 256) ((uint32_t *)smem)[i] = i * 2654435761u;
    __syncthreads();
    // MODE 0: lane * 3 / 2 bytes (the 3:2 scaler's tap pattern), 1: lane * 4, 2: lane * 16, 3: every lane the same address, 4: lane * 8
    uint32_t addr = (MODE == 0 ? (lane * 3) / 2 : MODE == 1 ? lane * 4 : MODE == 2 ? lane * 16 : MODE == 3 ? 0 : lane * 8) + wave * 2048;
    if (OP == 1) addr &= ~1u;
    if (OP == 2 || OP == 6) addr &= ~3u;
    if (OP == 5) addr &= ~7u;
    if (OP == 7 || OP == 8) addr &= ~15u;
    uint32_t r0 = 1, r1 = 2, r2 = 3, r3 = 4, r4 = 5, r5 = 6, r6 = 7, r7 = 8;
    u2 d0 = {0, 0}, d1 = d0, d2 = d0, d3 = d0, d4 = d0, d5 = d0, d6 = d0, d7 = d0;
    u4 q0 = {1, 2, 3, 4}, q1 = q0, q2 = q0, q3 = q0, q4 = q0, q5 = q0, q6 = q0, q7 = q0;
    for (int it = 0; it < iters; it++) {
        if (OP == 0) { REP16(L8(I_U8)) }
        if (OP == 1) { REP16(L8(I_U16)) }
        if (OP == 2) { REP16(L8(I_B32)) }
        if (OP == 3) { REP16(L8(I_D16)) }
        if (OP == 4) { REP16(L8(I_D16H)) }
        if (OP == 5) { REP16(asm volatile(I_B64(%0) "\n" I_B64(%1) "\n" I_B64(%2) "\n" I_B64(%3) "\n" I_B64(%4) "\n" I_B64(%5) "\n" I_B64(%6) "\n" I_B64(%7) "\n s_waitcnt lgkmcnt(0)"
                                     : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), "+v"(d4), "+v"(d5), "+v"(d6), "+v"(d7) : "v"(addr), "v"(seed));) }
        if (OP == 6) { REP16(asm volatile(I_R2(%0) "\n" I_R2(%1) "\n" I_R2(%2) "\n" I_R2(%3) "\n" I_R2(%4) "\n" I_R2(%5) "\n" I_R2(%6) "\n" I_R2(%7) "\n s_waitcnt lgkmcnt(0)"
                                     : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), "+v"(d4), "+v"(d5), "+v"(d6), "+v"(d7) : "v"(addr), "v"(seed));) }
        if (OP == 7) { REP16(L8W(I_B128)) }
        if (OP == 8) { REP16(L8W(I_W128)) }
        if (OP == 9) { REP16(asm volatile(I_RL(%0) "\n" I_RL(%1) "\n" I_RL(%2) "\n" I_RL(%3) : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3) : : "s20", "s21");) }
        if (OP == 10) { REP16(L8(I_ALIGN)) }
        if (OP == 11) { REP16(L8(I_DPP)) }
        if (OP == 12) { REP16(L8(I_BPERM)) }
        if (OP == 13) { REP16(L8(I_SWZ)) }
    }
    uint32_t x = r0 ^ r1 ^ r2 ^ r3 ^ r4 ^ r5 ^ r6 ^ r7 ^ d0.x ^ d1.y ^ d2.x ^ d3.y ^ d4.x ^ d5.y ^ d6.x ^ d7.y ^ q0.x ^ q1.y ^ q2.z ^ q3.w ^ q4.x ^ q5.y ^ q6.z ^ q7.w;
    if (x == 0x12345678u) out[0] = x;
}

template <int OP, int MODE>
void run(const char *name, uint32_t *d_out, int waves_per_simd, int per_group = 8) {
    const int iters = 50;
    dim3 block(256), grid(256 * waves_per_simd);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL((bench<OP, MODE>), grid, block, 16384, 0, d_out, iters, 5u);
    (void)hipDeviceSynchronize();
    float best = 1e30f;
    for (int rep = 0; rep < 3; rep++) {
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL((bench<OP, MODE>), grid, block, 16384, 0, d_out, iters, 5u);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        best = ms < best ? ms : best;
    }
    const double n_inst = (double)iters * 16.0 * per_group * waves_per_simd * 4;      // wave instructions per CU
    const double ns = best * 1e6 / n_inst;
    printf("%-34s waves/SIMD=%d  %.3f ns per wave-instruction per CU (%.2f clk @2.4GHz)\n", name, waves_per_simd, ns, ns * 2.4);
    fflush(stdout);
}
// sustained mode (ubench_lds.bin <case> <seconds>): one form for seconds, so that the power governor settles — sample rocm-smi beside it
// (tools/power_classes.sh; profiles/r05_notes.md section 10)
#include <chrono>
template <int OP, int MODE>
void sustained(const char *name, uint32_t *d_out, double seconds) {
    const int iters = 1000, waves_per_simd = 4;
    dim3 block(256), grid(256 * waves_per_simd);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL((bench<OP, MODE>), grid, block, 16384, 0, d_out, iters, 5u);
    (void)hipDeviceSynchronize();
    const auto t0 = std::chrono::steady_clock::now();
    double last_ms = 0;
    while (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() < seconds) {
        (void)hipEventRecord(e0);
        for (int k = 0; k < 8; k++) hipLaunchKernelGGL((bench<OP, MODE>), grid, block, 16384, 0, d_out, iters, 5u);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        last_ms = ms / 8;
    }
    const double n_inst = (double)iters * 16.0 * 8 * waves_per_simd * 4;
    printf("%-34s sustained %.3f ns per wave-instruction per CU\n", name, last_ms * 1e6 / n_inst);
}
#define RUN(OP, MODE, NAME) for (int w : {2, 4}) run<OP, MODE>(NAME, d_out, w);
int main(int argc, char **argv) {
    uint32_t *d_out; (void)hipMalloc(&d_out, 1024);
    if (argc > 2) {
        const double sec = atof(argv[2]);
        switch (atoi(argv[1])) {
            case 0: sustained<0, 0>("ds_read_u8 lane*1.5", d_out, sec); break;
            case 1: sustained<1, 0>("ds_read_u16 lane*1.5&~1", d_out, sec); break;
            case 2: sustained<2, 1>("ds_read_b32 lane*4", d_out, sec); break;
            case 3: sustained<5, 4>("ds_read_b64 lane*8", d_out, sec); break;
            case 4: sustained<6, 1>("ds_read2_b32 lane*4", d_out, sec); break;
            case 5: sustained<7, 2>("ds_read_b128 lane*16", d_out, sec); break;
            case 6: sustained<8, 2>("ds_write_b128 lane*16", d_out, sec); break;
            case 7: sustained<0, 3>("ds_read_u8 broadcast", d_out, sec); break;
            default: printf("unknown case\n");
        }
        return 0;
    }
    RUN(0, 0, "ds_read_u8 lane*1.5") RUN(0, 1, "ds_read_u8 lane*4") RUN(0, 3, "ds_read_u8 broadcast")
    RUN(1, 0, "ds_read_u16 lane*1.5&~1") RUN(2, 0, "ds_read_b32 lane*1.5&~3") RUN(2, 1, "ds_read_b32 lane*4")
    RUN(3, 0, "ds_read_u8_d16 lane*1.5") RUN(4, 0, "ds_read_u8_d16_hi lane*1.5")
    RUN(5, 4, "ds_read_b64 lane*8") RUN(5, 0, "ds_read_b64 lane*1.5&~7") RUN(6, 0, "ds_read2_b32 lane*1.5&~3") RUN(6, 1, "ds_read2_b32 lane*4")
    RUN(7, 2, "ds_read_b128 lane*16") RUN(7, 3, "ds_read_b128 broadcast") RUN(8, 2, "ds_write_b128 lane*16")
    for (int w : {2, 4}) run<9, 0>("v_readlane_b32 (per SIMD x4)", d_out, w, 8);
    RUN(10, 0, "v_alignbyte_b32 (x4 SIMDs)") RUN(11, 0, "v_mov_b32_dpp quad_perm (x4 SIMDs)")
    RUN(12, 1, "ds_bpermute_b32") RUN(13, 1, "ds_swizzle_b32")
    return 0;
}

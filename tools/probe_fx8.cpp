// tools/probe_fx8.cpp — what the fixed-point sampler's packed tap reads rely on, checked on the device (gfx950):
//   1. ds_read_u16 / ds_read_u16_d16 / ds_read_u16_d16_hi at ODD byte addresses and ds_read_b32 / ds_read2st64_b32 at addresses that are
//      only 2-byte aligned return the bytes at that address (unaligned LDS access), and what they cost next to their aligned forms;
//   2. v_dot4_u32_u8 = sum of four unsigned byte products + the 32-bit addend; v_perm_b32's byte selection;
//   3. throughput of v_dot4_u32_u8 and v_perm_b32 alone and interleaved with v_fma_f32 (does it pair like v_cvt_f32_ubyte?).
// Build: hipcc --offload-arch=gfx950 -O2 tools/probe_fx8.cpp -o tools/probe_fx8.bin
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstring>

__global__ void check(uint32_t *out) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const int lane = threadIdx.x;
    for (int i = lane; i < 4096; i += 64) smem[i] = (uint8_t)(i * 7 + (i >> 8) * 13 + 1);
    __syncthreads();
    const uint32_t a = (uint32_t)(size_t)smem + 1 + lane * 3;        // odd and even addresses, never a multiple of 4 in a row
    uint32_t u16 = 0xAAAAAAAAu, d16 = 0xAAAAAAAAu, b32 = 0, r2a = 0, r2b = 0;
    const uint32_t a2 = (uint32_t)(size_t)smem + 2 + lane * 6;       // even, 2 mod 4 for even lanes
    asm volatile("ds_read_u16 %0, %5\n\tds_read_u16_d16 %1, %5 offset:256\n\tds_read_u16_d16_hi %1, %5 offset:512\n\t"
                 "ds_read_b32 %2, %6\n\tds_read2st64_b32 %3, %6 offset0:1 offset1:2\n\ts_waitcnt lgkmcnt(0)"
                 : "=&v"(u16), "+v"(d16), "=&v"(b32), "=&v"(*(uint64_t *)&r2a) : "v"(0), "v"(a), "v"(a2));
    (void)r2b;
    out[lane * 8 + 0] = u16; out[lane * 8 + 1] = d16; out[lane * 8 + 2] = b32;
    uint64_t r2; memcpy(&r2, &r2a, 8);
    out[lane * 8 + 3] = (uint32_t)r2; out[lane * 8 + 4] = (uint32_t)(r2 >> 32);
    const uint32_t p = 0x04030201u * (lane + 1), w = 0xFF80017Fu ^ (lane * 0x01010101u);
    out[lane * 8 + 5] = __builtin_amdgcn_udot4(p, w, 0xFFFF0000u + lane, false);
    out[lane * 8 + 6] = __builtin_amdgcn_perm(0x77665544u + lane, 0x33221100u + lane, 0x06040200u);
    out[lane * 8 + 7] = __builtin_amdgcn_perm(0x77665544u + lane, 0x33221100u + lane, 0x07050301u);
}

#define REP4(x) x x x x
#define REP16(x) REP4(REP4(x))
#define L8(INS) asm volatile(INS(%0) "\n" INS(%1) "\n" INS(%2) "\n" INS(%3) "\n" INS(%4) "\n" INS(%5) "\n" INS(%6) "\n" INS(%7) "\n s_waitcnt lgkmcnt(0)" \
    : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7) : "v"(addr), "v"(seed));
#define I_U16(d)   "ds_read_u16 " #d ", %8 offset:2"
#define I_U16H(d)  "ds_read_u16_d16_hi " #d ", %8 offset:2"
#define I_B32(d)   "ds_read_b32 " #d ", %8 offset:4"
#define I_DOT(d)   "v_dot4_u32_u8 " #d ", " #d ", %9, " #d
#define I_PERM(d)  "v_perm_b32 " #d ", " #d ", %9, %8"
#define I_FMA(d)   "v_fma_f32 " #d ", " #d ", %9, " #d
#define I_DOTF(d)  "v_dot4_u32_u8 " #d ", " #d ", %9, " #d "\n v_fma_f32 %9, %9, %9, %9"
#define I_PERMF(d) "v_perm_b32 " #d ", " #d ", %9, %8\n v_fma_f32 %9, %9, %9, %9"
#define I_SHR(d)   "v_lshrrev_b32 " #d ", 8, " #d
#define I_DOTS(d)  "v_dot4_u32_u8 " #d ", " #d ", %9, " #d "\n v_lshrrev_b32 %9, 8, %9"
#define I_MUL24(d) "v_mul_u32_u24 " #d ", %9, " #d
#define I_MAD24(d) "v_mad_i32_i24 " #d ", " #d ", %9, " #d

// MODE: address pattern.  0: (lane * 3 / 2) & ~1 (aligned u16), 1: lane * 3 / 2 (u16 at odd addresses for every other lane pair),
// 2: (lane * 3) & ~3 (aligned b32), 3: (lane * 3) & ~1 (b32 at 2 mod 4 for half the lanes)
template <int OP, int MODE>
__global__ __launch_bounds__(256) void bench(uint32_t *out, int iters, uint32_t seed) {
    extern __shared__ uint8_t smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < 4096; i += 256) ((uint32_t *)smem)[i] = i * 2654435761u;
    __syncthreads();
    uint32_t addr = (MODE == 0 ? ((lane * 3) / 2) & ~1 : MODE == 1 ? (lane * 3) / 2 : MODE == 2 ? (lane * 3) & ~3 : (lane * 3) & ~1) + wave * 2048;
    uint32_t r0 = 1, r1 = 2, r2 = 3, r3 = 4, r4 = 5, r5 = 6, r6 = 7, r7 = 8;
    for (int it = 0; it < iters; it++) {
        if (OP == 0) { REP16(L8(I_U16)) }
        if (OP == 1) { REP16(L8(I_U16H)) }
        if (OP == 2) { REP16(L8(I_B32)) }
        if (OP == 3) { REP16(L8(I_DOT)) }
        if (OP == 4) { REP16(L8(I_PERM)) }
        if (OP == 5) { REP16(L8(I_FMA)) }
        if (OP == 6) { REP16(L8(I_DOTF)) }
        if (OP == 7) { REP16(L8(I_PERMF)) }
        if (OP == 8) { REP16(L8(I_DOTS)) }
        if (OP == 9) { REP16(L8(I_MUL24)) }
        if (OP == 10) { REP16(L8(I_MAD24)) }
        if (OP == 11) { REP16(L8(I_SHR)) }
    }
    uint32_t x = r0 ^ r1 ^ r2 ^ r3 ^ r4 ^ r5 ^ r6 ^ r7 ^ seed;
    if (x == 0x12345678u) out[0] = x;
}

template <int OP, int MODE>
void run(const char *name, uint32_t *d_out, int waves_per_simd, int per_group, bool per_cu) {
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
        float ms; (void)hipEventElapsedTime(&ms, e0, e1); best = ms < best ? ms : best;
    }
    // instructions per SIMD (VALU) or per CU (LDS): waves_per_simd waves x (4 SIMDs for per-CU) x iters x 16 x per_group
    const double n = (double)waves_per_simd * (per_cu ? 4 : 1) * iters * 16 * per_group;
    printf("%-44s waves/SIMD=%d  %.3f ns per wave-instruction per %s\n", name, waves_per_simd, best * 1e6 / n, per_cu ? "CU" : "SIMD");
}

int main() {
    uint32_t *d; (void)hipMalloc(&d, 64 * 8 * 4);
    hipLaunchKernelGGL(check, dim3(1), dim3(64), 8192, 0, d);
    uint32_t h[64 * 8]; (void)hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
    auto B = [](int i) { return (uint32_t)(uint8_t)(i * 7 + (i >> 8) * 13 + 1); };
    int bad[8] = {0};
    for (int l = 0; l < 64; l++) {
        const int a = 1 + l * 3, a2 = 2 + l * 6;
        const uint32_t e_u16 = B(a) | B(a + 1) << 8;
        const uint32_t e_d16 = B(a + 256) | B(a + 257) << 8 | B(a + 512) << 16 | B(a + 513) << 24;
        auto W = [&](int p) { return B(p) | B(p + 1) << 8 | B(p + 2) << 16 | B(p + 3) << 24; };
        const uint32_t p = 0x04030201u * (l + 1), w = 0xFF80017Fu ^ (l * 0x01010101u);
        uint32_t dot = 0xFFFF0000u + l;
        for (int k = 0; k < 4; k++) dot += ((p >> (8 * k)) & 255) * ((w >> (8 * k)) & 255);
        const uint32_t s1 = 0x33221100u + l, s0 = 0x77665544u + l;
        auto by = [&](int s) { return s < 4 ? (s1 >> (8 * s)) & 255 : (s0 >> (8 * (s - 4))) & 255; };
        const uint32_t pe = by(0) | by(2) << 8 | by(4) << 16 | by(6) << 24, po = by(1) | by(3) << 8 | by(5) << 16 | by(7) << 24;
        const uint32_t exp[8] = { e_u16, e_d16, W(a2), W(a2 + 256), W(a2 + 512), dot, pe, po };
        for (int k = 0; k < 8; k++) if (h[l * 8 + k] != exp[k]) { if (!bad[k]++) printf("MISMATCH item %d lane %d: got %08x want %08x\n", k, l, h[l * 8 + k], exp[k]); }
    }
    const char *names[8] = { "ds_read_u16 (odd addresses)", "ds_read_u16_d16 + _d16_hi (odd addresses)", "ds_read_b32 (2 mod 4)", "ds_read2st64_b32 lo (2 mod 4)",
                             "ds_read2st64_b32 hi (2 mod 4)", "v_dot4_u32_u8 + addend", "v_perm_b32 0x06040200", "v_perm_b32 0x07050301" };
    for (int k = 0; k < 8; k++) printf("%-44s %s\n", names[k], bad[k] ? "WRONG" : "ok (64 lanes)");
    for (int w : {2, 4}) {
        run<0, 0>("ds_read_u16 aligned (lane*1.5 & ~1)", d, w, 8, true);
        run<0, 1>("ds_read_u16 any byte (lane*1.5)", d, w, 8, true);
        run<1, 1>("ds_read_u16_d16_hi any byte", d, w, 8, true);
        run<2, 2>("ds_read_b32 aligned (lane*3 & ~3)", d, w, 8, true);
        run<2, 3>("ds_read_b32 2-byte aligned (lane*3 & ~1)", d, w, 8, true);
    }
    for (int w : {4, 6}) {
        run<5, 0>("v_fma_f32", d, w, 8, false);
        run<3, 0>("v_dot4_u32_u8", d, w, 8, false);
        run<4, 0>("v_perm_b32", d, w, 8, false);
        run<9, 0>("v_mul_u32_u24", d, w, 8, false);
        run<10, 0>("v_mad_i32_i24", d, w, 8, false);
        run<11, 0>("v_lshrrev_b32", d, w, 8, false);
        run<6, 0>("v_dot4_u32_u8 + v_fma_f32 (per pair: x2)", d, w, 16, false);
        run<7, 0>("v_perm_b32 + v_fma_f32 (per pair: x2)", d, w, 16, false);
        run<8, 0>("v_dot4_u32_u8 + v_lshrrev_b32 (per pair: x2)", d, w, 16, false);
    }
    return 0;
}

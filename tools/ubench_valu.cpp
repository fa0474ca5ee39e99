// tools/ubench_valu.cpp — issue cost (cycles per wave64 instruction on one SIMD) of the
// VALU/LDS instructions the pixel kernels lean on, measured with s_memtime on gfx950.
// Build: hipcc --offload-arch=gfx950 -O2 tools/ubench_valu.cpp -o /tmp/ubench_valu
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

#define REP8(x) x x x x x x x x
#define REP64(x) REP8(REP8(x))

template <int OP>
__global__ void bench(uint64_t *out, int iters, float seed) {
    float a0 = seed, a1 = seed + 1, a2 = seed + 2, a3 = seed + 3;
    float b0 = seed * 2, b1 = seed * 3;
    uint32_t i0 = (uint32_t)seed + threadIdx.x, i1 = i0 * 3 + 1, i2 = i0 ^ 0x55, i3 = i0 + 77;
    uint64_t t0 = __builtin_readcyclecounter();
    uint64_t s0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; it++) {
        if (OP == 0) { REP64(asm volatile("v_mul_f32 %0, %0, %2\n v_mul_f32 %1, %1, %2" : "+v"(a0), "+v"(a1) : "v"(b0));) }
        if (OP == 1) { REP64(asm volatile("v_fma_f32 %0, %0, %2, %3\n v_fma_f32 %1, %1, %2, %3" : "+v"(a0), "+v"(a1) : "v"(b0), "v"(b1));) }
        if (OP == 2) { REP64(asm volatile("v_cvt_f32_ubyte0 %0, %2\n v_cvt_f32_ubyte1 %1, %3" : "=v"(a0), "=v"(a1) : "v"(i0), "v"(i1));) }
        if (OP == 3) { REP64(asm volatile("v_rndne_f32 %0, %0\n v_rndne_f32 %1, %1" : "+v"(a0), "+v"(a1));) }
        if (OP == 4) { REP64(asm volatile("v_med3_f32 %0, %0, %2, %3\n v_med3_f32 %1, %1, %2, %3" : "+v"(a0), "+v"(a1) : "v"(b0), "v"(b1));) }
        if (OP == 5) { REP64(asm volatile("v_cvt_u32_f32 %0, %2\n v_cvt_u32_f32 %1, %3" : "=v"(i0), "=v"(i1) : "v"(a0), "v"(a1));) }
        if (OP == 6) { REP64(asm volatile("v_mul_lo_u32 %0, %0, %2\n v_mul_lo_u32 %1, %1, %2" : "+v"(i0), "+v"(i1) : "v"(i2));) }
        if (OP == 7) { REP64(asm volatile("v_mul_i32_i24 %0, %0, %2\n v_mul_i32_i24 %1, %1, %2" : "+v"(i0), "+v"(i1) : "v"(i2));) }
        if (OP == 8) { REP64(asm volatile("v_mad_i32_i24 %0, %0, %2, %3\n v_mad_i32_i24 %1, %1, %2, %3" : "+v"(i0), "+v"(i1) : "v"(i2), "v"(i3));) }
        if (OP == 9) { REP64(asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(*(double *)&a0) : "v"(*(double *)&b0));  asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(*(double *)&a2) : "v"(*(double *)&b0));) }
        if (OP == 10) { REP64(asm volatile("v_add_u32 %0, %0, %2\n v_add_u32 %1, %1, %2" : "+v"(i0), "+v"(i1) : "v"(i2));) }
        if (OP == 11) { REP64(asm volatile("v_perm_b32 %0, %0, %2, %3\n v_perm_b32 %1, %1, %2, %3" : "+v"(i0), "+v"(i1) : "v"(i2), "v"(i3));) }
        if (OP == 12) { REP64(asm volatile("v_bfe_u32 %0, %2, 8, 8\n v_bfe_u32 %1, %3, 16, 8" : "=v"(i0), "=v"(i1) : "v"(i2), "v"(i3));) }
        if (OP == 13) { REP64(asm volatile("v_cvt_pk_u8_f32 %0, %2, 0, %0\n v_cvt_pk_u8_f32 %1, %3, 1, %1" : "+v"(i0), "+v"(i1) : "v"(a0), "v"(a1));) }
        if (OP == 14) { REP64(asm volatile("v_lshl_add_u32 %0, %0, 2, %2\n v_lshl_add_u32 %1, %1, 2, %2" : "+v"(i0), "+v"(i1) : "v"(i2));) }
        if (OP == 15) { REP64(asm volatile("v_mad_u32_u24 %0, %0, %2, %3\n v_mad_u32_u24 %1, %1, %2, %3" : "+v"(i0), "+v"(i1) : "v"(i2), "v"(i3));) }
        if (OP == 16) { REP64(asm volatile("v_sub_f32 %0, %0, %2\n v_add_f32 %1, %1, %2" : "+v"(a0), "+v"(a1) : "v"(b0));) }
        if (OP == 17) { REP64(asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(*(double *)&a0) : "v"(*(double *)&b0));  asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(*(double *)&a2) : "v"(*(double *)&b0));) }
        if (OP == 18) { REP64(asm volatile("v_cvt_f32_u32 %0, %2\n v_cvt_f32_u32 %1, %3" : "=v"(a0), "=v"(a1) : "v"(i0), "v"(i1));) }
        if (OP == 19) { REP64(asm volatile("v_max_i32 %0, %0, %2\n v_min_i32 %1, %1, %2" : "+v"(i0), "+v"(i1) : "v"(i2));) }
        if (OP == 20) { REP64(asm volatile("v_med3_i32 %0, %0, %2, %3\n v_med3_i32 %1, %1, %2, %3" : "+v"(i0), "+v"(i1) : "v"(i2), "v"(i3));) }
        if (OP == 21) { REP64(asm volatile("v_ashrrev_i32 %0, 16, %0\n v_lshrrev_b32 %1, 8, %1" : "+v"(i0), "+v"(i1));) }
        if (OP == 22) { REP64(asm volatile("v_fma_mix_f32 %0, %2, %3, %0 op_sel_hi:[0,1,0]\n v_fma_mix_f32 %1, %2, %3, %1 op_sel:[0,1,0] op_sel_hi:[0,1,0]" : "+v"(a0), "+v"(a1) : "v"(b0), "v"(i2));) }
        if (OP == 23) { REP64(asm volatile("v_cvt_f16_u16_sdwa %0, %2 dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:BYTE_1\n v_cvt_f16_u16_sdwa %1, %3 dst_sel:WORD_0 dst_unused:UNUSED_PRESERVE src0_sel:BYTE_2" : "+v"(i0), "+v"(i1) : "v"(i2), "v"(i3));) }
        if (OP == 24) { REP64(asm volatile("v_cvt_f16_u16 %0, %2\n v_cvt_f16_u16 %1, %3" : "=v"(i0), "=v"(i1) : "v"(i2), "v"(i3));) }
    }
    uint64_t s1 = __builtin_amdgcn_s_memtime();
    (void)t0;
    if (threadIdx.x == 0) out[blockIdx.x] = s1 - s0;
    if (a0 + a1 + a2 + a3 + (float)(i0 + i1) == 12345.678f) out[0] = 1;  // keep results live
}

template <int OP>
void run(const char *name, uint64_t *d_out, int waves_per_simd) {
    int iters = 200;
    // one block = waves_per_simd*4 waves on one CU; 256 blocks -> one per CU
    dim3 block(256), grid(256 * waves_per_simd);
    hipLaunchKernelGGL(bench<OP>, grid, block, 0, 0, d_out, iters, 1.5f);
    hipDeviceSynchronize();
    hipLaunchKernelGGL(bench<OP>, grid, block, 0, 0, d_out, iters, 1.5f);
    hipDeviceSynchronize();
    std::vector<uint64_t> h(256);
    hipMemcpy(h.data(), d_out, 256 * 8, hipMemcpyDeviceToHost);
    double avg = 0;
    for (auto v : h) avg += (double)v;
    avg /= 256;
    // s_memtime ticks at a constant 100 MHz; convert with the kernel wall clock instead:
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    hipLaunchKernelGGL(bench<OP>, grid, block, 0, 0, d_out, iters, 1.5f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double n_inst = (double)iters * 128.0;             // per wave
    double ns_per_inst = ms * 1e6 / (n_inst * waves_per_simd);   // per SIMD, serialised over its waves
    printf("%-18s waves/SIMD=%d  %.3f ns per wave-instruction per SIMD  (= %.2f cycles @2.4GHz)  memtime ticks %.0f\n",
           name, waves_per_simd, ns_per_inst, ns_per_inst * 2.4, avg);
}

int main() {
    uint64_t *d_out; hipMalloc(&d_out, 256 * 8);
    for (int w : {4, 8}) {
        run<22>("v_fma_mix_f32", d_out, w); run<23>("v_cvt_f16_u16_sdwa", d_out, w); run<24>("v_cvt_f16_u16", d_out, w);
        run<0>("v_mul_f32", d_out, w); run<1>("v_fma_f32", d_out, w); run<16>("v_add/sub_f32", d_out, w);
        run<9>("v_pk_mul_f32", d_out, w); run<17>("v_pk_fma_f32", d_out, w);
        run<2>("v_cvt_f32_ubyteN", d_out, w); run<18>("v_cvt_f32_u32", d_out, w); run<3>("v_rndne_f32", d_out, w);
        run<4>("v_med3_f32", d_out, w); run<5>("v_cvt_u32_f32", d_out, w); run<13>("v_cvt_pk_u8_f32", d_out, w);
        run<6>("v_mul_lo_u32", d_out, w); run<7>("v_mul_i32_i24", d_out, w); run<8>("v_mad_i32_i24", d_out, w);
        run<15>("v_mad_u32_u24", d_out, w); run<10>("v_add_u32", d_out, w); run<14>("v_lshl_add_u32", d_out, w);
        run<11>("v_perm_b32", d_out, w); run<12>("v_bfe_u32", d_out, w); run<19>("v_max/min_i32", d_out, w);
        run<20>("v_med3_i32", d_out, w); run<21>("v_shift", d_out, w);
    }
    return 0;
}

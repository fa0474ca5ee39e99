// tools/ubench_vmem.cpp — per-lane gather loads straight from global memory (no LDS staging) on gfx950: instruction throughput per
// CU of global_load_ubyte / ushort / dword with the 3:2 scaler's address pattern (lane * 1.5 bytes, rows `pitch` apart), for
// working sets that live in the CU's L1 (one 4 KB window per wave), in L2 (windows walking through 64 KB per wave) and beyond.
// Build: hipcc --offload-arch=gfx950 -O2 tools/ubench_vmem.cpp -o tools/ubench_vmem.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

template <int BYTES, int UNROLL>
__global__ __launch_bounds__(256) void bench(const uint8_t *src, uint32_t *out, int iters, int pitch, int window, int span) {
    const int lane = threadIdx.x & 63;
    const int wave = blockIdx.x * 4 + (threadIdx.x >> 6);
    const uint8_t *base = src + (size_t)wave * span;
    uint32_t off = (uint32_t)((lane * 3) / 2) & ~(uint32_t)(BYTES == 4 ? 1 : 0);     // dword loads at even offsets (chroma pairs)
    uint32_t acc = 0, row = 0;
    for (int it = 0; it < iters; it++) {
        uint32_t v[UNROLL];
#pragma unroll
        for (int k = 0; k < UNROLL; k++) {
            const uint8_t *p = base + row + off;
            if (BYTES == 1) v[k] = *(const __attribute__((address_space(1))) uint8_t *)(uintptr_t)p;
            if (BYTES == 2) v[k] = *(const __attribute__((address_space(1))) uint16_t *)(uintptr_t)p;
            if (BYTES == 4) v[k] = *(const __attribute__((address_space(1))) uint32_t *)(uintptr_t)p;
            row += pitch;
            if (row >= (uint32_t)window) row -= window;
        }
#pragma unroll
        for (int k = 0; k < UNROLL; k++) acc += v[k];
        if ((it & 15) == 15) { base += window; if (base >= src + (size_t)(wave + 1) * span) base = src + (size_t)wave * span; }
    }
    if (acc == 0x12345678u) out[0] = acc;
}

template <int BYTES>
void run(const char *name, const uint8_t *src, uint32_t *d_out, int waves_per_simd, int window, int span) {
    const int iters = 400, UNROLL = 8;
    dim3 block(256), grid(256 * waves_per_simd);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL((bench<BYTES, UNROLL>), grid, block, 0, 0, src, d_out, iters, 160, window, span);
    (void)hipDeviceSynchronize();
    float best = 1e30f;
    for (int rep = 0; rep < 3; rep++) {
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL((bench<BYTES, UNROLL>), grid, block, 0, 0, src, d_out, iters, 160, window, span);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        best = ms < best ? ms : best;
    }
    const double n_inst = (double)iters * UNROLL * waves_per_simd * 4;      // load instructions per CU
    const double ns = best * 1e6 / n_inst;
    printf("%-40s waves/SIMD=%d  %.3f ns per wave-load per CU (%.2f clk @2.4GHz)\n", name, waves_per_simd, ns, ns * 2.4);
    fflush(stdout);
}

int main() {
    const size_t total = (size_t)256 * 8 * 4 * 65536;      // 64 KB per wave at 8 waves per SIMD
    uint8_t *src; (void)hipMalloc(&src, total + 4096); (void)hipMemset(src, 1, total + 4096);
    uint32_t *d_out; (void)hipMalloc(&d_out, 1024);
    for (int w : {2, 5, 8}) {
        run<1>("global_load_ubyte  L1 (4 KB window/wave)", src, d_out, w, 4096, 4096);
        run<2>("global_load_ushort L1 (4 KB window/wave)", src, d_out, w, 4096, 4096);
        run<4>("global_load_dword  L1 (4 KB window/wave)", src, d_out, w, 4096, 4096);
        run<2>("global_load_ushort 4 KB windows over 64 KB", src, d_out, w, 4096, 65536);
        run<4>("global_load_dword  4 KB windows over 64 KB", src, d_out, w, 4096, 65536);
    }
    return 0;
}

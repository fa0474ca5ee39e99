// kernels_lanczos.hip.cpp — separable Lanczos-3 resampler for 4-component
// planes (no reference counterpart).
//
// One block produces a TW x TH output tile.  Phase 1 runs the horizontal pass
// for every source row the tile's vertical taps touch and parks the float4
// results in LDS; phase 2 runs the vertical pass out of LDS and stores packed
// BGRA.  Both passes accumulate with one fused multiply-add per tap, taps in
// ascending order from 0.0f — the same chain the oracle evaluates with fmaf().
#include "pixel_math.hip.h"

#pragma clang fp contract(off)

namespace chv {

constexpr int LZ_TW = 32;
constexpr int LZ_TH = 16;

__global__ __launch_bounds__(256) void lanczos3_bgra(DPlane dst, DPlane src,
                                                     const int32_t *__restrict__ fx, const float *__restrict__ wx, int tx,
                                                     const int32_t *__restrict__ fy, const float *__restrict__ wy, int ty,
                                                     int max_rows) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float4 *hrow = (float4 *)smem;  // [max_rows][LZ_TW]
    const int ox0 = blockIdx.x * LZ_TW, oy0 = blockIdx.y * LZ_TH;
    const int oy_last = min(oy0 + LZ_TH, dst.h) - 1;
    const int row0 = fy[oy0];
    const int nrows = min(fy[oy_last] + ty - row0, max_rows);
    const int tid = threadIdx.x;

    // phase 1: horizontal pass into LDS
    for (int idx = tid; idx < nrows * LZ_TW; idx += 256) {
        int r = idx / LZ_TW, i = idx % LZ_TW;
        int ox = ox0 + i;
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        if (ox < dst.w) {
            int sy = min(max(row0 + r, 0), src.h - 1);
            const uint8_t *rowp = src.ptr + (size_t)sy * src.pitch;
            int f = fx[ox];
            const float *w = wx + (size_t)ox * tx;
            for (int k = 0; k < tx; k++) {
                int sx = min(max(f + k, 0), src.w - 1);
                uint32_t p = *(const uint32_t *)(rowp + (size_t)sx * 4);
                float wk = w[k];
                acc.x = __builtin_fmaf(wk, (float)(p & 255), acc.x);
                acc.y = __builtin_fmaf(wk, (float)((p >> 8) & 255), acc.y);
                acc.z = __builtin_fmaf(wk, (float)((p >> 16) & 255), acc.z);
                acc.w = __builtin_fmaf(wk, (float)(p >> 24), acc.w);
            }
        }
        hrow[r * LZ_TW + i] = acc;
    }
    __syncthreads();

    // phase 2: vertical pass out of LDS
    for (int idx = tid; idx < LZ_TW * LZ_TH; idx += 256) {
        int j = idx / LZ_TW, i = idx % LZ_TW;
        int ox = ox0 + i, oy = oy0 + j;
        if (ox >= dst.w || oy >= dst.h) continue;
        int rbase = fy[oy] - row0;
        const float *w = wy + (size_t)oy * ty;
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int k = 0; k < ty; k++) {
            float4 h = hrow[(rbase + k) * LZ_TW + i];
            float wk = w[k];
            acc.x = __builtin_fmaf(wk, h.x, acc.x);
            acc.y = __builtin_fmaf(wk, h.y, acc.y);
            acc.z = __builtin_fmaf(wk, h.z, acc.z);
            acc.w = __builtin_fmaf(wk, h.w, acc.w);
        }
        uint32_t o = to_code_raw(acc.x) | (to_code_raw(acc.y) << 8) | (to_code_raw(acc.z) << 16) | (to_code_raw(acc.w) << 24);
        *(uint32_t *)(dst.ptr + (size_t)oy * dst.pitch + (size_t)ox * 4) = o;
    }
}

hipError_t launch_lanczos(const DPlane &dst, const DPlane &src, const int32_t *fx, const float *wx,
                          int tx, const int32_t *fy, const float *wy, int ty, hipStream_t stream) {
    // rows of horizontal results one tile can need: first[] advances by at most
    // ceil(scale) per output row
    double scale = (double)src.h / (double)dst.h;
    int max_rows = (int)((LZ_TH - 1) * scale + 2) + ty;
    size_t lds = (size_t)max_rows * LZ_TW * sizeof(float4);
    if (lds > 160 * 1024) return hipErrorInvalidValue;
    if (lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute((const void *)lanczos3_bgra, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
    }
    dim3 grid((dst.w + LZ_TW - 1) / LZ_TW, (dst.h + LZ_TH - 1) / LZ_TH);
    hipLaunchKernelGGL(lanczos3_bgra, grid, dim3(256), lds, stream, dst, src, fx, wx, tx, fy, wy, ty, max_rows);
    return hipGetLastError();
}

}  // namespace chv

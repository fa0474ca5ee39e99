// kernels_lanczos.hip.cpp — separable Lanczos-3 resampler for 4-component planes
// (no reference counterpart).
//
// One block produces a 32 x 16 output tile in three LDS-resident phases:
//   A. stage the source rectangle the tile's taps touch (raw BGRA texels, 16-byte coalesced
//      global loads; columns/rows outside the picture are resolved by clamping the LOAD
//      address, so the taps below never clamp),
//   B. horizontal pass: a thread owns one output column (its tap weights live in registers)
//      and walks the staged rows; float4 results go to a second LDS array,
//   C. vertical pass out of that array, packed BGRA store.
// Both passes accumulate with one fused multiply-add per tap, taps in ascending order from
// 0.0f — the same chain the oracle evaluates with fmaf().
#include "pixel_math.hip.h"

#pragma clang fp contract(off)

namespace chv {

constexpr int LZ_TW = 32;
constexpr int LZ_TH = 16;
constexpr int LZ_MAXT = 24;      // taps held in registers (scale <= 4); more taps use the slow loop

template <int TAPS_IN_REGS>
__global__ __launch_bounds__(256) void lanczos3_bgra(DPlane dst, DPlane src,
                                                     const int32_t *__restrict__ fx, const float *__restrict__ wx, int tx,
                                                     const int32_t *__restrict__ fy, const float *__restrict__ wy, int ty,
                                                     int max_rows, int max_cols) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float4 *hrow = (float4 *)smem;                                   // [max_rows][LZ_TW]
    uint32_t *stile = (uint32_t *)(smem + (size_t)max_rows * LZ_TW * sizeof(float4));   // [max_rows][max_cols]
    const int ox0 = blockIdx.x * LZ_TW, oy0 = blockIdx.y * LZ_TH;
    const int ox_last = min(ox0 + LZ_TW, dst.w) - 1, oy_last = min(oy0 + LZ_TH, dst.h) - 1;
    const int row0 = fy[oy0];
    const int nrows = min(fy[oy_last] + ty - row0, max_rows);
    const int col0 = fx[ox0];
    const int ncols = min(fx[ox_last] + tx - col0, max_cols);
    const int tid = threadIdx.x;

    // phase A: source rectangle -> LDS (CLAMP_TO_EDGE applied to the load address).  32 lanes walk a row in
    // 16-byte vectors (4 texels; texels are 4-byte aligned, which is all global_load_dwordx4 needs), 8 rows at a
    // time; only vectors that stick out of the picture fall back to per-texel clamped loads.
    {
        const int v = tid & 31, rg = tid >> 5;
        const int nvec = (ncols + 3) >> 2;                       // max_cols is a multiple of 4: the tail vector fits
        for (int r = rg; r < nrows; r += 8) {
            const int sy = min(max(row0 + r, 0), src.h - 1);
            const uint8_t *srow = src.ptr + (size_t)sy * src.pitch;
            for (int vv = v; vv < nvec; vv += 32) {
                const int c = col0 + 4 * vv;
                uint4 t;
                if (c >= 0 && c + 4 <= src.w) t = gld<uint4>(srow + (size_t)c * 4);
                else {
                    t.x = gld<uint32_t>(srow + (size_t)min(max(c, 0), src.w - 1) * 4);
                    t.y = gld<uint32_t>(srow + (size_t)min(max(c + 1, 0), src.w - 1) * 4);
                    t.z = gld<uint32_t>(srow + (size_t)min(max(c + 2, 0), src.w - 1) * 4);
                    t.w = gld<uint32_t>(srow + (size_t)min(max(c + 3, 0), src.w - 1) * 4);
                }
                *(uint4 *)(stile + r * max_cols + 4 * vv) = t;
            }
        }
    }
    __syncthreads();

    // phase B: horizontal pass, one output column per thread
    {
        const int i = tid & (LZ_TW - 1), rg = tid >> 5;     // 8 row groups
        const int ox = min(ox0 + i, dst.w - 1);
        const int cbase = fx[ox] - col0;
        const float *w = wx + (size_t)ox * tx;
        float wr[TAPS_IN_REGS > 0 ? TAPS_IN_REGS : 1];
        if (TAPS_IN_REGS > 0) {
#pragma unroll
            for (int k = 0; k < TAPS_IN_REGS; k++) wr[k] = k < tx ? w[k] : 0.f;
        }
        for (int r = rg; r < nrows; r += 8) {
            const uint32_t *row = stile + r * max_cols + cbase;
            float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
            if (TAPS_IN_REGS > 0) {
#pragma unroll
                for (int k = 0; k < TAPS_IN_REGS; k++) {
                    if (k < tx) {
                        uint32_t p = row[k];
                        acc.x = __builtin_fmaf(wr[k], (float)(p & 255), acc.x);
                        acc.y = __builtin_fmaf(wr[k], (float)((p >> 8) & 255), acc.y);
                        acc.z = __builtin_fmaf(wr[k], (float)((p >> 16) & 255), acc.z);
                        acc.w = __builtin_fmaf(wr[k], (float)(p >> 24), acc.w);
                    }
                }
            } else {
                for (int k = 0; k < tx; k++) {
                    uint32_t p = row[k];
                    float wk = w[k];
                    acc.x = __builtin_fmaf(wk, (float)(p & 255), acc.x);
                    acc.y = __builtin_fmaf(wk, (float)((p >> 8) & 255), acc.y);
                    acc.z = __builtin_fmaf(wk, (float)((p >> 16) & 255), acc.z);
                    acc.w = __builtin_fmaf(wk, (float)(p >> 24), acc.w);
                }
            }
            hrow[r * LZ_TW + i] = acc;
        }
    }
    __syncthreads();

    // phase C: vertical pass out of LDS
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
        gst<uint32_t>(dst.ptr + (size_t)oy * dst.pitch + (size_t)ox * 4, o);
    }
}

hipError_t launch_lanczos(const DPlane &dst, const DPlane &src, const int32_t *fx, const float *wx,
                          int tx, const int32_t *fy, const float *wy, int ty, hipStream_t stream) {
    // rows / columns of source one tile can need: first[] advances by at most ceil(scale) per output
    double sy = (double)src.h / (double)dst.h, sxs = (double)src.w / (double)dst.w;
    int max_rows = (int)((LZ_TH - 1) * sy + 2) + ty;
    int max_cols = (int)((LZ_TW - 1) * sxs + 2) + tx;
    max_cols = (max_cols + 3) & ~3;
    size_t lds = (size_t)max_rows * LZ_TW * sizeof(float4) + (size_t)max_rows * max_cols * sizeof(uint32_t);
    if (lds > 160 * 1024) return hipErrorInvalidValue;
    dim3 grid((dst.w + LZ_TW - 1) / LZ_TW, (dst.h + LZ_TH - 1) / LZ_TH);
    auto launch = [&](auto kernel) -> hipError_t {
        if (lds > 64 * 1024) {
            hipError_t e = hipFuncSetAttribute((const void *)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            if (e != hipSuccess) return e;
        }
        hipLaunchKernelGGL(kernel, grid, dim3(256), lds, stream, dst, src, fx, wx, tx, fy, wy, ty, max_rows, max_cols);
        return hipGetLastError();
    };
    if (tx <= 12) return launch(lanczos3_bgra<12>);
    if (tx <= LZ_MAXT) return launch(lanczos3_bgra<LZ_MAXT>);
    return launch(lanczos3_bgra<0>);
}

}  // namespace chv

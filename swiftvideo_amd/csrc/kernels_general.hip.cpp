// kernels_general.hip.cpp — the general (any affine transform) tick kernels.
//
// One launch = a batch of mixer ticks (grid.z = tick).  Each thread owns one
// canvas pixel (BGRA target) or one 2x2 luma quad + its chroma sample (4:2:0
// targets: the quad's even/even pixel is the reference's `handleChroma` owner,
// kernels.cl.swift:76), keeps the canvas value in registers as 8-bit codes,
// applies every layer of the tick in z order and stores once.  Between layers
// the value is re-quantised exactly as the reference's per-layer kernels do
// through their UNORM8 canvas (write_imagef then read_imagef), so the result is
// byte-identical to clear + N x runComputeKernel(blends: true)
// (mix.video.swift:116-124, compute.cl.swift:288-335).
//
// Sources are gathered straight from global memory; the axis-aligned LDS-tiled
// fast paths live in kernels_fast.hip.cpp.
#include "pixel_math.hip.h"
#include "bgra_pixel.hip.h"
#include "yuv_pixel.hip.h"

#include <algorithm>

#pragma clang fp contract(off)

namespace chv {

// apply_layer_bgra: bgra_pixel.hip.h (shared with kernels_wave.hip.cpp)

// (ox, oy): the grid's first pixel — launches on canvases that are not cleared cover the union of their layers' bounding boxes only
// (launch_tick_general: a pixel outside every layer's box keeps its bytes)
__global__ __launch_bounds__(256) void tick_general_bgra(const DTick *__restrict__ ticks,
                                                         const DLayer *__restrict__ layers, int ox, int oy) {
    const DTick &T = ticks[blockIdx.z];
    int x = ox + blockIdx.x * 64 + threadIdx.x;
    int y = oy + blockIdx.y * 4 + threadIdx.y;
    if (x >= T.W || y >= T.H) return;
    const DPlane &D = T.dst.pl[0];
    if (x >= D.w || y >= D.h) return;
    uint8_t *dp = D.ptr + (size_t)y * D.pitch + (size_t)x * 4;
    // img_clear_bgra: (0,0,0,1), kernels.cl.swift:257-265
    uint32_t cur = T.clear_first ? 0xFF000000u : gld<uint32_t>(dp);
    float sx = (float)T.W, sy = (float)T.H;
    const DLayer *L = layers + T.first_layer;
    for (int l = 0; l < T.n_layers; l++) {
        const DLayer &Ly = L[l];
        if (x < Ly.bbox[0] || x >= Ly.bbox[2] || y < Ly.bbox[1] || y >= Ly.bbox[3]) continue;   // fails the border test for sure
        cur = apply_layer_bgra(Ly, x, y, sx, sy, cur);
    }
    gst<uint32_t>(dp, cur);
}

// ---------------------------------------------------------------------------
// 4:2:0 targets (NV12 / Y420P)
// ---------------------------------------------------------------------------
struct QuadState {
    uint32_t y[4];   // luma codes of (2qx+i, 2qy+j), index j*2+i
    uint32_t u, v;   // chroma codes at (qx, qy)
};

// apply_yuv_from_yuv / apply_yuv_from_rgb: yuv_pixel.hip.h (shared with kernels_wave_yuv.hip.cpp)

template <int TF>
__global__ __launch_bounds__(256) void tick_general_yuv(const DTick *__restrict__ ticks,
                                                        const DLayer *__restrict__ layers) {
    const DTick &T = ticks[blockIdx.z];
    int qx = blockIdx.x * 32 + threadIdx.x;
    int qy = blockIdx.y * 8 + threadIdx.y;
    int x0 = qx * 2, y0 = qy * 2;
    if (x0 >= T.W || y0 >= T.H) return;
    const DPlane &PY = T.dst.pl[0];
    const DPlane &PC = T.dst.pl[1];
    bool hx = x0 + 1 < T.W, hy = y0 + 1 < T.H;
    // gid/2 can fall outside the chroma plane on odd-sized canvases: such reads
    // are zero and such writes are dropped (oracle/ref_kernels.c, rd_near1/wr1)
    bool cvalid = qx < PC.w && qy < PC.h;
    uint8_t *py0 = PY.ptr + (size_t)y0 * PY.pitch + x0;
    uint8_t *py1 = py0 + PY.pitch;
    uint8_t *pu, *pv;
    if (TF == TF_NV12) { pu = PC.ptr + (size_t)qy * PC.pitch + (size_t)qx * 2; pv = pu + 1; }
    else {
        pu = PC.ptr + (size_t)qy * PC.pitch + qx;
        pv = T.dst.pl[2].ptr + (size_t)qy * T.dst.pl[2].pitch + qx;
    }
    QuadState s;
    if (T.clear_first) {
        // img_clear_nv12 / img_clear_y420p: Y = 0.0, chroma = 0.5 -> code 128 (RTE)
        s.y[0] = s.y[1] = s.y[2] = s.y[3] = 0;
        s.u = s.v = cvalid ? 128u : 0u;
    } else {
        s.y[0] = gld<uint8_t>(py0);
        s.y[1] = hx ? gld<uint8_t>(py0 + 1) : 0;
        s.y[2] = hy ? gld<uint8_t>(py1) : 0;
        s.y[3] = (hx && hy) ? gld<uint8_t>(py1 + 1) : 0;
        s.u = cvalid ? gld<uint8_t>(pu) : 0;
        s.v = cvalid ? gld<uint8_t>(pv) : 0;
    }
    float sx = (float)T.W, sy = (float)T.H;
    const DLayer *L = layers + T.first_layer;
    for (int l = 0; l < T.n_layers; l++) {
        const DLayer &Ly = L[l];
        if (x0 + 1 < Ly.bbox[0] || x0 >= Ly.bbox[2] || y0 + 1 < Ly.bbox[1] || y0 >= Ly.bbox[3]) continue;
        if (!cvalid) { s.u = 0; s.v = 0; }
        uint32_t du = 0, dv = 0;  // chroma of non-owner pixels: computed by the reference, never stored
#pragma unroll
        for (int k = 0; k < 4; k++) {
            int i = k & 1, j = k >> 1;
            if ((i && !hx) || (j && !hy)) continue;
            bool owner = (k == 0);
            if (Ly.kind == LK_YUV_FROM_RGB || Ly.kind == LK_YUV_FROM_RGB_INT)
                apply_yuv_from_rgb(Ly, x0 + i, y0 + j, sx, sy, owner, s.y[k], owner ? s.u : du, owner ? s.v : dv);
            else
                apply_yuv_from_yuv(Ly, x0 + i, y0 + j, sx, sy, owner, s.y[k], owner ? s.u : du, owner ? s.v : dv);
        }
    }
    gst<uint8_t>(py0, (uint8_t)s.y[0]);
    if (hx) gst<uint8_t>(py0 + 1, (uint8_t)s.y[1]);
    if (hy) gst<uint8_t>(py1, (uint8_t)s.y[2]);
    if (hx && hy) gst<uint8_t>(py1 + 1, (uint8_t)s.y[3]);
    if (cvalid) { gst<uint8_t>(pu, (uint8_t)s.u); gst<uint8_t>(pv, (uint8_t)s.v); }
}

// ---------------------------------------------------------------------------
// launchers (called from chipvideo.cpp)
// ---------------------------------------------------------------------------
// ticks_host / layers_host (may be null): the same descriptors on the host, for the grid of launches that do not clear
hipError_t launch_tick_general(int target_format, const DTick *ticks_host, const DLayer *layers_host, const DTick *ticks, const DLayer *layers,
                               int n_ticks, int maxW, int maxH, hipStream_t stream) {
    if (n_ticks <= 0) return hipSuccess;
    if (target_format == TF_BGRA) {
        int x0 = 0, y0 = 0, x1 = maxW, y1 = maxH;
        if (ticks_host && layers_host) {
            // no tick clears: the union of every layer's bounding box (what the second launch of a split batch — a rotated logo over videos the
            // streaming kernel composed — has to touch: a few thousand pixels of a 720p canvas)
            bool clears = false;
            int bx0 = maxW, by0 = maxH, bx1 = 0, by1 = 0;
            for (int i = 0; i < n_ticks && !clears; i++) {
                const DTick &T = ticks_host[i];
                clears = T.clear_first != 0;
                for (int l = 0; l < T.n_layers; l++) {
                    const DLayer &L = layers_host[T.first_layer + l];
                    bx0 = std::min(bx0, std::max(L.bbox[0], 0)); by0 = std::min(by0, std::max(L.bbox[1], 0));
                    bx1 = std::max(bx1, std::min(L.bbox[2], T.W)); by1 = std::max(by1, std::min(L.bbox[3], T.H));
                }
            }
            if (!clears) {
                if (bx1 <= bx0 || by1 <= by0) return hipSuccess;          // nothing any layer could touch
                x0 = bx0 & ~15; y0 = by0 & ~3; x1 = bx1; y1 = by1;        // (64-byte aligned rows of four)
            }
        }
        dim3 block(64, 4), grid((x1 - x0 + 63) / 64, (y1 - y0 + 3) / 4, n_ticks);
        hipLaunchKernelGGL(tick_general_bgra, grid, block, 0, stream, ticks, layers, x0, y0);
    } else {
        int qw = (maxW + 1) / 2, qh = (maxH + 1) / 2;
        dim3 block(32, 8), grid((qw + 31) / 32, (qh + 7) / 8, n_ticks);
        if (target_format == TF_NV12)
            hipLaunchKernelGGL(tick_general_yuv<TF_NV12>, grid, block, 0, stream, ticks, layers);
        else
            hipLaunchKernelGGL(tick_general_yuv<TF_Y420P>, grid, block, 0, stream, ticks, layers);
    }
    return hipGetLastError();
}

// ---------------------------------------------------------------------------
// device self-test of the primitive conversions (tests/test_gpu_primitives.py)
// out_f[0..255]   = unorm8(c)
// out_c[i]        = to_code(in_f[i])            i < n
// out_q[i]        = (float)num[i] / den[i]      i < n   (division used by `geometry`)
// ---------------------------------------------------------------------------
__global__ void selftest_kernel(float *out_f, const float *in_f, uint8_t *out_c, const float *num,
                                const float *den, float *out_q, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < 256) out_f[i] = unorm8((uint32_t)i);
    if (i < n) {
        out_c[i] = (uint8_t)to_code(in_f[i]);
        out_q[i] = num[i] / den[i];
    }
}
// out[i] = {pack_bgra_fixed(b,g,r), pack_bgra_fixed_pk(b,g,r)} for 16.16 channel sums
__global__ void selftest_pack_kernel(const int *b, const int *g, const int *r, uint32_t *out, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) { out[2 * i] = pack_bgra_fixed(b[i], g[i], r[i]); out[2 * i + 1] = pack_bgra_fixed_pk(b[i], g[i], r[i]); }
}
// out[i] = {to_code_raw x 4 packed with shifts, pack_codes(c0, c1, c2, c3)} for code-scale floats in[4i .. 4i+3]
__global__ void selftest_pack_codes_kernel(const float *in, uint32_t *out, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) {
        const float c0 = in[4 * i], c1 = in[4 * i + 1], c2 = in[4 * i + 2], c3 = in[4 * i + 3];
        out[2 * i] = to_code_raw(c0) | (to_code_raw(c1) << 8) | (to_code_raw(c2) << 16) | (to_code_raw(c3) << 24);
        out[2 * i + 1] = pack_codes(c0, c1, c2, c3);
    }
}
// Exhaustive self-test of the integer colour matrices, all 2^24 code triples of one matrix in one launch.
//   dir 0: i = Y << 16 | U << 8 | V  ->  out[i] = B | G << 8 | R << 16 | 255 << 24 through the plain form (yuv_to_bgra_word(Csc)); every OTHER
//          form the kernels use — offsets folded into one constant per channel, operands carrying the float adder's bias 0x4B400000 + code
//          (what code_biased hands the tiled / wave / stream kernels) packed by v_ashr_pk_u8_i32, and the float-code form of the blending
//          kernels packed by v_cvt_pk_u8_f32, the same with the offsets absorbed into the conversion biases
//          (yuv_to_bgr_floats_absorbed, where the matrix has such biases) — is compared with it on the device: *mism counts the triples on which any of them differs
//   dir 1: i = R << 16 | G << 8 | B  ->  out[i] = Y | U << 8 | V << 16 through rgb_to_yuv_int; *mism counts the triples on which the tick kernels'
//          form of the rows (biased operands, folded offsets, float codes) differs from it
__global__ void selftest_matrices_kernel(int dir, int csc, uint32_t *out, uint32_t *mism) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    const int a = (int)(i >> 16), b = (int)((i >> 8) & 255u), c = (int)(i & 255u);
    if (dir == 0) {
        const Csc &k = kCsc[csc & 3];
        const uint32_t w0 = yuv_to_bgra_word(k, a, b, c);
        const CscFolded f = csc_fold(k), fb = csc_fold_biased(k);
        const uint32_t w1 = yuv_to_bgra_word(f, a, b, c);
        const int ya = (int)code_biased((float)a), ua = (int)code_biased((float)b), va = (int)code_biased((float)c);
        const uint32_t w2 = yuv_to_bgra_word(fb, ya, ua, va);
        float pb, pg, pr;
        yuv_to_bgr_floats(fb, ya, ua, va, pb, pg, pr);
        const uint32_t w3 = pack_codes(pb, pg, pr, 0xFF000000u);
        // the absorbed form (csc_fold_absorbed: conversion biases that finish the red and blue channels in their multiply-adds), fed floats
        uint32_t w4 = w0;
        if (csc_absorbable(csc)) {
            float qb, qg, qr;
            yuv_to_bgr_floats_absorbed(csc_fold_absorbed(csc), (float)a, (float)b, (float)c, qb, qg, qr);
            w4 = pack_codes(qb, qg, qr, 0xFF000000u);
            // ... and its three conversions on samples BETWEEN codes (a + b / 256: the ties at b = 128 must go to the even code, as code_biased's)
            const CscAbsorbed q = csc_fold_absorbed(csc);
            const float f = (float)min(a, 254) + (float)b * (1.0f / 256.0f);
            const uint32_t want = code_biased(f) & 255u;
            const float mags[3] = { q.my, q.mu, q.mv };
            for (int m = 0; m < 3; m++) {
                const int32_t low = (int32_t)(__float_as_uint(f + mags[m]) & 0xFFFFFFu), bias = (int32_t)fabsf(mags[m]) - (1 << 23);
                const int32_t code = mags[m] > 0.f ? low - bias : bias - low;
                if (code != (int32_t)want) w4 = ~w0;
            }
        }
        out[i] = w0;
        if (w1 != w0 || w2 != w0 || w3 != w0 || w4 != w0) atomicAdd(mism, 1u);
    } else {
        uint32_t y, u, v;
        rgb_to_yuv_int(kR2Y[csc & 3], a, b, c, y, u, v);
        // (the three codes leave as the kernels consume them — converted to floats — and are packed by v_cvt_pk_u8_f32: written as
        // `y | u << 8 | v << 16`, hipcc 7.2 selects v_ashr_pk_u8_i32 for the clip8 pairs and ORs the third code into a register whose upper
        // half that instruction does not leave zero — the defect pixel_math.hip.h::pack_bgra_fixed documents; it showed here as 15.5 M wrong
        // V codes in the first version of this very test)
        out[i] = pack_codes((float)y, (float)u, (float)v, 0u);
        // the form the tick kernels use: operands that keep code_biased's 2^22, row offsets that take it out again (r2y_base_biased), the
        // code as a float through clamp + v_cvt_f32_ubyte2 (fixed_to_codef)
        const R2Y &k = kR2Y[csc & 3];
        const int ra = (int)code_biased((float)a), ga = (int)code_biased((float)b), ba = (int)code_biased((float)c);
        const float fy = fixed_to_codef(r2y_row(k.y[0], k.y[1], k.y[2], r2y_base_biased(k.y[0], k.y[1], k.y[2], (k.yoff << 16) + 32768), ra, ga, ba));
        const float fu = fixed_to_codef(r2y_row(k.u[0], k.u[1], k.u[2], r2y_base_biased(k.u[0], k.u[1], k.u[2], (128 << 16) + 32768), ra, ga, ba));
        const float fv = fixed_to_codef(r2y_row(k.v[0], k.v[1], k.v[2], r2y_base_biased(k.v[0], k.v[1], k.v[2], (128 << 16) + 32768), ra, ga, ba));
        if (fy != (float)y || fu != (float)u || fv != (float)v) atomicAdd(mism, 1u);
    }
}
hipError_t launch_selftest_matrices(int dir, int csc, uint32_t *out, uint32_t *mism, hipStream_t stream) {
    hipLaunchKernelGGL(selftest_matrices_kernel, dim3((1u << 24) / 256), dim3(256), 0, stream, dir, csc, out, mism);
    return hipGetLastError();
}
hipError_t launch_selftest_pack_codes(const float *in, uint32_t *out, int n, hipStream_t stream) {
    hipLaunchKernelGGL(selftest_pack_codes_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, in, out, n);
    return hipGetLastError();
}
hipError_t launch_selftest_pack(const int *b, const int *g, const int *r, uint32_t *out, int n, hipStream_t stream) {
    hipLaunchKernelGGL(selftest_pack_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, b, g, r, out, n);
    return hipGetLastError();
}
hipError_t launch_selftest(float *out_f, const float *in_f, uint8_t *out_c, const float *num,
                           const float *den, float *out_q, int n, hipStream_t stream) {
    int m = n > 256 ? n : 256;
    hipLaunchKernelGGL(selftest_kernel, dim3((m + 255) / 256), dim3(256), 0, stream, out_f, in_f, out_c,
                       num, den, out_q, n);
    return hipGetLastError();
}

}  // namespace chv

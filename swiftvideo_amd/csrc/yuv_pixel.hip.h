// yuv_pixel.hip.h — one layer applied to one pixel of a 4:2:0 canvas, exactly as the reference's OpenCL kernels do it
// (unit-scale arithmetic, taps gathered from global memory).  Used by the general kernel for every pixel and by the
// wave-per-strip kernel for strips that are not entirely inside a picture.
#pragma once
#include "pixel_math.hip.h"

#pragma clang fp contract(off)

namespace chv {

// YUV source: kernels.cl.swift:78-105 (img_nv12_nv12), :141-170, :219-252
CHV_DEV void apply_yuv_from_yuv(const DLayer &L, int x, int y, float sx, float sy, bool owner,
                                uint32_t &cy, uint32_t &cu, uint32_t &cv) {
    const float *U = L.u;
    Geo g = geometry_for(L, x, y, sx, sy);
    if (!g.in_border) return;
    float curY = unorm8(cy);
    if (g.in_tx && g.in_uv) {
        Lin2 ly = lin_setup(L.src.pl[0], g.u, g.v);
        float luma = lin_fetch(L.src.pl[0], ly, 0);
        float alpha = U[U_OPACITY];
        cy = to_code(curY * (1.f - alpha) + luma * alpha);
        if (owner) {
            float cb, cr;
            Lin2 lc = lin_setup(L.src.pl[1], g.u, g.v);  // same normalized uv on the half-size plane
            if (L.kind == LK_YUV_FROM_NV12) {
                cb = lin_fetch(L.src.pl[1], lc, 0);
                cr = lin_fetch(L.src.pl[1], lc, 1);
            } else {
                cb = lin_fetch(L.src.pl[1], lc, 0);
                Lin2 lv = lin_setup(L.src.pl[2], g.u, g.v);
                cr = lin_fetch(L.src.pl[2], lv, 0);
            }
            cu = to_code(unorm8(cu) * (1.f - alpha) + cb * alpha);
            cv = to_code(unorm8(cv) * (1.f - alpha) + cr * alpha);
        }
        return;
    }
    float fy, fu, fv;
    rgb2yuv(U[U_FILL + 0], U[U_FILL + 1], U[U_FILL + 2], fy, fu, fv);
    float alpha = U[U_OPACITY] * U[U_FILL + 3];
    cy = to_code(clampf(curY * (1.f - alpha) + fy * alpha, 0.f, 1.f));
    if (owner) {
        cu = to_code(clampf(unorm8(cu) * (1.f - alpha) + fu * alpha, -1.f, 1.f));
        cv = to_code(clampf(unorm8(cv) * (1.f - alpha) + fv * alpha, -1.f, 1.f));
    }
}

// RGB source, integer matrix (img_{bgra,rgba}_{nv12,y420p}_int; spec owned by this repository; statement of
// record: oracle/ref_kernels.c::px_rgb_to_yuv_int): code-scale sample and blend like the BGRA-target family, 16.16 matrix on
// the quantised sample, fill painted inside the whole border quad.  Chroma only at owner pixels (cu / cv untouched otherwise).
CHV_DEV void apply_yuv_from_rgb_int(const DLayer &L, int x, int y, float sx, float sy, bool owner,
                                    uint32_t &cy, uint32_t &cu, uint32_t &cv) {
    const float *U = L.u;
    Geo g = geometry_for(L, x, y, sx, sy);
    if (!g.in_border) return;
    const R2Y &k = kR2Y[L.csc & 3];
    const float af = U[U_OPACITY] * U[U_FILL + 3], iaf = 1.f - af;
    uint32_t fy, fu, fv;
    rgb_to_yuv_int(k, (int)to_code_raw(U[U_FILL + 0] * 255.0f), (int)to_code_raw(U[U_FILL + 1] * 255.0f), (int)to_code_raw(U[U_FILL + 2] * 255.0f), fy, fu, fv);
    float r0 = clampf(__builtin_fmaf((float)fy, af, (float)cy * iaf), 0.f, 255.f);
    float r1 = clampf(__builtin_fmaf((float)fu, af, (float)cu * iaf), 0.f, 255.f);
    float r2 = clampf(__builtin_fmaf((float)fv, af, (float)cv * iaf), 0.f, 255.f);
    if (g.in_tx && g.in_uv) {
        const DPlane &P = L.src.pl[0];
        Lin2 l = lin_setup(P, g.u, g.v);
        const uint32_t t00 = gld<uint32_t>(P.ptr + l.o00), t10 = gld<uint32_t>(P.ptr + l.o10);
        const uint32_t t01 = gld<uint32_t>(P.ptr + l.o01), t11 = gld<uint32_t>(P.ptr + l.o11);
        const float q0 = cs_mix(l, (float)(t00 & 255), (float)(t10 & 255), (float)(t01 & 255), (float)(t11 & 255));
        const float q1 = cs_mix(l, (float)((t00 >> 8) & 255), (float)((t10 >> 8) & 255), (float)((t01 >> 8) & 255), (float)((t11 >> 8) & 255));
        const float q2 = cs_mix(l, (float)((t00 >> 16) & 255), (float)((t10 >> 16) & 255), (float)((t01 >> 16) & 255), (float)((t11 >> 16) & 255));
        const float q3 = cs_mix(l, (float)(t00 >> 24), (float)(t10 >> 24), (float)(t01 >> 24), (float)(t11 >> 24));
        uint32_t py, pu, pv;
        rgb_to_yuv_int(k, (int)to_code_raw(L.swizzle ? q2 : q0), (int)to_code_raw(q1), (int)to_code_raw(L.swizzle ? q0 : q2), py, pu, pv);
        const float a = q3 * (U[U_OPACITY] * kInv255), ia = 1.f - a;
        r0 = __builtin_fmaf((float)py, a, r0 * ia);
        r1 = __builtin_fmaf((float)pu, a, r1 * ia);
        r2 = __builtin_fmaf((float)pv, a, r2 * ia);
    }
    cy = to_code_raw(r0);
    if (owner) { cu = to_code_raw(r1); cv = to_code_raw(r2); }
}

// RGB source: kernels.cl.swift:495-530 (img_bgra_nv12) and its three siblings
CHV_DEV void apply_yuv_from_rgb(const DLayer &L, int x, int y, float sx, float sy, bool owner,
                                uint32_t &cy, uint32_t &cu, uint32_t &cv) {
    if (L.kind == LK_YUV_FROM_RGB_INT) { apply_yuv_from_rgb_int(L, x, y, sx, sy, owner, cy, cu, cv); return; }
    const float *U = L.u;
    Geo g = geometry_for(L, x, y, sx, sy);
    if (!g.in_border || !g.in_tx) return;
    float alpha = U[U_OPACITY] * U[U_FILL + 3];
    float fy, fu, fv;
    rgb2yuv(U[U_FILL + 0] * alpha, U[U_FILL + 1] * alpha, U[U_FILL + 2] * alpha, fy, fu, fv);
    float rx = unorm8(cy) * (1.f - alpha) + fy * alpha;
    float ry = clampf(unorm8(cu) * (1.f - alpha) + fu * alpha, -1.f, 1.f);
    float rz = clampf(unorm8(cv) * (1.f - alpha) + fv * alpha, -1.f, 1.f);
    if (g.in_uv) {
        const DPlane &P = L.src.pl[0];
        Lin2 l = lin_setup(P, g.u, g.v);
        uint32_t t00 = gld<uint32_t>(P.ptr + l.o00), t10 = gld<uint32_t>(P.ptr + l.o10);
        uint32_t t01 = gld<uint32_t>(P.ptr + l.o01), t11 = gld<uint32_t>(P.ptr + l.o11);
        float q0 = lin_mix(l, unorm8(t00 & 255), unorm8(t10 & 255), unorm8(t01 & 255), unorm8(t11 & 255));
        float q1 = lin_mix(l, unorm8((t00 >> 8) & 255), unorm8((t10 >> 8) & 255),
                           unorm8((t01 >> 8) & 255), unorm8((t11 >> 8) & 255));
        float q2 = lin_mix(l, unorm8((t00 >> 16) & 255), unorm8((t10 >> 16) & 255),
                           unorm8((t01 >> 16) & 255), unorm8((t11 >> 16) & 255));
        float q3 = lin_mix(l, unorm8(t00 >> 24), unorm8(t10 >> 24), unorm8(t01 >> 24), unorm8(t11 >> 24));
        float r = L.swizzle ? q2 : q0, gg = q1, b = L.swizzle ? q0 : q2;  // .zyxw for bgra, :518
        float a2 = q3 * U[U_OPACITY];
        float yy, uu, vv;
        rgb2yuv(r * a2, gg * a2, b * a2, yy, uu, vv);
        rx = rx * (1.f - a2) + yy * a2;
        ry = ry * (1.f - a2) + uu * a2;
        rz = rz * (1.f - a2) + vv * a2;
    }
    cy = to_code(rx);
    if (owner) { cu = to_code(ry); cv = to_code(rz); }
}

}  // namespace chv

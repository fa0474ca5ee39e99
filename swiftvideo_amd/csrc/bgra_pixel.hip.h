// bgra_pixel.hip.h — one layer applied to one pixel of a BGRA canvas, any transform: the general kernel's pixel code
// (kernels_general.hip.cpp), shared with the wave kernel, which routes the layers it cannot stage (rotation, shear,
// unbounded matrices) through it pixel by pixel.
#pragma once
#include "pixel_math.hip.h"

#pragma clang fp contract(off)

namespace chv {

CHV_DEV uint32_t apply_layer_bgra(const DLayer &L, int x, int y, float sx, float sy, uint32_t cur) {
    const float *U = L.u;
    if (L.kind == LK_BGRA_METAL) {
        // kernels.metal:52-62
        float rx = U[U_INSIZE + 0] / U[U_OUTSIZE + 0];
        float ry = U[U_INSIZE + 1] / U[U_OUTSIZE + 1];
        int ix = min(max((int)((float)x * rx), 0), L.src.pl[0].w - 1);
        int iy = min(max((int)((float)y * ry), 0), L.src.pl[0].h - 1);
        uint32_t s = gld<uint32_t>(L.src.pl[0].ptr + (size_t)iy * L.src.pl[0].pitch + (size_t)ix * 4);
        float a = unorm8(s >> 24);
        float ia = 1.0f - a;
        uint32_t o0 = to_code(unorm8(s & 255) * a + unorm8(cur & 255) * ia);
        uint32_t o1 = to_code(unorm8((s >> 8) & 255) * a + unorm8((cur >> 8) & 255) * ia);
        uint32_t o2 = to_code(unorm8((s >> 16) & 255) * a + unorm8((cur >> 16) & 255) * ia);
        return o0 | (o1 << 8) | (o2 << 16) | 0xFF000000u;
    }
    // BGRA-target family: code-scale arithmetic (pixel_math.hip.h)
    Geo g = geometry_for(L, x, y, sx, sy);
    if (!g.in_border) return cur;
    // fill colour under the picture / on the border, straight alpha
    float af = U[U_OPACITY] * U[U_FILL + 3];
    float iaf = 1.f - af;
    float r0 = clampf(__builtin_fmaf(U[U_FILL + 2] * 255.0f, af, (float)(cur & 255) * iaf), 0.f, 255.f);          // B
    float r1 = clampf(__builtin_fmaf(U[U_FILL + 1] * 255.0f, af, (float)((cur >> 8) & 255) * iaf), 0.f, 255.f);   // G
    float r2 = clampf(__builtin_fmaf(U[U_FILL + 0] * 255.0f, af, (float)((cur >> 16) & 255) * iaf), 0.f, 255.f);  // R
    if (g.in_tx && g.in_uv) {
        float p0, p1, p2, a;
        if (L.kind == LK_BGRA_FROM_RGB) {
            const DPlane &P = L.src.pl[0];
            Lin2 l = lin_setup(P, g.u, g.v);
            uint32_t t00 = gld<uint32_t>(P.ptr + l.o00), t10 = gld<uint32_t>(P.ptr + l.o10);
            uint32_t t01 = gld<uint32_t>(P.ptr + l.o01), t11 = gld<uint32_t>(P.ptr + l.o11);
            float q0 = cs_mix(l, (float)(t00 & 255), (float)(t10 & 255), (float)(t01 & 255), (float)(t11 & 255));
            float q1 = cs_mix(l, (float)((t00 >> 8) & 255), (float)((t10 >> 8) & 255),
                              (float)((t01 >> 8) & 255), (float)((t11 >> 8) & 255));
            float q2 = cs_mix(l, (float)((t00 >> 16) & 255), (float)((t10 >> 16) & 255),
                              (float)((t01 >> 16) & 255), (float)((t11 >> 16) & 255));
            float q3 = cs_mix(l, (float)(t00 >> 24), (float)(t10 >> 24), (float)(t01 >> 24), (float)(t11 >> 24));
            p0 = L.swizzle ? q2 : q0; p1 = q1; p2 = L.swizzle ? q0 : q2;
            a = q3 * (U[U_OPACITY] * kInv255);
        } else {
            Lin2 ly = lin_setup(L.src.pl[0], g.u, g.v);
            Lin2 lc = lin_setup(L.src.pl[1], g.u, g.v);
            float fy = cs_fetch(L.src.pl[0], ly, 0), fu, fv;
            if (L.kind == LK_BGRA_FROM_NV12) {
                fu = cs_fetch(L.src.pl[1], lc, 0);
                fv = cs_fetch(L.src.pl[1], lc, 1);
            } else {
                fu = cs_fetch(L.src.pl[1], lc, 0);
                Lin2 lv = lin_setup(L.src.pl[2], g.u, g.v);
                fv = cs_fetch(L.src.pl[2], lv, 0);
            }
            uint32_t w = yuv_to_bgra_word(kCsc[L.csc & 3], (int)to_code_raw(fy), (int)to_code_raw(fu), (int)to_code_raw(fv));
            p0 = (float)(w & 255); p1 = (float)((w >> 8) & 255); p2 = (float)((w >> 16) & 255);
            a = 1.0f * U[U_OPACITY];
        }
        float ia = 1.f - a;
        r0 = __builtin_fmaf(p0, a, r0 * ia);
        r1 = __builtin_fmaf(p1, a, r1 * ia);
        r2 = __builtin_fmaf(p2, a, r2 * ia);
    }
    return pack_codes(r0, r1, r2, 0xFF000000u);
}

}  // namespace chv

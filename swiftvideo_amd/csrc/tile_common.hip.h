// tile_common.hip.h — building blocks shared by the axis-aligned LDS-staged kernels (kernels_fast.hip.cpp: one YUV layer ->
// BGRA, block-tiled; wave_common.hip.h / kernels_wave*.hip.cpp: N layers, one wave per strip): staging of source rectangles
// with CLAMP_TO_EDGE resolved at staging time, per-axis table entries, the LDS / global YUV samplers.
#pragma once
#include "pixel_math.hip.h"

#pragma clang fp contract(off)

namespace chv {

constexpr int NTHREADS = 256;
constexpr int LDS_BUDGET = 64 * 1024;

enum { AX_BORDER = 1, AX_TX = 2, AX_UV = 4, AX_ALL = 7 };


// Summarise a group of `1 << gshift` consecutive lanes (one wave of columns: 64; one tile's
// rows: 16).  Positions are monotone in the pixel index (every step of the coordinate
// arithmetic is a monotone rounding of a monotone function) and the "fully inside"
// entries form an interval, so the extremes sit at its first and last lane.
CHV_DEV void group_summary(int *out, int gshift, bool in_canvas, int fl, int iy, int ic) {
    const int lane = threadIdx.x & 63;
    const int g0 = (lane >> gshift) << gshift;
    const unsigned long long gmask = (gshift == 6) ? ~0ull : (((1ull << (1 << gshift)) - 1ull) << g0);
    const unsigned long long valid = __ballot(in_canvas && fl == AX_ALL) & gmask;
    const unsigned long long partial = __ballot(in_canvas && fl != AX_ALL) & gmask;
    int first = valid ? __ffsll((long long)valid) - 1 : g0;
    int last = valid ? 63 - __clzll((long long)valid) : g0;
    int y_a = __shfl(iy, first), y_b = __shfl(iy, last);
    int c_a = __shfl(ic, first), c_b = __shfl(ic, last);
    if (lane == g0) {
        out[0] = valid ? min(y_a, y_b) : 0x7fffffff;
        out[1] = valid ? max(y_a, y_b) + 1 : -0x7fffffff;
        out[2] = valid ? min(c_a, c_b) : 0x7fffffff;
        out[3] = valid ? max(c_a, c_b) + 1 : -0x7fffffff;
        out[4] = valid != 0;
        out[5] = partial == 0;
    }
}

// unclamped tap-0 position and weight of one axis of the linear filter (cf. lin_axis)
CHV_DEV void lin_axis_raw(float s, int w, int &i0, float &a) {
    float um = s * (float)w - 0.5f;
    float fl = __builtin_floorf(um);
    a = um - fl;
    i0 = (int)fl;
}

// the four bytes of a texel as floats on the code scale (0..255, exact): the BGRA-target family samples code
// values (pixel_math.hip.h), so a tap is a plain v_cvt_f32_ubyteN per byte
CHV_DEV float4 codes4(uint32_t w) {
    return make_float4((float)(w & 255), (float)((w >> 8) & 255), (float)((w >> 16) & 255), (float)(w >> 24));
}

// Does the 16-byte vector at byte offset `off` of row `row` lie inside the plane's allocation as one
// aligned load?  Not when it starts outside the row, and not when it would run past the end of the LAST
// row (planes on the tiled paths have base and pitch 16-byte aligned and rows of >= 16 bytes, host-checked).
CHV_DEV bool vec_loadable(const DPlane &P, int row, int off) {
    const int row_bytes = P.w * P.comps;
    return off >= 0 && off < row_bytes && (row < P.h - 1 || off + 16 <= row_bytes);
}
// the last row's tail vector, read bytewise (bytes past the row's payload are don't-care)
CHV_DEV uint4 load_tail_vec(const DPlane &P, int row, int off) {
    const int row_bytes = P.w * P.comps;
    const uint8_t *s = P.ptr + (size_t)row * P.pitch + off;
    uint32_t w[4] = { 0, 0, 0, 0 };
    for (int k = 0; k < 16 && off + k < row_bytes; k++) w[k >> 2] |= (uint32_t)gld<uint8_t>(s + k) << ((k & 3) * 8);
    return make_uint4(w[0], w[1], w[2], w[3]);
}

// Replace the texels of a 16-byte vector that lie outside the row (left of texel 0 when
// the vector is the padding vector, at or beyond the row end otherwise) by the nearest
// edge texel: CLAMP_TO_EDGE resolved once at staging time.  BPT = bytes per texel (1, 2, 4).
template <int BPT>
CHV_DEV uint4 patch_edges(uint4 val, const DPlane &P, int row, int off) {
    const int row_bytes = P.w * BPT;
    if (off >= 0 && off + 16 <= row_bytes) return val;
    const uint8_t *s = P.ptr + (size_t)row * P.pitch;
    uint32_t w[4] = { val.x, val.y, val.z, val.w };
    if (off < 0) {
        // padding vector in front of texel 0: only its last texel slot is ever addressed
        uint32_t e = BPT == 1 ? (uint32_t)gld<uint8_t>(s) << 24 : BPT == 2 ? (uint32_t)gld<uint16_t>(s) << 16 : gld<uint32_t>(s);
        w[3] = (w[3] & (BPT == 1 ? 0x00FFFFFFu : BPT == 2 ? 0x0000FFFFu : 0u)) | e;
    } else {
        uint32_t e = BPT == 1 ? (uint32_t)gld<uint8_t>(s + row_bytes - 1) * 0x01010101u
                   : BPT == 2 ? (uint32_t)gld<uint16_t>(s + row_bytes - 2) * 0x00010001u
                              : gld<uint32_t>(s + row_bytes - 4);
        int nvalid = max(row_bytes - off, 0);     // bytes of this vector inside the row
#pragma unroll
        for (int d = 0; d < 4; d++) {
            int nb = min(max(nvalid - 4 * d, 0), 4);
            uint32_t mask = nb >= 4 ? 0xFFFFFFFFu : ((1u << (8 * nb)) - 1u);
            w[d] = (w[d] & mask) | (e & ~mask);
        }
    }
    return make_uint4(w[0], w[1], w[2], w[3]);
}

// Staging of one plane's source rectangle, split in two so that the global loads of the
// NEXT tile are in flight while the current tile is being computed:
//   stage_load : raw 16-byte vectors -> registers (no dependent instruction)
//   stage_store: CLAMP_TO_EDGE patching, LDS write
// Slot i = tid + n * NTHREADS maps to row i / nslot, vector i % nslot of that row; LDS row r
// holds source row clamp(r_lo + r).  With `edge` (block-uniform: the rectangle touches a
// picture edge) vectors -1 .. nvec are staged, with the outside texels replicated.
struct StageGeom {
    int r_lo, rows;     // first (unclamped) source row, number of LDS rows
    int b0;             // first source byte of vector 0 (16-byte aligned)
    int nvec;           // vectors that hold picture bytes
    int nslot;          // slots per row = vectors staged per row (nvec, or nvec + 2 with `edge`)
    int inv20;          // ceil(2^20 / nslot): slot i -> row (i * inv20) >> 20, exact for i < 1024, nslot < 1024
    int edge;
};
CHV_DEV void stage_slots_init(StageGeom &g) {
    g.nslot = g.edge ? g.nvec + 2 : g.nvec;
    g.inv20 = ((1 << 20) + g.nslot - 1) / g.nslot;
}
CHV_DEV int stage_slots(const StageGeom &g) { return g.rows * g.nslot; }
CHV_DEV void stage_slot(const StageGeom &g, int i, int &r, int &vv) {
    r = (int)(((unsigned)i * (unsigned)g.inv20) >> 20);
    vv = i - r * g.nslot;
}

template <int N>
CHV_DEV void stage_load(uint4 (&regs)[N], const DPlane &P, const StageGeom &g, int tid, int base = 0) {
    // Exactly one global_load_dwordx4 per slot, straight into its final register: control flow that merges
    // differently-produced values here makes the compiler copy — and therefore wait for — the loaded
    // registers on the spot.  Vectors that are not loadable as such (outside the row: their texels are
    // replaced by patch_edges anyway; the last row's tail: re-read bytewise in stage_store) load the row's
    // first vector instead.
    // (Issuing these loads from inline asm — untracked, with hand-placed waits — measured 4 % faster on cfg2, and a
    // static check of the generated code then found the register allocator copying half of a destination vector
    // while its load was still in flight.  The loads stay tracked; callers use touch_regs() to place the wait.)
#pragma unroll
    for (int n = 0; n < N; n++) {
        int i = base + tid + n * NTHREADS, r, vv;
        stage_slot(g, i, r, vv);
        if (i < 1024 && r < g.rows) {
            int row = min(max(g.r_lo + r, 0), P.h - 1);
            int off = g.b0 + (g.edge ? vv - 1 : vv) * 16;
            if (g.edge) off = vec_loadable(P, row, off) ? off : 0;
            regs[n] = gld<uint4>(P.ptr + (size_t)row * P.pitch + off);
        }
    }
}

// BPT = bytes per source texel (1, 2, 4; selects the edge patching).  The 16 source bytes stay bytes: LDS byte 16 + k of
// a row = source byte b0 + k (float tiles were measured and dropped for every format, profiles/r01_notes.md).
// swap02 (BPT = 4): bytes 0 and 2 of every texel are exchanged on the way (RGBA sources staged as BGRA)
template <int BPT, int N>
CHV_DEV void stage_store(const uint4 (&regs)[N], uint8_t *lds, int lds_pitch, const DPlane &P, const StageGeom &g, int tid, int base = 0,
                         bool swap02 = false) {
#pragma unroll
    for (int n = 0; n < N; n++) {
        int i = base + tid + n * NTHREADS, r, vv;
        stage_slot(g, i, r, vv);
        if (i < 1024 && r < g.rows) {
            int v = g.edge ? vv - 1 : vv;
            uint4 val = regs[n];
            if (g.edge) {
                int row = min(max(g.r_lo + r, 0), P.h - 1);
                int off = g.b0 + v * 16;
                if (off >= 0 && off < P.w * BPT && !vec_loadable(P, row, off)) val = load_tail_vec(P, row, off);
                val = patch_edges<BPT>(val, P, row, off);
            }
            if (BPT == 4 && swap02) {     // RGBA texels become BGRA in LDS (block-uniform): the tap loops need no channel select
                val.x = __builtin_amdgcn_perm(val.x, val.x, 0x03000102u); val.y = __builtin_amdgcn_perm(val.y, val.y, 0x03000102u);
                val.z = __builtin_amdgcn_perm(val.z, val.z, 0x03000102u); val.w = __builtin_amdgcn_perm(val.w, val.w, 0x03000102u);
            }
            *(uint4 *)(lds + r * lds_pitch + 16 + v * 16) = val;
        }
    }
}

// Slots beyond the prefetch registers' capacity (rectangles at a picture edge carry two padding vectors per row):
// loaded and written on the spot, latency exposed — rare, and still one HBM read per source byte.
template <int BPT>
CHV_DEV void stage_tail(uint8_t *lds, int lds_pitch, const DPlane &P, const StageGeom &g, int tid, int first) {
    for (int base = first; base < stage_slots(g); base += NTHREADS) {
        uint4 t[1] = { make_uint4(0, 0, 0, 0) };
        stage_load(t, P, g, tid, base);
        stage_store<BPT>(t, lds, lds_pitch, P, g, tid, base);
    }
}

// ---- per-axis table entries and YUV samplers shared by the YUV -> BGRA tiled kernels ----------------------------
// x-dependent half of `geometry` + the sampler's x axis, evaluated at row 0 (under axis
// alignment the x components do not depend on y; signs of zero apart, which no later
// operation observes)
CHV_DEV void axis_entry_x(const float *__restrict__ U, int x, float sx, float sy, int wy, int wc,
                          int &iy, float &ay, int &ic, float &ac, int &flags) {
    float ou = (float)x / sx, ov = 0.0f / sy;
    float nx = ou * 2.f - 1.f, ny = ov * 2.f - 1.f;
    float t0 = dot4(nx, ny, 0.f, 1.f, U + U_TRANSFORM + 0);
    float t1 = dot4(nx, ny, 0.f, 1.f, U + U_TRANSFORM + 4);
    float t2 = dot4(nx, ny, 0.f, 1.f, U + U_TRANSFORM + 8);
    float t3 = dot4(nx, ny, 0.f, 1.f, U + U_TRANSFORM + 12);
    float b0 = dot4(nx, ny, 0.f, 1.f, U + U_BORDER + 0);
    float u = dot4(t0, t1, t2, t3, U + U_TEXTURE + 0);
    flags = ((b0 >= 0.f && b0 <= 1.f) ? AX_BORDER : 0) | ((t0 >= 0.f && t0 <= 1.f) ? AX_TX : 0) |
            ((u >= 0.f && u <= 1.f) ? AX_UV : 0);
    lin_axis_raw(u, wy, iy, ay);
    lin_axis_raw(u, wc, ic, ac);
}
CHV_DEV void axis_entry_y(const float *__restrict__ U, int y, float sx, float sy, int hy, int hc,
                          int &iy, float &ay, int &ic, float &ac, int &flags) {
    float ou = 0.0f / sx, ov = (float)y / sy;
    float nx = ou * 2.f - 1.f, ny = ov * 2.f - 1.f;
    float t0 = dot4(nx, ny, 0.f, 1.f, U + U_TRANSFORM + 0);
    float t1 = dot4(nx, ny, 0.f, 1.f, U + U_TRANSFORM + 4);
    float t2 = dot4(nx, ny, 0.f, 1.f, U + U_TRANSFORM + 8);
    float t3 = dot4(nx, ny, 0.f, 1.f, U + U_TRANSFORM + 12);
    float b1 = dot4(nx, ny, 0.f, 1.f, U + U_BORDER + 4);
    float v = dot4(t0, t1, t2, t3, U + U_TEXTURE + 4);
    flags = ((b1 >= 0.f && b1 <= 1.f) ? AX_BORDER : 0) | ((t1 >= 0.f && t1 <= 1.f) ? AX_TX : 0) |
            ((v >= 0.f && v <= 1.f) ? AX_UV : 0);
    lin_axis_raw(v, hy, iy, ay);
    lin_axis_raw(v, hc, ic, ac);
}

// One pixel of the BGRA-target family from already-sampled, quantised YUV (code-scale
// arithmetic, pixel_math.hip.h): fill under the picture, then the picture itself.
CHV_DEV uint32_t blend_bgra_general(uint32_t c, const float *__restrict__ U, bool in_pic, uint32_t w) {
    const float opacity = U[U_OPACITY];
    const float af = opacity * U[U_FILL + 3];
    const float iaf = 1.f - af;
    float r0 = clampf(__builtin_fmaf(U[U_FILL + 2] * 255.0f, af, (float)(c & 255) * iaf), 0.f, 255.f);
    float r1 = clampf(__builtin_fmaf(U[U_FILL + 1] * 255.0f, af, (float)((c >> 8) & 255) * iaf), 0.f, 255.f);
    float r2 = clampf(__builtin_fmaf(U[U_FILL + 0] * 255.0f, af, (float)((c >> 16) & 255) * iaf), 0.f, 255.f);
    if (in_pic) {
        const float a = 1.0f * opacity, ia = 1.f - a;
        r0 = __builtin_fmaf((float)(w & 255), a, r0 * ia);
        r1 = __builtin_fmaf((float)((w >> 8) & 255), a, r1 * ia);
        r2 = __builtin_fmaf((float)((w >> 16) & 255), a, r2 * ia);
    }
    return pack_codes(r0, r1, r2, 0xFF000000u);
}

// NV12 sample at one pixel from the staged byte tiles, on the code scale: luma bytes, interleaved chroma (u | v << 8
// per texel); tap 1 is the next texel, the next row is one LDS pitch further.
CHV_DEV void sample_nv12_lds_bytes(const uint8_t *smem, int ya, int ypitch, int ca, int cpitch,
                                   float w00, float w10, float w01, float w11,
                                   float c00, float c10, float c01, float c11,
                                   float &fy, float &fu, float &fv) {
    const uint8_t *py = smem + ya;
    fy = cs_mix(w00, w10, w01, w11, (float)py[0], (float)py[1], (float)py[ypitch], (float)py[ypitch + 1]);
    const uint32_t q00 = *(const uint16_t *)(smem + ca), q10 = *(const uint16_t *)(smem + ca + 2);
    const uint32_t q01 = *(const uint16_t *)(smem + ca + cpitch), q11 = *(const uint16_t *)(smem + ca + cpitch + 2);
    fu = cs_mix(c00, c10, c01, c11, (float)(q00 & 255), (float)(q10 & 255), (float)(q01 & 255), (float)(q11 & 255));
    fv = cs_mix(c00, c10, c01, c11, (float)(q00 >> 8), (float)(q10 >> 8), (float)(q01 >> 8), (float)(q11 >> 8));
}

// The same sample with every tap read as its own byte and fed to v_fma_mix_f32 as a binary16 denormal (tap_h, pixel_math.hip.h):
// no conversion instructions, 12 instead of 8 LDS reads.  The eight weights arrive multiplied by kTapScale.
// CHV_TAPS (measurement): 0 = every tap its own byte read (12 LDS reads, no conversions); 1 = luma bytes through v_fma_mix_f32,
// chroma as four aligned (u, v) pair reads widened with v_cvt_f32_ubyte0/1 (8 LDS reads, 8 conversions); CHV_ABL & 8 (timing only):
// the chroma taps reuse the luma registers (no chroma LDS reads at all)
#ifndef CHV_TAPS
#define CHV_TAPS 0
#endif
#ifndef CHV_ABL
#define CHV_ABL_T 0
#else
#define CHV_ABL_T CHV_ABL
#endif
constexpr float kChromaTapScale = CHV_TAPS == 1 ? 1.0f : kTapScale;      // what sample_nv12_lds_mix expects its chroma weights multiplied by
CHV_DEV void sample_nv12_lds_mix(const uint8_t *smem, int ya, int ypitch, int ca, int cpitch,
                                 float w00, float w10, float w01, float w11,
                                 float c00, float c10, float c01, float c11,
                                 float &fy, float &fu, float &fv) {
    const uint8_t *py = smem + ya, *pc = smem + ca;
    const chv_half y00 = tap_h(py), y10 = tap_h(py + 1), y01 = tap_h(py + ypitch), y11 = tap_h(py + ypitch + 1);
    fy = cs_mix_h(w00, w10, w01, w11, y00, y10, y01, y11);
    if (CHV_ABL_T & 8) {
        fu = cs_mix_h(c00, c10, c01, c11, y10, y00, y11, y01);
        fv = cs_mix_h(c00, c10, c01, c11, y01, y11, y00, y10);
    } else if (CHV_TAPS == 1) {
        // (the chroma weights arrive UNSCALED in this mode: kChromaTapScale)
        const uint32_t q00 = *(const uint16_t *)pc, q10 = *(const uint16_t *)(pc + 2);
        const uint32_t q01 = *(const uint16_t *)(pc + cpitch), q11 = *(const uint16_t *)(pc + cpitch + 2);
        fu = cs_mix(c00, c10, c01, c11, (float)(q00 & 255), (float)(q10 & 255), (float)(q01 & 255), (float)(q11 & 255));
        fv = cs_mix(c00, c10, c01, c11, (float)(q00 >> 8), (float)(q10 >> 8), (float)(q01 >> 8), (float)(q11 >> 8));
    } else {
        fu = cs_mix_h(c00, c10, c01, c11, tap_h(pc), tap_h(pc + 2), tap_h(pc + cpitch), tap_h(pc + cpitch + 2));
        fv = cs_mix_h(c00, c10, c01, c11, tap_h(pc + 1), tap_h(pc + 3), tap_h(pc + cpitch + 1), tap_h(pc + cpitch + 3));
    }
}
CHV_DEV void sample_y420p_lds_mix(const uint8_t *smem, int ya, int ypitch, int ca, int voff, int cpitch,
                                  float w00, float w10, float w01, float w11,
                                  float c00, float c10, float c01, float c11,
                                  float &fy, float &fu, float &fv) {
    const uint8_t *py = smem + ya, *pu = smem + ca, *pv = smem + ca + voff;
    fy = cs_mix_h(w00, w10, w01, w11, tap_h(py), tap_h(py + 1), tap_h(py + ypitch), tap_h(py + ypitch + 1));
    fu = cs_mix_h(c00, c10, c01, c11, tap_h(pu), tap_h(pu + 1), tap_h(pu + cpitch), tap_h(pu + cpitch + 1));
    fv = cs_mix_h(c00, c10, c01, c11, tap_h(pv), tap_h(pv + 1), tap_h(pv + cpitch), tap_h(pv + cpitch + 1));
}

// the same sample straight from the source planes (tile did not fit the LDS budget);
// SV != nullptr: planar chroma (y420p: U from SC, V from SV), else interleaved (NV12)
CHV_DEV void sample_nv12_global(const DPlane &SY, const DPlane &SC, const DPlane *SV, int ix, int iy, int cx, int cy,
                                float w00, float w10, float w01, float w11,
                                float c00, float c10, float c01, float c11,
                                float &fy, float &fu, float &fv) {
    int x0 = min(max(ix, 0), SY.w - 1), x1 = min(max(ix + 1, 0), SY.w - 1);
    int y0 = min(max(iy, 0), SY.h - 1), y1 = min(max(iy + 1, 0), SY.h - 1);
    const uint8_t *p0 = SY.ptr + (size_t)y0 * SY.pitch, *p1 = SY.ptr + (size_t)y1 * SY.pitch;
    fy = cs_mix(w00, w10, w01, w11, (float)gld<uint8_t>(p0 + x0), (float)gld<uint8_t>(p0 + x1), (float)gld<uint8_t>(p1 + x0), (float)gld<uint8_t>(p1 + x1));
    int u0 = min(max(cx, 0), SC.w - 1), u1 = min(max(cx + 1, 0), SC.w - 1);
    int v0 = min(max(cy, 0), SC.h - 1), v1 = min(max(cy + 1, 0), SC.h - 1);
    const uint8_t *q0 = SC.ptr + (size_t)v0 * SC.pitch, *q1 = SC.ptr + (size_t)v1 * SC.pitch;
    uint32_t a00, a10, a01, a11;
    if (SV) {
        const uint8_t *z0 = SV->ptr + (size_t)v0 * SV->pitch, *z1 = SV->ptr + (size_t)v1 * SV->pitch;
        a00 = gld<uint8_t>(q0 + u0) | (gld<uint8_t>(z0 + u0) << 8); a10 = gld<uint8_t>(q0 + u1) | (gld<uint8_t>(z0 + u1) << 8);
        a01 = gld<uint8_t>(q1 + u0) | (gld<uint8_t>(z1 + u0) << 8); a11 = gld<uint8_t>(q1 + u1) | (gld<uint8_t>(z1 + u1) << 8);
    } else {
        a00 = gld<uint16_t>(q0 + u0 * 2); a10 = gld<uint16_t>(q0 + u1 * 2);
        a01 = gld<uint16_t>(q1 + u0 * 2); a11 = gld<uint16_t>(q1 + u1 * 2);
    }
    fu = cs_mix(c00, c10, c01, c11, (float)(a00 & 255), (float)(a10 & 255), (float)(a01 & 255), (float)(a11 & 255));
    fv = cs_mix(c00, c10, c01, c11, (float)(a00 >> 8), (float)(a10 >> 8), (float)(a01 >> 8), (float)(a11 >> 8));
}

// the same for planar chroma (y420p): U tile at `ca`, V tile `voff` bytes further, one byte per texel
CHV_DEV void sample_y420p_lds_bytes(const uint8_t *smem, int ya, int ypitch, int ca, int voff, int cpitch,
                                    float w00, float w10, float w01, float w11,
                                    float c00, float c10, float c01, float c11,
                                    float &fy, float &fu, float &fv) {
    const uint8_t *py = smem + ya, *pu = smem + ca, *pv = smem + ca + voff;
    fy = cs_mix(w00, w10, w01, w11, (float)py[0], (float)py[1], (float)py[ypitch], (float)py[ypitch + 1]);
    fu = cs_mix(c00, c10, c01, c11, (float)pu[0], (float)pu[1], (float)pu[cpitch], (float)pu[cpitch + 1]);
    fv = cs_mix(c00, c10, c01, c11, (float)pv[0], (float)pv[1], (float)pv[cpitch], (float)pv[cpitch + 1]);
}

}  // namespace chv

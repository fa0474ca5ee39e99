// tile_common.hip.h — building blocks shared by the axis-aligned LDS-tiled kernels
// (kernels_fast.hip.cpp: NV12 -> BGRA; kernels_fast_rgb.hip.cpp: BGRA/RGBA layers -> BGRA).
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

}  // namespace chv

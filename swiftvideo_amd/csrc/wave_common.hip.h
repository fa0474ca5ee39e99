// wave_common.hip.h — what the wave-per-strip kernels share (kernels_wave.hip.cpp: BGRA canvases;
// kernels_wave_yuv.hip.cpp: NV12 / y420p canvases): strip bookkeeping, per-layer geometry (column entry in registers, row
// entries in a wave-private LDS table), wave-private staging of the source rectangles.  See kernels_wave.hip.cpp for the
// design and the measurements behind it.
#pragma once
#include "tile_common.hip.h"

#pragma clang fp contract(off)

namespace chv {

constexpr int WTW = 64;                 // strip width: one lane per column
// 1: layers of a batch take their per-strip geometry from the batch's tables (geom_cache.h, WaveStrip::setup_cached); 0: always computed in place
#ifndef CHV_GEOM_CACHE
#define CHV_GEOM_CACHE 1
#endif

// Waves per block.  The waves of a block share nothing but the launch (every wave has its own LDS region and there is no block
// barrier on the data path), so the block size only sets the granularity at which LDS is handed out: one wave per block
// (1, 2 and 4 measured the same where LDS is not the limit).
#ifndef CHV_WAVE_WAVES
#define CHV_WAVE_WAVES 1
#endif
constexpr int WAVES = CHV_WAVE_WAVES;
constexpr int WAVE_BLOCK = WAVES * 64;
// Per strip height H (rows per lane; 8 for BGRA canvases, 16 for 4:2:0 canvases whose codes pack four to a register):
// staging registers (16-byte vectors) per lane, 64 slots each, sized for the rectangles of a 1.5x downscale (YUV) / of a
// native-resolution picture (RGB); larger rectangles finish through wstage_tail:
//   RGB layer: [0 .. WN_RGB) plane 0;   YUV layer: [0 .. WN_Y) luma, then WN_C for chroma / U, then WN_C for V (planar)
template <int H>
struct WaveCfg {
    static_assert(H == 8 || H == 16, "strip height: 8 or 16 rows");
    static constexpr int WN_Y = H / 4, WN_C = H / 8, WN_RGB = H / 4 + 1;
    static constexpr int WNR = (WN_RGB > WN_Y + 2 * WN_C) ? WN_RGB : (WN_Y + 2 * WN_C);
    // per wave: the current layer's row entries — 8 dwords per row (A, B: what the per-pixel paths read), then one compact
    // 4-dword entry per row for the branch-free row loops (see RowFast)
    static constexpr int ROWTAB_BYTES = H * 32 + H * 16;
};

// ---- wave-level helpers ---------------------------------------------------------------------------------------
// ---- a lone tick's descriptors as kernel ARGUMENTS ------------------------------------------------------------------------------------------------
// A transient launch (chv_composite, a held pass: what an unmodified VideoMixer issues) used to copy its descriptor slot to device memory in
// front of the kernel: a second command on the stream, 5 us of a 30 us tick.  The strip kernels are the library's largest instantiations, so
// instead of by-value twins (tick_bgra_stream_one) every instantiation of tick_yuv_wave carries a trailing argument `WaveOne one` that no code names: launched
// with `ticks == nullptr` the kernel reads its tick and layers from where that argument lies in the kernarg segment (constant-address-space
// memory the runtime wrote with the launch), through the same pointers and the same scalar loads as a batch's descriptors.
// (WaveOne: device_types.h)
// the kernels' explicit arguments as the kernarg segment lays them out (declaration order, natural alignment): where `one` starts
struct WaveKernArgs { const DTick *ticks; const DLayer *layers; int32_t n_ticks, strips_x, strips_y; uint32_t strips_magic, strips_x_magic; int32_t p0pitch, p0rows, p1pitch, p1rows, planar_any; WaveOne one; };
CHV_DEV void wave_one_descriptors(const DTick *__restrict__ &ticks, const DLayer *__restrict__ &layers) {
    if (ticks == nullptr) {
        const uint64_t ka = (uint64_t)(uintptr_t)__builtin_amdgcn_kernarg_segment_ptr() + offsetof(WaveKernArgs, one);
        ticks = (const DTick *)(const __attribute__((address_space(1))) DTick *)(uintptr_t)ka;
        layers = (const DLayer *)(const __attribute__((address_space(1))) DLayer *)(uintptr_t)(ka + sizeof(DTick));
    }
}

CHV_DEV void wave_lds_fence() {
    // The wave's own LDS writes are visible to its later reads (LDS operations of a wave execute in order); this only keeps
    // the compiler from moving LDS accesses across the point.  The fence names the LDS address space: a fence over all
    // memory makes every later read of the (read-only, uniform) tick and layer descriptors "possibly clobbered", and the
    // compiler then fetches them with per-lane global loads instead of scalar loads — 90 vector loads per wave and
    // +35 % run time on the 4:2:0 mixer workload (profiles/r02_notes.md).
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront", "local");
    __builtin_amdgcn_wave_barrier();
}
CHV_DEV int rl(int v, int lane) { return __builtin_amdgcn_readlane(v, lane); }
CHV_DEV float rlf(float v, int lane) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), lane)); }

// Summary of the entries of `mask`'s lanes (columns: all 64; rows: lanes 0..WTH-1): positions are monotone in the pixel
// index and the "fully inside" entries form an interval, so the extremes sit at its first and last lane (cf.
// group_summary, tile_common.hip.h).  All results are wave-uniform (scalar registers).
struct AxisSum { int lo, hi, clo, chi; bool any, all; };
CHV_DEV AxisSum axis_summary(unsigned long long mask, bool in_canvas, int fl, int iy, int ic) {
    const unsigned long long valid = __ballot(in_canvas && fl == AX_ALL) & mask;
    const unsigned long long partial = __ballot(in_canvas && fl != AX_ALL) & mask;
    AxisSum s;
    s.any = valid != 0; s.all = partial == 0;
    const int first = s.any ? __ffsll((long long)valid) - 1 : 0;
    const int last = s.any ? 63 - __clzll((long long)valid) : 0;
    const int ya = rl(iy, first), yb = rl(iy, last), ca = rl(ic, first), cb = rl(ic, last);
    s.lo = min(ya, yb); s.hi = max(ya, yb) + 1;
    s.clo = min(ca, cb); s.chi = max(ca, cb) + 1;
    return s;
}

// per-layer state of a strip: column entry (every lane) and staging geometry (uniform); the row entries live in LDS
// A strip's staging rectangle.  `pair` (uniform): the rectangle holds, for every row j of the strip, that row's OWN two tap rows at LDS rows
// 2 j and 2 j + 1 (pair_ry: lane j's first tap row) instead of the contiguous source rows r_lo .. r_lo + rows - 1 — for vertical reductions
// beyond ~1.6:1, where consecutive canvas rows share no tap row and a contiguous rectangle would stage (and fetch) rows nothing reads:
// 16 instead of 27 luma rows for an 8-row strip at 3:1, which is LDS (waves per SIMD) and HBM traffic.
struct WGeom : StageGeom {
    int pair;
    int r_hi1;                   // pair form: the last first-tap row the rectangle was validated for (clamp of a row's own tap row)
};
#ifndef CHV_WAVE_PAIR
#define CHV_WAVE_PAIR 1
#endif
// source row of LDS row r (before CLAMP_TO_EDGE).  Pair form (PAIR instantiations only: stage_impl<EDGE, true>): `pry` holds, in lane j, strip row
// j's first tap row; executed by ALL lanes (a lane shuffle).  (Kept out of the other instantiations altogether — and free of pointers: a first
// version that looked the rows up in the LDS row table through a pointer argument made hipcc fetch every plane descriptor field with vector
// loads, each behind a full wait: cfg3 1.28 -> 2.43 ms.)
template <bool PAIR>
CHV_DEV int wstage_row(const WGeom &g, int r, int pry) {
    if constexpr (!PAIR) return g.r_lo + r;
    else {
        const int own = __builtin_amdgcn_ds_bpermute(((r >> 1) & 63) * 4, pry) + (r & 1);
        return g.pair ? own : g.r_lo + r;
    }
}

struct WLayer {
    int cyo, cco;            // staged layers: LDS byte offset of tap 0 inside a staged row (luma / RGB texel, chroma);
                             // unstaged layers: the unclamped tap-0 texel positions themselves
    float cya, cca;          // weight of tap 1
    int cfl;
    WGeom g0, g1;
    bool staged, all_inside;
    bool unit_rows;          // staged, and tap row 0 of every row of the strip is tap row 1 of the row above (source rows advance
                             // one per canvas row: native-resolution layers) — the row loops then convert every texel row once
};
// row entry in the wave's LDS table: two 16-byte halves
//   A = {yoff, coff, rfl, -}: LDS byte offset of the row's first tap row in the plane-0 / chroma rectangle (unstaged: the
//                              unclamped tap-0 row positions), flags
//   B = {yb, 1 - yb, cb, 1 - cb}: weights of tap row 1 (luma / RGB, chroma) and their complements

// The compact entry the branch-free row loops read: C = {yb, cb, yoff | coff << 16, rfl} at rowtab[2 H + j] (staged layers: both
// offsets are LDS byte offsets < 2^16).  A row then costs one ds_read_b64 and two ds_read_u16 (three dwords written to the register
// file per lane) instead of two broadcast ds_read_b128 (eight): LDS return data competes with the VALU for the register file's write
// ports, and the two wide reads per row measured ~9 ns each per wave on a SIMD (profiles/r03_notes.md) — a quarter of the pipeline
// tick's time.  The complements 1 - yb, 1 - cb are recomputed (one v_sub_f32 each: the operation that produced the table's values).
struct RowFast { float yb, iyb, cb, icb; int yoff, coff; };
template <int H, bool CHROMA>
CHV_DEV RowFast row_fast(const uint4 *rowtab, int j) {
    const uint8_t *e = (const uint8_t *)(rowtab + 2 * H + j);
    RowFast r;
    if constexpr (CHROMA) {
        const float2 w = *(const float2 *)e;
        r.yb = w.x; r.cb = w.y;
        r.coff = (int)((const uint16_t *)e)[5];
    } else {
        r.yb = *(const float *)e; r.cb = 0.f; r.coff = 0;
    }
    r.yoff = (int)((const uint16_t *)e)[4];
    r.iyb = 1.0f - r.yb; r.icb = 1.0f - r.cb;
    return r;
}
template <int H>
CHV_DEV uint32_t row_fast_flags(const uint4 *rowtab, int j) { return ((const uint32_t *)(rowtab + 2 * H + j))[3]; }

// EDGE = false: the caller knows (uniformly) that the rectangle touches no picture edge — no clamping, no patching, no
// padding vector: the compact instantiation most strips run
template <int OFF, int N, int NR, bool EDGE, bool PAIR>
CHV_DEV void wstage_load(uint4 (&regs)[NR], const DPlane &P, const WGeom &g, int lane, int pry) {
    // exactly one global_load_dwordx4 per slot, straight into its final register (see stage_load, tile_common.hip.h)
#pragma unroll
    for (int n = 0; n < N; n++) {
        int i = lane + n * 64, r, vv;
        stage_slot(g, i, r, vv);
        const int srow = wstage_row<PAIR>(g, r, pry);
        if (r < g.rows) {
            if constexpr (EDGE) {
                int row = min(max(srow, 0), P.h - 1);
                int off = g.b0 + (g.edge ? vv - 1 : vv) * 16;
                if (g.edge) off = vec_loadable(P, row, off) ? off : 0;
                regs[OFF + n] = gld<uint4>(P.ptr + (size_t)row * P.pitch + off);
            } else {
                regs[OFF + n] = gld<uint4>(P.ptr + (size_t)srow * P.pitch + (g.b0 + vv * 16));
            }
        }
    }
}
// one slot: CLAMP_TO_EDGE patching (edge rectangles only), optional RGBA -> BGRA, LDS write
template <int BPT, bool EDGE, bool PAIR>
CHV_DEV void wstage_put(uint4 val, int i, uint8_t *lds, int lds_pitch, const DPlane &P, const WGeom &g, bool swap02, int pry) {
    int r, vv;
    stage_slot(g, i, r, vv);
    const int srow = wstage_row<PAIR>(g, r, pry);
    if (i < 1024 && r < g.rows) {
        int v = (EDGE && g.edge) ? vv - 1 : vv;
        if (EDGE && g.edge) {
            int row = min(max(srow, 0), P.h - 1);
            int off = g.b0 + v * 16;
            if (off >= 0 && off < P.w * BPT && !vec_loadable(P, row, off)) val = load_tail_vec(P, row, off);
            val = patch_edges<BPT>(val, P, row, off);
        }
        if (BPT == 4 && swap02) {
            val.x = __builtin_amdgcn_perm(val.x, val.x, 0x03000102u); val.y = __builtin_amdgcn_perm(val.y, val.y, 0x03000102u);
            val.z = __builtin_amdgcn_perm(val.z, val.z, 0x03000102u); val.w = __builtin_amdgcn_perm(val.w, val.w, 0x03000102u);
        }
        *(uint4 *)(lds + r * lds_pitch + 16 + v * 16) = val;
    }
}
template <int BPT, int OFF, int N, int NR, bool EDGE, bool PAIR>
CHV_DEV void wstage_store(const uint4 (&regs)[NR], uint8_t *lds, int lds_pitch, const DPlane &P, const WGeom &g, int lane, bool swap02, int pry) {
#pragma unroll
    for (int n = 0; n < N; n++) wstage_put<BPT, EDGE, PAIR>(regs[OFF + n], lane + n * 64, lds, lds_pitch, P, g, swap02, pry);
}
// Slots beyond the registers' share of a plane (stronger downscales, rectangles at a picture edge): further rounds of
// WTAIL loads in flight, one wait, WTAIL LDS writes (not unrolled beyond that: the edge patching is large code).
#ifndef CHV_WTAIL
#define CHV_WTAIL 2
#endif
constexpr int WTAIL = CHV_WTAIL;
template <int BPT, int N, bool EDGE, bool PAIR>
CHV_DEV void wstage_tail(uint8_t *lds, int lds_pitch, const DPlane &P, const WGeom &g, int lane, bool swap02, int pry) {
#pragma unroll 1
    for (int base = N * 64; base < stage_slots(g); base += WTAIL * 64) {
        uint4 t[WTAIL];
#pragma unroll
        for (int n = 0; n < WTAIL; n++) {
            int i = base + n * 64 + lane, r, vv;
            stage_slot(g, i, r, vv);
            t[n] = make_uint4(0, 0, 0, 0);
            const int srow = wstage_row<PAIR>(g, r, pry);
            if (i < 1024 && r < g.rows) {
                if constexpr (EDGE) {
                    int row = min(max(srow, 0), P.h - 1);
                    int off = g.b0 + (g.edge ? vv - 1 : vv) * 16;
                    if (g.edge) off = vec_loadable(P, row, off) ? off : 0;
                    t[n] = gld<uint4>(P.ptr + (size_t)row * P.pitch + off);
                } else {
                    t[n] = gld<uint4>(P.ptr + (size_t)srow * P.pitch + (g.b0 + vv * 16));
                }
            }
        }
        if constexpr (WTAIL == 2) {
#pragma unroll 1
            for (int n = 0; n < WTAIL; n++) wstage_put<BPT, EDGE, PAIR>(n == 0 ? t[0] : t[WTAIL - 1], base + n * 64 + lane, lds, lds_pitch, P, g, swap02, pry);
        } else {
#pragma unroll
            for (int n = 0; n < WTAIL; n++) wstage_put<BPT, EDGE, PAIR>(t[n], base + n * 64 + lane, lds, lds_pitch, P, g, swap02, pry);
        }
    }
}

// the same for the rest of TWO rectangles at once (an NV12 picture's luma and chroma): WTAIL loads of each in flight, one wait per round
#ifndef CHV_WAVE_NV12_WIDE
#define CHV_WAVE_NV12_WIDE 1
#endif
template <bool EDGE, bool PAIR>
CHV_DEV uint4 wstage_tail_load(const DPlane &P, const WGeom &g, int i, int pry) {
    int r, vv;
    stage_slot(g, i, r, vv);
    uint4 t = make_uint4(0, 0, 0, 0);
    const int srow = wstage_row<PAIR>(g, r, pry);
    if (i < 1024 && r < g.rows) {
        if constexpr (EDGE) {
            int row = min(max(srow, 0), P.h - 1);
            int off = g.b0 + (g.edge ? vv - 1 : vv) * 16;
            if (g.edge) off = vec_loadable(P, row, off) ? off : 0;
            t = gld<uint4>(P.ptr + (size_t)row * P.pitch + off);
        } else {
            t = gld<uint4>(P.ptr + (size_t)srow * P.pitch + (g.b0 + vv * 16));
        }
    }
    return t;
}
template <int BPT0, int BPT1, int N0, int N1, bool EDGE, bool PAIR>
CHV_DEV void wstage_tail2(uint8_t *lds0, int pitch0, const DPlane &P0, const WGeom &g0, int pry0,
                          uint8_t *lds1, int pitch1, const DPlane &P1, const WGeom &g1, int pry1, int lane) {
    const int n0 = stage_slots(g0), n1 = stage_slots(g1);
#pragma unroll 1
    for (int b0 = N0 * 64, b1 = N1 * 64; b0 < n0 || b1 < n1; b0 += WTAIL * 64, b1 += WTAIL * 64) {
        uint4 t0[WTAIL], t1[WTAIL];
#pragma unroll
        for (int n = 0; n < WTAIL; n++) {
            t0[n] = wstage_tail_load<EDGE, PAIR>(P0, g0, min(b0 + n * 64 + lane, 1024), pry0);
            t1[n] = wstage_tail_load<EDGE, PAIR>(P1, g1, min(b1 + n * 64 + lane, 1024), pry1);
        }
#pragma unroll
        for (int n = 0; n < WTAIL; n++) {
            wstage_put<BPT0, EDGE, PAIR>(t0[n], min(b0 + n * 64 + lane, 1024), lds0, pitch0, P0, g0, false, pry0);
            wstage_put<BPT1, EDGE, PAIR>(t1[n], min(b1 + n * 64 + lane, 1024), lds1, pitch1, P1, g1, false, pry1);
        }
    }
}

// ---- shift-and-mask staging of narrow interior rectangles ------------------------------------------------------------
// Rectangles that touch no picture edge and are at most 8 vectors wide (luma and chroma of YUV pictures up to ~1.9x downscale: the
// pipeline, cfg2, every native-resolution video layer): lane -> (row of the round, vector) is lane >> sh, lane & mask with
// 2^sh >= nvec, a round covers 64 >> sh rows.  No slot division, no bounds branches: lanes beyond the last vector / the last
// row are clamped onto it and load and write the same bytes as their neighbour.  ~7 VALU instructions per slot where the general
// slot map (stage_slot: two multiplies, bounds tests, 64-bit addresses) costs ~20; the wave kernels are VALU-bound
// (profiles/r03_notes.md), so this is run time.
struct P2Map { int rsub, vcol, rstep, ok; };
template <int N>
CHV_DEV P2Map p2_map(const WGeom &g, int lane) {
    // sh = ceil(log2(nvec)), uniform
    const int sh = g.nvec <= 1 ? 0 : 32 - __builtin_clz((unsigned)(g.nvec - 1));
    P2Map m;
    m.rstep = 64 >> sh;
    m.ok = sh <= 6 && N * m.rstep >= g.rows && !g.pair;
    m.rsub = lane >> sh;
    m.vcol = min(lane & ((1 << sh) - 1), g.nvec - 1) * 16;
    return m;
}
template <int OFF, int N, int NR, bool SKIP>
CHV_DEV void wstage_load_p2(uint4 (&regs)[NR], const DPlane &P, const WGeom &g, const P2Map &m) {
    const uint8_t *base = P.ptr + (size_t)g.r_lo * P.pitch + g.b0;          // (uniform: scalar unit)
#pragma unroll
    for (int n = 0; n < N; n++) {
        // (SKIP — INTERIOR bit 3, off everywhere: rounds past the rectangle's last row skipped by a uniform branch.  The branches cost
        // more than the duplicate loads they save: pipeline 1.514 -> 1.546 ms, mixed 0.743 -> 0.783, mixer_nv12 0.637 -> 0.664)
        if (SKIP && n > 0 && n * m.rstep >= g.rows) break;
        const int r = min(m.rsub + n * m.rstep, g.rows - 1);
        regs[OFF + n] = gld_at<uint4>(base, __umul24((uint32_t)r, (uint32_t)P.pitch) + (uint32_t)m.vcol);       // (24-bit multiply: v_mul_lo_u32 issues at a quarter of the rate; pitches < 2^24 on this path, host-checked)
    }
}
template <int OFF, int N, int NR, bool SKIP>
CHV_DEV void wstage_store_p2(const uint4 (&regs)[NR], uint8_t *lds, int lds_pitch, const WGeom &g, const P2Map &m) {
#pragma unroll
    for (int n = 0; n < N; n++) {
        if (SKIP && n > 0 && n * m.rstep >= g.rows) break;
        const int r = min(m.rsub + n * m.rstep, g.rows - 1);
        *(uint4 *)(lds + (__umul24((uint32_t)r, (uint32_t)lds_pitch) + 16u + (uint32_t)m.vcol)) = regs[OFF + n];
    }
}

CHV_DEV float ub0(uint32_t w) { return (float)(w & 255u); }
CHV_DEV float ub1(uint32_t w) { return (float)((w >> 8) & 255u); }
CHV_DEV float ub2(uint32_t w) { return (float)((w >> 16) & 255u); }
CHV_DEV float ub3(uint32_t w) { return (float)(w >> 24); }


// source classes of a layer, whatever the canvas: one 4-byte plane; luma + interleaved chroma; three planes
CHV_DEV bool src_is_rgb(int kind) { return kind == LK_BGRA_FROM_RGB || kind == LK_YUV_FROM_RGB || kind == LK_YUV_FROM_RGB_INT; }
CHV_DEV bool src_is_planar(int kind) { return kind == LK_BGRA_FROM_Y420P || kind == LK_YUV_FROM_Y420P; }

// One wave's strip of one tick: WTW columns (lane = column) x WTH rows of the canvas.
// INTERIOR: bit 0 — YUV-source rectangles, bit 1 — RGB-source rectangles that touch no picture edge are staged by the compact
// instantiation (stage_impl<false>); a measured choice per kernel (kernels_wave.hip.cpp)
// KINDS: the source classes the launch contains (bit 0 NV12, bit 1 y420p, bit 2 RGB; host-checked).  A launch of one class runs
// an instantiation that holds no code for the others — the kernels are 70-100 KB of code, and the executed footprint counts
// (the NV12-only instantiation: pipeline -3.4 %).
// ---- geometry tables (geom_cache.h): what setup() derives for a layer, stored per strip column / per strip row / per strip -------------
// One table per geometry class of a batch (same matrices, source plane sizes, canvas size), built by geom_precompute (kernels_wave_yuv.hip.cpp)
// with setup() itself and read back by setup_cached() in the CACHED instantiations of the tick kernels — which contain no set-up code at all.
// A table covers every strip of its canvas, staged or not.  A DLayer carries its class's table address in pad2 (0: none).
struct GeomHdr { int32_t strips_x, strips_y, wth, row_bytes; uint32_t flags_off, cols_off, rows_off, pad; };     // 32 bytes, at the table's base
struct GeomCol {                       // a strip column: the lanes' column entries and the column halves of the rectangles
    uint4 a[64];                       // staged form: { cyo, cco, bits(cya), bits(cca) }
    uint4 b[64];                       // { tap-0 column of the unstaged form (luma / RGB), (chroma), cfl, - }
    int32_t s[16];                     // g0.b0, g0.nvec, inv20(nvec), inv20(nvec + 2), then the same four of g1
};
// a strip row (row_bytes each): uint4 rowtab[3 * WTH] as setup() leaves it in LDS for a STAGED layer, the same for an unstaged one (row
// positions instead of LDS offsets), then int32 s[16]: g0.r_lo, g0.rows, g0.pair, g0.r_hi1, the same four of g1, unit_rows
enum : uint32_t { GF_STAGED = 1, GF_ALL_INSIDE = 2, GF_EDGE0 = 4, GF_EDGE1 = 8 };       // the flag word of a strip (row-major, strips_x per row)
struct GeomRaw { int cy, cc, ry, rc, rfl; float rya, rca; };      // (setup<true>: what the unstaged form is made of)
struct GeomJob {                       // one class for geom_precompute: a representative layer (its plane POINTERS are not used), the canvas, the table
    DLayer layer;
    int32_t W, H, strips_x, strips_y, first_block, pad;
    uint8_t *table;
};

template <int WTH, int INTERIOR = 0, int KINDS = 7>
struct WaveStrip {
    // (bit 3: the launch also has layers that are not staged at all — any transform, applied per pixel by the kernels)
    static CHV_DEV bool is_rgb(int kind) { return (KINDS & 15) == 4 ? true : (KINDS & 4) ? src_is_rgb(kind) : false; }
    static CHV_DEV bool is_planar(int kind) { return (KINDS & 3) == 2 ? !is_rgb(kind) : (KINDS & 2) ? src_is_planar(kind) : false; }
    using Cfg = WaveCfg<WTH>;
    static constexpr int WN_Y = Cfg::WN_Y, WN_C = Cfg::WN_C, WN_RGB = Cfg::WN_RGB, WNR = Cfg::WNR, ROWTAB_BYTES = Cfg::ROWTAB_BYTES;
    const DTick *T;
    const DLayer *L;
    int nl;
    int lane, x0, y0, x;
    bool col_in, row_in;
    float sx, sy, nx, ny;              // launch domain; NDC coordinates of this lane's column / of row `lane`
    int xe, ye;                        // the same pixel indices, clamped into the canvas
    uint8_t *smem;                     // this wave's LDS region: row table, plane-0 rectangle, chroma / U, (planar) V
    uint4 *rowtab;
    int base0, base1, voff, p0pitch, p0rows, p1pitch, p1rows;
    int ylim, clim, plim;              // bytes a YUV layer's luma / NV12 chroma / planar chroma-plane rectangle row may take (the pitches, or the columns of the side-by-side layout)

    // XCD-aware numbering: block b runs on XCD b % 8; every XCD gets one contiguous range of the launch's strips (whole
    // frames when there are >= 8 ticks), so halo rows are shared through that XCD's L2.  false: nothing to do for this wave.
    // n / d and n % d for wave-uniform operands from M = floor(2^32 / d), which the host passes in (launch_wave_layers):
    // n M / 2^32 = n / d - n e / (d 2^32) with e = 2^32 mod d < d and n < 2^32, i.e. less than 1 below n / d — the estimate is
    // the quotient or one less, one correction step.  All on the scalar unit; the compiler's expansion of a 32-bit division
    // is ~30 instructions through v_rcp_iflag_f32 and v_readfirstlane, twice per strip.
    static CHV_DEV void udivmod(uint32_t n, uint32_t d, uint32_t M, int &q, int &r) {
        uint32_t qq = __umulhi(n, M), rr = n - qq * d;
        if (rr >= d) { qq++; rr -= d; }
        q = (int)qq; r = (int)rr;
    }
    // the wave's LDS region and the launch's rectangle layout (everything that does not depend on which strip this is)
    CHV_DEV void init_layout(uint8_t *smem_all, int p0pitch_, int p0rows_, int p1pitch_, int p1rows_, int planar_any) {
        const int tid = threadIdx.x;
        lane = tid & 63;
        const int wave = WAVES == 1 ? 0 : __builtin_amdgcn_readfirstlane(tid >> 6);      // (one wave per block: no per-wave LDS offset arithmetic)
        p0pitch = p0pitch_; p0rows = p0rows_ & 0xFFFF; p1pitch = p1pitch_; p1rows = p1rows_ & 0xFFFF;
        // planar_any: bit 0 — the launch has planar pictures; bit 7: the SIDE-BY-SIDE layout (launch_wave_layers) — one region whose rows hold a YUV
        // layer's luma columns (bits 8-19, units of 16 bytes), then its chroma: the (u, v) rows of an NV12 picture, or the U and V rows of a planar
        // one, each bits 20-31 wide — instead of a plane-0 region followed by chroma regions; RGB rectangles use the whole rows
        const int ycols = ((planar_any >> 8) & 0xFFF) * 16, pcols = ((planar_any >> 20) & 0xFFF) * 16;
        int wbytes;
        base0 = ROWTAB_BYTES;
        if (planar_any & 128) {
            p1pitch = p0pitch; voff = pcols; base1 = base0 + ycols; ylim = ycols; clim = p0pitch - ycols; plim = pcols;
            wbytes = ROWTAB_BYTES + p0pitch * max(p0rows, p1rows);
        } else {
            voff = p1rows * p1pitch; base1 = base0 + p0rows * p0pitch; ylim = p0pitch; clim = p1pitch; plim = p1pitch;
            wbytes = ROWTAB_BYTES + p0rows * p0pitch + voff * ((planar_any & 1) ? 2 : 1);
        }
        smem = smem_all + wave * wbytes;
        rowtab = (uint4 *)smem;
    }
    // this wave's strip: canvas of W x H pixels, strip origin (x0_, y0_)
    CHV_DEV void init_strip(int W, int H, int x0_, int y0_) {
        x0 = x0_; y0 = y0_;
        sx = (float)W; sy = (float)H;
        x = x0 + lane;
        col_in = x < W;
        row_in = lane < WTH && y0 + lane < H;
        // NDC coordinates (layer-independent: gid / size * 2 - 1, kernels.cl.swift:70-72)
        xe = min(x, W - 1); ye = min(y0 + min(lane, WTH - 1), H - 1);
        nx = ((float)xe / sx) * 2.f - 1.f; ny = ((float)ye / sy) * 2.f - 1.f;
    }
    // ONE (tick_bgra_wave_one): the launch is one tick whose layers follow it in the kernarg segment — tick 0, first_layer 0: `T` and `L` are the
    // kernarg base plus constants (no scalar registers of their own)
    template <bool ONE = false>
    CHV_DEV bool init(const DTick *ticks, const DLayer *layers, int n_ticks, int strips_x, int strips_y, uint32_t strips_magic, uint32_t strips_x_magic,
                      uint8_t *smem_all, int p0pitch_, int p0rows_, int p1pitch_, int p1rows_, int planar_any) {
        // (bits 16-31 of the two row counts: the launch's ORIGIN in strips — a launch that continues on composed canvases covers only the strips
        // its layers' bounding boxes touch, launch_wave_layers)
        const int osx = (int)((uint32_t)p0rows_ >> 16), osy = (int)((uint32_t)p1rows_ >> 16);
        init_layout(smem_all, p0pitch_, p0rows_, p1pitch_, p1rows_, planar_any);
        const int wave = WAVES == 1 ? 0 : __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
        const int strips = strips_x * strips_y;
        const int total = strips * n_ticks;
        const int bid = blockIdx.x;
        const int xcd = bid & 7, slot = bid >> 3;
        const int per_xcd = (total + 7) >> 3;                           // strips per XCD
        const int widx = slot * WAVES + wave;                           // this wave's strip within the XCD's range
        const int index = xcd * per_xcd + widx;
        if (widx >= per_xcd || index >= total) return false;
        int tick, strip, sxi, syi;
        if constexpr (ONE) { tick = 0; strip = index; }
        else udivmod((uint32_t)index, (uint32_t)strips, strips_magic, tick, strip);
        udivmod((uint32_t)strip, (uint32_t)strips_x, strips_x_magic, syi, sxi);
        T = ticks + tick;
        const int x0_ = (osx + sxi) * WTW, y0_ = (osy + syi) * WTH;
        if (x0_ >= T->W || y0_ >= T->H) return false;
        L = ONE ? layers : layers + T->first_layer;
        nl = T->n_layers;
        init_strip(T->W, T->H, x0_, y0_);
        return true;
    }

    // The geometry of layer l from its class's table instead of from its matrices (setup() below).  Same values, bit for bit: the table was
    // filled by setup() itself (geom_precompute), for every strip of the canvas — staged rectangles or not.
    CHV_DEV void setup_cached(int l, WLayer &w) const {
        const DLayer &Ly = L[l];
        const uint64_t tp = ((uint64_t)(uint32_t)Ly.pad2[1] << 32) | (uint64_t)(uint32_t)Ly.pad2[0];          // (uniform: the descriptor's scalar loads)
        // header, flag word and the scalar halves of both records: scalar loads through the constant address space (cld); the lanes' column
        // entries and the row table: coalesced vector loads on a uniform base
        const int hsx = cld<int32_t>(tp + 0), row_bytes = cld<int32_t>(tp + 12);
        const uint32_t flags_off = cld<uint32_t>(tp + 16), cols_off = cld<uint32_t>(tp + 20), rows_off = cld<uint32_t>(tp + 24);
        const int sxa = x0 >> 6, sya = WTH == 16 ? y0 >> 4 : y0 >> 3;           // (WTW = 64)
        const uint32_t f = cld<uint32_t>(tp + flags_off + 4u * (uint32_t)(sya * hsx + sxa));
        const bool st = (f & GF_STAGED) != 0;
        const uint64_t C = tp + cols_off + (uint64_t)sxa * sizeof(GeomCol), R = tp + rows_off + (uint64_t)sya * (uint64_t)row_bytes;
        const uint8_t *Cp = (const uint8_t *)C, *Rp = (const uint8_t *)R;
        const uint4 ca = gld_at<uint4>(Cp, (uint32_t)lane * 16u), cb = gld_at<uint4>(Cp, 1024u + (uint32_t)lane * 16u);
        w.cyo = st ? (int)ca.x : (int)cb.x; w.cco = st ? (int)ca.y : (int)cb.y;
        w.cya = __uint_as_float(ca.z); w.cca = __uint_as_float(ca.w); w.cfl = (int)cb.z;
        if (lane < 3 * WTH) rowtab[lane] = gld_at<uint4>(Rp, (st ? 0u : (uint32_t)(3 * WTH * 16)) + (uint32_t)lane * 16u);
        const uint64_t rs = R + 2 * 3 * WTH * 16, cs = C + 2048;
        const bool e0 = (f & GF_EDGE0) != 0, e1 = (f & GF_EDGE1) != 0;
        w.g0.r_lo = cld<int32_t>(rs + 0); w.g0.rows = cld<int32_t>(rs + 4); w.g0.pair = cld<int32_t>(rs + 8); w.g0.r_hi1 = cld<int32_t>(rs + 12);
        w.g1.r_lo = cld<int32_t>(rs + 16); w.g1.rows = cld<int32_t>(rs + 20); w.g1.pair = cld<int32_t>(rs + 24); w.g1.r_hi1 = cld<int32_t>(rs + 28);
        const int nv0 = cld<int32_t>(cs + 4), nv1 = cld<int32_t>(cs + 20);
        w.g0.b0 = cld<int32_t>(cs + 0); w.g0.nvec = nv0; w.g0.edge = e0 ? 1 : 0; w.g0.nslot = e0 ? nv0 + 2 : nv0; w.g0.inv20 = e0 ? cld<int32_t>(cs + 12) : cld<int32_t>(cs + 8);
        w.g1.b0 = cld<int32_t>(cs + 16); w.g1.nvec = nv1; w.g1.edge = e1 ? 1 : 0; w.g1.nslot = e1 ? nv1 + 2 : nv1; w.g1.inv20 = e1 ? cld<int32_t>(cs + 28) : cld<int32_t>(cs + 24);
        w.staged = st;
        w.all_inside = (f & GF_ALL_INSIDE) != 0;
        w.unit_rows = st && cld<int32_t>(rs + 32) != 0;
    }
    // (geom_precompute) what setup<true>() left in `w`, `raw` and the LDS row table, into the class's table.  Every strip writes the unstaged form
    // of its column and of its row and its flag word; a strip whose rectangles are staged the staged form as well — by every strip of the
    // column / of the row with the same bytes.
    CHV_DEV void geom_store(uint8_t *tab, int sxa, int sya, const WLayer &w, const GeomRaw &raw) const {
        const uint64_t tp = (uint64_t)(uintptr_t)tab;
        const int hsx = cld<int32_t>(tp + 0), row_bytes = cld<int32_t>(tp + 12);
        const uint32_t flags_off = cld<uint32_t>(tp + 16), cols_off = cld<uint32_t>(tp + 20), rows_off = cld<uint32_t>(tp + 24);
        uint8_t *Cp = tab + cols_off + (size_t)sxa * sizeof(GeomCol), *Rp = tab + rows_off + (size_t)sya * (size_t)row_bytes;
        gst_at<uint4>(Cp, 1024u + (uint32_t)lane * 16u, make_uint4((uint32_t)raw.cy, (uint32_t)raw.cc, (uint32_t)w.cfl, 0u));
        if (lane < WTH) {
            // the unstaged row entries (setup(): yoff = ry, coff = rc)
            uint8_t *Rr = Rp + 3 * WTH * 16;
            gst_at<uint4>(Rr, (uint32_t)(2 * lane) * 16u, make_uint4((uint32_t)raw.ry, (uint32_t)raw.rc, (uint32_t)raw.rfl, 0u));
            gst_at<uint4>(Rr, (uint32_t)(2 * lane + 1) * 16u, make_uint4(__float_as_uint(raw.rya), __float_as_uint(1.0f - raw.rya), __float_as_uint(raw.rca), __float_as_uint(1.0f - raw.rca)));
            gst_at<uint4>(Rr, (uint32_t)(2 * WTH + lane) * 16u,
                          make_uint4(__float_as_uint(raw.rya), __float_as_uint(raw.rca), ((uint32_t)raw.ry & 0xFFFFu) | ((uint32_t)raw.rc << 16), (uint32_t)raw.rfl));
        }
        if (w.staged) {
            gst_at<uint4>(Cp, (uint32_t)lane * 16u, make_uint4((uint32_t)w.cyo, (uint32_t)w.cco, __float_as_uint(w.cya), __float_as_uint(w.cca)));
            if (lane < 3 * WTH) gst_at<uint4>(Rp, (uint32_t)lane * 16u, rowtab[lane]);
        } else {
            // (a column no strip of which is staged still needs its weights: they are the same in both forms)
            gst_at<uint2>(Cp, (uint32_t)lane * 16u + 8u, make_uint2(__float_as_uint(w.cya), __float_as_uint(w.cca)));
        }
        if (lane == 0) {
            if (w.staged) {
                auto inv = [](int nslot) { return nslot > 0 ? ((1 << 20) + nslot - 1) / nslot : 0; };      // stage_slots_init's
                const int32_t rsv[9] = { w.g0.r_lo, w.g0.rows, w.g0.pair, w.g0.r_hi1, w.g1.r_lo, w.g1.rows, w.g1.pair, w.g1.r_hi1, w.unit_rows ? 1 : 0 };
                const int32_t csv[8] = { w.g0.b0, w.g0.nvec, inv(w.g0.nvec), inv(w.g0.nvec + 2), w.g1.b0, w.g1.nvec, inv(w.g1.nvec), inv(w.g1.nvec + 2) };
                for (int k = 0; k < 9; k++) gst_at<int32_t>(Rp, (uint32_t)(2 * 3 * WTH * 16 + 4 * k), rsv[k]);
                for (int k = 0; k < 8; k++) gst_at<int32_t>(Cp, (uint32_t)(2048 + 4 * k), csv[k]);
            }
            gst_at<uint32_t>(tab + flags_off, 4u * (uint32_t)(sya * hsx + sxa),
                             (w.staged ? GF_STAGED : 0u) | (w.all_inside ? GF_ALL_INSIDE : 0u) | ((w.staged && w.g0.edge) ? GF_EDGE0 : 0u) | ((w.staged && w.g1.edge) ? GF_EDGE1 : 0u));
        }
    }

    // layers whose border quad cannot touch the strip are skipped (uniform test against the host-computed bounding box)
    CHV_DEV int next_hit(int l) const {
        for (; l < nl; l++) {
            const int *bb = L[l].bbox;
            if (!(x0 + WTW <= bb[0] || x0 >= bb[2] || y0 + WTH <= bb[1] || y0 >= bb[3])) break;
        }
        return l;
    }

    // per-layer geometry: column entry, row entries (into the LDS table), summaries, staging rectangles
    template <bool RAW = false>
    CHV_DEV void setup(int l, WLayer &w, GeomRaw *raw = nullptr) const {
        constexpr unsigned long long ROWMASK = WTH >= 64 ? ~0ull : ((1ull << WTH) - 1ull);
        const DLayer &Ly = L[l];
        const bool rgb = is_rgb(Ly.kind);
        const DPlane &S0 = Ly.src.pl[0];
        const DPlane &S1 = Ly.src.pl[rgb ? 0 : 1];
        int fl, rfl, cy, cc, ry, rc;
        float rya, rca;
        {
            // every layer on this path is flagged LF_AXIS_ALIGNED | LF_BOUNDED (host-checked, wave_layers_eligible): the entries the
            // axis-alignment flag guarantees to be zero contribute exact zeros, so the short form gives the bits of the full dot
            // products (geometry_axis, pixel_math.hip.h)
            const float *U = Ly.u;
            const float t3 = U[U_TRANSFORM + 15];
            const float t0 = nx * U[U_TRANSFORM + 0] + U[U_TRANSFORM + 3], t1 = ny * U[U_TRANSFORM + 5] + U[U_TRANSFORM + 7];
            const float b0 = nx * U[U_BORDER + 0] + U[U_BORDER + 3], b1 = ny * U[U_BORDER + 5] + U[U_BORDER + 7];
            const float u = t0 * U[U_TEXTURE + 0] + t3 * U[U_TEXTURE + 3], v = t1 * U[U_TEXTURE + 5] + t3 * U[U_TEXTURE + 7];
            fl = ((b0 >= 0.f && b0 <= 1.f) ? AX_BORDER : 0) | ((t0 >= 0.f && t0 <= 1.f) ? AX_TX : 0) | ((u >= 0.f && u <= 1.f) ? AX_UV : 0);
            rfl = ((b1 >= 0.f && b1 <= 1.f) ? AX_BORDER : 0) | ((t1 >= 0.f && t1 <= 1.f) ? AX_TX : 0) | ((v >= 0.f && v <= 1.f) ? AX_UV : 0);
            lin_axis_raw(u, S0.w, cy, w.cya); lin_axis_raw(u, S1.w, cc, w.cca);
            lin_axis_raw(v, S0.h, ry, rya); lin_axis_raw(v, S1.h, rc, rca);
        }
        const AxisSum cs = axis_summary(~0ull, col_in, fl, cy, cc);
        w.cfl = col_in ? fl : AX_ALL;                               // past the canvas edge: never stored; copy of the last column
        const AxisSum rs = axis_summary(ROWMASK, row_in, rfl, ry, rc);
        if (!row_in) rfl = AX_ALL;
        if constexpr (RAW) { raw->cy = cy; raw->cc = cc; raw->ry = ry; raw->rc = rc; raw->rfl = rfl; raw->rya = rya; raw->rca = rca; }
        w.all_inside = cs.any && cs.all && rs.any && rs.all;
        w.staged = false;
        int cyo = cy, cco = cc, yoff = ry, coff = rc;              // unstaged: the positions themselves
        if (cs.any && rs.any) {
            bool ok;
            int c0off, c1off = 0, r1off = 0;
            {
                const int sh = rgb ? 2 : 4;                          // log2(texels per 16-byte vector)
                const int tpv = 1 << sh;
                const int col0 = max(cs.lo, 0) & ~(tpv - 1);
                const int nvec = ((min(cs.hi, S0.w - 1) - col0) >> sh) + 1;
                w.g0.r_lo = rs.lo; w.g0.rows = rs.hi - rs.lo + 1; w.g0.b0 = col0 << (4 - sh); w.g0.nvec = nvec;
                w.g0.pair = CHV_WAVE_PAIR && WTH == 8 && w.g0.rows > 2 * WTH;
                w.g0.r_hi1 = max(rs.hi - 1, rs.lo);
                if (w.g0.pair) w.g0.rows = 2 * WTH;
                w.g0.edge = cs.lo < 0 || cs.hi >= S0.w || rs.lo < 0 || rs.hi >= S0.h - 1 + (int)(col0 + nvec * tpv <= S0.w);
                stage_slots_init(w.g0);
                ok = (nvec + 2) * 16 <= (rgb ? p0pitch : ylim) && w.g0.rows <= p0rows && stage_slots(w.g0) <= 1024;
                // (positions clamped into the range of the fully-inside entries: lanes outside the picture then read staged
                // bytes, whatever they are, instead of addresses outside the rectangle; inside lanes are unaffected)
                c0off = base0 + 16 + ((min(max(cy, cs.lo), cs.hi - 1) - col0) << (4 - sh));     // byte of tap 0 in LDS row 0
            }
            if (ok && !rgb) {
                const int sh = is_planar(Ly.kind) ? 4 : 3;
                const int tpv = 1 << sh;
                const int col0 = max(cs.clo, 0) & ~(tpv - 1);
                const int nvec = ((min(cs.chi, S1.w - 1) - col0) >> sh) + 1;
                w.g1.r_lo = rs.clo; w.g1.rows = rs.chi - rs.clo + 1; w.g1.b0 = col0 << (4 - sh); w.g1.nvec = nvec;
                w.g1.pair = CHV_WAVE_PAIR && WTH == 8 && w.g1.rows > 2 * WTH;
                w.g1.r_hi1 = max(rs.chi - 1, rs.clo);
                if (w.g1.pair) w.g1.rows = 2 * WTH;
                w.g1.edge = cs.clo < 0 || cs.chi >= S1.w || rs.clo < 0 || rs.chi >= S1.h - 1 + (int)(col0 + nvec * tpv <= S1.w);
                stage_slots_init(w.g1);
                ok = (nvec + 2) * 16 <= (is_planar(Ly.kind) ? plim : clim) && w.g1.rows <= p1rows && stage_slots(w.g1) <= 1024;
                c1off = base1 + 16 + ((min(max(cc, cs.clo), cs.chi - 1) - col0) << (4 - sh));
                r1off = w.g1.pair ? 2 * min(lane, WTH - 1) * p1pitch : (min(max(rc, rs.clo), rs.chi - 1) - rs.clo) * p1pitch;
            }
            if (ok) {
                w.staged = true;
                cyo = c0off; cco = c1off;
                yoff = w.g0.pair ? 2 * min(lane, WTH - 1) * p0pitch : (min(max(ry, rs.lo), rs.hi - 1) - rs.lo) * p0pitch; coff = r1off;
            }
        }
        w.cyo = cyo; w.cco = cco;
        {
            // lane j + 1's row offset (lanes 0 .. WTH - 1 sit in one DPP row of 16)
            const int ynext = __builtin_amdgcn_update_dpp(yoff, yoff, 0x101 /* row_shl:1 */, 0xf, 0xf, false);
            w.unit_rows = w.staged && __ballot(lane < WTH - 1 && ynext - yoff != p0pitch) == 0;
        }
        if (lane < WTH) {
            rowtab[2 * lane] = make_uint4((uint32_t)yoff, (uint32_t)coff, (uint32_t)rfl, 0u);
            rowtab[2 * lane + 1] = make_uint4(__float_as_uint(rya), __float_as_uint(1.0f - rya), __float_as_uint(rca), __float_as_uint(1.0f - rca));
            rowtab[2 * WTH + lane] = make_uint4(__float_as_uint(rya), __float_as_uint(rca), ((uint32_t)yoff & 0xFFFFu) | ((uint32_t)coff << 16), (uint32_t)rfl);
        }
    }

    // issue the global loads of layer l's rectangles, wait once for all of them, write them to this wave's LDS region
    template <bool EDGE, bool PAIR = false>
    CHV_DEV void stage_impl(int l, const WLayer &w) const {
        const DLayer &Ly = L[l];
        // pair form: lane j's own first tap rows (the row geometry of setup(), clamped into the rows the rectangles were validated for)
        int pry0 = 0, pry1 = 0;
        if constexpr (PAIR) {
            const float *U = Ly.u;
            const float t1 = ny * U[U_TRANSFORM + 5] + U[U_TRANSFORM + 7];
            const float v = t1 * U[U_TEXTURE + 5] + U[U_TRANSFORM + 15] * U[U_TEXTURE + 7];
            float a_;
            lin_axis_raw(v, Ly.src.pl[0].h, pry0, a_);
            lin_axis_raw(v, Ly.src.pl[is_rgb(Ly.kind) ? 0 : 1].h, pry1, a_);
            pry0 = min(max(pry0, w.g0.r_lo), w.g0.r_hi1);
            pry1 = min(max(pry1, w.g1.r_lo), w.g1.r_hi1);
        }
        uint4 regs[WNR];
        if (is_rgb(Ly.kind)) {
            wstage_load<0, WN_RGB, WNR, EDGE, PAIR>(regs, Ly.src.pl[0], w.g0, lane, pry0);
            touch_regs(regs);             // one wait for all of the layer's loads (see touch_regs, pixel_math.hip.h)
            // staged texels are byte-swapped where the layer asks for it (RGBA source on a BGRA canvas: -> BGRA; BGRA source on a
            // 4:2:0 canvas, kernels.cl.swift:518 `.zyxw`: -> RGBA), so the tap loops need no channel select
            wstage_store<4, 0, WN_RGB, WNR, EDGE, PAIR>(regs, smem + base0, p0pitch, Ly.src.pl[0], w.g0, lane, Ly.swizzle != 0, pry0);
            wstage_tail<4, WN_RGB, EDGE, PAIR>(smem + base0, p0pitch, Ly.src.pl[0], w.g0, lane, Ly.swizzle != 0, pry0);
        } else if (!is_planar(Ly.kind)) {
            // (two planes: the interleaved chroma rectangle takes the registers a planar picture's V plane would — in pair form it has as many
            // slots as the luma rectangle — and what is left of BOTH rectangles finishes in one loop, one wait per round of it: a grid
            // quadrant's 2 x 208 slots are two waits instead of four)
            constexpr int WN_C2 = CHV_WAVE_NV12_WIDE ? 2 * WN_C : WN_C;
            wstage_load<0, WN_Y, WNR, EDGE, PAIR>(regs, Ly.src.pl[0], w.g0, lane, pry0);
            wstage_load<WN_Y, WN_C2, WNR, EDGE, PAIR>(regs, Ly.src.pl[1], w.g1, lane, pry1);
            touch_regs(regs);
            wstage_store<1, 0, WN_Y, WNR, EDGE, PAIR>(regs, smem + base0, p0pitch, Ly.src.pl[0], w.g0, lane, false, pry0);
            wstage_store<2, WN_Y, WN_C2, WNR, EDGE, PAIR>(regs, smem + base1, p1pitch, Ly.src.pl[1], w.g1, lane, false, pry1);
            if constexpr (CHV_WAVE_NV12_WIDE) {
                wstage_tail2<1, 2, WN_Y, WN_C2, EDGE, PAIR>(smem + base0, p0pitch, Ly.src.pl[0], w.g0, pry0, smem + base1, p1pitch, Ly.src.pl[1], w.g1, pry1, lane);
            } else {
                wstage_tail<1, WN_Y, EDGE, PAIR>(smem + base0, p0pitch, Ly.src.pl[0], w.g0, lane, false, pry0);
                wstage_tail<2, WN_C, EDGE, PAIR>(smem + base1, p1pitch, Ly.src.pl[1], w.g1, lane, false, pry1);
            }
        } else {
            wstage_load<0, WN_Y, WNR, EDGE, PAIR>(regs, Ly.src.pl[0], w.g0, lane, pry0);
            wstage_load<WN_Y, WN_C, WNR, EDGE, PAIR>(regs, Ly.src.pl[1], w.g1, lane, pry1);
            wstage_load<WN_Y + WN_C, WN_C, WNR, EDGE, PAIR>(regs, Ly.src.pl[2], w.g1, lane, pry1);
            touch_regs(regs);
            wstage_store<1, 0, WN_Y, WNR, EDGE, PAIR>(regs, smem + base0, p0pitch, Ly.src.pl[0], w.g0, lane, false, pry0);
            wstage_store<1, WN_Y, WN_C, WNR, EDGE, PAIR>(regs, smem + base1, p1pitch, Ly.src.pl[1], w.g1, lane, false, pry1);
            wstage_store<1, WN_Y + WN_C, WN_C, WNR, EDGE, PAIR>(regs, smem + base1 + voff, p1pitch, Ly.src.pl[2], w.g1, lane, false, pry1);
            wstage_tail<1, WN_Y, EDGE, PAIR>(smem + base0, p0pitch, Ly.src.pl[0], w.g0, lane, false, pry0);
            wstage_tail<1, WN_C, EDGE, PAIR>(smem + base1, p1pitch, Ly.src.pl[1], w.g1, lane, false, pry1);
            wstage_tail<1, WN_C, EDGE, PAIR>(smem + base1 + voff, p1pitch, Ly.src.pl[2], w.g1, lane, false, pry1);
        }
    }
    // YUV pictures, both rectangles interior and narrow: shift-and-mask slot map (wstage_load_p2)
    CHV_DEV bool stage_p2(int l, const WLayer &w) const {
        const DLayer &Ly = L[l];
        const P2Map m0 = p2_map<WN_Y>(w.g0, lane), m1 = p2_map<WN_C>(w.g1, lane);
        if (!(m0.ok && m1.ok)) return false;
        uint4 regs[WNR];
        const bool planar = is_planar(Ly.kind);
        wstage_load_p2<0, WN_Y, WNR, (INTERIOR & 8) != 0>(regs, Ly.src.pl[0], w.g0, m0);
        wstage_load_p2<WN_Y, WN_C, WNR, (INTERIOR & 8) != 0>(regs, Ly.src.pl[1], w.g1, m1);
        if (planar) wstage_load_p2<WN_Y + WN_C, WN_C, WNR, (INTERIOR & 8) != 0>(regs, Ly.src.pl[2], w.g1, m1);
        touch_regs(regs);                 // one wait for all of the layer's loads
        wstage_store_p2<0, WN_Y, WNR, (INTERIOR & 8) != 0>(regs, smem + base0, p0pitch, w.g0, m0);
        wstage_store_p2<WN_Y, WN_C, WNR, (INTERIOR & 8) != 0>(regs, smem + base1, p1pitch, w.g1, m1);
        if (planar) wstage_store_p2<WN_Y + WN_C, WN_C, WNR, (INTERIOR & 8) != 0>(regs, smem + base1 + voff, p1pitch, w.g1, m1);
        return true;
    }
    // (rectangles that touch no picture edge — most strips — take the instantiation without clamping and patching code)
    CHV_DEV void stage(int l, const WLayer &w) const {
        const bool rgb = is_rgb(L[l].kind);
        if constexpr (CHV_WAVE_PAIR && WTH == 8) {
            // (8-row strips only: the 16-row instantiations are at their register limit, and launches whose rectangles are this tall run 8-row
            // strips anyway (launch_wave_layers).  Its own instantiation: the row lookup costs the other paths registers)
            if (w.g0.pair || (!rgb && w.g1.pair)) {
                if (INTERIOR != 0 && !w.g0.edge && (rgb || !w.g1.edge)) stage_impl<false, true>(l, w);        // (no clamping, no patching)
                else stage_impl<true, true>(l, w);
                return;
            }
        }
        if constexpr ((INTERIOR & 4) != 0) {
            if (!rgb && !w.g0.edge && !w.g1.edge && stage_p2(l, w)) return;
        }
        if constexpr (INTERIOR != 0) {
            if ((INTERIOR & (rgb ? 2 : 1)) && !w.g0.edge && (rgb || !w.g1.edge)) { stage_impl<false>(l, w); return; }
        }
        stage_impl<true>(l, w);
    }
};

}  // namespace chv

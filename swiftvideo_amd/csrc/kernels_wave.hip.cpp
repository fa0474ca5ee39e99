// kernels_wave.hip.cpp — axis-aligned tick kernel for BGRA canvases, one WAVE per canvas strip, no block barriers.
//
// Layers: any mix of NV12 / y420p / BGRA / RGBA pictures, any number of them per tick — the tick VideoMixer.mix issues
// when decoded YUV streams and RGB overlays land on one BGRA canvas (mix.video.swift:114-124; findKernel :142-146 ->
// img_nv12_bgra / img_y420p_bgra / img_{bgra,rgba}_bgra_tx).
//
// Why a wave and not a block (profiles/r02_notes.md): with a 64x32 tile per 256-thread block the layer loop passes two
// block barriers per layer with 8 pixels of work per thread in between, the per-column table entries are re-read from
// LDS for every pixel, and the kernel sits at 129 VALU instructions per pixel-layer with waves waiting 62 % of the time.
// Here
//   * lane = canvas column: a wave owns a strip of 64 columns x WTH rows and keeps its WTH pixels per lane in registers
//     (packed BGRA codes) across all layers — the canvas is written once, every layer's source bytes leave HBM once
//     (strips of one frame run on one XCD, so halo rows are shared through that XCD's L2);
//   * the column half of the reference's coordinate arithmetic is evaluated once per lane and layer and stays in
//     registers for the lane's WTH pixels; the row half is evaluated by lanes 0..WTH-1, parked in a small wave-private
//     LDS table and fetched per row with two broadcast ds_read_b128 (v_readlane would cost six VALU issue slots per
//     row, and the VALU is what bounds this kernel);
//   * the source rectangle of the strip is staged into a WAVE-PRIVATE LDS region (bytes: luma + (u,v) pairs / U + V /
//     4-byte texels, edge texels replicated), so there is no block barrier anywhere: a wave loads its rectangle, writes
//     it, reads its taps; the waves of a SIMD drift apart and overlap each other's memory and arithmetic phases
//     (issuing a layer's loads before the previous layer's pixels — two sets of entries live — measured no faster);
//   * taps are read with naturally aligned ds_read_u8 / ds_read_u16 and widened with v_cvt_f32_ubyteN (unaligned
//     ds_read_u16 / ds_read_b32 covering both taps of a row compile and give the right bytes on gfx950, but execute one
//     lane at a time: SQ_LDS_IDX_ACTIVE went from 122 M to 962 M cycles per launch, profiles/r02_notes.md).
// Same instruction sequence as the general kernel for the coordinates and the same operations on the pixels, so the
// bytes are those of kernels_general.hip.cpp (= oracle/ref_kernels.c::px_to_bgra layer by layer).
#include "tile_common.hip.h"

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <type_traits>

#pragma clang fp contract(off)

namespace chv {

#ifndef CHV_WAVE_ROWS
#define CHV_WAVE_ROWS 8
#endif
constexpr int WTW = 64;                 // strip width: one lane per column
constexpr int WTH = CHV_WAVE_ROWS;      // strip height: rows per lane
// staging registers (16-byte vectors) per lane, 64 slots each, sized for the rectangles of a 1.5x downscale (YUV) / of a
// native-resolution picture (RGB); larger rectangles finish through the on-the-spot tail of wstage_store:
//   RGB layer: [0 .. WN_RGB) plane 0;   YUV layer: [0 .. WN_Y) luma, then WN_C for chroma / U, then WN_C for V (planar)
constexpr int WN_Y = WTH / 4, WN_C = WTH / 8, WN_RGB = WTH / 4 + 1;
constexpr int WNR = (WN_RGB > WN_Y + 2 * WN_C) ? WN_RGB : (WN_Y + 2 * WN_C);
static_assert(WTH == 8 || WTH == 16, "strip height: 8 or 16 rows");
constexpr int WAVES = NTHREADS / 64;
constexpr int ROWTAB_BYTES = WTH * 32;  // per wave: the current layer's row entries, 8 dwords per row

// ---- wave-level helpers ---------------------------------------------------------------------------------------
CHV_DEV void wave_lds_fence() {
    // the wave's own LDS writes are visible to its later reads (LDS operations of a wave execute in order); this only
    // keeps the compiler from moving accesses across the point
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
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
struct WLayer {
    int cyo, cco;            // staged layers: LDS byte offset of tap 0 inside a staged row (luma / RGB texel, chroma);
                             // unstaged layers: the unclamped tap-0 texel positions themselves
    float cya, cca;          // weight of tap 1
    int cfl;
    StageGeom g0, g1;
    bool staged, all_inside;
};
// row entry in the wave's LDS table: two 16-byte halves
//   A = {yoff, coff, rfl, -}: LDS byte offset of the row's first tap row in the plane-0 / chroma rectangle (unstaged: the
//                              unclamped tap-0 row positions), flags
//   B = {yb, 1 - yb, cb, 1 - cb}: weights of tap row 1 (luma / RGB, chroma) and their complements

template <int OFF, int N>
CHV_DEV void wstage_load(uint4 (&regs)[WNR], const DPlane &P, const StageGeom &g, int lane) {
    // exactly one global_load_dwordx4 per slot, straight into its final register (see stage_load, tile_common.hip.h)
#pragma unroll
    for (int n = 0; n < N; n++) {
        int i = lane + n * 64, r, vv;
        stage_slot(g, i, r, vv);
        if (r < g.rows) {
            int row = min(max(g.r_lo + r, 0), P.h - 1);
            int off = g.b0 + (g.edge ? vv - 1 : vv) * 16;
            if (g.edge) off = vec_loadable(P, row, off) ? off : 0;
            regs[OFF + n] = gld<uint4>(P.ptr + (size_t)row * P.pitch + off);
        }
    }
}
// one slot: CLAMP_TO_EDGE patching (edge rectangles only), optional RGBA -> BGRA, LDS write
template <int BPT>
CHV_DEV void wstage_put(uint4 val, int i, uint8_t *lds, int lds_pitch, const DPlane &P, const StageGeom &g, bool swap02) {
    int r, vv;
    stage_slot(g, i, r, vv);
    if (i < 1024 && r < g.rows) {
        int v = g.edge ? vv - 1 : vv;
        if (g.edge) {
            int row = min(max(g.r_lo + r, 0), P.h - 1);
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
template <int BPT, int OFF, int N>
CHV_DEV void wstage_store(const uint4 (&regs)[WNR], uint8_t *lds, int lds_pitch, const DPlane &P, const StageGeom &g, int lane, bool swap02) {
#pragma unroll
    for (int n = 0; n < N; n++) wstage_put<BPT>(regs[OFF + n], lane + n * 64, lds, lds_pitch, P, g, swap02);
}
// Slots beyond the registers' share of a plane (stronger downscales, rectangles at a picture edge): further rounds of
// WTAIL loads in flight, one wait, WTAIL LDS writes (not unrolled beyond that: the edge patching is large code).
constexpr int WTAIL = 2;
template <int BPT, int N>
CHV_DEV void wstage_tail(uint8_t *lds, int lds_pitch, const DPlane &P, const StageGeom &g, int lane, bool swap02) {
#pragma unroll 1
    for (int base = N * 64; base < stage_slots(g); base += WTAIL * 64) {
        uint4 t[WTAIL];
#pragma unroll
        for (int n = 0; n < WTAIL; n++) {
            int i = base + n * 64 + lane, r, vv;
            stage_slot(g, i, r, vv);
            t[n] = make_uint4(0, 0, 0, 0);
            if (i < 1024 && r < g.rows) {
                int row = min(max(g.r_lo + r, 0), P.h - 1);
                int off = g.b0 + (g.edge ? vv - 1 : vv) * 16;
                if (g.edge) off = vec_loadable(P, row, off) ? off : 0;
                t[n] = gld<uint4>(P.ptr + (size_t)row * P.pitch + off);
            }
        }
#pragma unroll 1
        for (int n = 0; n < WTAIL; n++) wstage_put<BPT>(n == 0 ? t[0] : t[WTAIL - 1], base + n * 64 + lane, lds, lds_pitch, P, g, swap02);
    }
}

CHV_DEV float ub0(uint32_t w) { return (float)(w & 255u); }
CHV_DEV float ub1(uint32_t w) { return (float)((w >> 8) & 255u); }
CHV_DEV float ub2(uint32_t w) { return (float)((w >> 16) & 255u); }
CHV_DEV float ub3(uint32_t w) { return (float)(w >> 24); }

// Integer colour matrix on biased codes, channels returned as float codes (cf. yuv_to_bgra_word)
CHV_DEV void yuv_to_bgr_floats(const CscFolded &k, int y, int u, int v, float &fb, float &fg, float &fr) {
    int32_t t = __mul24(y, k.cy);
    int32_t r = mad24_uniform(v, k.crv, t) + k.kr;
    int32_t g = mad24_uniform(v, k.ncgv, mad24_uniform(u, k.ncgu, t)) + k.kg;
    int32_t b = mad24_uniform(u, k.cbu, t) + k.kb;
    uint32_t bg, ra;
    asm("v_ashr_pk_u8_i32 %0, %1, %2, 16" : "=v"(bg) : "v"(b), "v"(g));          // byte0 = B, byte1 = G
    asm("v_ashr_pk_u8_i32 %0, %1, %2, 16" : "=v"(ra) : "v"(r), "v"(0));          // byte0 = R
    fb = ub0(bg); fg = ub1(bg); fr = ub0(ra);
}

#ifndef CHV_WAVE_MINW
#define CHV_WAVE_MINW 5
#endif
// CHV_WAVE_FENCE = n > 0: keep the scheduler from interleaving more than n rows of a lane's pixels in the branch-free loops
// (fewer live temporaries)
#ifndef CHV_WAVE_FENCE
#define CHV_WAVE_FENCE 0
#endif
#define WAVE_ROW_FENCE(j) do { if (CHV_WAVE_FENCE > 0 && (j) > 0 && (j) % (CHV_WAVE_FENCE > 0 ? CHV_WAVE_FENCE : 1) == 0) __builtin_amdgcn_sched_barrier(0); } while (0)
template <bool CLEAR>
__global__ __launch_bounds__(NTHREADS, CHV_WAVE_MINW) void tick_bgra_wave(const DTick *__restrict__ ticks,
                                                                         const DLayer *__restrict__ layers,
                                                                         int n_ticks, int strips_x, int strips_y,
                                                                         int p0pitch, int p0rows, int p1pitch, int p1rows, int planar_any) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem_all[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // this wave's private LDS region: row table, [p0rows][p0pitch] plane 0, [p1rows][p1pitch] chroma / U, (planar) V
    const int voff = p1rows * p1pitch;
    const int wbytes = ROWTAB_BYTES + p0rows * p0pitch + voff * (planar_any ? 2 : 1);
    uint8_t *smem = smem_all + wave * wbytes;
    uint4 *rowtab = (uint4 *)smem;
    const int base0 = ROWTAB_BYTES, base1 = base0 + p0rows * p0pitch;

    // XCD-aware numbering: block b runs on XCD b % 8; every XCD gets one contiguous range of the launch's strips
    const int strips = strips_x * strips_y;
    const int total = strips * n_ticks;
    const int nblocks = (int)gridDim.x, per_xcd_blocks = nblocks >> 3;
    const int bid = blockIdx.x;
    const int xcd = bid & 7, slot = bid >> 3;
    const int per_xcd = (total + 7) >> 3;                           // strips per XCD
    const int widx = slot * WAVES + wave;                           // this wave's strip within the XCD's range
    (void)per_xcd_blocks;
    const int index = xcd * per_xcd + widx;
    if (widx >= per_xcd || index >= total) return;                  // (no block barrier anywhere: waves may leave)
    const int tick = index / strips;
    const int strip = index - tick * strips;
    const DTick &T = ticks[tick];
    const int x0 = (strip % strips_x) * WTW, y0 = (strip / strips_x) * WTH;
    if (x0 >= T.W || y0 >= T.H) return;
    const DLayer *L = layers + T.first_layer;
    const int nl = T.n_layers;
    const DPlane &D = T.dst.pl[0];
    const float sx = (float)T.W, sy = (float)T.H;
    const int x = x0 + lane;
    const bool col_in = x < T.W;
    constexpr unsigned long long ROWMASK = WTH >= 64 ? ~0ull : ((1ull << WTH) - 1ull);

    // ---- per-layer geometry: column entry, row entries, summaries, staging rectangles ----------------------------
    // NDC coordinates of this lane's column and of row `lane` (layer-independent: gid / size * 2 - 1, kernels.cl.swift:70-72)
    const int xe = min(x, T.W - 1), ye = min(y0 + min(lane, WTH - 1), T.H - 1);
    const float nx = ((float)xe / sx) * 2.f - 1.f, ny = ((float)ye / sy) * 2.f - 1.f;
    const bool row_in = lane < WTH && y0 + lane < T.H;
    auto layer_setup = [&](int l, WLayer &w) {
        const DLayer &Ly = L[l];
        const bool rgb = Ly.kind == LK_BGRA_FROM_RGB;
        const DPlane &S0 = Ly.src.pl[0];
        const DPlane &S1 = Ly.src.pl[rgb ? 0 : 1];
        int fl, rfl, cy, cc, ry, rc;
        float rya, rca;
        if ((Ly.flags & (LF_AXIS_ALIGNED | LF_BOUNDED)) == (LF_AXIS_ALIGNED | LF_BOUNDED)) {
            // bounded matrices: the entries the axis-alignment flag guarantees to be zero contribute exact zeros, so the short
            // form gives the bits of the full dot products (geometry_axis, pixel_math.hip.h)
            const float *U = Ly.u;
            const float t3 = U[U_TRANSFORM + 15];
            const float t0 = nx * U[U_TRANSFORM + 0] + U[U_TRANSFORM + 3], t1 = ny * U[U_TRANSFORM + 5] + U[U_TRANSFORM + 7];
            const float b0 = nx * U[U_BORDER + 0] + U[U_BORDER + 3], b1 = ny * U[U_BORDER + 5] + U[U_BORDER + 7];
            const float u = t0 * U[U_TEXTURE + 0] + t3 * U[U_TEXTURE + 3], v = t1 * U[U_TEXTURE + 5] + t3 * U[U_TEXTURE + 7];
            fl = ((b0 >= 0.f && b0 <= 1.f) ? AX_BORDER : 0) | ((t0 >= 0.f && t0 <= 1.f) ? AX_TX : 0) | ((u >= 0.f && u <= 1.f) ? AX_UV : 0);
            rfl = ((b1 >= 0.f && b1 <= 1.f) ? AX_BORDER : 0) | ((t1 >= 0.f && t1 <= 1.f) ? AX_TX : 0) | ((v >= 0.f && v <= 1.f) ? AX_UV : 0);
            lin_axis_raw(u, S0.w, cy, w.cya); lin_axis_raw(u, S1.w, cc, w.cca);
            lin_axis_raw(v, S0.h, ry, rya); lin_axis_raw(v, S1.h, rc, rca);
        } else {
            axis_entry_x(Ly.u, xe, sx, sy, S0.w, S1.w, cy, w.cya, cc, w.cca, fl);
            axis_entry_y(Ly.u, ye, sx, sy, S0.h, S1.h, ry, rya, rc, rca, rfl);
        }
        const AxisSum cs = axis_summary(~0ull, col_in, fl, cy, cc);
        w.cfl = col_in ? fl : AX_ALL;                               // past the canvas edge: never stored; copy of the last column
        const AxisSum rs = axis_summary(ROWMASK, row_in, rfl, ry, rc);
        if (!row_in) rfl = AX_ALL;
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
                w.g0.edge = cs.lo < 0 || cs.hi >= S0.w || rs.lo < 0 || rs.hi >= S0.h - 1 + (int)(col0 + nvec * tpv <= S0.w);
                stage_slots_init(w.g0);
                ok = (nvec + 2) * 16 <= p0pitch && w.g0.rows <= p0rows && stage_slots(w.g0) <= 1024;
                c0off = base0 + 16 + ((cy - col0) << (4 - sh));     // byte of tap 0 in LDS row 0
            }
            if (ok && !rgb) {
                const int sh = Ly.kind == LK_BGRA_FROM_Y420P ? 4 : 3;
                const int tpv = 1 << sh;
                const int col0 = max(cs.clo, 0) & ~(tpv - 1);
                const int nvec = ((min(cs.chi, S1.w - 1) - col0) >> sh) + 1;
                w.g1.r_lo = rs.clo; w.g1.rows = rs.chi - rs.clo + 1; w.g1.b0 = col0 << (4 - sh); w.g1.nvec = nvec;
                w.g1.edge = cs.clo < 0 || cs.chi >= S1.w || rs.clo < 0 || rs.chi >= S1.h - 1 + (int)(col0 + nvec * tpv <= S1.w);
                stage_slots_init(w.g1);
                ok = (nvec + 2) * 16 <= p1pitch && w.g1.rows <= p1rows && stage_slots(w.g1) <= 1024;
                c1off = base1 + 16 + ((cc - col0) << (4 - sh));
                r1off = (rc - rs.clo) * p1pitch;
            }
            if (ok) {
                w.staged = true;
                cyo = c0off; cco = c1off;
                yoff = (ry - rs.lo) * p0pitch; coff = r1off;
            }
        }
        w.cyo = cyo; w.cco = cco;
        if (lane < WTH) {
            rowtab[2 * lane] = make_uint4((uint32_t)yoff, (uint32_t)coff, (uint32_t)rfl, 0u);
            rowtab[2 * lane + 1] = make_uint4(__float_as_uint(rya), __float_as_uint(1.0f - rya), __float_as_uint(rca), __float_as_uint(1.0f - rca));
        }
    };

    uint4 regs[WNR];
    auto prefetch = [&](int l, const WLayer &w) {                  // issue the global loads of layer l's rectangles
        const DLayer &Ly = L[l];
        if (Ly.kind == LK_BGRA_FROM_RGB) {
            wstage_load<0, WN_RGB>(regs, Ly.src.pl[0], w.g0, lane);
        } else {
            wstage_load<0, WN_Y>(regs, Ly.src.pl[0], w.g0, lane);
            wstage_load<WN_Y, WN_C>(regs, Ly.src.pl[1], w.g1, lane);
            if (Ly.kind == LK_BGRA_FROM_Y420P) wstage_load<WN_Y + WN_C, WN_C>(regs, Ly.src.pl[2], w.g1, lane);
        }
    };
    auto commit = [&](int l, const WLayer &w) {                    // registers -> this wave's LDS region; then the tails
        const DLayer &Ly = L[l];
        if (Ly.kind == LK_BGRA_FROM_RGB) {
            wstage_store<4, 0, WN_RGB>(regs, smem + base0, p0pitch, Ly.src.pl[0], w.g0, lane, Ly.swizzle != 0);   // RGBA -> BGRA on the way
            wstage_tail<4, WN_RGB>(smem + base0, p0pitch, Ly.src.pl[0], w.g0, lane, Ly.swizzle != 0);
        } else if (Ly.kind == LK_BGRA_FROM_NV12) {
            wstage_store<1, 0, WN_Y>(regs, smem + base0, p0pitch, Ly.src.pl[0], w.g0, lane, false);
            wstage_store<2, WN_Y, WN_C>(regs, smem + base1, p1pitch, Ly.src.pl[1], w.g1, lane, false);
            wstage_tail<1, WN_Y>(smem + base0, p0pitch, Ly.src.pl[0], w.g0, lane, false);
            wstage_tail<2, WN_C>(smem + base1, p1pitch, Ly.src.pl[1], w.g1, lane, false);
        } else {
            wstage_store<1, 0, WN_Y>(regs, smem + base0, p0pitch, Ly.src.pl[0], w.g0, lane, false);
            wstage_store<1, WN_Y, WN_C>(regs, smem + base1, p1pitch, Ly.src.pl[1], w.g1, lane, false);
            wstage_store<1, WN_Y + WN_C, WN_C>(regs, smem + base1 + voff, p1pitch, Ly.src.pl[2], w.g1, lane, false);
            wstage_tail<1, WN_Y>(smem + base0, p0pitch, Ly.src.pl[0], w.g0, lane, false);
            wstage_tail<1, WN_C>(smem + base1, p1pitch, Ly.src.pl[1], w.g1, lane, false);
            wstage_tail<1, WN_C>(smem + base1 + voff, p1pitch, Ly.src.pl[2], w.g1, lane, false);
        }
    };

    // layers whose border quad cannot touch the strip are skipped (uniform test against the host-computed bounding box)
    auto next_hit = [&](int l) {
        for (; l < nl; l++) {
            const int *bb = L[l].bbox;
            if (!(x0 + WTW <= bb[0] || x0 >= bb[2] || y0 + WTH <= bb[1] || y0 >= bb[3])) break;
        }
        return l;
    };

    // ---- canvas pixels of this lane: packed BGRA codes, row j in cv[j] -----------------------------------------------
    uint32_t cv[WTH];
#pragma unroll
    for (int j = 0; j < WTH; j++) cv[j] = 0xFF000000u;               // img_clear_bgra: (0,0,0,1)
    if (!CLEAR && col_in) {
#pragma unroll
        for (int j = 0; j < WTH; j++)
            if (y0 + j < T.H) cv[j] = gld<uint32_t>(D.ptr + (size_t)(y0 + j) * D.pitch + (size_t)x * 4);
    }

    WLayer cur;
    int l = next_hit(0);

    while (l < nl) {
        const DLayer &Ly = L[l];
        layer_setup(l, cur);              // (overwrites the row table: the previous layer's pixels are done)
        if (cur.staged) {
            prefetch(l, cur);
            touch_regs(regs);             // one wait for all of the layer's loads (see touch_regs)
            commit(l, cur);
        }
        wave_lds_fence();
        const int ln = next_hit(l + 1);

        {
            const float *U = Ly.u;
            const float opacity = U[U_OPACITY];
            const bool rgb = Ly.kind == LK_BGRA_FROM_RGB;
            const bool planar = Ly.kind == LK_BGRA_FROM_Y420P;
            const bool nofill = (Ly.flags & LF_NO_FILL) != 0;
            // opacity in [0,1] and no fill: every blend is a convex combination of code values, so neither the clamp of the
            // fill step nor the saturation of the store can trigger; with every pixel of the strip inside the picture the
            // loop is branch-free
            const bool fast = cur.staged && cur.all_inside && nofill && opacity >= 0.f && opacity <= 1.f;
            if (fast && rgb) {
                const float ka = opacity * kInv255;
                const float a = cur.cya, ia = 1.0f - a;
#pragma unroll
                for (int j = 0; j < WTH; j++) {
                    WAVE_ROW_FENCE(j);
                    const uint4 ra = rowtab[2 * j], rb = rowtab[2 * j + 1];
                    const float b = __uint_as_float(rb.x), ib = __uint_as_float(rb.y);
                    const uint8_t *p0 = smem + ((int)ra.x + cur.cyo);
                    const uint8_t *p1 = p0 + p0pitch;
                    const uint32_t u00 = ((const uint32_t *)p0)[0], u10 = ((const uint32_t *)p0)[1];
                    const uint32_t u01 = ((const uint32_t *)p1)[0], u11 = ((const uint32_t *)p1)[1];
                    const float w00 = ia * ib, w10 = a * ib, w01 = ia * b, w11 = a * b;
                    const float q0 = cs_mix(w00, w10, w01, w11, ub0(u00), ub0(u10), ub0(u01), ub0(u11));
                    const float q1 = cs_mix(w00, w10, w01, w11, ub1(u00), ub1(u10), ub1(u01), ub1(u11));
                    const float q2 = cs_mix(w00, w10, w01, w11, ub2(u00), ub2(u10), ub2(u01), ub2(u11));
                    const float q3 = cs_mix(w00, w10, w01, w11, ub3(u00), ub3(u10), ub3(u01), ub3(u11));
                    const float al = q3 * ka, ial = 1.f - al;
                    const uint32_t c = cv[j];                                      // staged texels are BGRA whatever the source order
                    cv[j] = pack_codes(__builtin_fmaf(q0, al, ub0(c) * ial), __builtin_fmaf(q1, al, ub1(c) * ial),
                                       __builtin_fmaf(q2, al, ub2(c) * ial), 0xFF000000u);
                }
            } else if (fast) {
                const CscFolded cscb = csc_fold_biased(kCsc[Ly.csc & 3]);
                const float al = 1.0f * opacity, ial = 1.f - al;
                const float ya = cur.cya, iya = 1.0f - ya, ca = cur.cca, ica = 1.0f - ca;
                auto yuv_fast = [&](auto planar_c, auto opaque_c) {
                    constexpr bool PL = decltype(planar_c)::value, OP = decltype(opaque_c)::value;
#pragma unroll
                    for (int j = 0; j < WTH; j++) {
                        WAVE_ROW_FENCE(j);
                        const uint4 ra = rowtab[2 * j], rb = rowtab[2 * j + 1];
                        const float yb = __uint_as_float(rb.x), iyb = __uint_as_float(rb.y), cbw = __uint_as_float(rb.z), icb = __uint_as_float(rb.w);
                        const int yo = (int)ra.x + cur.cyo, co = (int)ra.y + cur.cco;
                        const float w00 = iya * iyb, w10 = ya * iyb, w01 = iya * yb, w11 = ya * yb;
                        const float c00 = ica * icb, c10 = ca * icb, c01 = ica * cbw, c11 = ca * cbw;
                        float fy, fu, fv;
                        if constexpr (PL) sample_y420p_lds_bytes(smem, yo, p0pitch, co, voff, p1pitch, w00, w10, w01, w11, c00, c10, c01, c11, fy, fu, fv);
                        else sample_nv12_lds_bytes(smem, yo, p0pitch, co, p1pitch, w00, w10, w01, w11, c00, c10, c01, c11, fy, fu, fv);
                        if constexpr (OP) {
                            cv[j] = yuv_to_bgra_word(cscb, (int)code_biased(fy), (int)code_biased(fu), (int)code_biased(fv));   // fma(p, 1, c * 0) = p exactly
                        } else {
                            float pb, pg, pr;
                            yuv_to_bgr_floats(cscb, (int)code_biased(fy), (int)code_biased(fu), (int)code_biased(fv), pb, pg, pr);
                            const uint32_t c = cv[j];
                            cv[j] = pack_codes(__builtin_fmaf(pb, al, ub0(c) * ial), __builtin_fmaf(pg, al, ub1(c) * ial),
                                               __builtin_fmaf(pr, al, ub2(c) * ial), 0xFF000000u);
                        }
                    }
                };
                const bool opaque = (Ly.flags & LF_OPAQUE) != 0;
                if (planar) { if (opaque) yuv_fast(std::true_type{}, std::true_type{}); else yuv_fast(std::true_type{}, std::false_type{}); }
                else        { if (opaque) yuv_fast(std::false_type{}, std::true_type{}); else yuv_fast(std::false_type{}, std::false_type{}); }
            } else {
                // strips on a picture / border edge, fill colours, opacities outside [0,1], unstaged rectangles: one row at a
                // time, one copy of the code (the canvas registers are reached through select chains)
                const float af = opacity * U[U_FILL + 3], iaf = 1.f - af;
                const float f_b = U[U_FILL + 2] * 255.0f, f_g = U[U_FILL + 1] * 255.0f, f_r = U[U_FILL + 0] * 255.0f;
                const Csc &csc = kCsc[Ly.csc & 3];
                const DPlane &S0 = Ly.src.pl[0];
                const DPlane &S1 = Ly.src.pl[rgb ? 0 : 1];
                const DPlane &S2 = Ly.src.pl[planar ? 2 : (rgb ? 0 : 1)];
#pragma unroll 1
                for (int j = 0; j < WTH; j++) {
                    const uint4 ra = rowtab[2 * j], rb = rowtab[2 * j + 1];
                    const int fl = cur.cfl & (int)ra.z;
                    if (!(fl & AX_BORDER)) continue;
                    const int ry = (int)ra.x, rc = (int)ra.y;       // staged: LDS row offsets; unstaged: tap-0 row positions
                    const float b = __uint_as_float(rb.x), ib = __uint_as_float(rb.y), cbw = __uint_as_float(rb.z), icb = __uint_as_float(rb.w);
                    uint32_t c = cv[0];
#pragma unroll
                    for (int s = 1; s < WTH; s++) c = j == s ? cv[s] : c;
                    float r0 = clampf(__builtin_fmaf(f_b, af, ub0(c) * iaf), 0.f, 255.f);
                    float r1 = clampf(__builtin_fmaf(f_g, af, ub1(c) * iaf), 0.f, 255.f);
                    float r2 = clampf(__builtin_fmaf(f_r, af, ub2(c) * iaf), 0.f, 255.f);
                    if ((fl & (AX_TX | AX_UV)) == (AX_TX | AX_UV)) {
                        const float a = cur.cya, ia = 1.0f - a;
                        const float w00 = ia * ib, w10 = a * ib, w01 = ia * b, w11 = a * b;
                        float p0, p1, p2, al;
                        if (rgb) {
                            uint32_t u00, u10, u01, u11;
                            if (cur.staged) {
                                const uint8_t *q0 = smem + (ry + cur.cyo);
                                const uint8_t *q1 = q0 + p0pitch;
                                u00 = ((const uint32_t *)q0)[0]; u10 = ((const uint32_t *)q0)[1]; u01 = ((const uint32_t *)q1)[0]; u11 = ((const uint32_t *)q1)[1];
                            } else {
                                int xa = min(max(cur.cyo, 0), S0.w - 1), xb = min(max(cur.cyo + 1, 0), S0.w - 1);     // unstaged: cyo is the texel position
                                int ya = min(max(ry, 0), S0.h - 1), yb = min(max(ry + 1, 0), S0.h - 1);
                                auto ld = [&](int xx, int yy) { return gld<uint32_t>(S0.ptr + (size_t)yy * S0.pitch + (size_t)xx * 4); };
                                u00 = ld(xa, ya); u10 = ld(xb, ya); u01 = ld(xa, yb); u11 = ld(xb, yb);
                            }
                            const float s0 = cs_mix(w00, w10, w01, w11, ub0(u00), ub0(u10), ub0(u01), ub0(u11));
                            const float s1 = cs_mix(w00, w10, w01, w11, ub1(u00), ub1(u10), ub1(u01), ub1(u11));
                            const float s2 = cs_mix(w00, w10, w01, w11, ub2(u00), ub2(u10), ub2(u01), ub2(u11));
                            const float s3 = cs_mix(w00, w10, w01, w11, ub3(u00), ub3(u10), ub3(u01), ub3(u11));
                            const bool swz = !cur.staged && Ly.swizzle;     // taps gathered from global memory keep the source order
                            p0 = swz ? s2 : s0; p1 = s1; p2 = swz ? s0 : s2;
                            al = s3 * (opacity * kInv255);
                        } else {
                            const float ca = cur.cca, ica = 1.0f - ca;
                            float fy, fu, fv;
                            if (cur.staged) {
                                const int ya = ry + cur.cyo, cao = rc + cur.cco;
                                if (planar) sample_y420p_lds_bytes(smem, ya, p0pitch, cao, voff, p1pitch, w00, w10, w01, w11,
                                                                   ica * icb, ca * icb, ica * cbw, ca * cbw, fy, fu, fv);
                                else sample_nv12_lds_bytes(smem, ya, p0pitch, cao, p1pitch, w00, w10, w01, w11,
                                                           ica * icb, ca * icb, ica * cbw, ca * cbw, fy, fu, fv);
                            } else {
                                sample_nv12_global(S0, S1, planar ? &S2 : nullptr, cur.cyo, ry, cur.cco, rc, w00, w10, w01, w11,
                                                   ica * icb, ca * icb, ica * cbw, ca * cbw, fy, fu, fv);
                            }
                            const uint32_t wd = yuv_to_bgra_word(csc, (int)to_code_raw(fy), (int)to_code_raw(fu), (int)to_code_raw(fv));
                            p0 = ub0(wd); p1 = ub1(wd); p2 = ub2(wd);
                            al = 1.0f * opacity;
                        }
                        const float ial = 1.f - al;
                        r0 = __builtin_fmaf(p0, al, r0 * ial);
                        r1 = __builtin_fmaf(p1, al, r1 * ial);
                        r2 = __builtin_fmaf(p2, al, r2 * ial);
                    }
                    const uint32_t res = pack_codes(r0, r1, r2, 0xFF000000u);       // RTE, saturated, NaN -> 0; alpha forced to 1
#pragma unroll
                    for (int s = 0; s < WTH; s++) cv[s] = j == s ? res : cv[s];
                }
            }
        }
        wave_lds_fence();                 // the taps of layer l are read before the next layer's rectangle overwrites them
        l = ln;
    }

    if (col_in) {
#pragma unroll
        for (int j = 0; j < WTH; j++)
            if (y0 + j < T.H) gst<uint32_t>(D.ptr + (size_t)(y0 + j) * D.pitch + (size_t)x * 4, cv[j]);
    }
}

// ---------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------
static bool finite16w(const float *m) {
    for (int i = 0; i < 16; i++) if (!(m[i] - m[i] == 0.f)) return false;
    return true;
}
static bool aligned16w(const DPlane &p) { return (((uintptr_t)p.ptr) & 15) == 0 && (p.pitch & 15) == 0 && p.w * p.comps >= 16; }

struct WaveDims { int p0pitch, p0rows, p1pitch, p1rows; };

// LDS rectangles one strip of this layer can touch, from the layer's scale factors
static WaveDims wave_dims(const DTick &T, const DLayer &L) {
    const float *U = L.u;
    double sxr = std::fabs((double)U[U_TEXTURE + 0] * (double)U[U_TRANSFORM + 0] * 2.0 / (double)T.W);
    double syr = std::fabs((double)U[U_TEXTURE + 5] * (double)U[U_TRANSFORM + 5] * 2.0 / (double)T.H);
    WaveDims d{ 0, 0, 0, 0 };
    const bool rgb = L.kind == LK_BGRA_FROM_RGB, planar = L.kind == LK_BGRA_FROM_Y420P;
    const int bpt0 = rgb ? 4 : 1;
    int span0 = (int)std::ceil(WTW * sxr * L.src.pl[0].w) + 4;           // texels incl. tap 1 and rounding slack
    d.p0pitch = ((span0 * bpt0 + 15) / 16 + 3) * 16;                      // vectors + alignment + 2 pad vectors
    d.p0rows = (int)std::ceil(WTH * syr * L.src.pl[0].h) + 3;
    if (!rgb) {
        const int bpt1 = planar ? 1 : 2;
        int span1 = (int)std::ceil(WTW * sxr * L.src.pl[1].w) + 4;
        d.p1pitch = ((span1 * bpt1 + 15) / 16 + 3) * 16;
        d.p1rows = (int)std::ceil(WTH * syr * L.src.pl[1].h) + 3;
    }
    return d;
}
static size_t wave_lds(const WaveDims &d, bool planar) {
    return (size_t)WAVES * ((size_t)ROWTAB_BYTES + (size_t)d.p0pitch * d.p0rows + (size_t)d.p1pitch * d.p1rows * (planar ? 2 : 1));
}

bool wave_layers_eligible(const DTick *ticks, const DLayer *layers, int n_ticks) {
    for (int i = 0; i < n_ticks; i++) {
        const DTick &T = ticks[i];
        if (T.n_layers < 1 || T.clear_first != ticks[0].clear_first) return false;
        if (!aligned16w(T.dst.pl[0])) return false;
        for (int l = 0; l < T.n_layers; l++) {
            const DLayer &L = layers[T.first_layer + l];
            const bool rgb = L.kind == LK_BGRA_FROM_RGB, nv12 = L.kind == LK_BGRA_FROM_NV12, planar = L.kind == LK_BGRA_FROM_Y420P;
            if (!(rgb || nv12 || planar) || !(L.flags & LF_AXIS_ALIGNED)) return false;
            if (!finite16w(L.u + U_TRANSFORM) || !finite16w(L.u + U_TEXTURE) || !finite16w(L.u + U_BORDER)) return false;
            const int np = rgb ? 1 : nv12 ? 2 : 3;
            for (int p = 0; p < np; p++) if (!aligned16w(L.src.pl[p])) return false;
            if (wave_lds(wave_dims(T, L), planar) > (size_t)LDS_BUDGET) return false;
        }
    }
    return true;
}

hipError_t launch_wave_layers(const DTick *ticks_host, const DLayer *layers_host, const DTick *ticks, const DLayer *layers,
                              int n_ticks, int maxW, int maxH, hipStream_t stream) {
    WaveDims m{ 0, 0, 0, 0 };
    bool planar = false;
    for (int i = 0; i < n_ticks; i++) {
        for (int l = 0; l < ticks_host[i].n_layers; l++) {
            const DLayer &L = layers_host[ticks_host[i].first_layer + l];
            WaveDims d = wave_dims(ticks_host[i], L);
            m.p0pitch = std::max(m.p0pitch, d.p0pitch); m.p0rows = std::max(m.p0rows, d.p0rows);
            m.p1pitch = std::max(m.p1pitch, d.p1pitch); m.p1rows = std::max(m.p1rows, d.p1rows);
            planar = planar || L.kind == LK_BGRA_FROM_Y420P;
        }
    }
    size_t lds = wave_lds(m, planar);
    if (lds > (size_t)LDS_BUDGET) {
        // per-layer maxima combined exceed the budget: shrink the row counts; rectangles that do not fit fall back to
        // unstaged taps inside the kernel
        const size_t per_row = (size_t)WAVES * ((size_t)m.p0pitch + (size_t)m.p1pitch * (planar ? 2 : 1));
        int rows = std::max(1, (int)((LDS_BUDGET - WAVES * ROWTAB_BYTES) / per_row));
        m.p0rows = std::min(m.p0rows, rows); m.p1rows = std::min(m.p1rows, rows);
        lds = wave_lds(m, planar);
    }
    int strips_x = (maxW + WTW - 1) / WTW, strips_y = (maxH + WTH - 1) / WTH;
    long total = (long)n_ticks * strips_x * strips_y;
    long per_xcd = (total + 7) / 8;
    long blocks_per_xcd = (per_xcd + WAVES - 1) / WAVES;
    dim3 grid((unsigned)(blocks_per_xcd * 8));
    if (ticks_host[0].clear_first)
        hipLaunchKernelGGL(tick_bgra_wave<true>, grid, dim3(NTHREADS), lds, stream, ticks, layers, n_ticks, strips_x, strips_y,
                           m.p0pitch, m.p0rows, m.p1pitch, m.p1rows, planar ? 1 : 0);
    else
        hipLaunchKernelGGL(tick_bgra_wave<false>, grid, dim3(NTHREADS), lds, stream, ticks, layers, n_ticks, strips_x, strips_y,
                           m.p0pitch, m.p0rows, m.p1pitch, m.p1rows, planar ? 1 : 0);
    return hipGetLastError();
}

}  // namespace chv

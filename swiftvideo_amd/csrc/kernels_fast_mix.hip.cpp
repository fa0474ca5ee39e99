// kernels_fast_mix.hip.cpp — axis-aligned, LDS-tiled tick kernel for BGRA canvases whose layers are ANY mix of
// NV12 / y420p / BGRA / RGBA pictures: the literal "NV12 -> BGRA + scale + N-layer composite" tick that
// VideoMixer.mix issues when decoded YUV streams and RGB overlays land on one BGRA canvas
// (mix.video.swift:114-124, findKernel :142-146 -> img_nv12_bgra / img_y420p_bgra / img_{bgra,rgba}_bgra_tx).
//
// Structure (that of kernels_fast_rgb.hip.cpp, with a per-layer sampler):
//   a block owns a 64x32 canvas tile and keeps its pixels (8 per thread) in registers as float codes across all
//   layers: the canvas is written once and every layer's source bytes leave HBM once (algorithmic minimum);
//   phase 0   per-layer column/row tables of the reference's coordinate arithmetic for the luma (or RGB) plane and the
//             chroma plane (same instruction sequence as the general kernel => same bits);
//   per layer [the prefetched source rectangles go to LDS as bytes: luma + (u,v) pairs / U + V tiles / 4-byte texels,
//             edge texels replicated | barrier | the NEXT layer's global loads are issued into registers | 2x2 taps
//             from LDS, v_cvt_f32_ubyteN, code-scale FMAs, integer colour matrix (YUV), blend, re-quantise | barrier].
//   Layers whose bounding box misses the tile are skipped; layers underneath an opaque YUV picture that covers the
//   whole tile are skipped as well (their contribution is multiplied by exactly 0: fma(p, 1, r * 0) = p).
// Every path produces exactly the bytes of kernels_general.hip.cpp (= oracle/ref_kernels.c::px_to_bgra, layer by
// layer with the canvas re-quantised in between).
#include "tile_common.hip.h"

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <type_traits>

#pragma clang fp contract(off)

namespace chv {

constexpr int MTW = 64;           // tile width  (output pixels): 16 threads x MCOLS columns (txi + 16k)
constexpr int MTH = 32;           // tile height (output rows):   16 thread rows x MROWS rows (ly + 16r)
constexpr int MCOLS = 4, MROWS = 2, MPX = MCOLS * MROWS;
constexpr int MMAXL = 8;          // layers per tick this path accepts
constexpr int MNR = 4;            // prefetch registers (16-byte vectors) per thread:
                                  //   RGB layer: [0..2] plane 0;  YUV layer: [0..1] luma, [2] chroma (U), [3] V (planar)

struct MixLayerTable {
    int cy[MTW]; float cya[MTW];      // plane 0 (luma / RGB texels) column: unclamped tap-0 position, weight of tap 1
    int cc[MTW]; float cca[MTW];      // chroma column (YUV sources)
    int cfl[MTW];
    int ry[MTH]; float rya[MTH];
    int rc[MTH]; float rca[MTH];
    int rfl[MTH];
    int csum[8];                      // {min p0, max p0 + 1, min chroma, max chroma + 1, any inside, all inside, -, -}
    int rsum[8];
};

struct MixGeom {
    StageGeom g0, g1;                 // plane 0; chroma plane(s)
    int col0, ccol0;                  // first staged texel column of plane 0 / of the chroma plane
    bool staged;
};

template <int OFF, int N>
CHV_DEV void mstage_load(uint4 (&regs)[MNR], const DPlane &P, const StageGeom &g, int tid) {
    // as stage_load (tile_common.hip.h): exactly one global_load_dwordx4 per slot, straight into its final register
#pragma unroll
    for (int n = 0; n < N; n++) {
        int i = tid + n * NTHREADS, r, vv;
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
CHV_DEV void mstage_put(uint4 val, int i, uint8_t *lds, int lds_pitch, const DPlane &P, const StageGeom &g, bool swap02) {
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
CHV_DEV void mstage_store(const uint4 (&regs)[MNR], uint8_t *lds, int lds_pitch, const DPlane &P, const StageGeom &g, int tid, bool swap02) {
#pragma unroll
    for (int n = 0; n < N; n++) mstage_put<BPT>(regs[OFF + n], tid + n * NTHREADS, lds, lds_pitch, P, g, swap02);
    // slots beyond the registers' capacity: loaded and written on the spot (latency exposed; rare: strong downscales,
    // rectangles at a picture edge)
    for (int base = N * NTHREADS; base < stage_slots(g); base += NTHREADS) {
        int i = base + tid, r, vv;
        stage_slot(g, i, r, vv);
        uint4 val = make_uint4(0, 0, 0, 0);
        if (i < 1024 && r < g.rows) {
            int row = min(max(g.r_lo + r, 0), P.h - 1);
            int off = g.b0 + (g.edge ? vv - 1 : vv) * 16;
            if (g.edge) off = vec_loadable(P, row, off) ? off : 0;
            val = gld<uint4>(P.ptr + (size_t)row * P.pitch + off);
        }
        mstage_put<BPT>(val, i, lds, lds_pitch, P, g, swap02);
    }
}

// store conversion of a code-scale value kept as a float: RTE, saturated, NaN -> 0
CHV_DEV float mix_to_codef(float v) {
    v = __builtin_rintf(v);
    return __builtin_fminf(__builtin_fmaxf(v, 0.0f), 255.0f);
}

// Integer colour matrix on biased codes, channels returned as float codes: the two v_ashr_pk_u8_i32
// saturate and narrow, v_cvt_f32_ubyteN widens — the packed BGRA word is never formed.
CHV_DEV void yuv_to_bgr_floats(const CscFolded &k, int y, int u, int v, float &fb, float &fg, float &fr) {
    int32_t t = __mul24(y, k.cy);
    int32_t r = mad24_uniform(v, k.crv, t) + k.kr;
    int32_t g = mad24_uniform(v, k.ncgv, mad24_uniform(u, k.ncgu, t)) + k.kg;
    int32_t b = mad24_uniform(u, k.cbu, t) + k.kb;
    uint32_t bg, ra;
    asm("v_ashr_pk_u8_i32 %0, %1, %2, 16" : "=v"(bg) : "v"(b), "v"(g));          // byte0 = B, byte1 = G
    asm("v_ashr_pk_u8_i32 %0, %1, %2, 16" : "=v"(ra) : "v"(r), "v"(0));          // byte0 = R
    fb = (float)(bg & 255u); fg = (float)((bg >> 8) & 255u); fr = (float)(ra & 255u);
}

#ifndef CHV_MIX_MINW
#define CHV_MIX_MINW 4
#endif
// CHV_MIX_ROWFENCE: keep the scheduler from interleaving the rows of a thread's pixels in the branch-free loops (fewer
// live temporaries; the 8 pixels of a thread otherwise want ~127 VGPRs)
#ifndef CHV_MIX_ROWFENCE
#define CHV_MIX_ROWFENCE 0
#endif
#if CHV_MIX_ROWFENCE
#define MIX_ROW_FENCE() __builtin_amdgcn_sched_barrier(0)
#else
#define MIX_ROW_FENCE() ((void)0)
#endif
#if CHV_MIX_ROWFENCE == 2     // additionally between the pixel pairs of a row
#define MIX_PAIR_FENCE(k) do { if ((k) == 2) __builtin_amdgcn_sched_barrier(0); } while (0)
#else
#define MIX_PAIR_FENCE(k) ((void)0)
#endif
template <bool CLEAR>
__global__ __launch_bounds__(NTHREADS, CHV_MIX_MINW) void tick_mix_layers_tiled(const DTick *__restrict__ ticks,
                                                                                const DLayer *__restrict__ layers,
                                                                                int n_ticks, int tiles_x, int tiles_y,
                                                                                int p0pitch, int p0rows, int p1pitch, int p1rows,
                                                                                int max_layers) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    MixLayerTable *tabs = (MixLayerTable *)smem;                              // [max_layers] (the launch's deepest tick)
    int *scratch = (int *)(smem + sizeof(MixLayerTable) * max_layers);        // sink for summaries of absent layers
    const int base0 = (int)(sizeof(MixLayerTable) * max_layers) + 64;         // [p0rows][p0pitch] luma bytes / 4-byte texels
    const int base1 = base0 + p0rows * p0pitch;                               // [p1rows][p1pitch] (u,v) pairs (NV12) / U tile (planar)
    const int voff = p1rows * p1pitch;                                        // planar: the V tile follows the U tile

    // XCD-aware numbering: block b runs on XCD b % 8; every XCD gets one contiguous range of the launch's tiles
    const int bid = blockIdx.x;
    const int xcd = bid & 7, slot = bid >> 3;
    const int tiles = tiles_x * tiles_y;
    const int total = tiles * n_ticks, per_xcd = (total + 7) >> 3;
    const int index = xcd * per_xcd + slot;
    if (slot >= per_xcd || index >= total) return;
    const int tick = index / tiles;
    const int tile = index - tick * tiles;
    const DTick &T = ticks[tick];
    const int x0 = (tile % tiles_x) * MTW, y0 = (tile / tiles_x) * MTH;
    if (x0 >= T.W || y0 >= T.H) return;
    const DLayer *L = layers + T.first_layer;
    const int nl = T.n_layers;
    const DPlane &D = T.dst.pl[0];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const float sx = (float)T.W, sy = (float)T.H;

    // ---- phase 0: tables of every layer ---------------------------------------------------
    for (int l = wave; l < nl; l += 4) {                         // one wave = the 64 columns of one layer
        const DPlane &S0 = L[l].src.pl[0];
        const DPlane &S1 = L[l].src.pl[L[l].kind == LK_BGRA_FROM_RGB ? 0 : 1];
        int x = x0 + lane, iy, ic, fl; float ay, ac;
        axis_entry_x(L[l].u, min(x, T.W - 1), sx, sy, S0.w, S1.w, iy, ay, ic, ac, fl);
        group_summary(tabs[l].csum, 6, x < T.W, fl, iy, ic);
        if (x >= T.W) fl = AX_ALL;                               // past the canvas edge: never stored; copy of the last column
        tabs[l].cy[lane] = iy; tabs[l].cya[lane] = ay; tabs[l].cc[lane] = ic; tabs[l].cca[lane] = ac; tabs[l].cfl[lane] = fl;
    }
    static_assert(MTH == 32, "row tables: one wave covers the 32 rows of two layers");
    for (int l2 = wave * 2; l2 < nl; l2 += 8) {
        int l = l2 + (lane >> 5), j = lane & 31;
        int lc = min(l, nl - 1);
        const DPlane &S0 = L[lc].src.pl[0];
        const DPlane &S1 = L[lc].src.pl[L[lc].kind == LK_BGRA_FROM_RGB ? 0 : 1];
        int y = y0 + j, iy, ic, fl; float ay, ac;
        axis_entry_y(L[lc].u, min(y, T.H - 1), sx, sy, S0.h, S1.h, iy, ay, ic, ac, fl);
        group_summary(l < nl ? tabs[l].rsum : scratch, 5, l < nl && y < T.H, fl, iy, ic);
        if (y >= T.H) fl = AX_ALL;
        if (l < nl) { tabs[l].ry[j] = iy; tabs[l].rya[j] = ay; tabs[l].rc[j] = ic; tabs[l].rca[j] = ac; tabs[l].rfl[j] = fl; }
    }
    __syncthreads();

    // staging geometry of layer l's source rectangles for this tile.  The summaries come out of LDS (vector registers);
    // they are block-uniform, so they are moved to scalar registers and everything derived from them stays scalar.
    auto layer_geom = [&](int l, MixGeom &m) {
        const MixLayerTable &t = tabs[l];
        const DLayer &Ly = L[l];
        m.staged = false; m.col0 = 0; m.ccol0 = 0;
        m.g0.r_lo = 0; m.g1.r_lo = 0;
        const int c0 = __builtin_amdgcn_readfirstlane(t.csum[0]), c1 = __builtin_amdgcn_readfirstlane(t.csum[1]);
        const int r0 = __builtin_amdgcn_readfirstlane(t.rsum[0]), r1 = __builtin_amdgcn_readfirstlane(t.rsum[1]);
        if (!(c1 > c0 && r1 > r0)) return;
        const bool rgb = Ly.kind == LK_BGRA_FROM_RGB;
        {
            const DPlane &S = Ly.src.pl[0];
            const int sh = rgb ? 2 : 4;                          // log2(texels per 16-byte vector)
            const int tpv = 1 << sh;
            m.col0 = max(c0, 0) & ~(tpv - 1);
            const int nvec = ((min(c1, S.w - 1) - m.col0) >> sh) + 1;
            m.g0.r_lo = r0; m.g0.rows = r1 - r0 + 1; m.g0.b0 = m.col0 << (4 - sh); m.g0.nvec = nvec;
            m.g0.edge = c0 < 0 || c1 >= S.w || r0 < 0 || r1 >= S.h - 1 + (int)(m.col0 + nvec * tpv <= S.w);
            stage_slots_init(m.g0);
            if (!((nvec + 2) * 16 <= p0pitch && m.g0.rows <= p0rows && stage_slots(m.g0) <= 1024)) return;
        }
        if (!rgb) {
            const DPlane &S = Ly.src.pl[1];
            const int sh = Ly.kind == LK_BGRA_FROM_Y420P ? 4 : 3;
            const int tpv = 1 << sh;
            const int lo = __builtin_amdgcn_readfirstlane(t.csum[2]), hi = __builtin_amdgcn_readfirstlane(t.csum[3]);
            const int q0 = __builtin_amdgcn_readfirstlane(t.rsum[2]), q1 = __builtin_amdgcn_readfirstlane(t.rsum[3]);
            m.ccol0 = max(lo, 0) & ~(tpv - 1);
            const int nvec = ((min(hi, S.w - 1) - m.ccol0) >> sh) + 1;
            m.g1.r_lo = q0; m.g1.rows = q1 - q0 + 1; m.g1.b0 = m.ccol0 << (4 - sh); m.g1.nvec = nvec;
            m.g1.edge = lo < 0 || hi >= S.w || q0 < 0 || q1 >= S.h - 1 + (int)(m.ccol0 + nvec * tpv <= S.w);
            stage_slots_init(m.g1);
            if (!((nvec + 2) * 16 <= p1pitch && m.g1.rows <= p1rows && stage_slots(m.g1) <= 1024)) return;
        }
        m.staged = true;
    };
    auto all_inside = [&](int l) {      // every canvas pixel of the tile lies inside layer l's picture
        const MixLayerTable &t = tabs[l];
        return __builtin_amdgcn_readfirstlane(t.csum[4] & t.csum[5] & t.rsum[4] & t.rsum[5]) != 0;
    };

    uint4 regs[MNR];
    auto prefetch = [&](int l, const MixGeom &m) {               // issue the global loads of layer l's rectangles
        const DLayer &Ly = L[l];
        if (Ly.kind == LK_BGRA_FROM_RGB) {
            mstage_load<0, 3>(regs, Ly.src.pl[0], m.g0, tid);
        } else {
            mstage_load<0, 2>(regs, Ly.src.pl[0], m.g0, tid);
            mstage_load<2, 1>(regs, Ly.src.pl[1], m.g1, tid);
            if (Ly.kind == LK_BGRA_FROM_Y420P) mstage_load<3, 1>(regs, Ly.src.pl[2], m.g1, tid);
        }
    };
    auto commit = [&](int l, const MixGeom &m) {                 // registers -> LDS
        const DLayer &Ly = L[l];
        if (Ly.kind == LK_BGRA_FROM_RGB) {
            mstage_store<4, 0, 3>(regs, smem + base0, p0pitch, Ly.src.pl[0], m.g0, tid, Ly.swizzle != 0);   // RGBA -> BGRA on the way
        } else if (Ly.kind == LK_BGRA_FROM_NV12) {
            mstage_store<1, 0, 2>(regs, smem + base0, p0pitch, Ly.src.pl[0], m.g0, tid, false);
            mstage_store<2, 2, 1>(regs, smem + base1, p1pitch, Ly.src.pl[1], m.g1, tid, false);
        } else {
            mstage_store<1, 0, 2>(regs, smem + base0, p0pitch, Ly.src.pl[0], m.g0, tid, false);
            mstage_store<1, 2, 1>(regs, smem + base1, p1pitch, Ly.src.pl[1], m.g1, tid, false);
            mstage_store<1, 3, 1>(regs, smem + base1 + voff, p1pitch, Ly.src.pl[2], m.g1, tid, false);
        }
    };

    // ---- which layers touch this tile -----------------------------------------------------------------
    // layers whose border quad cannot touch the tile are skipped (block-uniform test against the host-computed
    // bounding box); `next_hit` walks the remaining ones in z order
    auto next_hit = [&](int l) {
        for (; l < nl; l++) {
            const int *bb = L[l].bbox;
            if (!(x0 + MTW <= bb[0] || x0 >= bb[2] || y0 + MTH <= bb[1] || y0 >= bb[3])) break;
        }
        return l;
    };
    // the topmost opaque YUV picture that covers every canvas pixel of the tile hides all that lies beneath it:
    // result = fma(p, 1, r * 0) = p whatever r (r is finite: the fill step clamps), so start there
    int l_first = 0;
    for (int l = nl - 1; l > 0; l--) {
        const DLayer &Ly = L[l];
        if (Ly.kind != LK_BGRA_FROM_RGB && (Ly.flags & LF_OPAQUE) && all_inside(l)) { l_first = l; break; }
    }
    const bool covered = l_first > 0 ||
        (nl > 0 && L[0].kind != LK_BGRA_FROM_RGB && (L[0].flags & LF_OPAQUE) && all_inside(0));

    // ---- canvas pixels of this thread, as float code values -----------------------------------
    const int txi = tid & 15, ly = tid >> 4;
    const int xq = x0 + txi, yq = y0 + ly;              // this thread's pixels: (xq + 16*k, yq + 16*r)
    const bool active = xq < T.W && yq < T.H;
    float cb[MPX], cg[MPX], cr[MPX];                    // index r * MCOLS + k
    uint32_t orig_a[MROWS];                             // the MCOLS original alpha bytes of a row, packed
    unsigned touched = CLEAR ? ~0u : 0u;                // bit r*4+k; untouched pixels keep their original alpha byte
#pragma unroll
    for (int q = 0; q < MPX; q++) { cb[q] = 0.f; cg[q] = 0.f; cr[q] = 0.f; }   // img_clear_bgra: (0,0,0,1)
#pragma unroll
    for (int r = 0; r < MROWS; r++) orig_a[r] = 0;
    if (!CLEAR && active && !covered) {
#pragma unroll
        for (int r = 0; r < MROWS; r++) {
            if (yq + 16 * r >= T.H) continue;
            const uint8_t *drow = D.ptr + (size_t)(yq + 16 * r) * D.pitch;
#pragma unroll
            for (int k = 0; k < MCOLS; k++) {
                uint32_t cur = (xq + 16 * k < T.W) ? gld<uint32_t>(drow + (size_t)(xq + 16 * k) * 4) : 0;
                cb[r * MCOLS + k] = (float)(cur & 255); cg[r * MCOLS + k] = (float)((cur >> 8) & 255); cr[r * MCOLS + k] = (float)((cur >> 16) & 255);
                orig_a[r] |= (cur >> 24) << (8 * k);
            }
        }
    }

    int l = next_hit(l_first);
    if (l < nl) { MixGeom pm; layer_geom(l, pm); if (pm.staged) prefetch(l, pm); }

    while (l < nl) {
        const DLayer &Ly = L[l];
        const MixLayerTable &t = tabs[l];
        // the layer's geometry is recomputed here (a few scalar instructions) rather than carried through the previous
        // layer's pixel loop; only what the pixel loop needs of it stays live
        MixGeom m;
        layer_geom(l, m);
        touch_regs(regs);                 // the wait for the prefetch, on every path (see touch_regs)
        if (m.staged) commit(l, m);
        __syncthreads();
        const int ln = next_hit(l + 1);
        if (ln < nl) { MixGeom nm; layer_geom(ln, nm); if (nm.staged) prefetch(ln, nm); }

        if (active) {
            const float *U = Ly.u;
            const float opacity = U[U_OPACITY];
            const bool rgb = Ly.kind == LK_BGRA_FROM_RGB;
            const bool planar = Ly.kind == LK_BGRA_FROM_Y420P;
            const bool nofill = (Ly.flags & LF_NO_FILL) != 0;
            // opacity in [0,1] and no fill: every blend is a convex combination of code values, so neither the clamp of
            // the fill step nor the saturation of the store can trigger; with every pixel of the tile inside the picture
            // the loop is branch-free
            const bool fast = m.staged && __builtin_amdgcn_readfirstlane(t.csum[5] & t.rsum[5]) != 0 && nofill && opacity >= 0.f && opacity <= 1.f;
            const int off0 = base0 + ((rgb ? 4 : 16) - m.col0) * (rgb ? 4 : 1);      // + position * bytes per texel = LDS byte of a plane-0 texel in row 0
            const int ctb = planar ? 1 : 2;
            const int off1 = base1 + ((planar ? 16 : 8) - m.ccol0) * ctb;
            if (fast && rgb) {
                const float ka = opacity * kInv255;
#pragma unroll
                for (int r = 0; r < MROWS; r++) {
                    if (r) MIX_ROW_FENCE();
                    const int lr = ly + 16 * r;
                    const float b = t.rya[lr], ib = 1.0f - b;
                    const int rowoff = off0 + (t.ry[lr] - m.g0.r_lo) * p0pitch;
#pragma unroll
                    for (int k = 0; k < MCOLS; k++) {
                        MIX_PAIR_FENCE(k);
                        const int c = txi + 16 * k, q = r * MCOLS + k;
                        const float a = t.cya[c], ia = 1.0f - a;
                        const int cpo = rowoff + t.cy[c] * 4;
                        const float w00 = ia * ib, w10 = a * ib, w01 = ia * b, w11 = a * b;
                        const uint32_t *p0 = (const uint32_t *)(smem + cpo);
                        const uint32_t *p1 = (const uint32_t *)(smem + cpo + p0pitch);
                        const float4 t00 = codes4(p0[0]), t10 = codes4(p0[1]), t01 = codes4(p1[0]), t11 = codes4(p1[1]);
                        const float q0 = cs_mix(w00, w10, w01, w11, t00.x, t10.x, t01.x, t11.x);
                        const float q1 = cs_mix(w00, w10, w01, w11, t00.y, t10.y, t01.y, t11.y);
                        const float q2 = cs_mix(w00, w10, w01, w11, t00.z, t10.z, t01.z, t11.z);
                        const float q3 = cs_mix(w00, w10, w01, w11, t00.w, t10.w, t01.w, t11.w);
                        const float al = q3 * ka, ial = 1.f - al;
                        cb[q] = code_rintf(__builtin_fmaf(q0, al, cb[q] * ial));     // staged texels are BGRA whatever the source order
                        cg[q] = code_rintf(__builtin_fmaf(q1, al, cg[q] * ial));
                        cr[q] = code_rintf(__builtin_fmaf(q2, al, cr[q] * ial));
                    }
                }
                touched = ~0u;
            } else if (fast) {
                const CscFolded cscb = csc_fold_biased(kCsc[Ly.csc & 3]);
                const float al = 1.0f * opacity, ial = 1.f - al;
                auto yuv_fast = [&](auto planar_c, auto opaque_c) {
                    constexpr bool PL = decltype(planar_c)::value, OP = decltype(opaque_c)::value;
#pragma unroll
                    for (int r = 0; r < MROWS; r++) {
                        if (r) MIX_ROW_FENCE();
                        const int lr = ly + 16 * r;
                        const float yb = t.rya[lr], iyb = 1.0f - yb, cbw = t.rca[lr], icb = 1.0f - cbw;
                        const int yrow = off0 + (t.ry[lr] - m.g0.r_lo) * p0pitch;
                        const int crow = off1 + (t.rc[lr] - m.g1.r_lo) * p1pitch;
#pragma unroll
                        for (int k = 0; k < MCOLS; k++) {
                            MIX_PAIR_FENCE(k);
                            const int c = txi + 16 * k, q = r * MCOLS + k;
                            const float ya = t.cya[c], iya = 1.0f - ya, ca = t.cca[c], ica = 1.0f - ca;
                            float fy, fu, fv;
                            if constexpr (PL)
                                sample_y420p_lds_bytes(smem, yrow + t.cy[c], p0pitch, crow + t.cc[c], voff, p1pitch,
                                                       iya * iyb, ya * iyb, iya * yb, ya * yb, ica * icb, ca * icb, ica * cbw, ca * cbw, fy, fu, fv);
                            else
                                sample_nv12_lds_bytes(smem, yrow + t.cy[c], p0pitch, crow + t.cc[c] * 2, p1pitch,
                                                      iya * iyb, ya * iyb, iya * yb, ya * yb, ica * icb, ca * icb, ica * cbw, ca * cbw, fy, fu, fv);
                            float pb, pg, pr;
                            yuv_to_bgr_floats(cscb, (int)code_biased(fy), (int)code_biased(fu), (int)code_biased(fv), pb, pg, pr);
                            if constexpr (OP) {
                                cb[q] = pb; cg[q] = pg; cr[q] = pr;               // fma(p, 1, c * 0) = p exactly
                            } else {
                                cb[q] = code_rintf(__builtin_fmaf(pb, al, cb[q] * ial));
                                cg[q] = code_rintf(__builtin_fmaf(pg, al, cg[q] * ial));
                                cr[q] = code_rintf(__builtin_fmaf(pr, al, cr[q] * ial));
                            }
                        }
                    }
                };
                const bool opaque = (Ly.flags & LF_OPAQUE) != 0;
                if (planar) { if (opaque) yuv_fast(std::true_type{}, std::true_type{}); else yuv_fast(std::true_type{}, std::false_type{}); }
                else        { if (opaque) yuv_fast(std::false_type{}, std::true_type{}); else yuv_fast(std::false_type{}, std::false_type{}); }
                touched = ~0u;
            } else {
                // tiles on a picture / border edge, fill colours, opacities outside [0,1], unstaged rectangles: one pixel at a
                // time, one copy of the code (the canvas registers are reached through select chains)
                const float af = opacity * U[U_FILL + 3], iaf = 1.f - af;
                const float f_b = U[U_FILL + 2] * 255.0f, f_g = U[U_FILL + 1] * 255.0f, f_r = U[U_FILL + 0] * 255.0f;
                const Csc &csc = kCsc[Ly.csc & 3];
                const DPlane &S0 = Ly.src.pl[0];
                const DPlane &S1 = Ly.src.pl[rgb ? 0 : 1];
                const DPlane &S2 = Ly.src.pl[planar ? 2 : (rgb ? 0 : 1)];
#pragma unroll 1
                for (int q = 0; q < MPX; q++) {
                    const int lr = ly + 16 * (q / MCOLS), c = txi + 16 * (q % MCOLS);
                    const int fl = t.cfl[c] & t.rfl[lr];
                    if (!(fl & AX_BORDER)) continue;
                    touched |= 1u << q;
                    float vb = cb[0], vg = cg[0], vr = cr[0];
#pragma unroll
                    for (int s = 1; s < MPX; s++) { vb = q == s ? cb[s] : vb; vg = q == s ? cg[s] : vg; vr = q == s ? cr[s] : vr; }
                    float r0 = clampf(__builtin_fmaf(f_b, af, vb * iaf), 0.f, 255.f);
                    float r1 = clampf(__builtin_fmaf(f_g, af, vg * iaf), 0.f, 255.f);
                    float r2 = clampf(__builtin_fmaf(f_r, af, vr * iaf), 0.f, 255.f);
                    if ((fl & (AX_TX | AX_UV)) == (AX_TX | AX_UV)) {
                        const float a = t.cya[c], ia = 1.0f - a, b = t.rya[lr], ib = 1.0f - b;
                        const float w00 = ia * ib, w10 = a * ib, w01 = ia * b, w11 = a * b;
                        float p0, p1, p2, al;
                        if (rgb) {
                            uint32_t u00, u10, u01, u11;
                            if (m.staged) {
                                const int cpo = off0 + (t.ry[lr] - m.g0.r_lo) * p0pitch + t.cy[c] * 4;
                                const uint32_t *q0 = (const uint32_t *)(smem + cpo);
                                const uint32_t *q1 = (const uint32_t *)(smem + cpo + p0pitch);
                                u00 = q0[0]; u10 = q0[1]; u01 = q1[0]; u11 = q1[1];
                            } else {
                                int xa = min(max(t.cy[c], 0), S0.w - 1), xb = min(max(t.cy[c] + 1, 0), S0.w - 1);
                                int ya = min(max(t.ry[lr], 0), S0.h - 1), yb = min(max(t.ry[lr] + 1, 0), S0.h - 1);
                                auto ld = [&](int xx, int yy) { return gld<uint32_t>(S0.ptr + (size_t)yy * S0.pitch + (size_t)xx * 4); };
                                u00 = ld(xa, ya); u10 = ld(xb, ya); u01 = ld(xa, yb); u11 = ld(xb, yb);
                            }
                            const float4 t00 = codes4(u00), t10 = codes4(u10), t01 = codes4(u01), t11 = codes4(u11);
                            const float s0 = cs_mix(w00, w10, w01, w11, t00.x, t10.x, t01.x, t11.x);
                            const float s1 = cs_mix(w00, w10, w01, w11, t00.y, t10.y, t01.y, t11.y);
                            const float s2 = cs_mix(w00, w10, w01, w11, t00.z, t10.z, t01.z, t11.z);
                            const float s3 = cs_mix(w00, w10, w01, w11, t00.w, t10.w, t01.w, t11.w);
                            const bool swz = !m.staged && Ly.swizzle;     // taps gathered from global memory keep the source order
                            p0 = swz ? s2 : s0; p1 = s1; p2 = swz ? s0 : s2;
                            al = s3 * (opacity * kInv255);
                        } else {
                            const float ca = t.cca[c], ica = 1.0f - ca, cbw = t.rca[lr], icb = 1.0f - cbw;
                            float fy, fu, fv;
                            if (m.staged) {
                                const int ya = off0 + (t.ry[lr] - m.g0.r_lo) * p0pitch + t.cy[c];
                                const int cao = off1 + (t.rc[lr] - m.g1.r_lo) * p1pitch + t.cc[c] * ctb;
                                if (planar) sample_y420p_lds_bytes(smem, ya, p0pitch, cao, voff, p1pitch, w00, w10, w01, w11,
                                                                   ica * icb, ca * icb, ica * cbw, ca * cbw, fy, fu, fv);
                                else sample_nv12_lds_bytes(smem, ya, p0pitch, cao, p1pitch, w00, w10, w01, w11,
                                                           ica * icb, ca * icb, ica * cbw, ca * cbw, fy, fu, fv);
                            } else {
                                sample_nv12_global(S0, S1, planar ? &S2 : nullptr, t.cy[c], t.ry[lr], t.cc[c], t.rc[lr], w00, w10, w01, w11,
                                                   ica * icb, ca * icb, ica * cbw, ca * cbw, fy, fu, fv);
                            }
                            const uint32_t w = yuv_to_bgra_word(csc, (int)to_code_raw(fy), (int)to_code_raw(fu), (int)to_code_raw(fv));
                            p0 = (float)(w & 255); p1 = (float)((w >> 8) & 255); p2 = (float)((w >> 16) & 255);
                            al = 1.0f * opacity;
                        }
                        const float ial = 1.f - al;
                        r0 = __builtin_fmaf(p0, al, r0 * ial);
                        r1 = __builtin_fmaf(p1, al, r1 * ial);
                        r2 = __builtin_fmaf(p2, al, r2 * ial);
                    }
                    r0 = mix_to_codef(r0); r1 = mix_to_codef(r1); r2 = mix_to_codef(r2);
#pragma unroll
                    for (int s = 0; s < MPX; s++) { cb[s] = q == s ? r0 : cb[s]; cg[s] = q == s ? r1 : cg[s]; cr[s] = q == s ? r2 : cr[s]; }
                }
            }
        }
        __syncthreads();
        l = ln;
    }

    if (active) {
#pragma unroll
        for (int r = 0; r < MROWS; r++) {
            if (yq + 16 * r >= T.H) continue;
            uint8_t *drow = D.ptr + (size_t)(yq + 16 * r) * D.pitch;
#pragma unroll
            for (int k = 0; k < MCOLS; k++) {
                const int q = r * MCOLS + k;
                const uint32_t a8 = ((touched >> q) & 1u) ? 0xFF000000u : (((orig_a[r] >> (8 * k)) & 255u) << 24);
                const uint32_t w = pack_codes(cb[q], cg[q], cr[q], a8);     // codes are integral floats in [0, 255] here
                if (xq + 16 * k < T.W) gst<uint32_t>(drow + (size_t)(xq + 16 * k) * 4, w);
            }
        }
    }
}

// ---------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------
static bool finite16m(const float *m) {
    for (int i = 0; i < 16; i++) if (!(m[i] - m[i] == 0.f)) return false;
    return true;
}
static bool aligned16m(const DPlane &p) { return (((uintptr_t)p.ptr) & 15) == 0 && (p.pitch & 15) == 0 && p.w * p.comps >= 16; }

struct MixDims { int p0pitch, p0rows, p1pitch, p1rows; };

// LDS rectangles one tile of this layer can touch, from the layer's scale factors
static MixDims mix_tile_dims(const DTick &T, const DLayer &L) {
    const float *U = L.u;
    double sxr = std::fabs((double)U[U_TEXTURE + 0] * (double)U[U_TRANSFORM + 0] * 2.0 / (double)T.W);
    double syr = std::fabs((double)U[U_TEXTURE + 5] * (double)U[U_TRANSFORM + 5] * 2.0 / (double)T.H);
    MixDims d{ 0, 0, 0, 0 };
    const bool rgb = L.kind == LK_BGRA_FROM_RGB, planar = L.kind == LK_BGRA_FROM_Y420P;
    const int bpt0 = rgb ? 4 : 1;
    int span0 = (int)std::ceil(MTW * sxr * L.src.pl[0].w) + 4;           // texels incl. tap 1 and rounding slack
    d.p0pitch = ((span0 * bpt0 + 15) / 16 + 3) * 16;                      // vectors + alignment + 2 pad vectors
    d.p0rows = (int)std::ceil(MTH * syr * L.src.pl[0].h) + 3;
    if (!rgb) {
        const int bpt1 = planar ? 1 : 2;
        int span1 = (int)std::ceil(MTW * sxr * L.src.pl[1].w) + 4;
        d.p1pitch = ((span1 * bpt1 + 15) / 16 + 3) * 16;
        d.p1rows = (int)std::ceil(MTH * syr * L.src.pl[1].h) + 3;
    }
    return d;
}
static size_t mix_lds(const MixDims &d, int n_layers, bool planar) {
    return sizeof(MixLayerTable) * (size_t)n_layers + 64 + (size_t)d.p0pitch * d.p0rows + (size_t)d.p1pitch * d.p1rows * (planar ? 2 : 1);
}

bool mix_layers_eligible(const DTick *ticks, const DLayer *layers, int n_ticks) {
    for (int i = 0; i < n_ticks; i++) {
        const DTick &T = ticks[i];
        if (T.n_layers < 1 || T.n_layers > MMAXL || T.clear_first != ticks[0].clear_first) return false;
        if (!aligned16m(T.dst.pl[0])) return false;
        for (int l = 0; l < T.n_layers; l++) {
            const DLayer &L = layers[T.first_layer + l];
            const bool rgb = L.kind == LK_BGRA_FROM_RGB, nv12 = L.kind == LK_BGRA_FROM_NV12, planar = L.kind == LK_BGRA_FROM_Y420P;
            if (!(rgb || nv12 || planar) || !(L.flags & LF_AXIS_ALIGNED)) return false;
            if (!finite16m(L.u + U_TRANSFORM) || !finite16m(L.u + U_TEXTURE) || !finite16m(L.u + U_BORDER)) return false;
            const int np = rgb ? 1 : nv12 ? 2 : 3;
            for (int p = 0; p < np; p++) if (!aligned16m(L.src.pl[p])) return false;
            if (mix_lds(mix_tile_dims(T, L), MMAXL, planar) > (size_t)LDS_BUDGET) return false;
        }
    }
    return true;
}

hipError_t launch_mix_layers(const DTick *ticks_host, const DLayer *layers_host, const DTick *ticks, const DLayer *layers,
                             int n_ticks, int maxW, int maxH, hipStream_t stream) {
    MixDims m{ 0, 0, 0, 0 };
    int max_layers = 1;
    bool planar = false;
    for (int i = 0; i < n_ticks; i++) {
        max_layers = std::max(max_layers, ticks_host[i].n_layers);
        for (int l = 0; l < ticks_host[i].n_layers; l++) {
            const DLayer &L = layers_host[ticks_host[i].first_layer + l];
            MixDims d = mix_tile_dims(ticks_host[i], L);
            m.p0pitch = std::max(m.p0pitch, d.p0pitch); m.p0rows = std::max(m.p0rows, d.p0rows);
            m.p1pitch = std::max(m.p1pitch, d.p1pitch); m.p1rows = std::max(m.p1rows, d.p1rows);
            planar = planar || L.kind == LK_BGRA_FROM_Y420P;
        }
    }
    size_t lds = mix_lds(m, max_layers, planar);
    if (lds > (size_t)LDS_BUDGET) {
        // per-layer maxima combined exceed the budget: shrink the row counts; rectangles that do not fit fall back to
        // unstaged taps inside the kernel
        const size_t fixed = sizeof(MixLayerTable) * (size_t)max_layers + 64;
        const size_t per_row = (size_t)m.p0pitch + (size_t)m.p1pitch * (planar ? 2 : 1);
        int rows = std::max(1, (int)((LDS_BUDGET - fixed) / per_row));
        m.p0rows = std::min(m.p0rows, rows); m.p1rows = std::min(m.p1rows, rows);
        lds = mix_lds(m, max_layers, planar);
    }
    int tiles_x = (maxW + MTW - 1) / MTW, tiles_y = (maxH + MTH - 1) / MTH;
    int per_xcd = (n_ticks * tiles_x * tiles_y + 7) / 8;
    dim3 grid((unsigned)(per_xcd * 8));
    if (ticks_host[0].clear_first)
        hipLaunchKernelGGL(tick_mix_layers_tiled<true>, grid, dim3(NTHREADS), lds, stream, ticks, layers, n_ticks, tiles_x, tiles_y,
                           m.p0pitch, m.p0rows, m.p1pitch, m.p1rows, max_layers);
    else
        hipLaunchKernelGGL(tick_mix_layers_tiled<false>, grid, dim3(NTHREADS), lds, stream, ticks, layers, n_ticks, tiles_x, tiles_y,
                           m.p0pitch, m.p0rows, m.p1pitch, m.p1rows, max_layers);
    return hipGetLastError();
}

}  // namespace chv

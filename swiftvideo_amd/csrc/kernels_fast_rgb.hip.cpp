// kernels_fast_rgb.hip.cpp — axis-aligned, LDS-tiled tick kernel for BGRA canvases whose
// layers are all BGRA/RGBA pictures (img_bgra_bgra_tx / img_rgba_bgra_tx; BASELINE
// configs 3 and 5: N full-size layers alpha-composited in one pass).
//
// A block owns a 64x32 tile of the canvas and keeps its pixels (8 per thread) in
// registers across all layers, so the canvas is written once and every layer is read
// once: the per-tick HBM traffic is the algorithmic minimum (layers + canvas).
//   phase 0  per-layer column/row tables of the reference's coordinate arithmetic
//            (same instruction sequence as the general kernel => same bits)
//   per layer: [store the prefetched source rectangle to LDS as it is (4-byte texels, edge
//            texels replicated) | barrier | issue the next layer's global loads | read 2x2
//            taps from LDS, convert (v_cvt_f32_ubyteN), blend, re-quantise | barrier]
// The tile stays in bytes on purpose: float4 texels made the LDS rectangle 4x larger (4 px per
// thread at 5 blocks per CU) and the kernel LDS- and latency-bound (profiles/r01_notes.md).
// Between layers the value is re-quantised (RTE through the float adder) exactly as the
// per-layer kernels do through their 8-bit canvas.
#include "tile_common.hip.h"

#include <algorithm>
#include <cmath>

#pragma clang fp contract(off)

namespace chv {

constexpr int RTW = 64;           // tile width  (output pixels): 16 threads x 4 px (columns txi + 16k, so that
                                  // lane-adjacent LDS reads hit adjacent texels: no bank conflicts)
constexpr int RTH = 32;           // tile height (output rows):   16 thread rows x RROWS
constexpr int RROWS = RTH / 16;   // rows per thread (ly + 16*r)
constexpr int RCOLS = 4;          // columns per thread
constexpr int RMAXL = 8;          // layers per tick this path accepts
#ifndef CHV_RGB_RNV
#define CHV_RGB_RNV 3
#endif
constexpr int RNV = CHV_RGB_RNV;  // prefetch registers (16-byte vectors) per thread

struct RgbLayerTable {
    int cp[RTW]; float ca[RTW]; int cfl[RTW];     // column: unclamped tap-0 texel, weight of tap 1, flags
    int rp[RTH]; float ra[RTH]; int rfl[RTH];     // row
    int csum[8];                                  // {min, max+1, -, -, any inside, all inside, -, -}
    int rsum[8];
};

CHV_DEV void axis_entry_x1(const float *__restrict__ U, int x, float sx, float sy, int w, int &ip, float &a, int &flags) {
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
    lin_axis_raw(u, w, ip, a);
}
CHV_DEV void axis_entry_y1(const float *__restrict__ U, int y, float sx, float sy, int h, int &ip, float &a, int &flags) {
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
    lin_axis_raw(v, h, ip, a);
}

// store conversion of a code-scale value kept as a float: RTE, saturated, NaN -> 0
CHV_DEV float to_codef(float v) {
    v = __builtin_rintf(v);
    return __builtin_fminf(__builtin_fmaxf(v, 0.0f), 255.0f);
}

#ifndef CHV_RGB_MINW
#define CHV_RGB_MINW 5
#endif
#ifndef CHV_RGB_ROWPAD
#define CHV_RGB_ROWPAD 3
#endif
template <bool CLEAR>
__global__ __launch_bounds__(NTHREADS, CHV_RGB_MINW) void tick_rgb_layers_tiled(const DTick *__restrict__ ticks,
                                                                   const DLayer *__restrict__ layers,
                                                                   int n_ticks, int tiles_x, int tiles_y,
                                                                   int tpitch, int trows, int max_layers) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    RgbLayerTable *tabs = (RgbLayerTable *)smem;                 // [max_layers] (the launch's deepest tick)
    int *scratch = (int *)(smem + sizeof(RgbLayerTable) * max_layers);  // sink for summaries of absent layers
    const int tbase = (int)(sizeof(RgbLayerTable) * max_layers) + 64; // [trows][tpitch] source texels (bytes)

    // XCD-aware numbering: block b runs on XCD b % 8; give every XCD one contiguous range of the
    // launch's tiles (whole frames when there are >= 8 ticks) so that halos are shared through its L2
    const int bid = blockIdx.x;
    const int xcd = bid & 7, slot = bid >> 3;
    const int tiles = tiles_x * tiles_y;
    const int total = tiles * n_ticks, per_xcd = (total + 7) >> 3;
    const int index = xcd * per_xcd + slot;
    if (slot >= per_xcd || index >= total) return;
    const int tick = index / tiles;
    const int tile = index - tick * tiles;
    const DTick &T = ticks[tick];
    const int x0 = (tile % tiles_x) * RTW, y0 = (tile / tiles_x) * RTH;
    if (x0 >= T.W || y0 >= T.H) return;
    const DLayer *L = layers + T.first_layer;
    const int nl = T.n_layers;
    const DPlane &D = T.dst.pl[0];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const float sx = (float)T.W, sy = (float)T.H;

    // ---- phase 0: tables of every layer ---------------------------------------------------
    for (int l = wave; l < nl; l += 4) {                         // one wave = the 64 columns of one layer
        const DPlane &S = L[l].src.pl[0];
        int x = x0 + lane, ip, fl; float a;
        axis_entry_x1(L[l].u, min(x, T.W - 1), sx, sy, S.w, ip, a, fl);
        group_summary(tabs[l].csum, 6, x < T.W, fl, ip, ip);
        if (x >= T.W) fl = AX_ALL;
        tabs[l].cp[lane] = ip; tabs[l].ca[lane] = a; tabs[l].cfl[lane] = fl;
    }
    constexpr int LPW = 64 / RTH;                                // layers whose rows one wave covers
    constexpr int RSH = RTH == 32 ? 5 : 4;
    static_assert(RTH == 16 || RTH == 32, "row tables: 16 or 32 rows per tile");
    for (int l4 = wave * LPW; l4 < nl; l4 += 4 * LPW) {          // one wave = the RTH rows of LPW layers
        int l = l4 + (lane >> RSH), j = lane & (RTH - 1);
        int lc = min(l, nl - 1);
        const DPlane &S = L[lc].src.pl[0];
        int y = y0 + j, ip, fl; float a;
        axis_entry_y1(L[lc].u, min(y, T.H - 1), sx, sy, S.h, ip, a, fl);
        group_summary(l < nl ? tabs[l].rsum : scratch, RSH, l < nl && y < T.H, fl, ip, ip);
        if (y >= T.H) fl = AX_ALL;
        if (l < nl) { tabs[l].rp[j] = ip; tabs[l].ra[j] = a; tabs[l].rfl[j] = fl; }
    }
    __syncthreads();

    // staging geometry of layer l's source rectangle for this tile
    auto layer_geom = [&](int l, StageGeom &g, int &col0) -> bool {
        const RgbLayerTable &t = tabs[l];
        const DPlane &S = L[l].src.pl[0];
        if (!(t.csum[1] > t.csum[0] && t.rsum[1] > t.rsum[0])) return false;
        const int lo = t.csum[0], hi = t.csum[1];
        col0 = max(lo, 0) & ~3;                                  // 4 texels per 16-byte vector
        const int nvec = (min(hi, S.w - 1) - col0) / 4 + 1;
        g.r_lo = t.rsum[0]; g.rows = t.rsum[1] - t.rsum[0] + 1; g.b0 = col0 * 4; g.nvec = nvec;
        g.edge = lo < 0 || hi >= S.w || t.rsum[0] < 0 || t.rsum[1] >= S.h - 1 + (int)(col0 + nvec * 4 <= S.w);
        stage_slots_init(g);
        return (nvec + 2) * 16 <= tpitch && g.rows <= trows && stage_slots(g) <= RNV * NTHREADS;
    };

    // ---- canvas pixels of this thread, as float code values -----------------------------------
    const int txi = tid & 15, ly = tid >> 4;
    const int xq = x0 + txi, yq = y0 + ly;              // this thread's pixels: (xq + 16*k, yq + 16*r)
    const bool active = xq < T.W && yq < T.H;
    float cb[RROWS][RCOLS], cg[RROWS][RCOLS], cr[RROWS][RCOLS];
    uint32_t orig_a[RROWS];                             // the RCOLS original alpha bytes of a row, packed
    unsigned touched = CLEAR ? ~0u : 0u;                // bit r*4+k; untouched pixels keep their original alpha byte
#pragma unroll
    for (int r = 0; r < RROWS; r++)
#pragma unroll
        for (int k = 0; k < RCOLS; k++) { cb[r][k] = 0.f; cg[r][k] = 0.f; cr[r][k] = 0.f; }   // img_clear_bgra: (0,0,0,1)
    if (!CLEAR && active) {
#pragma unroll
        for (int r = 0; r < RROWS; r++) {
            orig_a[r] = 0;
            if (yq + 16 * r >= T.H) continue;
            const uint8_t *drow = D.ptr + (size_t)(yq + 16 * r) * D.pitch;
#pragma unroll
            for (int k = 0; k < RCOLS; k++) {
                uint32_t cur = (xq + 16 * k < T.W) ? gld<uint32_t>(drow + (size_t)(xq + 16 * k) * 4) : 0;
                cb[r][k] = (float)(cur & 255); cg[r][k] = (float)((cur >> 8) & 255); cr[r][k] = (float)((cur >> 16) & 255);
                orig_a[r] |= (cur >> 24) << (8 * k);
            }
        }
    }

    // layers whose border quad cannot touch this tile are skipped (block-uniform test against the
    // host-computed bounding box); `next_hit` walks the remaining ones in z order
    auto next_hit = [&](int l) {
        for (; l < nl; l++) {
            const int *bb = L[l].bbox;
            if (!(x0 + RTW <= bb[0] || x0 >= bb[2] || y0 + RTH <= bb[1] || y0 >= bb[3])) break;
        }
        return l;
    };
    uint4 regs[RNV];
    StageGeom g, ng;
    int col0 = 0, ncol0 = 0;
    int l = next_hit(0);
    bool staged = l < nl && layer_geom(l, g, col0);
    if (staged) stage_load(regs, L[l].src.pl[0], g, tid);

    while (l < nl) {
        const DLayer &Ly = L[l];
        const DPlane &S = Ly.src.pl[0];
        const RgbLayerTable &t = tabs[l];
        touch_regs(regs);                 // the wait for the prefetch, on every path (see touch_regs)
        if (staged) stage_store<4>(regs, smem + tbase, tpitch, S, g, tid, 0, Ly.swizzle != 0);   // RGBA -> BGRA on the way
        __syncthreads();
        const int ln = next_hit(l + 1);
        bool nstaged = false;
        if (ln < nl) {
            nstaged = layer_geom(ln, ng, ncol0);
            if (nstaged) stage_load(regs, L[ln].src.pl[0], ng, tid);
        }
        if (active) {
            const float *U = Ly.u;
            const float opacity = U[U_OPACITY];
            const bool nofill = (Ly.flags & LF_NO_FILL) != 0;
            // opacity in [0,1] and no fill: every blend is a convex combination of code values, so
            // neither the clamp of the fill step nor the saturation of the store can trigger; with
            // every pixel of the tile inside the picture the loop is branch-free
            const bool fast = staged && t.csum[5] && t.rsum[5] && nofill && opacity >= 0.f && opacity <= 1.f;
            const float ka = opacity * kInv255;
            // column entries are re-read from the LDS tables per pixel: the LDS pipe is mostly idle in this
            // kernel and holding them across the rows costs 12 VGPRs (spills at 96)
            const int coloff = (4 - col0) * 4;
            if (fast) {
#pragma unroll
                for (int r = 0; r < RROWS; r++) {
                    const int lr = ly + 16 * r;
                    const float b = t.ra[lr], ib = 1.0f - b;
                    const int rowoff = tbase + (t.rp[lr] - g.r_lo) * tpitch;
#pragma unroll
                    for (int k = 0; k < RCOLS; k++) {
                        const int c = txi + 16 * k;
                        const float a = t.ca[c], ia = 1.0f - a;
                        const int cpo = t.cp[c] * 4 + coloff;
                        const float w00 = ia * ib, w10 = a * ib, w01 = ia * b, w11 = a * b;
                        const uint32_t *p0 = (const uint32_t *)(smem + rowoff + cpo);
                        const uint32_t *p1 = (const uint32_t *)(smem + rowoff + tpitch + cpo);
                        const float4 t00 = codes4(p0[0]), t10 = codes4(p0[1]), t01 = codes4(p1[0]), t11 = codes4(p1[1]);
                        const float q0 = cs_mix(w00, w10, w01, w11, t00.x, t10.x, t01.x, t11.x);
                        const float q1 = cs_mix(w00, w10, w01, w11, t00.y, t10.y, t01.y, t11.y);
                        const float q2 = cs_mix(w00, w10, w01, w11, t00.z, t10.z, t01.z, t11.z);
                        const float q3 = cs_mix(w00, w10, w01, w11, t00.w, t10.w, t01.w, t11.w);
                        const float al = q3 * ka, ial = 1.f - al;
                        cb[r][k] = code_rintf(__builtin_fmaf(q0, al, cb[r][k] * ial));     // staged texels are BGRA whatever the source order
                        cg[r][k] = code_rintf(__builtin_fmaf(q1, al, cg[r][k] * ial));
                        cr[r][k] = code_rintf(__builtin_fmaf(q2, al, cr[r][k] * ial));
                    }
                }
                touched = ~0u;
            } else {
                const float af = opacity * U[U_FILL + 3], iaf = 1.f - af;
                const float f_b = U[U_FILL + 2] * 255.0f, f_g = U[U_FILL + 1] * 255.0f, f_r = U[U_FILL + 0] * 255.0f;
#pragma unroll
                for (int r = 0; r < RROWS; r++) {
                    const int lr = ly + 16 * r;
                    const float b = t.ra[lr], ib = 1.0f - b;
                    const int rowoff = tbase + (t.rp[lr] - g.r_lo) * tpitch;
                    const int rfl = t.rfl[lr];
#pragma unroll
                    for (int k = 0; k < RCOLS; k++) {
                        const int c = txi + 16 * k;
                        const int fl = t.cfl[c] & rfl;
                        if (!(fl & AX_BORDER)) continue;
                        touched |= 1u << (r * 4 + k);
                        const bool in_pic = (fl & (AX_TX | AX_UV)) == (AX_TX | AX_UV);
                        float r0 = clampf(__builtin_fmaf(f_b, af, cb[r][k] * iaf), 0.f, 255.f);
                        float r1 = clampf(__builtin_fmaf(f_g, af, cg[r][k] * iaf), 0.f, 255.f);
                        float r2 = clampf(__builtin_fmaf(f_r, af, cr[r][k] * iaf), 0.f, 255.f);
                        if (in_pic) {
                            const float a = t.ca[c], ia = 1.0f - a;
                            const int cpo = t.cp[c] * 4 + coloff;
                            const float w00 = ia * ib, w10 = a * ib, w01 = ia * b, w11 = a * b;
                            uint32_t u00, u10, u01, u11;
                            if (staged) {
                                const uint32_t *p0 = (const uint32_t *)(smem + rowoff + cpo);
                                const uint32_t *p1 = (const uint32_t *)(smem + rowoff + tpitch + cpo);
                                u00 = p0[0]; u10 = p0[1]; u01 = p1[0]; u11 = p1[1];
                            } else {
                                int xa = min(max(t.cp[c], 0), S.w - 1), xb = min(max(t.cp[c] + 1, 0), S.w - 1);
                                int ya = min(max(t.rp[lr], 0), S.h - 1), yb = min(max(t.rp[lr] + 1, 0), S.h - 1);
                                auto ld = [&](int xx, int yy) { return gld<uint32_t>(S.ptr + (size_t)yy * S.pitch + (size_t)xx * 4); };
                                u00 = ld(xa, ya); u10 = ld(xb, ya); u01 = ld(xa, yb); u11 = ld(xb, yb);
                            }
                            const float4 t00 = codes4(u00), t10 = codes4(u10), t01 = codes4(u01), t11 = codes4(u11);
                            const float q0 = cs_mix(w00, w10, w01, w11, t00.x, t10.x, t01.x, t11.x);
                            const float q1 = cs_mix(w00, w10, w01, w11, t00.y, t10.y, t01.y, t11.y);
                            const float q2 = cs_mix(w00, w10, w01, w11, t00.z, t10.z, t01.z, t11.z);
                            const float q3 = cs_mix(w00, w10, w01, w11, t00.w, t10.w, t01.w, t11.w);
                            const bool swz = !staged && Ly.swizzle;       // taps gathered from global memory keep the source order
                            const float pb = swz ? q2 : q0, pr = swz ? q0 : q2;
                            const float al = q3 * ka, ial = 1.f - al;
                            r0 = __builtin_fmaf(pb, al, r0 * ial);
                            r1 = __builtin_fmaf(q1, al, r1 * ial);
                            r2 = __builtin_fmaf(pr, al, r2 * ial);
                        }
                        cb[r][k] = to_codef(r0); cg[r][k] = to_codef(r1); cr[r][k] = to_codef(r2);
                    }
                }
            }
        }
        __syncthreads();
        staged = nstaged; g = ng; col0 = ncol0; l = ln;
    }

    if (active) {
#pragma unroll
        for (int r = 0; r < RROWS; r++) {
            if (yq + 16 * r >= T.H) continue;
            uint8_t *drow = D.ptr + (size_t)(yq + 16 * r) * D.pitch;
#pragma unroll
            for (int k = 0; k < RCOLS; k++) {
                const uint32_t a8 = ((touched >> (r * 4 + k)) & 1u) ? 0xFF000000u : (((orig_a[r] >> (8 * k)) & 255u) << 24);
                const uint32_t w = pack_codes(cb[r][k], cg[r][k], cr[r][k], a8);     // codes are integral floats in [0, 255] here
                if (xq + 16 * k < T.W) gst<uint32_t>(drow + (size_t)(xq + 16 * k) * 4, w);
            }
        }
    }
}

// ---------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------
static bool finite16r(const float *m) {
    for (int i = 0; i < 16; i++) if (!(m[i] - m[i] == 0.f)) return false;
    return true;
}
static bool aligned16r(const DPlane &p) { return (((uintptr_t)p.ptr) & 15) == 0 && (p.pitch & 15) == 0 && p.w * p.comps >= 16; }

static void rgb_tile_dims(const DTick &T, const DLayer &L, int *pitch, int *rows) {
    const float *U = L.u;
    double sxr = std::fabs((double)U[U_TEXTURE + 0] * (double)U[U_TRANSFORM + 0] * 2.0 / (double)T.W);
    double syr = std::fabs((double)U[U_TEXTURE + 5] * (double)U[U_TRANSFORM + 5] * 2.0 / (double)T.H);
    int span = (int)std::ceil(RTW * sxr * L.src.pl[0].w) + 4;
    *pitch = ((span + 3) / 4 + 3) * 16;                 // 4-byte texels, 4 per vector, alignment + 2 pad vectors
    // rows a tile's taps span: <= ceil((RTH-1)*scale) + 2 (tap 1 of the last row) <= ceil(RTH*scale) + 2
    *rows = (int)std::ceil(RTH * syr * L.src.pl[0].h) + CHV_RGB_ROWPAD;
}

bool rgb_layers_eligible(const DTick *ticks, const DLayer *layers, int n_ticks) {
    for (int i = 0; i < n_ticks; i++) {
        const DTick &T = ticks[i];
        if (T.n_layers < 1 || T.n_layers > RMAXL || T.clear_first != ticks[0].clear_first) return false;
        if (!aligned16r(T.dst.pl[0])) return false;
        for (int l = 0; l < T.n_layers; l++) {
            const DLayer &L = layers[T.first_layer + l];
            if (L.kind != LK_BGRA_FROM_RGB || !(L.flags & LF_AXIS_ALIGNED)) return false;
            if (!finite16r(L.u + U_TRANSFORM) || !finite16r(L.u + U_TEXTURE) || !finite16r(L.u + U_BORDER)) return false;
            if (!aligned16r(L.src.pl[0])) return false;
            int pitch, rows;
            rgb_tile_dims(T, L, &pitch, &rows);
            if (sizeof(RgbLayerTable) * RMAXL + 64 + (size_t)pitch * rows > (size_t)LDS_BUDGET) return false;
        }
    }
    return true;
}

hipError_t launch_rgb_layers(const DTick *ticks_host, const DLayer *layers_host, const DTick *ticks, const DLayer *layers,
                             int n_ticks, int maxW, int maxH, hipStream_t stream) {
    int pitch = 0, rows = 0, max_layers = 1;
    for (int i = 0; i < n_ticks; i++) {
        max_layers = std::max(max_layers, ticks_host[i].n_layers);
        for (int l = 0; l < ticks_host[i].n_layers; l++) {
            int p, r;
            rgb_tile_dims(ticks_host[i], layers_host[ticks_host[i].first_layer + l], &p, &r);
            pitch = std::max(pitch, p); rows = std::max(rows, r);
        }
    }
    const size_t tab_bytes = sizeof(RgbLayerTable) * max_layers + 64;
    size_t lds = tab_bytes + (size_t)pitch * rows;
    if (lds > (size_t)LDS_BUDGET) {
        rows = std::max(1, (int)((LDS_BUDGET - tab_bytes) / pitch));
        lds = tab_bytes + (size_t)pitch * rows;
    }
    int tiles_x = (maxW + RTW - 1) / RTW, tiles_y = (maxH + RTH - 1) / RTH;
    int per_xcd = (n_ticks * tiles_x * tiles_y + 7) / 8;
    dim3 grid((unsigned)(per_xcd * 8));
    if (ticks_host[0].clear_first)
        hipLaunchKernelGGL(tick_rgb_layers_tiled<true>, grid, dim3(NTHREADS), lds, stream, ticks, layers, n_ticks, tiles_x, tiles_y, pitch, rows, max_layers);
    else
        hipLaunchKernelGGL(tick_rgb_layers_tiled<false>, grid, dim3(NTHREADS), lds, stream, ticks, layers, n_ticks, tiles_x, tiles_y, pitch, rows, max_layers);
    return hipGetLastError();
}

}  // namespace chv

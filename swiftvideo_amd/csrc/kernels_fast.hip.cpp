// kernels_fast.hip.cpp — axis-aligned, LDS-tiled tick kernels for gfx950.
//
// When a layer's matrices have no rotation/shear (every BASELINE configuration),
// tx.x / border.x / uv.x depend on the pixel column only and tx.y / border.y /
// uv.y on the row only.  A block then
//   0. evaluates the reference's coordinate arithmetic once per tile column and
//      once per tile row (same instruction sequence as the general kernel, so the
//      bits are identical) and parks tap offsets + weights in LDS tables,
//   1. stages the source rectangle the tile's taps touch into LDS with 16-byte
//      coalesced global loads (each source byte leaves HBM once); luma and chroma stay
//      bytes there (code-scale arithmetic: a tap is one v_cvt_f32_ubyte) — NV12: a tile of
//      (u, v) byte pairs, y420p: a U tile and a V tile,
//   2. lets every thread produce PXT horizontally adjacent pixels x RPT rows from LDS
//      and store them as one 8- or 16-byte BGRA write per row.
// Blocks are numbered so that all tiles of one frame run on one XCD (block b runs
// on XCD b % 8): tile halos are shared through that XCD's L2.
//
// Where the time goes on gfx950 (profiles/r01_notes.md): neither the VALU (~70 % busy) nor
// HBM is saturated; arithmetic, LDS traffic and the read / write phases overlap only partly
// across the 6 resident blocks of a CU.  The inner loop avoids v_pk_* (slower than two
// scalar ops, tools/ubench_valu.cpp).
//
// Every path here produces exactly the bytes of kernels_general.hip.cpp; the host
// picks a path per batch (select_fast_path) and falls back to the general kernel.
#include "tile_common.hip.h"
#include "switches.h"
#include <vector>

#include <algorithm>
#include <cmath>
#include <cstdlib>

#pragma clang fp contract(off)

namespace chv {

enum FastPath : int { FP_NONE = -1, FP_NV12_BGRA_TILED = 0, FP_Y420P_BGRA_TILED = 1, FP_WAVE_LAYERS = 2, FP_WAVE_NV12 = 3, FP_WAVE_Y420P = 4, FP_STREAM = 5, FP_CLEAR_BGRA = 6, FP_STREAM_NV12 = 7, FP_STREAM_Y420P = 8, FP_COUNT };

// kernels_wave.hip.cpp / kernels_wave_yuv.hip.cpp
bool wave_layers_eligible(int target_format, const DTick *ticks, const DLayer *layers, int n_ticks, bool tail = false);
// kernels_stream.hip.cpp
bool bgra_stream_eligible(const DTick *ticks, const DLayer *layers, int n_ticks);
hipError_t launch_bgra_stream(const DTick *ticks_host, const DLayer *layers_host, const DTick *ticks, const DLayer *layers, int n_ticks, int maxW, int maxH, hipStream_t stream);
bool yuv_stream_eligible(int tf, const DTick *ticks, const DLayer *layers, int n_ticks, bool transient);
hipError_t launch_yuv_stream(int tf, const DTick *ticks_host, const DLayer *layers_host, const DTick *ticks, const DLayer *layers, int n_ticks, int maxW, int maxH, hipStream_t stream);
hipError_t launch_wave_layers(int target_format, const DTick *ticks_host, const DLayer *layers_host, const DTick *ticks, const DLayer *layers,
                              int n_ticks, int maxW, int maxH, hipStream_t stream);

#ifndef CHV_TW
#define CHV_TW 128
#endif
constexpr int TW = CHV_TW;       // tile width  (output pixels), a multiple of 64
#ifndef CHV_PXT
#define CHV_PXT 2
#endif
constexpr int PXT = CHV_PXT;     // horizontally adjacent pixels per thread (4: one 16-byte store, 2: one 8-byte store)
constexpr int TXT = TW / PXT;    // threads across a tile row
constexpr int TYT = NTHREADS / TXT;   // tile rows covered per pass
// Tile height (output rows) is a kernel template parameter THV: 32 rows for launches that fill the chip anyway (more pixels
// per barrier interval: -6 % on cfg2), 16 rows for small launches such as a single mixer tick (twice the blocks: 9.6 instead
// of 12.3 us for one 720p tick).  RPT = THV / TYT rows per thread.
constexpr int TH_SMALL = 16, TH_LARGE = 32;
// NTHREADS = 256: 32 x 8 threads, each 4 px x 2 rows per tile

#ifndef CHV_KT
#define CHV_KT 4
#endif
constexpr int KT = CHV_KT;             // tiles per block: a vertical strip of KT tiles shares its column tables

// Per-strip tables, structure-of-arrays so that lane-adjacent reads are conflict free.
// Tap positions are the UNCLAMPED i0 = floor(u - 0.5) of the linear filter (tap 1 is
// i0 + 1): the staged tile replicates the edge texels, so CLAMP_TO_EDGE costs nothing
// in the inner loop and tap 1 always sits right next to tap 0.
template <int TH>
struct TileTablesT {
    int cy[TW]; float cya[TW];     // luma column:   tap-0 position, weight of tap 1
    int cc[TW]; float cca[TW];     // chroma column
    int cfl[TW];
    int ry[KT * TH]; float rya[KT * TH];     // luma rows of the KT tiles
    int rc[KT * TH]; float rca[KT * TH];     // chroma rows
    int rfl[KT * TH];
    // {min luma, max luma + 1, min chroma, max chroma + 1, any entry fully inside,
    //  every in-canvas entry fully inside, -, -}
    int csum[TW / 64][8];          // per column wave
    int rsum[KT][8];               // per tile of the strip
};

// axis_entry_x / axis_entry_y, blend_bgra_general and the LDS / global YUV samplers live in tile_common.hip.h
// (shared with the wave kernels)

// ---------------------------------------------------------------------------
// FP_NV12_BGRA_TILED / FP_Y420P_BGRA_TILED: one LK_BGRA_FROM_{NV12,Y420P} layer per tick, axis aligned.
// A block walks a vertical strip of KT tiles: column tables once, row tables for all KT
// tiles at once, then per tile  [LDS write of the prefetched rectangle | barrier | issue
// the next tile's global loads | compute + store | barrier].
// CLEAR: canvas starts as img_clear_bgra's value instead of being read.
// ---------------------------------------------------------------------------
// NYV / NCV: prefetch registers (16-byte vectors) per thread for the luma / chroma rectangle of the next tile (planar
// sources: NCV each for the U and the V plane).  The host picks (2, 1) when the worst-case rectangle fits, else (3, 2);
// slots beyond the registers' capacity go through stage_tail.  THV: tile height, 16 or 32 rows (launch_tick_fast).
// Occupancy: the SIMD's issue rate keeps rising with resident waves (tools/ubench_issue.cpp), so the NV12 variants that
// matter (small prefetch, and the 32-row bench kernel) are held to 80 VGPRs = 6 waves per SIMD; planar sources need
// more registers (three planes in flight) and run 5 or 4 waves — an 80-VGPR cap there spills and is 65 % slower.
#ifndef CHV_MINW
#define CHV_MINW ((!PLANAR && (NYV == 2 || THV == 32)) ? 6 : (NYV == 3 && PLANAR) ? 4 : 5)
#endif
// ONE: a launch of one tick whose descriptors are kernel arguments (tick_yuv_bgra_tiled_one below; see tick_bgra_stream_one)
template <bool CLEAR, bool PLANAR, int NYV, int NCV, int THV, bool ONE>
CHV_DEV void tiled_body(const DTick *__restrict__ ticks, const DLayer *__restrict__ layers,
                        int n_ticks, int tiles_x, int strips_y, int kt, int ypitch, int yrows, int cpitch, int crows) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    constexpr int TH = THV, RPT = TH / TYT;
    using TileTables = TileTablesT<THV>;
    TileTables &tb = *(TileTables *)smem;
    const int ybase = (int)sizeof(TileTables);          // [yrows][ypitch] luma bytes
    const int cbase = ybase + yrows * ypitch;           // [crows][cpitch] chroma bytes: (u, v) pairs (NV12) / U tile, then V tile (planar)

    // XCD-aware numbering: consecutive blocks go to consecutive XCDs, so give each
    // XCD whole frames: block -> (xcd, slot) -> (tick = group*8 + xcd, strip)
    const int bid = blockIdx.x;
    const int xcd = bid & 7, slot = bid >> 3;
    const int strips = tiles_x * strips_y;
    const int total = strips * n_ticks, per_xcd = (total + 7) >> 3;
    const int index = xcd * per_xcd + slot;          // one contiguous range of strips per XCD
    if (slot >= per_xcd || index >= total) return;
    const int tick = index / strips;
    const int strip = index - tick * strips;
    const DTick &T = ticks[ONE ? 0 : tick];
    const int x0 = (strip % tiles_x) * TW, ys0 = (strip / tiles_x) * (kt * TH);   // kt <= KT tiles per strip (host's choice)
    if (x0 >= T.W || ys0 >= T.H) return;
    const int ntiles = min(kt, (T.H - ys0 + TH - 1) / TH);
    const DLayer &L = layers[ONE ? 0 : T.first_layer];
    const float *U = L.u;
    const DPlane &SY = L.src.pl[0];
    const DPlane &SC = L.src.pl[1];
    const DPlane &SV = L.src.pl[PLANAR ? 2 : 1];
    constexpr int CVEC = PLANAR ? 16 : 8;           // chroma texels per 16-byte source vector
    // Chroma stays bytes in LDS and is converted per tap: 8 more v_cvt_f32_ubyte per pixel than with float pairs staged once
    // per texel, but a 4x smaller LDS image, plain 16-byte staging writes and far fewer bank conflicts (NV12: -2 % on cfg2;
    // y420p: 0.554 -> see profiles/r01_notes.md).  NV12: one tile of (u, v) byte pairs; y420p: a U tile and a V tile.
    constexpr int CTB = PLANAR ? 1 : 2;             // LDS bytes per chroma texel (planar: a U tile and a V tile)
    const int voff = crows * cpitch;                // planar: the V tile follows the U tile
    const DPlane &D = T.dst.pl[0];
    const int tid = threadIdx.x;
    const float sx = (float)T.W, sy = (float)T.H;

    // ---- phase 0: column entries of the strip, row entries of all its tiles ------------
    if (tid < TW) {
        int x = x0 + tid;
        int iy, ic, fl; float ay, ac;
        axis_entry_x(U, min(x, T.W - 1), sx, sy, SY.w, SC.w, iy, ay, ic, ac, fl);
        group_summary(tb.csum[tid >> 6], 6, x < T.W, fl, iy, ic);   // TW / 64 column waves
        if (x >= T.W) fl = AX_ALL;   // past the canvas edge: never stored; copy of the last column
        tb.cy[tid] = iy; tb.cya[tid] = ay; tb.cc[tid] = ic; tb.cca[tid] = ac; tb.cfl[tid] = fl;
    } else if (tid < TW + kt * TH) {
        int j = tid - TW, y = ys0 + j;
        int iy, ic, fl; float ay, ac;
        axis_entry_y(U, min(y, T.H - 1), sx, sy, SY.h, SC.h, iy, ay, ic, ac, fl);
        constexpr int TH_SHIFT = TH == 8 ? 3 : TH == 16 ? 4 : 5;
        static_assert((1 << TH_SHIFT) == TH, "tile height: 8, 16 or 32 rows");
        group_summary(tb.rsum[j >> TH_SHIFT], TH_SHIFT, y < T.H, fl, iy, ic);
        if (y >= T.H) fl = AX_ALL;
        tb.ry[j] = iy; tb.rya[j] = ay; tb.rc[j] = ic; tb.rca[j] = ac; tb.rfl[j] = fl;
    }
    __syncthreads();

    // column geometry of the staged rectangle (the same for every tile of the strip)
    int ylo = tb.csum[0][0], yhi = tb.csum[0][1], clo = tb.csum[0][2], chi = tb.csum[0][3];
    bool cols_inside = tb.csum[0][5] != 0;
#pragma unroll
    for (int w = 1; w < TW / 64; w++) {
        ylo = min(ylo, tb.csum[w][0]); yhi = max(yhi, tb.csum[w][1]);
        clo = min(clo, tb.csum[w][2]); chi = max(chi, tb.csum[w][3]);
        cols_inside = cols_inside && tb.csum[w][5];
    }
    const bool cols_any = yhi > ylo;
    const int ycol0 = max(ylo, 0) & ~15;                       // luma: byte == texel, 16 per vector
    const int ccol0 = max(clo, 0) & ~(CVEC - 1);               // chroma: CVEC texels per 16-byte vector
    const int ynv = (min(yhi, SY.w - 1) - ycol0) / 16 + 1;
    const int cnv = (min(chi, SC.w - 1) - ccol0) / CVEC + 1;
    const bool cols_fit = (ynv + 2) * 16 <= ypitch && (cnv + 2) * CVEC * CTB <= cpitch;

    // per-tile staging geometry from the tile's row summary
    auto tile_geom = [&](int j, StageGeom &gy, StageGeom &gc) -> bool {
        const int *rs = tb.rsum[j];
        if (!(cols_any && rs[1] > rs[0])) return false;
        gy.r_lo = rs[0]; gy.rows = rs[1] - rs[0] + 1; gy.b0 = ycol0; gy.nvec = ynv;
        gc.r_lo = rs[2]; gc.rows = rs[3] - rs[2] + 1; gc.b0 = ccol0 * (PLANAR ? 1 : 2); gc.nvec = cnv;
        // interior rectangles: every tap and every 16-byte vector lies inside the planes
        gy.edge = ylo < 0 || yhi >= SY.w || rs[0] < 0 || rs[1] >= SY.h - 1 + (int)(ycol0 + ynv * 16 <= SY.w);
        gc.edge = clo < 0 || chi >= SC.w || rs[2] < 0 || rs[3] >= SC.h - 1 + (int)(ccol0 + cnv * CVEC <= SC.w);
        stage_slots_init(gy);
        stage_slots_init(gc);
        return cols_fit && gy.rows <= yrows && gc.rows <= crows &&
               // byte tiles take what the prefetch registers cannot hold through stage_tail (slot numbers < 1024)
               stage_slots(gy) <= 1024 && stage_slots(gc) <= 1024;
    };

    uint4 yregs[NYV], cregs[NCV], vregs[PLANAR ? NCV : 1];
    StageGeom gy, gc, ngy, ngc;
    bool staged = tile_geom(0, gy, gc);
    if (staged) {
        stage_load(yregs, SY, gy, tid); stage_load(cregs, SC, gc, tid);
        if (PLANAR) stage_load(vregs, SV, gc, tid);
    }

    // column entries of this thread's four pixels (shared by all its rows); LDS byte
    // offsets inside a staged tile row
    const int txi = tid % TXT, tyi = tid / TXT;
    const int xq = x0 + txi * PXT;
    const bool full4 = xq + PXT - 1 < T.W;
    int cyo[PXT], cco[PXT];
    float cya[PXT], icya[PXT], cca[PXT], icca[PXT];
#pragma unroll
    for (int k = 0; k < PXT; k++) {
        int c = txi * PXT + k;
        // (times 2^24: the fast rows feed their taps to v_fma_mix_f32 as binary16 denormals — tap_h, pixel_math.hip.h; the slow
        // per-pixel path below re-reads the tables)
        cya[k] = tb.cya[c]; icya[k] = (1.0f - cya[k]) * kTapScale; cya[k] *= kTapScale;
        cca[k] = tb.cca[c]; icca[k] = (1.0f - cca[k]) * (PLANAR ? kTapScale : kChromaTapScale); cca[k] *= (PLANAR ? kTapScale : kChromaTapScale);
        cyo[k] = tb.cy[c] - ycol0 + 16; cco[k] = (tb.cc[c] - ccol0 + CVEC) * CTB;
    }
    const CscFolded csc = csc_fold(kCsc[L.csc & 3]);
    // the fast rows use the matrix with its red and blue offsets absorbed into the conversion biases (pixel_math.hip.h); a layer whose matrix
    // has no such biases (BT.601 full range: the host sends it to the strip kernel, select_single_purpose) takes the per-pixel rows here
    const bool absorbable = csc_absorbable(L.csc);
    const CscAbsorbed csca = csc_fold_absorbed(L.csc);
    const bool opaque = (L.flags & LF_OPAQUE) != 0;

    for (int j = 0; j < ntiles; j++) {
#ifndef CHV_TILED_PRIO
#define CHV_TILED_PRIO 1
#endif
        // issue priority for the wait-and-write phase and the next tile's loads, back to 0 for the pixel rows (cfg2 -1.5 %; from
        // non-volatile asm with a token operand — the builtin is a side effect hipcc orders everything around)
        int ptok = j;
        if (CHV_TILED_PRIO) asm("s_setprio 3" : "+s"(ptok));
        // ---- phase 1: the prefetched rectangle of tile j goes to LDS ------------------------
        touch_regs(yregs); touch_regs(cregs);          // the wait for the prefetch, on every path (see touch_regs)
        if (PLANAR) touch_regs(vregs);
        if (staged) {
            stage_store<1>(yregs, smem + ybase, ypitch, SY, gy, tid);
            stage_store<(PLANAR ? 1 : 2)>(cregs, smem + cbase, cpitch, SC, gc, tid);
            if constexpr (PLANAR) stage_store<1>(vregs, smem + cbase + voff, cpitch, SV, gc, tid);
            if (stage_slots(gy) > NYV * NTHREADS) stage_tail<1>(smem + ybase, ypitch, SY, gy, tid, NYV * NTHREADS);
            if (stage_slots(gc) > NCV * NTHREADS) {
                stage_tail<(PLANAR ? 1 : 2)>(smem + cbase, cpitch, SC, gc, tid, NCV * NTHREADS);
                if constexpr (PLANAR) stage_tail<1>(smem + cbase + voff, cpitch, SV, gc, tid, NCV * NTHREADS);
            }
        }
        __syncthreads();
        // ---- prefetch tile j+1 while tile j is computed ---------------------------------------
        bool nstaged = false;
        if (j + 1 < ntiles) {
            nstaged = tile_geom(j + 1, ngy, ngc);
            if (nstaged) {
                stage_load(yregs, SY, ngy, tid); stage_load(cregs, SC, ngc, tid);
                if (PLANAR) stage_load(vregs, SV, ngc, tid);
            }
        }

        // ---- phase 2: 4 px x 2 rows per thread ------------------------------------------------
        const bool uniform_inside = staged && cols_inside && tb.rsum[j][5];
        if (CHV_TILED_PRIO) asm("s_setprio 0" : "+s"(ptok));
        const int yr0 = gy.r_lo + (CHV_TILED_PRIO ? ptok - j : 0), cr0 = gc.r_lo;
        // one row of the common case: every pixel of the tile is inside the picture and the layer is
        // opaque, so result = fma(px, 1, cur*0) = px exactly: the colour-matrix word is the output
        // (no canvas read, no float round trip)
        auto fast_row = [&](int ly, uint32_t (&outw)[PXT]) {
            const int ry = tb.ry[ly], rc = tb.rc[ly];
            const float yb = tb.rya[ly], iyb = 1.0f - yb, cb = tb.rca[ly], icb = 1.0f - cb;
            const int yrow = ybase + (ry - yr0) * ypitch, crow = cbase + (rc - cr0) * cpitch;
#pragma unroll
            for (int k = 0; k < PXT; k++) {
                float fy, fu, fv;
                if constexpr (PLANAR)
                    sample_y420p_lds_mix(smem, yrow + cyo[k], ypitch, crow + cco[k], voff, cpitch,
                                         icya[k] * iyb, cya[k] * iyb, icya[k] * yb, cya[k] * yb,
                                         icca[k] * icb, cca[k] * icb, icca[k] * cb, cca[k] * cb, fy, fu, fv);
                else
                    sample_nv12_lds_mix(smem, yrow + cyo[k], ypitch, crow + cco[k], cpitch,
                                        icya[k] * iyb, cya[k] * iyb, icya[k] * yb, cya[k] * yb,
                                        icca[k] * icb, cca[k] * icb, icca[k] * cb, cca[k] * cb, fy, fu, fv);
                outw[k] = yuv_to_bgra_word_absorbed(csca, fy, fu, fv);
            }
        };
        auto store_row = [&](uint8_t *drow, const uint32_t (&outw)[PXT]) {
            // Large launches (the 32-row instantiation: >= 1024 blocks, the canvases together exceed every cache) use streaming
            // (nt) stores — each canvas is written once and not read back by the launch: -1.5 % on cfg2.  Small launches (a
            // mixer tick, whose canvas the next kernel or a download reads right away) use plain stores.
            constexpr bool STREAM = THV == TH_LARGE;
            if (PXT == 4) {
                const uint4 v = make_uint4(outw[0], outw[1 % PXT], outw[2 % PXT], outw[PXT - 1]);
                if (STREAM) gst_stream(drow + (uint32_t)(xq * 4), v); else gst<uint4>(drow + (uint32_t)(xq * 4), v);
            } else if (PXT == 2) {
                const uint2 v = make_uint2(outw[0], outw[PXT - 1]);
                if (STREAM) gst_stream(drow + (uint32_t)(xq * 4), v); else gst<uint2>(drow + (uint32_t)(xq * 4), v);
            } else {
                if (STREAM) gst_stream(drow + (uint32_t)(xq * 4), outw[0]); else gst<uint32_t>(drow + (uint32_t)(xq * 4), outw[0]);
            }
        };
        const bool fast_tile = uniform_inside && opaque && absorbable;
        const bool whole_tile = fast_tile && full4 && ys0 + (j + 1) * TH <= T.H;
        uint32_t outw[RPT][PXT];
        if (whole_tile) {
            // whole tile on the common path: one straight-line block for all RPT rows, so the
            // LDS reads of a later row are in flight while an earlier row is computed
#pragma unroll
            for (int rr = 0; rr < RPT; rr++) fast_row(j * TH + tyi + rr * TYT, outw[rr]);
        }
        if (whole_tile) {
#pragma unroll
            for (int rr = 0; rr < RPT; rr++) store_row(D.ptr + (size_t)(ys0 + j * TH + tyi + rr * TYT) * D.pitch, outw[rr]);
        } else if (xq < T.W) {
#pragma unroll 1
            for (int rr = 0; rr < RPT; rr++) {
                const int ly = j * TH + tyi + rr * TYT;
                const int y = ys0 + ly;
                if (y >= T.H) continue;
                uint8_t *drow = D.ptr + (size_t)y * D.pitch;
                uint32_t outw[PXT];
                if (fast_tile) {
                    fast_row(ly, outw);
                    if (full4) store_row(drow, outw);
                    else for (int k = 0; k < PXT; k++) if (xq + k < T.W) gst<uint32_t>(drow + (uint32_t)((xq + k) * 4), outw[k]);
                    continue;
                }
                // tiles on a picture/border edge, translucent layers, unstaged tiles: one pixel at a
                // time, entries re-read from the tables
                const int ry = tb.ry[ly], rc = tb.rc[ly], rfl = tb.rfl[ly];
                const float yb = tb.rya[ly], iyb = 1.0f - yb, cb = tb.rca[ly], icb = 1.0f - cb;
                const int yrow = ybase + (ry - yr0) * ypitch, crow = cbase + (rc - cr0) * cpitch;   // staged only
#pragma unroll 1
                for (int k = 0; k < PXT; k++) {
                    if (xq + k >= T.W) break;
                    const int c = txi * PXT + k;
                    const int fl = tb.cfl[c] & rfl;
                    uint8_t *dp = drow + (uint32_t)((xq + k) * 4);
                    uint32_t cpx = CLEAR ? 0xFF000000u : gld<uint32_t>(dp);
                    if (fl & AX_BORDER) {
                        const bool in_pic = (fl & (AX_TX | AX_UV)) == (AX_TX | AX_UV);
                        uint32_t w = 0;
                        if (in_pic) {
                            const int pyx = tb.cy[c], pcx = tb.cc[c];
                            const float ya = tb.cya[c], iya = 1.0f - ya, ca = tb.cca[c], ica = 1.0f - ca;
                            float fy, fu, fv;
                            if (staged && PLANAR)
                                sample_y420p_lds_bytes(smem, yrow + (pyx - ycol0 + 16), ypitch, crow + (pcx - ccol0 + CVEC) * CTB, voff, cpitch,
                                                       iya * iyb, ya * iyb, iya * yb, ya * yb, ica * icb, ca * icb, ica * cb, ca * cb, fy, fu, fv);
                            else if (staged)
                                sample_nv12_lds_bytes(smem, yrow + (pyx - ycol0 + 16), ypitch, crow + (pcx - ccol0 + CVEC) * CTB, cpitch,
                                                      iya * iyb, ya * iyb, iya * yb, ya * yb, ica * icb, ca * icb, ica * cb, ca * cb, fy, fu, fv);
                            else
                                sample_nv12_global(SY, SC, PLANAR ? &SV : nullptr, pyx, ry, pcx, rc,
                                                   iya * iyb, ya * iyb, iya * yb, ya * yb, ica * icb, ca * icb, ica * cb, ca * cb, fy, fu, fv);
                            w = yuv_to_bgra_word(csc, (int)to_code_raw(fy), (int)to_code_raw(fu), (int)to_code_raw(fv));
                        }
                        cpx = blend_bgra_general(cpx, U, in_pic, w);
                    }
                    gst<uint32_t>(dp, cpx);
                }
            }
        }
        __syncthreads();   // tile j's LDS rectangle is free again
        staged = nstaged; gy = ngy; gc = ngc;
    }
}

template <bool CLEAR, bool PLANAR, int NYV, int NCV, int THV>
__global__ __launch_bounds__(NTHREADS, CHV_MINW) void tick_yuv_bgra_tiled(const DTick *__restrict__ ticks,
                                                                  const DLayer *__restrict__ layers,
                                                                  int n_ticks, int tiles_x, int strips_y, int kt,
                                                                  int ypitch, int yrows, int cpitch, int crows) {
    tiled_body<CLEAR, PLANAR, NYV, NCV, THV, false>(ticks, layers, n_ticks, tiles_x, strips_y, kt, ypitch, yrows, cpitch, crows);
}

// one tick, its descriptors by value (440 bytes of kernel arguments): a transient launch — chv_run_kernel as an unchanged mix.video.swift
// issues it, layer by layer — has no descriptor copy in front of it and no tick -> layer chain of dependent loads in its waves
struct TiledOne {
    DTick t;
    DLayer l;
};
template <bool CLEAR, bool PLANAR, int NYV, int NCV, int THV>
__global__ __launch_bounds__(NTHREADS, CHV_MINW) void tick_yuv_bgra_tiled_one(const TiledOne a, int tiles_x, int strips_y, int kt,
                                                                      int ypitch, int yrows, int cpitch, int crows) {
    tiled_body<CLEAR, PLANAR, NYV, NCV, THV, true>(&a.t, &a.l, 1, tiles_x, strips_y, kt, ypitch, yrows, cpitch, crows);
}

// ---------------------------------------------------------------------------
// host side: path selection and launch geometry
// ---------------------------------------------------------------------------
struct TileDims { int ypitch, yrows, cpitch, crows; size_t lds; };
static size_t tables_bytes(int th) { return th == TH_LARGE ? sizeof(TileTablesT<TH_LARGE>) : sizeof(TileTablesT<TH_SMALL>); }

static bool finite16(const float *m) {
    for (int i = 0; i < 16; i++) if (!(m[i] - m[i] == 0.f)) return false;
    return true;
}

// LDS rectangle one tile of this layer can touch, from the layer's scale factors.
static TileDims tile_dims(const DTick &T, const DLayer &L, int TH) {
    const float *U = L.u;
    // |d(uv)/d(pixel)| as a fraction of the source per output pixel
    double sxr = std::fabs((double)U[U_TEXTURE + 0] * (double)U[U_TRANSFORM + 0] * 2.0 / (double)T.W);
    double syr = std::fabs((double)U[U_TEXTURE + 5] * (double)U[U_TRANSFORM + 5] * 2.0 / (double)T.H);
    TileDims d;
    int yspan = (int)std::ceil(TW * sxr * L.src.pl[0].w) + 4;   // texels incl. tap 1 and rounding slack
    int cspan = (int)std::ceil(TW * sxr * L.src.pl[1].w) + 4;
    d.ypitch = ((yspan + 15) / 16 + 3) * 16;                    // luma bytes: vectors + alignment + 2 pad vectors
    const bool planar = L.kind == LK_BGRA_FROM_Y420P;
    d.cpitch = planar ? ((cspan + 15) / 16 + 3) * 16 : ((cspan + 7) / 8 + 3) * 16;   // chroma bytes: U and V tiles (planar) / (u, v) pairs (NV12)
    // rows a tile's taps span: <= ceil((TH-1)*scale) + 2 (tap 1 of the last row) <= ceil(TH*scale) + 2
    d.yrows = (int)std::ceil(TH * syr * L.src.pl[0].h) + 3;
    d.crows = (int)std::ceil(TH * syr * L.src.pl[1].h) + 3;
    d.lds = tables_bytes(TH) + (size_t)d.ypitch * d.yrows + (size_t)d.cpitch * d.crows * (planar ? 2 : 1);
    return d;
}

// what the staged loads need: 16-byte aligned base and pitch, rows of at least one 16-byte vector
static bool aligned16(const DPlane &p) { return (((uintptr_t)p.ptr) & 15) == 0 && (p.pitch & 15) == 0 && p.w * p.comps >= 16; }

// img_clear_bgra on its own (the first launch of an unchanged mix.video.swift tick): every canvas pixel becomes (0, 0, 0, 1).  The plane
// travels by value — a transient launch, no descriptor copy in front of it.
__global__ __launch_bounds__(256) void canvas_clear_bgra(const DPlane D, int W, int H) {
    const int y = blockIdx.y, x = ((int)blockIdx.x * 256 + (int)threadIdx.x) * 4;
    const int w = min(W, D.w);
    if (y >= min(H, D.h) || x >= w) return;
    uint8_t *row = D.ptr + (size_t)y * D.pitch;
    if (x + 4 <= w && ((((uintptr_t)row) + (size_t)x * 4) & 15) == 0) {
        *(uint4 *)(row + (size_t)x * 4) = make_uint4(0xFF000000u, 0xFF000000u, 0xFF000000u, 0xFF000000u);
    } else {
        for (int i = x; i < min(x + 4, w); i++) *(uint32_t *)(row + (size_t)i * 4) = 0xFF000000u;
    }
}

const char *fast_path_name(int path) {
    switch (path) {
    case FP_NV12_BGRA_TILED: return "tick_nv12_bgra_tiled";
    case FP_Y420P_BGRA_TILED: return "tick_y420p_bgra_tiled";
    case FP_WAVE_LAYERS: return "tick_bgra_wave";
    case FP_STREAM: return "tick_bgra_stream";
    case FP_CLEAR_BGRA: return "canvas_clear_bgra";
    case FP_WAVE_NV12: return "tick_yuv_wave<nv12>";
    case FP_WAVE_Y420P: return "tick_yuv_wave<y420p>";
    case FP_STREAM_NV12: return "tick_yuv_stream<nv12>";
    case FP_STREAM_Y420P: return "tick_yuv_stream<y420p>";
    default: return "none";
    }
}

// the single-purpose kernel: exactly one YUV layer per tick (cfg2 / cfg4): block-tiled, strips of four tiles with the next
// tile's rectangle prefetched — 0.40-0.46 ms per 256 cfg2 ticks against 0.78 for the wave-per-strip kernel.  (RGB-only ticks
// had a block-tiled kernel of their own until the wave kernel overtook it: cfg3 1.92 vs 2.23 ms, cfg5 3.36 vs 3.67.)
static int select_single_purpose(const DTick *ticks, const DLayer *layers, int n_ticks) {
    for (int i = 0; i < n_ticks; i++) {
        const DTick &T = ticks[i];
        if (T.n_layers != 1 || T.clear_first != ticks[0].clear_first) return FP_NONE;
        const DLayer &L = layers[T.first_layer];
        if ((L.kind != LK_BGRA_FROM_NV12 && L.kind != LK_BGRA_FROM_Y420P) || !(L.flags & LF_AXIS_ALIGNED)) return FP_NONE;
        if (L.kind != layers[ticks[0].first_layer].kind) return FP_NONE;
        if (!csc_absorbable(L.csc) && switches().bgra_path.load(std::memory_order_relaxed) != 2) return FP_NONE;      // (its fast rows need the absorbed matrix: the strip kernel is the faster one there)
        if (L.kind == LK_BGRA_FROM_Y420P && !aligned16(L.src.pl[2])) return FP_NONE;
        if (!finite16(L.u + U_TRANSFORM) || !finite16(L.u + U_TEXTURE) || !finite16(L.u + U_BORDER)) return FP_NONE;
        if (!aligned16(T.dst.pl[0]) || !aligned16(L.src.pl[0]) || !aligned16(L.src.pl[1])) return FP_NONE;
        if (tile_dims(T, L, TH_SMALL).lds > (size_t)LDS_BUDGET) return FP_NONE;
    }
    return layers[ticks[0].first_layer].kind == LK_BGRA_FROM_Y420P ? FP_Y420P_BGRA_TILED : FP_NV12_BGRA_TILED;
}

int select_fast_path(int target_format, const DTick *ticks, const DLayer *layers, int n_ticks, bool transient) {
    if (n_ticks <= 0) return FP_NONE;
    // A/B switches for measurements and tests (switches.h: environment read once, chv_debug_set_switch afterwards):
    //   force_general (CHV_FORCE_GENERAL=1)  everything through the general kernels
    //   bgra_path (CHV_BGRA_PATH=wave)       BGRA canvases: the wave-per-strip kernel also where the single-purpose kernel (exactly
    //                                        one YUV layer per tick) would be chosen;  =tiled: the single-purpose kernel wherever it applies
    const int bp = switches().bgra_path.load(std::memory_order_relaxed);
    if (switches().force_general.load(std::memory_order_relaxed)) return FP_NONE;
    // 4:2:0 canvases (the reference's own kernels): one wave per strip, or the general quad kernel
    // 4:2:0 canvases: cleared ticks of 1..4 axis-aligned layers without fill paint can stream their source rows (kernels_stream_yuv.hip.cpp: rows
    // outermost, a ring per layer, launches of every size); by default where that measured faster, everything else keeps the strip kernel
    if (target_format != TF_BGRA) {
        // Lone ticks of video layers from 720p up go to the strip kernel since ITS descriptors travel as a kernel argument too (WaveOne): 19.3 /
        // 20.3 / 30.6 us per 720p / 1080p / 2160p tick with the host wait against 20.5 / 22.5 / 35.7 through the rings; at 360p the streaming
        // kernel's short chunks keep it ahead (17.8 against 19.0), and the encoder side's integer frames are level (profiles/r06_notes.md section 15).
        // (Asked before the streaming kernel's own, longer list of conditions: where the strips can take such a tick they do either way.)
        if (transient && n_ticks == 1 && switches().yuv_stream.load(std::memory_order_relaxed) == 1 && (long)ticks[0].W * ticks[0].H >= 500000 &&
            ticks[0].n_layers >= 1 && ticks[0].n_layers <= WAVE_ONE_LAYERS && ticks[0].clear_first) {
            bool video_only = true;
            for (int l = 0; l < ticks[0].n_layers; l++) {
                const int k = layers[ticks[0].first_layer + l].kind;
                video_only = video_only && (k == LK_YUV_FROM_NV12 || k == LK_YUV_FROM_Y420P);
            }
            if (video_only && wave_layers_eligible(target_format, ticks, layers, n_ticks)) return target_format == TF_NV12 ? FP_WAVE_NV12 : FP_WAVE_Y420P;
        }
        if (yuv_stream_eligible(target_format, ticks, layers, n_ticks, transient)) return target_format == TF_NV12 ? FP_STREAM_NV12 : FP_STREAM_Y420P;
    }
    if (target_format != TF_BGRA)
        return wave_layers_eligible(target_format, ticks, layers, n_ticks) ? (target_format == TF_NV12 ? FP_WAVE_NV12 : FP_WAVE_Y420P) : FP_NONE;
    // ticks of 2..4 full-frame NV12 layers of one geometry on a cleared canvas: rows outermost, layers innermost (kernels_stream.hip.cpp);
    // one-layer ticks: a transient launch (chv_composite / chv_run_kernel: 19.7 against 22.6 us with the host wait through the tiled kernel —
    // its descriptors travel as kernel arguments, tools/tick_rows_sweep.sh; a one-tick BATCH reads them through pointers and is level with
    // the tiled kernel, which keeps it), otherwise on request (CHV_BGRA_PATH=stream; cfg2's 256-tick launches: 0.55 against 0.44 ms).
    // Launches of every size: the streaming kernel takes its chunk height as an
    // argument, and with 4-row chunks a lone 720p tick is 10.9 us on the chip against ~17 through 8-row strips of the strip kernel
    // (26.4 against 33.7 us per tick with the host wait; tools/tick_rows_sweep.sh, tools/stream_threshold_sweep.sh).
    if (transient && n_ticks == 1 && ticks[0].n_layers == 0 && ticks[0].clear_first && (((uintptr_t)ticks[0].dst.pl[0].ptr) & 3) == 0 &&
        (ticks[0].dst.pl[0].pitch & 3) == 0)
        return FP_CLEAR_BGRA;
    // A lone tick of ONE NV12 layer on a cleared canvas, now that the strip kernel has a twin that takes such a tick as a kernel argument
    // (tick_bgra_wave_one): the strips are ahead of the streaming kernel's twin on canvases from ~1.4 Mpixel up (a 1080p canvas: 20.8 against
    // 23.2 us with the host wait; at 720p the streaming kernel keeps it, 19.4 against 21.3) and of the tiled kernel's twin wherever the
    // streaming kernel does not apply (a picture-in-picture inset: 22.5 against 24.1 us at 720p, 21.2 against 23.4 at 1080p) —
    // tools/lone_bgra_routes.py.  Planar sources stay (their strip instantiation has no twin).
    const bool lone_nv12 = transient && bp == 0 && n_ticks == 1 && ticks[0].n_layers == 1 && ticks[0].clear_first &&
                           (layers[ticks[0].first_layer].kind == LK_BGRA_FROM_NV12 || layers[ticks[0].first_layer].kind == LK_BGRA_FROM_Y420P);
    if ((bp == 0 || bp == 3) && bgra_stream_eligible(ticks, layers, n_ticks)) {
        if (lone_nv12 && (long)ticks[0].W * ticks[0].H >= 1400000 && wave_layers_eligible(TF_BGRA, ticks, layers, n_ticks)) return FP_WAVE_LAYERS;
        if (bp == 3 || ticks[0].n_layers >= 2 || transient) return FP_STREAM;
    } else if (lone_nv12 && wave_layers_eligible(TF_BGRA, ticks, layers, n_ticks)) return FP_WAVE_LAYERS;
    if (bp != 1) {
        int p = select_single_purpose(ticks, layers, n_ticks);
        // planar sources in launches that fill the chip: the y420p-only instantiation of the wave kernel is the faster one
        // (cfg2_y420p 0.574 -> 0.532 ms; NV12 stays: 0.467 tiled vs 0.487 wave); small launches keep the tiled kernel's short strips
        if (p == FP_Y420P_BGRA_TILED && bp != 2) {
            long strips = 0;
            for (int i = 0; i < n_ticks; i++) strips += (long)((ticks[i].W + 63) / 64) * ((ticks[i].H + 15) / 16);
            if (strips >= 8192 && wave_layers_eligible(TF_BGRA, ticks, layers, n_ticks)) return FP_WAVE_LAYERS;
        }
        if (p != FP_NONE) return p;
    }
    // any mix of NV12 / y420p / BGRA / RGBA layers, any number of them.  (Launches that do not clear take the strip kernel also when ALL their
    // layers need its per-pixel path — rotated overlays added to composed canvases —: its grid covers the layers' boxes only, and measured
    // against the general kernel on the same boxes it is the faster one, 81 against 96 us for pipeline_logo's 128 rotated 320 x 180 logos.)
    return wave_layers_eligible(TF_BGRA, ticks, layers, n_ticks, !ticks[0].clear_first) ? FP_WAVE_LAYERS : FP_NONE;
}

// A batch whose ticks are "2..4 full-frame videos of one geometry, then something else" (a rotated logo, overlays, a fifth layer): how many
// leading layers of EVERY tick the streaming kernel takes as a launch of its own (cleared), the rest following in a second launch that continues
// on the canvas — the layer-by-layer semantics already are that (the canvas is re-quantised between layers either way).  0: no.
// The point: one layer the strip machinery has to apply per pixel (KINDS bit 3) otherwise puts the WHOLE tick on the strip kernel's most general
// instantiation (four waves per SIMD): the pipeline tick + a rotated logo 1.268 ms per 128 ticks against 0.62 for its four videos alone.
int split_stream_prefix(const DTick *ticks, const DLayer *layers, int n_ticks) {
    if (n_ticks < 1 || !switches().stream.load(std::memory_order_relaxed) || switches().bgra_path.load(std::memory_order_relaxed) != 0) return 0;
    if (switches().force_general.load(std::memory_order_relaxed)) return 0;
    int least = 1 << 30;
    for (int i = 0; i < n_ticks; i++) least = std::min(least, ticks[i].n_layers);
    std::vector<DTick> head(ticks, ticks + n_ticks);
    for (int k = std::min(4, least - 1); k >= 2; k--) {            // (every tick keeps at least one layer for the second launch)
        for (int i = 0; i < n_ticks; i++) head[(size_t)i].n_layers = k;
        if (bgra_stream_eligible(head.data(), layers, n_ticks)) return k;
    }
    return 0;
}
int fast_path_stream_bgra() { return FP_STREAM; }
// the path of a split batch's second launch
int select_tail_path(int target_format, const DTick *ticks, const DLayer *layers, int n_ticks) {
    const int p = select_fast_path(target_format, ticks, layers, n_ticks, false);
    if (p == FP_NONE && target_format == TF_BGRA && !switches().force_general.load(std::memory_order_relaxed) &&
        wave_layers_eligible(TF_BGRA, ticks, layers, n_ticks, true)) return FP_WAVE_LAYERS;
    return p;
}

bool fast_path_is_wave(int path) { return path == FP_WAVE_LAYERS || path == FP_WAVE_NV12 || path == FP_WAVE_Y420P; }
bool fast_path_by_value(int path) { return path == FP_STREAM || path == FP_STREAM_NV12 || path == FP_STREAM_Y420P || path == FP_NV12_BGRA_TILED || path == FP_Y420P_BGRA_TILED || path == FP_CLEAR_BGRA; }

hipError_t launch_tick_fast(int path, const DTick *ticks_host, const DLayer *layers_host,
                            const DTick *ticks, const DLayer *layers, int n_ticks,
                            int maxW, int maxH, hipStream_t stream) {
    if (path == FP_CLEAR_BGRA) {
        const DTick &T = ticks_host[0];
        if (T.W <= 0 || T.H <= 0) return hipSuccess;
        hipLaunchKernelGGL(canvas_clear_bgra, dim3((unsigned)((T.W + 1023) / 1024), (unsigned)T.H), dim3(256), 0, stream, T.dst.pl[0], T.W, T.H);
        return hipGetLastError();
    }
    if (path == FP_STREAM) return launch_bgra_stream(ticks_host, layers_host, ticks, layers, n_ticks, maxW, maxH, stream);
    if (path == FP_STREAM_NV12) return launch_yuv_stream(TF_NV12, ticks_host, layers_host, ticks, layers, n_ticks, maxW, maxH, stream);
    if (path == FP_STREAM_Y420P) return launch_yuv_stream(TF_Y420P, ticks_host, layers_host, ticks, layers, n_ticks, maxW, maxH, stream);
    if (path == FP_WAVE_LAYERS) return launch_wave_layers(TF_BGRA, ticks_host, layers_host, ticks, layers, n_ticks, maxW, maxH, stream);
    if (path == FP_WAVE_NV12) return launch_wave_layers(TF_NV12, ticks_host, layers_host, ticks, layers, n_ticks, maxW, maxH, stream);
    if (path == FP_WAVE_Y420P) return launch_wave_layers(TF_Y420P, ticks_host, layers_host, ticks, layers, n_ticks, maxW, maxH, stream);
    if (path != FP_NV12_BGRA_TILED && path != FP_Y420P_BGRA_TILED) return hipErrorNotSupported;
    const bool clear = ticks_host[0].clear_first != 0, planar = path == FP_Y420P_BGRA_TILED;
    const int tiles_x = (maxW + TW - 1) / TW;
    // tile height: 32 rows when that still gives every CU a few blocks and the rectangles fit the LDS budget, else 16
    auto dims_for = [&](int th, TileDims &m) -> size_t {
        m = TileDims{ 0, 0, 0, 0, 0 };
        for (int i = 0; i < n_ticks; i++) {
            TileDims d = tile_dims(ticks_host[i], layers_host[ticks_host[i].first_layer], th);
            m.ypitch = std::max(m.ypitch, d.ypitch); m.yrows = std::max(m.yrows, d.yrows);
            m.cpitch = std::max(m.cpitch, d.cpitch); m.crows = std::max(m.crows, d.crows);
        }
        return tables_bytes(th) + (size_t)m.ypitch * m.yrows + (size_t)m.cpitch * m.crows * (planar ? 2 : 1);
    };
    // staging slots a tile can need: rows x vectors per row; `pad` = 2 counts the padding vectors of rectangles at a
    // picture edge (the worst case), 0 the interior tiles
    auto slots_fit = [&](const TileDims &m, int ny, int nc, int pad) {
        return m.yrows * (m.ypitch / 16 - 2 + pad) <= ny * NTHREADS && m.crows * (m.cpitch / 16 - 2 + pad) <= nc * NTHREADS;
    };
    TileDims m;
    int th = TH_LARGE;
    size_t lds = dims_for(th, m);
    const long blocks_large = (long)n_ticks * tiles_x * ((maxH + TH_LARGE - 1) / TH_LARGE);
    // 32-row tiles when the launch is large and interior rectangles fit the prefetch registers; edge rectangles may spill
    // into stage_tail as long as slot numbers stay below 1024.
    // CHV_TILE_ROWS (A/B and test switch): 16 = never; 32 = whatever the launch size, and even when interior rectangles
    // overflow the prefetch registers (everything beyond them then goes through stage_tail).
    const int rows_forced = switches().tile_rows.load(std::memory_order_relaxed);
    const bool tail_ok = slots_fit(m, 4, 4, 2);
    const bool large_wanted = rows_forced == TH_LARGE || (rows_forced != TH_SMALL && blocks_large >= 1024 && slots_fit(m, 3, 2, 0));
    if (!(large_wanted && tail_ok && lds <= (size_t)LDS_BUDGET)) {
        th = TH_SMALL;
        lds = dims_for(th, m);
    }
    if (lds > (size_t)LDS_BUDGET) {
        // per-tick maxima combined exceed the budget: shrink to it; tiles that do not fit
        // fall back to unstaged taps inside the kernel
        m.yrows = std::max(1, (int)((LDS_BUDGET - tables_bytes(th)) / 2 / m.ypitch));
        m.crows = std::max(1, (int)((LDS_BUDGET - tables_bytes(th)) / 2 / (m.cpitch * (planar ? 2 : 1))));
        lds = tables_bytes(th) + (size_t)m.ypitch * m.yrows + (size_t)m.cpitch * m.crows * (planar ? 2 : 1);
    }
    // strips of kt tiles: KT amortises the column tables best, but a small launch (one mixer tick = 120 strips of 4)
    // would leave most CUs idle — shorter strips until there are a few blocks per CU
    int kt = KT;
    while (kt > 1 && (long)n_ticks * tiles_x * ((maxH + kt * th - 1) / (kt * th)) < 1024) kt >>= 1;
    int tiles_y = (maxH + kt * th - 1) / (kt * th);
    int per_xcd = (n_ticks * tiles_x * tiles_y + 7) / 8;
    dim3 grid((unsigned)(per_xcd * 8));
    TiledOne one;
    if (!ticks) {
        // one tick, descriptors as kernel arguments (launch_transient)
        if (n_ticks != 1) return hipErrorInvalidValue;
        one.t = ticks_host[0];
        one.l = layers_host[ticks_host[0].first_layer];
        one.t.first_layer = 0;
    }
#define CHV_LAUNCH(C, P, NY, NC, H) do { \
        if (ticks) hipLaunchKernelGGL((tick_yuv_bgra_tiled<C, P, NY, NC, H>), grid, dim3(NTHREADS), lds, stream, ticks, layers, \
                                      n_ticks, tiles_x, tiles_y, kt, m.ypitch, m.yrows, m.cpitch, m.crows); \
        else hipLaunchKernelGGL((tick_yuv_bgra_tiled_one<C, P, NY, NC, H>), grid, dim3(NTHREADS), lds, stream, one, \
                                tiles_x, tiles_y, kt, m.ypitch, m.yrows, m.cpitch, m.crows); } while (0)
#define CHV_LAUNCH_N(C, P) do { if (th == TH_LARGE) { if (small) CHV_LAUNCH(C, P, 2, 1, TH_LARGE); else CHV_LAUNCH(C, P, 3, 2, TH_LARGE); } \
                                else { if (small) CHV_LAUNCH(C, P, 2, 1, TH_SMALL); else CHV_LAUNCH(C, P, 3, 2, TH_SMALL); } } while (0)
    const bool small = slots_fit(m, 2, 1, 2);
    if (clear && planar) CHV_LAUNCH_N(true, true);
    else if (clear) CHV_LAUNCH_N(true, false);
    else if (planar) CHV_LAUNCH_N(false, true);
    else CHV_LAUNCH_N(false, false);
#undef CHV_LAUNCH_N
#undef CHV_LAUNCH
    return hipGetLastError();
}

}  // namespace chv

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
#include "wave_common.hip.h"
#include "bgra_pixel.hip.h"
#include "switches.h"

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <type_traits>

// CHV_ABL: timing-only ablations (results are wrong): 1 = no staging, 2 = no pixel rows, 4 = no canvas stores
#ifndef CHV_ABL
#define CHV_ABL 0
#endif
// Rectangles that touch no picture edge staged by the instantiation without clamping / patching code: YUV sources only (bit 0).
// Measured in one call: pipeline 1.834 -> 1.734 ms, mixed 0.839 -> 0.799 with it; RGB sources (bit 1) cfg3 1.444 -> 1.611 and the
// 4:2:0 kernel 0.484 -> 0.507 (y420p_main) against it — the register allocation of the row loops shifts with the code around them.
// bit 2: narrow interior YUV rectangles through the shift-and-mask slot map (wstage_load_p2)
#ifndef CHV_WAVE_INTERIOR
#define CHV_WAVE_INTERIOR 5
#endif
#ifndef CHV_WAVE_PRIO
#define CHV_WAVE_PRIO 1
#endif
// RGB rectangles that touch no picture edge and need no byte swap are filled by LDS-DMA (global_load_lds_dwordx4): no staging registers, no slot
// arithmetic, no LDS write instructions — cfg3 1.3035 -> 1.2485 ms, cfg5 2.140 -> 2.050 (same call, profiles/r05_notes.md section 9, where the
// version that also PREFETCHED the next layer's rectangle into a second region is recorded: the LDS it takes costs more waves than the overlap
// returns).  0: off (the A/B; CHV_WAVE_DMA=0 in the environment, or chv_debug_set_switch("CHV_WAVE_DMA", "0"), does the same at run time).
#ifndef CHV_DMA_MUTATE
#define CHV_DMA_MUTATE 0      // (tests of the tests: a non-zero value shifts what the DMA fetches)
#endif
#ifndef CHV_WAVE_DMA
#define CHV_WAVE_DMA 1
#endif
// 0: this translation unit — the kernels that compute their geometry, the launcher, the build flags; 1: kernels_wave_cached.hip.cpp — the
// instantiations that read it from a batch's tables (geom_cache.h), nothing else
#ifndef CHV_WAVE_TU
#define CHV_WAVE_TU 0
#endif
#pragma clang fp contract(off)

namespace chv {

// one LDS-DMA instruction: the lane's 16 bytes from ITS global address to LDS at m0 + lane * 16 (tools/probe_lds_dma.cpp; kernels_stream.hip.cpp).
// M0 is set here every time; tests/test_device_code_contract.py checks that nothing else in the object touches it.
// NOT `asm volatile`, no memory clobber: either makes every later read of the tick / layer descriptors "possibly clobbered" and the compiler
// fetches them per lane (75 instead of 16 global_load_dword in the RGB-only instantiation, cfg3 1.35 -> 2.41 ms measured).  What orders these
// instructions against the LDS reads is a token instead: a VGPR the asm statements pretend to update — it goes in after depending on the
// previous layer's pixels (wave_dma_after), and the row loops' LDS addresses depend on it after the wait (wave_dma_wait).
CHV_DEV void wave_dma16(const uint8_t *p, bool active, uint32_t m0, int &tok) {
    if (active) asm("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, off" : "+v"(tok) : "s"(m0), "v"(p));
}
CHV_DEV void wave_dma_wait(int &tok) { asm("s_waitcnt vmcnt(0)" : "+v"(tok)); }
template <int N>
CHV_DEV void wave_dma_after(int &tok, const uint32_t (&cv)[N]) {
    if constexpr (N == 16)
        asm("" : "+v"(tok) : "v"(cv[0]), "v"(cv[1]), "v"(cv[2]), "v"(cv[3]), "v"(cv[4]), "v"(cv[5]), "v"(cv[6]), "v"(cv[7]), "v"(cv[8]), "v"(cv[9]), "v"(cv[10]),
            "v"(cv[11]), "v"(cv[12]), "v"(cv[13]), "v"(cv[14]), "v"(cv[15]));
    else
        asm("" : "+v"(tok) : "v"(cv[0]), "v"(cv[1]), "v"(cv[2]), "v"(cv[3]), "v"(cv[4]), "v"(cv[5]), "v"(cv[6]), "v"(cv[7]));
}

// Integer colour matrix on biased codes, channels returned as float codes (cf. yuv_to_bgra_word).
// clip8(x >> 16) = byte 2 of clamp(x, 0, 0xFFFFFF): v_med3_i32 + v_cvt_f32_ubyte2 per channel (two slow-class instructions,
// 3.6 ns per wave) — v_ashr_pk_u8_i32 + v_cvt_f32_ubyteN measured 3.45 + 1.8 per channel pair / channel
// (the packing instruction issues at a quarter of the rate, tools/ubench_tput.cpp).
// (yuv_to_bgr_floats: pixel_math.hip.h — shared with tick_bgra_stream and the exhaustive device self-test of the matrices)

#ifndef CHV_WAVE_MINW
#define CHV_WAVE_MINW 6
#endif
#ifndef CHV_WAVE_MINW16
#define CHV_WAVE_MINW16 5
#endif
// CHV_WAVE_FENCE = n > 0: keep the scheduler from interleaving more than n rows of a lane's pixels in the branch-free loops
// (fewer live temporaries)
#ifndef CHV_WAVE_FENCE
#define CHV_WAVE_FENCE 0
#endif
#define WAVE_ROW_FENCE(j) do { if (CHV_WAVE_FENCE > 0 && (j) > 0 && (j) % (CHV_WAVE_FENCE > 0 ? CHV_WAVE_FENCE : 1) == 0) __builtin_amdgcn_sched_barrier(0); } while (0)
// Strip height WTH (rows per lane), a template parameter picked per launch (launch_wave_layers): 16 rows halve a strip's fixed
// costs per pixel (pipeline 2.14 -> 1.86 ms, cfg3 1.91 -> 1.68, cfg5 3.17 -> 2.92 at 96 VGPRs = 5 waves per SIMD) but double
// the rows that go through the per-pixel path where a layer's edge crosses a strip (ticks with small overlays: 0.88 -> 1.05 ms),
// so they are used when every layer of the launch covers (almost) the whole canvas; 8 rows otherwise (80 VGPRs, 6 waves).      // strip height: rows per lane (16: -14 % on the 4 x NV12 pipeline at 128 VGPRs, but the LDS
                                        // footprint of 4-byte texel rectangles then halves the occupancy of mixed ticks: 3.0 vs 0.85 ms)
#ifndef CHV_WAVE_COVER
#define CHV_WAVE_COVER 1
#endif
#ifndef CHV_WAVE_PIXEL_UNROLL
#define CHV_WAVE_PIXEL_UNROLL 4      // rows of a per-pixel layer in flight together (their gathers are dependent chains of L2 round trips)
#endif
// CACHED: the per-layer geometry comes from the batch's tables (WaveStrip::setup_cached) — these instantiations contain no set-up code; they are
// compiled in a translation unit of their own (kernels_wave_cached.hip.cpp) and launched for batches whose staged layers all have tables
template <int WTH, bool CLEAR, int KINDS, bool CACHED>
__global__ __launch_bounds__(WAVE_BLOCK, (WTH == 16 ? ((KINDS & 8) ? 4 : CHV_WAVE_MINW16) : ((KINDS & 8) ? 5 : CHV_WAVE_MINW))) void tick_bgra_wave(const DTick *__restrict__ ticks,
                                                                         const DLayer *__restrict__ layers,
                                                                         int n_ticks, int strips_x, int strips_y, uint32_t strips_magic, uint32_t strips_x_magic,
                                                                         int p0pitch, int p0rows, int p1pitch, int p1rows, int planar_any) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem_all[];
    using Strip = WaveStrip<WTH, CHV_WAVE_INTERIOR, KINDS>;
    Strip S;
    if (!S.init(ticks, layers, n_ticks, strips_x, strips_y, strips_magic, strips_x_magic, smem_all, p0pitch, p0rows, p1pitch, p1rows, planar_any)) return;   // (no block barrier anywhere: waves may leave)
    p1pitch = S.p1pitch;                 // (the side-by-side layout keeps chroma in the rows of the plane-0 region: WaveStrip::init)
    const DTick &T = *S.T;
    const DLayer *L = S.L;
    const int nl = S.nl, lane = S.lane, x = S.x, y0 = S.y0;
    const bool col_in = S.col_in;
    // canvas plane BY VALUE, read once (see kernels_wave_yuv.hip.cpp: descriptor reads after the first canvas store would be
    // vector loads with a full wait each)
    const DPlane D = T.dst.pl[0];
    const int TH = T.H;
    uint8_t *smem = S.smem;
    const uint4 *rowtab = S.rowtab;
    const int voff = S.voff;
    (void)lane;

    // (the layer index is wave-uniform; saying so keeps the descriptor reads on the scalar unit: left to its divergence analysis
    // the compiler fetched every uniform of a layer with per-lane global loads — 90 vector loads per wave)
    // A strip that lies inside the inner box of an opaque picture (LF_COVERS: a picture-in-picture inset, a quadrant of a grid) starts at that
    // layer: whatever is beneath does not show.  (CHV_WAVE_COVER=0: the A/B.)
    int first = 0;
    if (CHV_WAVE_COVER) {
        // (the tick says which of its layers can cover: no bit, no descriptor reads)
        uint32_t cm = (uint32_t)T.cover_mask & ~1u;
        const int x1 = min(S.x0 + WTW, T.W), y1 = min(y0 + WTH, TH);
        while (cm) {
            const int k = 31 - __builtin_clz(cm);
            cm &= ~(1u << k);
            if (k >= nl) continue;
            const DLayer &K = L[k];
            if (S.x0 >= K.ibox[0] && x1 <= K.ibox[2] && y0 >= K.ibox[1] && y1 <= K.ibox[3]) { first = k; break; }
        }
    }
    int l = __builtin_amdgcn_readfirstlane(S.next_hit(first));
    // a strip no layer touches on a canvas that is not cleared keeps its pixels: nothing to read, nothing to write (the second launch of a
    // split batch — a logo or overlays over videos the streaming kernel composed — leaves most strips this way)
    if (!CLEAR && l >= nl) return;

    // ---- canvas pixels of this lane: packed BGRA codes, row j in cv[j] -----------------------------------------------
    uint32_t cv[WTH];
#pragma unroll
    for (int j = 0; j < WTH; j++) cv[j] = 0xFF000000u;               // img_clear_bgra: (0,0,0,1)
    if (!CLEAR && col_in) {
#pragma unroll
        for (int j = 0; j < WTH; j++)
            if (y0 + j < TH) cv[j] = gld_at<uint32_t>(D.ptr + (size_t)(y0 + j) * D.pitch, (uint32_t)x * 4u);
    }

    WLayer cur;
    bool have_geom = false;            // `cur` and the row table hold the geometry of the layer handled just before (LF_SAME_GEOM)
    // DMA staging (planar_any bit 5: on): lane -> (row, vector) of one instruction — a region's rows are p0pitch bytes = P16 vectors apart, an
    // instruction fills 64 / P16 of them; `tok`: see wave_dma16
    constexpr bool DMA = CHV_WAVE_DMA && ((KINDS & 15) == 4 || (CHV_WAVE_DMA > 1 && (KINDS & 4) != 0));
    const bool dma_on = DMA && (planar_any & 32) != 0;
    const int P16 = p0pitch >> 4, RPI = P16 > 0 ? 64 / P16 : 0;
    int lr = 0, lv = lane, tok = 0;
    if (DMA) {
        while (lv >= P16 && lr < 64) { lv -= P16; lr++; }
        asm("v_mov_b32 %0, 0" : "=v"(tok));              // (0, but the compiler does not know)
    }
    while (l < nl) {
        const DLayer &Ly = L[l];
        if constexpr ((KINDS & 8) != 0) {
            // Layers the strip machinery cannot stage (rotation, shear, unbounded matrices — the launch has some: KINDS bit 3)
            // are applied pixel by pixel with the general kernel's code, in z order with everything else: one rotated logo does
            // not send the whole tick to the general kernel.
            if (Ly.kind == LK_BGRA_METAL || (Ly.flags & (LF_AXIS_ALIGNED | LF_BOUNDED)) != (LF_AXIS_ALIGNED | LF_BOUNDED)) {
                const float gsx = (float)T.W, gsy = (float)TH;
#pragma unroll CHV_WAVE_PIXEL_UNROLL
                for (int j = 0; j < WTH; j++) {
                    const int y = y0 + j;
                    uint32_t c = cv[0];
#pragma unroll
                    for (int k = 1; k < WTH; k++) c = j == k ? cv[k] : c;
                    if (col_in && y < TH && x >= Ly.bbox[0] && x < Ly.bbox[2] && y >= Ly.bbox[1] && y < Ly.bbox[3]) c = apply_layer_bgra(Ly, x, y, gsx, gsy, c);
#pragma unroll
                    for (int k = 0; k < WTH; k++) cv[k] = j == k ? c : cv[k];
                }
                l = __builtin_amdgcn_readfirstlane(S.next_hit(l + 1));
                have_geom = false;
                continue;
            }
        }
        // Issue priority for the latency-bound phases (geometry, staging): a wave in them has few instructions to issue and long
        // waits between them, so letting it go first whenever it can shortens its chain, and more of the resident waves are in
        // their row loops at any time (pipeline -1.4 %, cfg3 -2.2 %).  Through non-volatile asm with a token operand: the
        // __builtin_amdgcn_s_setprio call counts as a side effect after which hipcc reads the descriptors per lane (+44 %).
        int ptok = l;
        if (CHV_WAVE_PRIO) asm("s_setprio 3" : "+s"(ptok));
        // Layers whose geometry inputs are bit-identical to their predecessor's (LF_SAME_GEOM, host-checked: equal bounding boxes, so
        // the predecessor was a hit for this strip as well) keep its column entry, row table and rectangles; only the planes change.
        // (overwrites the row table: the previous layer's pixels are done.  From the batch's geometry table where there is one — setup_cached)
        if (!(have_geom && (Ly.flags & LF_SAME_GEOM))) { if constexpr (CACHED) S.setup_cached(l, cur); else S.setup(ptok, cur); }
        have_geom = true;
        // a rectangle LDS-DMA can fill: texels in canvas order, away from every picture edge, contiguous rows (not the pair form), at most eight
        // instructions' worth of them
        if (DMA && dma_on && cur.staged && Strip::is_rgb(Ly.kind) && Ly.swizzle == 0 && !cur.g0.edge && !cur.g0.pair && RPI > 0 && cur.g0.rows <= 8 * RPI && cur.g0.nvec + 1 <= P16 && !(CHV_ABL & 1)) {
            const DPlane &P = Ly.src.pl[0];
            const uint8_t *base = P.ptr + (size_t)cur.g0.r_lo * P.pitch + cur.g0.b0;                  // (uniform)
            const uint32_t lds0 = (uint32_t)(size_t)(smem + S.base0);
            const bool lane_ok = lr < RPI && lv >= 1 && lv <= cur.g0.nvec;
            wave_dma_after(tok, cv);          // (the region's last readers — the rows of the layer before — are done)
            for (int r0 = 0; r0 < cur.g0.rows; r0 += RPI) {
                const int row = r0 + lr;
                wave_dma16(base + (size_t)row * P.pitch + (size_t)((lv - 1) * 16) + CHV_DMA_MUTATE, lane_ok && row < cur.g0.rows,
                           (uint32_t)__builtin_amdgcn_readfirstlane((int)(lds0 + (uint32_t)(r0 * p0pitch))), tok);
            }
            wave_dma_wait(tok);
        } else if (cur.staged && !(CHV_ABL & 1)) S.stage(l, cur);
        wave_lds_fence();
        const int ln = __builtin_amdgcn_readfirstlane(S.next_hit(l + 1));
        if (CHV_WAVE_PRIO) { asm("s_setprio 0" : "+s"(ptok)); cur.cyo += ptok - l; }       // (ptok - l = 0, opaque: pins the asm here)
        const int cyo_r = DMA ? cur.cyo + tok : cur.cyo;         // (tap 0 of this lane's column; after the pin above, and through the token: not before the wait)

        if (CHV_ABL & 2) cv[0] += (uint32_t)(cur.cyo ^ cur.cco ^ __float_as_int(cur.cya) ^ __float_as_int(cur.cca) ^ cur.cfl);
        else {
            const float *U = Ly.u;
            const float opacity = U[U_OPACITY];
            const bool rgb = Strip::is_rgb(Ly.kind);                 // (compile-time constants in the single-class instantiations)
            const bool planar = !rgb && Strip::is_planar(Ly.kind);
            const bool nofill = (Ly.flags & LF_NO_FILL) != 0;
            // opacity in [0,1] and no fill: every blend is a convex combination of code values, so neither the clamp of the
            // fill step nor the saturation of the store can trigger; with every pixel of the strip inside the picture the
            // loop is branch-free.  Strips that a layer's edge crosses (8-row kernel, i.e. launches with layers smaller than the
            // canvas) run the same loop in a MASKED instantiation, still branch-free: every row is computed (row offsets are
            // clamped into the staged rectangle), and a pixel takes the result only if its column and its row are inside the
            // picture; inside the border quad but outside the picture it only gets its alpha byte forced (with no fill,
            // clamp(fma(f, 0, c * 1)) = c).  A per-row branch instead of the selects measured 10 % slower on ticks with overlays:
            // it keeps the rows' LDS reads from overlapping.
#ifndef CHV_WAVE_MASKED
#define CHV_WAVE_MASKED 1
#endif
#ifndef CHV_WAVE_CARRY
#define CHV_WAVE_CARRY 1
#endif
#ifndef CHV_WAVE_MASK16
#define CHV_WAVE_MASK16 0
#endif
            constexpr bool CAN_MASK = CHV_WAVE_MASKED && (WTH == 8 || CHV_WAVE_MASK16);
            const bool fast = cur.staged && (CAN_MASK || cur.all_inside) && nofill && opacity >= 0.f && opacity <= 1.f;
            const bool lane_pic = cur.cfl == AX_ALL, lane_border = (cur.cfl & AX_BORDER) != 0;
            // MASKED: the pixel's new value, given its row's flags (uniform, from the row table)
            auto commit_px = [&](auto masked_c, int j, uint32_t old, uint32_t nv) {
                if constexpr (!decltype(masked_c)::value) return nv;
                else {
                    const uint32_t rfl = row_fast_flags<WTH>(rowtab, j);
                    const bool pic = lane_pic && rfl == (uint32_t)AX_ALL;
                    const bool border = lane_border && (rfl & AX_BORDER) != 0;
                    return pic ? nv : (border ? (old | 0xFF000000u) : old);
                }
            };
            if (fast && rgb) {
                const float ka = opacity * kInv255;
                const float a = cur.cya, ia = 1.0f - a;
                auto rgb_rows = [&](auto masked_c) {
#pragma unroll
                    for (int j = 0; j < WTH; j++) {
                        WAVE_ROW_FENCE(j);
                        const RowFast rw = row_fast<WTH, false>(rowtab, j);
                        const float b = rw.yb, ib = rw.iyb;
                        const uint8_t *p0 = smem + (rw.yoff + cyo_r);
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
                        cv[j] = commit_px(masked_c, j, c, pack_codes(__builtin_fmaf(q0, al, ub0(c) * ial), __builtin_fmaf(q1, al, ub1(c) * ial),
                                                                      __builtin_fmaf(q2, al, ub2(c) * ial), 0xFF000000u));
                    }
                };
                // Native-resolution layers (source rows advance one per canvas row, checked on the row table): the lower tap row
                // of a pixel is the upper tap row of the pixel below it, so its eight code-to-float conversions — a third of the
                // row's slow-class instructions — and its LDS reads are carried down the lane instead of repeated.
                auto rgb_rows_carried = [&]() {
                    const uint8_t *p = smem + (row_fast<WTH, false>(rowtab, 0).yoff + cyo_r);
                    uint32_t ut0 = ((const uint32_t *)p)[0], ut1 = ((const uint32_t *)p)[1];
                    float t00 = ub0(ut0), t01 = ub1(ut0), t02 = ub2(ut0), t03 = ub3(ut0);
                    float t10 = ub0(ut1), t11 = ub1(ut1), t12 = ub2(ut1), t13 = ub3(ut1);
#pragma unroll
                    for (int j = 0; j < WTH; j++) {
                        WAVE_ROW_FENCE(j);
                        const float b = *(const float *)(rowtab + 2 * WTH + j), ib = 1.0f - b;
                        p += p0pitch;
                        const uint32_t ub_0 = ((const uint32_t *)p)[0], ub_1 = ((const uint32_t *)p)[1];
                        const float b00 = ub0(ub_0), b01 = ub1(ub_0), b02 = ub2(ub_0), b03 = ub3(ub_0);
                        const float b10 = ub0(ub_1), b11 = ub1(ub_1), b12 = ub2(ub_1), b13 = ub3(ub_1);
                        const float w00 = ia * ib, w10 = a * ib, w01 = ia * b, w11 = a * b;
                        const float q0 = cs_mix(w00, w10, w01, w11, t00, t10, b00, b10);
                        const float q1 = cs_mix(w00, w10, w01, w11, t01, t11, b01, b11);
                        const float q2 = cs_mix(w00, w10, w01, w11, t02, t12, b02, b12);
                        const float q3 = cs_mix(w00, w10, w01, w11, t03, t13, b03, b13);
                        const float al = q3 * ka, ial = 1.f - al;
                        const uint32_t c = cv[j];
                        cv[j] = pack_codes(__builtin_fmaf(q0, al, ub0(c) * ial), __builtin_fmaf(q1, al, ub1(c) * ial),
                                           __builtin_fmaf(q2, al, ub2(c) * ial), 0xFF000000u);
                        t00 = b00; t01 = b01; t02 = b02; t03 = b03; t10 = b10; t11 = b11; t12 = b12; t13 = b13;
                    }
                };
                if (CHV_WAVE_CARRY && cur.unit_rows && cur.all_inside) rgb_rows_carried();
                else if constexpr (CAN_MASK) { if (cur.all_inside) rgb_rows(std::false_type{}); else rgb_rows(std::true_type{}); }
                else rgb_rows(std::false_type{});
            } else if (fast) {
                const CscFolded cscb = csc_fold_biased(kCsc[Ly.csc & 3]);
                const float al = 1.0f * opacity, ial = 1.f - al;
                // column weights times 2^24: the taps enter v_fma_mix_f32 as binary16 denormals (tap_h, pixel_math.hip.h); the
                // products with the row weights below are the reference's products times 2^24, exactly
                const float iya0 = 1.0f - cur.cya, ica0 = 1.0f - cur.cca;
                const float ya = cur.cya * kTapScale, iya = iya0 * kTapScale;
                const float cts = planar ? kTapScale : kChromaTapScale, ca = cur.cca * cts, ica = ica0 * cts;
                auto yuv_fast = [&](auto planar_c, auto opaque_c, auto masked_c) {
                    constexpr bool PL = decltype(planar_c)::value, OP = decltype(opaque_c)::value;
#pragma unroll
                    for (int j = 0; j < WTH; j++) {
                        WAVE_ROW_FENCE(j);
                        const RowFast rw = row_fast<WTH, true>(rowtab, (CHV_ABL & 16) ? 0 : j);
                        const float yb = rw.yb, iyb = rw.iyb, cbw = rw.cb, icb = rw.icb;
                        const int yo = rw.yoff + cur.cyo + ((CHV_ABL & 16) ? j * p0pitch : 0), co = rw.coff + cur.cco;
                        const float w00 = iya * iyb, w10 = ya * iyb, w01 = iya * yb, w11 = ya * yb;
                        const float c00 = ica * icb, c10 = ca * icb, c01 = ica * cbw, c11 = ca * cbw;
                        float fy, fu, fv;
                        if constexpr (PL) sample_y420p_lds_mix(smem, yo, p0pitch, co, voff, p1pitch, w00, w10, w01, w11, c00, c10, c01, c11, fy, fu, fv);
                        else sample_nv12_lds_mix(smem, yo, p0pitch, co, p1pitch, w00, w10, w01, w11, c00, c10, c01, c11, fy, fu, fv);
                        if constexpr (OP) {
                            cv[j] = commit_px(masked_c, j, cv[j], yuv_to_bgra_word(cscb, (int)code_biased(fy), (int)code_biased(fu), (int)code_biased(fv)));   // fma(p, 1, c * 0) = p exactly
                        } else {
                            float pb, pg, pr;
                            yuv_to_bgr_floats(cscb, (int)code_biased(fy), (int)code_biased(fu), (int)code_biased(fv), pb, pg, pr);
                            const uint32_t c = cv[j];
                            cv[j] = commit_px(masked_c, j, c, pack_codes(__builtin_fmaf(pb, al, ub0(c) * ial), __builtin_fmaf(pg, al, ub1(c) * ial),
                                                                          __builtin_fmaf(pr, al, ub2(c) * ial), 0xFF000000u));
                        }
                    }
                };
                const bool opaque = (Ly.flags & LF_OPAQUE) != 0;
                auto by_mask = [&](auto planar_c, auto opaque_c) {
                    if constexpr (CAN_MASK) { if (cur.all_inside) yuv_fast(planar_c, opaque_c, std::false_type{}); else yuv_fast(planar_c, opaque_c, std::true_type{}); }
                    else yuv_fast(planar_c, opaque_c, std::false_type{});
                };
                if (planar) { if (opaque) by_mask(std::true_type{}, std::true_type{}); else by_mask(std::true_type{}, std::false_type{}); }
                else        { if (opaque) by_mask(std::false_type{}, std::true_type{}); else by_mask(std::false_type{}, std::false_type{}); }
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
                                const uint8_t *q0 = smem + (ry + cyo_r);
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

    if (CHV_ABL & 4) {
#pragma unroll
        for (int j = 0; j < WTH; j++) asm volatile("" :: "v"(cv[j]));
    } else if (col_in) {
#pragma unroll
        for (int j = 0; j < WTH; j++)
            if (y0 + j < TH) gst_at<uint32_t>(D.ptr + (size_t)(y0 + j) * D.pitch, (uint32_t)x * 4u, cv[j]);     // (row base: scalar)
    }
}

// ---------------------------------------------------------------------------
// launch (geometry, LDS sizing and eligibility of both wave kernels: kernels_wave_yuv.hip.cpp)
// ---------------------------------------------------------------------------
#define CHV_STR2(x) #x
#define CHV_STR(x) CHV_STR2(x)
// what this translation unit was built with (chv_build_flags; a timing-only CHV_ABL build must never ship)
#if CHV_WAVE_TU == 0
const char *bgra_wave_build_flags() { return "tick_bgra_wave:abl=" CHV_STR(CHV_ABL) ",waves_per_block=" CHV_STR(CHV_WAVE_WAVES) ",strip_rows=8|16"; }
#endif

template <bool CACHED>
hipError_t launch_bgra_wave_t(int rows, bool clear, dim3 grid, size_t lds, hipStream_t stream, const DTick *ticks, const DLayer *layers, int n_ticks,
                              int strips_x, int strips_y, uint32_t strips_magic, uint32_t strips_x_magic, int p0pitch, int p0rows, int p1pitch, int p1rows, int planar, int kinds) {
#define CHV_LAUNCH_B(R, C, K) hipLaunchKernelGGL((tick_bgra_wave<R, C, K, CACHED>), grid, dim3(WAVE_BLOCK), lds, stream, ticks, layers, n_ticks, strips_x, strips_y, \
                                                 strips_magic, strips_x_magic, p0pitch, p0rows, p1pitch, p1rows, planar)
    /* (launches of RGB layers only — cfg3, cfg5: stacks of ONE geometry, set up once per strip for all layers, power-bound — gain nothing from
       tables and would pay for them in counted traffic: launch_wave_layers never asks for the CACHED form of KINDS = 4, and it is not built) */
#define CHV_LAUNCH_BK(R, C) do { if (kinds == 1) CHV_LAUNCH_B(R, C, 1); else if (kinds == 2) CHV_LAUNCH_B(R, C, 2); \
                                 else if (kinds == 4) { if constexpr (!CACHED) CHV_LAUNCH_B(R, C, 4); } else if (kinds == 5) CHV_LAUNCH_B(R, C, 5); \
                                 else if (kinds & 8) CHV_LAUNCH_B(R, C, 15); \
                                 else CHV_LAUNCH_B(R, C, 7); } while (0)      /* (y420p + RGB alone: its instantiation spills, 7 does not) */
    if (rows == 16) { if (clear) CHV_LAUNCH_BK(16, true); else CHV_LAUNCH_BK(16, false); }
    else            { if (clear) CHV_LAUNCH_BK(8, true); else CHV_LAUNCH_BK(8, false); }
#undef CHV_LAUNCH_BK
#undef CHV_LAUNCH_B
    return hipGetLastError();
}

#define CHV_BGRA_WAVE_ARGS int rows, bool clear, dim3 grid, size_t lds, hipStream_t stream, const DTick *ticks, const DLayer *layers, int n_ticks, \
                           int strips_x, int strips_y, uint32_t strips_magic, uint32_t strips_x_magic, int p0pitch, int p0rows, int p1pitch, int p1rows, int planar, int kinds
#if CHV_WAVE_TU == 0
template hipError_t launch_bgra_wave_t<false>(CHV_BGRA_WAVE_ARGS);
extern template hipError_t launch_bgra_wave_t<true>(CHV_BGRA_WAVE_ARGS);            // kernels_wave_cached.hip.cpp

hipError_t launch_bgra_wave(int rows, bool clear, dim3 grid, size_t lds, hipStream_t stream, const DTick *ticks, const DLayer *layers, int n_ticks,
                            int strips_x, int strips_y, uint32_t strips_magic, uint32_t strips_x_magic, int p0pitch, int p0rows, int p1pitch, int p1rows, int planar, int kinds, bool cached) {
    // bit 5 of `planar`: rectangles are staged by DMA where their shape allows (the RGB-only instantiation)
    {
        const int dma_on = switches().wave_dma.load(std::memory_order_relaxed);        // (CHV_WAVE_DMA=0 / chv_debug_set_switch: register staging)
        if (CHV_WAVE_DMA && dma_on && (kinds == 4 || (CHV_WAVE_DMA > 1 && (kinds & 4)))) planar |= 32;
    }
    return cached ? launch_bgra_wave_t<true>(rows, clear, grid, lds, stream, ticks, layers, n_ticks, strips_x, strips_y, strips_magic, strips_x_magic, p0pitch, p0rows, p1pitch, p1rows, planar, kinds)
                  : launch_bgra_wave_t<false>(rows, clear, grid, lds, stream, ticks, layers, n_ticks, strips_x, strips_y, strips_magic, strips_x_magic, p0pitch, p0rows, p1pitch, p1rows, planar, kinds);
}
#else
template hipError_t launch_bgra_wave_t<true>(CHV_BGRA_WAVE_ARGS);
#endif

}  // namespace chv

// kernels_wave_yuv.hip.cpp — axis-aligned tick kernel for 4:2:0 canvases (NV12, y420p), one wave per canvas strip.
//
// These are the reference's own kernels — img_nv12_nv12, img_y420p_nv12, img_y420p_y420p, img_{bgra,rgba}_{nv12,y420p}
// (kernels.cl.swift:47-532) — and the default Linux canvas is 4:2:0 (composer.swift:52-56; x264 wants y420p,
// enc.video.ffmpeg.swift:211-224).  The general quad kernel (kernels_general.hip.cpp) evaluates the geometry per pixel and
// gathers every tap from global memory: ~260 VALU instructions per canvas pixel, HBM fetch 2.4x the algorithmic bytes
// (profiles/r01_notes.md).  Here the structure of kernels_wave.hip.cpp is applied to the reference's unit-scale arithmetic:
//   * lane = canvas column, a wave owns 64 columns x WTH rows; luma codes of the lane's pixels packed four to a register,
//     the chroma sample of every 2x2 quad held by the quad's even/even pixel — the reference's `handleChroma` owner
//     (kernels.cl.swift:76) — so even lanes carry (u, v) codes for the even rows;
//   * column entries in registers, row entries in the wave's LDS table, source rectangles staged wave-privately as bytes
//     (wave_common.hip.h); strips of one frame on one XCD;
//   * UNORM8 loads are c / 255 correctly rounded (OpenCL 1.2 section 8.3.1.1; two-term product, pixel_math.hip.h), every sum
//     keeps the reference's order and roundings (no contraction);
//   * strips that lie entirely inside a layer's picture run branch-free; every other strip (picture or border edges, fill
//     paint, unstaged rectangles) applies the layer pixel by pixel with the general kernel's own code (yuv_pixel.hip.h).
// Bytes: those of kernels_general.hip.cpp = oracle/ref_kernels.c (px_yuv_to_yuv, px_rgb_to_yuv), layer by layer.
#include "wave_common.hip.h"
#include "yuv_pixel.hip.h"
#include "switches.h"
#include "geom_cache.h"
// 0: this translation unit — the kernels that compute their geometry, eligibility, the launcher, the geometry tables' builder; 1:
// kernels_wave_yuv_cached.hip.cpp — the instantiations that read their geometry from a batch's tables, nothing else
#ifndef CHV_WAVE_TU
#define CHV_WAVE_TU 0
#endif

#include <atomic>
#include <map>
#include <unordered_map>
#include <mutex>
#include <string>
#include <vector>

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <type_traits>

// CHV_ABL: timing-only ablations (results are wrong): 1 = no staging, 2 = no pixel rows, 4 = no canvas stores
#ifndef CHV_ABL
#define CHV_ABL 0
#endif
#ifndef CHV_WAVE_PRIO
#define CHV_WAVE_PRIO 1
#endif
#pragma clang fp contract(off)

namespace chv {

// UNORM8 loads: c / 255.0f, correctly rounded — the two-term product of pixel_math.hip.h (cvt + mul + fma per byte; a single multiply is wrong
// for 126 of 256 codes).  A 256-entry LDS table in their place was measured in rounds 2 and 6 and is gone from the source
// (profiles/r06_unorm_table_experiment.patch, r06_notes.md section 3): it removes 8 % of the launch's vector instructions and is 6 % SLOWER on
// conflict-free content (gradients: 4.9 M bank-conflict cycles), 6-7 % on low-pass noise (34 M), 6-10 % on random bytes (121 M) — an LDS read
// occupies the CU's one LDS pipe for as long as one of its four SIMDs would have spent on the arithmetic.
CHV_DEV float T8(uint32_t byte) { return unorm8(byte); }
// the same for byte K of a packed word
template <int K>
CHV_DEV float T8k(uint32_t w) {
    return unorm8f(K == 0 ? (float)(w & 255u) : K == 1 ? (float)((w >> 8) & 255u) : K == 2 ? (float)((w >> 16) & 255u) : (float)(w >> 24));
}
// convert_uchar_sat_rte(f * 255) into byte K of w (v_cvt_pk_u8_f32: RTE, clamp to [0, 255], NaN -> 0 = to_code)
template <int K>
CHV_DEV uint32_t put_code(uint32_t w, float f) {
    const float v = f * 255.0f;
    if (K == 0) asm("v_cvt_pk_u8_f32 %0, %1, 0, %0" : "+v"(w) : "v"(v));
    if (K == 1) asm("v_cvt_pk_u8_f32 %0, %1, 1, %0" : "+v"(w) : "v"(v));
    if (K == 2) asm("v_cvt_pk_u8_f32 %0, %1, 2, %0" : "+v"(w) : "v"(v));
    if (K == 3) asm("v_cvt_pk_u8_f32 %0, %1, 3, %0" : "+v"(w) : "v"(v));
    return w;
}
// row(integral_constant<int, j>) for j = 0 .. YTH - 1, in order (the row index is a compile-time constant in the body: byte
// positions of the packed canvas codes are immediates)
template <typename F, int... J>
CHV_DEV void for_each_row_impl(F &f, std::integer_sequence<int, J...>) { (f(std::integral_constant<int, J>{}), ...); }
template <int N, typename F>
CHV_DEV void for_rows(F &f) { for_each_row_impl(f, std::make_integer_sequence<int, N>{}); }

// code-scale variants (the integer-matrix RGB kind): byte K of a word as a float; to_code_raw of a code-scale value into byte K
template <int K>
CHV_DEV float ubk(uint32_t w) { return K == 0 ? ub0(w) : K == 1 ? ub1(w) : K == 2 ? ub2(w) : ub3(w); }
template <int K>
CHV_DEV uint32_t put_code_raw(uint32_t w, float v) {
    if (K == 0) asm("v_cvt_pk_u8_f32 %0, %1, 0, %0" : "+v"(w) : "v"(v));
    if (K == 1) asm("v_cvt_pk_u8_f32 %0, %1, 1, %0" : "+v"(w) : "v"(v));
    if (K == 2) asm("v_cvt_pk_u8_f32 %0, %1, 2, %0" : "+v"(w) : "v"(v));
    if (K == 3) asm("v_cvt_pk_u8_f32 %0, %1, 3, %0" : "+v"(w) : "v"(v));
    return w;
}
CHV_DEV float mix4(float w00, float w10, float w01, float w11, float t00, float t10, float t01, float t11) {
    return ((w00 * t00 + w10 * t10) + w01 * t01) + w11 * t11;      // lin_mix's order (OpenCL 1.2 section 8.2)
}

#ifndef CHV_WAVEY_CARRY
#define CHV_WAVEY_CARRY 1
#endif
// 1: uncleared launches cover the strips their layers' bounding boxes touch, not the canvas (launch_wave_layers); 0: the A/B
#ifndef CHV_WAVE_BBOX_GRID
#define CHV_WAVE_BBOX_GRID 1
#endif
// (bit 2: narrow interior YUV rectangles through the shift-and-mask slot map, wstage_load_p2; bit 3 — skip its unneeded rounds —
// measured and left off: mixer_y420p 0.649 -> 0.663 ms, mixer_nv12 0.637 -> 0.664 with it)
#ifndef CHV_WAVEY_INTERIOR
#define CHV_WAVEY_INTERIOR 5
#endif
#ifndef CHV_WAVEY_WIDE_STORES
#define CHV_WAVEY_WIDE_STORES 1
#endif
#ifndef CHV_WAVEY_INTERIOR_OWN
#define CHV_WAVEY_INTERIOR_OWN 4
#endif
// The instantiations that also carry the RGB-overlay rows run 5 waves per SIMD (96 VGPRs, nothing in scratch): at 6 waves they
// kept five or six registers in scratch, which showed as 1.5x write traffic (profiles/r02_mixer_y420p_rocprofv3.txt) and was no
// faster (mixer_nv12 0.741 -> 0.690 ms, mixer_y420p 0.787 -> 0.767 with 5; the own-format instantiation stays at 6: 0.410 vs 0.430)
#ifndef CHV_WAVEY_MINW_MIXED
#define CHV_WAVEY_MINW_MIXED 5
#endif
#ifndef CHV_WAVEY_MINW
#define CHV_WAVEY_MINW 6
#endif
// Strip height YTH on 4:2:0 canvases, a template parameter picked per launch (launch_wave_layers).  A strip's fixed costs
// (index arithmetic, per-layer geometry, staging, stores: ~430 VALU instructions) exceed the pixel work of one opaque layer
// over 64 x 8 pixels (~300), and the canvas codes pack four to a register, so 16 rows cost only 4 more registers: 16-row
// strips execute 12 % fewer instructions per pixel.  They were SLOWER (1.12 vs 0.86 ms per 128 ticks of y420p_main) as long
// as strips crossed by an overlay's edge took the per-pixel path and the build dropped to 5 waves per SIMD; with the masked
// rows and 6 waves (80 VGPRs, 9 spilled) they are faster everywhere: y420p_main 0.668 -> 0.495 ms, mixer_y420p 0.863 ->
// 0.829, mixer_nv12 0.828 -> 0.792 (profiles/r02_notes.md).  8-row strips remain for small launches (more waves) and for
// source rectangles whose 16-row version would not leave room for two strips per 64 KB of LDS.
// KINDS: source classes in the launch (wave_common.hip.h): 1 / 2 = pictures of the canvas' own format only (NV12 / y420p), 5 / 6 = those
// plus RGB overlays (the reference's usual mixer), 7 = any
// CACHED: the per-layer geometry comes from the batch's tables (WaveStrip::setup_cached) — no set-up code in these instantiations, which are
// compiled in kernels_wave_yuv_cached.hip.cpp and launched for batches whose staged layers all have tables
template <int TF, bool CLEAR, int YTH, int KINDS, bool CACHED>
__global__ __launch_bounds__(WAVE_BLOCK, ((KINDS == 1 || KINDS == 2) ? CHV_WAVEY_MINW : CHV_WAVEY_MINW_MIXED)) void tick_yuv_wave(const DTick *__restrict__ ticks,
                                                                         const DLayer *__restrict__ layers,
                                                                         int n_ticks, int strips_x, int strips_y, uint32_t strips_magic, uint32_t strips_x_magic,
                                                                         int p0pitch, int p0rows, int p1pitch, int p1rows, int planar_any, const WaveOne one) {
    wave_one_descriptors(ticks, layers);
    constexpr int YLW = YTH / 4;                     // registers holding the lane's luma codes (4 rows per register)
    extern __shared__ __attribute__((aligned(16))) uint8_t smem_all[];
    // (compact staging of interior rectangles: measured better for the mixed-class instantiations, worse for the own-format one)
    using Strip = WaveStrip<YTH, (KINDS == 1 || KINDS == 2) ? CHV_WAVEY_INTERIOR_OWN : CHV_WAVEY_INTERIOR, KINDS>;
    Strip S;
    if (!S.init(ticks, layers, n_ticks, strips_x, strips_y, strips_magic, strips_x_magic, smem_all, p0pitch, p0rows, p1pitch, p1rows, planar_any)) return;
    p1pitch = S.p1pitch;                 // (the side-by-side layout keeps chroma in the rows of the plane-0 region: WaveStrip::init)
    const DTick &T = *S.T;
    const DLayer *L = S.L;
    const int nl = S.nl, x = S.x, x0 = S.x0, y0 = S.y0;
    const bool col_in = S.col_in;
    const uint8_t *smem = S.smem;
    const uint4 *rowtab = S.rowtab;
    const int voff = S.voff;
    // canvas planes BY VALUE, read once: the stores at the end go through integer-cast pointers, after the first of them the
    // compiler treats the descriptors as possibly overwritten and would re-read pitch and height before every row's store
    // with a vector load and a full `s_waitcnt vmcnt(0)` (which also drains the previous store)
    const DPlane PY = T.dst.pl[0], PC = T.dst.pl[1], PV = T.dst.pl[TF == TF_Y420P ? 2 : 1];
    const int TH = T.H;
    const float sx = S.sx, sy = S.sy;

    // ---- canvas codes of this lane: luma row j in byte j & 3 of ly[j >> 2].  Chroma: the two lanes of a column pair (2k, 2k + 1)
    //      share quad column k; the even lane holds the strip's EVEN chroma rows, the odd lane the ODD ones — chroma row
    //      jj = 2 m + (lane & 1) (canvas chroma row y0 / 2 + jj) in byte m of nu / nv.  The reference evaluates a quad's chroma at
    //      its even/even pixel (`handleChroma`, kernels.cl.swift:76): with the chroma rows dealt out like this every lane works
    //      on one (at the even lane's column entry and the even row's row entry) instead of half the lanes on two. -------------
    constexpr int YCM = YTH / 4;                             // chroma rows per lane
    const int par = x & 1;                                   // (x0 is a multiple of 64: lane parity = column parity)
    const bool owner_lane = par == 0 && col_in;              // (canvas sizes are even on this path: host-checked)
    const int qx = x >> 1, qy0 = y0 >> 1;
    uint32_t ly[YLW], nu = 0x80808080u, nv = 0x80808080u;                  // chroma = 0.5 -> 128 (RTE)
#pragma unroll
    for (int k = 0; k < YLW; k++) ly[k] = 0;                               // img_clear_*: Y = 0.0
    // per-lane byte offsets of this lane's chroma rows relative to the (uniform) base of chroma row qy0 + 2 m
    const uint32_t coff_c = (uint32_t)par * (uint32_t)PC.pitch + (TF == TF_NV12 ? (uint32_t)qx * 2u : (uint32_t)qx);
    const uint32_t coff_v = (uint32_t)par * (uint32_t)PV.pitch + (uint32_t)qx;
    if (!CLEAR && col_in) {
#pragma unroll
        for (int j = 0; j < YTH; j++) {
            if (y0 + j < TH) ly[j >> 2] |= (uint32_t)gld_at<uint8_t>(PY.ptr + (size_t)(y0 + j) * PY.pitch, (uint32_t)x) << (8 * (j & 3));
        }
        nu = 0; nv = 0;
#pragma unroll
        for (int m = 0; m < YCM; m++) {
            // (uniform conditions only: chroma rows past the canvas are read from its last row instead of skipped per lane; they are
            // never stored)
            const int qr = min(qy0 + 2 * m, (TH >> 1) - 1);
            const bool odd_row = qy0 + 2 * m + 1 < (TH >> 1);
            const uint32_t oc = odd_row ? coff_c : coff_c - (uint32_t)par * (uint32_t)PC.pitch;
            uint32_t ub, vb;
            if (TF == TF_NV12) {
                const uint32_t p = gld_at<uint16_t>(PC.ptr + (size_t)qr * PC.pitch, oc);
                ub = p & 255u; vb = p >> 8;
            } else {
                const uint32_t ov = odd_row ? coff_v : coff_v - (uint32_t)par * (uint32_t)PV.pitch;
                ub = gld_at<uint8_t>(PC.ptr + (size_t)qr * PC.pitch, oc);
                vb = gld_at<uint8_t>(PV.ptr + (size_t)qr * PV.pitch, ov);
            }
            nu |= ub << (8 * m); nv |= vb << (8 * m);
        }
    }

    WLayer cur;
    bool have_geom = false;            // `cur` and the row table hold the geometry of the layer handled just before (LF_SAME_GEOM)
    // (the layer index is wave-uniform; saying so keeps the descriptor reads on the scalar unit: left to its divergence analysis
    // the compiler fetched every uniform of a layer with per-lane global loads — 90 vector loads per wave)
    int l = __builtin_amdgcn_readfirstlane(S.next_hit(0));
    while (l < nl) {
        const DLayer &Ly = L[l];
        // Issue priority for the latency-bound phases (geometry, staging): a wave in them has few instructions to issue and long
        // waits between them, so letting it go first whenever it can shortens its chain, and more of the resident waves are in
        // their row loops at any time (pipeline -1.4 %, cfg3 -2.2 %).  Through non-volatile asm with a token operand: the
        // __builtin_amdgcn_s_setprio call counts as a side effect after which hipcc reads the descriptors per lane (+44 %).
        // layers the strip machinery cannot stage (rotation, shear, unbounded matrices; the launch has some: KINDS bit 3): straight to
        // the per-pixel path below, no geometry tables, no staging
        bool general_layer = false;
        if constexpr ((KINDS & 8) != 0) general_layer = (Ly.flags & (LF_AXIS_ALIGNED | LF_BOUNDED)) != (LF_AXIS_ALIGNED | LF_BOUNDED);
        int ptok = l;
        if (CHV_WAVE_PRIO) asm("s_setprio 3" : "+s"(ptok));
        // (LF_SAME_GEOM: geometry inputs bit-identical to the predecessor's — its column entry, row table and rectangles stand)
        // (setup overwrites the row table: the previous layer's pixels are done.  From the batch's geometry table where there is one — setup_cached)
        if (!general_layer) { if (!(have_geom && (Ly.flags & LF_SAME_GEOM))) { if constexpr (CACHED) S.setup_cached(l, cur); else S.setup(ptok, cur); } have_geom = true; }
        else { have_geom = false; cur.staged = false; cur.all_inside = false; cur.unit_rows = false; cur.cfl = 0; cur.cyo = 0; cur.cco = 0; cur.cya = 0.f; cur.cca = 0.f; }
        // Strips entirely inside the picture, and — when the layer paints no fill (alpha of opacity x fill exactly 0: pixels of
        // the border quad outside the picture then keep their codes, to_code(c / 255) = c) — strips a picture edge crosses as
        // well: rows outside the picture are skipped (uniform branch), lanes outside keep their codes.
        const bool nofill = (Ly.flags & LF_NO_FILL) != 0;
#ifndef CHV_WAVEY_MASKED
#define CHV_WAVEY_MASKED 1
#endif
#ifndef CHV_WAVEY_INT_ROWS
#define CHV_WAVEY_INT_ROWS 1        // A/B: 0 sends the integer-matrix RGB kind through the per-pixel path (taps from global memory)
#endif
        const bool fast = !general_layer && cur.staged && (cur.all_inside || (CHV_WAVEY_MASKED && nofill)) && (CHV_WAVEY_INT_ROWS || Ly.kind != LK_YUV_FROM_RGB_INT);
        const bool lane_pic = cur.cfl == AX_ALL;
        // a pixel takes a row's result if its column and the row are inside the picture (row flags: uniform, from the row table);
        // every row is computed (branch-free: a branch per row keeps the rows' LDS reads from overlapping), row offsets are
        // clamped into the staged rectangle
        auto take = [&](uint32_t rfl) { return lane_pic && rfl == (uint32_t)AX_ALL; };
        if (fast && !(CHV_ABL & 1)) S.stage(l, cur);
        wave_lds_fence();
        const int ln = __builtin_amdgcn_readfirstlane(S.next_hit(l + 1));
        if (CHV_WAVE_PRIO) { asm("s_setprio 0" : "+s"(ptok)); cur.cyo += ptok - l; }       // (ptok - l = 0, opaque: pins the asm here)
        const float *U = Ly.u;

        if (CHV_ABL & 2) ly[0] += (uint32_t)(cur.cyo ^ cur.cco ^ __float_as_int(cur.cya) ^ __float_as_int(cur.cca) ^ cur.cfl);
        else if (fast && !Strip::is_rgb(Ly.kind)) {
            // ---- YUV picture over the whole strip (kernels.cl.swift:78-94): cur * (1 - opacity) + sample * opacity ----
            const float alpha = U[U_OPACITY], ialpha = 1.f - alpha;
            const float a = cur.cya, ia = 1.0f - a;
            // the even lane's column entry in both lanes of a pair (quad_perm [0, 0, 2, 2])
            const int cco_q = __builtin_amdgcn_update_dpp(cur.cco, cur.cco, 0xA0, 0xf, 0xf, false);
            const float cca_q = __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(cur.cca), __float_as_int(cur.cca), 0xA0, 0xf, 0xf, false));
            const float icaq = 1.0f - cca_q;
            const int picw = lane_pic && col_in ? 1 : 0;
            const bool pic_q = __builtin_amdgcn_update_dpp(picw, picw, 0xA0, 0xf, 0xf, false) != 0;
            auto body = [&](auto planar_c, auto opaque_c) {
                constexpr bool PL = decltype(planar_c)::value, OP = decltype(opaque_c)::value;
                // (unit_rows: native-resolution layers — the lower luma tap row of a pixel is the upper one of the pixel below, its
                // two UNORM8 conversions, three instructions each, are carried down the lane)
                const bool carry = CHV_WAVEY_CARRY && cur.unit_rows;
                float t0 = 0.f, t1 = 0.f;
                if (carry) { const uint8_t *py = smem + (row_fast<YTH, false>(rowtab, 0).yoff + cur.cyo); t0 = T8(py[0]); t1 = T8(py[1]); }
                auto row = [&](auto jc, auto carry_c) {
                    constexpr int j = decltype(jc)::value;
                    constexpr bool CARRY = decltype(carry_c)::value;
                    const RowFast rw = row_fast<YTH, false>(rowtab, j);       // (compact row entries: wave_common.hip.h)
                    const bool tk = take(row_fast_flags<YTH>(rowtab, j));
                    const float b = rw.yb, ib = rw.iyb;
                    const uint8_t *py = smem + (rw.yoff + cur.cyo);
                    float luma;
                    if constexpr (CARRY) {
                        const float b0 = T8(py[p0pitch]), b1 = T8(py[p0pitch + 1]);
                        luma = mix4(ia * ib, a * ib, ia * b, a * b, t0, t1, b0, b1);
                        t0 = b0; t1 = b1;
                    } else {
                        luma = mix4(ia * ib, a * ib, ia * b, a * b, T8(py[0]), T8(py[1]), T8(py[p0pitch]), T8(py[p0pitch + 1]));
                    }
                    uint32_t &lw = ly[j >> 2];
                    // opacity == 1: cur * 0 + luma * 1 = luma exactly
                    const float v = OP ? luma : T8k<j & 3>(lw) * ialpha + luma * alpha;
                    const uint32_t nlw = put_code<j & 3>(lw, v);
                    lw = tk ? nlw : lw;
                    if constexpr ((j & 3) == 0) {
                        // chroma rows 2 m (even lanes) and 2 m + 1 (odd lanes), m = j / 4: sampled at the uv of the quad's even/even
                        // pixel — the even lane's column entry (cco_q, cca_q), the row entry of luma row 4 m + 2 par — on the
                        // half-size plane(s)
                        constexpr int m = j >> 2;
                        const RowFast qw = row_fast<YTH, true>(rowtab, 4 * m + 2 * par);
                        const bool tkc = pic_q && row_fast_flags<YTH>(rowtab, 4 * m + 2 * par) == (uint32_t)AX_ALL;
                        const float cbw = qw.cb, icb = qw.icb;
                        const uint8_t *pc = smem + (qw.coff + cco_q);
                        const float c00 = icaq * icb, c10 = cca_q * icb, c01 = icaq * cbw, c11 = cca_q * cbw;
                        float fu, fv;
                        if constexpr (PL) {
                            fu = mix4(c00, c10, c01, c11, T8(pc[0]), T8(pc[1]), T8(pc[p1pitch]), T8(pc[p1pitch + 1]));
                            const uint8_t *pv = pc + voff;
                            fv = mix4(c00, c10, c01, c11, T8(pv[0]), T8(pv[1]), T8(pv[p1pitch]), T8(pv[p1pitch + 1]));
                        } else {
                            fu = mix4(c00, c10, c01, c11, T8(pc[0]), T8(pc[2]), T8(pc[p1pitch]), T8(pc[p1pitch + 2]));
                            fv = mix4(c00, c10, c01, c11, T8(pc[1]), T8(pc[3]), T8(pc[p1pitch + 1]), T8(pc[p1pitch + 3]));
                        }
                        const uint32_t nnu = put_code<m>(nu, OP ? fu : T8k<m>(nu) * ialpha + fu * alpha);
                        const uint32_t nnv = put_code<m>(nv, OP ? fv : T8k<m>(nv) * ialpha + fv * alpha);
                        nu = tkc ? nnu : nu; nv = tkc ? nnv : nv;
                    }
                };
                if (carry) { auto r = [&](auto jc) { row(jc, std::true_type{}); }; for_rows<YTH>(r); }
                else { auto r = [&](auto jc) { row(jc, std::false_type{}); }; for_rows<YTH>(r); }
            };
            const bool planar = Strip::is_planar(Ly.kind), opaque = (Ly.flags & LF_OPAQUE) != 0;
            if (planar) { if (opaque) body(std::true_type{}, std::true_type{}); else body(std::true_type{}, std::false_type{}); }
            else        { if (opaque) body(std::false_type{}, std::true_type{}); else body(std::false_type{}, std::false_type{}); }
        } else if (fast && Ly.kind == LK_YUV_FROM_RGB_INT) {
            // ---- RGB picture, integer BT.601 / 709 matrix (img_*_int; statement of record: yuv_pixel.hip.h::
            //      apply_yuv_from_rgb_int = oracle px_rgb_to_yuv_int): everything on the 0..255 code scale — bilinear sample with
            //      fused multiply-adds, the sample rounded to codes, the 16.16 matrix on the codes, canvas blended by alpha x opacity,
            //      fill painted first.  The encoder side's full-frame BGRA -> NV12 / y420p conversion runs here. ----
            const R2Y &k = kR2Y[Ly.csc & 3];
            const float opacity = U[U_OPACITY];
            const float ka = opacity * kInv255;
            const float af = opacity * U[U_FILL + 3], iaf = 1.f - af;
            uint32_t fyc, fuc, fvc;
            rgb_to_yuv_int(k, (int)to_code_raw(U[U_FILL + 0] * 255.0f), (int)to_code_raw(U[U_FILL + 1] * 255.0f), (int)to_code_raw(U[U_FILL + 2] * 255.0f), fyc, fuc, fvc);
            const float fyf = (float)fyc, fuf = (float)fuc, fvf = (float)fvc;
            const float a = cur.cya, ia = 1.0f - a;
            // native-resolution pictures (source rows advance one per canvas row: the encoder-side conversion): a pixel's lower tap row
            // is the upper tap row of the pixel below — its eight conversions and two LDS reads are carried down the lane
            // (measured: encode_nv12 0.638 -> 0.622 ms per 128 frames, but the eight carried registers put 2-6 registers of the 16-row
            // mixed-class instantiations in scratch and the mixer workloads lose 2 %: off)
#ifndef CHV_WAVEY_INT_CARRY
#define CHV_WAVEY_INT_CARRY 0
#endif
            const bool carry = CHV_WAVEY_INT_CARRY && cur.unit_rows;
            float t00 = 0.f, t01 = 0.f, t02 = 0.f, t03 = 0.f, t10 = 0.f, t11 = 0.f, t12 = 0.f, t13 = 0.f;
            if (carry) {
                const uint8_t *p0 = smem + (row_fast<YTH, false>(rowtab, 0).yoff + cur.cyo);
                const uint32_t u00 = ((const uint32_t *)p0)[0], u10 = ((const uint32_t *)p0)[1];
                t00 = ub0(u00); t01 = ub1(u00); t02 = ub2(u00); t03 = ub3(u00);
                t10 = ub0(u10); t11 = ub1(u10); t12 = ub2(u10); t13 = ub3(u10);
            }
            auto int_rows = [&](auto fill_c, auto carry_c) {
                constexpr bool FILL = decltype(fill_c)::value;      // !FILL: clamp(fma(f, 0, c * 1), 0, 255) = c for a code c
                constexpr bool CARRY = decltype(carry_c)::value;
                auto row = [&](auto jc) {
                    constexpr int j = decltype(jc)::value;
                    const RowFast rw = row_fast<YTH, false>(rowtab, j);
                    const bool tk = take(row_fast_flags<YTH>(rowtab, j));
                    const float b = rw.yb, ib = rw.iyb;
                    const uint8_t *p0 = smem + (rw.yoff + cur.cyo);
                    const uint32_t u01 = ((const uint32_t *)(p0 + p0pitch))[0], u11 = ((const uint32_t *)(p0 + p0pitch))[1];
                    const float w00 = ia * ib, w10 = a * ib, w01 = ia * b, w11 = a * b;
                    // staged texels are R, G, B, A whatever the source order
                    const float b00 = ub0(u01), b01 = ub1(u01), b02 = ub2(u01), b03 = ub3(u01);
                    const float b10 = ub0(u11), b11 = ub1(u11), b12 = ub2(u11), b13 = ub3(u11);
                    if constexpr (!CARRY) {
                        const uint32_t u00 = ((const uint32_t *)p0)[0], u10 = ((const uint32_t *)p0)[1];
                        t00 = ub0(u00); t01 = ub1(u00); t02 = ub2(u00); t03 = ub3(u00);
                        t10 = ub0(u10); t11 = ub1(u10); t12 = ub2(u10); t13 = ub3(u10);
                    }
                    const float q0 = cs_mix(w00, w10, w01, w11, t00, t10, b00, b10);
                    const float q1 = cs_mix(w00, w10, w01, w11, t01, t11, b01, b11);
                    const float q2 = cs_mix(w00, w10, w01, w11, t02, t12, b02, b12);
                    const float q3 = cs_mix(w00, w10, w01, w11, t03, t13, b03, b13);
                    if constexpr (CARRY) { t00 = b00; t01 = b01; t02 = b02; t03 = b03; t10 = b10; t11 = b11; t12 = b12; t13 = b13; }
                    // to_code_raw of a convex combination of codes: no clamp can trigger; rint through the float adder
                    // (the multiplier's operands keep the adder's bias: r2y_base_biased, pixel_math.hip.h)
                    const int cr = (int)code_biased(q0), cg = (int)code_biased(q1), cb = (int)code_biased(q2);
                    const float a2 = q3 * ka, ia2 = 1.f - a2;
                    const float py = fixed_to_codef(r2y_row(k.y[0], k.y[1], k.y[2], r2y_base_biased(k.y[0], k.y[1], k.y[2], (k.yoff << 16) + 32768), cr, cg, cb));
                    uint32_t &lw = ly[j >> 2];
                    const float cyf = ubk<j & 3>(lw);
                    const float r0 = FILL ? clampf(__builtin_fmaf(fyf, af, cyf * iaf), 0.f, 255.f) : cyf;
                    const uint32_t nlw = put_code_raw<j & 3>(lw, __builtin_fmaf(py, a2, r0 * ia2));
                    lw = tk ? nlw : lw;
                    if constexpr ((j & 1) == 0) {
                        // chroma of the quad: the even lane's pixel of this (even) row; chroma row jj = j / 2 lives in the even lane
                        // (jj even) or in its odd neighbour (jj odd: the values travel one lane up, quad_perm [0, 0, 2, 2])
                        constexpr int jj = j >> 1, m = jj >> 1;
                        float pu = fixed_to_codef(r2y_row(k.u[0], k.u[1], k.u[2], r2y_base_biased(k.u[0], k.u[1], k.u[2], (128 << 16) + 32768), cr, cg, cb));
                        float pv = fixed_to_codef(r2y_row(k.v[0], k.v[1], k.v[2], r2y_base_biased(k.v[0], k.v[1], k.v[2], (128 << 16) + 32768), cr, cg, cb));
                        float sa = a2, sia = ia2;
                        int stk = (tk && owner_lane) ? 1 : 0;
                        if constexpr ((jj & 1) != 0) {
                            pu = __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(pu), __float_as_int(pu), 0xA0, 0xf, 0xf, false));
                            pv = __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(pv), __float_as_int(pv), 0xA0, 0xf, 0xf, false));
                            sa = __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(a2), __float_as_int(a2), 0xA0, 0xf, 0xf, false));
                            sia = __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(ia2), __float_as_int(ia2), 0xA0, 0xf, 0xf, false));
                            stk = __builtin_amdgcn_update_dpp(stk, stk, 0xA0, 0xf, 0xf, false);
                        }
                        const bool mine = stk != 0 && par == (jj & 1);
                        const float cuf = ubk<m>(nu), cvf = ubk<m>(nv);
                        const float r1 = FILL ? clampf(__builtin_fmaf(fuf, af, cuf * iaf), 0.f, 255.f) : cuf;
                        const float r2 = FILL ? clampf(__builtin_fmaf(fvf, af, cvf * iaf), 0.f, 255.f) : cvf;
                        const uint32_t nnu = put_code_raw<m>(nu, __builtin_fmaf(pu, sa, r1 * sia));
                        const uint32_t nnv = put_code_raw<m>(nv, __builtin_fmaf(pv, sa, r2 * sia));
                        nu = mine ? nnu : nu; nv = mine ? nnv : nv;
                    }
                };
                for_rows<YTH>(row);
            };
            if (carry) { if (nofill) int_rows(std::false_type{}, std::true_type{}); else int_rows(std::true_type{}, std::true_type{}); }
            else { if (nofill) int_rows(std::false_type{}, std::false_type{}); else int_rows(std::true_type{}, std::false_type{}); }
        } else if (fast) {
            // ---- RGB picture over the whole strip (kernels.cl.swift:509-529): fill pre-blend, sample, rgb2yuv of the
            //      pre-multiplied pixel, blend by alpha x opacity; staged texels are R,G,B,A whatever the source order ----
            const float opacity = U[U_OPACITY];
            const float af = opacity * U[U_FILL + 3], iaf = 1.f - af;
            float fy, fu, fv;
            rgb2yuv(U[U_FILL + 0] * af, U[U_FILL + 1] * af, U[U_FILL + 2] * af, fy, fu, fv);
            const float fya = fy * af, fua = fu * af, fva = fv * af;
            const float a = cur.cya, ia = 1.0f - a;
            const bool carry = CHV_WAVEY_CARRY && cur.unit_rows;       // (as above: eight conversions per row carried down the lane)
            float t00 = 0.f, t01 = 0.f, t02 = 0.f, t03 = 0.f, t10 = 0.f, t11 = 0.f, t12 = 0.f, t13 = 0.f;
            if (carry) {
                const uint8_t *p0 = smem + (row_fast<YTH, false>(rowtab, 0).yoff + cur.cyo);
                const uint32_t u00 = ((const uint32_t *)p0)[0], u10 = ((const uint32_t *)p0)[1];
                t00 = T8k<0>(u00); t01 = T8k<1>(u00); t02 = T8k<2>(u00); t03 = T8k<3>(u00);
                t10 = T8k<0>(u10); t11 = T8k<1>(u10); t12 = T8k<2>(u10); t13 = T8k<3>(u10);
            }
            auto row = [&](auto jc, auto carry_c) {
                constexpr int j = decltype(jc)::value;
                constexpr bool CARRY = decltype(carry_c)::value;
                const RowFast rw = row_fast<YTH, false>(rowtab, j);
                const bool tk = take(row_fast_flags<YTH>(rowtab, j));
                const float b = rw.yb, ib = rw.iyb;
                const uint8_t *p0 = smem + (rw.yoff + cur.cyo);
                const uint32_t u01 = ((const uint32_t *)(p0 + p0pitch))[0], u11 = ((const uint32_t *)(p0 + p0pitch))[1];
                const float w00 = ia * ib, w10 = a * ib, w01 = ia * b, w11 = a * b;
                const float b00 = T8k<0>(u01), b01 = T8k<1>(u01), b02 = T8k<2>(u01), b03 = T8k<3>(u01);
                const float b10 = T8k<0>(u11), b11 = T8k<1>(u11), b12 = T8k<2>(u11), b13 = T8k<3>(u11);
                if constexpr (!CARRY) {
                    const uint32_t u00 = ((const uint32_t *)p0)[0], u10 = ((const uint32_t *)p0)[1];
                    t00 = T8k<0>(u00); t01 = T8k<1>(u00); t02 = T8k<2>(u00); t03 = T8k<3>(u00);
                    t10 = T8k<0>(u10); t11 = T8k<1>(u10); t12 = T8k<2>(u10); t13 = T8k<3>(u10);
                }
                const float r = mix4(w00, w10, w01, w11, t00, t10, b00, b10);
                const float g = mix4(w00, w10, w01, w11, t01, t11, b01, b11);
                const float bl = mix4(w00, w10, w01, w11, t02, t12, b02, b12);
                const float q3 = mix4(w00, w10, w01, w11, t03, t13, b03, b13);
                if constexpr (CARRY) { t00 = b00; t01 = b01; t02 = b02; t03 = b03; t10 = b10; t11 = b11; t12 = b12; t13 = b13; }
                const float a2 = q3 * opacity, ia2 = 1.f - a2;
                float yy, uu, vv;
                rgb2yuv(r * a2, g * a2, bl * a2, yy, uu, vv);
                uint32_t &lw = ly[j >> 2];
                const float rx = T8k<j & 3>(lw) * iaf + fya;
                const uint32_t nlw = put_code<j & 3>(lw, rx * ia2 + yy * a2);
                lw = tk ? nlw : lw;
                if constexpr ((j & 1) == 0) {
                    // the quad's chroma comes from the even lane's pixel of this (even) row; chroma row jj = j / 2 lives in the even
                    // lane (jj even) or in its odd neighbour (jj odd: the values travel one lane up, quad_perm [0, 0, 2, 2])
                    constexpr int jj = j >> 1, m = jj >> 1;
                    float su = uu, sv = vv, sa = a2, sia = ia2;
                    int stk = (tk && owner_lane) ? 1 : 0;
                    if constexpr ((jj & 1) != 0) {
                        su = __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(uu), __float_as_int(uu), 0xA0, 0xf, 0xf, false));
                        sv = __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(vv), __float_as_int(vv), 0xA0, 0xf, 0xf, false));
                        sa = __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(a2), __float_as_int(a2), 0xA0, 0xf, 0xf, false));
                        sia = __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(ia2), __float_as_int(ia2), 0xA0, 0xf, 0xf, false));
                        stk = __builtin_amdgcn_update_dpp(stk, stk, 0xA0, 0xf, 0xf, false);
                    }
                    const bool mine = stk != 0 && par == (jj & 1);
                    const float ry = clampf(T8k<m>(nu) * iaf + fua, -1.f, 1.f);
                    const float rz = clampf(T8k<m>(nv) * iaf + fva, -1.f, 1.f);
                    const uint32_t nnu = put_code<m>(nu, ry * sia + su * sa);
                    const uint32_t nnv = put_code<m>(nv, rz * sia + sv * sa);
                    nu = mine ? nnu : nu; nv = mine ? nnv : nv;
                }
            };
            if (carry) { auto r = [&](auto jc) { row(jc, std::true_type{}); }; for_rows<YTH>(r); }
            else { auto r = [&](auto jc) { row(jc, std::false_type{}); }; for_rows<YTH>(r); }
        } else if (col_in) {
            // ---- any other strip: the layer pixel by pixel, the general kernel's code (geometry per pixel, taps from global
            //      memory); chroma of non-owner pixels is computed by the reference and never stored ----
#pragma unroll 1
            for (int j = 0; j < YTH; j++) {
                const int y = y0 + j;
                if (y >= TH) break;
                const int sh = 8 * (j & 3), csh = 8 * (j >> 2);          // luma byte; chroma byte m = (j / 2) / 2 of the holding lane
                uint32_t lw = ly[0];
#pragma unroll
                for (int k = 1; k < YLW; k++) lw = (j >> 2) == k ? ly[k] : lw;
                uint32_t cy = (lw >> sh) & 255u;
                const bool owner = owner_lane && (j & 1) == 0;
                // the owner pixel's chroma codes: its own (chroma row j / 2 even) or its odd neighbour's (odd; quad_perm [1, 1, 3, 3])
                const bool from_odd = ((j >> 1) & 1) != 0;
                const uint32_t ownu = (nu >> csh) & 255u, ownv = (nv >> csh) & 255u;
                const uint32_t nbu = (uint32_t)__builtin_amdgcn_update_dpp((int)ownu, (int)ownu, 0xF5, 0xf, 0xf, false);
                const uint32_t nbv = (uint32_t)__builtin_amdgcn_update_dpp((int)ownv, (int)ownv, 0xF5, 0xf, 0xf, false);
                uint32_t pu = owner ? (from_odd ? nbu : ownu) : 0u, pv = owner ? (from_odd ? nbv : ownv) : 0u;
                if (Strip::is_rgb(Ly.kind)) apply_yuv_from_rgb(Ly, x, y, sx, sy, owner, cy, pu, pv);
                else apply_yuv_from_yuv(Ly, x, y, sx, sy, owner, cy, pu, pv);
                lw = (lw & ~(255u << sh)) | (cy << sh);
#pragma unroll
                for (int k = 0; k < YLW; k++) ly[k] = (j >> 2) == k ? lw : ly[k];
                // back to the lane that holds the row (all lanes of the wave execute this: the loop runs over uniform j)
                const int ow = owner ? 1 : 0;
                const uint32_t bu = (uint32_t)__builtin_amdgcn_update_dpp((int)pu, (int)pu, 0xA0, 0xf, 0xf, false);
                const uint32_t bv = (uint32_t)__builtin_amdgcn_update_dpp((int)pv, (int)pv, 0xA0, 0xf, 0xf, false);
                const bool bo = __builtin_amdgcn_update_dpp(ow, ow, 0xA0, 0xf, 0xf, false) != 0;
                const bool mine = (j & 1) == 0 && bo && par == (from_odd ? 1 : 0);
                if (mine) { nu = (nu & ~(255u << csh)) | (bu << csh); nv = (nv & ~(255u << csh)) | (bv << csh); }
            }
        }
        wave_lds_fence();                 // the taps of layer l are read before the next layer's setup overwrites table and rectangles
        l = ln;
    }

    if (CHV_ABL & 4) {
#pragma unroll
        for (int k = 0; k < YLW; k++) asm volatile("" :: "v"(ly[k]));
        asm volatile("" :: "v"(nu), "v"(nv));
    } else {
        // Stores.  A vector memory instruction occupies the CU's address unit for 16 cycles whatever it moves (tools/ubench_vmem.cpp),
        // and a strip written byte by byte is 16 + 8 of them — a quarter of this kernel's time (profiles/r03_notes.md).  Strips that
        // lie entirely inside the canvas (uniform test) transpose their codes inside the wave first — luma: a 4 x 4 byte transpose in
        // every quad of lanes turns "4 rows of one column" into "4 columns of one row"; chroma: the two lanes of a column pair
        // regroup their 8 rows into rows 0-3 / 4-7, one ds_bpermute packs the columns, then the same transpose — and store dwords:
        // YTH / 4 instructions for luma, one per chroma plane (one 8-byte store for NV12's interleaved plane).
        const bool wide = CHV_WAVEY_WIDE_STORES && x0 + WTW <= T.W && y0 + YTH <= TH &&
                          ((((uintptr_t)PY.ptr | (uintptr_t)PC.ptr | (uintptr_t)PV.ptr) & 3u) == 0) && (((PY.pitch | PC.pitch | PV.pitch) & 3) == 0);
        const int lane = S.lane;
        if (wide) {
            const uint32_t sel1 = (lane & 1) ? 0x03070105u : 0x06020400u, sel2 = (lane & 2) ? 0x03020706u : 0x05040100u;
            auto quad_transpose = [&](uint32_t v) {
                const uint32_t p1 = (uint32_t)__builtin_amdgcn_update_dpp(dpp_old(), (int)v, 0xB1 /* quad_perm [1,0,3,2] */, 0xf, 0xf, false);
                const uint32_t a = __builtin_amdgcn_perm(p1, v, sel1);
                const uint32_t p2 = (uint32_t)__builtin_amdgcn_update_dpp(dpp_old(), (int)a, 0x4E /* quad_perm [2,3,0,1] */, 0xf, 0xf, false);
                return __builtin_amdgcn_perm(p2, a, sel2);
            };
            const uint32_t loff = (uint32_t)(lane & 3) * (uint32_t)PY.pitch + (uint32_t)(x0 + (lane & ~3));
#pragma unroll
            for (int k = 0; k < YLW; k++) gst_at<uint32_t>(PY.ptr + (size_t)(y0 + 4 * k) * PY.pitch, loff, quad_transpose(ly[k]));
        } else if (col_in) {
#pragma unroll
            for (int j = 0; j < YTH; j++)
                if (y0 + j < TH) gst_at<uint8_t>(PY.ptr + (size_t)(y0 + j) * PY.pitch, (uint32_t)x, (uint8_t)((ly[j >> 2] >> (8 * (j & 3))) & 255u));
        }
        if (wide && YTH == 16) {
            // lane 2k: rows 0, 2, 4, 6 of chroma column k; lane 2k + 1: rows 1, 3, 5, 7  ->  lane 2k: rows 0-3, lane 2k + 1: rows 4-7
            const uint32_t selp = (lane & 1) ? 0x03070206u : 0x05010400u;
            const int src = ((lane & ~7) + 2 * (lane & 3) + ((lane >> 2) & 1)) * 4;         // ds_bpermute: lane 8c + i <- column 4c + i (rows 0-3), lane 8c + 4 + i <- the same column (rows 4-7)
            const uint32_t sel1 = (lane & 1) ? 0x03070105u : 0x06020400u, sel2 = (lane & 2) ? 0x03020706u : 0x05040100u;
            auto regroup = [&](uint32_t v) {
                const uint32_t p = (uint32_t)__builtin_amdgcn_update_dpp(dpp_old(), (int)v, 0xB1, 0xf, 0xf, false);
                uint32_t a = __builtin_amdgcn_perm(p, v, selp);
                a = (uint32_t)__builtin_amdgcn_ds_bpermute(src, (int)a);
                const uint32_t p1 = (uint32_t)__builtin_amdgcn_update_dpp(dpp_old(), (int)a, 0xB1, 0xf, 0xf, false);
                const uint32_t b = __builtin_amdgcn_perm(p1, a, sel1);
                const uint32_t p2 = (uint32_t)__builtin_amdgcn_update_dpp(dpp_old(), (int)b, 0x4E, 0xf, 0xf, false);
                return __builtin_amdgcn_perm(p2, b, sel2);                                   // lane 8c + i: row i (+ 4 for lanes 8c + 4 ..), columns 4c .. 4c + 3
            };
            const uint32_t tu = regroup(nu), tv = regroup(nv);
            const uint32_t crow = (uint32_t)((lane & 3) + 4 * ((lane >> 2) & 1)), ccol = (uint32_t)((x0 >> 1) + 4 * (lane >> 3));
            if (TF == TF_NV12) {
                const uint2 w = make_uint2(__builtin_amdgcn_perm(tv, tu, 0x05010400u), __builtin_amdgcn_perm(tv, tu, 0x07030602u));     // u0 v0 u1 v1 | u2 v2 u3 v3
                gst_at<chv_u32x2>(PC.ptr + (size_t)qy0 * PC.pitch, crow * (uint32_t)PC.pitch + ccol * 2u, chv_u32x2{ w.x, w.y });
            } else {
                gst_at<uint32_t>(PC.ptr + (size_t)qy0 * PC.pitch, crow * (uint32_t)PC.pitch + ccol, tu);
                gst_at<uint32_t>(PV.ptr + (size_t)qy0 * PV.pitch, crow * (uint32_t)PV.pitch + ccol, tv);
            }
        } else if (col_in) {
#pragma unroll
            for (int m = 0; m < YCM; m++) {
                if (y0 + 4 * m + 2 * par < TH) {
                    const uint32_t ub = (nu >> (8 * m)) & 255u, vb = (nv >> (8 * m)) & 255u;
                    if (TF == TF_NV12) gst_at<uint16_t>(PC.ptr + (size_t)(qy0 + 2 * m) * PC.pitch, coff_c, (uint16_t)(ub | (vb << 8)));
                    else {
                        gst_at<uint8_t>(PC.ptr + (size_t)(qy0 + 2 * m) * PC.pitch, coff_c, (uint8_t)ub);
                        gst_at<uint8_t>(PV.ptr + (size_t)(qy0 + 2 * m) * PV.pitch, coff_v, (uint8_t)vb);
                    }
                }
            }
        }
    }
}

// the launch of one instantiation (launch_wave_layers has sized everything)
template <bool CACHED>
hipError_t launch_yuv_wave_t(int target_format, bool clear, int WTH, int kinds, dim3 grid, size_t lds, hipStream_t stream, const DTick *ticks, const DLayer *layers, int n_ticks,
                             int strips_x, int strips_y, uint32_t strips_magic, uint32_t strips_x_magic, int p0pitch, int p0rows_arg, int p1pitch, int p1rows_arg, int planar_any, const WaveOne *one_arg) {
    static const WaveOne none{};
    const WaveOne &one = one_arg ? *one_arg : none;            // (with `ticks` == nullptr the kernel reads its descriptors from this argument: wave_one_descriptors)
#define CHV_LAUNCH_Y(TFV, C, R, K) hipLaunchKernelGGL((tick_yuv_wave<TFV, C, R, K, CACHED>), grid, dim3(WAVE_BLOCK), lds, stream, ticks, layers, n_ticks, strips_x, strips_y, \
                                                      strips_magic, strips_x_magic, p0pitch, p0rows_arg, p1pitch, p1rows_arg, planar_any, one)
#define CHV_LAUNCH_YK(TFV, C, R, OWN) do { if (kinds == OWN) CHV_LAUNCH_Y(TFV, C, R, OWN); else if (kinds == (OWN | 4)) CHV_LAUNCH_Y(TFV, C, R, (OWN | 4)); \
                                           else if (kinds & 8) CHV_LAUNCH_Y(TFV, C, R, 15); else CHV_LAUNCH_Y(TFV, C, R, 7); } while (0)
#define CHV_LAUNCH_YR(TFV, C, OWN) do { if (WTH == 16) CHV_LAUNCH_YK(TFV, C, 16, OWN); else CHV_LAUNCH_YK(TFV, C, 8, OWN); } while (0)
    if (target_format == TF_NV12) { if (clear) CHV_LAUNCH_YR(TF_NV12, true, 1); else CHV_LAUNCH_YR(TF_NV12, false, 1); }
    else { if (clear) CHV_LAUNCH_YR(TF_Y420P, true, 2); else CHV_LAUNCH_YR(TF_Y420P, false, 2); }
#undef CHV_LAUNCH_YK
#undef CHV_LAUNCH_YR
#undef CHV_LAUNCH_Y
    return hipGetLastError();
}
#define CHV_YUV_WAVE_ARGS int target_format, bool clear, int WTH, int kinds, dim3 grid, size_t lds, hipStream_t stream, const DTick *ticks, const DLayer *layers, int n_ticks, \
                          int strips_x, int strips_y, uint32_t strips_magic, uint32_t strips_x_magic, int p0pitch, int p0rows_arg, int p1pitch, int p1rows_arg, int planar_any, \
                          const WaveOne *one_arg
#if CHV_WAVE_TU != 0
template hipError_t launch_yuv_wave_t<true>(CHV_YUV_WAVE_ARGS);
#else
template hipError_t launch_yuv_wave_t<false>(CHV_YUV_WAVE_ARGS);
extern template hipError_t launch_yuv_wave_t<true>(CHV_YUV_WAVE_ARGS);              // kernels_wave_yuv_cached.hip.cpp


// ---------------------------------------------------------------------------
// host side: eligibility and launch of both wave kernels
// ---------------------------------------------------------------------------
static bool finite16w(const float *m) {
    for (int i = 0; i < 16; i++) if (!(m[i] - m[i] == 0.f)) return false;
    return true;
}
// (pitch < 2^24: the staging address arithmetic uses 24-bit multiplies)
static bool aligned16w(const DPlane &p) { return (((uintptr_t)p.ptr) & 15) == 0 && (p.pitch & 15) == 0 && p.w * p.comps >= 16 && p.pitch < (1 << 24) && p.h < (1 << 24); }
static bool host_src_rgb(int kind) { return kind == LK_BGRA_FROM_RGB || kind == LK_YUV_FROM_RGB || kind == LK_YUV_FROM_RGB_INT; }
static bool host_src_planar(int kind) { return kind == LK_BGRA_FROM_Y420P || kind == LK_YUV_FROM_Y420P; }
static bool host_src_nv12(int kind) { return kind == LK_BGRA_FROM_NV12 || kind == LK_YUV_FROM_NV12; }

struct WaveDims { int p0pitch, p0rows, p1pitch, p1rows; };
static int strip_rows(int) { return 8; }      // (the height every eligible tick must fit; launches may pick 16, see launch_wave_layers)

// LDS rectangles one strip of this layer can touch, from the layer's scale factors
// rows of slack in a staged rectangle beyond ceil(strip rows x vertical ratio): tap row 1, and the ratio's rounding on both sides
#ifndef CHV_P0ROWS_SLACK
#define CHV_P0ROWS_SLACK 3
#endif
static WaveDims wave_dims(const DTick &T, const DLayer &L, int WTH) {
    const float *U = L.u;
    double sxr = std::fabs((double)U[U_TEXTURE + 0] * (double)U[U_TRANSFORM + 0] * 2.0 / (double)T.W);
    double syr = std::fabs((double)U[U_TEXTURE + 5] * (double)U[U_TRANSFORM + 5] * 2.0 / (double)T.H);
    WaveDims d{ 0, 0, 0, 0 };
    const bool rgb = host_src_rgb(L.kind), planar = host_src_planar(L.kind);
    const int bpt0 = rgb ? 4 : 1;
    int span0 = (int)std::ceil(WTW * sxr * L.src.pl[0].w) + 4;           // texels incl. tap 1 and rounding slack
    d.p0pitch = ((span0 * bpt0 + 15) / 16 + 3) * 16;                      // vectors + alignment + 2 pad vectors
    // (rectangles taller than two rows per strip row are staged as the rows' own tap-row pairs: WGeom::pair, wave_common.hip.h)
    d.p0rows = std::min((int)std::ceil(WTH * syr * L.src.pl[0].h) + CHV_P0ROWS_SLACK, (CHV_WAVE_PAIR && WTH == 8) ? 2 * WTH : (1 << 30));
    if (!rgb) {
        const int bpt1 = planar ? 1 : 2;
        int span1 = (int)std::ceil(WTW * sxr * L.src.pl[1].w) + 4;
        d.p1pitch = ((span1 * bpt1 + 15) / 16 + 3) * 16;
        d.p1rows = std::min((int)std::ceil(WTH * syr * L.src.pl[1].h) + 3, (CHV_WAVE_PAIR && WTH == 8) ? 2 * WTH : (1 << 30));
    }
    return d;
}
static size_t wave_lds(const WaveDims &d, bool planar, int target_format, int rows) {
    return            (size_t)WAVES * ((size_t)rows * 48 + (size_t)d.p0pitch * d.p0rows + (size_t)d.p1pitch * d.p1rows * (planar ? 2 : 1));
}

// tail: the launch continues on canvases another launch has composed (the second part of a split batch): strips no layer touches leave at once,
// so the strip kernel is the better choice also where every layer needs the per-pixel path
bool wave_layers_eligible(int target_format, const DTick *ticks, const DLayer *layers, int n_ticks, bool tail) {
    bool any_general = false, any_staged = false;
    for (int i = 0; i < n_ticks; i++) {
        const DTick &T = ticks[i];
        if (T.n_layers < 1 || T.clear_first != ticks[0].clear_first) return false;
        if (target_format == TF_BGRA) {
            if (!aligned16w(T.dst.pl[0])) return false;
        } else {
            // 4:2:0: even canvas, chroma planes exactly half size (odd sizes, where gid/2 leaves the chroma plane, stay on the general kernel)
            const int np = target_format == TF_NV12 ? 2 : 3;
            if ((T.W & 1) || (T.H & 1) || T.dst.pl[0].w != T.W || T.dst.pl[0].h != T.H) return false;
            for (int p = 1; p < np; p++) if (T.dst.pl[p].w != T.W / 2 || T.dst.pl[p].h != T.H / 2) return false;
        }
        for (int l = 0; l < T.n_layers; l++) {
            const DLayer &L = layers[T.first_layer + l];
            const bool rgb = host_src_rgb(L.kind), nv12 = host_src_nv12(L.kind), planar = host_src_planar(L.kind);
            // img_bgra_bgra (kernels.metal:52-62 — what an unchanged VideoMixer.findKernel resolves a BGRA layer on a BGRA canvas to):
            // nearest sampling, nothing to stage; applied per pixel inside the wave kernel like a rotated layer, so that a
            // reference-default tick of video layers + a BGRA overlay keeps the strip path for its video layers
            if (L.kind == LK_BGRA_METAL && target_format == TF_BGRA) { any_general = true; continue; }
            if (!(rgb || nv12 || planar)) return false;
            // layers that cannot be staged (rotation, shear, unbounded matrices) are applied per pixel inside the kernel: nothing to check
            if ((L.flags & (LF_AXIS_ALIGNED | LF_BOUNDED)) != (LF_AXIS_ALIGNED | LF_BOUNDED)) { any_general = true; continue; }
            any_staged = true;
            if (!finite16w(L.u + U_TRANSFORM) || !finite16w(L.u + U_TEXTURE) || !finite16w(L.u + U_BORDER)) return false;
            const int np = rgb ? 1 : nv12 ? 2 : 3;
            for (int p = 0; p < np; p++) if (!aligned16w(L.src.pl[p])) return false;
            if (planar && (L.src.pl[2].w != L.src.pl[1].w || L.src.pl[2].h != L.src.pl[1].h)) return false;   // one staging geometry for U and V
            if (wave_lds(wave_dims(T, L, strip_rows(target_format)), planar, target_format, strip_rows(target_format)) > (size_t)LDS_BUDGET) return false;
        }
    }
    // (a launch whose layers ALL need the per-pixel path gains nothing here: the general kernel it is)
    return any_staged || !any_general || tail;
}

#define CHV_STR2(x) #x
#define CHV_STR(x) CHV_STR2(x)
const char *yuv_wave_build_flags() { return "tick_yuv_wave:abl=" CHV_STR(CHV_ABL) ",strip_rows=8|16"; }

// kernels_wave.hip.cpp
hipError_t launch_bgra_wave(int rows, bool clear, dim3 grid, size_t lds, hipStream_t stream, const DTick *ticks, const DLayer *layers, int n_ticks,
                            int strips_x, int strips_y, uint32_t strips_magic, uint32_t strips_x_magic, int p0pitch, int p0rows, int p1pitch, int p1rows, int planar, int kinds, bool cached,
                            const WaveOne *one_arg);

// ---------------------------------------------------------------------------
// geometry tables of a batch (geom_cache.h; device side: wave_common.hip.h)
// ---------------------------------------------------------------------------
// One wave per (class, strip): the strip kernels' own set-up on a representative layer of the class, stored where the tick kernels look it up.
// KINDS = 7: the generic source-class tests (the single-class instantiations' shortcuts give the same answers for the kinds they see).
template <int WTH>
__global__ __launch_bounds__(64) void geom_precompute(const GeomJob *__restrict__ jobs, int n_jobs, int p0pitch, int p0rows, int p1pitch, int p1rows, int planar_any) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem_all[];
    const int b = (int)blockIdx.x;
    int j = 0;
    while (j + 1 < n_jobs && b >= jobs[j + 1].first_block) j++;
    j = __builtin_amdgcn_readfirstlane(j);
    const GeomJob &J = jobs[j];
    const int strip = b - J.first_block;
    const int sya = strip / J.strips_x, sxa = strip - sya * J.strips_x;
    if (sya >= J.strips_y) return;
    WaveStrip<WTH, 0, 7> S;
    S.init_layout(smem_all, p0pitch, p0rows, p1pitch, p1rows, planar_any);
    S.T = nullptr; S.L = &J.layer; S.nl = 1;
    S.init_strip(J.W, J.H, sxa * WTW, sya * WTH);
    WLayer w;
    GeomRaw raw;
    memset(&w, 0, sizeof w);
    memset(&raw, 0, sizeof raw);
    S.template setup<true>(0, w, &raw);
    wave_lds_fence();
    S.geom_store(J.table, sxa, sya, w, raw);
}

GeomCache *&geom_cache_current() {
    static thread_local GeomCache *cur = nullptr;
    return cur;
}
void geom_cache_release(GeomCache &c) {
    if (c.owns) {
        if (c.tables) (void)hipFree(c.tables);
        if (c.jobs) (void)hipFree(c.jobs);
    }
    c.tables = c.jobs = nullptr;
    c.owns = true;
    c.built = false; c.bytes = 0; c.classes = 0;
}

// ---- the device's store (geom_cache.h) ----
struct GeomStore {
    std::mutex mu;
    std::unordered_map<std::string, void *> tables;          // (class key + configuration key) -> the class's table
    std::unordered_map<std::string, int> sightings;          // launches that asked for it and did not find it
    std::vector<void *> owned;                               // allocations given to the store (never freed: the store lives as long as the process)
    size_t bytes = 0;
    bool full = false;                                       // a build did not fit any more: nothing asks for builds on the store's behalf from here on
    std::atomic<uint64_t> patched{0};                        // (geom_store_counter; bumped without the lock by the lone tick's memo)
    uint64_t batch_hits = 0, builds = 0;
};
static GeomStore &geom_store() {
    static GeomStore stores[16];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) { (void)hipGetLastError(); dev = 0; }
    return stores[dev & 15];
}
GeomTransient &geom_transient_current() {
    static thread_local GeomTransient t;
    return t;
}
uint64_t geom_store_counter(int which) {
    GeomStore &st = geom_store();
    std::lock_guard<std::mutex> lk(st.mu);
    switch (which) {
    case 0: return st.patched.load(std::memory_order_relaxed);
    case 1: return st.batch_hits;
    case 2: return st.builds;
    case 3: return (uint64_t)st.bytes;
    case 4: return (uint64_t)st.tables.size();
    }
    return 0;
}
// the inputs of a layer's set-up as bytes: the three matrices, the source planes' sizes and layout class, the canvas size (false: the layer is
// applied per pixel — no set-up, no table)
struct GeomRawKey { float u[48]; int32_t w0, h0, w1, h1, cls, W, H; };
static bool geom_class_raw(const DTick &T, const DLayer &L, GeomRawKey &k) {
    if (L.kind == LK_BGRA_METAL || (L.flags & (LF_AXIS_ALIGNED | LF_BOUNDED)) != (LF_AXIS_ALIGNED | LF_BOUNDED)) return false;
    memset(&k, 0, sizeof k);
    memcpy(k.u, L.u, sizeof k.u);
    const bool rgb = host_src_rgb(L.kind);
    k.w0 = L.src.pl[0].w; k.h0 = L.src.pl[0].h; k.w1 = rgb ? 0 : L.src.pl[1].w; k.h1 = rgb ? 0 : L.src.pl[1].h;
    k.cls = rgb ? 2 : host_src_planar(L.kind) ? 1 : 0; k.W = T.W; k.H = T.H;
    return true;
}
static bool geom_class_key(const DTick &T, const DLayer &L, std::string &out) {
    GeomRawKey k;
    if (!geom_class_raw(T, L, k)) return false;
    out.assign((const char *)&k, sizeof k);
    return true;
}
// The scene a thread's last covered LONE tick was (its classes' raw keys, the launch configuration, the tables): the next tick of the scene —
// the steady state of a mixer — compares bytes and takes the pointers; no strings, no hashing, no lock (tables are never freed, so a covered
// answer stays right).  0.6 us of a lone tick's 4.8 us of host time (tools/host_enqueue_probe.py).
struct GeomLoneMemo { int dev = -1, n = 0, tf = -1; GeomConfig cfg{}; bool has[WAVE_ONE_LAYERS + 2]; GeomRawKey key[WAVE_ONE_LAYERS + 2]; void *tab[WAVE_ONE_LAYERS + 2]; };
static GeomLoneMemo &geom_lone_memo() {
    static thread_local GeomLoneMemo m;
    return m;
}
// what of a launch configuration a class's table depends on (everything but the size of the batch's layer array)
static std::string geom_config_key(const GeomConfig &c) {
    const int32_t v[9] = { c.target_format, c.wth, c.p0pitch, c.p0rows, c.p1pitch, c.p1rows, c.planar_any, c.strips_x, c.strips_y };
    return std::string((const char *)v, sizeof v);
}

// (Re)build the tables of the batch being launched for this launch configuration and point its device layers at them; with the switch off,
// take the pointers out again.  Everything is ordered on `stream` in front of the tick kernel.
static hipError_t geom_cache_prepare(GeomCache &gc, const GeomConfig &cfg, const DTick *ticks_host, int n_ticks, size_t rowtab_lds, hipStream_t stream, bool *covered) {
    *covered = false;
    const bool on = CHV_GEOM_CACHE && switches().geom_cache.load(std::memory_order_relaxed) != 0;
    DLayer *hl = gc.h_layers;
    if (!hl || !gc.d_layers || gc.n_layers <= 0) return hipSuccess;
    if (!on) {
        if (gc.patched) {
            // (rare: a switch flipped between two runs of a batch.  Nothing in flight may still read the layers: wait, then copy synchronously)
            hipError_t e = hipStreamSynchronize(stream);
            if (e != hipSuccess) return e;
            for (int i = 0; i < gc.n_layers; i++) hl[i].pad2[0] = hl[i].pad2[1] = 0;
            e = hipMemcpy(gc.d_layers, hl, sizeof(DLayer) * (size_t)gc.n_layers, hipMemcpyHostToDevice);
            if (e != hipSuccess) return e;
            gc.patched = false; gc.built = false;
        }
        return hipSuccess;
    }
    if (gc.built && gc.config == cfg) { *covered = gc.patched; return hipSuccess; }
    // classes: layers whose set-up inputs are the same bytes — the three matrices, the source planes' sizes and layout class, the canvas size
    std::map<std::string, int> index;
    std::vector<GeomJob> jobs;
    std::vector<std::string> keys;
    std::vector<int> cls_of((size_t)gc.n_layers, -1);
    for (int i = 0; i < n_ticks; i++) {
        const DTick &T = ticks_host[i];
        for (int l = 0; l < T.n_layers; l++) {
            const int li = T.first_layer + l;
            if (li < 0 || li >= gc.n_layers) continue;
            const DLayer &L = hl[li];
            std::string ks;
            if (!geom_class_key(T, L, ks)) continue;             // applied per pixel: no set-up
            auto it = index.find(ks);
            if (it == index.end()) {
                GeomJob J;
                memset(&J, 0, sizeof J);
                J.layer = L;
                J.layer.pad2[0] = J.layer.pad2[1] = 0;
                J.W = T.W; J.H = T.H;
                J.strips_x = (T.W + WTW - 1) / WTW; J.strips_y = (T.H + cfg.wth - 1) / cfg.wth;
                it = index.emplace(ks, (int)jobs.size()).first;
                jobs.push_back(J);
                keys.push_back(ks);
            }
            cls_of[(size_t)li] = it->second;
        }
    }
    // The device's store first: tables another batch (or lone tick) of this geometry and configuration left there.  All found: the layers are
    // pointed at them (one drain + one copy of the layer array, no allocation, no kernel) whatever this batch has seen.
    const std::string ck = geom_config_key(cfg);
    bool known = !jobs.empty();                 // every class has been asked for before (by anything on this device)
    {
        GeomStore &st = geom_store();
        std::vector<void *> found(jobs.size(), nullptr);
        bool all = !jobs.empty() && jobs.size() <= 256;
        {
            std::lock_guard<std::mutex> lk(st.mu);
            for (size_t c = 0; c < jobs.size(); c++) {
                const std::string full = keys[c] + ck;
                auto it = st.tables.find(full);
                if (it != st.tables.end()) found[c] = it->second; else all = false;
                auto sg = st.sightings.find(full);
                if (sg == st.sightings.end() || sg->second < 2 || st.full) known = false;
            }
        }
        if (all) {
            hipError_t es = hipStreamSynchronize(stream);
            if (es != hipSuccess) return es;
            geom_cache_release(gc);
            for (int i = 0; i < gc.n_layers; i++) {
                hl[i].pad2[0] = hl[i].pad2[1] = 0;
                if (cls_of[(size_t)i] < 0) continue;
                const uint64_t tp = (uint64_t)(uintptr_t)found[(size_t)cls_of[(size_t)i]];
                hl[i].pad2[0] = (int32_t)(uint32_t)(tp & 0xFFFFFFFFu); hl[i].pad2[1] = (int32_t)(uint32_t)(tp >> 32);
            }
            es = hipMemcpy(gc.d_layers, hl, sizeof(DLayer) * (size_t)gc.n_layers, hipMemcpyHostToDevice);
            if (es != hipSuccess) return es;
            gc.owns = false; gc.tables = found[0]; gc.patched = true; gc.built = true; gc.config = cfg; gc.classes = (int)jobs.size();
            { std::lock_guard<std::mutex> lk(st.mu); st.batch_hits++; }
            *covered = true;
            return hipSuccess;
        }
    }
    // A batch that is run once (a host that builds one per tick) never pays for tables: they are built at the SECOND launch with a configuration
    // (draining the stream, three small copies and a kernel: tens of microseconds) — the second launch of this batch, or a launch of geometry the
    // store has been asked for before —, the first one computes its geometry in place.
    if (!(gc.seen && gc.seen_config == cfg) && !gc.force_build && !known && switches().geom_cache.load(std::memory_order_relaxed) != 2) {
        gc.seen = true; gc.seen_config = cfg;
        if (gc.patched) {          // (tables of another configuration: the kernels about to run compute in place and never look, but the pointers go)
            hipError_t e0 = hipStreamSynchronize(stream);
            if (e0 != hipSuccess) return e0;
            for (int i = 0; i < gc.n_layers; i++) hl[i].pad2[0] = hl[i].pad2[1] = 0;
            e0 = hipMemcpy(gc.d_layers, hl, sizeof(DLayer) * (size_t)gc.n_layers, hipMemcpyHostToDevice);
            if (e0 != hipSuccess) return e0;
            gc.patched = false; gc.built = false;
        }
        return hipSuccess;
    }
    // A (re)build happens once per batch and launch configuration: an earlier run of the batch may still be reading the layers, and the host
    // buffers below are pageable — the stream is drained first and every copy is a synchronous one (an asynchronous copy from pageable memory
    // may read its source after this function has returned).
    hipError_t e = hipStreamSynchronize(stream);
    if (e != hipSuccess) return e;
    geom_cache_release(gc);
    for (int i = 0; i < gc.n_layers; i++) hl[i].pad2[0] = hl[i].pad2[1] = 0;
    // (a batch of a thousand distinct geometries gains nothing from tables that are each used once)
    if (!jobs.empty() && jobs.size() <= 256) {
        const size_t row_bytes = (size_t)2 * 3 * cfg.wth * 16 + 64;      // the staged and the unstaged row table, 16 scalars
        std::vector<size_t> offs(jobs.size());
        size_t total = 0;
        int blocks = 0;
        for (size_t c = 0; c < jobs.size(); c++) {
            GeomJob &J = jobs[c];
            offs[c] = total;
            const size_t flags = ((size_t)J.strips_x * J.strips_y * 4 + 15) & ~(size_t)15;
            total += sizeof(GeomHdr) + flags + (size_t)J.strips_x * sizeof(GeomCol) + (size_t)J.strips_y * row_bytes;
            total = (total + 255) & ~(size_t)255;
            J.first_block = blocks;
            blocks += J.strips_x * J.strips_y;
        }
        e = hipMalloc(&gc.tables, total);
        if (e == hipSuccess) e = hipMalloc(&gc.jobs, sizeof(GeomJob) * jobs.size());

        if (e == hipSuccess) {
            std::vector<uint8_t> image(total, 0);              // zeroed: a strip's flag word 0 = "not in the table"
            for (size_t c = 0; c < jobs.size(); c++) {
                GeomJob &J = jobs[c];
                J.table = (uint8_t *)gc.tables + offs[c];
                GeomHdr H;
                memset(&H, 0, sizeof H);
                H.strips_x = J.strips_x; H.strips_y = J.strips_y; H.wth = cfg.wth; H.row_bytes = (int32_t)row_bytes;
                H.flags_off = (uint32_t)sizeof(GeomHdr);
                H.cols_off = H.flags_off + (uint32_t)(((size_t)J.strips_x * J.strips_y * 4 + 15) & ~(size_t)15);
                H.rows_off = H.cols_off + (uint32_t)((size_t)J.strips_x * sizeof(GeomCol));
                memcpy(image.data() + offs[c], &H, sizeof H);
            }
            e = hipMemcpy(gc.tables, image.data(), total, hipMemcpyHostToDevice);
        }
        if (e == hipSuccess) e = hipMemcpy(gc.jobs, jobs.data(), sizeof(GeomJob) * jobs.size(), hipMemcpyHostToDevice);
        if (e == hipSuccess) {
            (void)hipGetLastError();
            if (cfg.wth == 16) hipLaunchKernelGGL(geom_precompute<16>, dim3((unsigned)blocks), dim3(64), rowtab_lds, stream, (const GeomJob *)gc.jobs, (int)jobs.size(),
                                                  cfg.p0pitch, cfg.p0rows, cfg.p1pitch, cfg.p1rows, cfg.planar_any);
            else hipLaunchKernelGGL(geom_precompute<8>, dim3((unsigned)blocks), dim3(64), rowtab_lds, stream, (const GeomJob *)gc.jobs, (int)jobs.size(),
                                    cfg.p0pitch, cfg.p0rows, cfg.p1pitch, cfg.p1rows, cfg.planar_any);
            e = hipGetLastError();
        }
        if (e == hipSuccess) {
            for (int i = 0; i < gc.n_layers; i++) {
                if (cls_of[(size_t)i] < 0) continue;
                const uint64_t tp = (uint64_t)(uintptr_t)jobs[(size_t)cls_of[(size_t)i]].table;
                hl[i].pad2[0] = (int32_t)(uint32_t)(tp & 0xFFFFFFFFu); hl[i].pad2[1] = (int32_t)(uint32_t)(tp >> 32);
            }
            gc.bytes = total; gc.classes = (int)jobs.size();
            gc.owns = true;
            // The tables go to the device's store (another stream may read them: the precompute kernel has to be done first).
            if (hipStreamSynchronize(stream) == hipSuccess) {
                GeomStore &st = geom_store();
                std::lock_guard<std::mutex> lk(st.mu);
                const size_t give = total + sizeof(GeomJob) * jobs.size();
                if (st.bytes + give <= kGeomStoreBytes) {
                    for (size_t c = 0; c < jobs.size(); c++) st.tables.emplace(keys[c] + ck, (void *)jobs[c].table);      // (a class another thread gave meanwhile keeps its first table)
                    st.owned.push_back(gc.tables); st.owned.push_back(gc.jobs);
                    st.bytes += give;
                    st.builds++;
                    gc.owns = false;
                } else st.full = true;
            } else (void)hipGetLastError();
        } else {
            (void)hipGetLastError();
            geom_cache_release(gc);             // no tables: the kernels compute their geometry as before
            e = hipSuccess;
        }
    }
    hipError_t e2 = hipMemcpy(gc.d_layers, hl, sizeof(DLayer) * (size_t)gc.n_layers, hipMemcpyHostToDevice);
    if (e2 != hipSuccess) return e2;
    gc.patched = gc.tables != nullptr;        // (every layer the kernels set up has a table, or none has: classes are all-or-nothing)
    gc.built = true;
    gc.force_build = false;
    gc.config = cfg;
    *covered = gc.patched;
    return e;
}

// What a launch of the strip kernels is shaped like — strip height, LDS layout, source classes — from its descriptors alone (launch_wave_layers
// launches with it; geom_store_patch asks for it before the descriptors go to the device: the tables it looks up were built for one shape).
struct WavePlan { int WTH; WaveDims m; bool planar; int kinds, side; size_t lds; };
static WavePlan plan_wave_layers(int target_format, const DTick *ticks_host, const DLayer *layers_host, int n_ticks, int maxW, int maxH) {
    // Strip height.  16 rows when the launch has enough strips to fill the chip's wave slots with them and the taller
    // rectangles leave room for two strips per 64 KB of LDS; on BGRA canvases (whose 16-row instantiation has no masked rows:
    // registers) only when every layer covers (almost) the whole canvas, so that hardly any strip is crossed by a layer's
    // edge.  8 rows otherwise.  CHV_WAVE_ROWS=8|16 is the A/B and test switch.
    int WTH = strip_rows(target_format);
    {
        const int forced = switches().wave_rows.load(std::memory_order_relaxed);
        bool tall = (long)n_ticks * ((maxW + WTW - 1) / WTW) * ((maxH + 15) / 16) >= 8192;
        if (target_format == TF_BGRA) {
            for (int i = 0; i < n_ticks && tall; i++) {
                const DTick &T = ticks_host[i];
                for (int l = 0; l < T.n_layers && tall; l++) {
                    const int32_t *bb = layers_host[T.first_layer + l].bbox;
                    tall = (double)(bb[2] - bb[0]) * (double)(bb[3] - bb[1]) >= 0.9 * (double)T.W * (double)T.H;
                }
            }
        }
        if (forced == 8 || forced == 16) tall = forced == 16;
        if (tall) WTH = 16;
    }
    WaveDims m{ 0, 0, 0, 0 };
    bool planar = false;
    int kinds = 0;               // source classes in the launch: bit 0 NV12, bit 1 y420p, bit 2 RGB
    auto measure = [&](int rows) {
        m = WaveDims{ 0, 0, 0, 0 };
        planar = false;
        for (int i = 0; i < n_ticks; i++) {
            for (int l = 0; l < ticks_host[i].n_layers; l++) {
                const DLayer &L = layers_host[ticks_host[i].first_layer + l];
                if (L.kind == LK_BGRA_METAL || (L.flags & (LF_AXIS_ALIGNED | LF_BOUNDED)) != (LF_AXIS_ALIGNED | LF_BOUNDED)) { kinds |= 8; continue; }     // not staged
                WaveDims d = wave_dims(ticks_host[i], L, rows);
                m.p0pitch = std::max(m.p0pitch, d.p0pitch); m.p0rows = std::max(m.p0rows, d.p0rows);
                m.p1pitch = std::max(m.p1pitch, d.p1pitch); m.p1rows = std::max(m.p1rows, d.p1rows);
                planar = planar || host_src_planar(L.kind);
                kinds |= host_src_rgb(L.kind) ? 4 : host_src_planar(L.kind) ? 2 : 1;
            }
        }
        return wave_lds(m, planar, target_format, rows);
    };
    size_t lds = measure(WTH);
    if (WTH == 16 && lds > (size_t)LDS_BUDGET / 2) { WTH = 8; lds = measure(WTH); }      // (keep at least two tall strips' worth per 64 KB)
    // Side-by-side layout: a YUV layer's luma, chroma / U and V rectangles next to each other in the rows of ONE region whose rows are as wide as the
    // widest need of the launch — an RGB rectangle (the mixer canvas: a video layer's 128 + 96 + 96 bytes in an overlay's 320) or luma + chroma of
    // the widest video layer — instead of a plane-0 region followed by chroma regions.  The region is then max(rows) x pitch instead of the sum of
    // the regions; taken where that is smaller (18 -> 23 waves' worth of LDS per CU on the mixer canvas, 20 -> 26 for video + overlays on BGRA).
    int side = 0;
    if ((kinds & 3) && lds <= (size_t)LDS_BUDGET) {
        int yneed = 0, c_nv12 = 0, c_planar = 0;
        for (int i = 0; i < n_ticks; i++)
            for (int l = 0; l < ticks_host[i].n_layers; l++) {
                const DLayer &L = layers_host[ticks_host[i].first_layer + l];
                if (L.kind == LK_BGRA_METAL || (L.flags & (LF_AXIS_ALIGNED | LF_BOUNDED)) != (LF_AXIS_ALIGNED | LF_BOUNDED) || host_src_rgb(L.kind)) continue;
                const WaveDims d = wave_dims(ticks_host[i], L, WTH);
                yneed = std::max(yneed, d.p0pitch);
                if (host_src_planar(L.kind)) c_planar = std::max(c_planar, d.p1pitch); else c_nv12 = std::max(c_nv12, d.p1pitch);
            }
        const int planes = planar ? 2 : 1;
        const int pitch = std::max(m.p0pitch, yneed + std::max(c_nv12, 2 * c_planar));
        const size_t separate = (size_t)m.p0pitch * m.p0rows + (size_t)m.p1pitch * m.p1rows * planes;
        const size_t beside = (size_t)pitch * std::max(m.p0rows, m.p1rows);
        if (yneed > 0 && beside < separate && (yneed >> 4) < 4096 && (c_planar >> 4) < 4096) {
            // bits 8-19: luma columns / 16; bits 20-31: columns of ONE planar chroma plane / 16 (0: no planar picture in the launch)
            side = ((yneed >> 4) << 8) | ((c_planar >> 4) << 20) | (1 << 7);
            m.p0pitch = pitch;
            lds = lds - (size_t)WAVES * separate + (size_t)WAVES * beside;
        }
    }
    if (lds > (size_t)LDS_BUDGET) {
        // per-layer maxima combined exceed the budget: shrink the row counts; rectangles that do not fit fall back to
        // unstaged taps inside the kernel
        const size_t fixed = (size_t)WAVES * WTH * 48;     // (row table: WaveCfg::ROWTAB_BYTES)
        const size_t per_row = (size_t)WAVES * ((size_t)m.p0pitch + (size_t)m.p1pitch * (planar ? 2 : 1));
        int rows = std::max(1, (int)((LDS_BUDGET - fixed) / per_row));
        m.p0rows = std::min(m.p0rows, rows); m.p1rows = std::min(m.p1rows, rows);
        lds = wave_lds(m, planar, target_format, WTH);
    }
#ifdef CHV_EXP_LDS_PAD
    // measurement-only variant (tools/build_variant.sh … -DCHV_EXP_LDS_PAD): extra LDS per block caps the waves per SIMD — how much
    // occupancy the strip kernels need (profiles/r03_notes.md section 7)
    if (const char *pad = getenv("CHV_LDS_PAD")) { fprintf(stderr, "[chv] lds %zu + pad %d, rows %d\n", lds, atoi(pad), WTH); lds += (size_t)atoi(pad); }
#endif
    m.p0rows = std::min(m.p0rows, 0xFFFF); m.p1rows = std::min(m.p1rows, 0xFFFF);      // (far beyond what LDS holds: such rectangles are not staged anyway)
    return WavePlan{ WTH, m, planar, kinds, side, lds };
}
static GeomConfig plan_config(const WavePlan &P, int target_format, int maxW, int maxH, int n_layers_total) {
    return GeomConfig{ target_format, P.WTH, P.m.p0pitch, P.m.p0rows, P.m.p1pitch, P.m.p1rows, (P.planar ? 1 : 0) | P.side, (maxW + WTW - 1) / WTW, (maxH + P.WTH - 1) / P.WTH, n_layers_total };
}

bool geom_store_patch(int target_format, const DTick *ticks_host, DLayer *layers_host, int n_ticks, int maxW, int maxH, int n_layers_total,
                      GeomConfig *cfg_out, bool *want_build) {
    *want_build = false;
    const int mode = CHV_GEOM_CACHE ? switches().geom_cache.load(std::memory_order_relaxed) : 0;
    if (n_ticks < 1 || !layers_host) return false;
    const WavePlan P = plan_wave_layers(target_format, ticks_host, layers_host, n_ticks, maxW, maxH);
    *cfg_out = plan_config(P, target_format, maxW, maxH, n_layers_total);
    auto zero = [&]() {
        for (int i = 0; i < n_ticks; i++)
            for (int l = 0; l < ticks_host[i].n_layers; l++) { DLayer &L = layers_host[ticks_host[i].first_layer + l]; L.pad2[0] = L.pad2[1] = 0; }
    };
    if (!mode || P.kinds == 4 || (P.kinds & 7) == 0) { zero(); return false; }        // (off; launches of RGB layers only keep computing in place; nothing staged)
    // a lone tick of the scene this thread's last covered lone tick was
    constexpr int MEMO_MAX = WAVE_ONE_LAYERS + 2;
    const bool lone = n_ticks == 1 && ticks_host[0].n_layers >= 1 && ticks_host[0].n_layers <= MEMO_MAX;
    GeomRawKey raw[MEMO_MAX];
    bool has[MEMO_MAX];
    int dev = 0;
    if (lone) {
        if (hipGetDevice(&dev) != hipSuccess) { (void)hipGetLastError(); dev = 0; }
        const DTick &T = ticks_host[0];
        for (int l = 0; l < T.n_layers; l++) has[l] = geom_class_raw(T, layers_host[T.first_layer + l], raw[l]);
        GeomLoneMemo &m = geom_lone_memo();
        bool same = m.dev == dev && m.n == T.n_layers && m.tf == target_format && m.cfg == *cfg_out;
        for (int l = 0; same && l < T.n_layers; l++) same = m.has[l] == has[l] && (!has[l] || memcmp(&m.key[l], &raw[l], sizeof raw[l]) == 0);
        if (same) {
            for (int l = 0; l < T.n_layers; l++) {
                DLayer &L = layers_host[T.first_layer + l];
                const uint64_t tp = has[l] ? (uint64_t)(uintptr_t)m.tab[l] : 0;
                L.pad2[0] = (int32_t)(uint32_t)(tp & 0xFFFFFFFFu); L.pad2[1] = (int32_t)(uint32_t)(tp >> 32);
            }
            geom_store().patched.fetch_add(1, std::memory_order_relaxed);
            return true;
        }
    }
    const std::string ck = geom_config_key(*cfg_out);
    GeomStore &st = geom_store();
    std::vector<std::pair<int, void *>> hits;
    bool all = true, known = true;
    int classes = 0;
    {
        std::lock_guard<std::mutex> lk(st.mu);
        // this call's classes (a sighting is a LAUNCH that asked, whatever its number of ticks): the first sixteen as raw keys compared by bytes,
        // starting from where the same layer of the previous tick was found (a group's ticks repeat their predecessor's geometry: one memcmp
        // per layer — a batch of 128 mixer ticks used to pay 27 us of strings and hashing here); further ones through a map
        struct Seen { GeomRawKey k; void *tab; };
        Seen seen[16];
        int n_seen = 0;
        std::string ks;
        std::unordered_map<std::string, void *> asked;
        auto lookup = [&](const std::string &key) -> void * {
            const std::string full = key + ck;
            auto it = st.tables.find(full);
            void *tab = it != st.tables.end() ? it->second : nullptr;
            if (!tab) {
                // (an animated layer is a new geometry every tick, seen once: the count of sightings is bounded by starting over)
                if (st.sightings.size() >= kGeomStoreSightings) st.sightings.clear();
                int &n = st.sightings[full];
                if (n < (1 << 20)) n++;
                if (n < 2 || st.full) known = false;
            }
            classes++;
            return tab;
        };
        for (int i = 0; i < n_ticks; i++) {
            const DTick &T = ticks_host[i];
            for (int l = 0; l < T.n_layers; l++) {
                const int li = T.first_layer + l;
                GeomRawKey rk;
                if (!geom_class_raw(T, layers_host[li], rk)) continue;
                void *tab = nullptr;
                int at = -1;
                for (int q = 0; q < n_seen && at < 0; q++) {
                    const int j = (l + q) % n_seen;                  // (layer l of a tick is usually class l of the group)
                    if (memcmp(&seen[j].k, &rk, sizeof rk) == 0) at = j;
                }
                if (at >= 0) tab = seen[at].tab;
                else {
                    ks.assign((const char *)&rk, sizeof rk);
                    if (n_seen < 16) {
                        tab = lookup(ks);
                        seen[n_seen].k = rk; seen[n_seen].tab = tab; n_seen++;
                    } else {
                        auto f = asked.find(ks);
                        if (f != asked.end()) tab = f->second;
                        else { tab = lookup(ks); asked.emplace(ks, tab); }
                    }
                }
                if (tab) hits.emplace_back(li, tab); else all = false;
            }
        }
    }
    if (!all || hits.empty()) {
        zero();
        *want_build = !hits.empty() || classes > 0 ? (known || mode == 2) && !all : false;
        return false;
    }
    zero();
    st.patched.fetch_add(1, std::memory_order_relaxed);
    for (auto &h : hits) {
        const uint64_t tp = (uint64_t)(uintptr_t)h.second;
        layers_host[h.first].pad2[0] = (int32_t)(uint32_t)(tp & 0xFFFFFFFFu); layers_host[h.first].pad2[1] = (int32_t)(uint32_t)(tp >> 32);
    }
    if (lone) {
        GeomLoneMemo &m = geom_lone_memo();
        const DTick &T = ticks_host[0];
        m.dev = dev; m.n = T.n_layers; m.tf = target_format; m.cfg = *cfg_out;
        for (int l = 0; l < T.n_layers; l++) {
            m.has[l] = has[l];
            if (has[l]) m.key[l] = raw[l];
            const DLayer &L = layers_host[T.first_layer + l];
            m.tab[l] = (void *)(uintptr_t)(((uint64_t)(uint32_t)L.pad2[1] << 32) | (uint64_t)(uint32_t)L.pad2[0]);
        }
    }
    return true;
}

// May this lone tick travel as a kernel argument (launch_transient: no descriptor slot, no copy)?  4:2:0 canvases: every instantiation of
// tick_yuv_wave takes one; BGRA canvases: tick_bgra_wave_one exists for cleared canvases, 8-row strips and launches without per-pixel layers.
bool wave_layers_by_value(int target_format, const DTick *tick_host, const DLayer *layers_host) {
    if (tick_host->n_layers < 1 || tick_host->n_layers > WAVE_ONE_LAYERS) return false;
    if (target_format != TF_BGRA) return true;
    if (!tick_host->clear_first) return false;
    const WavePlan P = plan_wave_layers(target_format, tick_host, layers_host, 1, tick_host->W, tick_host->H);
    return P.WTH == 8 && !(P.kinds & 8) && P.kinds != 4;
}

hipError_t launch_wave_layers(int target_format, const DTick *ticks_host, const DLayer *layers_host, const DTick *ticks, const DLayer *layers,
                              int n_ticks, int maxW, int maxH, hipStream_t stream) {
    const WavePlan P = plan_wave_layers(target_format, ticks_host, layers_host, n_ticks, maxW, maxH);
    const int WTH = P.WTH, kinds = P.kinds, side = P.side;
    WaveDims m = P.m;
    const bool planar = P.planar;
    size_t lds = P.lds;
    int strips_x = (maxW + WTW - 1) / WTW, strips_y = (maxH + WTH - 1) / WTH;
    const bool clear = ticks_host[0].clear_first != 0;
    // A launch that continues on canvases something else composed (the second launch of a split batch: a logo or overlays over the videos the
    // streaming kernel did; layers added to a canvas outside a clear) only has work where its layers are: the grid covers the strips the union of
    // the layers' bounding boxes touches instead of the canvas (a 320 x 180 logo on 720p: ~40 of 1800 strips per tick; the other waves used to
    // start, find no layer and leave — 115 us of a 730 us pipeline_logo batch).  The origin travels in the high halves of the two row counts.
    int origin_x = 0, origin_y = 0;
    if (!clear && CHV_WAVE_BBOX_GRID) {
        int bx0 = maxW, by0 = maxH, bx1 = 0, by1 = 0;
        for (int i = 0; i < n_ticks; i++)
            for (int l = 0; l < ticks_host[i].n_layers; l++) {
                const int32_t *bb = layers_host[ticks_host[i].first_layer + l].bbox;
                bx0 = std::min(bx0, std::max(bb[0], 0)); by0 = std::min(by0, std::max(bb[1], 0));
                bx1 = std::max(bx1, std::min(bb[2], maxW)); by1 = std::max(by1, std::min(bb[3], maxH));
            }
        if (bx1 > bx0 && by1 > by0) {
            origin_x = bx0 / WTW; origin_y = by0 / WTH;
            strips_x = (bx1 + WTW - 1) / WTW - origin_x; strips_y = (by1 + WTH - 1) / WTH - origin_y;
        } else { strips_x = 1; strips_y = 1; }                     // (nothing visible: one strip per tick, which finds no layer and leaves)
        if (origin_x > 0xFFFF || origin_y > 0xFFFF) { origin_x = origin_y = 0; strips_x = (maxW + WTW - 1) / WTW; strips_y = (maxH + WTH - 1) / WTH; }
    }
    const int p0rows_arg = m.p0rows | (origin_x << 16), p1rows_arg = m.p1rows | (origin_y << 16);
    // the batch's geometry tables for this configuration (a transient launch has none: its kernels compute their geometry in place)
    bool cached = false;                       // -> the CACHED instantiations (no set-up code): every staged layer of the launch has its table
    // (not for launches of RGB layers only: stacks of one geometry are set up once per strip for all their layers — cfg3 -1 %, cfg5 +-0 with
    // tables, for 5 % more counted traffic: profiles/r06_notes.md section 9)
    GeomCache *gc = kinds != 4 ? geom_cache_current() : nullptr;
    if (gc) {
        const GeomConfig cfg = plan_config(P, target_format, maxW, maxH, gc->n_layers);
        hipError_t ge = geom_cache_prepare(*gc, cfg, ticks_host, n_ticks, (size_t)WTH * 48, stream, &cached);
        if (ge != hipSuccess) return ge;
    } else if (kinds != 4) {
        // a lone tick: geom_store_patch pointed its layers at the store's tables before they went to the device, for exactly this shape
        GeomTransient &gt = geom_transient_current();
        cached = gt.covered && gt.cfg == plan_config(P, target_format, maxW, maxH, gt.cfg.n_layers);
    }
    // floor(2^32 / d) for the kernels' scalar divisions by the strips per tick and per row (WaveStrip::udivmod)
    auto magic = [](uint32_t d) { return d <= 1 ? 0xFFFFFFFFu : (uint32_t)((1ull << 32) / d); };
    const uint32_t strips_magic = magic((uint32_t)(strips_x * strips_y)), strips_x_magic = magic((uint32_t)strips_x);
    long total = (long)n_ticks * strips_x * strips_y;
    long per_xcd = (total + 7) / 8;
    long blocks_per_xcd = (per_xcd + WAVES - 1) / WAVES;
    dim3 grid((unsigned)(blocks_per_xcd * 8));
    // a lone tick without descriptors in device memory (launch_transient, which asked wave_layers_by_value first): they travel as the 4:2:0
    // kernels' last argument / as the first argument of tick_bgra_wave_one
    WaveOne one_store;
    const WaveOne *one = nullptr;
    if (!ticks) {
        if (n_ticks != 1 || ticks_host[0].n_layers > WAVE_ONE_LAYERS) return hipErrorInvalidValue;
        one_store.t = ticks_host[0];
        one_store.t.first_layer = 0;
        if (one_store.t.n_layers > 0) memcpy(one_store.l, layers_host + ticks_host[0].first_layer, sizeof(DLayer) * (size_t)one_store.t.n_layers);
        one = &one_store;
        layers = nullptr;
    }
    if (target_format == TF_BGRA) {
        return launch_bgra_wave(WTH, clear, grid, lds, stream, ticks, layers, n_ticks, strips_x, strips_y, strips_magic, strips_x_magic, m.p0pitch, p0rows_arg, m.p1pitch, p1rows_arg,
                                (planar ? 1 : 0) | side, kinds, cached, one);
    }
    return cached ? launch_yuv_wave_t<true>(target_format, clear, WTH, kinds, grid, lds, stream, ticks, layers, n_ticks, strips_x, strips_y, strips_magic, strips_x_magic,
                                            m.p0pitch, p0rows_arg, m.p1pitch, p1rows_arg, (planar ? 1 : 0) | side, one)
                  : launch_yuv_wave_t<false>(target_format, clear, WTH, kinds, grid, lds, stream, ticks, layers, n_ticks, strips_x, strips_y, strips_magic, strips_x_magic,
                                             m.p0pitch, p0rows_arg, m.p1pitch, p1rows_arg, (planar ? 1 : 0) | side, one);
}
#endif      // CHV_WAVE_TU == 0

}  // namespace chv

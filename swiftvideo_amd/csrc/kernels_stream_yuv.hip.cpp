// kernels_stream_yuv.hip.cpp — tick_yuv_stream: ticks of 1..4 axis-aligned layers onto a cleared 4:2:0 canvas (NV12 / y420p), canvas rows
// outermost and the layers innermost, every layer's source rows streamed through small LDS rings by LDS-DMA.
//
// These are the reference's own kernels (img_nv12_nv12, img_y420p_nv12, img_y420p_y420p, img_{bgra,rgba}_{nv12,y420p},
// kernels.cl.swift:186-255,267-335,469-532) on the canvas format the reference defaults to (composer.swift:52-56), plus the integer
// RGB -> YUV kernels of DESIGN.md 4.5.  tick_yuv_wave (kernels_wave_yuv.hip.cpp) runs them one wave per 64 x 16 strip: per strip and
// layer a geometry evaluation, a rectangle staged through registers into LDS, sixteen unrolled rows — about 380 ns of fixed work per
// strip against 44 ns per pixel row of an opaque layer (profiles/r03_notes.md section 3).  Here the structure that took the BGRA
// headline from 0.31 to 0.41 of the HBM roofline (kernels_stream.hip.cpp) is applied to them, generalised to layers of DIFFERENT
// geometry (a full-canvas video under small overlays is the reference mixer's usual tick):
//   * a wave owns a 64-column strip over a chunk of rows (one wave per block, no barrier: the LDS is handed out per wave) and works in
//     TRIPS of four canvas rows: the lane's four luma codes of a trip are one packed register, the trip's chroma sample of the lane's
//     column pair one code each for U and V (even lanes: chroma row 2m, odd lanes: 2m + 1 — the reference's `handleChroma` owner is the
//     quad's even/even pixel, kernels.cl.swift:76, so both lanes sample at the EVEN lane's column);
//   * every layer has its own column entry (registers), its own row table (LDS, 32 rows at a time, with a two-dword summary per 8-row
//     STEP) and its own rings: luma 24 rows x 128 B in batches of 8, chroma 12 rows in batches of 4 (NV12: 128-byte rows of (u, v) pairs;
//     planar: U and V rows of 96 bytes side by side, one load instruction for both), RGB texels 12 (encoder side) or 8 (beside video
//     layers) rows x 320 B in batches of 3 or 2.  A batch is one `global_load_lds_dwordx4` (lane -> row, vector; LDS address M0 + lane x
//     16), requested as soon as the taps have left the oldest batch and awaited BY COUNT (`s_waitcnt vmcnt(n)`, n = loads issued since:
//     loads complete in order among themselves) when a tap row reaches it.  Residency is decided once per step for the YUV layers, per
//     row for RGB layers (their rows are 2.5x as long);
//   * the bulk of a mixer tick — steps in which only an opaque same-size video layer touches the strip — and the encoder-side frame take
//     their own short loops (fast_step, fast_rgb_trip: no per-layer tests, the lower tap row of a pixel carried down the lane as the
//     upper tap row of the pixel below); the operations per pixel are the general loops';
//   * layers that miss the strip's columns are dropped at the chunk start, trips outside a layer's rows cost a flag test;
//   * the luma codes of a trip leave as ONE dword store per lane (a 4 x 4 byte transpose inside every quad of lanes), issued after the
//     NEXT trip's first wait (gfx950 counts loads and stores in one counter); chroma leaves once per 16 rows as in tick_yuv_wave.
// The arithmetic per pixel and layer is tick_yuv_wave's, operation for operation (the reference's unit-scale Khronos arithmetic:
// c / 255 correctly rounded per tap, unfused sums in source order; code-scale fused taps and the 16.16 matrix for the `_int`
// kind), so the bytes are those of oracle/ref_kernels.c::px_yuv_to_yuv / px_rgb_to_yuv / px_rgb_to_yuv_int, layer by layer.
//
// Eligibility (yuv_stream_eligible): cleared canvas with W % 8 == 0 and H % 4 == 0, 1..4 layers, every layer axis-aligned, bounded,
// without fill paint and without flips, horizontal reduction <= 1.7 (YUV) / 1.17 (RGB), vertical <= 2.1 (a step of 8 rows must span at most 15 source rows: yuv_stream_eligible), source rows a multiple of
// 16 bytes, at most 16 KB of rings per wave.  Everything else keeps tick_yuv_wave.
#include "wave_common.hip.h"
#include "switches.h"

#include <algorithm>
#include <cmath>
#include <cstring>
#include <type_traits>
#include <utility>

#pragma clang fp contract(off)

namespace chv {

#ifndef CHV_YS_ABL
#define CHV_YS_ABL 0             // timing-only (wrong pixels): 1 no ring fills, 2 no canvas stores, 4 no luma stores, 8 no chroma stores
#endif
#ifndef CHV_YS_BLOCK
#define CHV_YS_BLOCK 1           // waves (neighbouring strips) per block: 1 / 2 / 4 = 0.370 / 0.374 / 0.404 ms per 128 ticks of y420p_main (the kernel is
                                 // bound by its instructions, not by its traffic: the finer the LDS is handed out, the better)
#endif
#ifndef CHV_YS_ROUNDS
#define CHV_YS_ROUNDS 12         // chunk height of large launches: enough chunks for this many rounds of waves
#endif
#ifndef CHV_YS_ROWS_FIXED
#define CHV_YS_ROWS_FIXED 0
#endif
#ifndef CHV_YS_SMALL_WAVES
#define CHV_YS_SMALL_WAVES 4800  // waves a small launch is cut into
#endif
#ifndef CHV_YS_WAVES_OWN
#define CHV_YS_WAVES_OWN 6       // waves per SIMD the own-format instantiations are compiled for
#endif
#ifndef CHV_YS_WAVES_MIXED
#define CHV_YS_WAVES_MIXED 4       // (the rings of a video layer and two overlays are 10 KB per wave: the LDS holds four waves per SIMD anyway, and at 96
                                 // registers the three-layer instantiation kept a dozen in scratch — scratch accesses share `vmcnt` with the ring
                                 // fills, and hipcc drains the counter in front of every reload: 0.98 -> 1.21 ms per 128 mixer ticks)
#endif
#ifndef CHV_YS_STORE
#define CHV_YS_STORE 0           // luma stores: 0 — a lane writes 4 columns of row (lane & 3) (16-byte pieces of four rows per quarter wave);
                                 // 1 — one more lane shuffle first: a quarter wave writes 64 contiguous bytes of ONE row; bit 1 (2, 3): nontemporal
#endif
#ifndef CHV_YS_RGBFAST
#define CHV_YS_RGBFAST 1         // the encoder-side frame's own trip loop (fast_rgb_trip); 0: the general row loop (A/B)
#endif
#ifndef CHV_YS_WAVES_RGBINT
#define CHV_YS_WAVES_RGBINT 6
#endif
#ifndef CHV_YS_CARRY
#define CHV_YS_CARRY 1           // native-resolution rows: a pixel's lower tap row is the upper one of the pixel below (conversions carried)
#endif

constexpr int YS_WAVES = CHV_YS_BLOCK;
constexpr int YS_TAB = 32;                       // row entries per table fill (lane = row)
// per layer: packed (row positions + flags), luma / RGB row weight, chroma row weight — 16 dwords each — and two dwords per 8-row step
constexpr int YS_TAB_DW = 3 * YS_TAB + 2 * (YS_TAB / 8);
constexpr int YS_TAB_BYTES = YS_TAB_DW * 4;
constexpr int YS_MAXL = 4;

// kinds a launch contains (template parameter KINDS; the instantiation holds no code for the others)
enum { YK_NV12 = 1, YK_PLANAR = 2, YK_RGB = 4, YK_RGBINT = 8 };

// ---- ring classes: rows of PV vectors, filled B rows at a time (one load instruction), NB batches; SPLIT > 0: vectors [0, SPLIT) of a row come
//      from plane a, the rest from plane b (planar chroma: a U row and a V row side by side).  Rows lie linearly: position p at p x PITCH. ----
template <int PV_, int B_, int NB_, int SPLIT_>
struct RingClass {
    static constexpr int PV = PV_, B = B_, NB = NB_, SPLIT = SPLIT_;
    static constexpr int PITCH = PV_ * 16, SLOT = B_ * PITCH, ROWS = NB_ * B_, BYTES = ROWS * PITCH;
    static_assert(B_ * PV_ <= 64, "one load instruction per batch");
    static_assert(ROWS <= 32 && NB_ <= 4, "ring bookkeeping: a byte per batch");
};
using RingY = RingClass<8, 8, 3, 0>;             // luma: 24 rows x 128 B, eight rows per load
using RingC2 = RingClass<8, 4, 3, 0>;            // NV12 chroma: 12 rows of (u, v) pairs
using RingCP = RingClass<12, 4, 3, 6>;           // planar chroma: 12 rows of U (96 B) | V (96 B)
// RGB texels, rows of 320 B: four batches of three rows in RGB-only launches (the encoder side), of two rows where video layers share the LDS
// (a mixer's overlays: the ring follows every row, and with two batches of three a row's request was two rows ahead of its taps — 600 ns of
// waiting per overlay row)
template <int B> using RingRGB = RingClass<20, B, 4, 0>;
constexpr int ys_rgb_b(int kinds) { return (kinds & (YK_NV12 | YK_PLANAR)) ? 2 : 3; }

CHV_DEV bool ys_is_rgb(int kind) { return kind == LK_YUV_FROM_RGB || kind == LK_YUV_FROM_RGB_INT; }
constexpr int ys_layer_bytes_c(bool rgb, bool planar, int rgb_b) {
    return rgb ? rgb_b * 4 * 320 : RingY::BYTES + (planar ? RingCP::BYTES : RingC2::BYTES);
}

struct RingState {
    int a0;          // first virtual source row of the window (rows -1 .. h: CLAMP_TO_EDGE through the row of the address); far below 0: not started
    uint32_t seq;    // the load counter right after the request of each batch of the window, a byte each, oldest batch in byte 0
    int last;        // last row any tap of the chunk can read: nothing beyond it is requested
    int misc;        // bits 0-7: ring position (row) of a0; bits 8-10: batches of the window, from a0, that have arrived
};
constexpr int YS_NOT_STARTED = -(1 << 29);

// One batch: lane -> (row of the batch, vector); the lane's 16 bytes go from plane base + its own 32-bit byte offset to LDS at M0 + lane x 16
// (tools/probe_lds_dma.cpp).  M0 is set by hand right in front of the load; tests/test_device_code_contract.py checks that nothing else in
// this object touches it.
// (None of the asm statements of this file clobbers "memory": after such a statement hipcc re-reads every tick / layer descriptor field it
// needs with a scalar load and a full wait — six of them per trip put ~450 ns of waiting into every canvas row of the first version.  The LDS
// side is ordered by wave_lds_fence, which names the LDS address space only; volatile asm statements keep their order among themselves.)
CHV_DEV void ys_dma(const uint8_t *base, uint32_t voff, bool active, uint32_t m0) {
    // (the scalar operands pass through v_readfirstlane: when hipcc has moved a uniform value's computation to the vector unit, an "s" constraint
    // does not bring it back — it substitutes the VGPR into the text, which does not assemble)
    const uint32_t m0s = (uint32_t)__builtin_amdgcn_readfirstlane((int)m0);
    const uint64_t b = (uint64_t)(uintptr_t)base;
    const uint64_t bs = (uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)b) | ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(b >> 32)) << 32);
    if (active && !(CHV_YS_ABL & 1))
        asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" :: "s"(m0s), "v"(voff), "s"(bs));
}
// Loads complete in order among themselves: once at most `younger` vector-memory operations are outstanding, where `younger` loads were issued
// after the batch in question, that batch has landed — whatever the canvas stores in between did (they share the counter, complete out of
// order and can only lengthen the wait).  `s_waitcnt` takes its count as an immediate: a compare chain, written out by hand (hipcc turns the
// same chain of uniform ifs into three times the scalar instructions, and this kernel's first version was bound by its scalar instructions —
// the scalar unit is shared by the four SIMDs of a CU).
CHV_DEV void ys_await(int younger) {
    asm volatile(
        "s_cmp_lt_u32 %0, 4\n\t"
        "s_cbranch_scc1 1f\n\t"
        "s_cmp_lt_u32 %0, 6\n\t"
        "s_cbranch_scc1 2f\n\t"
        "s_cmp_lt_u32 %0, 7\n\t"
        "s_cbranch_scc1 3f\n\t"
        "s_waitcnt vmcnt(7)\n\t"
        "s_branch 9f\n"
        "3:\n\ts_waitcnt vmcnt(6)\n\t"
        "s_branch 9f\n"
        "2:\n\ts_cmp_lt_u32 %0, 5\n\t"
        "s_cbranch_scc1 4f\n\t"
        "s_waitcnt vmcnt(5)\n\t"
        "s_branch 9f\n"
        "4:\n\ts_waitcnt vmcnt(4)\n\t"
        "s_branch 9f\n"
        "1:\n\ts_cmp_lt_u32 %0, 2\n\t"
        "s_cbranch_scc1 5f\n\t"
        "s_cmp_lt_u32 %0, 3\n\t"
        "s_cbranch_scc1 6f\n\t"
        "s_waitcnt vmcnt(3)\n\t"
        "s_branch 9f\n"
        "6:\n\ts_waitcnt vmcnt(2)\n\t"
        "s_branch 9f\n"
        "5:\n\ts_cmp_lt_u32 %0, 1\n\t"
        "s_cbranch_scc1 7f\n\t"
        "s_waitcnt vmcnt(1)\n\t"
        "s_branch 9f\n"
        "7:\n\ts_waitcnt vmcnt(0)\n"
        "9:\n" :: "s"(__builtin_amdgcn_readfirstlane(younger)) : "scc");
    wave_lds_fence();
}
// before a batch is overwritten: the taps of its rows have been read
CHV_DEV void ys_taps_done() {
    wave_lds_fence();
    asm volatile("s_waitcnt lgkmcnt(0)");
}

// What a ring is filled from: up to two planes of one shape (planar chroma: U and V); the strip's first byte column and the vectors of a row
// some tap can read travel packed (col0 | nvec << 24) in the layer's state
struct RingSrc {
    const uint8_t *pa, *pb;
    int pitch, h, rowbytes;      // source plane: bytes per row step, rows, payload bytes per row (a multiple of 16)
    int col0, nvec;
};
CHV_DEV RingSrc ring_src(const DPlane &a, const DPlane &b, int colvec) {
    return RingSrc{ a.ptr, b.ptr, a.pitch, a.h, a.w * a.comps, colvec & 0xFFFFFF, (int)((uint32_t)colvec >> 24) };
}

// rows row0 .. row0 + B - 1 into the batch at LDS address slot_addr; returns the load instructions issued
template <class RC>
CHV_DEV int ring_request(uint32_t slot_addr, const RingSrc &s, int row0, int last, int lane) {
    int q, vv;
    if constexpr ((RC::PV & (RC::PV - 1)) == 0) { q = lane / RC::PV; vv = lane & (RC::PV - 1); }      // (shifts and masks)
    else { q = (lane * (256 / RC::PV + 1)) >> 8; vv = lane - q * RC::PV; static_assert(((63 * (256 / RC::PV + 1)) >> 8) == 63 / RC::PV, "lane / PV"); }
    const int r = min(max(row0 + q, 0), s.h - 1);
    const uint32_t roff = __umul24((uint32_t)r, (uint32_t)s.pitch);
    if constexpr (RC::SPLIT > 0) {
        // a row of the ring: SPLIT vectors of plane a, then the same of plane b (one shape, host-checked).  Both halves are addressed from plane
        // a's base, the distance between the planes travels in the lane's 32-bit offset (planes of one allocation; unrelated allocations more
        // than 2 GB apart take an instruction each)
        const int pl = vv >= RC::SPLIT ? 1 : 0, vp = vv - pl * RC::SPLIT;
        const bool active = q < RC::B && vp < s.nvec && row0 + q <= last;
        const uint32_t off = roff + (uint32_t)min(s.col0 + 16 * vp, s.rowbytes - 16);
        const long long dist = (long long)(s.pb - s.pa);
        if (dist >= 0 && dist < (1ll << 31)) {
            ys_dma(s.pa, off + (pl ? (uint32_t)dist : 0u), active, slot_addr);
            return 1;
        }
        ys_dma(s.pa, off, active && pl == 0, slot_addr);
        ys_dma(s.pb, off, active && pl == 1, slot_addr);
        return 2;
    } else {
        const bool active = q < RC::B && vv < s.nvec && row0 + q <= last;
        ys_dma(s.pa, roff + (uint32_t)min(s.col0 + 16 * vv, s.rowbytes - 16), active, slot_addr);
        return 1;
    }
}

// Source rows lo .. hi of the layer resident in the ring (lo <= hi, hi - lo <= ROWS - B; all arguments wave-uniform).  `src()` yields the RingSrc
// (descriptor reads: only evaluated when a batch is requested).
template <class RC, class SrcFn>
CHV_DEV void ring_ensure(RingState &R, uint32_t lds_base, SrcFn &&src, int lo, int hi, int &issued, int lane) {
    int pos0 = R.misc & 255, landk = R.misc >> 8;
    if (lo >= R.a0 + 2 * RC::ROWS) {
        // first use in this chunk (or a jump far past the window): the window starts at lo
        ys_taps_done();
        const RingSrc s = src();
        R.a0 = lo; pos0 = 0; landk = 0; R.seq = 0;
#pragma unroll 1
        for (int k = 0; k < RC::NB; k++) {
            const int row0 = lo + k * RC::B;
            if (row0 <= R.last) issued += ring_request<RC>(lds_base + (uint32_t)(k * RC::SLOT), s, row0, R.last, lane);
            R.seq |= ((uint32_t)issued & 255u) << (8 * k);
        }
    } else if (lo >= R.a0 + RC::B) {
        // the taps have left the oldest batch(es): their rows of the ring take the rows behind the window
        ys_taps_done();
        const RingSrc s = src();
        do {
            const int row0 = R.a0 + RC::ROWS;
            if (row0 <= R.last) issued += ring_request<RC>(lds_base + (uint32_t)(pos0 * RC::PITCH), s, row0, R.last, lane);
            R.seq = (R.seq >> 8) | (((uint32_t)issued & 255u) << (8 * (RC::NB - 1)));
            pos0 = pos0 + RC::B == RC::ROWS ? 0 : pos0 + RC::B;
            R.a0 += RC::B;
            landk = landk > 0 ? landk - 1 : 0;
        } while (lo >= R.a0 + RC::B);
    }
    if (hi >= R.a0 + landk * RC::B) {
        const int d = hi - R.a0;
        int k;
        if constexpr ((RC::B & (RC::B - 1)) == 0) k = d / RC::B;
        else { static_assert(RC::B == 3 && RC::ROWS <= 16, "d / 3 by multiplication, d < 16"); k = ((d & 15) * 11) >> 5; }
        k = k > RC::NB - 1 ? RC::NB - 1 : k;
        ys_await((int)(((uint32_t)issued - (R.seq >> (8 * k))) & 255u));
        landk = k + 1;
    }
    R.misc = pos0 | (landk << 8);
}
// The same for the usual step of a picture drawn at its own size — the taps have left exactly the oldest batch, one batch follows the window —
// without the loops and the general bookkeeping (this kernel is bound by the instructions it issues, the scalar ones included); anything else
// takes ring_ensure
template <class RC, class SrcFn>
CHV_DEV void ring_ensure_step(RingState &R, uint32_t lds_base, SrcFn &&src, int lo, int hi, int &issued, int lane) {
    static_assert(RC::NB == 3, "three batches");
    const int d = lo - R.a0;
    if (d >= RC::B && d < 2 * RC::B) {
        int pos0 = R.misc & 255, landk = R.misc >> 8;
        ys_taps_done();
        const int row0 = R.a0 + RC::ROWS;
        if (row0 <= R.last) { const RingSrc s = src(); issued += ring_request<RC>(lds_base + (uint32_t)(pos0 * RC::PITCH), s, row0, R.last, lane); }
        R.seq = (R.seq >> 8) | (((uint32_t)issued & 255u) << 16);
        pos0 = pos0 + RC::B == RC::ROWS ? 0 : pos0 + RC::B;
        R.a0 += RC::B;
        landk = landk > 0 ? landk - 1 : 0;
        const int e = hi - R.a0;
        const int k = e >= 2 * RC::B ? 2 : e >= RC::B ? 1 : 0;
        if (k >= landk) {
            ys_await((int)(((uint32_t)issued - (R.seq >> (8 * k))) & 255u));
            landk = k + 1;
        }
        R.misc = pos0 | (landk << 8);
    } else {
        ring_ensure<RC>(R, lds_base, src, lo, hi, issued, lane);
    }
}
// LDS byte offset (inside the ring) of source row r, a0 <= r < a0 + ROWS
template <class RC>
CHV_DEV int ring_row(const RingState &R, int r) {
    // (wave-uniform, written without min / max so that it stays on the scalar unit — hipcc selects v_med3_i32 for a clamp and the whole
    // address chain then runs once per lane)
    int p = (R.misc & 255) + (r - R.a0);
    p = p >= RC::ROWS ? p - RC::ROWS : p;
    return __builtin_amdgcn_readfirstlane(p * RC::PITCH);
}
// the next row of the ring
template <class RC>
CHV_DEV int ring_next(int off) { return off + RC::PITCH == RC::BYTES ? 0 : off + RC::PITCH; }

// ---- pixel arithmetic (tick_yuv_wave's, kernels_wave_yuv.hip.cpp) ----------------------------------------------------------------
CHV_DEV float ys_t8(uint32_t byte) { return unorm8(byte); }                       // c / 255.0f, correctly rounded
template <int K>
CHV_DEV float ys_t8k(uint32_t w) { return unorm8f(K == 0 ? (float)(w & 255u) : K == 1 ? (float)((w >> 8) & 255u) : K == 2 ? (float)((w >> 16) & 255u) : (float)(w >> 24)); }
template <int K>
CHV_DEV float ys_ubk(uint32_t w) { return K == 0 ? ub0(w) : K == 1 ? ub1(w) : K == 2 ? ub2(w) : ub3(w); }
// convert_uchar_sat_rte(f * 255) into byte K of w (v_cvt_pk_u8_f32: RTE, clamp to [0, 255], NaN -> 0 = to_code)
template <int K>
CHV_DEV uint32_t ys_put(uint32_t w, float f) {
    const float v = f * 255.0f;
    if (K == 0) asm("v_cvt_pk_u8_f32 %0, %1, 0, %0" : "+v"(w) : "v"(v));
    if (K == 1) asm("v_cvt_pk_u8_f32 %0, %1, 1, %0" : "+v"(w) : "v"(v));
    if (K == 2) asm("v_cvt_pk_u8_f32 %0, %1, 2, %0" : "+v"(w) : "v"(v));
    if (K == 3) asm("v_cvt_pk_u8_f32 %0, %1, 3, %0" : "+v"(w) : "v"(v));
    return w;
}
template <int K>
CHV_DEV uint32_t ys_put_raw(uint32_t w, float v) {
    if (K == 0) asm("v_cvt_pk_u8_f32 %0, %1, 0, %0" : "+v"(w) : "v"(v));
    if (K == 1) asm("v_cvt_pk_u8_f32 %0, %1, 1, %0" : "+v"(w) : "v"(v));
    if (K == 2) asm("v_cvt_pk_u8_f32 %0, %1, 2, %0" : "+v"(w) : "v"(v));
    if (K == 3) asm("v_cvt_pk_u8_f32 %0, %1, 3, %0" : "+v"(w) : "v"(v));
    return w;
}
CHV_DEV float ys_mix4(float w00, float w10, float w01, float w11, float t00, float t10, float t01, float t11) {
    return ((w00 * t00 + w10 * t10) + w01 * t01) + w11 * t11;      // lin_mix's order (OpenCL 1.2 section 8.2)
}
CHV_DEV int ys_dpp_even(int v) { return __builtin_amdgcn_update_dpp(v, v, 0xA0 /* quad_perm [0,0,2,2] */, 0xf, 0xf, false); }
CHV_DEV float ys_dpp_even(float v) { return __int_as_float(ys_dpp_even(__float_as_int(v))); }

template <typename F, int... J>
CHV_DEV void ys_seq_impl(F &f, std::integer_sequence<int, J...>) { (f(std::integral_constant<int, J>{}), ...); }
template <int N, typename F>
CHV_DEV void ys_seq(F &f) { ys_seq_impl(f, std::make_integer_sequence<int, N>{}); }

// per-lane column entry of one layer
struct YsCol {
    uint32_t off;      // byte offsets of the two tap columns inside a ring row, CLAMP_TO_EDGE resolved: plane 0 (luma byte / RGB texel) in bits
                       // 0-8 and 9-17, chroma at the column pair's EVEN lane in bits 18-24 and 25-31
    float a, ca;       // weight of tap column 1 (luma / RGB; chroma at the even lane)
};
CHV_DEV int ys_o0(uint32_t off) { return (int)(off & 511u); }
CHV_DEV int ys_o1(uint32_t off) { return (int)((off >> 9) & 511u); }
CHV_DEV int ys_c0(uint32_t off) { return (int)((off >> 18) & 127u); }
CHV_DEV int ys_c1(uint32_t off) { return (int)(off >> 25); }

// v_cvt_pk_u8_f32 with the byte position in a register (wave-uniform k)
CHV_DEV uint32_t ys_put_raw_k(uint32_t w, float v, int k) {
    asm("v_cvt_pk_u8_f32 %0, %1, %2, %0" : "+v"(w) : "v"(v), "s"(k));
    return w;
}

// waves per SIMD an instantiation is compiled for
// (the encoder-side instantiation carries eight texel values down the lane across trips: 96 registers, nothing in scratch)
constexpr int ys_min_waves(int kinds, int nl) { return (kinds & (YK_RGB | YK_RGBINT)) ? (kinds == YK_RGBINT ? CHV_YS_WAVES_RGBINT : CHV_YS_WAVES_MIXED) : nl >= 2 ? CHV_YS_WAVES_OWN - 1 : CHV_YS_WAVES_OWN; }

// ONE: one tick whose descriptors are kernel ARGUMENTS (tick_yuv_stream_one)
template <int TF, int NL, int KINDS, bool ONE>
CHV_DEV void ys_body(const DTick *__restrict__ ticks, const DLayer *__restrict__ layers, int n_ticks, int strips_x, int chunks_y, int rows_per_chunk, int wave_bytes) {
    static_assert(NL >= 1 && NL <= YS_MAXL, "layers per tick");
    constexpr int RGB_B = ys_rgb_b(KINDS);
    using RingR = RingRGB<RGB_B>;
    constexpr bool HAS_YUV = (KINDS & (YK_NV12 | YK_PLANAR)) != 0, HAS_RGB = (KINDS & (YK_RGB | YK_RGBINT)) != 0;
    extern __shared__ __attribute__((aligned(16))) uint8_t ys_lds_all[];
    const int wave = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
    uint8_t *lds = ys_lds_all + wave * wave_bytes;
    const uint32_t lds0 = (uint32_t)(size_t)lds;
    const int lane = threadIdx.x & 63;
    // XCD-aware numbering: block b runs on XCD b % 8; an XCD owns a contiguous range of blocks, a block YS_WAVES consecutive strips of the
    // launch's (tick, chunk, strip) list — neighbours along a row except where a row of strips ends inside the block
    const int per_tick = chunks_y * strips_x;
    const int total_w = n_ticks * per_tick, total = (total_w + YS_WAVES - 1) / YS_WAVES;
    const int b = blockIdx.x, per_xcd = (total + 7) >> 3;
    const int idx = (b & 7) * per_xcd + (b >> 3);
    if ((b >> 3) >= per_xcd || idx >= total) return;
    const int widx = idx * YS_WAVES + wave;
    if (widx >= total_w) return;
    const int tick = widx / per_tick, rem = widx - tick * per_tick;
    const int chunk = rem / strips_x, strip = rem - chunk * strips_x;
    const DTick &T = ticks[ONE ? 0 : tick];
    const DLayer *L = layers + (ONE ? 0 : T.first_layer);
    const int x0 = strip * 64, y0 = chunk * rows_per_chunk;
    const int TW = T.W, TH = T.H;
    if (x0 >= TW || y0 >= TH) return;
    const int nrows = min(rows_per_chunk, TH - y0);                   // (a multiple of 4: H % 4 == 0, chunks of 4 k rows)
    const int nl = min(T.n_layers, NL);
    const DPlane PY = T.dst.pl[0], PC = T.dst.pl[1], PV = T.dst.pl[TF == TF_Y420P ? 2 : 1];
    const float sx = (float)TW, sy = (float)TH;
    const int x = x0 + lane, xe = min(x, TW - 1);
    const bool col_in = x < TW;
    const float nx = ((float)xe / sx) * 2.f - 1.f;
    const int par = lane & 1;

    // (which source class a layer is, by the launch's KINDS where that settles it)
    auto layer_rgb = [&](const DLayer &Ly) { return HAS_RGB && (!HAS_YUV || ys_is_rgb(Ly.kind)); };
    auto layer_planar = [&](const DLayer &Ly) { return ((KINDS & YK_PLANAR) != 0) && (((KINDS & YK_NV12) == 0) || Ly.kind == LK_YUV_FROM_Y420P); };

    // ---- per layer: LDS slots, column entry, what its rings are filled from -------------------------------------------------------
    YsCol col[NL];
    RingState rY[NL], rC[NL];
    int cvY[NL], cvC[NL];       // col0 | nvec << 24 of the plane-0 / chroma ring (RingSrc)
    // one or two layer slots: what the rings are filled from stays in scalar registers (a request per trip and ring: re-reading the plane
    // descriptors each time put two scalar-load round trips into every trip's chain); deeper instantiations have no registers for that
    constexpr bool KEEP_SRC = NL <= 2;
    RingSrc sY[KEEP_SRC ? NL : 1], sC[KEEP_SRC ? NL : 1];
    int lbase[NL];              // LDS byte offset of the layer's rings (plane-0 ring first, then chroma)
    int trips[NL];              // chunk rows [lo, hi) the layer's bounding box covers: lo | hi << 16
    int lf[NL];                 // what the row loops ask of a layer, read once: bit 0 RGB, 1 planar, 2 integer matrix, 3 swizzle, 4 opaque; bits 8-9 colourspace
    float opac[NL];
    // of the 32 rows of the current table: rowm bit r — row r is inside the picture; unitm bit r — row r + 1 taps the source row behind row r's;
    // crowm bit r — the chroma row of luma row r + 2 is the one behind luma row r's; fullm bit r (r % 4 == 0) — the trip at row r takes the
    // short form of the row loops (luma: bit r, chroma: bit r + 1)
    uint32_t rowm[NL], unitm[NL], crowm[NL], fullm[NL];
    uint32_t picmask = 0;       // bit l: this lane's column is inside layer l's picture; bit 8 + l: the same for the pair's even lane
    int hit = 0;                // bit l: layer l can touch this strip's columns and this chunk's rows; bit 8 + l: its chroma as well
    int lds_used = 0;
    auto setup_layer = [&](auto lc) {
        constexpr int l = decltype(lc)::value;
        rY[l] = RingState{ YS_NOT_STARTED, 0u, 0, 0 }; rC[l] = RingState{ YS_NOT_STARTED, 0u, 0, 0 };
        col[l] = YsCol{ 0u, 0.f, 0.f };
        cvY[l] = 0; cvC[l] = 0; trips[l] = 0; lf[l] = 0; opac[l] = 0.f; rowm[l] = 0; unitm[l] = 0; crowm[l] = 0; fullm[l] = 0;
        if constexpr (KEEP_SRC) { sY[l] = RingSrc{ nullptr, nullptr, 0, 1, 16, 0, 0 }; sC[l] = sY[l]; }
        lbase[l] = lds_used;
        if (l >= nl) return;
        const DLayer &Ly = L[l];
        const bool rgb = layer_rgb(Ly);
        const bool planar = !rgb && layer_planar(Ly);
        lds_used += ys_layer_bytes_c(rgb, planar, RGB_B);
        {
            const bool is_int = rgb && ((KINDS & YK_RGBINT) != 0) && (((KINDS & YK_RGB) == 0) || Ly.kind == LK_YUV_FROM_RGB_INT);
            lf[l] = (rgb ? 1 : 0) | (planar ? 2 : 0) | (is_int ? 4 : 0) | (Ly.swizzle != 0 ? 8 : 0) | ((Ly.flags & LF_OPAQUE) ? 16 : 0) | ((Ly.csc & 3) << 8);
            opac[l] = Ly.u[U_OPACITY];
        }
        const int *bb = Ly.bbox;
        if (x0 + 64 <= bb[0] || x0 >= bb[2] || y0 + nrows <= bb[1] || y0 >= bb[3]) return;
        trips[l] = min(max(bb[1] - y0, 0), nrows) | (min(max(bb[3] - y0, 0), nrows) << 16);
        const float *U = Ly.u;
        const DPlane S0 = Ly.src.pl[0], S1 = Ly.src.pl[rgb ? 0 : 1];
        const float t3 = U[U_TRANSFORM + 15];
        const float t0 = nx * U[U_TRANSFORM + 0] + U[U_TRANSFORM + 3];
        const float b0 = nx * U[U_BORDER + 0] + U[U_BORDER + 3];
        const float u = t0 * U[U_TEXTURE + 0] + t3 * U[U_TEXTURE + 3];
        const int cfl = ((b0 >= 0.f && b0 <= 1.f) ? AX_BORDER : 0) | ((t0 >= 0.f && t0 <= 1.f) ? AX_TX : 0) | ((u >= 0.f && u <= 1.f) ? AX_UV : 0);
        int cy, cc;
        float cya, cca;
        lin_axis_raw(u, S0.w, cy, cya); lin_axis_raw(u, S1.w, cc, cca);
        const bool lane_pic = cfl == AX_ALL && col_in;
        const unsigned long long valid = __ballot(lane_pic);
        if (valid == 0) return;
        const int first = __ffsll((long long)valid) - 1, lastl = 63 - __clzll((long long)valid);
        const int bpt = rgb ? 4 : 1;
        const int pitch0 = rgb ? RingR::PITCH : RingY::PITCH, pv0 = rgb ? RingR::PV : RingY::PV;
        const int cy_f = min(max(rl(cy, first), 0), S0.w - 1), cy_l = min(max(rl(cy, lastl) + 1, 0), S0.w - 1);
        const int col0 = (cy_f * bpt) & ~15;
        const int o0 = min(max(min(max(cy, 0), S0.w - 1) * bpt - col0, 0), pitch0 - bpt);
        const int o1 = min(max(min(max(cy + 1, 0), S0.w - 1) * bpt - col0, 0), pitch0 - bpt);
        int c0 = 0, c1 = 0;
        col[l].a = cya;
        cvY[l] = col0 | (min(((cy_l * bpt + bpt - 1 - col0) >> 4) + 1, pv0) << 24);
        picmask |= lane_pic ? (1u << l) : 0u;
        hit |= 1 << l;
        // every column of the strip inside the picture, tap column 1 right behind tap column 0 and both weighted one half (a picture drawn at
        // its own size, away from its left and right edges): the row loops' short form
        if (valid == ~0ull && __ballot(o1 != o0 + bpt || cya != 0.5f) == 0) hit |= 1 << (16 + l);
        if constexpr (KEEP_SRC) sY[l] = ring_src(S0, S0, cvY[l]);
        // the chunk's last tap rows (nothing past them is requested: a chunk's overshoot is another wave's first rows)
        const int ye = min(y0 + nrows - 1, TH - 1);
        const float ny = ((float)ye / sy) * 2.f - 1.f;
        const float t1 = ny * U[U_TRANSFORM + 5] + U[U_TRANSFORM + 7];
        const float v = t1 * U[U_TEXTURE + 5] + t3 * U[U_TEXTURE + 7];
        int ry, rc;
        float a_;
        lin_axis_raw(v, S0.h, ry, a_); lin_axis_raw(v, S1.h, rc, a_);
        rY[l].last = min(max(__builtin_amdgcn_readfirstlane(ry), -1), S0.h - 1) + 1;
        if (!rgb) {
            // chroma: both lanes of a column pair sample at the EVEN lane's column
            const int cc_q = ys_dpp_even(cc);
            const float cca_q = ys_dpp_even(cca);
            const int pw = lane_pic ? 1 : 0;
            const bool pic_q = ys_dpp_even(pw) != 0;
            const unsigned long long validq = __ballot(pic_q);
            picmask |= pic_q ? (1u << (8 + l)) : 0u;
            col[l].ca = cca_q;
            if (validq != 0) {
                const int fq = __ffsll((long long)validq) - 1, lq = 63 - __clzll((long long)validq);
                const int bpc = planar ? 1 : 2;
                const int pitch1 = planar ? RingCP::PITCH : RingC2::PITCH, pv1 = planar ? RingCP::PV : RingC2::PV;
                const int cc_f = min(max(rl(cc_q, fq), 0), S1.w - 1), cc_l = min(max(rl(cc_q, lq) + 1, 0), S1.w - 1);
                const int ccol0 = (cc_f * bpc) & ~15;
                c0 = min(max(min(max(cc_q, 0), S1.w - 1) * bpc - ccol0, 0), pitch1 - bpc);
                c1 = min(max(min(max(cc_q + 1, 0), S1.w - 1) * bpc - ccol0, 0), pitch1 - bpc);
                cvC[l] = ccol0 | (min(((cc_l * bpc + bpc - 1 - ccol0) >> 4) + 1, pv1) << 24);
                rC[l].last = min(max(__builtin_amdgcn_readfirstlane(rc), -1), S1.h - 1) + 1;
                hit |= 1 << (8 + l);
                if (validq == ~0ull && __ballot(c1 != c0 + bpc || cca_q != 0.5f) == 0) hit |= 1 << (24 + l);
                if constexpr (KEEP_SRC) sC[l] = ring_src(S1, Ly.src.pl[planar ? 2 : 1], cvC[l]);
            }
        }
        col[l].off = (uint32_t)o0 | ((uint32_t)o1 << 9) | ((uint32_t)c0 << 18) | ((uint32_t)c1 << 25);
    };
    ys_seq<NL>(setup_layer);
    // (only layer 0 can touch this strip and chunk: an opaque YUV picture whose columns take the short form of the row loops, luma and chroma)
    const bool fast0 = CHV_YS_CARRY && HAS_YUV && (hit & 0xFF) == 1 && (lf[0] & 17) == 16 && (hit & (1 << 16)) != 0 && (hit & (1 << 24)) != 0;
    // (the same for the encoder side's frame: only layer 0, an integer-matrix RGB picture drawn at its own size over the whole strip)
    const bool fastR0 = CHV_YS_CARRY && CHV_YS_RGBFAST && KINDS == YK_RGBINT && (hit & 0xFF) == 1 && (lf[0] & 5) == 5 && (hit & (1 << 16)) != 0;
    // (the rings of a tick's layers, then one row table per layer)
    uint32_t *rowtab = (uint32_t *)(lds + (wave_bytes - NL * YS_TAB_BYTES));

    // chroma codes of the 16-row group being assembled: byte m = trip m of the group (even lanes: chroma row 2 m, odd lanes: 2 m + 1)
    uint32_t nu = 0x80808080u, nv = 0x80808080u;
    uint32_t pend_lw = 0;           // the previous trip's luma dword (transposed), stored after this trip's first wait
    int pend_row = -1;
    int issued = 0;
    const uint32_t loff = (CHV_YS_STORE & 1) ? (uint32_t)(lane >> 4) * (uint32_t)PY.pitch + (uint32_t)(x0 + 4 * (lane & 15))
                                             : (uint32_t)(lane & 3) * (uint32_t)PY.pitch + (uint32_t)(x0 + (lane & ~3));
    const bool lcol_ok = (CHV_YS_STORE & 1) ? x0 + 4 * (lane & 15) < TW : x0 + (lane & ~3) < TW;      // (W % 8 == 0: a lane's four columns are inside together)
    auto flush = [&]() {
        if (pend_row >= 0) {
            if (CHV_YS_ABL & 6) asm volatile("" :: "v"(pend_lw));
            else if (lcol_ok) {
                if (CHV_YS_STORE & 2) gst_stream(PY.ptr + (size_t)pend_row * PY.pitch + loff, pend_lw);
                else gst_at<uint32_t>(PY.ptr + (size_t)pend_row * PY.pitch, loff, pend_lw);
            }
            pend_row = -1;
        }
    };

    // the end of a trip: its luma codes transposed inside every quad of lanes and left pending (stored after the next trip's waits), its chroma codes
    // into the group's registers, the group stored every fourth trip
    auto finish_trip = [&](int j0, uint32_t lw, uint32_t cu, uint32_t cv) {
        // ---- the trip's luma: a 4 x 4 byte transpose inside every quad of lanes turns "4 rows of one column" into "4 columns of one row" ----
        {
            const uint32_t sel1 = (lane & 1) ? 0x03070105u : 0x06020400u, sel2 = (lane & 2) ? 0x03020706u : 0x05040100u;
            const uint32_t p1 = (uint32_t)__builtin_amdgcn_update_dpp(dpp_old(), (int)lw, 0xB1 /* quad_perm [1,0,3,2] */, 0xf, 0xf, false);
            const uint32_t aa = __builtin_amdgcn_perm(p1, lw, sel1);
            const uint32_t p2 = (uint32_t)__builtin_amdgcn_update_dpp(dpp_old(), (int)aa, 0x4E /* quad_perm [2,3,0,1] */, 0xf, 0xf, false);
            pend_lw = __builtin_amdgcn_perm(p2, aa, sel2);
            // (lane 4 c + r holds row r, columns 4 c .. 4 c + 3; lane 16 r + c takes it: a quarter wave then holds 64 contiguous bytes of one row)
            if (CHV_YS_STORE & 1) pend_lw = (uint32_t)__builtin_amdgcn_ds_bpermute((4 * (lane & 15) + (lane >> 4)) * 4, (int)pend_lw);
            pend_row = y0 + j0;
        }
        // ---- chroma: byte m of the group's registers; stored every fourth trip ----------------------------------------------------
        {
            const int m = (j0 >> 2) & 3;
            const uint32_t sel = 0x03020100u ^ ((uint32_t)(m ^ 4) << (8 * m));       // byte m <- byte 0 of the trip's code, the others stay
            nu = __builtin_amdgcn_perm(cu, nu, sel);
            nv = __builtin_amdgcn_perm(cv, nv, sel);
            if (m == 3 || j0 + 4 >= nrows) {
                // lane 2k: rows 0, 2, 4, 6 of chroma column k; lane 2k + 1: rows 1, 3, 5, 7  ->  lane 8c + i: row i (+ 4 for lanes 8c + 4 ..) of
                // columns 4c .. 4c + 3 (kernels_wave_yuv.hip.cpp)
                const uint32_t selp = (lane & 1) ? 0x03070206u : 0x05010400u;
                const int srcl = ((lane & ~7) + 2 * (lane & 3) + ((lane >> 2) & 1)) * 4;
                const uint32_t sel1 = (lane & 1) ? 0x03070105u : 0x06020400u, sel2 = (lane & 2) ? 0x03020706u : 0x05040100u;
                auto regroup = [&](uint32_t v) {
                    const uint32_t p = (uint32_t)__builtin_amdgcn_update_dpp(dpp_old(), (int)v, 0xB1, 0xf, 0xf, false);
                    uint32_t q = __builtin_amdgcn_perm(p, v, selp);
                    q = (uint32_t)__builtin_amdgcn_ds_bpermute(srcl, (int)q);
                    const uint32_t p1 = (uint32_t)__builtin_amdgcn_update_dpp(dpp_old(), (int)q, 0xB1, 0xf, 0xf, false);
                    const uint32_t bq = __builtin_amdgcn_perm(p1, q, sel1);
                    const uint32_t p2 = (uint32_t)__builtin_amdgcn_update_dpp(dpp_old(), (int)bq, 0x4E, 0xf, 0xf, false);
                    return __builtin_amdgcn_perm(p2, bq, sel2);
                };
                const uint32_t tu = regroup(nu), tv = regroup(nv);
                const int g0 = j0 & ~15;                                             // first row of the group inside the chunk
                const int crow = (lane & 3) + 4 * ((lane >> 2) & 1);
                const uint32_t ccol = (uint32_t)((x0 >> 1) + 4 * (lane >> 3));
                const int qy = ((y0 + g0) >> 1) + crow;
                const bool ok = crow < 2 * (m + 1) && x0 + 8 * (lane >> 3) < TW;
                if (CHV_YS_ABL & 10) asm volatile("" :: "v"(tu), "v"(tv));
                else if (ok) {
                    if (TF == TF_NV12) {
                        const uint2 w = make_uint2(__builtin_amdgcn_perm(tv, tu, 0x05010400u), __builtin_amdgcn_perm(tv, tu, 0x07030602u));     // u0 v0 u1 v1 | u2 v2 u3 v3
                        gst_at<chv_u32x2>(PC.ptr + (size_t)qy * PC.pitch, ccol * 2u, chv_u32x2{ w.x, w.y });
                    } else {
                        gst_at<uint32_t>(PC.ptr + (size_t)qy * PC.pitch, ccol, tu);
                        gst_at<uint32_t>(PV.ptr + (size_t)qy * PV.pitch, ccol, tv);
                    }
                }
                nu = 0x80808080u; nv = 0x80808080u;
            }
        }
    };

    // One trip: four canvas rows through every layer that touches them
    auto trip = [&](int j0) {
        constexpr bool FAST = false;
        const int jt = j0 & (YS_TAB - 1);
        uint32_t lw = 0;                                 // img_clear_*: Y = 0.0
        uint32_t cu = 128u, cv = 128u;                   // chroma = 0.5 -> 128 (RTE)

        auto layer = [&](auto lc) {
            constexpr int l = decltype(lc)::value;
            if (!FAST && !(hit & (1 << l))) return;
            const uint32_t act4 = FAST ? 15u : (rowm[l] >> jt) & 15u;
            if (!act4) return;
            const uint32_t *tab = rowtab + l * YS_TAB_DW + jt;       // [0 ..]: packed, [YS_TAB ..]: luma / RGB row weights, [2 YS_TAB ..]: chroma
            const bool lane_pic = (picmask >> l) & 1u;
            const bool rgb = !FAST && HAS_RGB && (!HAS_YUV || (lf[l] & 1) != 0);
            const uint8_t *ldsY = lds + lbase[l];
            const int o0 = ys_o0(col[l].off), o1 = ys_o1(col[l].off);

            if (!rgb) {
                if constexpr (HAS_YUV) {
                    // ---- YUV picture (kernels.cl.swift:78-94): cur * (1 - opacity) + sample * opacity, luma at every pixel, chroma at
                    //      the quad's even/even pixel ----
                    const bool planar = ((KINDS & YK_PLANAR) != 0) && (((KINDS & YK_NV12) == 0) || (lf[l] & 2) != 0);
                    const float alpha = opac[l], ialpha = 1.f - alpha;
                    const bool opaque = (lf[l] & 16) != 0;
                    const float a = col[l].a, ia = 1.0f - a;
                    const bool act[4] = { (act4 & 1) != 0, (act4 & 2) != 0, (act4 & 4) != 0, (act4 & 8) != 0 };
                    const uint4 w4 = *(const uint4 *)(tab + YS_TAB);
                    const float rya[4] = { __uint_as_float(w4.x), __uint_as_float(w4.y), __uint_as_float(w4.z), __uint_as_float(w4.w) };
                    // rows of the trip that are consecutive source rows (native resolution): the lower tap row of a pixel is the upper one
                    // of the pixel below — its two UNORM8 conversions (three instructions each) are carried down the lane.
                    // `full`: such a trip over a strip in the short form — the full-canvas video under a mixer's overlays: every pixel takes
                    // its row's result (no selects), tap column 1 is an immediate offset, and with both column weights one half the four weight
                    // products are two
                    const bool full = FAST || ((fullm[l] >> jt) & 1u) != 0;
                    const bool unit = full || (CHV_YS_CARRY && opaque && act4 == 15u && ((unitm[l] >> jt) & 7u) == 7u);
                    const uint32_t pk0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)tab[0]);
                    const int ry0 = (int)(pk0 & 8191u) - 1;
                    auto luma_unit = [&](auto full_c) {
                        constexpr bool FULL = decltype(full_c)::value;
                        int q = ring_row<RingY>(rY[l], ry0);
                        const uint8_t *p = ldsY + (q + o0);
                        float t0 = ys_t8(p[0]), t1 = FULL ? ys_t8(p[1]) : ys_t8(ldsY[q + o1]);
                        auto row = [&](auto kc) {
                            constexpr int k = decltype(kc)::value;
                            const float bw = rya[k], ib = 1.0f - bw;
                            q = ring_next<RingY>(q);
                            const uint8_t *p1 = ldsY + (q + o0);
                            const float b0 = ys_t8(p1[0]), b1 = FULL ? ys_t8(p1[1]) : ys_t8(ldsY[q + o1]);
                            float v;
                            if constexpr (FULL) { const float wt = 0.5f * ib, wb = 0.5f * bw; v = ys_mix4(wt, wt, wb, wb, t0, t1, b0, b1); }
                            else v = ys_mix4(ia * ib, a * ib, ia * bw, a * bw, t0, t1, b0, b1);
                            t0 = b0; t1 = b1;
                            const uint32_t nlw = ys_put<k>(lw, v);          // opacity == 1: cur * 0 + luma * 1 = luma exactly
                            if constexpr (FULL) lw = nlw; else lw = lane_pic ? nlw : lw;
                        };
                        ys_seq<4>(row);
                    };
                    auto luma_any = [&](auto opaque_c) {
                        constexpr bool OP = decltype(opaque_c)::value;
                        const uint4 p4 = *(const uint4 *)tab;
                        const int ry[4] = { ry0, (int)((uint32_t)__builtin_amdgcn_readfirstlane((int)p4.y) & 8191u) - 1,
                                            (int)((uint32_t)__builtin_amdgcn_readfirstlane((int)p4.z) & 8191u) - 1,
                                            (int)((uint32_t)__builtin_amdgcn_readfirstlane((int)p4.w) & 8191u) - 1 };
                        // (rows outside the picture are computed and not taken: any resident row will do for them)
                        const int rin = act[0] ? ry[0] : act[1] ? ry[1] : act[2] ? ry[2] : ry[3];
                        auto row = [&](auto kc) {
                            constexpr int k = decltype(kc)::value;
                            const int r = act[k] ? ry[k] : rin;
                            const float bw = rya[k], ib = 1.0f - bw;
                            const int q0 = ring_row<RingY>(rY[l], r), q1 = ring_next<RingY>(q0);
                            const uint8_t *p0 = ldsY + q0, *p1 = ldsY + q1;
                            const float v = ys_mix4(ia * ib, a * ib, ia * bw, a * bw, ys_t8(p0[o0]), ys_t8(p0[o1]), ys_t8(p1[o0]), ys_t8(p1[o1]));
                            const float o = OP ? v : ys_t8k<k>(lw) * ialpha + v * alpha;
                            const uint32_t nlw = ys_put<k>(lw, o);
                            lw = (lane_pic && act[k]) ? nlw : lw;
                        };
                        ys_seq<4>(row);
                    };
                    if (full) luma_unit(std::true_type{});
                    else if (unit) luma_unit(std::false_type{});
                    else if (opaque) luma_any(std::true_type{});
                    else luma_any(std::false_type{});

                    if (FAST || ((hit & (1 << (8 + l))) != 0 && (act4 & 5) != 0)) {
                        // chroma rows 2 m (even lanes) and 2 m + 1 (odd lanes): sampled at the uv of the quad's even/even pixel — the even lane's
                        // column entry, the row entry of luma row 4 m + 2 par — on the half-size plane(s)
                        const bool pic_q = (picmask >> (8 + l)) & 1u;
                        const bool tkc = pic_q && (par ? act[2] : act[0]);
                        const uint4 cw4 = *(const uint4 *)(tab + 2 * YS_TAB);
                        const float cbw = __uint_as_float(par ? cw4.z : cw4.x), icb = 1.0f - cbw;
                        const uint8_t *ldsC = ldsY + RingY::BYTES;
                        const int c0 = ys_c0(col[l].off);
                        const bool fullc = FAST || ((fullm[l] >> jt) & 2u) != 0;
                        const bool cunit = fullc || ((act4 & 5u) == 5u && ((crowm[l] >> jt) & 1u) != 0);
                        // (the short form, as for luma: opaque, every pair of the strip inside the picture, tap column 1 right behind tap column 0,
                        // column weights one half, the odd lanes' chroma row the one behind the even lanes')
                        int rc0 = (int)((pk0 >> 13) & 8191u) - 1, rc2 = rc0 + 1;
                        if (!cunit) {
                            rc2 = (int)(((uint32_t)__builtin_amdgcn_readfirstlane((int)tab[2]) >> 13) & 8191u) - 1;
                            if (!act[0]) rc0 = rc2;
                            if (!act[2]) rc2 = rc0;
                        }
                        auto chroma_rows = [&](auto planar_c, auto full_c) {
                            constexpr bool PL = decltype(planar_c)::value, FULL = decltype(full_c)::value;
                            using RC = std::conditional_t<PL, RingCP, RingC2>;
                            constexpr int BPC = PL ? 1 : 2, VO = PL ? RingCP::SPLIT * 16 : 1;       // tap column 1, the V sample
                            const int q00 = ring_row<RC>(rC[l], rc0), q01 = ring_next<RC>(q00);
                            const int q20 = FULL ? q01 : ring_row<RC>(rC[l], rc2), q21 = ring_next<RC>(q20);
                            const uint8_t *p0 = ldsC + ((par ? q20 : q00) + c0), *p1 = ldsC + ((par ? q21 : q01) + c0);
                            float fu, fv;
                            if constexpr (FULL) {
                                const float wt = 0.5f * icb, wb = 0.5f * cbw;
                                fu = ys_mix4(wt, wt, wb, wb, ys_t8(p0[0]), ys_t8(p0[BPC]), ys_t8(p1[0]), ys_t8(p1[BPC]));
                                fv = ys_mix4(wt, wt, wb, wb, ys_t8(p0[VO]), ys_t8(p0[VO + BPC]), ys_t8(p1[VO]), ys_t8(p1[VO + BPC]));
                                cu = ys_put<0>(cu, fu); cv = ys_put<0>(cv, fv);
                            } else {
                                const int dc = ys_c1(col[l].off) - c0;
                                const float ca = col[l].ca, ica = 1.0f - ca;
                                const float c00 = ica * icb, c10 = ca * icb, c01 = ica * cbw, c11 = ca * cbw;
                                fu = ys_mix4(c00, c10, c01, c11, ys_t8(p0[0]), ys_t8(p0[dc]), ys_t8(p1[0]), ys_t8(p1[dc]));
                                fv = ys_mix4(c00, c10, c01, c11, ys_t8(p0[VO]), ys_t8(p0[VO + dc]), ys_t8(p1[VO]), ys_t8(p1[VO + dc]));
                                const uint32_t nnu = ys_put<0>(cu, opaque ? fu : ys_t8k<0>(cu) * ialpha + fu * alpha);
                                const uint32_t nnv = ys_put<0>(cv, opaque ? fv : ys_t8k<0>(cv) * ialpha + fv * alpha);
                                cu = tkc ? nnu : cu; cv = tkc ? nnv : cv;
                            }
                        };
                        if (planar) { if (fullc) chroma_rows(std::true_type{}, std::true_type{}); else chroma_rows(std::true_type{}, std::false_type{}); }
                        else        { if (fullc) chroma_rows(std::false_type{}, std::true_type{}); else chroma_rows(std::false_type{}, std::false_type{}); }
                    }
                }
            } else {
                if constexpr (HAS_RGB) {
                    // ---- RGB picture: the reference's float rows (kernels.cl.swift:509-529; no fill paint on this path: the fill pre-blend is
                    //      the identity) or the integer BT.601 / 709 matrix (img_*_int).  Row by row (a runtime loop: a row of
                    //      an RGB layer is 100-130 instructions, and four unrolled copies per layer slot are 100 KB of code): a row of four-byte
                    //      texels is a third of a batch, so the ring follows every row. ----
                    const DLayer &Ly = L[l];
                    const uint32_t ringY = lds0 + (uint32_t)lbase[l];
                    auto srcY = [&]() { if constexpr (KEEP_SRC) return sY[l]; else return ring_src(Ly.src.pl[0], Ly.src.pl[0], cvY[l]); };
                    const bool is_int = ((KINDS & YK_RGBINT) != 0) && (((KINDS & YK_RGB) == 0) || (lf[l] & 4) != 0);
                    const bool swz = (lf[l] & 8) != 0;          // texels are taken as R, G, B, A whatever the source order
                    const float opacity = opac[l];
                    const float a = col[l].a, ia = 1.0f - a;
                    const bool owner_lane = par == 0 && col_in;
                    auto fix = [&](uint32_t w) { return swz ? __builtin_amdgcn_perm(w, w, 0x03000102u) : w; };
                    auto rows = [&](auto int_c) {
                        constexpr bool INT = decltype(int_c)::value;
                        const R2Y &kk = kR2Y[(lf[l] >> 8) & 3];
                        const float ka = opacity * kInv255;
                        int have_row = -0x40000000;                  // source row whose texels t.. hold (carried down the lane)
                        float t00 = 0.f, t01 = 0.f, t02 = 0.f, t03 = 0.f, t10 = 0.f, t11 = 0.f, t12 = 0.f, t13 = 0.f;
#pragma unroll 1
                        for (int k = 0; k < 4; k++) {
                            if (!((act4 >> k) & 1)) continue;
                            const int ryk = (int)((uint32_t)__builtin_amdgcn_readfirstlane((int)tab[k]) & 8191u) - 1;
                            ring_ensure<RingR>(rY[l], ringY, srcY, ryk, ryk + 1, issued, lane);
                            const bool tk = lane_pic;
                            const float bw = __uint_as_float(tab[YS_TAB + k]), ib = 1.0f - bw;
                            const int q0 = ring_row<RingR>(rY[l], ryk), q1 = ring_next<RingR>(q0);
                            const uint8_t *p1 = ldsY + q1;
                            const uint32_t u01 = fix(*(const uint32_t *)(p1 + o0)), u11 = fix(*(const uint32_t *)(p1 + o1));
                            if (!(CHV_YS_CARRY && have_row == ryk)) {
                                const uint8_t *p0 = ldsY + q0;
                                const uint32_t u00 = fix(*(const uint32_t *)(p0 + o0)), u10 = fix(*(const uint32_t *)(p0 + o1));
                                if constexpr (INT) {
                                    t00 = ub0(u00); t01 = ub1(u00); t02 = ub2(u00); t03 = ub3(u00);
                                    t10 = ub0(u10); t11 = ub1(u10); t12 = ub2(u10); t13 = ub3(u10);
                                } else {
                                    t00 = ys_t8k<0>(u00); t01 = ys_t8k<1>(u00); t02 = ys_t8k<2>(u00); t03 = ys_t8k<3>(u00);
                                    t10 = ys_t8k<0>(u10); t11 = ys_t8k<1>(u10); t12 = ys_t8k<2>(u10); t13 = ys_t8k<3>(u10);
                                }
                            }
                            const float w00 = ia * ib, w10 = a * ib, w01 = ia * bw, w11 = a * bw;
                            float b00, b01, b02, b03, b10, b11, b12, b13;
                            if constexpr (INT) {
                                b00 = ub0(u01); b01 = ub1(u01); b02 = ub2(u01); b03 = ub3(u01);
                                b10 = ub0(u11); b11 = ub1(u11); b12 = ub2(u11); b13 = ub3(u11);
                            } else {
                                b00 = ys_t8k<0>(u01); b01 = ys_t8k<1>(u01); b02 = ys_t8k<2>(u01); b03 = ys_t8k<3>(u01);
                                b10 = ys_t8k<0>(u11); b11 = ys_t8k<1>(u11); b12 = ys_t8k<2>(u11); b13 = ys_t8k<3>(u11);
                            }
                            const float curf = (float)__builtin_amdgcn_ubfe(lw, (uint32_t)(8 * k), 8u);      // the pixel's luma code
                            const bool even_row = (k & 1) == 0;
                            if constexpr (INT) {
                                const float q0f = cs_mix(w00, w10, w01, w11, t00, t10, b00, b10);
                                const float q1f = cs_mix(w00, w10, w01, w11, t01, t11, b01, b11);
                                const float q2f = cs_mix(w00, w10, w01, w11, t02, t12, b02, b12);
                                const float q3f = cs_mix(w00, w10, w01, w11, t03, t13, b03, b13);
                                // to_code_raw of a convex combination of codes: no clamp can trigger; rint through the float adder
                                // (the multiplier's operands keep the adder's bias: r2y_base_biased, pixel_math.hip.h)
                                const int cr = (int)code_biased(q0f), cg = (int)code_biased(q1f), cb = (int)code_biased(q2f);
                                const float a2 = q3f * ka, ia2 = 1.f - a2;
                                const float py = fixed_to_codef(r2y_row(kk.y[0], kk.y[1], kk.y[2], r2y_base_biased(kk.y[0], kk.y[1], kk.y[2], (kk.yoff << 16) + 32768), cr, cg, cb));
                                const uint32_t nlw = ys_put_raw_k(lw, __builtin_fmaf(py, a2, curf * ia2), k);
                                lw = tk ? nlw : lw;
                                if (even_row) {
                                    // chroma of the quad: the even lane's pixel of this (even) row; the trip's second chroma row lives in the odd
                                    // lane (the values travel one lane up, quad_perm [0, 0, 2, 2])
                                    float pu = fixed_to_codef(r2y_row(kk.u[0], kk.u[1], kk.u[2], r2y_base_biased(kk.u[0], kk.u[1], kk.u[2], (128 << 16) + 32768), cr, cg, cb));
                                    float pv = fixed_to_codef(r2y_row(kk.v[0], kk.v[1], kk.v[2], r2y_base_biased(kk.v[0], kk.v[1], kk.v[2], (128 << 16) + 32768), cr, cg, cb));
                                    float sa = a2, sia = ia2;
                                    int stk = (tk && owner_lane) ? 1 : 0;
                                    if (k == 2) { pu = ys_dpp_even(pu); pv = ys_dpp_even(pv); sa = ys_dpp_even(a2); sia = ys_dpp_even(ia2); stk = ys_dpp_even(stk); }
                                    const bool mine = stk != 0 && par == (k >> 1);
                                    const uint32_t nnu = ys_put_raw<0>(cu, __builtin_fmaf(pu, sa, ub0(cu) * sia));
                                    const uint32_t nnv = ys_put_raw<0>(cv, __builtin_fmaf(pv, sa, ub0(cv) * sia));
                                    cu = mine ? nnu : cu; cv = mine ? nnv : cv;
                                }
                            } else {
                                const float r = ys_mix4(w00, w10, w01, w11, t00, t10, b00, b10);
                                const float g = ys_mix4(w00, w10, w01, w11, t01, t11, b01, b11);
                                const float bl = ys_mix4(w00, w10, w01, w11, t02, t12, b02, b12);
                                const float q3f = ys_mix4(w00, w10, w01, w11, t03, t13, b03, b13);
                                const float a2 = q3f * opacity, ia2 = 1.f - a2;
                                float yy, uu, vv;
                                rgb2yuv(r * a2, g * a2, bl * a2, yy, uu, vv);
                                // (no fill: `cur * (1 - 0) + f * 0` is cur, and its clamp to [-1, 1] is the identity on a code / 255)
                                const uint32_t nlw = ys_put_raw_k(lw, (unorm8f(curf) * ia2 + yy * a2) * 255.0f, k);
                                lw = tk ? nlw : lw;
                                if (even_row) {
                                    float su = uu, sv = vv, sa = a2, sia = ia2;
                                    int stk = (tk && owner_lane) ? 1 : 0;
                                    if (k == 2) { su = ys_dpp_even(uu); sv = ys_dpp_even(vv); sa = ys_dpp_even(a2); sia = ys_dpp_even(ia2); stk = ys_dpp_even(stk); }
                                    const bool mine = stk != 0 && par == (k >> 1);
                                    const uint32_t nnu = ys_put<0>(cu, ys_t8k<0>(cu) * sia + su * sa);
                                    const uint32_t nnv = ys_put<0>(cv, ys_t8k<0>(cv) * sia + sv * sa);
                                    cu = mine ? nnu : cu; cv = mine ? nnv : cv;
                                }
                            }
                            t00 = b00; t01 = b01; t02 = b02; t03 = b03; t10 = b10; t11 = b11; t12 = b12; t13 = b13;
                            have_row = ryk + 1;
                        }
                    };
                    if (is_int) { if constexpr ((KINDS & YK_RGBINT) != 0) rows(std::true_type{}); }
                    else { if constexpr ((KINDS & YK_RGB) != 0) rows(std::false_type{}); }
                }
            }
        };
        ys_seq<NL>(layer);
        finish_trip(j0, lw, cu, cv);
    };

    // A trip of the encoder side's frame — only layer 0, an RGB picture through the integer matrix (img_*_int), drawn at its own size
    // over the whole strip onto the cleared canvas, its four rows inside the picture and on consecutive source rows: one ring request and one
    // wait for the trip instead of one per row, the four rows unrolled with their byte positions as immediates, no per-pixel selects on luma,
    // both column weights one half (two weight products instead of four), the lower texel row of a pixel carried down the lane as the upper one
    // of the pixel below — across trips too —, the matrix as 24-bit multiply-adds (codes < 2^8, coefficients < 2^16: exact).  Cleared canvas and
    // first layer: the luma blend fma(p, a, 0 * (1 - a)) is RN(p * a).  The same operations per pixel as the general row loop otherwise.
    float rt00 = 0.f, rt01 = 0.f, rt02 = 0.f, rt03 = 0.f, rt10 = 0.f, rt11 = 0.f, rt12 = 0.f, rt13 = 0.f;
    int rt_row = -0x40000000;                        // the source row whose texels rt.. hold
    auto fast_rgb_trip = [&](int j0, auto swz_c) {
        constexpr bool SWZ = decltype(swz_c)::value;
        const int jt = j0 & (YS_TAB - 1);
        const uint32_t *tab = rowtab + jt;
        const DLayer &Ly = L[0];
        const uint32_t ringY = lds0 + (uint32_t)lbase[0];
        const uint8_t *ldsY = lds + lbase[0];
        const int ry0 = (int)((uint32_t)__builtin_amdgcn_readfirstlane((int)tab[0]) & 8191u) - 1;
        auto srcY = [&]() { if constexpr (KEEP_SRC) return sY[0]; else return ring_src(Ly.src.pl[0], Ly.src.pl[0], cvY[0]); };
        const int hi = ry0 + 4 < rY[0].last ? ry0 + 4 : rY[0].last;
        ring_ensure<RingR>(rY[0], ringY, srcY, ry0, hi, issued, lane);
        flush();
        const uint4 w4 = *(const uint4 *)(tab + YS_TAB);
        const float rya[4] = { __uint_as_float(w4.x), __uint_as_float(w4.y), __uint_as_float(w4.z), __uint_as_float(w4.w) };
        const int o0 = ys_o0(col[0].off);
        const R2Y &kk = kR2Y[(lf[0] >> 8) & 3];
        const int ky0 = kk.y[0], ky1 = kk.y[1], ky2 = kk.y[2], ku0 = kk.u[0], ku1 = kk.u[1], ku2 = kk.u[2], kv0 = kk.v[0], kv1 = kk.v[1], kv2 = kk.v[2];
        // (row offsets for operands that keep the float adder's bias: r2y_base_biased, pixel_math.hip.h)
        const int cy_ = r2y_base_biased(ky0, ky1, ky2, (kk.yoff << 16) + 32768), ccu_ = r2y_base_biased(ku0, ku1, ku2, (128 << 16) + 32768),
                  ccv_ = r2y_base_biased(kv0, kv1, kv2, (128 << 16) + 32768);
        const float ka = opac[0] * kInv255;
        // (texels are taken as R, G, B, A whatever the source order: a swizzled source has R in byte 2 and B in byte 0 — which byte a
        // conversion reads is free, a v_perm_b32 per texel in front of the conversions was not)
        auto red = [](uint32_t w) { return SWZ ? ub2(w) : ub0(w); };
        auto blue = [](uint32_t w) { return SWZ ? ub0(w) : ub2(w); };
        int q = ring_row<RingR>(rY[0], ry0);
        if (rt_row != ry0) {
            const uint8_t *p0 = ldsY + (q + o0);
            const uint32_t u00 = *(const uint32_t *)p0, u10 = *(const uint32_t *)(p0 + 4);
            rt00 = red(u00); rt01 = ub1(u00); rt02 = blue(u00); rt03 = ub3(u00);
            rt10 = red(u10); rt11 = ub1(u10); rt12 = blue(u10); rt13 = ub3(u10);
        }
        uint32_t lw = 0, cu = 128u, cv = 128u;
        auto row = [&](auto kc) {
            constexpr int k = decltype(kc)::value;
            const float bw = rya[k], ib = 1.0f - bw;
            q = ring_next<RingR>(q);
            const uint8_t *p1 = ldsY + (q + o0);
            const uint32_t u01 = *(const uint32_t *)p1, u11 = *(const uint32_t *)(p1 + 4);
            const float b00 = red(u01), b01 = ub1(u01), b02 = blue(u01), b03 = ub3(u01);
            const float b10 = red(u11), b11 = ub1(u11), b12 = blue(u11), b13 = ub3(u11);
            const float wt = 0.5f * ib, wb = 0.5f * bw;
            const float q0f = cs_mix(wt, wt, wb, wb, rt00, rt10, b00, b10);
            const float q1f = cs_mix(wt, wt, wb, wb, rt01, rt11, b01, b11);
            const float q2f = cs_mix(wt, wt, wb, wb, rt02, rt12, b02, b12);
            const float q3f = cs_mix(wt, wt, wb, wb, rt03, rt13, b03, b13);
            rt00 = b00; rt01 = b01; rt02 = b02; rt03 = b03; rt10 = b10; rt11 = b11; rt12 = b12; rt13 = b13;
            // to_code_raw of a convex combination of codes: no clamp can trigger; rint through the float adder
            const int cr = (int)code_biased(q0f), cg = (int)code_biased(q1f), cb = (int)code_biased(q2f);
            const float a2 = q3f * ka, ia2 = 1.f - a2;
            const float py = fixed_to_codef(mad24_uniform(cr, ky0, mad24_uniform(cg, ky1, mad24_uniform(cb, ky2, cy_))));
            lw = ys_put_raw<k>(lw, py * a2);
            if constexpr ((k & 1) == 0) {
                // chroma of the quad: the even lane's pixel of this (even) row; the trip's second chroma row lives in the odd lane (the values
                // travel one lane up, quad_perm [0, 0, 2, 2])
                float pu = fixed_to_codef(mad24_uniform(cr, ku0, mad24_uniform(cg, ku1, mad24_uniform(cb, ku2, ccu_))));
                float pv = fixed_to_codef(mad24_uniform(cr, kv0, mad24_uniform(cg, kv1, mad24_uniform(cb, kv2, ccv_))));
                float sa = a2, sia = ia2;
                if constexpr (k == 2) { pu = ys_dpp_even(pu); pv = ys_dpp_even(pv); sa = ys_dpp_even(a2); sia = ys_dpp_even(ia2); }
                const bool mine = par == (k >> 1);            // (every column of the strip is inside the picture and the canvas)
                const uint32_t nnu = ys_put_raw<0>(cu, __builtin_fmaf(pu, sa, ub0(cu) * sia));
                const uint32_t nnv = ys_put_raw<0>(cv, __builtin_fmaf(pv, sa, ub0(cv) * sia));
                cu = mine ? nnu : cu; cv = mine ? nnv : cv;
            }
        };
        ys_seq<4>(row);
        rt_row = ry0 + 4;
        finish_trip(j0, lw, cu, cv);
    };

    // A step of eight canvas rows in which only layer 0 — an opaque YUV picture drawn at its own size over the whole strip: the full-canvas video
    // of a mixer tick, the bulk of its pixels — touches the strip, every row taps the source row behind its predecessor's, and the columns take the
    // short form (tap column 1 right behind tap column 0, both weights one half): no per-layer tests, one ring position per plane walked row by
    // row, the lower tap row of a pixel carried down the lane as the upper one of the pixel below across both trips.  The same operations per
    // pixel as the general row loops.  (~100 instructions per canvas row went through the general path, 45 of them scalar; the scalar unit is
    // shared by the four SIMDs of a CU and the kernel is bound by what it issues.)
    auto fast_step = [&](int j0, auto planar_c) {
        constexpr bool PL = decltype(planar_c)::value;
        using RCc = std::conditional_t<PL, RingCP, RingC2>;
        constexpr int l = 0;
        constexpr int BPC = PL ? 1 : 2, VO = PL ? RingCP::SPLIT * 16 : 1;
        const int jt = j0 & (YS_TAB - 1);
        const uint32_t *tab = rowtab + jt;
        const DLayer &Ly = L[0];
        const uint2 sm = *(const uint2 *)(rowtab + 3 * YS_TAB + (jt >> 2));
        const uint32_t sy_ = (uint32_t)__builtin_amdgcn_readfirstlane((int)sm.x), sc_ = (uint32_t)__builtin_amdgcn_readfirstlane((int)sm.y);
        const uint32_t ringY = lds0 + (uint32_t)lbase[0];
        const int lo = (int)(sy_ & 8191u) - 1, clo = (int)(sc_ & 8191u) - 1;
        {
            const int hi0 = (int)((sy_ >> 13) & 8191u), hi = hi0 < rY[0].last ? hi0 : rY[0].last;
            const int chi0 = (int)((sc_ >> 13) & 8191u), chi = chi0 < rC[0].last ? chi0 : rC[0].last;
            auto srcY = [&]() { if constexpr (KEEP_SRC) return sY[l]; else return ring_src(Ly.src.pl[0], Ly.src.pl[0], cvY[l]); };
            auto srcC = [&]() { if constexpr (KEEP_SRC) return sC[l]; else return ring_src(Ly.src.pl[1], Ly.src.pl[PL ? 2 : 1], cvC[l]); };
            ring_ensure_step<RingY>(rY[0], ringY, srcY, lo, hi, issued, lane);
            ring_ensure_step<RCc>(rC[0], ringY + (uint32_t)RingY::BYTES, srcC, clo, chi, issued, lane);
        }
        flush();                                         // (the previous trip's luma, after this step's waits)
        const uint8_t *ldsY = lds + lbase[0], *ldsC = ldsY + RingY::BYTES;
        const int o0 = ys_o0(col[0].off), c0 = ys_c0(col[0].off);
        int q = ring_row<RingY>(rY[0], lo), qc = ring_row<RCc>(rC[0], clo);
        float t0, t1;
        { const uint8_t *p = ldsY + (q + o0); t0 = ys_t8(p[0]); t1 = ys_t8(p[1]); }
#pragma unroll
        for (int h = 0; h < 2; h++) {
            const uint4 w4 = *(const uint4 *)(tab + YS_TAB + 4 * h);
            const float rya[4] = { __uint_as_float(w4.x), __uint_as_float(w4.y), __uint_as_float(w4.z), __uint_as_float(w4.w) };
            uint32_t lw = 0, cu = 128u, cv = 128u;
            auto row = [&](auto kc) {
                constexpr int k = decltype(kc)::value;
                const float bw = rya[k], ib = 1.0f - bw;
                q = ring_next<RingY>(q);
                const uint8_t *p1 = ldsY + (q + o0);
                const float b0 = ys_t8(p1[0]), b1 = ys_t8(p1[1]);
                const float wt = 0.5f * ib, wb = 0.5f * bw;
                const float v = ys_mix4(wt, wt, wb, wb, t0, t1, b0, b1);
                t0 = b0; t1 = b1;
                lw = ys_put<k>(lw, v);
            };
            ys_seq<4>(row);
            {
                // chroma rows 2 m (even lanes: taps qc, qc + 1) and 2 m + 1 (odd lanes: qc + 1, qc + 2)
                const uint4 cw4 = *(const uint4 *)(tab + 2 * YS_TAB + 4 * h);
                const float cbw = __uint_as_float(par ? cw4.z : cw4.x), icb = 1.0f - cbw;
                const int q1 = ring_next<RCc>(qc), q2 = ring_next<RCc>(q1);
                const uint8_t *p0 = ldsC + ((par ? q1 : qc) + c0), *p1 = ldsC + ((par ? q2 : q1) + c0);
                const float wt = 0.5f * icb, wb = 0.5f * cbw;
                const float fu = ys_mix4(wt, wt, wb, wb, ys_t8(p0[0]), ys_t8(p0[BPC]), ys_t8(p1[0]), ys_t8(p1[BPC]));
                const float fv = ys_mix4(wt, wt, wb, wb, ys_t8(p0[VO]), ys_t8(p0[VO + BPC]), ys_t8(p1[VO]), ys_t8(p1[VO + BPC]));
                cu = ys_put<0>(cu, fu); cv = ys_put<0>(cv, fv);
                qc = q2;
            }
            if (h == 1) flush();
            finish_trip(j0 + 4 * h, lw, cu, cv);
        }
    };

    for (int jv = 0; jv < nrows; jv += 4) {
        // (the row counter is wave-uniform, and says so: left to itself hipcc kept it — and with it every mask shift, every ring decision of the
        // step and all their branches — on the vector unit, with exec-mask branches around each `if`)
        const int j0 = __builtin_amdgcn_readfirstlane(jv);
        const int jt = j0 & (YS_TAB - 1);
        // ---- row entries of every layer that can touch the strip, YS_TAB rows at a time: lane = row ----------------------------
        if (jt == 0) {
            wave_lds_fence();
            auto fill = [&](auto lc) {
                constexpr int l = decltype(lc)::value;
                if (!(hit & (1 << l))) return;
                rowm[l] = 0; unitm[l] = 0; crowm[l] = 0; fullm[l] = 0;
                if (j0 + YS_TAB <= (trips[l] & 0xFFFF) || j0 >= (trips[l] >> 16)) return;
                const DLayer &Ly = L[l];
                const float *U = Ly.u;
                const bool rgb = HAS_RGB && (!HAS_YUV || (lf[l] & 1) != 0);
                const int h0 = Ly.src.pl[0].h, h1 = Ly.src.pl[rgb ? 0 : 1].h;
                const int ye = min(y0 + j0 + min(lane, YS_TAB - 1), TH - 1);
                const float ny = ((float)ye / sy) * 2.f - 1.f;
                const float t3 = U[U_TRANSFORM + 15];
                const float t1 = ny * U[U_TRANSFORM + 5] + U[U_TRANSFORM + 7];
                const float b1 = ny * U[U_BORDER + 5] + U[U_BORDER + 7];
                const float v = t1 * U[U_TEXTURE + 5] + t3 * U[U_TEXTURE + 7];
                const int rfl = ((b1 >= 0.f && b1 <= 1.f) ? AX_BORDER : 0) | ((t1 >= 0.f && t1 <= 1.f) ? AX_TX : 0) | ((v >= 0.f && v <= 1.f) ? AX_UV : 0);
                int ry, rc;
                float rya, rca;
                lin_axis_raw(v, h0, ry, rya); lin_axis_raw(v, h1, rc, rca);
                // (rows inside the picture have positions -1 .. h - 1; the positions of the others only bound what a step keeps resident: the
                // clamped values rise with the row like the real ones)
                ry = min(max(ry, -1), h0 - 1); rc = min(max(rc, -1), h1 - 1);
                const uint32_t packed = (uint32_t)(ry + 1) | ((uint32_t)(rc + 1) << 13) | ((uint32_t)rfl << 26);
                uint32_t *t = rowtab + l * YS_TAB_DW;
                const bool in_tab = lane < YS_TAB;
                if (in_tab) { t[lane] = packed; t[YS_TAB + lane] = __float_as_uint(rya); t[2 * YS_TAB + lane] = __float_as_uint(rca); }
                // what the rows of this table are, as wave-uniform masks (lane + n through LDS-free lane shuffles: ds_bpermute reads any lane)
                const int ry1 = __builtin_amdgcn_ds_bpermute((lane + 1) * 4, ry), rc2 = __builtin_amdgcn_ds_bpermute((lane + 2) * 4, rc);
                const int ry7 = __builtin_amdgcn_ds_bpermute((lane + 7) * 4, ry), rc6 = __builtin_amdgcn_ds_bpermute((lane + 6) * 4, rc);
                const uint32_t am = (uint32_t)__ballot(in_tab && rfl == AX_ALL), um = (uint32_t)__ballot(lane < YS_TAB - 1 && ry1 == ry + 1);
                const uint32_t cm = (uint32_t)__ballot(lane < YS_TAB - 2 && rc2 == rc + 1);
                rowm[l] = am; unitm[l] = um; crowm[l] = cm;
                {
                    // trips in the short form (see the row loops): opaque layer, the strip's 64 columns in the short form, the trip's four rows
                    // inside the picture and on consecutive source rows; chroma: rows 0 and 2 inside, the second chroma row behind the first
                    const uint32_t a4 = am & (am >> 1) & (am >> 2) & (am >> 3), u3 = um & (um >> 1) & (um >> 2);
                    uint32_t fm = 0;
                    if ((lf[l] & 16) != 0 && CHV_YS_CARRY) {
                        if (hit & (1 << (16 + l))) fm |= a4 & u3 & 0x11111111u;
                        if (hit & (1 << (24 + l))) fm |= (am & (am >> 2) & cm & 0x11111111u) << 1;
                    }
                    fullm[l] = fm;
                }
                // per 8-row step: the first and last tap rows, luma and chroma (rows 0 and 7 / 0 and 6 of the step)
                if (in_tab && (lane & 7) == 0) {
                    t[3 * YS_TAB + (lane >> 2)] = (uint32_t)(ry + 1) | ((uint32_t)(ry7 + 1) << 13);
                    t[3 * YS_TAB + (lane >> 2) + 1] = (uint32_t)(rc + 1) | ((uint32_t)(rc6 + 1) << 13);
                }
            };
            ys_seq<NL>(fill);
            wave_lds_fence();
        }
        // ---- residency, once per 8-row step: the YUV layers' rings take the rows the step's two trips tap -------------------------------
        if constexpr (HAS_YUV) {
            if ((j0 & 7) == 0) {
                // the short way for the bulk of a mixer tick's pixels (fast_step): all eight rows inside the picture, each on the source row behind
                // its predecessor's, chroma rows likewise
                if (fast0 && j0 + 8 <= nrows && ((rowm[0] >> jt) & 0xFFu) == 0xFFu && ((unitm[0] >> jt) & 0x7Fu) == 0x7Fu && ((crowm[0] >> jt) & 0x15u) == 0x15u) {
                    if ((KINDS & YK_PLANAR) != 0 && (((KINDS & YK_NV12) == 0) || (lf[0] & 2) != 0)) fast_step(j0, std::true_type{});
                    else if constexpr ((KINDS & YK_NV12) != 0) fast_step(j0, std::false_type{});
                    jv += 4;
                    continue;
                }
                auto step = [&](auto lc) {
                    constexpr int l = decltype(lc)::value;
                    if (!(hit & (1 << l))) return;
                    const bool rgb = HAS_RGB && (lf[l] & 1) != 0;
                    if (rgb) return;
                    const uint32_t stepact = (rowm[l] >> jt) & 0xFFu;
                    if (!stepact) return;
                    const DLayer &Ly = L[l];
                    const uint2 sm = *(const uint2 *)(rowtab + l * YS_TAB_DW + 3 * YS_TAB + (jt >> 2));
                    const uint32_t sy_ = (uint32_t)__builtin_amdgcn_readfirstlane((int)sm.x), sc_ = (uint32_t)__builtin_amdgcn_readfirstlane((int)sm.y);
                    const uint32_t ringY = lds0 + (uint32_t)lbase[l];
                    {
                        const int lo = (int)(sy_ & 8191u) - 1, hi0 = (int)((sy_ >> 13) & 8191u), hi = hi0 < rY[l].last ? hi0 : rY[l].last;
                        auto srcY = [&]() { if constexpr (KEEP_SRC) return sY[l]; else return ring_src(Ly.src.pl[0], Ly.src.pl[0], cvY[l]); };
                        ring_ensure<RingY>(rY[l], ringY, srcY, lo, hi, issued, lane);
                    }
                    if ((hit & (1 << (8 + l))) != 0 && (stepact & 0x55) != 0) {
                        const bool planar = ((KINDS & YK_PLANAR) != 0) && (((KINDS & YK_NV12) == 0) || (lf[l] & 2) != 0);
                        const int clo = (int)(sc_ & 8191u) - 1, chi0 = (int)((sc_ >> 13) & 8191u), chi = chi0 < rC[l].last ? chi0 : rC[l].last;
                        auto srcC = [&]() { if constexpr (KEEP_SRC) return sC[l]; else return ring_src(Ly.src.pl[1], Ly.src.pl[planar ? 2 : 1], cvC[l]); };
                        if (planar) ring_ensure<RingCP>(rC[l], ringY + (uint32_t)RingY::BYTES, srcC, clo, chi, issued, lane);
                        else ring_ensure<RingC2>(rC[l], ringY + (uint32_t)RingY::BYTES, srcC, clo, chi, issued, lane);
                    }
                };
                ys_seq<NL>(step);
            }
        }
        if constexpr (KINDS == YK_RGBINT) {
            if (fastR0 && j0 + 4 <= nrows && ((rowm[0] >> jt) & 15u) == 15u && ((unitm[0] >> jt) & 7u) == 7u) {
                if ((lf[0] & 8) != 0) fast_rgb_trip(j0, std::true_type{}); else fast_rgb_trip(j0, std::false_type{});
                continue;
            }
        }
        flush();                                         // (the previous trip's luma, after this trip's waits)
        trip(j0);
    }
    flush();
}

template <int TF, int NL, int KINDS>
__global__ __launch_bounds__(64 * YS_WAVES, ys_min_waves(KINDS, NL))
void tick_yuv_stream(const DTick *__restrict__ ticks, const DLayer *__restrict__ layers, int n_ticks, int strips_x, int chunks_y, int rows_per_chunk, int wave_bytes) {
    ys_body<TF, NL, KINDS, false>(ticks, layers, n_ticks, strips_x, chunks_y, rows_per_chunk, wave_bytes);
}

// one tick, descriptors by value (96 + NL x 368 bytes of kernel arguments): no descriptor copy in front of a lone tick's launch, no
// tick -> first_layer -> layer chain of dependent loads in front of its waves
template <int NL>
struct YsOne {
    DTick t;
    DLayer l[NL];
};
template <int TF, int NL, int KINDS>
__global__ __launch_bounds__(64 * YS_WAVES, ys_min_waves(KINDS, NL) > 5 ? 5 : ys_min_waves(KINDS, NL))
void tick_yuv_stream_one(const YsOne<NL> a, int strips_x, int chunks_y, int rows_per_chunk, int wave_bytes) {
    ys_body<TF, NL, KINDS, true>(&a.t, a.l, 1, strips_x, chunks_y, rows_per_chunk, wave_bytes);
}

// ---------------------------------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------------------------------
#define CHV_STR2(x) #x
#define CHV_STR(x) CHV_STR2(x)
const char *yuv_stream_build_flags() { return "tick_yuv_stream:abl=" CHV_STR(CHV_YS_ABL) ",strips_per_block=" CHV_STR(CHV_YS_BLOCK) ",rounds=" CHV_STR(CHV_YS_ROUNDS) ",carry=" CHV_STR(CHV_YS_CARRY); }

static bool ys_plane_ok(const DPlane &p) {
    return p.ptr && (((uintptr_t)p.ptr) & 15) == 0 && (p.pitch & 15) == 0 && ((p.w * p.comps) & 15) == 0 && p.w * p.comps >= 16 && p.h >= 1 && p.h <= 8190 &&
           (long)p.pitch * p.h < (1L << 31);
}
static bool ys_kind_rgb(int k) { return k == LK_YUV_FROM_RGB || k == LK_YUV_FROM_RGB_INT; }

// source classes of a launch (template parameter KINDS) and the LDS a wave needs for its widest tick
struct YsPlan { int kinds, nl, wave_bytes; };
static YsPlan ys_plan(const DTick *ticks, const DLayer *layers, int n_ticks) {
    YsPlan p{ 0, 1, 0 };
    for (int i = 0; i < n_ticks; i++) {
        p.nl = std::max(p.nl, ticks[i].n_layers);
        for (int l = 0; l < ticks[i].n_layers; l++) {
            const int k = layers[ticks[i].first_layer + l].kind;
            p.kinds |= k == LK_YUV_FROM_NV12 ? YK_NV12 : k == LK_YUV_FROM_Y420P ? YK_PLANAR : k == LK_YUV_FROM_RGB ? YK_RGB : YK_RGBINT;
        }
    }
    return p;
}
// the instantiation a launch runs: (kinds, layers) rounded up to what is compiled
static void ys_round(int tf, YsPlan &p) {
    const int own = tf == TF_NV12 ? YK_NV12 : YK_PLANAR;
    if (p.kinds == own) p.nl = p.nl <= 1 ? 1 : p.nl <= 2 ? 2 : 4;
    else if (p.kinds == YK_RGBINT && p.nl == 1) { }
    else if ((p.kinds & ~(own | YK_RGB)) == 0) { p.kinds = own | YK_RGB; p.nl = p.nl <= 3 ? 3 : 4; }
    else { p.kinds = YK_NV12 | YK_PLANAR | YK_RGB | YK_RGBINT; p.nl = 4; }
    if (p.kinds == own && p.nl == 4) { p.kinds = own | YK_RGB; }
}
static int ys_wave_bytes(const YsPlan &p, const DTick *ticks, const DLayer *layers, int n_ticks) {
    const int nb = ys_rgb_b(p.kinds);
    int rings = 0;
    for (int i = 0; i < n_ticks; i++) {
        int r = 0;
        for (int l = 0; l < ticks[i].n_layers; l++) {
            const int k = layers[ticks[i].first_layer + l].kind;
            // (the kernel's own rule: a launch without RGB kinds / without YUV kinds never asks the layer)
            const bool rgb = (p.kinds & (YK_RGB | YK_RGBINT)) && (!(p.kinds & (YK_NV12 | YK_PLANAR)) || ys_kind_rgb(k));
            const bool planar = !rgb && (p.kinds & YK_PLANAR) && (!(p.kinds & YK_NV12) || k == LK_YUV_FROM_Y420P);
            r += ys_layer_bytes_c(rgb, planar, nb);
        }
        rings = std::max(rings, r);
    }
    return rings + p.nl * YS_TAB_BYTES;
}

// transient: one tick, launched once (chv_composite / chv_run_kernel)
bool yuv_stream_eligible(int tf, const DTick *ticks, const DLayer *layers, int n_ticks, bool transient) {
    const int mode = switches().yuv_stream.load(std::memory_order_relaxed);
    if (n_ticks < 1 || !mode) return false;
    if (tf != TF_NV12 && tf != TF_Y420P) return false;
    if (mode == 1) {
        // Where the streaming kernel is the faster one today (profiles/r04_notes.md; same-call A/Bs against tick_yuv_wave):
        //   * launches whose layers are all integer-matrix RGB pictures — the encoder side's full-frame conversion: 0.567 against 0.640 ms per
        //     128 frames of 1080p BGRA -> NV12, a lone frame 13.3 against 16.7 us;
        //   * lone ticks of video layers only (descriptors as kernel arguments, 4- to 16-row chunks): 13.9 against 15.4 us for a 1080p tick.
        // Batches of video layers are level (0.378 against 0.354 ms per 128 ticks) and stay with the strip kernel; ticks with float RGB
        // overlays are slower through the rings (a 1080p mixer tick: 31.8 against 21.4 us; 1.08 against 0.65 ms per 128) and stay there too.
        bool all_int = true, any_rgb = false;
        for (int i = 0; i < n_ticks; i++)
            for (int l = 0; l < ticks[i].n_layers; l++) {
                const int k = layers[ticks[i].first_layer + l].kind;
                all_int = all_int && k == LK_YUV_FROM_RGB_INT;
                any_rgb = any_rgb || k == LK_YUV_FROM_RGB || k == LK_YUV_FROM_RGB_INT;
            }
        if (!(all_int || (transient && n_ticks == 1 && !any_rgb))) return false;
    }
    for (int i = 0; i < n_ticks; i++) {
        const DTick &T = ticks[i];
        if (!T.clear_first || T.n_layers < 1 || T.n_layers > YS_MAXL) return false;
        if ((T.W & 7) || (T.H & 3) || T.W < 8 || T.H < 4 || T.dst.pl[0].w != T.W || T.dst.pl[0].h != T.H) return false;
        const int np = tf == TF_NV12 ? 2 : 3;
        for (int p = 0; p < np; p++) {
            const DPlane &P = T.dst.pl[p];
            if (!P.ptr || (((uintptr_t)P.ptr) & 3) || (P.pitch & 3)) return false;
            if (p > 0 && (P.w != T.W / 2 || P.h != T.H / 2)) return false;
        }
        for (int l = 0; l < T.n_layers; l++) {
            const DLayer &Y = layers[T.first_layer + l];
            const bool rgb = ys_kind_rgb(Y.kind), nv12 = Y.kind == LK_YUV_FROM_NV12, planar = Y.kind == LK_YUV_FROM_Y420P;
            if (!(rgb || nv12 || planar)) return false;
            if (nv12 && tf != TF_NV12) return false;
            const int need = LF_AXIS_ALIGNED | LF_BOUNDED | LF_NO_FILL;
            if ((Y.flags & need) != need) return false;
            if (!(Y.u[U_OPACITY] - Y.u[U_OPACITY] == 0.f)) return false;
            const int npl = rgb ? 1 : nv12 ? 2 : 3;
            for (int p = 0; p < npl; p++) if (!ys_plane_ok(Y.src.pl[p])) return false;
            if (rgb && Y.src.pl[0].comps != 4) return false;
            if (!rgb && (Y.src.pl[0].comps != 1 || Y.src.pl[1].comps != (nv12 ? 2 : 1))) return false;
            if (planar && (Y.src.pl[2].w != Y.src.pl[1].w || Y.src.pl[2].h != Y.src.pl[1].h || Y.src.pl[2].pitch != Y.src.pl[1].pitch || Y.src.pl[2].comps != 1)) return false;
            // source texels per canvas pixel: u = (x / W * 2 - 1) * T0 * X0 + ...  =>  du/dx * w = 2 T0 X0 w / W
            const double kx = 2.0 * (double)Y.u[U_TRANSFORM + 0] * (double)Y.u[U_TEXTURE + 0], ky = 2.0 * (double)Y.u[U_TRANSFORM + 5] * (double)Y.u[U_TEXTURE + 5];
            if (!(kx > 0.0) || !(ky > 0.0)) return false;                                   // flips: the rings assume rising positions
            const double sxr = kx * Y.src.pl[0].w / (double)T.W, syr = ky * Y.src.pl[0].h / (double)T.H;
            // An 8-row step asks the 24-row luma ring for the rows ry(row 0) .. ry(row 7) + 1, and ring_ensure holds hi - lo <= ROWS - B = 16:
            // ry(7) - ry(0) <= ceil(7 * syr) must stay <= 15, i.e. syr < 15 / 7 = 2.143 (at 2.2 a step could span 17 rows and row 7 would tap a
            // slot already holding the row 24 below: stale luma).  2.1 is the bound the fixed cases and the fuzzers exercise.
            if (!std::isfinite(sxr) || !std::isfinite(syr) || syr > 2.1) return false;
            // bytes of a ring row a strip's taps span: 63 steps + tap 1 + rounding slack, the start's alignment
            if (rgb) { if (!((63.0 * sxr + 3.0) * 4.0 + 12.0 <= 320.0)) return false; }
            else {
                if (!(63.0 * sxr + 3.0 + 15.0 <= 128.0)) return false;                      // luma; NV12 chroma: half the texels, two bytes each
                const double sxc = kx * Y.src.pl[1].w / (double)T.W;
                if (nv12 && !((63.0 * sxc + 3.0) * 2.0 + 14.0 <= 128.0)) return false;
                if (planar && !(63.0 * sxc + 3.0 + 15.0 <= 96.0)) return false;
            }
        }
    }
    YsPlan p = ys_plan(ticks, layers, n_ticks);
    ys_round(tf, p);
    // (a wave's rings and tables: beyond 16 KB the LDS holds fewer than ten waves per CU, and the strip kernel is the better choice)
    return ys_wave_bytes(p, ticks, layers, n_ticks) <= 16 * 1024;
}

template <int TF, int NL, int KINDS>
static void ys_launch(dim3 grid, size_t lds, hipStream_t stream, const DTick *ticks_host, const DLayer *layers_host, const DTick *ticks, const DLayer *layers,
                      int n_ticks, int strips_x, int chunks_y, int rows, int wave_bytes) {
    if (!ticks) {
        YsOne<NL> a;
        std::memset((void *)&a, 0, sizeof a);
        a.t = ticks_host[0];
        a.t.first_layer = 0;
        for (int l = 0; l < NL && l < ticks_host[0].n_layers; l++) a.l[l] = layers_host[ticks_host[0].first_layer + l];
        hipLaunchKernelGGL((tick_yuv_stream_one<TF, NL, KINDS>), grid, dim3(64 * YS_WAVES), lds, stream, a, strips_x, chunks_y, rows, wave_bytes);
    } else {
        hipLaunchKernelGGL((tick_yuv_stream<TF, NL, KINDS>), grid, dim3(64 * YS_WAVES), lds, stream, ticks, layers, n_ticks, strips_x, chunks_y, rows, wave_bytes);
    }
}

// ticks == nullptr: one tick, launched with its descriptors (ticks_host[0], layers_host) as kernel arguments
hipError_t launch_yuv_stream(int tf, const DTick *ticks_host, const DLayer *layers_host, const DTick *ticks, const DLayer *layers, int n_ticks, int maxW, int maxH, hipStream_t stream) {
    YsPlan p = ys_plan(ticks_host, layers_host, n_ticks);
    ys_round(tf, p);
    const int wave_bytes = (ys_wave_bytes(p, ticks_host, layers_host, n_ticks) + 15) & ~15;
    const int strips_x = (maxW + 63) / 64;
    // rows per chunk (a multiple of 4; of 16 in launches that fill the chip, so that chroma leaves in full groups): enough chunks for
    // CHV_YS_ROUNDS rounds of waves; small launches (a Swift VideoMixer issues ONE tick and waits) are cut into about 4 800 short chains
    const int waves = ys_min_waves(p.kinds, p.nl);
    const long want = 1024L * waves * CHV_YS_ROUNDS;
    const long chunks = std::max<long>(1, want / std::max<long>(1, (long)n_ticks * strips_x));
    const long wave_rows = (long)n_ticks * strips_x * maxH;
    long rows_small = std::min<long>(16, std::max<long>(4, (wave_rows + CHV_YS_SMALL_WAVES - 1) / CHV_YS_SMALL_WAVES));
    rows_small = (rows_small + 3) & ~3L;
    long rows_large = ((maxH + chunks - 1) / chunks + 15) & ~15L;
    int rows = (int)std::max<long>(rows_small, rows_large > 16 ? rows_large : 0);
    if (CHV_YS_ROWS_FIXED > 0) rows = (CHV_YS_ROWS_FIXED + 3) & ~3;
    rows = std::max(4, rows);
    const int chunks_y = (maxH + rows - 1) / rows;
    const long total = ((long)n_ticks * chunks_y * strips_x + YS_WAVES - 1) / YS_WAVES;
    dim3 grid((unsigned)(((total + 7) / 8) * 8));
    const size_t lds = (size_t)YS_WAVES * (size_t)wave_bytes;
    if (!ticks && (n_ticks != 1 || !layers_host)) return hipErrorInvalidValue;
#define CHV_YS_GO(TFV, NLV, KV) ys_launch<TFV, NLV, KV>(grid, lds, stream, ticks_host, layers_host, ticks, layers, n_ticks, strips_x, chunks_y, rows, wave_bytes)
#define CHV_YS_TF(TFV, OWN) do { \
        if (p.kinds == OWN && p.nl == 1) CHV_YS_GO(TFV, 1, OWN); \
        else if (p.kinds == OWN && p.nl == 2) CHV_YS_GO(TFV, 2, OWN); \
        else if (p.kinds == YK_RGBINT && p.nl == 1) CHV_YS_GO(TFV, 1, YK_RGBINT); \
        else if (p.kinds == (OWN | YK_RGB) && p.nl == 3) CHV_YS_GO(TFV, 3, (OWN | YK_RGB)); \
        else if (p.kinds == (OWN | YK_RGB)) CHV_YS_GO(TFV, 4, (OWN | YK_RGB)); \
        else CHV_YS_GO(TFV, 4, (YK_NV12 | YK_PLANAR | YK_RGB | YK_RGBINT)); } while (0)
    if (tf == TF_NV12) CHV_YS_TF(TF_NV12, YK_NV12); else CHV_YS_TF(TF_Y420P, YK_PLANAR);
#undef CHV_YS_TF
#undef CHV_YS_GO
    return hipGetLastError();
}

}  // namespace chv

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
    constexpr bool ONE = false;
#include "kernels_wave_body.hip.inc"
}

// ONE: a lone tick on a cleared BGRA canvas (8-row strips, up to WAVE_ONE_LAYERS layers) whose descriptors are the kernel's first ARGUMENT: `ticks`
// and `layers` point into the kernarg segment (constant address space: scalar loads whatever happens in between), no ring slot and no copy in
// front of the launch.  A twin instead of the trailing argument tick_yuv_wave carries (wave_common.hip.h): a pointer CHOSEN at run time cannot be
// reloaded from the kernarg segment at will, and the two scalar registers that cost put 2 - 14 vector registers of the 8-row instantiations
// (80 VGPRs: six waves) into scratch — and a kernel with a scratch segment is dispatched 2 us later than one without (measured on the twin at six
// waves: 28.5 against 26.4 us between events).  The twins are built for FIVE waves (96 VGPRs, nothing in scratch): a lone tick up to 1080p is
// at most 4 080 strips on 5 120 wave slots, all resident at once either way.  KINDS 1 / 2 / 5 / 7: what launch_bgra_wave_t picks for NV12
// video, planar video, NV12 video + RGB overlays and everything else without per-pixel layers.
template <int KINDS, bool CACHED>
__global__ __launch_bounds__(WAVE_BLOCK, CHV_WAVE_MINW - 1) void tick_bgra_wave_one(const WaveOne, int strips_x, int strips_y, uint32_t strips_magic, uint32_t strips_x_magic,
                                                                                  int p0pitch, int p0rows, int p1pitch, int p1rows, int planar_any) {
    constexpr int WTH = 8;
    constexpr bool CLEAR = true, ONE = true;
    const uint64_t ka = (uint64_t)(uintptr_t)__builtin_amdgcn_kernarg_segment_ptr();             // (`WaveOne` is the first argument: offset 0)
    const DTick *__restrict__ ticks = (const DTick *)(const CHV_CONSTANT DTick *)(uintptr_t)ka;
    const DLayer *__restrict__ layers = (const DLayer *)(const CHV_CONSTANT DLayer *)(uintptr_t)(ka + sizeof(DTick));
    const int n_ticks = 1;
#include "kernels_wave_body.hip.inc"
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
                              int strips_x, int strips_y, uint32_t strips_magic, uint32_t strips_x_magic, int p0pitch, int p0rows, int p1pitch, int p1rows, int planar, int kinds,
                              const WaveOne *one_arg) {
    if (one_arg) {
        // a lone tick whose descriptors are the kernel's first argument (launch_wave_layers asked wave_layers_by_value first)
        if (rows != 8 || !clear || (kinds & 8) || kinds == 4) return hipErrorNotSupported;
#define CHV_LAUNCH_ONE(K) hipLaunchKernelGGL((tick_bgra_wave_one<K, CACHED>), grid, dim3(WAVE_BLOCK), lds, stream, *one_arg, strips_x, strips_y, strips_magic, strips_x_magic, \
                                             p0pitch, p0rows, p1pitch, p1rows, planar)
        if (kinds == 1) CHV_LAUNCH_ONE(1); else if (kinds == 2) CHV_LAUNCH_ONE(2); else if (kinds == 5) CHV_LAUNCH_ONE(5); else CHV_LAUNCH_ONE(7);
#undef CHV_LAUNCH_ONE
        return hipGetLastError();
    }
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
                           int strips_x, int strips_y, uint32_t strips_magic, uint32_t strips_x_magic, int p0pitch, int p0rows, int p1pitch, int p1rows, int planar, int kinds, \
                           const WaveOne *one_arg
#if CHV_WAVE_TU == 0
template hipError_t launch_bgra_wave_t<false>(CHV_BGRA_WAVE_ARGS);
extern template hipError_t launch_bgra_wave_t<true>(CHV_BGRA_WAVE_ARGS);            // kernels_wave_cached.hip.cpp

hipError_t launch_bgra_wave(int rows, bool clear, dim3 grid, size_t lds, hipStream_t stream, const DTick *ticks, const DLayer *layers, int n_ticks,
                            int strips_x, int strips_y, uint32_t strips_magic, uint32_t strips_x_magic, int p0pitch, int p0rows, int p1pitch, int p1rows, int planar, int kinds, bool cached,
                            const WaveOne *one_arg) {
    // bit 5 of `planar`: rectangles are staged by DMA where their shape allows (the RGB-only instantiation)
    {
        const int dma_on = switches().wave_dma.load(std::memory_order_relaxed);        // (CHV_WAVE_DMA=0 / chv_debug_set_switch: register staging)
        if (CHV_WAVE_DMA && dma_on && (kinds == 4 || (CHV_WAVE_DMA > 1 && (kinds & 4)))) planar |= 32;
    }
    return cached ? launch_bgra_wave_t<true>(rows, clear, grid, lds, stream, ticks, layers, n_ticks, strips_x, strips_y, strips_magic, strips_x_magic, p0pitch, p0rows, p1pitch, p1rows, planar, kinds, one_arg)
                  : launch_bgra_wave_t<false>(rows, clear, grid, lds, stream, ticks, layers, n_ticks, strips_x, strips_y, strips_magic, strips_x_magic, p0pitch, p0rows, p1pitch, p1rows, planar, kinds, one_arg);
}
#else
template hipError_t launch_bgra_wave_t<true>(CHV_BGRA_WAVE_ARGS);
#endif

}  // namespace chv

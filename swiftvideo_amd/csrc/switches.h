// switches.h — measurement / test switches of the path selection (host side only).
//
// The environment is read ONCE, at the first use in the process (getenv on a hot path is undefined behaviour next to a
// setenv in another thread, and the mixer and uploader threads run concurrently): CHV_FORCE_GENERAL, CHV_BGRA_PATH,
// CHV_WAVE_ROWS, CHV_TILE_ROWS, CHV_SAME_GEOM, CHV_DESC, CHV_STREAM, CHV_YUV_STREAM, CHV_WAVE_DMA, CHV_PASS_FUSE, CHV_GEOM_CACHE.  Tests and A/B tools change them afterwards through chv_debug_set_switch (include/chipvideo.h),
// never through the environment.  Every value is an atomic int; 0 = "the library decides".
#pragma once
#include <atomic>

namespace chv {

struct Switches {
    std::atomic<int> force_general{0};   // 1: every tick through the general kernels
    std::atomic<int> bgra_path{0};       // 1: wave kernel also for one-YUV-layer BGRA ticks; 2: the tiled kernel wherever it applies
    std::atomic<int> wave_rows{0};       // 8 | 16: strip height of the wave kernels
    std::atomic<int> tile_rows{0};       // 16 | 32: tile height of the tiled YUV -> BGRA kernel
    std::atomic<int> same_geom{1};       // 0: do not share a layer's geometry with its predecessor (A/B of LF_SAME_GEOM)
    std::atomic<int> desc_host{0};       // CHV_DESC, a transient launch's descriptors: 0 kernel arguments where the kernel takes them (tick_bgra_stream_one), else a device
                                         // copy; 1 (host) read from the pinned host ring; 2 (device) the device copy always (A/Bs)
    std::atomic<int> stream{1};          // 0: never tick_bgra_stream (A/B; CHV_BGRA_PATH=stream: also for one-layer ticks)
    std::atomic<int> yuv_stream{1};      // CHV_YUV_STREAM: 0 never tick_yuv_stream (4:2:0 canvases keep tick_yuv_wave); 1 (default) where it measured faster
                                         // (kernels_stream_yuv.hip.cpp::yuv_stream_eligible); force: every eligible launch (A/Bs and tests)
    std::atomic<int> wave_dma{1};        // CHV_WAVE_DMA: 0 the RGB-only strip kernel stages its rectangles through registers like the others (A/B, and the
                                         // fuzzers' way to the non-DMA staging of that instantiation); 1 (default) by LDS-DMA where the shape allows
    std::atomic<int> geom_cache{1};      // CHV_GEOM_CACHE: 0 the strip kernels compute every layer's per-strip geometry in place also in batches (A/B, and the
                                         // fuzzers' way to that path); 1 (default) batches keep it in tables built at their SECOND launch with a
                                         // configuration (geom_cache.h: a batch run once never pays); eager: at the first one (the test suite, whose
                                         // batches mostly run once, so that its fuzzers reach the table-reading kernels)
    std::atomic<int> pass_fuse{1};       // CHV_PASS_FUSE: 1 (default) picture kernels issued inside chv_pass_begin ... chv_pass_end are held and leave as the one
                                         // fused launch chv_composite would make of them (chipvideo.cpp: PendingPass); 0 every chv_run_kernel launches at once
};
Switches &switches();                    // (chipvideo.cpp; initialised from the environment on first use)

}  // namespace chv

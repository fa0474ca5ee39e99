// geom_cache.h — per-batch tables of the strip kernels' per-layer geometry (host side: what chv_batch owns; device side: wave_common.hip.h).
//
// WaveStrip::setup (wave_common.hip.h) derives, per layer and strip, the column entries (per lane), the row table (LDS) and the staging
// rectangles from nothing but the layer's three matrices, the source planes' sizes, the canvas size, the strip's position and the launch's LDS
// layout — ~230 vector + ~300 scalar instructions, 30 % of the launch on the strip kernels' one- and two-layer ticks
// (profiles/r06_notes.md section 9).  In a batch the same (layer geometry, strip) pair recurs in every tick: the values are computed ONCE per
// batch and launch configuration by a small kernel that runs that very code (same inputs, same instructions, same bits), stored factored into
// one record per strip COLUMN and one per strip ROW (they are functions of the column / of the row alone; what mixes both — staged or not,
// inside or not, rectangle at a picture edge or not — is a flag word per strip), and the tick kernels load them.
#pragma once
#include <stddef.h>
#include <stdint.h>

namespace chv {

struct DLayer;

// what the tables were built for: any difference rebuilds them (strip height and LDS layout are chosen per LAUNCH, launch_wave_layers)
struct GeomConfig {
    int32_t target_format, wth, p0pitch, p0rows, p1pitch, p1rows, planar_any, strips_x, strips_y, n_layers;
    bool operator==(const GeomConfig &o) const {
        return target_format == o.target_format && wth == o.wth && p0pitch == o.p0pitch && p0rows == o.p0rows && p1pitch == o.p1pitch && p1rows == o.p1rows &&
               planar_any == o.planar_any && strips_x == o.strips_x && strips_y == o.strips_y && n_layers == o.n_layers;
    }
};

struct GeomCache {
    bool built = false;          // tables hold `config`'s geometry and the batch's device layers point at them
    bool patched = false;        // the device layers carry table pointers (cleared again when the switch goes off)
    GeomConfig config{};
    bool seen = false;           // a launch with `seen_config` has been issued (tables are built at the second one: a batch run once never pays)
    GeomConfig seen_config{};
    void *tables = nullptr;      // device: every class's tables, back to back
    void *jobs = nullptr;        // device: the precompute kernel's job list (kept: freed with the cache)
    size_t bytes = 0;
    int classes = 0;
    // the batch's layer descriptors (chv_batch): the host copy is patched and re-sent when tables are (re)built
    DLayer *d_layers = nullptr;
    DLayer *h_layers = nullptr;
    int n_layers = 0;
};

// the cache of the batch whose launch is being issued by this thread (nullptr: a transient launch — geometry is computed in the kernel)
GeomCache *&geom_cache_current();
// frees the device memory of a cache (chv_batch_destroy; the device is current)
void geom_cache_release(GeomCache &c);

}  // namespace chv

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
    bool owns = true;            // `tables` / `jobs` are this cache's to free (false: given to the device's store, or taken from it)
    bool force_build = false;    // build at the next launch whatever has been seen (a lone tick whose geometry the store has seen before)
    void *tables = nullptr;      // device: every class's tables, back to back
    void *jobs = nullptr;        // device: the precompute kernel's job list (kept: freed with the cache)
    size_t bytes = 0;
    int classes = 0;
    // the batch's layer descriptors (chv_batch): the host copy is patched and re-sent when tables are (re)built
    DLayer *d_layers = nullptr;
    DLayer *h_layers = nullptr;
    int n_layers = 0;
};

// ---- the device's STORE of tables ----------------------------------------------------------------------------------------------------
// A batch is bound to its pictures' addresses, so a host that batches builds a new batch for every group of frames and runs it once: tables owned
// by the batch would serve benchmarks only.  What recurs in a real pipeline is the GEOMETRY (a scene's layers keep their matrices and sizes for
// seconds), so a table, once built, is given to a per-device store keyed by (the class's set-up inputs, the launch configuration) and lives
// there (bounded: kGeomStoreBytes; beyond it batches keep their tables to themselves, as before).  Anything about to be launched through the
// strip kernels asks the store BEFORE its descriptors go to the device — a batch at creation, a lone tick before its descriptor slot is copied —
// and, where every staged layer has a table, gets its layers pointed at them for nothing.  A geometry seen for the second time (by any batch or
// tick of the device) is built at that launch, as a batch's second launch always did.
constexpr size_t kGeomStoreBytes = (size_t)256 << 20;
constexpr size_t kGeomStoreSightings = 8192;      // geometries asked for and not found that the store remembers (animated layers: one per tick)
// Point `layers_host` (the launch's layers, about to be copied to the device) at the store's tables: true when EVERY layer the kernels set up
// has one (the layers are patched), false otherwise (their table pointers are zeroed).  `cfg`: the launch configuration the strip kernels'
// launcher will use (n_layers = n_layers_total).  `want_build`: not covered, but every missing geometry has been seen before (or tables are
// built eagerly): worth building at this launch.  Counts the sighting.
bool geom_store_patch(int target_format, const DTick *ticks_host, DLayer *layers_host, int n_ticks, int maxW, int maxH, int n_layers_total,
                      GeomConfig *cfg, bool *want_build);
// the outcome of geom_store_patch for the TRANSIENT launch this thread is about to issue (no batch: launch_wave_layers looks here)
struct GeomTransient { bool covered = false; GeomConfig cfg{}; };
GeomTransient &geom_transient_current();
// the store of the current device in numbers (chv_debug_get_counter; tests and probes): 0 launches whose layers were all pointed at the store's
// tables before their descriptors travelled (batches at creation, lone ticks), 1 batches pointed at them at a launch, 2 builds given to the
// store, 3 bytes it holds, 4 tables (geometry classes x launch configurations) it holds
uint64_t geom_store_counter(int which);

// the cache of the batch whose launch is being issued by this thread (nullptr: a transient launch — geometry is computed in the kernel)
GeomCache *&geom_cache_current();
// frees the device memory of a cache (chv_batch_destroy; the device is current)
void geom_cache_release(GeomCache &c);

}  // namespace chv

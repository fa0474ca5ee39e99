// tests/stubhip/stub_launchers.cpp — stand-ins for what chipvideo.cpp imports from the kernel translation units (path selection and
// launchers), for the sanitizer builds of the host runtime (tests/test_sanitizers.py).  TEST INFRASTRUCTURE.
//
// A "launch" is a closure on the stream (stub_runtime.cpp) that, when the stream gets to it, reads the tick and layer descriptors it was given,
// checks them, touches the first and last byte of every plane they name (so a buffer freed or a descriptor slot recycled too early is a
// sanitizer report) and stamps the canvas.  Launches whose descriptors the host still owns at launch time (launch_tick_fast: the host copies)
// remember a checksum and compare it with what the DEVICE copy holds when the launch runs: a ring slot overwritten in flight is an abort.
// Path selection keeps the product's shape where the host logic depends on it: BGRA ticks of 1..4 YUV layers on a cleared canvas take the
// by-value route (kernel arguments, no descriptor copy), everything else with layers the ring + device-twin route, layerless clears their
// own; CHV_FORCE_GENERAL sends everything to launch_tick_general.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstring>
#include <vector>

#include "../../swiftvideo_amd/csrc/device_types.h"
#include "../../swiftvideo_amd/csrc/switches.h"
#include "../../swiftvideo_amd/csrc/geom_cache.h"
#include <map>
#include <mutex>
#include <cstring>

namespace chv {
enum { FP_NONE = -1, FP_WAVE = 2, FP_STREAM = 5, FP_CLEAR = 6 };

static uint64_t fnv(const void *p, size_t n, uint64_t h = 1469598103934665603ull) {
    const unsigned char *b = (const unsigned char *)p;
    for (size_t i = 0; i < n; i++) { h ^= b[i]; h *= 1099511628211ull; }
    return h;
}
static void touch(const DPlane &p, bool write) {
    if (!p.ptr || p.w <= 0 || p.h <= 0) return;
    volatile uint8_t *a = p.ptr, *z = p.ptr + (size_t)(p.h - 1) * p.pitch + (size_t)p.w * p.comps - 1;
    uint8_t x = *a, y = *z;
    if (write) { *a = (uint8_t)(x + 1); *z = (uint8_t)(y + 1); }
}
static void run_tick(const DTick &T, const DLayer *L) {
    if (T.n_layers < 0 || T.n_layers > 4096 || T.W <= 0 || T.H <= 0) { fprintf(stderr, "stub kernel: corrupt tick descriptor (%d layers, %dx%d)\n", T.n_layers, T.W, T.H); abort(); }
    for (int l = 0; l < T.n_layers; l++) {
        const DLayer &Y = L[T.first_layer + l];
        if (Y.kind < 0 || Y.kind > 7) { fprintf(stderr, "stub kernel: corrupt layer descriptor (kind %d)\n", Y.kind); abort(); }
        for (int p = 0; p < 3; p++) touch(Y.src.pl[p], false);
    }
    for (int p = 0; p < 3; p++) touch(T.dst.pl[p], true);
}
static hipError_t enqueue_ticks(const DTick *ticks_host, const DLayer *layers_host, const DTick *ticks, const DLayer *layers, int n, hipStream_t st) {
    if (stubhip_launch_should_fail()) return hipErrorLaunchFailure;
    if (!ticks) {
        // descriptors by value (kernel arguments): copied NOW
        int nl = ticks_host[0].n_layers;
        DTick t = ticks_host[0];
        std::vector<DLayer> ls(layers_host + t.first_layer, layers_host + t.first_layer + nl);
        t.first_layer = 0;
        stubhip_enqueue(st, [t, ls] { run_tick(t, ls.data()); });
        return hipSuccess;
    }
    uint64_t want = 0;
    int total_layers = 0;
    if (ticks_host) {
        want = fnv(ticks_host, sizeof(DTick) * (size_t)n);
        for (int i = 0; i < n; i++) total_layers = ticks_host[i].first_layer + ticks_host[i].n_layers > total_layers ? ticks_host[i].first_layer + ticks_host[i].n_layers : total_layers;
    }
    const bool check = ticks_host != nullptr;
    stubhip_enqueue(st, [=] {
        if (check && fnv(ticks, sizeof(DTick) * (size_t)n) != want) { fprintf(stderr, "stub kernel: the tick descriptors changed between launch and execution (a ring slot recycled in flight)\n"); abort(); }
        for (int i = 0; i < n; i++) run_tick(ticks[i], layers);
    });
    return hipSuccess;
}

GeomCache *&geom_cache_current() { static thread_local GeomCache *cur = nullptr; return cur; }
void geom_cache_release(GeomCache &) {}                    // (the stand-in launchers build no tables)
// A stand-in for the device's store of tables (geom_cache.h), so that the library's hooks run under the sanitizers: the second launch of a scene
// "builds" (chipvideo.cpp hands its launch a temporary GeomCache), from the third one on its layers are "covered" (pointed at a table).
GeomTransient &geom_transient_current() { static thread_local GeomTransient t; return t; }
bool geom_store_patch(int, const DTick *ticks_host, DLayer *layers_host, int n_ticks, int, int, int n_layers_total, GeomConfig *cfg, bool *want_build) {
    static std::mutex mu;
    static std::map<uint64_t, int> seen;
    uint64_t h = 1469598103934665603ull;
    for (int i = 0; i < n_ticks; i++)
        for (int l = 0; l < ticks_host[i].n_layers; l++) h = h * 1099511628211ull ^ fnv(layers_host[ticks_host[i].first_layer + l].u, sizeof layers_host[0].u);
    int c;
    { std::lock_guard<std::mutex> lk(mu); c = ++seen[h]; }
    memset(cfg, 0, sizeof *cfg);
    cfg->n_layers = n_layers_total;
    *want_build = c == 2;
    for (int i = 0; i < n_ticks; i++)
        for (int l = 0; l < ticks_host[i].n_layers; l++) { DLayer &L = layers_host[ticks_host[i].first_layer + l]; L.pad2[0] = c >= 3 ? 0x1000 : 0; L.pad2[1] = 0; }
    return c >= 3;
}
uint64_t geom_store_counter(int) { return 0; }
bool fast_path_is_wave(int path) { return path == FP_WAVE; }
bool wave_layers_by_value(int, const DTick *t, const DLayer *) { return t->n_layers >= 1 && t->n_layers <= WAVE_ONE_LAYERS; }
const char *bgra_wave_build_flags() { return "stub:abl=0"; }
const char *yuv_wave_build_flags() { return "stub:abl=0"; }
const char *bgra_stream_build_flags() { return "stub:abl=0"; }
const char *yuv_stream_build_flags() { return "stub:abl=0"; }
const char *lanczos_build_flags() { return "stub:abl=0"; }
hipError_t launch_tick_general(int, const DTick *, const DLayer *, const DTick *ticks, const DLayer *layers, int n_ticks, int, int, hipStream_t stream) {
    return enqueue_ticks(nullptr, nullptr, ticks, layers, n_ticks, stream);
}
hipError_t launch_selftest(float *, const float *, uint8_t *, const float *, const float *, float *, int, hipStream_t) { return hipErrorNotSupported; }
hipError_t launch_selftest_pack(const int *, const int *, const int *, uint32_t *, int, hipStream_t) { return hipErrorNotSupported; }
hipError_t launch_selftest_pack_codes(const float *, uint32_t *, int, hipStream_t) { return hipErrorNotSupported; }
hipError_t launch_selftest_matrices(int, int, uint32_t *, uint32_t *, hipStream_t) { return hipErrorNotSupported; }
const char *fast_path_name(int path) { return path == FP_STREAM ? "stub_by_value" : path == FP_WAVE ? "stub_ring" : path == FP_CLEAR ? "stub_clear" : "none"; }
bool fast_path_by_value(int path) { return path == FP_STREAM || path == FP_CLEAR; }
int split_stream_prefix(const DTick *ticks, const DLayer *layers, int n_ticks) {
    // "2..4 videos, then something else" in every tick -> two launches (the product's split batches)
    int k = 4;
    for (int i = 0; i < n_ticks; i++) {
        int v = 0;
        while (v < ticks[i].n_layers && v < 4 && (layers[ticks[i].first_layer + v].kind == LK_BGRA_FROM_NV12 || layers[ticks[i].first_layer + v].kind == LK_BGRA_FROM_Y420P)) v++;
        if (v >= ticks[i].n_layers) v = ticks[i].n_layers - 1;
        k = v < k ? v : k;
    }
    return k >= 2 ? k : 0;
}
int fast_path_stream_bgra() { return FP_STREAM; }
int select_fast_path(int target_format, const DTick *ticks, const DLayer *layers, int n_ticks, bool transient) {
    if (n_ticks <= 0 || switches().force_general.load()) return FP_NONE;
    if (transient && n_ticks == 1 && ticks[0].n_layers == 0 && ticks[0].clear_first && target_format == TF_BGRA) return FP_CLEAR;
    bool videos = target_format == TF_BGRA;
    for (int i = 0; i < n_ticks && videos; i++) {
        videos = ticks[i].clear_first && ticks[i].n_layers >= 1 && ticks[i].n_layers <= 4;
        for (int l = 0; l < ticks[i].n_layers && videos; l++) { int k = layers[ticks[i].first_layer + l].kind; videos = k == LK_BGRA_FROM_NV12 || k == LK_BGRA_FROM_Y420P; }
    }
    if (videos) return FP_STREAM;
    for (int i = 0; i < n_ticks; i++) if (ticks[i].n_layers < 1) return FP_NONE;
    return FP_WAVE;
}
int select_tail_path(int target_format, const DTick *ticks, const DLayer *layers, int n_ticks) { return select_fast_path(target_format, ticks, layers, n_ticks, false); }
hipError_t launch_tick_fast(int path, const DTick *ticks_host, const DLayer *layers_host, const DTick *ticks, const DLayer *layers, int n_ticks, int, int, hipStream_t stream) {
    if (path == FP_CLEAR) {
        if (stubhip_launch_should_fail()) return hipErrorLaunchFailure;
        DPlane d = ticks_host[0].dst.pl[0];
        stubhip_enqueue(stream, [d] { touch(d, true); });
        return hipSuccess;
    }
    return enqueue_ticks(ticks_host, layers_host, ticks, layers, n_ticks, stream);
}
hipError_t launch_snd_s16i(int16_t *out, const int16_t *const *in, int count, int n, const float *, const float *, hipStream_t stream) {
    if (stubhip_launch_should_fail()) return hipErrorLaunchFailure;
    std::vector<const int16_t *> ins(in, in + count);
    stubhip_enqueue(stream, [=] { if (n > 0) { volatile int16_t *o = out; o[0] = (int16_t)(o[0] + 1); o[n - 1] = (int16_t)(o[n - 1] + 1); for (auto p : ins) { volatile const int16_t *q = p; (void)q[0]; (void)q[n - 1]; } } });
    return hipSuccess;
}
hipError_t launch_me_fullsearch(const DPlane &out, const DPlane &ref, const DPlane &cur, const int32_t *, const int32_t *, const int32_t *, const float *, hipStream_t stream) {
    if (stubhip_launch_should_fail()) return hipErrorLaunchFailure;
    stubhip_enqueue(stream, [=] { touch(ref, false); touch(cur, false); touch(out, true); });
    return hipSuccess;
}
hipError_t launch_lanczos(const DPlane &dst, const DPlane &src, const int32_t *fx, const float *wx, int tx, const int32_t *fy, const float *wy, int ty, hipStream_t stream,
                          const DPlane *batch, int n_batch, int, int, int) {
    if (stubhip_launch_should_fail()) return hipErrorLaunchFailure;
    // the coefficient tables live in device memory owned by the context's cache: read them when the launch RUNS
    stubhip_enqueue(stream, [=] {
        volatile const int32_t *a = fx, *b = fy; volatile const float *c = wx, *d = wy;
        (void)a[0]; (void)b[0]; (void)c[0]; (void)d[0]; (void)tx; (void)ty;
        if (batch) { for (int i = 0; i < n_batch; i++) { touch(batch[2 * i + 1], false); touch(batch[2 * i], true); } }
        else { touch(src, false); touch(dst, true); }
    });
    return hipSuccess;
}
}  // namespace chv

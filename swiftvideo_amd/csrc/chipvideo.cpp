// chipvideo.cpp — host runtime behind include/chipvideo.h.
//
// Owns devices, contexts (one HIP stream each), device buffers, pitched
// transfers with a pinned staging ring, descriptor building/validation and
// kernel dispatch.  There is no CPU pixel path in this library: every entry
// that would touch pixels either launches a gfx950 kernel or returns an error.
#include "../../include/chipvideo.h"

#include <hip/hip_runtime.h>
#include <hip/hiprtc.h>

#include <algorithm>
#include <atomic>
#include <cctype>
#include <cstdarg>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <vector>

#include "device_types.h"
#include "switches.h"
#include "geom_cache.h"

namespace chv {
const char *bgra_wave_build_flags();      // kernels_wave.hip.cpp
const char *yuv_wave_build_flags();       // kernels_wave_yuv.hip.cpp
const char *bgra_stream_build_flags();   // kernels_stream.hip.cpp
const char *yuv_stream_build_flags();    // kernels_stream_yuv.hip.cpp
const char *lanczos_build_flags();       // kernels_lanczos.hip.cpp
hipError_t launch_tick_general(int target_format, const DTick *ticks_host, const DLayer *layers_host, const DTick *ticks, const DLayer *layers,
                               int n_ticks, int maxW, int maxH, hipStream_t stream);
hipError_t launch_selftest(float *out_f, const float *in_f, uint8_t *out_c, const float *num,
                           const float *den, float *out_q, int n, hipStream_t stream);
hipError_t launch_selftest_pack(const int *b, const int *g, const int *r, uint32_t *out, int n, hipStream_t stream);
hipError_t launch_selftest_pack_codes(const float *in, uint32_t *out, int n, hipStream_t stream);
hipError_t launch_selftest_matrices(int dir, int csc, uint32_t *out, uint32_t *mism, hipStream_t stream);
// kernels_fast.hip.cpp
const char *fast_path_name(int path);
bool wave_layers_by_value(int target_format, const DTick *tick_host, const DLayer *layers_host);      // kernels_wave_yuv.hip.cpp
bool fast_path_is_wave(int path);                // the strip kernels (they take per-layer geometry tables: geom_cache.h)
bool fast_path_by_value(int path);
int split_stream_prefix(const DTick *ticks, const DLayer *layers, int n_ticks);      // kernels_fast.hip.cpp
int fast_path_stream_bgra();
int select_tail_path(int target_format, const DTick *ticks, const DLayer *layers, int n_ticks);      // a lone tick's descriptors can travel as kernel arguments (ticks == layers == nullptr)
int select_fast_path(int target_format, const DTick *ticks, const DLayer *layers, int n_ticks, bool transient = false);   // transient: one tick, launched once
hipError_t launch_tick_fast(int path, const DTick *ticks_host, const DLayer *layers_host,
                            const DTick *ticks, const DLayer *layers, int n_ticks,
                            int maxW, int maxH, hipStream_t stream);
// kernels_idle.hip.cpp
hipError_t launch_snd_s16i(int16_t *out, const int16_t *const *in, int count, int n, const float *gain, const float *fade, hipStream_t stream);
hipError_t launch_me_fullsearch(const DPlane &out, const DPlane &ref, const DPlane &cur, const int32_t *block, const int32_t *window, const int32_t *image,
                                const float *cost256, hipStream_t stream);
// kernels_lanczos.hip.cpp
hipError_t launch_lanczos(const DPlane &dst, const DPlane &src, const int32_t *fx, const float *wx,
                          int tx, const int32_t *fy, const float *wy, int ty, hipStream_t stream,
                          const DPlane *batch, int n_batch, int stride_x, int first_x, int stride_y);
}  // namespace chv

using namespace chv;

static void g_detail_set(const char *msg);

// ---------------------------------------------------------------------------
// switches (switches.h) and build flags
// ---------------------------------------------------------------------------
static int parse_switch(const char *name, const char *value, int *out) {
    const std::string n = name ? name : "", v = value ? value : "";
    if (n == "CHV_FORCE_GENERAL") { *out = v == "1" ? 1 : 0; return 0; }
    if (n == "CHV_BGRA_PATH") { *out = v == "wave" ? 1 : v == "tiled" ? 2 : v == "stream" ? 3 : 0; return 1; }
    if (n == "CHV_WAVE_ROWS") { *out = v == "8" ? 8 : v == "16" ? 16 : 0; return 2; }
    if (n == "CHV_TILE_ROWS") { *out = v == "16" ? 16 : v == "32" ? 32 : 0; return 3; }
    if (n == "CHV_SAME_GEOM") { *out = v == "0" ? 0 : 1; return 4; }
    if (n == "CHV_DESC") { *out = v == "host" ? 1 : v == "device" ? 2 : 0; return 5; }
    if (n == "CHV_STREAM") { *out = v == "0" ? 0 : 1; return 6; }
    if (n == "CHV_YUV_STREAM") { *out = v == "0" ? 0 : (v == "force" || v == "2") ? 2 : 1; return 7; }
    if (n == "CHV_WAVE_DMA") { *out = v == "0" ? 0 : 1; return 8; }
    if (n == "CHV_PASS_FUSE") { *out = v == "0" ? 0 : 1; return 9; }
    if (n == "CHV_GEOM_CACHE") { *out = v == "0" ? 0 : v == "eager" ? 2 : 1; return 10; }
    return -1;
}
static void store_switch(Switches &s, int which, int val) {
    std::atomic<int> *slots[11] = { &s.force_general, &s.bgra_path, &s.wave_rows, &s.tile_rows, &s.same_geom, &s.desc_host, &s.stream, &s.yuv_stream, &s.wave_dma,
                                    &s.pass_fuse, &s.geom_cache };
    slots[which]->store(val, std::memory_order_relaxed);
}
Switches &chv::switches() {
    static Switches s;
    static std::once_flag once;
    std::call_once(once, [] {
        static const char *const names[11] = { "CHV_FORCE_GENERAL", "CHV_BGRA_PATH", "CHV_WAVE_ROWS", "CHV_TILE_ROWS", "CHV_SAME_GEOM", "CHV_DESC", "CHV_STREAM", "CHV_YUV_STREAM",
                                               "CHV_WAVE_DMA", "CHV_PASS_FUSE", "CHV_GEOM_CACHE" };
        for (const char *n : names) {
            const char *v = getenv(n);
            int val = 0, which = v ? parse_switch(n, v, &val) : -1;
            if (which >= 0) store_switch(s, which, val);
        }
    });
    return s;
}
extern "C" int chv_debug_set_switch(const char *name, const char *value) {
    int val = 0;
    Switches &s = switches();                     // (environment first, so that a later first use cannot overwrite this)
    const int which = parse_switch(name, value, &val);
    if (which < 0) { g_detail_set("unknown switch"); return CHV_ERR_INVALID_VALUE; }
    if (!value || !*value) val = (which == 4 || which == 6 || which == 7 || which == 8 || which == 9 || which == 10) ? 1 : 0;      // empty / NULL: back to "the library decides"
    store_switch(s, which, val);
    return CHV_OK;
}
extern "C" int chv_debug_get_counter(const char *name, unsigned long long *value) {
    static const char *const names[] = { "geom_store_patched", "geom_store_batch_hits", "geom_store_builds", "geom_store_bytes", "geom_store_tables" };
    if (!name || !value) { g_detail_set("null argument"); return CHV_ERR_INVALID_VALUE; }
    for (int i = 0; i < 5; i++)
        if (!strcmp(name, names[i])) { *value = (unsigned long long)geom_store_counter(i); return CHV_OK; }
    g_detail_set("unknown counter");
    return CHV_ERR_INVALID_VALUE;
}
extern "C" const char *chv_build_flags(void) {
    // what the library was built with: the architecture the Makefile compiled for, the compiler (the hand-scheduled kernels — LDS-DMA through
    // M0, counted vmcnt waits, prefetch registers in flight — were validated against THIS hipcc; tests/test_device_code_contract.py refuses
    // another major version until tools/check_inflight.py and the GPU suite have been re-run with it), every timing-only ablation macro
    static const std::string flags = std::string("arch=" CHV_ARCH ";hipcc=" CHV_HIPCC_VERSION ";clang=") + std::to_string(__clang_major__) + "." + std::to_string(__clang_minor__) +
                                     ";fp_contract=off;" + bgra_wave_build_flags() + ";" + yuv_wave_build_flags() + ";" + bgra_stream_build_flags() + ";" +
                                     yuv_stream_build_flags() + ";" + lanczos_build_flags();
    return flags.c_str();
}

// ---------------------------------------------------------------------------
// errors
// ---------------------------------------------------------------------------
static thread_local std::string g_detail;
static void g_detail_set(const char *msg) { g_detail = msg; }

static int fail(int code, const char *fmt, ...) __attribute__((format(printf, 2, 3)));
static int fail(int code, const char *fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_detail = buf;
    return code;
}

// HIP error -> ComputeError-shaped status (the role of check(), compute.cuda.swift:102-112)
static int hip_fail(hipError_t e, const char *what) {
    int code = CHV_ERR_UNKNOWN;
    switch (e) {
    case hipErrorOutOfMemory: code = CHV_ERR_OUT_OF_MEMORY; break;
    case hipErrorInvalidValue: code = CHV_ERR_INVALID_VALUE; break;
    case hipErrorInvalidDevice: code = CHV_ERR_INVALID_DEVICE; break;
    case hipErrorNoDevice: code = CHV_ERR_DEVICE_NOT_AVAILABLE; break;
    case hipErrorInvalidContext: code = CHV_ERR_INVALID_CONTEXT; break;
    case hipErrorNotSupported: code = CHV_ERR_NOT_IMPLEMENTED; break;
    default: break;
    }
    return fail(code, "%s: %s (%d)", what, hipGetErrorString(e), (int)e);
}
#define HIP_TRY(expr)                                         \
    do {                                                      \
        hipError_t _e = (expr);                               \
        if (_e != hipSuccess) return hip_fail(_e, #expr);     \
    } while (0)

extern "C" const char *chv_error_string(int s) {
    switch (s) {
    case CHV_OK: return "success";
    case CHV_ERR_INVALID_VALUE: return "invalidValue";
    case CHV_ERR_OUT_OF_MEMORY: return "outOfMemory";
    case CHV_ERR_INVALID_CONTEXT: return "invalidContext";
    case CHV_ERR_BAD_TARGET: return "badTarget";
    case CHV_ERR_BAD_INPUT: return "badInputData";
    case CHV_ERR_NOT_IMPLEMENTED: return "notImplemented";
    case CHV_ERR_KERNEL_NOT_FOUND: return "computeKernelNotFound";
    case CHV_ERR_DEVICE_NOT_AVAILABLE: return "deviceNotAvailable";
    case CHV_ERR_INVALID_DEVICE: return "invalidDevice";
    case CHV_ERR_INVALID_OPERATION: return "invalidOperation";
    case CHV_ERR_BAD_CONTEXT_STATE: return "badContextState";
    case CHV_ERR_INVALID_PLATFORM: return "invalidPlatform";
    default: return "unknownError";
    }
}
extern "C" const char *chv_last_error_detail(void) { return g_detail.c_str(); }
extern "C" int chv_version(void) { return CHV_VERSION; }

// ---------------------------------------------------------------------------
// kernel name table (defaultComputeKernelFromString, compute.swift:90-110)
// ---------------------------------------------------------------------------
struct KernelName { const char *name; int id; };
static const KernelName kNames[] = {
    { "img_nv12_nv12", CHV_K_IMG_NV12_NV12 },   { "img_bgra_nv12", CHV_K_IMG_BGRA_NV12 },
    { "img_rgba_nv12", CHV_K_IMG_RGBA_NV12 },   { "img_bgra_bgra", CHV_K_IMG_BGRA_BGRA },
    { "img_y420p_y420p", CHV_K_IMG_Y420P_Y420P }, { "img_y420p_nv12", CHV_K_IMG_Y420P_NV12 },
    { "img_clear_nv12", CHV_K_IMG_CLEAR_NV12 }, { "img_clear_yuvs", CHV_K_IMG_CLEAR_YUVS },
    { "img_clear_bgra", CHV_K_IMG_CLEAR_BGRA }, { "img_clear_rgba", CHV_K_IMG_CLEAR_BGRA },
    { "img_rgba_y420p", CHV_K_IMG_RGBA_Y420P }, { "img_bgra_y420p", CHV_K_IMG_BGRA_Y420P },
    { "img_clear_y420p", CHV_K_IMG_CLEAR_Y420P },
    // names findKernel can synthesise for BGRA canvases (mix.video.swift:142-146)
    { "img_nv12_bgra", CHV_K_IMG_NV12_BGRA },   { "img_y420p_bgra", CHV_K_IMG_Y420P_BGRA },
    { "img_bgra_bgra_tx", CHV_K_IMG_BGRA_BGRA_TX }, { "img_rgba_bgra_tx", CHV_K_IMG_RGBA_BGRA_TX },
    // an RGBA layer on a BGRA canvas: no reference backend has a kernel of that name; it resolves to the transform-aware
    // one (as "img_clear_rgba" resolves to img_clear_bgra, compute.swift:101)
    { "img_rgba_bgra", CHV_K_IMG_RGBA_BGRA_TX },
    { "img_bgra_nv12_int", CHV_K_IMG_BGRA_NV12_INT }, { "img_rgba_nv12_int", CHV_K_IMG_RGBA_NV12_INT },
    { "img_bgra_y420p_int", CHV_K_IMG_BGRA_Y420P_INT }, { "img_rgba_y420p_int", CHV_K_IMG_RGBA_Y420P_INT },
    // the two cases of the enum the reference's table leaves out (compute.swift:91-105; nothing dispatches them there): resolvable here, so that
    // every case that has a kernel can be asked for by its name
    { "snd_s16i_s16i", CHV_K_SND_S16I_S16I }, { "me_fullsearch", CHV_K_ME_FULLSEARCH },
};

extern "C" int chv_kernel_from_string(const char *name, int *kernel) {
    if (!name || !kernel) return fail(CHV_ERR_INVALID_VALUE, "null argument");
    for (const auto &k : kNames)
        if (!strcmp(k.name, name)) { *kernel = k.id; return CHV_OK; }
    return fail(CHV_ERR_INVALID_VALUE, "no default kernel named '%s'", name);
}

extern "C" const char *chv_kernel_name(int kernel) {
    switch (kernel) {
    case CHV_K_IMG_NV12_NV12: return "img_nv12_nv12";
    case CHV_K_IMG_BGRA_NV12: return "img_bgra_nv12";
    case CHV_K_IMG_RGBA_NV12: return "img_rgba_nv12";
    case CHV_K_IMG_BGRA_BGRA: return "img_bgra_bgra";
    case CHV_K_IMG_Y420P_Y420P: return "img_y420p_y420p";
    case CHV_K_IMG_Y420P_NV12: return "img_y420p_nv12";
    case CHV_K_IMG_CLEAR_NV12: return "img_clear_nv12";
    case CHV_K_IMG_CLEAR_YUVS: return "img_clear_yuvs";
    case CHV_K_IMG_CLEAR_BGRA: return "img_clear_bgra";
    case CHV_K_IMG_CLEAR_Y420P: return "img_clear_y420p";
    case CHV_K_IMG_CLEAR_RGBA: return "img_clear_rgba";
    case CHV_K_IMG_RGBA_Y420P: return "img_rgba_y420p";
    case CHV_K_IMG_BGRA_Y420P: return "img_bgra_y420p";
    case CHV_K_SND_S16I_S16I: return "snd_s16i_s16i";
    case CHV_K_ME_FULLSEARCH: return "me_fullsearch";
    case CHV_K_IMG_NV12_BGRA: return "img_nv12_bgra";
    case CHV_K_IMG_Y420P_BGRA: return "img_y420p_bgra";
    case CHV_K_IMG_BGRA_BGRA_TX: return "img_bgra_bgra_tx";
    case CHV_K_IMG_RGBA_BGRA_TX: return "img_rgba_bgra_tx";
    case CHV_K_IMG_BGRA_NV12_INT: return "img_bgra_nv12_int";
    case CHV_K_IMG_RGBA_NV12_INT: return "img_rgba_nv12_int";
    case CHV_K_IMG_BGRA_Y420P_INT: return "img_bgra_y420p_int";
    case CHV_K_IMG_RGBA_Y420P_INT: return "img_rgba_y420p_int";
    default: return NULL;
    }
}

// ---------------------------------------------------------------------------
// objects
// ---------------------------------------------------------------------------
// Device-side Lanczos tables of one (in, out) size pair.  Shared ownership: the cache holds one reference, every call that
// looked the pair up holds another until its launch is enqueued — an eviction by another context of the device can therefore
// never free tables between a lookup and the launch that reads them.  The memory goes back with hipFree, which waits for the
// work already queued on the device (i.e. for the launches that read the tables).
struct LanczosTable {
    int device = 0;
    int taps = 0;
    int32_t *first = nullptr;  // device
    float *weights = nullptr;  // device
    int stride = 0;            // first[o + 1] - first[o] when that is the same for every o (2 for the 2:1 class), else 0
    int first0 = 0;            // first[0]
    LanczosTable() = default;
    LanczosTable(const LanczosTable &) = delete;
    LanczosTable &operator=(const LanczosTable &) = delete;
    ~LanczosTable() {
        if (first || weights) { (void)hipSetDevice(device); (void)hipFree(first); (void)hipFree(weights); }
    }
};
typedef std::shared_ptr<LanczosTable> LanczosRef;
struct LanczosEntry {
    LanczosRef tab;
    uint64_t last_use = 0;     // LRU stamp
};
static constexpr size_t kLanczosCacheEntries = 64;   // (in, out) size pairs kept per device; least recently used goes first
static constexpr size_t kLanczosRetireBatch = 32;    // evicted tables are freed this many at a time (hipFree synchronises the device)

// State shared by a context and everything created with chv_context_share
// (the role of InternalContext, compute.cl.swift:60-74).
struct DeviceShared {
    int device = 0;
    std::mutex mu;
    std::map<std::pair<int, int>, LanczosEntry> lanczos;  // (in, out) -> tables
    std::vector<LanczosRef> lanczos_retired;               // evicted, not yet freed (see lanczos_table)
    uint64_t lanczos_clock = 0;
    ~DeviceShared() { (void)hipSetDevice(device); }         // (the tables free themselves)
};

// A run-time compiled kernel (`ComputeKernel.custom`): the module lives as long as any context's library names it.
struct CustomKernel {
    int device = 0;
    hipModule_t module = nullptr;
    hipFunction_t fn = nullptr;
    ~CustomKernel() {
        if (module) { (void)hipSetDevice(device); (void)hipModuleUnload(module); }
    }
};

struct StagingSlot {
    void *host = nullptr;
    size_t cap = 0;
    hipEvent_t done = nullptr;
    bool pending = false;
};

// The descriptor ring is walked slot by slot, and its slots are guarded in GROUPS of kDescGroup: one event after the launch that used a
// group's last slot stands for every launch of the group (a stream runs in order), and it is waited for when the walk enters the group
// again, kDescSlots launches later.  An event per launch — what this was until round 3 — put a marker between every two kernels of a
// tick issued layer by layer: an unchanged mix.video.swift tick (clear + 4 launches + wait) took 77-84 us with them and 55 without.
struct DescGroup {
    hipEvent_t done = nullptr;
    bool pending = false;     // `done` has been recorded and not yet waited for
    bool unguarded = false;   // the event behind the group's last slot could not be recorded: the walk drains the stream before it enters the group again
};

static constexpr int kStagingSlots = 4;
static constexpr int kDescSlots = 64, kDescGroup = 8;
static constexpr size_t kDescSlotBytes = sizeof(DTick) + CHV_MAX_LAYERS * sizeof(DLayer);

struct chv_context {
    uint32_t magic = 0x43485643;  // 'CHVC'
    int device = 0;
    hipStream_t stream = nullptr;
    std::shared_ptr<DeviceShared> shared;
    int pass_depth = 0;                // open chv_pass_begin brackets (an upload helper that brackets its copies inside the mixer's pass nests)
    StagingSlot staging[kStagingSlots];
    int next_staging = 0;
    // descriptor ring in pinned, device-mapped host memory
    uint8_t *desc_host = nullptr;
    uint8_t *desc_dev = nullptr;       // the same ring in device memory: a transient launch's descriptors are copied there on the stream
    DescGroup desc_group[kDescSlots / kDescGroup];
    int next_desc = 0;
    // `library` of the reference's ComputeContext (compute.cl.swift:66-73): name -> built kernel
    std::map<std::string, std::shared_ptr<CustomKernel>> library;
    // The picture kernels of the pass in progress that have been accepted but not launched yet (PendingPass below): inside
    // chv_pass_begin ... chv_pass_end nothing has to be visible before the pass ends, so `clear + N layer kernels on one target` — what an
    // unchanged VideoMixer issues per tick, mix.video.swift:116-124 — leaves as the ONE launch chv_composite would have made of it.
    struct PendingPass {
        bool active = false;
        chv_image target;
        int clear_first = 0;
        int clear_tf = -1;                     // the target format a clear kernel fixed (kernel_shape), -1: the layers decide
        std::vector<chv_layer> layers;         // POD copies: the caller's structs need not outlive chv_run_kernel
        std::vector<chv_buffer *> pins;        // every buffer the pending kernels name, held against chv_buffer_free until they are launched
    } pending;
};

struct chv_buffer {
    uint32_t magic = 0x43485642;  // 'CHVB'
    void *ptr = nullptr;
    size_t size = 0;
    int device = 0;
    bool owned = true;
    // cross-context ordering: an asynchronous upload records `ready` on the uploading
    // context's stream; a kernel launched from another context's stream waits on it first
    // (the reference gets this ordering from blocking copies, compute.cl.swift:440-450)
    // (guarded by `mu`: the uploading thread and the launching thread are different threads in the reference's
    // pipeline — a Bus runner and the mixer's queue)
    std::mutex mu;
    hipEvent_t ready = nullptr;
    std::atomic<hipStream_t> ready_stream{nullptr};      // (atomic: launches look without the lock first — most buffers have nothing pending)
    uint64_t ready_seq = 0;
    // Deferred passes (chv_context::PendingPass): kernels that were accepted but not launched yet hold their buffers.  chv_buffer_free on a held
    // buffer marks it `doomed` and returns; the release that drops the last hold frees it (OpenCL keeps a cl_mem alive the same way from
    // clSetKernelArg / enqueue to completion — a ComputeBuffer's deinit may run the moment runComputeKernel returns, compute.cl.swift:55-57).
    // Both fields are guarded by g_pin_mu.
    int pins = 0;
    bool doomed = false;
};
static std::mutex g_pin_mu;
// (passes and kernels, below) launch what the pass in progress has accepted so far; every entry point that touches the context's stream starts with it
static int flush_pending(chv_context *c);
#define FLUSH_PENDING(c) do { if ((c)->pending.active) { int frc_ = flush_pending(c); if (frc_) return frc_; } } while (0)

// after an asynchronous copy into `b` was enqueued on `stream`: record the buffer's event
static int mark_uploaded(chv_buffer *b, hipStream_t stream);
// before `stream` reads or overwrites `b`: wait (on the stream, not on the host) for a pending upload from another stream
static int wait_for_buffer(chv_buffer *b, hipStream_t stream);

struct chv_event {
    hipEvent_t ev = nullptr;
    int device = 0;
};

struct BatchDep {
    chv_buffer *buf;
    hipStream_t stream;   // the stream that last waited for this buffer on behalf of the batch
    uint64_t seq;         // ... and the upload it saw
};

// ---- descriptor blocks of batches -----------------------------------------------------------------------------------------------------------
// A host that groups its mixers builds a batch for every group tick, runs it once and frees it (tools/batch_create_probe.cpp): two hipMalloc, two
// synchronous copies and two hipFree (which wait for the device) per tick were 45 of the 105 us a group of eight ticks took.  A batch's
// descriptors — ticks | layers | the second launch's ticks — now live in ONE block of a per-device pool: device memory with a pinned host twin
// and an event.  Creation fills the twin and sends it with one asynchronous copy on the creating context's stream; every run records the
// block's event behind its launches (a run on another stream first makes that stream wait for it); a block taken from the pool again waits
// for its event on the host before the twin is overwritten (a host that waited for its tick never blocks there).  Blocks are never returned
// to the system below kDescPoolBytes per device; beyond it, and when pinned memory runs out, a batch allocates and frees as before.
struct DescBlock {
    uint8_t *dev = nullptr, *host = nullptr;
    size_t cap = 0;
    hipEvent_t ev = nullptr;
    bool recorded = false;         // `ev` has been recorded at least once since the block was (re)taken
    hipStream_t last = nullptr;    // the stream it was recorded on last
};
constexpr size_t kDescPoolBytes = (size_t)64 << 20;
struct DescPool { std::mutex mu; std::vector<DescBlock> free_blocks; size_t bytes = 0; };
// (never destroyed: freeing device memory from a static destructor would call into a runtime that may be gone already)
static DescPool &desc_pool(int device) { static DescPool *pools = new DescPool[16]; return pools[device & 15]; }
// (the device is current)
static bool desc_block_acquire(int device, size_t need, DescBlock *out) {
    DescPool &P = desc_pool(device);
    {
        std::lock_guard<std::mutex> lk(P.mu);
        size_t best = P.free_blocks.size();
        for (size_t i = 0; i < P.free_blocks.size(); i++)
            if (P.free_blocks[i].cap >= need && (best == P.free_blocks.size() || P.free_blocks[i].cap < P.free_blocks[best].cap)) best = i;
        if (best != P.free_blocks.size()) {
            *out = P.free_blocks[best];
            P.free_blocks.erase(P.free_blocks.begin() + (long)best);
            return true;
        }
        size_t cap = (size_t)64 << 10;
        while (cap < need) cap <<= 1;
        if (P.bytes + cap > kDescPoolBytes) return false;
        P.bytes += cap;
        need = cap;
    }
    DescBlock b;
    b.cap = need;
    hipError_t e = hipMalloc((void **)&b.dev, b.cap);
    if (e == hipSuccess) e = hipHostMalloc((void **)&b.host, b.cap, hipHostMallocDefault);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&b.ev, hipEventDisableTiming);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        if (b.dev) (void)hipFree(b.dev);
        if (b.host) (void)hipHostFree(b.host);
        std::lock_guard<std::mutex> lk(P.mu);
        P.bytes -= b.cap;
        return false;
    }
    *out = b;
    return true;
}
static void desc_block_release(int device, DescBlock &b) {
    if (!b.dev) return;
    DescPool &P = desc_pool(device);
    std::lock_guard<std::mutex> lk(P.mu);
    P.free_blocks.push_back(b);
    b = DescBlock{};
}

struct chv_batch {
    DescBlock blk;                 // pooled descriptors (d_ticks / d_layers / d_ticks2 point into blk.dev); blk.dev == nullptr: allocations of its own
    int device = 0;
    int n_ticks = 0, n_layers = 0;
    int target_format = 0;
    int maxW = 0, maxH = 0;
    int fast_path = -1;
    // a second launch that continues on the canvases (split_stream_prefix: the leading video layers through the streaming kernel, then the rest)
    int fast_path2 = -2;           // -2: none
    DTick *d_ticks2 = nullptr;
    std::vector<DTick> h_ticks2;
    DTick *d_ticks = nullptr;
    DLayer *d_layers = nullptr;
    std::vector<BatchDep> deps;    // distinct buffers the descriptors point into + the last upload waited for, per stream
    std::vector<DTick> h_ticks;    // host copies: launch geometry of the fast paths
    std::vector<DLayer> h_layers;
    std::string kernel_name;
    GeomCache geom;                // the strip kernels' per-layer geometry, computed once per launch configuration (geom_cache.h)
};

static bool ctx_ok(chv_context *c) { return c && c->magic == 0x43485643 && c->stream; }
static bool buf_ok(const chv_buffer *b) { return b && b->magic == 0x43485642 && b->ptr; }

// ---------------------------------------------------------------------------
// devices
// ---------------------------------------------------------------------------
extern "C" int chv_device_count(int *count) {
    if (!count) return fail(CHV_ERR_INVALID_VALUE, "null argument");
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) { *count = 0; return hip_fail(e, "hipGetDeviceCount"); }
    *count = n;
    return CHV_OK;
}

extern "C" int chv_device_info_get(int device, chv_device_info *info) {
    if (!info) return fail(CHV_ERR_INVALID_VALUE, "null argument");
    int n = 0;
    int rc = chv_device_count(&n);
    if (rc) return rc;
    if (device < 0 || device >= n) return fail(CHV_ERR_INVALID_DEVICE, "device %d of %d", device, n);
    hipDeviceProp_t p;
    HIP_TRY(hipGetDeviceProperties(&p, device));
    memset(info, 0, sizeof *info);
    info->index = device;
    info->device_type = 0;
    info->vendor_id = 0x1002;
    info->compute_units = p.multiProcessorCount;
    info->supports_images = 0;
    info->total_memory = p.totalGlobalMem;
    snprintf(info->name, sizeof info->name, "%s", p.name);
    snprintf(info->arch, sizeof info->arch, "%s", p.gcnArchName);
    // kernels in this library are built for gfx950 only
    info->available = strncmp(p.gcnArchName, "gfx950", 6) == 0;
    return CHV_OK;
}

// ---------------------------------------------------------------------------
// contexts
// ---------------------------------------------------------------------------
static int context_new(int device, std::shared_ptr<DeviceShared> shared, chv_context **out) {
    HIP_TRY(hipSetDevice(device));
    std::unique_ptr<chv_context> c(new chv_context);
    c->device = device;
    c->shared = std::move(shared);
    HIP_TRY(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
    hipError_t e = hipHostMalloc((void **)&c->desc_host, kDescSlots * kDescSlotBytes, hipHostMallocMapped);
    if (e != hipSuccess) { (void)hipStreamDestroy(c->stream); return hip_fail(e, "hipHostMalloc(descriptors)"); }
    e = hipMalloc((void **)&c->desc_dev, kDescSlots * kDescSlotBytes);
    if (e != hipSuccess) { (void)hipHostFree(c->desc_host); (void)hipStreamDestroy(c->stream); return hip_fail(e, "hipMalloc(descriptors)"); }
    *out = c.release();
    return CHV_OK;
}

extern "C" int chv_context_create(int device, chv_context **out) {
    if (!out) return fail(CHV_ERR_INVALID_VALUE, "null argument");
    *out = nullptr;
    chv_device_info info;
    int rc = chv_device_info_get(device, &info);
    if (rc) return rc;
    if (!info.available)
        return fail(CHV_ERR_DEVICE_NOT_AVAILABLE, "device %d is %s; this library carries gfx950 code only", device, info.arch);
    auto shared = std::make_shared<DeviceShared>();
    shared->device = device;
    return context_new(device, shared, out);
}

extern "C" int chv_context_share(chv_context *parent, chv_context **out) {
    if (!out) return fail(CHV_ERR_INVALID_VALUE, "null argument");
    *out = nullptr;
    if (!ctx_ok(parent)) return fail(CHV_ERR_INVALID_CONTEXT, "bad parent context");
    int rc = context_new(parent->device, parent->shared, out);
    if (rc == CHV_OK) (*out)->library = parent->library;      // ComputeContext(other:) copies the library, compute.cl.swift:82-87
    return rc;
}

extern "C" int chv_context_destroy(chv_context *c) {
    if (!ctx_ok(c)) return fail(CHV_ERR_INVALID_CONTEXT, "bad context");
    (void)hipSetDevice(c->device);
    (void)flush_pending(c);             // (a pass left open: its kernels were accepted, they run)
    (void)hipStreamSynchronize(c->stream);
    for (auto &s : c->staging) {
        if (s.done) (void)hipEventDestroy(s.done);
        if (s.host) (void)hipHostFree(s.host);
    }
    for (auto &d : c->desc_group) if (d.done) (void)hipEventDestroy(d.done);
    if (c->desc_host) (void)hipHostFree(c->desc_host);
    if (c->desc_dev) (void)hipFree(c->desc_dev);
    (void)hipStreamDestroy(c->stream);
    c->magic = 0;
    c->stream = nullptr;
    delete c;
    return CHV_OK;
}

extern "C" int chv_context_device(chv_context *c, int *device) {
    if (!ctx_ok(c) || !device) return fail(CHV_ERR_INVALID_CONTEXT, "bad context");
    *device = c->device;
    return CHV_OK;
}
// NUMA node of the device's PCIe root (sysfs), -1 when the platform does not say: a host that pins its upload ring on that
// node (first touch from a thread bound there) keeps H2D copies off the inter-socket link
extern "C" int chv_context_numa_node(chv_context *c, int *node) {
    if (!ctx_ok(c) || !node) return fail(CHV_ERR_INVALID_CONTEXT, "bad context");
    *node = -1;
    char bus[64] = { 0 };
    if (hipDeviceGetPCIBusId(bus, (int)sizeof bus, c->device) != hipSuccess) { (void)hipGetLastError(); return CHV_OK; }
    for (char *p = bus; *p; p++) *p = (char)tolower((unsigned char)*p);
    const std::string path = std::string("/sys/bus/pci/devices/") + bus + "/numa_node";
    if (FILE *f = fopen(path.c_str(), "r")) {
        int n = -1;
        if (fscanf(f, "%d", &n) == 1) *node = n;
        fclose(f);
    }
    return CHV_OK;
}
extern "C" int chv_context_stream(chv_context *c, void **s) {
    if (!ctx_ok(c) || !s) return fail(CHV_ERR_INVALID_CONTEXT, "bad context");
    FLUSH_PENDING(c);       // (whoever asks for the stream is about to put work behind what the pass holds)
    *s = (void *)c->stream;
    return CHV_OK;
}

// ---------------------------------------------------------------------------
// buffers and transfers
// ---------------------------------------------------------------------------
extern "C" int chv_buffer_alloc(chv_context *c, size_t bytes, chv_buffer **out) {
    if (!out) return fail(CHV_ERR_INVALID_VALUE, "null argument");
    *out = nullptr;
    if (!ctx_ok(c)) return fail(CHV_ERR_INVALID_CONTEXT, "bad context");
    if (bytes == 0) return fail(CHV_ERR_INVALID_VALUE, "zero-sized buffer");
    HIP_TRY(hipSetDevice(c->device));
    void *p = nullptr;
    HIP_TRY(hipMalloc(&p, bytes));
    chv_buffer *b = new chv_buffer;
    b->ptr = p; b->size = bytes; b->device = c->device; b->owned = true;
    *out = b;
    return CHV_OK;
}

extern "C" int chv_buffer_wrap(chv_context *c, void *device_ptr, size_t bytes, chv_buffer **out) {
    if (!out) return fail(CHV_ERR_INVALID_VALUE, "null argument");
    *out = nullptr;
    if (!ctx_ok(c)) return fail(CHV_ERR_INVALID_CONTEXT, "bad context");
    if (!device_ptr || bytes == 0) return fail(CHV_ERR_INVALID_VALUE, "null/empty device memory");
    chv_buffer *b = new chv_buffer;
    b->ptr = device_ptr; b->size = bytes; b->device = c->device; b->owned = false;
    *out = b;
    return CHV_OK;
}

static int buffer_free_now(chv_buffer *b);
extern "C" int chv_buffer_free(chv_buffer *b) {
    if (!b || b->magic != 0x43485642) return fail(CHV_ERR_INVALID_VALUE, "bad buffer");
    {
        std::lock_guard<std::mutex> lock(g_pin_mu);
        if (b->doomed) return fail(CHV_ERR_INVALID_VALUE, "buffer freed twice");
        if (b->pins > 0) { b->doomed = true; return CHV_OK; }      // a pending pass still names it: freed when that pass has been launched
    }
    return buffer_free_now(b);
}
static int buffer_free_now(chv_buffer *b) {
    if (b->owned && b->ptr) {
        // any thread, any time: make the owning device current first
        // (compute.cuda.swift:82-88 pushes the context in deinit for the same reason)
        (void)hipSetDevice(b->device);
        (void)hipFree(b->ptr);  // hipFree waits for in-flight work on the allocation
    }
    if (b->ready) { (void)hipSetDevice(b->device); (void)hipEventDestroy(b->ready); b->ready = nullptr; }
    b->magic = 0;
    b->ptr = nullptr;
    delete b;
    return CHV_OK;
}

extern "C" int chv_buffer_info(chv_buffer *b, void **device_ptr, size_t *bytes) {
    if (!buf_ok(b)) return fail(CHV_ERR_INVALID_VALUE, "bad buffer");
    if (device_ptr) *device_ptr = b->ptr;
    if (bytes) *bytes = b->size;
    return CHV_OK;
}

extern "C" int chv_plane_alloc(chv_context *c, int width, int height, int components,
                               chv_buffer **out, size_t *pitch) {
    if (!out || !pitch) return fail(CHV_ERR_INVALID_VALUE, "null argument");
    if (width <= 0 || height <= 0) return fail(CHV_ERR_INVALID_OPERATION, "plane size %dx%d", width, height);
    if (components != 1 && components != 2 && components != 4)
        return fail(CHV_ERR_BAD_INPUT, "components must be 1, 2 or 4, got %d", components);
    // rows start on a 128-byte cache line; widths that already are a multiple of 128 bytes stay packed,
    // so a packed host plane of the same stride uploads as one linear copy
    size_t row = (size_t)width * components;
    size_t p = (row + 127) & ~(size_t)127;
    *pitch = p;
    return chv_buffer_alloc(c, p * (size_t)height, out);
}

static int check_span(const chv_buffer *b, size_t offset, size_t pitch, size_t width_bytes, size_t rows,
                      const char *what) {
    if (!buf_ok(b)) return fail(CHV_ERR_BAD_INPUT, "%s: bad buffer", what);
    if (rows == 0 || width_bytes == 0) return fail(CHV_ERR_INVALID_VALUE, "%s: empty region", what);
    if (pitch < width_bytes) return fail(CHV_ERR_BAD_INPUT, "%s: pitch %zu < row bytes %zu", what, pitch, width_bytes);
    size_t end = 0, total = 0;
    if (__builtin_mul_overflow(rows - 1, pitch, &end) || __builtin_add_overflow(end, width_bytes, &end) ||
        __builtin_add_overflow(end, offset, &end) || __builtin_mul_overflow(rows, width_bytes, &total))
        return fail(CHV_ERR_BAD_INPUT, "%s: region size overflows", what);
    if (end > b->size) return fail(CHV_ERR_BAD_INPUT, "%s: region ends at %zu, buffer has %zu bytes", what, end, b->size);
    return CHV_OK;
}

extern "C" int chv_upload(chv_context *c, chv_buffer *dst, size_t dst_offset, size_t dst_pitch,
                          const void *src, size_t src_pitch, size_t width_bytes, size_t rows, int async) {
    if (!ctx_ok(c)) return fail(CHV_ERR_INVALID_CONTEXT, "bad context");
    FLUSH_PENDING(c);
    if (!src) return fail(CHV_ERR_BAD_INPUT, "null source");
    int rc = check_span(dst, dst_offset, dst_pitch, width_bytes, rows, "upload");
    if (rc) return rc;
    if (src_pitch < width_bytes) return fail(CHV_ERR_BAD_INPUT, "source pitch %zu < row bytes %zu", src_pitch, width_bytes);
    HIP_TRY(hipSetDevice(c->device));
    uint8_t *d = (uint8_t *)dst->ptr + dst_offset;
    auto copy_h2d = [&](const void *from, size_t from_pitch) -> hipError_t {
        // one contiguous block on both sides -> a linear copy
        if (dst_pitch == width_bytes && from_pitch == width_bytes) return hipMemcpyAsync(d, from, width_bytes * rows, hipMemcpyHostToDevice, c->stream);
        return hipMemcpy2DAsync(d, dst_pitch, from, from_pitch, width_bytes, rows, hipMemcpyHostToDevice, c->stream);
    };
    // an earlier asynchronous upload into the same buffer from another context must land first (stream order
    // only covers this context's own copies)
    rc = wait_for_buffer(dst, c->stream);
    if (rc) return rc;
    if (!async) {
        HIP_TRY(copy_h2d(src, src_pitch));
        HIP_TRY(hipStreamSynchronize(c->stream));
        std::lock_guard<std::mutex> lock(dst->mu);
        dst->ready_stream = nullptr;   // complete: nobody has to wait
        return CHV_OK;
    }
    if (async == 2) {
        // caller-owned pinned memory (chv_host_alloc): no staging copy
        HIP_TRY(copy_h2d(src, src_pitch));
        return mark_uploaded(dst, c->stream);
    }
    // stage into pinned memory so the caller's bytes are only borrowed for this call
    StagingSlot &s = c->staging[c->next_staging];
    c->next_staging = (c->next_staging + 1) % kStagingSlots;
    if (s.pending) { HIP_TRY(hipEventSynchronize(s.done)); s.pending = false; }
    size_t need = width_bytes * rows;
    if (s.cap < need) {
        if (s.host) { HIP_TRY(hipHostFree(s.host)); s.host = nullptr; s.cap = 0; }
        size_t cap = (need + (1u << 20) - 1) & ~(size_t)((1u << 20) - 1);   // need = rows * width_bytes, overflow-checked by check_span
        HIP_TRY(hipHostMalloc(&s.host, cap, hipHostMallocDefault));
        s.cap = cap;
    }
    if (!s.done) HIP_TRY(hipEventCreateWithFlags(&s.done, hipEventDisableTiming));
    const uint8_t *sp = (const uint8_t *)src;
    uint8_t *hp = (uint8_t *)s.host;
    if (src_pitch == width_bytes) memcpy(hp, sp, need);
    else for (size_t r = 0; r < rows; r++) memcpy(hp + r * width_bytes, sp + r * src_pitch, width_bytes);
    HIP_TRY(copy_h2d(hp, width_bytes));
    HIP_TRY(hipEventRecord(s.done, c->stream));
    s.pending = true;
    return mark_uploaded(dst, c->stream);
}

static int mark_uploaded(chv_buffer *b, hipStream_t stream) {
    std::lock_guard<std::mutex> lock(b->mu);
    if (!b->ready) HIP_TRY(hipEventCreateWithFlags(&b->ready, hipEventDisableTiming));
    HIP_TRY(hipEventRecord(b->ready, stream));
    b->ready_stream = stream;
    b->ready_seq++;
    return CHV_OK;
}

static int wait_for_buffer(chv_buffer *b, hipStream_t stream) {
    std::lock_guard<std::mutex> lock(b->mu);
    if (!b->ready_stream) return CHV_OK;
    if (hipEventQuery(b->ready) == hipSuccess) { b->ready_stream = nullptr; return CHV_OK; }   // landed: nobody has to wait any more
    (void)hipGetLastError();                                                                    // hipErrorNotReady is not an error here
    if (b->ready_stream != stream) HIP_TRY(hipStreamWaitEvent(stream, b->ready, 0));
    return CHV_OK;
}

extern "C" int chv_host_alloc(chv_context *c, size_t bytes, void **out) {
    if (!out) return fail(CHV_ERR_INVALID_VALUE, "null argument");
    *out = nullptr;
    if (!ctx_ok(c)) return fail(CHV_ERR_INVALID_CONTEXT, "bad context");
    if (bytes == 0) return fail(CHV_ERR_INVALID_VALUE, "zero-sized allocation");
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipHostMalloc(out, bytes, hipHostMallocDefault));
    return CHV_OK;
}

extern "C" int chv_host_free(chv_context *c, void *ptr) {
    if (!ctx_ok(c)) return fail(CHV_ERR_INVALID_CONTEXT, "bad context");
    if (!ptr) return fail(CHV_ERR_INVALID_VALUE, "null pointer");
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipHostFree(ptr));
    return CHV_OK;
}

static int download_enqueue(chv_context *c, void *dst, size_t dst_pitch, chv_buffer *src,
                            size_t src_offset, size_t src_pitch, size_t width_bytes, size_t rows) {
    if (!ctx_ok(c)) return fail(CHV_ERR_INVALID_CONTEXT, "bad context");
    FLUSH_PENDING(c);
    if (!dst) return fail(CHV_ERR_BAD_INPUT, "null destination");
    int rc = check_span(src, src_offset, src_pitch, width_bytes, rows, "download");
    if (rc) return rc;
    if (dst_pitch < width_bytes) return fail(CHV_ERR_BAD_INPUT, "destination pitch %zu < row bytes %zu", dst_pitch, width_bytes);
    HIP_TRY(hipSetDevice(c->device));
    // a picture uploaded asynchronously on another context (GPUBarrierUpload's) and read back through this one
    // (GPUBarrierDownload's, compute.swift:175-255): order the read behind the upload
    rc = wait_for_buffer(src, c->stream);
    if (rc) return rc;
    const uint8_t *from = (const uint8_t *)src->ptr + src_offset;
    // one contiguous block on both sides -> a linear copy (a slab of frames: the link gives its full rate to copies of 8 MB and more)
    if (dst_pitch == width_bytes && src_pitch == width_bytes) HIP_TRY(hipMemcpyAsync(dst, from, width_bytes * rows, hipMemcpyDeviceToHost, c->stream));
    else HIP_TRY(hipMemcpy2DAsync(dst, dst_pitch, from, src_pitch, width_bytes, rows, hipMemcpyDeviceToHost, c->stream));
    return CHV_OK;
}

extern "C" int chv_download(chv_context *c, void *dst, size_t dst_pitch, chv_buffer *src,
                            size_t src_offset, size_t src_pitch, size_t width_bytes, size_t rows) {
    int rc = download_enqueue(c, dst, dst_pitch, src, src_offset, src_pitch, width_bytes, rows);
    if (rc) return rc;
    HIP_TRY(hipStreamSynchronize(c->stream));
    return CHV_OK;
}

// The copy is ordered on the context's stream and the call returns at once: `dst` is pinned memory (chv_host_alloc) whose bytes are the
// picture's once the stream has passed the copy — chv_event_record + chv_event_synchronize, or chv_pass_end(wait) on this context.  Work of
// OTHER contexts that writes `src` (the mixer's tick) is ordered in front of it with chv_event_record there and chv_event_wait here, as
// between any two contexts: a download barrier running on a context of its own (GPUBarrierDownload, compute.swift:217-255) then overlaps the
// D2H copy of tick t with the kernels of tick t + 1.
// `dst` must be pinned (chv_host_alloc, or memory the caller registered with the runtime): into pageable memory the runtime stages the copy and
// blocks the host, and "valid once the stream has passed the copy" holds only loosely — such a destination gets the synchronous path instead
// (the bytes are there when the call returns; nothing the caller does afterwards can be wrong).
static bool host_range_is_pinned(const void *p, size_t bytes) {
    if (!p || bytes == 0) return false;
    for (const uint8_t *q : { (const uint8_t *)p, (const uint8_t *)p + bytes - 1 }) {
        hipPointerAttribute_t a;
        memset(&a, 0, sizeof a);
        if (hipPointerGetAttributes(&a, q) != hipSuccess) { (void)hipGetLastError(); return false; }
        if (a.type != hipMemoryTypeHost) return false;
    }
    return true;
}
extern "C" int chv_download_async(chv_context *c, void *dst, size_t dst_pitch, chv_buffer *src,
                                  size_t src_offset, size_t src_pitch, size_t width_bytes, size_t rows) {
    int rc = download_enqueue(c, dst, dst_pitch, src, src_offset, src_pitch, width_bytes, rows);
    if (rc) return rc;
    if (rows && width_bytes && !host_range_is_pinned(dst, (rows - 1) * dst_pitch + width_bytes)) HIP_TRY(hipStreamSynchronize(c->stream));
    return CHV_OK;
}

// ---------------------------------------------------------------------------
// descriptor building
// ---------------------------------------------------------------------------
struct KernelShape {
    int src_planes;      // 0 for clear
    int src_comps[3];
    int target_format;   // TargetFormat
    int kind;            // LayerKind
    int swizzle;
    bool is_clear;
};

static int kernel_shape(int kernel, KernelShape *s) {
    memset(s, 0, sizeof *s);
    auto yuv_nv12 = [&](int tf, int kind) { s->src_planes = 2; s->src_comps[0] = 1; s->src_comps[1] = 2; s->target_format = tf; s->kind = kind; };
    auto yuv_420p = [&](int tf, int kind) { s->src_planes = 3; s->src_comps[0] = s->src_comps[1] = s->src_comps[2] = 1; s->target_format = tf; s->kind = kind; };
    auto rgb = [&](int tf, int kind, int swz) { s->src_planes = 1; s->src_comps[0] = 4; s->target_format = tf; s->kind = kind; s->swizzle = swz; };
    switch (kernel) {
    case CHV_K_IMG_NV12_NV12: yuv_nv12(TF_NV12, LK_YUV_FROM_NV12); break;
    case CHV_K_IMG_Y420P_NV12: yuv_420p(TF_NV12, LK_YUV_FROM_Y420P); break;
    case CHV_K_IMG_Y420P_Y420P: yuv_420p(TF_Y420P, LK_YUV_FROM_Y420P); break;
    case CHV_K_IMG_BGRA_NV12: rgb(TF_NV12, LK_YUV_FROM_RGB, 1); break;   // .zyxw, kernels.cl.swift:518
    case CHV_K_IMG_RGBA_NV12: rgb(TF_NV12, LK_YUV_FROM_RGB, 0); break;
    case CHV_K_IMG_BGRA_Y420P: rgb(TF_Y420P, LK_YUV_FROM_RGB, 1); break;
    case CHV_K_IMG_RGBA_Y420P: rgb(TF_Y420P, LK_YUV_FROM_RGB, 0); break;
    case CHV_K_IMG_BGRA_BGRA: rgb(TF_BGRA, LK_BGRA_METAL, 0); break;
    case CHV_K_IMG_NV12_BGRA: yuv_nv12(TF_BGRA, LK_BGRA_FROM_NV12); break;
    case CHV_K_IMG_Y420P_BGRA: yuv_420p(TF_BGRA, LK_BGRA_FROM_Y420P); break;
    case CHV_K_IMG_BGRA_BGRA_TX: rgb(TF_BGRA, LK_BGRA_FROM_RGB, 0); break;
    case CHV_K_IMG_RGBA_BGRA_TX: rgb(TF_BGRA, LK_BGRA_FROM_RGB, 1); break;
    case CHV_K_IMG_BGRA_NV12_INT: rgb(TF_NV12, LK_YUV_FROM_RGB_INT, 1); break;
    case CHV_K_IMG_RGBA_NV12_INT: rgb(TF_NV12, LK_YUV_FROM_RGB_INT, 0); break;
    case CHV_K_IMG_BGRA_Y420P_INT: rgb(TF_Y420P, LK_YUV_FROM_RGB_INT, 1); break;
    case CHV_K_IMG_RGBA_Y420P_INT: rgb(TF_Y420P, LK_YUV_FROM_RGB_INT, 0); break;
    case CHV_K_IMG_CLEAR_NV12: s->is_clear = true; s->target_format = TF_NV12; break;
    case CHV_K_IMG_CLEAR_Y420P: s->is_clear = true; s->target_format = TF_Y420P; break;
    case CHV_K_IMG_CLEAR_BGRA: case CHV_K_IMG_CLEAR_RGBA: s->is_clear = true; s->target_format = TF_BGRA; break;
    case CHV_K_IMG_CLEAR_YUVS:
        // an enum case for which no backend of the reference has a kernel either
        return fail(CHV_ERR_KERNEL_NOT_FOUND, "kernel %s has no implementation", chv_kernel_name(kernel));
    case CHV_K_SND_S16I_S16I: case CHV_K_ME_FULLSEARCH:
        // (not layer kernels: chv_run_kernel runs them directly, run_idle_kernel)
        return fail(CHV_ERR_INVALID_OPERATION, "%s is not a picture-layer kernel: issue it through chv_run_kernel", chv_kernel_name(kernel));
    default:
        return fail(CHV_ERR_KERNEL_NOT_FOUND, "unknown kernel id %d", kernel);
    }
    return CHV_OK;
}

// buffers referenced by the descriptors being built (for upload -> kernel ordering)
static thread_local std::vector<chv_buffer *> *g_deps = nullptr;

struct DepScope {
    std::vector<chv_buffer *> bufs;
    DepScope() { g_deps = &bufs; }
    ~DepScope() { g_deps = nullptr; }
    // distinct buffers (a picture's planes usually share one allocation)
    std::vector<BatchDep> deps() const {
        // (a 256-tick batch names 2 300 planes of some hundreds of buffers, and is built afresh for every group tick of a host that batches: the
        // quadratic walk this replaces was 130 of the 300 us chv_batch_create took for it, tools/batch_create_probe.cpp.  A picture's planes
        // share an allocation — consecutive repeats go first —, the rest through a small open-addressing set.)
        std::vector<BatchDep> out;
        const size_t n = bufs.size();
        if (n <= 24) {
            for (chv_buffer *b : bufs) {
                bool seen = false;
                for (const BatchDep &d : out) seen = seen || d.buf == b;
                if (!seen) out.push_back(BatchDep{ b, nullptr, 0 });
            }
            return out;
        }
        size_t cap = 64;
        while (cap < 2 * n) cap <<= 1;
        std::vector<chv_buffer *> tab(cap, nullptr);
        chv_buffer *last = nullptr;
        for (chv_buffer *b : bufs) {
            if (b == last) continue;
            last = b;
            size_t h = (size_t)(((uintptr_t)b >> 6) * 0x9E3779B97F4A7C15ull >> 32) & (cap - 1);
            while (tab[h] && tab[h] != b) h = (h + 1) & (cap - 1);
            if (!tab[h]) { tab[h] = b; out.push_back(BatchDep{ b, nullptr, 0 }); }
        }
        return out;
    }
};

// Make `stream` wait for pending asynchronous uploads into the buffers a launch reads.  A batch remembers, per buffer,
// the (stream, upload) it last waited for, so replaying it on the same stream with no new upload costs nothing; a
// different stream, or a newer upload, waits again.
static int wait_for_uploads(hipStream_t stream, std::vector<BatchDep> &deps) {
    for (BatchDep &d : deps) {
        chv_buffer *b = d.buf;
        if (!b->ready_stream.load(std::memory_order_acquire)) continue;          // nothing pending (a batch of 256 mixers names a thousand buffers)
        std::lock_guard<std::mutex> lock(b->mu);
        if (!b->ready_stream || b->ready_stream == stream) continue;
        if (d.stream == stream && d.seq == b->ready_seq) continue;
        if (hipEventQuery(b->ready) == hipSuccess) { b->ready_stream = nullptr; continue; }
        (void)hipGetLastError();
        HIP_TRY(hipStreamWaitEvent(stream, b->ready, 0));
        d.stream = stream; d.seq = b->ready_seq;
    }
    return CHV_OK;
}

static int plane_to_device(const chv_plane &p, int comps, int device, DPlane *out, int err, const char *what, int idx) {
    if (!buf_ok(p.buffer)) return fail(err, "%s plane %d: no device buffer", what, idx);
    if (p.buffer->device != device) return fail(err, "%s plane %d lives on device %d, context on %d", what, idx, p.buffer->device, device);
    if (p.width <= 0 || p.height <= 0) return fail(err, "%s plane %d: size %dx%d", what, idx, p.width, p.height);
    if (p.components != comps) return fail(err, "%s plane %d: %d components, kernel expects %d", what, idx, p.components, comps);
    const int64_t row_bytes = (int64_t)p.width * comps;
    if (row_bytes > 0x3fffffff || p.height > 0x3fffffff) return fail(err, "%s plane %d: size %dx%d is out of range", what, idx, p.width, p.height);
    if ((int64_t)p.pitch < row_bytes) return fail(err, "%s plane %d: pitch %d < %lld", what, idx, p.pitch, (long long)row_bytes);
    size_t end = 0;
    if (__builtin_mul_overflow((size_t)(p.height - 1), (size_t)p.pitch, &end) || __builtin_add_overflow(end, (size_t)row_bytes, &end) ||
        __builtin_add_overflow(end, p.offset, &end))
        return fail(err, "%s plane %d: extent overflows", what, idx);
    if (end > p.buffer->size) return fail(err, "%s plane %d: extent %zu exceeds buffer size %zu", what, idx, end, p.buffer->size);
    if (comps == 4 && (((uintptr_t)p.buffer->ptr + p.offset) & 3 || (p.pitch & 3)))
        return fail(err, "%s plane %d: 4-component planes must be 4-byte aligned", what, idx);
    out->ptr = (uint8_t *)p.buffer->ptr + p.offset;
    out->w = p.width; out->h = p.height; out->pitch = p.pitch; out->comps = comps;
    if (g_deps) g_deps->push_back(p.buffer);
    return CHV_OK;
}

static int target_to_device(const chv_image *t, int target_format, int device, DImage *out) {
    if (!t) return fail(CHV_ERR_BAD_TARGET, "null target");
    static const int np[3] = { 2, 3, 1 };
    static const int comps[3][3] = { { 1, 2, 0 }, { 1, 1, 1 }, { 4, 0, 0 } };
    if (t->n_planes != np[target_format])
        return fail(CHV_ERR_BAD_TARGET, "target has %d planes, kernel writes %d", t->n_planes, np[target_format]);
    memset(out, 0, sizeof *out);
    for (int i = 0; i < t->n_planes; i++) {
        int rc = plane_to_device(t->planes[i], comps[target_format][i], device, &out->pl[i], CHV_ERR_BAD_TARGET, "target", i);
        if (rc) return rc;
    }
    return CHV_OK;
}

// host-side analysis of a layer's uniforms for fast-path selection
static int32_t layer_flags(const float *u) {
    int32_t f = 0;
    auto axis = [&](const float *m) {
        // row 0 must not depend on y/z, row 1 not on x/z, rows 2,3 = (0,0,*,*), so that
        // component 0 is a function of x only and component 1 of y only
        return m[1] == 0.f && m[2] == 0.f && m[4] == 0.f && m[6] == 0.f;
    };
    const float *T = u + U_TRANSFORM, *X = u + U_TEXTURE, *B = u + U_BORDER;
    // uv = textureTx . tx uses all four tx components: tx.z/tx.w must not depend on x or y
    bool tzw_const = T[8] == 0.f && T[9] == 0.f && T[12] == 0.f && T[13] == 0.f;
    if (axis(T) && axis(B) && axis(X) && tzw_const) f |= LF_AXIS_ALIGNED;
    // no fill contribution: alpha is exactly zero AND the colour is finite (0 * inf would be NaN)
    // (checked on fill * 255, the code-scale value the BGRA-target family multiplies by the alpha)
    auto finite255 = [](float c) { float v = c * 255.0f; return v - v == 0.f; };
    bool fill_finite = finite255(u[U_FILL]) && finite255(u[U_FILL + 1]) && finite255(u[U_FILL + 2]);
    if (u[U_OPACITY] * u[U_FILL + 3] == 0.f && fill_finite) f |= LF_NO_FILL;
    if (u[U_OPACITY] == 1.f) f |= LF_OPAQUE;
    // bounded matrices: with |entry| < 2^60 and |nx|, |ny| <= 3 no product of the geometry prologue overflows, so the
    // zero entries of an axis-aligned layer contribute exact zeros (geometry_axis, pixel_math.hip.h)
    bool bounded = true;
    for (int i = 0; i < 48; i++) bounded = bounded && std::fabs(u[i]) < 0x1p60f;      // false for NaN as well
    if (bounded) f |= LF_BOUNDED;
    return f;
}

// Conservative bounding box of {pixels whose border coordinates lie in [0,1]^2}.  Axis-aligned
// layers only (border.x depends on x alone); everything else gets the whole canvas.  The box is
// computed in double with a 2-pixel margin, far more than the float evaluation can differ by.
static void layer_bbox(DLayer *L, int W, int H) {
    L->bbox[0] = 0; L->bbox[1] = 0; L->bbox[2] = W; L->bbox[3] = H;
    const float *B = L->u + U_BORDER;
    if (!(L->flags & LF_AXIS_ALIGNED)) {
        // Rotated / sheared layers: the border coordinates are affine in the pixel — b = A n + t with n = (2 x / W - 1, 2 y / H - 1), z = 0, w = 1
        // (geometry(), pixel_math.hip.h) — so the pixels a layer can touch lie in the parallelogram A^-1 ([0,1]^2 - t).  Its box, in double with the
        // same 2-pixel margin; entries large enough for the float evaluation to differ by more than that (or a singular A) keep the whole canvas.
        // (Until round 4 these layers had the whole canvas for a box: a rotated logo made EVERY strip of the tick run its per-pixel loop.)
        const double a = B[0], b = B[1], c = B[4], d = B[5], tx = B[3], ty = B[7];
        double big = 0.0;
        for (double v : { a, b, c, d, tx, ty }) { if (!(v - v == 0.0)) return; big = std::max(big, std::fabs(v)); }
        const double det = a * d - b * c;
        if (!(big < 1048576.0) || !(std::fabs(det) > 1e-9 * std::max(1.0, big * big))) return;
        // The device evaluates b = A n + t in float (error ~ 4e-7 big) and the box is the image of [0,1]^2 under A^-1, which amplifies that by
        // ~ big / |det|: the float-accepted pixels can stick out of the double parallelogram by ~ 8e-7 big^2 / |det| * W / 2 pixels — more than
        // the 2-pixel margin once big^2 / |det| passes a few thousand (thin or strongly sheared layers with a large translation term).  Those keep
        // the whole canvas, as every such layer did before round 4.
        if (!(big * big / std::fabs(det) < 1e3)) return;
        double x0 = 1e300, x1 = -1e300, y0 = 1e300, y1 = -1e300;
        for (int k = 0; k < 4; k++) {
            const double u = (k & 1) - tx, v = (k >> 1) - ty;
            const double nx = (d * u - b * v) / det, ny = (-c * u + a * v) / det;
            const double px = (nx + 1.0) * W / 2.0, py = (ny + 1.0) * H / 2.0;
            x0 = std::min(x0, px); x1 = std::max(x1, px); y0 = std::min(y0, py); y1 = std::max(y1, py);
        }
        if (!(x0 - x0 == 0.0 && x1 - x1 == 0.0 && y0 - y0 == 0.0 && y1 - y1 == 0.0)) return;
        auto clampi = [](double v, int n) { return (int32_t)std::min(std::max(v, 0.0), (double)n); };
        L->bbox[0] = clampi(std::floor(x0) - 2.0, W); L->bbox[2] = clampi(std::ceil(x1) + 3.0, W);
        L->bbox[1] = clampi(std::floor(y0) - 2.0, H); L->bbox[3] = clampi(std::ceil(y1) + 3.0, H);
        return;
    }
    auto range = [](double k, double c, int n, int32_t *lo, int32_t *hi) {
        // 0 <= k * (2*x/n - 1) + c <= 1
        if (!(k - k == 0.0) || !(c - c == 0.0)) return;
        if (k == 0.0) { if (c < -1e-3 || c > 1.0 + 1e-3) { *lo = 0; *hi = 0; } return; }
        double a = ((0.0 - c) / k + 1.0) * n / 2.0, b = ((1.0 - c) / k + 1.0) * n / 2.0;
        if (a > b) std::swap(a, b);
        double l = std::floor(a) - 2.0, h = std::ceil(b) + 3.0;
        *lo = (int32_t)std::min(std::max(l, 0.0), (double)n);
        *hi = (int32_t)std::min(std::max(h, 0.0), (double)n);
    };
    // row 0: b0 = nx*B[0] + (ny*B[1] + 0*B[2]) + B[3] with B[1] = B[2] = 0
    range((double)B[0], (double)B[3], W, &L->bbox[0], &L->bbox[2]);
    range((double)B[5], (double)B[7], H, &L->bbox[1], &L->bbox[3]);
}

// LF_COVERS + ibox (device_types.h): the canvas pixels an opaque YUV-source layer replaces for sure — where border, transform and texture
// coordinates all lie in [0, 1] — shrunk by two pixels per side (the same margin layer_bbox adds).  Axis-aligned, bounded layers on BGRA canvases.
static void layer_inner_box(DLayer *L, int W, int H, int target_format) {
    L->ibox[0] = L->ibox[1] = L->ibox[2] = L->ibox[3] = 0;
    L->flags &= ~LF_COVERS;
    const int need = LF_AXIS_ALIGNED | LF_BOUNDED | LF_OPAQUE;
    if (target_format != TF_BGRA || (L->flags & need) != need) return;
    if (L->kind != LK_BGRA_FROM_NV12 && L->kind != LK_BGRA_FROM_Y420P) return;          // (RGB sources carry per-pixel alpha)
    const float *T = L->u + U_TRANSFORM, *X = L->u + U_TEXTURE, *B = L->u + U_BORDER;
    // per axis: v(n) = k n + c in [0, 1] for the three coordinates; n = 2 p / size - 1
    auto inner = [](double k, double c, int n, double *lo, double *hi) {
        if (!(k - k == 0.0) || !(c - c == 0.0) || k == 0.0) { *lo = 1.0; *hi = 0.0; return; }
        double a = ((0.0 - c) / k + 1.0) * n / 2.0, b = ((1.0 - c) / k + 1.0) * n / 2.0;
        if (a > b) std::swap(a, b);
        *lo = std::max(*lo, std::ceil(a) + 2.0); *hi = std::min(*hi, std::floor(b) - 2.0);
    };
    const double t3 = T[15];
    // The device evaluates u = fl(fl(t0 X0) + fl(t3 X3)) in float; with large, mutually cancelling terms its error can exceed the 2-pixel
    // shrink, and a strip judged covered would drop the layers beneath pixels that are in fact outside the texture range.  Modest magnitudes only
    // (the animator's matrices are O(canvas / picture): single digits to a few hundred).
    for (double m : { (double)T[3] * X[0], t3 * X[3], (double)T[7] * X[5], t3 * X[7], (double)T[3], (double)T[7], (double)B[3], (double)B[7],
                      (double)T[0] * X[0], (double)T[5] * X[5], (double)B[0], (double)B[5] })
        if (!(std::fabs(m) < 1e3)) return;
    double xl = 0.0, xh = W, yl = 0.0, yh = H;
    inner(B[0], B[3], W, &xl, &xh); inner(T[0], T[3], W, &xl, &xh); inner((double)T[0] * X[0], (double)T[3] * X[0] + t3 * X[3], W, &xl, &xh);
    inner(B[5], B[7], H, &yl, &yh); inner(T[5], T[7], H, &yl, &yh); inner((double)T[5] * X[5], (double)T[7] * X[5] + t3 * X[7], H, &yl, &yh);
    if (!(xl < xh && yl < yh)) return;
    L->ibox[0] = (int32_t)xl; L->ibox[2] = (int32_t)xh; L->ibox[1] = (int32_t)yl; L->ibox[3] = (int32_t)yh;
    L->flags |= LF_COVERS;
}

static int layer_to_device(const chv_layer &l, int device, int *target_format, DLayer *out) {
    KernelShape s;
    int rc = kernel_shape(l.kernel, &s);
    if (rc) return rc;
    if (s.is_clear) return fail(CHV_ERR_INVALID_OPERATION, "%s is not a layer kernel", chv_kernel_name(l.kernel));
    if (*target_format < 0) *target_format = s.target_format;
    else if (*target_format != s.target_format)
        return fail(CHV_ERR_BAD_TARGET, "%s does not write this target format", chv_kernel_name(l.kernel));
    if (l.image.n_planes < s.src_planes)
        return fail(CHV_ERR_BAD_INPUT, "%s needs %d input planes, image has %d", chv_kernel_name(l.kernel), s.src_planes, l.image.n_planes);
    memset(out, 0, sizeof *out);
    for (int i = 0; i < s.src_planes; i++) {
        rc = plane_to_device(l.image.planes[i], s.src_comps[i], device, &out->src.pl[i], CHV_ERR_BAD_INPUT, "input", i);
        if (rc) return rc;
    }
    static_assert(sizeof(chv_uniforms) == 236, "ImageUniforms must be 236 bytes (compute.swift:76-86)");
    memcpy(out->u, &l.uniforms, sizeof(chv_uniforms));
    if (s.kind == LK_BGRA_METAL) {
        if (!(out->u[U_OUTSIZE] > 0.f) || !(out->u[U_OUTSIZE + 1] > 0.f))
            return fail(CHV_ERR_INVALID_VALUE, "img_bgra_bgra needs a positive outputSize in the uniforms");
    }
    out->kind = s.kind;
    out->swizzle = s.swizzle;
    out->csc = l.opts.colorspace & 3;
    out->flags = layer_flags(out->u);
    return CHV_OK;
}

static int tick_to_device(const chv_tick &t, int device, int forced_target_format, DTick *dt,
                          std::vector<DLayer> *layers, int *target_format_out) {
    // (no upper bound here: a batch's descriptors live in device memory; chv_composite splits a tick into launches of
    // at most CHV_MAX_LAYERS layers, the capacity of a slot of its descriptor ring)
    if (t.n_layers < 0 || t.n_layers > 4096) return fail(CHV_ERR_INVALID_VALUE, "%d layers", t.n_layers);
    if (t.n_layers > 0 && !t.layers) return fail(CHV_ERR_BAD_INPUT, "null layers");
    int tf = forced_target_format;
    int first = (int)layers->size();
    if (layers->capacity() < layers->size() + (size_t)t.n_layers) layers->reserve(std::max(layers->capacity() * 2, layers->size() + (size_t)t.n_layers));
    for (int i = 0; i < t.n_layers; i++) {
        DLayer dl;
        int rc = layer_to_device(t.layers[i], device, &tf, &dl);
        if (rc) return rc;
        layers->push_back(dl);
    }
    if (tf < 0) {
        // no layers: infer the clear flavour from the target's plane structure
        if (t.target.n_planes == 1) tf = TF_BGRA;
        else if (t.target.n_planes == 2) tf = TF_NV12;
        else if (t.target.n_planes == 3) tf = TF_Y420P;
        else return fail(CHV_ERR_BAD_TARGET, "target has %d planes", t.target.n_planes);
    }
    memset(dt, 0, sizeof *dt);
    int rc = target_to_device(&t.target, tf, device, &dt->dst);
    if (rc) return rc;
    dt->W = dt->dst.pl[0].w;
    dt->H = dt->dst.pl[0].h;
    // (the boxes are functions of the matrices, the kind, the flags and the canvas: a group of mixers repeats a handful of geometries a thousand
    // times per batch, a lone mixer the same ones every tick — the last few answers are kept per thread; a dozen double divisions per miss)
    struct BoxMemo { float m[48]; int32_t kind, flags, W, H, tf, bbox[4], ibox[4], flags_out; bool used; };
    static thread_local BoxMemo memo[8];
    static thread_local unsigned memo_next = 0;
    for (int i = first; i < (int)layers->size(); i++) {
        DLayer &L = (*layers)[i];
        BoxMemo *hit = nullptr;
        for (BoxMemo &M : memo)
            if (M.used && M.kind == L.kind && M.flags == L.flags && M.W == dt->W && M.H == dt->H && M.tf == tf && memcmp(M.m, L.u, sizeof M.m) == 0) { hit = &M; break; }
        if (hit) {
            memcpy(L.bbox, hit->bbox, sizeof L.bbox); memcpy(L.ibox, hit->ibox, sizeof L.ibox); L.flags = hit->flags_out;
        } else {
            BoxMemo &M = memo[memo_next++ & 7];
            memcpy(M.m, L.u, sizeof M.m); M.kind = L.kind; M.flags = L.flags; M.W = dt->W; M.H = dt->H; M.tf = tf;
            if (L.kind == LK_BGRA_METAL) { L.bbox[0] = 0; L.bbox[1] = 0; L.bbox[2] = dt->W; L.bbox[3] = dt->H; }
            else layer_bbox(&L, dt->W, dt->H);
            layer_inner_box(&L, dt->W, dt->H, tf);
            memcpy(M.bbox, L.bbox, sizeof L.bbox); memcpy(M.ibox, L.ibox, sizeof L.ibox); M.flags_out = L.flags; M.used = true;
        }
        if ((L.flags & LF_COVERS) && i - first < 31) dt->cover_mask |= 1 << (i - first);
    }
    // LF_SAME_GEOM (device_types.h): a layer whose geometry inputs equal its predecessor's
    if (switches().same_geom.load(std::memory_order_relaxed)) {
        for (int i = first + 1; i < (int)layers->size(); i++) {
            DLayer &L = (*layers)[i];
            const DLayer &P = (*layers)[i - 1];
            bool same = L.kind == P.kind && L.kind != LK_BGRA_METAL && memcmp(L.u, P.u, 48 * sizeof(float)) == 0 &&
                        memcmp(L.bbox, P.bbox, sizeof L.bbox) == 0 &&
                        ((L.flags ^ P.flags) & (LF_AXIS_ALIGNED | LF_BOUNDED)) == 0;
            for (int p = 0; p < 3 && same; p++) {
                const DPlane &a = L.src.pl[p], &b = P.src.pl[p];
                same = a.w == b.w && a.h == b.h && a.pitch == b.pitch && a.comps == b.comps && (a.ptr == nullptr) == (b.ptr == nullptr);
            }
            if (same) L.flags |= LF_SAME_GEOM;
        }
    }
    dt->clear_first = t.clear_first ? 1 : 0;
    dt->n_layers = t.n_layers;
    dt->first_layer = first;
    *target_format_out = tf;
    return CHV_OK;
}

// One slot of the descriptor ring for the lifetime of the object: taken once the device is done with what the slot's group held (see
// DescGroup), given back on EVERY path out of the scope — a launch that fails after taking the group's last slot still leaves the event
// (or the `unguarded` mark) behind the launches of the group that were queued before it.  The walk advances only when the wait succeeded.
struct DescSlot {
    chv_context *c;
    int slot = -1;
    int rc = CHV_OK;
    explicit DescSlot(chv_context *ctx) : c(ctx) {
        const int s = c->next_desc;
        DescGroup &g = c->desc_group[s / kDescGroup];
        if (s % kDescGroup == 0) {
            hipError_t e = hipSuccess;
            if (g.unguarded) e = hipStreamSynchronize(c->stream);
            else if (g.pending) e = hipEventSynchronize(g.done);
            if (e != hipSuccess) { rc = hip_fail(e, "waiting for a descriptor slot"); return; }
            g.pending = false; g.unguarded = false;
        }
        c->next_desc = (s + 1) % kDescSlots;
        slot = s;
    }
    ~DescSlot() {
        if (slot < 0 || slot % kDescGroup != kDescGroup - 1) return;
        DescGroup &g = c->desc_group[slot / kDescGroup];
        hipError_t e = hipSuccess;
        if (!g.done) e = hipEventCreateWithFlags(&g.done, hipEventDisableTiming);
        if (e == hipSuccess) e = hipEventRecord(g.done, c->stream);
        if (e == hipSuccess) g.pending = true;
        else { g.pending = false; g.unguarded = true; (void)hipGetLastError(); }
    }
    DescSlot(const DescSlot &) = delete;
    DescSlot &operator=(const DescSlot &) = delete;
};

// launch one transient tick through the pinned descriptor ring
static int launch_transient(chv_context *c, const DTick &tick_in, const std::vector<DLayer> &layers, int tf) {
    HIP_TRY(hipSetDevice(c->device));
    // the tick as the kernels see it: first_layer = 0, its layers behind it
    DTick tick = tick_in;
    tick.first_layer = 0;
    const DTick *ht = &tick;
    const DLayer *hl = layers.data();
    // Where the kernel reads the descriptors from.  Every wave of the launch fetches its tick and, layer by layer, that layer's
    // planes, uniforms and flags with scalar loads — a handful of dependent loads per layer.  From the pinned host ring each of them
    // is a trip across PCIe (~1.5 us, uncached): a 4-layer 720p tick took 31 us of kernel time, 5 us per layer, whatever was done on
    // the chip (tools/tick_latency.py).  So the slot is copied to its twin in device memory on the launch's own stream first
    // (one small asynchronous copy), and the waves read L2 / scalar-cache resident descriptors.  CHV_DESC=host keeps the old way (A/B).
    // The streaming kernel goes one better for a lone tick: its descriptors travel as kernel ARGUMENTS (tick_bgra_stream_one) — no copy
    // in front of the launch, and every field a scalar load from one base instead of the tick -> layer chain.  CHV_DESC=device keeps the
    // copy for it too (A/B).
    const size_t used = sizeof(DTick) + layers.size() * sizeof(DLayer);
    const int desc_mode = switches().desc_host.load(std::memory_order_relaxed);       // 0 default, 1 host ring, 2 device twin always
    int path = select_fast_path(tf, ht, hl, 1, desc_mode == 0);
    (void)hipGetLastError();   // the launchers report through hipGetLastError(): drop whatever an earlier, unrelated call left there
    if (fast_path_by_value(path) && desc_mode == 0) {
        // descriptors as kernel arguments: the ring is not involved
        hipError_t e = launch_tick_fast(path, ht, hl, nullptr, nullptr, 1, ht->W, ht->H, c->stream);
        return e == hipSuccess ? CHV_OK : hip_fail(e, "kernel launch");
    }
    // The strip kernels take a lone tick of up to WAVE_ONE_LAYERS layers as an ARGUMENT (wave_common.hip.h: wave_one_descriptors; tick_bgra_wave_one): no ring
    // slot, no copy in front of the launch.  The layers are pointed at the device's geometry tables first, where its store has them for this
    // scene; a launch that is to BUILD tables (a scene's second sighting) needs its layers in device memory and goes through the slot below.
    const bool wave = fast_path_is_wave(path);
    GeomTransient &gt = geom_transient_current();
    gt.covered = false;
    if (wave && desc_mode == 0 && wave_layers_by_value(tf, &tick, layers.data())) {
        WaveOne one;
        one.t = tick;
        memcpy(one.l, layers.data(), layers.size() * sizeof(DLayer));
        bool build = false;
        gt.covered = geom_store_patch(tf, &one.t, one.l, 1, one.t.W, one.t.H, (int)layers.size(), &gt.cfg, &build);
        if (!build) {
            hipError_t e = launch_tick_fast(path, &one.t, one.l, nullptr, nullptr, 1, one.t.W, one.t.H, c->stream);
            gt.covered = false;
            return e == hipSuccess ? CHV_OK : hip_fail(e, "kernel launch");
        }
        gt.covered = false;
    }
    DescSlot ds(c);
    if (ds.rc) return ds.rc;
    const int slot = ds.slot;
    uint8_t *base = c->desc_host + (size_t)slot * kDescSlotBytes;
    DTick *st = (DTick *)base;
    DLayer *sl = (DLayer *)(base + sizeof(DTick));
    *st = tick;
    if (!layers.empty()) memcpy(sl, layers.data(), layers.size() * sizeof(DLayer));
    // the strip kernels' geometry tables, where the device's store has them for this scene: the layers are pointed at them before they travel
    // (not with CHV_DESC=host: a build would copy the slot onto itself)
    bool build_tables = false;
    if (wave && !layers.empty() && desc_mode != 1)
        gt.covered = geom_store_patch(tf, st, sl, 1, st->W, st->H, (int)layers.size(), &gt.cfg, &build_tables);
    DTick *dt = nullptr;
    if (desc_mode == 1) {
        HIP_TRY(hipHostGetDevicePointer((void **)&dt, st, 0));
    } else {
        dt = (DTick *)(c->desc_dev + (size_t)slot * kDescSlotBytes);
        HIP_TRY(hipMemcpyAsync(dt, st, used, hipMemcpyHostToDevice, c->stream));
    }
    DLayer *dl = (DLayer *)((uint8_t *)dt + sizeof(DTick));
    hipError_t e;
    if (build_tables) {
        // a geometry the device's store has been asked for before and does not have yet: this launch builds its tables (as a batch's second launch
        // does) and gives them to the store — the next tick of the scene finds them before its descriptors are copied
        GeomCache tmp;
        tmp.d_layers = dl; tmp.h_layers = sl; tmp.n_layers = (int)layers.size(); tmp.force_build = true;
        geom_cache_current() = &tmp;
        e = launch_tick_fast(path, st, sl, dt, dl, 1, st->W, st->H, c->stream);
        geom_cache_current() = nullptr;
        if (tmp.owns && tmp.tables) (void)hipStreamSynchronize(c->stream);          // (the store was full: the tables die with this launch)
        geom_cache_release(tmp);
    } else {
        e = path >= 0 ? launch_tick_fast(path, st, sl, dt, dl, 1, st->W, st->H, c->stream)
                      : launch_tick_general(tf, st, sl, dt, dl, 1, st->W, st->H, c->stream);
    }
    geom_transient_current().covered = false;
    if (e != hipSuccess) return hip_fail(e, "kernel launch");
    return CHV_OK;
}

// ---------------------------------------------------------------------------
// passes and kernels
// ---------------------------------------------------------------------------
static int composite_now(chv_context *c, const chv_image *target, int clear_first, int forced_tf, const chv_layer *layers, int n_layers);

static void pending_release(chv_context *c) {
    std::vector<chv_buffer *> dead;
    {
        std::lock_guard<std::mutex> lock(g_pin_mu);
        for (chv_buffer *b : c->pending.pins)
            if (--b->pins == 0 && b->doomed) dead.push_back(b);
    }
    for (chv_buffer *b : dead) (void)buffer_free_now(b);          // (hipFree waits for the launch that just went out)
    c->pending.pins.clear();
    c->pending.layers.clear();
    c->pending.active = false;
    c->pending.clear_first = 0;
    c->pending.clear_tf = -1;
}

// Launch what the pass has accepted so far.  Called by every entry point that puts work on the context's stream, hands the stream out, or
// waits for it — the pending kernels keep their place in the stream's order.  Their arguments were validated when they were accepted, so what
// can still fail here is the launch itself.
static int flush_pending(chv_context *c) {
    if (!c->pending.active) return CHV_OK;
    const int rc = composite_now(c, &c->pending.target, c->pending.clear_first, c->pending.clear_tf,
                                 c->pending.layers.empty() ? nullptr : c->pending.layers.data(), (int)c->pending.layers.size());
    pending_release(c);
    return rc;
}

static bool same_target(const chv_image &a, const chv_image &b) {
    if (a.n_planes != b.n_planes) return false;
    for (int i = 0; i < a.n_planes && i < 3; i++) {
        const chv_plane &p = a.planes[i], &q = b.planes[i];
        if (p.buffer != q.buffer || p.offset != q.offset || p.pitch != q.pitch || p.width != q.width || p.height != q.height || p.components != q.components)
            return false;
    }
    return true;
}

extern "C" int chv_pass_begin(chv_context *c) {
    if (!ctx_ok(c)) return fail(CHV_ERR_INVALID_CONTEXT, "bad context");
    // no device work: like OpenCL, a pass is just a bracket (compute.cl.swift:234-237).  Brackets nest (uploadComputePicture opens its own,
    // compute.cl.swift:433,453, downloadComputePicture :470,492): kernels are held while any is open, every chv_pass_end launches what is held
    if (c->pass_depth < (1 << 20)) c->pass_depth++;
    return CHV_OK;
}

extern "C" int chv_pass_end(chv_context *c, int wait) {
    if (!ctx_ok(c)) return fail(CHV_ERR_INVALID_CONTEXT, "bad context");
    if (c->pass_depth > 0) c->pass_depth--;
    HIP_TRY(hipSetDevice(c->device));
    // the pass's picture kernels, held back since chv_run_kernel accepted them: one fused launch where they are `clear + layers on one target`
    FLUSH_PENDING(c);
    // (hipStreamSynchronize already polls before it blocks: a hand-written hipStreamQuery loop in front of it measured the
    // same 16-17 us for an empty tick, tools/tick_latency.py — that floor is the launch + completion path of the runtime)
    if (wait) HIP_TRY(hipStreamSynchronize(c->stream));
    else (void)hipStreamQuery(c->stream);  // nudge submission, the clFlush analogue
    return CHV_OK;
}

// snd_s16i_s16i / me_fullsearch (kernels_idle.hip.cpp): buffers and luma planes bound as the reference binds them, no tick descriptors
static int run_idle_kernel(chv_context *c, int kernel, const chv_image *target, const chv_image *inputs, int n_inputs,
                           const void *uniforms, size_t uniforms_size) {
    FLUSH_PENDING(c);
    DepScope deps;
    if (target->n_planes < 1) return fail(CHV_ERR_BAD_TARGET, "%s: target without planes", chv_kernel_name(kernel));
    if (n_inputs < 0 || (n_inputs > 0 && !inputs)) return fail(CHV_ERR_BAD_INPUT, "null inputs");
    for (int i = 0; i < n_inputs; i++) if (inputs[i].n_planes < 1) return fail(CHV_ERR_BAD_INPUT, "%s: input %d has no planes", chv_kernel_name(kernel), i);
    int rc;
    if (kernel == CHV_K_SND_S16I_S16I) {
        if (!uniforms || uniforms_size != sizeof(chv_snd_uniforms))
            return fail(CHV_ERR_INVALID_VALUE, "snd_s16i_s16i needs the 100-byte BufferUniforms, got %zu bytes", uniforms_size);
        chv_snd_uniforms u;
        memcpy(&u, uniforms, sizeof u);
        if (u.input_count < 0 || u.input_count > 8) return fail(CHV_ERR_INVALID_VALUE, "snd_s16i_s16i: inputCount %d", u.input_count);
        if (n_inputs < u.input_count) return fail(CHV_ERR_BAD_INPUT, "snd_s16i_s16i: inputCount %d, %d buffers given", u.input_count, n_inputs);
        DPlane out;
        if ((rc = plane_to_device(target->planes[0], 2, c->device, &out, CHV_ERR_BAD_TARGET, "target", 0))) return rc;
        if (out.h > 1 && out.pitch != out.w * 2) return fail(CHV_ERR_BAD_TARGET, "snd_s16i_s16i: rows of samples must be contiguous");
        if (((uintptr_t)out.ptr) & 1) return fail(CHV_ERR_BAD_TARGET, "snd_s16i_s16i: samples must be 2-byte aligned");
        // the kernel's sample count is an int (the reference's get_global_size(0)): a buffer of 2^31 samples or more is refused, not truncated
        if ((int64_t)out.w * (int64_t)out.h > (int64_t)INT32_MAX)
            return fail(CHV_ERR_BAD_TARGET, "snd_s16i_s16i: %lld samples exceed the kernel's 32-bit count", (long long)((int64_t)out.w * out.h));
        const size_t span = (size_t)out.w * (size_t)out.h * 2;
        const int16_t *in[8] = { nullptr };
        for (int i = 0; i < u.input_count; i++) {
            DPlane p;
            if ((rc = plane_to_device(inputs[i].planes[0], 2, c->device, &p, CHV_ERR_BAD_INPUT, "input", i))) return rc;
            if (p.w != out.w || p.h != out.h || (p.h > 1 && p.pitch != p.w * 2) || (((uintptr_t)p.ptr) & 1))
                return fail(CHV_ERR_BAD_INPUT, "snd_s16i_s16i: input %d does not have the output's shape", i);
            // The source accumulates INTO out[gid] input after input (kernels.cl.swift:548-560), so an input that is the output would read the
            // partially mixed sample; the kernel keeps the accumulator in a register and reads every input once.  Not a case any caller has
            // (the inputs are other tracks' buffers): overlapping buffers are refused rather than mixed differently.
            if ((uintptr_t)p.ptr < (uintptr_t)out.ptr + span && (uintptr_t)out.ptr < (uintptr_t)p.ptr + span)
                return fail(CHV_ERR_BAD_INPUT, "snd_s16i_s16i: input %d overlaps the output buffer", i);
            in[i] = (const int16_t *)p.ptr;
        }
        auto dp = deps.deps();
        HIP_TRY(hipSetDevice(c->device));
        if ((rc = wait_for_uploads(c->stream, dp))) return rc;
        HIP_TRY(launch_snd_s16i((int16_t *)out.ptr, in, u.input_count, out.w * out.h, u.input_gains, u.input_fade, c->stream));
        return CHV_OK;
    }
    if (!uniforms || uniforms_size != sizeof(chv_me_uniforms))
        return fail(CHV_ERR_INVALID_VALUE, "me_fullsearch needs the 24-byte MotionEstimationUniforms, got %zu bytes", uniforms_size);
    chv_me_uniforms u;
    memcpy(&u, uniforms, sizeof u);
    if (n_inputs != 2) return fail(CHV_ERR_BAD_INPUT, "me_fullsearch takes the reference and the current picture, got %d images", n_inputs);
    if (u.block_size[0] < 1 || u.block_size[1] < 1 || u.block_size[0] > 64 || u.block_size[1] > 64 || u.search_window_size[0] < 0 || u.search_window_size[1] < 0)
        return fail(CHV_ERR_INVALID_VALUE, "me_fullsearch: block %dx%d, window %dx%d", u.block_size[0], u.block_size[1], u.search_window_size[0], u.search_window_size[1]);
    DPlane out, ref, cur;
    if ((rc = plane_to_device(target->planes[0], 4, c->device, &out, CHV_ERR_BAD_TARGET, "target", 0))) return rc;
    if ((rc = plane_to_device(inputs[0].planes[0], 1, c->device, &ref, CHV_ERR_BAD_INPUT, "reference picture", 0))) return rc;
    if ((rc = plane_to_device(inputs[1].planes[0], 1, c->device, &cur, CHV_ERR_BAD_INPUT, "current picture", 0))) return rc;
    // deltaCost2 (kernels.metal:135-142) per vector component, by |v|: lambda * (log2(|v| + 1) * 2 + 0.718 + (v != 0)) + 0.5 through the HOST's
    // log2f, the table oracle/ref_kernels.c builds the same way — no device libm in the comparison
    static float cost[256];
    static std::once_flag once;
    std::call_once(once, [] {
        for (int d = 0; d < 256; d++) {
            const float l2 = log2f((float)d + 1.0f), rounding = d != 0 ? 1.0f : 0.0f;
            cost[d] = 4.0f * (l2 * 2.0f + 0.718f + rounding) + 0.5f;
        }
    });
    auto dp = deps.deps();
    HIP_TRY(hipSetDevice(c->device));
    if ((rc = wait_for_uploads(c->stream, dp))) return rc;
    HIP_TRY(launch_me_fullsearch(out, ref, cur, u.block_size, u.search_window_size, u.image_size, cost, c->stream));
    return CHV_OK;
}

extern "C" int chv_run_kernel(chv_context *c, int kernel, const chv_image *target,
                              const chv_image *inputs, int n_inputs,
                              const void *uniforms, size_t uniforms_size, int blends,
                              const chv_kernel_opts *opts) {
    if (!ctx_ok(c)) return fail(CHV_ERR_INVALID_CONTEXT, "bad context");
    if (!target) return fail(CHV_ERR_BAD_TARGET, "null target");
    if (kernel == CHV_K_SND_S16I_S16I || kernel == CHV_K_ME_FULLSEARCH) return run_idle_kernel(c, kernel, target, inputs, n_inputs, uniforms, uniforms_size);
    KernelShape s;
    int rc = kernel_shape(kernel, &s);
    if (rc) return rc;
    chv_tick t;
    memset(&t, 0, sizeof t);
    t.target = *target;
    chv_layer layer;
    if (s.is_clear) {
        t.clear_first = 1;
        t.n_layers = 0;
    } else {
        if (n_inputs != 1 || !inputs)
            return fail(CHV_ERR_BAD_INPUT, "%s takes exactly one input image, got %d", chv_kernel_name(kernel), n_inputs);
        if (!uniforms || uniforms_size != sizeof(chv_uniforms))
            return fail(CHV_ERR_INVALID_VALUE, "%s needs the 236-byte ImageUniforms, got %zu bytes", chv_kernel_name(kernel), uniforms_size);
        if (!blends)
            return fail(CHV_ERR_INVALID_OPERATION, "%s reads the current target; call it with blends = true", chv_kernel_name(kernel));
        memset(&layer, 0, sizeof layer);
        layer.kernel = kernel;
        layer.image = inputs[0];
        memcpy(&layer.uniforms, uniforms, sizeof(chv_uniforms));
        if (opts) layer.opts = *opts;
        t.clear_first = 0;
        t.n_layers = 1;
        t.layers = &layer;
    }
    DTick dt;
    std::vector<DLayer> dl;
    int tf = -1;
    DepScope deps;
    rc = tick_to_device(t, c->device, s.is_clear ? s.target_format : -1, &dt, &dl, &tf);      // (every argument error surfaces here, in this call)
    if (rc) return rc;
    if (c->pass_depth > 0 && switches().pass_fuse.load(std::memory_order_relaxed)) {
        // Inside a pass: accept the kernel, launch it with the rest of the pass (flush_pending).  A kernel on another target, or a clear
        // after something else, first sends out what is held — the stream sees the kernels in the order they were issued.
        chv_context::PendingPass &pp = c->pending;
        // ... and so does a layer that READS the canvas being held (issued kernel by kernel it would sample what the held layers wrote; inside one
        // fused launch the canvas lives in registers and memory still has the old bytes), and a pass deeper than a batch's descriptor allows
        bool reads_held_canvas = false;
        if (pp.active && !s.is_clear)
            for (int i = 0; i < inputs[0].n_planes && i < 3; i++)
                for (int k = 0; k < pp.target.n_planes && k < 3; k++)
                    reads_held_canvas = reads_held_canvas || inputs[0].planes[i].buffer == pp.target.planes[k].buffer;
        if (pp.active && (s.is_clear || reads_held_canvas || pp.layers.size() >= 1024 || !same_target(pp.target, *target) || (pp.clear_tf >= 0 && pp.clear_tf != tf)))
            FLUSH_PENDING(c);
        if (!pp.active) {
            pp.active = true;
            pp.target = *target;
            pp.clear_first = 0;
            pp.clear_tf = -1;
        }
        if (s.is_clear) { pp.clear_first = 1; pp.clear_tf = s.target_format; }
        else pp.layers.push_back(layer);
        {
            std::lock_guard<std::mutex> lock(g_pin_mu);
            for (chv_buffer *b : deps.bufs) { b->pins++; pp.pins.push_back(b); }
        }
        return CHV_OK;
    }
    FLUSH_PENDING(c);
    auto dp = deps.deps();
    rc = wait_for_uploads(c->stream, dp);
    if (rc) return rc;
    return launch_transient(c, dt, dl, tf);
}

extern "C" int chv_composite(chv_context *c, const chv_image *target, int clear_first,
                             const chv_layer *layers, int n_layers) {
    if (!ctx_ok(c)) return fail(CHV_ERR_INVALID_CONTEXT, "bad context");
    if (!target) return fail(CHV_ERR_BAD_TARGET, "null target");
    if (n_layers < 0) return fail(CHV_ERR_INVALID_VALUE, "%d layers", n_layers);
    if (n_layers > 0 && !layers) return fail(CHV_ERR_BAD_INPUT, "null layers");
    FLUSH_PENDING(c);
    return composite_now(c, target, clear_first, -1, layers, n_layers);
}

// (`forced_tf`: the target format a clear kernel of a deferred pass fixed; -1: the layers' kernels decide, the plane count if there are none)
static int composite_now(chv_context *c, const chv_image *target, int clear_first, int forced_tf, const chv_layer *layers, int n_layers) {
    // A mixer composes any number of layers (mix.video.swift:116-124).  One launch takes up to CHV_MAX_LAYERS of them;
    // a deeper tick becomes several launches on the context's stream — the first clears, the others continue on the
    // canvas.  Byte-identical to one pass: the canvas is re-quantised between layers either way.
    // All chunks are validated before the first one is launched, so a bad layer leaves the canvas untouched.
    const int n_chunks = n_layers <= CHV_MAX_LAYERS ? 1 : (n_layers + CHV_MAX_LAYERS - 1) / CHV_MAX_LAYERS;
    std::vector<DTick> dts((size_t)n_chunks);
    std::vector<std::vector<DLayer>> dls((size_t)n_chunks);
    int tf = forced_tf;
    DepScope deps;
    for (int k = 0; k < n_chunks; k++) {
        chv_tick t;
        memset(&t, 0, sizeof t);
        t.target = *target;
        t.clear_first = k == 0 ? clear_first : 0;
        t.n_layers = std::min(CHV_MAX_LAYERS, n_layers - k * CHV_MAX_LAYERS);
        t.layers = layers ? layers + (size_t)k * CHV_MAX_LAYERS : nullptr;
        int tfk = tf;
        int rc = tick_to_device(t, c->device, tf, &dts[k], &dls[k], &tfk);
        if (rc) return rc;
        tf = tfk;
    }
    auto dp = deps.deps();
    int rc = wait_for_uploads(c->stream, dp);
    if (rc) return rc;
    for (int k = 0; k < n_chunks; k++) {
        rc = launch_transient(c, dts[k], dls[k], tf);
        if (rc) return rc;
    }
    return CHV_OK;
}

// ---------------------------------------------------------------------------
// batches
// ---------------------------------------------------------------------------
extern "C" int chv_batch_create(chv_context *c, const chv_tick *ticks, int n_ticks, chv_batch **out) {
    if (!out) return fail(CHV_ERR_INVALID_VALUE, "null argument");
    *out = nullptr;
    if (!ctx_ok(c)) return fail(CHV_ERR_INVALID_CONTEXT, "bad context");
    if (!ticks || n_ticks <= 0) return fail(CHV_ERR_INVALID_VALUE, "empty batch");
    if (n_ticks > 65535) return fail(CHV_ERR_INVALID_VALUE, "at most 65535 ticks per batch");
    std::vector<DTick> dts((size_t)n_ticks);
    std::vector<DLayer> dls;
    {
        size_t total = 0;
        for (int i = 0; i < n_ticks; i++) total += ticks[i].n_layers > 0 && ticks[i].n_layers <= 4096 ? (size_t)ticks[i].n_layers : 0;
        dls.reserve(total);
    }
    int tf0 = -1, maxW = 0, maxH = 0;
    DepScope deps;
    deps.bufs.reserve((size_t)n_ticks * 3 + dls.capacity() * 3);
    for (int i = 0; i < n_ticks; i++) {
        int tf = -1;
        int rc = tick_to_device(ticks[i], c->device, -1, &dts[i], &dls, &tf);
        if (rc) return rc;
        if (tf0 < 0) tf0 = tf;
        else if (tf != tf0) return fail(CHV_ERR_BAD_TARGET, "all ticks of a batch must share the target format");
        if (dts[i].W > maxW) maxW = dts[i].W;
        if (dts[i].H > maxH) maxH = dts[i].H;
    }
    HIP_TRY(hipSetDevice(c->device));
    std::unique_ptr<chv_batch> b(new chv_batch);
    b->device = c->device; b->n_ticks = n_ticks; b->n_layers = (int)dls.size();
    b->target_format = tf0; b->maxW = maxW; b->maxH = maxH;
    // Everything that decides what the descriptors look like happens on the host first — the route, a split into two launches, the geometry
    // tables the device's store has for the scene — so that they travel once.
    b->fast_path = select_fast_path(tf0, dts.data(), dls.data(), n_ticks);
    auto path_name = [&](int path) { return path >= 0 ? std::string(fast_path_name(path))
                                                       : std::string(tf0 == TF_BGRA ? "tick_general_bgra" : (tf0 == TF_NV12 ? "tick_general_yuv<nv12>" : "tick_general_yuv<y420p>")); };
    b->kernel_name = path_name(b->fast_path);
    if (tf0 == TF_BGRA && b->fast_path != fast_path_stream_bgra()) {
        const int k = split_stream_prefix(dts.data(), dls.data(), n_ticks);
        if (k > 0) {
            std::vector<DTick> head = dts, tail = dts;
            for (int i = 0; i < n_ticks; i++) {
                head[(size_t)i].n_layers = k;
                head[(size_t)i].cover_mask &= (1 << k) - 1;
                tail[(size_t)i].first_layer += k; tail[(size_t)i].n_layers -= k; tail[(size_t)i].clear_first = 0;
                tail[(size_t)i].cover_mask = (int32_t)((uint32_t)tail[(size_t)i].cover_mask >> k);
            }
            const int p1 = select_fast_path(tf0, head.data(), dls.data(), n_ticks), p2 = select_tail_path(tf0, tail.data(), dls.data(), n_ticks);
            if (p1 == fast_path_stream_bgra()) {
                b->fast_path = p1; b->fast_path2 = p2;
                b->h_ticks2 = std::move(tail);
                dts = std::move(head);
                b->kernel_name = path_name(p1) + " + " + path_name(p2);
            }
        }
    }
    // The strip kernels' geometry tables, where the device's store has them for this scene (a batch is bound to its pictures: a host builds one per
    // group of frames, and the second one of a scene onwards starts with tables): the layers are pointed at them before they are sent.
    {
        const bool w1 = fast_path_is_wave(b->fast_path), w2 = b->fast_path2 != -2 && fast_path_is_wave(b->fast_path2);
        if ((w1 || w2) && !dls.empty()) {
            GeomConfig cfg{};
            bool want_build = false;
            const std::vector<DTick> &tk = w1 ? dts : b->h_ticks2;
            if (geom_store_patch(tf0, tk.data(), dls.data(), n_ticks, b->maxW, b->maxH, (int)dls.size(), &cfg, &want_build)) {
                b->geom.built = true; b->geom.patched = true; b->geom.config = cfg; b->geom.owns = false; b->geom.tables = nullptr;
            }
        }
    }
    // ticks | layers | the second launch's ticks: one block of the device's pool (DescBlock), one asynchronous copy on this context's stream
    const size_t tb = sizeof(DTick) * dts.size(), lb = sizeof(DLayer) * (dls.size() ? dls.size() : 1), t2b = sizeof(DTick) * b->h_ticks2.size();
    const size_t lo = (tb + 255) & ~(size_t)255, t2o = (lo + lb + 255) & ~(size_t)255, total = t2o + t2b;
    hipError_t e = hipSuccess;
    if (desc_block_acquire(c->device, total, &b->blk)) {
        DescBlock &K = b->blk;
        if (K.recorded) e = hipEventSynchronize(K.ev);                 // (its last user's copy and launches: long done for a host that waited for its tick)
        if (e == hipSuccess) {
            memcpy(K.host, dts.data(), tb);
            if (!dls.empty()) memcpy(K.host + lo, dls.data(), sizeof(DLayer) * dls.size());
            if (t2b) memcpy(K.host + t2o, b->h_ticks2.data(), t2b);
            e = hipMemcpyAsync(K.dev, K.host, total, hipMemcpyHostToDevice, c->stream);
        }
        if (e == hipSuccess) e = hipEventRecord(K.ev, c->stream);
        if (e != hipSuccess) { desc_block_release(c->device, K); return hip_fail(e, "descriptors of a batch"); }
        K.recorded = true; K.last = c->stream;
        b->d_ticks = (DTick *)K.dev; b->d_layers = (DLayer *)(K.dev + lo); b->d_ticks2 = t2b ? (DTick *)(K.dev + t2o) : nullptr;
    } else {
        e = hipMalloc((void **)&b->d_ticks, tb);
        if (e == hipSuccess) e = hipMalloc((void **)&b->d_layers, lb);
        if (e == hipSuccess && t2b) e = hipMalloc((void **)&b->d_ticks2, t2b);
        if (e == hipSuccess) e = hipMemcpy(b->d_ticks, dts.data(), tb, hipMemcpyHostToDevice);
        if (e == hipSuccess && !dls.empty()) e = hipMemcpy(b->d_layers, dls.data(), sizeof(DLayer) * dls.size(), hipMemcpyHostToDevice);
        if (e == hipSuccess && t2b) e = hipMemcpy(b->d_ticks2, b->h_ticks2.data(), t2b, hipMemcpyHostToDevice);
        if (e != hipSuccess) {
            if (b->d_ticks) (void)hipFree(b->d_ticks);
            if (b->d_layers) (void)hipFree(b->d_layers);
            if (b->d_ticks2) (void)hipFree(b->d_ticks2);
            return hip_fail(e, "descriptors of a batch");
        }
    }
    b->h_ticks = std::move(dts);
    b->h_layers = std::move(dls);
    b->deps = deps.deps();
    *out = b.release();
    return CHV_OK;
}

extern "C" int chv_batch_run(chv_context *c, chv_batch *b) {
    if (!ctx_ok(c)) return fail(CHV_ERR_INVALID_CONTEXT, "bad context");
    FLUSH_PENDING(c);
    if (!b || !b->d_ticks) return fail(CHV_ERR_INVALID_VALUE, "bad batch");
    if (b->device != c->device) return fail(CHV_ERR_INVALID_CONTEXT, "batch belongs to device %d", b->device);
    HIP_TRY(hipSetDevice(c->device));
    int wrc = wait_for_uploads(c->stream, b->deps);
    if (wrc) return wrc;
    (void)hipGetLastError();   // see launch_transient
    // (pooled descriptors: their copy went out on the creating context's stream, and the block's event marks the last thing that used it)
    if (b->blk.dev && b->blk.last != c->stream) HIP_TRY(hipStreamWaitEvent(c->stream, b->blk.ev, 0));
    // ... and on every way out from here the block's event goes behind whatever this call queued (the launchers may queue more than the tick
    // kernels — a table build —, and a failing second launch leaves the first one in flight): what the next user of the block waits for
    struct BlockMark {
        chv_batch *bb; hipStream_t st;
        ~BlockMark() { if (bb->blk.dev) { if (hipEventRecord(bb->blk.ev, st) == hipSuccess) bb->blk.last = st; else (void)hipGetLastError(); } }
    } mark{ b, c->stream };
    // (the strip kernels' launcher finds the batch's geometry tables through this: geom_cache.h)
    struct CacheScope {
        explicit CacheScope(chv_batch *bb) {
            bb->geom.d_layers = bb->d_layers; bb->geom.h_layers = bb->h_layers.data(); bb->geom.n_layers = (int)bb->h_layers.size();
            geom_cache_current() = &bb->geom;
        }
        ~CacheScope() { geom_cache_current() = nullptr; }
    } scope(b);
    hipError_t e = b->fast_path >= 0
        ? launch_tick_fast(b->fast_path, b->h_ticks.data(), b->h_layers.data(), b->d_ticks, b->d_layers, b->n_ticks, b->maxW, b->maxH, c->stream)
        : launch_tick_general(b->target_format, b->h_ticks.data(), b->h_layers.data(), b->d_ticks, b->d_layers, b->n_ticks, b->maxW, b->maxH, c->stream);
    if (e != hipSuccess) return hip_fail(e, "kernel launch");
    if (b->fast_path2 != -2) {
        e = b->fast_path2 >= 0
            ? launch_tick_fast(b->fast_path2, b->h_ticks2.data(), b->h_layers.data(), b->d_ticks2, b->d_layers, b->n_ticks, b->maxW, b->maxH, c->stream)
            : launch_tick_general(b->target_format, b->h_ticks2.data(), b->h_layers.data(), b->d_ticks2, b->d_layers, b->n_ticks, b->maxW, b->maxH, c->stream);
        if (e != hipSuccess) return hip_fail(e, "kernel launch (second part of the batch)");
    }
    return CHV_OK;
}

extern "C" int chv_batch_destroy(chv_batch *b) {
    if (!b) return fail(CHV_ERR_INVALID_VALUE, "null batch");
    (void)hipSetDevice(b->device);
    if (b->blk.dev) desc_block_release(b->device, b->blk);           // (back to the pool: whoever takes it next waits for its event)
    else {
        (void)hipFree(b->d_ticks);
        (void)hipFree(b->d_layers);
        if (b->d_ticks2) (void)hipFree(b->d_ticks2);
    }
    geom_cache_release(b->geom);
    b->d_ticks = nullptr;
    delete b;
    return CHV_OK;
}

extern "C" int chv_batch_describe(chv_batch *b, char *kernel_name, size_t cap, int *n_launches) {
    if (!b) return fail(CHV_ERR_INVALID_VALUE, "null batch");
    if (kernel_name && cap) snprintf(kernel_name, cap, "%s", b->kernel_name.c_str());
    if (n_launches) *n_launches = b->fast_path2 != -2 ? 2 : 1;
    return CHV_OK;
}

// ---------------------------------------------------------------------------
// Lanczos-3: tables in double on the host, resampling on the GPU
// ---------------------------------------------------------------------------
// ---------------------------------------------------------------------------
// custom kernels (compute.swift:72-73, compute.cl.swift:153-232) through hipRTC
// ---------------------------------------------------------------------------
static const char kCustomPrelude[] = R"CHV(
// ---- CHIPVideo custom-kernel prelude (prefixed to every source given to chv_kernel_build) ----
typedef unsigned char uint8_t;
typedef int int32_t;
typedef unsigned int uint32_t;
struct chv_dev_plane { uint8_t *ptr; int32_t width, height, pitch, components; };
struct chv_dev_image { chv_dev_plane planes[3]; int32_t n_planes, format; };
struct chv_custom_args {
    chv_dev_image target, current;
    chv_dev_image inputs[4];
    int32_t n_inputs, uniforms_size;
    uint8_t uniforms[256];
};
// ImageUniforms, compute.swift:76-86 (236 bytes); each matrix holds the rows of M^-1
struct ImageUniforms {
    float transform[16], textureTransform[16], borderMatrix[16];
    float fillColor[4], inputSize[2], outputSize[2];
    float opacity, imageTime, targetTime;
};
#define CHV_GID_X ((int)(blockIdx.x * blockDim.x + threadIdx.x))
#define CHV_GID_Y ((int)(blockIdx.y * blockDim.y + threadIdx.y))
// the launch is rounded up to 16x16 blocks: leave when outside plane `p`
#define CHV_GUARD(p) do { if (CHV_GID_X >= (p).width || CHV_GID_Y >= (p).height) return; } while (0)
// vecmat4, kernels.cl.swift:27: (dot(v, row0), dot(v, row1), dot(v, row2), dot(v, row3))
__device__ inline float4 chv_vecmat4(float4 v, const float *m) {
    return make_float4(((v.x * m[0] + v.y * m[1]) + v.z * m[2]) + v.w * m[3], ((v.x * m[4] + v.y * m[5]) + v.z * m[6]) + v.w * m[7],
                       ((v.x * m[8] + v.y * m[9]) + v.z * m[10]) + v.w * m[11], ((v.x * m[12] + v.y * m[13]) + v.z * m[14]) + v.w * m[15]);
}
// UNORM_INT8 conversions of OpenCL 1.2 section 8.3.1.1
__device__ inline float chv_unorm8(uint8_t c) { return (float)c / 255.0f; }
__device__ inline uint8_t chv_to_unorm8(float f) {
    float v = rintf(f * 255.0f);
    return (uint8_t)fminf(fmaxf(v, 0.0f), 255.0f);         // NaN -> 0
}
// read_imagef, nearest, unnormalized integer coordinates (zero outside the plane)
__device__ inline float chv_read(const chv_dev_plane &p, int x, int y, int c) {
    if (x < 0 || y < 0 || x >= p.width || y >= p.height) return 0.0f;
    return chv_unorm8(p.ptr[(size_t)y * p.pitch + (size_t)x * p.components + c]);
}
// write_imagef of one component (dropped outside the plane)
__device__ inline void chv_write(const chv_dev_plane &p, int x, int y, int c, float f) {
    if (x < 0 || y < 0 || x >= p.width || y >= p.height) return;
    p.ptr[(size_t)y * p.pitch + (size_t)x * p.components + c] = chv_to_unorm8(f);
}
// read_imagef, LINEAR | CLAMP_TO_EDGE | NORMALIZED (OpenCL 1.2 section 8.2), component c
__device__ inline float chv_sample(const chv_dev_plane &p, float u, float v, int c) {
    float um = u * (float)p.width - 0.5f, vm = v * (float)p.height - 0.5f;
    float fu = floorf(um), fv = floorf(vm);
    float a = um - fu, b = vm - fv;
    int i0 = min(max((int)fu, 0), p.width - 1), i1 = min(max((int)fu + 1, 0), p.width - 1);
    int j0 = min(max((int)fv, 0), p.height - 1), j1 = min(max((int)fv + 1, 0), p.height - 1);
    const uint8_t *r0 = p.ptr + (size_t)j0 * p.pitch + c, *r1 = p.ptr + (size_t)j1 * p.pitch + c;
    float t00 = chv_unorm8(r0[i0 * p.components]), t10 = chv_unorm8(r0[i1 * p.components]);
    float t01 = chv_unorm8(r1[i0 * p.components]), t11 = chv_unorm8(r1[i1 * p.components]);
    return (((1.0f - a) * (1.0f - b) * t00 + a * (1.0f - b) * t10) + (1.0f - a) * b * t01) + a * b * t11;
}
// ---- end of prelude ----
#line 1 "custom_kernel"
)CHV";

extern "C" const char *chv_custom_prelude(void) { return kCustomPrelude; }

extern "C" int chv_kernel_build(chv_context *c, const char *name, const char *source) {
    if (!ctx_ok(c)) return fail(CHV_ERR_INVALID_CONTEXT, "bad context");
    if (!name || !*name || !source) return fail(CHV_ERR_INVALID_VALUE, "null kernel name or source");
    (void)hipSetDevice(c->device);
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, c->device));
    std::string text = std::string(kCustomPrelude) + source;
    hiprtcProgram prog = nullptr;
    if (hiprtcCreateProgram(&prog, text.c_str(), name, 0, nullptr, nullptr) != HIPRTC_SUCCESS)
        return fail(CHV_ERR_UNKNOWN, "hiprtcCreateProgram failed");
    std::string arch = std::string("--offload-arch=") + prop.gcnArchName;
    const char *opts[] = { arch.c_str(), "-O3", "-ffp-contract=off", "-std=c++17" };
    hiprtcResult cr = hiprtcCompileProgram(prog, 4, opts);
    if (cr != HIPRTC_SUCCESS) {
        size_t n = 0;
        std::string log;
        if (hiprtcGetProgramLogSize(prog, &n) == HIPRTC_SUCCESS && n > 1) { log.resize(n); (void)hiprtcGetProgramLog(prog, &log[0]); }
        (void)hiprtcDestroyProgram(&prog);
        if (log.size() > 3000) log.resize(3000);
        return fail(CHV_ERR_BAD_INPUT, "Unable to create kernel named %s (%s)\nBuild log:\n%s", name, hiprtcGetErrorString(cr), log.c_str());
    }
    size_t code_size = 0;
    std::vector<char> code;
    if (hiprtcGetCodeSize(prog, &code_size) != HIPRTC_SUCCESS || code_size == 0) { (void)hiprtcDestroyProgram(&prog); return fail(CHV_ERR_UNKNOWN, "hiprtcGetCodeSize failed"); }
    code.resize(code_size);
    hiprtcResult gr = hiprtcGetCode(prog, code.data());
    (void)hiprtcDestroyProgram(&prog);
    if (gr != HIPRTC_SUCCESS) return fail(CHV_ERR_UNKNOWN, "hiprtcGetCode failed");
    auto k = std::make_shared<CustomKernel>();
    k->device = c->device;
    hipError_t e = hipModuleLoadData(&k->module, code.data());
    if (e != hipSuccess) return hip_fail(e, "hipModuleLoadData");
    e = hipModuleGetFunction(&k->fn, k->module, name);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        return fail(CHV_ERR_BAD_INPUT, "Unable to create kernel named %s: the source defines no extern \"C\" __global__ function of that name", name);
    }
    c->library[name] = std::move(k);     // merging { $1 }: the new kernel replaces an older one (compute.cl.swift:190-194)
    return CHV_OK;
}

static int image_to_device_any(const chv_image *img, int device, chv_dev_image *out, int err, const char *what) {
    memset(out, 0, sizeof *out);
    if (!img) return fail(err, "null %s image", what);
    if (img->n_planes < 1 || img->n_planes > 3) return fail(err, "%s image has %d planes", what, img->n_planes);
    for (int i = 0; i < img->n_planes; i++) {
        const chv_plane &p = img->planes[i];
        if (p.components != 1 && p.components != 2 && p.components != 4) return fail(err, "%s plane %d: %d components", what, i, p.components);
        DPlane d;
        int rc = plane_to_device(p, p.components, device, &d, err, what, i);
        if (rc) return rc;
        out->planes[i].ptr = d.ptr; out->planes[i].width = d.w; out->planes[i].height = d.h;
        out->planes[i].pitch = d.pitch; out->planes[i].components = d.comps;
    }
    out->n_planes = img->n_planes;
    out->format = img->format;
    return CHV_OK;
}

extern "C" int chv_run_custom(chv_context *c, const char *name, const chv_image *target,
                              const chv_image *inputs, int n_inputs,
                              const void *uniforms, size_t uniforms_size, int blends) {
    if (!ctx_ok(c)) return fail(CHV_ERR_INVALID_CONTEXT, "bad context");
    FLUSH_PENDING(c);
    if (!name) return fail(CHV_ERR_INVALID_VALUE, "null kernel name");
    auto it = c->library.find(name);
    if (it == c->library.end()) return fail(CHV_ERR_KERNEL_NOT_FOUND, "no custom kernel named %s in this context's library", name);
    if (n_inputs < 0 || n_inputs > CHV_CUSTOM_MAX_INPUTS) return fail(CHV_ERR_BAD_INPUT, "%d input images (max %d)", n_inputs, CHV_CUSTOM_MAX_INPUTS);
    if (n_inputs > 0 && !inputs) return fail(CHV_ERR_BAD_INPUT, "null inputs");
    if (uniforms_size > CHV_CUSTOM_MAX_UNIFORMS || (uniforms_size && !uniforms))
        return fail(CHV_ERR_INVALID_VALUE, "%zu uniform bytes (max %d)", uniforms_size, CHV_CUSTOM_MAX_UNIFORMS);
    (void)hipSetDevice(c->device);
    chv_custom_args a;
    memset(&a, 0, sizeof a);
    DepScope deps;
    int rc = image_to_device_any(target, c->device, &a.target, CHV_ERR_BAD_TARGET, "target");
    if (rc) return rc;
    if (blends) a.current = a.target;
    for (int i = 0; i < n_inputs; i++) {
        rc = image_to_device_any(&inputs[i], c->device, &a.inputs[i], CHV_ERR_BAD_INPUT, "input");
        if (rc) return rc;
    }
    a.n_inputs = n_inputs;
    a.uniforms_size = (int32_t)uniforms_size;
    if (uniforms_size) memcpy(a.uniforms, uniforms, uniforms_size);
    auto dp = deps.deps();
    rc = wait_for_uploads(c->stream, dp);
    if (rc) return rc;
    size_t size = sizeof a;
    void *config[] = { HIP_LAUNCH_PARAM_BUFFER_POINTER, &a, HIP_LAUNCH_PARAM_BUFFER_SIZE, &size, HIP_LAUNCH_PARAM_END };
    const unsigned gx = (unsigned)(a.target.planes[0].width + 15) / 16, gy = (unsigned)(a.target.planes[0].height + 15) / 16;
    HIP_TRY(hipModuleLaunchKernel(it->second->fn, gx, gy, 1, 16, 16, 1, 0, c->stream, nullptr, config));
    return CHV_OK;
}

static double sinc_pi(double t) {
    if (t == 0.0) return 1.0;
    double pt = 3.14159265358979323846 * t;
    return std::sin(pt) / pt;
}

static int lanczos_host_table(int in_size, int out_size, int *taps_out, std::vector<int32_t> *first,
                              std::vector<float> *weights) {
    double scale = (double)in_size / (double)out_size;
    double fs = scale > 1.0 ? scale : 1.0;
    double support = 3.0 * fs;
    int taps = 2 * (int)std::ceil(support);
    if (taps > 256) return fail(CHV_ERR_INVALID_VALUE, "Lanczos ratio %d:%d needs %d taps (max 256)", in_size, out_size, taps);
    first->resize(out_size);
    weights->resize((size_t)out_size * taps);
    std::vector<double> w(taps);
    for (int o = 0; o < out_size; o++) {
        double center = ((double)o + 0.5) * scale - 0.5;
        int f = (int)std::floor(center - support) + 1;
        double sum = 0.0;
        for (int k = 0; k < taps; k++) {
            double t = ((double)(f + k) - center) / fs;
            double v = (t > -3.0 && t < 3.0) ? sinc_pi(t) * sinc_pi(t / 3.0) : 0.0;
            w[k] = v;
            sum += v;
        }
        (*first)[o] = f;
        for (int k = 0; k < taps; k++) (*weights)[(size_t)o * taps + k] = (float)(w[k] / sum);
    }
    *taps_out = taps;
    return CHV_OK;
}

static int lanczos_table(chv_context *c, int in_size, int out_size, LanczosRef *out) {
    DeviceShared &sh = *c->shared;
    auto key = std::make_pair(in_size, out_size);
    {
        std::lock_guard<std::mutex> lock(sh.mu);
        auto it = sh.lanczos.find(key);
        if (it != sh.lanczos.end()) { it->second.last_use = ++sh.lanczos_clock; *out = it->second.tab; return CHV_OK; }
    }
    // build and upload outside the lock: other contexts of the device keep running meanwhile
    std::vector<int32_t> first;
    std::vector<float> weights;
    LanczosRef t = std::make_shared<LanczosTable>();
    t->device = c->device;
    int rc = lanczos_host_table(in_size, out_size, &t->taps, &first, &weights);
    if (rc) return rc;
    t->first0 = first[0];
    t->stride = first.size() > 1 ? first[1] - first[0] : 0;
    for (size_t o = 1; o + 1 < first.size() && t->stride; o++) if (first[o + 1] - first[o] != t->stride) t->stride = 0;
    HIP_TRY(hipMalloc((void **)&t->first, first.size() * sizeof(int32_t)));
    hipError_t e = hipMalloc((void **)&t->weights, weights.size() * sizeof(float));
    if (e != hipSuccess) t->weights = nullptr;
    if (e == hipSuccess)
        e = hipMemcpy(t->first, first.data(), first.size() * sizeof(int32_t), hipMemcpyHostToDevice);
    if (e == hipSuccess)
        e = hipMemcpy(t->weights, weights.data(), weights.size() * sizeof(float), hipMemcpyHostToDevice);
    if (e != hipSuccess) return hip_fail(e, "lanczos table upload");     // (t frees what it holds)
    // Evicted tables are parked and freed kLanczosRetireBatch at a time — an animated resize makes a new (in, out) pair per
    // frame, and one hipFree per miss would put a device-wide synchronisation into every tick.  Only tables nobody else
    // holds are freed: a context between its lookup and its launch keeps its reference, and the table with it.
    std::vector<LanczosRef> to_free;
    {
        std::lock_guard<std::mutex> lock(sh.mu);
        auto it = sh.lanczos.find(key);
        if (it != sh.lanczos.end()) {                       // another context built the same table meanwhile: keep theirs
            it->second.last_use = ++sh.lanczos_clock;
            *out = it->second.tab;
            to_free.push_back(std::move(t));
        } else {
            if (sh.lanczos.size() >= kLanczosCacheEntries) {
                auto lru = sh.lanczos.begin();
                for (auto jt = sh.lanczos.begin(); jt != sh.lanczos.end(); ++jt) if (jt->second.last_use < lru->second.last_use) lru = jt;
                sh.lanczos_retired.push_back(std::move(lru->second.tab));
                sh.lanczos.erase(lru);
            }
            LanczosEntry en;
            en.tab = t; en.last_use = ++sh.lanczos_clock;
            sh.lanczos[key] = en;
            *out = std::move(t);
            if (sh.lanczos_retired.size() >= kLanczosRetireBatch) {
                std::vector<LanczosRef> keep;
                for (LanczosRef &r : sh.lanczos_retired) (r.use_count() == 1 ? to_free : keep).push_back(std::move(r));
                sh.lanczos_retired.swap(keep);
            }
        }
    }
    to_free.clear();      // outside the lock: ~LanczosTable -> hipFree, which waits for queued work that may still read the tables
    return CHV_OK;
}

extern "C" int chv_scale_lanczos(chv_context *c, const chv_image *dst, const chv_image *src) {
    if (!ctx_ok(c)) return fail(CHV_ERR_INVALID_CONTEXT, "bad context");
    FLUSH_PENDING(c);
    if (!dst || dst->n_planes != 1) return fail(CHV_ERR_BAD_TARGET, "Lanczos target must be one 4-component plane");
    if (!src || src->n_planes != 1) return fail(CHV_ERR_BAD_INPUT, "Lanczos source must be one 4-component plane");
    DPlane d, s;
    DepScope deps;
    int rc = plane_to_device(dst->planes[0], 4, c->device, &d, CHV_ERR_BAD_TARGET, "target", 0);
    if (rc) return rc;
    rc = plane_to_device(src->planes[0], 4, c->device, &s, CHV_ERR_BAD_INPUT, "input", 0);
    if (rc) return rc;
    HIP_TRY(hipSetDevice(c->device));
    auto dp = deps.deps();
    rc = wait_for_uploads(c->stream, dp);
    if (rc) return rc;
    LanczosRef tx, ty;      // held until the launch is enqueued (see LanczosTable)
    rc = lanczos_table(c, s.w, d.w, &tx);
    if (rc) return rc;
    rc = lanczos_table(c, s.h, d.h, &ty);
    if (rc) return rc;
    (void)hipGetLastError();
    hipError_t e = launch_lanczos(d, s, tx->first, tx->weights, tx->taps, ty->first, ty->weights, ty->taps, c->stream, nullptr, 0, tx->stride, tx->first0, ty->stride);
    if (e != hipSuccess) return hip_fail(e, "lanczos launch");
    return CHV_OK;
}

// Many resizes of ONE geometry (all sources of one size, all targets of one size) in one launch per chunk: what a host with
// several streams per device issues per tick (a 2160p -> 1080p pass is 28 us of device time, of the order of a launch).
extern "C" int chv_scale_lanczos_batch(chv_context *c, const chv_image *dsts, const chv_image *srcs, int n) {
    if (!ctx_ok(c)) return fail(CHV_ERR_INVALID_CONTEXT, "bad context");
    FLUSH_PENDING(c);
    if (n <= 0 || !dsts || !srcs) return fail(CHV_ERR_INVALID_VALUE, "empty batch");
    std::vector<DPlane> pairs((size_t)2 * n);
    DepScope deps;
    for (int i = 0; i < n; i++) {
        if (dsts[i].n_planes != 1) return fail(CHV_ERR_BAD_TARGET, "Lanczos target %d must be one 4-component plane", i);
        if (srcs[i].n_planes != 1) return fail(CHV_ERR_BAD_INPUT, "Lanczos source %d must be one 4-component plane", i);
        int rc = plane_to_device(dsts[i].planes[0], 4, c->device, &pairs[2 * i], CHV_ERR_BAD_TARGET, "target", i);
        if (rc) return rc;
        rc = plane_to_device(srcs[i].planes[0], 4, c->device, &pairs[2 * i + 1], CHV_ERR_BAD_INPUT, "input", i);
        if (rc) return rc;
        if (pairs[2 * i].w != pairs[0].w || pairs[2 * i].h != pairs[0].h || pairs[2 * i + 1].w != pairs[1].w || pairs[2 * i + 1].h != pairs[1].h)
            return fail(CHV_ERR_INVALID_VALUE, "pair %d: %dx%d -> %dx%d, the batch is %dx%d -> %dx%d (one geometry per batch)", i,
                        pairs[2 * i + 1].w, pairs[2 * i + 1].h, pairs[2 * i].w, pairs[2 * i].h, pairs[1].w, pairs[1].h, pairs[0].w, pairs[0].h);
    }
    HIP_TRY(hipSetDevice(c->device));
    auto dp = deps.deps();
    int rc = wait_for_uploads(c->stream, dp);
    if (rc) return rc;
    LanczosRef tx, ty;
    rc = lanczos_table(c, pairs[1].w, pairs[0].w, &tx);
    if (rc) return rc;
    rc = lanczos_table(c, pairs[1].h, pairs[0].h, &ty);
    if (rc) return rc;
    // the pairs travel through the pinned, device-mapped descriptor ring (a slot per chunk), like a transient tick's descriptors
    const int per_slot = CHV_LANCZOS_BATCH_CHUNK;
    static_assert((size_t)CHV_LANCZOS_BATCH_CHUNK * 2 * sizeof(DPlane) <= kDescSlotBytes, "a chunk's plane pairs must fit one descriptor slot");
    for (int first = 0; first < n; first += per_slot) {
        const int m = std::min(per_slot, n - first);
        DescSlot ds(c);
        if (ds.rc) return ds.rc;
        DPlane *host = (DPlane *)(c->desc_host + (size_t)ds.slot * kDescSlotBytes);
        memcpy(host, pairs.data() + 2 * (size_t)first, sizeof(DPlane) * 2 * (size_t)m);
        DPlane *dev = nullptr;
        HIP_TRY(hipHostGetDevicePointer((void **)&dev, host, 0));
        (void)hipGetLastError();
        hipError_t e = launch_lanczos(pairs[0], pairs[1], tx->first, tx->weights, tx->taps, ty->first, ty->weights, ty->taps, c->stream, dev, m, tx->stride, tx->first0, ty->stride);
        if (e != hipSuccess) return hip_fail(e, "lanczos launch");
    }
    return CHV_OK;
}

// ---------------------------------------------------------------------------
// events
// ---------------------------------------------------------------------------
extern "C" int chv_event_create(chv_context *c, chv_event **out) {
    if (!out) return fail(CHV_ERR_INVALID_VALUE, "null argument");
    *out = nullptr;
    if (!ctx_ok(c)) return fail(CHV_ERR_INVALID_CONTEXT, "bad context");
    HIP_TRY(hipSetDevice(c->device));
    hipEvent_t ev;
    HIP_TRY(hipEventCreate(&ev));
    chv_event *e = new chv_event;
    e->ev = ev; e->device = c->device;
    *out = e;
    return CHV_OK;
}
extern "C" int chv_event_record(chv_context *c, chv_event *ev) {
    if (!ctx_ok(c)) return fail(CHV_ERR_INVALID_CONTEXT, "bad context");
    FLUSH_PENDING(c);
    if (!ev || !ev->ev) return fail(CHV_ERR_INVALID_VALUE, "bad event");
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipEventRecord(ev->ev, c->stream));
    return CHV_OK;
}
extern "C" int chv_event_wait(chv_context *c, chv_event *ev) {
    if (!ctx_ok(c)) return fail(CHV_ERR_INVALID_CONTEXT, "bad context");
    FLUSH_PENDING(c);
    if (!ev || !ev->ev) return fail(CHV_ERR_INVALID_VALUE, "bad event");
    if (ev->device != c->device) return fail(CHV_ERR_INVALID_VALUE, "event belongs to device %d", ev->device);
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipStreamWaitEvent(c->stream, ev->ev, 0));
    return CHV_OK;
}
extern "C" int chv_event_synchronize(chv_event *ev) {
    if (!ev || !ev->ev) return fail(CHV_ERR_INVALID_VALUE, "bad event");
    HIP_TRY(hipSetDevice(ev->device));
    HIP_TRY(hipEventSynchronize(ev->ev));
    return CHV_OK;
}
extern "C" int chv_event_elapsed_ms(chv_event *a, chv_event *b, float *ms) {
    if (!a || !b || !ms) return fail(CHV_ERR_INVALID_VALUE, "null argument");
    HIP_TRY(hipSetDevice(a->device));
    HIP_TRY(hipEventElapsedTime(ms, a->ev, b->ev));
    return CHV_OK;
}
extern "C" int chv_event_destroy(chv_event *ev) {
    if (!ev) return fail(CHV_ERR_INVALID_VALUE, "null event");
    (void)hipSetDevice(ev->device);
    if (ev->ev) (void)hipEventDestroy(ev->ev);
    delete ev;
    return CHV_OK;
}
extern "C" int chv_device_synchronize(chv_context *c) {
    if (!ctx_ok(c)) return fail(CHV_ERR_INVALID_CONTEXT, "bad context");
    FLUSH_PENDING(c);
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipDeviceSynchronize());
    return CHV_OK;
}

// ---------------------------------------------------------------------------
// device self-test of the primitive conversions; not part of chipvideo.h
// (bound by tests/test_gpu_primitives.py only)
// ---------------------------------------------------------------------------
extern "C" int chv_selftest_primitives(chv_context *c, float *unorm_out /*256*/, const float *in_f,
                                       uint8_t *codes_out, const float *num, const float *den,
                                       float *quot_out, int n) {
    if (!ctx_ok(c)) return fail(CHV_ERR_INVALID_CONTEXT, "bad context");
    if (n < 0 || !unorm_out) return fail(CHV_ERR_INVALID_VALUE, "bad arguments");
    HIP_TRY(hipSetDevice(c->device));
    float *d_f = nullptr, *d_in = nullptr, *d_num = nullptr, *d_den = nullptr, *d_q = nullptr;
    uint8_t *d_c = nullptr;
    size_t m = n > 0 ? (size_t)n : 1;
    HIP_TRY(hipMalloc((void **)&d_f, 256 * sizeof(float)));
    HIP_TRY(hipMalloc((void **)&d_in, m * sizeof(float)));
    HIP_TRY(hipMalloc((void **)&d_num, m * sizeof(float)));
    HIP_TRY(hipMalloc((void **)&d_den, m * sizeof(float)));
    HIP_TRY(hipMalloc((void **)&d_q, m * sizeof(float)));
    HIP_TRY(hipMalloc((void **)&d_c, m));
    if (n > 0) {
        HIP_TRY(hipMemcpy(d_in, in_f, n * sizeof(float), hipMemcpyHostToDevice));
        HIP_TRY(hipMemcpy(d_num, num, n * sizeof(float), hipMemcpyHostToDevice));
        HIP_TRY(hipMemcpy(d_den, den, n * sizeof(float), hipMemcpyHostToDevice));
    }
    hipError_t e = launch_selftest(d_f, d_in, d_c, d_num, d_den, d_q, n, c->stream);
    if (e != hipSuccess) return hip_fail(e, "selftest launch");
    HIP_TRY(hipStreamSynchronize(c->stream));
    HIP_TRY(hipMemcpy(unorm_out, d_f, 256 * sizeof(float), hipMemcpyDeviceToHost));
    if (n > 0) {
        HIP_TRY(hipMemcpy(codes_out, d_c, n, hipMemcpyDeviceToHost));
        HIP_TRY(hipMemcpy(quot_out, d_q, n * sizeof(float), hipMemcpyDeviceToHost));
    }
    (void)hipFree(d_f); (void)hipFree(d_in); (void)hipFree(d_num); (void)hipFree(d_den); (void)hipFree(d_q); (void)hipFree(d_c);
    return CHV_OK;
}

extern "C" int chv_selftest_pack(chv_context *c, const int *b, const int *g, const int *r, uint32_t *out /*2n*/, int n) {
    if (!ctx_ok(c)) return fail(CHV_ERR_INVALID_CONTEXT, "bad context");
    if (n <= 0) return fail(CHV_ERR_INVALID_VALUE, "bad arguments");
    HIP_TRY(hipSetDevice(c->device));
    int *d_b = nullptr, *d_g = nullptr, *d_r = nullptr; uint32_t *d_o = nullptr;
    HIP_TRY(hipMalloc((void **)&d_b, n * 4)); HIP_TRY(hipMalloc((void **)&d_g, n * 4)); HIP_TRY(hipMalloc((void **)&d_r, n * 4));
    HIP_TRY(hipMalloc((void **)&d_o, n * 8));
    HIP_TRY(hipMemcpy(d_b, b, n * 4, hipMemcpyHostToDevice)); HIP_TRY(hipMemcpy(d_g, g, n * 4, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(d_r, r, n * 4, hipMemcpyHostToDevice));
    hipError_t e = launch_selftest_pack(d_b, d_g, d_r, d_o, n, c->stream);
    if (e != hipSuccess) return hip_fail(e, "selftest launch");
    HIP_TRY(hipStreamSynchronize(c->stream));
    HIP_TRY(hipMemcpy(out, d_o, n * 8, hipMemcpyDeviceToHost));
    (void)hipFree(d_b); (void)hipFree(d_g); (void)hipFree(d_r); (void)hipFree(d_o);
    return CHV_OK;
}

// Exhaustive device self-test of the integer BT.601 / 709 matrices (tests/test_gpu_matrices.py): direction 0 YUV -> RGB (out[Y << 16 | U << 8 | V]
// = the BGRA word), 1 RGB -> YUV (out[R << 16 | G << 8 | B] = Y | U << 8 | V << 16), all 2^24 triples of colourspace `csc`; *mismatches = triples
// on which the forms of the YUV -> RGB matrix the kernels use disagree among themselves (direction 0).
extern "C" int chv_selftest_matrices(chv_context *c, int direction, int csc, uint32_t *out /* 1 << 24 */, uint32_t *mismatches) {
    if (!ctx_ok(c)) return fail(CHV_ERR_INVALID_CONTEXT, "bad context");
    if (!out || !mismatches || direction < 0 || direction > 1 || csc < 0 || csc > 3) return fail(CHV_ERR_INVALID_VALUE, "bad arguments");
    HIP_TRY(hipSetDevice(c->device));
    uint32_t *d_o = nullptr, *d_m = nullptr;
    const size_t bytes = sizeof(uint32_t) << 24;
    HIP_TRY(hipMalloc((void **)&d_o, bytes));
    hipError_t e = hipMalloc((void **)&d_m, sizeof(uint32_t));
    if (e == hipSuccess) e = hipMemsetAsync(d_m, 0, sizeof(uint32_t), c->stream);
    if (e == hipSuccess) e = launch_selftest_matrices(direction, csc, d_o, d_m, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    if (e == hipSuccess) e = hipMemcpy(out, d_o, bytes, hipMemcpyDeviceToHost);
    if (e == hipSuccess) e = hipMemcpy(mismatches, d_m, sizeof(uint32_t), hipMemcpyDeviceToHost);
    (void)hipFree(d_o); if (d_m) (void)hipFree(d_m);
    return e == hipSuccess ? CHV_OK : hip_fail(e, "selftest matrices");
}

extern "C" int chv_selftest_pack_codes(chv_context *c, const float *in /*4n*/, uint32_t *out /*2n*/, int n) {
    if (!ctx_ok(c)) return fail(CHV_ERR_INVALID_CONTEXT, "bad context");
    if (n <= 0 || !in || !out) return fail(CHV_ERR_INVALID_VALUE, "bad arguments");
    HIP_TRY(hipSetDevice(c->device));
    float *d_in = nullptr; uint32_t *d_o = nullptr;
    HIP_TRY(hipMalloc((void **)&d_in, (size_t)n * 16));
    hipError_t e = hipMalloc((void **)&d_o, (size_t)n * 8);
    if (e == hipSuccess) e = hipMemcpy(d_in, in, (size_t)n * 16, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = launch_selftest_pack_codes(d_in, d_o, n, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    if (e == hipSuccess) e = hipMemcpy(out, d_o, (size_t)n * 8, hipMemcpyDeviceToHost);
    (void)hipFree(d_in); if (d_o) (void)hipFree(d_o);
    return e == hipSuccess ? CHV_OK : hip_fail(e, "selftest pack_codes");
}

// tests/stubhip/stub_runtime.cpp — the CPU stand-in behind tests/stubhip/hip/hip_runtime.h (see there).  TEST INFRASTRUCTURE.
#include <hip/hip_runtime.h>
#include <hip/hiprtc.h>

#include <atomic>
#include <cstdio>
#include <chrono>
#include <deque>
#include <map>
#include <mutex>
#include <set>
#include <string>
#include <vector>

namespace {
std::recursive_mutex g_mu;                    // one lock for the whole "device": operations of different streams execute one at a time
std::map<const void *, size_t> g_host, g_dev; // live pinned / device allocations (base -> size)
std::set<stubhip_stream *> g_streams;
thread_local int t_device = 0;
thread_local hipError_t t_last = hipSuccess;
std::atomic<int> g_fail_in{0};
std::atomic<long> g_executed{0};
int device_count() { const char *e = getenv("STUBHIP_DEVICES"); int n = e ? atoi(e) : 2; return n < 0 ? 0 : n; }
hipError_t err(hipError_t e) { if (e != hipSuccess) t_last = e; return e; }
}  // namespace

struct stubhip_stream {
    std::deque<std::function<void()>> q;
    unsigned long issued = 0, done = 0;       // operations enqueued / executed
    int device = 0;
};
struct stubhip_event {
    stubhip_stream *stream = nullptr;
    unsigned long seq = 0;                    // complete once stream->done >= seq
    bool recorded = false;
    double when = 0.0;
};
struct stubhip_function { std::string name; };
struct stubhip_module { std::string image; std::vector<stubhip_function *> fns; };
struct stubhip_program { std::string src, name, log; bool ok = false; };

static stubhip_stream g_null_stream;           // hipStream_t 0

static stubhip_stream *S(hipStream_t s) { return s ? s : &g_null_stream; }
// run queued operations of `s` until `upto` of them are done
static void drain(stubhip_stream *s, unsigned long upto) {
    std::lock_guard<std::recursive_mutex> lock(g_mu);
    while (s->done < upto && !s->q.empty()) {
        auto fn = std::move(s->q.front());
        s->q.pop_front();
        fn();
        s->done++;
        g_executed++;
    }
}

void stubhip_enqueue(hipStream_t st, std::function<void()> fn) {
    std::lock_guard<std::recursive_mutex> lock(g_mu);
    stubhip_stream *s = S(st);
    s->q.push_back(std::move(fn));
    s->issued++;
    if (s->q.size() > 2048) drain(s, s->issued - 1024);       // a device does make progress on its own — but far later than a 64-slot ring wraps
}
void stubhip_fail_launch_after(int n) { g_fail_in = n; }
static std::atomic<long> g_launches{0};
long stubhip_launches() { return g_launches.load(); }         // every launcher asks stubhip_launch_should_fail() exactly once
bool stubhip_launch_should_fail() {
    g_launches++;
    int v = g_fail_in.load();
    while (v > 0) { if (g_fail_in.compare_exchange_weak(v, v - 1)) return v == 1; }
    return false;
}
long stubhip_ops_executed() { return g_executed.load(); }

hipError_t hipGetDeviceCount(int *n) { *n = device_count(); return *n > 0 ? hipSuccess : err(hipErrorNoDevice); }
hipError_t hipGetDeviceProperties(hipDeviceProp_t *p, int dev) {
    if (dev < 0 || dev >= device_count()) return err(hipErrorInvalidDevice);
    memset(p, 0, sizeof *p);
    snprintf(p->name, sizeof p->name, "stub MI355X #%d", dev);
    const char *arch = getenv("STUBHIP_ARCH");
    snprintf(p->gcnArchName, sizeof p->gcnArchName, "%s", arch ? arch : "gfx950:sramecc+:xnack-");
    p->totalGlobalMem = (size_t)288 << 30; p->multiProcessorCount = 256; p->pciBusID = 5 + dev;
    return hipSuccess;
}
hipError_t hipDeviceGetPCIBusId(char *buf, int len, int dev) { snprintf(buf, (size_t)len, "0000:%02x:00.0", 5 + dev); return hipSuccess; }
hipError_t hipSetDevice(int dev) { if (dev < 0 || dev >= device_count()) return err(hipErrorInvalidDevice); t_device = dev; return hipSuccess; }
hipError_t hipGetLastError() { hipError_t e = t_last; t_last = hipSuccess; return e; }
const char *hipGetErrorString(hipError_t e) { return e == hipSuccess ? "no error" : e == hipErrorLaunchFailure ? "launch failure (injected)" : "stub error"; }

hipError_t hipMalloc(void **p, size_t n) {
    *p = nullptr;
    if (n == 0) return hipSuccess;
    if (n > ((size_t)1 << 34)) return err(hipErrorOutOfMemory);
    *p = malloc(n);
    if (!*p) return err(hipErrorOutOfMemory);
    memset(*p, 0xCD, n);
    std::lock_guard<std::recursive_mutex> lock(g_mu);
    g_dev[*p] = n;
    return hipSuccess;
}
static void drain_everything() {
    std::lock_guard<std::recursive_mutex> lock(g_mu);
    drain(&g_null_stream, g_null_stream.issued);
    for (stubhip_stream *s : g_streams) drain(s, s->issued);
}
hipError_t hipFree(void *p) {
    if (!p) return hipSuccess;
    std::lock_guard<std::recursive_mutex> lock(g_mu);
    drain_everything();                        // hipFree synchronises the device
    if (!g_dev.erase(p)) return err(hipErrorInvalidValue);
    free(p);
    return hipSuccess;
}
hipError_t hipHostMalloc(void **p, size_t n, unsigned) {
    *p = malloc(n ? n : 1);
    if (!*p) return err(hipErrorOutOfMemory);
    std::lock_guard<std::recursive_mutex> lock(g_mu);
    g_host[*p] = n;
    return hipSuccess;
}
hipError_t hipHostFree(void *p) {
    std::lock_guard<std::recursive_mutex> lock(g_mu);
    drain_everything();
    if (!g_host.erase(p)) return err(hipErrorInvalidValue);
    free(p);
    return hipSuccess;
}
hipError_t hipHostGetDevicePointer(void **dev, void *host, unsigned) { *dev = host; return hipSuccess; }
hipError_t hipPointerGetAttributes(hipPointerAttribute_t *a, const void *p) {
    std::lock_guard<std::recursive_mutex> lock(g_mu);
    for (auto *m : { &g_host, &g_dev }) {
        auto it = m->upper_bound(p);
        if (it == m->begin()) continue;
        --it;
        if ((const char *)p < (const char *)it->first + it->second) {
            a->type = m == &g_host ? hipMemoryTypeHost : hipMemoryTypeDevice; a->device = 0;
            a->devicePointer = (void *)p; a->hostPointer = m == &g_host ? (void *)p : nullptr;
            return hipSuccess;
        }
    }
    return err(hipErrorInvalidValue);
}
hipError_t hipMemcpy(void *d, const void *s, size_t n, hipMemcpyKind) {
    drain_everything();                        // (blocking copies run behind everything queued, as on the null stream)
    if (n) memcpy(d, s, n);
    return hipSuccess;
}
hipError_t hipMemcpyAsync(void *d, const void *s, size_t n, hipMemcpyKind k, hipStream_t st) {
    // pageable host memory is staged by the real runtime before the call returns: copy now; pinned / device memory is read LATER
    bool pinned = true;
    if (k == hipMemcpyHostToDevice) { hipPointerAttribute_t a; pinned = hipPointerGetAttributes(&a, s) == hipSuccess; t_last = hipSuccess; }
    if (!pinned) { std::vector<char> tmp((const char *)s, (const char *)s + n); stubhip_enqueue(st, [d, tmp] { if (!tmp.empty()) memcpy(d, tmp.data(), tmp.size()); }); return hipSuccess; }
    stubhip_enqueue(st, [d, s, n] { if (n) memcpy(d, s, n); });
    return hipSuccess;
}
hipError_t hipMemcpy2DAsync(void *d, size_t dp, const void *s, size_t sp, size_t w, size_t h, hipMemcpyKind k, hipStream_t st) {
    if (dp < w || sp < w) return err(hipErrorInvalidValue);
    bool pinned = true;
    if (k == hipMemcpyHostToDevice) { hipPointerAttribute_t a; pinned = hipPointerGetAttributes(&a, s) == hipSuccess; t_last = hipSuccess; }
    if (!pinned) {
        std::vector<char> tmp(w * h);
        for (size_t r = 0; r < h; r++) memcpy(tmp.data() + r * w, (const char *)s + r * sp, w);
        stubhip_enqueue(st, [d, dp, tmp, w, h] { for (size_t r = 0; r < h; r++) memcpy((char *)d + r * dp, tmp.data() + r * w, w); });
        return hipSuccess;
    }
    stubhip_enqueue(st, [=] { for (size_t r = 0; r < h; r++) memcpy((char *)d + r * dp, (const char *)s + r * sp, w); });
    return hipSuccess;
}
hipError_t hipMemsetAsync(void *d, int v, size_t n, hipStream_t st) { stubhip_enqueue(st, [=] { memset(d, v, n); }); return hipSuccess; }

hipError_t hipStreamCreateWithFlags(hipStream_t *s, unsigned) {
    *s = new stubhip_stream; (*s)->device = t_device;
    std::lock_guard<std::recursive_mutex> lock(g_mu);
    g_streams.insert(*s);
    return hipSuccess;
}
hipError_t hipStreamDestroy(hipStream_t s) {
    if (!s) return err(hipErrorInvalidValue);
    std::lock_guard<std::recursive_mutex> lock(g_mu);
    drain(s, s->issued);
    g_streams.erase(s);
    delete s;
    return hipSuccess;
}
hipError_t hipStreamSynchronize(hipStream_t s) { std::lock_guard<std::recursive_mutex> lock(g_mu); drain(S(s), S(s)->issued); return hipSuccess; }
hipError_t hipStreamQuery(hipStream_t s) {
    std::lock_guard<std::recursive_mutex> lock(g_mu);
    stubhip_stream *t = S(s);
    if (t->done < t->issued) drain(t, t->done + 2);            // polling sees progress
    if (t->done < t->issued) { return hipErrorNotReady; }
    return hipSuccess;
}
hipError_t hipDeviceSynchronize() { drain_everything(); return hipSuccess; }

static double now_ms() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
hipError_t hipEventCreate(hipEvent_t *e) { *e = new stubhip_event; return hipSuccess; }
hipError_t hipEventCreateWithFlags(hipEvent_t *e, unsigned) { return hipEventCreate(e); }
hipError_t hipEventDestroy(hipEvent_t e) { if (!e) return err(hipErrorInvalidValue); std::lock_guard<std::recursive_mutex> lock(g_mu); delete e; return hipSuccess; }
hipError_t hipEventRecord(hipEvent_t e, hipStream_t s) {
    if (!e) return err(hipErrorInvalidValue);
    std::lock_guard<std::recursive_mutex> lock(g_mu);
    stubhip_stream *t = S(s);
    stubhip_enqueue(s, [e] { e->when = now_ms(); });
    e->stream = t; e->seq = t->issued; e->recorded = true;
    return hipSuccess;
}
static bool event_done(stubhip_event *e) { return !e->recorded || !g_streams.count(e->stream) ? true : e->stream->done >= e->seq; }
hipError_t hipEventSynchronize(hipEvent_t e) {
    if (!e) return err(hipErrorInvalidValue);
    std::lock_guard<std::recursive_mutex> lock(g_mu);
    if (e->recorded && (e->stream == &g_null_stream || g_streams.count(e->stream))) drain(e->stream, e->seq);
    return hipSuccess;
}
hipError_t hipEventQuery(hipEvent_t e) {
    if (!e) return err(hipErrorInvalidValue);
    std::lock_guard<std::recursive_mutex> lock(g_mu);
    if (e->recorded && e->stream != &g_null_stream && !g_streams.count(e->stream)) return hipSuccess;       // (its stream is gone: everything ran)
    if (e->recorded && e->stream->done < e->seq) drain(e->stream, e->stream->done + 2);
    return (e->recorded && e->stream->done < e->seq) ? hipErrorNotReady : hipSuccess;
}
hipError_t hipStreamWaitEvent(hipStream_t s, hipEvent_t e, unsigned) {
    if (!e) return err(hipErrorInvalidValue);
    std::lock_guard<std::recursive_mutex> lock(g_mu);
    if (!e->recorded) return hipSuccess;
    stubhip_stream *src = e->stream;
    const unsigned long seq = e->seq;                           // (the wait is for the record made BEFORE this call, whatever happens to the event later)
    stubhip_enqueue(s, [src, seq] { if (src == &g_null_stream || g_streams.count(src)) drain(src, seq); });
    return hipSuccess;
}
hipError_t hipEventElapsedTime(float *ms, hipEvent_t a, hipEvent_t b) {
    if (!a || !b) return err(hipErrorInvalidValue);
    std::lock_guard<std::recursive_mutex> lock(g_mu);
    if (!event_done(a) || !event_done(b)) return err(hipErrorNotReady);
    *ms = (float)(b->when - a->when);
    return hipSuccess;
}

hipError_t hipModuleLoadData(hipModule_t *m, const void *image) { *m = new stubhip_module{ std::string((const char *)image), {} }; return hipSuccess; }
hipError_t hipModuleUnload(hipModule_t m) { for (auto *f : m->fns) delete f; delete m; return hipSuccess; }
hipError_t hipModuleGetFunction(hipFunction_t *f, hipModule_t m, const char *name) {
    if (m->image.find(name) == std::string::npos) return err(hipErrorInvalidValue);
    *f = new stubhip_function{ name };          // (owned by its module)
    m->fns.push_back(*f);
    return hipSuccess;
}
hipError_t hipModuleLaunchKernel(hipFunction_t, unsigned, unsigned, unsigned, unsigned, unsigned, unsigned, unsigned, hipStream_t st, void **, void **extra) {
    if (stubhip_launch_should_fail()) return err(hipErrorLaunchFailure);
    // the argument block is copied at launch, like kernel arguments
    std::vector<char> args;
    if (extra && extra[0] == HIP_LAUNCH_PARAM_BUFFER_POINTER) { size_t n = *(size_t *)extra[3]; args.assign((char *)extra[1], (char *)extra[1] + n); }
    stubhip_enqueue(st, [args] { (void)args; });
    return hipSuccess;
}

hiprtcResult hiprtcCreateProgram(hiprtcProgram *p, const char *src, const char *name, int, const char **, const char **) {
    *p = new stubhip_program; (*p)->src = src ? src : ""; (*p)->name = name ? name : ""; return HIPRTC_SUCCESS;
}
hiprtcResult hiprtcCompileProgram(hiprtcProgram p, int, const char **) {
    p->ok = p->src.find("__global__") != std::string::npos && p->src.find("#error") == std::string::npos;
    p->log = p->ok ? "" : "stub hiprtc: no __global__ function in the source";
    return p->ok ? HIPRTC_SUCCESS : HIPRTC_ERROR_COMPILATION;
}
hiprtcResult hiprtcGetProgramLogSize(hiprtcProgram p, size_t *n) { *n = p->log.size() + 1; return HIPRTC_SUCCESS; }
hiprtcResult hiprtcGetProgramLog(hiprtcProgram p, char *log) { memcpy(log, p->log.c_str(), p->log.size() + 1); return HIPRTC_SUCCESS; }
hiprtcResult hiprtcGetCodeSize(hiprtcProgram p, size_t *n) { *n = p->src.size() + 1; return HIPRTC_SUCCESS; }
hiprtcResult hiprtcGetCode(hiprtcProgram p, char *code) { memcpy(code, p->src.c_str(), p->src.size() + 1); return HIPRTC_SUCCESS; }
hiprtcResult hiprtcDestroyProgram(hiprtcProgram *p) { delete *p; *p = nullptr; return HIPRTC_SUCCESS; }
const char *hiprtcGetErrorString(hiprtcResult r) { return r == HIPRTC_SUCCESS ? "HIPRTC_SUCCESS" : "HIPRTC_ERROR_COMPILATION"; }

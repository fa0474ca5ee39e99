// tests/stubhip/hip/hip_runtime.h — a CPU stand-in for the part of the HIP runtime that swiftvideo_amd/csrc/chipvideo.cpp (the host side of
// libchipvideo.so) uses, so that the C ABI's host logic — contexts, buffer lifetimes, the descriptor ring, upload / download ordering, batches,
// error paths — can run under AddressSanitizer, UBSan and ThreadSanitizer on a machine without a GPU (tests/test_sanitizers.py).
// TEST INFRASTRUCTURE: nothing under swiftvideo_amd/ includes it; the product is built against /opt/rocm.
//
// Model: a stream is an in-order queue of closures executed LAZILY — at synchronisation points, when another stream waits for one of its
// events, or a few at a time when the host polls (hipEventQuery / hipStreamQuery): work really is outstanding when an asynchronous call
// returns, so a descriptor slot, a staging buffer or a device allocation that the host recycles too early is read after the fact by the
// deferred operation — and the sanitizer (or the launcher stub's descriptor check, stub_launchers.cpp) sees it.
#pragma once
#include <cstddef>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <functional>

typedef int hipError_t;
enum { hipSuccess = 0, hipErrorInvalidValue = 1, hipErrorOutOfMemory = 2, hipErrorInvalidDevice = 101, hipErrorNoDevice = 100,
       hipErrorInvalidContext = 201, hipErrorNotReady = 600, hipErrorNotSupported = 801, hipErrorLaunchFailure = 719 };
enum hipMemcpyKind { hipMemcpyHostToHost = 0, hipMemcpyHostToDevice = 1, hipMemcpyDeviceToHost = 2, hipMemcpyDeviceToDevice = 3 };
enum { hipHostMallocDefault = 0, hipHostMallocMapped = 2, hipStreamNonBlocking = 1, hipEventDisableTiming = 2 };
enum hipMemoryType { hipMemoryTypeUnregistered = 0, hipMemoryTypeHost = 1, hipMemoryTypeDevice = 2 };

struct stubhip_stream;
struct stubhip_event;
typedef stubhip_stream *hipStream_t;
typedef stubhip_event *hipEvent_t;
typedef struct stubhip_module *hipModule_t;
typedef struct stubhip_function *hipFunction_t;

struct hipDeviceProp_t {
    char name[256];
    char gcnArchName[256];
    size_t totalGlobalMem;
    int multiProcessorCount;
    int pciBusID, pciDeviceID, pciDomainID;
};
struct hipPointerAttribute_t { hipMemoryType type; int device; void *devicePointer; void *hostPointer; };
struct dim3 { unsigned x, y, z; dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {} };

hipError_t hipGetDeviceCount(int *n);
hipError_t hipGetDeviceProperties(hipDeviceProp_t *p, int dev);
hipError_t hipDeviceGetPCIBusId(char *buf, int len, int dev);
hipError_t hipSetDevice(int dev);
hipError_t hipGetLastError();
const char *hipGetErrorString(hipError_t e);
hipError_t hipMalloc(void **p, size_t n);
template <class T> hipError_t hipMalloc(T **p, size_t n) { return hipMalloc((void **)p, n); }
hipError_t hipFree(void *p);
hipError_t hipHostMalloc(void **p, size_t n, unsigned flags);
template <class T> hipError_t hipHostMalloc(T **p, size_t n, unsigned flags) { return hipHostMalloc((void **)p, n, flags); }
hipError_t hipHostFree(void *p);
hipError_t hipHostGetDevicePointer(void **dev, void *host, unsigned flags);
template <class T> hipError_t hipHostGetDevicePointer(T **dev, void *host, unsigned flags) { return hipHostGetDevicePointer((void **)dev, host, flags); }
hipError_t hipPointerGetAttributes(hipPointerAttribute_t *a, const void *p);
hipError_t hipMemcpy(void *d, const void *s, size_t n, hipMemcpyKind k);
hipError_t hipMemcpyAsync(void *d, const void *s, size_t n, hipMemcpyKind k, hipStream_t st);
hipError_t hipMemcpy2DAsync(void *d, size_t dp, const void *s, size_t sp, size_t w, size_t h, hipMemcpyKind k, hipStream_t st);
hipError_t hipMemsetAsync(void *d, int v, size_t n, hipStream_t st);
hipError_t hipStreamCreateWithFlags(hipStream_t *s, unsigned flags);
hipError_t hipStreamDestroy(hipStream_t s);
hipError_t hipStreamSynchronize(hipStream_t s);
hipError_t hipStreamQuery(hipStream_t s);
hipError_t hipStreamWaitEvent(hipStream_t s, hipEvent_t e, unsigned flags);
hipError_t hipDeviceSynchronize();
hipError_t hipEventCreate(hipEvent_t *e);
hipError_t hipEventCreateWithFlags(hipEvent_t *e, unsigned flags);
hipError_t hipEventDestroy(hipEvent_t e);
hipError_t hipEventRecord(hipEvent_t e, hipStream_t s);
hipError_t hipEventSynchronize(hipEvent_t e);
hipError_t hipEventQuery(hipEvent_t e);
hipError_t hipEventElapsedTime(float *ms, hipEvent_t a, hipEvent_t b);
hipError_t hipModuleLoadData(hipModule_t *m, const void *image);
hipError_t hipModuleUnload(hipModule_t m);
hipError_t hipModuleGetFunction(hipFunction_t *f, hipModule_t m, const char *name);
hipError_t hipModuleLaunchKernel(hipFunction_t f, unsigned gx, unsigned gy, unsigned gz, unsigned bx, unsigned by, unsigned bz, unsigned shmem,
                                 hipStream_t st, void **params, void **extra);

// ---- test hooks (not HIP) ----
// enqueue a closure on a stream (the launcher stubs' "kernels"); executed in stream order like a copy
void stubhip_enqueue(hipStream_t st, std::function<void()> fn);
// the n-th launcher call from now on (1 = the next) reports hipErrorLaunchFailure instead of enqueuing; 0 = never
void stubhip_fail_launch_after(int n);
bool stubhip_launch_should_fail();
long stubhip_launches();
long stubhip_ops_executed();
#define HIP_LAUNCH_PARAM_BUFFER_POINTER ((void *)0x01)
#define HIP_LAUNCH_PARAM_BUFFER_SIZE ((void *)0x02)
#define HIP_LAUNCH_PARAM_END ((void *)0x03)

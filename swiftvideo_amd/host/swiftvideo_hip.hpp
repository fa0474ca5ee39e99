// swiftvideo_hip.hpp — C++17 host-side mirror of SwiftVideo's compute/picture operator
// surface over the CHIPVideo C ABI (include/chipvideo.h).
//
// The reference is compiled Swift; with no Swift toolchain available, this header is the
// native spelling of the backend contract (compute.cl.swift:36-499) and of the operators
// that sit directly on it (compute.swift, mix.video.swift, sample.pict.linux.swift): same
// names, argument meaning and error behaviour, so that native callers and tests read like
// the reference's call sites:
//
//     auto ctx  = sv::makeComputeContext(sv::ComputeDeviceType::GPU);
//     auto gpu  = sv::uploadComputePicture(ctx, pict);
//     ctx = sv::usingContext(ctx, [&](sv::ComputeContext c) {
//         return sv::applyComputeImage(c, gpu, backing, sv::ComputeKernel::img_bgra_nv12); });
//     auto out  = sv::downloadComputePicture(ctx, backing);
//
// Header-only; link with -lchipvideo.  No pixel is touched on the host.
#pragma once

#include <algorithm>
#include <array>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <functional>
#include <map>
#include <memory>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#include "../../include/chipvideo.h"

namespace sv {

// ---- ComputeError (compute.swift:22-39) -------------------------------------------------------
struct ComputeError : std::runtime_error {
    int status;              // chv_status
    std::string caseName;    // Swift case name
    ComputeError(int s, const std::string &detail)
        : std::runtime_error(std::string("ComputeError.") + chv_error_string(s) + (detail.empty() ? "" : ": " + detail)),
          status(s), caseName(chv_error_string(s)) {}
};
inline void check(int status) {
    if (status != CHV_OK) throw ComputeError(status, chv_last_error_detail());
}

enum class ComputeDeviceType { GPU, CPU, Accelerator, Default };   // compute.swift:41-46

// ---- ComputeKernel (compute.swift:49-74) ---------------------------------------------------------
enum class ComputeKernel : int {
    img_nv12_nv12 = CHV_K_IMG_NV12_NV12, img_bgra_nv12 = CHV_K_IMG_BGRA_NV12, img_rgba_nv12 = CHV_K_IMG_RGBA_NV12,
    img_bgra_bgra = CHV_K_IMG_BGRA_BGRA, img_y420p_y420p = CHV_K_IMG_Y420P_Y420P, img_y420p_nv12 = CHV_K_IMG_Y420P_NV12,
    img_clear_nv12 = CHV_K_IMG_CLEAR_NV12, img_clear_yuvs = CHV_K_IMG_CLEAR_YUVS, img_clear_bgra = CHV_K_IMG_CLEAR_BGRA,
    img_clear_y420p = CHV_K_IMG_CLEAR_Y420P, img_clear_rgba = CHV_K_IMG_CLEAR_RGBA, img_rgba_y420p = CHV_K_IMG_RGBA_Y420P,
    img_bgra_y420p = CHV_K_IMG_BGRA_Y420P, snd_s16i_s16i = CHV_K_SND_S16I_S16I, me_fullsearch = CHV_K_ME_FULLSEARCH,
    img_nv12_bgra = CHV_K_IMG_NV12_BGRA, img_y420p_bgra = CHV_K_IMG_Y420P_BGRA,
    img_bgra_bgra_tx = CHV_K_IMG_BGRA_BGRA_TX, img_rgba_bgra_tx = CHV_K_IMG_RGBA_BGRA_TX,
    // integer BT.601/709 RGB -> YUV onto 4:2:0 canvases
    img_bgra_nv12_int = CHV_K_IMG_BGRA_NV12_INT, img_rgba_nv12_int = CHV_K_IMG_RGBA_NV12_INT,
    img_bgra_y420p_int = CHV_K_IMG_BGRA_Y420P_INT, img_rgba_y420p_int = CHV_K_IMG_RGBA_Y420P_INT
};
inline std::string describing(ComputeKernel k) {           // String(describing:)
    const char *n = chv_kernel_name((int)k);
    return n ? n : "";
}
// compute.swift:90-110; throws ComputeError.invalidValue
inline ComputeKernel defaultComputeKernelFromString(const std::string &str) {
    int k = -1;
    check(chv_kernel_from_string(str.c_str(), &k));
    return (ComputeKernel)k;
}

// ---- pixel formats and planes (sample.pict.swift:20-58) --------------------------------------
enum class PixelFormat : int {
    nv12 = CHV_FMT_NV12, nv21 = CHV_FMT_NV21, yuvs = CHV_FMT_YUVS, zvuy = CHV_FMT_ZVUY, y420p = CHV_FMT_Y420P,
    y422p = CHV_FMT_Y422P, y444p = CHV_FMT_Y444P, RGBA = CHV_FMT_RGBA, BGRA = CHV_FMT_BGRA, invalid = CHV_FMT_INVALID
};
inline std::string lowercasedName(PixelFormat f) {        // String(describing:).lowercased(), mix.video.swift:143-144
    switch (f) {
    case PixelFormat::nv12: return "nv12"; case PixelFormat::nv21: return "nv21"; case PixelFormat::yuvs: return "yuvs";
    case PixelFormat::zvuy: return "zvuy"; case PixelFormat::y420p: return "y420p"; case PixelFormat::y422p: return "y422p";
    case PixelFormat::y444p: return "y444p"; case PixelFormat::RGBA: return "rgba"; case PixelFormat::BGRA: return "bgra";
    default: return "invalid";
    }
}
enum class BufferType { shared, cpu, gpu, invalid };
struct Vector2 { float x = 0, y = 0; };
struct Vector4 { float x = 0, y = 0, z = 0, w = 0; };
struct Plane {
    Vector2 size; int stride = 0; int bitDepth = 8; int components = 1;
};

// ---- Matrix4: what applyComputeImage needs from VectorMath (compute.swift:151-155) ---------------
// Column-vector convention: (M v)_i = sum_j m[i][j] v_j; translation in m[i][3].
struct Matrix4 {
    std::array<double, 16> m{ 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1 };
    static Matrix4 identity() { return Matrix4(); }
    static Matrix4 translation(double x, double y, double z = 0) { Matrix4 r; r.m[3] = x; r.m[7] = y; r.m[11] = z; return r; }
    static Matrix4 scale(double x, double y, double z = 1) { Matrix4 r; r.m[0] = x; r.m[5] = y; r.m[10] = z; return r; }
    static Matrix4 rotationZ(double t) { Matrix4 r; r.m[0] = std::cos(t); r.m[1] = -std::sin(t); r.m[4] = std::sin(t); r.m[5] = std::cos(t); return r; }
    // Matrix4(ortho), animator.pic.swift:326-332: canvas pixels -> NDC
    static Matrix4 ortho(double w, double h) { Matrix4 r; r.m[0] = 2 / w; r.m[5] = 2 / h; r.m[3] = -1; r.m[7] = -1; r.m[11] = 1; return r; }
    Matrix4 operator*(const Matrix4 &b) const {
        Matrix4 r;
        for (int i = 0; i < 4; i++) for (int j = 0; j < 4; j++) {
            double s = 0;
            for (int k = 0; k < 4; k++) s += m[i * 4 + k] * b.m[k * 4 + j];
            r.m[i * 4 + j] = s;
        }
        return r;
    }
    Matrix4 inverse() const {            // Gauss-Jordan with partial pivoting
        double a[4][8];
        for (int i = 0; i < 4; i++) for (int j = 0; j < 4; j++) { a[i][j] = m[i * 4 + j]; a[i][j + 4] = i == j ? 1.0 : 0.0; }
        for (int c = 0; c < 4; c++) {
            int p = c;
            for (int r = c + 1; r < 4; r++) if (std::fabs(a[r][c]) > std::fabs(a[p][c])) p = r;
            if (a[p][c] == 0.0) throw ComputeError(CHV_ERR_INVALID_VALUE, "singular matrix");
            if (p != c) for (int j = 0; j < 8; j++) std::swap(a[p][j], a[c][j]);
            double d = a[c][c];
            for (int j = 0; j < 8; j++) a[c][j] /= d;
            for (int r = 0; r < 4; r++) if (r != c) {
                double f = a[r][c];
                if (f != 0.0) for (int j = 0; j < 8; j++) a[r][j] -= f * a[c][j];
            }
        }
        Matrix4 r;
        for (int i = 0; i < 4; i++) for (int j = 0; j < 4; j++) r.m[i * 4 + j] = a[i][j + 4];
        return r;
    }
    // the 16 floats a kernel reads for this matrix: row i of M^-1 at [4i..4i+3]
    // (VectorMath's M.inverse.transpose stored column-major is exactly that)
    void kernelRows(float out[16]) const {
        Matrix4 inv = inverse();
        for (int i = 0; i < 16; i++) out[i] = (float)inv.m[i];
    }
};

// ---- devices / contexts (compute.cl.swift:36-151) --------------------------------------------------
struct ComputeDevice {
    int deviceId = 0; bool available = false; ComputeDeviceType deviceType = ComputeDeviceType::GPU;
    int vendorId = 0; std::string vendorName, arch; bool supportsImages = false;
};
inline std::vector<ComputeDevice> availableComputeDevices() {
    std::vector<ComputeDevice> out;
    int n = 0;
    if (chv_device_count(&n) != CHV_OK) return out;
    for (int i = 0; i < n; i++) {
        chv_device_info info;
        if (chv_device_info_get(i, &info) != CHV_OK) continue;
        ComputeDevice d;
        d.deviceId = i; d.available = info.available != 0; d.vendorId = info.vendor_id; d.vendorName = info.name;
        d.arch = info.arch; d.supportsImages = info.supports_images != 0;
        out.push_back(d);
    }
    return out;
}

// value type holding a shared handle, like the Swift struct (compute.cl.swift:75-105)
struct ComputeContext {
    std::shared_ptr<chv_context> handle;
    ComputeDevice device;
    chv_context *get() const {
        if (!handle) throw ComputeError(CHV_ERR_INVALID_CONTEXT, "context destroyed");
        return handle.get();
    }
};
inline ComputeContext wrapContext(chv_context *c, const ComputeDevice &d) {
    // destroyComputeContext() releases explicitly; the deleter covers contexts that are just dropped
    return ComputeContext{ std::shared_ptr<chv_context>(c, [](chv_context *p) { if (p) chv_context_destroy(p); }), d };
}
inline ComputeContext createComputeContext(const ComputeDevice &device) {
    chv_context *c = nullptr;
    check(chv_context_create(device.deviceId, &c));
    return wrapContext(c, device);
}
inline ComputeContext createComputeContext(const ComputeContext &sharing) {      // createComputeContext(sharing:)
    chv_context *c = nullptr;
    check(chv_context_share(sharing.get(), &c));
    return wrapContext(c, sharing.device);
}
inline bool hasAvailableComputeDevices(ComputeDeviceType t) {
    for (auto &d : availableComputeDevices()) if (d.deviceType == t && d.available) return true;
    return false;
}
inline ComputeContext makeComputeContext(ComputeDeviceType forType, int index = 0) {   // compute.swift:121-129
    int seen = 0;
    for (auto &d : availableComputeDevices())
        if (d.deviceType == forType && d.available && seen++ == index) return createComputeContext(d);
    throw ComputeError(CHV_ERR_DEVICE_NOT_AVAILABLE, "no available compute device of the requested type");
}
inline void destroyComputeContext(ComputeContext &ctx) { ctx.handle.reset(); }
inline ComputeContext beginComputePass(ComputeContext ctx) { check(chv_pass_begin(ctx.get())); return ctx; }
inline ComputeContext endComputePass(ComputeContext ctx, bool waitForCompletion) {
    check(chv_pass_end(ctx.get(), waitForCompletion ? 1 : 0));
    return ctx;
}
// compute.swift:131-134.  (When `fun` throws, the bracket is closed on the way out: the library holds a pass's kernels until its end.)
inline ComputeContext usingContext(ComputeContext ctx, const std::function<ComputeContext(ComputeContext)> &fun) {
    ComputeContext c = beginComputePass(ctx);
    try {
        c = fun(c);
    } catch (...) {
        (void)chv_pass_end(ctx.get(), 0);
        throw;
    }
    return endComputePass(c, true);
}

// ---- buffers and samples (compute.cl.swift:46-58, sample.pict.linux.swift:23-311) ---------------
struct ComputeBuffer {
    chv_buffer *handle = nullptr; size_t size = 0; size_t pitch = 0;
    ComputeBuffer(chv_buffer *h, size_t s, size_t p) : handle(h), size(s), pitch(p) {}
    ComputeBuffer(const ComputeBuffer &) = delete;
    ~ComputeBuffer() { if (handle) chv_buffer_free(handle); }
};
using ComputeBufferRef = std::shared_ptr<ComputeBuffer>;
using Data = std::vector<uint8_t>;

struct ImageBuffer {
    PixelFormat pixelFormat = PixelFormat::invalid;
    BufferType bufferType = BufferType::invalid;
    Vector2 size;
    std::vector<ComputeBufferRef> computeTextures;
    std::vector<std::shared_ptr<Data>> buffers;     // one per plane, stride * rows bytes
    std::vector<Plane> planes;
    // HIP backend: the planes of one picture are regions of ONE device allocation (computeTextures holds the same
    // buffer for every plane) at these byte offsets / pitches, so that a picture whose host planes are contiguous
    // (sample.pict.linux.swift:296-311) is uploaded with one copy instead of one per plane
    std::vector<size_t> gpuOffsets, gpuPitches;
};

// sample.pict.linux.swift:275-294
inline std::vector<Plane> planesForFormat(PixelFormat f, Vector2 size) {
    int w = (int)size.x;
    Vector2 half{ (float)((int)size.x / 2), (float)((int)size.y / 2) };
    switch (f) {
    case PixelFormat::nv12: return { Plane{ size, w, 8, 1 }, Plane{ half, w, 8, 2 } };
    case PixelFormat::BGRA: case PixelFormat::RGBA: return { Plane{ size, w * 4, 8, 4 } };
    case PixelFormat::y420p: return { Plane{ size, w, 8, 1 }, Plane{ half, w / 2, 8, 1 }, Plane{ half, w / 2, 8, 1 } };
    default: throw ComputeError(CHV_ERR_BAD_INPUT, "Invalid pixel format");
    }
}

struct PictureSample {
    std::shared_ptr<ImageBuffer> img;
    Matrix4 matrix, textureMatrix, borderMatrix;
    Vector4 fillColor; float opacity = 1.0f; int zIndex = 0;
    std::string assetId, workspaceId, revision;
    double time = 0, pts = 0;
    PixelFormat pixelFormat() const { return img->pixelFormat; }
    BufferType bufferType() const { return img->bufferType; }
    Vector2 size() const { return img->size; }
};

// sample.pict.linux.swift:254-273
inline PictureSample createPictureSample(Vector2 size, PixelFormat format, const std::string &assetId = "",
                                         const std::string &workspaceId = "") {
    if (!(size.x > 0 && size.y > 0)) throw ComputeError(CHV_ERR_INVALID_OPERATION, "size must be positive");
    auto img = std::make_shared<ImageBuffer>();
    img->pixelFormat = format; img->bufferType = BufferType::cpu; img->size = size;
    img->planes = planesForFormat(format, size);
    for (auto &p : img->planes) img->buffers.push_back(std::make_shared<Data>((size_t)p.stride * std::max(1, (int)p.size.y), 0));
    PictureSample s;
    s.img = img; s.assetId = assetId; s.workspaceId = workspaceId; s.borderMatrix = s.matrix;
    return s;
}

inline int planeComponents(const Plane &p) { return p.components >= 3 ? 4 : p.components; }

// compute.cl.swift:421-459
inline PictureSample uploadComputePicture(const ComputeContext &ctx, const PictureSample &pict, int maxPlanes = 3,
                                          bool retainCpuBuffer = true, bool asynchronous = false) {
    if (pict.bufferType() != BufferType::cpu) return pict;
    const ImageBuffer &image = *pict.img;
    size_t n = image.planes.size();
    if (!(n > 0 && n <= 3)) throw ComputeError(CHV_ERR_BAD_INPUT, "Input image must have 1, 2, or 3 planes");
    if (n != image.buffers.size()) throw ComputeError(CHV_ERR_BAD_INPUT, "Input image must have the same number of buffers as planes");
    auto out = std::make_shared<ImageBuffer>(image);
    out->computeTextures.clear(); out->gpuOffsets.clear(); out->gpuPitches.clear();
    const size_t np = std::min(n, (size_t)maxPlanes);
    size_t total = 0;
    for (size_t i = 0; i < np; i++) {                  // 128-byte aligned pitches, planes back to back
        const Plane &p = image.planes[i];
        if (p.size.x <= 0 || p.size.y <= 0) throw ComputeError(CHV_ERR_INVALID_OPERATION, "empty plane");
        size_t pitch = ((size_t)p.size.x * planeComponents(p) + 127) / 128 * 128;
        out->gpuPitches.push_back(pitch); out->gpuOffsets.push_back(total);
        total += pitch * (size_t)p.size.y;
    }
    chv_buffer *h = nullptr;
    check(chv_buffer_alloc(ctx.get(), total, &h));
    auto tex = std::make_shared<ComputeBuffer>(h, total, 0);
    for (size_t i = 0; i < np; i++) out->computeTextures.push_back(tex);
    beginComputePass(ctx);
    // planes that are adjacent with equal pitch and width on both sides (NV12's luma + chroma, y420p's two chroma
    // planes) travel as one pitched copy
    for (size_t i = 0; i < np;) {
        const Plane &p = image.planes[i];
        const size_t wb = (size_t)p.size.x * planeComponents(p);
        size_t rows = (size_t)p.size.y, j = i + 1;
        const uint8_t *src = image.buffers[i]->data();
        while (j < np) {
            const Plane &q = image.planes[j];
            if (out->gpuPitches[j] != out->gpuPitches[i] || q.stride != p.stride || (size_t)q.size.x * planeComponents(q) != wb ||
                out->gpuOffsets[j] != out->gpuOffsets[i] + out->gpuPitches[i] * rows || image.buffers[j]->data() != src + (size_t)p.stride * rows)
                break;
            rows += (size_t)q.size.y; j++;
        }
        check(chv_upload(ctx.get(), h, out->gpuOffsets[i], out->gpuPitches[i], src, (size_t)p.stride, wb, rows, asynchronous ? 1 : 0));
        i = j;
    }
    endComputePass(ctx, true);
    out->bufferType = BufferType::gpu;
    if (!retainCpuBuffer) out->buffers.clear();
    PictureSample r = pict;
    r.img = out;
    return r;
}

// compute.cl.swift:461-498
inline PictureSample downloadComputePicture(const ComputeContext &ctx, const PictureSample &pict, bool retainGpuBuffer = false) {
    if (pict.bufferType() != BufferType::gpu) return pict;
    const ImageBuffer &image = *pict.img;
    auto out = std::make_shared<ImageBuffer>(image);
    out->buffers.clear();
    beginComputePass(ctx);
    for (size_t i = 0; i < image.computeTextures.size(); i++) {
        const Plane &p = image.planes[i];
        int comps = planeComponents(p);
        auto buf = i < image.buffers.size() ? image.buffers[i] : std::make_shared<Data>((size_t)p.stride * (size_t)p.size.y, 0);
        check(chv_download(ctx.get(), buf->data(), (size_t)p.stride, image.computeTextures[i]->handle, image.gpuOffsets[i],
                           image.gpuPitches[i], (size_t)p.size.x * comps, (size_t)p.size.y));
        out->buffers.push_back(buf);
    }
    endComputePass(ctx, true);
    out->bufferType = BufferType::cpu;
    if (!retainGpuBuffer) { out->computeTextures.clear(); out->gpuOffsets.clear(); out->gpuPitches.clear(); }
    PictureSample r = pict;
    r.img = out;
    return r;
}

// The D2H half of a download barrier that overlaps the next tick (chv_download_async; swift/compute.hip.swift::downloadComputePictureAsync): the
// picture's planes, tightly packed one after the other, into PINNED host memory (chv_host_alloc); the bytes are there once the context's stream
// has passed the copies — endComputePass(ctx, true), or an event recorded behind them.  Returns the bytes the picture takes.
inline size_t downloadComputePictureAsync(const ComputeContext &ctx, const PictureSample &pict, void *pinned) {
    if (pict.bufferType() != BufferType::gpu || !pict.img) throw ComputeError(CHV_ERR_BAD_INPUT, "Missing device image");
    const ImageBuffer &image = *pict.img;
    size_t offset = 0;
    for (size_t i = 0; i < image.computeTextures.size(); i++) {
        const Plane &p = image.planes[i];
        const size_t rowBytes = (size_t)p.size.x * (size_t)planeComponents(p), rows = (size_t)p.size.y;
        check(chv_download_async(ctx.get(), (uint8_t *)pinned + offset, rowBytes, image.computeTextures[i]->handle, image.gpuOffsets[i],
                                 image.gpuPitches[i], rowBytes, rows));
        offset += rowBytes * rows;
    }
    return offset;
}

// ---- kernels (compute.swift:76-86,145-170; compute.cl.swift:250-344) ------------------------------
using ImageUniforms = chv_uniforms;        // 236 bytes, same layout as the Swift struct

inline bool describe(const PictureSample &s, chv_image *d, int maxPlanes = 3) {
    if (!s.img || s.img->bufferType != BufferType::gpu || s.img->computeTextures.empty()) return false;
    std::memset(d, 0, sizeof *d);
    d->format = (int)s.img->pixelFormat; d->width = (int)s.img->size.x; d->height = (int)s.img->size.y;
    int n = std::min((int)s.img->computeTextures.size(), maxPlanes);
    d->n_planes = n;
    for (int i = 0; i < n; i++) {
        const Plane &p = s.img->planes[i];
        d->planes[i] = chv_plane{ s.img->computeTextures[i]->handle, s.img->gpuOffsets[i], (int)p.size.x, (int)p.size.y,
                                  (int)s.img->gpuPitches[i], planeComponents(p) };
    }
    return true;
}

inline ComputeContext runComputeKernel(ComputeContext ctx, const std::vector<PictureSample> &images, const PictureSample &target,
                                       ComputeKernel kernel, int maxPlanes = 3, const ImageUniforms *uniforms = nullptr,
                                       bool blends = false, int colorspace = CHV_CSC_BT601_LIMITED) {
    chv_image t;
    if (!describe(target, &t)) throw ComputeError(CHV_ERR_BAD_TARGET, "target has no GPU image buffer");
    std::vector<chv_image> in(images.size());
    for (size_t i = 0; i < images.size(); i++)
        if (!describe(images[i], &in[i], maxPlanes)) throw ComputeError(CHV_ERR_BAD_INPUT, "Bad input image");
    chv_kernel_opts opts{ colorspace, { 0, 0, 0 } };
    check(chv_run_kernel(ctx.get(), (int)kernel, &t, in.data(), (int)in.size(), uniforms, uniforms ? sizeof(ImageUniforms) : 0,
                         blends ? 1 : 0, &opts));
    return ctx;
}

// ---- the two kernels of the enum that work on buffers (compute.swift:67,70; kernels.cl.swift:534-562, kernels.metal:129-267) ----
// A ComputeBuffer bound where a kernel expects an image: one plane of `width` x `height` texels of `components` bytes — interleaved-stereo
// int16 samples are 2-byte texels (snd_s16i_s16i), a luma plane 1-byte ones, me_fullsearch's output one RGBA8 texel per block.
struct BufferImage { ComputeBufferRef buffer; int width = 0, height = 1, components = 2; size_t offset = 0; };
using BufferUniforms = chv_snd_uniforms;               // kernels.cl.swift:536-541
using MotionEstimationUniforms = chv_me_uniforms;      // kernels.metal:33-37
inline chv_image describe(const BufferImage &b) {
    chv_image d;
    std::memset(&d, 0, sizeof d);
    d.format = CHV_FMT_INVALID; d.width = b.width; d.height = b.height; d.n_planes = 1;
    d.planes[0] = chv_plane{ b.buffer ? b.buffer->handle : nullptr, b.offset, b.width, b.height, b.width * b.components, b.components };
    return d;
}
inline ComputeBufferRef uploadComputeBuffer(const ComputeContext &ctx, const void *src, size_t bytes) {      // compute.cl.swift:361-379
    chv_buffer *h = nullptr;
    check(chv_buffer_alloc(ctx.get(), bytes, &h));
    auto b = std::make_shared<ComputeBuffer>(h, bytes, 0);
    check(chv_upload(ctx.get(), h, 0, bytes, src, bytes, bytes, 1, 0));
    return b;
}
inline void downloadComputeBuffer(const ComputeContext &ctx, const ComputeBufferRef &src, void *dst) {       // compute.cl.swift:381-396
    check(chv_download(ctx.get(), dst, src->size, src->handle, 0, src->size, src->size, 1));
}
template <typename T>
inline ComputeContext runComputeKernel(ComputeContext ctx, const std::vector<BufferImage> &images, const BufferImage &target, ComputeKernel kernel,
                                       const T &uniforms) {
    chv_image t = describe(target);
    std::vector<chv_image> in;
    for (const BufferImage &b : images) in.push_back(describe(b));
    check(chv_run_kernel(ctx.get(), (int)kernel, &t, in.data(), (int)in.size(), &uniforms, sizeof(T), 0, nullptr));
    return ctx;
}

// ---- ComputeKernel.custom(name:) (compute.swift:72-73) + buildComputeKernel (compute.cl.swift:153-195) ----
struct CustomKernel { std::string name; };

// `source` is HIP C++ (hipRTC), prefixed with chv_custom_prelude(); it defines
// extern "C" __global__ void <name>(chv_custom_args a).  Failure: ComputeError.badInputData with the build log.
inline ComputeContext buildComputeKernel(ComputeContext ctx, const std::string &name, const std::string &source) {
    check(chv_kernel_build(ctx.get(), name.c_str(), source.c_str()));
    return ctx;
}

// runComputeKernel<T> with a kernel from the context's library; `uniforms` may be any trivially copyable value
template <typename T = ImageUniforms>
inline ComputeContext runComputeKernel(ComputeContext ctx, const std::vector<PictureSample> &images, const PictureSample &target,
                                       const CustomKernel &kernel, int maxPlanes = 3, const T *uniforms = nullptr, bool blends = false) {
    chv_image t;
    if (!describe(target, &t)) throw ComputeError(CHV_ERR_BAD_TARGET, "target has no GPU image buffer");
    std::vector<chv_image> in(images.size());
    for (size_t i = 0; i < images.size(); i++)
        if (!describe(images[i], &in[i], maxPlanes)) throw ComputeError(CHV_ERR_BAD_INPUT, "Bad input image");
    check(chv_run_custom(ctx.get(), kernel.name.c_str(), &t, in.data(), (int)in.size(), uniforms, uniforms ? sizeof(T) : 0, blends ? 1 : 0));
    return ctx;
}

inline ImageUniforms imageUniformsFor(const PictureSample &image, const PictureSample &target) {   // compute.swift:147-161
    ImageUniforms u;
    std::memset(&u, 0, sizeof u);
    image.matrix.kernelRows(u.transform);
    image.textureMatrix.kernelRows(u.texture_transform);
    image.borderMatrix.kernelRows(u.border_matrix);
    u.fill_color[0] = image.fillColor.x; u.fill_color[1] = image.fillColor.y; u.fill_color[2] = image.fillColor.z; u.fill_color[3] = image.fillColor.w;
    u.input_size[0] = image.size().x; u.input_size[1] = image.size().y;
    u.output_size[0] = target.size().x; u.output_size[1] = target.size().y;
    u.opacity = image.opacity; u.image_time = (float)image.time; u.target_time = (float)target.time;
    return u;
}

inline ComputeContext applyComputeImage(ComputeContext ctx, const PictureSample &image, const PictureSample &target,
                                        ComputeKernel kernel, int colorspace = CHV_CSC_BT601_LIMITED) {
    ImageUniforms u = imageUniformsFor(image, target);
    return runComputeKernel(ctx, { image }, target, kernel, 3, &u, true, colorspace);
}

struct TickLayer { ComputeKernel kernel; PictureSample image; ImageUniforms uniforms; int colorspace = CHV_CSC_BT601_LIMITED; };

// one mixer tick in one launch (chv_composite)
inline ComputeContext compositeTick(ComputeContext ctx, const PictureSample &target, const std::vector<TickLayer> &layers,
                                    bool clearFirst = true) {
    chv_image t;
    if (!describe(target, &t)) throw ComputeError(CHV_ERR_BAD_TARGET, "target has no GPU image buffer");
    std::vector<chv_layer> ls(layers.size());
    for (size_t i = 0; i < layers.size(); i++) {
        std::memset(&ls[i], 0, sizeof ls[i]);
        ls[i].kernel = (int)layers[i].kernel;
        if (!describe(layers[i].image, &ls[i].image)) throw ComputeError(CHV_ERR_BAD_INPUT, "Bad input image");
        ls[i].uniforms = layers[i].uniforms;
        ls[i].opts.colorspace = layers[i].colorspace;
    }
    check(chv_composite(ctx.get(), &t, clearFirst ? 1 : 0, ls.data(), (int)ls.size()));
    return ctx;
}

// separable Lanczos-3 resample BGRA -> BGRA (chv_scale_lanczos; no reference counterpart)
inline ComputeContext scaleLanczos(ComputeContext ctx, const PictureSample &dst, const PictureSample &src) {
    chv_image d, s;
    if (!describe(dst, &d)) throw ComputeError(CHV_ERR_BAD_TARGET, "target has no GPU image buffer");
    if (!describe(src, &s)) throw ComputeError(CHV_ERR_BAD_INPUT, "Bad input image");
    check(chv_scale_lanczos(ctx.get(), &d, &s));
    return ctx;
}

// n resizes of one geometry as one launch (chv_scale_lanczos_batch): same bytes as n scaleLanczos calls
inline ComputeContext scaleLanczos(ComputeContext ctx, const std::vector<std::pair<PictureSample, PictureSample>> &dstSrcPairs) {
    std::vector<chv_image> d(dstSrcPairs.size()), s(dstSrcPairs.size());
    for (size_t i = 0; i < dstSrcPairs.size(); i++) {
        if (!describe(dstSrcPairs[i].first, &d[i])) throw ComputeError(CHV_ERR_BAD_TARGET, "target has no GPU image buffer");
        if (!describe(dstSrcPairs[i].second, &s[i])) throw ComputeError(CHV_ERR_BAD_INPUT, "Bad input image");
    }
    if (!d.empty()) check(chv_scale_lanczos_batch(ctx.get(), d.data(), s.data(), (int)d.size()));
    return ctx;
}

// Many independent ticks as ONE launch (chv_batch_*): what a host with several mixers / streams on a device
// (composer.swift:203-224) uses instead of one chv_composite per tick; byte-identical to running them one by one.
struct Tick { PictureSample target; bool clearFirst = true; std::vector<TickLayer> layers; };
class TickBatch {
public:
    TickBatch(const ComputeContext &ctx, const std::vector<Tick> &ticks) : keep_(ticks) {
        std::vector<chv_tick> ts(ticks.size());
        layers_.resize(ticks.size());
        for (size_t i = 0; i < ticks.size(); i++) {
            std::memset(&ts[i], 0, sizeof ts[i]);
            if (!describe(ticks[i].target, &ts[i].target)) throw ComputeError(CHV_ERR_BAD_TARGET, "target has no GPU image buffer");
            layers_[i].resize(ticks[i].layers.size());
            for (size_t l = 0; l < ticks[i].layers.size(); l++) {
                chv_layer &d = layers_[i][l];
                std::memset(&d, 0, sizeof d);
                d.kernel = (int)ticks[i].layers[l].kernel;
                if (!describe(ticks[i].layers[l].image, &d.image)) throw ComputeError(CHV_ERR_BAD_INPUT, "Bad input image");
                d.uniforms = ticks[i].layers[l].uniforms;
                d.opts.colorspace = ticks[i].layers[l].colorspace;
            }
            ts[i].clear_first = ticks[i].clearFirst ? 1 : 0;
            ts[i].n_layers = (int)layers_[i].size();
            ts[i].layers = layers_[i].data();
        }
        check(chv_batch_create(ctx.get(), ts.data(), (int)ts.size(), &batch_));
        char name[128] = { 0 };
        check(chv_batch_describe(batch_, name, sizeof name, nullptr));
        kernelName = name;
    }
    TickBatch(const TickBatch &) = delete;
    ~TickBatch() { if (batch_) chv_batch_destroy(batch_); }
    ComputeContext run(ComputeContext ctx) const { check(chv_batch_run(ctx.get(), batch_)); return ctx; }   // inside a compute pass
    std::string kernelName;
private:
    chv_batch *batch_ = nullptr;
    std::vector<Tick> keep_;                       // the pictures stay alive as long as the device descriptors do
    std::vector<std::vector<chv_layer>> layers_;
};

// ---- pipeline operators (compute.swift:175-255) -------------------------------------------------------
struct EventError { std::string source; int code = 0; std::string description; std::string assetId; };
template <typename T> struct EventBox {          // event.swift:63-95: .just / .nothing / .error / .gone
    enum Kind { just, nothing, error, gone } kind = nothing;
    T value{}; EventError err;
};

class GPUBarrierUpload {
public:
    explicit GPUBarrierUpload(const ComputeContext &context, bool retainCpuBuffer = true)
        : context_(createComputeContext(context)), retain_(retainCpuBuffer) {}
    EventBox<PictureSample> operator()(const PictureSample &s) const {
        EventBox<PictureSample> r;
        if (s.bufferType() == BufferType::cpu) {
            try { r.value = uploadComputePicture(context_, s, 3, retain_); r.kind = r.just; }
            catch (const ComputeError &e) { r.kind = r.error; r.err = EventError{ "barrier.upload", -1, e.what(), s.assetId }; }
        } else { r.value = s; r.kind = r.just; }
        return r;
    }
private:
    ComputeContext context_; bool retain_;
};

class GPUBarrierDownload {
public:
    explicit GPUBarrierDownload(const ComputeContext &context, bool retainGpuBuffer = true)
        : context_(createComputeContext(context)), retain_(retainGpuBuffer) {}
    EventBox<PictureSample> operator()(const PictureSample &s) const {
        EventBox<PictureSample> r;
        if (s.bufferType() == BufferType::gpu) {
            try { r.value = downloadComputePicture(context_, s, retain_); r.kind = r.just; }
            catch (const ComputeError &e) { r.kind = r.error; r.err = EventError{ "barrier.download", -1, e.what(), s.assetId }; }
        } else { r.value = s; r.kind = r.just; }
        return r;
    }
private:
    ComputeContext context_; bool retain_;
};

// ---- VideoMixer (mix.video.swift:21-184) without the clock: push() is the Source's set closure
//      (:57-75), mix(at) one tick (:95-140) ------------------------------------------------------------
class VideoMixer {
public:
    static constexpr int numberBackingImages = 10;     // mix.video.swift:167
    VideoMixer(const std::string &workspaceId, Vector2 outputSize, PixelFormat outputFormat, const ComputeContext &computeContext,
               const std::string &assetId = "mixer", bool fused = true, bool bgraTransformAware = false)
        : clContext_(createComputeContext(computeContext)), backingSize_(outputSize), backingFormat_(outputFormat),
          idWorkspace_(workspaceId), idAsset_(assetId), fused_(fused), bgraTx_(bgraTransformAware) {}

    const std::string &assetId() const { return idAsset_; }

    EventBox<PictureSample> push(const PictureSample &pic) {
        EventBox<PictureSample> r;
        if (pic.assetId != idAsset_) { samples_[0][pic.revision] = pic; r.kind = r.nothing; }
        else { r.kind = r.just; r.value = pic; }
        return r;
    }

    ComputeKernel findKernel(const PictureSample *image, const PictureSample &target) const {   // mix.video.swift:142-146
        std::string inp = image ? lowercasedName(image->pixelFormat()) : "clear";
        std::string outp = lowercasedName(target.pixelFormat());
        std::string name = "img_" + inp + "_" + outp;
        if (bgraTx_ && image && outp == "bgra" && (inp == "bgra" || inp == "rgba")) name += "_tx";
        return defaultComputeKernelFromString(name);
    }

    const ComputeContext &context() const { return clContext_; }

    // the tick of this mixer as data (next backing image, z-sorted layers): for VideoMixerGroup
    Tick prepareTick() {
        Tick t;
        t.target = getBacking();
        t.clearFirst = true;
        for (auto &im : sortedImages()) t.layers.push_back(TickLayer{ findKernel(&im, t.target), im, imageUniformsFor(im, t.target) });
        return t;
    }
    void finishTick() { samples_[1] = samples_[0]; samples_[0].clear(); }   // mix.video.swift:104-107

    EventBox<PictureSample> mix(double at) {
        EventBox<PictureSample> out;
        try {
            PictureSample backing = getBacking();
            std::vector<PictureSample> images = sortedImages();
            if (fused_) {
                std::vector<TickLayer> layers;
                for (auto &im : images) layers.push_back(TickLayer{ findKernel(&im, backing), im, imageUniformsFor(im, backing) });
                usingContext(clContext_, [&](ComputeContext c) { return compositeTick(c, backing, layers, true); });
            } else {
                usingContext(clContext_, [&](ComputeContext c) {
                    c = runComputeKernel(c, {}, backing, findKernel(nullptr, backing));
                    for (auto &im : images) c = applyComputeImage(c, im, backing, findKernel(&im, backing));
                    return c;
                });
            }
            out.kind = out.just; out.value = backing; out.value.pts = at; out.value.time = at; out.value.assetId = idAsset_;
        } catch (const ComputeError &e) {
            out.kind = out.error; out.err = EventError{ "mix.video", -2, std::string("Compute error ") + e.what(), idAsset_ };
        }
        finishTick();
        return out;
    }

private:
    std::vector<PictureSample> sortedImages() const {
        std::map<std::string, PictureSample> merged = samples_[1];
        for (auto &kv : samples_[0]) merged[kv.first] = kv.second;       // lhs wins, mix.video.swift:114
        std::vector<PictureSample> images;
        for (auto &kv : merged) images.push_back(kv.second);
        std::stable_sort(images.begin(), images.end(), [](const PictureSample &a, const PictureSample &b) { return a.zIndex < b.zIndex; });
        return images;
    }
    PictureSample getBacking() {                      // mix.video.swift:148-165
        if ((int)backing_.size() < numberBackingImages) {
            PictureSample image = createPictureSample(backingSize_, backingFormat_, idAsset_, idWorkspace_);
            backing_.push_back(uploadComputePicture(clContext_, image));
            return backing_.back();
        }
        PictureSample image = backing_[currentBacking_];
        currentBacking_ = (currentBacking_ + 1) % (int)backing_.size();
        return image;
    }
    ComputeContext clContext_;
    std::vector<PictureSample> backing_;
    int currentBacking_ = 0;
    Vector2 backingSize_; PixelFormat backingFormat_;
    std::string idWorkspace_, idAsset_;
    bool fused_, bgraTx_;
    std::map<std::string, PictureSample> samples_[2];
};


// Several VideoMixers of one device ticked together: one launch per canvas format and one host wait instead of one
// launch and one wait per mixer; each element of the result is what that mixer's own mix(at) returns.
class VideoMixerGroup {
public:
    explicit VideoMixerGroup(std::vector<VideoMixer *> mixers) : mixers_(std::move(mixers)) {}
    std::vector<EventBox<PictureSample>> mix(double at) {
        std::vector<EventBox<PictureSample>> out(mixers_.size());
        try {
            std::map<int, std::vector<Tick>> byFormat;
            std::vector<PictureSample> backings;
            for (VideoMixer *m : mixers_) {
                Tick t = m->prepareTick();
                backings.push_back(t.target);
                byFormat[(int)t.target.pixelFormat()].push_back(std::move(t));
            }
            std::vector<std::unique_ptr<TickBatch>> batches;
            for (auto &kv : byFormat) batches.emplace_back(new TickBatch(mixers_[0]->context(), kv.second));
            usingContext(mixers_[0]->context(), [&](ComputeContext c) { for (auto &b : batches) c = b->run(c); return c; });
            for (size_t i = 0; i < mixers_.size(); i++) {
                out[i].kind = out[i].just; out[i].value = backings[i];
                out[i].value.pts = at; out[i].value.time = at; out[i].value.assetId = mixers_[i]->assetId();
            }
        } catch (const ComputeError &e) {
            for (size_t i = 0; i < mixers_.size(); i++) {
                out[i].kind = out[i].error;
                out[i].err = EventError{ "mix.video", -2, std::string("Compute error ") + e.what(), mixers_[i]->assetId() };
            }
        }
        for (VideoMixer *m : mixers_) m->finishTick();
        return out;
    }
private:
    std::vector<VideoMixer *> mixers_;
};

// ---- PictureFilter: the Tx<PictureSample, PictureSample> the reference sketches and leaves commented out
//      (filter.pict.swift:20-47).  Converts a picture to outputFormat at outputSize on the device: one
//      full-canvas layer through the composite kernels (colour conversion + bilinear scale in one launch),
//      or a separable Lanczos-3 resample (BGRA -> BGRA).  CPU samples are uploaded first; results land in a
//      ring of device images like the mixer's (mix.video.swift:148-167). ------------------------------------
class PictureFilter {
public:
    enum class Scaler { bilinear, lanczos };
    static constexpr int numberBackingImages = 10;
    PictureFilter(Vector2 outputSize, PixelFormat outputFormat, const ComputeContext &computeContext,
                  Scaler scaler = Scaler::bilinear, int colorspace = CHV_CSC_BT601_LIMITED)
        : context_(createComputeContext(computeContext)), size_(outputSize), format_(outputFormat), scaler_(scaler), colorspace_(colorspace) {}

    ComputeKernel findKernel(const PictureSample &image) const {      // VideoMixer.findKernel's naming rule
        std::string inp = lowercasedName(image.pixelFormat()), outp = lowercasedName(format_);
        std::string name = "img_" + inp + "_" + outp;
        if (outp == "bgra" && (inp == "bgra" || inp == "rgba")) name += "_tx";
        // an RGB picture onto a 4:2:0 format: the integer BT.601/709 matrix of `colorspace` unless the
        // reference's float full-range kernels are asked for
        if (integerMatrix && (outp == "nv12" || outp == "y420p") && (inp == "bgra" || inp == "rgba")) name += "_int";
        return defaultComputeKernelFromString(name);
    }
    bool integerMatrix = true;

    EventBox<PictureSample> operator()(const PictureSample &sample) {
        EventBox<PictureSample> r;
        try {
            PictureSample src = sample.bufferType() == BufferType::cpu ? uploadComputePicture(context_, sample) : sample;
            PictureSample dst = getBacking(sample);
            if (scaler_ == Scaler::lanczos) {
                if (src.pixelFormat() != PixelFormat::BGRA || format_ != PixelFormat::BGRA)
                    throw ComputeError(CHV_ERR_NOT_IMPLEMENTED, "lanczos: BGRA -> BGRA only");
                usingContext(context_, [&](ComputeContext c) { return scaleLanczos(c, dst, src); });
            } else {
                // a full-canvas opaque layer: the unit quad stretched over the canvas in NDC, no border, no fill
                PictureSample full = src;
                Matrix4 quad; quad.m[0] = 2; quad.m[5] = 2; quad.m[3] = -1; quad.m[7] = -1; quad.m[11] = 1;
                full.matrix = quad; full.borderMatrix = quad; full.textureMatrix = Matrix4::identity();
                full.fillColor = Vector4{}; full.opacity = 1.0f;
                std::vector<TickLayer> layers{ TickLayer{ findKernel(src), full, imageUniformsFor(full, dst), colorspace_ } };
                usingContext(context_, [&](ComputeContext c) { return compositeTick(c, dst, layers, true); });
            }
            r.kind = r.just; r.value = sample;            // time stamps, ids and transform state carry over
            r.value.img = dst.img;
        } catch (const ComputeError &e) {
            r.kind = r.error; r.err = EventError{ "filter.pict", -2, std::string("Compute error ") + e.what(), sample.assetId };
        }
        return r;
    }

private:
    PictureSample getBacking(const PictureSample &like) {
        if ((int)backing_.size() < numberBackingImages) {
            PictureSample image = createPictureSample(size_, format_, like.assetId, like.workspaceId);
            backing_.push_back(uploadComputePicture(context_, image));
            return backing_.back();
        }
        PictureSample image = backing_[current_];
        current_ = (current_ + 1) % (int)backing_.size();
        return image;
    }
    ComputeContext context_;
    Vector2 size_; PixelFormat format_; Scaler scaler_; int colorspace_;
    std::vector<PictureSample> backing_; int current_ = 0;
};

// ---- PictureAnimator (animator.pic.swift:28-128, 149-272): the caller that produces the matrices ----
// Including elements attached to a parent element through parent anchors (:149-193).
// The 4x4 algebra is VectorMath's in the reference (un-vendored, un-pinned); conventions as in Matrix4 above.
enum class AspectMode { aspectNone, aspectFit, aspectFill };
enum class PictureOrigin { originTopLeft, originCenter };
enum PictureAnchor : unsigned { anchorTopLeft = 1, anchorTopRight = 2, anchorBottomLeft = 4, anchorBottomRight = 8 };   // bit set

struct ElementState {                       // Proto/Composition.proto:56-71, picture fields
    double picPos[3] = { 0, 0, 0 };
    double size[2] = { 0, 0 };
    double textureOffset[2] = { 0, 0 };
    double rotation = 0, transparency = 0;
    AspectMode picAspect = AspectMode::aspectNone;
    PictureOrigin picOrigin = PictureOrigin::originTopLeft;
    bool hasFillColor = false;
    double fillColor[4] = { 0, 0, 0, 0 };   // r g b a
    double borderSize[4] = { 0, 0, 0, 0 };  // l t r b
    bool hidden = false;
    unsigned parentAnchor = 0;              // PictureAnchor bits; 0 -> anchorTopLeft (:64)
};

struct ComputedPictureState { Matrix4 matrix, textureMatrix, borderMatrix; Vector4 fillColor; float opacity = 1; };

inline Matrix4 computeTextureMatrix(Vector2 sampleSize, const double geometrySize[2], const double textureOffset[2], AspectMode aspect) {
    double origAspect = sampleSize.x / sampleSize.y, geomAspect = geometrySize[0] / geometrySize[1];
    double scalex, scaley;
    if (aspect == AspectMode::aspectFit) {
        scalex = origAspect > geomAspect ? 1.0 : origAspect / geomAspect;
        scaley = origAspect <= geomAspect ? 1.0 : geomAspect / origAspect;
    } else if (aspect == AspectMode::aspectFill) {
        scalex = origAspect <= geomAspect ? 1.0 : origAspect / geomAspect;
        scaley = origAspect > geomAspect ? 1.0 : geomAspect / origAspect;
    } else return Matrix4::identity();
    return Matrix4::translation(textureOffset[0] + (1.0 - scalex) / 2, textureOffset[1] + (1.0 - scaley) / 2) * Matrix4::scale(scalex, scaley);
}

inline ElementState computeElementState(const ElementState &a, const ElementState &b, double pct) {   // :195-205
    auto lerp = [pct](double x, double y) { return x + (y - x) * pct; };
    ElementState r = a;
    for (int i = 0; i < 3; i++) r.picPos[i] = lerp(a.picPos[i], b.picPos[i]);
    for (int i = 0; i < 2; i++) { r.size[i] = lerp(a.size[i], b.size[i]); r.textureOffset[i] = lerp(a.textureOffset[i], b.textureOffset[i]); }
    r.rotation = lerp(a.rotation, b.rotation); r.transparency = lerp(a.transparency, b.transparency);
    r.picAspect = b.picAspect; r.picOrigin = b.picOrigin;
    for (int i = 0; i < 4; i++) {
        r.fillColor[i] = lerp(a.hasFillColor ? a.fillColor[i] : 0.0, b.hasFillColor ? b.fillColor[i] : 0.0);
        r.borderSize[i] = lerp(a.borderSize[i], b.borderSize[i]);
    }
    r.hasFillColor = a.hasFillColor || b.hasFillColor;
    return r;
}

// :149-193 — position and size of an element whose corners follow its parent's corners.
// parentPos: the parent's translation; d: parent size now minus parent size at attachment.
inline void computePositionSize(const double basePos[2], const double baseSize[2], const double parentPos[2], const double d[2],
                                unsigned anchors, double pos[2], double size[2]) {
    const double rx = basePos[0] + parentPos[0], ry = basePos[1] + parentPos[1];
    double v[3][2] = { { rx, ry }, { rx + baseSize[0], ry }, { rx, ry + baseSize[1] } };
    if (anchors & anchorBottomRight) {
        for (auto &q : v) { q[0] += d[0]; q[1] += d[1]; }
        if (anchors & anchorBottomLeft) { v[0][0] = rx; v[2][0] = rx; }
        if (anchors & anchorTopRight) { v[0][1] = ry; v[1][1] = ry; }
        if (anchors & anchorTopLeft) {
            v[0][0] = rx; v[0][1] = ry;
            v[1][0] = rx + baseSize[0] + d[0]; v[1][1] = ry;
            v[2][0] = rx; v[2][1] = ry + baseSize[1] + d[1];
        }
    } else if (anchors & anchorTopRight) {
        v[1][0] += d[0];
        if (!(anchors & anchorTopLeft) && !(anchors & anchorBottomLeft)) { v[0][0] += d[0]; v[2][0] += d[0]; }
        else if (anchors & anchorBottomLeft) v[2][1] += d[1];
    } else if (anchors & anchorBottomLeft) {
        v[2][1] += d[1];
        if (!(anchors & anchorTopLeft)) { v[1][1] += d[1]; v[0][1] += d[1]; }
    }
    pos[0] = v[0][0]; pos[1] = v[0][1];
    size[0] = v[1][0] - v[0][0]; size[1] = v[2][1] - v[0][1];
}

// lengths of the first two columns' xy parts = the element's size under rotation (:243-249)
inline void columnScale(const Matrix4 &m, double out[2]) {
    out[0] = std::hypot(m.m[0], m.m[4]);
    out[1] = std::hypot(m.m[1], m.m[5]);
}

// :229-272.  parent: the parent's (un-projected) matrix or nullptr; initialParent: the parent's state
// when this element was attached, or nullptr.
inline ComputedPictureState computePictureState(Vector2 sampleSize, const ElementState &state, const Matrix4 *parent = nullptr,
                                                unsigned anchors = anchorTopLeft, const ComputedPictureState *initialParent = nullptr) {
    double parentPos[2] = { 0, 0 }, parentSize[2] = { 0, 0 }, initialSize[2] = { 0, 0 };
    if (parent) { parentPos[0] = parent->m[3]; parentPos[1] = parent->m[7]; columnScale(*parent, parentSize); }
    if (initialParent) columnScale(initialParent->matrix, initialSize);
    const double delta[2] = { parentSize[0] - initialSize[0], parentSize[1] - initialSize[1] };
    double ax = state.picOrigin == PictureOrigin::originTopLeft ? 0.0 : -state.size[0] / 2;
    double ay = state.picOrigin == PictureOrigin::originTopLeft ? 0.0 : -state.size[1] / 2;
    double rel[2], size[2];
    computePositionSize(state.picPos, state.size, parentPos, delta, anchors, rel, size);
    double px = rel[0] + ax, py = rel[1] + ay;
    const double *b = state.borderSize;
    ComputedPictureState c;
    c.matrix = Matrix4::translation(px, py) * Matrix4::rotationZ(state.rotation) * Matrix4::scale(size[0], size[1]);
    c.textureMatrix = computeTextureMatrix(sampleSize, size, state.textureOffset, state.picAspect);
    c.borderMatrix = Matrix4::translation(px - b[0], py - b[1]) * Matrix4::rotationZ(state.rotation) *
                     Matrix4::scale(b[0] + size[0] + b[2], b[1] + size[1] + b[3]);
    if (state.hasFillColor) c.fillColor = Vector4{ (float)state.fillColor[0], (float)state.fillColor[1], (float)state.fillColor[2], (float)state.fillColor[3] };
    c.opacity = (float)(1.0 - state.transparency);
    return c;
}

class PictureAnimator {                     // Tx<PictureSample, PictureSample>, :28-128
public:
    PictureAnimator(Vector2 canvasSize, const ElementState &state, const std::string &revision = "",
                    const PictureAnimator *parent = nullptr, unsigned parentAnchors = anchorTopLeft)
        : canvas_(canvasSize), state_(state), revision_(revision), parent_(parent), anchors_(parentAnchors) {}
    void setParent(const PictureAnimator *p) { parent_ = p; }
    // immediate switch (duration <= 0, :56-66): anchors follow the new state, the attachment is re-initialised
    void setState(const ElementState &s) {
        state_ = s; has_initial_ = false;
        anchors_ = s.parentAnchor ? s.parentAnchor : (unsigned)anchorTopLeft;
    }
    ComputedPictureState computedState(const PictureSample &sample, const ComputedPictureState *parentState = nullptr) const {   // :84-105
        return computePictureState(sample.size(), state_, parentState ? &parentState->matrix : nullptr, anchors_,
                                   has_initial_ ? &initial_ : nullptr);
    }
    EventBox<PictureSample> operator()(const PictureSample &sample) {
        EventBox<PictureSample> r;
        if (state_.hidden) { r.kind = r.nothing; return r; }
        // the parent's state is computed without ITS parent (:112); the attachment state is recorded only
        // after the first sample went through (:115-117)
        ComputedPictureState ps;
        if (parent_) ps = parent_->computedState(sample);
        ComputedPictureState cs = computedState(sample, parent_ ? &ps : nullptr);
        if (parent_ && !has_initial_) { initial_ = ps; has_initial_ = true; }
        Matrix4 proj = Matrix4::ortho(canvas_.x, canvas_.y);
        r.kind = r.just; r.value = sample;
        r.value.matrix = proj * cs.matrix; r.value.textureMatrix = cs.textureMatrix; r.value.borderMatrix = proj * cs.borderMatrix;
        r.value.fillColor = cs.fillColor; r.value.opacity = cs.opacity * (parent_ ? ps.opacity : 1.0f);
        if (!revision_.empty()) r.value.revision = revision_;
        return r;
    }
private:
    Vector2 canvas_; ElementState state_; std::string revision_;
    const PictureAnimator *parent_ = nullptr; unsigned anchors_ = anchorTopLeft;
    ComputedPictureState initial_; bool has_initial_ = false;
};

}  // namespace sv

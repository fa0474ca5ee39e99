/*
 * chipvideo.h — C ABI of CHIPVideo, the MI355X (gfx950) compute backend for
 * SwiftVideo's picture path.
 *
 * This header is what a SwiftPM system-library target `CHIPVideo` exposes to
 * `compute.hip.swift` (the way Sources/CCUDA/module.modulemap:1-6 exposes
 * cuda.h to compute.cuda.swift).  Every entry point names the reference
 * interface it replaces; citations are into unpause-live/SwiftVideo,
 * `Sources/SwiftVideo/` unless a directory is given.
 *
 * Conventions
 *  - plain C, no C++/torch types; every function returns a chv_status
 *    (0 = success), never aborts, never throws across the boundary
 *    (errors surface in Swift as `ComputeError`, compute.swift:22-39);
 *  - a chv_context is used by one thread at a time; different contexts on the
 *    same device may be entered concurrently (GPUBarrierUpload/Download run on
 *    Bus runner threads with their own shared context, compute.swift:177,234);
 *  - chv_buffer_free may be called from any thread at any time
 *    (ComputeBuffer.deinit, compute.cl.swift:55-57);
 *  - pixel work only ever runs on the GPU: there is no CPU fallback and a
 *    missing/unsupported device is an error, not a slow path.
 */
#ifndef CHIPVIDEO_H
#define CHIPVIDEO_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CHV_VERSION 0x000100

/* ---- status codes: one per ComputeError case, compute.swift:22-39 -------- */
typedef enum chv_status {
    CHV_OK = 0,
    CHV_ERR_INVALID_VALUE = 1,        /* .invalidValue                       */
    CHV_ERR_OUT_OF_MEMORY = 2,        /* .outOfMemory                        */
    CHV_ERR_INVALID_CONTEXT = 3,      /* .invalidContext                     */
    CHV_ERR_BAD_TARGET = 4,           /* .badTarget                          */
    CHV_ERR_BAD_INPUT = 5,            /* .badInputData                       */
    CHV_ERR_NOT_IMPLEMENTED = 6,      /* .notImplemented                     */
    CHV_ERR_KERNEL_NOT_FOUND = 7,     /* .computeKernelNotFound              */
    CHV_ERR_DEVICE_NOT_AVAILABLE = 8, /* .deviceNotAvailable                 */
    CHV_ERR_INVALID_DEVICE = 9,       /* .invalidDevice                      */
    CHV_ERR_INVALID_OPERATION = 10,   /* .invalidOperation                   */
    CHV_ERR_BAD_CONTEXT_STATE = 11,   /* .badContextState                    */
    CHV_ERR_INVALID_PLATFORM = 12,    /* .invalidPlatform                    */
    CHV_ERR_UNKNOWN = 13              /* .unknownError                       */
} chv_status;

/* Static string for a status (the checkCLError/check() tables,
 * compute.cl.swift:683-702, compute.cuda.swift:102-112). */
const char *chv_error_string(int status);
/* Thread-local detail text of the last failing call on this thread ("" if none). */
const char *chv_last_error_detail(void);
int chv_version(void);
/* What the library was built with: "arch=<the Makefile's ARCH>;hipcc=<HIP version>;clang=<major.minor>;fp_contract=off;tick_bgra_wave:abl=0,...;
 * ...;lanczos3:abl=0".  `abl` != 0 marks a timing-only ablation build whose pixels are wrong by design (profiles/r02_notes.md section 6):
 * tests/test_abi.py asserts 0 for every kernel family, bench.py prints the string.  The compiler version is there because the streaming and
 * Lanczos strip kernels are scheduled by hand against one hipcc (tests/test_device_code_contract.py pins major.minor). */
const char *chv_build_flags(void);
/* Measurement / test hook: path-selection switches.  Names and values are those of the environment variables read once at
 * first use (CHV_FORCE_GENERAL=1, CHV_BGRA_PATH=wave|tiled|stream, CHV_WAVE_ROWS=8|16, CHV_TILE_ROWS=16|32, CHV_SAME_GEOM=0,
 * CHV_DESC=host|device, CHV_STREAM=0, CHV_YUV_STREAM=0|force, CHV_WAVE_DMA=0, CHV_PASS_FUSE=0, CHV_GEOM_CACHE=0|eager);
 * NULL or "" restores the default.  Process-wide, atomic; not part of the Swift-facing contract. */
int chv_debug_set_switch(const char *name, const char *value);
/* Measurement / test hook: counters of the current device's store of strip-kernel geometry tables (csrc/geom_cache.h): "geom_store_patched"
 * (launches whose layers were pointed at stored tables before their descriptors travelled: batches at creation, lone ticks),
 * "geom_store_batch_hits", "geom_store_builds", "geom_store_bytes", "geom_store_tables".  Unknown name -> CHV_ERR_INVALID_VALUE. */
int chv_debug_get_counter(const char *name, unsigned long long *value);

/* ---- kernels: `enum ComputeKernel`, compute.swift:49-74 ------------------ */
typedef enum chv_kernel {
    CHV_K_IMG_NV12_NV12 = 0,
    CHV_K_IMG_BGRA_NV12 = 1,
    CHV_K_IMG_RGBA_NV12 = 2,
    CHV_K_IMG_BGRA_BGRA = 3,    /* semantics of kernels.metal:52-62 */
    CHV_K_IMG_Y420P_Y420P = 4,
    CHV_K_IMG_Y420P_NV12 = 5,
    CHV_K_IMG_CLEAR_NV12 = 6,
    CHV_K_IMG_CLEAR_YUVS = 7,   /* enum case without a kernel in any backend */
    CHV_K_IMG_CLEAR_BGRA = 8,
    CHV_K_IMG_CLEAR_Y420P = 9,
    CHV_K_IMG_CLEAR_RGBA = 10,  /* name resolves to CLEAR_BGRA, compute.swift:101 */
    CHV_K_IMG_RGBA_Y420P = 11,
    CHV_K_IMG_BGRA_Y420P = 12,
    CHV_K_SND_S16I_S16I = 13,   /* audio mix; chv_snd_uniforms below */
    CHV_K_ME_FULLSEARCH = 14,   /* block motion search; chv_me_uniforms below */
    /* Kernels VideoMixer.findKernel can name (mix.video.swift:142-146) but no
     * reference backend implements; specification in DESIGN.md section 4. */
    CHV_K_IMG_NV12_BGRA = 32,
    CHV_K_IMG_Y420P_BGRA = 33,
    CHV_K_IMG_BGRA_BGRA_TX = 34, /* transform/opacity/fill-aware BGRA over BGRA */
    CHV_K_IMG_RGBA_BGRA_TX = 35,
    /* The encoder side of "integer BT.601/709 YUV <-> RGB": an RGB picture onto a 4:2:0 canvas through the 16.16 integer
     * matrix of chv_kernel_opts.colorspace (the reference's img_bgra_nv12 family is a float full-range matrix with a 0.113
     * blue weight, kernels.cl.swift:96-99).  Specification in DESIGN.md section 4.5. */
    CHV_K_IMG_BGRA_NV12_INT = 36,
    CHV_K_IMG_RGBA_NV12_INT = 37,
    CHV_K_IMG_BGRA_Y420P_INT = 38,
    CHV_K_IMG_RGBA_Y420P_INT = 39
} chv_kernel;

/* defaultComputeKernelFromString, compute.swift:90-110, plus the entries of the
 * compute.swift hunk in INTEGRATION.md section 1: the eight names above and
 * "img_rgba_bgra" (-> CHV_K_IMG_RGBA_BGRA_TX; what VideoMixer.findKernel,
 * mix.video.swift:142-146, synthesises for an RGBA layer on a BGRA canvas).
 * Unknown name -> CHV_ERR_INVALID_VALUE, as the reference throws. */
int chv_kernel_from_string(const char *name, int *kernel);
/* String(describing: ComputeKernel) — the round trip computeTests.swift:9-39 checks. */
const char *chv_kernel_name(int kernel);

/* ---- pixel formats: `enum PixelFormat`, sample.pict.swift:20-33 ---------- */
typedef enum chv_pixel_format {
    CHV_FMT_NV12 = 0, CHV_FMT_NV21 = 1, CHV_FMT_YUVS = 2, CHV_FMT_ZVUY = 3,
    CHV_FMT_Y420P = 4, CHV_FMT_Y422P = 5, CHV_FMT_Y444P = 6,
    CHV_FMT_RGBA = 7, CHV_FMT_BGRA = 8, CHV_FMT_INVALID = 11
} chv_pixel_format;

/* Integer YUV->RGB matrices for CHV_K_IMG_{NV12,Y420P}_BGRA. */
typedef enum chv_colorspace {
    CHV_CSC_BT601_LIMITED = 0, CHV_CSC_BT709_LIMITED = 1,
    CHV_CSC_BT601_FULL = 2, CHV_CSC_BT709_FULL = 3
} chv_colorspace;

/* ---- devices: ComputeDevice / availableComputeDevices,
 *      compute.cl.swift:36-44,107-109 ----------------------------------- */
typedef struct chv_device_info {
    int32_t index;
    int32_t available;        /* ComputeDevice.available  */
    int32_t device_type;      /* 0 = GPU (ComputeDeviceType, compute.swift:41-46) */
    int32_t vendor_id;        /* PCI vendor, 0x1002       */
    int32_t compute_units;
    int32_t supports_images;  /* 0: gfx950 has no image path; planes are linear */
    uint64_t total_memory;
    char name[128];
    char arch[32];            /* "gfx950..." */
} chv_device_info;

int chv_device_count(int *count);
int chv_device_info_get(int device, chv_device_info *info);

/* ---- contexts: ComputeContext, compute.cl.swift:75-105 ------------------ */
typedef struct chv_context chv_context;

/* createComputeContext(_:logger:), compute.cl.swift:115-145.  One HIP stream
 * per context; all kernels are part of the library (no per-context build). */
int chv_context_create(int device, chv_context **out);
/* createComputeContext(sharing:), compute.cl.swift:111-113: same device and
 * allocator, a NEW stream (the reference makes a new command queue, :82-87). */
int chv_context_share(chv_context *parent, chv_context **out);
/* destroyComputeContext, compute.cl.swift:147-151 */
int chv_context_destroy(chv_context *ctx);
int chv_context_device(chv_context *ctx, int *device);
/* Raw hipStream_t of the context, for callers that interleave their own work. */
int chv_context_stream(chv_context *ctx, void **hip_stream);
/* NUMA node of the device's PCIe root complex (sysfs), -1 if unknown: where a host should pin its upload ring. */
int chv_context_numa_node(chv_context *ctx, int *node);

/* ---- device memory: ComputeBuffer, compute.cl.swift:46-58 ---------------- */
typedef struct chv_buffer chv_buffer;

/* createBuffer, compute.cl.swift:522-529 */
int chv_buffer_alloc(chv_context *ctx, size_t bytes, chv_buffer **out);
/* Adopt device memory owned by someone else (e.g. a decoder surface); never freed here. */
int chv_buffer_wrap(chv_context *ctx, void *device_ptr, size_t bytes, chv_buffer **out);
/* ComputeBuffer.deinit, compute.cl.swift:55-57; callable from any thread.  A buffer that kernels of a pass in progress name (chv_pass_begin)
 * is released when they have been launched; the call returns at once either way.  A second free of the same buffer is CHV_ERR_INVALID_VALUE
 * while the first is pending, undefined afterwards (as for any freed handle). */
int chv_buffer_free(chv_buffer *buf);
int chv_buffer_info(chv_buffer *buf, void **device_ptr, size_t *bytes);
/* One plane of createTexture (compute.cl.swift:532-581): `components` bytes
 * per texel (1 = R8, 2 = RG8, 4 = RGBA8).  Linear, pitch is 128-byte (cache line) aligned. */
int chv_plane_alloc(chv_context *ctx, int width, int height, int components,
                    chv_buffer **out, size_t *pitch);

/* uploadComputeBuffer / the per-plane clEnqueueWriteImage of
 * uploadComputePicture (compute.cl.swift:361-379, 434-452): pitched H2D copy.
 * async = 0: returns when the copy is complete (reference behaviour).
 * async = 1: the host bytes are staged into pinned memory before returning, so
 *            `src` is only borrowed for the call; the copy is ordered on the
 *            context's stream.
 * async = 2: `src` is pinned memory from chv_host_alloc that the caller leaves
 *            unchanged until the context's stream has passed the copy (e.g. the
 *            next chv_pass_end(wait)); no staging copy is made — the path for
 *            decoders that write straight into pinned frames.
 * After an asynchronous upload, kernels launched from ANY context of the device
 * that read the plane wait for the copy on their own stream (an event per
 * buffer); no host-side wait is needed between upload and use. */
int chv_upload(chv_context *ctx, chv_buffer *dst, size_t dst_offset, size_t dst_pitch,
               const void *src, size_t src_pitch, size_t width_bytes, size_t rows, int async);
/* Pinned (page-locked) host memory for async = 2 uploads. */
int chv_host_alloc(chv_context *ctx, size_t bytes, void **out);
int chv_host_free(chv_context *ctx, void *ptr);

/* downloadComputeBuffer / downloadComputePicture (compute.cl.swift:381-396,
 * 461-498): pitched D2H copy, always complete on return. */
int chv_download(chv_context *ctx, void *dst, size_t dst_pitch, chv_buffer *src,
                 size_t src_offset, size_t src_pitch, size_t width_bytes, size_t rows);
/* The same copy without the wait (the D2H half of GPUBarrierDownload running on a context of its own, compute.swift:217-255, so that
 * the read-back of tick t overlaps the kernels of tick t + 1): ordered on ctx's stream behind pending asynchronous uploads of `src`;
 * `dst` is pinned memory from chv_host_alloc and holds the picture once ctx's stream has passed the copy (chv_event_record +
 * chv_event_synchronize, or chv_pass_end(ctx, wait)).  Kernels of ANOTHER context that write `src` are ordered in front of the copy
 * with chv_event_record (there) + chv_event_wait (here).  Adjacent planes / frames of one allocation travel as one linear copy when
 * both pitches equal width_bytes. */
int chv_download_async(chv_context *ctx, void *dst, size_t dst_pitch, chv_buffer *src,
                       size_t src_offset, size_t src_pitch, size_t width_bytes, size_t rows);

/* ---- images: POD view of ImageBuffer.planes + computeTextures,
 *      sample.pict.linux.swift:23-72 ------------------------------------- */
typedef struct chv_plane {
    chv_buffer *buffer;
    size_t offset;        /* bytes from the start of `buffer` */
    int32_t width, height;/* texels (Plane.size)  */
    int32_t pitch;        /* bytes  (Plane.stride) */
    int32_t components;   /* Plane.components.count */
} chv_plane;

typedef struct chv_image {
    int32_t format;       /* chv_pixel_format */
    int32_t width, height;
    int32_t n_planes;
    chv_plane planes[3];
} chv_image;

/* ImageUniforms, compute.swift:76-86: 236 bytes, no padding.  Each matrix is
 * M.inverse.transpose as uploaded by applyComputeImage (compute.swift:149-161),
 * i.e. floats [4i..4i+3] are what the kernels dot with to get component i. */
typedef struct chv_uniforms {
    float transform[16];
    float texture_transform[16];
    float border_matrix[16];
    float fill_color[4];
    float input_size[2];
    float output_size[2];
    float opacity;
    float image_time;
    float target_time;
} chv_uniforms;

typedef struct chv_kernel_opts {
    int32_t colorspace;   /* chv_colorspace; YUV->BGRA kernels only */
    int32_t reserved[3];
} chv_kernel_opts;

/* Uniforms of the two kernels of `enum ComputeKernel` no caller of the reference dispatches (compute.swift:67,70); both run through
 * chv_run_kernel in the reference's bind order [outputs][inputs][uniforms] (compute.cl.swift:288-327):
 *   CHV_K_SND_S16I_S16I (kernels.cl.swift:534-562)  target: ONE plane of 2-byte texels = interleaved-stereo int16 samples (width x height
 *       samples; rows contiguous when height > 1), updated in place (out[gid] += ...); inputs: inputCount..8 images of the same shape;
 *       uniforms: the 100-byte chv_snd_uniforms.  A float product below -32768 wraps as on the CPU the kernel string was compiled for
 *       (OpenCL leaves it undefined; oracle/ref_kernels.c::snd_cvt).
 *   CHV_K_ME_FULLSEARCH (kernels.metal:129-267)  target: one RGBA8 texel per block, (mv.x, 0.5, mv.y, 1) normalised to [0, 1]; inputs[0] =
 *       reference picture, inputs[1] = current picture (plane 0 of each, 1-component: the luma plane of an NV12 / y420p picture);
 *       uniforms: the 24-byte chv_me_uniforms, blockSize 1..64.  Bit-for-bit the Metal source, its sliding-window SAD included. */
typedef struct chv_snd_uniforms {     /* BufferUniforms, kernels.cl.swift:536-541 */
    int32_t input_count;
    int32_t input_offsets[8];         /* (not read by the kernel) */
    float input_gains[8];
    float input_fade[8];
} chv_snd_uniforms;
typedef struct chv_me_uniforms {      /* MotionEstimationUniforms, kernels.metal:33-37 */
    int32_t block_size[2];
    int32_t search_window_size[2];
    int32_t image_size[2];
} chv_me_uniforms;

/* ---- compute passes ----------------------------------------------------- */
/* beginComputePass, compute.cl.swift:234-237.
 * Between chv_pass_begin and chv_pass_end the picture kernels issued through chv_run_kernel are ACCEPTED — every argument check runs in the
 * call and its error comes back from it — and HELD: nothing has to be visible before the pass ends (usingContext, compute.swift:131-134), so
 * `img_clear_* + N layer kernels on one target`, what an unchanged VideoMixer issues per tick (mix.video.swift:116-124), leaves as the ONE
 * launch chv_composite would have made of it: same bytes (that equality is chv_composite's definition), one launch instead of N + 1.  What is
 * held goes out, in issue order, at chv_pass_end — or before anything else that touches ctx's stream: an upload or download through ctx, a
 * batch, a custom or buffer kernel, an event, chv_context_stream, a kernel on another target, a clear after layers.  Buffers named by held
 * kernels may be passed to chv_buffer_free before the pass ends (a ComputeBuffer's deinit can run as soon as runComputeKernel returns,
 * compute.cl.swift:55-57): the free takes effect once they have been launched.  Work of OTHER contexts is ordered against a pass's kernels at the
 * pass's end, as against any kernel: events, or the per-buffer upload events.  Brackets nest (uploadComputePicture opens its own around its
 * copies, compute.cl.swift:433,453): kernels are held while any bracket is open, and every chv_pass_end launches what is held.  CHV_PASS_FUSE=0 (environment / chv_debug_set_switch) launches
 * every kernel in its call, as rounds 1-5 did. */
int chv_pass_begin(chv_context *ctx);
/* runComputeKernel (both overloads), compute.cl.swift:250-344.  Launch domain
 * is the target's plane-0 size (:329).  `inputs`/`n_inputs`: the images array;
 * `uniforms`/`uniforms_size`: the Swift struct bytes (236 for ImageUniforms,
 * 0/NULL for none); `blends`: bind the target again as the read-only current
 * image (:301-313).  `opts` may be NULL. */
int chv_run_kernel(chv_context *ctx, int kernel, const chv_image *target,
                   const chv_image *inputs, int n_inputs,
                   const void *uniforms, size_t uniforms_size, int blends,
                   const chv_kernel_opts *opts);
/* endComputePass, compute.cl.swift:346-359: launches what the pass holds (above; a launch error of those kernels is returned here), then
 * wait != 0 -> block until the stream is idle (clFinish), else just make sure work is submitted (clFlush). */
int chv_pass_end(chv_context *ctx, int wait);

/* ---- one mixer tick in one launch --------------------------------------- */
/* What VideoMixer.mix does per tick (mix.video.swift:116-124): clear the
 * backing image, then applyComputeImage for each layer in z order.
 * chv_composite produces byte-identical output to that sequence of
 * chv_run_kernel calls, but reads every layer once and writes the canvas once. */
typedef struct chv_layer {
    int32_t kernel;               /* img_<fmt>_<target fmt> */
    chv_image image;
    chv_uniforms uniforms;
    chv_kernel_opts opts;
} chv_layer;

/* Layers per launch of chv_composite; a deeper tick is issued as several launches
 * on the context's stream (same bytes: the canvas is 8-bit between layers anyway).
 * Ticks of a batch (chv_batch_create) may have any number of layers. */
#define CHV_MAX_LAYERS 16

int chv_composite(chv_context *ctx, const chv_image *target, int clear_first,
                  const chv_layer *layers, int n_layers);

/* Many independent ticks (streams / frames) in one launch: job i composites
 * layers[first_layer[i] .. first_layer[i]+n_layers[i]) onto targets[i].
 * A batch is immutable once created and can be run any number of times
 * (canvas and upload rings make the same descriptors recur every tick). */
typedef struct chv_batch chv_batch;
typedef struct chv_tick {
    chv_image target;
    int32_t clear_first;
    int32_t n_layers;
    const chv_layer *layers;
} chv_tick;
int chv_batch_create(chv_context *ctx, const chv_tick *ticks, int n_ticks, chv_batch **out);
int chv_batch_run(chv_context *ctx, chv_batch *batch);
int chv_batch_destroy(chv_batch *batch);
/* Name of the device kernel a batch dispatches to and its launch count (for profiling).  A batch whose ticks start with 2..4
 * full-frame videos of one geometry and go on with other layers runs as TWO launches on the context's stream ("tick_bgra_stream +
 * tick_bgra_wave": the videos, then the rest continuing on the canvas); the bytes are those of one pass. */
int chv_batch_describe(chv_batch *batch, char *kernel_name, size_t cap, int *n_launches);

/* ---- custom kernels ------------------------------------------------------ */
/* `ComputeKernel.custom(name:)` + buildComputeKernel (compute.swift:72-73,
 * compute.cl.swift:153-195, getComputeKernel :218-232): user source compiled at
 * run time and kept in the context's library under `name`.  Here the source
 * is HIP C++ compiled with hipRTC for the context's device; it is prefixed
 * with chv_custom_prelude() (the counterpart of kOpenCLKernelMatrixFuncs,
 * kernels.cl.swift:25-35, plus the image builtins OpenCL gives a kernel for
 * free) and must define
 *     extern "C" __global__ void <name>(chv_custom_args a)
 * The argument block carries what the reference binds positionally
 * (compute.cl.swift:288-335): the target planes, the same planes again as
 * `current` when `blends` (n_planes = 0 otherwise), the input images, the
 * uniforms' bytes.  Launch domain: one thread per texel of target plane 0 in
 * 16x16 blocks, rounded up — kernels test their coordinates (CHV_GUARD). */
#define CHV_CUSTOM_MAX_INPUTS 4
#define CHV_CUSTOM_MAX_UNIFORMS 256
typedef struct chv_dev_plane {
    uint8_t *ptr;                     /* device address of texel (0, 0) */
    int32_t width, height, pitch, components;
} chv_dev_plane;
typedef struct chv_dev_image {
    chv_dev_plane planes[3];
    int32_t n_planes, format;
} chv_dev_image;
typedef struct chv_custom_args {
    chv_dev_image target, current;
    chv_dev_image inputs[CHV_CUSTOM_MAX_INPUTS];
    int32_t n_inputs, uniforms_size;
    uint8_t uniforms[CHV_CUSTOM_MAX_UNIFORMS];
} chv_custom_args;
/* The text every custom source is prefixed with (struct definitions above, vecmat4, unorm8 load/store,
 * nearest and linear samplers with the semantics of the built-in kernels). */
const char *chv_custom_prelude(void);
/* buildComputeKernel.  Replaces an earlier kernel of the same name in this context's library; contexts made
 * with chv_context_share afterwards inherit the library.  Compile or lookup failure:
 * CHV_ERR_BAD_INPUT ("Unable to create kernel named ..."), build log in chv_last_error_detail(). */
int chv_kernel_build(chv_context *ctx, const char *name, const char *source);
/* runComputeKernel(kernel: .custom(name)).  CHV_ERR_KERNEL_NOT_FOUND if `name` is not in the library;
 * at most CHV_CUSTOM_MAX_INPUTS images and CHV_CUSTOM_MAX_UNIFORMS uniform bytes. */
int chv_run_custom(chv_context *ctx, const char *name, const chv_image *target,
                   const chv_image *inputs, int n_inputs,
                   const void *uniforms, size_t uniforms_size, int blends);

/* ---- resampling --------------------------------------------------------- */
/* Separable Lanczos-3 resize of a 4-component image (BGRA or RGBA) from `src`
 * to `dst` size.  No reference counterpart; DESIGN.md section 4.4.  Reductions whose 8 x 4 output tile needs more than 160 KB of staged
 * source (about 24:1 on one axis, about 17:1 on both at once) are refused with CHV_ERR_INVALID_VALUE: nothing is written. */
int chv_scale_lanczos(chv_context *ctx, const chv_image *dst, const chv_image *src);
/* n resizes of one geometry (every src of one size, every dst of one size) in one launch per CHV_LANCZOS_BATCH_CHUNK pairs;
 * same bytes as n calls of chv_scale_lanczos.  Other geometries in the list -> CHV_ERR_INVALID_VALUE, nothing is launched. */
#define CHV_LANCZOS_BATCH_CHUNK 64
int chv_scale_lanczos_batch(chv_context *ctx, const chv_image *dsts, const chv_image *srcs, int n);

/* ---- timing (what the "gpu.upload"/"mix.video.compose" StatsReport timers
 *      measure on the host, compute.swift:185-187, mix.video.swift:110-126,
 *      measured on the stream) ------------------------------------------- */
typedef struct chv_event chv_event;
int chv_event_create(chv_context *ctx, chv_event **out);
int chv_event_record(chv_context *ctx, chv_event *ev);
/* Make all later work of `ctx`'s stream wait for `ev` (recorded on any context of the device). */
int chv_event_wait(chv_context *ctx, chv_event *ev);
int chv_event_synchronize(chv_event *ev);
int chv_event_elapsed_ms(chv_event *start, chv_event *stop, float *ms);
int chv_event_destroy(chv_event *ev);
/* Block until every stream of the context's device is idle. */
int chv_device_synchronize(chv_context *ctx);

#ifdef __cplusplus
}
#endif
#endif /* CHIPVIDEO_H */

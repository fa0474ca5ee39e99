/*
 * oracle/ref_kernels.h — CPU restatement of the SwiftVideo pixel kernels.
 *
 * TEST INFRASTRUCTURE, NOT PRODUCT.  Only tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg may load this library, and only as the checker
 * (or as the timed CPU baseline).  Nothing under swiftvideo_amd/ links, loads
 * or calls it; the product path fails loudly if its HIP library is missing.
 *
 * PARITY STATUS: "parity unpinned".
 *   The reference (unpause-live/SwiftVideo) ships no numeric test, golden
 *   vector or fixture for this path (its only compute test is the kernel-name
 *   table, Tests/swiftVideoInternalTests/computeTests.swift:9-39, which
 *   tests/test_kernel_names.py replays).  The kernel bodies below follow the
 *   reference OpenCL-C / Metal sources line by line (citations at each
 *   function), but the image sampler arithmetic (read_imagef / write_imagef)
 *   lives in the OpenCL runtime, a third-party dependency that is neither
 *   vendored in the reference nor pinned to a version (Package.swift:40 links
 *   "OpenCL"; TestEnvironment.dockerfile:259 installs only the ICD loader).
 *   It is restated here from the published Khronos OpenCL 1.2 specification,
 *   section 8.2 (linear filtering, CLAMP_TO_EDGE, normalized coordinates) and
 *   section 8.3.1.1 (UNORM_INT8 conversion rules).
 *
 * All coordinates/arith are IEEE binary32 with no fused contraction (build
 * with -ffp-contract=off), operations in reference source order, dot() summed
 * left to right ((x+y)+z)+w as in the reference's only explicit definition
 * (kernels.cuda.swift:45-47).  The kernels whose specification this repository
 * owns (ids 32.., the BGRA-target family, and Lanczos-3) use explicit fmaf()
 * where their specification says "fused" (DESIGN.md section 4): the BGRA-target
 * family evaluates samples, fill and blend on the 0..255 code scale.
 */
#ifndef ORACLE_REF_KERNELS_H
#define ORACLE_REF_KERNELS_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Kernel ids.  0..14 follow the declaration order of the reference's
 * `enum ComputeKernel` (Sources/SwiftVideo/compute.swift:49-74).
 * 32.. are kernels the reference names (mix.video.swift:142-146 synthesises
 * "img_nv12_bgra") but does not implement; their spec is owned by this repo
 * (DESIGN.md section 4). */
enum {
    ORC_IMG_NV12_NV12 = 0,   /* kernels.cl.swift:47-109  */
    ORC_IMG_BGRA_NV12 = 1,   /* kernels.cl.swift:469-532 */
    ORC_IMG_RGBA_NV12 = 2,   /* kernels.cl.swift:405-467 */
    ORC_IMG_BGRA_BGRA = 3,   /* kernels.metal:52-62      */
    ORC_IMG_Y420P_Y420P = 4, /* kernels.cl.swift:186-255 */
    ORC_IMG_Y420P_NV12 = 5,  /* kernels.cl.swift:110-173 */
    ORC_IMG_CLEAR_NV12 = 6,  /* kernels.cl.swift:38-46   */
    ORC_IMG_CLEAR_YUVS = 7,  /* enum case only, no kernel anywhere */
    ORC_IMG_CLEAR_BGRA = 8,  /* kernels.cl.swift:257-265 */
    ORC_IMG_CLEAR_Y420P = 9, /* kernels.cl.swift:174-185 */
    ORC_IMG_CLEAR_RGBA = 10, /* compute.swift:101 maps the name to img_clear_bgra */
    ORC_IMG_RGBA_Y420P = 11, /* kernels.cl.swift:336-403 */
    ORC_IMG_BGRA_Y420P = 12, /* kernels.cl.swift:267-335 */
    ORC_SND_S16I_S16I = 13,  /* kernels.cl.swift:534-562: orc_snd_s16i_s16i (buffers, not images) */
    ORC_ME_FULLSEARCH = 14,  /* kernels.metal:129-267: orc_me_fullsearch */
    /* spec owned by this repo */
    ORC_IMG_NV12_BGRA = 32,
    ORC_IMG_Y420P_BGRA = 33,
    ORC_IMG_BGRA_BGRA_TX = 34,
    ORC_IMG_RGBA_BGRA_TX = 35,
    /* integer BT.601/709 RGB -> YUV onto 4:2:0 canvases (the encoder side) */
    ORC_IMG_BGRA_NV12_INT = 36,
    ORC_IMG_RGBA_NV12_INT = 37,
    ORC_IMG_BGRA_Y420P_INT = 38,
    ORC_IMG_RGBA_Y420P_INT = 39,
    /* envelope evaluators of 32..35 on the unit scale, in the reference family's style
     * (tests only; see px_to_bgra_unit) */
    ORC_ENV_NV12_BGRA_UNIT = 64,
    ORC_ENV_Y420P_BGRA_UNIT = 65,
    ORC_ENV_BGRA_BGRA_UNIT = 66,
    ORC_ENV_RGBA_BGRA_UNIT = 67
};

enum { ORC_OK = 0, ORC_ERR_INVALID_VALUE = 1, ORC_ERR_NOT_IMPLEMENTED = 6,
       ORC_ERR_BAD_TARGET = 4, ORC_ERR_BAD_INPUT = 5 };

/* YUV->RGB integer matrices for the 32/33 kernels. */
enum { ORC_CSC_BT601_LIMITED = 0, ORC_CSC_BT709_LIMITED = 1,
       ORC_CSC_BT601_FULL = 2, ORC_CSC_BT709_FULL = 3 };

/* One 8-bit UNORM plane in host memory: `comps` interleaved bytes per texel
 * (1 = CL_R, 2 = CL_RG, 4 = CL_RGBA; compute.cl.swift:545-558). */
typedef struct {
    uint8_t *data;
    int32_t w, h;      /* texels */
    int32_t pitch;     /* bytes per row */
    int32_t comps;
} orc_plane;

/* The 236-byte ImageUniforms blob (compute.swift:76-86; kernels.cl.swift:49-59). */
typedef struct {
    float transform[16];
    float textureTx[16];
    float borderMatrix[16];
    float fillColor[4];
    float inSize[2];
    float outSize[2];
    float opacity;
    float sampleTime;
    float targetTime;
} orc_uniforms;

/* Run one kernel over the whole target (global size = target plane 0 size,
 * compute.cl.swift:329-335).  `target` planes are read as `cur*` and written
 * as `out*` in place (blends:true binds the same memory twice,
 * compute.cl.swift:288-313).  `uniforms` may be NULL for the clear kernels.
 * `threads` row-partitions the work (>=1). */
int orc_run_kernel(int kernel, const orc_plane *target, int n_target,
                   const orc_plane *inputs, int n_inputs,
                   const orc_uniforms *uniforms, int csc, int threads);

/* Separable Lanczos-3 resample of a 4-component plane (spec owned by this
 * repo).  Coefficient tables come from
 * orc_lanczos_table(). */
int orc_lanczos_table(int in_size, int out_size, int *taps_out,
                      int32_t *first /*[out_size]*/, float *weights /*[out_size*taps]*/,
                      int max_taps);
int orc_lanczos_bgra(const orc_plane *dst, const orc_plane *src, int threads);

/* Single-pixel helpers exported for exhaustive unit tests. */
uint8_t orc_store_unorm8(float f);
float orc_load_unorm8(uint8_t c);
void orc_yuv2rgb_int(int csc, uint8_t y, uint8_t u, uint8_t v, uint8_t rgb[3]);
void orc_rgb2yuv_int(int csc, uint8_t r, uint8_t g, uint8_t b, uint8_t yuv[3]);

/* The two idle kernels (compute.swift:67,70).  BufferUniforms, kernels.cl.swift:536-541; MotionEstimationUniforms, kernels.metal:33-37. */
typedef struct { int32_t inputCount; int32_t inputOffsets[8]; float inputGains[8]; float inputFade[8]; } orc_snd_uniforms;
typedef struct { int32_t blockSize[2], searchWindowSize[2], imageSize[2]; } orc_me_uniforms;
int orc_snd_s16i_s16i(int16_t *out, int n, const int16_t *const *in, const orc_snd_uniforms *u);
/* out: one RGBA8 texel per block (w = blocks across, h = blocks down); ref, cur: 1-component planes */
int orc_me_fullsearch(const orc_plane *out, const orc_plane *ref, const orc_plane *cur, const orc_me_uniforms *u);
/* per-component motion-vector cost for |v| = 0 .. n-1 (deltaCost2, kernels.metal:135-142) through the host's log2f */
void orc_me_cost_table(float *table, int n);

#ifdef __cplusplus
}
#endif
#endif

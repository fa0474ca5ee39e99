// device_types.h — descriptors shared by the host runtime and the gfx950 kernels.
//
// A "tick" is one VideoMixer.mix pass (mix.video.swift:116-124): an optional
// clear of the canvas followed by n_layers applyComputeImage calls in z order.
// Descriptors live in device memory; a kernel block reads the one it owns with
// wave-uniform (scalar) loads.
#pragma once
#include <stdint.h>

namespace chv {

struct DPlane {
    uint8_t *ptr;
    int32_t w, h;      // texels
    int32_t pitch;     // bytes
    int32_t comps;     // bytes per texel
};

struct DImage {
    DPlane pl[3];
};

// How a layer is applied; derived on the host from (kernel id, target format).
enum LayerKind : int32_t {
    LK_YUV_FROM_NV12 = 0,   // img_nv12_nv12
    LK_YUV_FROM_Y420P = 1,  // img_y420p_nv12 / img_y420p_y420p
    LK_YUV_FROM_RGB = 2,    // img_{bgra,rgba}_{nv12,y420p}; swizzle flag for bgra
    LK_BGRA_FROM_NV12 = 3,  // img_nv12_bgra
    LK_BGRA_FROM_Y420P = 4, // img_y420p_bgra
    LK_BGRA_FROM_RGB = 5,   // img_{bgra,rgba}_bgra_tx; swizzle flag for rgba
    LK_BGRA_METAL = 6,      // img_bgra_bgra, kernels.metal:52-62
    LK_YUV_FROM_RGB_INT = 7 // img_{bgra,rgba}_{nv12,y420p}_int: integer BT.601/709 RGB -> YUV, code-scale blend; swizzle flag for bgra
};

struct DLayer {
    DImage src;
    float u[59];        // ImageUniforms, compute.swift:76-86
    int32_t kind;       // LayerKind
    int32_t swizzle;    // 1: exchange byte 0 and byte 2 of the sampled texel
    int32_t csc;        // chv_colorspace
    int32_t flags;      // LF_* below (host-side analysis; fast paths only)
    // Conservative canvas-pixel bounding box [x0, x1) x [y0, y1) of the layer's border quad
    // (host-side, a few pixels of margin): pixels outside it fail the border test for sure,
    // so kernels may skip the layer there without evaluating the geometry.
    int32_t bbox[4];
    int32_t pad;
    // LF_COVERS layers: canvas pixels [x0, x1) x [y0, y1) that lie inside border, transform AND texture range for sure (host-side, a few pixels
    // of margin): there the layer's opaque sample replaces whatever is beneath, so a strip inside this box starts at this layer.
    int32_t ibox[4];
    int32_t pad2[2];
};
static_assert(sizeof(DLayer) % 8 == 0, "DLayer arrays follow a DTick in descriptor slots and kernel arguments");

enum LayerFlags : int32_t {
    LF_AXIS_ALIGNED = 1,  // no rotation/shear: tx.x/uv.x depend on x only, tx.y/uv.y on y only
    LF_NO_FILL = 2,       // opacity * fillColor.w == 0 exactly
    LF_OPAQUE = 4,        // opacity == 1 exactly
    LF_BOUNDED = 8,       // all 48 matrix entries finite and < 2^60 in magnitude: no product in the prologue overflows
    // Same geometry as the layer below it in the tick: the three matrices are bit-identical, the source planes have the same
    // sizes, pitches and layout class, the bounding boxes are equal (only opacity, fill colour, times, colourspace and the plane
    // POINTERS may differ).  Every geometry value the strip kernels derive (column entries, row table, rectangles, slot maps) is
    // a function of exactly these inputs, so they keep the predecessor's — same inputs, same bits.  True for every layer but
    // the first of the BASELINE composite configurations (N full-canvas pictures of one size: mix.video.swift:114-124).
    LF_SAME_GEOM = 16,
    // An opaque picture without per-pixel alpha (a YUV source, opacity 1) whose inner box `ibox` is not empty: inside it the layers beneath do not
    // show — a picture-in-picture inset, the quadrants of a grid over a background
    LF_COVERS = 32
};

enum TargetFormat : int32_t { TF_NV12 = 0, TF_Y420P = 1, TF_BGRA = 2 };

struct DTick {
    DImage dst;
    int32_t W, H;          // launch domain = plane 0 size (compute.cl.swift:329)
    int32_t clear_first;
    int32_t n_layers;
    int32_t first_layer;   // index into the batch's DLayer array
    int32_t cover_mask;    // bit l: layer l of the tick is flagged LF_COVERS (most ticks: 0 — the strip kernels look no further)
};

// A lone tick and its layers as ONE kernel argument of the strip kernels (wave_common.hip.h: wave_one_descriptors): a transient launch of up to
// WAVE_ONE_LAYERS layers carries its descriptors in the kernarg segment instead of copying them to device memory in front of the kernel.
constexpr int WAVE_ONE_LAYERS = 6;
struct WaveOne {
    DTick t;
    DLayer l[WAVE_ONE_LAYERS];
};
static_assert(sizeof(WaveOne) <= 3072, "kernel arguments: 4 KB in all");

// uniforms blob offsets (floats)
enum { U_TRANSFORM = 0, U_TEXTURE = 16, U_BORDER = 32, U_FILL = 48, U_INSIZE = 52,
       U_OUTSIZE = 54, U_OPACITY = 56, U_IMAGETIME = 57, U_TARGETTIME = 58 };

}  // namespace chv

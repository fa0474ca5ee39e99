/*
 * oracle/ref_kernels.c — CPU restatement of the SwiftVideo pixel kernels.
 *
 * TEST INFRASTRUCTURE, NOT PRODUCT (see ref_kernels.h).  PARITY UNPINNED: the
 * reference has no numeric vectors for this path; kernel bodies follow the
 * reference sources cited at each function, the image sampler follows the
 * Khronos OpenCL 1.2 specification text (sections 8.2, 8.3.1.1).
 *
 * Build: gcc -O3 -march=native -ffp-contract=off -fno-fast-math -fPIC -shared
 *        -pthread ref_kernels.c -lm        (see oracle/Makefile)
 */
#include "ref_kernels.h"

#include <math.h>
#include <pthread.h>
#include <stdlib.h>
#include <string.h>


#define INL static inline __attribute__((always_inline))

typedef struct { float x, y, z, w; } f4;

/* ------------------------------------------------------------------------ */
/* OpenCL builtins used by the kernels                                       */
/* ------------------------------------------------------------------------ */

/* dot(float4,float4): summation order as written out in the reference's CUDA
 * prelude, kernels.cuda.swift:45-47. */
INL float dot4(f4 a, const float *row) {
    return ((a.x * row[0] + a.y * row[1]) + a.z * row[2]) + a.w * row[3];
}

/* vecmat4 macro, kernels.cl.swift:27 */
INL f4 vecmat4(f4 v, const float *m) {
    f4 r = { dot4(v, m + 0), dot4(v, m + 4), dot4(v, m + 8), dot4(v, m + 12) };
    return r;
}

INL float clampf(float v, float lo, float hi) { return fminf(fmaxf(v, lo), hi); }
INL int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

/* UNORM_INT8 -> float, OpenCL 1.2 section 8.3.1.1: c / 255.0f */
float orc_load_unorm8(uint8_t c) { return (float)c / 255.0f; }
INL float ld8(uint8_t c) { return (float)c / 255.0f; }

/* float -> UNORM_INT8, OpenCL 1.2 section 8.3.1.1:
 * convert_uchar_sat_rte(f * 255.0f); NaN converts to 0. */
INL uint8_t st8(float f) {
    float v = f * 255.0f;
    if (!(v == v)) return 0;
    v = rintf(v); /* default rounding mode: to nearest even */
    if (v <= 0.0f) return 0;
    if (v >= 255.0f) return 255;
    return (uint8_t)(int)v;
}
uint8_t orc_store_unorm8(float f) { return st8(f); }

/* RN(1/255), the alpha scale of the code-scale family */
#define ORC_INV255 0x1.010102p-8f

/* Same conversion for a value already on the 0..255 code scale. */
INL uint8_t st8_code(float v) {
    if (!(v == v)) return 0;
    v = rintf(v);
    if (v <= 0.0f) return 0;
    if (v >= 255.0f) return 255;
    return (uint8_t)(int)v;
}

/* read_imagef(img, CLK_NORMALIZED_COORDS_FALSE|CLK_ADDRESS_NONE|CLK_FILTER_NEAREST, int2)
 * — `curSampler`, kernels.cl.swift:62.  Out-of-range coordinates are undefined
 * with ADDRESS_NONE; they only occur for gid/2 on odd-sized 4:2:0 canvases and
 * are defined here (and in the HIP path) to read zero. */
INL const uint8_t *texel(const orc_plane *p, int x, int y) {
    return p->data + (size_t)y * (size_t)p->pitch + (size_t)x * (size_t)p->comps;
}
INL int inside(const orc_plane *p, int x, int y) {
    return x >= 0 && y >= 0 && x < p->w && y < p->h;
}
INL float rd_near1(const orc_plane *p, int x, int y, int c) {
    if (!inside(p, x, y)) return 0.0f;
    return ld8(texel(p, x, y)[c]);
}

/* Linear filter coordinates, OpenCL 1.2 section 8.2 (normalized coords,
 * CLAMP_TO_EDGE, LINEAR): u = s*w; i0 = floor(u-0.5); i1 = i0+1;
 * a = frac(u-0.5); addresses clamped to [0, w-1]. */
typedef struct { int i0, i1; float a; } lin1;
INL lin1 lin_coord(float s, int w) {
    lin1 r;
    float u = s * (float)w;
    float um = u - 0.5f;
    float fl = floorf(um);
    r.a = um - fl;
    int i = (int)fl;
    r.i0 = clampi(i, 0, w - 1);
    r.i1 = clampi(i + 1, 0, w - 1);
    return r;
}

/* read_imagef(img, `sampler` = NORMALIZED|CLAMP_TO_EDGE|LINEAR, float2),
 * kernels.cl.swift:61.  T = (1-a)(1-b)T00 + a(1-b)T10 + (1-a)b T01 + ab T11,
 * summed in that order.  Returns component c of the filtered texel. */
typedef struct { lin1 x, y; float w00, w10, w01, w11; } lin2;
INL lin2 lin_setup(const orc_plane *p, float s, float t) {
    lin2 l;
    l.x = lin_coord(s, p->w);
    l.y = lin_coord(t, p->h);
    float ia = 1.0f - l.x.a, ib = 1.0f - l.y.a;
    l.w00 = ia * ib;
    l.w10 = l.x.a * ib;
    l.w01 = ia * l.y.a;
    l.w11 = l.x.a * l.y.a;
    return l;
}
INL float lin_fetch(const orc_plane *p, const lin2 *l, int c) {
    float t00 = ld8(texel(p, l->x.i0, l->y.i0)[c]);
    float t10 = ld8(texel(p, l->x.i1, l->y.i0)[c]);
    float t01 = ld8(texel(p, l->x.i0, l->y.i1)[c]);
    float t11 = ld8(texel(p, l->x.i1, l->y.i1)[c]);
    return ((l->w00 * t00 + l->w10 * t10) + l->w01 * t01) + l->w11 * t11;
}

/* write_imagef: out-of-range writes are dropped. */
INL void wr1(const orc_plane *p, int x, int y, int c, float v) {
    if (inside(p, x, y)) ((uint8_t *)texel(p, x, y))[c] = st8(v);
}

/* rgb2yuv rows, kernels.cl.swift:96-99 (note 0.113, not BT.601's 0.114) */
static const float RGB2YUV[16] = {
    0.299f, 0.587f, 0.113f, 0.f,
    -0.169f, -0.331f, 0.5f, 0.5f,
    0.5f, -0.419f, -0.081f, 0.5f,
    0.f, 0.f, 0.f, 1.f };

/* ------------------------------------------------------------------------ */
/* Integer YUV -> RGB (spec owned by this repo)        */
/* 16.16 fixed point; the BT.601 limited-range row is the widely published    */
/* {76309, 104597, 25675, 53279, 132201} set.                                 */
/* ------------------------------------------------------------------------ */
typedef struct { int32_t yoff, cy, crv, cgu, cgv, cbu; } csc_t;
static const csc_t CSC[4] = {
    { 16, 76309, 104597, 25675, 53279, 132201 }, /* BT.601 limited */
    { 16, 76309, 117489, 13975, 34925, 138438 }, /* BT.709 limited */
    { 0, 65536, 91881, 22553, 46802, 116130 },   /* BT.601 full    */
    { 0, 65536, 103206, 12276, 30679, 121609 },  /* BT.709 full    */
};
INL uint8_t clip8(int32_t v) { return (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v)); }
INL void yuv2rgb_int(const csc_t *k, int y, int u, int v, uint8_t *r, uint8_t *g, uint8_t *b) {
    int32_t c = k->cy * (y - k->yoff) + 32768;
    int32_t d = u - 128, e = v - 128;
    /* >> on negative values is an arithmetic shift (floor) */
    *r = clip8((c + k->crv * e) >> 16);
    *g = clip8((c - k->cgu * d - k->cgv * e) >> 16);
    *b = clip8((c + k->cbu * d) >> 16);
}
void orc_yuv2rgb_int(int csc, uint8_t y, uint8_t u, uint8_t v, uint8_t rgb[3]) {
    yuv2rgb_int(&CSC[csc & 3], y, u, v, &rgb[0], &rgb[1], &rgb[2]);
}

/* ------------------------------------------------------------------------ */
/* Integer RGB -> YUV (spec owned by this repo)        */
/* 16.16 fixed point, the transpose of the matrices above: coefficients       */
/* round(K * 65536 * 219/255) (luma) / round(K * 65536 * 224/255) (chroma)    */
/* for limited range, green adjusted so that every luma row sums to           */
/* round(65536 * 219/255) = 56284 (65536 full range) and every chroma row     */
/* to 0 exactly: white is (235, 128, 128) / (255, 128, 128), grey has         */
/* neutral chroma, and the BT.601 limited row gives the colour-bar values     */
/* red (81, 90, 240), green (145, 54, 34), blue (41, 240, 110).               */
/* ------------------------------------------------------------------------ */
typedef struct { int32_t yoff, y[3], u[3], v[3]; } r2y_t;
static const r2y_t R2Y[4] = {
    { 16, { 16829, 33039, 6416 }, { -9714, -19070, 28784 }, { 28784, -24103, -4681 } },  /* BT.601 limited */
    { 16, { 11966, 40254, 4064 }, { -6596, -22188, 28784 }, { 28784, -26145, -2639 } },  /* BT.709 limited */
    { 0, { 19595, 38470, 7471 }, { -11058, -21710, 32768 }, { 32768, -27439, -5329 } },  /* BT.601 full    */
    { 0, { 13933, 46871, 4732 }, { -7509, -25259, 32768 }, { 32768, -29763, -3005 } },   /* BT.709 full    */
};
INL void rgb2yuv_int(const r2y_t *k, int r, int g, int b, uint8_t *y, uint8_t *u, uint8_t *v) {
    *y = clip8((k->y[0] * r + k->y[1] * g + k->y[2] * b + (k->yoff << 16) + 32768) >> 16);
    *u = clip8((k->u[0] * r + k->u[1] * g + k->u[2] * b + (128 << 16) + 32768) >> 16);
    *v = clip8((k->v[0] * r + k->v[1] * g + k->v[2] * b + (128 << 16) + 32768) >> 16);
}
void orc_rgb2yuv_int(int csc, uint8_t r, uint8_t g, uint8_t b, uint8_t yuv[3]) {
    rgb2yuv_int(&R2Y[csc & 3], r, g, b, &yuv[0], &yuv[1], &yuv[2]);
}

/* ------------------------------------------------------------------------ */
/* Kernel bodies                                                             */
/* ------------------------------------------------------------------------ */
enum { SRC_NV12, SRC_Y420P, SRC_BGRA, SRC_RGBA };
enum { DST_NV12, DST_Y420P, DST_BGRA };

typedef struct {
    int kernel, csc;
    const orc_plane *t;   /* target planes (out == cur) */
    const orc_plane *in;  /* input planes */
    const orc_uniforms *u;
    int W, H;
} job_t;

/* Shared geometry prologue of the OpenCL composite family
 * (identical in all seven kernels; kernels.cl.swift:70-77 = :495-502). */
typedef struct { f4 tx, border, uv; int in_border, in_tx, in_uv; } geom_t;
INL geom_t geometry(const job_t *j, int x, int y) {
    geom_t g;
    float sx = (float)j->W, sy = (float)j->H;       /* get_global_size */
    float ou = (float)x / sx, ov = (float)y / sy;   /* out_uv, not pixel-centred */
    f4 np = { ou * 2.f - 1.f, ov * 2.f - 1.f, 0.f, 1.f };
    g.tx = vecmat4(np, j->u->transform);
    g.border = vecmat4(np, j->u->borderMatrix);
    g.in_border = g.border.x >= 0.f && g.border.y >= 0.f && g.border.x <= 1.f && g.border.y <= 1.f;
    g.uv = vecmat4(g.tx, j->u->textureTx);
    g.in_tx = g.tx.x >= 0.0f && g.tx.y >= 0.0f && g.tx.x <= 1.0f && g.tx.y <= 1.0f;
    g.in_uv = g.uv.x >= 0.f && g.uv.y >= 0.f && g.uv.x <= 1.f && g.uv.y <= 1.f;
    return g;
}

/* current chroma of the canvas at gid/2 */
INL void read_cur_chroma(int DST, const job_t *j, int cx, int cy, float *cu, float *cv) {
    if (DST == DST_NV12) { *cu = rd_near1(&j->t[1], cx, cy, 0); *cv = rd_near1(&j->t[1], cx, cy, 1); }
    else { *cu = rd_near1(&j->t[1], cx, cy, 0); *cv = rd_near1(&j->t[2], cx, cy, 0); }
}
INL void write_chroma(int DST, const job_t *j, int cx, int cy, float u, float v) {
    if (DST == DST_NV12) { wr1(&j->t[1], cx, cy, 0, u); wr1(&j->t[1], cx, cy, 1, v); }
    else { wr1(&j->t[1], cx, cy, 0, u); wr1(&j->t[2], cx, cy, 0, v); }
}

/* YUV-source kernels: img_nv12_nv12 (kernels.cl.swift:63-108),
 * img_y420p_nv12 (:126-172), img_y420p_y420p (:202-254). */
INL void px_yuv_to_yuv(int SRC, int DST, const job_t *j, int x, int y) {
    geom_t g = geometry(j, x, y);
    int hc = (x % 2) == 0 && (y % 2) == 0;              /* handleChroma */
    if (!g.in_border) return;
    float curY = rd_near1(&j->t[0], x, y, 0);
    float curU = 0.f, curV = 0.f;
    if (hc) read_cur_chroma(DST, j, x / 2, y / 2, &curU, &curV);
    if (g.in_tx && g.in_uv) {
        lin2 ly = lin_setup(&j->in[0], g.uv.x, g.uv.y);
        float luma = lin_fetch(&j->in[0], &ly, 0);
        float alpha = j->u->opacity;
        wr1(&j->t[0], x, y, 0, curY * (1.f - alpha) + luma * alpha);
        if (hc) {
            float cb, cr;
            /* chroma planes are sampled at the SAME normalized uv */
            lin2 lc = lin_setup(&j->in[1], g.uv.x, g.uv.y);
            if (SRC == SRC_NV12) { cb = lin_fetch(&j->in[1], &lc, 0); cr = lin_fetch(&j->in[1], &lc, 1); }
            else {
                cb = lin_fetch(&j->in[1], &lc, 0);
                lin2 lv = lin_setup(&j->in[2], g.uv.x, g.uv.y);
                cr = lin_fetch(&j->in[2], &lv, 0);
            }
            write_chroma(DST, j, x / 2, y / 2,
                         curU * (1.f - alpha) + cb * alpha,
                         curV * (1.f - alpha) + cr * alpha);
        }
        return;
    }
    /* inside the border quad but outside the picture: paint the fill colour */
    f4 fc_in = { j->u->fillColor[0], j->u->fillColor[1], j->u->fillColor[2], 1.0f };
    f4 fc = vecmat4(fc_in, RGB2YUV);
    float alpha = j->u->opacity * j->u->fillColor[3];
    wr1(&j->t[0], x, y, 0, clampf(curY * (1.f - alpha) + fc.x * alpha, 0.f, 1.f));
    if (hc) {
        write_chroma(DST, j, x / 2, y / 2,
                     clampf(curU * (1.f - alpha) + fc.y * alpha, -1.f, 1.f),
                     clampf(curV * (1.f - alpha) + fc.z * alpha, -1.f, 1.f));
    }
}

/* RGB-source kernels: img_bgra_nv12 (kernels.cl.swift:485-531),
 * img_rgba_nv12 (:421-466), img_bgra_y420p (:283-334), img_rgba_y420p (:352-402). */
INL void px_rgb_to_yuv(int SRC, int DST, const job_t *j, int x, int y) {
    geom_t g = geometry(j, x, y);
    int hc = (x % 2) == 0 && (y % 2) == 0;
    if (!g.in_border) return;
    float curY = rd_near1(&j->t[0], x, y, 0);
    float curU = 0.f, curV = 0.f;  /* uninitialised in the reference when !hc; never stored then */
    if (hc) read_cur_chroma(DST, j, x / 2, y / 2, &curU, &curV);
    if (!g.in_tx) return;          /* nothing is written outside tx (:509-529) */
    float alpha = j->u->opacity * j->u->fillColor[3];
    f4 fin = { j->u->fillColor[0] * alpha, j->u->fillColor[1] * alpha, j->u->fillColor[2] * alpha, 1.0f };
    f4 fc = vecmat4(fin, RGB2YUV);
    float rx = curY * (1.f - alpha) + fc.x * alpha;
    float ry = clampf(curU * (1.f - alpha) + fc.y * alpha, -1.f, 1.f);
    float rz = clampf(curV * (1.f - alpha) + fc.z * alpha, -1.f, 1.f);
    if (g.in_uv) {
        lin2 l = lin_setup(&j->in[0], g.uv.x, g.uv.y);
        float p0 = lin_fetch(&j->in[0], &l, 0), p1 = lin_fetch(&j->in[0], &l, 1);
        float p2 = lin_fetch(&j->in[0], &l, 2), p3 = lin_fetch(&j->in[0], &l, 3);
        /* Linux textures are CL_RGBA for both byte orders (compute.cl.swift:548-558);
         * the bgra kernels swizzle .zyxw (kernels.cl.swift:518) */
        float r = SRC == SRC_BGRA ? p2 : p0, gch = p1, b = SRC == SRC_BGRA ? p0 : p2;
        float a2 = p3 * j->u->opacity;
        f4 pin = { r * a2, gch * a2, b * a2, 1.0f };
        f4 yuv = vecmat4(pin, RGB2YUV);
        rx = rx * (1.f - a2) + yuv.x * a2;
        ry = ry * (1.f - a2) + yuv.y * a2;
        rz = rz * (1.f - a2) + yuv.z * a2;
    }
    wr1(&j->t[0], x, y, 0, rx);
    if (hc) write_chroma(DST, j, x / 2, y / 2, ry, rz);
}

/* Metal img_bgra_bgra, kernels.metal:52-62: nearest scale by inputSize/outputSize
 * (truncation), source-alpha "over", output alpha forced to 1, transform and
 * opacity ignored. Channel-symmetric, so it works on memory-order bytes. */
INL void px_bgra_bgra_metal(const job_t *j, int x, int y) {
    float rx = j->u->inSize[0] / j->u->outSize[0];
    float ry = j->u->inSize[1] / j->u->outSize[1];
    float ipx = (float)x * rx, ipy = (float)y * ry;
    int sx = clampi((int)ipx, 0, j->in[0].w - 1);   /* uint2(inPos) */
    int sy = clampi((int)ipy, 0, j->in[0].h - 1);
    const uint8_t *s = texel(&j->in[0], sx, sy);
    uint8_t *d = (uint8_t *)texel(&j->t[0], x, y);
    float a = ld8(s[3]);
    float ia = 1.0f - a;
    for (int c = 0; c < 3; c++) d[c] = st8(ld8(s[c]) * a + ld8(d[c]) * ia);
    d[3] = st8(1.0f);
}

/* BGRA-target family (spec owned by this repo):
 * the structure of the reference's YUV-source kernels (kernels.cl.swift:78-105)
 * with the canvas in BGRA, per-pixel source alpha as in kernels.metal:59 and
 * the YUV->RGB step done in integer on the quantised sample.
 *
 * Arithmetic of this family is specified on the 0..255 CODE scale with fused
 * multiply-adds (one rounding each, fmaf), not on the unit scale of the
 * reference's OpenCL kernels:
 *   sample_c = fma(w11,T11, fma(w01,T01, fma(w10,T10, w00*T00)))   T = texel bytes as floats,
 *              geometry, tap addresses and weights exactly those of the LINEAR sampler above
 *   YUV sources: codes = RTE(sample) per plane -> integer matrix -> p_c = B,G,R bytes
 *   RGB sources: p_c = sample_c (not rounded), a = sample_A * (opacity * (1/255))
 *   fill:   r_c = clamp(fma(fill_c*255, af, cur_c*(1-af)), 0, 255),  af = opacity*fill_A
 *   blend:  r_c = fma(p_c, a, r_c*(1-a))
 *   store:  RTE, saturated, NaN -> 0; alpha byte 255
 * It differs from the unit-scale evaluation (c/255 per tap, sequential roundings,
 * *255 at the store) by at most one code of the sampled value, at rounding ties
 * (tests/test_oracle_golden.py::test_code_scale_sampler_vs_unit_scale). */
INL float cs_fetch(const orc_plane *p, const lin2 *l, int c) {
    float t00 = (float)texel(p, l->x.i0, l->y.i0)[c];
    float t10 = (float)texel(p, l->x.i1, l->y.i0)[c];
    float t01 = (float)texel(p, l->x.i0, l->y.i1)[c];
    float t11 = (float)texel(p, l->x.i1, l->y.i1)[c];
    return fmaf(l->w11, t11, fmaf(l->w01, t01, fmaf(l->w10, t10, l->w00 * t00)));
}
INL void px_to_bgra(int SRC, const job_t *j, int x, int y) {
    geom_t g = geometry(j, x, y);
    if (!g.in_border) return;
    uint8_t *d = (uint8_t *)texel(&j->t[0], x, y);
    /* fill colour under the picture / on the border, straight alpha (:96-105) */
    float af = j->u->opacity * j->u->fillColor[3];
    float iaf = 1.f - af;
    float r0 = clampf(fmaf(j->u->fillColor[2] * 255.0f, af, (float)d[0] * iaf), 0.f, 255.f); /* B */
    float r1 = clampf(fmaf(j->u->fillColor[1] * 255.0f, af, (float)d[1] * iaf), 0.f, 255.f); /* G */
    float r2 = clampf(fmaf(j->u->fillColor[0] * 255.0f, af, (float)d[2] * iaf), 0.f, 255.f); /* R */
    if (g.in_tx && g.in_uv) {
        float p0, p1, p2, a;
        if (SRC == SRC_BGRA || SRC == SRC_RGBA) {
            lin2 l = lin_setup(&j->in[0], g.uv.x, g.uv.y);
            float q0 = cs_fetch(&j->in[0], &l, 0), q1 = cs_fetch(&j->in[0], &l, 1);
            float q2 = cs_fetch(&j->in[0], &l, 2), q3 = cs_fetch(&j->in[0], &l, 3);
            p0 = SRC == SRC_BGRA ? q0 : q2; p1 = q1; p2 = SRC == SRC_BGRA ? q2 : q0;
            a = q3 * (j->u->opacity * ORC_INV255);
        } else {
            lin2 ly = lin_setup(&j->in[0], g.uv.x, g.uv.y);
            lin2 lc = lin_setup(&j->in[1], g.uv.x, g.uv.y);
            float fy = cs_fetch(&j->in[0], &ly, 0), fu, fv;
            if (SRC == SRC_NV12) { fu = cs_fetch(&j->in[1], &lc, 0); fv = cs_fetch(&j->in[1], &lc, 1); }
            else {
                fu = cs_fetch(&j->in[1], &lc, 0);
                lin2 lv = lin_setup(&j->in[2], g.uv.x, g.uv.y);
                fv = cs_fetch(&j->in[2], &lv, 0);
            }
            /* quantise as a store to an 8-bit YUV image would */
            uint8_t R, G, B;
            yuv2rgb_int(&CSC[j->csc & 3], st8_code(fy), st8_code(fu), st8_code(fv), &R, &G, &B);
            p0 = (float)B; p1 = (float)G; p2 = (float)R;
            a = 1.0f * j->u->opacity;
        }
        float ia = 1.f - a;
        r0 = fmaf(p0, a, r0 * ia);
        r1 = fmaf(p1, a, r1 * ia);
        r2 = fmaf(p2, a, r2 * ia);
    }
    d[0] = st8_code(r0); d[1] = st8_code(r1); d[2] = st8_code(r2); d[3] = 255;
}

/* Integer RGB -> YUV family (spec owned by this repo): img_{bgra,rgba}_{nv12,y420p}_int, the mirror
 * image of px_to_bgra for the encoder side (BGRA canvas -> 4:2:0 picture for x264, composer.swift:52-56,
 * enc.video.ffmpeg.swift:211-224).  Geometry, tap addresses and weights are the composite family's (geometry(), lin_setup);
 * chroma belongs to the quad's even/even pixel and is computed from THAT pixel's sample (`handleChroma`,
 * kernels.cl.swift:76 — with the reference's non-centred out_uv a same-size layer samples at gid - 0.5, i.e. the chroma
 * sample already is a 2 x 2 box average of the source).  Arithmetic on the code scale with fused multiply-adds, like 4.1:
 *   fill:   F = rgb2yuv_int(RTE_sat(fill.rgb * 255));  r_k = clamp(fma(F_k, af, cur_k * (1 - af)), 0, 255),  af = opacity * fill.a
 *   sample: s_c = fma(w11,T11, fma(w01,T01, fma(w10,T10, w00*T00)));  (R, G, B) = RTE_sat(s);  P = rgb2yuv_int(R, G, B)
 *   blend:  a = s_A * (opacity * RN(1/255));  r_k = fma(P_k, a, r_k * (1 - a));  store RTE_sat
 * (both inside the border quad; the picture only where tx and uv are inside [0,1]^2).  k = Y at every pixel, U and V at owners. */
INL void px_rgb_to_yuv_int(int SRC, int DST, const job_t *j, int x, int y) {
    geom_t g = geometry(j, x, y);
    int hc = (x % 2) == 0 && (y % 2) == 0;
    if (!g.in_border) return;
    const r2y_t *k = &R2Y[j->csc & 3];
    uint8_t *dy = (uint8_t *)texel(&j->t[0], x, y);
    uint8_t *du = 0, *dv = 0;
    if (hc && inside(&j->t[1], x / 2, y / 2)) {
        if (DST == DST_NV12) { du = (uint8_t *)texel(&j->t[1], x / 2, y / 2); dv = du + 1; }
        else { du = (uint8_t *)texel(&j->t[1], x / 2, y / 2); dv = (uint8_t *)texel(&j->t[2], x / 2, y / 2); }
    }
    float af = j->u->opacity * j->u->fillColor[3];
    float iaf = 1.f - af;
    uint8_t fy, fu, fv;
    rgb2yuv_int(k, st8_code(j->u->fillColor[0] * 255.0f), st8_code(j->u->fillColor[1] * 255.0f), st8_code(j->u->fillColor[2] * 255.0f), &fy, &fu, &fv);
    float r0 = clampf(fmaf((float)fy, af, (float)dy[0] * iaf), 0.f, 255.f);
    float r1 = du ? clampf(fmaf((float)fu, af, (float)du[0] * iaf), 0.f, 255.f) : 0.f;
    float r2 = dv ? clampf(fmaf((float)fv, af, (float)dv[0] * iaf), 0.f, 255.f) : 0.f;
    if (g.in_tx && g.in_uv) {
        lin2 l = lin_setup(&j->in[0], g.uv.x, g.uv.y);
        float q0 = cs_fetch(&j->in[0], &l, 0), q1 = cs_fetch(&j->in[0], &l, 1);
        float q2 = cs_fetch(&j->in[0], &l, 2), q3 = cs_fetch(&j->in[0], &l, 3);
        uint8_t R = st8_code(SRC == SRC_BGRA ? q2 : q0), G = st8_code(q1), B = st8_code(SRC == SRC_BGRA ? q0 : q2);
        uint8_t py, pu, pv;
        rgb2yuv_int(k, R, G, B, &py, &pu, &pv);
        float a = q3 * (j->u->opacity * ORC_INV255), ia = 1.f - a;
        r0 = fmaf((float)py, a, r0 * ia);
        r1 = fmaf((float)pu, a, r1 * ia);
        r2 = fmaf((float)pv, a, r2 * ia);
    }
    dy[0] = st8_code(r0);
    if (du) du[0] = st8_code(r1);
    if (dv) dv[0] = st8_code(r2);
}

/* ENVELOPE EVALUATOR (tests only; ids 64..67, never dispatched by the product): the BGRA-target family evaluated
 * the way the reference's own kernels evaluate theirs — on the UNIT scale, texels through ld8 (c / 255), the Khronos
 * filter with sequential roundings (lin_fetch), unfused blends in the source order of px_yuv_to_yuv / px_rgb_to_yuv
 * (cur * (1 - a) + p * a), st8 (* 255, RTE) at the store.  Same geometry, same integer colour matrix on the quantised
 * YUV sample, same structure (fill under the picture, picture over it).  tests/test_oracle_golden.py measures how far
 * the code-scale specification above is from this evaluation, per layer and through stacks of layers. */
INL void px_to_bgra_unit(int SRC, const job_t *j, int x, int y) {
    geom_t g = geometry(j, x, y);
    if (!g.in_border) return;
    uint8_t *d = (uint8_t *)texel(&j->t[0], x, y);
    float af = j->u->opacity * j->u->fillColor[3];
    float r0 = clampf(ld8(d[0]) * (1.f - af) + j->u->fillColor[2] * af, 0.f, 1.f);   /* B */
    float r1 = clampf(ld8(d[1]) * (1.f - af) + j->u->fillColor[1] * af, 0.f, 1.f);   /* G */
    float r2 = clampf(ld8(d[2]) * (1.f - af) + j->u->fillColor[0] * af, 0.f, 1.f);   /* R */
    if (g.in_tx && g.in_uv) {
        float p0, p1, p2, a;
        if (SRC == SRC_BGRA || SRC == SRC_RGBA) {
            lin2 l = lin_setup(&j->in[0], g.uv.x, g.uv.y);
            float q0 = lin_fetch(&j->in[0], &l, 0), q1 = lin_fetch(&j->in[0], &l, 1);
            float q2 = lin_fetch(&j->in[0], &l, 2), q3 = lin_fetch(&j->in[0], &l, 3);
            p0 = SRC == SRC_BGRA ? q0 : q2; p1 = q1; p2 = SRC == SRC_BGRA ? q2 : q0;
            a = q3 * j->u->opacity;
        } else {
            lin2 ly = lin_setup(&j->in[0], g.uv.x, g.uv.y);
            lin2 lc = lin_setup(&j->in[1], g.uv.x, g.uv.y);
            float fy = lin_fetch(&j->in[0], &ly, 0), fu, fv;
            if (SRC == SRC_NV12) { fu = lin_fetch(&j->in[1], &lc, 0); fv = lin_fetch(&j->in[1], &lc, 1); }
            else {
                fu = lin_fetch(&j->in[1], &lc, 0);
                lin2 lv = lin_setup(&j->in[2], g.uv.x, g.uv.y);
                fv = lin_fetch(&j->in[2], &lv, 0);
            }
            uint8_t R, G, B;
            yuv2rgb_int(&CSC[j->csc & 3], st8(fy), st8(fu), st8(fv), &R, &G, &B);
            p0 = ld8(B); p1 = ld8(G); p2 = ld8(R);
            a = 1.0f * j->u->opacity;
        }
        r0 = r0 * (1.f - a) + p0 * a;
        r1 = r1 * (1.f - a) + p1 * a;
        r2 = r2 * (1.f - a) + p2 * a;
    }
    d[0] = st8(r0); d[1] = st8(r1); d[2] = st8(r2); d[3] = 255;
}

/* Clear kernels: img_clear_nv12 (kernels.cl.swift:38-46), img_clear_y420p
 * (:174-185), img_clear_bgra (:257-265).  Every work-item writes chroma at
 * gid/2 (idempotent). */
INL void px_clear(int DST, const job_t *j, int x, int y) {
    if (DST == DST_BGRA) {
        if (inside(&j->t[0], x, y)) {
            uint8_t *d = (uint8_t *)texel(&j->t[0], x, y);
            d[0] = st8(0.f); d[1] = st8(0.f); d[2] = st8(0.f); d[3] = st8(1.f);
        }
        return;
    }
    wr1(&j->t[0], x, y, 0, 0.0f);
    write_chroma(DST, j, x / 2, y / 2, 0.5f, 0.5f);
}

/* ------------------------------------------------------------------------ */
/* Row-range drivers (one work-item per target pixel, global = [W, H])        */
/* ------------------------------------------------------------------------ */
#define ROWS(body) for (int y = y0; y < y1; y++) for (int x = 0; x < j->W; x++) { body; }

static void run_rows(const job_t *j, int y0, int y1) {
    switch (j->kernel) {
    case ORC_IMG_NV12_NV12:   ROWS(px_yuv_to_yuv(SRC_NV12, DST_NV12, j, x, y)) break;
    case ORC_IMG_Y420P_NV12:  ROWS(px_yuv_to_yuv(SRC_Y420P, DST_NV12, j, x, y)) break;
    case ORC_IMG_Y420P_Y420P: ROWS(px_yuv_to_yuv(SRC_Y420P, DST_Y420P, j, x, y)) break;
    case ORC_IMG_BGRA_NV12:   ROWS(px_rgb_to_yuv(SRC_BGRA, DST_NV12, j, x, y)) break;
    case ORC_IMG_RGBA_NV12:   ROWS(px_rgb_to_yuv(SRC_RGBA, DST_NV12, j, x, y)) break;
    case ORC_IMG_BGRA_Y420P:  ROWS(px_rgb_to_yuv(SRC_BGRA, DST_Y420P, j, x, y)) break;
    case ORC_IMG_RGBA_Y420P:  ROWS(px_rgb_to_yuv(SRC_RGBA, DST_Y420P, j, x, y)) break;
    case ORC_IMG_BGRA_BGRA:   ROWS(px_bgra_bgra_metal(j, x, y)) break;
    case ORC_IMG_BGRA_NV12_INT:  ROWS(px_rgb_to_yuv_int(SRC_BGRA, DST_NV12, j, x, y)) break;
    case ORC_IMG_RGBA_NV12_INT:  ROWS(px_rgb_to_yuv_int(SRC_RGBA, DST_NV12, j, x, y)) break;
    case ORC_IMG_BGRA_Y420P_INT: ROWS(px_rgb_to_yuv_int(SRC_BGRA, DST_Y420P, j, x, y)) break;
    case ORC_IMG_RGBA_Y420P_INT: ROWS(px_rgb_to_yuv_int(SRC_RGBA, DST_Y420P, j, x, y)) break;
    case ORC_IMG_NV12_BGRA:   ROWS(px_to_bgra(SRC_NV12, j, x, y)) break;
    case ORC_IMG_Y420P_BGRA:  ROWS(px_to_bgra(SRC_Y420P, j, x, y)) break;
    case ORC_IMG_BGRA_BGRA_TX: ROWS(px_to_bgra(SRC_BGRA, j, x, y)) break;
    case ORC_IMG_RGBA_BGRA_TX: ROWS(px_to_bgra(SRC_RGBA, j, x, y)) break;
    case ORC_ENV_NV12_BGRA_UNIT:   ROWS(px_to_bgra_unit(SRC_NV12, j, x, y)) break;
    case ORC_ENV_Y420P_BGRA_UNIT:  ROWS(px_to_bgra_unit(SRC_Y420P, j, x, y)) break;
    case ORC_ENV_BGRA_BGRA_UNIT:   ROWS(px_to_bgra_unit(SRC_BGRA, j, x, y)) break;
    case ORC_ENV_RGBA_BGRA_UNIT:   ROWS(px_to_bgra_unit(SRC_RGBA, j, x, y)) break;
    case ORC_IMG_CLEAR_NV12:  ROWS(px_clear(DST_NV12, j, x, y)) break;
    case ORC_IMG_CLEAR_Y420P: ROWS(px_clear(DST_Y420P, j, x, y)) break;
    case ORC_IMG_CLEAR_BGRA:
    case ORC_IMG_CLEAR_RGBA:  ROWS(px_clear(DST_BGRA, j, x, y)) break;
    default: break;
    }
}

typedef struct { const job_t *j; int y0, y1; } slice_t;
static void *slice_main(void *p) { slice_t *s = (slice_t *)p; run_rows(s->j, s->y0, s->y1); return 0; }

/* Rows are split on even boundaries so a 2x2 chroma quad has one owner thread. */
static void run_threads(const job_t *j, int threads, void *(*fn)(void *), int H) {
    if (threads < 1) threads = 1;
    if (threads > 256) threads = 256;
    int pairs = (H + 1) / 2;
    if (threads > pairs) threads = pairs > 0 ? pairs : 1;
    slice_t sl[256];
    pthread_t th[256];
    for (int t = 0; t < threads; t++) {
        sl[t].j = j;
        sl[t].y0 = (int)((long)pairs * t / threads) * 2;
        sl[t].y1 = (int)((long)pairs * (t + 1) / threads) * 2;
        if (sl[t].y1 > H) sl[t].y1 = H;
    }
    if (threads == 1) { fn(&sl[0]); return; }
    for (int t = 0; t < threads; t++) pthread_create(&th[t], 0, fn, &sl[t]);
    for (int t = 0; t < threads; t++) pthread_join(th[t], 0);
}

static int plane_ok(const orc_plane *p, int comps) {
    return p && p->data && p->w > 0 && p->h > 0 && p->comps == comps && p->pitch >= p->w * comps;
}

/* plane signature checks mirror the argument lists of the kernels
 * (bind order compute.cl.swift:288-327) */
static int check_planes(const orc_plane *p, int n, int fmt) {
    switch (fmt) {
    case SRC_NV12:  return n == 2 && plane_ok(&p[0], 1) && plane_ok(&p[1], 2);
    case SRC_Y420P: return n == 3 && plane_ok(&p[0], 1) && plane_ok(&p[1], 1) && plane_ok(&p[2], 1);
    default:        return n == 1 && plane_ok(&p[0], 4);
    }
}

int orc_run_kernel(int kernel, const orc_plane *target, int n_target,
                   const orc_plane *inputs, int n_inputs,
                   const orc_uniforms *uniforms, int csc, int threads) {
    int sfmt = -1, dfmt = -1, clear = 0;
    switch (kernel) {
    case ORC_IMG_NV12_NV12: sfmt = SRC_NV12; dfmt = SRC_NV12; break;
    case ORC_IMG_Y420P_NV12: sfmt = SRC_Y420P; dfmt = SRC_NV12; break;
    case ORC_IMG_Y420P_Y420P: sfmt = SRC_Y420P; dfmt = SRC_Y420P; break;
    case ORC_IMG_BGRA_NV12: case ORC_IMG_RGBA_NV12: case ORC_IMG_BGRA_NV12_INT: case ORC_IMG_RGBA_NV12_INT: sfmt = SRC_BGRA; dfmt = SRC_NV12; break;
    case ORC_IMG_BGRA_Y420P: case ORC_IMG_RGBA_Y420P: case ORC_IMG_BGRA_Y420P_INT: case ORC_IMG_RGBA_Y420P_INT: sfmt = SRC_BGRA; dfmt = SRC_Y420P; break;
    case ORC_IMG_BGRA_BGRA: case ORC_IMG_BGRA_BGRA_TX: case ORC_IMG_RGBA_BGRA_TX:
        sfmt = SRC_BGRA; dfmt = SRC_BGRA; break;
    case ORC_IMG_NV12_BGRA: case ORC_ENV_NV12_BGRA_UNIT: sfmt = SRC_NV12; dfmt = SRC_BGRA; break;
    case ORC_IMG_Y420P_BGRA: case ORC_ENV_Y420P_BGRA_UNIT: sfmt = SRC_Y420P; dfmt = SRC_BGRA; break;
    case ORC_ENV_BGRA_BGRA_UNIT: case ORC_ENV_RGBA_BGRA_UNIT: sfmt = SRC_BGRA; dfmt = SRC_BGRA; break;
    case ORC_IMG_CLEAR_NV12: clear = 1; dfmt = SRC_NV12; break;
    case ORC_IMG_CLEAR_Y420P: clear = 1; dfmt = SRC_Y420P; break;
    case ORC_IMG_CLEAR_BGRA: case ORC_IMG_CLEAR_RGBA: clear = 1; dfmt = SRC_BGRA; break;
    case ORC_IMG_CLEAR_YUVS: case ORC_SND_S16I_S16I: case ORC_ME_FULLSEARCH:
        return ORC_ERR_NOT_IMPLEMENTED;
    default: return ORC_ERR_INVALID_VALUE;
    }
    if (!target || !check_planes(target, n_target, dfmt)) return ORC_ERR_BAD_TARGET;
    if (!clear) {
        if (!inputs || !check_planes(inputs, n_inputs, sfmt)) return ORC_ERR_BAD_INPUT;
        if (!uniforms) return ORC_ERR_INVALID_VALUE;
    }
    job_t j = { kernel, csc, target, inputs, uniforms, target[0].w, target[0].h };
    run_threads(&j, threads, slice_main, j.H);
    return ORC_OK;
}

/* ------------------------------------------------------------------------ */
/* Lanczos-3 (spec owned by this repo)                 */
/* ------------------------------------------------------------------------ */
static double sinc_pi(double t) {
    if (t == 0.0) return 1.0;
    double pt = 3.14159265358979323846 * t;
    return sin(pt) / pt;
}

int orc_lanczos_table(int in_size, int out_size, int *taps_out,
                      int32_t *first, float *weights, int max_taps) {
    if (in_size <= 0 || out_size <= 0) return ORC_ERR_INVALID_VALUE;
    double scale = (double)in_size / (double)out_size;
    double fs = scale > 1.0 ? scale : 1.0;
    double support = 3.0 * fs;
    int taps = 2 * (int)ceil(support);
    if (taps > max_taps) return ORC_ERR_INVALID_VALUE;
    *taps_out = taps;
    for (int o = 0; o < out_size; o++) {
        double center = ((double)o + 0.5) * scale - 0.5;
        int f = (int)floor(center - support) + 1;
        double w[256], sum = 0.0;
        for (int k = 0; k < taps; k++) {
            double t = ((double)(f + k) - center) / fs;
            double v = (t > -3.0 && t < 3.0) ? sinc_pi(t) * sinc_pi(t / 3.0) : 0.0;
            w[k] = v; sum += v;
        }
        first[o] = f;
        for (int k = 0; k < taps; k++) weights[(size_t)o * taps + k] = (float)(w[k] / sum);
    }
    return ORC_OK;
}

typedef struct {
    const orc_plane *dst, *src;
    const int32_t *fx, *fy; const float *wx, *wy; int tx, ty;
    float *tmp; /* [src->h][dst->w][4] horizontal pass result, code scale */
} lz_job;
typedef struct { const lz_job *j; int y0, y1; } lz_slice;

static void *lz_hpass(void *p) {
    lz_slice *s = (lz_slice *)p; const lz_job *j = s->j;
    int ow = j->dst->w, iw = j->src->w;
    for (int y = s->y0; y < s->y1; y++) {
        for (int o = 0; o < ow; o++) {
            float acc[4] = { 0.f, 0.f, 0.f, 0.f };
            for (int k = 0; k < j->tx; k++) {
                int i = clampi(j->fx[o] + k, 0, iw - 1);
                const uint8_t *t = texel(j->src, i, y);
                float w = j->wx[(size_t)o * j->tx + k];
                for (int c = 0; c < 4; c++) acc[c] = fmaf(w, (float)t[c], acc[c]);
            }
            memcpy(j->tmp + ((size_t)y * ow + o) * 4, acc, sizeof acc);
        }
    }
    return 0;
}
static void *lz_vpass(void *p) {
    lz_slice *s = (lz_slice *)p; const lz_job *j = s->j;
    int ow = j->dst->w, ih = j->src->h;
    for (int o = s->y0; o < s->y1; o++) {
        for (int x = 0; x < ow; x++) {
            float acc[4] = { 0.f, 0.f, 0.f, 0.f };
            for (int k = 0; k < j->ty; k++) {
                int i = clampi(j->fy[o] + k, 0, ih - 1);
                const float *t = j->tmp + ((size_t)i * ow + x) * 4;
                float w = j->wy[(size_t)o * j->ty + k];
                for (int c = 0; c < 4; c++) acc[c] = fmaf(w, t[c], acc[c]);
            }
            uint8_t *d = (uint8_t *)texel(j->dst, x, o);
            for (int c = 0; c < 4; c++) d[c] = st8_code(acc[c]);
        }
    }
    return 0;
}
static void lz_threads(const lz_job *j, int threads, void *(*fn)(void *), int n) {
    if (threads < 1) threads = 1;
    if (threads > 256) threads = 256;
    if (threads > n) threads = n;
    lz_slice sl[256]; pthread_t th[256];
    for (int t = 0; t < threads; t++) {
        sl[t].j = j; sl[t].y0 = (int)((long)n * t / threads); sl[t].y1 = (int)((long)n * (t + 1) / threads);
    }
    if (threads == 1) { fn(&sl[0]); return; }
    for (int t = 0; t < threads; t++) pthread_create(&th[t], 0, fn, &sl[t]);
    for (int t = 0; t < threads; t++) pthread_join(th[t], 0);
}

int orc_lanczos_bgra(const orc_plane *dst, const orc_plane *src, int threads) {
    if (!plane_ok(dst, 4)) return ORC_ERR_BAD_TARGET;
    if (!plane_ok(src, 4)) return ORC_ERR_BAD_INPUT;
    enum { MAXT = 256 };
    lz_job j; memset(&j, 0, sizeof j);
    j.dst = dst; j.src = src;
    int32_t *fx = malloc(sizeof(int32_t) * dst->w), *fy = malloc(sizeof(int32_t) * dst->h);
    float *wx = malloc(sizeof(float) * (size_t)dst->w * MAXT), *wy = malloc(sizeof(float) * (size_t)dst->h * MAXT);
    float *tmp = malloc(sizeof(float) * 4 * (size_t)dst->w * src->h);
    int rc = ORC_ERR_INVALID_VALUE;
    if (fx && fy && wx && wy && tmp &&
        orc_lanczos_table(src->w, dst->w, &j.tx, fx, wx, MAXT) == ORC_OK &&
        orc_lanczos_table(src->h, dst->h, &j.ty, fy, wy, MAXT) == ORC_OK) {
        j.fx = fx; j.fy = fy; j.wx = wx; j.wy = wy; j.tmp = tmp;
        lz_threads(&j, threads, lz_hpass, src->h);
        lz_threads(&j, threads, lz_vpass, dst->h);
        rc = ORC_OK;
    }
    free(fx); free(fy); free(wx); free(wy); free(tmp);
    return rc;
}

/* ------------------------------------------------------------------------ */
/* The two idle kernels of `enum ComputeKernel` (compute.swift:67,70; no caller in the reference dispatches them)   */
/* ------------------------------------------------------------------------ */

/* snd_s16i_s16i, kernels.cl.swift:534-562.  One work-item per interleaved-stereo sample:
 *   channel = gid % 2
 *   for i < inputCount:  value = min((float)in_i[gid] * gain_i * (channel == 0 ? 1 - fade_i : fade_i), 32767.f);  out[gid] += (short)value
 * Products left to right, no contraction.  `min` only caps the top (:557), so a value below -32768 reaches the conversion: OpenCL C leaves
 * float -> short out of range undefined; fixed here as what the compiled kernel string does on x86-64 (and the HIP kernel on gfx950): convert
 * to int32 toward zero (out of int32 range and NaN: INT32_MIN) and keep the low 16 bits.  The += wraps modulo 2^16.  inputOffsets is not read
 * by the kernel (:548-560). */
INL int16_t snd_cvt(float v) {
    int32_t i = (v >= -2147483648.0f && v < 2147483648.0f) ? (int32_t)v : INT32_MIN;   /* cvttss2si; NaN fails both tests */
    return (int16_t)(uint16_t)(uint32_t)i;
}
int orc_snd_s16i_s16i(int16_t *out, int n, const int16_t *const *in, const orc_snd_uniforms *u) {
    if (!out || !u || n < 0 || u->inputCount < 0 || u->inputCount > 8) return ORC_ERR_INVALID_VALUE;
    for (int i = 0; i < u->inputCount; i++) if (!in || !in[i]) return ORC_ERR_BAD_INPUT;
    for (int gid = 0; gid < n; gid++) {
        int channel = gid % 2;
        for (int i = 0; i < u->inputCount; i++) {
            float k = channel == 0 ? 1.f - u->inputFade[i] : u->inputFade[i];
            float x = (float)in[i][gid] * u->inputGains[i] * k;
            float value = 32767.f < x ? 32767.f : x;                     /* min(x, y) = y < x ? y : x, OpenCL 1.2 6.12.4 */
            out[gid] = (int16_t)(uint16_t)((uint16_t)out[gid] + (uint16_t)snd_cvt(value));
        }
    }
    return ORC_OK;
}

/* me_fullsearch, kernels.metal:129-267 (Metal only; restated by hand).  One thread per block of the CURRENT picture: every candidate
 * position of the block inside the search area of the REFERENCE picture is scored as deltaCost2(mv) + SAD * 256 (:236-237), candidates
 * visited column by column, top to bottom (:229-256), the first strict minimum kept (:241); the clamped vector is normalised to
 * [0, 1] and written as (mv.x, 0.5, mv.y, 1) (:262-264).
 *
 * Faithful to the source INCLUDING its sliding-window SAD (:152-166): when the previous candidate's top-row SAD is > 0 the function
 * sums the new top row, then runs a second loop for the bottom row whose counters were left at the end by the first — it adds nothing —
 * and returns previousSad - previousSide.  A clean full search is NOT what the reference computes: that running value falls below zero in
 * tall search windows, and the first negative score ends the search through earlyExit (:221,228-232,247-249).
 * Texel reads are R8Unorm -> c / 255.0f (ld8); reads outside a picture (an origin block hanging over the right / bottom edge) return 0
 * (Metal leaves them undefined).  Sums accumulate in float in source order: row-major over the block (:168-183).
 * deltaCost2 (:135-142) goes through log2: the table of per-component costs is built by the HOST's log2f for both the oracle and the
 * product (the HIP kernel receives the same table), so that no device libm enters the comparison. */
static float me_component_cost(int d) {           /* lambda * (log2(|v| + 1) * 2 + 0.718 + (v != 0)) + 0.5, :136-141 */
    const float lambda = 4.0f;
    float l2 = log2f((float)d + 1.0f);
    float rounding = d != 0 ? 1.0f : 0.0f;
    return lambda * (l2 * 2.0f + 0.718f + rounding) + 0.5f;
}
void orc_me_cost_table(float *table, int n) { for (int d = 0; d < n; d++) table[d] = me_component_cost(d); }
INL float me_rd(const orc_plane *p, int x, int y) { return (x >= 0 && y >= 0 && x < p->w && y < p->h) ? ld8(texel(p, x, y)[0]) : 0.0f; }
typedef struct { float side, sad; } me_sad_t;
static me_sad_t me_sad(const int b1[4], const int b2[4], const orc_plane *t1, const orc_plane *t2, float prevSide, float prevSad) {
    int p1x = b1[0], p1y = b1[1], p2x = b2[0], p2y = b2[1];
    float sum = 0.f, top = 0.f;
    if (prevSide > 0.f) {
        float bottom = 0.f;
        for (; p1x < b1[2] && p2x < b2[2]; p1x++, p2x++) top += fabsf(me_rd(t1, p1x, b1[1]) - me_rd(t2, p2x, b2[1]));
        for (; p1x < b1[2] && p2x < b2[2]; p1x++, p2x++) bottom += fabsf(me_rd(t1, p1x, b1[3] - 1) - me_rd(t2, p2x, b2[3] - 1));   /* (never entered) */
        sum = prevSad - prevSide + bottom;
    } else {
        while (p1y < b1[3] && p2y < b2[3]) {
            p1x = b1[0]; p2x = b2[0];
            while (p1x < b1[2] && p2x < b2[2]) {
                float d = fabsf(me_rd(t1, p1x, p1y) - me_rd(t2, p2x, p2y));
                sum += d;
                if (p1y == b1[1] && p2y == b2[1]) top += d;
                p1x++; p2x++;
            }
            p1y++; p2y++;
        }
    }
    me_sad_t r = { top, sum };
    return r;
}
INL int me_clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }     /* Metal clamp(x, lo, hi) = min(max(x, lo), hi) */
INL float me_clampf(float v, float lo, float hi) { return fminf(fmaxf(v, lo), hi); }
int orc_me_fullsearch(const orc_plane *out, const orc_plane *ref, const orc_plane *cur, const orc_me_uniforms *u) {
    if (!plane_ok(out, 4)) return ORC_ERR_BAD_TARGET;
    if (!plane_ok(ref, 1) || !plane_ok(cur, 1) || !u) return ORC_ERR_BAD_INPUT;
    const int bsx = u->blockSize[0], bsy = u->blockSize[1];
    if (bsx < 1 || bsy < 1 || bsx > 64 || bsy > 64 || u->searchWindowSize[0] < 0 || u->searchWindowSize[1] < 0) return ORC_ERR_INVALID_VALUE;
    float cost[256];
    orc_me_cost_table(cost, 256);
    const int swx = u->searchWindowSize[0] < 64 ? u->searchWindowSize[0] : 64, swy = u->searchWindowSize[1] < 64 ? u->searchWindowSize[1] : 64;   /* MAX_SEARCH_SIZE */
    const float maxx = (float)(u->searchWindowSize[0] / 2), maxy = (float)(u->searchWindowSize[1] / 2);
    for (int by = 0; by < out->h; by++) for (int bx = 0; bx < out->w; bx++) {
        const int ob[4] = { bx * bsx, by * bsy, bx * bsx + bsx, by * bsy + bsy };
        const int left = me_clampi(ob[0] + bsx / 2 - swx / 2, 0, u->imageSize[0]), top = me_clampi(ob[1] + bsy / 2 - swy / 2, 0, u->imageSize[1]);
        const int right = me_clampi(left + swx, 0, u->imageSize[0]), bottom = me_clampi(top + swy, 0, u->imageSize[1]);
        int rb[4] = { left, top, left + bsx, top + bsy };
        float bestScore = 3.402823466e+38f, bmx = 0.f, bmy = 0.f, side = 0.f, prev = 0.f;
        const float threshold = 0.f;           /* :221 */
        int earlyExit = 0;                     /* :228 */
        while (rb[2] < right && !earlyExit) {
            rb[1] = top; rb[3] = rb[1] + bsy;
            while (rb[3] < bottom && !earlyExit) {
                me_sad_t s = me_sad(ob, rb, cur, ref, side, prev);
                const int mx = ob[0] - rb[0], my = ob[1] - rb[1];
                const int ax = mx < 0 ? -mx : mx, ay = my < 0 ? -my : my;
                const float score = 4.0f * (cost[ax > 255 ? 255 : ax] + cost[ay > 255 ? 255 : ay]) + s.sad * 256.0f;
                prev = s.sad; side = s.side;
                if (score < bestScore) { bestScore = score; bmx = me_clampf((float)mx, -maxx, maxx); bmy = me_clampf((float)my, -maxy, maxy); }
                /* :247-249.  A score DOES go negative: the sliding form above returns previousSad - previousSide row after row, so down one
                 * column the "SAD" loses one top-row SAD per candidate and falls below zero once the column is a little taller than the
                 * block (every window >= about three block heights).  The first such candidate in visiting order (columns left to right,
                 * rows top to bottom) has just become the best one — every earlier score was >= 0 — and ends the whole search. */
                if (score < threshold) earlyExit = 1;
                rb[1]++; rb[3]++;
            }
            side = 0.f; prev = 0.f; rb[0]++; rb[2]++;
        }
        bmx = bmx / maxx; bmy = bmy / maxy;
        bmx = bmx * 0.5f + 0.5f; bmy = bmy * 0.5f + 0.5f;
        uint8_t *d = (uint8_t *)texel(out, bx, by);
        d[0] = st8(bmx); d[1] = st8(0.5f); d[2] = st8(bmy); d[3] = st8(1.0f);
    }
    return ORC_OK;
}

/*
 * oracle/clref/cl_shim.c — host-side OpenCL-1.2 image builtins + NDRange driver
 * for the reference's OpenCL-C kernel strings compiled to x86-64 objects.
 *
 * TEST INFRASTRUCTURE, NOT PRODUCT.  This is a CROSS-CHECK, NOT A REFERENCE
 * BUILD: the kernel *bodies* linked against this file are the reference's own
 * (extracted at build time from /root/reference/Sources/SwiftVideo/
 * kernels.cl.swift, never committed), but the eleven OpenCL builtins they call
 * are supplied HERE, written from the Khronos OpenCL 1.2 specification
 * (sections 6.12.14, 8.2, 8.3.1.1), because the image ships no OpenCL CPU
 * runtime.  It therefore pins only the kernel-body arithmetic of
 * oracle/ref_kernels.c (operation order, constants, branch structure), not the
 * sampler; parity stays "unpinned" (see ref_kernels.h).
 *
 * Compile with the same clang that compiled the kernels so that the
 * ext_vector_type ABI matches.
 */
#include <math.h>
#include <stdint.h>
#include <string.h>

typedef float float4 __attribute__((ext_vector_type(4)));
typedef float float2 __attribute__((ext_vector_type(2)));
typedef int int2 __attribute__((ext_vector_type(2)));

/* pointee chosen by the shim for the opaque image2d_t */
typedef struct {
    uint8_t *data;
    int32_t w, h, pitch, comps;
} shim_image;

/* sampler bits as clang's opencl-c-base.h encodes them */
#define S_NORMALIZED 1
#define S_ADDR_MASK 0xE
#define S_CLAMP_TO_EDGE 2
#define S_LINEAR 0x20

static __thread unsigned g_gid[2];
static unsigned g_gsize[2];

unsigned long get_global_id_(unsigned d) __asm__("_Z13get_global_idj");
unsigned long get_global_id_(unsigned d) { return d < 2 ? g_gid[d] : 0; }
unsigned long get_global_size_(unsigned d) __asm__("_Z15get_global_sizej");
unsigned long get_global_size_(unsigned d) { return d < 2 ? g_gsize[d] : 1; }

float dot_(float4 a, float4 b) __asm__("_Z3dotDv4_fS_");
float dot_(float4 a, float4 b) {
#pragma clang fp contract(off)
    return ((a.x * b.x + a.y * b.y) + a.z * b.z) + a.w * b.w; /* kernels.cuda.swift:45-47 */
}
float clamp_f(float v, float lo, float hi) __asm__("_Z5clampfff");
float clamp_f(float v, float lo, float hi) { return fminf(fmaxf(v, lo), hi); }
float4 clamp_v(float4 v, float lo, float hi) __asm__("_Z5clampDv4_fff");
float4 clamp_v(float4 v, float lo, float hi) {
    float4 r;
    r.x = clamp_f(v.x, lo, hi); r.y = clamp_f(v.y, lo, hi);
    r.z = clamp_f(v.z, lo, hi); r.w = clamp_f(v.w, lo, hi);
    return r;
}
float min_f(float a, float b) __asm__("_Z3minff");
float min_f(float a, float b) { return fminf(a, b); }

void *__translate_sampler_initializer(int v) { return (void *)(intptr_t)(v | 0x10000); }

/* channel order fill, OpenCL 1.2 section 5.3.1.1: CL_R -> (r,0,0,1),
 * CL_RG -> (r,g,0,1), CL_RGBA -> (r,g,b,a); UNORM_INT8 -> c/255.0f (8.3.1.1) */
static float4 fetch(const shim_image *im, int x, int y) {
    float4 r = { 0.f, 0.f, 0.f, 1.f };
    if (x < 0 || y < 0 || x >= im->w || y >= im->h) {
        /* undefined for ADDRESS_NONE; defined as zero data (see ref_kernels.c) */
        r.w = im->comps == 4 ? 0.f : 1.f;
        return r;
    }
    const uint8_t *p = im->data + (size_t)y * im->pitch + (size_t)x * im->comps;
    r.x = (float)p[0] / 255.0f;
    if (im->comps >= 2) r.y = (float)p[1] / 255.0f;
    if (im->comps == 4) { r.z = (float)p[2] / 255.0f; r.w = (float)p[3] / 255.0f; }
    return r;
}

static int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

float4 read_imagef_i(const shim_image *im, void *smp, int2 c)
    __asm__("_Z11read_imagef14ocl_image2d_ro11ocl_samplerDv2_i");
float4 read_imagef_i(const shim_image *im, void *smp, int2 c) {
    (void)smp; /* unnormalized, nearest */
    return fetch(im, c.x, c.y);
}

float4 read_imagef_f(const shim_image *im, void *smp, float2 c)
    __asm__("_Z11read_imagef14ocl_image2d_ro11ocl_samplerDv2_f");
float4 read_imagef_f(const shim_image *im, void *smp, float2 c) {
#pragma clang fp contract(off)
    int bits = (int)(intptr_t)smp;
    float u = c.x, v = c.y;
    if (bits & S_NORMALIZED) { u = u * (float)im->w; v = v * (float)im->h; }
    if (!(bits & S_LINEAR)) {
        return fetch(im, clampi((int)floorf(u), 0, im->w - 1), clampi((int)floorf(v), 0, im->h - 1));
    }
    /* section 8.2, linear filter */
    float um = u - 0.5f, vm = v - 0.5f;
    float fu = floorf(um), fv = floorf(vm);
    float a = um - fu, b = vm - fv;
    int i0 = clampi((int)fu, 0, im->w - 1), i1 = clampi((int)fu + 1, 0, im->w - 1);
    int j0 = clampi((int)fv, 0, im->h - 1), j1 = clampi((int)fv + 1, 0, im->h - 1);
    float4 t00 = fetch(im, i0, j0), t10 = fetch(im, i1, j0);
    float4 t01 = fetch(im, i0, j1), t11 = fetch(im, i1, j1);
    float w00 = (1.0f - a) * (1.0f - b), w10 = a * (1.0f - b);
    float w01 = (1.0f - a) * b, w11 = a * b;
    float4 r;
    r.x = ((w00 * t00.x + w10 * t10.x) + w01 * t01.x) + w11 * t11.x;
    r.y = ((w00 * t00.y + w10 * t10.y) + w01 * t01.y) + w11 * t11.y;
    r.z = ((w00 * t00.z + w10 * t10.z) + w01 * t01.z) + w11 * t11.z;
    r.w = ((w00 * t00.w + w10 * t10.w) + w01 * t01.w) + w11 * t11.w;
    return r;
}

static uint8_t cvt(float f) { /* convert_uchar_sat_rte(f * 255.0f), NaN -> 0 */
#pragma clang fp contract(off)
    float v = f * 255.0f;
    if (v != v) return 0;
    v = rintf(v);
    if (v <= 0.f) return 0;
    if (v >= 255.f) return 255;
    return (uint8_t)(int)v;
}
static void store(const shim_image *im, int2 c, float4 v) {
    if (c.x < 0 || c.y < 0 || c.x >= im->w || c.y >= im->h) return;
    uint8_t *p = im->data + (size_t)c.y * im->pitch + (size_t)c.x * im->comps;
    p[0] = cvt(v.x);
    if (im->comps >= 2) p[1] = cvt(v.y);
    if (im->comps == 4) { p[2] = cvt(v.z); p[3] = cvt(v.w); }
}
void write_imagef_wo(const shim_image *im, int2 c, float4 v)
    __asm__("_Z12write_imagef14ocl_image2d_woDv2_iDv4_f");
void write_imagef_wo(const shim_image *im, int2 c, float4 v) { store(im, c, v); }
void write_imagef_rw(const shim_image *im, int2 c, float4 v)
    __asm__("_Z12write_imagef14ocl_image2d_rwDv2_iDv4_f");
void write_imagef_rw(const shim_image *im, int2 c, float4 v) { store(im, c, v); }

/* ---- kernels (defined by the compiled reference objects) ---------------- */
typedef const shim_image *IMG;
void img_clear_nv12(IMG, IMG);
void img_clear_y420p(IMG, IMG, IMG);
void img_clear_bgra(IMG);
void img_nv12_nv12(IMG, IMG, IMG, IMG, IMG, IMG, const void *);
void img_y420p_nv12(IMG, IMG, IMG, IMG, IMG, IMG, IMG, const void *);
void img_y420p_y420p(IMG, IMG, IMG, IMG, IMG, IMG, IMG, IMG, IMG, const void *);
void img_bgra_y420p(IMG, IMG, IMG, IMG, IMG, IMG, IMG, const void *);
void img_rgba_y420p(IMG, IMG, IMG, IMG, IMG, IMG, IMG, const void *);
void img_bgra_nv12(IMG, IMG, IMG, IMG, IMG, const void *);
void img_rgba_nv12(IMG, IMG, IMG, IMG, IMG, const void *);

/* NDRange driver: global = [t[0].w, t[0].h] (compute.cl.swift:329-335);
 * argument order [outputs][cur = outputs][inputs][uniforms] (:288-327). */
int clref_run(const char *name, const shim_image *t, int nt, const shim_image *in, int nin,
              const void *uniforms) {
    if (nt < 1) return 4;
    g_gsize[0] = (unsigned)t[0].w; g_gsize[1] = (unsigned)t[0].h;
    int id = -1;
    static const char *names[] = { "img_clear_nv12", "img_clear_y420p", "img_clear_bgra",
        "img_nv12_nv12", "img_y420p_nv12", "img_y420p_y420p", "img_bgra_y420p",
        "img_rgba_y420p", "img_bgra_nv12", "img_rgba_nv12" };
    for (int i = 0; i < 10; i++) if (!strcmp(name, names[i])) id = i;
    if (id < 0) return 1;
    static const int need_t[] = { 2, 3, 1, 2, 2, 3, 3, 3, 2, 2 };
    static const int need_in[] = { 0, 0, 0, 2, 3, 3, 1, 1, 1, 1 };
    if (nt != need_t[id] || nin != need_in[id]) return 5;
    for (unsigned y = 0; y < g_gsize[1]; y++) for (unsigned x = 0; x < g_gsize[0]; x++) {
        g_gid[0] = x; g_gid[1] = y;
        switch (id) {
        case 0: img_clear_nv12(&t[0], &t[1]); break;
        case 1: img_clear_y420p(&t[0], &t[1], &t[2]); break;
        case 2: img_clear_bgra(&t[0]); break;
        case 3: img_nv12_nv12(&t[0], &t[1], &t[0], &t[1], &in[0], &in[1], uniforms); break;
        case 4: img_y420p_nv12(&t[0], &t[1], &t[0], &t[1], &in[0], &in[1], &in[2], uniforms); break;
        case 5: img_y420p_y420p(&t[0], &t[1], &t[2], &t[0], &t[1], &t[2], &in[0], &in[1], &in[2], uniforms); break;
        case 6: img_bgra_y420p(&t[0], &t[1], &t[2], &t[0], &t[1], &t[2], &in[0], uniforms); break;
        case 7: img_rgba_y420p(&t[0], &t[1], &t[2], &t[0], &t[1], &t[2], &in[0], uniforms); break;
        case 8: img_bgra_nv12(&t[0], &t[1], &t[0], &t[1], &in[0], uniforms); break;
        case 9: img_rgba_nv12(&t[0], &t[1], &t[0], &t[1], &in[0], uniforms); break;
        }
    }
    return 0;
}

/* snd_s16i_s16i (kernels.cl.swift:534-562): buffers, a 1-D range of n work-items; the kernel needs get_global_id and min from above only.
 * Unused input pointers may be NULL (the kernel indexes inputs[i] for i < inputCount only). */
void snd_s16i_s16i(short *, const void *, short *, short *, short *, short *, short *, short *, short *, short *);
int clref_run_snd(short *out, int n, short *const *in, const void *uniforms) {
    g_gsize[0] = (unsigned)n; g_gsize[1] = 1;
    for (int gid = 0; gid < n; gid++) {
        g_gid[0] = (unsigned)gid; g_gid[1] = 0;
        snd_s16i_s16i(out, uniforms, in[0], in[1], in[2], in[3], in[4], in[5], in[6], in[7]);
    }
    return 0;
}

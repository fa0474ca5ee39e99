// pixel_math.hip.h — device-side arithmetic of the SwiftVideo picture kernels
// for gfx950.  All float math is IEEE binary32, evaluated in the reference's
// source order with contraction disabled (this file is compiled with
// -ffp-contract=off and the pragma below); the few fused operations are
// written explicitly with __builtin_fmaf where a single rounding is intended.
//
// Semantics (bit-exact contract, checked by tests/ against oracle/):
//   sampler      — Khronos OpenCL 1.2 section 8.2, LINEAR / CLAMP_TO_EDGE /
//                  normalized coords, as bound by kernels.cl.swift:61
//   unorm8 load  — c / 255.0f correctly rounded      (OpenCL 1.2 section 8.3.1.1)
//   unorm8 store — convert_uchar_sat_rte(f * 255.0f), NaN -> 0
//   dot()        — ((x+y)+z)+w, kernels.cuda.swift:45-47
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "device_types.h"

#pragma clang fp contract(off)

namespace chv {

#define CHV_DEV __device__ __forceinline__

// Global-memory accessors.  Plane pointers reach the kernels through descriptor tables, so the compiler sees
// generic pointers and emits FLAT loads/stores — which count in lgkmcnt as well as vmcnt: every wait for an
// LDS read then also waits for the outstanding global prefetch, and the prefetch hides nothing.  Going
// through an address_space(1) pointer makes them global_load / global_store (vmcnt only).
#if defined(__HIP_DEVICE_COMPILE__)
#define CHV_GLOBAL __attribute__((address_space(1)))
#else
#define CHV_GLOBAL
#endif
typedef uint32_t chv_u32x2 __attribute__((ext_vector_type(2)));
typedef uint32_t chv_u32x4 __attribute__((ext_vector_type(4)));
template <typename T> CHV_DEV T gld(const void *p) { return *(const CHV_GLOBAL T *)(uintptr_t)p; }
template <> CHV_DEV uint2 gld<uint2>(const void *p) { chv_u32x2 v = *(const CHV_GLOBAL chv_u32x2 *)(uintptr_t)p; return make_uint2(v.x, v.y); }
template <> CHV_DEV uint4 gld<uint4>(const void *p) { chv_u32x4 v = *(const CHV_GLOBAL chv_u32x4 *)(uintptr_t)p; return make_uint4(v.x, v.y, v.z, v.w); }
template <typename T> CHV_DEV void gst(void *p, T v) { *(CHV_GLOBAL T *)(uintptr_t)p = v; }
// Read-only data at a wave-uniform address that no launch in flight writes (the geometry tables of a batch, geom_cache.h): through the constant
// address space these are scalar loads (s_load_dword*, several merged into one), whatever the compiler can prove about where the address came from.
#if defined(__HIP_DEVICE_COMPILE__)
#define CHV_CONSTANT __attribute__((address_space(4)))
#else
#define CHV_CONSTANT
#endif
template <typename T> CHV_DEV T cld(uint64_t addr) { return *(const CHV_CONSTANT T *)(uintptr_t)addr; }
// wave-uniform base + 32-bit unsigned per-lane byte offset: selects the `global_* v_off, ..., s[base:base+1]` addressing form (the row
// base stays on the scalar unit; the generic pointer + size_t form costs a v_mad_i64_i32 per access)
template <typename T> CHV_DEV T gld_at(const uint8_t *base, uint32_t off) { return *(const CHV_GLOBAL T *)((const CHV_GLOBAL uint8_t *)(uintptr_t)base + off); }
template <typename T> CHV_DEV void gst_at(uint8_t *base, uint32_t off, T v) { *(CHV_GLOBAL T *)((CHV_GLOBAL uint8_t *)(uintptr_t)base + off) = v; }
// streaming stores (nt): for data written once per launch and not read back by it
CHV_DEV void gst_stream(void *p, uint2 v) { chv_u32x2 t = { v.x, v.y }; __builtin_nontemporal_store(t, (CHV_GLOBAL chv_u32x2 *)(uintptr_t)p); }
CHV_DEV void gst_stream(void *p, uint4 v) { chv_u32x4 t = { v.x, v.y, v.z, v.w }; __builtin_nontemporal_store(t, (CHV_GLOBAL chv_u32x4 *)(uintptr_t)p); }
CHV_DEV void gst_stream(void *p, uint32_t v) { __builtin_nontemporal_store(v, (CHV_GLOBAL uint32_t *)(uintptr_t)p); }
// The `old` operand of a DPP move whose every lane reads a valid lane (quad_perm, all rows and banks): never used, and a register that no
// instruction had to initialise (an empty asm "defines" it) — update_dpp(0, ...) costs a v_mov_b32 per move.
CHV_DEV int dpp_old() { int v; asm("" : "=v"(v)); return v; }
// An unconditional (empty) use of prefetched registers.  hipcc places its s_waitcnt for a load in front of the
// first use it sees on a path and merges paths pessimistically: with the prefetch consumed only under conditions
// (staged? lane owns a slot?) the registers stay "maybe pending" on the paths that skip the use, and every load
// of the NEXT prefetch then gets `s_waitcnt vmcnt(0)` in front of it, which serialises the loads.  A use on
// every path, right where the wait belongs anyway, settles it.
// The asm has outputs ("+v") and is NOT volatile on purpose: an `asm volatile` (and any asm without outputs is
// implicitly volatile) counts as a possible write to memory, after which the compiler no longer proves the tick / layer
// descriptors unclobbered and reads every uniform of a layer with per-lane global loads instead of scalar loads
// (tick_yuv_wave: 90 vector loads per wave, profiles/r02_notes.md).
template <int N>
CHV_DEV void touch_regs(uint4 (&regs)[N]) {
#pragma unroll
    for (int n = 0; n < N; n++) asm("" : "+v"(regs[n].x), "+v"(regs[n].y), "+v"(regs[n].z), "+v"(regs[n].w));
}
// the volatile, input-only form (kernels that read no descriptors after it: it constrains the register allocator less)
template <int N>
CHV_DEV void touch_regs_volatile(const uint4 (&regs)[N]) {
#pragma unroll
    for (int n = 0; n < N; n++) asm volatile("" :: "v"(regs[n].x), "v"(regs[n].y), "v"(regs[n].z), "v"(regs[n].w));
}

// c / 255.0f, correctly rounded, without a divide: with r_hi = RN(1/255) and
// r_lo = RN(1/255 - r_hi), RN(c*r_hi + RN(c*r_lo)) equals RN(c/255) for all 256 codes
// (checked in exact rational arithmetic by tests/test_host_logic.py and on device by
// tests/test_gpu_primitives.py).
CHV_DEV float unorm8f(float f) {
    const float r_hi = 0x1.010102p-8f;
    const float r_lo = -0x1.fdfdfep-33f;
    return __builtin_fmaf(f, r_hi, f * r_lo);
}
CHV_DEV float unorm8(uint32_t c) { return unorm8f((float)c); }

// convert_uchar_sat_rte(f * 255.0f); fmaxf drops a NaN operand, so NaN -> 0.
CHV_DEV uint32_t to_code(float f) {
    float v = __builtin_rintf(f * 255.0f);
    v = __builtin_fminf(__builtin_fmaxf(v, 0.0f), 255.0f);
    return (uint32_t)v;
}
// same for a value already on the 0..255 code scale (Lanczos output)
CHV_DEV uint32_t to_code_raw(float v) {
    v = __builtin_rintf(v);
    v = __builtin_fminf(__builtin_fmaxf(v, 0.0f), 255.0f);
    return (uint32_t)v;
}

// Four (three) code-scale values -> one packed word, byte k = to_code_raw(ck): gfx950's v_cvt_pk_u8_f32 clamps to
// [0, 255], rounds to nearest even, turns NaN into 0 and leaves the other bytes of its destination alone
// (tools/probe_cvt_pk_u8.cpp, profiles/r01_probe_cvt_pk_u8.txt) — to_code_raw plus the packing in one instruction per byte.
CHV_DEV uint32_t pack_codes(float c0, float c1, float c2, uint32_t w) {
    asm("v_cvt_pk_u8_f32 %0, %1, 0, %0" : "+v"(w) : "v"(c0));
    asm("v_cvt_pk_u8_f32 %0, %1, 1, %0" : "+v"(w) : "v"(c1));
    asm("v_cvt_pk_u8_f32 %0, %1, 2, %0" : "+v"(w) : "v"(c2));
    return w;
}
CHV_DEV uint32_t pack_codes(float c0, float c1, float c2, float c3) {
    uint32_t w = pack_codes(c0, c1, c2, 0u);
    asm("v_cvt_pk_u8_f32 %0, %1, 3, %0" : "+v"(w) : "v"(c3));
    return w;
}

// to_code for a value known to lie in [0, 1 + a few ulp] and not NaN (a convex
// combination of unorm8 samples): the saturation cannot trigger, so it is dropped.
CHV_DEV uint32_t to_code_unit(float f) {
    return (uint32_t)__builtin_rintf(f * 255.0f);
}
// The same rounding through the float adder: x + 1.5*2^23 is rounded to nearest-even at
// unit spacing, so the low mantissa bits of the sum are rint(x) for 0 <= x < 2^22.
// Returns the raw bits 0x4B400000 + rint(f*255): two full-rate ops, no v_rndne/v_cvt.
CHV_DEV uint32_t to_code_unit_biased(float f) {
    return __float_as_uint(f * 255.0f + 12582912.0f);
}
constexpr uint32_t kCodeBias = 0x4B400000u;

CHV_DEV float clampf(float v, float lo, float hi) {
    return __builtin_fminf(__builtin_fmaxf(v, lo), hi);
}

CHV_DEV float dot4(float x, float y, float z, float w, const float *__restrict__ r) {
    return ((x * r[0] + y * r[1]) + z * r[2]) + w * r[3];
}

// Geometry prologue shared by the composite family (kernels.cl.swift:70-77).
struct Geo {
    float tx, ty;    // tx.xy
    float u, v;      // uv.xy
    bool in_border, in_tx, in_uv;
};

CHV_DEV Geo geometry(const float *__restrict__ U, int x, int y, float sx, float sy) {
    Geo g;
    float ou = (float)x / sx;          // true division: out_uv = gid / size
    float ov = (float)y / sy;
    float nx = ou * 2.f - 1.f, ny = ov * 2.f - 1.f;
    float t0 = dot4(nx, ny, 0.f, 1.f, U + U_TRANSFORM + 0);
    float t1 = dot4(nx, ny, 0.f, 1.f, U + U_TRANSFORM + 4);
    float t2 = dot4(nx, ny, 0.f, 1.f, U + U_TRANSFORM + 8);
    float t3 = dot4(nx, ny, 0.f, 1.f, U + U_TRANSFORM + 12);
    float b0 = dot4(nx, ny, 0.f, 1.f, U + U_BORDER + 0);
    float b1 = dot4(nx, ny, 0.f, 1.f, U + U_BORDER + 4);
    g.tx = t0; g.ty = t1;
    g.u = dot4(t0, t1, t2, t3, U + U_TEXTURE + 0);
    g.v = dot4(t0, t1, t2, t3, U + U_TEXTURE + 4);
    g.in_border = b0 >= 0.f && b1 >= 0.f && b0 <= 1.f && b1 <= 1.f;
    g.in_tx = t0 >= 0.f && t1 >= 0.f && t0 <= 1.f && t1 <= 1.f;
    g.in_uv = g.u >= 0.f && g.v >= 0.f && g.u <= 1.f && g.v <= 1.f;
    return g;
}

// The same prologue for layers flagged LF_AXIS_ALIGNED | LF_BOUNDED (no rotation/shear, bounded entries).  The
// matrix entries the flag guarantees to be exactly zero contribute products that are +-0, and x + (+-0) == x, so
//   dot4((nx,ny,0,1), row0) = fl(fl(nx*m0) + m3)          dot4((nx,ny,0,1), row1) = fl(fl(ny*m5) + m7)
//   dot4(..., row2) = m11   dot4(..., row3) = m15          (rows 2,3 = (0,0,*,*); 0*m10 = 0, 1*m11 = m11)
//   dot4(tx, X row0) = fl(fl(t0*X0) + fl(t3*X3))           dot4(tx, X row1) = fl(fl(t1*X5) + fl(t3*X7))
// are the values `geometry` computes, bit for bit up to the sign of a zero (which no later operation observes):
// 12 instead of 56 multiply/adds per pixel and layer.  tests/test_gpu_fuzz.py compares both against the oracle.
CHV_DEV Geo geometry_axis(const float *__restrict__ U, int x, int y, float sx, float sy) {
    Geo g;
    float ou = (float)x / sx;
    float ov = (float)y / sy;
    float nx = ou * 2.f - 1.f, ny = ov * 2.f - 1.f;
    float t0 = nx * U[U_TRANSFORM + 0] + U[U_TRANSFORM + 3];
    float t1 = ny * U[U_TRANSFORM + 5] + U[U_TRANSFORM + 7];
    float t3 = U[U_TRANSFORM + 15];
    float b0 = nx * U[U_BORDER + 0] + U[U_BORDER + 3];
    float b1 = ny * U[U_BORDER + 5] + U[U_BORDER + 7];
    g.tx = t0; g.ty = t1;
    g.u = t0 * U[U_TEXTURE + 0] + t3 * U[U_TEXTURE + 3];
    g.v = t1 * U[U_TEXTURE + 5] + t3 * U[U_TEXTURE + 7];
    g.in_border = b0 >= 0.f && b1 >= 0.f && b0 <= 1.f && b1 <= 1.f;
    g.in_tx = t0 >= 0.f && t1 >= 0.f && t0 <= 1.f && t1 <= 1.f;
    g.in_uv = g.u >= 0.f && g.v >= 0.f && g.u <= 1.f && g.v <= 1.f;
    return g;
}
CHV_DEV Geo geometry_for(const DLayer &L, int x, int y, float sx, float sy) {
    constexpr int fast = LF_AXIS_ALIGNED | LF_BOUNDED;
    return (L.flags & fast) == fast ? geometry_axis(L.u, x, y, sx, sy) : geometry(L.u, x, y, sx, sy);
}

// One axis of the linear filter: u = s*w; i0 = floor(u-0.5); a = frac(u-0.5).
struct Lin1 {
    int i0, i1;
    float a;
};
CHV_DEV Lin1 lin_axis(float s, int w) {
    Lin1 r;
    float um = s * (float)w - 0.5f;
    float fl = __builtin_floorf(um);
    r.a = um - fl;
    int i = (int)fl;
    r.i0 = min(max(i, 0), w - 1);
    r.i1 = min(max(i + 1, 0), w - 1);
    return r;
}

struct Lin2 {
    int o00, o10, o01, o11;  // byte offsets of the four taps' texels
    float w00, w10, w01, w11;
};
CHV_DEV Lin2 lin_setup(const DPlane &p, float s, float t) {
    Lin1 lx = lin_axis(s, p.w), ly = lin_axis(t, p.h);
    Lin2 l;
    int r0 = ly.i0 * p.pitch, r1 = ly.i1 * p.pitch;
    int c0 = lx.i0 * p.comps, c1 = lx.i1 * p.comps;
    l.o00 = r0 + c0; l.o10 = r0 + c1; l.o01 = r1 + c0; l.o11 = r1 + c1;
    float ia = 1.0f - lx.a, ib = 1.0f - ly.a;
    l.w00 = ia * ib;
    l.w10 = lx.a * ib;
    l.w01 = ia * ly.a;
    l.w11 = lx.a * ly.a;
    return l;
}
CHV_DEV float lin_mix(const Lin2 &l, float t00, float t10, float t01, float t11) {
    return ((l.w00 * t00 + l.w10 * t10) + l.w01 * t01) + l.w11 * t11;
}
CHV_DEV float lin_fetch(const DPlane &p, const Lin2 &l, int c) {
    const uint8_t *b = p.ptr + c;
    return lin_mix(l, unorm8(gld<uint8_t>(b + l.o00)), unorm8(gld<uint8_t>(b + l.o10)), unorm8(gld<uint8_t>(b + l.o01)), unorm8(gld<uint8_t>(b + l.o11)));
}

// ---- code-scale arithmetic of the BGRA-target family (spec owned by this repo) ----
// Samples, fill and blend of img_{nv12,y420p,bgra,rgba}_bgra(_tx) are specified on the 0..255 code
// scale with fused multiply-adds (one rounding each); geometry, tap addresses and weights are those
// of the LINEAR sampler above.  oracle/ref_kernels.c::px_to_bgra is the statement of record.
constexpr float kInv255 = 0x1.010102p-8f;   // RN(1/255): alpha scale of RGB sources
CHV_DEV float cs_mix(float w00, float w10, float w01, float w11, float t00, float t10, float t01, float t11) {
    return __builtin_fmaf(w11, t11, __builtin_fmaf(w01, t01, __builtin_fmaf(w10, t10, w00 * t00)));
}
CHV_DEV float cs_mix(const Lin2 &l, float t00, float t10, float t01, float t11) {
    return cs_mix(l.w00, l.w10, l.w01, l.w11, t00, t10, t01, t11);
}
CHV_DEV float cs_fetch(const DPlane &p, const Lin2 &l, int c) {
    const uint8_t *b = p.ptr + c;
    return cs_mix(l, (float)gld<uint8_t>(b + l.o00), (float)gld<uint8_t>(b + l.o10), (float)gld<uint8_t>(b + l.o01), (float)gld<uint8_t>(b + l.o11));
}
// ---- taps without a conversion instruction --------------------------------------------------------------------------
// A staged byte b, zero-extended to 16 bits, IS the binary16 denormal b * 2^-24 (exact for b < 1024).  v_fma_mix_f32 widens
// a binary16 operand inside the multiplier, so with the weight pre-multiplied by 2^24 (exact: a power of two, no overflow
// for weights in [0, 1])
//     fma_mix(h = 0x00bb, w * 2^24, acc)  =  RN(b * 2^-24 * w * 2^24 + acc)  =  fmaf(w, (float)b, acc)       bit for bit,
// and `fma(w, T, +0)` is `w * T` for the non-negative operands of the filter.  One slow-class instruction (1.8 ns per wave on
// a SIMD) where v_cvt_f32_ubyteN + v_fma_f32 cost 1.8 + 1.1 (tools/ubench_tput.cpp, profiles/r03_ubench_tput_gfx950.txt);
// the kernels' FP mode keeps binary16 denormals (.amdhsa_float_denorm_mode_16_64 3, asserted by tests/test_device_code_contract.py)
// and tests/test_gpu_primitives.py checks the identity on the device for every byte.
typedef _Float16 chv_half;
constexpr float kTapScale = 16777216.0f;       // 2^24
CHV_DEV chv_half tap_h(uint32_t byte) { return __builtin_bit_cast(chv_half, (unsigned short)byte); }
CHV_DEV chv_half tap_h(const uint8_t *p) { return tap_h((uint32_t)*p); }
// cs_mix with the four weights already multiplied by kTapScale
CHV_DEV float cs_mix_h(float w00, float w10, float w01, float w11, chv_half t00, chv_half t10, chv_half t01, chv_half t11) {
    return __builtin_fmaf(w11, (float)t11, __builtin_fmaf(w01, (float)t01, __builtin_fmaf(w10, (float)t10, __builtin_fmaf(w00, (float)t00, 0.0f))));
}

// RTE of a code-scale value known to lie in [0, 255 + a few ulp] and not NaN (a convex combination
// of codes), through the float adder: returns the raw bits 0x4B400000 + rint(v)
CHV_DEV uint32_t code_biased(float v) { return __float_as_uint(v + 12582912.0f); }
// the same as a float code value
CHV_DEV float code_rintf(float v) { return (v + 12582912.0f) - 12582912.0f; }

// rgb2yuv rows, kernels.cl.swift:96-99 (0.113 is the reference's value).
CHV_DEV void rgb2yuv(float r, float g, float b, float &y, float &u, float &v) {
    // dot((r,g,b,1), row): the .w lane multiplies 1.0 by the row's offset
    y = ((r * 0.299f + g * 0.587f) + b * 0.113f) + 1.0f * 0.f;
    u = ((r * -0.169f + g * -0.331f) + b * 0.5f) + 1.0f * 0.5f;
    v = ((r * 0.5f + g * -0.419f) + b * -0.081f) + 1.0f * 0.5f;
}

// Integer YUV -> RGB, 16.16 fixed point.
struct Csc { int32_t yoff, cy, crv, cgu, cgv, cbu; };
__device__ __constant__ const Csc kCsc[4] = {
    { 16, 76309, 104597, 25675, 53279, 132201 },  // BT.601 limited
    { 16, 76309, 117489, 13975, 34925, 138438 },  // BT.709 limited
    { 0, 65536, 91881, 22553, 46802, 116130 },    // BT.601 full
    { 0, 65536, 103206, 12276, 30679, 121609 },   // BT.709 full
};
CHV_DEV uint32_t clip8(int32_t v) { return (uint32_t)min(max(v, 0), 255); }

// Integer RGB -> YUV, 16.16 fixed point: the encoder-side counterpart of kCsc.  Luma rows sum to 56284
// (= round(65536 * 219 / 255); 65536 full range), chroma rows to 0: white -> (235, 128, 128), grey has neutral chroma.
struct R2Y { int32_t yoff, y[3], u[3], v[3]; };
__device__ __constant__ const R2Y kR2Y[4] = {
    { 16, { 16829, 33039, 6416 }, { -9714, -19070, 28784 }, { 28784, -24103, -4681 } },  // BT.601 limited
    { 16, { 11966, 40254, 4064 }, { -6596, -22188, 28784 }, { 28784, -26145, -2639 } },  // BT.709 limited
    { 0, { 19595, 38470, 7471 }, { -11058, -21710, 32768 }, { 32768, -27439, -5329 } },  // BT.601 full
    { 0, { 13933, 46871, 4732 }, { -7509, -25259, 32768 }, { 32768, -29763, -3005 } },   // BT.709 full
};
// One row of the matrix: c0 r + c1 g + c2 b + base.  Coefficients are below 2^16 in magnitude and codes below 2^8, so every product fits
// the 24-bit multiplier exactly (v_mad_i32_i24, half rate) — the plain `int` products compile to v_mul_lo_u32, a quarter-rate instruction
// (tools/ubench_tput.cpp: 3.45 against 1.8 ns per wave), five of them per pixel row in the strip kernel's integer-matrix rows.
CHV_DEV int32_t r2y_row(int32_t c0, int32_t c1, int32_t c2, int32_t base, int r, int g, int b) {
    return __mul24(c0, r) + (__mul24(c1, g) + (__mul24(c2, b) + base));
}
// The kernels' form of a row.  The operands are code_biased's raw bits — the 24-bit multiplier sees 2^22 + code, no mask in front of it — and the
// row's offset takes 2^22 x (c0 + c1 + c2) out again, in wrap-around arithmetic (r2y_base_biased); the row's code leaves as a float through
// clamp-to-24-bits + v_cvt_f32_ubyte2, which is clip8(sum >> 16) (pack_bgra_fixed) without the shift.
CHV_DEV int32_t r2y_base_biased(int32_t c0, int32_t c1, int32_t c2, int32_t base) { return (int32_t)((uint32_t)base - ((uint32_t)(c0 + c1 + c2) << 22)); }
CHV_DEV float fixed_to_codef(int32_t sum16) {
    const int32_t c = min(max(sum16, 0), 0xFFFFFF);
    float f;
    asm("v_cvt_f32_ubyte2 %0, %1" : "=v"(f) : "v"(c));
    return f;
}
CHV_DEV void rgb_to_yuv_int(const R2Y &k, int r, int g, int b, uint32_t &y, uint32_t &u, uint32_t &v) {
    y = clip8(r2y_row(k.y[0], k.y[1], k.y[2], (k.yoff << 16) + 32768, r, g, b) >> 16);
    u = clip8(r2y_row(k.u[0], k.u[1], k.u[2], (128 << 16) + 32768, r, g, b) >> 16);
    v = clip8(r2y_row(k.v[0], k.v[1], k.v[2], (128 << 16) + 32768, r, g, b) >> 16);
}

// Pack three 16.16 fixed-point channels into a memory-order BGRA word:
// clip8(x >> 16) == clamp(x, 0, 0xFFFFFF) >> 16, so saturate first and then move the
// integer byte of each channel into place.  (Written this way on purpose: for the
// ashr-then-clamp form hipcc 7.2 selects gfx950's v_ashr_pk_u8_i32 and then ORs the third
// channel into a register whose upper half that instruction does not leave zero.)
CHV_DEV uint32_t pack_bgra_fixed(int32_t b16, int32_t g16, int32_t r16) {
    uint32_t b = (uint32_t)min(max(b16, 0), 0xFFFFFF);
    uint32_t g = (uint32_t)min(max(g16, 0), 0xFFFFFF);
    uint32_t r = (uint32_t)min(max(r16, 0), 0xFFFFFF);
    return (b >> 16) | ((g >> 8) & 0xFF00u) | (r & 0xFF0000u) | 0xFF000000u;
}

// The same packing with gfx950's v_ashr_pk_u8_i32 (shift right, saturate to u8, pack two
// channels per instruction) and one v_perm_b32: {B,G} and {R,255} pairs, then a byte gather.
// Only the low 16 bits of each pair are used (the instruction leaves the upper half undefined
// for our purposes).  Checked against pack_bgra_fixed on device by tests/test_gpu_primitives.py.
CHV_DEV uint32_t pack_bgra_fixed_pk(int32_t b16, int32_t g16, int32_t r16) {
    // (inline asm rather than __builtin_amdgcn_ashr_pk_u8_i32: the builtin's 16-bit result type makes
    // the compiler mask the register before the v_perm, which ignores the upper bytes anyway)
    uint32_t bg, ra;
    asm("v_ashr_pk_u8_i32 %0, %1, %2, 16" : "=v"(bg) : "v"(b16), "v"(g16));          // byte0 = B, byte1 = G
    asm("v_ashr_pk_u8_i32 %0, %1, %2, 16" : "=v"(ra) : "v"(r16), "v"(0x00FF0000));   // byte0 = R, byte1 = 255
    // v_perm_b32(S0, S1, sel): selector bytes 0..3 pick from S1, 4..7 from S0
    return __builtin_amdgcn_perm(ra, bg, 0x05040100u);
}

// returns memory-order BGRA word: B | G<<8 | R<<16 | 255<<24
CHV_DEV uint32_t yuv_to_bgra_word(const Csc &k, int y, int u, int v) {
    int32_t c = k.cy * (y - k.yoff) + 32768;
    int32_t d = u - 128, e = v - 128;
    return pack_bgra_fixed(c + k.cbu * d, c - k.cgu * d - k.cgv * e, c + k.crv * e);
}

// The same matrix with the offsets folded into one constant per channel and 24-bit
// multiplies (v_mad_i32_i24, full rate): all intermediate sums stay below 2^27, so the
// regrouping is exact integer arithmetic.
CHV_DEV int32_t mad24(int32_t a, int32_t b, int32_t c) { return __mul24(a, b) + c; }
// v_mad_i32_i24 spelled out: left to itself hipcc 7.2 splits multiply-adds into v_mul_i32_i24 + v_add3_u32,
// one more multiplier-pipe instruction per term; `coef` must be wave-uniform (an SGPR)
CHV_DEV int32_t mad24_uniform(int32_t a, int32_t coef, int32_t c) {
    int32_t d;
    asm("v_mad_i32_i24 %0, %1, %2, %3" : "=v"(d) : "v"(a), "s"(coef), "v"(c));
    return d;
}
struct CscFolded { int32_t cy, crv, ncgu, ncgv, cbu, kr, kg, kb; };
CHV_DEV CscFolded csc_fold(const Csc &k) {
    CscFolded f;
    int32_t base = 32768 - k.cy * k.yoff;
    f.cy = k.cy; f.crv = k.crv; f.ncgu = -k.cgu; f.ncgv = -k.cgv; f.cbu = k.cbu;
    f.kr = base - 128 * k.crv;
    f.kg = base + 128 * (k.cgu + k.cgv);
    f.kb = base - 128 * k.cbu;
    return f;
}
// Operands carry the bias of to_code_unit_biased: v_mul_i32_i24 reads only bits [23:0]
// (= 2^22 + code, still a positive 24-bit number) and the resulting constant
// 2^22 * coefficient is folded into the channel offsets, in wrap-around arithmetic.
CHV_DEV CscFolded csc_fold_biased(const Csc &k) {
    CscFolded f = csc_fold(k);
    const uint32_t b = 1u << 22;
    f.kr = (int32_t)((uint32_t)f.kr - b * (uint32_t)f.cy - b * (uint32_t)f.crv);
    f.kg = (int32_t)((uint32_t)f.kg - b * (uint32_t)f.cy - b * (uint32_t)f.ncgu - b * (uint32_t)f.ncgv);
    f.kb = (int32_t)((uint32_t)f.kb - b * (uint32_t)f.cy - b * (uint32_t)f.cbu);
    return f;
}
CHV_DEV uint32_t yuv_to_bgra_word(const CscFolded &k, int y, int u, int v) {
    // one luma product shared by the three channels, chroma terms as 24-bit multiply-adds
    // (v_mad_i32_i24), channel offsets as plain adds: 5 multiplier-pipe ops + 3 adds per pixel
    int32_t t = __mul24(y, k.cy);
    int32_t r, g, b;
    r = mad24_uniform(v, k.crv, t) + k.kr;
    g = mad24_uniform(v, k.ncgv, mad24_uniform(u, k.ncgu, t)) + k.kg;
    b = mad24_uniform(u, k.cbu, t) + k.kb;
    return pack_bgra_fixed_pk(b, g, r);
}

// The same matrix on biased codes (csc_fold_biased), the channels returned as float codes 0..255 for a blend that follows (tick_bgra_wave,
// tick_bgra_stream): clamp to 24 bits, then the integer byte of each 16.16 channel through v_cvt_f32_ubyte2.
CHV_DEV void yuv_to_bgr_floats(const CscFolded &k, int y, int u, int v, float &fb, float &fg, float &fr) {
    int32_t t = __mul24(y, k.cy);
    int32_t r = mad24_uniform(v, k.crv, t) + k.kr;
    int32_t g = mad24_uniform(v, k.ncgv, mad24_uniform(u, k.ncgu, t)) + k.kg;
    int32_t b = mad24_uniform(u, k.cbu, t) + k.kb;
    // (v_cvt_f32_ubyte2 spelled out: left alone hipcc picks v_cvt_f32_u32_sdwa src0_sel:WORD_1, and SDWA forms — like v_fma_mix_f32,
    // the 24-bit multiplies and v_perm_b32 — do not pair with a neighbouring f32 instruction, while v_cvt_f32_ubyteN, v_med3_i32 and
    // v_cvt_pk_u8_f32 do: tools/ubench_pair.cpp, profiles/r03_ubench_pair_gfx950.txt)
    const int32_t cb = min(max(b, 0), 0xFFFFFF), cg = min(max(g, 0), 0xFFFFFF), cr = min(max(r, 0), 0xFFFFFF);
    asm("v_cvt_f32_ubyte2 %0, %1" : "=v"(fb) : "v"(cb));
    asm("v_cvt_f32_ubyte2 %0, %1" : "=v"(fg) : "v"(cg));
    asm("v_cvt_f32_ubyte2 %0, %1" : "=v"(fr) : "v"(cr));
}

// ---- the same matrix with the red and blue channel offsets ABSORBED into the float -> code conversion -------------------------------
// code_biased adds 1.5 x 2^23 to a sample and hands the raw bits to the 24-bit multiplier, which sees 2^22 + code; csc_fold_biased folds
// 2^22 x coefficient into the three channel offsets.  Any other even bias B works the same way (M = 2^23 + B: bits 0x4B000000 + B + rint(f),
// ties still to the even code; B + 255 < 2^23), and so does M = -(2^23 + B): the operand is B - rint(f) and the layer's coefficients for that
// operand change sign.  A bias adds coefficient x bias to every channel its operand feeds, so biases (by, bu, bv) with
//     crv bv + cy by = KR   and   cbu bu + cy by = KB   (mod 2^32)
// leave the red and the blue channel FINISHED after their last multiply-add: two vector adds fewer per pixel and layer, the third offset
// becomes kg' = KG - (cy by - cgu bu - cgv bv).  tools/csc_absorb_search.py finds them (126 / 127 / 0 / 256 solutions for the four matrices:
// BT.601 full range has none — its luma coefficient 2^16 leaves the luma bias sixteen useful bits) and tests/test_host_logic.py re-derives
// the table.  Same integers into the same clamp: the same bytes (tests/test_gpu_matrices.py runs every code triple through both forms).
// The table holds the folded form: coefficients with their operand's sign, kg', and the three conversion constants +-(2^23 + |bias|)
// (printed by the tool; scalar loads from constant memory in the kernels).
struct CscAbsorbed { int32_t cy, crv, ncgu, ncgv, cbu, kg; float my, mu, mv; };
__device__ __constant__ const CscAbsorbed kCscAbsorbed[4] = {
    { 76309, 104597, -25675, -53279, 132201, 1649788258, 10041594.0f, 13672062.0f, 9933686.0f },      // BT.601 limited: biases 1652986, 5283454, 1545078
    { 76309, 117489, -13975, -34925, 138438, -1793622274, 8659076.0f, 15468090.0f, 11137308.0f },     // BT.709 limited: biases 270468, 7079482, 2748700
    { 0, 0, 0, 0, 0, 0, 0.f, 0.f, 0.f },                                                              // BT.601 full: no such biases
    { 65536, -103206, 12276, 30679, -121609, -223690752, 8400986.0f, -15564928.0f, -9977984.0f },     // BT.709 full: biases 12378, -7176320, -1589376
};
constexpr bool csc_absorbable(int csc) { return (csc & 3) != 2; }
CHV_DEV CscAbsorbed csc_fold_absorbed(int csc) { return kCscAbsorbed[csc & 3]; }
// The channels as 16.16 sums clamped to [0, 2^24): the code is byte 2 — as a binary16 read from the register's HIGH half, code x 2^-24
// (code_h below: the denormal trick of tap_h), which is how tick_bgra_stream's blend consumes it: one v_fma_mix_f32 per channel where
// v_cvt_f32_ubyte2 + v_fmac_f32 were two.
CHV_DEV void yuv_to_bgr_fixed_absorbed(const CscAbsorbed &k, float fy, float fu, float fv, int32_t &cb, int32_t &cg, int32_t &cr) {
    const int y = (int)__float_as_uint(fy + k.my), u = (int)__float_as_uint(fu + k.mu), v = (int)__float_as_uint(fv + k.mv);
    const int32_t t = __mul24(y, k.cy);
    const int32_t r = mad24_uniform(v, k.crv, t);
    const int32_t g = mad24_uniform(v, k.ncgv, mad24_uniform(u, k.ncgu, t)) + k.kg;
    const int32_t b = mad24_uniform(u, k.cbu, t);
    cb = min(max(b, 0), 0xFFFFFF); cg = min(max(g, 0), 0xFFFFFF); cr = min(max(r, 0), 0xFFFFFF);
}
// ... and packed into a memory-order BGRA word (yuv_to_bgra_word on biased operands, two adds shorter): the tiled kernel's opaque pixel
CHV_DEV uint32_t yuv_to_bgra_word_absorbed(const CscAbsorbed &k, float fy, float fu, float fv) {
    const int y = (int)__float_as_uint(fy + k.my), u = (int)__float_as_uint(fu + k.mu), v = (int)__float_as_uint(fv + k.mv);
    const int32_t t = __mul24(y, k.cy);
    const int32_t r = mad24_uniform(v, k.crv, t);
    const int32_t g = mad24_uniform(v, k.ncgv, mad24_uniform(u, k.ncgu, t)) + k.kg;
    const int32_t b = mad24_uniform(u, k.cbu, t);
    return pack_bgra_fixed_pk(b, g, r);
}
CHV_DEV chv_half code_h(int32_t fixed24) { return __builtin_bit_cast(chv_half, (unsigned short)((uint32_t)fixed24 >> 16)); }
// yuv_to_bgr_floats on FLOAT samples in code scale (the conversion is part of the form)
CHV_DEV void yuv_to_bgr_floats_absorbed(const CscAbsorbed &k, float fy, float fu, float fv, float &fb, float &fg, float &fr) {
    const int y = (int)__float_as_uint(fy + k.my), u = (int)__float_as_uint(fu + k.mu), v = (int)__float_as_uint(fv + k.mv);
    const int32_t t = __mul24(y, k.cy);
    const int32_t r = mad24_uniform(v, k.crv, t);
    const int32_t g = mad24_uniform(v, k.ncgv, mad24_uniform(u, k.ncgu, t)) + k.kg;
    const int32_t b = mad24_uniform(u, k.cbu, t);
    const int32_t cb = min(max(b, 0), 0xFFFFFF), cg = min(max(g, 0), 0xFFFFFF), cr = min(max(r, 0), 0xFFFFFF);
    asm("v_cvt_f32_ubyte2 %0, %1" : "=v"(fb) : "v"(cb));
    asm("v_cvt_f32_ubyte2 %0, %1" : "=v"(fg) : "v"(cg));
    asm("v_cvt_f32_ubyte2 %0, %1" : "=v"(fr) : "v"(cr));
}

}  // namespace chv

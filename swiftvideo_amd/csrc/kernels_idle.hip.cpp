// kernels_idle.hip.cpp — the two kernels of `enum ComputeKernel` (compute.swift:67,70) that no caller of the reference dispatches:
// snd_s16i_s16i (kernels.cl.swift:534-562, OpenCL) and me_fullsearch (kernels.metal:129-267, Metal only).  Built so that every case of the
// enum resolves to something runnable; bit-exact against oracle/ref_kernels.c::orc_snd_s16i_s16i / orc_me_fullsearch, and
// snd_s16i_s16i also against the reference's own kernel string compiled for x86-64 (oracle/clref).
#include "pixel_math.hip.h"

#include <cstring>

#pragma clang fp contract(off)

namespace chv {

// ---------------------------------------------------------------------------------------------------------------------
// snd_s16i_s16i: out[gid] += (short) min(in_i[gid] * gain_i * (gid even ? 1 - fade_i : fade_i), 32767.f) for every input i
// ---------------------------------------------------------------------------------------------------------------------
struct SndArgs {
    int16_t *out;
    const int16_t *in[8];
    int32_t n, count, vec;          // vec: every pointer is 16-byte aligned (eight samples per 128-bit access)
    float gain[8], fade[8];
};

// float -> short as the reference's kernel does it where it was compiled (oracle/ref_kernels.c::snd_cvt): toward zero into 32 bits — the
// conversion instruction saturates below -2^31 to INT32_MIN and turns NaN into 0, both of which have zero low halves like x86's INT32_MIN —
// then the low 16 bits.  `min` caps the top only (kernels.cl.swift:557).
CHV_DEV uint32_t snd_term(int16_t s, float gain, float k) {
    const float x = (float)s * gain * k;
    const float v = 32767.f < x ? 32767.f : x;
    return (uint32_t)__float2int_rz(v) & 0xFFFFu;
}

__global__ __launch_bounds__(256) void snd_s16i_s16i_kernel(const SndArgs a) {
    const int g0 = ((int)blockIdx.x * 256 + (int)threadIdx.x) * 8;
    if (g0 >= a.n) return;
    if (a.vec && g0 + 8 <= a.n) {
        // eight samples = four stereo frames: one 128-bit access per buffer
        uint4 o = gld<uint4>(a.out + g0);
        uint32_t w[4] = { o.x, o.y, o.z, o.w };
        for (int i = 0; i < a.count; i++) {
            const uint4 v = gld<uint4>(a.in[i] + g0);
            const uint32_t s[4] = { v.x, v.y, v.z, v.w };
            const float kl = 1.f - a.fade[i], kr = a.fade[i];
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const uint32_t l = ((w[q] & 0xFFFFu) + snd_term((int16_t)(s[q] & 0xFFFFu), a.gain[i], kl)) & 0xFFFFu;
                const uint32_t r = ((w[q] >> 16) + snd_term((int16_t)(s[q] >> 16), a.gain[i], kr)) & 0xFFFFu;
                w[q] = l | (r << 16);
            }
        }
        *(uint4 *)(a.out + g0) = make_uint4(w[0], w[1], w[2], w[3]);
        return;
    }
    for (int g = g0; g < min(g0 + 8, a.n); g++) {
        uint32_t acc = (uint16_t)a.out[g];
        for (int i = 0; i < a.count; i++) acc = (acc + snd_term(a.in[i][g], a.gain[i], (g & 1) == 0 ? 1.f - a.fade[i] : a.fade[i])) & 0xFFFFu;
        a.out[g] = (int16_t)(uint16_t)acc;
    }
}

hipError_t launch_snd_s16i(int16_t *out, const int16_t *const *in, int count, int n, const float *gain, const float *fade, hipStream_t stream) {
    if (n <= 0) return hipSuccess;
    SndArgs a;
    memset(&a, 0, sizeof a);
    a.out = out; a.n = n; a.count = count;
    a.vec = (((uintptr_t)out) & 15) == 0;
    for (int i = 0; i < count; i++) {
        a.in[i] = in[i]; a.gain[i] = gain[i]; a.fade[i] = fade[i];
        a.vec = a.vec && (((uintptr_t)in[i]) & 15) == 0;
    }
    const int threads = (n + 7) / 8;
    (void)hipGetLastError();                   // (a stale error of an earlier call is not this launch's)
    hipLaunchKernelGGL(snd_s16i_s16i_kernel, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, stream, a);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------------------------
// me_fullsearch: one WAVE per block of the current picture; lane = one column of candidate positions in the reference picture's search area
// (at most 63: MAX_SEARCH_SIZE 64, kernels.metal:204), walked top to bottom exactly as the source walks it — including its sliding-window
// SAD, which after a candidate with a non-zero top-row SAD returns previousSad - previousSide (:152-166, the second loop never runs), and the
// early exit that value triggers when it falls below zero (threshold 0, :221,247-249: a wave ballot picks the first column that got there).  Sums
// accumulate in float in source order, so there is nothing to reduce across lanes but the final (score, visiting order) minimum.
// The current block and the search area are staged in LDS as bytes (<= 4 KB each); texel values are c / 255.0f like every R8Unorm read here.
// ---------------------------------------------------------------------------------------------------------------------
struct MeArgs {
    DPlane out, ref, cur;
    int32_t bsx, bsy, swx, swy, imw, imh;      // swx / swy: min(searchWindowSize, 64)
    float maxx, maxy;                          // float(searchWindowSize / 2)
    float cost[256];                           // per-component vector cost by |v| (deltaCost2 through the HOST's log2f: the table the oracle uses)
};

__global__ __launch_bounds__(64) void me_fullsearch_kernel(const MeArgs a) {
    __shared__ uint8_t s_cur[64 * 64], s_ref[64 * 64];
    const int bx = blockIdx.x, by = blockIdx.y, lane = threadIdx.x;
    const int ox = bx * a.bsx, oy = by * a.bsy;
    auto clampi = [](int v, int lo, int hi) { return min(max(v, lo), hi); };
    const int left = clampi(ox + a.bsx / 2 - a.swx / 2, 0, a.imw), top = clampi(oy + a.bsy / 2 - a.swy / 2, 0, a.imh);
    const int right = clampi(left + a.swx, 0, a.imw), bottom = clampi(top + a.swy, 0, a.imh);
    const int aw = right - left, ah = bottom - top;                                   // search area, <= 64 x 64
    // stage: reads outside a picture are 0 (an origin block over the right / bottom edge; the search area is clamped to imageSize, which may
    // exceed the plane)
    for (int i = lane; i < a.bsx * a.bsy; i += 64) {
        const int x = ox + i % a.bsx, y = oy + i / a.bsx;
        s_cur[i] = (x < a.cur.w && y < a.cur.h) ? gld<uint8_t>(a.cur.ptr + (size_t)y * a.cur.pitch + x) : (uint8_t)0;
    }
    for (int i = lane; i < aw * ah; i += 64) {
        const int x = left + i % aw, y = top + i / aw;
        s_ref[i] = (x < a.ref.w && y < a.ref.h) ? gld<uint8_t>(a.ref.ptr + (size_t)y * a.ref.pitch + x) : (uint8_t)0;
    }
    __syncthreads();
    float best = 3.402823466e+38f, bmx = 0.f, bmy = 0.f;
    bool negative = false;                                       // this column reached a score < threshold = 0 (:221,247-249)
    const int ncols = aw - a.bsx, nrows = ah - a.bsy;            // candidates: refBlock.z < searchArea.z, refBlock.w < searchArea.w (strict, :229,232)
    if (lane < ncols) {
        float side = 0.f, prev = 0.f;
        for (int r = 0; r < nrows; r++) {
            float sum = 0.f, tp = 0.f;
            if (side > 0.f) {
                for (int x = 0; x < a.bsx; x++) tp += __builtin_fabsf(unorm8((uint32_t)s_cur[x]) - unorm8((uint32_t)s_ref[r * aw + lane + x]));
                sum = prev - side + 0.f;
            } else {
                for (int y = 0; y < a.bsy; y++)
                    for (int x = 0; x < a.bsx; x++) {
                        const float d = __builtin_fabsf(unorm8((uint32_t)s_cur[y * a.bsx + x]) - unorm8((uint32_t)s_ref[(r + y) * aw + lane + x]));
                        sum += d;
                        if (y == 0) tp += d;
                    }
            }
            const int mx = ox - (left + lane), my = oy - (top + r);
            const float score = 4.0f * (a.cost[min(abs(mx), 255)] + a.cost[min(abs(my), 255)]) + sum * 256.0f;
            prev = sum; side = tp;
            if (score < best) {
                best = score;
                bmx = __builtin_fminf(__builtin_fmaxf((float)mx, -a.maxx), a.maxx);
                bmy = __builtin_fminf(__builtin_fmaxf((float)my, -a.maxy), a.maxy);
            }
            // earlyExit: the running "SAD" of the sliding form loses a top-row SAD per row and does go negative in tall windows.  Every score
            // before this one in the column was >= 0, so this candidate has just become the column's best; the column stops here
            if (score < 0.f) { negative = true; break; }
        }
    }
    // The source visits columns left to right and ends the WHOLE search at the first negative score: the lowest column that has one wins with
    // that candidate (all scores of the columns before it are >= 0 and lose to it; the columns after it are never visited).
    const unsigned long long negs = __ballot(negative);
    if (negs != 0ull) {
        const int first = __ffsll((long long)negs) - 1;
        bmx = __shfl(bmx, first); bmy = __shfl(bmy, first);
        best = -3.402823466e+38f;                                // (every lane now holds the winner: the reduction below keeps it)
    }
    // otherwise the first strict minimum in visiting order: columns left to right, so among equal scores the lowest lane
    int who = lane;
    for (int off = 32; off >= 1; off >>= 1) {
        const float ob = __shfl_xor(best, off);
        const int ow = __shfl_xor(who, off);
        const float ox_ = __shfl_xor(bmx, off), oy_ = __shfl_xor(bmy, off);
        if (ob < best || (ob == best && ow < who)) { best = ob; who = ow; bmx = ox_; bmy = oy_; }
    }
    if (lane == 0 && bx < a.out.w && by < a.out.h) {
        bmx = bmx / a.maxx; bmy = bmy / a.maxy;
        bmx = bmx * 0.5f + 0.5f; bmy = bmy * 0.5f + 0.5f;
        const uint32_t w = to_code(bmx) | (to_code(0.5f) << 8) | (to_code(bmy) << 16) | (to_code(1.0f) << 24);
        *(uint32_t *)(a.out.ptr + (size_t)by * a.out.pitch + (size_t)bx * 4) = w;
    }
}

hipError_t launch_me_fullsearch(const DPlane &out, const DPlane &ref, const DPlane &cur, const int32_t *block, const int32_t *window, const int32_t *image,
                                const float *cost256, hipStream_t stream) {
    if (out.w <= 0 || out.h <= 0) return hipSuccess;
    MeArgs a;
    a.out = out; a.ref = ref; a.cur = cur;
    a.bsx = block[0]; a.bsy = block[1];
    a.swx = window[0] < 64 ? window[0] : 64; a.swy = window[1] < 64 ? window[1] : 64;
    a.imw = image[0]; a.imh = image[1];
    a.maxx = (float)(window[0] / 2); a.maxy = (float)(window[1] / 2);
    memcpy(a.cost, cost256, sizeof a.cost);
    (void)hipGetLastError();
    hipLaunchKernelGGL(me_fullsearch_kernel, dim3((unsigned)out.w, (unsigned)out.h), dim3(64), 0, stream, a);
    return hipGetLastError();
}

}  // namespace chv

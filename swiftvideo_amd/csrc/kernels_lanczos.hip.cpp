// kernels_lanczos.hip.cpp — separable Lanczos-3 resampler for 4-component planes
// (no reference counterpart).
//
// One block produces a 32 x 16 output tile in three LDS-resident phases:
//   A. stage the source rectangle the tile's taps touch (raw BGRA texels, 16-byte coalesced
//      global loads; columns/rows outside the picture are resolved by clamping the LOAD
//      address, so the taps below never clamp),
//   B. horizontal pass: a thread owns one output column (its tap weights live in registers)
//      and walks the staged rows; float4 results go to a second LDS array,
//   C. vertical pass out of that array, packed BGRA store.
// Both passes accumulate with one fused multiply-add per tap, taps in ascending order from
// 0.0f — the same chain the oracle evaluates with fmaf().
#include "pixel_math.hip.h"

#include <algorithm>
#include <type_traits>
#include <utility>

#pragma clang fp contract(off)

namespace chv {

// Output tile per block: 32 x 16 normally; 8 x 4 for large reduction factors, whose source rectangle
// (tile * scale + taps in each direction) would not fit the LDS otherwise.
constexpr int LZ_MAXT = 24;      // taps held in registers (scale <= 4); more taps use the slow loop
// `ks` tiles per block, side by side (the host's choice, see launch_lanczos): tile t+1's source rectangle is prefetched into
#ifndef CHV_LZ_MIN_ROWS
#define CHV_LZ_MIN_ROWS 8        // (measured: 4, 8, 16 -> 27.5 / 23.3 / 30.8 us for one 1080p -> 720p resize, wall time with the host wait)
#endif
#ifndef CHV_LZ_STRIP_ANY
#define CHV_LZ_STRIP_ANY 1   // A/B: 0 keeps every ratio outside the 2:1 class on the tile kernel
#endif
#ifndef CHV_LZ_STRIP
#define CHV_LZ_STRIP 1
#endif
constexpr int LZ_NPRE = 6;       // registers (one 16-byte vector per row rg + 8n, n < LZ_NPRE) while tile t is filtered

// EXACT: tx == ty == TAPS_IN_REGS, the tap loops are straight-line code.  Otherwise taps beyond the table's count
// re-read the last tap with weight 0 (fma(0, finite, acc) == acc): still no branch per tap, at one address add each.
template <int TAPS_IN_REGS, bool EXACT, bool PREFETCH, int LZ_TW, int LZ_TH>
__global__ __launch_bounds__(256, (TAPS_IN_REGS > 12 ? 2 : EXACT ? 4 : 3)) void lanczos3_bgra(DPlane dst, DPlane src,
                                                     const int32_t *__restrict__ fx, const float *__restrict__ wx, int tx,
                                                     const int32_t *__restrict__ fy, const float *__restrict__ wy, int ty,
                                                     int max_rows, int max_cols, int ks, const DPlane *__restrict__ batch) {
    // batch != nullptr: grid.z images of one geometry in one launch, (dst, src) pairs in device-visible memory
    if (batch) { dst = batch[2 * blockIdx.z]; src = batch[2 * blockIdx.z + 1]; }
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float4 *hrow = (float4 *)smem;                                   // [max_rows][LZ_TW]
    uint32_t *stile = (uint32_t *)(smem + (size_t)max_rows * LZ_TW * sizeof(float4));   // [max_rows][max_cols]
    const int oy0 = blockIdx.y * LZ_TH;
    const int oy_last = min(oy0 + LZ_TH, dst.h) - 1;
    const int row0 = fy[oy0];
    const int nrows = min(fy[oy_last] + ty - row0, max_rows);
    const int tid = threadIdx.x;
    const int v = tid & 31, rg = tid >> 5;                           // staging: 32 lanes walk a row in 16-byte vectors, 8 rows at a time

    // source rectangle of a tile: first column, width in 16-byte vectors (4 texels; max_cols is a multiple of 4)
    auto tile_geom = [&](int ox0, int &col0, int &nvec) {
        const int ox_last = min(ox0 + LZ_TW, dst.w) - 1;
        col0 = fx[ox0];
        nvec = (min(fx[ox_last] + tx - col0, max_cols) + 3) >> 2;
    };
    // One vector (4 texels from column c) of source row r with CLAMP_TO_EDGE applied at staging time, so the taps
    // below never clamp.  Split in two so that the load is a single instruction on every path (a prefetch that merges
    // differently-produced values makes the compiler wait for it on the spot): load_vec reads the aligned-in-range
    // vector at clamp(c, 0, w-4) (texels are 4-byte aligned, all global_load_dwordx4 needs; src.w >= 4 host-checked),
    // fix_vec re-orders its texels for the vectors that stick out of the picture.
    auto load_vec = [&](int r, int c) {
        const int sy = min(max(row0 + r, 0), src.h - 1);
        return gld<uint4>(src.ptr + (size_t)sy * src.pitch + (size_t)min(max(c, 0), src.w - 4) * 4);
    };
    auto fix_vec = [&](uint4 L, int c) {
        const int cc = min(max(c, 0), src.w - 4);
        if (c == cc) return L;
        auto pick = [&](int k) {
            const int idx = min(max(c + k, 0), src.w - 1) - cc;
            return idx == 0 ? L.x : idx == 1 ? L.y : idx == 2 ? L.z : L.w;
        };
        return make_uint4(pick(0), pick(1), pick(2), pick(3));
    };

    // Vertical pass, per thread: output column tid % LZ_TW of every tile, rows cj + n * CRG.  Row positions and
    // row weights do not depend on the tile, so they are fetched once per block (a load per tap inside the tap
    // loop made the vertical pass a chain of dependent memory latencies: 2/3 of the kernel's time).
    constexpr int CT = TAPS_IN_REGS > 0 ? TAPS_IN_REGS : 1;
    constexpr int CRG = 256 / LZ_TW;                      // row groups of the vertical pass
    constexpr int CNR = (LZ_TH + CRG - 1) / CRG;          // rows per thread: 2 (32 x 16 tiles) or 1 (8 x 4)
    const int ci = tid & (LZ_TW - 1), cj = tid / LZ_TW;
    const bool rows_in_regs = EXACT ? true : (TAPS_IN_REGS > 0 && ty <= TAPS_IN_REGS);
    float wrow[CNR][CT];
    int rbase_c[CNR];
#pragma unroll
    for (int n = 0; n < CNR; n++) {
        const int oy = min(oy0 + cj + n * CRG, dst.h - 1);
        rbase_c[n] = fy[oy] - row0;
        if (rows_in_regs) {
#pragma unroll
            for (int k = 0; k < CT; k++) {
                const float wk = wy[(size_t)oy * ty + (EXACT ? k : min(k, ty - 1))];
                wrow[n][k] = (EXACT || k < ty) ? wk : 0.f;
            }
        }
    }

    uint4 pre[LZ_NPRE];
    int col0 = 0, nvec = 0;
    const int first_ox0 = blockIdx.x * (ks * LZ_TW);
    if (PREFETCH && first_ox0 < dst.w) {
        tile_geom(first_ox0, col0, nvec);
#pragma unroll
        for (int n = 0; n < LZ_NPRE; n++) if (rg + 8 * n < nrows && v < nvec) pre[n] = load_vec(rg + 8 * n, col0 + 4 * v);
    }

    for (int t = 0; t < ks; t++) {
    const int ox0 = first_ox0 + t * LZ_TW;
    if (ox0 >= dst.w) break;

    // this thread's output column in the horizontal pass: tap position and weights, fetched before the staging
    // so that their latency hides behind it
    if (!PREFETCH) tile_geom(ox0, col0, nvec);
    const int bx = min(ox0 + (tid & (LZ_TW - 1)), dst.w - 1);
    const int cbase = fx[bx] - col0;
    const float *w = wx + (size_t)bx * tx;
    float wr[TAPS_IN_REGS > 0 ? TAPS_IN_REGS : 1];
    if (TAPS_IN_REGS > 0) {
#pragma unroll
        for (int k = 0; k < TAPS_IN_REGS; k++) {
            const float wk = w[EXACT ? k : min(k, tx - 1)];
            wr[k] = (EXACT || k < tx) ? wk : 0.f;
        }
    }

    // phase A: source rectangle -> LDS
    if (PREFETCH) {
        touch_regs_volatile(pre);                 // the wait for the prefetch, on every path (see touch_regs)
        // block-uniform test: a rectangle inside the picture's columns needs no texel re-ordering
        if (col0 >= 0 && col0 + 4 * nvec <= src.w) {
#pragma unroll
            for (int n = 0; n < LZ_NPRE; n++) if (rg + 8 * n < nrows && v < nvec) *(uint4 *)(stile + (rg + 8 * n) * max_cols + 4 * v) = pre[n];
        } else {
#pragma unroll
            for (int n = 0; n < LZ_NPRE; n++) if (rg + 8 * n < nrows && v < nvec) *(uint4 *)(stile + (rg + 8 * n) * max_cols + 4 * v) = fix_vec(pre[n], col0 + 4 * v);
        }
    } else {
        for (int r = rg; r < nrows; r += 8)
            for (int vv = v; vv < nvec; vv += 32) {
                // no prefetch (large scale factors, or pictures narrower than one vector): gather texel by texel
                const int c = col0 + 4 * vv;
                const uint8_t *srow = src.ptr + (size_t)min(max(row0 + r, 0), src.h - 1) * src.pitch;
                uint4 t4;
                t4.x = gld<uint32_t>(srow + (size_t)min(max(c, 0), src.w - 1) * 4);
                t4.y = gld<uint32_t>(srow + (size_t)min(max(c + 1, 0), src.w - 1) * 4);
                t4.z = gld<uint32_t>(srow + (size_t)min(max(c + 2, 0), src.w - 1) * 4);
                t4.w = gld<uint32_t>(srow + (size_t)min(max(c + 3, 0), src.w - 1) * 4);
                *(uint4 *)(stile + r * max_cols + 4 * vv) = t4;
            }
    }
    __syncthreads();
    if (PREFETCH && t + 1 < ks && ox0 + LZ_TW < dst.w) {
        tile_geom(ox0 + LZ_TW, col0, nvec);
#pragma unroll
        for (int n = 0; n < LZ_NPRE; n++) if (rg + 8 * n < nrows && v < nvec) pre[n] = load_vec(rg + 8 * n, col0 + 4 * v);
    }

    // phase B: horizontal pass, one output column per thread
    {
        constexpr int RGS = 256 / LZ_TW;                   // row groups: 8 for 32-wide tiles, 32 for 8-wide ones
        const int i = tid & (LZ_TW - 1), rg = tid / LZ_TW;
        if (TAPS_IN_REGS > 0) {
            // the taps of row r + RGS are read from LDS while row r is filtered: left to itself the compiler issues
            // one LDS read per tap pair and waits for it on the spot (6 exposed LDS latencies per row)
            constexpr int T = TAPS_IN_REGS > 0 ? TAPS_IN_REGS : 1;
            uint32_t p[T], q[T];
            auto load_row = [&](int r, uint32_t (&d)[T]) {
                const uint32_t *row = stile + r * max_cols + cbase;
#pragma unroll
                for (int k = 0; k < T; k++) d[k] = row[EXACT ? k : min(k, tx - 1)];
            };
            int r = rg;
            if (r < nrows) load_row(r, p);
            for (; r < nrows; r += RGS) {
                if (r + RGS < nrows) load_row(r + RGS, q);
                __builtin_amdgcn_sched_barrier(0);
                float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
                for (int k = 0; k < T; k++) {
                    acc.x = __builtin_fmaf(wr[k], (float)(p[k] & 255), acc.x);
                    acc.y = __builtin_fmaf(wr[k], (float)((p[k] >> 8) & 255), acc.y);
                    acc.z = __builtin_fmaf(wr[k], (float)((p[k] >> 16) & 255), acc.z);
                    acc.w = __builtin_fmaf(wr[k], (float)(p[k] >> 24), acc.w);
                }
                hrow[r * LZ_TW + i] = acc;
#pragma unroll
                for (int k = 0; k < T; k++) p[k] = q[k];
            }
        } else {
            for (int r = rg; r < nrows; r += RGS) {
                const uint32_t *row = stile + r * max_cols + cbase;
                float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
                for (int k = 0; k < tx; k++) {
                    uint32_t p = row[k];
                    float wk = w[k];
                    acc.x = __builtin_fmaf(wk, (float)(p & 255), acc.x);
                    acc.y = __builtin_fmaf(wk, (float)((p >> 8) & 255), acc.y);
                    acc.z = __builtin_fmaf(wk, (float)((p >> 16) & 255), acc.z);
                    acc.w = __builtin_fmaf(wk, (float)(p >> 24), acc.w);
                }
                hrow[r * LZ_TW + i] = acc;
            }
        }
    }
    __syncthreads();

    // phase C: vertical pass out of LDS
    if (rows_in_regs) {
#pragma unroll
        for (int n = 0; n < CNR; n++) {
            const int j = cj + n * CRG, ox = ox0 + ci, oy = oy0 + j;
            if (j >= LZ_TH || ox >= dst.w || oy >= dst.h) continue;
            const float4 *col = hrow + rbase_c[n] * LZ_TW + ci;
            float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
            for (int k = 0; k < CT; k++) {
                const float4 h = col[(EXACT ? k : min(k, ty - 1)) * LZ_TW];
                const float wk = wrow[n][k];
                acc.x = __builtin_fmaf(wk, h.x, acc.x);
                acc.y = __builtin_fmaf(wk, h.y, acc.y);
                acc.z = __builtin_fmaf(wk, h.z, acc.z);
                acc.w = __builtin_fmaf(wk, h.w, acc.w);
            }
            uint32_t o = pack_codes(acc.x, acc.y, acc.z, acc.w);
            gst<uint32_t>(dst.ptr + (size_t)oy * dst.pitch + (size_t)ox * 4, o);
        }
    } else {
        for (int idx = tid; idx < LZ_TW * LZ_TH; idx += 256) {
            int j = idx / LZ_TW, i = idx % LZ_TW;
            int ox = ox0 + i, oy = oy0 + j;
            if (ox >= dst.w || oy >= dst.h) continue;
            int rbase = fy[oy] - row0;
            const float *w = wy + (size_t)oy * ty;
            float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
            for (int k = 0; k < ty; k++) {
                float4 h = hrow[(rbase + k) * LZ_TW + i];
                float wk = w[k];
                acc.x = __builtin_fmaf(wk, h.x, acc.x);
                acc.y = __builtin_fmaf(wk, h.y, acc.y);
                acc.z = __builtin_fmaf(wk, h.z, acc.z);
                acc.w = __builtin_fmaf(wk, h.w, acc.w);
            }
            uint32_t o = pack_codes(acc.x, acc.y, acc.z, acc.w);
            gst<uint32_t>(dst.ptr + (size_t)oy * dst.pitch + (size_t)ox * 4, o);
        }
    }
    // the next tile's phase A writes `stile` only (every wave is past phase B) and is followed by a barrier before its
    // phase B overwrites `hrow`, which slower waves may still be reading here
    }   // tiles of the strip
}

// ---------------------------------------------------------------------------------------------------------------------
// lanczos3_strip2 — reductions of the 2:1 class (12 taps on both axes, every output advancing by exactly two source texels:
// 2160p -> 1080p, 1440p -> 720p, ...), ONE WAVE per strip of 64 output columns x `rows_per_wave` output rows, no block barrier.
//
// A lane owns an output column for BOTH passes: the horizontal results of its column for the 12 source rows a vertical
// filter needs are a window of 12 float4 in the lane's registers (a source row enters, the oldest leaves; indices are static
// in a loop unrolled over the window), so the float intermediate never goes through LDS and nothing has to wait for another
// wave.  Per source row: lanes 0..35 write the row's 144 staged texels (16-byte vectors prefetched four rows ahead into
// registers; CLAMP_TO_EDGE resolved here) to a two-row LDS ring, every lane reads its 12 taps as 6 or 7 aligned pairs
// (the tap offset 2 * lane + c has the parity of c for the whole launch: ODD) and runs the same fused chain as
// lanczos3_bgra; every second source row one output row is finished: the vertical chain over the window, packed store.
// The tile kernel above recomputes 12 halo rows per 16 output rows (2.75 horizontally filtered rows per output row) and
// synchronises four waves twice per tile; here it is 2 + 10 / rows_per_wave, and the SIMDs never idle at a barrier.
constexpr int LS_NV = 36;          // 16-byte vectors of one staged source row: 2 * 63 + 3 + 12 = 141 texels at most
#ifndef CHV_LS_PRE
#define CHV_LS_PRE 4
#endif
#ifndef CHV_LS_WAVES
#define CHV_LS_WAVES 4     // waves per SIMD the register allocation leaves room for
#endif
#ifndef CHV_LS_ABL
#define CHV_LS_ABL 0      // timing-only ablations (wrong pixels): 1 no staging at all, 2 no global loads, 4 taps from registers instead of LDS
#endif
constexpr int LS_PRE = CHV_LS_PRE;   // source rows in flight (registers) ahead of the row being filtered (a divisor of 12)

template <bool ODD>
__global__ __launch_bounds__(64, CHV_LS_WAVES) void lanczos3_strip2(DPlane dst, DPlane src, const int32_t *__restrict__ fx, const float *__restrict__ wx,
                                                         const int32_t *__restrict__ fy, const float *__restrict__ wy, int rows_per_wave,
                                                         int strips, int chunks, int total, const DPlane *__restrict__ batch) {
    // XCD-aware numbering (as in the strip kernels of the composite): block b runs on XCD b % 8, and every XCD gets one contiguous
    // range of (image, row chunk, strip) — neighbouring strips and chunks share their halo columns and rows through that XCD's L2
    // instead of fetching them once per XCD (the staged rows start 32 bytes before a 512-byte boundary: with strips dealt round-robin
    // FETCH_SIZE was 1.67x the source bytes)
    const int b = blockIdx.x, per_xcd = (total + 7) >> 3;
    const int idx = (b & 7) * per_xcd + (b >> 3);
    if ((b >> 3) >= per_xcd || idx >= total) return;
    const int image = idx / (strips * chunks), rem = idx - image * (strips * chunks);
    const int chunk = rem / strips, strip = rem - chunk * strips;
    if (batch) { dst = batch[2 * image]; src = batch[2 * image + 1]; }
    __shared__ __attribute__((aligned(16))) uint4 ring[2][LS_NV];
    const int lane = threadIdx.x;
    const int ox0 = strip * 64, j0 = chunk * rows_per_wave;
    if (ox0 >= dst.w || j0 >= dst.h) return;
    const int nrows = min(rows_per_wave, dst.h - j0);
    const int x = ox0 + lane, xe = min(x, dst.w - 1);
    const int col0 = __builtin_amdgcn_readfirstlane(fx[ox0]);
    const int col0a = col0 & ~3;                                  // (rounds towards -inf: the staged row starts on a 16-byte vector)
    const int cb = fx[xe] - col0a;                                // tap 0 of this lane, in texels from the start of the staged row
    float wr[12];
#pragma unroll
    for (int k = 0; k < 12; k++) wr[k] = wx[(size_t)xe * 12 + k];
    const int row0 = __builtin_amdgcn_readfirstlane(fy[j0]);
    const bool edge = col0a < 0 || col0a + 4 * LS_NV > src.w;     // (uniform) some staged vector sticks out of the picture
    const int S = 2 * nrows + 10;                                 // source rows this strip filters
    const int vc = col0a + 4 * lane, vcc = min(max(vc, 0), src.w - 4);
    const bool loader = lane < LS_NV;
    // The row loads are issued and awaited by hand.  gfx950 counts loads and stores in ONE counter (vmcnt) and completes them out of
    // order with respect to each other, so with the output stores in the loop hipcc has to drain the counter (`s_waitcnt vmcnt(0)`)
    // before every use of a prefetched row: the prefetch depth collapses to nothing and the kernel runs at VALU time PLUS memory time
    // (0.49 ms per 24 images, 0.34 without the loads).  Loads complete in order among themselves: once at most LS_PRE - 1 operations
    // are outstanding, the oldest of LS_PRE loads has arrived whatever the stores did — that is the wait below.
    auto issue = [&](int s) {
        const int sy = min(max(row0 + s, 0), src.h - 1);
        const uint8_t *p = src.ptr + (size_t)sy * src.pitch + (size_t)vcc * 4;
        chv_u32x4 v;
        asm volatile("global_load_dwordx4 %0, %1, off" : "=&v"(v) : "v"(p) : "memory");
        return v;
    };
    auto arrived = [&](chv_u32x4 &v) {            // (the "+v" ties the wait to the row's registers: no use can move above it)
        asm volatile("s_waitcnt vmcnt(%1)" : "+v"(v) : "n"(LS_PRE - 1) : "memory");
    };
    auto fix = [&](chv_u32x4 L) {                                  // texel k of the vector = texel clamp(vc + k) of the row
        auto pick = [&](int k) {
            const int idx = min(max(vc + k, 0), src.w - 1) - vcc;
            return idx == 0 ? L.x : idx == 1 ? L.y : idx == 2 ? L.z : L.w;
        };
        chv_u32x4 r = { pick(0), pick(1), pick(2), pick(3) };
        return r;
    };
    chv_u32x4 pre[LS_PRE];
#pragma unroll
    for (int p = 0; p < LS_PRE; p++) pre[p] = chv_u32x4{ 0u, 0u, 0u, 0u };
    if (loader) {
#pragma unroll
        for (int p = 0; p < LS_PRE; p++) pre[p] = issue(p);
    }
    float4 h[12];
#pragma unroll
    for (int t = 0; t < 12; t++) h[t] = make_float4(0.f, 0.f, 0.f, 0.f);
    const uint32_t taps_off = (uint32_t)(cb >> 1) * 8u;
    for (int g = 0; 12 * g < S; g++) {
#pragma unroll
        for (int t = 0; t < 12; t++) {
            const int s = 12 * g + t;
            if (s >= S) break;                                    // (uniform)
            // stage source row s, request row s + LS_PRE
            if (loader && !(CHV_LS_ABL & 1)) {
                if (!(CHV_LS_ABL & 2)) arrived(pre[t % LS_PRE]);
                const chv_u32x4 v = edge ? fix(pre[t % LS_PRE]) : pre[t % LS_PRE];
                *(chv_u32x4 *)&ring[t & 1][lane] = v;
                if (!(CHV_LS_ABL & 2)) pre[t % LS_PRE] = issue(s + LS_PRE);
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront", "local");
            __builtin_amdgcn_wave_barrier();
            // horizontal pass of this lane's column
            const uint2 *q = (const uint2 *)((const uint8_t *)ring[t & 1] + taps_off);
            uint32_t e[14];
#pragma unroll
            for (int i = 0; i < 6 + (ODD ? 1 : 0); i++) {
                if (CHV_LS_ABL & 4) { e[2 * i] = (uint32_t)(s * 77 + lane + i); e[2 * i + 1] = (uint32_t)(s * 31 + lane * 3 + i); continue; }
                const uint2 pr = q[i]; e[2 * i] = pr.x; e[2 * i + 1] = pr.y;
            }
            float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
            for (int k = 0; k < 12; k++) {
                const uint32_t p = e[k + (ODD ? 1 : 0)];
                acc.x = __builtin_fmaf(wr[k], (float)(p & 255), acc.x);
                acc.y = __builtin_fmaf(wr[k], (float)((p >> 8) & 255), acc.y);
                acc.z = __builtin_fmaf(wr[k], (float)((p >> 16) & 255), acc.z);
                acc.w = __builtin_fmaf(wr[k], (float)(p >> 24), acc.w);
            }
            h[t] = acc;
            if (t & 1) {
                // source rows up to s = 2 j + 11 are in the window: output row j, rows (t + 1 + k) % 12 of the window in tap order
                const int j = (s - 11) >> 1;
                if (j >= 0 && j < nrows) {                        // (uniform)
                    const float *wrow = wy + (size_t)(j0 + j) * 12;
                    float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
                    for (int k = 0; k < 12; k++) {
                        const float4 hk = h[(t + 1 + k) % 12];
                        const float wk = wrow[k];
                        o.x = __builtin_fmaf(wk, hk.x, o.x);
                        o.y = __builtin_fmaf(wk, hk.y, o.y);
                        o.z = __builtin_fmaf(wk, hk.z, o.z);
                        o.w = __builtin_fmaf(wk, hk.w, o.w);
                    }
                    if (x < dst.w) gst<uint32_t>(dst.ptr + (size_t)(j0 + j) * dst.pitch + (size_t)x * 4, pack_codes(o.x, o.y, o.z, o.w));
                }
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");              // (the rows requested past the strip's last one: nothing leaves in flight)
}

// ---------------------------------------------------------------------------------------------------------------------
// lanczos3_strip<T> — any ratio whose two axes have the same tap count T <= 22 and whose staged source row fits 64 vectors
// (reductions up to 3.5:1, every enlargement): the structure of lanczos3_strip2 — one wave per strip, a lane owns an output
// column for both passes, the horizontal results of the last T source rows in a window of T float4 in the lane's REGISTERS —
// for ratios where the number of new source rows per output row varies (3:2 alternates one and two; enlargements finish
// several output rows per source row).  The loop runs over SOURCE rows, T per trip, so that window indices and prefetch slots
// are static in every copy of the body (a slot or a window row chosen at run time would make the compiler copy registers at
// the joins, and a prefetch register whose load is in flight must not be copied: tools/check_inflight.py); after each source
// row a uniform test finishes the output rows whose last source row it was — their window starts at row (t + 1) % T.
// Same chains, same bytes as lanczos3_bgra.
template <typename F, int... Is>
CHV_DEV bool lg_all_of(F &&f, std::integer_sequence<int, Is...>) { return (... && f(std::integral_constant<int, Is>{})); }
template <int T> struct LgPre { static constexpr int value = T % 4 == 0 ? 4 : T % 3 == 0 ? 3 : 2; };      // prefetch depth: a divisor of T

template <int T>
__global__ __launch_bounds__(64, (T >= 14 ? 3 : 4)) void lanczos3_strip(DPlane dst, DPlane src, const int32_t *__restrict__ fx, const float *__restrict__ wx,
                                                                        const int32_t *__restrict__ fy, const float *__restrict__ wy, int rows_per_wave,
                                                                        int strips, int chunks, int total, int nv, const DPlane *__restrict__ batch) {
    constexpr int PRE = LgPre<T>::value;
    const int b = blockIdx.x, per_xcd = (total + 7) >> 3;         // XCD-aware numbering, as in lanczos3_strip2
    const int idx = (b & 7) * per_xcd + (b >> 3);
    if ((b >> 3) >= per_xcd || idx >= total) return;
    const int image = idx / (strips * chunks), rem = idx - image * (strips * chunks);
    const int chunk = rem / strips, strip = rem - chunk * strips;
    if (batch) { dst = batch[2 * image]; src = batch[2 * image + 1]; }
    extern __shared__ __attribute__((aligned(16))) uint8_t lsm[];
    chv_u32x4 *stage = (chv_u32x4 *)lsm;                          // [2][nv]
    const int lane = threadIdx.x;
    const int ox0 = strip * 64, j0 = chunk * rows_per_wave;
    if (ox0 >= dst.w || j0 >= dst.h) return;
    const int nrows = min(rows_per_wave, dst.h - j0);
    const int x = ox0 + lane, xe = min(x, dst.w - 1);
    const int col0 = __builtin_amdgcn_readfirstlane(fx[ox0]);
    const int col0a = col0 & ~3;
    const int cb = fx[xe] - col0a;
    float wr[T];
#pragma unroll
    for (int k = 0; k < T; k++) wr[k] = wx[(size_t)xe * T + k];
    const int row0 = __builtin_amdgcn_readfirstlane(fy[j0]);
    const bool edge = col0a < 0 || col0a + 4 * nv > src.w;
    const int vc = col0a + 4 * lane, vcc = min(max(vc, 0), src.w - 4);
    const bool loader = lane < nv;
    auto fix = [&](chv_u32x4 L) {
        auto pick = [&](int k) {
            const int i2 = min(max(vc + k, 0), src.w - 1) - vcc;
            return i2 == 0 ? L.x : i2 == 1 ? L.y : i2 == 2 ? L.z : L.w;
        };
        chv_u32x4 r = { pick(0), pick(1), pick(2), pick(3) };
        return r;
    };
    // (the load writes its destination registers some time AFTER the statement: the asm names the prefetch slot itself as its output,
    // with no temporary in between that the compiler could copy from before the data has landed)
#define LG_ISSUE(SLOT, S) do { const int sy_ = min(max(row0 + (S), 0), src.h - 1); \
                               const uint8_t *p_ = src.ptr + (size_t)sy_ * src.pitch + (size_t)vcc * 4; \
                               asm volatile("global_load_dwordx4 %0, %1, off" : "=&v"(SLOT) : "v"(p_) : "memory"); } while (0)
    chv_u32x4 pre[PRE];
#pragma unroll
    for (int p = 0; p < PRE; p++) pre[p] = chv_u32x4{ 0u, 0u, 0u, 0u };
    if (loader) {
#pragma unroll
        for (int p = 0; p < PRE; p++) LG_ISSUE(pre[p], p);
    }
    float4 h[T];
#pragma unroll
    for (int t = 0; t < T; t++) h[t] = make_float4(0.f, 0.f, 0.f, 0.f);
    const uint32_t tap0 = (uint32_t)cb * 4u;
    const int S = __builtin_amdgcn_readfirstlane(fy[j0 + nrows - 1]) - row0 + T;       // source rows this strip filters
    int jcur = 0, fcur = 0;                                       // next output row to finish, its first source row (fy[j0] - row0 = 0)
    for (int g = 0; g * T < S; g++) {
        // (expanded through a fold expression, not `#pragma unroll`: with the early exit and the inner loop the unroller gives up from
        // 10 taps on, and a window indexed at run time lives in scratch memory)
        auto body = [&](auto tc) -> bool {
            constexpr int t = decltype(tc)::value;
            const int s = g * T + t;
            if (s >= S) return false;                             // (uniform)
            if (loader) {
                asm volatile("s_waitcnt vmcnt(%1)" : "+v"(pre[t % PRE]) : "n"(PRE - 1) : "memory");
                stage[(t & 1) * nv + lane] = edge ? fix(pre[t % PRE]) : pre[t % PRE];
                LG_ISSUE(pre[t % PRE], s + PRE);
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront", "local");
            __builtin_amdgcn_wave_barrier();
            const uint32_t *row = (const uint32_t *)((const uint8_t *)(stage + (t & 1) * nv) + tap0);
            uint32_t e[T];
#pragma unroll
            for (int k = 0; k < T; k++) e[k] = row[k];
            float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
            for (int k = 0; k < T; k++) {
                acc.x = __builtin_fmaf(wr[k], (float)(e[k] & 255), acc.x);
                acc.y = __builtin_fmaf(wr[k], (float)((e[k] >> 8) & 255), acc.y);
                acc.z = __builtin_fmaf(wr[k], (float)((e[k] >> 16) & 255), acc.z);
                acc.w = __builtin_fmaf(wr[k], (float)(e[k] >> 24), acc.w);
            }
            h[t] = acc;
            // output rows whose last source row this was: source rows s - T + 1 .. s are window rows (t + 1) % T, (t + 2) % T, ...
            while (jcur < nrows && fcur + T - 1 == s) {           // (uniform; at most once per source row when reducing)
                const float *wrow = wy + (size_t)(j0 + jcur) * T;
                float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
                for (int k = 0; k < T; k++) {
                    const float4 hk = h[(t + 1 + k) % T];
                    const float wk = wrow[k];
                    o.x = __builtin_fmaf(wk, hk.x, o.x);
                    o.y = __builtin_fmaf(wk, hk.y, o.y);
                    o.z = __builtin_fmaf(wk, hk.z, o.z);
                    o.w = __builtin_fmaf(wk, hk.w, o.w);
                }
                if (x < dst.w) gst<uint32_t>(dst.ptr + (size_t)(j0 + jcur) * dst.pitch + (size_t)x * 4, pack_codes(o.x, o.y, o.z, o.w));
                jcur++;
                if (jcur < nrows) fcur = __builtin_amdgcn_readfirstlane(fy[j0 + jcur]) - row0;
            }
            return true;
        };
        lg_all_of(body, std::make_integer_sequence<int, T>{});
    }
#pragma unroll
    for (int p = 0; p < PRE; p++) asm volatile("s_waitcnt vmcnt(0)" : "+v"(pre[p]) :: "memory");
#undef LG_ISSUE
}

hipError_t launch_lanczos(const DPlane &dst, const DPlane &src, const int32_t *fx, const float *wx,
                          int tx, const int32_t *fy, const float *wy, int ty, hipStream_t stream,
                          const DPlane *batch, int n_batch, int stride_x, int first_x, int stride_y) {
    // uniform tap stride 2 on both axes, 12 taps: the wave-per-strip kernel (CHV_LZ_STRIP=0 builds keep the tile kernel: A/B)
    if (CHV_LZ_STRIP && tx == 12 && ty == 12 && stride_x == 2 && stride_y == 2 && src.w >= 4 * LS_NV) {
        const int strips = (dst.w + 63) / 64;
        const long want = 4L * 1024 * 4;                       // waves: four rounds of four per SIMD
        long r = ((long)dst.h * strips * (batch ? n_batch : 1) + want - 1) / want;
        // (every chunk re-filters 10 warm-up rows: 24-row chunks as long as they still give every SIMD four waves, short ones for a lone
        // picture, whose wall time is a wave's serial chain — 2160p -> 1080p alone 37.5 -> 29.6 us)
        const long waves24 = (long)((dst.h + 23) / 24) * strips * (batch ? n_batch : 1);
        const int rows = (int)std::min<long>(std::max<long>(r, waves24 >= 4096 ? 24 : CHV_LZ_MIN_ROWS), 180);
        const int chunks = (dst.h + rows - 1) / rows, total = strips * chunks * (batch ? n_batch : 1);
        dim3 grid((unsigned)(((total + 7) / 8) * 8));
        if (first_x & 1) hipLaunchKernelGGL(lanczos3_strip2<true>, grid, dim3(64), 0, stream, dst, src, fx, wx, fy, wy, rows, strips, chunks, total, batch);
        else hipLaunchKernelGGL(lanczos3_strip2<false>, grid, dim3(64), 0, stream, dst, src, fx, wx, fy, wy, rows, strips, chunks, total, batch);
        return hipGetLastError();
    }
    // rows / columns of source one tile can need: first[] advances by at most ceil(scale) per output
    const double sy = (double)src.h / (double)dst.h, sxs = (double)src.w / (double)dst.w;
    // equal tap counts on both axes, a staged row of at most 64 vectors: the general wave-per-strip kernel
    if (CHV_LZ_STRIP_ANY && tx == ty && tx <= 22 && (tx & 1) == 0 && src.w >= 4) {
        const int nv = ((int)(63 * sxs) + 1 + 3 + tx + 3) / 4 + 1;          // fx[x + 63] - fx[x] <= floor(63 scale) + 1; 3 texels of alignment
        if (nv <= 64) {
            const int strips = (dst.w + 63) / 64;
            // rows per wave: enough waves for three rounds of four per SIMD when the launch is large; a small launch (one picture) gets
            // short chunks instead — every chunk re-filters tx - 2 warm-up rows, but a wave's serial chain is what a lone resize waits for
            const long want = 4L * 1024 * 3;
            long r = ((long)dst.h * strips * (batch ? n_batch : 1) + want - 1) / want;
            const int rows = (int)std::min<long>(std::max<long>(r, CHV_LZ_MIN_ROWS), 256);
            const int chunks = (dst.h + rows - 1) / rows, total = strips * chunks * (batch ? n_batch : 1);
            dim3 grid((unsigned)(((total + 7) / 8) * 8));
            const size_t lds = (size_t)2 * nv * 16;                       // the two-row staging ring; the window is in registers
#define CHV_LZ_GO(TT) hipLaunchKernelGGL(lanczos3_strip<TT>, grid, dim3(64), lds, stream, dst, src, fx, wx, fy, wy, rows, strips, chunks, total, nv, batch)
            switch (tx) {
            case 6: CHV_LZ_GO(6); break;   case 8: CHV_LZ_GO(8); break;   case 10: CHV_LZ_GO(10); break; case 12: CHV_LZ_GO(12); break;
            case 14: CHV_LZ_GO(14); break; case 16: CHV_LZ_GO(16); break; case 18: CHV_LZ_GO(18); break; case 20: CHV_LZ_GO(20); break;
            default: CHV_LZ_GO(22); break;
            }
#undef CHV_LZ_GO
            return hipGetLastError();
        }
    }
    auto dims = [&](int tw, int th, int *max_rows, int *max_cols) -> size_t {
        *max_rows = (int)((th - 1) * sy + 2) + ty;
        *max_cols = ((int)((tw - 1) * sxs + 2) + tx + 3) & ~3;
        return (size_t)*max_rows * tw * sizeof(float4) + (size_t)*max_rows * *max_cols * sizeof(uint32_t);
    };
    int max_rows, max_cols;
    size_t lds = dims(32, 16, &max_rows, &max_cols);
    const bool small_tiles = lds > 64 * 1024;                 // reduction factors beyond about 4
    if (small_tiles) lds = dims(8, 4, &max_rows, &max_cols);
    if (lds > 160 * 1024) return hipErrorInvalidValue;        // beyond about 24:1 on one axis, about 17:1 on both
    const int tw = small_tiles ? 8 : 32, th = small_tiles ? 4 : 16;
    // Tiles per block: a block lives for `ks` tile times and the chip holds 256 CUs x floor(160 KB / lds) blocks at once,
    // so the launch takes about ceil(blocks / resident) * ks tile times; pick the ks in 2..8 that minimises it (ties: the
    // longer strip, fewer cold starts of the prefetch).  2160p -> 1080p: 4 (1020 blocks on 1024 slots).
    const int tiles_x = (dst.w + tw - 1) / tw, tiles_y = (dst.h + th - 1) / th;
    const long resident = 256L * std::max<long>(1, (long)(160 * 1024 / lds));
    int ks = 2; long best = -1;
    for (int k = 2; k <= 8; k++) {
        const long blocks = (long)((tiles_x + k - 1) / k) * tiles_y * (batch ? n_batch : 1);
        const long cost = ((blocks + resident - 1) / resident) * k;
        if (best < 0 || cost <= best) { best = cost; ks = k; }
    }
    dim3 grid((tiles_x + ks - 1) / ks, tiles_y, batch ? n_batch : 1);
    // register prefetch of the next tile's rectangle: one vector per thread and 8-row group
    const bool prefetch = !small_tiles && max_rows <= 8 * LZ_NPRE && max_cols / 4 <= 32 && src.w >= 4;
    auto launch = [&](auto kernel) -> hipError_t {
        if (lds > 64 * 1024) {
            hipError_t e = hipFuncSetAttribute((const void *)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            if (e != hipSuccess) return e;
        }
        hipLaunchKernelGGL(kernel, grid, dim3(256), lds, stream, dst, src, fx, wx, tx, fy, wy, ty, max_rows, max_cols, ks, batch);
        return hipGetLastError();
    };
    if (small_tiles) return tx <= LZ_MAXT ? launch(lanczos3_bgra<LZ_MAXT, false, false, 8, 4>) : launch(lanczos3_bgra<0, false, false, 8, 4>);
    if (tx == 12 && ty == 12)      // 2:1 (2160p -> 1080p) and its neighbourhood
        return prefetch ? launch(lanczos3_bgra<12, true, true, 32, 16>) : launch(lanczos3_bgra<12, true, false, 32, 16>);
    if (tx == 6 && ty == 6)        // enlargements
        return prefetch ? launch(lanczos3_bgra<6, true, true, 32, 16>) : launch(lanczos3_bgra<6, true, false, 32, 16>);
    if (tx <= 12) return prefetch ? launch(lanczos3_bgra<12, false, true, 32, 16>) : launch(lanczos3_bgra<12, false, false, 32, 16>);
    if (tx <= LZ_MAXT) return prefetch ? launch(lanczos3_bgra<LZ_MAXT, false, true, 32, 16>) : launch(lanczos3_bgra<LZ_MAXT, false, false, 32, 16>);
    return launch(lanczos3_bgra<0, false, false, 32, 16>);
}

#define CHV_LSTR2(x) #x
#define CHV_LSTR(x) CHV_LSTR2(x)
// what this translation unit was built with (chv_build_flags; a timing-only CHV_LS_ABL build must never ship)
const char *lanczos_build_flags() { return "lanczos3:abl=" CHV_LSTR(CHV_LS_ABL); }

}  // namespace chv

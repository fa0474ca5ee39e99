// kernels_stream.hip.cpp — tick_bgra_stream: ticks of 1..4 full-frame NV12 (or, all of them, y420p) video layers of ONE geometry onto a cleared BGRA canvas
// (the bench headline's tick, cfg2's tick; VideoMixer.mix of N camera feeds scaled to the canvas, mix.video.swift:114-124), rows
// outermost and the layers innermost.
//
// tick_bgra_wave (kernels_wave.hip.cpp) stages one layer's rectangle for a 64 x 16 strip (6.7 KB), runs the rows of that layer,
// stages the next one: the eight weight products of a pixel row, its row-table reads and the canvas pixel's trip through packed
// codes are paid once per row AND layer, because four rectangles do not fit beside each other at five waves per SIMD
// (profiles/r03_notes.md section 7).  Here nothing is staged as a rectangle.  A wave owns a 64-column strip of the canvas over a
// chunk of rows and streams every layer's source rows through a small LDS ring per plane — 8 luma rows and 4 chroma rows of 128
// bytes, 1.5 KB per layer — filled four (two) rows at a time by `global_load_lds_dwordx4`: a lane's 16 bytes go straight from its own
// global address to LDS at M0 + lane * 16 (tools/probe_lds_dma.cpp), no registers, no LDS write instruction.  With every layer's
// rows resident the loop runs over canvas rows: row entry, weights, tap addresses once, then per layer 12 taps, the integer
// matrix and one blend on float codes, one pack and one store at the end.  Same operations per pixel and layer as
// apply_layer_bgra / tick_bgra_wave (the blend result of a layer is rounded to codes exactly as its store would round it), so the
// same bytes.
//
// Eligibility is decided on the host (launch_bgra_stream_eligible): cleared BGRA canvas, every tick the same number (<= 4) of NV12
// layers, layers 1.. flagged LF_SAME_GEOM, axis-aligned bounded matrices without flips, no fill paint, opacities in [0, 1],
// horizontal reduction <= 1.7 (a strip's source bytes fit the ring's 128-byte rows), plane rows a multiple of 16 bytes.
#include "wave_common.hip.h"
#include "switches.h"

#include <algorithm>
#include <type_traits>
#include <cmath>

#pragma clang fp contract(off)

namespace chv {

#ifndef CHV_ST_ABL
#define CHV_ST_ABL 0        // timing-only (wrong pixels): 1 no ring fills, 2 no canvas stores, 4 no layer arithmetic (rings and stores only)
#endif
#ifndef CHV_STREAM_ABSORB
#define CHV_STREAM_ABSORB 1       // 0: the plain form of the colour matrix for every launch (A/B builds)
#endif
#ifndef CHV_STREAM_PMIX
#define CHV_STREAM_PMIX 1
#endif
#ifndef CHV_STREAM_ROWS_FIXED
#define CHV_STREAM_ROWS_FIXED 0
#endif
#ifndef CHV_STREAM_SMALL_WAVES
#define CHV_STREAM_SMALL_WAVES 4800       // waves a small launch is cut into (launch_bgra_stream)
#endif
#ifndef CHV_STREAM_SMALL_ROWS_MAX
#define CHV_STREAM_SMALL_ROWS_MAX 12
#endif
#ifndef CHV_STREAM_MIN_ROWS
#define CHV_STREAM_MIN_ROWS 4
#endif
#ifndef CHV_STREAM_ROUNDS
#define CHV_STREAM_ROUNDS 12      // chunk height: enough chunks for this many rounds of waves.  With one strip per block and untrimmed requests
                                  // short chunks won (neighbouring strips drift apart over a tall chunk and fetch shared lines twice: 240 rows 1.59 ms
                                  // and 1.75x the algorithmic bytes, 30 rows 1.34 ms); with four strips per block and trimmed requests: 3 / 6 / 12 /
                                  // 18 / 24 / 48 rounds (240 .. 16 rows) = 1.294 / 1.256 / 1.250 / 1.276 / 1.282 / 1.347 ms, traffic 1.00x at 60 rows
#endif
#ifndef CHV_STREAM_WAVES
#define CHV_STREAM_WAVES 6
#endif

constexpr float kRintBias = 8388608.0f;       // 2^23 (stream_body: the rounding between two layers)
constexpr int ST_PITCH = 128;                 // bytes per ring row (8 vectors)
constexpr int ST_YROWS = 8, ST_CROWS = 4;     // ring rows: luma (batches of 4), chroma (batches of 2)
// LDS layout (NL layers): luma [2 batch slots][NL layers][4 rows][128], then chroma [2 batch slots][NL layers][2 rows][128] — the layers of one
// batch slot lie next to each other so that ONE load instruction fills the rows of two layers (luma: 2 x 4 rows x 8 vectors = 64 lanes) or of
// all four (chroma: 4 x 2 x 8): a vector memory instruction occupies the CU's address unit for 16 cycles whatever it moves
// (tools/ubench_vmem.cpp), and with one instruction per layer and plane that unit was a fifth of the kernel's time.
constexpr int ST_YL = 4 * ST_PITCH, ST_CL = 2 * ST_PITCH;       // bytes of one layer inside a batch slot: 512, 256
// Planar sources (y420p: what FFmpeg's software decoders emit, dec.video.ffmpeg.swift:187-221): the chroma region holds U rows and, behind
// them, V rows — each [2 batch slots][NL layers][2 rows][96] (a strip's chroma texels at up to 1.7 : 1 fit six vectors) — filled by one load
// instruction per plane kind (NL x 2 rows x 6 vectors = 48 lanes at four layers).
constexpr int ST_CPP = 96, ST_CLP = 2 * ST_CPP;                 // planar chroma: bytes per ring row, bytes of one layer inside a batch slot
template <int NL, bool PL> constexpr int st_layer_bytes() { return ST_PITCH * ST_YROWS + (PL ? 2 * 2 * ST_CLP : ST_PITCH * ST_CROWS); }       // per layer: 1536 / 1792
constexpr int ST_TAB = 32;                    // row entries computed at a time (lane = row)
#ifndef CHV_STREAM_BLOCK
#define CHV_STREAM_BLOCK 4
#endif
constexpr int ST_WAVES = CHV_STREAM_BLOCK;    // waves (neighbouring strips) per block

// integer colour matrix on biased codes -> float codes: yuv_to_bgr_floats (pixel_math.hip.h)

// One load instruction: lane -> (layer li, row rr of the batch, vector vec); `p` is the lane's own 16-byte source address, its LDS
// destination is m0 + lane * 16 (tools/probe_lds_dma.cpp).
// (M0 is a reserved register to hipcc: it cannot be named as a clobber — "may lead to undefined behaviour" — so the statement sets it itself
// every time, and tests/test_device_code_contract.py checks on the built code that nothing else in this kernel reads or writes M0.)
CHV_DEV void st_dma(const uint8_t *p, bool active, uint32_t m0) {
    if (active && !(CHV_ST_ABL & 1))
        asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off" :: "s"(m0), "v"(p) : "memory");
}

// A row's store: the row's address stays on the scalar unit (gst_at asks for the same, and here hipcc adds the lane's offset to it with a
// 64-bit vector add per row instead).  The stores share the counter the ring's loads are awaited by; they only ever make a wait longer.
CHV_DEV void st_store(uint8_t *row, uint32_t off, uint32_t v) {
    asm volatile("global_store_dword %0, %1, %2" :: "v"(off), "v"(v), "s"(row) : "memory");
}

// ONE: a launch of one tick whose descriptors are kernel ARGUMENTS (tick_bgra_stream_one below) — `ticks` / `layers` point into the kernarg
// segment, every field is a scalar load at a constant offset from one base, issued together: no tick -> first_layer -> layer chain of
// dependent loads in front of a lone tick's waves, and no descriptor copy in front of the launch.
// ABS: every layer's colour matrix has absorbing biases (csc_fold_absorbed, pixel_math.hip.h; launch_bgra_stream decides)
template <int NL, bool ONE, bool PL, bool ABS>
CHV_DEV void stream_body(const DTick *__restrict__ ticks, const DLayer *__restrict__ layers, int n_ticks, int strips_x, int chunks_y, int rows_per_chunk) {
    // ST_WAVES independent waves per block, on neighbouring strips (no barrier anywhere): their source windows overlap by a vector or two,
    // and waves of one block start together on one CU — the shared lines are fetched once (HBM traffic 1.47x -> see profiles/r03_notes.md)
    extern __shared__ __attribute__((aligned(16))) uint8_t lds_all[];
    constexpr int LBYTES = st_layer_bytes<NL, PL>();
    constexpr int CPITCH = PL ? ST_CPP : ST_PITCH, CLB = PL ? ST_CLP : ST_CL;             // chroma ring row, one layer inside a chroma batch slot
    constexpr int VOFF = PL ? 2 * NL * ST_CLP : 1;                                       // from a U sample to its V sample
    constexpr int WBYTES = NL * LBYTES + ST_TAB * (int)(sizeof(uint4) + sizeof(uint32_t));       // (NV12, four layers: 6784 — six blocks of four waves per CU)
    const int wave = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
    uint8_t *lds = lds_all + wave * WBYTES;
    uint4 *rowtab = (uint4 *)(lds + NL * LBYTES);                 // [ST_TAB] what vector instructions use: {luma row weight, chroma row weight, 1 - luma, 1 - chroma}
    uint32_t *rowpos = (uint32_t *)(rowtab + ST_TAB);             // [ST_TAB] what the scalar unit uses: luma tap row (16 bits) | chroma tap row (15 bits) << 16 | row inside the picture << 31
    const uint32_t lds0 = (uint32_t)(size_t)lds;                  // LDS byte address of the rings (the DMA's M0)
    const int lane = threadIdx.x & 63;
    // XCD-aware numbering: block b runs on XCD b % 8; an XCD owns a contiguous range of (tick, chunk, group of ST_WAVES strips)
    const int groups_x = (strips_x + ST_WAVES - 1) / ST_WAVES;
    const int total = n_ticks * chunks_y * groups_x;
    const int b = blockIdx.x, per_xcd = (total + 7) >> 3;
    const int idx = (b & 7) * per_xcd + (b >> 3);
    if ((b >> 3) >= per_xcd || idx >= total) return;
    const int tick = idx / (chunks_y * groups_x), rem = idx - tick * (chunks_y * groups_x);
    const int chunk = rem / groups_x, strip = (rem - chunk * groups_x) * ST_WAVES + wave;
    if (strip >= strips_x) return;
    const DTick &T = ticks[ONE ? 0 : tick];
    const DLayer *L = layers + (ONE ? 0 : T.first_layer);
    const int x0 = strip * 64, y0 = chunk * rows_per_chunk;
    if (x0 >= T.W || y0 >= T.H) return;
    const int nrows = min(rows_per_chunk, T.H - y0);
    const DPlane D = T.dst.pl[0];
    const float sx = (float)T.W, sy = (float)T.H;
    const float *U = L[0].u;                                       // (layers 1.. : LF_SAME_GEOM — the same 48 geometry inputs, plane shapes, bounding box)
    const DPlane SY = L[0].src.pl[0], SC = L[0].src.pl[1];

    // ---- this lane's column: tap positions and weights (WaveStrip::setup, wave_common.hip.h) ---------------------------------
    const int x = x0 + lane, xe = min(x, T.W - 1);
    const float nx = ((float)xe / sx) * 2.f - 1.f;
    const float t3 = U[U_TRANSFORM + 15];
    const float t0 = nx * U[U_TRANSFORM + 0] + U[U_TRANSFORM + 3];
    const float b0 = nx * U[U_BORDER + 0] + U[U_BORDER + 3];
    const float u = t0 * U[U_TEXTURE + 0] + t3 * U[U_TEXTURE + 3];
    const int cfl = ((b0 >= 0.f && b0 <= 1.f) ? AX_BORDER : 0) | ((t0 >= 0.f && t0 <= 1.f) ? AX_TX : 0) | ((u >= 0.f && u <= 1.f) ? AX_UV : 0);
    int cy, cc;
    float cya, cca;
    lin_axis_raw(u, SY.w, cy, cya); lin_axis_raw(u, SC.w, cc, cca);
    // first source column of the strip (no flips: lane 0 has the smallest positions), as the start of the staged 128 bytes
    const int cy_first = __builtin_amdgcn_readfirstlane(cy), cc_first = __builtin_amdgcn_readfirstlane(cc);
    const int ycol0 = min(max(cy_first, 0), SY.w - 1) & ~15;
    constexpr int BPC = PL ? 1 : 2;                                                     // bytes per chroma texel of the ring: a U (V) byte / a (U, V) pair
    const int ccol0 = (min(max(cc_first, 0), SC.w - 1) * BPC) & ~15;
    // byte offsets of the two tap columns inside a ring row, CLAMP_TO_EDGE in x; lanes past the staged bytes (columns outside the canvas
    // or outside the picture, never stored / never taken) read whatever is there
    const int oy0 = min(max(min(max(cy, 0), SY.w - 1) - ycol0, 0), ST_PITCH - 1), oy1 = min(max(min(max(cy + 1, 0), SY.w - 1) - ycol0, 0), ST_PITCH - 1);
    const int oc0 = min(max(min(max(cc, 0), SC.w - 1) * BPC - ccol0, 0), CPITCH - BPC), oc1 = min(max(min(max(cc + 1, 0), SC.w - 1) * BPC - ccol0, 0), CPITCH - BPC);
    // (the chroma offsets are even, and a compiler that knows it fuses a pair's U and V byte reads into one 16-bit read and splits it
    // again with two more vector instructions per tap pair: every tap its own byte read is the cheaper form here, r03_notes.md section 2)
    int oc0v = oc0, oc1v = oc1;
    asm("" : "+v"(oc0v), "+v"(oc1v));
    const float iya = (1.0f - cya) * kTapScale, ya = cya * kTapScale, ica = (1.0f - cca) * kTapScale, ca = cca * kTapScale;
    const bool lane_pic = cfl == AX_ALL && x < T.W;
    // 16-byte vectors of a ring row that some tap of the strip can read (the last lane has the largest offsets): the rest is not requested
    const int nvecY = (__builtin_amdgcn_readlane(oy1, 63) >> 4) + 1, nvecC = ((__builtin_amdgcn_readlane(oc1, 63) + BPC - 1) >> 4) + 1;

    // per-layer constants (wave-uniform), read once: the asm statements below clobber "memory", and every descriptor read after one of
    // them would be a fresh scalar load with its latency in the middle of the ring logic
    std::conditional_t<ABS, CscAbsorbed, CscFolded> csc[NL];
    float al[NL], ial[NL], nrb[NL], al24[NL];
    const uint8_t *planeY[NL], *planeC[NL], *planeV[NL];
#pragma unroll
    for (int l = 0; l < NL; l++) {
        if constexpr (ABS) csc[l] = csc_fold_absorbed(L[l].csc);
        else csc[l] = csc_fold_biased(kCsc[L[l].csc & 3]);
        al[l] = 1.0f * L[l].u[U_OPACITY]; ial[l] = 1.f - al[l];
        // Between two layers the canvas pixel is rounded to codes (what the first layer's store would have kept) and multiplied by the next
        // layer's 1 - opacity.  The rounding is one add of 2^23 (x + 2^23 = 2^23 + rint(x) exactly, ties to even, for 0 <= x < 2^22); taking the
        // 2^23 out again and the multiply are ONE fused multiply-add: fma(x + 2^23, ia, -2^23 ia) rounds the exact (rint(x) + 2^23) ia - 2^23 ia
        // = rint(x) ia — the product the separate multiply rounds — because 2^23 ia is exact (a power of two times a float).  The constant
        // lives in a vector register: a VOP3 instruction reads one scalar operand, and that is the layer's 1 - opacity.
        al24[l] = al[l] * kTapScale;                     // (exact: a power of two; the layer's pixel arrives scaled by 2^-24, see the blend)
        nrb[l] = -kRintBias * ial[l];
        asm volatile("" : "+v"(nrb[l]));
        planeY[l] = L[l].src.pl[0].ptr; planeC[l] = L[l].src.pl[1].ptr; planeV[l] = L[l].src.pl[PL ? 2 : 1].ptr;
    }
    // this lane's place in a batch: (row, vector) and its byte column, luma and chroma
    // (planar chroma: a layer's two rows of six vectors are twelve lanes)
    const int liC = PL ? (lane * 43) >> 9 : lane >> 4, inC = PL ? lane - liC * 12 : lane & 15;      // lane -> layer, place inside the layer's rows
    static_assert(((63 * 43) >> 9) == 5 && ((47 * 43) >> 9) == 3 && ((48 * 43) >> 9) == 4 && ((11 * 43) >> 9) == 0 && ((12 * 43) >> 9) == 1, "lane / 12");
    const int rrY = (lane >> 3) & 3, rrC = PL ? (inC >= 6 ? 1 : 0) : (lane >> 3) & 1, vecC = PL ? inC - 6 * rrC : lane & 7;
    const int colY = min(ycol0 + 16 * (lane & 7), SY.w - 16), colC = min(ccol0 + 16 * vecC, SC.w * BPC - 16);

    // the chunk's last tap rows: nothing past them is requested (a chunk's overshoot is another wave's first rows: fetched twice)
    int lastY, lastC;
    {
        const int ye = min(y0 + nrows - 1, T.H - 1);
        const float ny = ((float)ye / sy) * 2.f - 1.f;
        const float t1 = ny * U[U_TRANSFORM + 5] + U[U_TRANSFORM + 7];
        const float v = t1 * U[U_TEXTURE + 5] + t3 * U[U_TEXTURE + 7];
        int ry, rc;
        float a_;
        lin_axis_raw(v, SY.h, ry, a_); lin_axis_raw(v, SC.h, rc, a_);
        lastY = __builtin_amdgcn_readfirstlane(ry) + 1; lastC = __builtin_amdgcn_readfirstlane(rc) + 1;
    }
    uint32_t pending = 0u;                                        // the previous row's pixel, stored after this row's wait
    uint32_t alpha_word = 0xFF000000u;                            // img_clear_bgra's pixel; the word the colour bytes are packed into
    asm volatile("" : "+v"(alpha_word));
    int issued = 0, seqY = 0, seqC = 0;                           // load instructions issued so far; the count right after the newest luma / chroma batch
    int nextY = 0, nextC = 0, landY = 0, landC = 0, baseY = 0, baseC = 0;      // ring state: rows below next* are requested, below land* have arrived
    for (int j = 0; j < nrows; j++) {
        // ---- row entries, ST_TAB at a time: lane = row (WaveStrip::setup) ------------------------------------------------
        if ((j & (ST_TAB - 1)) == 0) {
            wave_lds_fence();
            const int ye = min(y0 + j + min(lane, ST_TAB - 1), T.H - 1);
            const float ny = ((float)ye / sy) * 2.f - 1.f;
            const float t1 = ny * U[U_TRANSFORM + 5] + U[U_TRANSFORM + 7];
            const float b1 = ny * U[U_BORDER + 5] + U[U_BORDER + 7];
            const float v = t1 * U[U_TEXTURE + 5] + t3 * U[U_TEXTURE + 7];
            const int rfl = ((b1 >= 0.f && b1 <= 1.f) ? AX_BORDER : 0) | ((t1 >= 0.f && t1 <= 1.f) ? AX_TX : 0) | ((v >= 0.f && v <= 1.f) ? AX_UV : 0);
            int ry, rc;
            float rya, rca;
            lin_axis_raw(v, SY.h, ry, rya); lin_axis_raw(v, SC.h, rc, rca);
            if (lane < ST_TAB) {
                rowtab[lane] = make_uint4(__float_as_uint(rya), __float_as_uint(rca), __float_as_uint(1.0f - rya), __float_as_uint(1.0f - rca));
                // (rows of the picture: -1 .. plane rows - 1, which the host holds to 16 / 15 signed bits; rows outside it are not stored, and
                // clamped — still rising with the canvas row — they steer the rings as their unclamped values would)
                rowpos[lane] = ((uint32_t)min(max(ry, -32768), 32767) & 0xFFFFu) | (((uint32_t)min(max(rc, -16384), 16383) & 0x7FFFu) << 16) | (rfl == AX_ALL ? 0x80000000u : 0u);
            }
            wave_lds_fence();
        }
        const uint4 re = rowtab[j & (ST_TAB - 1)];
        const int rp = __builtin_amdgcn_readfirstlane((int)rowpos[j & (ST_TAB - 1)]);
        const int ry = (int)((uint32_t)rp << 16) >> 16, rc = (int)((uint32_t)rp << 1) >> 17;
        const bool row_pic = rp < 0;
        // (the row weights and their complements are used by vector instructions only: they stay the broadcast registers the LDS read
        // returned — no v_readfirstlane, and the two subtractions were done once per row by the table's lane)
        const float yb = __uint_as_float(re.x), cbw = __uint_as_float(re.y);
        // ---- residency: source rows ry, ry + 1 (luma) and rc, rc + 1 (chroma) of every layer ------------------------------
        // A ring holds two batches.  The next batch is REQUESTED as soon as the taps have left the older of the two (ry has entered the
        // newer one) and AWAITED only when a tap row reaches it — about two canvas rows later at a 1.5 : 1 reduction, time the other waves
        // of the SIMD fill; waiting right after the request put every wave to sleep for a memory latency every 1.3 rows (pipeline 1.57 ms).
        if (j == 0) {
            baseY = ry; baseC = rc; nextY = ry; nextC = rc; landY = ry; landC = rc;       // the rings start at the chunk's first tap rows
        }
        {
            // Requests are awaited by COUNT: loads complete in order among themselves, so once at most n vector-memory operations are
            // outstanding, where n is the number of loads issued after the batch a tap row needs, that batch has landed — whatever the
            // canvas stores in between did (they share the counter and complete out of order; they can only make the wait longer).
            // A full drain instead would also wait for the other plane's request of a row ago.
            auto await = [&](int younger) {                        // (uniform)
                if (younger <= 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                else if (younger == 1) asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
                else if (younger == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
                else asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
                wave_lds_fence();
            };
            const bool needY = ry + 1 >= landY, needC = rc + 1 >= landC;
            const bool wantY = ry + ST_YROWS / 2 >= nextY && nextY <= lastY, wantC = rc + ST_CROWS / 2 >= nextC && nextC <= lastC;     // (uniform)
            // (one test for the rows on which nothing happens — two in five at a 1.5 : 1 reduction: a row's nine uniform branches were a
            // tenth of its time)
            if (needY | needC | wantY | wantC) {
            if (needY) { await(issued - seqY); landY = nextY; }
            if (needC) { await(issued - seqC); landC = nextC; }
            if (wantY || wantC) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");    // (the taps of the rows being overwritten have been read)
            // (a while: the first row fills both batches; strong vertical reductions skip rows)
            while (ry + ST_YROWS / 2 >= nextY && nextY <= lastY) {
                // four luma rows of every layer, two layers per instruction; the planes of all layers have one shape (LF_SAME_GEOM)
                const uint32_t slot = (uint32_t)((((nextY - baseY) >> 2) & 1) * (NL * ST_YL));
                const int r = min(max(nextY + rrY, 0), SY.h - 1);
                const size_t roff = (size_t)r * SY.pitch + (size_t)colY;
#pragma unroll
                for (int l0 = 0; l0 < NL; l0 += 2) {
                    const uint8_t *pa = planeY[l0], *pb = planeY[l0 + 1 < NL ? l0 + 1 : l0];
                    st_dma((lane < 32 ? pa : pb) + roff, (lane < 32 || l0 + 1 < NL) && (lane & 7) < nvecY && nextY + rrY <= lastY, lds0 + slot + (uint32_t)(l0 * ST_YL));
                }
                nextY += 4; issued += (NL + 1) / 2; seqY = issued;
            }
            while (rc + ST_CROWS / 2 >= nextC && nextC <= lastC) {
                // two chroma rows of every layer in one instruction
                const uint32_t slot = (uint32_t)(2 * NL * ST_YL + (((nextC - baseC) >> 1) & 1) * (NL * CLB));
                const int r = min(max(nextC + rrC, 0), SC.h - 1);
                const size_t roff = (size_t)r * SC.pitch + (size_t)colC;
                const int li = liC;
                const bool act = li < NL && vecC < nvecC && nextC + rrC <= lastC;
                const uint8_t *pl = li == 0 ? planeC[0] : li == 1 ? planeC[NL > 1 ? 1 : 0] : li == 2 ? planeC[NL > 2 ? 2 : 0] : planeC[NL > 3 ? 3 : 0];
                st_dma(pl + roff, act, lds0 + slot);
                if constexpr (PL) {
                    // the V planes (the shape of the U planes, host-checked) into the region behind the U rows
                    const uint8_t *pv = li == 0 ? planeV[0] : li == 1 ? planeV[NL > 1 ? 1 : 0] : li == 2 ? planeV[NL > 2 ? 2 : 0] : planeV[NL > 3 ? 3 : 0];
                    st_dma(pv + roff, act, lds0 + slot + (uint32_t)(2 * NL * ST_CLP));
                    issued += 1;
                }
                nextC += 2; issued += 1; seqC = issued;
            }
            // (first row of a chunk, rows skipped by a strong reduction: what was just requested is needed now)
            if (ry + 1 >= landY || rc + 1 >= landC) { await(0); landY = nextY; landC = nextC; }
            }
        }
        // The previous row's pixels are stored HERE, right after this row's wait: gfx950 has one counter for loads and stores, the wait above
        // drains it, and a store issued just before it would be waited for every time (a write latency per wait); issued now it has
        // until the next wait, a row or two away.
        if (j > 0 && x < T.W && !(CHV_ST_ABL & 2)) st_store(D.ptr + (size_t)(y0 + j - 1) * D.pitch, (uint32_t)x * 4u, pending);
        // ---- tap addresses and weights, once for all layers -----------------------------------------------------------------
        auto yoff = [&](int q) { return ((q >> 2) & 1) * (NL * ST_YL) + (q & 3) * ST_PITCH; };                       // layer 0's copy of luma row baseY + q
        auto coff = [&](int q) { return 2 * NL * ST_YL + ((q >> 1) & 1) * (NL * CLB) + (q & 1) * CPITCH; };
        const int sY0 = yoff(ry - baseY), sY1 = yoff(ry + 1 - baseY), sC0 = coff(rc - baseC), sC1 = coff(rc + 1 - baseC);
        const uint8_t *pY00 = lds + (oy0 + sY0), *pY10 = lds + (oy1 + sY0), *pY01 = lds + (oy0 + sY1), *pY11 = lds + (oy1 + sY1);
        const uint8_t *pC00 = lds + (oc0v + sC0), *pC10 = lds + (oc1v + sC0), *pC01 = lds + (oc0v + sC1), *pC11 = lds + (oc1v + sC1);
        const float iyb = __uint_as_float(re.z), icb = __uint_as_float(re.w);
        const float w00 = iya * iyb, w10 = ya * iyb, w01 = iya * yb, w11 = ya * yb;
        const float c00 = ica * icb, c10 = ca * icb, c01 = ica * cbw, c11 = ca * cbw;
        float r0 = 0.f, r1 = 0.f, r2 = 0.f;                          // img_clear_bgra: (0, 0, 0, 1) — the canvas pixel as float codes
#pragma unroll
        for (int l = 0; l < ((CHV_ST_ABL & 4) ? 0 : NL); l++) {
            constexpr int dummy = 0; (void)dummy;
            const int lo = l * ST_YL, lc = l * CLB;
            const float fy = cs_mix_h(w00, w10, w01, w11, tap_h(pY00 + lo), tap_h(pY10 + lo), tap_h(pY01 + lo), tap_h(pY11 + lo));
            const float fu = cs_mix_h(c00, c10, c01, c11, tap_h(pC00 + lc), tap_h(pC10 + lc), tap_h(pC01 + lc), tap_h(pC11 + lc));
            const float fv = cs_mix_h(c00, c10, c01, c11, tap_h(pC00 + lc + VOFF), tap_h(pC10 + lc + VOFF), tap_h(pC01 + lc + VOFF), tap_h(pC11 + lc + VOFF));
            if constexpr (ABS && CHV_STREAM_PMIX) {
                // the layer's pixel enters the blend as a binary16 read from the high half of its clamped 16.16 sum (code x 2^-24, exact;
                // the opacity carries the 2^24): p * a + inner in one v_fma_mix_f32, the same real numbers into the same single rounding
                int32_t cb, cg, cr;
                yuv_to_bgr_fixed_absorbed(csc[l], fy, fu, fv, cb, cg, cr);
                const float a24 = al24[l];
                if (l == 0) {
                    r0 = __builtin_fmaf(a24, (float)code_h(cb), 0.0f); r1 = __builtin_fmaf(a24, (float)code_h(cg), 0.0f); r2 = __builtin_fmaf(a24, (float)code_h(cr), 0.0f);
                } else {
                    r0 = __builtin_fmaf(a24, (float)code_h(cb), __builtin_fmaf(r0, ial[l], nrb[l]));
                    r1 = __builtin_fmaf(a24, (float)code_h(cg), __builtin_fmaf(r1, ial[l], nrb[l]));
                    r2 = __builtin_fmaf(a24, (float)code_h(cr), __builtin_fmaf(r2, ial[l], nrb[l]));
                }
                if (l + 1 < NL) { r0 += kRintBias; r1 += kRintBias; r2 += kRintBias; }
                continue;
            }
            float pb, pg, pr;
            if constexpr (ABS) yuv_to_bgr_floats_absorbed(csc[l], fy, fu, fv, pb, pg, pr);
            else yuv_to_bgr_floats(csc[l], (int)code_biased(fy), (int)code_biased(fu), (int)code_biased(fv), pb, pg, pr);
            if (l == 0) {                // the cleared canvas: fma(p, a, 0 * (1 - a)) = RN(p * a) for a in [0, 1]
                r0 = pb * al[0]; r1 = pg * al[0]; r2 = pr * al[0];
            } else {                     // (r holds 2^23 + the codes the previous layer's store would have kept: see nrb above)
                r0 = __builtin_fmaf(pb, al[l], __builtin_fmaf(r0, ial[l], nrb[l]));
                r1 = __builtin_fmaf(pg, al[l], __builtin_fmaf(r1, ial[l], nrb[l]));
                r2 = __builtin_fmaf(pr, al[l], __builtin_fmaf(r2, ial[l], nrb[l]));
            }
            if (l + 1 < NL) { r0 += kRintBias; r1 += kRintBias; r2 += kRintBias; }
        }
        // (pack_codes with the alpha word as a separate source: tied to the destination it is re-materialised every row)
        uint32_t out;
        asm("v_cvt_pk_u8_f32 %0, %1, 0, %2" : "=v"(out) : "v"(r0), "v"(alpha_word));
        asm("v_cvt_pk_u8_f32 %0, %1, 1, %0" : "+v"(out) : "v"(r1));
        asm("v_cvt_pk_u8_f32 %0, %1, 2, %0" : "+v"(out) : "v"(r2));
        // a pixel outside the picture keeps the cleared canvas (inside the border quad its alpha is forced: the same word)
        const uint32_t res = (lane_pic && row_pic) ? out : alpha_word;
        pending = res;
    }
    if (nrows > 0 && x < T.W && !(CHV_ST_ABL & 2)) st_store(D.ptr + (size_t)(y0 + nrows - 1) * D.pitch, (uint32_t)x * 4u, pending);
}

template <int NL, bool PL, bool ABS>
__global__ __launch_bounds__(64 * ST_WAVES, CHV_STREAM_WAVES) void tick_bgra_stream(const DTick *__restrict__ ticks, const DLayer *__restrict__ layers, int n_ticks,
                                                                          int strips_x, int chunks_y, int rows_per_chunk) {
    stream_body<NL, false, PL, ABS>(ticks, layers, n_ticks, strips_x, chunks_y, rows_per_chunk);
}

// one tick, descriptors by value (96 + NL x 368 bytes of kernel arguments)
template <int NL>
struct StreamOne {
    DTick t;
    DLayer l[NL];
};
template <int NL, bool PL, bool ABS>
__global__ __launch_bounds__(64 * ST_WAVES, CHV_STREAM_WAVES) void tick_bgra_stream_one(const StreamOne<NL> a, int strips_x, int chunks_y, int rows_per_chunk) {
    stream_body<NL, true, PL, ABS>(&a.t, a.l, 1, strips_x, chunks_y, rows_per_chunk);
}

// ---------------------------------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------------------------------
static bool stream_plane_ok(const DPlane &p) {
    return (((uintptr_t)p.ptr) & 15) == 0 && (p.pitch & 15) == 0 && ((p.w * p.comps) & 15) == 0 && p.w * p.comps >= 16 && p.h >= 1;
}

// every tick: cleared canvas, the same number (1..4) of NV12 -> BGRA layers of one geometry, axis-aligned and bounded, no flips, no fill,
// opacities in [0, 1], a strip's source columns within one 128-byte ring row
#define CHV_STR2(x) #x
#define CHV_STR(x) CHV_STR2(x)
// what this translation unit was built with (chv_build_flags; a timing-only CHV_ST_ABL build must never ship)
const char *bgra_stream_build_flags() { return "tick_bgra_stream:abl=" CHV_STR(CHV_ST_ABL) ",strips_per_block=" CHV_STR(CHV_STREAM_BLOCK) ",rounds=" CHV_STR(CHV_STREAM_ROUNDS); }

bool bgra_stream_eligible(const DTick *ticks, const DLayer *layers, int n_ticks) {
    if (n_ticks < 1 || !switches().stream.load(std::memory_order_relaxed)) return false;
    const int nl = ticks[0].n_layers;
    if (nl < 1 || nl > 4) return false;
    for (int i = 0; i < n_ticks; i++) {
        const DTick &T = ticks[i];
        if (T.n_layers != nl || !T.clear_first) return false;
        if ((((uintptr_t)T.dst.pl[0].ptr) & 3) != 0 || (T.dst.pl[0].pitch & 3) != 0) return false;
        for (int l = 0; l < nl; l++) {
            const DLayer &Y = layers[T.first_layer + l];
            // (NV12 or planar sources, one class per launch: LF_SAME_GEOM already says so inside a tick)
            const int kind0 = layers[ticks[0].first_layer].kind;
            if ((kind0 != LK_BGRA_FROM_NV12 && kind0 != LK_BGRA_FROM_Y420P) || Y.kind != kind0) return false;
            const bool planar = kind0 == LK_BGRA_FROM_Y420P;
            const int need = LF_AXIS_ALIGNED | LF_BOUNDED | LF_NO_FILL | (l ? LF_SAME_GEOM : 0);
            if ((Y.flags & need) != need) return false;
            const float op = Y.u[U_OPACITY];
            if (!(op >= 0.f && op <= 1.f)) return false;
            if (!stream_plane_ok(Y.src.pl[0]) || !stream_plane_ok(Y.src.pl[1])) return false;
            if (Y.src.pl[0].h > 32767 || Y.src.pl[1].h > 16383) return false;          // (the row table packs tap rows into 16 / 15 signed bits)
            if (Y.src.pl[1].comps != (planar ? 1 : 2) || Y.src.pl[0].comps != 1) return false;
            if (planar) {
                const DPlane &cu = Y.src.pl[1], &cv = Y.src.pl[2];
                if (!stream_plane_ok(cv) || cv.comps != 1 || cv.w != cu.w || cv.h != cu.h || cv.pitch != cu.pitch) return false;       // one ring geometry for U and V
            }
            if (l == 0) {
                // source texels per canvas pixel: u = (x / W * 2 - 1) * T0 * X0 + ...  =>  du/dx * w = 2 T0 X0 w / W
                const double kx = 2.0 * (double)Y.u[U_TRANSFORM + 0] * (double)Y.u[U_TEXTURE + 0], ky = 2.0 * (double)Y.u[U_TRANSFORM + 5] * (double)Y.u[U_TEXTURE + 5];
                if (!(kx > 0.0) || !(ky > 0.0)) return false;                                   // flips: the rings assume rising positions
                const double sx = kx * Y.src.pl[0].w / (double)T.W;
                if (!(63.0 * sx + 2.0 + 15.0 + 1.0 <= 128.0)) return false;                     // luma bytes of a strip (NV12 chroma: the same count)
                if (planar && !(63.0 * (kx * Y.src.pl[1].w / (double)T.W) + 2.0 + 15.0 + 1.0 <= (double)ST_CPP)) return false;     // a plane's chroma bytes of a strip
                // source rows per canvas row: the rings are advanced batch by batch (four luma rows per load), so the work per canvas row grows
                // with the vertical reduction — a picture squeezed into a few canvas rows (a zoom animation's first frames) would issue
                // hundreds of loads of rows nobody taps per canvas row, and tick_bgra_wave culls by bounding box instead
                const double sy = ky * Y.src.pl[0].h / (double)T.H;
                if (!std::isfinite(sy) || sy > 4.0) return false;
            }
        }
    }
    return true;
}

// ticks == nullptr: one tick, launched with its descriptors (ticks_host[0], layers_host) as kernel arguments
hipError_t launch_bgra_stream(const DTick *ticks_host, const DLayer *layers_host, const DTick *ticks, const DLayer *layers, int n_ticks, int maxW, int maxH, hipStream_t stream) {
    const int nl = ticks_host[0].n_layers;
    const int strips_x = (maxW + 63) / 64;
    // rows per chunk.  Launches that fill the chip: tall chunks amortise the per-chunk geometry and the first ring fill — enough chunks for
    // CHV_STREAM_ROUNDS rounds of waves.  Small launches (a Swift VideoMixer issues ONE tick and waits): a wave's rows are a serial chain — a lone
    // wave issues an instruction every ~2.3 ns and sits out every memory round trip itself — so the chain is cut short, down to 4 rows,
    // until about 4 800 waves share the launch (tools/stream_rows_sweep.sh, 720p ticks, us per launch at 4 / 6 / 8 / 12 / 16 / 24 rows:
    // one tick 10.9 / 12.4 / 13.2 / 18.0 / 20.0 / 28.2; two 18.9 / 17.6 / 18.8 / 21.9 / 24.0 / 33.7; eight 54.2 / 51.2 / 51.8 / 51.0 / 54.6 / 52.3;
    // sixteen 104.5 / 97.2 / 95.7 / 91.6 / 93.8 / 94.3).
    const long want = 1024L * CHV_STREAM_WAVES * CHV_STREAM_ROUNDS;
    const long chunks = std::max<long>(1, want / std::max<long>(1, (long)n_ticks * strips_x));
    const long wave_rows = (long)n_ticks * strips_x * maxH;
    const long rows_small = std::min<long>(CHV_STREAM_SMALL_ROWS_MAX, std::max<long>(CHV_STREAM_MIN_ROWS, (wave_rows + CHV_STREAM_SMALL_WAVES - 1) / CHV_STREAM_SMALL_WAVES));
    int rows = (int)std::max<long>(rows_small, (maxH + chunks - 1) / chunks);
    if (CHV_STREAM_ROWS_FIXED > 0) rows = CHV_STREAM_ROWS_FIXED;      // (sweeps: tools/build_variant.sh ... -DCHV_STREAM_ROWS_FIXED=n)
    rows = std::max(1, std::min(rows, maxH));
    const int chunks_y = (maxH + rows - 1) / rows;
    const long total = (long)n_ticks * chunks_y * ((strips_x + ST_WAVES - 1) / ST_WAVES);
    dim3 grid((unsigned)(((total + 7) / 8) * 8));
    const bool planar = layers_host[ticks_host[0].first_layer].kind == LK_BGRA_FROM_Y420P;
    const size_t layer_bytes = planar ? (size_t)st_layer_bytes<1, true>() : (size_t)st_layer_bytes<1, false>();
    const size_t lds = (size_t)ST_WAVES * ((size_t)nl * layer_bytes + ST_TAB * (sizeof(uint4) + sizeof(uint32_t)));
    // the absorbed form of the colour matrix (pixel_math.hip.h) when every layer's matrix has one — all but BT.601 full range
    bool absorb = CHV_STREAM_ABSORB != 0;
    for (int i = 0; i < n_ticks && absorb; i++)
        for (int l = 0; l < nl; l++) absorb = absorb && csc_absorbable(layers_host[ticks_host[i].first_layer + l].csc);
    if (!ticks) {
        // one tick, descriptors as kernel arguments (launch_transient)
        if (n_ticks != 1 || !layers_host) return hipErrorInvalidValue;
        auto go = [&](auto tag) {
            constexpr int NL = decltype(tag)::value;
            StreamOne<NL> a;
            a.t = ticks_host[0];
            a.t.first_layer = 0;
            for (int l = 0; l < NL; l++) a.l[l] = layers_host[ticks_host[0].first_layer + l];
            auto fire = [&](auto pl, auto ab) {
                hipLaunchKernelGGL((tick_bgra_stream_one<NL, decltype(pl)::value, decltype(ab)::value>), grid, dim3(64 * ST_WAVES), lds, stream, a, strips_x, chunks_y, rows);
            };
            if (planar) { if (absorb) fire(std::true_type{}, std::true_type{}); else fire(std::true_type{}, std::false_type{}); }
            else { if (absorb) fire(std::false_type{}, std::true_type{}); else fire(std::false_type{}, std::false_type{}); }
        };
        switch (nl) {
        case 1: go(std::integral_constant<int, 1>{}); break;
        case 2: go(std::integral_constant<int, 2>{}); break;
        case 3: go(std::integral_constant<int, 3>{}); break;
        default: go(std::integral_constant<int, 4>{}); break;
        }
        return hipGetLastError();
    }
#define CHV_ST_GO2(N, P, A) hipLaunchKernelGGL((tick_bgra_stream<N, P, A>), grid, dim3(64 * ST_WAVES), lds, stream, ticks, layers, n_ticks, strips_x, chunks_y, rows)
#define CHV_ST_GO(N) do { if (planar) { if (absorb) CHV_ST_GO2(N, true, true); else CHV_ST_GO2(N, true, false); } \
                          else { if (absorb) CHV_ST_GO2(N, false, true); else CHV_ST_GO2(N, false, false); } } while (0)
    switch (nl) {
    case 1: CHV_ST_GO(1); break;
    case 2: CHV_ST_GO(2); break;
    case 3: CHV_ST_GO(3); break;
    default: CHV_ST_GO(4); break;
    }
#undef CHV_ST_GO
#undef CHV_ST_GO2
    return hipGetLastError();
}

}  // namespace chv

// tools/ubench_issue.cpp — VALU issue rate on one gfx950 SIMD as a function of (a) resident waves per
// SIMD and (b) independent dependency chains per wave.  Answers: how many ready waves / how much ILP
// does a VALU-bound pixel kernel need before the SIMD issues at its best rate?
// Build: hipcc --offload-arch=gfx950 -O2 tools/ubench_issue.cpp -o /tmp/ubench_issue
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

#define REP4(x) x x x x
#define REP16(x) REP4(REP4(x))

// CH chains of dependent v_mul_f32 (OP 0) or v_mul_i32_i24 (OP 1), round-robin; 64 instructions per REP
template <int OP, int CH>
__global__ void bench(uint64_t *out, int iters, float seed) {
    float a[8]; uint32_t q[8];
    for (int i = 0; i < 8; i++) { a[i] = seed + i; q[i] = threadIdx.x + i; }
    float b = seed * 2; uint32_t m = threadIdx.x | 3;
    uint64_t s0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; it++) {
        if (OP == 0) {
            if (CH == 1) { REP16(REP4(asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[0]) : "v"(b));)) }
            if (CH == 2) { REP16(REP4(asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[0]) : "v"(b));) REP4(asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[1]) : "v"(b));)) }
            if (CH == 2) {}
            if (CH == 4) { REP16(asm volatile("v_mul_f32 %0, %0, %4\n v_mul_f32 %1, %1, %4\n v_mul_f32 %2, %2, %4\n v_mul_f32 %3, %3, %4" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]) : "v"(b));) }
            if (CH == 8) { REP4(REP4(asm volatile("v_mul_f32 %0, %0, %4\n v_mul_f32 %1, %1, %4\n v_mul_f32 %2, %2, %4\n v_mul_f32 %3, %3, %4" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]) : "v"(b));
                                     asm volatile("v_mul_f32 %0, %0, %4\n v_mul_f32 %1, %1, %4\n v_mul_f32 %2, %2, %4\n v_mul_f32 %3, %3, %4" : "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]) : "v"(b));)) }
        } else {
            if (CH == 1) { REP16(REP4(asm volatile("v_mul_i32_i24 %0, %0, %1" : "+v"(q[0]) : "v"(m));)) }
            if (CH == 2) { REP16(REP4(asm volatile("v_mul_i32_i24 %0, %0, %1" : "+v"(q[0]) : "v"(m));) REP4(asm volatile("v_mul_i32_i24 %0, %0, %1" : "+v"(q[1]) : "v"(m));)) }
            if (CH == 4) { REP16(asm volatile("v_mul_i32_i24 %0, %0, %4\n v_mul_i32_i24 %1, %1, %4\n v_mul_i32_i24 %2, %2, %4\n v_mul_i32_i24 %3, %3, %4" : "+v"(q[0]), "+v"(q[1]), "+v"(q[2]), "+v"(q[3]) : "v"(m));) }
            if (CH == 8) { REP4(REP4(asm volatile("v_mul_i32_i24 %0, %0, %4\n v_mul_i32_i24 %1, %1, %4\n v_mul_i32_i24 %2, %2, %4\n v_mul_i32_i24 %3, %3, %4" : "+v"(q[0]), "+v"(q[1]), "+v"(q[2]), "+v"(q[3]) : "v"(m));
                                     asm volatile("v_mul_i32_i24 %0, %0, %4\n v_mul_i32_i24 %1, %1, %4\n v_mul_i32_i24 %2, %2, %4\n v_mul_i32_i24 %3, %3, %4" : "+v"(q[4]), "+v"(q[5]), "+v"(q[6]), "+v"(q[7]) : "v"(m));)) }
        }
    }
    uint64_t s1 = __builtin_readcyclecounter();
    if (threadIdx.x % 64 == 0) out[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = s1 - s0;
    float acc = 0; uint32_t qa = 0;
    for (int i = 0; i < 8; i++) { acc += a[i]; qa += q[i]; }
    if (acc + (float)qa == 12345.678f) out[0] = 1;
}

// CH==2 in the form above is 4 dependent then 4 dependent: replace by true round robin
template <int OP>
__global__ void bench2(uint64_t *out, int iters, float seed) {
    float a0 = seed, a1 = seed + 1; uint32_t q0 = threadIdx.x, q1 = threadIdx.x + 1;
    float b = seed * 2; uint32_t m = threadIdx.x | 3;
    uint64_t s0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; it++) {
        if (OP == 0) { REP16(asm volatile("v_mul_f32 %0, %0, %2\n v_mul_f32 %1, %1, %2\n v_mul_f32 %0, %0, %2\n v_mul_f32 %1, %1, %2" : "+v"(a0), "+v"(a1) : "v"(b));) }
        else { REP16(asm volatile("v_mul_i32_i24 %0, %0, %2\n v_mul_i32_i24 %1, %1, %2\n v_mul_i32_i24 %0, %0, %2\n v_mul_i32_i24 %1, %1, %2" : "+v"(q0), "+v"(q1) : "v"(m));) }
    }
    uint64_t s1 = __builtin_readcyclecounter();
    if (threadIdx.x % 64 == 0) out[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = s1 - s0;
    if (a0 + a1 + (float)(q0 + q1) == 12345.678f) out[0] = 1;
}

static uint64_t *d_out;

template <typename K>
void run(const char *name, K kernel, int chains, int waves_per_simd) {
    const int iters = 400;
    // one block per CU (256 CUs), 4 * waves_per_simd waves per block when <= 16 waves, else 2 blocks per CU
    int wpb = 4 * waves_per_simd, blocks = 256;
    if (wpb > 16) { wpb /= 2; blocks *= 2; }
    dim3 block(64 * wpb), grid(blocks);
    size_t lds = wpb > 8 || blocks > 256 ? 65536 : 0;   // keep >2 blocks from sharing a CU
    hipLaunchKernelGGL(kernel, grid, block, lds, 0, d_out, iters, 1.5f);
    hipDeviceSynchronize();
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    hipLaunchKernelGGL(kernel, grid, block, lds, 0, d_out, iters, 1.5f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    int nw = blocks * wpb;
    std::vector<uint64_t> h(nw);
    hipMemcpy(h.data(), d_out, nw * 8, hipMemcpyDeviceToHost);
    double avg = 0; for (auto v : h) avg += (double)v; avg /= nw;
    double n_inst = (double)iters * 64.0;
    printf("%-14s chains=%d waves/SIMD=%d  wall %.3f ns/instr/SIMD   per-wave %.1f clk/instr  -> %.2f clk/instr/SIMD\n",
           name, chains, waves_per_simd, ms * 1e6 / (n_inst * waves_per_simd), avg / n_inst, avg / n_inst / waves_per_simd);
}

int main() {
    hipMalloc(&d_out, 8192 * 8);
    for (int w : {1, 2, 3, 4, 5, 6, 8}) {
        run("v_mul_f32", bench<0, 1>, 1, w); run("v_mul_f32", bench2<0>, 2, w);
        run("v_mul_f32", bench<0, 4>, 4, w); run("v_mul_f32", bench<0, 8>, 8, w);
    }
    for (int w : {1, 2, 4, 5, 6, 8}) {
        run("v_mul_i32_i24", bench<1, 1>, 1, w); run("v_mul_i32_i24", bench2<1>, 2, w);
        run("v_mul_i32_i24", bench<1, 4>, 4, w); run("v_mul_i32_i24", bench<1, 8>, 8, w);
    }
    return 0;
}

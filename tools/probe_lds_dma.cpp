// tools/probe_lds_dma.cpp — where does global_load_lds_dwordx4 put a lane's 16 bytes?  (gfx950; hipcc --offload-arch=gfx950 -O3)
// Loads 64 x 16 bytes from a pattern buffer with M0 = base (+ an instruction offset), partial exec, then dumps the LDS.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
__global__ void k(const unsigned char *p, unsigned *out, int base, int active) {
    __shared__ __attribute__((aligned(16))) unsigned lds[1024];
    for (int i = threadIdx.x; i < 1024; i += 64) lds[i] = 0xDEADBEEFu;
    __syncthreads();
    const unsigned char *q = p + (63 - threadIdx.x) * 16;          // reversed lane order in memory: lane i reads chunk 63 - i
    unsigned m0v = (unsigned)(size_t)lds + (unsigned)base;
    if ((int)threadIdx.x < active)
        asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off offset:0" :: "s"(__builtin_amdgcn_readfirstlane(m0v)), "v"(q) : "memory");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int i = threadIdx.x; i < 1024; i += 64) out[i] = lds[i];
}
int main() {
    std::vector<uint32_t> h(64 * 4);
    for (int i = 0; i < 64; i++) for (int j = 0; j < 4; j++) h[i * 4 + j] = (uint32_t)(i << 8 | j);      // chunk i, dword j
    unsigned char *d; unsigned *o; hipMalloc(&d, 1024); hipMalloc(&o, 4096);
    hipMemcpy(d, h.data(), 1024, hipMemcpyHostToDevice);
    for (int t = 0; t < 3; t++) {
        int base = t == 1 ? 512 : 0, active = t == 2 ? 20 : 64;
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, o, base, active);
        std::vector<uint32_t> r(1024); hipMemcpy(r.data(), o, 4096, hipMemcpyDeviceToHost);
        printf("base %d active %d:", base, active);
        int shown = 0;
        for (int i = 0; i < 1024 && shown < 14; i++) if (r[i] != 0xDEADBEEFu && (i % 4 == 0)) { printf(" lds[%d]=%x", i, r[i]); shown++; }
        int cnt = 0; for (int i = 0; i < 1024; i++) cnt += r[i] != 0xDEADBEEFu;
        printf("  (%d dwords written)\n", cnt);
    }
    return 0;
}

// tools/probe_cvt_pk_u8.cpp — what v_cvt_pk_u8_f32 does with ties, out-of-range values, NaN and infinities on gfx950.
// Build: hipcc --offload-arch=gfx950 -O2 tools/probe_cvt_pk_u8.cpp -o tools/probe_cvt_pk_u8.bin
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdint>
#include <cstring>
#include <vector>

__global__ void k(const float *x, uint32_t *out, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint32_t d = 0xAABBCCDDu;
    asm volatile("v_cvt_pk_u8_f32 %0, %1, 1, %0" : "+v"(d) : "v"(x[i]));
    out[i] = d;
}

int main() {
    std::vector<float> v = { 0.f, -0.f, 0.49999997f, 0.5f, 0.50000006f, 1.5f, 2.5f, 3.5f, 254.5f, 255.f, 255.49998f, 255.5f, 256.f, 300.f,
                             1e10f, -0.4f, -0.5f, -0.6f, -1.f, -1e10f, INFINITY, -INFINITY, NAN, 127.99999f, 128.5f, 0.99999994f, 1.0000001f };
    // exhaustive check of two candidate models over a fine grid
    for (int i = 0; i <= 256 * 64 + 128; i++) v.push_back((float)i / 64.0f - 0.5f);
    int n = (int)v.size();
    float *dx; uint32_t *dout;
    hipMalloc(&dx, n * 4); hipMalloc(&dout, n * 4);
    hipMemcpy(dx, v.data(), n * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3((n + 255) / 256), dim3(256), 0, 0, dx, dout, n);
    std::vector<uint32_t> o(n);
    hipMemcpy(o.data(), dout, n * 4, hipMemcpyDeviceToHost);
    int bad_rte = 0, bad_trunc = 0, bad_half_up = 0, bad_other_bytes = 0;
    for (int i = 0; i < n; i++) {
        float x = v[i];
        uint32_t b = (o[i] >> 8) & 255;
        if ((o[i] & 0xFFFF00FFu) != 0xAABB00DDu) bad_other_bytes++;
        if (i < 27) printf("x = %-14.9g -> %u\n", x, b);
        if (std::isnan(x)) continue;
        float c = fminf(fmaxf(x, 0.f), 255.f);
        uint32_t rte = (uint32_t)nearbyintf(c), tr = (uint32_t)c, hu = (uint32_t)floorf(c + 0.5f);
        bad_rte += rte != b; bad_trunc += tr != b; bad_half_up += hu != b;
    }
    printf("n = %d: mismatches vs clamp+RTE %d, vs clamp+truncate %d, vs clamp+floor(x+0.5) %d; other bytes disturbed %d\n",
           n, bad_rte, bad_trunc, bad_half_up, bad_other_bytes);
    return 0;
}

// tools/probe_kernarg.cpp — what a kernel ARGUMENT BLOCK of a given size costs a launch: an empty kernel taking a struct of N bytes by value,
// launched back to back (host time per launch, nothing waited for) and one at a time with a host wait (wall per launch).
// hipcc --offload-arch=gfx950 -O2 -o tools/probe_kernarg.bin tools/probe_kernarg.cpp ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
template <int N> struct Blob { unsigned char b[N]; };
template <int N> __global__ void k(const Blob<N> a, int *out) { if (out) *out = a.b[N - 1]; }
template <int N> static void run() {
    Blob<N> a{};
    hipStream_t s; (void)hipStreamCreate(&s);
    for (int i = 0; i < 50; i++) hipLaunchKernelGGL(k<N>, dim3(1), dim3(64), 0, s, a, (int *)nullptr);
    (void)hipStreamSynchronize(s);
    double best = 1e9, bestw = 1e9;
    for (int rep = 0; rep < 5; rep++) {
        auto t0 = std::chrono::steady_clock::now();
        for (int i = 0; i < 300; i++) hipLaunchKernelGGL(k<N>, dim3(1), dim3(64), 0, s, a, (int *)nullptr);
        auto t1 = std::chrono::steady_clock::now();
        (void)hipStreamSynchronize(s);
        double us = std::chrono::duration<double, std::micro>(t1 - t0).count() / 300;
        if (us < best) best = us;
        auto w0 = std::chrono::steady_clock::now();
        for (int i = 0; i < 300; i++) { hipLaunchKernelGGL(k<N>, dim3(1), dim3(64), 0, s, a, (int *)nullptr); (void)hipStreamSynchronize(s); }
        auto w1 = std::chrono::steady_clock::now();
        double wus = std::chrono::duration<double, std::micro>(w1 - w0).count() / 300;
        if (wus < bestw) bestw = wus;
    }
    printf("argument block %4d bytes: %5.2f us of host time per launch back to back, %5.2f us per launch + wait\n", N, best, bestw);
    (void)hipStreamDestroy(s);
}
int main() { run<16>(); run<464>(); run<1200>(); run<1568>(); run<2304>(); run<3040>(); run<4000>(); run<16>(); return 0; }

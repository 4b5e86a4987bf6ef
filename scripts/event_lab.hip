// What does per-kernel timing cost?  N dependent kernels of ~20 us each on one stream, three ways:
//   plain       no timing;
//   record      hipEventRecord before and after every kernel (two marker packets per kernel);
//   ext         hipExtLaunchKernelGGL(start, stop): the events take the dispatch packet's own begin/end timestamps.
// Also checks hipEventElapsedTime across two different ext-launched kernels (first kernel's begin -> second kernel's end).
// build: hipcc -O2 --offload-arch=gfx950 scripts/event_lab.hip -o build_lab/event_lab
#include <hip/hip_ext.h>
#include <hip/hip_runtime.h>
#include <stdio.h>

#include <chrono>
#include <vector>

__global__ void spin(float *p, int iters) {
    float v = p[threadIdx.x + blockIdx.x * blockDim.x];
    for (int i = 0; i < iters; i++) v = v * 1.0001f + 0.5f;
    p[threadIdx.x + blockIdx.x * blockDim.x] = v;
}

#define CK(x)                                                                  \
    do {                                                                       \
        hipError_t err_ = (x);                                                    \
        if (err_ != hipSuccess) {                                              \
            printf("%s -> %s\n", #x, hipGetErrorString(err_));                 \
            return 1;                                                          \
        }                                                                      \
    } while (0)

int main() {
    const int N = 400, iters = 6000;
    float *p;
    CK(hipMalloc(&p, 256 * 256 * 4));
    CK(hipMemset(p, 0, 256 * 256 * 4));
    hipStream_t s;
    CK(hipStreamCreate(&s));
    std::vector<hipEvent_t> ev(2 * N);
    for (auto &e : ev) CK(hipEventCreate(&e));
    auto wall = [&](int mode) -> double {
        CK(hipStreamSynchronize(s));
        auto t0 = std::chrono::steady_clock::now();
        for (int i = 0; i < N; i++) {
            if (mode == 0) {
                hipLaunchKernelGGL(spin, dim3(256), dim3(256), 0, s, p, iters);
            } else if (mode == 1) {
                (void)hipEventRecord(ev[2 * i], s);
                hipLaunchKernelGGL(spin, dim3(256), dim3(256), 0, s, p, iters);
                (void)hipEventRecord(ev[2 * i + 1], s);
            } else {
                hipExtLaunchKernelGGL(spin, dim3(256), dim3(256), 0, s, ev[2 * i], ev[2 * i + 1], 0, p, iters);
            }
        }
        CK(hipStreamSynchronize(s));
        return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
    };
    for (int rep = 0; rep < 2; rep++)
        for (int mode = 0; mode < 3; mode++) {
            double us = wall(mode);
            double sum = 0;
            if (mode) {
                for (int i = 0; i < N; i++) {
                    float ms = 0;
                    CK(hipEventElapsedTime(&ms, ev[2 * i], ev[2 * i + 1]));
                    sum += ms * 1e3;
                }
            }
            printf("%-6s %d kernels: wall %.1f us (%.2f us per kernel); sum of event intervals %.1f us\n",
                   mode == 0 ? "plain" : mode == 1 ? "record" : "ext", N, us, us / N, sum);
        }
    // first kernel's begin -> third kernel's end, with an un-timed kernel between
    hipExtLaunchKernelGGL(spin, dim3(256), dim3(256), 0, s, ev[0], nullptr, 0, p, iters);
    hipLaunchKernelGGL(spin, dim3(256), dim3(256), 0, s, p, iters);
    hipExtLaunchKernelGGL(spin, dim3(256), dim3(256), 0, s, nullptr, ev[1], 0, p, iters);
    CK(hipStreamSynchronize(s));
    float ms = 0;
    hipError_t e = hipEventElapsedTime(&ms, ev[0], ev[1]);
    printf("begin(k1) -> end(k3) across three kernels: %s, %.1f us\n", hipGetErrorString(e), ms * 1e3);
    // rebinding: stop event bound to two successive kernels, the last binding should win
    hipExtLaunchKernelGGL(spin, dim3(256), dim3(256), 0, s, ev[2], ev[3], 0, p, iters);
    hipExtLaunchKernelGGL(spin, dim3(256), dim3(256), 0, s, nullptr, ev[3], 0, p, iters);
    CK(hipStreamSynchronize(s));
    e = hipEventElapsedTime(&ms, ev[2], ev[3]);
    printf("begin(k1) -> end(k2) with the stop event bound twice: %s, %.1f us\n", hipGetErrorString(e), ms * 1e3);
    return 0;
}

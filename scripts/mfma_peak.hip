// Register-only fp32 MFMA loop: what v_mfma_f32_32x32x2_f32 sustains on this device (practical ceiling for the
// conv kernels).  hipcc --offload-arch=gfx950 -O3 scripts/mfma_peak.hip -o /tmp/mfma_peak && /tmp/mfma_peak
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int WAVES>
__global__ void __launch_bounds__(WAVES * 64) k(float *out, int iters, float a0, float b0) {
    f32x16 acc[4];
    for (int i = 0; i < 4; i++) for (int r = 0; r < 16; r++) acc[i][r] = 0.f;
    float a = a0 + threadIdx.x * 1e-3f, b = b0 - threadIdx.x * 1e-3f;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int u = 0; u < 8; u++) {
            acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(b, a, acc[1], 0, 0, 0);
            acc[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, a, acc[2], 0, 0, 0);
            acc[3] = __builtin_amdgcn_mfma_f32_32x32x2f32(b, b, acc[3], 0, 0, 0);
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    unsigned long long r1 = __builtin_amdgcn_s_memrealtime();
    float s = 0.f;
    for (int i = 0; i < 4; i++) for (int r = 0; r < 16; r++) s += acc[i][r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        out[0] = (float)(t1 - t0);
        out[1] = (float)(r1 - r0);
    }
}
template <int WAVES>
void run(int blocks, const char *name) {
    float *d;
    hipMalloc(&d, blocks * WAVES * 64 * 4);
    const int iters = 20000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 3; rep++) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(k<WAVES>, dim3(blocks), dim3(WAVES * 64), 0, 0, d, iters, 1.01f, 0.99f);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        float h[2]; hipMemcpy(h, d, 8, hipMemcpyDeviceToHost);
        double flops = (double)blocks * WAVES * iters * 32 * (2.0 * 32 * 32 * 2);
        printf("%s blocks=%d waves/blk=%d: %.2f ms  %.1f TFLOP/s  clock %.0f MHz (memtime/memrealtime*100MHz)\n", name, blocks, WAVES, ms,
               flops / ms / 1e9, h[0] / h[1] * 100.0);
    }
    hipFree(d);
}
int main() {
    run<4>(256, "1 wave/SIMD ");
    run<4>(512, "2 waves/SIMD");
    run<8>(256, "2 waves/SIMD (8-wave blocks)");
    return 0;
}

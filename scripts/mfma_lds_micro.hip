// What does the LDS-fed fp32 MFMA stream of the conv kernels sustain, feature by feature?
//   V0 registers only | V1 + operand fragments from LDS (ds_read2_b32, one k-pair ahead) | V2 + barrier every 64 MFMAs
//   V3 = V2 with 4 extra idle waves per workgroup that only hit the barrier (the producer role)
// hipcc --offload-arch=gfx950 -O3 scripts/mfma_lds_micro.hip -o /tmp/mm && /tmp/mm
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int V>
__global__ void __launch_bounds__(V == 3 ? 512 : 256) k(float *out, int iters) {
    __shared__ float As[2][32][129];
    __shared__ float Bs[2][32][128];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < 2 * 32 * 129; i += blockDim.x) (&As[0][0][0])[i] = 1.0f + i * 1e-6f;
    for (int i = tid; i < 2 * 32 * 128; i += blockDim.x) (&Bs[0][0][0])[i] = 1.0f - i * 1e-6f;
    __syncthreads();
    if (V == 3 && wave >= 4) {  // idle "producer" waves: barrier only
        for (int it = 0; it < iters; it++) __syncthreads();
        return;
    }
    const int wm = wave >> 1, wn = wave & 1, l31 = lane & 31, lh = lane >> 5;
    f32x16 acc[2][2];
    for (int i = 0; i < 2; i++) for (int j = 0; j < 2; j++) for (int r = 0; r < 16; r++) acc[i][j][r] = 0.f;
    float a0 = 1.01f + lane * 1e-3f, b0 = 0.99f - lane * 1e-3f;
    for (int it = 0; it < iters; it++) {
        const int cur = it & 1;
        const float *as = &As[cur][0][0] + wm * 64 + l31;
        const float *bs = &Bs[cur][0][0] + wn * 64 + l31;
        float av[2][2], bv[2][2];
        if (V >= 1) { av[0][0] = as[lh * 129]; av[0][1] = as[lh * 129 + 32]; bv[0][0] = bs[lh * 128]; bv[0][1] = bs[lh * 128 + 32]; }
        else { av[0][0] = a0; av[0][1] = b0; bv[0][0] = b0; bv[0][1] = a0; av[1][0] = a0; av[1][1] = b0; bv[1][0] = b0; bv[1][1] = a0; }
#pragma unroll
        for (int kk = 0; kk < 16; kk++) {
            if (V >= 1 && kk + 1 < 16) {
                av[(kk + 1) & 1][0] = as[((kk + 1) * 2 + lh) * 129];
                av[(kk + 1) & 1][1] = as[((kk + 1) * 2 + lh) * 129 + 32];
                bv[(kk + 1) & 1][0] = bs[((kk + 1) * 2 + lh) * 128];
                bv[(kk + 1) & 1][1] = bs[((kk + 1) * 2 + lh) * 128 + 32];
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < 2; i++)
#pragma unroll
                for (int j = 0; j < 2; j++)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[kk & 1][i], bv[kk & 1][j], acc[i][j], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
        if (V >= 2) __syncthreads();
    }
    float s = 0.f;
    for (int i = 0; i < 2; i++) for (int j = 0; j < 2; j++) for (int r = 0; r < 16; r++) s += acc[i][j][r];
    out[blockIdx.x * 256 + tid] = s;
}
template <int V>
void run(const char *name) {
    const int blocks = 512 * 6, iters = 400;
    float *d; (void)hipMalloc(&d, blocks * 256 * 4);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int rep = 0; rep < 2; rep++) {
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL(k<V>, dim3(blocks), dim3(V == 3 ? 512 : 256), 0, 0, d, iters);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        double flops = (double)blocks * 4 * iters * 64 * 4096.0;
        if (rep) printf("%-40s %.2f ms  %.1f TFLOP/s (%.1f%% of 157.3)\n", name, ms, flops / ms / 1e9, flops / ms / 1e9 / 157.3 * 100);
    }
    (void)hipFree(d);
}
int main() {
    run<0>("V0 registers only");
    run<1>("V1 + LDS fragment reads");
    run<2>("V2 + barrier / 64 MFMAs");
    run<3>("V3 + 4 idle waves at the barrier");
    return 0;
}

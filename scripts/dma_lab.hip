// LDS-DMA (buffer_load_dwordx4 ... lds) throughput per CU as a function of the shape of one instruction's 1 KB:
// contiguous, 8 rows x 128 B, 16 rows x 64 B (row stride 1 KB or 16 KB), and of the number of loader waves per CU.
//   hipcc --offload-arch=gfx950 -O3 scripts/dma_lab.hip -o /tmp/dma_lab && /tmp/dma_lab
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((address_space(3))) void lds_void_t;

// PIECE = contiguous bytes per row piece (1024, 128, 64); STRIDE = bytes between rows
template <int WAVES, int PIECE>
__global__ void __launch_bounds__(WAVES * 64) k(const float *src, float *out, int iters, unsigned stride, unsigned span_mask) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(src), 0, 1 << 30, 0x00020000);
    constexpr int LPP = PIECE / 16;            // lanes per piece
    const unsigned lane_off = (unsigned)(lane / LPP) * stride + (unsigned)(lane % LPP) * 16u;
    unsigned base = (blockIdx.x * 7919u + w * 104729u) * 1024u;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int j = 0; j < 8; j++) {
            const unsigned off = ((base + (unsigned)(it * 8 + j) * 64u * stride) & span_mask) + lane_off;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (lds_void_t *)(lds + (w * 8 + j) * 1024), 16, (int)off, 0, 0, 0);
        }
        asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) out[blockIdx.x] = ((float *)lds)[5];
}

template <int WAVES, int PIECE>
void run(const float *src, float *out, unsigned stride, unsigned span_mask, const char *name) {
    const int iters = 4000;
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    float best = 1e30f;
    for (int rep = 0; rep < 3; rep++) {
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL((k<WAVES, PIECE>), dim3(256), dim3(WAVES * 64), WAVES * 8 * 1024, 0, src, out, iters, stride, span_mask);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        if (rep && ms < best) best = ms;
    }
    const double bytes = 256.0 * WAVES * iters * 8 * 1024.0;
    printf("%-44s waves/CU %d  %7.2f ms  %6.1f GB/s per CU  %5.2f TB/s chip\n", name, WAVES, best, bytes / best / 1e6 / 256, bytes / best / 1e9);
    fflush(stdout);
}

int main() {
    float *src, *out;
    (void)hipMalloc(&src, 1u << 30);
    (void)hipMemset(src, 0, 1u << 30);
    (void)hipMalloc(&out, 4096);
    const unsigned l2 = (4u << 20) - 1, big = (512u << 20) - 1;   // 4 MB window (L2 / MALL resident) vs 512 MB (HBM)
    run<4, 1024>(src, out, 1024, l2, "contiguous 1 KB, 4 MB window");
    run<4, 128>(src, out, 1024, l2, "8 rows x 128 B, stride 1 KB, 4 MB window");
    run<4, 128>(src, out, 16384, l2, "8 rows x 128 B, stride 16 KB, 4 MB window");
    run<4, 64>(src, out, 1024, l2, "16 rows x 64 B, stride 1 KB, 4 MB window");
    run<4, 64>(src, out, 128, l2, "16 rows x 64 B, stride 128 B, 4 MB window");
    run<1, 1024>(src, out, 1024, l2, "contiguous 1 KB, 4 MB window");
    run<2, 1024>(src, out, 1024, l2, "contiguous 1 KB, 4 MB window");
    run<8, 1024>(src, out, 1024, l2, "contiguous 1 KB, 4 MB window");
    run<4, 1024>(src, out, 1024, big, "contiguous 1 KB, 512 MB window");
    run<4, 128>(src, out, 16384, big, "8 rows x 128 B, stride 16 KB, 512 MB window");
    run<4, 64>(src, out, 1024, big, "16 rows x 64 B, stride 1 KB, 512 MB window");
    return 0;
}

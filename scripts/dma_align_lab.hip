// Does a 16-byte-per-lane LDS-DMA (buffer_load_dwordx4 ... lds) accept an LDS base that is only 8-byte (or 4-byte) aligned?
// One wave copies 1 KB from global to LDS at base + SHIFT bytes, reads LDS back with plain loads and writes it out.
// build: hipcc -O2 --offload-arch=gfx950 scripts/dma_align_lab.hip -o build_lab/dma_align_lab
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>

typedef __attribute__((address_space(3))) void lds_void_t;

__global__ void probe(const float *src, float *dst, int shift_floats) {
    __shared__ __attribute__((aligned(16))) float lds[1024];
    for (int i = threadIdx.x; i < 1024; i += 64) lds[i] = -1.f;
    __syncthreads();
    const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(src), 0, 4096, 0x00020000);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (lds_void_t *)(lds + shift_floats), 16, (int)(threadIdx.x * 16), 0, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int i = threadIdx.x; i < 1024; i += 64) dst[i] = lds[i];
}

int main() {
    std::vector<float> h(1024), out(1024);
    for (int i = 0; i < 1024; i++) h[i] = (float)i;
    float *s, *d;
    hipMalloc(&s, 4096);
    hipMalloc(&d, 4096);
    hipMemcpy(s, h.data(), 4096, hipMemcpyHostToDevice);
    for (int shift : {0, 4, 2, 1, 6}) {
        hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, s, d, shift);
        hipDeviceSynchronize();
        hipMemcpy(out.data(), d, 4096, hipMemcpyDeviceToHost);
        int ok = 0, first_bad = -1;
        for (int i = 0; i < 256; i++) {
            if (out[shift + i] == (float)i) ok++;
            else if (first_bad < 0) first_bad = i;
        }
        printf("LDS base + %2d bytes: %d of 256 floats landed where expected (first mismatch at %d: got %.0f); lds[0..7] =", shift * 4, ok,
               first_bad, first_bad >= 0 ? out[shift + first_bad] : 0.f);
        for (int i = 0; i < 8; i++) printf(" %.0f", out[i]);
        printf("\n");
    }
    return 0;
}

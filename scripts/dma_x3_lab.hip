// 12-byte-per-lane LDS-DMA (buffer_load_dwordx3 ... lds) on gfx950: where do the lanes' 12 bytes land (base + 12 * lane?), does a
// base that is only 16-byte aligned work, what do out-of-range lanes (offset bit 31) deposit, and does the scalar offset operand add.
// One wave copies 768 B from global to LDS, reads LDS back with plain loads and writes it out.
// build: hipcc -O2 --offload-arch=gfx950 scripts/dma_x3_lab.hip -o build_lab/dma_x3_lab
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>

typedef __attribute__((address_space(3))) void lds_void_t;

__global__ void probe(const float *src, float *dst, int shift_floats, int oob_from, int soff) {
    __shared__ __attribute__((aligned(16))) float lds[1024];
    for (int i = threadIdx.x; i < 1024; i += 64) lds[i] = -1.f;
    __syncthreads();
    const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(src), 0, 4096, 0x00020000);
    // lane l fetches the 12 bytes at 16 * l (so the source stride differs from the destination stride: 3 of every 4 floats)
    unsigned off = threadIdx.x * 16;
    if ((int)threadIdx.x >= oob_from) off |= 0x80000000u;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (lds_void_t *)(lds + shift_floats), 12, (int)off, soff, 0, 0);
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
    struct Case { int shift, oob_from, soff; };
    for (Case c : {Case{0, 64, 0}, Case{4, 64, 0}, Case{196, 64, 0}, Case{0, 40, 0}, Case{0, 64, 64}, Case{1, 64, 0}}) {
        hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, s, d, c.shift, c.oob_from, c.soff);
        hipDeviceSynchronize();
        hipMemcpy(out.data(), d, 4096, hipMemcpyDeviceToHost);
        int ok = 0, first_bad = -1;
        for (int l = 0; l < 64; l++)
            for (int e = 0; e < 3; e++) {
                const float want = l >= c.oob_from ? 0.f : (float)(l * 4 + e + c.soff / 4);
                if (out[c.shift + l * 3 + e] == want) ok++;
                else if (first_bad < 0) first_bad = l * 3 + e;
            }
        printf("base + %3d B, lanes >= %d out of range, soffset %d: %d of 192 floats as expected (first mismatch at float %d: got %.0f); after the block: %.0f %.0f; lds[0..7] =",
               c.shift * 4, c.oob_from, c.soff, ok, first_bad, first_bad >= 0 ? out[c.shift + first_bad] : 0.f, out[c.shift + 192], out[c.shift + 193]);
        for (int i = 0; i < 8; i++) printf(" %.0f", out[i]);
        printf("\n");
    }
    return 0;
}

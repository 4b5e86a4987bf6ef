// Where does an LDS-fed fp32 MFMA loop lose time on gfx950?  A lab for the consumer loop of the wave-specialised conv
// kernels (csrc/conv_igemm.hip): the same instruction stream (two ds_read2_b32 one k-pair ahead of four
// v_mfma_f32_32x32x2_f32, 16 k-pairs per K step) under different workgroup shapes, with and without the per-K-step
// barrier, idle or storing producer waves, constant or random operands.  Every variant runs >= 10 ms; prints TFLOP/s,
// % of 157.3 and the in-kernel clock (s_memtime / s_memrealtime).
//   hipcc --offload-arch=gfx950 -O3 scripts/mfma_loop_lab.hip -o /tmp/lab && /tmp/lab
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int BK = 32, AS = 129, BS = 128;

// MODE bits: 1 = operands from LDS, 2 = barrier per K step, 4 = producer waves store a tile into the idle buffer each step,
//            8 = producer waves also issue global loads each step, 16 = consumer LDS reads two k-pairs ahead
// SCHED: 0 = the k-pair's LDS reads clustered in ONE gap between MFMA groups (round-1 kernels); 1 = one LDS read per MFMA gap,
//        reads one k-pair ahead; 2 = one read per gap, two k-pairs ahead
template <int CONS, int PROD, int MODE, int SCHED = 0>
__global__ void __launch_bounds__((CONS + PROD) * 64) lab(float *out, const float *src, int iters, int rnd) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float *As = lds, *Bs = lds + 2 * BK * AS;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < 2 * BK * (AS + BS); i += blockDim.x)
        lds[i] = rnd ? src[(i * 7 + blockIdx.x) & 0xFFFFF] : 1.0f + i * 1e-6f;
    __syncthreads();
    unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    if (wave >= CONS) {  // producer role
        const int ptid = tid - CONS * 64;
        float4 ra[4], rb[4];
        for (int j = 0; j < 4; j++) { ra[j] = make_float4(1.f + j, 2.f, 3.f, 4.f); rb[j] = ra[j]; }
        const float4 *g = reinterpret_cast<const float4 *>(src);
        for (int it = 0; it < iters; it++) {
            if (MODE & 4) {
                const int buf = (it + 1) & 1;
                const int a_kq = ptid % 8, a_r0 = ptid / 8;
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    float *as = As + buf * BK * AS;
                    const int r = a_r0 + j * 32;
                    as[(a_kq * 4 + 0) * AS + r] = ra[j].x; as[(a_kq * 4 + 1) * AS + r] = ra[j].y;
                    as[(a_kq * 4 + 2) * AS + r] = ra[j].z; as[(a_kq * 4 + 3) * AS + r] = ra[j].w;
                }
                const int b_nq = ptid % 32, b_kr = ptid / 32;
#pragma unroll
                for (int j = 0; j < 4; j++)
                    *reinterpret_cast<float4 *>(&Bs[buf * BK * BS + (b_kr + j * 8) * BS + b_nq * 4]) = rb[j];
            }
            if (MODE & 8) {
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    ra[j] = g[(blockIdx.x * 977 + it * 4099 + j * 256 + ptid) & 0x3FFFF];
                    rb[j] = g[(blockIdx.x * 613 + it * 2053 + j * 256 + ptid + 1024) & 0x3FFFF];
                }
            }
            if (MODE & 2) __syncthreads();
        }
        if (ra[0].x == 12345.678f) out[0] = rb[1].y;
        return;
    }
    const int wm = wave >> 1, wn = wave & 1, l31 = lane & 31, lh = lane >> 5;
    f32x16 acc[2][2];
    for (int i = 0; i < 2; i++) for (int j = 0; j < 2; j++) for (int r = 0; r < 16; r++) acc[i][j][r] = 0.f;
    const float a0 = 1.01f + lane * 1e-3f, b0 = 0.99f - lane * 1e-3f;
    constexpr int AH = ((MODE & 16) || SCHED == 2) ? 2 : 1;  // read-ahead distance in k-pairs
    for (int it = 0; it < iters; it++) {
        const int cur = it & 1;
        const float *as = As + cur * BK * AS + (wm & 1) * 64 + l31;
        const float *bs = Bs + cur * BK * BS + wn * 64 + l31;
        float av[3][2], bv[3][2];
        if (MODE & 1) {
#pragma unroll
            for (int p = 0; p < AH; p++) {
                av[p][0] = as[(p * 2 + lh) * AS]; av[p][1] = as[(p * 2 + lh) * AS + 32];
                bv[p][0] = bs[(p * 2 + lh) * BS]; bv[p][1] = bs[(p * 2 + lh) * BS + 32];
            }
        } else {
#pragma unroll
            for (int p = 0; p < 3; p++) { av[p][0] = a0; av[p][1] = b0; bv[p][0] = b0; bv[p][1] = a0; }
        }
#pragma unroll
        for (int kk = 0; kk < 16; kk++) {
            const bool rd = (MODE & 1) && kk + AH < 16;
            constexpr int dummy = 0; (void)dummy;
            if (SCHED == 0) {
                if (rd) {
                    av[(kk + AH) % 3][0] = as[((kk + AH) * 2 + lh) * AS]; av[(kk + AH) % 3][1] = as[((kk + AH) * 2 + lh) * AS + 32];
                    bv[(kk + AH) % 3][0] = bs[((kk + AH) * 2 + lh) * BS]; bv[(kk + AH) % 3][1] = bs[((kk + AH) * 2 + lh) * BS + 32];
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int i = 0; i < 2; i++)
#pragma unroll
                    for (int j = 0; j < 2; j++)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[kk % 3][i], bv[kk % 3][j], acc[i][j], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            } else {
                // M00 | read A | M01 | read B | M10 | M11   (each '|' a hard scheduling fence)
                __builtin_amdgcn_sched_barrier(0);
                acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[kk % 3][0], bv[kk % 3][0], acc[0][0], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                if (rd) { av[(kk + AH) % 3][0] = as[((kk + AH) * 2 + lh) * AS]; av[(kk + AH) % 3][1] = as[((kk + AH) * 2 + lh) * AS + 32]; }
                __builtin_amdgcn_sched_barrier(0);
                acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[kk % 3][0], bv[kk % 3][1], acc[0][1], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                if (rd) { bv[(kk + AH) % 3][0] = bs[((kk + AH) * 2 + lh) * BS]; bv[(kk + AH) % 3][1] = bs[((kk + AH) * 2 + lh) * BS + 32]; }
                __builtin_amdgcn_sched_barrier(0);
                acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[kk % 3][1], bv[kk % 3][0], acc[1][0], 0, 0, 0);
                acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[kk % 3][1], bv[kk % 3][1], acc[1][1], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        if (MODE & 2) __syncthreads();
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    float s = 0.f;
    for (int i = 0; i < 2; i++) for (int j = 0; j < 2; j++) for (int r = 0; r < 16; r++) s += acc[i][j][r];
    out[16 + blockIdx.x * 512 + tid] = s;
    if (tid == 0 && blockIdx.x == 0) { out[0] = (float)(t1 - t0); out[1] = (float)(r1 - r0); }
}

static float *d_out, *d_src;

template <int CONS, int PROD, int MODE, int SCHED = 0>
void run(const char *name, int wgs_per_cu, int rnd) {
    const int blocks = 256 * wgs_per_cu;
    const size_t lds = sizeof(float) * 2 * BK * (AS + BS);
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&lab<CONS, PROD, MODE, SCHED>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    int occ = 0;
    (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, lab<CONS, PROD, MODE, SCHED>, (CONS + PROD) * 64, lds);
    // waves per SIMD doing MFMA = CONS * wgs_per_cu / 4; size the run to ~12 ms
    const double per_iter_us = 64.0 * 64.0 * (CONS * wgs_per_cu / 4.0) / 2400.0;
    const int iters = (int)(12000.0 / per_iter_us);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    float best = 1e30f, clk = 0;
    for (int rep = 0; rep < 3; rep++) {
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL((lab<CONS, PROD, MODE, SCHED>), dim3(blocks), dim3((CONS + PROD) * 64), lds, 0, d_out, d_src, iters, rnd);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        float h[2]; (void)hipMemcpy(h, d_out, 8, hipMemcpyDeviceToHost);
        if (rep && ms < best) { best = ms; clk = h[0] / h[1] * 100.f; }
    }
    const double flops = (double)blocks * CONS * iters * 64 * 4096.0;
    printf("%-58s wg/cu %d (occ %d) %s  %7.2f ms %6.1f TF %5.1f%%  clk %4.0f MHz\n", name, wgs_per_cu, occ, rnd ? "rand " : "const",
           best, flops / best / 1e9, flops / best / 1e9 / 157.3 * 100, clk);
    fflush(stdout);
}

// ---- wide-fragment variant: per-wave accumulator block (32*TM) x (32*TN); the LDS tiles are laid out [k][l31][TM] / [k][l31][TN]
// so that ONE ds_read_b64 / ds_read_b128 (16-bit immediate offset, no address arithmetic) brings all the A / B values a
// lane needs for a k-pair: 2 LDS instructions per TM*TN MFMAs.
template <int TM, int TN> struct Frag;
template <> struct Frag<2, 2> { typedef float2 A; typedef float2 B; };
template <> struct Frag<2, 4> { typedef float2 A; typedef float4 B; };
template <> struct Frag<4, 4> { typedef float4 A; typedef float4 B; };
__device__ __forceinline__ float fget(const float2 &v, int i) { return i ? v.y : v.x; }
__device__ __forceinline__ float fget(const float4 &v, int i) { return i == 0 ? v.x : i == 1 ? v.y : i == 2 ? v.z : v.w; }

template <int CONS, int PROD, int TM, int TN, int WM_WAVES>
__global__ void __launch_bounds__((CONS + PROD) * 64) labw(float *out, const float *src, int iters, int rnd) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    constexpr int WN_WAVES = CONS / WM_WAVES;
    constexpr int PA = 32 * TM + 4, PB = 32 * TN + 4;          // floats per k row of one wave-block (padded)
    constexpr int A_FLOATS = WM_WAVES * BK * PA, B_FLOATS = WN_WAVES * BK * PB;
    float *As = lds, *Bs = lds + 2 * A_FLOATS;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < 2 * (A_FLOATS + B_FLOATS); i += blockDim.x)
        lds[i] = rnd ? src[(i * 7 + blockIdx.x) & 0xFFFFF] : 1.0f + i * 1e-6f;
    __syncthreads();
    unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    if (PROD > 0 && wave >= CONS) {  // producer role: 16 global loads + transposing stores of a (WM_WAVES*32*TM) x 32 A tile, b128 stores of B
        const int ptid = tid - CONS * 64;
        constexpr int NLD = PROD > 0 ? (A_FLOATS + B_FLOATS) / 4 / ((PROD > 0 ? PROD : 1) * 64) + 1 : 1;
        float4 r[NLD];
        for (int j = 0; j < NLD; j++) r[j] = make_float4(1.f + j, 2.f, 3.f, 4.f);
        const float4 *g = reinterpret_cast<const float4 *>(src);
        for (int it = 0; it < iters; it++) {
            const int buf = (it + 1) & 1;
#pragma unroll
            for (int j = 0; j < NLD; j++) {
                float *dst = (j & 1) ? Bs + buf * B_FLOATS : As + buf * A_FLOATS;
                const int lim = (j & 1) ? B_FLOATS : A_FLOATS;
                const int o = ((j / 2) * PROD * 64 + ptid) * 4;
                if (j & 1) { if (o + 3 < lim) *reinterpret_cast<float4 *>(dst + o) = r[j]; }
                else if (o + 3 * 33 < lim) { dst[o / 4] = r[j].x; dst[o / 4 + 33] = r[j].y; dst[o / 4 + 66] = r[j].z; dst[o / 4 + 99] = r[j].w; }
            }
#pragma unroll
            for (int j = 0; j < NLD; j++) r[j] = g[(blockIdx.x * 977 + it * 4099 + j * 256 + ptid) & 0x3FFFF];  // coalesced 16 B per lane, L2-resident 4 MB
            __syncthreads();
        }
        if (r[0].x == 12345.678f) out[0] = r[1].y;
        return;
    }
    const int wm = wave % WM_WAVES, wn = wave / WM_WAVES, l31 = lane & 31, lh = lane >> 5;
    f32x16 acc[TM][TN];
    for (int i = 0; i < TM; i++) for (int j = 0; j < TN; j++) for (int r = 0; r < 16; r++) acc[i][j][r] = 0.f;
    typedef typename Frag<TM, TN>::A FA;
    typedef typename Frag<TM, TN>::B FB;
    for (int it = 0; it < iters; it++) {
        const int cur = it & 1;
        const float *as = As + cur * A_FLOATS + wm * BK * PA + lh * PA + l31 * TM;
        const float *bs = Bs + cur * B_FLOATS + wn * BK * PB + lh * PB + l31 * TN;
        FA av[2];
        FB bv[2];
        av[0] = *reinterpret_cast<const FA *>(as);
        bv[0] = *reinterpret_cast<const FB *>(bs);
#pragma unroll
        for (int kk = 0; kk < 16; kk++) {
            if (kk + 1 < 16) {
                av[(kk + 1) & 1] = *reinterpret_cast<const FA *>(as + (kk + 1) * 2 * PA);
                bv[(kk + 1) & 1] = *reinterpret_cast<const FB *>(bs + (kk + 1) * 2 * PB);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < TM; i++)
#pragma unroll
                for (int j = 0; j < TN; j++)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fget(av[kk & 1], i), fget(bv[kk & 1], j), acc[i][j], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
        __syncthreads();
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    float s = 0.f;
    for (int i = 0; i < TM; i++) for (int j = 0; j < TN; j++) for (int r = 0; r < 16; r++) s += acc[i][j][r];
    out[16 + (blockIdx.x * 1024 + tid) % (512 * 512)] = s;
    if (tid == 0 && blockIdx.x == 0) { out[0] = (float)(t1 - t0); out[1] = (float)(r1 - r0); }
}

template <int CONS, int PROD, int TM, int TN, int WM_WAVES>
void runw(const char *name, int wgs_per_cu, int rnd) {
    const int blocks = 256 * wgs_per_cu;
    constexpr int PA = 32 * TM + 4, PB = 32 * TN + 4;
    const size_t lds = sizeof(float) * 2 * (WM_WAVES * BK * PA + (CONS / WM_WAVES) * BK * PB);
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&labw<CONS, PROD, TM, TN, WM_WAVES>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    int occ = 0;
    (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, labw<CONS, PROD, TM, TN, WM_WAVES>, (CONS + PROD) * 64, lds);
    const double per_iter_us = 16.0 * TM * TN * 64.0 * (CONS * wgs_per_cu / 4.0) / 2400.0;
    const int iters = (int)(12000.0 / per_iter_us);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    float best = 1e30f, clk = 0;
    for (int rep = 0; rep < 3; rep++) {
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL((labw<CONS, PROD, TM, TN, WM_WAVES>), dim3(blocks), dim3((CONS + PROD) * 64), lds, 0, d_out, d_src, iters, rnd);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        float h[2]; (void)hipMemcpy(h, d_out, 8, hipMemcpyDeviceToHost);
        if (rep && ms < best) { best = ms; clk = h[0] / h[1] * 100.f; }
    }
    const double flops = (double)blocks * CONS * iters * 16.0 * TM * TN * 4096.0;
    printf("%-58s wg/cu %d (occ %d, lds %zu) %s %7.2f ms %6.1f TF %5.1f%%  clk %4.0f MHz\n", name, wgs_per_cu, occ, lds,
           rnd ? "rand " : "const", best, flops / best / 1e9, flops / best / 1e9 / 157.3 * 100, clk);
    fflush(stdout);
}

// ---- what does each kind of producer instruction cost the MFMA waves?  8 consumer waves (64x128 each, wide reads) + 4 probe
// waves that issue PN instructions of ONE kind per K step and then join the barrier.
//   PK: 0 nothing | 1 v_add_u32 | 2 ds_write_b32 | 3 ds_write_b128 | 4 buffer_load_dwordx4 -> VGPR | 5 buffer_load_dwordx4 -> LDS (DMA)
template <int PK, int PN>
__global__ void __launch_bounds__(768) labp(float *out, const float *src, int iters, int rnd) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    constexpr int CONS = 8, TM = 2, TN = 4, WM_WAVES = 4, WN_WAVES = 2;
    constexpr int PA = 32 * TM + 4, PB = 32 * TN + 4;
    constexpr int A_FLOATS = WM_WAVES * BK * PA, B_FLOATS = WN_WAVES * BK * PB;
    float *As = lds, *Bs = lds + 2 * A_FLOATS;
    float *scratch = lds + 2 * (A_FLOATS + B_FLOATS);   // 16 KB the probes write into
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < 2 * (A_FLOATS + B_FLOATS); i += blockDim.x)
        lds[i] = rnd ? src[(i * 7 + blockIdx.x) & 0xFFFFF] : 1.0f + i * 1e-6f;
    __syncthreads();
    unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    if (wave >= CONS) {
        const int ptid = tid - CONS * 64;
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(src), 0, 1 << 22, 0x00020000);
        unsigned v[8];
        for (int j = 0; j < 8; j++) v[j] = ptid * 3 + j;
        float4 r[4];
        for (int j = 0; j < 4; j++) r[j] = make_float4(1.f + j, 2.f, 3.f, 4.f);
        for (int it = 0; it < iters; it++) {
            unsigned base = ((blockIdx.x * 977 + it * 4099) & 0x3FFF) * 16;
#pragma unroll
            for (int n = 0; n < PN; n++) {
                if (PK == 1) { v[n & 7] += v[(n + 3) & 7] ^ (unsigned)it; asm volatile("" : "+v"(v[n & 7])); }
                if (PK == 2) scratch[(n & 15) * 256 + ptid] = r[n & 3].x;
                if (PK == 3) *reinterpret_cast<float4 *>(scratch + ((n & 3) * 256 + ptid) * 4) = r[n & 3];
                if (PK == 4) {
                    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
                    u32x4 q = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)(base + (n * 256 + ptid) * 16), 0, 0);
                    r[n & 3].x = __uint_as_float(q.x); r[n & 3].y = __uint_as_float(q.y); r[n & 3].z = __uint_as_float(q.z); r[n & 3].w = __uint_as_float(q.w);
                }
                if (PK == 5)
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void *)(scratch + (wave - CONS) * 1024 + (n & 3) * 256),
                                                             16, (int)(base + (n * 256 + ptid) * 16), 0, 0, 0);
            }
            if (PK == 4) { asm volatile("" :: "v"(r[0].x), "v"(r[1].x), "v"(r[2].x), "v"(r[3].x)); }
            __syncthreads();
        }
        if (v[0] == 0x12345u && r[0].x == 3.25f) out[1] = v[1] + r[1].y;
        return;
    }
    const int wm = wave % WM_WAVES, wn = wave / WM_WAVES, l31 = lane & 31, lh = lane >> 5;
    f32x16 acc[TM][TN];
    for (int i = 0; i < TM; i++) for (int j = 0; j < TN; j++) for (int r = 0; r < 16; r++) acc[i][j][r] = 0.f;
    for (int it = 0; it < iters; it++) {
        const int cur = it & 1;
        const float *as = As + cur * A_FLOATS + wm * BK * PA + lh * PA + l31 * TM;
        const float *bs = Bs + cur * B_FLOATS + wn * BK * PB + lh * PB + l31 * TN;
        float2 av[2];
        float4 bv[2];
        av[0] = *reinterpret_cast<const float2 *>(as);
        bv[0] = *reinterpret_cast<const float4 *>(bs);
#pragma unroll
        for (int kk = 0; kk < 16; kk++) {
            if (kk + 1 < 16) {
                av[(kk + 1) & 1] = *reinterpret_cast<const float2 *>(as + (kk + 1) * 2 * PA);
                bv[(kk + 1) & 1] = *reinterpret_cast<const float4 *>(bs + (kk + 1) * 2 * PB);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < TM; i++)
#pragma unroll
                for (int j = 0; j < TN; j++)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fget(av[kk & 1], i), fget(bv[kk & 1], j), acc[i][j], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
        __syncthreads();
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    float s = 0.f;
    for (int i = 0; i < TM; i++) for (int j = 0; j < TN; j++) for (int r = 0; r < 16; r++) s += acc[i][j][r];
    out[16 + (blockIdx.x * 1024 + tid) % (512 * 512)] = s;
    if (tid == 0 && blockIdx.x == 0) { out[0] = (float)(t1 - t0); out[1] = (float)(r1 - r0); }
}

template <int PK, int PN>
void runp(const char *name) {
    const int blocks = 256;
    constexpr int PA = 32 * 2 + 4, PB = 32 * 4 + 4;
    const size_t lds = sizeof(float) * 2 * (4 * BK * PA + 2 * BK * PB) + 16384;
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&labp<PK, PN>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    const double per_iter_us = 16.0 * 8 * 64.0 * 2 / 2400.0;
    const int iters = (int)(12000.0 / per_iter_us);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    float best = 1e30f, clk = 0;
    for (int rep = 0; rep < 3; rep++) {
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL((labp<PK, PN>), dim3(blocks), dim3(768), lds, 0, d_out, d_src, iters, 1);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        float h[2]; (void)hipMemcpy(h, d_out, 8, hipMemcpyDeviceToHost);
        if (rep && ms < best) { best = ms; clk = h[0] / h[1] * 100.f; }
    }
    const double flops = (double)blocks * 8 * iters * 16.0 * 8 * 4096.0;
    // cycles of MFMA-pipe time lost per probe instruction: a K step is 2 waves/SIMD x 128 MFMAs x 64 cycles = 16384 cycles
    const double frac = flops / best / 1e9 / 157.3;
    printf("%-44s PN %3d  %7.2f ms %6.1f TF %5.1f%%  clk %4.0f MHz\n", name, PN, best, flops / best / 1e9, frac * 100, clk);
    fflush(stdout);
}

int main() {
    (void)hipMalloc(&d_out, (16 + 512 * 512) * 4);
    (void)hipMalloc(&d_src, (1 << 20) * 4);
    float *h = (float *)malloc((1 << 20) * 4);
    srand(1);
    for (int i = 0; i < (1 << 20); i++) h[i] = (float)rand() / RAND_MAX * 2.f - 1.f;
    (void)hipMemcpy(d_src, h, (1 << 20) * 4, hipMemcpyHostToDevice);
    runp<0, 0>("8 cons 64x128 + 4 idle probe waves");
    runp<1, 32>("+ v_add_u32 per probe wave and K step");
    runp<1, 128>("+ v_add_u32");
    runp<2, 32>("+ ds_write_b32");
    runp<2, 128>("+ ds_write_b32");
    runp<3, 16>("+ ds_write_b128");
    runp<3, 64>("+ ds_write_b128");
    runp<4, 16>("+ buffer_load_dwordx4 -> VGPR");
    runp<4, 64>("+ buffer_load_dwordx4 -> VGPR");
    runp<5, 16>("+ buffer_load_dwordx4 -> LDS (DMA)");
    runp<5, 64>("+ buffer_load_dwordx4 -> LDS (DMA)");
    return 0;
}

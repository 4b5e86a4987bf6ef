// bf16 filter-gradient kernel on LDS-DMA, included by conv_bf16.hip (round 5): dw[k][n] = sum_m x[m + tap][k] dy[m][n] for
// bf16-STORED x and dy, fp32 sums, in the mould of this round's forward / data-gradient kernel (igemm_b16t.h; what its lab measured:
// v_mfma_f32_16x16x32_bf16 holds a higher clock than 32x32x16, and LDS-DMA instructions issued by dedicated waves cost the MFMA waves
// nothing).  The round-1 kernel (wgrad_bf16_kernel: 8 MFMA waves on 32x32x16 + 4 register-staged producer waves, two LDS stages of 32
// pixels) runs the big layers at 53 - 57 % of the dense matrix rate and stays for everything this kernel does not take.
//
//   * 256 (k) x 256 (n) filter tiles, the reduction runs over pixels, 32 per K step; 12 waves, one workgroup per CU: 8 MFMA waves
//     (4 along k x 2 along n, 64 x 128 accumulators each as 4 x 8 tiles of the 16 x 16 MFMA) + 4 DMA waves.
//   * both operands have the reduction index (the pixel) as their SLOW memory index, so a stage holds them as they lie in memory --
//     [32 pixels][256 channels] bf16 = 512 B per pixel row, 16 KB per operand, 32 KB per stage, ring of 5 = all 160 KB, fetched 3
//     steps ahead -- and the MFMA fragments (16 channels x 32 pixels, the pixels along the instruction's k) come out of it with the
//     transposing read ds_read_b64_tr_b16: a 16-lane group reads 4 pixel rows x 16 channels and receives them channel-major, two
//     reads per fragment.  A k tile lies inside ONE filter tap (Cin % 256 == 0), so the gather of x is a constant pixel shift and a
//     pixel's 256 channels are contiguous: a DMA instruction (64 lanes x 16 B) moves two pixel rows.
//   * bank conflicts: the four groups of a wave read the SAME 16 channels of different pixels, and the two groups of a 32-lane half
//     go through the LDS together: 8 pixel rows x 32 B must land on 8 different 32-byte bank segments, which rows of 512 B (two bank
//     rows each) never do by themselves.  (1) The 32-byte unit u (16 channels) of pixel row P is stored at unit position
//     u ^ ((P >> 1) & 7) -- the DMA lanes fetch their 16-byte chunks in that order -- and (2) the pixel <-> MFMA-k-slot assignment,
//     which is free as long as both operands share it, gives group g, lane quad q the pixels 2 (4 g + q) + r (r = 0, 1: the two reads)
//     -- so the 8 rows of a half-wave read have 8 different (P >> 1) & 7.  No padding: the stage stays 32 KB.
//   * MFMA waves: the forward kernel's schedule -- 8 columns (dy fragments) x 4 MFMAs per step, B fragments through a 3-deep rotation,
//     the barrier in front of column 4, the next step's fragments behind it.
//   * DMA waves: per step 4 + 4 instructions each; the pixel -> (row, column) arithmetic of the gather (two multiply-high divisions
//     per instruction) is theirs.  Pixels past the split's end and taps outside the image carry the out-of-range bit: zeros.
//   * split-M exactly as the round-1 kernel (fp32 partial filter tiles, summed in a fixed order by slab_sum): bit-reproducible.
// Requirements: x and dy stored bf16, Cin % 256 == 0 (conv3_2 .. fc7), Cout % 8 == 0.
#pragma once

namespace {

constexpr int WT_PX = 32;                        // pixels per K step
constexpr int WT_OPER = WT_PX * 512;             // one operand's step in LDS: 32 pixel rows x 256 channels x 2 B
constexpr int WT_STAGE = 2 * WT_OPER;
constexpr int WT_NBUF = 5, WT_AHEAD = 3;
constexpr int WT_THREADS = 12 * 64;
constexpr size_t WT_LDS = (size_t)WT_NBUF * WT_STAGE;
static_assert(WT_LDS == 160 * 1024, "the ring fills the LDS");

__device__ __forceinline__ bt_bf16x8 wt_frag(const char *p) {   // two transposing reads: pixels P (r = 0) and P + 1 (r = 1) of the lane's pair
    typedef __attribute__((address_space(3))) bf16x4 lds_bf16x4;
    const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4 *)(p));
    const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4 *)(p + 512));
    return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
}

__global__ void __launch_bounds__(WT_THREADS, 3) wgrad_b16t_kernel(WgradBf16Args a) {
    constexpr int NBUF = WT_NBUF, AHEAD = WT_AHEAD, STAGE = WT_STAGE, OPER = WT_OPER;
    extern __shared__ __attribute__((aligned(16))) char ldsb[];

    // XCD-aware block order as in wgrad_bf16_kernel: split-major, inside a split groups of GK k tiles x all n tiles
    int tile_k, tile_n, split;
    {
        const int ntile = a.tiles_k * a.tiles_n;
        const int nblk = ntile * a.splits;
        const int bid = blockIdx.x;
        const int xcd = bid & 7, q = nblk >> 3, r = nblk & 7;
        int t = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
        split = t / ntile;
        t -= split * ntile;
        constexpr int GK = 4;
        const int per_group = GK * a.tiles_n;
        const int g = t / per_group;
        const int first_k = g * GK;
        const int gsz = (a.tiles_k - first_k < GK) ? (a.tiles_k - first_k) : GK;
        const int in_g = t - g * per_group;
        tile_k = first_k + in_g % gsz;
        tile_n = in_g / gsz;
    }
    const int k0 = tile_k * 256, n0 = tile_n * 256;
    const int m_begin = split * a.rows_per_split;
    int m_end = m_begin + a.rows_per_split;
    if (m_end > a.M) m_end = a.M;
    const int nsteps = (m_end > m_begin) ? (m_end - m_begin + WT_PX - 1) / WT_PX : 0;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);

    if (wave >= 8) {
        // ======================================= DMA waves =======================================
        constexpr int IPW = 4;   // pairs of pixel rows per DMA wave, operand and step
        const int lw = wave - 8;
        const int tap = k0 / a.Cin, c0 = k0 - tap * a.Cin;
        const int dh = (tap / a.KW) * a.dil - a.pad_h, dw = (tap % a.KW) * a.dil - a.pad_w;
        const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(a.x), 0, (int)a.x_bytes, 0x00020000);
        const __amdgpu_buffer_rsrc_t yr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(a.dy), 0, (int)a.y_bytes, 0x00020000);
        constexpr unsigned kOOB = 0x80000000u;
        const int half = lane >> 5, cpos = lane & 31;
        // lane -> pixel of the step and 16-byte chunk of its row: pair p = lw * IPW + j, pixel 2 p + half, stored chunk position cpos
        // holds chunk cpos ^ ((p & 7) << 1)
        int px[IPW];
        unsigned xcol[IPW], ycol[IPW];
        bool yok[IPW];
#pragma unroll
        for (int j = 0; j < IPW; j++) {
            const int p = lw * IPW + j;
            const int c = cpos ^ ((p & 7) << 1);
            px[j] = 2 * p + half;
            xcol[j] = (unsigned)(c0 + c * 8) * 2u;
            ycol[j] = (unsigned)(n0 + c * 8) * 2u;
            yok[j] = n0 + c * 8 < a.N;
        }
        const int x_shift = (dh * a.W + dw) * a.Cin * 2;   // bytes from a pixel to the pixel its tap reads
        // BiasAddGrad rides along: the DMA waves add up the dy stage that has just landed -- wave lw owns the 16-byte chunks 8 lw ..
        // 8 lw + 7 (64 columns), lane -> chunk lane & 7 and the pixels 4 (lane >> 3) + r of the step (four ds_read_b128).  The DMA
        // waves are NOT idle enough to do that every step (measured: the launch 15 % slower when the workgroups of the first k tile
        // did it alone -- the MFMA waves wait for them at the barrier), so the workgroups that share a pixel range and columns, one
        // per k tile, divide the steps: step st belongs to k tile st % tiles_k, a workgroup's partial sums go to row (split, tile_k)
        const bool do_bias = a.bias_out != nullptr;
        int bias_st = tile_k;
        const int bchunk = lw * 8 + (lane & 7), bpg = lane >> 3;
        float bs[8];
#pragma unroll
        for (int e = 0; e < 8; e++) bs[e] = 0.f;
        auto bias_step = [&](int off) {
            const char *ys = ldsb + off + OPER + (4 * bpg) * 512;
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const int pos = bchunk ^ (((2 * bpg + (r >> 1)) & 7) << 1);   // (the stage's unit swizzle: pair index (P >> 1) & 7)
                const u32x4 v = *reinterpret_cast<const u32x4 *>(ys + r * 512 + pos * 16);
                const unsigned w4[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                for (int e = 0; e < 4; e++) {
                    bs[2 * e] += __uint_as_float(w4[e] << 16);
                    bs[2 * e + 1] += __uint_as_float(w4[e] & 0xffff0000u);
                }
            }
        };
        auto fill = [&](int off, int st) {
            const int mb = m_begin + st * WT_PX;
            char *xs = ldsb + off + lw * (IPW * 1024), *ys = ldsb + off + OPER + lw * (IPW * 1024);
#pragma unroll
            for (int j = 0; j < IPW; j++) {
                const int m = mb + px[j];
                const unsigned t2 = div_magic((unsigned)m, a.magic_w);
                const int ow = m - (int)t2 * a.W;
                const int oh = (int)(t2 - div_magic(t2, a.magic_h) * (unsigned)a.H);
                const bool ok = m < m_end && (unsigned)(oh + dh) < (unsigned)a.H && (unsigned)(ow + dw) < (unsigned)a.W;
                const unsigned off_x = (unsigned)((long long)m * a.Cin * 2 + x_shift) + xcol[j];
                __builtin_amdgcn_raw_ptr_buffer_load_lds(xr, (lds_void_t *)(xs + j * 1024), 16, (int)(ok ? off_x : kOOB), 0, 0, 0);
            }
#pragma unroll
            for (int j = 0; j < IPW; j++) {
                const int m = mb + px[j];
                const unsigned off_y = (unsigned)m * (unsigned)(a.N * 2) + ycol[j];
                __builtin_amdgcn_raw_ptr_buffer_load_lds(yr, (lds_void_t *)(ys + j * 1024), 16, (int)((m < m_end && yok[j]) ? off_y : kOOB), 0, 0, 0);
            }
        };
#pragma unroll
        for (int s = 0; s < AHEAD; s++) fill(s * STAGE, s);   // (steps past the end: every pixel >= m_end -> zeros)
        wait_vmcnt<(AHEAD - 1) * 2 * IPW>();
        __builtin_amdgcn_s_barrier();   // step 0 has landed
        int dst = AHEAD * STAGE, cur = 0;
        for (int st = 0; st < nsteps; st++) {
            fill(dst, st + AHEAD);
            if (do_bias && st == bias_st) {   // (stage st landed before the previous barrier; its slot is refilled two iterations from now)
                asm volatile("" ::: "memory");
                bias_step(cur);
                bias_st += a.tiles_k;
            }
            wait_vmcnt<(AHEAD - 1) * 2 * IPW>();   // step st + 1 has landed
            __builtin_amdgcn_s_barrier();
            dst = dst == (NBUF - 1) * STAGE ? 0 : dst + STAGE;
            cur = cur == (NBUF - 1) * STAGE ? 0 : cur + STAGE;
        }
        wait_vmcnt<0>();
        if (do_bias) {   // the eight pixel groups of the wave (lane bits 3 .. 5), in a fixed order
#pragma unroll
            for (int sft = 8; sft < 64; sft <<= 1)
#pragma unroll
                for (int e = 0; e < 8; e++) bs[e] += __shfl_xor(bs[e], sft);
            const int col = n0 + bchunk * 8;
            if (lane < 8 && col < a.N) {
                float *bo = a.bias_out + ((long long)split * a.tiles_k + tile_k) * a.N + col;
                *reinterpret_cast<float4 *>(bo) = make_float4(bs[0], bs[1], bs[2], bs[3]);
                *reinterpret_cast<float4 *>(bo + 4) = make_float4(bs[4], bs[5], bs[6], bs[7]);
            }
        }
        return;
    }

    // ======================================= MFMA waves =======================================
    const int wm = wave & 3, wn = wave >> 2;
    const int g = lane >> 4, q = (lane >> 2) & 3, p = lane & 3;
    const int pair = 4 * g + q, fq = pair & 7;
    // byte address (inside a stage) of the lane's part of fragment i: pixel row 2 pair (+ 1 for the second read: + 512), unit
    // (U + i) ^ fq.  U (4 wm resp. 8 wn), i and fq XOR bit-wise (U's bits lie above i's), so the address is ONE base register XOR
    // (i << 5): twelve ready-made addresses cost the MFMA loop 29 spilled registers, a v_xor per fragment costs it nothing
    const int ax = ((2 * pair) * 512 + p * 8) ^ (((wm * 4) ^ fq) << 5);
    const int bx = (OPER + (2 * pair) * 512 + wn * 256 + p * 8) ^ (fq << 5);
    bt_f32x4 acc[4][8];
#pragma unroll
    for (int i = 0; i < 4; i++)
#pragma unroll
        for (int j = 0; j < 8; j++)
#pragma unroll
            for (int r = 0; r < 4; r++) acc[i][j][r] = 0.f;
    bt_bf16x8 av[4], bv[3];
    __builtin_amdgcn_s_barrier();   // step 0 has landed
    asm volatile("" ::: "memory");
#define WT_SB __builtin_amdgcn_sched_barrier(0);
#define WT_RDA(I, BASE) { av[I] = wt_frag(ldsb + ((((BASE) + ax)) ^ ((I) << 5))); WT_SB }
#define WT_RDB(S, J, BASE) { bv[S] = wt_frag(ldsb + ((((BASE) + bx)) ^ ((J) << 5))); WT_SB }
#define WT_MF(I, J, S) { acc[I][J] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av[I], bv[S], acc[I][J], 0, 0, 0); WT_SB }
    if (nsteps > 0) {
        WT_RDA(0, 0) WT_RDA(1, 0) WT_RDA(2, 0) WT_RDA(3, 0) WT_RDB(0, 0, 0) WT_RDB(1, 1, 0)
    }
    // (igemm_b16t.h: column J = 4 MFMAs on B fragment J in rotation slot (J + O) % 3, fragment J + 2 requested in front of them -- from
    // the next stage once J + 2 >= 8, behind the barrier in front of column 4; O = 0, 2, 1 over three consecutive steps)
#define WT_COL(J, O)                                                                                                                   \
        if ((J) == 4) { __builtin_amdgcn_s_barrier(); asm volatile("" ::: "memory"); WT_SB }                                           \
        if ((J) + 2 < 8) { WT_RDB(((J) + 2 + (O)) % 3, (J) + 2, cur) } else { WT_RDB(((J) + 2 + (O)) % 3, (J) + 2 - 8, nx) }           \
        WT_MF(0, J, ((J) + (O)) % 3) if ((J) == 7) { WT_RDA(0, nx) }                                                                   \
        WT_MF(1, J, ((J) + (O)) % 3) if ((J) == 7) { WT_RDA(1, nx) }                                                                   \
        WT_MF(2, J, ((J) + (O)) % 3) if ((J) == 7) { WT_RDA(2, nx) }                                                                   \
        WT_MF(3, J, ((J) + (O)) % 3) if ((J) == 7) { WT_RDA(3, nx) }
#define WT_STEP(O)                                                                                                  \
        {                                                                                                           \
            const int nx = cur == (NBUF - 1) * STAGE ? 0 : cur + STAGE;                                             \
            WT_SB                                                                                                   \
            WT_COL(0, O) WT_COL(1, O) WT_COL(2, O) WT_COL(3, O) WT_COL(4, O) WT_COL(5, O) WT_COL(6, O) WT_COL(7, O) \
            cur = nx;                                                                                               \
        }
    {
        int cur = 0;
        for (int st = 0; st < nsteps; st += 3) {
            WT_STEP(0)
            if (st + 1 < nsteps) WT_STEP(2)
            if (st + 2 < nsteps) WT_STEP(1)
        }
    }
#undef WT_STEP
#undef WT_COL
#undef WT_MF
#undef WT_RDB
#undef WT_RDA
#undef WT_SB
    // ---- the fp32 tile (or this split's partial tile): C/D layout of the 16 x 16 MFMA -- column = lane & 15, row = 4 (lane >> 4) + r
    float *out = a.out + (long long)split * a.K * a.N;
    const int l15 = lane & 15, lq = lane >> 4;
#pragma unroll
    for (int i = 0; i < 4; i++)
#pragma unroll
        for (int j = 0; j < 8; j++) {
            const int col = n0 + wn * 128 + j * 16 + l15;
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const int row = k0 + wm * 64 + i * 16 + 4 * lq + r;
                if (col < a.N && row < a.K) out[(long long)row * a.N + col] = acc[i][j][r];
            }
        }
}

inline bool wgrad_b16t_ok(const emadapt_conv_desc *d, int x_bf16, int dy_bf16) {
    return x_bf16 && dy_bf16 && d->Cin % 256 == 0 && d->Cout % 8 == 0 && d->Cout >= 128;
}

int launch_wgrad_b16t(const WgradBf16Args &a, hipStream_t stream) {
    static DynLdsAttr attr;
    const int arc = attr.ensure(reinterpret_cast<const void *>(&wgrad_b16t_kernel), WT_LDS);
    if (arc) return arc;
    const long long nblk = (long long)a.tiles_k * a.tiles_n * a.splits;
    if (nblk > 0x7fffffffLL) return EMADAPT_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(wgrad_b16t_kernel, dim3((unsigned)nblk), dim3(WT_THREADS), WT_LDS, stream, a);
    return launch_status();
}

}  // namespace

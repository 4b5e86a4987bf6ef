// Second-generation fp32 implicit-GEMM kernel (forward and data gradient), included by conv_igemm.hip.
//
// Same arithmetic as igemm_kernel / igemm_ws_kernel (v_mfma_f32_32x32x2_f32, k-ordered fp32 FMA chains, K order =
// (channel chunk, tap), bounds-checked gathers, fused epilogue, split-K slabs), re-laid-out around what the matrix pipe
// of gfx950 actually tolerates (profiles/r02_mfma_loop_lab.txt):
//
//   * the fp32 MFMA runs on the vector ALU's datapath: every VALU / LDS instruction a MFMA wave issues costs ~10 cycles
//     of matrix time, and VALU issued by ANY wave of the SIMD costs its issue cycles; LDS stores and global loads issued
//     by other waves are free.  The round-1 loop (64x64 per wave, two ds_read2_b32 + two v_add per four MFMAs) tops out
//     at 84-90 %; one ds_read_b64 + one ds_read_b128 per EIGHT MFMAs with immediate offsets reaches 97.5 %.
//   * so every MFMA wave owns a 64 x 128 accumulator block (TM = 2, TN = 4 tiles of 32x32, 128 VGPRs) and the LDS tiles
//     are interleaved so that ONE wide read returns all the values a lane needs for a k-pair:
//         A tile  [wave-row block][k][l31][TM]   -> ds_read_b64  (rows l31 and l31 + 32 of the block, same k)
//         B tile  [wave-col block][k][l31*TN..]  -> ds_read_b128 (four ADJACENT output columns: the lane's column
//                                                   in the four 32x32 tiles is l31*4 + j, not l31 + 32*j)
//     The consumers' instruction stream is LDS reads with 16-bit immediate offsets and MFMAs, nothing else.
//   * the column interleave makes the accumulators of a lane four consecutive outputs of a row: the epilogue is direct
//     16-byte stores (512 contiguous bytes per row and wave), no staging through LDS.
//   * the producers (separate waves: gather, transpose into the interleaved tiles, plain ds_write) carry ~3 VALU per
//     load: the tap-validity of each of a thread's rows is a bit mask computed once per tile, the per-step tap / channel
//     offsets are wave-uniform (SALU).
//
// Configurations <WMW, WNW, BK, PW>: consumer waves WMW x WNW, producer waves PW
//   <4, 2, 32, 4>  256 x 256 x 32, 12 waves, one workgroup per CU   (fc6 / fc7: N >= 1024)
//   <2, 2, 16, 2>  128 x 256 x 16,  6 waves, two workgroups per CU  (conv3 .. conv5: 256 <= N < 1024)
// Requirements (plan_launches falls back to the round-1 kernels otherwise): Cx % BK == 0, KH*KW <= 32.
#pragma once
#include "igemm_common.h"

namespace {

template <int WMW, int WNW, int BK, int PW, bool BT>
__global__ void __launch_bounds__((WMW * WNW + PW) * 64, 3) igemm2_kernel(IgemmArgs a) {
    constexpr int CONS = WMW * WNW;          // consumer (MFMA) waves
    constexpr int BM = WMW * 64, BN = WNW * 128;
    constexpr int TM = 2, TN = 4;
    constexpr int PA = 64 + 4, PB = 128 + 4;  // floats per k row of one wave block (16-byte aligned rows)
    constexpr int A_BUF = WMW * BK * PA, B_BUF = WNW * BK * PB;
    constexpr int NKK = BK / 2;
    constexpr int NPT = PW * 64;              // producer threads
    constexpr int KQ = BK / 4;                // float4 quads along k
    constexpr int A_RSTEP = NPT / KQ;         // rows covered by one pass of the producers
    constexpr int A_N = BM / A_RSTEP;
    constexpr int QN = BN / 4;
    constexpr int B_KSTEP = NPT / QN;
    constexpr int B_N = BT ? (BN / A_RSTEP) : (BK / B_KSTEP);
    static_assert(A_N >= 1 && B_N >= 1 && NPT % KQ == 0 && NPT % QN == 0, "tile / producer shape");

    extern __shared__ __attribute__((aligned(16))) float lds[];
    float *As = lds;                  // [2][WMW][BK][PA]
    float *Bs = lds + 2 * A_BUF;      // [2][WNW][BK][PB]

    // XCD-aware tile order (see igemm_kernel)
    int tile_m, tile_n, split;
    {
        const int ntile = a.tiles_m * a.tiles_n;
        const int nblk = ntile * a.splits;
        const int bid = blockIdx.x;
        const int xcd = bid & 7, q = nblk >> 3, r = nblk & 7;
        int t = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
        split = t / ntile;
        t -= split * ntile;
        constexpr int GM = 8;
        const int per_group = GM * a.tiles_n;
        const int g = t / per_group;
        const int first_m = g * GM;
        const int gsz = (a.tiles_m - first_m < GM) ? (a.tiles_m - first_m) : GM;
        const int in_g = t - g * per_group;
        tile_m = first_m + in_g % gsz;
        tile_n = in_g / gsz;
    }
    const int m0 = a.m_begin + tile_m * BM, n0 = tile_n * BN;
    const int nk_all = a.K / BK;  // Cx % BK == 0: every K step is one tap x one BK-channel chunk
    const int kt0 = (int)((long long)nk_all * split / a.splits);
    const int nk = (int)((long long)nk_all * (split + 1) / a.splits) - kt0;

    if (threadIdx.x >= CONS * 64) {
        // ================================ PRODUCER wavefronts ================================
        const int tid = threadIdx.x - CONS * 64;
        if (a.pace != 99) __builtin_amdgcn_s_setprio(3);
        const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(a.x), 0, (int)a.x_bytes, 0x00020000);
        const __amdgpu_buffer_rsrc_t wr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(a.w), 0, (int)a.w_bytes, 0x00020000);
        constexpr unsigned kOOB = 0x80000000u;
        const int a_kq = tid % KQ, a_r0 = tid / KQ;
        // per row: byte offset of (pixel, channel quad a_kq) and the mask of taps that fall outside the image
        unsigned a_off[A_N], a_bad[A_N];
        int a_dst[A_N];   // float index of (row, k = a_kq*4) inside one A buffer
#pragma unroll
        for (int j = 0; j < A_N; j++) {
            const int r = a_r0 + j * A_RSTEP;
            const int m = m0 + r;
            a_dst[j] = ((r >> 6) * BK + a_kq * 4) * PA + (r & 31) * TM + ((r >> 5) & 1);
            if (m < a.m_end) {
                const int t2 = m / a.W;
                const int oh = t2 % a.H, ow = m - t2 * a.W;
                a_off[j] = (unsigned)(m * a.Cx + a_kq * 4) * 4u;
                unsigned rowbad = 0, colbad = 0;
                const int KH = a.ntaps / a.KW;
                for (int kh = 0; kh < KH; kh++) {
                    const int ih = oh + a.sign * (kh * a.dil - a.pad_h);
                    rowbad |= ((unsigned)ih < (unsigned)a.H ? 0u : 1u) << kh;
                }
                for (int kw = 0; kw < a.KW; kw++) {
                    const int iw = ow + a.sign * (kw * a.dil - a.pad_w);
                    colbad |= ((unsigned)iw < (unsigned)a.W ? 0u : 1u) << kw;
                }
                unsigned bad = 0;
                const unsigned full = a.KW >= 32 ? 0xFFFFFFFFu : ((1u << a.KW) - 1u);
                for (int kh = 0; kh < KH; kh++) bad |= (((rowbad >> kh) & 1u) ? full : colbad) << (kh * a.KW);
                a_bad[j] = bad;
            } else {
                a_off[j] = 0;
                a_bad[j] = 0xFFFFFFFFu;
            }
        }
        // filter loads: per-thread constant part of the offset (bit 31 = column out of range), wave-uniform part per step
        unsigned b_off[B_N];
        int b_dst[B_N];
        if constexpr (!BT) {
            const int b_nq = tid % QN, b_kr = tid / QN;
            const int n = n0 + b_nq * 4;
#pragma unroll
            for (int j = 0; j < B_N; j++) {
                const int kr = b_kr + j * B_KSTEP;
                b_off[j] = ((unsigned)(kr * a.N + n) * 4u) | (n < a.N ? 0u : kOOB);
                b_dst[j] = (((b_nq * 4) >> 7) * BK + kr) * PB + ((b_nq * 4) & 127);
            }
        } else {
            const int b_nr = tid / KQ;
#pragma unroll
            for (int j = 0; j < B_N; j++) {
                const int nl = b_nr + j * A_RSTEP;
                const int n = n0 + nl;
                b_off[j] = ((unsigned)(n * a.Cx + a_kq * 4) * 4u) | (n < a.N ? 0u : kOOB);
                b_dst[j] = ((nl >> 7) * BK + a_kq * 4) * PB + (nl & 127);
            }
        }
        float4 ra[A_N], rb[B_N];
#ifdef EMADAPT_ABLATE_BUILD
        for (int j = 0; j < A_N; j++) ra[j] = make_float4(1.f, 2.f, 3.f, 4.f);
        for (int j = 0; j < B_N; j++) rb[j] = make_float4(1.f, 2.f, 3.f, 4.f);
#endif

        // One K step of tile movement: park tile kt+1 (registers) in LDS buffer `buf`, re-arm every register with its load
        // of tile `kt_next`.  The parts are issued in CH chunks with the producers asleep in between (`pace` x 64 cycles):
        // a burst of 40 LDS stores and 16 loads per wave right after the barrier sits in front of the consumers' first
        // fragment reads and costs the matrix pipe 15 % (ablation, profiles/r02_igemm_v2_ablation.txt); spread over the
        // K step it is hidden.
        constexpr int CH = A_N > B_N ? A_N : B_N;
        auto move_tile = [&](int buf, int kt, bool store, bool load) {
            const unsigned chunk = __umulhi((unsigned)kt, a.magic_nt);
            const unsigned tap_u = a.ntaps > 1 ? (unsigned)kt - chunk * (unsigned)a.ntaps : 0u;
            const unsigned chunk_u = a.ntaps > 1 ? chunk : (unsigned)kt;
            const bool past = (int)chunk_u * BK >= a.Cx;   // K steps past the end are zero-filled
            const unsigned kh = div_magic(tap_u, a.magic_kw);
            const int kw = (int)(tap_u - kh * (unsigned)a.KW);
            const int dh = a.sign * ((int)kh * a.dil - a.pad_h);
            const int dw = a.sign * (kw * a.dil - a.pad_w);
            const unsigned toff = (unsigned)(((dh * a.W + dw) * a.Cx + (int)chunk_u * BK) * 4);
            const unsigned sh = past ? 0u : 31u - tap_u;   // moves this tap's "bad" bit to bit 31
            unsigned woff;
            if constexpr (!BT) woff = (unsigned)(((int)tap_u * a.Cx + (int)chunk_u * BK) * a.N) * 4u;
            else woff = (unsigned)((int)tap_u * a.N * a.Cx + (int)chunk_u * BK) * 4u;
            float *as = As + buf * A_BUF;
            float *bs = Bs + buf * B_BUF;
#pragma unroll
            for (int j = 0; j < CH; j++) {
                if (j < A_N) {
                    if (store && !ABL(2)) {
                        as[a_dst[j] + 0 * PA] = ra[j].x;
                        as[a_dst[j] + 1 * PA] = ra[j].y;
                        as[a_dst[j] + 2 * PA] = ra[j].z;
                        as[a_dst[j] + 3 * PA] = ra[j].w;
                    }
                    if (load && !ABL(1)) {
                        const unsigned bad = past ? kOOB : ((a_bad[j] << sh) & kOOB);
                        ra[j] = buf_ld4(xr, (a_off[j] + toff) | bad);
                    }
                }
                if (j < B_N) {
                    if (store && !ABL(2)) {
                        if constexpr (!BT) {
                            *reinterpret_cast<float4 *>(bs + b_dst[j]) = rb[j];
                        } else {
                            bs[b_dst[j] + 0 * PB] = rb[j].x;
                            bs[b_dst[j] + 1 * PB] = rb[j].y;
                            bs[b_dst[j] + 2 * PB] = rb[j].z;
                            bs[b_dst[j] + 3 * PB] = rb[j].w;
                        }
                    }
                    if (load && !ABL(1)) rb[j] = buf_ld4(wr, past ? kOOB : b_off[j] + woff);
                }
                if (store && load)
                    for (int z = 0; z < a.pace; z++) __builtin_amdgcn_s_sleep(1);
            }
        };
        move_tile(0, kt0, false, true);
        move_tile(0, kt0 + 1 < kt0 + nk ? kt0 + 1 : nk_all, true, true);   // K steps past this split's range load zeros
        __syncthreads();
#ifdef EMADAPT_ABLATE_BUILD
        unsigned long long t_work = 0, t_wait = 0, t0 = __builtin_amdgcn_s_memtime();
#endif
        for (int kt = 0; kt < nk; kt++) {
            move_tile((kt + 1) & 1, kt + 2 < nk ? kt0 + kt + 2 : nk_all, true, true);
#ifdef EMADAPT_ABLATE_BUILD
            unsigned long long t1 = __builtin_amdgcn_s_memtime();
            t_work += t1 - t0;
#endif
            __syncthreads();
#ifdef EMADAPT_ABLATE_BUILD
            t0 = __builtin_amdgcn_s_memtime();
            t_wait += t0 - t1;
#endif
        }
#ifdef EMADAPT_ABLATE_BUILD
        if (ABL(32) && blockIdx.x == 17 && tid == 0)
            printf("producer: %d steps, work %llu cycles/step, barrier wait %llu cycles/step\n", nk, t_work / nk, t_wait / nk);
#endif
        return;
    }

    // ================================ CONSUMER wavefronts ================================
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wm = wave % WMW, wn = wave / WMW;
    const int l31 = lane & 31, lh = lane >> 5;
    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; i++)
#pragma unroll
        for (int j = 0; j < TN; j++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[i][j][r] = 0.f;
    __syncthreads();
#ifdef EMADAPT_ABLATE_BUILD
    unsigned long long t_work = 0, t_wait = 0, t0 = __builtin_amdgcn_s_memtime();
#endif
    for (int kt = 0; kt < nk; kt++) {
        const int cur = kt & 1;
        const float *as = As + cur * A_BUF + (wm * BK + lh) * PA + l31 * TM;
        const float *bs = Bs + cur * B_BUF + (wn * BK + lh) * PB + l31 * TN;
        float2 av[2];
        float4 bv[2];
        av[0] = *reinterpret_cast<const float2 *>(as);
        bv[0] = *reinterpret_cast<const float4 *>(bs);
#pragma unroll
        for (int kk = 0; kk < NKK; kk++) {
            if (kk + 1 < NKK) {
                av[(kk + 1) & 1] = *reinterpret_cast<const float2 *>(as + (kk + 1) * 2 * PA);
                bv[(kk + 1) & 1] = *reinterpret_cast<const float4 *>(bs + (kk + 1) * 2 * PB);
            }
            __builtin_amdgcn_sched_barrier(0);
            const float2 a2 = av[kk & 1];
            const float4 b4 = bv[kk & 1];
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a2.x, b4.x, acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a2.x, b4.y, acc[0][1], 0, 0, 0);
            acc[0][2] = __builtin_amdgcn_mfma_f32_32x32x2f32(a2.x, b4.z, acc[0][2], 0, 0, 0);
            acc[0][3] = __builtin_amdgcn_mfma_f32_32x32x2f32(a2.x, b4.w, acc[0][3], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a2.y, b4.x, acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a2.y, b4.y, acc[1][1], 0, 0, 0);
            acc[1][2] = __builtin_amdgcn_mfma_f32_32x32x2f32(a2.y, b4.z, acc[1][2], 0, 0, 0);
            acc[1][3] = __builtin_amdgcn_mfma_f32_32x32x2f32(a2.y, b4.w, acc[1][3], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
#ifdef EMADAPT_ABLATE_BUILD
        unsigned long long t1 = __builtin_amdgcn_s_memtime();
        t_work += t1 - t0;
#endif
        __syncthreads();
#ifdef EMADAPT_ABLATE_BUILD
        t0 = __builtin_amdgcn_s_memtime();
        t_wait += t0 - t1;
#endif
    }
#ifdef EMADAPT_ABLATE_BUILD
    if (ABL(32) && blockIdx.x == 17 && (tid == 0 || tid == 256))
        printf("consumer wave %d: %d steps, work %llu cycles/step, barrier wait %llu cycles/step\n", wave, nk, t_work / nk, t_wait / nk);
#endif

    // ---- epilogue: a lane holds, for each of its 2 x 16 rows, four consecutive output columns -> 16-byte stores.
    // C/D layout of the 32x32 MFMA: row in tile = (r&3) + 8*(r>>2) + 4*lh, column = l31 (here: column l31*4 + j).
    const int col = n0 + wn * 128 + l31 * TN;
    if (col >= a.N) return;   // N % 4 == 0: the quad is all in or all out
    float *slab = a.splits > 1 ? a.slab + ((long long)split * (a.m_end - a.m_begin) - a.m_begin) * a.N : nullptr;
    float4 bias4 = make_float4(0.f, 0.f, 0.f, 0.f);
    if (!slab && a.epi != EMADAPT_EPI_NONE && a.bias) bias4 = *reinterpret_cast<const float4 *>(a.bias + col);
#pragma unroll
    for (int i = 0; i < TM; i++) {
#pragma unroll
        for (int r = 0; r < 16; r++) {
            const int row = m0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
            if (row >= a.m_end) continue;
            const long long o = (long long)row * a.N + col;
            float v[4] = {acc[i][0][r], acc[i][1][r], acc[i][2][r], acc[i][3][r]};
            if (slab) {
                *reinterpret_cast<float4 *>(slab + o) = make_float4(v[0], v[1], v[2], v[3]);
                continue;
            }
            v[0] += bias4.x; v[1] += bias4.y; v[2] += bias4.z; v[3] += bias4.w;   // (bias4 = 0 when there is no bias)
            if (a.epi == EMADAPT_EPI_BIAS_RELU) {
#pragma unroll
                for (int e = 0; e < 4; e++) v[e] = v[e] > 0.f ? v[e] : 0.f;
                if (a.keep < 1.0f) {
                    unsigned mk = 0;
                    if (a.mask) mk = *reinterpret_cast<const unsigned *>(a.mask + o);   // four keep bytes (o % 4 == 0)
#pragma unroll
                    for (int e = 0; e < 4; e++) {
                        const bool kp = a.mask ? (((mk >> (8 * e)) & 0xFFu) != 0)
                                               : dropout_keep(a.seed, (unsigned long long)(o + e), a.keep);
                        v[e] = kp ? v[e] / a.keep : 0.f;  // tf.nn.dropout divides by keep_prob
                    }
                }
            }
            if (a.gate) {
                const float4 g = *reinterpret_cast<const float4 *>(a.gate + o);
                v[0] = g.x > 0.f ? v[0] * a.scale : 0.f;
                v[1] = g.y > 0.f ? v[1] * a.scale : 0.f;
                v[2] = g.z > 0.f ? v[2] * a.scale : 0.f;
                v[3] = g.w > 0.f ? v[3] * a.scale : 0.f;
            }
            *reinterpret_cast<float4 *>(a.y + o) = make_float4(v[0], v[1], v[2], v[3]);
        }
    }
}

template <int WMW, int WNW, int BK, int PW, bool BT>
int launch_igemm2(IgemmArgs &a, hipStream_t stream) {
    constexpr int BM = WMW * 64, BN = WNW * 128;
    constexpr int THREADS = (WMW * WNW + PW) * 64;
    constexpr size_t lds = sizeof(float) * 2 * (WMW * BK * (64 + 4) + WNW * BK * (128 + 4));
    static DynLdsAttr attr;
    int rc = attr.ensure(reinterpret_cast<const void *>(&igemm2_kernel<WMW, WNW, BK, PW, BT>), lds);
    if (rc) return rc;
    a.tiles_m = (int)ceil_div(a.m_end - a.m_begin, BM);
    a.tiles_n = (int)ceil_div(a.N, BN);
    const long long nblk = (long long)a.tiles_m * a.tiles_n * a.splits;
    if (nblk > 0x7fffffffLL) return EMADAPT_ERR_UNSUPPORTED;
    // producers sleep between the chunks of a K step's tile movement: ~60 % of the step's MFMA time in all
    // (a K step = BK/2 k-pairs x 8 MFMAs x 64 cycles per consumer wave, two consumer waves per SIMD)
    {
        constexpr int CH = (BM * (BK / 4) / (PW * 64)) > 8 ? (BM * (BK / 4) / (PW * 64)) : 8;
        const int force = tune(TUNE_PACE);
        a.pace = force ? (force < 0 ? 0 : force) : (BK / 2) * 8 * 64 * 2 * 6 / 10 / 64 / CH;
    }
    hipLaunchKernelGGL((igemm2_kernel<WMW, WNW, BK, PW, BT>), dim3((unsigned)nblk), dim3(THREADS), lds, stream, a);
    rc = launch_status();
    if (rc || a.splits <= 1) return rc;
    return launch_splitk_epilogue(a, stream);
}

}  // namespace

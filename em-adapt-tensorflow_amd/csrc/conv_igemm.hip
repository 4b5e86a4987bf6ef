// Implicit-GEMM convolution on the gfx950 matrix cores: forward and data-gradient.
//
// Replaces tf.nn.conv2d / tf.nn.atrous_conv2d (+bias_add, relu, dropout) -- deeplab.py:58-59,65-66,71,
// 92-96,100,104 -- and the Conv2DBackpropInput half of the autodiff of deeplab.py:192.
//
//   out[m][n] = sum_k A[m][k] * Bw[k][n]       m = (b,oh,ow) pixel, k = (tap, c), n = output channel
//
// A is never materialised: each workgroup gathers its [BM x BK] slice of the im2col matrix straight
// from the NHWC activation (16-byte loads along the channel axis, zero fill outside the image;
// dilation is just a different tap offset) and transposes it into a k-major LDS tile, so that the
// MFMA operand reads (one fp32 per lane: lane&31 = row, lane>>5 = k parity) are conflict-free
// ds_read_b32.  Bw is the HWIO filter as it lies in memory for the forward pass; for the data
// gradient the same kernel reads the filter "transposed" (k = (tap, co), n = ci, contiguous along k)
// and mirrors the tap offsets, so no transposed copy of the weights is ever written.
//
// Arithmetic: v_mfma_f32_32x32x2_f32 -- exact fp32 multiply-add chains, accumulation in k order.
// Tiling: 256 threads = 4 wavefronts, each owning a 64x64 accumulator block (2x2 MFMA tiles, 64
// accumulator VGPRs); global->register prefetch of tile k+1 overlaps the MFMAs of tile k; two LDS
// buffers, one barrier per K step; two workgroups per CU.  blockIdx is remapped so that the eight
// workgroups that share an XCD (and its L2) work on neighbouring tiles (8 M-tiles x all N-tiles).
// Epilogue (bias / ReLU / dropout, or the ReLU-gradient gate for dgrad) is fused.
#include <type_traits>

#include "igemm_common.h"
#include "igemm_v2.h"

namespace {

// Tile configurations: <BM, BN, BK, WN>.  Every wave owns a 64 x WN accumulator block (WN = 64: 64 VGPRs, WN = 128:
// 128 VGPRs); THREADS = (BM/64)*(BN/WN)*64.
//   128x128x32, WN 64,  256 threads, 2 workgroups/CU  -- default
//   256x64x16,  WN 64,  256 threads                   -- N <= 64 layers
//   256x256x32, WN 128, 512 threads, 1 workgroup/CU   -- N >= 1024 layers (fc6/fc7): half the global-load
//                                                         instructions per MFMA of the 128x128 tile
template <int BM, int BN, int BK, int WN, bool BT>
__global__ void __launch_bounds__((BM / 64) * (BN / WN) * 64, 2) igemm_kernel(IgemmArgs a) {
    constexpr int WM = 64;                         // per-wave accumulator block is WM x WN
    constexpr int WAVES_N = BN / WN;               // waves along N
    constexpr int THREADS = (BM / WM) * (BN / WN) * 64;
    static_assert(THREADS == 256 || THREADS == 512, "4 or 8 wavefronts per workgroup");
    constexpr int TM = WM / 32, TN = WN / 32;
    constexpr int AS = BM + 1;                     // LDS pitch of the k-major A tile (odd: conflict-free transposing stores)
    constexpr int BS = BT ? BN + 1 : BN;
    constexpr int KQ = BK / 4;                     // float4 quads along k
    constexpr int A_RSTEP = THREADS / KQ;          // rows covered by one pass of the workgroup's threads
    constexpr int A_N = BM / A_RSTEP;              // A float4 loads per thread
    constexpr int QN = BN / 4;
    constexpr int B_KSTEP = THREADS / QN;
    constexpr int B_N = BT ? (BN / A_RSTEP) : (BK / B_KSTEP);
    static_assert(A_N >= 1 && B_N >= 1, "tile too small for the workgroup");

    extern __shared__ __attribute__((aligned(16))) float lds[];
    float *As = lds;                        // [2][BK][AS]
    float *Bs = lds + 2 * BK * AS;          // [2][BK][BS]   (2*BK*AS*4 bytes is a multiple of 16)

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wm = wave / WAVES_N, wn = wave % WAVES_N;

    // ---- XCD-aware tile order: blocks b, b+8, b+16.. share an XCD; give each XCD a contiguous run of
    //      tiles, ordered as groups of 8 M-tiles x all N-tiles (operands of neighbours hit in its L2).
    int tile_m, tile_n, split;
    {
        const int ntile = a.tiles_m * a.tiles_n;
        const int nblk = ntile * a.splits;
        const int bid = blockIdx.x;
        const int xcd = bid & 7, q = nblk >> 3, r = nblk & 7;
        int t = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
        split = t / ntile;  // all tiles of split 0 first, then split 1, ...
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

    // ---- per-thread gather state for the A rows it loads ----
    const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(a.x), 0, (int)a.x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t wr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(a.w), 0, (int)a.w_bytes, 0x00020000);
    const int a_kq = tid % KQ;
    const int a_r0 = tid / KQ;
    int a_base[A_N];  // element offset of the row's pixel: ((b*H+oh)*W+ow)*Cx
    int a_oh[A_N], a_ow[A_N];
#pragma unroll
    for (int j = 0; j < A_N; j++) {
        const int m = m0 + a_r0 + j * A_RSTEP;
        if (m < a.m_end) {
            const int t2 = m / a.W;
            a_base[j] = m * a.Cx;
            a_oh[j] = t2 % a.H;
            a_ow[j] = m - t2 * a.W;
        } else {
            a_base[j] = 0;
            a_oh[j] = -(1 << 20);
            a_ow[j] = -(1 << 20);
        }
    }
    // B loader coordinates
    const int b_nq = tid % QN, b_kr = tid / QN;       // !BT: float4 along n
    const int b_nr = tid / KQ;                        //  BT: float4 along k (same k quad as A)

    // two register sets: while set P receives the global loads of tile kt+2, set P^1 (tile kt+1, loaded one
    // iteration earlier) is parked in LDS -- a load has more than a full K step of MFMAs to land.
    float4 ra[2][A_N], rb[2][B_N];

    // Out-of-range accesses are marked by setting bit 31 of the byte offset: every buffer is < 2 GiB, so the
    // hardware range check returns zeros for them -- no branch, no exec masking.
    constexpr unsigned kOOB = 0x80000000u;
    struct Tap { unsigned tap; int c, dh, dw, off; bool valid; };
    // this thread's k quad of K step kt: k = kt*BK + a_kq*4 -> (tap = kh*KW + kw, channel c), exact for k < 2^20
    // K-step order.  Linear: k = kt*BK + .. = (tap, c) with the channel fastest.  tap_inner (Cx % BK == 0): K step
    // kt = (channel chunk kt / ntaps, tap kt % ntaps): all taps of one 128-byte channel slice run back to back, so
    // the slice of the tile's pixels (+halo) is fetched once and then hit in L1/L2 by the other taps.
    auto tap_of = [&](int kt) {
        Tap t;
        if (a.tap_inner) {
            const unsigned chunk = __umulhi((unsigned)kt, a.magic_nt);
            t.tap = (unsigned)kt - chunk * (unsigned)a.ntaps;
            t.c = (int)chunk * BK + a_kq * 4;
            if ((int)chunk * BK >= a.Cx) t.tap = (unsigned)a.ntaps;  // past the end of K
        } else {
            const unsigned k = (unsigned)(kt * BK + a_kq * 4);
            t.tap = __umulhi(k, a.magic_cx);
            t.c = (int)(k - t.tap * (unsigned)a.Cx);
        }
        const unsigned kh = div_magic(t.tap, a.magic_kw);
        const int kw = (int)(t.tap - kh * (unsigned)a.KW);
        t.valid = (int)t.tap < a.ntaps;
        t.dh = a.sign * ((int)kh * a.dil - a.pad_h);
        t.dw = a.sign * (kw * a.dil - a.pad_w);
        t.off = (t.dh * a.W + t.dw) * a.Cx + t.c;
        return t;
    };
    auto load_a = [&](float4 &dst, int j, const Tap &t) {
        const int ih = a_oh[j] + t.dh, iw = a_ow[j] + t.dw;
        const bool ok = t.valid && (unsigned)ih < (unsigned)a.H && (unsigned)iw < (unsigned)a.W;
        const unsigned off = (unsigned)(a_base[j] + t.off) * 4u;
        dst = buf_ld4(xr, off | (ok ? 0u : kOOB));
    };
    auto load_b = [&](float4 &dst, int j, int kt, const Tap &t) {
        if constexpr (!BT) {
            // row of the [K][N] filter matrix: linear order kt*BK + r; tap-inner order tap*Cx + chunk*BK + r
            const int r = b_kr + j * B_KSTEP;
            const int kb = a.tap_inner ? (int)t.tap * a.Cx + (t.c - a_kq * 4) + r : kt * BK + r;
            const int n = n0 + b_nq * 4;
            const unsigned off = (unsigned)(kb * a.N + n) * 4u;
            dst = buf_ld4(wr, off | ((kb < a.K && n < a.N) ? 0u : kOOB));
        } else {
            const int n = n0 + b_nr + j * A_RSTEP;
            const unsigned off = (unsigned)(((int)t.tap * a.N + n) * a.Cx + t.c) * 4u;
            dst = buf_ld4(wr, off | ((t.valid && n < a.N) ? 0u : kOOB));
        }
    };
    auto store_a = [&](int buf, int j, const float4 &v) {
        float *as = As + buf * BK * AS;
        const int r = a_r0 + j * A_RSTEP;
        as[(a_kq * 4 + 0) * AS + r] = v.x;
        as[(a_kq * 4 + 1) * AS + r] = v.y;
        as[(a_kq * 4 + 2) * AS + r] = v.z;
        as[(a_kq * 4 + 3) * AS + r] = v.w;
    };
    auto store_b = [&](int buf, int j, const float4 &v) {
        float *bs = Bs + buf * BK * BS;
        if constexpr (!BT) {
            *reinterpret_cast<float4 *>(&bs[(b_kr + j * B_KSTEP) * BS + b_nq * 4]) = v;
        } else {
            const int n = b_nr + j * A_RSTEP;
            bs[(a_kq * 4 + 0) * BS + n] = v.x;
            bs[(a_kq * 4 + 1) * BS + n] = v.y;
            bs[(a_kq * 4 + 2) * BS + n] = v.z;
            bs[(a_kq * 4 + 3) * BS + n] = v.w;
        }
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; i++)
#pragma unroll
        for (int j = 0; j < TN; j++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[i][j][r] = 0.f;

    // Software pipeline.  Inside the MFMA stream of K step kt -- NKK k-pairs of TM*TN MFMAs, 64 cycles each -- the
    // tile movement is spread out in program order:
    //   k-pairs 1 .. LD_END      : global loads of tile kt+2 into register set P (a few per pair, with their
    //                              address arithmetic)
    //   k-pairs ST0 .. NKK-1     : one part per pair of "park tile kt+1 (register set P^1) in the idle LDS buffer"
    // so the LDS writes, VALU and VMEM issue in the shadow of the MFMAs instead of forming a phase of their own
    // in which the matrix pipe idles.  Everything is unconditional (tiles past K are zero-filled by the range
    // check and land in an LDS buffer nobody reads), so an iteration is one branch-free block, one barrier.
    // The 128-accumulator configuration (WN = 128) has no room for two register sets: it uses ONE set, stores first
    // (k-pairs 1..NST), then re-loads (k-pairs NST+1..); its k-pairs are twice as long (8 MFMAs), so a load still
    // has >= 2500 cycles before it is needed.
    constexpr bool TWO_SETS = (WN == 64);
    constexpr int NKK = BK / 2;
    constexpr int NST = A_N + B_N;              // tile parts (one load / one LDS store each)
    constexpr int ST0 = TWO_SETS ? NKK - NST : 1;            // first k-pair that stores
    constexpr int LD0 = TWO_SETS ? 1 : 1 + NST;             // first k-pair that loads
    constexpr int LDN = TWO_SETS ? ST0 - 1 : NKK - LD0;     // number of k-pairs that load
    static_assert(ST0 >= 1 && LDN >= 1, "K step too short for the software pipeline");
    constexpr int LPK = (NST + LDN - 1) / LDN;              // loads per k-pair
    const int nk_all = (a.K + BK - 1) / BK;
    const int kt0 = (int)((long long)nk_all * split / a.splits);        // this workgroup's K steps [kt0, kt0 + nk)
    const int nk = (int)((long long)nk_all * (split + 1) / a.splits) - kt0;
    {
        const Tap t0 = tap_of(kt0);
#pragma unroll
        for (int j = 0; j < A_N; j++) load_a(ra[0][j], j, t0);
#pragma unroll
        for (int j = 0; j < B_N; j++) load_b(rb[0][j], j, kt0, t0);
        const Tap t1 = tap_of(kt0 + 1);
        if constexpr (TWO_SETS) {
#pragma unroll
            for (int j = 0; j < A_N; j++) load_a(ra[1][j], j, t1);
#pragma unroll
            for (int j = 0; j < B_N; j++) load_b(rb[1][j], j, kt0 + 1, t1);
        }
#pragma unroll
        for (int j = 0; j < A_N; j++) store_a(0, j, ra[0][j]);
#pragma unroll
        for (int j = 0; j < B_N; j++) store_b(0, j, rb[0][j]);
        if constexpr (!TWO_SETS) {
#pragma unroll
            for (int j = 0; j < A_N; j++) load_a(ra[0][j], j, t1);
#pragma unroll
            for (int j = 0; j < B_N; j++) load_b(rb[0][j], j, kt0 + 1, t1);
        }
    }
    __syncthreads();

    const int l31 = lane & 31, lh = lane >> 5;
    auto k_step = [&](auto parity, int kt) {
        constexpr int P = decltype(parity)::value;  // = kt & 1: LDS buffer being multiplied, register set being loaded
        const float *as = As + P * BK * AS + wm * WM + l31;
        const float *bs = Bs + P * BK * BS + wn * WN + l31;
        Tap tnext = {};
        // LDS fragment reads run one k-pair ahead of the MFMAs that consume them
        float av[2][TM], bv[2][TN];
#pragma unroll
        for (int i = 0; i < TM; i++) av[0][i] = as[lh * AS + i * 32];
#pragma unroll
        for (int j = 0; j < TN; j++) bv[0][j] = bs[lh * BS + j * 32];
#pragma unroll
        for (int kk = 0; kk < NKK; kk++) {
            if (kk + 1 < NKK) {
#pragma unroll
                for (int i = 0; i < TM; i++) av[(kk + 1) & 1][i] = as[((kk + 1) * 2 + lh) * AS + i * 32];
#pragma unroll
                for (int j = 0; j < TN; j++) bv[(kk + 1) & 1][j] = bs[((kk + 1) * 2 + lh) * BS + j * 32];
            }
            // hard scheduling fence: the reads above stay ahead of this k-pair's MFMAs (one k-pair of latency cover)
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < TM; i++)
#pragma unroll
                for (int j = 0; j < TN; j++)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[kk & 1][i], bv[kk & 1][j], acc[i][j], 0, 0, 0);
            // tile movement slotted into this k-pair (program order: a part's LDS store precedes its re-load)
            constexpr int LS = TWO_SETS ? P : 0;      // register set receiving tile kt+2
            constexpr int SS = TWO_SETS ? P ^ 1 : 0;  // register set holding tile kt+1
            if (kk >= ST0 && kk < ST0 + NST && !ABL(2)) {
                const int sidx = kk - ST0;
                if (sidx < A_N) store_a(P ^ 1, sidx, ra[SS][sidx < A_N ? sidx : 0]);
                else store_b(P ^ 1, sidx - A_N, rb[SS][sidx >= A_N && sidx - A_N < B_N ? sidx - A_N : 0]);
            }
            if (kk >= LD0 && kk < LD0 + LDN && !ABL(1)) {
#pragma unroll
                for (int q = 0; q < LPK; q++) {
                    const int lidx = (kk - LD0) * LPK + q;
                    if (lidx == 0) tnext = tap_of(kt + 2);
                    if (lidx < A_N) { if (!ABL(8)) load_a(ra[LS][lidx < A_N ? lidx : 0], lidx, tnext); }
                    else if (lidx < NST) { if (!ABL(16)) load_b(rb[LS][lidx - A_N < B_N && lidx >= A_N ? lidx - A_N : 0], lidx - A_N, kt + 2, tnext); }
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        if (!ABL(4)) __syncthreads();
    };
    for (int kt = 0; kt < nk; kt += 2) {  // k_step's argument is the absolute K step (the parity is relative)
        k_step(std::integral_constant<int, 0>{}, kt0 + kt);
        if (kt + 1 < nk) k_step(std::integral_constant<int, 1>{}, kt0 + kt + 1);
    }

    // ---- fused epilogue through LDS (igemm_common.h: store_acc_tile): the tiles are dead after the last barrier.
    // split-K partials are stored compactly: slab[split][row - m_begin][col]
    constexpr int LDS_FLOATS = 2 * BK * (AS + BS);
    constexpr int ROWS = ((THREADS / 64) * 32 * (WN + 4) <= LDS_FLOATS) ? 32 : 16;
    static_assert((THREADS / 64) * ROWS * (WN + 4) <= LDS_FLOATS, "epilogue slabs must fit in the tile LDS");
    float *slab = a.splits > 1 ? a.slab + ((long long)split * (a.m_end - a.m_begin) - a.m_begin) * a.N : nullptr;
    store_acc_tile<TM, TN, ROWS, false>(a, acc, lds + wave * (ROWS * (WN + 4)), lane, m0 + wm * WM, n0 + wn * WN, slab);
}

// ------------------------------------------------------------------------------------------------------------------
// Wave-specialised variant of the 128x128x32 tile: 512 threads = 4 CONSUMER wavefronts (LDS fragment reads + MFMA,
// nothing else in their instruction stream) + 4 PRODUCER wavefronts (im2col gather, filter loads, LDS stores).  A wave
// issues in order, so in the unified kernel every global load / address computation / ds_write sits in front of that
// wave's next MFMA; with the roles split the matrix pipe of a SIMD is fed by a wave that never waits on VMEM, while
// the producer's VALU/VMEM/LDS work runs on the separate vector pipes.  The two roles live in disjoint branches, so the
// register allocation is max(consumer, producer), not the sum (2 workgroups = 16 waves per CU need <= 128 VGPRs).
// Same LDS layout, K order, split-K and epilogue as igemm_kernel; one barrier per K step joins both roles:
//   producer, step kt : park tile kt+1 (registers) in LDS buffer (kt+1)&1, issue the loads of tile kt+2, barrier
//   consumer, step kt : 64 MFMAs on LDS buffer kt&1, barrier
template <bool BT>
__global__ void __launch_bounds__(512, 4) igemm_ws_kernel(IgemmArgs a) {
    constexpr int BM = 128, BN = 128, BK = 32, WM = 64, WN = 64;
    constexpr int WAVES_N = BN / WN;
    constexpr int TM = WM / 32, TN = WN / 32;
    constexpr int AS = BM + 1;
    constexpr int BS = BT ? BN + 1 : BN;
    constexpr int KQ = BK / 4;
    constexpr int A_RSTEP = 256 / KQ;
    constexpr int A_N = BM / A_RSTEP;
    constexpr int QN = BN / 4;
    constexpr int B_KSTEP = 256 / QN;
    constexpr int B_N = BT ? (BN / A_RSTEP) : (BK / B_KSTEP);
    constexpr int NKK = BK / 2;

    extern __shared__ __attribute__((aligned(16))) float lds[];
    float *As = lds;
    float *Bs = lds + 2 * BK * AS;

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
    const int nk_all = (a.K + BK - 1) / BK;
    const int kt0 = (int)((long long)nk_all * split / a.splits);
    const int nk = (int)((long long)nk_all * (split + 1) / a.splits) - kt0;

    if (threadIdx.x >= 256) {
        // ================================ PRODUCER wavefronts ================================
        const int tid = threadIdx.x - 256;
        const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(a.x), 0, (int)a.x_bytes, 0x00020000);
        const __amdgpu_buffer_rsrc_t wr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(a.w), 0, (int)a.w_bytes, 0x00020000);
        constexpr unsigned kOOB = 0x80000000u;
        const int a_kq = tid % KQ, a_r0 = tid / KQ;
        int a_base[A_N], a_oh[A_N], a_ow[A_N];
#pragma unroll
        for (int j = 0; j < A_N; j++) {
            const int m = m0 + a_r0 + j * A_RSTEP;
            if (m < a.m_end) {
                const int t2 = m / a.W;
                a_base[j] = m * a.Cx;
                a_oh[j] = t2 % a.H;
                a_ow[j] = m - t2 * a.W;
            } else {
                a_base[j] = 0;
                a_oh[j] = -(1 << 20);
                a_ow[j] = -(1 << 20);
            }
        }
        const int b_nq = tid % QN, b_kr = tid / QN;
        const int b_nr = tid / KQ;
        float4 ra[A_N], rb[B_N];

        auto load_tile = [&](int kt) {
            unsigned tap;
            int c;
            if (a.tap_inner) {
                const unsigned chunk = __umulhi((unsigned)kt, a.magic_nt);
                tap = (unsigned)kt - chunk * (unsigned)a.ntaps;
                c = (int)chunk * BK + a_kq * 4;
                if ((int)chunk * BK >= a.Cx) tap = (unsigned)a.ntaps;
            } else {
                const unsigned k = (unsigned)(kt * BK + a_kq * 4);
                tap = __umulhi(k, a.magic_cx);
                c = (int)(k - tap * (unsigned)a.Cx);
            }
            const unsigned kh = div_magic(tap, a.magic_kw);
            const int kw = (int)(tap - kh * (unsigned)a.KW);
            const bool kvalid = (int)tap < a.ntaps;
            const int dh = a.sign * ((int)kh * a.dil - a.pad_h);
            const int dw = a.sign * (kw * a.dil - a.pad_w);
            const int toff = (dh * a.W + dw) * a.Cx + c;
#pragma unroll
            for (int j = 0; j < A_N; j++) {
                const int ih = a_oh[j] + dh, iw = a_ow[j] + dw;
                const bool ok = kvalid && (unsigned)ih < (unsigned)a.H && (unsigned)iw < (unsigned)a.W;
                ra[j] = buf_ld4(xr, ((unsigned)(a_base[j] + toff) * 4u) | (ok ? 0u : kOOB));
            }
#pragma unroll
            for (int j = 0; j < B_N; j++) {
                if constexpr (!BT) {
                    const int r = b_kr + j * B_KSTEP;
                    const int kb = a.tap_inner ? (int)tap * a.Cx + (c - a_kq * 4) + r : kt * BK + r;
                    const int n = n0 + b_nq * 4;
                    rb[j] = buf_ld4(wr, ((unsigned)(kb * a.N + n) * 4u) | ((kb < a.K && n < a.N) ? 0u : kOOB));
                } else {
                    const int n = n0 + b_nr + j * A_RSTEP;
                    rb[j] = buf_ld4(wr, ((unsigned)(((int)tap * a.N + n) * a.Cx + c) * 4u) | ((kvalid && n < a.N) ? 0u : kOOB));
                }
            }
        };
        auto store_tile = [&](int buf) {
            float *as = As + buf * BK * AS;
            float *bs = Bs + buf * BK * BS;
#pragma unroll
            for (int j = 0; j < A_N; j++) {
                const int r = a_r0 + j * A_RSTEP;
                as[(a_kq * 4 + 0) * AS + r] = ra[j].x;
                as[(a_kq * 4 + 1) * AS + r] = ra[j].y;
                as[(a_kq * 4 + 2) * AS + r] = ra[j].z;
                as[(a_kq * 4 + 3) * AS + r] = ra[j].w;
            }
#pragma unroll
            for (int j = 0; j < B_N; j++) {
                if constexpr (!BT) {
                    *reinterpret_cast<float4 *>(&bs[(b_kr + j * B_KSTEP) * BS + b_nq * 4]) = rb[j];
                } else {
                    const int n = b_nr + j * A_RSTEP;
                    bs[(a_kq * 4 + 0) * BS + n] = rb[j].x;
                    bs[(a_kq * 4 + 1) * BS + n] = rb[j].y;
                    bs[(a_kq * 4 + 2) * BS + n] = rb[j].z;
                    bs[(a_kq * 4 + 3) * BS + n] = rb[j].w;
                }
            }
        };
        load_tile(kt0);
        store_tile(0);
        load_tile(kt0 + 1);  // K steps past the end are zero-filled by the range check
        __syncthreads();
        for (int kt = 0; kt < nk; kt++) {
            store_tile((kt + 1) & 1);
            load_tile(kt0 + kt + 2);
            __syncthreads();
        }
        return;
    }

    // ================================ CONSUMER wavefronts ================================
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WAVES_N, wn = wave % WAVES_N;
    const int l31 = lane & 31, lh = lane >> 5;
    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; i++)
#pragma unroll
        for (int j = 0; j < TN; j++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[i][j][r] = 0.f;
    __syncthreads();
    for (int kt = 0; kt < nk; kt++) {
        const int cur = kt & 1;
        const float *as = As + cur * BK * AS + wm * WM + l31;
        const float *bs = Bs + cur * BK * BS + wn * WN + l31;
        float av[2][TM], bv[2][TN];
#pragma unroll
        for (int i = 0; i < TM; i++) av[0][i] = as[lh * AS + i * 32];
#pragma unroll
        for (int j = 0; j < TN; j++) bv[0][j] = bs[lh * BS + j * 32];
#pragma unroll
        for (int kk = 0; kk < NKK; kk++) {
            if (kk + 1 < NKK) {
#pragma unroll
                for (int i = 0; i < TM; i++) av[(kk + 1) & 1][i] = as[((kk + 1) * 2 + lh) * AS + i * 32];
#pragma unroll
                for (int j = 0; j < TN; j++) bv[(kk + 1) & 1][j] = bs[((kk + 1) * 2 + lh) * BS + j * 32];
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < TM; i++)
#pragma unroll
                for (int j = 0; j < TN; j++)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[kk & 1][i], bv[kk & 1][j], acc[i][j], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
        __syncthreads();
    }
    // epilogue through LDS (igemm_common.h: store_acc_tile); split-K partials: slab[split][row - m_begin][col]
    static_assert(4 * 32 * (WN + 4) <= 2 * BK * (AS + BS), "epilogue slabs must fit in the tile LDS");
    float *slab = a.splits > 1 ? a.slab + ((long long)split * (a.m_end - a.m_begin) - a.m_begin) * a.N : nullptr;
    store_acc_tile<TM, TN, 32, false>(a, acc, lds + wave * (32 * (WN + 4)), lane, m0 + wm * WM, n0 + wn * WN, slab);
}

template <bool BT>
int launch_igemm_ws(IgemmArgs &a, hipStream_t stream) {
    constexpr int BM = 128, BN = 128, BK = 32;
    constexpr int AS = BM + 1, BS = BT ? BN + 1 : BN;
    constexpr size_t lds = sizeof(float) * 2 * BK * (AS + BS);
    static DynLdsAttr attr;
    int rc = attr.ensure(reinterpret_cast<const void *>(&igemm_ws_kernel<BT>), lds);
    if (rc) return rc;
    a.tiles_m = (int)ceil_div(a.m_end - a.m_begin, BM);
    a.tiles_n = (int)ceil_div(a.N, BN);
    const long long nblk = (long long)a.tiles_m * a.tiles_n * a.splits;
    if (nblk > 0x7fffffffLL) return EMADAPT_ERR_UNSUPPORTED;
    hipLaunchKernelGGL((igemm_ws_kernel<BT>), dim3((unsigned)nblk), dim3(512), lds, stream, a);
    rc = launch_status();
    if (rc || a.splits <= 1) return rc;
    return launch_splitk_epilogue(a, stream);
}

template <int BM, int BN, int BK, int WN, bool BT>
int launch_igemm(IgemmArgs &a, hipStream_t stream) {
    constexpr int THREADS = (BM / 64) * (BN / WN) * 64;
    constexpr int AS = BM + 1, BS = BT ? BN + 1 : BN;
    constexpr size_t lds = sizeof(float) * 2 * BK * (AS + BS);
    static_assert((2 * BK * AS * 4) % 16 == 0, "B tile must stay 16-byte aligned");
    static DynLdsAttr attr;
    int rc = attr.ensure(reinterpret_cast<const void *>(&igemm_kernel<BM, BN, BK, WN, BT>), lds);
    if (rc) return rc;
    a.tiles_m = (int)ceil_div(a.m_end - a.m_begin, BM);
    a.tiles_n = (int)ceil_div(a.N, BN);
    const long long nblk = (long long)a.tiles_m * a.tiles_n * a.splits;
    if (nblk > 0x7fffffffLL) return EMADAPT_ERR_UNSUPPORTED;
    hipLaunchKernelGGL((igemm_kernel<BM, BN, BK, WN, BT>), dim3((unsigned)nblk), dim3(THREADS), lds, stream, a);
    rc = launch_status();
    if (rc || a.splits <= 1) return rc;
    return launch_splitk_epilogue(a, stream);
}

// How one convolution is launched.  Tile variants ("fp32_tile" numbering of emadapt_tune_set / emadapt_conv_plan):
//   1: 256x64x16 unified kernel (N <= 64)
//   2: 256x256x32 unified kernel: what variant 4 falls back to (Cx % 16 != 0, more than 32 taps)
//   3: 128x128x32 wave-specialised kernel, split-K when the tile count is small; also the "tail" of variants 2 / 4
//   4: 256x256x16 second generation (igemm_v2.h: LDS-DMA loaders, 64x128 per MFMA wave), one workgroup per CU
// Variants 2 / 4 are used when their tiles fill at least one whole round of the CUs (fc6 / fc7 forward, fc7 / fc8 data
// gradient, conv3_x): they take the rows that fill WHOLE rounds, and the remaining rows (the partial last round: 160 of
// fc6's 1696 tiles) go to a second launch of variant 3 with split-K, so that no CU idles while a handful of big tiles
// finish.  Split-K (fp32 partial slabs + a fixed-order reduce) fills whole rounds when the tile count is small.
// Measured in one process per layer (scripts/conv_bench.py --tiles=..., profiles/r02_fp32_tile_variants.txt).
struct LaunchPlan {
    int cfg;         // tile variant of the main launch
    int main_rows;   // rows [0, main_rows) go to the main launch
    int main_splits;
    int tail_cfg;    // tile variant of the launch over rows [main_rows, M)
    int tail_splits; // its split-K factor (0 = no tail)
    int bands;       // variant 5 over packed rows (RowPack): pixel classes of the launch; their tap boxes' padding is skipped
};

inline bool v2_ok(const IgemmArgs &a, int bk) { return a.Cx % bk == 0 && a.ntaps <= 32 && a.K >= 2 * bk; }

int splits_for(const IgemmArgs &a, int cfg, long long rows) {
    switch (cfg) {
        case 1: return choose_splits(rows, a.N, a.K, 256, 64, 16);
        case 3: return choose_splits(rows, a.N, a.K, 128, 128, 32);
        default: return 1;
    }
}

LaunchPlan plan_launches(const IgemmArgs &a) {
    const int force = tune(TUNE_FP32_TILE);  // 0 = by shape; the tests force every variant onto small shapes
    const int cus = plan_cus();
    LaunchPlan p;
    p.main_rows = a.M;
    p.tail_cfg = 0;
    p.tail_splits = 0;
    p.main_splits = 1;
    p.bands = 0;
    // 256x256 tiles pay when they fill at least one whole round of the CUs and N uses the tile's width; with fewer tiles
    // (fc6's data gradient: 212 tiles, K = 65536) variant 4 runs split-K when the reduction is long enough to carry the
    // slab traffic -- decided by the estimated times of the two plans (rates: sustained TFLOP/s of each main loop)
    const long long t256 = ceil_div(a.M, 256) * ceil_div(a.N, 256);
    const bool big = a.N >= 256 && t256 >= cus;
    bool split4 = false;
    if (!big && a.N >= 256 && v2_ok(a, 16)) {
        double t3 = 0, t4 = 0;
        (void)choose_splits(a.M, a.N, a.K, 128, 128, 32, 512.0, 132e12, &t3);
        (void)choose_splits(a.M, a.N, a.K, 256, 256, 16, 256.0, 142e12, &t4);
        split4 = t4 < t3 * 0.98;
    }
    // (K <= 32 -- fc8's data gradient, K = 24 -- is one K step of any kernel: the launch is output traffic, and the small tiles
    //  of variant 1, several workgroups per CU, keep it moving: 0.245 -> 0.208 ms, profiles/r03_fp32_experiments.txt)
    int cfg = (a.N <= 64 || a.K <= 32) ? 1 : big ? (v2_ok(a, 16) ? 4 : 2) : split4 ? 4 : 3;
    // Variant 5 = the 256x256 kernel with the stream-K work split (igemm_v2.h): whenever the tile count is not a whole number
    // of rounds and every CU still gets a few dozen K steps.  N <= 1024: wider layers (N = 4096) keep the rows-of-whole-rounds + tail
    // plan -- not for workspace (since round 4 a stream-K launch needs 64 MB whatever the layer) but because on the stream-K unit
    // line no two workgroups of an XCD share a K phase, which costs fc7 2.7 % in fp32 and 16 % on the plane kernels.
    const int nk16 = a.K / 16;
    if (cfg != 1 && v2_ok(a, 16) && a.N >= 256 && a.N <= 1024 && t256 % cus != 0 && t256 * nk16 / cus >= 32) cfg = 5;
    if (force >= 1 && force <= 5 && !(force >= 4 && !v2_ok(a, 16))) cfg = force;
    // Dilated filters on small maps multiply a lot of padding (fc6, 4x4 rate 4 on 41x41: filter rows -6 / +6 are outside the
    // image for 6 of its 41 rows each): over packed rows (igemm_common.h RowPack) a stream-K launch skips the filter rows and
    // columns that are wholly outside, whatever N is (fp32_loader = 2 keeps the dense reduction: tests, A/B).
    if ((force == 0 || force == 5) && cfg != 1 && v2_ok(a, 16) && a.N >= 256 && tune(TUNE_FP32_LOADER) != 2) {
        IgemmArgs b = a;
        const int tn = (int)ceil_div(a.N, 256);
        if (build_row_pack(b, tn) && b.pk.run_u0[b.pk.nruns] / cus >= 32) {
            p.cfg = 5;
            p.bands = b.pk.n;
            p.main_splits = streamk_levels_packed(b, tn, cus);
            return p;
        }
    }
    p.cfg = cfg;
    if (cfg == 5) {
        p.main_splits = streamk_levels(t256, nk16, cus);   // slab levels (0: every tile is run by one workgroup)
    } else if (cfg == 4 && t256 < cus) {
        p.main_splits = choose_splits(a.M, a.N, a.K, 256, 256, 16, 256.0, 142e12);
    } else if (cfg == 2 || cfg == 4) {
        const long long tn = ceil_div(a.N, 256), tm = ceil_div(a.M, 256);
        const long long full = (tm * tn) / cus;  // whole rounds of one 256x256 workgroup per CU
        if (full >= 1 && (tm * tn) % cus != 0) {
            const long long tm_full = full * cus / tn;
            if (tm_full >= 1 && tm_full < tm) {
                p.main_rows = (int)(tm_full * 256);
                p.tail_cfg = 3;
                p.tail_splits = splits_for(a, p.tail_cfg, a.M - p.main_rows);
                // the rows of the partial last round as a stream-K launch of the same 256x256 kernel (variant 4 only: its
                // reduction is a whole number of 16-wide steps): few tiles, all workgroups on the same rows -- the filter
                // stays L2-resident there -- and slab levels of tail rows only
                const long long tail_tiles = (tm - tm_full) * tn;
                if (cfg == 4 && tail_tiles * nk16 / cus >= 32 && tune(TUNE_FP32_TILE) == 0) {
                    p.tail_cfg = 5;
                    p.tail_splits = streamk_levels(tail_tiles, nk16, cus);
                }
            }
        }
    } else {
        p.main_splits = splits_for(a, cfg, a.M);
    }
    return p;
}

// (packed rows: the GEMM row -> pixel table sits behind the slab levels)
inline size_t rowmap_bytes(const IgemmArgs &a) { return (size_t)ceil_div(a.M, 256) * 256 * sizeof(int); }

// (stream-K launches, variant 5: arrival words + one 256 KB slab per workgroup when tiles are shared -- igemm_v2.h -- whatever
//  the layer's size; rounds 2-3 kept `levels` [M][N] images: 881 MB for fc6)
size_t igemm_workspace_bytes(const IgemmArgs &a) {
    const LaunchPlan p = plan_launches(a);
    const size_t sk = streamk_workspace_bytes(plan_cus());
    if (p.cfg == 5) return (p.main_splits > 0 ? sk : 0) + (p.bands ? rowmap_bytes(a) : 0);
    size_t b = p.main_splits > 1 ? (size_t)p.main_splits * p.main_rows * a.N * sizeof(float) : 0;
    if (p.tail_cfg == 5) b = b > (p.tail_splits > 0 ? sk : 0) ? b : (p.tail_splits > 0 ? sk : 0);   // (the launches run one after the other)
    else if (p.tail_splits > 1) b += (size_t)p.tail_splits * (a.M - p.main_rows) * a.N * sizeof(float);
    return b;
}

// a.sk_flags / a.slab of a stream-K launch from the start of the workspace
inline void set_streamk_workspace(IgemmArgs &a, void *workspace, bool shared_tiles) {
    a.sk_flags = shared_tiles ? static_cast<unsigned *>(workspace) : nullptr;
    a.slab = shared_tiles ? reinterpret_cast<float *>(static_cast<char *>(workspace) + streamk_flag_bytes(plan_cus())) : nullptr;
}

// > 0: the second-generation kernel's loaders take the scalar-offset form (igemm_v2.h, NT); 0: the generic loaders.  The scalar
// form reads the activation through a buffer that starts up to (pad_h * W + pad_w) pixels before the tensor: its extent must
// still fit the 31 bits of a buffer offset.
int specialised_taps(const IgemmArgs &a) {
    if (tune(TUNE_FP32_LOADER) == 1) return 0;
    const long long reach = ((long long)(a.ntaps / a.KW - 1) * a.dil * a.W + (long long)(a.KW - 1) * a.dil) * a.Cx * 4;
    if ((long long)a.x_bytes + reach >= (1LL << 31)) return 0;
    return a.ntaps;
}

template <bool BT>
int launch_variant(int cfg, IgemmArgs &a, hipStream_t stream) {
    a.tap_inner = (a.ntaps > 1 && a.Cx % (cfg == 1 ? 16 : 32) == 0) ? 1 : 0;   // (variants 4 / 5 always use that order)
    switch (cfg) {
        case 1: return launch_igemm<256, 64, 16, 64, BT>(a, stream);
        case 2: return launch_igemm<256, 256, 32, 128, BT>(a, stream);
        case 4: {
            if (specialised_taps(a) > 0) return launch_igemm2<4, 2, 16, 4, 4, 3, BT, 1, false>(a, stream);
            return launch_igemm2<4, 2, 16, 4, 4, 3, BT, 0, false>(a, stream);
        }
        case 5: {   // (a.splits = workgroups, a.slab = slab levels: dispatch_igemm)
            if (specialised_taps(a) > 0) return launch_igemm2<4, 2, 16, 4, 4, 3, BT, 1, true>(a, stream);
            return launch_igemm2<4, 2, 16, 4, 4, 3, BT, 0, true>(a, stream);
        }
        default: return launch_igemm_ws<BT>(a, stream);
    }
}

template <bool BT>
int dispatch_igemm(IgemmArgs &a, void *workspace, size_t workspace_bytes, hipStream_t stream, ProfScope &prof) {
#ifdef EMADAPT_ABLATE_BUILD
    a.ablate = tune(TUNE_ABLATE);
#else
    a.ablate = 0;
#endif
    LaunchPlan p = plan_launches(a);
    if (!workspace || workspace_bytes < igemm_workspace_bytes(a)) {  // no room for slabs / the row table: unsplit, same result
        if (p.cfg == 5 && (p.main_splits > 0 || p.bands)) { p.cfg = 3; p.bands = 0; }
        if (p.cfg != 5) p.main_splits = 1;
        if (p.tail_cfg == 5 && p.tail_splits > 0) p.tail_cfg = 3;
        if (p.tail_cfg != 5 && p.tail_splits > 1) p.tail_splits = 1;
    }
    a.pk.n = 0;
    a.pk.rowmap = nullptr;
    if (p.cfg == 5) {
        if (p.bands) {
            double exec = 1.0;
            if (!build_row_pack(a, (int)ceil_div(a.N, 256), &exec)) return EMADAPT_ERR_INVALID;   // (the plan was made from the same geometry)
            prof.executed_fraction(exec);
            a.pk.rowmap = reinterpret_cast<const int *>(static_cast<char *>(workspace) + (p.main_splits > 0 ? streamk_workspace_bytes(plan_cus()) : 0));
        }
        a.m_begin = 0;
        a.m_end = a.M;
        a.splits = plan_cus();
        set_streamk_workspace(a, workspace, p.main_splits > 0);
        return launch_variant<BT>(5, a, stream);
    }
    a.m_begin = 0;
    a.m_end = p.main_rows;
    a.splits = p.main_splits;
    a.sk_flags = nullptr;
    a.slab = p.main_splits > 1 ? static_cast<float *>(workspace) : nullptr;
    int rc = launch_variant<BT>(p.cfg, a, stream);
    if (rc || p.tail_cfg == 0) return rc;
    a.m_begin = p.main_rows;
    a.m_end = a.M;
    if (p.tail_cfg == 5) {
        a.splits = plan_cus();
        set_streamk_workspace(a, workspace, p.tail_splits > 0);
    } else {
        a.splits = p.tail_splits;
        a.slab = p.tail_splits > 1 ? static_cast<float *>(workspace) : nullptr;  // the main launch of a split plan uses no slab
    }
    return launch_variant<BT>(p.tail_cfg, a, stream);
}

}  // namespace

namespace emadapt {
// SAME padding of a stride-1 (dilated) convolution: TF pads (dilated_k - 1)/2 before, the rest after
// (nn_ops._with_space_to_batch_base_paddings): 3x3 rate r -> r; 4x4 rate 4 -> 6 (taps -6,-2,+2,+6).
int conv_pad_before(int k, int dil) { return ((k - 1) * dil) / 2; }

double exec_fraction_fp32(const emadapt_conv_desc *d, int kind) {
    if (kind == 2) return exec_fraction_fp32_wgrad(d);
    if (check_desc(d) != EMADAPT_OK || d->B == 0) return 1.0;
    IgemmArgs a;
    fill_geometry(a, d, kind == 1);
    const LaunchPlan p = plan_launches(a);
    double e = 1.0;
    if (p.cfg == 5 && p.bands && build_row_pack(a, (int)ceil_div(a.N, 256), &e)) return e;
    return 1.0;
}
}  // namespace emadapt


extern "C" {

// batch-sliced launches (tensors of 2 GiB and more): every slice is planned on its own, and the ragged last slice (B % bs images)
// may take a plan with MORE workspace than the full ones (the cost model's split counts are not monotone in M): the larger of the two
static size_t sliced_workspace_bytes(const emadapt_conv_desc *d, size_t (*per_slice)(const emadapt_conv_desc *)) {
    emadapt_conv_desc s = *d;
    s.B = slice_batch(d, false);
    if (s.B >= d->B) return 0;
    size_t w = per_slice(&s);
    if (d->B % s.B != 0) {
        s.B = d->B % s.B;
        const size_t w2 = per_slice(&s);
        w = w2 > w ? w2 : w;
    }
    return w;
}

size_t emadapt_conv_fwd_workspace_bytes(const emadapt_conv_desc *d) {
    if (d && d->B > 0 && check_desc(d) == EMADAPT_ERR_UNSUPPORTED && slice_batch(d, false) > 0)
        return sliced_workspace_bytes(d, emadapt_conv_fwd_workspace_bytes);
    if (check_desc(d) != EMADAPT_OK || d->B == 0) return 0;
    IgemmArgs a;
    fill_geometry(a, d, false);
    return igemm_workspace_bytes(a);
}

size_t emadapt_conv_dgrad_workspace_bytes(const emadapt_conv_desc *d) {
    if (d && d->B > 0 && check_desc(d) == EMADAPT_ERR_UNSUPPORTED && slice_batch(d, false) > 0)
        return sliced_workspace_bytes(d, emadapt_conv_dgrad_workspace_bytes);
    if (check_desc(d) != EMADAPT_OK || d->B == 0) return 0;
    IgemmArgs a;
    fill_geometry(a, d, true);
    return igemm_workspace_bytes(a);
}

int emadapt_conv_plan(const emadapt_conv_desc *d, int dgrad, int plan[4]) {
    int rc = check_desc(d);
    if (rc) return rc;
    if (!plan) return EMADAPT_ERR_INVALID;
    IgemmArgs a;
    fill_geometry(a, d, dgrad != 0);
    const LaunchPlan p = plan_launches(a);
    plan[0] = p.cfg;
    plan[1] = p.main_rows;
    plan[2] = p.main_splits;
    plan[3] = p.cfg == 5 && p.bands ? p.bands : p.tail_splits;
    return EMADAPT_OK;
}

int emadapt_conv_executed_fraction(const emadapt_conv_desc *d, int kind, int arithmetic, double *fraction) {
    if (!d || !fraction || kind < 0 || kind > 2 || arithmetic < 0 || arithmetic > 2) return EMADAPT_ERR_INVALID;
    *fraction = arithmetic == 0 ? exec_fraction_fp32(d, kind) : arithmetic == 2 ? exec_fraction_x6p(d, kind) : 1.0;
    return EMADAPT_OK;
}

int emadapt_conv_fwd(const emadapt_conv_desc *d, const float *x, const float *w, const float *bias, float *y,
                     int epilogue, float keep_prob, const uint8_t *drop_mask, uint64_t drop_seed, void *workspace,
                     size_t workspace_bytes, void *stream_) {
    int rc = check_desc(d);
    if (rc == EMADAPT_ERR_UNSUPPORTED && x && y) {   // a tensor of 2 GiB or more: slices of whole images (igemm_common.h slice_batch)
        const int bs = slice_batch(d, false);
        if (bs > 0 && bs < d->B) {
            const long long hw = (long long)d->H * d->W;
            for (int b0 = 0, s = 0; b0 < d->B; b0 += bs, s++) {
                emadapt_conv_desc dd = *d;
                dd.B = d->B - b0 < bs ? d->B - b0 : bs;
                // (the built-in dropout stream is a function of (seed, element index in the launch): every slice draws from a seed of its own)
                rc = emadapt_conv_fwd(&dd, x + b0 * hw * d->Cin, w, bias, y + b0 * hw * d->Cout, epilogue, keep_prob,
                                      drop_mask ? drop_mask + b0 * hw * d->Cout : nullptr, drop_seed + 0x9E3779B97F4A7C15ULL * (uint64_t)s,
                                      workspace, workspace_bytes, stream_);
                if (rc) return rc;
            }
            return EMADAPT_OK;
        }
    }
    if (rc) return rc;
    if (!x || !w || !y) return EMADAPT_ERR_INVALID;
    if (epilogue < EMADAPT_EPI_NONE || epilogue > EMADAPT_EPI_BIAS_RELU) return EMADAPT_ERR_INVALID;
    if (epilogue != EMADAPT_EPI_NONE && !bias) return EMADAPT_ERR_INVALID;
    if (!(keep_prob > 0.f) || keep_prob > 1.f) return EMADAPT_ERR_INVALID;
    if (keep_prob < 1.f && epilogue != EMADAPT_EPI_BIAS_RELU) return EMADAPT_ERR_INVALID;
    if (!aligned16(x) || !aligned16(w) || !aligned16(y) || !aligned16(workspace)) return EMADAPT_ERR_INVALID;
    if (d->B == 0) return EMADAPT_OK;
    IgemmArgs a;
    fill_geometry(a, d, false);
    a.x = x; a.w = w; a.y = y; a.bias = bias; a.gate = nullptr; a.mask = drop_mask;
    a.epi = epilogue; a.scale = 1.f; set_keep(a, keep_prob); a.seed = drop_seed;
    hipStream_t stream = as_stream(stream_);
    ProfScope prof(EMADAPT_K_IGEMM_FWD, 2.0 * a.M * (double)a.K * a.N, 4.0 * ((double)a.M * a.Cx + (double)a.K * a.N + (double)a.M * a.N), stream);   // bytes: every operand once
    return dispatch_igemm<false>(a, workspace, workspace_bytes, stream, prof);
}

int emadapt_conv_dgrad(const emadapt_conv_desc *d, const float *dy, const float *w, float *dx, const float *gate,
                       float gate_scale, void *workspace, size_t workspace_bytes, void *stream_) {
    int rc = check_desc(d);
    if (rc == EMADAPT_ERR_UNSUPPORTED && dy && dx) {   // a tensor of 2 GiB or more: slices of whole images
        const int bs = slice_batch(d, false);
        if (bs > 0 && bs < d->B) {
            const long long hw = (long long)d->H * d->W;
            for (int b0 = 0; b0 < d->B; b0 += bs) {
                emadapt_conv_desc dd = *d;
                dd.B = d->B - b0 < bs ? d->B - b0 : bs;
                rc = emadapt_conv_dgrad(&dd, dy + b0 * hw * d->Cout, w, dx + b0 * hw * d->Cin, gate ? gate + b0 * hw * d->Cin : nullptr,
                                        gate_scale, workspace, workspace_bytes, stream_);
                if (rc) return rc;
            }
            return EMADAPT_OK;
        }
    }
    if (rc) return rc;
    if (!dy || !w || !dx) return EMADAPT_ERR_INVALID;
    if (!aligned16(dy) || !aligned16(w) || !aligned16(dx) || !aligned16(workspace)) return EMADAPT_ERR_INVALID;
    if (d->B == 0) return EMADAPT_OK;
    IgemmArgs a;
    fill_geometry(a, d, true);
    a.x = dy; a.w = w; a.y = dx; a.bias = nullptr; a.gate = gate; a.mask = nullptr;
    a.epi = EMADAPT_EPI_NONE; a.scale = gate_scale; set_keep(a, 1.f); a.seed = 0;
    hipStream_t stream = as_stream(stream_);
    ProfScope prof(EMADAPT_K_IGEMM_DGRAD, 2.0 * a.M * (double)a.K * a.N, 4.0 * ((double)a.M * a.Cx + (double)a.K * a.N + (double)a.M * a.N * (gate ? 2.0 : 1.0)), stream);
    return dispatch_igemm<true>(a, workspace, workspace_bytes, stream, prof);
}

}  // extern "C"

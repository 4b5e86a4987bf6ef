// Filter gradient of the (dilated) convolutions on the gfx950 matrix cores.
//
// Replaces the Conv2DBackpropFilter half of the autodiff of deeplab.py:192:
//
//   dw[k][n] = sum_m A[m][k] * dy[m][n]        k = (tap, ci), n = co, m = (b,oh,ow) pixel
//
// i.e. a GEMM whose reduction runs over the pixels.  Both operands are read in their natural memory
// layout -- the im2col slice A[m][k..k+127] is contiguous along the channel axis and dy[m][n..] is a
// row of the NHWC gradient -- and both land in LDS as [pixel][k or n] tiles with plain 16-byte stores;
// the MFMA operand reads (lane&31 along k/n, lane>>5 = pixel parity) are conflict-free ds_read_b32.
// Every thread's k quad (tap, ci) is fixed for the whole kernel, only the pixel advances.
//
// Layers whose [K x N] filter has few 128x128 tiles (conv1_x .. conv5_x) would leave most of the 256
// CUs idle, so the pixel range is split across workgroups ("split-M"); each split writes an fp32
// partial slab and a second kernel adds the slabs in a fixed order -- deterministic, no atomics.
#include "common.h"

namespace emadapt {
int launch_slab_sum(const float *slab, int nslab, long long n, float *out, hipStream_t stream);
int conv_pad_before(int k, int dil);
}  // namespace emadapt

namespace {

using namespace emadapt;

typedef float f32x16 __attribute__((ext_vector_type(16)));

struct WgradArgs {
    const float *x;   // [B,H,W,Cin]
    const float *dy;  // [M][N]
    float *out;       // dw or slab base: [splits][K][N]
    int H, W, Cin, N, KW, dil, pad_h, pad_w;
    int M, K, ntaps;
    int tiles_k, tiles_n, splits;
    int rows_per_split;  // multiple of BMR
    unsigned x_bytes, y_bytes;      // extents for the bounds-checked buffer loads (< 2 GiB each)
    unsigned magic_w, magic_h;      // ceil(2^32/W), ceil(2^32/H): pixel -> (oh, ow) by v_mul_hi_u32
    int skip;                       // wgrad2_kernel<UT>: walk only the pixels at which the tile's tap is inside the image
    unsigned char korder[256];      // with skip: k tile run by the workgroups of logical k index i (balances the taps' unequal
                                    // pixel counts over the XCDs: see order_k_tiles)
    float *bias_part;               // wgrad_ws_kernel: [splits][4 producer waves][N] column sums of dy (BiasAddGrad partials), or null
};

// 16-byte buffer load with hardware range checking (offsets with bit 31 set are out of range -> zeros).
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float4 buf_ld4(__amdgpu_buffer_rsrc_t r, unsigned byte_off) {
    const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, (int)byte_off, 0, 0);
    return make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
}

constexpr int BMR = 32;  // pixels per reduction step

// Wave-specialised (see igemm_ws_kernel in conv_igemm.hip): 512 threads = 4 consumer wavefronts (LDS fragment reads +
// MFMA only) + 4 producer wavefronts (gather of x, rows of dy, LDS stores); one barrier per reduction step.
template <int BN>
__global__ void __launch_bounds__(512, 4) wgrad_ws_kernel(WgradArgs a) {
    constexpr int BKO = 128;
    constexpr int WK = BKO / 2, WN = BN / 2;
    constexpr int TK = WK / 32, TN = WN / 32;
    constexpr int XQ = BKO / 4, X_RSTEP = 256 / XQ, X_N = BMR / X_RSTEP;
    constexpr int YQ = BN / 4, Y_RSTEP = 256 / YQ, Y_N = BMR / Y_RSTEP;
    constexpr int NKK = BMR / 2;

    extern __shared__ __attribute__((aligned(16))) float lds[];
    float *Xs = lds;                    // [2][BMR][BKO]
    float *Ys = lds + 2 * BMR * BKO;    // [2][BMR][BN]

    // XCD-aware block order (blocks b, b+8, .. share an XCD and its L2): each XCD takes a contiguous run of logical
    // blocks in split-major order -- so the workgroups of one XCD stream the SAME pixel range -- and inside a split
    // the filter tiles go in groups of 8 k-tiles x all n-tiles (k fastest): 64 co-resident workgroups then share
    // 8 x-panels and 8 dy-panels instead of 2 and 32.
    int tile_k, tile_n, split;
    {
        const int ntile = a.tiles_k * a.tiles_n;
        const int nblk = ntile * a.splits;
        const int bid = blockIdx.x;
        const int xcd = bid & 7, q = nblk >> 3, r = nblk & 7;
        int t = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
        split = t / ntile;
        t -= split * ntile;
        constexpr int GK = 8;
        const int per_group = GK * a.tiles_n;
        const int g = t / per_group;
        const int first_k = g * GK;
        const int gsz = (a.tiles_k - first_k < GK) ? (a.tiles_k - first_k) : GK;
        const int in_g = t - g * per_group;
        tile_k = first_k + in_g % gsz;
        tile_n = in_g / gsz;
    }
    const int k0 = tile_k * BKO, n0 = tile_n * BN;
    const int m_begin = split * a.rows_per_split;
    int m_end = m_begin + a.rows_per_split;
    if (m_end > a.M) m_end = a.M;
    const int nsteps = (m_end > m_begin) ? (m_end - m_begin + BMR - 1) / BMR : 0;

    if (threadIdx.x >= 256) {
        // ================================ PRODUCER wavefronts ================================
        const int tid = threadIdx.x - 256;
        const int x_kq = tid % XQ, x_r0 = tid / XQ;
        const int k = k0 + x_kq * 4;
        const bool kvalid = k < a.K;
        const int tap = kvalid ? k / a.Cin : 0;
        const int c = kvalid ? k % a.Cin : 0;
        const int dh = (tap / a.KW) * a.dil - a.pad_h;
        const int dw = (tap % a.KW) * a.dil - a.pad_w;
        const int x_off = (dh * a.W + dw) * a.Cin + c;
        const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(a.x), 0, (int)a.x_bytes, 0x00020000);
        const __amdgpu_buffer_rsrc_t yr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(a.dy), 0, (int)a.y_bytes, 0x00020000);
        constexpr unsigned kOOB = 0x80000000u;
        const int y_nq = tid % YQ, y_r0 = tid / YQ;
        const int y_n = n0 + y_nq * 4;
        const bool nvalid = y_n < a.N;
        float4 rx[X_N], ry[Y_N];
        // BiasAddGrad for free (round 4): the producers of the workgroups with k tile 0 see every dy row of their pixel range pass
        // through their registers exactly once; each thread keeps the column sums of its four columns (the separate bias-gradient
        // kernel re-read 1.4 GB of conv1_x / conv2_x gradients per step for the same numbers)
        const bool do_bias = a.bias_part != nullptr && tile_k == 0;
        float4 bsum = make_float4(0.f, 0.f, 0.f, 0.f);
        auto load_tile = [&](int mb) {
#pragma unroll
            for (int j = 0; j < X_N; j++) {
                const int m = mb + x_r0 + j * X_RSTEP;
                const unsigned t2 = div_magic((unsigned)m, a.magic_w);
                const int ow = m - (int)t2 * a.W;
                const int oh = (int)(t2 - div_magic(t2, a.magic_h) * (unsigned)a.H);
                const int ih = oh + dh, iw = ow + dw;
                const bool ok = kvalid && m < m_end && (unsigned)ih < (unsigned)a.H && (unsigned)iw < (unsigned)a.W;
                rx[j] = buf_ld4(xr, ((unsigned)(m * a.Cin + x_off) * 4u) | (ok ? 0u : kOOB));
            }
#pragma unroll
            for (int j = 0; j < Y_N; j++) {
                const int m = mb + y_r0 + j * Y_RSTEP;
                ry[j] = buf_ld4(yr, ((unsigned)(m * a.N + y_n) * 4u) | ((nvalid && m < m_end) ? 0u : kOOB));
            }
        };
        auto store_tile = [&](int buf) {
#pragma unroll
            for (int j = 0; j < X_N; j++)
                *reinterpret_cast<float4 *>(&Xs[buf * BMR * BKO + (x_r0 + j * X_RSTEP) * BKO + x_kq * 4]) = rx[j];
#pragma unroll
            for (int j = 0; j < Y_N; j++)
                *reinterpret_cast<float4 *>(&Ys[buf * BMR * BN + (y_r0 + j * Y_RSTEP) * BN + y_nq * 4]) = ry[j];
            if (do_bias) {   // (rows past the pixel range arrive as zeros)
#pragma unroll
                for (int j = 0; j < Y_N; j++) { bsum.x += ry[j].x; bsum.y += ry[j].y; bsum.z += ry[j].z; bsum.w += ry[j].w; }
            }
        };
        load_tile(m_begin);
        store_tile(0);
        load_tile(m_begin + BMR);  // rows >= m_end are zero-filled by the range check
        __syncthreads();
        for (int st = 0; st < nsteps; st++) {
            store_tile((st + 1) & 1);
            load_tile(m_begin + (st + 2) * BMR);
            __syncthreads();
        }
        if (do_bias) {
            // the lanes of a wave that hold the same four columns (lane % YQ) add up, lane < YQ writes the wave's row of partials;
            // waves and splits are added by the second stage in a fixed order
#pragma unroll
            for (int off = 32; off >= YQ; off >>= 1) {
                bsum.x += __shfl_xor(bsum.x, off, 64); bsum.y += __shfl_xor(bsum.y, off, 64);
                bsum.z += __shfl_xor(bsum.z, off, 64); bsum.w += __shfl_xor(bsum.w, off, 64);
            }
            if ((tid & 63) < YQ && nvalid)
                *reinterpret_cast<float4 *>(a.bias_part + ((long long)split * 4 + (tid >> 6)) * a.N + y_n) = bsum;
        }
        return;
    }

    // ================================ CONSUMER wavefronts ================================
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wk = wave >> 1, wn = wave & 1;
    const int l31 = lane & 31, lh = lane >> 5;
    f32x16 acc[TK][TN];
#pragma unroll
    for (int i = 0; i < TK; i++)
#pragma unroll
        for (int j = 0; j < TN; j++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[i][j][r] = 0.f;
    __syncthreads();
    for (int st = 0; st < nsteps; st++) {
        const int cur = st & 1;
        // a wave's 32-wide tiles sit 64 floats apart (k: wk*32 + i*64, n: wn*32 + j*64), not side by side: the two reads of a
        // k-pair are then ONE ds_read2st64_b32 whose row rides in the 8-bit offsets -- with adjacent tiles (ds_read2_b32,
        // offsets in dwords up to 255) every k-pair paid a v_add for its row address, one vector instruction per two MFMAs
        const float *xs = Xs + cur * BMR * BKO + wk * 32 + l31;
        const float *ys = Ys + cur * BMR * BN + wn * 32 + l31;
        float av[2][TK], bv[2][TN];
#pragma unroll
        for (int i = 0; i < TK; i++) av[0][i] = xs[lh * BKO + i * 64];
#pragma unroll
        for (int j = 0; j < TN; j++) bv[0][j] = ys[lh * BN + j * 64];
#pragma unroll
        for (int kk = 0; kk < NKK; kk++) {
            if (kk + 1 < NKK) {
#pragma unroll
                for (int i = 0; i < TK; i++) av[(kk + 1) & 1][i] = xs[((kk + 1) * 2 + lh) * BKO + i * 64];
#pragma unroll
                for (int j = 0; j < TN; j++) bv[(kk + 1) & 1][j] = ys[((kk + 1) * 2 + lh) * BN + j * 64];
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < TK; i++)
#pragma unroll
                for (int j = 0; j < TN; j++)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[kk & 1][i], bv[kk & 1][j], acc[i][j], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
        __syncthreads();
    }
    float *out = a.out + (long long)split * a.K * a.N;
#pragma unroll
    for (int i = 0; i < TK; i++)
#pragma unroll
        for (int j = 0; j < TN; j++) {
            const int col = n0 + wn * 32 + j * 64 + l31;
            const bool col_ok = col < a.N;
#pragma unroll
            for (int r = 0; r < 16; r++) {
                const int row = k0 + wk * 32 + i * 64 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                if (col_ok && row < a.K) out[(long long)row * a.N + col] = acc[i][j][r];
            }
        }
}


// ------------------------------------------------------------------------------------------------------------------
// Second generation (see igemm_v2.h for the measurements behind it): 256(k) x 256(n) filter tiles, 8 MFMA waves with a
// 64 x 128 accumulator block each + 4 loader waves, one workgroup per CU.  Both operand tiles lie in LDS as they lie
// in memory, [pixel][k] and [pixel][n] -- the reduction index is the slow one, so ONE ds_read_b64 / ds_read_b128 with an
// immediate offset brings a lane's operands of a k-pair (its tile rows are k = l31*2 + i, its tile columns n = l31*4 + j:
// adjacent, not 32 apart) -- and are filled by LDS-DMA, 16 pixels per step, through a ring of 4 buffers fetched 3 steps
// ahead.  A loader instruction is one pixel row of 256 k (resp. n): the pixel's (oh, ow) is wave-uniform (SALU).
// UT ("uniform tap", Cin % 256 == 0: every layer this kernel is chosen for): the 256 k of a tile are one tap's channels, so
// the tap's offset and its validity at the pixel are scalars too -- the per-lane offset is a constant, the row's offset rides
// in the instruction's scalar operand and an out-of-image tap selects an empty buffer descriptor (all lanes read zeros): the
// loaders issue no vector-ALU instruction per step.  Otherwise each lane adds its own tap offset and bounds check (4 VALU
// per pixel row).  The MFMA waves' loop is unrolled over the ring like igemm2_kernel's (LDS addresses = register + immediate).
typedef __attribute__((address_space(3))) void lds_void_t;

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

constexpr int W2_BMR = 16, W2_NBUF = 4, W2_DIST = 3, W2_CONS = 8, W2_PW = 4;

template <bool UT>
__global__ void __launch_bounds__((W2_CONS + W2_PW) * 64, 3) wgrad2_kernel(WgradArgs a) {
    constexpr int BKO = 256, BN = 256, BMR = W2_BMR, NBUF = W2_NBUF, DIST = W2_DIST, CONS = W2_CONS, PW = W2_PW;
    constexpr int TK = 2, TN = 4;
    constexpr int X_BUF = BMR * BKO, Y_BUF = BMR * BN;
    constexpr int IPW = BMR / PW;   // pixel rows (= DMA instructions) per loader wave, step and operand
    static_assert(NBUF >= DIST + 1 && DIST >= 2 && (DIST - 2) * 2 * IPW <= 63 && (BMR / 2) % 2 == 0, "ring depth / vmcnt range");

    extern __shared__ __attribute__((aligned(16))) float lds[];
    float *Xs = lds;                  // [NBUF][BMR][BKO]
    float *Ys = lds + NBUF * X_BUF;   // [NBUF][BMR][BN]

    // XCD-aware block order, as wgrad_ws_kernel
    int tile_k, tile_n, split;
    {
        const int ntile = a.tiles_k * a.tiles_n;
        const int nblk = ntile * a.splits;
        const int bid = blockIdx.x;
        const int xcd = bid & 7, q = nblk >> 3, r = nblk & 7;
        int t = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
        split = t / ntile;
        t -= split * ntile;
        constexpr int GK = 8;
        const int per_group = GK * a.tiles_n;
        const int g = t / per_group;
        const int first_k = g * GK;
        const int gsz = (a.tiles_k - first_k < GK) ? (a.tiles_k - first_k) : GK;
        const int in_g = t - g * per_group;
        tile_k = first_k + in_g % gsz;
        tile_n = in_g / gsz;
    }
    if (UT && a.skip) {
        // Pixel skipping makes the tiles unequal (a corner tap of a dilated filter walks fewer pixels than a central one), so
        // the block order is the tiles sorted by work, heaviest first (a.korder), with no XCD regrouping: the hardware deals
        // consecutive blocks to the XCDs in turn -- every XCD gets an even sample, in descending order, and a CU that
        // finishes a light tile of the first round picks up the heaviest one still queued (longest-processing-time order).
        const int per_k = a.tiles_n * a.splits;
        const int kr = (int)blockIdx.x / per_k, r = (int)blockIdx.x - kr * per_k;
        tile_k = a.korder[kr];
        tile_n = r / a.splits;
        split = r - tile_n * a.splits;
    }
    const int k0 = tile_k * BKO, n0 = tile_n * BN;
    // Pixels this workgroup reduces over.  UT with a.skip: a tile's 256 k are one tap's channels, and the pixels at which that tap
    // reads inside the image form a box [h0, h1) x [w0, w1) of every image -- outside it the x operand is all zeros.  The
    // reduction walks a compact index i over the boxes only (fc6, 4x4 rate 4 on 41x41 maps: 81 % of the pixel-tap pairs;
    // LargeFOV's 3x3 rate 12: 65 %); i -> (image, row, column) is scalar arithmetic in the loaders.
    const int tap_s = k0 / a.Cin;
    const int dh_s = (tap_s / a.KW) * a.dil - a.pad_h, dw_s = (tap_s % a.KW) * a.dil - a.pad_w;
    const bool compact = UT && a.skip != 0;
    int h0 = 0, w0 = 0, vw = a.W, pv = a.H * a.W;
    if (compact) {
        h0 = dh_s < 0 ? -dh_s : 0;
        w0 = dw_s < 0 ? -dw_s : 0;
        const int h1 = dh_s > 0 ? a.H - dh_s : a.H, w1 = dw_s > 0 ? a.W - dw_s : a.W;
        const int vh = h1 > h0 ? h1 - h0 : 0;
        vw = w1 > w0 ? w1 - w0 : 0;
        pv = vh * vw;
    }
    const int m_total = compact ? (a.M / (a.H * a.W)) * pv : a.M;
    const int rps = compact ? (int)((((long long)m_total + a.splits - 1) / a.splits + W2_BMR - 1) / W2_BMR) * W2_BMR : a.rows_per_split;
    const int m_begin = __builtin_amdgcn_readfirstlane(split * rps);
    int m_end = m_begin + rps;
    if (m_end > m_total) m_end = m_total;
    m_end = __builtin_amdgcn_readfirstlane(m_end);
    const int nsteps = (m_end > m_begin) ? (m_end - m_begin + BMR - 1) / BMR : 0;

    if (threadIdx.x >= CONS * 64) {
        // ================================ LOADER wavefronts ================================
        const int lane = threadIdx.x & 63;
        const int pw = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6) - CONS);
        const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(a.x), 0, (int)a.x_bytes, 0x00020000);
        const __amdgpu_buffer_rsrc_t yr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(a.dy), 0, (int)a.y_bytes, 0x00020000);
        constexpr unsigned kOOB = 0x80000000u;
        // this lane's k quad: fixed (tap, channel) for the whole kernel
        const int k = k0 + lane * 4;
        const bool kvalid = k < a.K;
        const int tap = kvalid ? k / a.Cin : 0;
        const int c = kvalid ? k % a.Cin : 0;
        const int dh = (tap / a.KW) * a.dil - a.pad_h;
        const int dw = (tap % a.KW) * a.dil - a.pad_w;
        const unsigned x_off = (unsigned)(((dh * a.W + dw) * a.Cin + c) * 4);
        const int n = n0 + lane * 4;
        const unsigned y_off = ((unsigned)n * 4u) | (n < a.N ? 0u : kOOB);
        // UT: scalars of this tile's tap, per-lane constants
        const int c0_s = k0 - tap_s * a.Cin;
        // compact index -> (image, row, column): divisions by the box sizes as multiply-high (the reciprocals cost one real
        // division each, once per kernel)
        const unsigned magic_pv = compact && pv > 1 ? (unsigned)__builtin_amdgcn_readfirstlane((int)((0x100000000ULL + (unsigned)pv - 1) / (unsigned)pv)) : 0u;
        const unsigned magic_vw = compact && vw > 1 ? (unsigned)__builtin_amdgcn_readfirstlane((int)((0x100000000ULL + (unsigned)vw - 1) / (unsigned)vw)) : 0u;
        const unsigned xv_c = ((unsigned)(c0_s + lane * 4) * 4u) | (kvalid ? 0u : kOOB);
        const __amdgpu_buffer_rsrc_t nullr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(a.x), 0, 0, 0x00020000);
        auto fill = [&](int buf, int st) {
            float *xs = Xs + buf * X_BUF, *ys = Ys + buf * Y_BUF;
#pragma unroll
            for (int j = 0; j < IPW; j++) {
                const int row = j * PW + pw;                       // pixel row of the step (wave-uniform)
                if constexpr (UT) {
                    const int i = __builtin_amdgcn_readfirstlane(m_begin + st * BMR + row);
                    const bool live = i < m_end;
                    int m, oh, ow;
                    if (compact) {
                        const unsigned img = div_magic((unsigned)i, magic_pv);
                        const unsigned r = (unsigned)i - img * (unsigned)pv;
                        const unsigned rh = div_magic(r, magic_vw);
                        oh = h0 + (int)rh;
                        ow = w0 + (int)(r - rh * (unsigned)vw);
                        m = ((int)img * a.H + oh) * a.W + ow;
                    } else {
                        m = i;
                        const unsigned t2 = div_magic((unsigned)m, a.magic_w);
                        ow = m - (int)t2 * a.W;
                        oh = (int)(t2 - div_magic(t2, a.magic_h) * (unsigned)a.H);
                    }
                    const bool ok = live && (unsigned)(oh + dh_s) < (unsigned)a.H && (unsigned)(ow + dw_s) < (unsigned)a.W;
                    const unsigned sx = ok ? (unsigned)((m + dh_s * a.W + dw_s) * a.Cin) * 4u : 0u;
                    const unsigned sy = live ? (unsigned)(m * a.N) * 4u : 0u;
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(ok ? xr : nullr, (lds_void_t *)(xs + row * BKO), 16, (int)xv_c, (int)sx, 0, 0);
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(live ? yr : nullr, (lds_void_t *)(ys + row * BN), 16, (int)y_off, (int)sy, 0, 0);
                    continue;
                }
                const int m = m_begin + st * BMR + row;
                const unsigned t2 = div_magic((unsigned)m, a.magic_w);   // pixel -> (oh, ow): scalar arithmetic
                const int ow = m - (int)t2 * a.W;
                const int oh = (int)(t2 - div_magic(t2, a.magic_h) * (unsigned)a.H);
                const bool live = m < m_end;
                const int ih = oh + dh, iw = ow + dw;
                const bool ok = live && kvalid && (unsigned)ih < (unsigned)a.H && (unsigned)iw < (unsigned)a.W;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(xr, (lds_void_t *)(xs + row * BKO), 16,
                                                         (int)(((unsigned)(m * a.Cin) * 4u + x_off) | (ok ? 0u : kOOB)), 0, 0, 0);
                __builtin_amdgcn_raw_ptr_buffer_load_lds(yr, (lds_void_t *)(ys + row * BN), 16,
                                                         (int)(live ? (unsigned)(m * a.N) * 4u + y_off : kOOB), 0, 0, 0);
            }
        };
#pragma unroll
        for (int d = 0; d < DIST; d++)
            if (d < nsteps) fill(d, d);
        // (the consumers read the first fragments of step st + 1 during step st: the barrier that ends step st - 1 must
        // already cover step st + 1, so the loaders wait for all but their DIST - 2 youngest fills; see igemm_v2.h)
        if (nsteps >= DIST) wait_vmcnt<(DIST - 2) * 2 * IPW>();
        else wait_vmcnt<0>();
        __builtin_amdgcn_s_barrier();
        for (int st = 0; st < nsteps; st++) {
            if (st + DIST < nsteps) {
                fill((st + DIST) % NBUF, st + DIST);
                wait_vmcnt<(DIST - 2) * 2 * IPW>();   // step st + 2 has landed
            } else {
                wait_vmcnt<0>();
            }
            __builtin_amdgcn_s_barrier();
        }
        return;
    }

    // ================================ CONSUMER wavefronts ================================
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wk = wave & 3, wn = wave >> 2;
    const int l31 = lane & 31, lh = lane >> 5;
    f32x16 acc[TK][TN];
#pragma unroll
    for (int i = 0; i < TK; i++)
#pragma unroll
        for (int j = 0; j < TN; j++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[i][j][r] = 0.f;
    // per-lane fragment addresses (floats from the start of LDS); the y one is opaque to the optimiser so that ring position and
    // row stay in the instructions' 16-bit offset fields instead of one address register per combination (see igemm_v2.h)
    const int x_idx = lh * BKO + wk * 64 + l31 * TK;
    int y_idx = NBUF * X_BUF + lh * BN + wn * 128 + l31 * TN;
    asm volatile("" : "+v"(y_idx));
    // fragments are fetched in groups of two k-pairs (pixel rows 4g + lh and 4g + 2 + lh): one two-address 8-byte read for x,
    // two 16-byte reads for dy per sixteen MFMAs
    constexpr int NG = BMR / 4;
    float2 xv[2][2];
    float4 yv[2][2];
    auto read_g = [&](int buf, int g, int slot) {
        const float *xs = lds + buf * X_BUF + x_idx + g * 4 * BKO;
        const float *ys = lds + buf * Y_BUF + y_idx + g * 4 * BN;
#pragma unroll
        for (int h = 0; h < 2; h++) {
            xv[slot][h] = *reinterpret_cast<const float2 *>(xs + h * 2 * BKO);
            yv[slot][h] = *reinterpret_cast<const float4 *>(ys + h * 2 * BN);
        }
    };
    __syncthreads();
    if (nsteps > 0) read_g(0, 0, 0);
    auto k_step = [&](int cur, int nxt, bool more) {
#pragma unroll
        for (int g = 0; g < NG; g++) {
            if (g + 1 < NG) read_g(cur, g + 1, (g + 1) & 1);
            else if (more) read_g(nxt, 0, 0);   // first fragments of the next step (landed before the previous barrier)
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int h = 0; h < 2; h++) {
                const float2 x2 = xv[g & 1][h];
                const float4 y4 = yv[g & 1][h];
                acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(x2.x, y4.x, acc[0][0], 0, 0, 0);
                acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(x2.x, y4.y, acc[0][1], 0, 0, 0);
                acc[0][2] = __builtin_amdgcn_mfma_f32_32x32x2f32(x2.x, y4.z, acc[0][2], 0, 0, 0);
                acc[0][3] = __builtin_amdgcn_mfma_f32_32x32x2f32(x2.x, y4.w, acc[0][3], 0, 0, 0);
                acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(x2.y, y4.x, acc[1][0], 0, 0, 0);
                acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(x2.y, y4.y, acc[1][1], 0, 0, 0);
                acc[1][2] = __builtin_amdgcn_mfma_f32_32x32x2f32(x2.y, y4.z, acc[1][2], 0, 0, 0);
                acc[1][3] = __builtin_amdgcn_mfma_f32_32x32x2f32(x2.y, y4.w, acc[1][3], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        __syncthreads();
    };
    for (int st = 0; st < nsteps; st += NBUF) {
#pragma unroll
        for (int u = 0; u < NBUF; u++) {
            if (st + u >= nsteps) break;
            k_step(u, (u + 1) % NBUF, st + u + 1 < nsteps);
        }
    }
    // tile row of the 32x32 MFMA = (r&3) + 8*(r>>2) + 4*lh -> filter row k0 + wk*64 + row*2 + i; the lane's four
    // accumulator tiles j are four adjacent filter columns: 16-byte stores
    float *out = a.out + (long long)split * a.K * a.N;
    const int col = n0 + wn * 128 + l31 * TN;
    if (col >= a.N) return;
#pragma unroll
    for (int i = 0; i < TK; i++)
#pragma unroll
        for (int r = 0; r < 16; r++) {
            const int row = k0 + wk * 64 + ((r & 3) + 8 * (r >> 2) + 4 * lh) * TK + i;
            if (row < a.K)
                *reinterpret_cast<float4 *>(out + (long long)row * a.N + col) =
                    make_float4(acc[i][0][r], acc[i][1][r], acc[i][2][r], acc[i][3][r]);
        }
}

struct Plan {
    int gen, bn, tiles_k, tiles_n, splits, rows_per_split;   // gen 2: wgrad2_kernel (256x256 tiles, one workgroup per CU)
};

// Split-M plan.  T filter tiles x S pixel splits workgroups run in ceil(T*S/512) rounds on the 512 resident slots
// (256 CUs x 2); S is chosen to fill whole rounds (e.g. conv4_x/conv5_x: T = 144, S = 7 -> 1008 = 1.97 rounds, or 14 ->
// 3.94) while charging the extra pass over the S partial slabs (~3 TB/s) against the kernel time (~115 TFLOP/s).
Plan make_plan(const emadapt_conv_desc *d) {
    Plan p;
    const long long M = (long long)d->B * d->H * d->W;
    const int K = d->KH * d->KW * d->Cin;
    const int force = tune(TUNE_WGRAD_GEN);   // 0 = by shape, 1 / 2 force a kernel generation (tests)
    // the 256x256 tiles win ~3.3 points wherever they are not padded (conv3_1's K = 1152 = 4.5 tiles loses 3)
    const bool fits = d->Cout >= 256 && K >= 256 && ceil_div(K, 256) * 256 * 100 <= (long long)K * 104 &&
                      ceil_div(d->Cout, 256) * 256 * 100 <= (long long)d->Cout * 104;
    p.gen = force ? force : fits ? 2 : 1;
    const int bk = p.gen == 2 ? 256 : 128, bmr = p.gen == 2 ? W2_BMR : BMR;
    p.bn = p.gen == 2 ? 256 : (d->Cout <= 64 ? 64 : 128);
    p.tiles_k = (int)ceil_div(K, bk);
    p.tiles_n = (int)ceil_div(d->Cout, p.bn);
    const long long tiles = (long long)p.tiles_k * p.tiles_n;
    const double rate = p.gen == 2 ? 135e12 : 115e12;
    const double t_kernel = 2.0 * (double)M * (p.tiles_k * (double)bk) * (p.tiles_n * (double)p.bn) / rate;
    const double kn_bytes = (double)K * d->Cout * 4.0;
    long long best = 1;
    double best_t = 1e30;
    const long long max_s = ceil_div(M, 256) < 2048 ? ceil_div(M, 256) : 2048;  // >= 256 pixels per split
    const double slots = (p.gen == 2 ? 1.0 : 2.0) * plan_cus();
    for (long long s = 1; s <= max_s; s++) {
        const double rounds = (double)tiles * s / slots;
        const double eff = rounds / (double)(long long)(rounds + 0.999999);
        double t = t_kernel / eff + 2e-6 * (double)(long long)(rounds + 0.999999);  // + per-round turnover
        if (s > 1) t += (double)(s + 1) * kn_bytes / 3e12 + 3e-6;
        if (t < best_t * 0.995) { best_t = t; best = s; }
        if (rounds > 24.0) break;
    }
    long long rps = ceil_div(M, best);
    rps = ceil_div(rps, bmr) * bmr;
    p.rows_per_split = (int)rps;
    p.splits = (int)ceil_div(M, rps);
    return p;
}

int check_desc(const emadapt_conv_desc *d) {
    if (!d) return EMADAPT_ERR_INVALID;
    if (d->B < 0 || d->H <= 0 || d->W <= 0 || d->Cin <= 0 || d->Cout <= 0 || d->KH <= 0 || d->KW <= 0 || d->dil <= 0)
        return EMADAPT_ERR_INVALID;
    if (d->Cin % 4 != 0 || d->Cout % 4 != 0) return EMADAPT_ERR_UNSUPPORTED;
    if ((long long)d->B * d->H * d->W >= (1LL << 30)) return EMADAPT_ERR_UNSUPPORTED;
    const long long M = (long long)d->B * d->H * d->W;
    if (M * (d->Cin > d->Cout ? d->Cin : d->Cout) * 4 >= (1LL << 31)) return EMADAPT_ERR_UNSUPPORTED;  // 32-bit byte offsets
    if (M * (d->W > d->H ? d->W : d->H) >= (1LL << 32)) return EMADAPT_ERR_UNSUPPORTED;                 // magic division range
    return EMADAPT_OK;
}

// With pixel skipping a k tile's work is the pixel count of its tap's box: corner taps of a dilated filter walk fewer pixels
// than central ones (fc6: 35x35 .. 39x39 of 41x41).  korder = the k tiles sorted by that work, heaviest first (stable).
void order_k_tiles(WgradArgs &a) {
    const int nt = a.tiles_k;
    int work[256], idx[256];
    for (int t = 0; t < nt; t++) {
        const int tap = t * 256 / a.Cin;
        const int dh = (tap / a.KW) * a.dil - a.pad_h, dw = (tap % a.KW) * a.dil - a.pad_w;
        const int vh = a.H - (dh < 0 ? -dh : dh), vw = a.W - (dw < 0 ? -dw : dw);
        work[t] = (vh > 0 ? vh : 0) * (vw > 0 ? vw : 0);
        idx[t] = t;
    }
    for (int i = 1; i < nt; i++) {   // insertion sort, descending work, stable
        const int v = idx[i];
        int j = i - 1;
        while (j >= 0 && work[idx[j]] < work[v]) { idx[j + 1] = idx[j]; j--; }
        idx[j + 1] = v;
    }
    for (int i = 0; i < nt; i++) a.korder[i] = (unsigned char)idx[i];
}

template <int BN>
int launch_wgrad(const WgradArgs &a, hipStream_t stream) {
    constexpr size_t lds = sizeof(float) * 2 * BMR * (128 + BN);
    static DynLdsAttr attr;
    int rc = attr.ensure(reinterpret_cast<const void *>(&wgrad_ws_kernel<BN>), lds);
    if (rc) return rc;
    const long long nblk = (long long)a.tiles_k * a.tiles_n * a.splits;
    if (nblk > 0x7fffffffLL) return EMADAPT_ERR_UNSUPPORTED;
    hipLaunchKernelGGL((wgrad_ws_kernel<BN>), dim3((unsigned)nblk), dim3(512), lds, stream, a);
    return launch_status();
}

template <bool UT>
int launch_wgrad2(const WgradArgs &a, hipStream_t stream) {
    constexpr size_t lds = sizeof(float) * W2_NBUF * W2_BMR * (256 + 256);
    static DynLdsAttr attr;
    int rc = attr.ensure(reinterpret_cast<const void *>(&wgrad2_kernel<UT>), lds);
    if (rc) return rc;
    const long long nblk = (long long)a.tiles_k * a.tiles_n * a.splits;
    if (nblk > 0x7fffffffLL) return EMADAPT_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(wgrad2_kernel<UT>, dim3((unsigned)nblk), dim3((W2_CONS + W2_PW) * 64), lds, stream, a);
    return launch_status();
}

}  // namespace

namespace emadapt {
double exec_fraction_fp32_wgrad(const emadapt_conv_desc *d) {
    if (check_desc(d) != EMADAPT_OK || d->B == 0) return 1.0;
    const Plan p = make_plan(d);
    const long long M = (long long)d->B * d->H * d->W;
    const bool ut = d->Cin % 256 == 0 && tune(TUNE_FP32_LOADER) != 1;
    const bool skip = ut && d->KH * d->KW > 1 && tune(TUNE_FP32_LOADER) != 2 && M * d->H * d->W < (1LL << 32) && p.tiles_k <= 256;
    if (!skip || p.gen != 2) return 1.0;
    return taps_inside_fraction(d->H, d->W, d->KH, d->KW, d->dil, conv_pad_before(d->KH, d->dil), conv_pad_before(d->KW, d->dil));
}
}  // namespace emadapt

extern "C" {

// Images per launch when the tensors are too large for one (igemm_common.h slice_batch, with this file's check_desc)
static int slice_batch(const emadapt_conv_desc *d, bool) {
    if (!d || d->B <= 0) return 0;
    emadapt_conv_desc one = *d;
    one.B = 1;
    if (check_desc(&one) != EMADAPT_OK) return 0;
    const long long hw = (long long)d->H * d->W, cmax = d->Cin > d->Cout ? d->Cin : d->Cout;
    long long b = ((1LL << 30) - 1) / hw;
    const long long b2 = ((1LL << 31) - 1) / (hw * cmax * 4), b3 = ((1LL << 32) - 1) / (hw * (d->W > d->H ? d->W : d->H));
    b = b2 < b ? b2 : b;
    b = b3 < b ? b3 : b;
    return b < 1 ? 0 : (int)(b < d->B ? b : d->B);
}

// batch-sliced filter gradient (tensors of 2 GiB and more): workspace = [one slice's own workspace][slices x KN floats].  The ragged
// last slice (B % bs images) is planned on its own and the split count of a plan is not monotone in the pixel count: size for the
// LARGER of the two slice plans, so that no slice can find the workspace short after earlier slices were enqueued (advisor, round 4).
static size_t wgrad_slice_ws(const emadapt_conv_desc *d, int bs) {
    emadapt_conv_desc s = *d;
    s.B = bs;
    size_t w = emadapt_conv_wgrad_workspace_bytes(&s);
    if (d->B % bs != 0) {
        s.B = d->B % bs;
        const size_t w2 = emadapt_conv_wgrad_workspace_bytes(&s);
        w = w2 > w ? w2 : w;
    }
    return (w + 255) & ~(size_t)255;
}

size_t emadapt_conv_wgrad_workspace_bytes(const emadapt_conv_desc *d) {
    if (d && d->B > 0 && check_desc(d) == EMADAPT_ERR_UNSUPPORTED) {
        const int bs = slice_batch(d, true);
        if (bs > 0 && bs < d->B)
            return wgrad_slice_ws(d, bs) + (size_t)ceil_div(d->B, bs) * d->KH * d->KW * d->Cin * d->Cout * sizeof(float);
    }
    if (check_desc(d) != EMADAPT_OK || d->B == 0) return 0;
    const Plan p = make_plan(d);
    if (p.splits <= 1) return 0;
    return (size_t)p.splits * d->KH * d->KW * d->Cin * d->Cout * sizeof(float);
}

// BiasAddGrad fused into the filter gradient: possible where the round-1 kernel runs (its producers hold every dy row in registers)
// and the tensors fit one launch
static bool wgrad_fuses_bias(const emadapt_conv_desc *d) {
    return check_desc(d) == EMADAPT_OK && d->B > 0 && make_plan(d).gen == 1 && tune(TUNE_WGRAD_BIAS) != 1;
}
static size_t wgrad_bias_offset(const emadapt_conv_desc *d) { return (emadapt_conv_wgrad_workspace_bytes(d) + 255) & ~(size_t)255; }

int emadapt_conv_wgrad_bias_fused(const emadapt_conv_desc *d) { return wgrad_fuses_bias(d) ? 1 : 0; }

size_t emadapt_conv_wgrad_bias_workspace_bytes(const emadapt_conv_desc *d) {
    if (!wgrad_fuses_bias(d)) return 0;
    return wgrad_bias_offset(d) + (size_t)make_plan(d).splits * 4 * d->Cout * sizeof(float);
}

static int wgrad_run(const emadapt_conv_desc *d, const float *x, const float *dy, float *dw, float *db, void *workspace,
                     size_t workspace_bytes, void *stream_);

int emadapt_conv_wgrad(const emadapt_conv_desc *d, const float *x, const float *dy, float *dw, void *workspace,
                       size_t workspace_bytes, void *stream_) {
    return wgrad_run(d, x, dy, dw, nullptr, workspace, workspace_bytes, stream_);
}

int emadapt_conv_wgrad_bias(const emadapt_conv_desc *d, const float *x, const float *dy, float *dw, float *db, void *workspace,
                            size_t workspace_bytes, void *stream_) {
    if (!d || !db || (reinterpret_cast<uintptr_t>(db) & 15)) return EMADAPT_ERR_INVALID;
    if (!wgrad_fuses_bias(d)) return EMADAPT_ERR_UNSUPPORTED;
    if (!workspace || workspace_bytes < emadapt_conv_wgrad_bias_workspace_bytes(d)) return EMADAPT_ERR_WORKSPACE;
    return wgrad_run(d, x, dy, dw, db, workspace, workspace_bytes, stream_);
}

}  // extern "C"

static int wgrad_run(const emadapt_conv_desc *d, const float *x, const float *dy, float *dw, float *db, void *workspace,
                     size_t workspace_bytes, void *stream_) {
    int rc = check_desc(d);
    if (rc == EMADAPT_ERR_UNSUPPORTED && x && dy && dw) {   // a tensor of 2 GiB or more: per-slice gradients, added in slice order
        const int bs = slice_batch(d, true);
        if (bs > 0 && bs < d->B) {
            if (!workspace || workspace_bytes < emadapt_conv_wgrad_workspace_bytes(d) || (reinterpret_cast<uintptr_t>(workspace) & 15))
                return EMADAPT_ERR_WORKSPACE;
            const long long hw = (long long)d->H * d->W, KN = (long long)d->KH * d->KW * d->Cin * d->Cout;
            const size_t ws_slice = wgrad_slice_ws(d, bs);
            float *parts = reinterpret_cast<float *>(static_cast<char *>(workspace) + ws_slice);
            int s = 0;
            for (int b0 = 0; b0 < d->B; b0 += bs, s++) {
                emadapt_conv_desc dd = *d;
                dd.B = d->B - b0 < bs ? d->B - b0 : bs;
                rc = emadapt_conv_wgrad(&dd, x + b0 * hw * d->Cin, dy + b0 * hw * d->Cout, parts + (long long)s * KN, workspace, ws_slice, stream_);
                if (rc) return rc;
            }
            return launch_slab_sum(parts, s, KN, dw, as_stream(stream_));
        }
    }
    if (rc) return rc;
    if (!x || !dy || !dw) return EMADAPT_ERR_INVALID;
    if ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(dy) | reinterpret_cast<uintptr_t>(dw) |
         reinterpret_cast<uintptr_t>(workspace)) & 15)
        return EMADAPT_ERR_INVALID;
    hipStream_t stream = as_stream(stream_);
    const long long KN = (long long)d->KH * d->KW * d->Cin * d->Cout;
    if (d->B == 0) {
        EMADAPT_HIP_TRY(hipMemsetAsync(dw, 0, KN * sizeof(float), stream));
        return EMADAPT_OK;
    }
    const Plan p = make_plan(d);
    if (p.splits > 1 && (!workspace || workspace_bytes < emadapt_conv_wgrad_workspace_bytes(d)))
        return EMADAPT_ERR_WORKSPACE;
    WgradArgs a;
    a.x = x; a.dy = dy; a.out = p.splits > 1 ? static_cast<float *>(workspace) : dw;
    a.H = d->H; a.W = d->W; a.Cin = d->Cin; a.N = d->Cout; a.KW = d->KW; a.dil = d->dil;
    a.pad_h = conv_pad_before(d->KH, d->dil); a.pad_w = conv_pad_before(d->KW, d->dil);
    a.M = d->B * d->H * d->W; a.ntaps = d->KH * d->KW; a.K = a.ntaps * d->Cin;
    a.tiles_k = p.tiles_k; a.tiles_n = p.tiles_n; a.splits = p.splits; a.rows_per_split = p.rows_per_split;
    a.x_bytes = (unsigned)((long long)a.M * d->Cin * 4); a.y_bytes = (unsigned)((long long)a.M * d->Cout * 4);
    a.magic_w = magic_u32(d->W); a.magic_h = magic_u32(d->H);
    a.bias_part = db ? reinterpret_cast<float *>(static_cast<char *>(workspace) + wgrad_bias_offset(d)) : nullptr;   // (gen 1 only: emadapt_conv_wgrad_bias)
    ProfScope prof(EMADAPT_K_IGEMM_WGRAD, 2.0 * a.M * (double)a.K * a.N, 4.0 * ((double)a.M * d->Cin + (double)a.M * a.N + (double)a.K * a.N), stream);
    const bool ut = d->Cin % 256 == 0 && tune(TUNE_FP32_LOADER) != 1;   // a 256-wide k tile lies inside one tap
    // skip the pixels at which a tile's tap is outside the image (fp32_loader = 2 keeps the dense walk: tests, A/B); the
    // compact index is divided by multiply-high: exact while pixels x box size < 2^32
    a.skip = ut && a.ntaps > 1 && tune(TUNE_FP32_LOADER) != 2 && (long long)a.M * d->H * d->W < (1LL << 32) && p.tiles_k <= 256;
    if (a.skip) {
        order_k_tiles(a);
        if (p.gen == 2) prof.executed_fraction(taps_inside_fraction(d->H, d->W, d->KH, d->KW, d->dil, a.pad_h, a.pad_w));
    }
    rc = p.gen == 2 ? (ut ? launch_wgrad2<true>(a, stream) : launch_wgrad2<false>(a, stream)) : (p.bn == 64) ? launch_wgrad<64>(a, stream) : launch_wgrad<128>(a, stream);
    if (rc) return rc;
    if (p.splits > 1) rc = launch_slab_sum(static_cast<const float *>(workspace), p.splits, KN, dw, stream);
    if (!rc && db) rc = launch_slab_sum(a.bias_part, p.splits * 4, d->Cout, db, stream);   // splits x producer waves, in order
    return rc;
}

// fp32-grade convolution GEMMs on the bf16 matrix pipe ("bf16x6"; opt-in, compute_dtype = "bf16x6").
//
// gfx950's bf16 MFMA runs at 16x the rate of its fp32 MFMA.  An fp32 number splits EXACTLY into three bf16 pieces,
//     x = p0 + p1 + p2,   p0 = bf16(x), p1 = bf16(x - p0), p2 = bf16(x - p0 - p1)      (8 + 8 + 8 mantissa bits; the two
// subtractions are exact in fp32), a product of two bf16 numbers is exact in fp32, and
//     x*y = p0 q0 + (p0 q1 + p1 q0) + (p1 q1 + p0 q2 + p2 q0) + [p1 q2 + p2 q1 + p2 q2 <= 2^-25 |xy|]
// so SIX v_mfma_f32_32x32x16_bf16 per 16 k, accumulated in fp32 like the native instruction, reproduce the fp32 product
// to better than one fp32 rounding (the dropped terms are below 2^-25 relative; v_mfma_f32_32x32x2_f32 itself rounds
// every product-sum to 2^-24).  The kernels are held to the same parity bar as the native fp32 kernels (2e-5 of the
// tensor maximum against the fp64 oracle on unrounded inputs, tests/test_gpu_x6.py) at 16/6 = 2.7x the matrix rate.
// Everything else -- layouts, epilogues, split-K / split-M, determinism, fp32 storage of every tensor -- is as in the
// fp32 kernels.  The default compute path remains the native fp32 MFMA (conv_igemm.hip, conv_wgrad.hip).
//
// Structure (see conv_bf16.hip for the bf16 building blocks): wave-specialised 128x128 tiles, K steps of 16 (one MFMA k
// group x 6 plane pairs x 4 accumulator tiles = 24 MFMAs per wave and step), 4 MFMA + 4 loader waves, two workgroups per
// CU.  Both operands are gathered along k as fp32 float4s and split into their three planes by the loader waves
// (v_cvt_pk_bf16_f32 + two exact subtractions) on the way into LDS: tiles are [plane][row][k] bf16, 32-byte rows
// with a chunk swizzle (conflict-free ds_write_b64 and ds_read_b128).  The forward pass reads the filter from a k-contiguous fp32
// image [tap][co][ci] (emadapt_filter_kmajor, refreshed once per optimizer step); the data gradient reads the HWIO master
// copy directly (k = co is contiguous).  The filter gradient stores [plane][pixel][k|n] and gathers its fragments with
// ds_read_b64_tr_b16, 16 pixels per step.
#include <stdlib.h>

#include <type_traits>

#include "igemm_common.h"

namespace emadapt {
int launch_slab_sum(const float *slab, int nslab, long long n, float *out, hipStream_t stream);
}

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

// four fp32 values -> their three bf16 planes (4 x bf16 = 8 bytes each)
__device__ __forceinline__ void split3(const float4 v, uint2 &p0, uint2 &p1, uint2 &p2) {
    const float x[4] = {v.x, v.y, v.z, v.w};
    bf16x4 a, b, c;
#pragma unroll
    for (int e = 0; e < 4; e++) {
        a[e] = (__bf16)x[e];
        const float r1 = x[e] - (float)a[e];  // exact
        b[e] = (__bf16)r1;
        const float r2 = r1 - (float)b[e];    // exact
        c[e] = (__bf16)r2;
    }
    p0 = *reinterpret_cast<uint2 *>(&a);
    p1 = *reinterpret_cast<uint2 *>(&b);
    p2 = *reinterpret_cast<uint2 *>(&c);
}

// The six plane pairs, in the order their fragments arrive from LDS (planes are loaded 0, 1, 2), so the first MFMAs of a K
// step start after 4 of its 12 fragment loads.  The order has no bearing on accuracy: the accumulator already carries
// the running sum of the previous K steps.
#define X6_PAIRS(F) F(0, 0) F(0, 1) F(1, 0) F(1, 1) F(0, 2) F(2, 0)

// ------------------------------------------------------------------------------------------------------------------
// k-contiguous fp32 image of the filter for the forward pass: HWIO [tap][ci][co] -> [tap][co][ci]
// ------------------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) filter_kmajor_kernel(const float *__restrict__ w, float *__restrict__ wk, int Cin,
                                                            int Cout, int tiles_ci, int tiles_co) {
    __shared__ float tile[32][33];
    int bid = blockIdx.x;
    const int tco = bid % tiles_co;
    bid /= tiles_co;
    const int tci = bid % tiles_ci;
    const int tap = bid / tiles_ci;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const int ci0 = tci * 32, co0 = tco * 32;
#pragma unroll
    for (int r = 0; r < 4; r++) {
        const int ci = ci0 + ty + 8 * r, co = co0 + tx;
        tile[ty + 8 * r][tx] = (ci < Cin && co < Cout) ? w[((long long)tap * Cin + ci) * Cout + co] : 0.f;
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < 4; r++) {
        const int co = co0 + ty + 8 * r, ci = ci0 + tx;
        if (co < Cout && ci < Cin) wk[((long long)tap * Cout + co) * Cin + ci] = tile[tx][ty + 8 * r];
    }
}

// ------------------------------------------------------------------------------------------------------------------
// Forward / data gradient: out[m][n] = sum_k A[m][k] B[n][k]; B = a.w is [tap][N][Cx] fp32 with k contiguous.
// ------------------------------------------------------------------------------------------------------------------
// LDS rows of 16 bf16 = 32 bytes, unpadded; the two 16-byte chunks of a row are swapped when bit 3 of the row is set.
// Loaders: 16 consecutive lanes store 4 rows x 4 quads = 128 contiguous bytes (conflict-free ds_write_b64; rows of a
// group share bit 3).  MFMA waves: the 16-lane groups of ds_read_b128 cover each row residue mod 8 twice, with opposite
// bit 3 -> 16 distinct 16-byte slots.  (A padded 48-byte pitch cost a third of the LDS cycles in store conflicts.)
constexpr int kPitchX = 32;

// Tile BM x 128: BM = 128 (4 MFMA + 4 loader waves, two workgroups per CU) or BM = 256 (8 + 4 waves, one per CU).  The
// loaders' split arithmetic is per loaded element while the MFMA count grows with the tile area: 7 VALU instructions
// per MFMA at 128x128 (measured; enough to fill the SIMD issue ports) against ~3.8 at 256x128.
template <int BM>
__global__ void __launch_bounds__((BM / 64 * 2 + 4) * 64, BM == 128 ? 4 : 3) igemm_x6_kernel(IgemmArgs a) {
    constexpr int BN = 128, BK = 16, WM = 64, WN = 64;
    constexpr int TM = 2, TN = 2;
    constexpr int CTHREADS = (BM / WM) * 2 * 64;  // MFMA threads; the 256 loader threads follow
    constexpr int KQ = BK / 4;            // 4 fp32 quads per row
    constexpr int RSTEP = 256 / KQ;       // 64 rows per pass of the loaders
    constexpr int A_N = BM / RSTEP, B_N = BN / RSTEP;   // fp32 quads per thread and K step
    constexpr int PLANE_A = BM * kPitchX, PLANE_B = BN * kPitchX;   // bytes per plane
    constexpr int TILE_A = 3 * PLANE_A, TILE_B = 3 * PLANE_B;       // bytes per operand tile

    extern __shared__ __attribute__((aligned(16))) char ldsb[];
    char *As = ldsb;                 // [2][3][BM][kPitchX]
    char *Bs = ldsb + 2 * TILE_A;    // [2][3][BN][kPitchX]

    int tile_m, tile_n, split;
    {
        const int ntile = a.tiles_m * a.tiles_n;
        const int nblk = ntile * a.splits;
        const int bid = blockIdx.x;
        const int xcd = bid & 7, q = nblk >> 3, r = nblk & 7;
        int t = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
        split = t / ntile;
        t -= split * ntile;
        constexpr int GM = BM == 128 ? 8 : 4;
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

    if (threadIdx.x >= CTHREADS) {
        // ================================ LOADER wavefronts ================================
        const int tid = threadIdx.x - CTHREADS;
        const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(a.x), 0, (int)a.x_bytes, 0x00020000);
        const __amdgpu_buffer_rsrc_t wr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(a.w), 0, (int)a.w_bytes, 0x00020000);
        constexpr unsigned kOOB = 0x80000000u;
        const int kq = tid % KQ, r0 = tid / KQ;
        unsigned a_hw[A_N];  // (oh << 16 | ow); rows past m_end fail every bounds check
#pragma unroll
        for (int j = 0; j < A_N; j++) {
            const int m = m0 + r0 + j * RSTEP;
            if (m < a.m_end) {
                const int t2 = m / a.W;
                a_hw[j] = ((unsigned)(t2 % a.H) << 16) | (unsigned)(m - t2 * a.W);
            } else {
                a_hw[j] = 0x40000000u;
            }
        }
        const int a_base0 = (m0 + r0) * a.Cx, a_rstride = RSTEP * a.Cx;
        float4 ra[A_N], rb[B_N];
        int toff, dh, dw, boff;
        bool kvalid;
        auto step_setup = [&](int kt) {
            unsigned tap;
            int c;
            if (a.tap_inner) {  // Cx % BK == 0: the whole K step lies in one tap
                const unsigned chunk = __umulhi((unsigned)kt, a.magic_nt);
                tap = (unsigned)kt - chunk * (unsigned)a.ntaps;
                if ((int)chunk * BK >= a.Cx) tap = (unsigned)a.ntaps;
                c = (int)chunk * BK + kq * 4;
            } else {
                const unsigned k = (unsigned)(kt * BK + kq * 4);
                tap = __umulhi(k, a.magic_cx);
                c = (int)(k - tap * (unsigned)a.Cx);
            }
            const unsigned kh = div_magic(tap, a.magic_kw);
            const int kw = (int)(tap - kh * (unsigned)a.KW);
            kvalid = (int)tap < a.ntaps;
            dh = a.sign * ((int)kh * a.dil - a.pad_h);
            dw = a.sign * (kw * a.dil - a.pad_w);
            toff = a_base0 + (dh * a.W + dw) * a.Cx + c;
            boff = ((int)tap * a.N + n0 + r0) * a.Cx + c;
        };
        auto load_a = [&](int j) {
            const int ih = (int)(a_hw[j] >> 16) + dh, iw = (int)(a_hw[j] & 0xffffu) + dw;
            const bool ok = kvalid && (unsigned)ih < (unsigned)a.H && (unsigned)iw < (unsigned)a.W;
            ra[j] = buf_ld4(xr, ((unsigned)(toff + j * a_rstride) * 4u) | (ok ? 0u : kOOB));
        };
        auto load_b = [&](int j) {
            const int n = n0 + r0 + j * RSTEP;
            rb[j] = buf_ld4(wr, ((unsigned)(boff + j * RSTEP * a.Cx) * 4u) | ((kvalid && n < a.N) ? 0u : kOOB));
        };
        auto store3 = [&](char *base, int plane, int row, const float4 &v) {
            uint2 p0, p1, p2;
            split3(v, p0, p1, p2);
            char *d = base + row * kPitchX + (((kq >> 1) ^ ((row >> 3) & 1)) * 16) + (kq & 1) * 8;
            *reinterpret_cast<uint2 *>(d) = p0;
            *reinterpret_cast<uint2 *>(d + plane) = p1;
            *reinterpret_cast<uint2 *>(d + 2 * plane) = p2;
        };
        step_setup(kt0);
#pragma unroll
        for (int j = 0; j < A_N; j++) load_a(j);
#pragma unroll
        for (int j = 0; j < B_N; j++) load_b(j);
        step_setup(kt0 + 1);  // K steps past the end are zero-filled by the range check
#pragma unroll
        for (int j = 0; j < A_N; j++) { store3(As, PLANE_A, r0 + j * RSTEP, ra[j]); load_a(j); }
#pragma unroll
        for (int j = 0; j < B_N; j++) { store3(Bs, PLANE_B, r0 + j * RSTEP, rb[j]); load_b(j); }
        __syncthreads();
        // steady state: park register j (split into its planes) and re-arm it with the load of the step after next
        for (int kt = 0; kt < nk; kt++) {
            const int buf = (kt + 1) & 1;
            step_setup(kt0 + kt + 2);
#pragma unroll
            for (int j = 0; j < A_N; j++) { store3(As + buf * TILE_A, PLANE_A, r0 + j * RSTEP, ra[j]); load_a(j); }
#pragma unroll
            for (int j = 0; j < B_N; j++) { store3(Bs + buf * TILE_B, PLANE_B, r0 + j * RSTEP, rb[j]); load_b(j); }
            __syncthreads();
        }
        return;
    }

    // ================================ MFMA wavefronts ================================
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
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
        const int chunk = (lh ^ ((l31 >> 3) & 1)) * 16;   // swizzled chunk (tile bases are multiples of 16 rows)
        const char *as = As + cur * TILE_A + (wm * WM + l31) * kPitchX + chunk;
        const char *bs = Bs + cur * TILE_B + (wn * WN + l31) * kPitchX + chunk;
        bf16x8 av[3][TM], bv[3][TN];
#pragma unroll
        for (int p = 0; p < 3; p++) {
#pragma unroll
            for (int i = 0; i < TM; i++) av[p][i] = *reinterpret_cast<const bf16x8 *>(as + p * PLANE_A + i * 32 * kPitchX);
#pragma unroll
            for (int j = 0; j < TN; j++) bv[p][j] = *reinterpret_cast<const bf16x8 *>(bs + p * PLANE_B + j * 32 * kPitchX);
        }
#define X6_MFMA(P, Q)                                                                                              \
    _Pragma("unroll") for (int i = 0; i < TM; i++) _Pragma("unroll") for (int j = 0; j < TN; j++) acc[i][j] =      \
        __builtin_amdgcn_mfma_f32_32x32x16_bf16(av[P][i], bv[Q][j], acc[i][j], 0, 0, 0);
        X6_PAIRS(X6_MFMA)
#undef X6_MFMA
        __syncthreads();
    }
    constexpr int SLAB = 32 * (WN + 4);
    static_assert(SLAB * 4 * (CTHREADS / 64) <= 2 * (TILE_A + TILE_B), "epilogue slabs must fit in the tile LDS");
    float *slab = a.splits > 1 ? a.slab + ((long long)split * (a.m_end - a.m_begin) - a.m_begin) * a.N : nullptr;
    store_acc_tile<TM, TN, 32, false>(a, acc, reinterpret_cast<float *>(ldsb) + wave * SLAB, lane, m0 + wm * WM, n0 + wn * WN, slab);
}

// 128x128 tiles by default.  The 256x128 configuration (one workgroup per CU) is kept selectable and parity-tested, but
// does not pay for the forward / data-gradient kernels: with three planes per operand these kernels are bound by LDS
// traffic (SQ_LDS_IDX_ACTIVE ~ the kernel's cycles), which the taller tile reduces by only a quarter while halving the
// number of resident MFMA waves that can cover each other's LDS round trips.
bool x6_big(const IgemmArgs &a) {
    (void)a;
    return tune(TUNE_X6_TILE) == 256;  // 128|256 (the parity tests run both)
}

int x6_splits(const IgemmArgs &a) {
    return x6_big(a) ? choose_splits(a.M, a.N, a.K, 256, 128, 16, 256.0, 250e12)
                     : choose_splits(a.M, a.N, a.K, 128, 128, 16, 512.0, 200e12);
}

template <int BM>
int launch_igemm_x6(IgemmArgs &a, hipStream_t stream) {
    constexpr size_t lds = 2 * 3 * (BM + 128) * kPitchX;
    constexpr int threads = (BM / 64 * 2 + 4) * 64;
    static DynLdsAttr attr;
    {
        const int arc = attr.ensure(reinterpret_cast<const void *>(&igemm_x6_kernel<BM>), lds);
        if (arc) return arc;
    }
    a.tiles_m = (int)ceil_div(a.M, BM);
    a.tiles_n = (int)ceil_div(a.N, 128);
    const long long nblk = (long long)a.tiles_m * a.tiles_n * a.splits;
    if (nblk > 0x7fffffffLL) return EMADAPT_ERR_UNSUPPORTED;
    hipLaunchKernelGGL((igemm_x6_kernel<BM>), dim3((unsigned)nblk), dim3(threads), lds, stream, a);
    return launch_status();
}

int dispatch_igemm_x6(IgemmArgs &a, void *workspace, size_t workspace_bytes, hipStream_t stream) {
    int splits = x6_splits(a);
    if (splits > 1 && (!workspace || workspace_bytes < (size_t)splits * a.M * a.N * sizeof(float))) splits = 1;
    a.ablate = 0;
    a.tap_inner = (a.ntaps > 1 && a.Cx % 16 == 0) ? 1 : 0;
    a.m_begin = 0;
    a.m_end = a.M;
    a.splits = splits;
    a.slab = splits > 1 ? static_cast<float *>(workspace) : nullptr;
    int rc = x6_big(a) ? launch_igemm_x6<256>(a, stream) : launch_igemm_x6<128>(a, stream);
    if (rc || a.splits <= 1) return rc;
    long long g = ((long long)a.M * a.N / 4 + 255) / 256;
    if (g > 8192) g = 8192;
    hipLaunchKernelGGL(splitk_epilogue_kernel, dim3((unsigned)g), dim3(256), 0, stream, a);
    return launch_status();
}

// ------------------------------------------------------------------------------------------------------------------
// Filter gradient: dw[k][n] = sum_m A[m][k] dy[m][n]; 128 x BN filter tiles, 16 pixels per reduction step, split-M.
// ------------------------------------------------------------------------------------------------------------------
struct WgradX6Args {
    const float *x, *dy;
    float *out;
    int H, W, Cin, N, KW, dil, pad_h, pad_w;
    int M, K, ntaps;
    int tiles_k, tiles_n, splits;
    int rows_per_split;  // multiple of 16
    unsigned x_bytes, y_bytes;
    unsigned magic_w, magic_h;
};

__device__ __forceinline__ bf16x8 tr_frag(const char *p0, const char *p1) {
    typedef __attribute__((address_space(3))) bf16x4 lds_bf16x4;
    const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4 *)(p0));
    const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4 *)(p1));
    return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
}

template <int BKO, int BN>
__global__ void __launch_bounds__((BKO / 64 * 2 + 4) * 64, BKO == 128 ? 4 : 3) wgrad_x6_kernel(WgradX6Args a) {
    constexpr int BMR = 16;
    constexpr int WK = 64, WN = BN / 2;
    constexpr int TK = WK / 32, TN = WN / 32;
    constexpr int CTHREADS = (BKO / WK) * 2 * 64;
    constexpr int XQ = BKO / 4, X_RSTEP = 256 / XQ, X_N = BMR / X_RSTEP;  // BKO 128: 32, 8, 2; BKO 256: 64, 4, 4
    constexpr int YQ = BN / 4, Y_RSTEP = 256 / YQ, Y_N = BMR / Y_RSTEP;   // BN 128: 32, 8, 2; BN 64: 16, 16, 1
    constexpr int PX = BKO * 2 + 64, PY = BN * 2 + 64;                    // LDS row pitches in bytes
    constexpr int PLANE_X = BMR * PX, PLANE_Y = BMR * PY;
    constexpr int TILE_X = 3 * PLANE_X, TILE_Y = 3 * PLANE_Y;

    extern __shared__ __attribute__((aligned(16))) char ldsb[];
    char *Xs = ldsb;                // [2][3][BMR][PX]
    char *Ys = ldsb + 2 * TILE_X;   // [2][3][BMR][PY]

    int tile_k, tile_n, split;
    {
        const int ntile = a.tiles_k * a.tiles_n;
        const int nblk = ntile * a.splits;
        const int bid = blockIdx.x;
        const int xcd = bid & 7, q = nblk >> 3, r = nblk & 7;
        int t = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
        split = t / ntile;
        t -= split * ntile;
        constexpr int GK = BKO == 128 ? 8 : 4;
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

    if (threadIdx.x >= CTHREADS) {
        // ================================ LOADER wavefronts ================================
        const int tid = threadIdx.x - CTHREADS;
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
        auto load_x = [&](int mb, int j) {
            const int m = mb + x_r0 + j * X_RSTEP;
            const unsigned t2 = div_magic((unsigned)m, a.magic_w);
            const int ow = m - (int)t2 * a.W;
            const int oh = (int)(t2 - div_magic(t2, a.magic_h) * (unsigned)a.H);
            const int ih = oh + dh, iw = ow + dw;
            const bool ok = kvalid && m < m_end && (unsigned)ih < (unsigned)a.H && (unsigned)iw < (unsigned)a.W;
            rx[j] = buf_ld4(xr, ((unsigned)(m * a.Cin + x_off) * 4u) | (ok ? 0u : kOOB));
        };
        auto load_y = [&](int mb, int j) {
            const int m = mb + y_r0 + j * Y_RSTEP;
            ry[j] = buf_ld4(yr, ((unsigned)(m * a.N + y_n) * 4u) | ((nvalid && m < m_end) ? 0u : kOOB));
        };
        auto store3 = [&](char *d, int plane, const float4 &v) {
            uint2 p0, p1, p2;
            split3(v, p0, p1, p2);
            *reinterpret_cast<uint2 *>(d) = p0;
            *reinterpret_cast<uint2 *>(d + plane) = p1;
            *reinterpret_cast<uint2 *>(d + 2 * plane) = p2;
        };
        auto store_x = [&](int buf, int j) { store3(Xs + buf * TILE_X + (x_r0 + j * X_RSTEP) * PX + x_kq * 8, PLANE_X, rx[j]); };
        auto store_y = [&](int buf, int j) { store3(Ys + buf * TILE_Y + (y_r0 + j * Y_RSTEP) * PY + y_nq * 8, PLANE_Y, ry[j]); };
#pragma unroll
        for (int j = 0; j < X_N; j++) load_x(m_begin, j);
#pragma unroll
        for (int j = 0; j < Y_N; j++) load_y(m_begin, j);
#pragma unroll
        for (int j = 0; j < X_N; j++) { store_x(0, j); load_x(m_begin + BMR, j); }  // rows >= m_end: zero-filled
#pragma unroll
        for (int j = 0; j < Y_N; j++) { store_y(0, j); load_y(m_begin + BMR, j); }
        __syncthreads();
        for (int st = 0; st < nsteps; st++) {
            const int buf = (st + 1) & 1;
            const int mb = m_begin + (st + 2) * BMR;
#pragma unroll
            for (int j = 0; j < X_N; j++) { store_x(buf, j); load_x(mb, j); }
#pragma unroll
            for (int j = 0; j < Y_N; j++) { store_y(buf, j); load_y(mb, j); }
            __syncthreads();
        }
        return;
    }

    // ================================ MFMA wavefronts ================================
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wk = wave >> 1, wn = wave & 1;
    const int l31 = lane & 31, lh = lane >> 5;
    const int grp = lane >> 4, q = (lane >> 2) & 3, p = lane & 3;
    const int frow = (grp >> 1) * 8 + q;            // pixel row inside the 16-pixel step (+4 for the upper half)
    const int fcol = ((grp & 1) * 16 + p * 4) * 2;  // byte offset of the lane's 4 columns
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
        const char *xs = Xs + cur * TILE_X + frow * PX + fcol + (wk * WK) * 2;
        const char *ys = Ys + cur * TILE_Y + frow * PY + fcol + (wn * WN) * 2;
        bf16x8 av[3][TK], bv[3][TN];
#pragma unroll
        for (int pl = 0; pl < 3; pl++) {
#pragma unroll
            for (int i = 0; i < TK; i++) av[pl][i] = tr_frag(xs + pl * PLANE_X + i * 64, xs + pl * PLANE_X + i * 64 + 4 * PX);
#pragma unroll
            for (int j = 0; j < TN; j++) bv[pl][j] = tr_frag(ys + pl * PLANE_Y + j * 64, ys + pl * PLANE_Y + j * 64 + 4 * PY);
        }
#define X6_MFMA(P, Q)                                                                                              \
    _Pragma("unroll") for (int i = 0; i < TK; i++) _Pragma("unroll") for (int j = 0; j < TN; j++) acc[i][j] =      \
        __builtin_amdgcn_mfma_f32_32x32x16_bf16(av[P][i], bv[Q][j], acc[i][j], 0, 0, 0);
        X6_PAIRS(X6_MFMA)
#undef X6_MFMA
        __syncthreads();
    }
    float *out = a.out + (long long)split * a.K * a.N;
#pragma unroll
    for (int i = 0; i < TK; i++)
#pragma unroll
        for (int j = 0; j < TN; j++) {
            const int col = n0 + wn * WN + j * 32 + l31;
            const bool col_ok = col < a.N;
#pragma unroll
            for (int r = 0; r < 16; r++) {
                const int row = k0 + wk * WK + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                if (col_ok && row < a.K) out[(long long)row * a.N + col] = acc[i][j][r];
            }
        }
}

struct PlanX6 {
    int bko, bn, tiles_k, tiles_n, splits, rows_per_split;
};

PlanX6 make_plan_x6(const emadapt_conv_desc *d) {
    PlanX6 p;
    const long long M = (long long)d->B * d->H * d->W;
    const int K = d->KH * d->KW * d->Cin;
    const int force = tune(TUNE_X6_TILE);  // 128|256 (the parity tests run both)
    bool big = K >= 256 && d->Cout >= 128;
    if (force) big = force == 256;
    p.bko = big ? 256 : 128;
    p.bn = big ? 128 : (d->Cout <= 64 ? 64 : 128);
    p.tiles_k = (int)ceil_div(K, p.bko);
    p.tiles_n = (int)ceil_div(d->Cout, p.bn);
    const long long tiles = (long long)p.tiles_k * p.tiles_n;
    const double slots = big ? 256.0 : 512.0;
    const double t_kernel = 2.0 * (double)M * (p.tiles_k * (double)p.bko) * (p.tiles_n * (double)p.bn) / (big ? 250e12 : 200e12);
    const double kn_bytes = (double)K * d->Cout * 4.0;
    long long best = 1;
    double best_t = 1e30;
    const long long max_s = ceil_div(M, 256) < 2048 ? ceil_div(M, 256) : 2048;
    for (long long s = 1; s <= max_s; s++) {
        const double rounds = (double)tiles * s / slots;
        const double eff = rounds / (double)(long long)(rounds + 0.999999);
        double t = t_kernel / eff + 2e-6 * (double)(long long)(rounds + 0.999999);
        if (s > 1) t += (double)(s + 1) * kn_bytes / 3e12 + 3e-6;
        if (t < best_t * 0.995) { best_t = t; best = s; }
        if (rounds > 24.0) break;
    }
    long long rps = ceil_div(M, best);
    rps = ceil_div(rps, 64) * 64;
    p.rows_per_split = (int)rps;
    p.splits = (int)ceil_div(M, rps);
    return p;
}

int check_desc_wgrad_x6(const emadapt_conv_desc *d) {
    int rc = check_desc(d);
    if (rc) return rc;
    const long long M = (long long)d->B * d->H * d->W;
    if (M * (d->W > d->H ? d->W : d->H) >= (1LL << 32)) return EMADAPT_ERR_UNSUPPORTED;  // magic division range
    return EMADAPT_OK;
}

template <int BKO, int BN>
int launch_wgrad_x6(const WgradX6Args &a, hipStream_t stream) {
    constexpr size_t lds = 2 * 3 * 16 * (BKO * 2 + 64 + BN * 2 + 64);
    constexpr int threads = (BKO / 64 * 2 + 4) * 64;
    static DynLdsAttr attr;
    {
        const int arc = attr.ensure(reinterpret_cast<const void *>(&wgrad_x6_kernel<BKO, BN>), lds);
        if (arc) return arc;
    }
    const long long nblk = (long long)a.tiles_k * a.tiles_n * a.splits;
    if (nblk > 0x7fffffffLL) return EMADAPT_ERR_UNSUPPORTED;
    hipLaunchKernelGGL((wgrad_x6_kernel<BKO, BN>), dim3((unsigned)nblk), dim3(threads), lds, stream, a);
    return launch_status();
}

}  // namespace

extern "C" {

int emadapt_filter_kmajor(const emadapt_conv_desc *d, const float *w, float *w_kmajor, void *stream_) {
    int rc = check_desc(d);
    if (rc) return rc;
    if (!w || !w_kmajor || !aligned16(w) || !aligned16(w_kmajor)) return EMADAPT_ERR_INVALID;
    const int tiles_ci = (int)ceil_div(d->Cin, 32), tiles_co = (int)ceil_div(d->Cout, 32);
    const long long nblk = (long long)d->KH * d->KW * tiles_ci * tiles_co;
    hipLaunchKernelGGL(filter_kmajor_kernel, dim3((unsigned)nblk), dim3(256), 0, as_stream(stream_), w, w_kmajor, d->Cin,
                       d->Cout, tiles_ci, tiles_co);
    return launch_status();
}

size_t emadapt_conv_fwd_x6_workspace_bytes(const emadapt_conv_desc *d) {
    if (check_desc(d) != EMADAPT_OK || d->B == 0) return 0;
    IgemmArgs a;
    fill_geometry(a, d, false);
    const int s = x6_splits(a);
    return s > 1 ? (size_t)s * a.M * a.N * sizeof(float) : 0;
}

size_t emadapt_conv_dgrad_x6_workspace_bytes(const emadapt_conv_desc *d) {
    if (check_desc(d) != EMADAPT_OK || d->B == 0) return 0;
    IgemmArgs a;
    fill_geometry(a, d, true);
    const int s = x6_splits(a);
    return s > 1 ? (size_t)s * a.M * a.N * sizeof(float) : 0;
}

int emadapt_conv_fwd_x6(const emadapt_conv_desc *d, const float *x, const float *w_kmajor, const float *bias, float *y,
                        int epilogue, float keep_prob, const uint8_t *drop_mask, uint64_t drop_seed, void *workspace,
                        size_t workspace_bytes, void *stream_) {
    int rc = check_desc(d);
    if (rc) return rc;
    if (!x || !w_kmajor || !y) return EMADAPT_ERR_INVALID;
    if (epilogue < EMADAPT_EPI_NONE || epilogue > EMADAPT_EPI_BIAS_RELU) return EMADAPT_ERR_INVALID;
    if (epilogue != EMADAPT_EPI_NONE && !bias) return EMADAPT_ERR_INVALID;
    if (!(keep_prob > 0.f) || keep_prob > 1.f) return EMADAPT_ERR_INVALID;
    if (keep_prob < 1.f && epilogue != EMADAPT_EPI_BIAS_RELU) return EMADAPT_ERR_INVALID;
    if (!aligned16(x) || !aligned16(w_kmajor) || !aligned16(y) || !aligned16(workspace)) return EMADAPT_ERR_INVALID;
    if (d->B == 0) return EMADAPT_OK;
    IgemmArgs a;
    fill_geometry(a, d, false);
    a.x = x; a.w = w_kmajor; a.y = y; a.bias = bias; a.gate = nullptr; a.mask = drop_mask;
    a.epi = epilogue; a.scale = 1.f; set_keep(a, keep_prob); a.seed = drop_seed;
    hipStream_t stream = as_stream(stream_);
    ProfScope prof(EMADAPT_K_IGEMM_FWD, 2.0 * a.M * (double)a.K * a.N, 4.0 * ((double)a.M * a.Cx + (double)a.K * a.N + (double)a.M * a.N), stream);
    return dispatch_igemm_x6(a, workspace, workspace_bytes, stream);
}

int emadapt_conv_dgrad_x6(const emadapt_conv_desc *d, const float *dy, const float *w, float *dx, const float *gate,
                          float gate_scale, void *workspace, size_t workspace_bytes, void *stream_) {
    int rc = check_desc(d);
    if (rc) return rc;
    if (!dy || !w || !dx) return EMADAPT_ERR_INVALID;
    if (!aligned16(dy) || !aligned16(w) || !aligned16(dx) || !aligned16(workspace)) return EMADAPT_ERR_INVALID;
    if (d->B == 0) return EMADAPT_OK;
    IgemmArgs a;
    fill_geometry(a, d, true);
    a.x = dy; a.w = w; a.y = dx; a.bias = nullptr; a.gate = gate; a.mask = nullptr;
    a.epi = EMADAPT_EPI_NONE; a.scale = gate_scale; set_keep(a, 1.f); a.seed = 0;
    hipStream_t stream = as_stream(stream_);
    ProfScope prof(EMADAPT_K_IGEMM_DGRAD, 2.0 * a.M * (double)a.K * a.N, 4.0 * ((double)a.M * a.Cx + (double)a.K * a.N + (double)a.M * a.N), stream);
    return dispatch_igemm_x6(a, workspace, workspace_bytes, stream);
}

size_t emadapt_conv_wgrad_x6_workspace_bytes(const emadapt_conv_desc *d) {
    if (check_desc_wgrad_x6(d) != EMADAPT_OK || d->B == 0) return 0;
    const PlanX6 p = make_plan_x6(d);
    if (p.splits <= 1) return 0;
    return (size_t)p.splits * d->KH * d->KW * d->Cin * d->Cout * sizeof(float);
}

int emadapt_conv_wgrad_x6(const emadapt_conv_desc *d, const float *x, const float *dy, float *dw, void *workspace,
                          size_t workspace_bytes, void *stream_) {
    int rc = check_desc_wgrad_x6(d);
    if (rc) return rc;
    if (!x || !dy || !dw) return EMADAPT_ERR_INVALID;
    if (!aligned16(x) || !aligned16(dy) || !aligned16(dw) || !aligned16(workspace)) return EMADAPT_ERR_INVALID;
    hipStream_t stream = as_stream(stream_);
    const long long KN = (long long)d->KH * d->KW * d->Cin * d->Cout;
    if (d->B == 0) {
        EMADAPT_HIP_TRY(hipMemsetAsync(dw, 0, KN * sizeof(float), stream));
        return EMADAPT_OK;
    }
    const PlanX6 p = make_plan_x6(d);
    if (p.splits > 1 && (!workspace || workspace_bytes < emadapt_conv_wgrad_x6_workspace_bytes(d)))
        return EMADAPT_ERR_WORKSPACE;
    WgradX6Args a;
    a.x = x; a.dy = dy; a.out = p.splits > 1 ? static_cast<float *>(workspace) : dw;
    a.H = d->H; a.W = d->W; a.Cin = d->Cin; a.N = d->Cout; a.KW = d->KW; a.dil = d->dil;
    a.pad_h = conv_pad_before(d->KH, d->dil); a.pad_w = conv_pad_before(d->KW, d->dil);
    a.M = d->B * d->H * d->W; a.ntaps = d->KH * d->KW; a.K = a.ntaps * d->Cin;
    a.tiles_k = p.tiles_k; a.tiles_n = p.tiles_n; a.splits = p.splits; a.rows_per_split = p.rows_per_split;
    a.x_bytes = (unsigned)((long long)a.M * d->Cin * 4); a.y_bytes = (unsigned)((long long)a.M * d->Cout * 4);
    a.magic_w = magic_u32(d->W); a.magic_h = magic_u32(d->H);
    ProfScope prof(EMADAPT_K_IGEMM_WGRAD, 2.0 * a.M * (double)a.K * a.N, 4.0 * ((double)a.M * d->Cin + (double)a.M * a.N + (double)a.K * a.N), stream);
    rc = p.bko == 256 ? launch_wgrad_x6<256, 128>(a, stream)
       : p.bn == 64  ? launch_wgrad_x6<128, 64>(a, stream)
                     : launch_wgrad_x6<128, 128>(a, stream);
    if (rc) return rc;
    if (p.splits > 1) rc = launch_slab_sum(static_cast<const float *>(workspace), p.splits, KN, dw, stream);
    return rc;
}

}  // extern "C"

// bf16-compute variants of the three convolution GEMMs (configurations C3/C4: "bf16 compute,
// fp32 master weights and accumulators").  Same call surface, layouts, epilogues and split-K/split-M machinery as
// the fp32 kernels (conv_igemm.hip, conv_wgrad.hip); what changes is the arithmetic of the inner product:
//
//   operands are rounded to bf16 (round-to-nearest-even, v_cvt_pk_bf16_f32), products are exact, sums are fp32
//   (v_mfma_f32_32x32x16_bf16: 16 k per instruction at half the cycles of the fp32 32x32x2 form -> 16x the rate).
//
// Activations and activation gradients are stored f32 or bf16 (template parameters XBF / YBF / DBF, EMADAPT_DT_* in the
// ABI): a bf16 tensor holds the values an f32 tensor would be rounded to, was rounded once by the epilogue that wrote it,
// and is copied into LDS as 16-byte vectors; an f32 tensor is converted by the PRODUCER wavefronts on its way into LDS.
// With the matrix pipe this fast the kernels are bound by operand bytes (L2->CU delivery, ~70 GB/s per CU), hence bf16
// storage, 256x256 tiles and vector epilogues.  The
// filters are converted once per optimizer step into two bf16 images whose reduction axis is contiguous, so that
// the forward pass and the data gradient read them with plain 16-byte loads and store them untransposed:
//     fwd image   [tap][co][ci8]   (ci padded to a multiple of 8, zero filled)   k = (tap, ci), n = co
//     dgrad image [tap][ci][co8]   (the HWIO filter itself, co padded to 8)      k = (tap, co), n = ci
//
// LDS tiles are [row][k] bf16 with a 144-byte row pitch: the 16-lane groups of ds_read_b128 then touch 16 different
// 16-byte slots of the 256-byte bank row (9 r mod 16 is a bijection on each group's rows) -- conflict-free without a
// swizzle -- and the producers' ds_write_b64 / ds_write_b128 cover contiguous 128-byte runs.
//
// The filter gradient reduces over pixels, the axis along which neither operand is contiguous; its LDS tiles are
// [pixel][k] and [pixel][n] exactly as loaded, and the MFMA fragments are gathered with the gfx950 transposing read
// ds_read_b64_tr_b16 (4 pixels x 16 columns per 16-lane group, delivered column-major).  Row pitch = columns*2 + 64
// bytes: the four pixel rows of one read sit 16 banks apart, so the 32-lane halves are conflict-free.
#include <stdlib.h>

#include <type_traits>

#include "igemm_common.h"

namespace emadapt {
int launch_slab_sum(const float *slab, int nslab, long long n, float *out, hipStream_t stream);
int launch_slab_sum2(const float *slab_a, int nslab_a, long long n_a, float *out_a, const float *slab_b, int nslab_b, long long n_b,
                     float *out_b, hipStream_t stream);
}

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));

inline int round_up8(int v) { return (v + 7) & ~7; }

__device__ __forceinline__ uint2 cvt4(float4 v) {
    bf16x4 c;
    c[0] = (__bf16)v.x; c[1] = (__bf16)v.y; c[2] = (__bf16)v.z; c[3] = (__bf16)v.w;
    return *reinterpret_cast<uint2 *>(&c);
}

// ------------------------------------------------------------------------------------------------------------------
// Filter conversion: fp32 HWIO -> the two bf16 images.  One 32x32 (ci, co) tile per workgroup, transposed through LDS.
// ------------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ void weights_to_bf16_block(int bid, const float *__restrict__ w, __bf16 *__restrict__ wf, __bf16 *__restrict__ wd, int Cin,
                                                      int Cout, int Cip, int Cop, int tiles_ci, int tiles_co) {
    __shared__ float tile[32][33];
    const int tco = bid % tiles_co;
    bid /= tiles_co;
    const int tci = bid % tiles_ci;
    const int tap = bid / tiles_ci;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
    const int ci0 = tci * 32, co0 = tco * 32;
#pragma unroll
    for (int r = 0; r < 4; r++) {
        const int ci = ci0 + ty + 8 * r, co = co0 + tx;
        const float v = (ci < Cin && co < Cout) ? w[((long long)tap * Cin + ci) * Cout + co] : 0.f;
        tile[ty + 8 * r][tx] = v;
        if (wd && ci < Cin && co < Cop) wd[((long long)tap * Cin + ci) * Cop + co] = (__bf16)v;
    }
    __syncthreads();
    if (!wf) return;
#pragma unroll
    for (int r = 0; r < 4; r++) {
        const int co = co0 + ty + 8 * r, ci = ci0 + tx;
        if (co < Cout && ci < Cip) wf[((long long)tap * Cout + co) * Cip + ci] = (__bf16)tile[tx][ty + 8 * r];
    }
}
__global__ void __launch_bounds__(256) weights_to_bf16_kernel(const float *__restrict__ w, __bf16 *__restrict__ wf,
                                                              __bf16 *__restrict__ wd, int Cin, int Cout, int Cip,
                                                              int Cop, int tiles_ci, int tiles_co) {
    weights_to_bf16_block((int)blockIdx.x, w, wf, wd, Cin, Cout, Cip, Cop, tiles_ci, tiles_co);
}
// Several filters in one launch (the small layers' refresh launches are all ramp and tail): layer k owns blocks [first[k], first[k + 1]).
constexpr int kW16MaxLayers = 16;
struct W16Layers {
    const float *w[kW16MaxLayers];
    __bf16 *wf[kW16MaxLayers], *wd[kW16MaxLayers];
    int Cin[kW16MaxLayers], Cout[kW16MaxLayers], tiles_ci[kW16MaxLayers], tiles_co[kW16MaxLayers], first[kW16MaxLayers + 1];
    int count;
};
__global__ void __launch_bounds__(256) weights_to_bf16_multi_kernel(W16Layers L) {
    const int bid = blockIdx.x;
    int k = 0;
#pragma unroll
    for (int q = 1; q < kW16MaxLayers; q++) k += (q < L.count && bid >= L.first[q]) ? 1 : 0;
    weights_to_bf16_block(bid - L.first[k], L.w[k], L.wf[k], L.wd[k], L.Cin[k], L.Cout[k], (L.Cin[k] + 7) & ~7, (L.Cout[k] + 7) & ~7, L.tiles_ci[k],
                          L.tiles_co[k]);
}

// ------------------------------------------------------------------------------------------------------------------
// Forward / data-gradient: out[m][n] = sum_k A[m][k] Bw[n][k].
// IgemmArgs is used as in the fp32 kernel with: w = bf16 image [tap][N][Cxp], Cxp = round_up(Cx, 8) = the k pitch of
// a tap (K = ntaps * Cxp; channels Cx..Cxp-1 are zero on both sides), magic_cx = magic(Cxp).
//
// Tile configurations <BM, BN, WN> (BK = 64; every consumer wave owns a 64 x WN block; 4 producer waves):
//   128x128, WN 64  : 4 + 4 waves, 2 workgroups/CU, 128 registers/wave
//   256x256, WN 128 : 8 + 4 waves, 1 workgroup/CU, 168 registers/wave.  What bounds these kernels is the delivery rate
//                     of the vector memory path (~70 GB/s per CU from L2, MI355X_MICROARCH.md "gather into LDS"), i.e.
//                     bytes per FLOP: 48 KB of operands per 128x128x64 step = 94 B/clk to keep the MFMAs busy against
//                     ~29 B/clk delivered (31 %); the 256x256 tile moves 96 KB per 4x the FLOPs (47 B/clk, 62 %).
// ------------------------------------------------------------------------------------------------------------------
constexpr int kPitchG = 144;  // bytes per LDS row of 64 bf16 (+8 pad)

// bias / ReLU / dropout, or the ReLU-gradient gate (stored activation, bf16 when GBF), on one output element.
template <bool GBF>
__device__ __forceinline__ float epilogue_t(const IgemmArgs &a, float v, int col, long long o) {
    if (a.epi != EMADAPT_EPI_NONE && a.bias) v += a.bias[col];
    if (a.epi == EMADAPT_EPI_BIAS_RELU) {
        v = relu_keep_nan(v);
        if (a.keep < 1.0f) {
            const bool kp = a.mask ? (a.mask[o] != 0) : dropout_keep(a.seed, (unsigned long long)o, a.keep, a.keep8);
            v = kp ? drop_scale(a, v) : 0.f;
        }
    }
    if (a.gate) {
        const float g = GBF ? (float)reinterpret_cast<const __bf16 *>(a.gate)[o] : a.gate[o];
        v = gate_keep_nan(g, v * a.scale);
    }
    return v;
}

template <bool YBF>
__device__ __forceinline__ void store_out(const IgemmArgs &a, long long o, float v) {
    if constexpr (YBF) reinterpret_cast<__bf16 *>(a.y)[o] = (__bf16)v;
    else a.y[o] = v;
}

// split-K second stage for the bf16-storage variants (the fp32-storage one is splitk_epilogue_kernel).
template <bool YBF>
__global__ void __launch_bounds__(256) splitk_epilogue_t_kernel(IgemmArgs a) {
    const long long mn = (long long)(a.m_end - a.m_begin) * a.N;
    const long long n4 = mn / 4;
    const long long o0 = (long long)a.m_begin * a.N;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long long)gridDim.x * blockDim.x) {
        float4 v = *reinterpret_cast<const float4 *>(a.slab + i * 4);
        for (int sp = 1; sp < a.splits; sp++) {
            const float4 u = *reinterpret_cast<const float4 *>(a.slab + sp * mn + i * 4);
            v.x += u.x; v.y += u.y; v.z += u.z; v.w += u.w;
        }
        const long long o = o0 + i * 4;
        const int col = (int)(o % a.N);
        v.x = epilogue_t<YBF>(a, v.x, col + 0, o + 0);
        v.y = epilogue_t<YBF>(a, v.y, col + 1, o + 1);
        v.z = epilogue_t<YBF>(a, v.z, col + 2, o + 2);
        v.w = epilogue_t<YBF>(a, v.w, col + 3, o + 3);
        if constexpr (YBF) *reinterpret_cast<uint2 *>(reinterpret_cast<__bf16 *>(a.y) + o) = cvt4(v);
        else *reinterpret_cast<float4 *>(a.y + o) = v;
    }
}

}  // namespace

#include "igemm_b16t.h"   // the LDS-DMA kernel for bf16-stored activations (uses epilogue_t / cvt4 / splitk_epilogue_t_kernel above)
#include "conv_d64.h"     // direct 3 x 3 convolution, 64 -> 64 channels, the filter in registers (conv1_2)
#include "conv_d4.h"      // direct 3 x 3 convolution of the 4-channel f32 image into 64 bf16 channels (conv1_1 forward)
#include "conv_k32.h"     // 1 x 1 convolution with at most 32 gathered channels, operands straight from memory (fc8's data gradient)

namespace {

// Storage types: XBF = the gathered activation (x, or dy for the data gradient) is bf16 in memory (Cx % 8 == 0; 16-byte
// loads carry 8 channels and go to LDS unconverted); YBF = the output (and the gate of the data gradient) is bf16.
// ABLM: diagnostic builds only (scripts/bf16_ablate.py) -- parts of the kernel compiled out; 0 in every shipped instantiation.
template <int BM, int BN, int WN, bool XBF, bool YBF, int ABLM = 0>
__global__ void __launch_bounds__(((BM / 64) * (BN / WN) + 4) * 64, BM == 128 ? 4 : 3) igemm_bf16_kernel(IgemmArgs a, int Cxp) {
    constexpr int BK = 64, WM = 64;
    constexpr int WAVES_N = BN / WN;
    constexpr int CTHREADS = (BM / WM) * WAVES_N * 64;  // consumer threads; the 256 producer threads follow
    constexpr int TM = WM / 32, TN = WN / 32;
    constexpr int KC = BK / 8;             // 8 bf16 chunks (16 B) per row
    constexpr int KQ = XBF ? KC : BK / 4;  // 16-byte pieces per A row: 8 bf16 chunks or 16 fp32 quads
    constexpr int A_RSTEP = 256 / KQ;      // rows per pass of the producers
    constexpr int A_N = BM / A_RSTEP;      // A loads per producer thread and K step
    constexpr int B_RSTEP = 256 / KC;      // 32
    constexpr int B_N = BN / B_RSTEP;
    constexpr int TILE_A = BM * kPitchG, TILE_B = BN * kPitchG;  // bytes per operand tile

    extern __shared__ __attribute__((aligned(16))) char ldsb[];
    char *As = ldsb;                 // [2][BM][kPitchG]
    char *Bs = ldsb + 2 * TILE_A;    // [2][BN][kPitchG]

    int tile_m, tile_n, split;
    {
        const int ntile = a.tiles_m * a.tiles_n;
        const int nblk = ntile * a.splits;
        const int bid = blockIdx.x;
        const int xcd = bid & 7, q = nblk >> 3, r = nblk & 7;
        int t = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
        split = t / ntile;
        t -= split * ntile;
        constexpr int GM = BM == 128 ? 8 : 4;  // M-tiles per group: the XCD's co-resident workgroups form a square-ish block
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
        // ================================ PRODUCER wavefronts ================================
        const int tid = threadIdx.x - CTHREADS;
        const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(a.x), 0, (int)a.x_bytes, 0x00020000);
        const __amdgpu_buffer_rsrc_t wr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(a.w), 0, (int)a.w_bytes, 0x00020000);
        constexpr unsigned kOOB = 0x80000000u;
        const int a_kq = tid % KQ, a_r0 = tid / KQ;
        // per-row pixel coordinates, packed (oh << 16 | ow); rows past m_end get oh = 0x4000 and fail every bounds check
        unsigned a_hw[A_N];
#pragma unroll
        for (int j = 0; j < A_N; j++) {
            const int m = m0 + a_r0 + j * A_RSTEP;
            if (m < a.m_end) {
                const int t2 = m / a.W;
                a_hw[j] = ((unsigned)(t2 % a.H) << 16) | (unsigned)(m - t2 * a.W);
            } else {
                a_hw[j] = 0x40000000u;
            }
        }
        const int a_base0 = (m0 + a_r0) * a.Cx;
        const int a_rstride = A_RSTEP * a.Cx;
        const int b_kc = tid % KC, b_r0 = tid / KC;
        u32x4 ra[A_N];
        u32x4 rb[B_N];
        constexpr int AEL = XBF ? 8 : 4;   // channels per A piece
        constexpr unsigned ABY = XBF ? 2u : 4u;

        // per-K-step addressing, shared by the loads of a step
        int toff, dh, dw;
        bool kvalid, bvalid;
        unsigned boff;
        auto step_setup = [&](int kt) {
            unsigned tap, tapb;
            int c, cb;
            if (a.tap_inner) {  // Cxp % BK == 0: the whole K step lies in one tap
                const unsigned chunk = __umulhi((unsigned)kt, a.magic_nt);
                tap = (unsigned)kt - chunk * (unsigned)a.ntaps;
                if ((int)chunk * BK >= Cxp) tap = (unsigned)a.ntaps;
                tapb = tap;
                c = (int)chunk * BK + a_kq * AEL;
                cb = (int)chunk * BK + b_kc * 8;
            } else {
                const unsigned k = (unsigned)(kt * BK + a_kq * AEL);
                tap = __umulhi(k, a.magic_cx);
                c = (int)(k - tap * (unsigned)Cxp);
                const unsigned kb = (unsigned)(kt * BK + b_kc * 8);
                tapb = __umulhi(kb, a.magic_cx);
                cb = (int)(kb - tapb * (unsigned)Cxp);
            }
            const unsigned kh = div_magic(tap, a.magic_kw);
            const int kw = (int)(tap - kh * (unsigned)a.KW);
            kvalid = (int)tap < a.ntaps && c < a.Cx;
            dh = a.sign * ((int)kh * a.dil - a.pad_h);
            dw = a.sign * (kw * a.dil - a.pad_w);
            toff = a_base0 + (dh * a.W + dw) * a.Cx + c;
            bvalid = (int)tapb < a.ntaps;
            boff = (unsigned)(((int)tapb * a.N + n0 + b_r0) * Cxp + cb) * 2u;
        };
        auto load_a = [&](int j) {
            const int ih = (int)(a_hw[j] >> 16) + dh, iw = (int)(a_hw[j] & 0xffffu) + dw;
            const bool ok = kvalid && (unsigned)ih < (unsigned)a.H && (unsigned)iw < (unsigned)a.W;
            if constexpr ((ABLM & 1) || (ABLM & 8)) return;   // (diagnostic builds: no loads / no A loads -- the registers keep what they held)
            ra[j] = __builtin_amdgcn_raw_buffer_load_b128(xr, (int)(((unsigned)(toff + j * a_rstride) * ABY) | (ok ? 0u : kOOB)), 0, 0);
        };
        auto load_b = [&](int j) {
            const int n = n0 + b_r0 + j * B_RSTEP;
            const unsigned off = boff + (unsigned)(j * B_RSTEP * Cxp) * 2u;
            if constexpr ((ABLM & 1) || (ABLM & 16)) return;
            rb[j] = __builtin_amdgcn_raw_buffer_load_b128(wr, (int)(off | ((bvalid && n < a.N) ? 0u : kOOB)), 0, 0);
        };
        auto store_a = [&](int buf, int j) {
            if constexpr ((ABLM & 2) != 0) return;   // no LDS stores
            if constexpr (XBF) {
                *reinterpret_cast<u32x4 *>(As + buf * TILE_A + (a_r0 + j * A_RSTEP) * kPitchG + a_kq * 16) = ra[j];
            } else {
                const float4 f = make_float4(__uint_as_float(ra[j].x), __uint_as_float(ra[j].y), __uint_as_float(ra[j].z),
                                             __uint_as_float(ra[j].w));
                *reinterpret_cast<uint2 *>(As + buf * TILE_A + (a_r0 + j * A_RSTEP) * kPitchG + a_kq * 8) = cvt4(f);
            }
        };
        auto store_b = [&](int buf, int j) {
            if constexpr ((ABLM & 2) != 0) return;
            *reinterpret_cast<u32x4 *>(Bs + buf * TILE_B + (b_r0 + j * B_RSTEP) * kPitchG + b_kc * 16) = rb[j];
        };
        step_setup(kt0);
#pragma unroll
        for (int j = 0; j < A_N; j++) load_a(j);
#pragma unroll
        for (int j = 0; j < B_N; j++) load_b(j);
        step_setup(kt0 + 1);  // K steps past the end are zero-filled by the range check
#pragma unroll
        for (int j = 0; j < A_N; j++) { store_a(0, j); load_a(j); }
#pragma unroll
        for (int j = 0; j < B_N; j++) { store_b(0, j); load_b(j); }
        __syncthreads();
        // steady state: register j is parked in LDS and immediately re-armed with the load of the step after next, so
        // the vector-memory pipe always has all but one of this thread's loads in flight
        for (int kt = 0; kt < nk; kt++) {
            const int buf = (kt + 1) & 1;
            step_setup(kt0 + kt + 2);
#pragma unroll
            for (int j = 0; j < A_N; j++) { store_a(buf, j); load_a(j); }
#pragma unroll
            for (int j = 0; j < B_N; j++) { store_b(buf, j); load_b(j); }
            if constexpr (!(ABLM & 4)) __syncthreads();
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
        const char *as = As + cur * TILE_A + (wm * WM + l31) * kPitchG + lh * 16;
        const char *bs = Bs + cur * TILE_B + (wn * WN + l31) * kPitchG + lh * 16;
        if constexpr (BM == 128) {
            // fragments of k-step ks+1 are fetched while the MFMAs of ks run
            bf16x8 av[2][TM], bv[2][TN];
#pragma unroll
            for (int i = 0; i < TM; i++) av[0][i] = *reinterpret_cast<const bf16x8 *>(as + i * 32 * kPitchG);
#pragma unroll
            for (int j = 0; j < TN; j++) bv[0][j] = *reinterpret_cast<const bf16x8 *>(bs + j * 32 * kPitchG);
#pragma unroll
            for (int ks = 0; ks < BK / 16; ks++) {
                if (ks + 1 < BK / 16) {
#pragma unroll
                    for (int i = 0; i < TM; i++)
                        av[(ks + 1) & 1][i] = *reinterpret_cast<const bf16x8 *>(as + i * 32 * kPitchG + (ks + 1) * 32);
#pragma unroll
                    for (int j = 0; j < TN; j++)
                        bv[(ks + 1) & 1][j] = *reinterpret_cast<const bf16x8 *>(bs + j * 32 * kPitchG + (ks + 1) * 32);
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int i = 0; i < TM; i++)
#pragma unroll
                    for (int j = 0; j < TN; j++)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av[ks & 1][i], bv[ks & 1][j], acc[i][j], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
        } else {
            // 128 accumulator registers leave room for one fragment set plus the next step's A fragments; the second
            // consumer wave of the SIMD covers the LDS latency of the B fragments
            bf16x8 av[2][TM], bv[TN];
            const bool rd = !(ABLM & 32) || kt == 0;   // (diagnostic builds, bit 32: fragments are read in the first K step only)
            if (rd) {
#pragma unroll
                for (int i = 0; i < TM; i++) av[0][i] = *reinterpret_cast<const bf16x8 *>(as + i * 32 * kPitchG);
            }
#pragma unroll
            for (int ks = 0; ks < BK / 16; ks++) {
                if (rd) {
#pragma unroll
                    for (int j = 0; j < TN; j++) bv[j] = *reinterpret_cast<const bf16x8 *>(bs + j * 32 * kPitchG + ks * 32);
                    if (ks + 1 < BK / 16) {
#pragma unroll
                        for (int i = 0; i < TM; i++)
                            av[(ks + 1) & 1][i] = *reinterpret_cast<const bf16x8 *>(as + i * 32 * kPitchG + (ks + 1) * 32);
                    }
                }
                if constexpr ((ABLM & 64) != 0) continue;   // bit 64: no MFMAs
#pragma unroll
                for (int j = 0; j < TN; j++)
#pragma unroll
                    for (int i = 0; i < TM; i++)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av[ks & 1][i], bv[j], acc[i][j], 0, 0, 0);
            }
        }
        if constexpr (!(ABLM & 4)) __syncthreads();
    }
    // ---- epilogue through LDS (igemm_common.h: store_acc_tile): the operand tiles are dead after the last barrier
    constexpr int SLAB = 32 * (WN + 4);  // floats per wave
    static_assert(SLAB * 4 * (BM / WM) * WAVES_N <= 2 * (BM + BN) * kPitchG, "epilogue slabs must fit in the tile LDS");
    float *slab = a.splits > 1 ? a.slab + ((long long)split * (a.m_end - a.m_begin) - a.m_begin) * a.N : nullptr;
    store_acc_tile<TM, TN, 32, YBF>(a, acc, reinterpret_cast<float *>(ldsb) + wave * SLAB, lane, m0 + wm * WM, n0 + wn * WN, slab);
}

void fill_geometry_bf16(IgemmArgs &a, int &Cxp, const emadapt_conv_desc *d, bool dgrad) {
    fill_geometry(a, d, dgrad);
    Cxp = round_up8(a.Cx);
    a.K = a.ntaps * Cxp;
    a.w_bytes = (unsigned)((long long)a.ntaps * a.N * Cxp * 2);
    a.magic_cx = magic_u32(Cxp);
}

// Launch plan: the 256x256 tile (one workgroup per CU) when the output is wide and tall enough to fill the chip with
// it, else 128x128 tiles (two per CU) with split-K when the tile count is small.
struct PlanG {
    int big;     // 1: 256x256 tiles
    int splits;
    int narrow;  // 1: 128x64 tiles (N <= 64: conv1_x, the data gradients into 64 channels -- round 4: the 128-wide tile multiplied and
                 //    loaded 64 columns of nothing there)
};

PlanG plan_bf16(const IgemmArgs &a) {
    int force = tune(TUNE_BF16_TILE);  // 128|256 forces a configuration (the parity tests run both on small shapes)
    if (force == 1 || force == 3) force = 0;   // (1: the automatic plan with the direct kernels off; 3: round 5's patch kernel, deleted in round 6 -- the automatic plan)
    PlanG p;
    const long long t256 = ceil_div(a.M, 256) * ceil_div(a.N, 256);
    p.big = (a.N >= 256 && t256 >= 192) ? 1 : 0;
    if (force) p.big = force == 256 ? 1 : 0;
    p.narrow = (!p.big && a.N <= 64 && force != 128) ? 1 : 0;
    p.splits = p.big ? choose_splits(a.M, a.N, a.K, 256, 256, 64, 256.0, 1000e12)
                     : choose_splits(a.M, a.N, a.K, 128, p.narrow ? 64 : 128, 64, 512.0, 450e12);
    return p;
}

size_t igemm_bf16_workspace_bytes(const IgemmArgs &a) {
    const int s = plan_bf16(a).splits;
    return s > 1 ? (size_t)s * a.M * a.N * sizeof(float) : 0;
}

template <int BM, int BN, int WN, bool XBF, bool YBF, int ABLM = 0>
int launch_igemm_bf16(IgemmArgs &a, int Cxp, hipStream_t stream) {
    constexpr size_t lds = 2 * (BM + BN) * kPitchG;
    constexpr int threads = ((BM / 64) * (BN / WN) + 4) * 64;
    static DynLdsAttr attr;
    {
        const int arc = attr.ensure(reinterpret_cast<const void *>(&igemm_bf16_kernel<BM, BN, WN, XBF, YBF, ABLM>), lds);
        if (arc) return arc;
    }
    a.tiles_m = (int)ceil_div(a.M, BM);
    a.tiles_n = (int)ceil_div(a.N, BN);
    const long long nblk = (long long)a.tiles_m * a.tiles_n * a.splits;
    if (nblk > 0x7fffffffLL) return EMADAPT_ERR_UNSUPPORTED;
    hipLaunchKernelGGL((igemm_bf16_kernel<BM, BN, WN, XBF, YBF, ABLM>), dim3((unsigned)nblk), dim3(threads), lds, stream, a, Cxp);
    int rc = launch_status();
    if (rc || a.splits <= 1) return rc;
    long long g = ((long long)a.M * a.N / 4 + 255) / 256;
    if (g > 8192) g = 8192;
    hipLaunchKernelGGL(splitk_epilogue_t_kernel<YBF>, dim3((unsigned)g), dim3(256), 0, stream, a);
    return launch_status();
}

template <bool XBF, bool YBF>
int dispatch_igemm_bf16(IgemmArgs &a, int Cxp, void *workspace, size_t workspace_bytes, hipStream_t stream) {
    // (round 5 carried an opt-in patch kernel for the 3 x 3 layers with N <= 128, conv_p3.h: 3 - 10 % ahead per layer in an isolated loop,
    // a tie in the step.  Round 6's in-step trace has it 3.6 % BEHIND on the four conv2 launches it would take -- the isolated loop ran on
    // operands resident in the Infinity Cache -- profiles/r06_p3_in_step.txt; deleted.)
    if constexpr (XBF && YBF) {   // fc8's data gradient (1 x 1, 24 gathered channels): one MFMA step, nothing staged
        if (tune(TUNE_BF16_TILE) == 0 && k32_shape_ok(a) && !a.bias && a.epi == EMADAPT_EPI_NONE) return launch_k32(a, stream);
    }
    if constexpr (XBF && YBF) {   // conv1_2 (64 -> 64 channels, 3 x 3): the direct kernel, unless a tile is forced
        if (tune(TUNE_BF16_TILE) == 0 && d64_shape_ok(a) && !(a.epi == EMADAPT_EPI_BIAS_RELU && (a.keep < 1.0f || a.mask))) {
            a.m_begin = 0;
            a.m_end = a.M;
            return launch_d64(a, a.sign < 0, stream);
        }
    }
    if constexpr (!XBF && YBF) {   // conv1_1 (the 4-channel f32 image -> 64 bf16 channels, 3 x 3): the direct kernel, unless a tile is forced
        if (tune(TUNE_BF16_TILE) == 0 && d4_shape_ok(a) && !a.gate && !(a.epi == EMADAPT_EPI_BIAS_RELU && (a.keep < 1.0f || a.mask))) return launch_d4(a, stream);
    }
    PlanG p = plan_bf16(a);
    if (p.splits > 1 && (!workspace || workspace_bytes < igemm_bf16_workspace_bytes(a))) p.splits = 1;
#ifdef EMADAPT_ABLATE_BUILD
    a.ablate = tune(TUNE_ABLATE);
#else
    a.ablate = 0;
#endif
    a.tap_inner = (a.ntaps > 1 && Cxp % 64 == 0) ? 1 : 0;
    a.m_begin = 0;
    a.m_end = a.M;
    a.splits = p.splits;
    a.slab = p.splits > 1 ? static_cast<float *>(workspace) : nullptr;
#ifdef EMADAPT_ABLATE_BUILD
    if constexpr (XBF && YBF) {   // diagnostic builds: the 256x256 kernel on bf16-stored tensors with parts compiled out
        if (p.big)
            switch (a.ablate) {
#define EMADAPT_ABL_CASE(m) case m: return launch_igemm_bf16<256, 256, 128, true, true, m>(a, Cxp, stream);
                EMADAPT_ABL_CASE(1) EMADAPT_ABL_CASE(8) EMADAPT_ABL_CASE(16) EMADAPT_ABL_CASE(2) EMADAPT_ABL_CASE(3) EMADAPT_ABL_CASE(7)
                EMADAPT_ABL_CASE(32) EMADAPT_ABL_CASE(64) EMADAPT_ABL_CASE(96) EMADAPT_ABL_CASE(35) EMADAPT_ABL_CASE(39) EMADAPT_ABL_CASE(4)
#undef EMADAPT_ABL_CASE
                default: break;
            }
    }
#endif
    return p.big      ? launch_igemm_bf16<256, 256, 128, XBF, YBF>(a, Cxp, stream)
           : p.narrow ? launch_igemm_bf16<128, 64, 32, XBF, YBF>(a, Cxp, stream)
                      : launch_igemm_bf16<128, 128, 64, XBF, YBF>(a, Cxp, stream);
}

// storage-type dispatch: (x, y) in {f32, bf16}^2
int dispatch_igemm_bf16_dt(IgemmArgs &a, int Cxp, int x_bf16, int y_bf16, void *workspace, size_t workspace_bytes,
                           hipStream_t stream) {
    if (y_bf16 && a.N % 8 != 0) return EMADAPT_ERR_UNSUPPORTED;  // 16-byte vectors of 8 bf16 channels
    if (x_bf16) {
        if (a.Cx % 8 != 0) return EMADAPT_ERR_UNSUPPORTED;  // 16-byte loads of 8 bf16 channels
        a.x_bytes = (unsigned)((long long)a.M * a.Cx * 2);
        return y_bf16 ? dispatch_igemm_bf16<true, true>(a, Cxp, workspace, workspace_bytes, stream)
                      : dispatch_igemm_bf16<true, false>(a, Cxp, workspace, workspace_bytes, stream);
    }
    return y_bf16 ? dispatch_igemm_bf16<false, true>(a, Cxp, workspace, workspace_bytes, stream)
                  : dispatch_igemm_bf16<false, false>(a, Cxp, workspace, workspace_bytes, stream);
}

// ------------------------------------------------------------------------------------------------------------------
// Filter gradient: dw[k][n] = sum_m A[m][k] dy[m][n]; 128 x BN filter tiles, 64 pixels per reduction step, split-M.
// ------------------------------------------------------------------------------------------------------------------
struct WgradBf16Args {
    const float *x;   // [B,H,W,Cin]
    const float *dy;  // [M][N]
    float *out;       // dw or slab base: [splits][K][N]
    float *bias_out;  // null, or the partial column sums of dy (BiasAddGrad): [splits][tiles_k][N] -- the workgroup of k tile t adds up
                      // the K steps st with st % tiles_k == t of its split and columns (db itself when splits * tiles_k == 1); needs dy stored bf16
    int H, W, Cin, N, KW, dil, pad_h, pad_w;
    int M, K, ntaps;
    int tiles_k, tiles_n, splits;
    int rows_per_split;  // multiple of kRowsUnit
    unsigned x_bytes, y_bytes;
    unsigned magic_w, magic_h;
};

constexpr int kRowsUnit = 64;  // rows_per_split is a multiple of this (covers both reduction step sizes)

__device__ __forceinline__ bf16x8 tr_frag(const char *p0, const char *p1) {
    typedef __attribute__((address_space(3))) bf16x4 lds_bf16x4;
    const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4 *)(p0));
    const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4 *)(p1));
    return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
}

// Tile configurations <BKO, BN, BMR>: BKO x BN filter tile, BMR pixels per reduction step; consumer waves own 64 x BN/2
// blocks ((BKO/64) x 2 of them), 4 producer waves.  LDS rows are pixels: pitch = columns*2 + 64 bytes, so that the four
// pixel rows of one transposing read sit 16 banks apart.
//   128 x 64|128, 64 px : 4 + 4 waves, 2 workgroups/CU
//   256 x 256,    32 px : 8 + 4 waves, 1 workgroup/CU (168 registers/wave) -- half the operand bytes per FLOP
// Storage types: XBF / DBF = x / dy are bf16 in memory (Cin resp. Cout % 8 == 0; 16-byte loads carry 8 channels).
template <int BKO, int BN, int BMR, bool XBF, bool DBF>
__global__ void __launch_bounds__((BKO / 64 * 2 + 4) * 64, BKO == 128 ? 4 : 3) wgrad_bf16_kernel(WgradBf16Args a) {
    constexpr int WK = 64, WN = BN / 2;
    constexpr int TK = WK / 32, TN = WN / 32;
    constexpr int CTHREADS = (BKO / WK) * 2 * 64;
    constexpr int XE = XBF ? 8 : 4, YE = DBF ? 8 : 4;  // channels per 16-byte piece
    constexpr int XQ = BKO / XE, X_RSTEP = 256 / XQ, X_N = BMR / X_RSTEP;
    constexpr int YQ = BN / YE, Y_RSTEP = 256 / YQ, Y_N = BMR / Y_RSTEP;
    constexpr int PX = BKO * 2 + 64, PY = BN * 2 + 64;  // LDS row pitches in bytes
    constexpr int TILE_X = BMR * PX, TILE_Y = BMR * PY;
    static_assert(X_N >= 1 && Y_N >= 1, "tile too small for the producers");

    extern __shared__ __attribute__((aligned(16))) char ldsb[];
    char *Xs = ldsb;                // [2][BMR][PX]
    char *Ys = ldsb + 2 * TILE_X;   // [2][BMR][PY]

    // XCD-aware block order (blocks b, b+8, .. share an XCD and its L2): each XCD takes a contiguous run of logical
    // blocks in split-major order -- so the workgroups of one XCD stream the SAME pixel range -- and inside a split
    // the filter tiles go in groups of GK k-tiles x all n-tiles (k fastest).
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
        // ================================ PRODUCER wavefronts ================================
        const int tid = threadIdx.x - CTHREADS;
        const int x_kq = tid % XQ, x_r0 = tid / XQ;
        const int k = k0 + x_kq * XE;
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
        const int y_n = n0 + y_nq * YE;
        const bool nvalid = y_n < a.N;
        u32x4 rx[X_N], ry[Y_N];
        // BiasAddGrad rides along (DBF): every dy piece of the workgroup's pixel range and columns passes through a producer register
        // exactly once (rows past the range arrive as zeros).  The workgroups that share a pixel range and columns -- one per k tile
        // -- divide the steps among them (step st belongs to k tile st % tiles_k); each lane keeps the sums of its 8 columns
        const bool do_bias = DBF && a.bias_out != nullptr;
        int bias_st = tile_k;   // the next step whose dy pieces this workgroup adds up
        float bs[8];
#pragma unroll
        for (int e = 0; e < 8; e++) bs[e] = 0.f;
        auto as_f4 = [](const u32x4 &v) {
            return make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
        };
        auto load_x = [&](int mb, int j) {
            const int m = mb + x_r0 + j * X_RSTEP;
            const unsigned t2 = div_magic((unsigned)m, a.magic_w);
            const int ow = m - (int)t2 * a.W;
            const int oh = (int)(t2 - div_magic(t2, a.magic_h) * (unsigned)a.H);
            const int ih = oh + dh, iw = ow + dw;
            const bool ok = kvalid && m < m_end && (unsigned)ih < (unsigned)a.H && (unsigned)iw < (unsigned)a.W;
            rx[j] = __builtin_amdgcn_raw_buffer_load_b128(xr, (int)(((unsigned)(m * a.Cin + x_off) * (XBF ? 2u : 4u)) | (ok ? 0u : kOOB)), 0, 0);
        };
        auto load_y = [&](int mb, int j) {
            const int m = mb + y_r0 + j * Y_RSTEP;
            ry[j] = __builtin_amdgcn_raw_buffer_load_b128(yr, (int)(((unsigned)(m * a.N + y_n) * (DBF ? 2u : 4u)) | ((nvalid && m < m_end) ? 0u : kOOB)), 0, 0);
        };
        auto store_x = [&](int buf, int j) {
            char *d = Xs + buf * TILE_X + (x_r0 + j * X_RSTEP) * PX + x_kq * (XE * 2);
            if constexpr (XBF) *reinterpret_cast<u32x4 *>(d) = rx[j];
            else *reinterpret_cast<uint2 *>(d) = cvt4(as_f4(rx[j]));
        };
        auto store_y = [&](int buf, int j, bool sum) {
            char *d = Ys + buf * TILE_Y + (y_r0 + j * Y_RSTEP) * PY + y_nq * (YE * 2);
            if constexpr (DBF) {
                *reinterpret_cast<u32x4 *>(d) = ry[j];
                if (sum) {
                    const unsigned w4[4] = {ry[j].x, ry[j].y, ry[j].z, ry[j].w};
#pragma unroll
                    for (int e = 0; e < 4; e++) {
                        bs[2 * e] += __uint_as_float(w4[e] << 16);
                        bs[2 * e + 1] += __uint_as_float(w4[e] & 0xffff0000u);
                    }
                }
            } else {
                *reinterpret_cast<uint2 *>(d) = cvt4(as_f4(ry[j]));
            }
        };
#pragma unroll
        for (int j = 0; j < X_N; j++) load_x(m_begin, j);
#pragma unroll
        for (int j = 0; j < Y_N; j++) load_y(m_begin, j);
#pragma unroll
        for (int j = 0; j < X_N; j++) { store_x(0, j); load_x(m_begin + BMR, j); }  // rows >= m_end: zero-filled
        {
            const bool sum = do_bias && bias_st == 0;
            if (sum) bias_st += a.tiles_k;
#pragma unroll
            for (int j = 0; j < Y_N; j++) { store_y(0, j, sum); load_y(m_begin + BMR, j); }
        }
        __syncthreads();
        // steady state: park register j in LDS, re-arm it with the load of the step after next (loads stay in flight)
        for (int st = 0; st < nsteps; st++) {
            const int buf = (st + 1) & 1;
            const int mb = m_begin + (st + 2) * BMR;
#pragma unroll
            for (int j = 0; j < X_N; j++) { store_x(buf, j); load_x(mb, j); }
            const bool sum = do_bias && bias_st == st + 1;   // (this iteration parks step st + 1)
            if (sum) bias_st += a.tiles_k;
#pragma unroll
            for (int j = 0; j < Y_N; j++) { store_y(buf, j, sum); load_y(mb, j); }
            __syncthreads();
        }
        if constexpr (DBF) {
            if (do_bias) {   // (workgroup-uniform: the MFMA waves meet the barrier below too)
                // lanes of a wave with the same columns (lane % YQ), then the four producer waves through the idle X stage, in a fixed order
#pragma unroll
                for (int sft = YQ; sft < 64; sft <<= 1)
#pragma unroll
                    for (int e = 0; e < 8; e++) bs[e] += __shfl_xor(bs[e], sft);
                float *sh = reinterpret_cast<float *>(ldsb);   // [4][BN]
                const int lane = tid & 63, pw = tid >> 6;
                if (lane < YQ) {
#pragma unroll
                    for (int e = 0; e < 8; e++) sh[pw * BN + lane * 8 + e] = bs[e];
                }
                __syncthreads();
                if (tid < BN && n0 + tid < a.N)
                    a.bias_out[((long long)split * a.tiles_k + tile_k) * a.N + n0 + tid] = (sh[tid] + sh[BN + tid]) + (sh[2 * BN + tid] + sh[3 * BN + tid]);
            }
        }
        return;
    }

    // ================================ CONSUMER wavefronts ================================
    // Transposing read: in the 16-lane group g = lane>>4, lane 4q+p supplies the address of pixel row q, columns
    // 4p..4p+3 of the group's 16 columns, and receives column (lane&15) of the four rows.  For the 32x32x16 operand
    // lane l holds column l&31 and pixels 8*(l>>5) + 0..7: group g covers columns 16*(g&1).. and pixel half g>>1.
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wk = wave >> 1, wn = wave & 1;
    const int l31 = lane & 31, lh = lane >> 5;
    const int grp = lane >> 4, q = (lane >> 2) & 3, p = lane & 3;
    const int frow = (grp >> 1) * 8 + q;              // pixel row inside a 16-pixel MFMA step (+4 for the upper half)
    const int fcol = ((grp & 1) * 16 + p * 4) * 2;    // byte offset of the lane's 4 columns
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
        bf16x8 av[2][TK], bv[TN];
#pragma unroll
        for (int i = 0; i < TK; i++) av[0][i] = tr_frag(xs + i * 64, xs + i * 64 + 4 * PX);
#pragma unroll
        for (int ms = 0; ms < BMR / 16; ms++) {
#pragma unroll
            for (int j = 0; j < TN; j++) bv[j] = tr_frag(ys + ms * 16 * PY + j * 64, ys + ms * 16 * PY + j * 64 + 4 * PY);
            if (ms + 1 < BMR / 16) {
                const int o = (ms + 1) * 16 * PX;
#pragma unroll
                for (int i = 0; i < TK; i++) av[(ms + 1) & 1][i] = tr_frag(xs + o + i * 64, xs + o + i * 64 + 4 * PX);
            }
#pragma unroll
            for (int j = 0; j < TN; j++)
#pragma unroll
                for (int i = 0; i < TK; i++)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av[ms & 1][i], bv[j], acc[i][j], 0, 0, 0);
        }
        __syncthreads();
    }
    if (DBF && a.bias_out != nullptr) __syncthreads();   // (the producers' bias-gradient exchange)
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

struct PlanW {
    int bko, bn, tiles_k, tiles_n, splits, rows_per_split;
};

// Split-M plan as in conv_wgrad.hip, with the bf16 kernels' rates; the 256x256 filter tile when the filter is at
// least that large (conv3_x .. fc7).
PlanW make_plan_bf16(const emadapt_conv_desc *d) {
    PlanW p;
    const long long M = (long long)d->B * d->H * d->W;
    const int K = d->KH * d->KW * d->Cin;
    const int force = tune(TUNE_BF16_TILE);  // 128|256 (the parity tests run both; 1 / 3 concern the forward / data-gradient patch kernel only)
    bool big = K >= 256 && d->Cout >= 256;
    if (force == 128 || force == 256) big = force == 256;
    p.bko = big ? 256 : 128;
    p.bn = big ? 256 : (d->Cout <= 64 ? 64 : 128);
    p.tiles_k = (int)ceil_div(K, p.bko);
    p.tiles_n = (int)ceil_div(d->Cout, p.bn);
    const long long tiles = (long long)p.tiles_k * p.tiles_n;
    const double slots = big ? 256.0 : 512.0;
    const double rate = big ? 800e12 : 450e12;
    const double t_kernel = 2.0 * (double)M * (p.tiles_k * (double)p.bko) * (p.tiles_n * (double)p.bn) / rate;
    const double kn_bytes = (double)K * d->Cout * 4.0;
    long long best = 1;
    double best_t = 1e30;
    const long long max_s = ceil_div(M, 512) < 2048 ? ceil_div(M, 512) : 2048;
    for (long long s = 1; s <= max_s; s++) {
        const double rounds = (double)tiles * s / slots;
        const double eff = rounds / (double)(long long)(rounds + 0.999999);
        double t = t_kernel / eff + 2e-6 * (double)(long long)(rounds + 0.999999);
        if (s > 1) t += (double)(s + 1) * kn_bytes / 3e12 + 3e-6;
        if (t < best_t * 0.995) { best_t = t; best = s; }
        if (rounds > 24.0) break;
    }
    long long rps = ceil_div(M, best);
    rps = ceil_div(rps, kRowsUnit) * kRowsUnit;
    p.rows_per_split = (int)rps;
    p.splits = (int)ceil_div(M, rps);
    return p;
}

int check_desc_wgrad(const emadapt_conv_desc *d) {
    int rc = check_desc(d);
    if (rc) return rc;
    const long long M = (long long)d->B * d->H * d->W;
    if (M * (d->W > d->H ? d->W : d->H) >= (1LL << 32)) return EMADAPT_ERR_UNSUPPORTED;  // magic division range
    return EMADAPT_OK;
}

template <int BKO, int BN, int BMR, bool XBF, bool DBF>
int launch_wgrad_bf16(const WgradBf16Args &a, hipStream_t stream) {
    constexpr size_t lds = 2 * BMR * (BKO * 2 + 64 + BN * 2 + 64);
    constexpr int threads = (BKO / 64 * 2 + 4) * 64;
    static DynLdsAttr attr;
    {
        const int arc = attr.ensure(reinterpret_cast<const void *>(&wgrad_bf16_kernel<BKO, BN, BMR, XBF, DBF>), lds);
        if (arc) return arc;
    }
    const long long nblk = (long long)a.tiles_k * a.tiles_n * a.splits;
    if (nblk > 0x7fffffffLL) return EMADAPT_ERR_UNSUPPORTED;
    hipLaunchKernelGGL((wgrad_bf16_kernel<BKO, BN, BMR, XBF, DBF>), dim3((unsigned)nblk), dim3(threads), lds, stream, a);
    return launch_status();
}

}  // namespace

#include "wgrad_b16t.h"   // the LDS-DMA filter-gradient kernel for bf16-stored tensors (uses WgradBf16Args, the bt_* vector types)
#include "wgrad_d64.h"    // direct filter gradient for 3 x 3, 64 -> 64 channels (conv1_2): the whole gradient in registers, operands fetched once

namespace {

// tune "wgrad_gen" = 1 keeps the round-1 kernel everywhere (A/B, parity of both)
template <bool XBF, bool DBF>
int launch_wgrad_bf16_plan(const PlanW &p, const WgradBf16Args &a, hipStream_t stream) {
    if constexpr (XBF && DBF) {
        if (p.bko == 256 && a.Cin % 256 == 0 && a.N % 8 == 0 && tune(TUNE_WGRAD_GEN) != 1) return launch_wgrad_b16t(a, stream);
    }
    return p.bko == 256 ? launch_wgrad_bf16<256, 256, 32, XBF, DBF>(a, stream)
         : p.bn == 64   ? launch_wgrad_bf16<128, 64, 64, XBF, DBF>(a, stream)
                        : launch_wgrad_bf16<128, 128, 64, XBF, DBF>(a, stream);
}

// ------------------------------------------------------------------------------------------------------------------
// LDS-DMA kernel (igemm_b16t.h): launch plan.  One workgroup per CU: whole rounds of tiles run as plain tiles; whatever is not a
// whole number of rounds runs stream-K (N <= 1024: the whole launch; wider outputs: the rows of the partial last round only, so
// the slab levels stay small).
// ------------------------------------------------------------------------------------------------------------------
struct PlanBT {
    int main_rows;     // rows of the first launch
    int main_sk;       // 1: stream-K
    int main_levels;   // its slab levels
    int tail_levels;   // rows [main_rows, M): a stream-K launch with this many slab levels (-1: a plain launch)
    int m224;          // 1: the (single, plain) launch runs 224 x 256 tiles (igemm_b16t.h M224)
};

void fill_b16t(IgemmArgs &a, const emadapt_conv_desc *d, bool dgrad) {
    fill_geometry(a, d, dgrad);
    a.x_bytes = (unsigned)((long long)a.M * a.Cx * 2);
    a.w_bytes = (unsigned)((long long)a.K / BT_BK * ceil_div(a.N, 256) * 256 * BT_ROWB);   // tiled filter image: rows padded to whole tiles
}

bool bt_ok(const emadapt_conv_desc *d, bool dgrad, int y_bf16) {
    if (check_desc(d) != EMADAPT_OK) return false;
    const long long M = (long long)d->B * d->H * d->W;
    const int Cx = dgrad ? d->Cout : d->Cin, N = dgrad ? d->Cin : d->Cout;
    const int ntaps = d->KH * d->KW;
    const long long img_x = M * Cx * 2, img_w = (long long)ntaps * (Cx / BT_BK) * ceil_div(N, 256) * 256 * BT_ROWB;
    // (per-lane offset + scalar offset must stay below bit 31, the out-of-range flag: image + the taps' reach + one pixel row)
    const long long reach = ((long long)(d->KH - 1) * d->dil * d->W + (long long)(d->KW - 1) * d->dil + 1) * Cx * 2;
    return Cx % BT_BK == 0 && ntaps <= 32 && N % (y_bf16 ? 8 : 4) == 0 && img_x + reach < (1LL << 31) && img_w < (1LL << 31) &&
           M * N < (1LL << 40);
}

PlanBT plan_b16t(const IgemmArgs &a) {
    const int cus = plan_cus();
    PlanBT p = {a.M, 0, 0, 0, 0};
    constexpr int BM = 256, BN = 256;
    const long long tn = ceil_div(a.N, BN), tm = ceil_div(a.M, BM), t = tm * tn;
    const int bk = BT_BK;
    const int nk = a.K / bk;
    // Round 6: a plain launch of ONE round that leaves CUs idle runs 224-row tiles when those still fit the round: 7 / 8 of the work each on
    // more of the CUs (M = 26,896 at N = 512: 242 workgroups instead of 212) -- tune "bf16t_sk" = 3 keeps the 256-row tiles (A/B)
    auto one_round = [&](PlanBT q) {
        const long long t224 = ceil_div(a.M, 224) * tn;
        if (t < cus && t224 <= cus && t224 > t && tune(TUNE_BF16T_SK) != 3) q.m224 = 1;
        return q;
    };
    if (t % cus == 0 || t * nk * bk / cus < 16 * BT_BK) return t <= cus ? one_round(p) : p;   // whole rounds, or too little work per CU to cut into pieces
    // one round that fills at least three quarters of the CUs (conv4_x / conv5_x: 212 tiles): plain tiles -- stream-K's second
    // prologue, published and combined accumulator image per CU cost more than the idle quarter (0.107 against 0.119 ms on conv4_2)
    if (t <= cus && 4 * t >= 3LL * cus) return one_round(p);
    // four and more rounds (fc6 / fc7: 6.6): plain tiles all the way -- the last round's idle CUs (a sixth of one round in seven) cost less than
    // the second launch, its prologue and the combined tiles of a stream-K tail: fc7 forward 0.697 -> 0.667 ms, data gradient 0.760 -> 0.727,
    // fc6 forward 1.201 -> 1.193 (same process, same tensors, A/B/A/B).  Fewer rounds keep whole rounds + stream-K tail below.
    if (t >= 4LL * cus && (tune(TUNE_BF16T_SK) == 0 || tune(TUNE_BF16T_SK) == 3)) return p;
    if (a.N <= 1024 || t < cus || tune(TUNE_BF16T_SK) == 1) {
        p.main_sk = 1;
        p.main_levels = streamk_levels(t, nk, cus);
        return p;
    }
    const long long tm_full = (t / cus) * cus / tn;
    if (tm_full >= 1 && tm_full < tm) {
        p.main_rows = (int)(tm_full * BM);
        const long long tail_tiles = (tm - tm_full) * tn;
        p.tail_levels = tail_tiles * nk * bk / cus >= 16 * BT_BK ? streamk_levels(tail_tiles, nk, cus) : -1;
    }
    return p;
}

// stream-K launches with shared tiles: [arrival words, padded][one 256 KB accumulator image per workgroup] (igemm_v2.h)
size_t bt_workspace_bytes(const IgemmArgs &a) {
    const PlanBT p = plan_b16t(a);
    const bool shared = (p.main_sk && p.main_levels > 0) || (p.main_rows < a.M && p.tail_levels > 0);
    return shared ? streamk_workspace_bytes(plan_cus()) : 0;
}

template <bool SK, bool YBF>
int launch_bt(IgemmArgs &a, hipStream_t stream) { return launch_igemm_b16t<SK, YBF>(a, stream); }

template <bool YBF>
int dispatch_b16t(IgemmArgs &a, void *workspace, size_t workspace_bytes, hipStream_t stream) {
    const PlanBT p = plan_b16t(a);
    if (bt_workspace_bytes(a) > 0 && (!workspace || workspace_bytes < bt_workspace_bytes(a))) return EMADAPT_ERR_WORKSPACE;
    a.ablate = 0;
    a.tap_inner = 1;
    a.pk.n = 0;
    a.pk.rowmap = nullptr;
    a.m_begin = 0;
    a.m_end = p.main_rows;
    auto sk_ws = [&](int levels) {   // the stream-K workspace: arrival words, then the accumulator images
        a.sk_flags = levels > 0 ? static_cast<unsigned *>(workspace) : nullptr;
        a.slab = levels > 0 ? reinterpret_cast<float *>(static_cast<char *>(workspace) + streamk_flag_bytes(plan_cus())) : nullptr;
    };
    if (p.main_sk) {
        a.splits = plan_cus();
        sk_ws(p.main_levels);
        return launch_bt<true, YBF>(a, stream);
    }
    a.splits = 1;
    a.slab = nullptr;
    if (p.m224) return launch_igemm_b16t<false, YBF, 0, true>(a, stream);
    int rc = launch_bt<false, YBF>(a, stream);
    if (rc || p.main_rows >= a.M) return rc;
    a.m_begin = p.main_rows;
    a.m_end = a.M;
    if (p.tail_levels < 0) return launch_bt<false, YBF>(a, stream);
    a.splits = plan_cus();
    sk_ws(p.tail_levels);
    return launch_bt<true, YBF>(a, stream);
}

}  // namespace

extern "C" {

size_t emadapt_weights_bf16_bytes(const emadapt_conv_desc *d, int which) {
    if (check_desc(d) != EMADAPT_OK) return 0;
    const size_t taps = (size_t)d->KH * d->KW;
    return which == 0 ? taps * d->Cout * round_up8(d->Cin) * 2 : taps * d->Cin * round_up8(d->Cout) * 2;
}

int emadapt_weights_to_bf16(const emadapt_conv_desc *d, const float *w, void *w_fwd_bf16, void *w_dgrad_bf16,
                            void *stream_) {
    int rc = check_desc(d);
    if (rc) return rc;
    if (!w || (!w_fwd_bf16 && !w_dgrad_bf16)) return EMADAPT_ERR_INVALID;
    if (!aligned16(w) || !aligned16(w_fwd_bf16) || !aligned16(w_dgrad_bf16)) return EMADAPT_ERR_INVALID;
    const int Cip = round_up8(d->Cin), Cop = round_up8(d->Cout);
    const int tiles_ci = (int)ceil_div(Cip, 32), tiles_co = (int)ceil_div(Cop, 32);
    const long long nblk = (long long)d->KH * d->KW * tiles_ci * tiles_co;
    hipLaunchKernelGGL(weights_to_bf16_kernel, dim3((unsigned)nblk), dim3(256), 0, as_stream(stream_), w,
                       static_cast<__bf16 *>(w_fwd_bf16), static_cast<__bf16 *>(w_dgrad_bf16), d->Cin, d->Cout, Cip,
                       Cop, tiles_ci, tiles_co);
    return launch_status();
}

int emadapt_weights_to_bf16_multi(int n, const emadapt_conv_desc *const *d, const float *const *w, void *const *w_fwd_bf16,
                                  void *const *w_dgrad_bf16, void *stream_) {
    if (n < 0 || n > kW16MaxLayers || (n && (!d || !w || !w_fwd_bf16 || !w_dgrad_bf16))) return EMADAPT_ERR_INVALID;
    W16Layers L;
    long long tot = 0;
    L.count = 0;
    for (int k = 0; k < n; k++) {
        int rc = check_desc(d[k]);
        if (rc) return rc;
        if (!w[k] || (!w_fwd_bf16[k] && !w_dgrad_bf16[k]) || !aligned16(w[k]) || !aligned16(w_fwd_bf16[k]) || !aligned16(w_dgrad_bf16[k]))
            return EMADAPT_ERR_INVALID;
        const int c = L.count++;
        L.w[c] = w[k]; L.wf[c] = static_cast<__bf16 *>(w_fwd_bf16[k]); L.wd[c] = static_cast<__bf16 *>(w_dgrad_bf16[k]);
        L.Cin[c] = d[k]->Cin; L.Cout[c] = d[k]->Cout;
        L.tiles_ci[c] = (int)ceil_div(round_up8(d[k]->Cin), 32); L.tiles_co[c] = (int)ceil_div(round_up8(d[k]->Cout), 32);
        L.first[c] = (int)tot;
        tot += (long long)d[k]->KH * d[k]->KW * L.tiles_ci[c] * L.tiles_co[c];
        if (tot > 0x7fffffffLL) return EMADAPT_ERR_UNSUPPORTED;
    }
    for (int k = L.count; k <= kW16MaxLayers; k++) L.first[k] = (int)tot;
    for (int k = L.count; k < kW16MaxLayers; k++) { L.w[k] = nullptr; L.wf[k] = nullptr; L.wd[k] = nullptr; L.Cin[k] = L.Cout[k] = L.tiles_ci[k] = L.tiles_co[k] = 0; }
    if (tot == 0) return EMADAPT_OK;
    hipLaunchKernelGGL(weights_to_bf16_multi_kernel, dim3((unsigned)tot), dim3(256), 0, as_stream(stream_), L);
    return launch_status();
}

static PlanW plan_wgrad_bf16(const emadapt_conv_desc *d, int x_dtype, int dy_dtype);

int emadapt_conv_bf16_plan(const emadapt_conv_desc *d, int kind, int plan[3]) {
    if (!plan || kind < 0 || kind > 2) return EMADAPT_ERR_INVALID;
    int rc = kind == 2 ? check_desc_wgrad(d) : check_desc(d);
    if (rc) return rc;
    if (kind == 2) {
        const PlanW p = plan_wgrad_bf16(d, 1, 1);   // (bf16-stored tensors assumed: what the model keeps)
        plan[0] = p.bko ? p.bko : 4; plan[1] = p.bko ? p.bn : 16; plan[2] = p.splits;   // (4 x 16: the direct kernel, 16 x 16-pixel tiles, plan[2] slabs)
        return EMADAPT_OK;
    }
    IgemmArgs a;
    int Cxp;
    fill_geometry_bf16(a, Cxp, d, kind == 1);
    if (tune(TUNE_BF16_TILE) == 0 && d->B > 0 && d64_shape_ok(a)) {   // (bf16-stored tensors assumed: what the model keeps)
        plan[0] = 4; plan[1] = 16; plan[2] = 16;
        return EMADAPT_OK;
    }
    if (tune(TUNE_BF16_TILE) == 0 && d->B > 0 && kind == 1 && k32_shape_ok(a)) {   // (bf16-stored tensors assumed; the forward has a bias)
        plan[0] = 6; plan[1] = 64; plan[2] = 256;
        return EMADAPT_OK;
    }
    if (tune(TUNE_BF16_TILE) == 0 && d->B > 0 && kind == 0 && d4_shape_ok(a)) {   // (the f32 image in, bf16 out: conv1_1 as the model runs it)
        plan[0] = 5; plan[1] = 16; plan[2] = 16;
        return EMADAPT_OK;
    }
    const PlanG p = plan_bf16(a);
    plan[0] = p.big ? 256 : 128;
    plan[1] = p.big ? 256 : p.narrow ? 64 : 128;
    plan[2] = p.splits;
    return EMADAPT_OK;
}

size_t emadapt_conv_fwd_bf16_workspace_bytes(const emadapt_conv_desc *d) {
    if (check_desc(d) != EMADAPT_OK || d->B == 0) return 0;
    IgemmArgs a;
    int Cxp;
    fill_geometry_bf16(a, Cxp, d, false);
    return igemm_bf16_workspace_bytes(a);
}

size_t emadapt_conv_dgrad_bf16_workspace_bytes(const emadapt_conv_desc *d) {
    if (check_desc(d) != EMADAPT_OK || d->B == 0) return 0;
    IgemmArgs a;
    int Cxp;
    fill_geometry_bf16(a, Cxp, d, true);
    return igemm_bf16_workspace_bytes(a);
}

int emadapt_conv_fwd_bf16(const emadapt_conv_desc *d, const void *x, int x_dtype, const void *w_fwd_bf16,
                          const float *bias, void *y, int y_dtype, int epilogue, float keep_prob,
                          const uint8_t *drop_mask, uint64_t drop_seed, void *workspace, size_t workspace_bytes,
                          void *stream_) {
    int rc = check_desc(d);
    if (rc) return rc;
    if (!x || !w_fwd_bf16 || !y) return EMADAPT_ERR_INVALID;
    if ((x_dtype | y_dtype) & ~1) return EMADAPT_ERR_INVALID;
    if (epilogue < EMADAPT_EPI_NONE || epilogue > EMADAPT_EPI_BIAS_RELU) return EMADAPT_ERR_INVALID;
    if (epilogue != EMADAPT_EPI_NONE && !bias) return EMADAPT_ERR_INVALID;
    if (!(keep_prob > 0.f) || keep_prob > 1.f) return EMADAPT_ERR_INVALID;
    if (keep_prob < 1.f && epilogue != EMADAPT_EPI_BIAS_RELU) return EMADAPT_ERR_INVALID;
    if (!aligned16(x) || !aligned16(w_fwd_bf16) || !aligned16(y) || !aligned16(workspace)) return EMADAPT_ERR_INVALID;
    if (d->B == 0) return EMADAPT_OK;
    IgemmArgs a;
    int Cxp;
    fill_geometry_bf16(a, Cxp, d, false);
    a.x = static_cast<const float *>(x); a.w = static_cast<const float *>(w_fwd_bf16); a.y = static_cast<float *>(y);
    a.bias = bias; a.gate = nullptr; a.mask = drop_mask;
    a.epi = epilogue; a.scale = 1.f; set_keep(a, keep_prob); a.seed = drop_seed;
    hipStream_t stream = as_stream(stream_);
    ProfScope prof(EMADAPT_K_IGEMM_FWD, 2.0 * a.M * (double)(a.ntaps * a.Cx) * a.N,
                   (double)a.M * a.Cx * (x_dtype ? 2 : 4) + 2.0 * a.ntaps * a.Cx * a.N + (double)a.M * a.N * (y_dtype ? 2 : 4), stream);
    return dispatch_igemm_bf16_dt(a, Cxp, x_dtype, y_dtype, workspace, workspace_bytes, stream);
}

int emadapt_conv_dgrad_bf16(const emadapt_conv_desc *d, const void *dy, int dy_dtype, const void *w_dgrad_bf16,
                            void *dx, int dx_dtype, const void *gate, float gate_scale, void *workspace,
                            size_t workspace_bytes, void *stream_) {
    int rc = check_desc(d);
    if (rc) return rc;
    if (!dy || !w_dgrad_bf16 || !dx) return EMADAPT_ERR_INVALID;
    if ((dy_dtype | dx_dtype) & ~1) return EMADAPT_ERR_INVALID;
    if (!aligned16(dy) || !aligned16(w_dgrad_bf16) || !aligned16(dx) || !aligned16(workspace)) return EMADAPT_ERR_INVALID;
    if (d->B == 0) return EMADAPT_OK;
    IgemmArgs a;
    int Cxp;
    fill_geometry_bf16(a, Cxp, d, true);
    a.x = static_cast<const float *>(dy); a.w = static_cast<const float *>(w_dgrad_bf16); a.y = static_cast<float *>(dx);
    a.bias = nullptr; a.gate = static_cast<const float *>(gate); a.mask = nullptr;
    a.epi = EMADAPT_EPI_NONE; a.scale = gate_scale; set_keep(a, 1.f); a.seed = 0;
    hipStream_t stream = as_stream(stream_);
    ProfScope prof(EMADAPT_K_IGEMM_DGRAD, 2.0 * a.M * (double)(a.ntaps * a.Cx) * a.N,
                   (double)a.M * a.Cx * (dy_dtype ? 2 : 4) + 2.0 * a.ntaps * a.Cx * a.N + (double)a.M * a.N * (dx_dtype ? 2 : 4) * (gate ? 2.0 : 1.0), stream);
    return dispatch_igemm_bf16_dt(a, Cxp, dy_dtype, dx_dtype, workspace, workspace_bytes, stream);
}

// The plan of a filter-gradient launch given the storage types: the direct kernel (wgrad_d64.h; bko == 0 marks it, one slab per workgroup)
// where it applies, else make_plan_bf16's.  tune "wgrad_gen" = 1 or any forced "bf16_tile" keep the split-M kernels.
static PlanW plan_wgrad_bf16(const emadapt_conv_desc *d, int x_dtype, int dy_dtype) {
    PlanW p = make_plan_bf16(d);
    if (wgrad_d64_shape_ok(d, x_dtype, dy_dtype) && tune(TUNE_WGRAD_GEN) != 1 && tune(TUNE_BF16_TILE) == 0) {
        p.bko = 0; p.bn = 64; p.tiles_k = 1; p.tiles_n = 1; p.splits = wgrad_d64_groups(d); p.rows_per_split = 0;
    }
    return p;
}

// (the queries do not know the storage types: the larger of the two plans' needs where the direct kernel could be chosen)
static size_t wgrad_bf16_ws_floats(const emadapt_conv_desc *d, bool bias) {
    size_t need = 0;
    for (int direct = 0; direct < 2; direct++) {
        const PlanW p = direct ? plan_wgrad_bf16(d, 1, 1) : make_plan_bf16(d);
        const size_t w_floats = p.splits > 1 ? (size_t)p.splits * d->KH * d->KW * d->Cin * d->Cout : 0;
        const size_t b_rows = bias ? (size_t)p.splits * p.tiles_k : 0;
        const size_t n = w_floats + (b_rows > 1 ? b_rows * d->Cout : 0);
        if (n > need) need = n;
    }
    return need;
}

size_t emadapt_conv_wgrad_bf16_workspace_bytes(const emadapt_conv_desc *d) {
    if (check_desc_wgrad(d) != EMADAPT_OK || d->B == 0) return 0;
    return wgrad_bf16_ws_floats(d, false) * sizeof(float);
}

static int wgrad_bf16_impl(const emadapt_conv_desc *d, const void *x, int x_dtype, const void *dy, int dy_dtype,
                           float *dw, float *db, void *workspace, size_t workspace_bytes, void *stream_) {
    int rc = check_desc_wgrad(d);
    if (rc) return rc;
    if (!x || !dy || !dw) return EMADAPT_ERR_INVALID;
    if ((x_dtype | dy_dtype) & ~1) return EMADAPT_ERR_INVALID;
    if ((x_dtype && d->Cin % 8 != 0) || (dy_dtype && d->Cout % 8 != 0)) return EMADAPT_ERR_UNSUPPORTED;
    if (!aligned16(x) || !aligned16(dy) || !aligned16(dw) || !aligned16(workspace) || !aligned16(db)) return EMADAPT_ERR_INVALID;
    hipStream_t stream = as_stream(stream_);
    const long long KN = (long long)d->KH * d->KW * d->Cin * d->Cout;
    if (d->B == 0) {
        EMADAPT_HIP_TRY(hipMemsetAsync(dw, 0, KN * sizeof(float), stream));
        if (db) EMADAPT_HIP_TRY(hipMemsetAsync(db, 0, (size_t)d->Cout * sizeof(float), stream));
        return EMADAPT_OK;
    }
    const PlanW p = plan_wgrad_bf16(d, x_dtype, dy_dtype);
    // workspace: [splits][K][N] partial filter gradients (splits > 1), then [splits][tiles_k][N] partial bias gradients (more than one row)
    const long long w_floats = p.splits > 1 ? (long long)p.splits * KN : 0;
    const int b_rows = db ? p.splits * p.tiles_k : 0;
    const long long b_floats = b_rows > 1 ? (long long)b_rows * d->Cout : 0;
    if (w_floats + b_floats > 0 && (!workspace || workspace_bytes < (size_t)(w_floats + b_floats) * sizeof(float))) return EMADAPT_ERR_WORKSPACE;
    WgradBf16Args a;
    a.x = static_cast<const float *>(x); a.dy = static_cast<const float *>(dy);
    a.out = p.splits > 1 ? static_cast<float *>(workspace) : dw;
    a.bias_out = !db ? nullptr : b_rows > 1 ? static_cast<float *>(workspace) + w_floats : db;
    a.H = d->H; a.W = d->W; a.Cin = d->Cin; a.N = d->Cout; a.KW = d->KW; a.dil = d->dil;
    a.pad_h = conv_pad_before(d->KH, d->dil); a.pad_w = conv_pad_before(d->KW, d->dil);
    a.M = d->B * d->H * d->W; a.ntaps = d->KH * d->KW; a.K = a.ntaps * d->Cin;
    a.tiles_k = p.tiles_k; a.tiles_n = p.tiles_n; a.splits = p.splits; a.rows_per_split = p.rows_per_split;
    a.x_bytes = (unsigned)((long long)a.M * d->Cin * (x_dtype ? 2 : 4));
    a.y_bytes = (unsigned)((long long)a.M * d->Cout * (dy_dtype ? 2 : 4));
    a.magic_w = magic_u32(d->W); a.magic_h = magic_u32(d->H);
    ProfScope prof(EMADAPT_K_IGEMM_WGRAD, 2.0 * a.M * (double)a.K * a.N, (double)a.x_bytes + (double)a.y_bytes + 4.0 * a.K * a.N, stream);
    if (p.bko == 0) rc = launch_wgrad_d64(a, stream);
    else
    rc = x_dtype ? (dy_dtype ? launch_wgrad_bf16_plan<true, true>(p, a, stream) : launch_wgrad_bf16_plan<true, false>(p, a, stream))
                 : (dy_dtype ? launch_wgrad_bf16_plan<false, true>(p, a, stream) : launch_wgrad_bf16_plan<false, false>(p, a, stream));
    if (rc) return rc;
    // one second-stage launch finishes both gradients (each in a fixed order).  (Round 6 measured the alternative the round-5 verdict
    // asked for -- the workgroups of a tile add up their partials INSIDE the launch, arrival word per tile, each workgroup 1 / S of the
    // tile, slab_sum's order, bit-identical -- and refused it: this launch moves a conv4 / conv5 layer's 75 MB in 12 us (6.3 TB/s, the
    // slabs still in the Infinity Cache); inside the launch the same bytes cost 24 us (every 16-byte load of another XCD's slab is
    // a trip to memory with a handful in flight per lane, and more in flight cost the MFMA loop accumulator spills) plus 5.4 us for
    // the fill node that zeroes the arrival words: 10.03 -> 10.20 ms per step.  DESIGN "round 6", profiles/r06_wgrad_combine.txt.)
    return launch_slab_sum2(static_cast<const float *>(workspace), p.splits, p.splits > 1 ? KN : 0, dw,
                            static_cast<const float *>(workspace) + w_floats, b_rows, b_floats ? d->Cout : 0, db, stream);
}

int emadapt_conv_wgrad_bf16(const emadapt_conv_desc *d, const void *x, int x_dtype, const void *dy, int dy_dtype,
                            float *dw, void *workspace, size_t workspace_bytes, void *stream_) {
    return wgrad_bf16_impl(d, x, x_dtype, dy, dy_dtype, dw, nullptr, workspace, workspace_bytes, stream_);
}

int emadapt_conv_wgrad_bias_bf16_fused(const emadapt_conv_desc *d, int x_dtype, int dy_dtype) {
    if (check_desc_wgrad(d) != EMADAPT_OK || ((x_dtype | dy_dtype) & ~1)) return 0;
    if (tune(TUNE_WGRAD_BIAS) == 1) return 0;
    return dy_dtype == EMADAPT_DT_BF16 && d->Cout % 8 == 0 && (!x_dtype || d->Cin % 8 == 0) ? 1 : 0;
}

size_t emadapt_conv_wgrad_bias_bf16_workspace_bytes(const emadapt_conv_desc *d) {
    if (check_desc_wgrad(d) != EMADAPT_OK || d->B == 0) return 0;
    return wgrad_bf16_ws_floats(d, true) * sizeof(float);
}

int emadapt_conv_wgrad_bias_bf16(const emadapt_conv_desc *d, const void *x, int x_dtype, const void *dy, int dy_dtype,
                                 float *dw, float *db, void *workspace, size_t workspace_bytes, void *stream_) {
    if (!db) return EMADAPT_ERR_INVALID;
    if (!emadapt_conv_wgrad_bias_bf16_fused(d, x_dtype, dy_dtype)) {
        const int rc = check_desc_wgrad(d);
        return rc ? rc : EMADAPT_ERR_UNSUPPORTED;
    }
    return wgrad_bf16_impl(d, x, x_dtype, dy, dy_dtype, dw, db, workspace, workspace_bytes, stream_);
}

/* ---- LDS-DMA kernel on bf16-stored activations and a TILED filter image (igemm_b16t.h) ---- */
int emadapt_conv_bf16t_supported(const emadapt_conv_desc *d, int kind, int y_dtype) {
    if (kind < 0 || kind > 1 || (y_dtype & ~1)) return 0;
    return bt_ok(d, kind == 1, y_dtype) ? 1 : 0;
}

/* Where the model sends a layer to this kernel instead of the register-staged one: the shapes it was measured faster on
 * (profiles/r05_bf16t_bench.txt, same process, same tensors) --
 *   * outputs of four and more rounds of 256 x 256 tiles on the CUs: fc6 / fc7 of the `ref` preset at batch 16 (forward 60.7 / 51.2 %
 *     of the dense matrix rate against 53.7 / 48.4 %, data gradient 55.3 / 46.5 against 50.8 / 45.0);
 *   * one and a half rounds and more with a long reduction (K >= 2304): conv3_2 / conv3_3 (411 tiles, stream-K with the in-launch
 *     combine: 41.6 / 40.2 % against 39.2 / 37.1), LargeFOV's fc6 forward (424 tiles: 46.5 against 40.0 %);
 *   * a thousand and more K steps per CU whatever the tile count: fc6's data gradient.
 *   * a single round of 212 tiles with K >= 4608 (conv4_2 .. conv5_3) as PLAIN tiles (as stream-K it only ties: two prologues, one
 *     published and one combined accumulator image per CU cost what the 17 % of idle CUs do).
 * Elsewhere the older kernel stays: short reductions (conv3_1, conv4_1, LargeFOV's fc7: 32 - 72 K steps per tile) lose to it, and
 * outputs narrower than a tile (conv1_2, conv2_x) waste the tile's MFMAs. */
int emadapt_conv_bf16t_pays(const emadapt_conv_desc *d, int kind, int y_dtype) {
    if (!emadapt_conv_bf16t_supported(d, kind, y_dtype) || d->B == 0) return 0;
    const long long M = (long long)d->B * d->H * d->W;
    const int N = kind == 1 ? d->Cin : d->Cout, Cx = kind == 1 ? d->Cout : d->Cin;
    const long long tiles = ceil_div(M, 256) * ceil_div(N, 256), K = (long long)d->KH * d->KW * Cx;
    const int cus = plan_cus();
    // (third class: few tiles with a very long reduction -- fc6's data gradient: 212 tiles x 2048 K steps = 1696 steps per CU as one
    // stream-K launch, 55.3 % against 50.8 % for the older kernel's six-way split-K; fourth: a single round of at least three quarters
    // of the CUs with K >= 4608 -- conv4_2 .. conv5_3 as plain tiles: 47 - 50 % against 44 - 45 %)
    // (round 6 tried a fifth class -- the same single round with 2304 <= K < 4608 on the new 224-row tiles: conv4_1's forward, 0.054 against
    // 0.062 ms in an isolated loop -- and measured a tie in the step, 10.11 / 10.15 against 10.12 / 10.17 ms: profiles/r06_m224_tiles.txt)
    return N >= 256 && (tiles >= 4LL * cus || (2 * tiles >= 3LL * cus && K >= 2304) || tiles * (K / 32) >= 1024LL * cus ||
                        (tiles <= cus && 4 * tiles >= 3LL * cus && K >= 4608)) ? 1 : 0;
}

size_t emadapt_filter_tiles_bf16_bytes(const emadapt_conv_desc *d, int dgrad) {
    if (check_desc(d) != EMADAPT_OK) return 0;
    const int rows = dgrad ? d->Cin : d->Cout, kdim = dgrad ? d->Cout : d->Cin;
    if (kdim % BT_BK != 0) return 0;
    return (size_t)d->KH * d->KW * (kdim / BT_BK) * (size_t)(ceil_div(rows, 256) * 256) * BT_ROWB;
}

int emadapt_filter_tiles_bf16(const emadapt_conv_desc *d, const float *w, void *tiles, int dgrad, void *stream_) {
    int rc = check_desc(d);
    if (rc) return rc;
    if (!w || !tiles || !aligned16(w) || !aligned16(tiles)) return EMADAPT_ERR_INVALID;
    const int rows = dgrad ? d->Cin : d->Cout, kdim = dgrad ? d->Cout : d->Cin;
    if (kdim % BT_BK != 0) return EMADAPT_ERR_UNSUPPORTED;
    const int rows_pad = (int)(ceil_div(rows, 256) * 256), taps = d->KH * d->KW;
    if (dgrad) {
        long long g = ((long long)taps * (kdim / BT_BK) * rows_pad * 4 + 255) / 256;
        if (g > 16384) g = 16384;
        hipLaunchKernelGGL(filter_tiles_b16_dgrad_kernel, dim3((unsigned)g), dim3(256), 0, as_stream(stream_), w, reinterpret_cast<uint4 *>(tiles),
                           d->Cin, d->Cout, taps, rows_pad);
    } else {
        const long long g = (long long)taps * (kdim / BT_BK) * (rows_pad / 64);
        if (g > 0x7fffffffLL) return EMADAPT_ERR_UNSUPPORTED;
        hipLaunchKernelGGL(filter_tiles_b16_fwd_kernel, dim3((unsigned)g), dim3(256), 0, as_stream(stream_), w, reinterpret_cast<uint4 *>(tiles),
                           d->Cin, d->Cout, rows_pad);
    }
    return launch_status();
}

int emadapt_filter_tiles_bf16_both(const emadapt_conv_desc *d, const float *w, void *tiles_fwd, void *tiles_dgrad, void *stream_) {
    int rc = check_desc(d);
    if (rc) return rc;
    if (!w || !tiles_fwd || !tiles_dgrad || !aligned16(w) || !aligned16(tiles_fwd) || !aligned16(tiles_dgrad)) return EMADAPT_ERR_INVALID;
    if (d->Cin % BT_BK != 0 || d->Cout % BT_BK != 0) return EMADAPT_ERR_UNSUPPORTED;
    if (d->Cin % 256 != 0 || d->Cout % 64 != 0) {   // (padding rows of the data-gradient image, or a ragged row block: the two single-image kernels)
        rc = emadapt_filter_tiles_bf16(d, w, tiles_fwd, 0, stream_);
        return rc ? rc : emadapt_filter_tiles_bf16(d, w, tiles_dgrad, 1, stream_);
    }
    const int rows_pad_f = (int)(ceil_div(d->Cout, 256) * 256), taps = d->KH * d->KW;
    if (rows_pad_f != d->Cout) {   // the forward image's padding rows (Cout % 256 != 0) are zeros: written by the single-image kernel only
        rc = emadapt_filter_tiles_bf16(d, w, tiles_fwd, 0, stream_);
        return rc ? rc : emadapt_filter_tiles_bf16(d, w, tiles_dgrad, 1, stream_);
    }
    const long long g = (long long)taps * (d->Cin / BT_BK) * (d->Cout / 64);
    if (g > 0x7fffffffLL) return EMADAPT_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(filter_tiles_b16_both_kernel, dim3((unsigned)g), dim3(256), 0, as_stream(stream_), w, reinterpret_cast<uint4 *>(tiles_fwd),
                       reinterpret_cast<uint4 *>(tiles_dgrad), d->Cin, d->Cout, rows_pad_f);
    return launch_status();
}

int emadapt_sgd_accumulate_momentum_tiles_supported(const emadapt_conv_desc *d) {
    if (check_desc(d)) return 0;
    const long long g = (long long)d->KH * d->KW * (d->Cin / BT_BK) * (d->Cout / 64) + ceil_div(d->Cout, 256);
    return d->Cin % 256 == 0 && d->Cout % 256 == 0 && g <= 0x7fffffffLL ? 1 : 0;
}

int emadapt_sgd_accumulate_momentum_tiles(const emadapt_conv_desc *d, float *w, float *v, const float *g, int64_t n_bias, float weight_decay,
                                          float gscale, float lr, float momentum, float ascale, void *tiles_fwd, void *tiles_dgrad,
                                          void *stream_) {
    int rc = check_desc(d);
    if (rc) return rc;
    if (!w || !v || !g || !tiles_fwd || !tiles_dgrad || n_bias < 0 || !aligned16(w) || !aligned16(v) || !aligned16(g) || !aligned16(tiles_fwd) ||
        !aligned16(tiles_dgrad))
        return EMADAPT_ERR_INVALID;
    if (!emadapt_sgd_accumulate_momentum_tiles_supported(d)) return EMADAPT_ERR_UNSUPPORTED;
    const int taps = d->KH * d->KW;
    const long long pieces = (long long)taps * (d->Cin / BT_BK) * (d->Cout / 64), nw = (long long)taps * d->Cin * d->Cout;
    const long long blocks = pieces + ceil_div((long long)n_bias, 256);
    if (blocks > 0x7fffffffLL) return EMADAPT_ERR_UNSUPPORTED;
    hipStream_t stream = as_stream(stream_);
    ProfScope prof(EMADAPT_K_OTHER, 0.0, (double)nw * 24.0 + (double)n_bias * 20.0, stream);
    hipLaunchKernelGGL(sgd_tiles_b16_both_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, w, v, g, reinterpret_cast<uint4 *>(tiles_fwd),
                       reinterpret_cast<uint4 *>(tiles_dgrad), d->Cin, d->Cout, d->Cout, (int)pieces, nw, (long long)n_bias, weight_decay, gscale, lr,
                       momentum, ascale, status_word());
    return launch_status();
}

int emadapt_sgd_momentum_payload_tiles(const emadapt_conv_desc *d, float *w, float *v, float *accum, const void *payload, int payload_dtype,
                                       int64_t n_bias, float lr, float momentum, float scale, void *tiles_fwd, void *tiles_dgrad, void *stream_) {
    int rc = check_desc(d);
    if (rc) return rc;
    if (!w || !v || !payload || !tiles_fwd || !tiles_dgrad || n_bias < 0 || (payload_dtype != EMADAPT_DT_F32 && payload_dtype != EMADAPT_DT_BF16))
        return EMADAPT_ERR_INVALID;
    if (!aligned16(w) || !aligned16(v) || !aligned16(accum) || !aligned16(tiles_fwd) || !aligned16(tiles_dgrad) ||
        reinterpret_cast<uintptr_t>(payload) % (payload_dtype == EMADAPT_DT_BF16 ? 8 : 16) != 0)
        return EMADAPT_ERR_INVALID;
    if (!emadapt_sgd_accumulate_momentum_tiles_supported(d)) return EMADAPT_ERR_UNSUPPORTED;
    const int taps = d->KH * d->KW;
    const long long pieces = (long long)taps * (d->Cin / BT_BK) * (d->Cout / 64), nw = (long long)taps * d->Cin * d->Cout;
    const long long blocks = pieces + ceil_div((long long)n_bias, 256);
    if (blocks > 0x7fffffffLL) return EMADAPT_ERR_UNSUPPORTED;
    hipStream_t stream = as_stream(stream_);
    const bool bf = payload_dtype == EMADAPT_DT_BF16;
    ProfScope prof(EMADAPT_K_OTHER, 0.0, (double)(nw + n_bias) * ((accum ? 24.0 : 20.0) + (bf ? 2.0 : 4.0)), stream);
    if (bf)
        hipLaunchKernelGGL(sgd_payload_tiles_b16_both_kernel<__bf16>, dim3((unsigned)blocks), dim3(256), 0, stream, w, v, accum,
                           static_cast<const __bf16 *>(payload), reinterpret_cast<uint4 *>(tiles_fwd), reinterpret_cast<uint4 *>(tiles_dgrad), d->Cin,
                           d->Cout, d->Cout, (int)pieces, nw, (long long)n_bias, lr, momentum, scale, status_word(), status_collective());
    else
        hipLaunchKernelGGL(sgd_payload_tiles_b16_both_kernel<float>, dim3((unsigned)blocks), dim3(256), 0, stream, w, v, accum,
                           static_cast<const float *>(payload), reinterpret_cast<uint4 *>(tiles_fwd), reinterpret_cast<uint4 *>(tiles_dgrad), d->Cin,
                           d->Cout, d->Cout, (int)pieces, nw, (long long)n_bias, lr, momentum, scale, status_word(), status_collective());
    return launch_status();
}

size_t emadapt_conv_bf16t_workspace_bytes(const emadapt_conv_desc *d, int kind) {
    if (kind < 0 || kind > 1 || !bt_ok(d, kind == 1, 0) || d->B == 0) return 0;
    IgemmArgs a;
    fill_b16t(a, d, kind == 1);
    return bt_workspace_bytes(a);
}

int emadapt_conv_bf16t_plan(const emadapt_conv_desc *d, int kind, int plan[4]) {
    if (!plan || kind < 0 || kind > 1) return EMADAPT_ERR_INVALID;
    if (!bt_ok(d, kind == 1, 0)) return EMADAPT_ERR_UNSUPPORTED;
    IgemmArgs a;
    fill_b16t(a, d, kind == 1);
    const PlanBT p = plan_b16t(a);
    plan[0] = p.main_rows; plan[1] = p.main_sk; plan[2] = p.main_levels; plan[3] = p.m224 ? 224 : p.tail_levels;
    return EMADAPT_OK;
}

int emadapt_conv_fwd_bf16t(const emadapt_conv_desc *d, const void *x_bf16, const void *w_tiles, const float *bias, void *y,
                           int y_dtype, int epilogue, float keep_prob, const uint8_t *drop_mask, uint64_t drop_seed,
                           void *workspace, size_t workspace_bytes, void *stream_) {
    int rc = check_desc(d);
    if (rc) return rc;
    if (!x_bf16 || !w_tiles || !y || (y_dtype & ~1)) return EMADAPT_ERR_INVALID;
    if (epilogue < EMADAPT_EPI_NONE || epilogue > EMADAPT_EPI_BIAS_RELU) return EMADAPT_ERR_INVALID;
    if (epilogue != EMADAPT_EPI_NONE && !bias) return EMADAPT_ERR_INVALID;
    if (!(keep_prob > 0.f) || keep_prob > 1.f) return EMADAPT_ERR_INVALID;
    if (keep_prob < 1.f && epilogue != EMADAPT_EPI_BIAS_RELU) return EMADAPT_ERR_INVALID;
    if (!aligned16(x_bf16) || !aligned16(w_tiles) || !aligned16(y) || !aligned16(workspace)) return EMADAPT_ERR_INVALID;
    if (!bt_ok(d, false, y_dtype)) return EMADAPT_ERR_UNSUPPORTED;
    if (d->B == 0) return EMADAPT_OK;
    IgemmArgs a;
    fill_b16t(a, d, false);
    a.x = static_cast<const float *>(x_bf16); a.w = static_cast<const float *>(w_tiles); a.y = static_cast<float *>(y);
    a.bias = bias; a.gate = nullptr; a.mask = drop_mask;
    a.epi = epilogue; a.scale = 1.f; set_keep(a, keep_prob); a.seed = drop_seed;
    hipStream_t stream = as_stream(stream_);
    ProfScope prof(EMADAPT_K_IGEMM_FWD, 2.0 * a.M * (double)a.K * a.N,
                   2.0 * ((double)a.M * a.Cx + (double)a.K * a.N) + (double)a.M * a.N * (y_dtype ? 2 : 4), stream);
    return y_dtype ? dispatch_b16t<true>(a, workspace, workspace_bytes, stream) : dispatch_b16t<false>(a, workspace, workspace_bytes, stream);
}

int emadapt_conv_dgrad_bf16t(const emadapt_conv_desc *d, const void *dy_bf16, const void *w_tiles, void *dx, int dx_dtype,
                             const void *gate, float gate_scale, void *workspace, size_t workspace_bytes, void *stream_) {
    int rc = check_desc(d);
    if (rc) return rc;
    if (!dy_bf16 || !w_tiles || !dx || (dx_dtype & ~1)) return EMADAPT_ERR_INVALID;
    if (!aligned16(dy_bf16) || !aligned16(w_tiles) || !aligned16(dx) || !aligned16(workspace)) return EMADAPT_ERR_INVALID;
    if (!bt_ok(d, true, dx_dtype)) return EMADAPT_ERR_UNSUPPORTED;
    if (d->B == 0) return EMADAPT_OK;
    IgemmArgs a;
    fill_b16t(a, d, true);
    a.x = static_cast<const float *>(dy_bf16); a.w = static_cast<const float *>(w_tiles); a.y = static_cast<float *>(dx);
    a.bias = nullptr; a.gate = static_cast<const float *>(gate); a.mask = nullptr;
    a.epi = EMADAPT_EPI_NONE; a.scale = gate_scale; set_keep(a, 1.f); a.seed = 0;
    hipStream_t stream = as_stream(stream_);
    ProfScope prof(EMADAPT_K_IGEMM_DGRAD, 2.0 * a.M * (double)a.K * a.N,
                   2.0 * ((double)a.M * a.Cx + (double)a.K * a.N) + (double)a.M * a.N * (dx_dtype ? 2 : 4) * (gate ? 2.0 : 1.0), stream);
    return dx_dtype ? dispatch_b16t<true>(a, workspace, workspace_bytes, stream) : dispatch_b16t<false>(a, workspace, workspace_bytes, stream);
}

}  // extern "C"

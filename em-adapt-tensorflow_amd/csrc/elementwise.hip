// HBM-bound kernels of the EM-Adapt step for gfx950: max-pool fwd/bwd, softmax cross-entropy
// fwd+bwd, bias gradient, L2 loss, gradient accumulation + momentum update, channel padding.
//
// All tensors are NHWC fp32 with C % 4 == 0, so every thread moves 16 bytes per access
// (global_load/store_dwordx4) and a wavefront covers 1 KiB of contiguous channels/pixels.
// Reductions are two-stage with a fixed summation order (no float atomics): results are
// bitwise reproducible run to run.
#include <float.h>
#include <math.h>

#include "common.h"

namespace {

using namespace emadapt;

__device__ __forceinline__ float4 ld4(const float *p) { return *reinterpret_cast<const float4 *>(p); }
__device__ __forceinline__ void st4(float *p, float4 v) { *reinterpret_cast<float4 *>(p) = v; }

// The same 4-channel accessors for bf16 storage (include/emadapt.h "Storage types"): 8-byte vectors, values widened
// to fp32 in registers and rounded to nearest-even on the way out.
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float4 ld4(const __bf16 *p) {
    const bf16x4 v = *reinterpret_cast<const bf16x4 *>(p);
    return make_float4((float)v[0], (float)v[1], (float)v[2], (float)v[3]);
}
__device__ __forceinline__ void st4(__bf16 *p, float4 v) {
    bf16x4 c;
    c[0] = (__bf16)v.x; c[1] = (__bf16)v.y; c[2] = (__bf16)v.z; c[3] = (__bf16)v.w;
    *reinterpret_cast<bf16x4 *>(p) = c;
}

// 16 bytes per thread for either storage type: VW<T> channels (4 f32 or 8 bf16), widened to fp32 in registers.
template <typename T> struct VW { static constexpr int N = 16 / (int)sizeof(T); };
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
__device__ __forceinline__ void ldv(const float *p, float (&v)[4]) {
    const float4 t = ld4(p);
    v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
}
__device__ __forceinline__ void stv(float *p, const float (&v)[4]) { st4(p, make_float4(v[0], v[1], v[2], v[3])); }
__device__ __forceinline__ void ldv(const __bf16 *p, float (&v)[8]) {
    const bf16x8 t = *reinterpret_cast<const bf16x8 *>(p);
#pragma unroll
    for (int e = 0; e < 8; e++) v[e] = (float)t[e];
}
__device__ __forceinline__ void stv(__bf16 *p, const float (&v)[8]) {
    bf16x8 t;
#pragma unroll
    for (int e = 0; e < 8; e++) t[e] = (__bf16)v[e];
    *reinterpret_cast<bf16x8 *>(p) = t;
}
template <int N> struct ByteVec;
template <> struct ByteVec<4> { typedef unsigned int type; };
template <> struct ByteVec<8> { typedef unsigned long long type; };

// ------------------------------------------------------------------------------------------
// tf.nn.max_pool 3x3, SAME (deeplab.py:77,82)
// ------------------------------------------------------------------------------------------
// I = index type: unsigned when the element count fits 31 bits (always, for DeepLab's maps), else long long -- the
// decomposition of the linear index costs four divisions by run-time divisors per thread, and 64-bit ones are ~10x the
// instructions of 32-bit ones on this hardware.
// GATED: the index byte also carries kPoolClosed when the winner is not > 0 -- the ReluGrad gate of the layer that produced x, which
// the backward kernels then apply without reading x again (the gate only matters at a pixel that wins a window, and there it is
// the window's maximum).
constexpr unsigned kPoolClosed = 0x10u;
template <typename T, typename I, bool GATED>
__global__ void __launch_bounds__(256) maxpool_fwd_kernel(const T *__restrict__ x, int B, int H, int W, int CV,
                                                        int stride, int OH, int OW, int pt, int pl,
                                                        T *__restrict__ y, unsigned char *__restrict__ idx) {
    constexpr int N = VW<T>::N;  // channels per thread; CV = C / N
    const I n = (I)B * OH * OW * CV;
    for (I t = (I)blockIdx.x * blockDim.x + threadIdx.x; t < n; t += (I)gridDim.x * blockDim.x) {
        const int cv = (int)(t % (I)CV);
        I r = t / (I)CV;
        const int ow = (int)(r % (I)OW);
        r /= (I)OW;
        const int oh = (int)(r % (I)OH);
        const int b = (int)(r / (I)OH);
        float best[N];
        int bi[N];
#pragma unroll
        for (int e = 0; e < N; e++) { best[e] = -FLT_MAX; bi[e] = -1; }
#pragma unroll
        for (int kh = 0; kh < 3; kh++) {
            const int h = oh * stride - pt + kh;
            if ((unsigned)h >= (unsigned)H) continue;
#pragma unroll
            for (int kw = 0; kw < 3; kw++) {
                const int w = ow * stride - pl + kw;
                if ((unsigned)w >= (unsigned)W) continue;
                float v[N];
                ldv(x + ((((long long)b * H + h) * W + w) * CV + cv) * N, v);
                const int tap = kh * 3 + kw;
#pragma unroll
                for (int e = 0; e < N; e++)
                    if (bi[e] < 0 || v[e] > best[e]) { best[e] = v[e]; bi[e] = tap; }  // first maximum wins
            }
        }
        stv(y + (long long)t * N, best);
        if (idx) {
            typename ByteVec<N>::type pk = 0;
#pragma unroll
            for (int e = 0; e < N; e++) {
                unsigned byte = (unsigned)bi[e] & 0xffu;
                if (GATED) byte |= best[e] > 0.f ? 0u : kPoolClosed;
                pk |= (typename ByteVec<N>::type)byte << (8 * e);
            }
            *reinterpret_cast<typename ByteVec<N>::type *>(idx + (long long)t * N) = pk;
        }
    }
}

// MaxPoolGrad (+ ReluGrad of the producing layer): gather form, one thread per input pixel x 4 channels.
template <typename T, typename I>
__global__ void __launch_bounds__(256) maxpool_bwd_kernel(const T *__restrict__ dy,
                                                        const unsigned char *__restrict__ idx,
                                                        const T *__restrict__ gate, int B, int H, int W, int CV,
                                                        int stride, int OH, int OW, int pt, int pl,
                                                        T *__restrict__ dx) {
    constexpr int N = VW<T>::N;
    const I n = (I)B * H * W * CV;
    for (I t = (I)blockIdx.x * blockDim.x + threadIdx.x; t < n; t += (I)gridDim.x * blockDim.x) {
        const int cv = (int)(t % (I)CV);
        I r = t / (I)CV;
        const int w = (int)(r % (I)W);
        r /= (I)W;
        const int h = (int)(r % (I)H);
        const int b = (int)(r / (I)H);
        float acc[N];
        unsigned won[N];   // index byte of a window this pixel won (they all carry the same gate bit: it is this pixel's)
#pragma unroll
        for (int e = 0; e < N; e++) { acc[e] = 0.f; won[e] = 0u; }
#pragma unroll
        for (int kh = 0; kh < 3; kh++) {
            const int th = h + pt - kh;
            if (th < 0 || th % stride != 0) continue;
            const int oh = th / stride;
            if (oh >= OH) continue;
#pragma unroll
            for (int kw = 0; kw < 3; kw++) {
                const int tw = w + pl - kw;
                if (tw < 0 || tw % stride != 0) continue;
                const int ow = tw / stride;
                if (ow >= OW) continue;
                const long long o = ((((long long)b * OH + oh) * OW + ow) * CV + cv) * N;
                const typename ByteVec<N>::type id = *reinterpret_cast<const typename ByteVec<N>::type *>(idx + o);
                float g[N];
                ldv(dy + o, g);
                const unsigned tap = kh * 3 + kw;
#pragma unroll
                for (int e = 0; e < N; e++) {
                    const unsigned byte = (unsigned)(id >> (8 * e)) & 0xffu;
                    if ((byte & 0xfu) == tap) { acc[e] += g[e]; won[e] = byte; }
                }
            }
        }
#pragma unroll
        for (int e = 0; e < N; e++) acc[e] = gate_keep_nan((won[e] & kPoolClosed) ? 0.f : 1.f, acc[e]);
        if (gate) {
            float a[N];
            ldv(gate + (long long)t * N, a);
#pragma unroll
            for (int e = 0; e < N; e++) acc[e] = gate_keep_nan(a[e], acc[e]);
        }
        stv(dx + (long long)t * N, acc);
    }
}

// The same gather, laid out by rows: one block walks (part of) one input row, so the image/row split of the index is
// scalar work and each thread divides once (by CV, a multiply-high).  The windows that cover an input pixel are enumerated
// directly -- for stride 2 the tap parity is fixed by the pixel's, which leaves at most 2 x 2 of the 9 taps, for stride 1
// all 3 x 3 -- in the same (kh, kw) order as above, so the sums round identically.
template <typename T, int STRIDE>
__global__ void __launch_bounds__(256) maxpool_bwd_rows_kernel(const T *__restrict__ dy,
                                                             const unsigned char *__restrict__ idx,
                                                             const T *__restrict__ gate, int H, int W, int CV, int OH,
                                                             int OW, int pt, int pl, unsigned cv_magic,
                                                             T *__restrict__ dx) {
    constexpr int N = VW<T>::N;
    constexpr int NC = STRIDE == 2 ? 2 : 3;   // candidate taps per dimension
    const int bh = blockIdx.x;
    const int b = bh / H, h = bh - b * H;
    int r_oh[NC], r_kh[NC];
#pragma unroll
    for (int a = 0; a < NC; a++) {
        const int kh = STRIDE == 2 ? ((h + pt) & 1) + 2 * a : a;
        const int th = h + pt - kh;
        const int oh = STRIDE == 2 ? th >> 1 : th;
        r_kh[a] = kh;
        r_oh[a] = (kh < 3 && th >= 0 && oh < OH) ? oh : -1;
    }
    const int row_n = W * CV;
    const long long row_base = (long long)bh * row_n;
    for (int t = blockIdx.y * 256 + threadIdx.x; t < row_n; t += gridDim.y * 256) {
        const int w = (int)div_magic((unsigned)t, cv_magic);
        const int cv = t - w * CV;
        float acc[N];
        unsigned won[N];
#pragma unroll
        for (int e = 0; e < N; e++) { acc[e] = 0.f; won[e] = 0u; }
#pragma unroll
        for (int a = 0; a < NC; a++) {
            if (r_oh[a] < 0) continue;   // uniform over the block
            const long long orow = ((long long)b * OH + r_oh[a]) * OW;
#pragma unroll
            for (int c = 0; c < NC; c++) {
                const int kw = STRIDE == 2 ? ((w + pl) & 1) + 2 * c : c;
                const int tw = w + pl - kw;
                const int ow = STRIDE == 2 ? tw >> 1 : tw;
                if (kw >= 3 || tw < 0 || ow >= OW) continue;
                const long long o = ((orow + ow) * CV + cv) * N;
                const typename ByteVec<N>::type id = *reinterpret_cast<const typename ByteVec<N>::type *>(idx + o);
                float g[N];
                ldv(dy + o, g);
                const unsigned tap = r_kh[a] * 3 + kw;
#pragma unroll
                for (int e = 0; e < N; e++) {
                    const unsigned byte = (unsigned)(id >> (8 * e)) & 0xffu;
                    if ((byte & 0xfu) == tap) { acc[e] += g[e]; won[e] = byte; }
                }
            }
        }
#pragma unroll
        for (int e = 0; e < N; e++) acc[e] = gate_keep_nan((won[e] & kPoolClosed) ? 0.f : 1.f, acc[e]);
        const long long xo = (row_base + t) * N;
        if (gate) {
            float g[N];
            ldv(gate + xo, g);
#pragma unroll
            for (int e = 0; e < N; e++) acc[e] = gate_keep_nan(g[e], acc[e]);
        }
        stv(dx + xo, acc);
    }
}

// Stride 2 by QUADS: one thread = the 2 x 2 input pixels (2a, 2a+1) x (2b, 2b+1) of the padded grid (hs = h + pt, ws = w + pl) x N
// channels.  The four pixels lie in the same four windows -- rows a (taps kh = hs - 2a: 0, 1) and a - 1 (tap 2), columns likewise --
// so a thread loads (index, dy) of 4 windows once and serves the 9 (pixel, window) pairs they form: the row form above loads and
// widens 4 / 2 / 2 / 1 windows for the same four pixels and runs both column parities through every wave (the kernel is bound by
// its vector ALU, not by memory: ~200 instructions per 16 bytes written there, ~130 here).  Every pixel adds its windows in the row
// form's order (kh, then kw, ascending), so the sums round identically.
template <typename T>
__global__ void __launch_bounds__(256) maxpool_bwd_quad_kernel(const T *__restrict__ dy, const unsigned char *__restrict__ idx,
                                                             const T *__restrict__ gate, int H, int W, int CV, int OH, int OW,
                                                             int pt, int pl, int NA, int NB, unsigned cv_magic, T *__restrict__ dx) {
    constexpr int N = VW<T>::N;
    typedef typename ByteVec<N>::type IdT;
    const int ba = blockIdx.x;
    const int b = ba / NA, a = ba - b * NA;
    const int h0 = 2 * a - pt;   // the quad's pixel rows: h0, h0 + 1
    const bool hok[2] = {h0 >= 0 && h0 < H, h0 + 1 >= 0 && h0 + 1 < H};
    const int oh_q[2] = {a, a - 1};
    const bool ohok[2] = {a < OH, a - 1 >= 0 && a - 1 < OH};
    const int row_n = NB * CV;
    for (int t = blockIdx.y * 256 + threadIdx.x; t < row_n; t += gridDim.y * 256) {
        const int bq = (int)div_magic((unsigned)t, cv_magic);
        const int cv = t - bq * CV;
        const int w0 = 2 * bq - pl;
        const bool wok[2] = {w0 >= 0 && w0 < W, w0 + 1 >= 0 && w0 + 1 < W};
        const bool owok[2] = {bq < OW, bq - 1 >= 0 && bq - 1 < OW};
        // window q = 2 * (row: 0 = a, 1 = a - 1) + (column: 0 = bq, 1 = bq - 1); a missing window matches no tap (low nibble 15)
        float g[4][N];
        IdT id[4];
#pragma unroll
        for (int q = 0; q < 4; q++) {
            id[q] = ~(IdT)0;
#pragma unroll
            for (int e = 0; e < N; e++) g[q][e] = 0.f;
            if (ohok[q >> 1] && owok[q & 1]) {
                const long long o = ((((long long)b * OH + oh_q[q >> 1]) * OW + (bq - (q & 1))) * CV + cv) * N;
                id[q] = *reinterpret_cast<const IdT *>(idx + o);
                ldv(dy + o, g[q]);
            }
        }
        float acc[4][N];     // pixel p = 2 * (row h0 + (p >> 1)) + (column w0 + (p & 1))
        unsigned won[4][N];
#pragma unroll
        for (int p = 0; p < 4; p++)
#pragma unroll
            for (int e = 0; e < N; e++) { acc[p][e] = 0.f; won[p][e] = 0u; }
#define EMADAPT_POOL_PAIR(P, Q, TAP)                                                  \
        _Pragma("unroll") for (int e = 0; e < N; e++) {                                \
            const unsigned byte = (unsigned)(id[Q] >> (8 * e)) & 0xffu;                \
            if ((byte & 0xfu) == (TAP)) { acc[P][e] += g[Q][e]; won[P][e] = byte; }    \
        }
        EMADAPT_POOL_PAIR(0, 0, 0u) EMADAPT_POOL_PAIR(0, 1, 2u) EMADAPT_POOL_PAIR(0, 2, 6u) EMADAPT_POOL_PAIR(0, 3, 8u)
        EMADAPT_POOL_PAIR(1, 0, 1u) EMADAPT_POOL_PAIR(1, 2, 7u)
        EMADAPT_POOL_PAIR(2, 0, 3u) EMADAPT_POOL_PAIR(2, 1, 5u)
        EMADAPT_POOL_PAIR(3, 0, 4u)
#undef EMADAPT_POOL_PAIR
#pragma unroll
        for (int p = 0; p < 4; p++) {
            if (!hok[p >> 1] || !wok[p & 1]) continue;
            const long long xo = ((((long long)b * H + (h0 + (p >> 1))) * W + (w0 + (p & 1))) * CV + cv) * N;
#pragma unroll
            for (int e = 0; e < N; e++) acc[p][e] = gate_keep_nan((won[p][e] & kPoolClosed) ? 0.f : 1.f, acc[p][e]);
            if (gate) {
                float gt[N];
                ldv(gate + xo, gt);
#pragma unroll
                for (int e = 0; e < N; e++) acc[p][e] = gate_keep_nan(gt[e], acc[p][e]);
            }
            stv(dx + xo, acc[p]);
        }
    }
}

// ------------------------------------------------------------------------------------------
// generic deterministic reductions
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ float block_sum_256(float v, float *sh) {
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
    __syncthreads();
    const float r = (sh[0] + sh[1]) + (sh[2] + sh[3]);
    __syncthreads();
    return r;
}

// out[0] = scale * sum(partials[0..n)) in index order chunks (single block).
__global__ void __launch_bounds__(256) final_sum_kernel(const float *__restrict__ partials, int n, float scale,
                                                      float *__restrict__ out) {
    __shared__ float sh[4];
    float v = 0.f;
    for (int i = threadIdx.x; i < n; i += 256) v += partials[i];
    v = block_sum_256(v, sh);
    if (threadIdx.x == 0) out[0] = v * scale;
}

// BiasAddGrad stage 1: partial[chunk][n] = sum over the chunk's rows of dy[m][n].
// block = CP column-vectors (16 bytes: 4 f32 or 8 bf16 channels) x (256/CP) row lanes, CP = power of two >=
// min(N/VW, 64): narrow layers (N = 64, 24) spend their idle column lanes on more rows.
template <typename T>
__global__ void __launch_bounds__(256) bias_grad_partial_kernel(const T *__restrict__ dy, long long M, int NV,
                                                              int cp_log2, long long rows_per_chunk,
                                                              float *__restrict__ partial) {
    constexpr int N = VW<T>::N;
    const int CP = 1 << cp_log2, RL = 256 >> cp_log2;
    const int tx = threadIdx.x & (CP - 1), ty = threadIdx.x >> cp_log2;
    const int q = blockIdx.x * CP + tx;
    const long long r0 = (long long)blockIdx.y * rows_per_chunk;
    long long r1 = r0 + rows_per_chunk;
    if (r1 > M) r1 = M;
    float acc[N];
#pragma unroll
    for (int e = 0; e < N; e++) acc[e] = 0.f;
    if (q < NV) {
        long long r = r0 + ty;
        for (; r + 3 * RL < r1; r += 4 * RL) {   // four independent 16-byte loads in flight per lane
            float v[4][N];
#pragma unroll
            for (int u = 0; u < 4; u++) ldv(dy + ((r + u * RL) * NV + q) * N, v[u]);
#pragma unroll
            for (int e = 0; e < N; e++) acc[e] += (v[0][e] + v[1][e]) + (v[2][e] + v[3][e]);
        }
        for (; r < r1; r += RL) {
            float v[N];
            ldv(dy + (r * NV + q) * N, v);
#pragma unroll
            for (int e = 0; e < N; e++) acc[e] += v[e];
        }
    }
    __shared__ float sh[256][N + 1];
#pragma unroll
    for (int e = 0; e < N; e++) sh[ty * CP + tx][e] = acc[e];
    __syncthreads();
    if (ty == 0 && q < NV) {
#pragma unroll
        for (int e = 0; e < N; e++) {
            float sum = sh[tx][e];
            for (int j = 1; j < RL; j++) sum += sh[j * CP + tx][e];
            partial[((long long)blockIdx.y * NV + q) * N + e] = sum;
        }
    }
}

// stage 2 (also used by the conv filter-gradient): out[i] = sum_s slab[s][i].
// block = 64 float4 columns x 4 slab lanes; lane g adds slabs g, g+4, g+8, .. (4 independent loads in flight per
// step), then the four lane sums are combined as ((l0+l1)+(l2+l3)): a fixed tree -> bitwise reproducible.
// Two independent segments in one launch (a filter gradient's slabs + the bias-gradient partial rows of the same launch,
// conv_bf16.hip): float4 index i < n4a belongs to segment a, the rest to segment b (its own slab count and row length n4b).
struct SlabSumSeg {
    const float *slab;
    float *out;
    long long n4;
    int nslab;
};

__global__ void __launch_bounds__(256) slab_sum_kernel(SlabSumSeg sa, SlabSumSeg sb) {
    __shared__ float4 sh[4][64];
    const int tx = threadIdx.x & 63, g = threadIdx.x >> 6;
    const long long n4 = sa.n4 + sb.n4;
    for (long long i0 = (long long)blockIdx.x * 64; i0 < n4; i0 += (long long)gridDim.x * 64) {
        const long long ig = i0 + tx;
        const bool second = ig >= sa.n4;
        const float *slab = second ? sb.slab : sa.slab;
        const long long sn4 = second ? sb.n4 : sa.n4, i = second ? ig - sa.n4 : ig;
        const int nslab = second ? sb.nslab : sa.nslab;
        float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
        if (ig < n4) {
            int k = g;
            for (; k + 12 < nslab; k += 16) {
                const float4 v0 = ld4(slab + ((long long)k * sn4 + i) * 4);
                const float4 v1 = ld4(slab + ((long long)(k + 4) * sn4 + i) * 4);
                const float4 v2 = ld4(slab + ((long long)(k + 8) * sn4 + i) * 4);
                const float4 v3 = ld4(slab + ((long long)(k + 12) * sn4 + i) * 4);
                s.x += v0.x; s.y += v0.y; s.z += v0.z; s.w += v0.w;
                s.x += v1.x; s.y += v1.y; s.z += v1.z; s.w += v1.w;
                s.x += v2.x; s.y += v2.y; s.z += v2.z; s.w += v2.w;
                s.x += v3.x; s.y += v3.y; s.z += v3.z; s.w += v3.w;
            }
            for (; k < nslab; k += 4) {
                const float4 v = ld4(slab + ((long long)k * sn4 + i) * 4);
                s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
            }
        }
        sh[g][tx] = s;
        __syncthreads();
        if (g == 0 && ig < n4) {
            const float4 a = sh[0][tx], b = sh[1][tx], c = sh[2][tx], d = sh[3][tx];
            float4 r;
            r.x = (a.x + b.x) + (c.x + d.x);
            r.y = (a.y + b.y) + (c.y + d.y);
            r.z = (a.z + b.z) + (c.z + d.z);
            r.w = (a.w + b.w) + (c.w + d.w);
            st4((second ? sb.out : sa.out) + i * 4, r);
        }
        __syncthreads();
    }
}

// ------------------------------------------------------------------------------------------
// sparse softmax cross-entropy + gradient (deeplab.py:182); one thread per pixel row
// ------------------------------------------------------------------------------------------
template <typename TD>
__global__ void __launch_bounds__(256) softmax_ce_kernel(const float *__restrict__ logits, long long ldl,
                                                       const long long *__restrict__ labels, long long M, int C,
                                                       float grad_scale, TD *__restrict__ dlogits,
                                                       float *__restrict__ partial) {
    __shared__ float sh[4];
    const long long r = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    float loss = 0.f;
    if (r < M) {
        const float *p = logits + r * ldl;
        float mx = p[0];
        for (int c = 1; c < C; c++) mx = fmaxf(mx, p[c]);
        float s = 0.f;
        for (int c = 0; c < C; c++) s += expf(p[c] - mx);
        const long long lab = labels[r];
        const float lse = logf(s);
        if (lab >= 0 && lab < C) loss = lse - (p[lab] - mx);
        if (dlogits) {
            TD *d = dlogits + r * ldl;
            const float inv = 1.0f / s;
            for (int c = 0; c < C; c++) {
                const float sm = expf(p[c] - mx) * inv;
                d[c] = (TD)((sm - (c == lab ? 1.f : 0.f)) * grad_scale);
            }
            for (int c = C; c < ldl; c++) d[c] = (TD)0.f;
        }
    }
    loss = block_sum_256(loss, sh);
    if (threadIdx.x == 0 && partial) partial[blockIdx.x] = loss;
}

__global__ void __launch_bounds__(256) l2_partial_kernel(const float *__restrict__ x, long long n,
                                                       float *__restrict__ partial) {
    __shared__ float sh[4];
    float v = 0.f;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n;
         i += (long long)gridDim.x * blockDim.x)
        v += x[i] * x[i];
    v = block_sum_256(v, sh);
    if (threadIdx.x == 0) partial[blockIdx.x] = v;
}

// ------------------------------------------------------------------------------------------
// optimiser (deeplab.py:202-205)
// ------------------------------------------------------------------------------------------
// (the pinned arithmetic -- mul_rn, opt_decay / opt_accum / opt_velocity / opt_step -- lives in common.h: conv_bf16.hip's fused update shares it)

__global__ void __launch_bounds__(256) sgd_accumulate_kernel(float *__restrict__ accum, const float *__restrict__ g,
                                                           const float *__restrict__ w, long long n,
                                                           long long n_decay, float wd, float scale, const unsigned *__restrict__ poison) {
    const bool bad = status_raised(poison);   // (collective mode only: common.h -- the failure travels with the gradient)
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n;
         i += (long long)gridDim.x * blockDim.x) {
        float gi = g[i];
        if (i < n_decay) gi = opt_decay(gi, wd, w[i]);
        accum[i] = bad ? poison_nan() : opt_accum(accum[i], gi, scale);
    }
}

__global__ void __launch_bounds__(256) sgd_accumulate_kernel_v4(float *__restrict__ accum,
                                                              const float *__restrict__ g,
                                                              const float *__restrict__ w, long long n4,
                                                              long long n_decay4, float wd, float scale, const unsigned *__restrict__ poison) {
    const bool bad = status_raised(poison);
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n4;
         i += (long long)gridDim.x * blockDim.x) {
        float4 gi = ld4(g + i * 4);
        if (i < n_decay4) {
            const float4 wi = ld4(w + i * 4);
            gi.x = opt_decay(gi.x, wd, wi.x); gi.y = opt_decay(gi.y, wd, wi.y); gi.z = opt_decay(gi.z, wd, wi.z); gi.w = opt_decay(gi.w, wd, wi.w);
        }
        float4 a = ld4(accum + i * 4);
        a.x = opt_accum(a.x, gi.x, scale); a.y = opt_accum(a.y, gi.y, scale); a.z = opt_accum(a.z, gi.z, scale); a.w = opt_accum(a.w, gi.w, scale);
        if (bad) a = make_float4(poison_nan(), poison_nan(), poison_nan(), poison_nan());
        st4(accum + i * 4, a);
    }
}

// (status_raised: common.h)
__global__ void __launch_bounds__(256) sgd_momentum_kernel(float *__restrict__ w, float *__restrict__ v,
                                                         float *__restrict__ accum, long long n, float lr, float mu,
                                                         float ascale, unsigned *__restrict__ status, int coll) {
    if (!coll && status_raised(status)) return;
    bool bad = false;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n;
         i += (long long)gridDim.x * blockDim.x) {
        float vi = v[i], wi = w[i];
        opt_update(wi, vi, accum[i], lr, mu, ascale, coll, bad);
        v[i] = vi;
        w[i] = wi;
        accum[i] = 0.f;
    }
    if (bad) status_raise(status, EMADAPT_STATUS_NAN_GRADIENT);
}

__global__ void __launch_bounds__(256) sgd_momentum_kernel_v4(float *__restrict__ w, float *__restrict__ v,
                                                            float *__restrict__ accum, long long n4, float lr,
                                                            float mu, float ascale, unsigned *__restrict__ status, int coll) {
    if (!coll && status_raised(status)) return;
    bool bad = false;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n4;
         i += (long long)gridDim.x * blockDim.x) {
        const float4 a = ld4(accum + i * 4);
        float4 vi = ld4(v + i * 4);
        float4 wi = ld4(w + i * 4);
        opt_update(wi.x, vi.x, a.x, lr, mu, ascale, coll, bad); opt_update(wi.y, vi.y, a.y, lr, mu, ascale, coll, bad);
        opt_update(wi.z, vi.z, a.z, lr, mu, ascale, coll, bad); opt_update(wi.w, vi.w, a.w, lr, mu, ascale, coll, bad);
        st4(v + i * 4, vi);
        st4(w + i * 4, wi);
        st4(accum + i * 4, make_float4(0.f, 0.f, 0.f, 0.f));
    }
    if (bad) status_raise(status, EMADAPT_STATUS_NAN_GRADIENT);
}

// The same update reading the SUMMED GRADIENT from the gathered payload of the data-parallel exchange (f32 or bf16, dist.py)
// instead of the accumulator -- the exchanged values never travel back into the accumulator, which is only zeroed (the clean op,
// deeplab.py:203).  VN elements per thread: 4 (f32 payload) or 8 (bf16 payload: 16 bytes).
template <typename P>
__global__ void __launch_bounds__(256) sgd_momentum_payload_kernel(float *__restrict__ w, float *__restrict__ v,
                                                                 float *__restrict__ accum, const P *__restrict__ payload,
                                                                 long long nv, float lr, float mu, float ascale,
                                                                 unsigned *__restrict__ status, int coll) {
    constexpr int VN = VW<P>::N;
    if (!coll && status_raised(status)) return;
    bool bad = false;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < nv; i += (long long)gridDim.x * blockDim.x) {
        float g[VN];
        ldv(payload + i * VN, g);
#pragma unroll
        for (int q = 0; q < VN / 4; q++) {
            const long long o = i * VN + q * 4;
            float4 vi = ld4(v + o);
            float4 wi = ld4(w + o);
            opt_update(wi.x, vi.x, g[q * 4 + 0], lr, mu, ascale, coll, bad); opt_update(wi.y, vi.y, g[q * 4 + 1], lr, mu, ascale, coll, bad);
            opt_update(wi.z, vi.z, g[q * 4 + 2], lr, mu, ascale, coll, bad); opt_update(wi.w, vi.w, g[q * 4 + 3], lr, mu, ascale, coll, bad);
            st4(v + o, vi);
            st4(w + o, wi);
            if (accum) st4(accum + o, make_float4(0.f, 0.f, 0.f, 0.f));   // (null: the exchange was fed without an accumulator, sgd_accumulate_pack)
        }
    }
    if (bad) status_raise(status, EMADAPT_STATUS_NAN_GRADIENT);
}
template <typename P>
__global__ void __launch_bounds__(256) sgd_momentum_payload_tail_kernel(float *__restrict__ w, float *__restrict__ v,
                                                                      float *__restrict__ accum, const P *__restrict__ payload,
                                                                      long long i0, long long n, float lr, float mu, float ascale,
                                                                      unsigned *__restrict__ status, int coll) {
    if (!coll && status_raised(status)) return;
    bool bad = false;
    for (long long i = i0 + (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        float vi = v[i], wi = w[i];
        opt_update(wi, vi, (float)payload[i], lr, mu, ascale, coll, bad);
        v[i] = vi;
        w[i] = wi;
        if (accum) accum[i] = 0.f;
    }
    if (bad) status_raise(status, EMADAPT_STATUS_NAN_GRADIENT);
}

// accumulate (deeplab.py:202) + apply_gradients (:204-205) + clean (:203) of ONE micro-batch in one pass, for accum_num == 1
// without a gradient exchange: the accumulator would only carry (g + wd w) * gscale from one kernel to the next.  The arithmetic
// is the two kernels' (0 + (g + wd w) gscale, then mu v + that * ascale), operation for operation: the weights come out bit for bit
// as from emadapt_sgd_accumulate followed by emadapt_sgd_momentum on a zero accumulator.  20 bytes per parameter instead of 40.
__global__ void __launch_bounds__(256) sgd_accumulate_momentum_kernel_v4(float *__restrict__ w, float *__restrict__ v,
                                                                       const float *__restrict__ g, long long n4, long long n_decay4,
                                                                       float wd, float gscale, float lr, float mu, float ascale,
                                                                       const unsigned *__restrict__ status) {
    if (status_raised(status)) return;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long long)gridDim.x * blockDim.x) {
        float4 gi = ld4(g + i * 4);
        float4 wi = ld4(w + i * 4);
        float4 vi = ld4(v + i * 4);
        if (i < n_decay4) { gi.x = opt_decay(gi.x, wd, wi.x); gi.y = opt_decay(gi.y, wd, wi.y); gi.z = opt_decay(gi.z, wd, wi.z); gi.w = opt_decay(gi.w, wd, wi.w); }
        const float4 a = make_float4(opt_accum(0.f, gi.x, gscale), opt_accum(0.f, gi.y, gscale), opt_accum(0.f, gi.z, gscale), opt_accum(0.f, gi.w, gscale));
        vi.x = opt_velocity(mu, vi.x, a.x, ascale); vi.y = opt_velocity(mu, vi.y, a.y, ascale);
        vi.z = opt_velocity(mu, vi.z, a.z, ascale); vi.w = opt_velocity(mu, vi.w, a.w, ascale);
        wi.x = opt_step(wi.x, lr, vi.x); wi.y = opt_step(wi.y, lr, vi.y); wi.z = opt_step(wi.z, lr, vi.z); wi.w = opt_step(wi.w, lr, vi.w);
        st4(v + i * 4, vi);
        st4(w + i * 4, wi);
    }
}
// The same pass over SEVERAL layers' segments of the flat buffers in one launch (the small layers' launches are all ramp and tail:
// 8 of them cost 35 us for 1.7 M parameters).  first4[k] = the segment's first float4 in the launch's index space (first4[count] =
// the total), at4[k] = where it starts in w / v / g, decay4[k] = its decayed float4s.
constexpr int kSgdMaxSegs = 16;
struct SgdSegs {
    long long first4[kSgdMaxSegs + 1], at4[kSgdMaxSegs], decay4[kSgdMaxSegs];
    int count;
};
__global__ void __launch_bounds__(256) sgd_accumulate_momentum_multi_kernel(float *__restrict__ w, float *__restrict__ v,
                                                                          const float *__restrict__ g, SgdSegs sg, float wd, float gscale,
                                                                          float lr, float mu, float ascale,
                                                                          const unsigned *__restrict__ status) {
    if (status_raised(status)) return;
    const long long n4 = sg.first4[sg.count];
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long long)gridDim.x * blockDim.x) {
        int k = 0;
#pragma unroll
        for (int q = 1; q < kSgdMaxSegs; q++) k += (q < sg.count && i >= sg.first4[q]) ? 1 : 0;
        const long long r = i - sg.first4[k], o = (sg.at4[k] + r) * 4;
        float4 gi = ld4(g + o);
        float4 wi = ld4(w + o);
        float4 vi = ld4(v + o);
        if (r < sg.decay4[k]) { gi.x = opt_decay(gi.x, wd, wi.x); gi.y = opt_decay(gi.y, wd, wi.y); gi.z = opt_decay(gi.z, wd, wi.z); gi.w = opt_decay(gi.w, wd, wi.w); }
        const float4 a = make_float4(opt_accum(0.f, gi.x, gscale), opt_accum(0.f, gi.y, gscale), opt_accum(0.f, gi.z, gscale), opt_accum(0.f, gi.w, gscale));
        vi.x = opt_velocity(mu, vi.x, a.x, ascale); vi.y = opt_velocity(mu, vi.y, a.y, ascale);
        vi.z = opt_velocity(mu, vi.z, a.z, ascale); vi.w = opt_velocity(mu, vi.w, a.w, ascale);
        wi.x = opt_step(wi.x, lr, vi.x); wi.y = opt_step(wi.y, lr, vi.y); wi.z = opt_step(wi.z, lr, vi.z); wi.w = opt_step(wi.w, lr, vi.w);
        st4(v + o, vi);
        st4(w + o, wi);
    }
}
__global__ void __launch_bounds__(256) sgd_accumulate_momentum_kernel(float *__restrict__ w, float *__restrict__ v,
                                                                    const float *__restrict__ g, long long n, long long n_decay,
                                                                    float wd, float gscale, float lr, float mu, float ascale,
                                                                    const unsigned *__restrict__ status) {
    if (status_raised(status)) return;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        float gi = g[i];
        const float wi = w[i];
        if (i < n_decay) gi = opt_decay(gi, wd, wi);
        const float a = opt_accum(0.f, gi, gscale);
        const float vi = opt_velocity(mu, v[i], a, ascale);
        v[i] = vi;
        w[i] = opt_step(wi, lr, vi);
    }
}

// ------------------------------------------------------------------------------------------
// accumulate + pack for accum_num == 1 in a data-parallel job: payload[i] = P(0 + (g + wd w) scale) -- what emadapt_sgd_accumulate would leave
// in a zeroed accumulator and emadapt_grad_pack would then round, operation for operation, without the accumulator's write, read and
// clean (12 of the 44 bytes the exchange path moved per parameter) and without the pack pass.
template <typename P>
__global__ void __launch_bounds__(256) sgd_accumulate_pack_kernel(const float *__restrict__ g, const float *__restrict__ w, long long nv,
                                                                long long n_decay, float wd, float scale, P *__restrict__ dst,
                                                                const unsigned *__restrict__ poison) {
    constexpr int VN = VW<P>::N;
    const bool bad = status_raised(poison);
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < nv; i += (long long)gridDim.x * blockDim.x) {
        float x[VN];
#pragma unroll
        for (int q = 0; q < VN / 4; q++) {
            const long long o = i * VN + q * 4;
            float4 gi = ld4(g + o);
            if (o < n_decay) {   // (n_decay % 4 == 0: checked by the host)
                const float4 wi = ld4(w + o);
                gi.x = opt_decay(gi.x, wd, wi.x); gi.y = opt_decay(gi.y, wd, wi.y); gi.z = opt_decay(gi.z, wd, wi.z); gi.w = opt_decay(gi.w, wd, wi.w);
            }
            x[q * 4 + 0] = opt_accum(0.f, gi.x, scale); x[q * 4 + 1] = opt_accum(0.f, gi.y, scale);
            x[q * 4 + 2] = opt_accum(0.f, gi.z, scale); x[q * 4 + 3] = opt_accum(0.f, gi.w, scale);
        }
        if (bad) {
#pragma unroll
            for (int e = 0; e < VN; e++) x[e] = poison_nan();
        }
        stv(dst + i * VN, x);
    }
}
template <typename P>
__global__ void __launch_bounds__(256) sgd_accumulate_pack_scalar_kernel(const float *__restrict__ g, const float *__restrict__ w, long long i0,
                                                                       long long n, long long n_decay, float wd, float scale, P *__restrict__ dst,
                                                                       const unsigned *__restrict__ poison) {
    const bool bad = status_raised(poison);
    for (long long i = i0 + (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        float gi = g[i];
        if (i < n_decay) gi = opt_decay(gi, wd, w[i]);
        dst[i] = (P)(bad ? poison_nan() : opt_accum(0.f, gi, scale));
    }
}

// Staging of the data-parallel gradient exchange (dist.py, algo "direct": reduce-scatter + all-gather as two all-to-all steps).
// The exchange maps the reference's micro-batches to ranks (deeplab.py:202: accum += g / accum_num); these kernels are the only
// arithmetic on that path and replace round 4's torch copy_/add_ sequences (6 + W launches and ~8 passes per bucket) by one pass each.
// ------------------------------------------------------------------------------------------
// pack: send[0:n] = P(accum[0:n]) (bf16: round to nearest even, once), send[n:npad] = 0 (whole shards for every rank)
template <typename P>
__global__ void __launch_bounds__(256) grad_pack_kernel(const float *__restrict__ src, long long n, P *__restrict__ dst, long long npad,
                                                        const unsigned *__restrict__ poison) {
    constexpr int VN = VW<P>::N;
    const bool bad = status_raised(poison);
    const long long nv = npad / VN;   // (npad % VN == 0 and both pointers 16-byte aligned: checked by the host, else the scalar kernel)
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < nv; i += (long long)gridDim.x * blockDim.x) {
        float x[VN];
#pragma unroll
        for (int q = 0; q < VN / 4; q++) {
            const long long o = i * VN + q * 4;
            float4 t = make_float4(0.f, 0.f, 0.f, 0.f);
            if (o + 4 <= n) t = ld4(src + o);
            else if (o < n) { t.x = src[o]; if (o + 1 < n) t.y = src[o + 1]; if (o + 2 < n) t.z = src[o + 2]; }
            x[q * 4 + 0] = t.x; x[q * 4 + 1] = t.y; x[q * 4 + 2] = t.z; x[q * 4 + 3] = t.w;
        }
        if (bad) {
#pragma unroll
            for (int e = 0; e < VN; e++) x[e] = poison_nan();   // (the padding too: nobody applies it)
        }
        stv(dst + i * VN, x);
    }
}
template <typename P>
__global__ void __launch_bounds__(256) grad_pack_scalar_kernel(const float *__restrict__ src, long long n, P *__restrict__ dst, long long npad,
                                                               const unsigned *__restrict__ poison) {
    const bool bad = status_raised(poison);
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < npad; i += (long long)gridDim.x * blockDim.x)
        dst[i] = (P)(bad ? poison_nan() : (i < n ? src[i] : 0.f));
}
// owner sum: out[i] = P( fp32(recv[0][i]) + fp32(recv[1][i]) + ... + fp32(recv[W-1][i]) ), the adds in fp32 in RANK ORDER (every
// element is summed exactly once, by its owner: all ranks end up with the same bits), the sum rounded once.  recv = [W][shard].
template <typename P>
__global__ void __launch_bounds__(256) grad_owner_sum_kernel(const P *__restrict__ recv, int W, long long shard, P *__restrict__ out) {
    constexpr int VN = VW<P>::N;
    const long long nv = shard / VN;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < nv; i += (long long)gridDim.x * blockDim.x) {
        float acc[VN];
        ldv(recv + i * VN, acc);
        for (int r = 1; r < W; r++) {
            float x[VN];
            ldv(recv + (long long)r * shard + i * VN, x);
#pragma unroll
            for (int e = 0; e < VN; e++) acc[e] = acc[e] + x[e];
        }
        stv(out + i * VN, acc);
    }
}
template <typename P>
__global__ void __launch_bounds__(256) grad_owner_sum_scalar_kernel(const P *__restrict__ recv, int W, long long shard, P *__restrict__ out) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < shard; i += (long long)gridDim.x * blockDim.x) {
        float acc = (float)recv[i];
        for (int r = 1; r < W; r++) acc = acc + (float)recv[(long long)r * shard + i];
        out[i] = (P)acc;
    }
}

__global__ void __launch_bounds__(256) copy_pad_kernel(const float *__restrict__ src, long long rows, int Cs,
                                                     float *__restrict__ dst, int Cd) {
    const long long n = rows * Cd;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n;
         i += (long long)gridDim.x * blockDim.x) {
        const long long r = i / Cd;
        const int c = (int)(i % Cd);
        dst[i] = c < Cs ? src[r * Cs + c] : 0.f;
    }
}

inline int grid_for(long long n, int block = 256, int cap = 2048) {
    long long g = ceil_div(n, block);
    if (g < 1) g = 1;
    if (g > cap) g = cap;
    return (int)g;
}

inline bool aligned16(const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

void same_pad(int in, int stride, int *out, int *pad_before) {
    *out = (in + stride - 1) / stride;
    int total = (*out - 1) * stride + 3 - in;
    if (total < 0) total = 0;
    *pad_before = total / 2;
}

constexpr int kSumPartials = 1024;  // stage-1 blocks of the scalar reductions

}  // namespace

namespace emadapt {
// used by conv_wgrad.hip
int launch_slab_sum(const float *slab, int nslab, long long n, float *out, hipStream_t stream) {
    if (n % 4 != 0) return EMADAPT_ERR_INVALID;
    const SlabSumSeg sa = {slab, out, n / 4, nslab}, sb = {nullptr, nullptr, 0, 0};
    hipLaunchKernelGGL(slab_sum_kernel, dim3(grid_for(n / 4, 64, 4096)), dim3(256), 0, stream, sa, sb);
    return launch_status();
}
// two independent sums in one launch: out_a[i] = sum over nslab_a slabs of n_a floats, out_b likewise (either may be empty: n = 0)
int launch_slab_sum2(const float *slab_a, int nslab_a, long long n_a, float *out_a, const float *slab_b, int nslab_b, long long n_b,
                     float *out_b, hipStream_t stream) {
    if (n_a % 4 != 0 || n_b % 4 != 0) return EMADAPT_ERR_INVALID;
    if (n_a + n_b == 0) return EMADAPT_OK;
    const SlabSumSeg sa = {slab_a, out_a, n_a / 4, nslab_a}, sb = {slab_b, out_b, n_b / 4, nslab_b};
    hipLaunchKernelGGL(slab_sum_kernel, dim3(grid_for((n_a + n_b) / 4, 64, 4096)), dim3(256), 0, stream, sa, sb);
    return launch_status();
}
}  // namespace emadapt

namespace {

template <typename T>
int maxpool_fwd_t(const T *x, int B, int H, int W, int C, int stride, T *y, uint8_t *idx, bool gated, void *stream_) {
    if (!x || !y || B < 0 || H <= 0 || W <= 0 || C <= 0) return EMADAPT_ERR_INVALID;
    constexpr int N = VW<T>::N;
    if (C % N != 0 || (stride != 1 && stride != 2)) return EMADAPT_ERR_UNSUPPORTED;
    if (!aligned16(x) || !aligned16(y)) return EMADAPT_ERR_INVALID;
    if (B == 0) return EMADAPT_OK;
    int OH, OW, pt, pl;
    same_pad(H, stride, &OH, &pt);
    same_pad(W, stride, &OW, &pl);
    const long long n = (long long)B * OH * OW * (C / N);
    hipStream_t stream = as_stream(stream_);
    ProfScope prof(EMADAPT_K_OTHER, 0.0, (double)B * H * W * C * sizeof(T) + (double)n * N * (sizeof(T) + 1.0), stream);
    if (gated && !idx) return EMADAPT_ERR_INVALID;
    const dim3 grid(grid_for(n, 256, 16384));
    const bool narrow = n < (1LL << 31) - (1LL << 23) && (long long)B * H * W * (C / N) < (1LL << 31);
    if (narrow && gated)
        hipLaunchKernelGGL((maxpool_fwd_kernel<T, unsigned, true>), grid, dim3(256), 0, stream, x, B, H, W, C / N, stride, OH, OW, pt, pl, y, idx);
    else if (narrow)
        hipLaunchKernelGGL((maxpool_fwd_kernel<T, unsigned, false>), grid, dim3(256), 0, stream, x, B, H, W, C / N, stride, OH, OW, pt, pl, y, idx);
    else if (gated)
        hipLaunchKernelGGL((maxpool_fwd_kernel<T, long long, true>), grid, dim3(256), 0, stream, x, B, H, W, C / N, stride, OH, OW, pt, pl, y, idx);
    else
        hipLaunchKernelGGL((maxpool_fwd_kernel<T, long long, false>), grid, dim3(256), 0, stream, x, B, H, W, C / N, stride, OH, OW, pt, pl, y, idx);
    return launch_status();
}

template <typename T>
int maxpool_bwd_t(const T *dy, const uint8_t *idx, const T *gate, int B, int H, int W, int C, int stride, T *dx,
                  void *stream_) {
    constexpr int N = VW<T>::N;
    if (!dy || !idx || !dx || B < 0 || H <= 0 || W <= 0 || C <= 0) return EMADAPT_ERR_INVALID;
    if (C % N != 0 || (stride != 1 && stride != 2)) return EMADAPT_ERR_UNSUPPORTED;
    if (B == 0) return EMADAPT_OK;
    int OH, OW, pt, pl;
    same_pad(H, stride, &OH, &pt);
    same_pad(W, stride, &OW, &pl);
    const long long n = (long long)B * H * W * (C / N);
    hipStream_t stream = as_stream(stream_);
    ProfScope prof(EMADAPT_K_OTHER, 0.0, (double)n * N * sizeof(T) * (gate ? 2.0 : 1.0) + (double)B * OH * OW * C * (sizeof(T) + 1.0), stream);
    const long long rows = (long long)B * H, row_n = (long long)W * (C / N);
    if (rows < (1LL << 31) && row_n * (C / N) < (1LL << 32) && row_n < (1LL << 30)) {   // (div_magic is exact for t * CV < 2^32)
        long long chunks = ceil_div(row_n, 256);
        const long long want = ceil_div(16384, rows);          // ~16 K blocks; rows of wide maps are walked in a loop
        if (chunks > want) chunks = want;
        const dim3 grid((unsigned)rows, (unsigned)chunks);
        if (stride == 2 && tune(TUNE_POOL_ROWS) == 0) {   // by quads (tune "pool_rows" = 1: the row form, the A/B)
            const int NA = (H + pt + 1) >> 1, NB = (W + pl + 1) >> 1;
            const long long qrows = (long long)B * NA, qrow_n = (long long)NB * (C / N);
            long long qchunks = ceil_div(qrow_n, 256);
            const long long qwant = ceil_div(16384, qrows);
            if (qchunks > qwant) qchunks = qwant;
            hipLaunchKernelGGL((maxpool_bwd_quad_kernel<T>), dim3((unsigned)qrows, (unsigned)qchunks), dim3(256), 0, stream, dy, idx, gate, H, W,
                               C / N, OH, OW, pt, pl, NA, NB, magic_u32(C / N), dx);
        } else if (stride == 2)
            hipLaunchKernelGGL((maxpool_bwd_rows_kernel<T, 2>), grid, dim3(256), 0, stream, dy, idx, gate, H, W, C / N, OH, OW,
                               pt, pl, magic_u32(C / N), dx);
        else
            hipLaunchKernelGGL((maxpool_bwd_rows_kernel<T, 1>), grid, dim3(256), 0, stream, dy, idx, gate, H, W, C / N, OH, OW,
                               pt, pl, magic_u32(C / N), dx);
    } else {   // very wide channel counts: the generic index arithmetic
        hipLaunchKernelGGL((maxpool_bwd_kernel<T, long long>), dim3(grid_for(n, 256, 16384)), dim3(256), 0, stream, dy, idx,
                           gate, B, H, W, C / N, stride, OH, OW, pt, pl, dx);
    }
    return launch_status();
}

}  // namespace

extern "C" {

int emadapt_maxpool_fwd(const float *x, int B, int H, int W, int C, int stride, float *y, uint8_t *idx,
                        void *stream_) {
    return maxpool_fwd_t<float>(x, B, H, W, C, stride, y, idx, false, stream_);
}

int emadapt_maxpool_bwd(const float *dy, const uint8_t *idx, const float *gate, int B, int H, int W, int C,
                        int stride, float *dx, void *stream_) {
    return maxpool_bwd_t<float>(dy, idx, gate, B, H, W, C, stride, dx, stream_);
}

int emadapt_maxpool_fwd_dt(const void *x, int dtype, int B, int H, int W, int C, int stride, void *y, uint8_t *idx,
                           void *stream_) {
    return emadapt_maxpool_fwd_gated(x, dtype, B, H, W, C, stride, y, idx, 0, stream_);
}

int emadapt_maxpool_fwd_gated(const void *x, int dtype, int B, int H, int W, int C, int stride, void *y, uint8_t *idx, int relu_gate,
                              void *stream_) {
    if (dtype == EMADAPT_DT_F32)
        return maxpool_fwd_t<float>(static_cast<const float *>(x), B, H, W, C, stride, static_cast<float *>(y), idx, relu_gate != 0, stream_);
    if (dtype == EMADAPT_DT_BF16)
        return maxpool_fwd_t<__bf16>(static_cast<const __bf16 *>(x), B, H, W, C, stride, static_cast<__bf16 *>(y), idx, relu_gate != 0, stream_);
    return EMADAPT_ERR_INVALID;
}

int emadapt_maxpool_bwd_dt(const void *dy, const uint8_t *idx, const void *gate, int dtype, int B, int H, int W, int C,
                           int stride, void *dx, void *stream_) {
    if (dtype == EMADAPT_DT_F32)
        return maxpool_bwd_t<float>(static_cast<const float *>(dy), idx, static_cast<const float *>(gate), B, H, W, C, stride,
                                    static_cast<float *>(dx), stream_);
    if (dtype == EMADAPT_DT_BF16)
        return maxpool_bwd_t<__bf16>(static_cast<const __bf16 *>(dy), idx, static_cast<const __bf16 *>(gate), B, H, W, C,
                                     stride, static_cast<__bf16 *>(dx), stream_);
    return EMADAPT_ERR_INVALID;
}

static long long bias_grad_max_chunks(int N) {
    long long c = 65536 / (N > 0 ? N : 1);  // narrow layers get more row chunks (same workspace)
    if (c < 256) c = 256;
    if (c > 1024) c = 1024;
    return c;
}

size_t emadapt_bias_grad_workspace_bytes(int64_t M, int N) {
    (void)M;
    return (size_t)bias_grad_max_chunks(N) * (size_t)(N > 0 ? N : 0) * sizeof(float) + 256;
}

}  // extern "C"

template <typename T>
static int bias_grad_t(const T *dy, int64_t M, int N, float *db, void *workspace, size_t workspace_bytes,
                       void *stream_) {
    constexpr int NC = VW<T>::N;
    if (!dy || !db || M <= 0 || N <= 0) return EMADAPT_ERR_INVALID;
    if (N % NC != 0) return EMADAPT_ERR_UNSUPPORTED;
    if (!workspace || workspace_bytes < emadapt_bias_grad_workspace_bytes(M, N)) return EMADAPT_ERR_WORKSPACE;
    const int N4 = N / NC;  // 16-byte column vectors per row
    int cp_log2 = 0;
    while ((1 << cp_log2) < N4 && cp_log2 < 6) cp_log2++;
    const int CP = 1 << cp_log2;
    const int gx = (int)ceil_div(N4, CP);
    long long chunks = 2048 / gx;
    if (chunks < 1) chunks = 1;
    if (chunks > bias_grad_max_chunks(N)) chunks = bias_grad_max_chunks(N);
    long long rows_per_chunk = ceil_div(M, chunks);
    const int RL = 256 / CP;
    if (rows_per_chunk < 4 * RL) rows_per_chunk = 4 * RL;
    chunks = ceil_div(M, rows_per_chunk);
    hipStream_t stream = as_stream(stream_);
    ProfScope prof(EMADAPT_K_OTHER, 0.0, (double)M * N * sizeof(T), stream);
    float *partial = static_cast<float *>(workspace);
    hipLaunchKernelGGL(bias_grad_partial_kernel<T>, dim3(gx, (unsigned)chunks), dim3(256), 0, stream, dy, (long long)M,
                       N4, cp_log2, rows_per_chunk, partial);
    int rc = launch_status();
    if (rc) return rc;
    return launch_slab_sum(partial, (int)chunks, N, db, stream);
}

extern "C" {

int emadapt_bias_grad(const float *dy, int64_t M, int N, float *db, void *workspace, size_t workspace_bytes,
                      void *stream_) {
    return bias_grad_t<float>(dy, M, N, db, workspace, workspace_bytes, stream_);
}

int emadapt_bias_grad_dt(const void *dy, int dtype, int64_t M, int N, float *db, void *workspace,
                         size_t workspace_bytes, void *stream_) {
    if (dtype == EMADAPT_DT_F32) return bias_grad_t<float>(static_cast<const float *>(dy), M, N, db, workspace, workspace_bytes, stream_);
    if (dtype == EMADAPT_DT_BF16) return bias_grad_t<__bf16>(static_cast<const __bf16 *>(dy), M, N, db, workspace, workspace_bytes, stream_);
    return EMADAPT_ERR_INVALID;
}

size_t emadapt_softmax_ce_workspace_bytes(int64_t M) { return (size_t)(ceil_div(M > 0 ? M : 1, 256)) * 4 + 256; }

}  // extern "C"

template <typename TD>
static int softmax_ce_t(const float *logits, int64_t ldl, const int64_t *labels, int64_t M, int C, float grad_scale,
                        TD *dlogits, float *loss_sum, void *workspace, size_t workspace_bytes, void *stream_) {
    if (!logits || !labels || M <= 0 || C <= 0 || ldl < C) return EMADAPT_ERR_INVALID;
    if (loss_sum && (!workspace || workspace_bytes < emadapt_softmax_ce_workspace_bytes(M)))
        return EMADAPT_ERR_WORKSPACE;
    const int grid = (int)ceil_div(M, 256);
    hipStream_t stream = as_stream(stream_);
    ProfScope prof(EMADAPT_K_OTHER, 0.0, (double)M * (ldl * 8.0 + 8.0), stream);
    float *partial = loss_sum ? static_cast<float *>(workspace) : nullptr;
    hipLaunchKernelGGL(softmax_ce_kernel<TD>, dim3(grid), dim3(256), 0, stream, logits, (long long)ldl,
                       reinterpret_cast<const long long *>(labels), (long long)M, C, grad_scale, dlogits, partial);
    int rc = launch_status();
    if (rc || !loss_sum) return rc;
    hipLaunchKernelGGL(final_sum_kernel, dim3(1), dim3(256), 0, stream, partial, grid, 1.0f, loss_sum);
    return launch_status();
}

extern "C" {

int emadapt_softmax_ce(const float *logits, int64_t ldl, const int64_t *labels, int64_t M, int C, float grad_scale,
                       float *dlogits, float *loss_sum, void *workspace, size_t workspace_bytes, void *stream_) {
    return softmax_ce_t<float>(logits, ldl, labels, M, C, grad_scale, dlogits, loss_sum, workspace, workspace_bytes, stream_);
}

int emadapt_softmax_ce_dt(const float *logits, int64_t ldl, const int64_t *labels, int64_t M, int C, float grad_scale,
                          void *dlogits, int dlogits_dtype, float *loss_sum, void *workspace, size_t workspace_bytes,
                          void *stream_) {
    if (dlogits_dtype == EMADAPT_DT_F32)
        return softmax_ce_t<float>(logits, ldl, labels, M, C, grad_scale, static_cast<float *>(dlogits), loss_sum, workspace, workspace_bytes, stream_);
    if (dlogits_dtype == EMADAPT_DT_BF16)
        return softmax_ce_t<__bf16>(logits, ldl, labels, M, C, grad_scale, static_cast<__bf16 *>(dlogits), loss_sum, workspace, workspace_bytes, stream_);
    return EMADAPT_ERR_INVALID;
}

size_t emadapt_l2_loss_workspace_bytes(int64_t n) {
    (void)n;
    return kSumPartials * sizeof(float);
}

int emadapt_l2_loss(const float *x, int64_t n, float *out, void *workspace, size_t workspace_bytes, void *stream_) {
    if (!x || !out || n < 0) return EMADAPT_ERR_INVALID;
    if (!workspace || workspace_bytes < emadapt_l2_loss_workspace_bytes(n)) return EMADAPT_ERR_WORKSPACE;
    hipStream_t stream = as_stream(stream_);
    const int grid = grid_for(n, 256, kSumPartials);
    float *partial = static_cast<float *>(workspace);
    ProfScope prof(EMADAPT_K_OTHER, 0.0, (double)n * 4.0, stream);
    hipLaunchKernelGGL(l2_partial_kernel, dim3(grid), dim3(256), 0, stream, x, (long long)n, partial);
    int rc = launch_status();
    if (rc) return rc;
    hipLaunchKernelGGL(final_sum_kernel, dim3(1), dim3(256), 0, stream, partial, grid, 0.5f, out);
    return launch_status();
}

int emadapt_sgd_accumulate(float *accum, const float *g, const float *w, int64_t n, int64_t n_decay,
                           float weight_decay, float scale, void *stream_) {
    if (!accum || !g || n < 0 || n_decay < 0 || n_decay > n) return EMADAPT_ERR_INVALID;
    if (n_decay > 0 && !w) return EMADAPT_ERR_INVALID;
    if (n == 0) return EMADAPT_OK;
    hipStream_t stream = as_stream(stream_);
    ProfScope prof(EMADAPT_K_OTHER, 0.0, (double)n * 12.0 + (double)n_decay * 4.0, stream);
    const unsigned *poison = status_collective() ? status_word() : nullptr;   // (collective mode: a raised word turns the output into NaNs, common.h)
    if (n % 4 == 0 && n_decay % 4 == 0 && aligned16(accum) && aligned16(g) && (!w || aligned16(w))) {
        hipLaunchKernelGGL(sgd_accumulate_kernel_v4, dim3(grid_for(n / 4, 256, 4096)), dim3(256), 0, stream, accum, g,
                           w, (long long)(n / 4), (long long)(n_decay / 4), weight_decay, scale, poison);
    } else {
        hipLaunchKernelGGL(sgd_accumulate_kernel, dim3(grid_for(n, 256, 4096)), dim3(256), 0, stream, accum, g, w,
                           (long long)n, (long long)n_decay, weight_decay, scale, poison);
    }
    return launch_status();
}

int emadapt_sgd_accumulate_pack(const float *g, const float *w, int64_t n, int64_t n_decay, float weight_decay, float scale,
                                void *dst, int dst_dtype, void *stream_) {
    if (!g || !dst || n < 0 || n_decay < 0 || n_decay > n || (dst_dtype != EMADAPT_DT_F32 && dst_dtype != EMADAPT_DT_BF16)) return EMADAPT_ERR_INVALID;
    if (n_decay > 0 && !w) return EMADAPT_ERR_INVALID;
    if (n == 0) return EMADAPT_OK;
    hipStream_t stream = as_stream(stream_);
    const bool bf = dst_dtype == EMADAPT_DT_BF16;
    ProfScope prof(EMADAPT_K_OTHER, 0.0, (double)n * (4.0 + (bf ? 2.0 : 4.0)) + (double)n_decay * 4.0, stream);
    const unsigned *poison = status_collective() ? status_word() : nullptr;
    const int vn = bf ? 8 : 4;
    const bool vec = aligned16(g) && aligned16(dst) && (!w || aligned16(w)) && n_decay % 4 == 0;
    const int64_t nvec = vec ? n / vn : 0;
    if (nvec > 0) {
        if (bf) hipLaunchKernelGGL(sgd_accumulate_pack_kernel<__bf16>, dim3(grid_for(nvec, 256, 4096)), dim3(256), 0, stream, g, w, (long long)nvec,
                                   (long long)n_decay, weight_decay, scale, static_cast<__bf16 *>(dst), poison);
        else hipLaunchKernelGGL(sgd_accumulate_pack_kernel<float>, dim3(grid_for(nvec, 256, 4096)), dim3(256), 0, stream, g, w, (long long)nvec,
                                (long long)n_decay, weight_decay, scale, static_cast<float *>(dst), poison);
    }
    if (nvec * vn < n) {
        if (bf) hipLaunchKernelGGL(sgd_accumulate_pack_scalar_kernel<__bf16>, dim3(grid_for(n - nvec * vn, 256, 4096)), dim3(256), 0, stream, g, w,
                                   (long long)(nvec * vn), (long long)n, (long long)n_decay, weight_decay, scale, static_cast<__bf16 *>(dst), poison);
        else hipLaunchKernelGGL(sgd_accumulate_pack_scalar_kernel<float>, dim3(grid_for(n - nvec * vn, 256, 4096)), dim3(256), 0, stream, g, w,
                                (long long)(nvec * vn), (long long)n, (long long)n_decay, weight_decay, scale, static_cast<float *>(dst), poison);
    }
    return launch_status();
}

int emadapt_sgd_momentum(float *w, float *v, float *accum, int64_t n, float lr, float momentum, float accum_scale,
                         void *stream_) {
    if (!w || !v || !accum || n < 0) return EMADAPT_ERR_INVALID;
    if (n == 0) return EMADAPT_OK;
    hipStream_t stream = as_stream(stream_);
    unsigned *status = status_word();
    const int coll = status_collective();
    ProfScope prof(EMADAPT_K_OTHER, 0.0, (double)n * 24.0, stream);
    if (n % 4 == 0 && aligned16(w) && aligned16(v) && aligned16(accum)) {
        hipLaunchKernelGGL(sgd_momentum_kernel_v4, dim3(grid_for(n / 4, 256, 4096)), dim3(256), 0, stream, w, v, accum,
                           (long long)(n / 4), lr, momentum, accum_scale, status, coll);
    } else {
        hipLaunchKernelGGL(sgd_momentum_kernel, dim3(grid_for(n, 256, 4096)), dim3(256), 0, stream, w, v, accum,
                           (long long)n, lr, momentum, accum_scale, status, coll);
    }
    return launch_status();
}

int emadapt_sgd_momentum_payload(float *w, float *v, float *accum, const void *payload, int payload_dtype, int64_t n, float lr,
                                 float momentum, float scale, void *stream_) {
    if (!w || !v || !payload || n < 0 || (payload_dtype != EMADAPT_DT_F32 && payload_dtype != EMADAPT_DT_BF16))   // (accum may be NULL: nothing to clean)
        return EMADAPT_ERR_INVALID;
    if (n == 0) return EMADAPT_OK;
    hipStream_t stream = as_stream(stream_);
    unsigned *status = status_word();
    const int coll = status_collective();
    const bool bf = payload_dtype == EMADAPT_DT_BF16;
    ProfScope prof(EMADAPT_K_OTHER, 0.0, (double)n * ((accum ? 20.0 : 16.0) + (bf ? 2.0 : 4.0)), stream);
    const int vn = bf ? 8 : 4;
    const bool vec = aligned16(w) && aligned16(v) && (!accum || aligned16(accum)) && aligned16(payload);
    const int64_t nvec = vec ? n / vn : 0;
    if (nvec > 0) {
        if (bf)
            hipLaunchKernelGGL(sgd_momentum_payload_kernel<__bf16>, dim3(grid_for(nvec, 256, 4096)), dim3(256), 0, stream, w, v, accum,
                               static_cast<const __bf16 *>(payload), (long long)nvec, lr, momentum, scale, status, coll);
        else
            hipLaunchKernelGGL(sgd_momentum_payload_kernel<float>, dim3(grid_for(nvec, 256, 4096)), dim3(256), 0, stream, w, v, accum,
                               static_cast<const float *>(payload), (long long)nvec, lr, momentum, scale, status, coll);
        int rc = launch_status();
        if (rc) return rc;
    }
    if (nvec * vn < n) {
        if (bf)
            hipLaunchKernelGGL(sgd_momentum_payload_tail_kernel<__bf16>, dim3(grid_for(n - nvec * vn, 256, 4096)), dim3(256), 0, stream, w, v,
                               accum, static_cast<const __bf16 *>(payload), (long long)(nvec * vn), (long long)n, lr, momentum, scale, status, coll);
        else
            hipLaunchKernelGGL(sgd_momentum_payload_tail_kernel<float>, dim3(grid_for(n - nvec * vn, 256, 4096)), dim3(256), 0, stream, w, v,
                               accum, static_cast<const float *>(payload), (long long)(nvec * vn), (long long)n, lr, momentum, scale, status, coll);
        return launch_status();
    }
    return EMADAPT_OK;
}

int emadapt_sgd_accumulate_momentum(float *w, float *v, const float *g, int64_t n, int64_t n_decay, float weight_decay,
                                    float gscale, float lr, float momentum, float ascale, void *stream_) {
    if (!w || !v || !g || n < 0 || n_decay < 0 || n_decay > n) return EMADAPT_ERR_INVALID;
    if (n == 0) return EMADAPT_OK;
    hipStream_t stream = as_stream(stream_);
    const unsigned *status = status_word();
    ProfScope prof(EMADAPT_K_OTHER, 0.0, (double)n * 20.0, stream);
    if (n % 4 == 0 && n_decay % 4 == 0 && aligned16(w) && aligned16(v) && aligned16(g)) {
        hipLaunchKernelGGL(sgd_accumulate_momentum_kernel_v4, dim3(grid_for(n / 4, 256, 4096)), dim3(256), 0, stream, w, v, g,
                           (long long)(n / 4), (long long)(n_decay / 4), weight_decay, gscale, lr, momentum, ascale, status);
    } else {
        hipLaunchKernelGGL(sgd_accumulate_momentum_kernel, dim3(grid_for(n, 256, 4096)), dim3(256), 0, stream, w, v, g, (long long)n,
                           (long long)n_decay, weight_decay, gscale, lr, momentum, ascale, status);
    }
    return launch_status();
}

int emadapt_sgd_accumulate_momentum_multi(float *w, float *v, const float *g, int nseg, const int64_t *start, const int64_t *n,
                                          const int64_t *n_decay, float weight_decay, float gscale, float lr, float momentum, float ascale,
                                          void *stream_) {
    if (!w || !v || !g || nseg < 0 || nseg > kSgdMaxSegs || (nseg && (!start || !n || !n_decay))) return EMADAPT_ERR_INVALID;
    if (!aligned16(w) || !aligned16(v) || !aligned16(g)) return EMADAPT_ERR_INVALID;
    SgdSegs sg;
    long long tot = 0;
    sg.count = 0;
    for (int k = 0; k < nseg; k++) {
        if (start[k] < 0 || n[k] < 0 || n_decay[k] < 0 || n_decay[k] > n[k]) return EMADAPT_ERR_INVALID;
        if (start[k] % 4 != 0 || n[k] % 4 != 0 || n_decay[k] % 4 != 0) return EMADAPT_ERR_UNSUPPORTED;   // (whole float4s: the caller takes such a segment to the one-segment entry)
        for (int j = 0; j < k; j++)
            if (start[k] < start[j] + n[j] && start[j] < start[k] + n[k]) return EMADAPT_ERR_INVALID;   // overlapping segments
        if (n[k] == 0) continue;
        sg.first4[sg.count] = tot; sg.at4[sg.count] = start[k] / 4; sg.decay4[sg.count] = n_decay[k] / 4;
        tot += n[k] / 4;
        sg.count++;
    }
    for (int k = sg.count; k <= kSgdMaxSegs; k++) sg.first4[k] = tot;
    for (int k = sg.count; k < kSgdMaxSegs; k++) { sg.at4[k] = 0; sg.decay4[k] = 0; }
    if (tot == 0) return EMADAPT_OK;
    hipStream_t stream = as_stream(stream_);
    ProfScope prof(EMADAPT_K_OTHER, 0.0, (double)tot * 80.0, stream);
    hipLaunchKernelGGL(sgd_accumulate_momentum_multi_kernel, dim3(grid_for(tot, 256, 4096)), dim3(256), 0, stream, w, v, g, sg, weight_decay, gscale, lr,
                       momentum, ascale, status_word());
    return launch_status();
}

int emadapt_grad_pack(const float *src, int64_t n, void *dst, int dst_dtype, int64_t npad, void *stream_) {
    if (!src || !dst || n < 0 || npad < n || (dst_dtype != EMADAPT_DT_F32 && dst_dtype != EMADAPT_DT_BF16)) return EMADAPT_ERR_INVALID;
    if (npad == 0) return EMADAPT_OK;
    hipStream_t stream = as_stream(stream_);
    const bool bf = dst_dtype == EMADAPT_DT_BF16;
    ProfScope prof(EMADAPT_K_OTHER, 0.0, (double)n * 4.0 + (double)npad * (bf ? 2.0 : 4.0), stream);
    const unsigned *poison = status_collective() ? status_word() : nullptr;
    const int vn = bf ? 8 : 4;
    const bool vec = npad % vn == 0 && aligned16(src) && aligned16(dst);
    if (bf) {
        __bf16 *d = static_cast<__bf16 *>(dst);
        if (vec) hipLaunchKernelGGL(grad_pack_kernel<__bf16>, dim3(grid_for(npad / vn, 256, 4096)), dim3(256), 0, stream, src, (long long)n, d, (long long)npad, poison);
        else hipLaunchKernelGGL(grad_pack_scalar_kernel<__bf16>, dim3(grid_for(npad, 256, 4096)), dim3(256), 0, stream, src, (long long)n, d, (long long)npad, poison);
    } else {
        float *d = static_cast<float *>(dst);
        if (vec) hipLaunchKernelGGL(grad_pack_kernel<float>, dim3(grid_for(npad / vn, 256, 4096)), dim3(256), 0, stream, src, (long long)n, d, (long long)npad, poison);
        else hipLaunchKernelGGL(grad_pack_scalar_kernel<float>, dim3(grid_for(npad, 256, 4096)), dim3(256), 0, stream, src, (long long)n, d, (long long)npad, poison);
    }
    return launch_status();
}

int emadapt_grad_owner_sum(const void *recv, int dtype, int world, int64_t shard, void *out, void *stream_) {
    if (!recv || !out || world < 1 || shard < 0 || (dtype != EMADAPT_DT_F32 && dtype != EMADAPT_DT_BF16)) return EMADAPT_ERR_INVALID;
    if (shard == 0) return EMADAPT_OK;
    hipStream_t stream = as_stream(stream_);
    const bool bf = dtype == EMADAPT_DT_BF16;
    ProfScope prof(EMADAPT_K_OTHER, 0.0, (double)shard * (world + 1) * (bf ? 2.0 : 4.0), stream);
    const int vn = bf ? 8 : 4;
    // (every rank's slice of recv must start on a 16-byte boundary for the vector form: shard % vn == 0)
    const bool vec = shard % vn == 0 && aligned16(recv) && aligned16(out);
    if (bf) {
        const __bf16 *r = static_cast<const __bf16 *>(recv);
        __bf16 *o = static_cast<__bf16 *>(out);
        if (vec) hipLaunchKernelGGL(grad_owner_sum_kernel<__bf16>, dim3(grid_for(shard / vn, 256, 4096)), dim3(256), 0, stream, r, world, (long long)shard, o);
        else hipLaunchKernelGGL(grad_owner_sum_scalar_kernel<__bf16>, dim3(grid_for(shard, 256, 4096)), dim3(256), 0, stream, r, world, (long long)shard, o);
    } else {
        const float *r = static_cast<const float *>(recv);
        float *o = static_cast<float *>(out);
        if (vec) hipLaunchKernelGGL(grad_owner_sum_kernel<float>, dim3(grid_for(shard / vn, 256, 4096)), dim3(256), 0, stream, r, world, (long long)shard, o);
        else hipLaunchKernelGGL(grad_owner_sum_scalar_kernel<float>, dim3(grid_for(shard, 256, 4096)), dim3(256), 0, stream, r, world, (long long)shard, o);
    }
    return launch_status();
}

int emadapt_copy_pad_channels(const float *src, int64_t rows, int Csrc, float *dst, int Cdst, void *stream_) {
    if (!src || !dst || rows < 0 || Csrc <= 0 || Cdst <= 0) return EMADAPT_ERR_INVALID;
    if (rows == 0) return EMADAPT_OK;
    hipStream_t stream = as_stream(stream_);
    ProfScope prof(EMADAPT_K_OTHER, 0.0, (double)rows * (Csrc + Cdst) * 4.0, stream);
    hipLaunchKernelGGL(copy_pad_kernel, dim3(grid_for(rows * Cdst, 256, 8192)), dim3(256), 0, stream, src,
                       (long long)rows, Csrc, dst, Cdst);
    return launch_status();
}

}  // extern "C"

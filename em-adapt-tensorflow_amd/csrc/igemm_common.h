// Shared pieces of the implicit-GEMM convolution kernels (fp32: conv_igemm.hip, bf16: conv_bf16.hip):
// launch arguments, bounds-checked loads, the fused epilogue, the split-K second stage and its cost model.
#pragma once
#include "common.h"

namespace emadapt {
int conv_pad_before(int k, int dil);
}

namespace {

using namespace emadapt;

typedef float f32x16 __attribute__((ext_vector_type(16)));

// Ablation switches (emadapt_tune_set("ablate", bits)) exist only in diagnostic builds (-DEMADAPT_ABLATE_BUILD,
// `python -m em_adapt_tensorflow_amd._build --ablate`): in the shipped kernel they would put scalar branches into
// every k-pair of the MFMA stream.
#ifdef EMADAPT_ABLATE_BUILD
#define ABL(bit) (a.ablate & (bit))
#else
#define ABL(bit) 0
#endif

// Row packing of the stream-K launches that skip padding (igemm_v2.h).  Filter row kh reads input row oh + dh(kh), filter
// column kw input column ow + dw(kw); for the output pixels near the edges of an image some filter rows / columns lie wholly
// outside it and their part of the reduction multiplies zeros.  The output rows that see the same set of filter rows form a
// band [oh0, oh0 + noh) (a 4x4 rate-4 filter on a 41-row map: [0,2) [2,6) [6,35) [35,39) [39,41)), likewise the columns;
// a CLASS is a row band x a column band, and its TAP BOX the filter rows x columns that reach into the image from its pixels.
// The GEMM's rows are the pixels class after class (image-major inside a class) -- every pixel exactly once, so there are as
// many M tiles as in the dense launch -- and a tile's K loop runs over the UNION of the tap boxes of the classes it holds rows
// of (a tile inside one class: that class's box; a tile across a class boundary multiplies the padding of the smaller boxes).
// Round 2 padded every class to whole tiles instead: 86.1 % of the dense products for fc6 at batch 16 against 82.3 % packed
// (80.7 % exact), and no gain at all below batch 12 (25 classes x a partial tile each: the reference's own batch is 6).
//   classes: GEMM rows [row0[c], row0[c+1]) are the pixels of class c; rowmap_kernel turns that into the table
//            rowmap[GEMM row] = pixel (in the workspace), which the loaders, epilogues and the fix-up pass read
//   runs:    consecutive M tiles with the same tap box; run r covers tiles [run_tile0[r], run_tile0[r+1]) and stream-K
//            units [run_u0[r], run_u0[r+1]) (units = K steps of all tiles, tile after tile, N tiles fastest)
// n = 0: the rows are the pixels in memory order, nothing is skipped.
constexpr int kMaxRowClasses = 36;
constexpr int kMaxTileRuns = 112;
struct RowPack {
    int n;                 // classes (0 = off)
    int nruns;
    const int *rowmap;     // [tiles_m * 256]; -1 = row past the last pixel
    int row0[kMaxRowClasses + 1];
    unsigned short oh0[kMaxRowClasses], noh[kMaxRowClasses], ow0[kMaxRowClasses], now[kMaxRowClasses];
    unsigned magic_img[kMaxRowClasses];   // ceil(2^32 / (noh * now)): class row -> image
    unsigned magic_now[kMaxRowClasses];   // ceil(2^32 / now)
    int run_tile0[kMaxTileRuns + 1];
    int run_u0[kMaxTileRuns + 1];
    unsigned char run_kh0[kMaxTileRuns], run_nkh[kMaxTileRuns], run_kw0[kMaxTileRuns], run_nkw[kMaxTileRuns];
};

struct IgemmArgs {
    const float *x;     // activation to gather from: [B,H,W,Cx]
    const float *w;     // filter, HWIO
    float *y;           // [M][N]
    const float *bias;  // [N] or null
    const float *gate;  // [M][N] or null (dgrad: stored activation; out = gate>0 ? out*scale : 0)
    const unsigned char *mask;  // [M][N] keep mask or null
    int H, W, Cx, N, KW, dil, sign, pad_h, pad_w;
    int M, K, ntaps;
    int epi;            // EMADAPT_EPI_*
    float scale;        // gate scale
    float keep;         // dropout keep probability (1 = off)
    float inv_keep;     // 1.0f / keep (set_keep): what the kept elements are multiplied by
    int keep8;          // keep * 256 when that is an integer in 1 .. 255 (set_keep), else 0: the built-in generator then draws four elements per hash
    unsigned long long seed;
    int tiles_m, tiles_n;
    unsigned x_bytes, w_bytes;  // extents for the bounds-checked buffer loads (< 2 GiB each)
    unsigned magic_cx, magic_kw;  // ceil(2^32/Cx), ceil(2^32/KW): k / Cx and tap / KW by one v_mul_hi_u32
    int m_begin, m_end;  // output rows (pixels) this launch covers: [m_begin, m_end)
    int splits;          // split-K: K steps are divided over `splits` workgroups per tile (partials -> slab)
    float *slab;         // [splits][M][N] fp32 partials when splits > 1 (stream-K launches of igemm2_kernel: one 256x256 accumulator image per workgroup)
    unsigned *sk_flags;  // stream-K launches of igemm2_kernel: one arrival word per workgroup, zeroed ahead of the launch
    unsigned *status;    // the device status word (common.h status_word): a head whose wait ran out raises EMADAPT_STATUS_STREAMK_TIMEOUT
    int sk_drop_flag;    // tests only (tune "sk_drop_flag"): publishers keep their arrival words down, heads give up after 2^12 polls
    int tap_inner;       // 1: K order = (channel chunk, tap) -- needs Cx % BK == 0 (host decides)
    unsigned magic_nt;   // ceil(2^32/ntaps)
    RowPack pk;          // stream-K launches that skip padding (pk.n > 0)
    unsigned magic_w;    // ceil(2^32 / W)
    int gm;      // M tiles per group of the plain-tile order (tune "raster_gm", default 8)
    int lockstep;  // 1: every wave issues its DMA burst at the same point (tune "dma_stagger" = 1; default 0: SIMD partners staggered)
    int ablate;  // diagnostic builds only (tune "ablate"): 1 = no global loads, 2 = no LDS stores, 4 = no barrier, 8 = no A loads, 16 = no B loads, 32 = no epilogue, 64 = no fragment reads (igemm2_kernel)
};

// 16-byte buffer load with hardware range checking: an offset at/after the end of the buffer returns 0,
// which is how out-of-image taps, K/M/N tails are zero-filled without any branch or exec masking.
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float4 buf_ld4(__amdgpu_buffer_rsrc_t r, unsigned byte_off) {
    const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, (int)byte_off, 0, 0);
    return make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
}

// keep decision of the built-in dropout generator: a function of (seed, element index, keep_prob) alone, whatever kernel or vector
// width evaluates it.  Two forms, chosen by the keep probability:
//   * keep_prob = t / 256 (IgemmArgs::keep8 = t; the reference only ever feeds 0.5, deeplab.py:266): FOUR elements to one 32-bit hash --
//     element idx takes byte idx & 3 of the hash of idx >> 2 and is kept when that byte is below t.  Round 6: the fc6 / fc7 forward
//     epilogues draw 110 M + 110 M decisions per step inside waves that share their SIMD with the MFMAs, and the 32-bit multiplies
//     of the hash run at a quarter of the vector rate -- three per four elements now instead of four per two, and the 16-bit form's
//     int -> float conversion, multiply and float compare per element are a bit-field extract and an integer compare.
//   * any other keep_prob: a uniform 16-bit draw per element, two elements to one hash (idx takes half idx & 1 of the hash of
//     idx >> 1; keep probabilities resolve to 2^-16) -- rounds 2-5's generator, unchanged.
// The hash (both forms): the seed is hashed on its own into two words (a kernel argument: scalar-ALU work once per wave); one is XORed
// into the unit index's linear mix before a 32-bit multiply-xorshift finaliser, the other added between the finaliser's two rounds.
// A 32-bit state means every seed draws from the same 2^32 values, but two seeds visit them in unrelated orders (XOR and add do not
// commute with the multiplies, so no seed's stream is a shifted copy of another's -- scripts/dropout_quality.py checks the cross-seed
// correlation, the keep rate, lag correlations along channels and pixels and the joint distribution of a quad's / an octet's
// decisions).
__device__ __forceinline__ unsigned fmix32(unsigned h) {
    h ^= h >> 16;
    h *= 0x85EBCA6Bu;
    h ^= h >> 13;
    h *= 0xC2B2AE35u;
    h ^= h >> 16;
    return h;
}
__device__ __forceinline__ unsigned drop_hash(unsigned long long seed, unsigned long long unit) {
    const unsigned s1 = fmix32((unsigned)seed ^ fmix32((unsigned)(seed >> 32) + 0x9E3779B9u));
    const unsigned s2 = fmix32(s1 + 0x7F4A7C15u);
    unsigned h = (unsigned)unit * 0x9E3779B1u;
    const unsigned hi = (unsigned)(unit >> 32);
    if (__builtin_expect(hi != 0u, 0)) h += hi * 0x27D4EB2Fu;   // (tensors of 2^32 units and more: a branch the waves of a launch all skip)
    h ^= s1;
    h ^= h >> 15;
    h *= 0x2C1B3C6Du;
    h += s2;
    h ^= h >> 12;
    h *= 0x297A2D39u;
    h ^= h >> 15;
    return h;
}
__device__ __forceinline__ bool dropout_keep(unsigned long long seed, unsigned long long idx, float keep, int keep8) {
    if (keep8) return (int)((drop_hash(seed, idx >> 2) >> (8u * ((unsigned)idx & 3u))) & 0xFFu) < keep8;
    const unsigned h = drop_hash(seed, idx >> 1);
    const unsigned r = (idx & 1) ? (h >> 16) : (h & 0xFFFFu);
    return (float)r * (1.0f / 65536.0f) < keep;
}

// tf.nn.dropout scales the kept elements by 1 / keep_prob.  TF 1.x did that as x / keep_prob up to 1.12 and as
// x * (1 / keep_prob) from 1.13 on (the reference pins no version); for keep_prob = 2^-n -- 0.5 in the reference's training loop,
// deeplab.py:266 -- the two are the same number bit for bit.  Here: x * inv_keep with inv_keep = 1.0f / keep_prob rounded once
// on the host.  (A true fp32 division is ~10 vector instructions per element, unrolled 128 times in every fused epilogue: it
// cost registers the accumulators needed -- the dropout section of the epilogues was where these kernels spilled.)
inline void set_keep(IgemmArgs &a, float keep_prob) {
    a.keep = keep_prob;
    a.inv_keep = 1.0f / keep_prob;
    const float t = keep_prob * 256.0f;
    a.keep8 = (t >= 1.0f && t <= 255.0f && t == (float)(int)t) ? (int)t : 0;
}
__device__ __forceinline__ float drop_scale(const IgemmArgs &a, float v) { return v * a.inv_keep; }

// The four elements idx0 .. idx0 + 3 (idx0 % 4 == 0: every fused epilogue handles whole quads) -- what dropout_keep() returns for
// each: one hash in the 8-bit form, two in the 16-bit form.
__device__ __forceinline__ void dropout_keep4(const IgemmArgs &a, unsigned long long idx0, bool (&k)[4]) {
    if (a.keep8) {
        const unsigned h = drop_hash(a.seed, idx0 >> 2);
#pragma unroll
        for (int e = 0; e < 4; e++) k[e] = (int)((h >> (8 * e)) & 0xFFu) < a.keep8;
    } else {
#pragma unroll
        for (int e = 0; e < 4; e += 2) {
            const unsigned h = drop_hash(a.seed, (idx0 + e) >> 1);
            k[e] = (float)(h & 0xFFFFu) * (1.0f / 65536.0f) < a.keep;
            k[e + 1] = (float)(h >> 16) * (1.0f / 65536.0f) < a.keep;
        }
    }
}

// bias / ReLU / dropout (tf.nn.dropout: x / keep_prob * mask) or the ReLU-gradient gate, on one output element.
__device__ __forceinline__ float epilogue(const IgemmArgs &a, float v, int col, long long o) {
    if (a.epi != EMADAPT_EPI_NONE && a.bias) v += a.bias[col];
    if (a.epi == EMADAPT_EPI_BIAS_RELU) {
        v = relu_keep_nan(v);
        if (a.keep < 1.0f) {
            const bool kp = a.mask ? (a.mask[o] != 0) : dropout_keep(a.seed, (unsigned long long)o, a.keep, a.keep8);
            v = kp ? drop_scale(a, v) : 0.f;  // tf.nn.dropout divides by keep_prob
        }
    }
    if (a.gate) v = gate_keep_nan(a.gate[o], v * a.scale);
    return v;
}

// Tile epilogue through LDS, shared by the fp32 and bf16 kernels.  The operand tiles are dead after the main loop's
// last barrier, so every MFMA wave takes a private LDS region (`stage`, ROWS x (TN*32 + 4) floats), parks ROWS rows of
// its accumulator block there ([row][col] fp32; C/D layout of the 32x32 MFMA: col = lane&31, row = (r&3) + 8*(r>>2) +
// 4*(lane>>5)) and reads them back as 8 consecutive columns per lane.  Bias / ReLU / dropout / gate then run on 8
// values, and the gate loads and output stores are 16-byte vectors over contiguous rows instead of one scalar per
// lane and accumulator register.  Same arithmetic, element for element, as epilogue() above.
//   ROWS = 32 or 16 (16 when 32-row slabs of all waves would not fit in the kernel's tile LDS)
//   YBF  = output (and gate) stored as bf16 (N % 8 == 0), else f32 (N % 4 == 0)
typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));

// GEMM row -> pixel.  PixelRows: the rows ARE the pixels (every kernel but the padding-skipping stream-K launches).
struct PixelRows {
    __device__ __forceinline__ int operator()(const IgemmArgs &a, int row) const { return row < a.m_end ? row : -1; }
};
// MapRows: the rows are packed class after class (RowPack); the table gives the pixel of a GEMM row, -1 past the last pixel.
struct MapRows {
    const int *map;
    __device__ __forceinline__ int operator()(const IgemmArgs &, int j) const { return map[j]; }
};

//   ADJ  = the lane's TN accumulator tiles are TN ADJACENT columns (column l31*TN + j) instead of l31 + 32*j
//   RM   = GEMM row -> pixel (row0 is then a GEMM row)
//   DROP = false: the dropout section is compiled out (data-gradient instantiations, which never drop: the fp32 stream-K data gradient's
//          MFMA loop picked up scratch reloads when the round-6 generator made that dead section larger)
template <int TM, int TN, int ROWS, bool YBF, bool ADJ = false, typename RM = PixelRows, bool DROP = true>
__device__ __forceinline__ void store_acc_tile(const IgemmArgs &a, const f32x16 (&acc)[TM][TN], float *stage, int lane,
                                               int row0, int col0, float *slab, RM rm = RM()) {
    constexpr int WN = TN * 32;
    constexpr int SP = WN + 4;   // slab pitch in floats (rows stay 16-byte aligned)
    constexpr int CPR = WN / 8;  // 8-column chunks per slab row
    static_assert(ROWS == 32 || ROWS == 16, "slab height");
    const int l31 = lane & 31, lh = lane >> 5;
#pragma unroll
    for (int i = 0; i < TM; i++) {  // (fully unrolled: a dynamic index would move the accumulators to scratch)
#pragma unroll
        for (int h = 0; h < 32 / ROWS; h++) {
#pragma unroll
            for (int j = 0; j < TN; j++)
#pragma unroll
                for (int rr = 0; rr < ROWS / 2; rr++) {
                    const int r = h * (ROWS / 2) + rr;  // accumulator register; its row inside the slab:
                    stage[((r & 3) + 8 * ((r >> 2) % (ROWS / 8)) + 4 * lh) * SP + (ADJ ? l31 * TN + j : j * 32 + l31)] = acc[i][j][r];
                }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
            for (int it = 0; it < ROWS * CPR / 64; it++) {
                const int idx = it * 64 + lane;
                const int row_l = idx / CPR, ch = idx % CPR;
                const int row = rm(a, row0 + i * 32 + h * ROWS + row_l);
                const int col = col0 + ch * 8;
                const float4 v0 = *reinterpret_cast<const float4 *>(stage + row_l * SP + ch * 8);
                const float4 v1 = *reinterpret_cast<const float4 *>(stage + row_l * SP + ch * 8 + 4);
                if (row < 0 || col >= a.N) continue;
                const bool hi = col + 4 < a.N;  // N % 4 == 0: the chunk holds one or two whole quads
                const long long o = (long long)row * a.N + col;
                if (slab) {
                    *reinterpret_cast<float4 *>(slab + o) = v0;
                    if (hi) *reinterpret_cast<float4 *>(slab + o + 4) = v1;
                    continue;
                }
                float v[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
                if (a.epi != EMADAPT_EPI_NONE && a.bias) {
                    const float4 b0 = *reinterpret_cast<const float4 *>(a.bias + col);
                    const float4 b1 = hi ? *reinterpret_cast<const float4 *>(a.bias + col + 4) : make_float4(0.f, 0.f, 0.f, 0.f);
                    v[0] += b0.x; v[1] += b0.y; v[2] += b0.z; v[3] += b0.w;
                    v[4] += b1.x; v[5] += b1.y; v[6] += b1.z; v[7] += b1.w;
                }
                if (a.epi == EMADAPT_EPI_BIAS_RELU) {
#pragma unroll
                    for (int e = 0; e < 8; e++) v[e] = relu_keep_nan(v[e]);
                    if (DROP && a.keep < 1.0f) {
#pragma unroll
                        for (int e = 0; e < 8; e += 4) {   // (o is a multiple of 4: o + e .. o + e + 3 is a quad of the generator)
                            if (e >= 4 && !hi) break;
                            bool k[4];
                            if (a.mask) {
#pragma unroll
                                for (int q = 0; q < 4; q++) k[q] = a.mask[o + e + q] != 0;
                            } else dropout_keep4(a, (unsigned long long)(o + e), k);
#pragma unroll
                            for (int q = 0; q < 4; q++) v[e + q] = k[q] ? drop_scale(a, v[e + q]) : 0.f;
                        }
                    }
                }
                if (a.gate) {
                    float g[8];
                    if constexpr (YBF) {  // the gate has the storage type of the output
                        const bf16x8_t gv = *reinterpret_cast<const bf16x8_t *>(reinterpret_cast<const __bf16 *>(a.gate) + o);
#pragma unroll
                        for (int e = 0; e < 8; e++) g[e] = (float)gv[e];
                    } else {
                        const float4 g0 = *reinterpret_cast<const float4 *>(a.gate + o);
                        const float4 g1 = hi ? *reinterpret_cast<const float4 *>(a.gate + o + 4) : make_float4(0.f, 0.f, 0.f, 0.f);
                        g[0] = g0.x; g[1] = g0.y; g[2] = g0.z; g[3] = g0.w; g[4] = g1.x; g[5] = g1.y; g[6] = g1.z; g[7] = g1.w;
                    }
#pragma unroll
                    for (int e = 0; e < 8; e++) v[e] = gate_keep_nan(g[e], v[e] * a.scale);
                }
                if constexpr (YBF) {
                    bf16x8_t ov;
#pragma unroll
                    for (int e = 0; e < 8; e++) ov[e] = (__bf16)v[e];
                    *reinterpret_cast<bf16x8_t *>(reinterpret_cast<__bf16 *>(a.y) + o) = ov;
                } else {
                    *reinterpret_cast<float4 *>(a.y + o) = make_float4(v[0], v[1], v[2], v[3]);
                    if (hi) *reinterpret_cast<float4 *>(a.y + o + 4) = make_float4(v[4], v[5], v[6], v[7]);
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        }
    }
}

// split-K second stage: y = epilogue(sum_s slab[s]) with the partials added in ascending s (deterministic).
__global__ void __launch_bounds__(256) splitk_epilogue_kernel(IgemmArgs a) {
    const long long mn = (long long)(a.m_end - a.m_begin) * a.N;  // one slab
    const long long n4 = mn / 4;
    const long long o0 = (long long)a.m_begin * a.N;
    const bool small = (long long)a.m_end * a.N < (1LL << 31);   // 32-bit column arithmetic (a 64-bit modulo is ~10x the work)
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long long)gridDim.x * blockDim.x) {
        float4 v = *reinterpret_cast<const float4 *>(a.slab + i * 4);
        for (int sp = 1; sp < a.splits; sp++) {
            const float4 u = *reinterpret_cast<const float4 *>(a.slab + sp * mn + i * 4);
            v.x += u.x; v.y += u.y; v.z += u.z; v.w += u.w;
        }
        const long long o = o0 + i * 4;
        const int col = small ? (int)((unsigned)o % (unsigned)a.N) : (int)(o % a.N);
        v.x = epilogue(a, v.x, col + 0, o + 0);
        v.y = epilogue(a, v.y, col + 1, o + 1);
        v.z = epilogue(a, v.z, col + 2, o + 2);
        v.w = epilogue(a, v.w, col + 3, o + 3);
        *reinterpret_cast<float4 *>(a.y + o) = v;
    }
}

inline int launch_splitk_epilogue(const IgemmArgs &a, hipStream_t stream) {
    long long g = ((long long)(a.m_end - a.m_begin) * a.N / 4 + 255) / 256;
    if (g > 8192) g = 8192;
    hipLaunchKernelGGL(splitk_epilogue_kernel, dim3((unsigned)g), dim3(256), 0, stream, a);
    return launch_status();
}

// Split-K factor: with T tiles on `slots` resident workgroup slots (256 CUs x 2) a launch runs ceil(T/slots) rounds;
// when T is a few hundred (the M = 26 896, N = 512 layers: 844 tiles -> 2 rounds at 82 %) splitting K by S makes T*S
// workgroups fill whole rounds (S = 3 -> 4.95 of 5).  The price is one extra pass over S+1 output-sized slabs,
// charged at ~3 TB/s against the kernel's time at ~110 TFLOP/s.
int choose_splits(long long M, int N, int K, int bm, int bn, int bk, double slots = 512.0, double rate = 110e12,
                  double *t_out = nullptr) {
    const int force = tune(TUNE_SPLITK);
    slots *= plan_cus() / 256.0;
    const long long tiles = ceil_div(M, bm) * ceil_div(N, bn);
    const int nk = (K + bk - 1) / bk;
    const double t_kernel = 2.0 * M * (double)K * N / rate;
    int best = 1;
    double best_t = 1e30;
    for (int s = 1; s <= 8; s++) {
        if (s > 1 && (nk / s < 16 || N % 4 != 0)) break;
        const double rounds = (double)tiles * s / slots;
        const double eff = rounds / (double)(long long)(rounds + 0.999999);
        double t = t_kernel / eff;
        if (s > 1) t += (double)(s + 1) * M * N * 4.0 / 3e12 + 3e-6;
        if (force > 0 ? (s == (force <= nk ? force : 1)) : (t < best_t * 0.97)) { best_t = t; best = s; }
    }
    if (t_out) *t_out = best_t;   // estimated seconds of the launch with the chosen factor
    return best;
}

int check_desc(const emadapt_conv_desc *d) {
    if (!d) return EMADAPT_ERR_INVALID;
    if (d->B < 0 || d->H <= 0 || d->W <= 0 || d->Cin <= 0 || d->Cout <= 0 || d->KH <= 0 || d->KW <= 0 || d->dil <= 0)
        return EMADAPT_ERR_INVALID;
    if (d->Cin % 4 != 0 || d->Cout % 4 != 0) return EMADAPT_ERR_UNSUPPORTED;
    if ((long long)d->B * d->H * d->W >= (1LL << 30)) return EMADAPT_ERR_UNSUPPORTED;
    // activations and filters are addressed with 32-bit byte offsets (bounds-checked buffer loads)
    if ((long long)d->B * d->H * d->W * (d->Cin > d->Cout ? d->Cin : d->Cout) * 4 >= (1LL << 31)) return EMADAPT_ERR_UNSUPPORTED;
    if ((long long)d->KH * d->KW * d->Cin * d->Cout * 4 >= (1LL << 31)) return EMADAPT_ERR_UNSUPPORTED;
    // k -> (tap, channel) by magic division: exact while K * C < 2^32 (common.h: div_magic)
    const long long cmax = (d->Cin > d->Cout ? d->Cin : d->Cout) + 7;
    if ((long long)d->KH * d->KW * cmax * cmax >= (1LL << 32)) return EMADAPT_ERR_UNSUPPORTED;
    return EMADAPT_OK;
}

inline bool aligned16(const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

// Tensors of 2 GiB and more (513x513 inputs at batch 32: conv1's 2.16 GB activations): the kernels address an operand with
// 32-bit byte offsets, so the fp32 entries cut such a launch into slices of whole images, each slice a launch of its own on
// bumped base pointers (images are independent in forward and data gradient; the filter gradient adds the slices' results).
// Returns the images per slice (== d->B when the whole batch is one launch), 0 when not even one image fits.
inline int slice_batch(const emadapt_conv_desc *d, bool wgrad) {
    if (!d || d->B <= 0) return 0;
    emadapt_conv_desc one = *d;
    one.B = 1;
    if (check_desc(&one) != EMADAPT_OK) return 0;
    const long long hw = (long long)d->H * d->W, cmax = d->Cin > d->Cout ? d->Cin : d->Cout;
    long long b = ((1LL << 30) - 1) / hw;
    const long long b2 = ((1LL << 31) - 1) / (hw * cmax * 4);
    b = b2 < b ? b2 : b;
    if (wgrad) {
        const long long b3 = ((1LL << 32) - 1) / (hw * (d->W > d->H ? d->W : d->H));   // magic-division range of the pixel index
        b = b3 < b ? b3 : b;
    }
    if (b < 1) return 0;
    return (int)(b < d->B ? b : d->B);
}

// geometry shared by the launch and the workspace query
void fill_geometry(IgemmArgs &a, const emadapt_conv_desc *d, bool dgrad) {
    a.H = d->H; a.W = d->W; a.KW = d->KW; a.dil = d->dil; a.sign = dgrad ? -1 : 1;
    a.Cx = dgrad ? d->Cout : d->Cin;
    a.N = dgrad ? d->Cin : d->Cout;
    a.pad_h = conv_pad_before(d->KH, d->dil); a.pad_w = conv_pad_before(d->KW, d->dil);
    a.M = d->B * d->H * d->W; a.ntaps = d->KH * d->KW; a.K = a.ntaps * a.Cx;
    a.x_bytes = (unsigned)((long long)a.M * a.Cx * 4); a.w_bytes = (unsigned)((long long)a.K * a.N * 4);
    a.magic_cx = magic_u32(a.Cx); a.magic_kw = magic_u32(a.KW); a.magic_nt = magic_u32(a.ntaps);
    a.magic_w = magic_u32(a.W); a.pk.n = 0; a.pk.rowmap = nullptr;
    a.gm = tune(TUNE_RASTER_GM) > 0 ? tune(TUNE_RASTER_GM) : 8;
    a.lockstep = tune(TUNE_DMA_STAGGER) == 1 ? 1 : 0;
    a.sk_drop_flag = tune(TUNE_SK_DROP_FLAG) == 1 ? 1 : 0;
    a.sk_flags = nullptr;
    a.status = status_word();
}

}  // namespace

// bf16 forward / data-gradient kernel on LDS-DMA ("b16t": bf16 operands, Tiled filter image), included by conv_bf16.hip.
//
// Round 5 rebuild.  Rounds 1-4 had three designs of this kernel within +-6 % of each other at 43-52 % of the dense bf16 matrix rate
// (register-staged producers; LDS-DMA with 64-byte pieces issued by the MFMA waves themselves; the same with 128-byte pieces), all of
// them feeding v_mfma_f32_32x32x16_bf16 from 64 x 128 accumulator blocks.  scripts/bf16_quad_lab.hip (profiles/r05_bf16_quad_lab.txt;
// a plain GEMM at fc7's shape, random operands, one process) measured what moves that number:
//     8 waves, 32x32x16 (round 4's structure)                      1158-1244 TF    46-49 %
//     4 waves, one per SIMD, 128 x 128 accumulators, 32x32x16      1150-1173 TF    46 %     (16x16x32: 915-960 TF -- nobody covers a
//                                                                                             wave's DMA issue or its fragment latency)
//     8 waves, 16x16x32                                            1294-1320 TF    51-52 %  (the chip holds a higher clock on this shape:
//                                                                                             MFMAs alone 1860-1960 vs 1710-1730 TF)
//     8 MFMA waves 16x16x32 + 4 DMA waves, ring of 5               1356-1385 TF    54-55 %  <- this kernel
// So: v_mfma_f32_16x16x32_bf16, and the MFMA waves issue nothing but MFMAs and fragment reads -- an LDS-DMA instruction costs its
// issuing wave 60-180 cycles of address-unit wait (MI355X_MICROARCH.md), which in a wave that also owns half a SIMD's MFMA stream is
// matrix time.  The shape is the fp32 second-generation kernel's (igemm_v2.h):
//
//   * 256 x 256 x 32 tiles, one workgroup of 12 waves per CU: 8 MFMA waves (4 along M x 2 along N, a 64 x 128 accumulator block
//     each = 4 x 8 MFMA tiles of 16 x 16, 128 registers) + 4 DMA waves that issue all 32 LDS-DMA instructions of a K step.
//   * LDS image of an operand's K step: [256 rows][64 B] dense (16 KB); a stage = A + B = 32 KB; ring of 5 stages = all 160 KB,
//     fetched 3 steps ahead.  A DMA instruction moves 64 consecutive 16-byte slots = 16 rows.
//   * bank conflicts: slot s of row r holds the row's 16-byte k chunk s ^ bt_swz(r), bt_swz(r) = (-(r >> 2)) & 3 (the DMA lanes
//     fetch the chunks in that order, the fragment address applies the same XOR).  Conflict-free for the 16x16x32 fragment read (lane
//     (l15, lq) reads chunk lq of row l15: the four row quads of a ds_read_b128 lane group land on four different slot columns --
//     round 4's (r >> 2) & 3 is 2-way here) AND for the 32x32x16 one at ANY row offset (scripts/bf16_quad_lab.hip header).
//   * the FILTER is stored tile by tile in exactly that LDS order (filter_tiles_b16_kernel, once per optimizer step):
//         [tap][k / 32][row, zero rows up to a multiple of 256][4 slots of 16 B, swizzled]
//     so a K step's 256 filter rows are 16 contiguous KB and the filter DMAs are plain 1 KB copies.
//   * MFMA waves: a K step is 8 columns (B fragments of 16 filter rows, k = 32) x 4 MFMAs on the wave's four A fragments.  B
//     fragments stream through a 3-deep register rotation (fragment j + 2 is requested in front of column j's MFMAs), the A
//     fragments of the next step replace the current ones behind their last MFMAs: 16 + 12 fragment registers beside 128 accumulators
//     = 165 of the 168 a 12-wave workgroup leaves a wave.  The step's barrier sits in front of column 4; the fragments of the next
//     step are read behind it (columns 6 and 7).  The rotation advances by 8 % 3 = 2 slots per step: three unrolled bodies.
//   * DMA waves: per step 4 + 4 instructions each into the stage read two steps ago (a step's later B fragments are read behind its
//     own barrier, so the stage of step kt - 1 is still in use when step kt's DMAs are issued: ring of 5 for 3 steps ahead), then a
//     counted vmcnt wait (two steps stay in flight) and the barrier that publishes step kt + 1.
//   * work split: plain tiles (XCD-aware order, optional split-K into slab levels) or stream-K (SK: the K steps of all tiles cut
//     into one piece per CU; tiles shared by several workgroups are combined inside the launch like the fp32 kernel's: write-through
//     accumulator image per workgroup, arrival word, the head's workgroup adds in ascending k and runs the fused epilogue).
// Requirements: activations stored bf16 (x_dtype = EMADAPT_DT_BF16), Cx % 32 == 0, KH*KW <= 32, N % 8 == 0 for bf16 outputs,
// images below 2 GiB.  Replaces round 4's igemm_b16t_kernel / igemm_b16w_kernel (8 waves issuing their own DMAs; 64- and 128-byte
// activation pieces) and their tall-tile instantiations, none of which the model used.
#pragma once
#include "igemm_v2.h"

namespace {

typedef __bf16 bt_bf16x8 __attribute__((ext_vector_type(8)));
typedef float bt_f32x4 __attribute__((ext_vector_type(4)));

constexpr int BT_BK = 32;
constexpr int BT_ROWB = 2 * BT_BK;             // bytes of one row's K step, in memory and in LDS
constexpr int BT_OPER = 256 * BT_ROWB;         // one operand's K step in LDS (16 KB)
constexpr int BT_STAGE = 2 * BT_OPER;          // A + B
constexpr int BT_NBUF = 5;
constexpr int BT_AHEAD = 3;                    // K steps the DMA waves run ahead of the MFMA waves
constexpr int BT_MWAVES = 8, BT_LWAVES = 4;
constexpr int BT_THREADS = (BT_MWAVES + BT_LWAVES) * 64;
constexpr int BT_IPW = BT_OPER / 1024 / BT_LWAVES;   // DMA instructions per DMA wave, operand and K step (4)
constexpr size_t BT_LDS = (size_t)BT_NBUF * BT_STAGE;
static_assert(BT_LDS == 160 * 1024, "the ring fills the LDS");

__host__ __device__ inline int bt_swz(int r) { return (-(r >> 2)) & 3; }

// ------------------------------------------------------------------------------------------------------------------
// Tiled bf16 image of a filter: [tap][k / 32][row (zero rows up to a multiple of 256)][4 slots], slot s of row r = the row's
// k chunk s ^ bt_swz(r) (8 bf16 each).  DGRAD = false: rows = Cout, k = Cin (forward); true: rows = Cin, k = Cout (data
// gradient).  One thread = one 16-byte slot.
// ------------------------------------------------------------------------------------------------------------------
// Data-gradient image (rows = Cin, k = Cout): a slot's 8 k values are 32 contiguous bytes of the HWIO filter -- two 16-byte loads.
__global__ void __launch_bounds__(256) filter_tiles_b16_dgrad_kernel(const float *__restrict__ w, uint4 *__restrict__ out, int Cin, int Cout,
                                                                     int taps, int rows_pad) {
    const int G = Cout / BT_BK;
    const long long total = (long long)taps * G * rows_pad * 4;
    for (long long idx = (long long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long long)gridDim.x * 256) {
        const int slot = (int)(idx & 3);
        long long t = idx >> 2;
        const int row = (int)(t % rows_pad);
        t /= rows_pad;
        const int g = (int)(t % G), tap = (int)(t / G);
        const int chunk = slot ^ bt_swz(row);
        float4 f0 = make_float4(0.f, 0.f, 0.f, 0.f), f1 = f0;
        if (row < Cin) {
            const float *p = w + ((long long)tap * Cin + row) * Cout + g * BT_BK + chunk * 8;   // (Cout % 32 == 0: 32-byte aligned)
            f0 = *reinterpret_cast<const float4 *>(p);
            f1 = *reinterpret_cast<const float4 *>(p + 4);
        }
        bt_bf16x8 v;
        v[0] = (__bf16)f0.x; v[1] = (__bf16)f0.y; v[2] = (__bf16)f0.z; v[3] = (__bf16)f0.w;
        v[4] = (__bf16)f1.x; v[5] = (__bf16)f1.y; v[6] = (__bf16)f1.z; v[7] = (__bf16)f1.w;
        out[idx] = *reinterpret_cast<uint4 *>(&v);
    }
}

// Forward image (rows = Cout, k = Cin): the filter is contiguous along the ROWS, so a block transposes a [32 k][64 rows] piece
// through LDS (coalesced 256-byte reads per k, one 16-byte slot written per thread).  (The first version read 8 values a filter row
// apart per thread: 0.3 ms per refresh of fc6's image -- more than the kernel it feeds saved.)
__global__ void __launch_bounds__(256) filter_tiles_b16_fwd_kernel(const float *__restrict__ w, uint4 *__restrict__ out, int Cin, int Cout,
                                                                   int rows_pad) {
    __shared__ float t[32][65];
    const int G = Cin / BT_BK, RB = rows_pad / 64;
    int bid = blockIdx.x;
    const int rb = bid % RB;
    bid /= RB;
    const int g = bid % G, tap = bid / G;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
#pragma unroll
    for (int r = 0; r < 8; r++) {
        const int k = ty + 4 * r, row = rb * 64 + tx;
        t[k][tx] = row < Cout ? w[((long long)tap * Cin + g * BT_BK + k) * Cout + row] : 0.f;
    }
    __syncthreads();
    const int row_l = threadIdx.x >> 2, slot = threadIdx.x & 3;
    const int row = rb * 64 + row_l;
    const int chunk = slot ^ bt_swz(row);
    bt_bf16x8 v;
#pragma unroll
    for (int e = 0; e < 8; e++) v[e] = (__bf16)t[chunk * 8 + e][row_l];
    out[(((long long)tap * G + g) * rows_pad + row) * 4 + slot] = *reinterpret_cast<uint4 *>(&v);
}

// Both images from ONE pass over the f32 filter (a layer whose forward and data gradient both run on this kernel refreshes both every
// optimizer step: the two kernels above read the 260 MB of filters twice).  The [32 ci][64 co] piece a block holds in LDS for the
// forward image's transpose is also 32 rows x 2 k groups x 4 slots of the data-gradient image (rows = ci, k = co): one more 16-byte
// slot per thread, straight out of the LDS rows.  Needs Cin % 256 == 0 (the data-gradient image then has no padding rows to zero)
// and Cout % 64 == 0.
__global__ void __launch_bounds__(256) filter_tiles_b16_both_kernel(const float *__restrict__ w, uint4 *__restrict__ out_f, uint4 *__restrict__ out_d,
                                                                    int Cin, int Cout, int rows_pad_f) {
    __shared__ float t[32][65];
    const int G = Cin / BT_BK, RB = Cout / 64, GD = Cout / BT_BK;
    int bid = blockIdx.x;
    const int rb = bid % RB;
    bid /= RB;
    const int g = bid % G, tap = bid / G;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
#pragma unroll
    for (int r = 0; r < 8; r++) {
        const int k = ty + 4 * r;
        t[k][tx] = w[((long long)tap * Cin + g * BT_BK + k) * Cout + rb * 64 + tx];
    }
    __syncthreads();
    {   // forward image: row = co, k = ci (filter_tiles_b16_fwd_kernel's second half)
        const int row_l = threadIdx.x >> 2, slot = threadIdx.x & 3;
        const int row = rb * 64 + row_l;
        const int chunk = slot ^ bt_swz(row);
        bt_bf16x8 v;
#pragma unroll
        for (int e = 0; e < 8; e++) v[e] = (__bf16)t[chunk * 8 + e][row_l];
        out_f[(((long long)tap * G + g) * rows_pad_f + row) * 4 + slot] = *reinterpret_cast<uint4 *>(&v);
    }
    {   // data-gradient image: row = ci, k = co -- thread -> (ci of the piece, k group of the piece's 64 co, slot)
        const int slot = threadIdx.x & 3, gl = (threadIdx.x >> 2) & 1, ci_l = threadIdx.x >> 3;
        const int row = g * BT_BK + ci_l;
        const int chunk = slot ^ bt_swz(row);
        bt_bf16x8 v;
#pragma unroll
        for (int e = 0; e < 8; e++) v[e] = (__bf16)t[ci_l][gl * 32 + chunk * 8 + e];
        out_d[(((long long)tap * GD + rb * 2 + gl) * Cin + row) * 4 + slot] = *reinterpret_cast<uint4 *>(&v);   // (rows_pad == Cin)
    }
}

// The optimizer step of a layer AND both images, one pass (round 5): accumulate + Momentum + clean of one micro-batch on the block's
// [32 ci][64 co] piece of (w, v, g) -- emadapt_sgd_accumulate_momentum's arithmetic, operation for operation (common.h opt_*) -- the new
// weights go to memory and to the LDS piece the two image writers of the kernel above read.  The refresh pass re-read every filter the
// optimizer had just written: 4 of the 28 bytes per parameter the two passes moved.  The blocks behind the last piece update the
// biases (no decay).  A raised status word leaves weights, momentum and images alone, like the plain optimizer kernels.
__global__ void __launch_bounds__(256) sgd_tiles_b16_both_kernel(float *__restrict__ w, float *__restrict__ v, const float *__restrict__ g,
                                                                 uint4 *__restrict__ out_f, uint4 *__restrict__ out_d, int Cin, int Cout, int rows_pad_f,
                                                                 int pieces, long long nw, long long nb, float wd, float gscale, float lr, float mu,
                                                                 float ascale, const unsigned *__restrict__ status) {
    __shared__ float t[32][65];
    if (status_raised(status)) return;
    if ((int)blockIdx.x >= pieces) {   // biases
        const long long i = ((long long)blockIdx.x - pieces) * 256 + threadIdx.x;
        if (i < nb) {
            const float a = opt_accum(0.f, g[nw + i], gscale);
            const float vi = opt_velocity(mu, v[nw + i], a, ascale);
            v[nw + i] = vi;
            w[nw + i] = opt_step(w[nw + i], lr, vi);
        }
        return;
    }
    const int G = Cin / BT_BK, RB = Cout / 64, GD = Cout / BT_BK;
    int bid = blockIdx.x;
    const int rb = bid % RB;
    bid /= RB;
    const int gk = bid % G, tap = bid / G;
    {
        const int c4 = threadIdx.x & 15, ty = threadIdx.x >> 4;
#pragma unroll
        for (int r = 0; r < 2; r++) {
            const int k = ty + 16 * r;
            const long long o = ((long long)tap * Cin + gk * BT_BK + k) * Cout + rb * 64 + c4 * 4;
            float4 gi = *reinterpret_cast<const float4 *>(g + o);
            float4 wi = *reinterpret_cast<const float4 *>(w + o);
            float4 vi = *reinterpret_cast<const float4 *>(v + o);
            gi.x = opt_decay(gi.x, wd, wi.x); gi.y = opt_decay(gi.y, wd, wi.y); gi.z = opt_decay(gi.z, wd, wi.z); gi.w = opt_decay(gi.w, wd, wi.w);
            const float4 a = make_float4(opt_accum(0.f, gi.x, gscale), opt_accum(0.f, gi.y, gscale), opt_accum(0.f, gi.z, gscale), opt_accum(0.f, gi.w, gscale));
            vi.x = opt_velocity(mu, vi.x, a.x, ascale); vi.y = opt_velocity(mu, vi.y, a.y, ascale);
            vi.z = opt_velocity(mu, vi.z, a.z, ascale); vi.w = opt_velocity(mu, vi.w, a.w, ascale);
            wi.x = opt_step(wi.x, lr, vi.x); wi.y = opt_step(wi.y, lr, vi.y); wi.z = opt_step(wi.z, lr, vi.z); wi.w = opt_step(wi.w, lr, vi.w);
            *reinterpret_cast<float4 *>(v + o) = vi;
            *reinterpret_cast<float4 *>(w + o) = wi;
            t[k][c4 * 4 + 0] = wi.x; t[k][c4 * 4 + 1] = wi.y; t[k][c4 * 4 + 2] = wi.z; t[k][c4 * 4 + 3] = wi.w;
        }
    }
    __syncthreads();
    {   // forward image: row = co, k = ci
        const int row_l = threadIdx.x >> 2, slot = threadIdx.x & 3;
        const int row = rb * 64 + row_l;
        const int chunk = slot ^ bt_swz(row);
        bt_bf16x8 o;
#pragma unroll
        for (int e = 0; e < 8; e++) o[e] = (__bf16)t[chunk * 8 + e][row_l];
        out_f[(((long long)tap * G + gk) * rows_pad_f + row) * 4 + slot] = *reinterpret_cast<uint4 *>(&o);
    }
    {   // data-gradient image: row = ci, k = co
        const int slot = threadIdx.x & 3, gl = (threadIdx.x >> 2) & 1, ci_l = threadIdx.x >> 3;
        const int row = gk * BT_BK + ci_l;
        const int chunk = slot ^ bt_swz(row);
        bt_bf16x8 o;
#pragma unroll
        for (int e = 0; e < 8; e++) o[e] = (__bf16)t[ci_l][gl * 32 + chunk * 8 + e];
        out_d[(((long long)tap * GD + rb * 2 + gl) * Cin + row) * 4 + slot] = *reinterpret_cast<uint4 *>(&o);   // (rows_pad == Cin)
    }
}

// The same for the data-parallel job's Momentum pass (emadapt_sgd_momentum_payload: the summed gradient arrives as the exchange's
// gathered payload, f32 or bf16, decay and 1 / accum_num already inside): v = mu v + payload ascale, w -= lr v, accum = 0 where there
// is an accumulator -- on the block's piece, then both images.
template <typename P>
__global__ void __launch_bounds__(256) sgd_payload_tiles_b16_both_kernel(float *__restrict__ w, float *__restrict__ v, float *__restrict__ accum,
                                                                         const P *__restrict__ payload, uint4 *__restrict__ out_f,
                                                                         uint4 *__restrict__ out_d, int Cin, int Cout, int rows_pad_f, int pieces,
                                                                         long long nw, long long nb, float lr, float mu, float ascale,
                                                                         unsigned *__restrict__ status, int coll) {
    __shared__ float t[32][65];
    if (!coll && status_raised(status)) return;
    bool bad = false;   // (collective mode, common.h: an element whose summed gradient is NaN is left alone, on every rank alike)
    if ((int)blockIdx.x >= pieces) {   // biases
        const long long i = ((long long)blockIdx.x - pieces) * 256 + threadIdx.x;
        if (i < nb) {
            float vi = v[nw + i], wi = w[nw + i];
            opt_update(wi, vi, (float)payload[nw + i], lr, mu, ascale, coll, bad);
            v[nw + i] = vi;
            w[nw + i] = wi;
            if (accum) accum[nw + i] = 0.f;
        }
        if (bad) status_raise(status, EMADAPT_STATUS_NAN_GRADIENT);
        return;
    }
    const int G = Cin / BT_BK, RB = Cout / 64, GD = Cout / BT_BK;
    int bid = blockIdx.x;
    const int rb = bid % RB;
    bid /= RB;
    const int gk = bid % G, tap = bid / G;
    {
        const int c4 = threadIdx.x & 15, ty = threadIdx.x >> 4;
#pragma unroll
        for (int r = 0; r < 2; r++) {
            const int k = ty + 16 * r;
            const long long o = ((long long)tap * Cin + gk * BT_BK + k) * Cout + rb * 64 + c4 * 4;
            float a[4];
            if constexpr (sizeof(P) == 4) {
                const float4 pv = *reinterpret_cast<const float4 *>(payload + o);
                a[0] = pv.x; a[1] = pv.y; a[2] = pv.z; a[3] = pv.w;
            } else {
                typedef __bf16 pk_bf16x4 __attribute__((ext_vector_type(4)));
                const pk_bf16x4 pv = *reinterpret_cast<const pk_bf16x4 *>(payload + o);
                a[0] = (float)pv[0]; a[1] = (float)pv[1]; a[2] = (float)pv[2]; a[3] = (float)pv[3];
            }
            float4 wi = *reinterpret_cast<const float4 *>(w + o);
            float4 vi = *reinterpret_cast<const float4 *>(v + o);
            opt_update(wi.x, vi.x, a[0], lr, mu, ascale, coll, bad); opt_update(wi.y, vi.y, a[1], lr, mu, ascale, coll, bad);
            opt_update(wi.z, vi.z, a[2], lr, mu, ascale, coll, bad); opt_update(wi.w, vi.w, a[3], lr, mu, ascale, coll, bad);
            *reinterpret_cast<float4 *>(v + o) = vi;
            *reinterpret_cast<float4 *>(w + o) = wi;
            if (accum) *reinterpret_cast<float4 *>(accum + o) = make_float4(0.f, 0.f, 0.f, 0.f);
            t[k][c4 * 4 + 0] = wi.x; t[k][c4 * 4 + 1] = wi.y; t[k][c4 * 4 + 2] = wi.z; t[k][c4 * 4 + 3] = wi.w;
        }
    }
    if (bad) status_raise(status, EMADAPT_STATUS_NAN_GRADIENT);
    __syncthreads();
    {   // forward image: row = co, k = ci
        const int row_l = threadIdx.x >> 2, slot = threadIdx.x & 3;
        const int row = rb * 64 + row_l;
        const int chunk = slot ^ bt_swz(row);
        bt_bf16x8 o;
#pragma unroll
        for (int e = 0; e < 8; e++) o[e] = (__bf16)t[chunk * 8 + e][row_l];
        out_f[(((long long)tap * G + gk) * rows_pad_f + row) * 4 + slot] = *reinterpret_cast<uint4 *>(&o);
    }
    {   // data-gradient image: row = ci, k = co
        const int slot = threadIdx.x & 3, gl = (threadIdx.x >> 2) & 1, ci_l = threadIdx.x >> 3;
        const int row = gk * BT_BK + ci_l;
        const int chunk = slot ^ bt_swz(row);
        bt_bf16x8 o;
#pragma unroll
        for (int e = 0; e < 8; e++) o[e] = (__bf16)t[ci_l][gl * 32 + chunk * 8 + e];
        out_d[(((long long)tap * GD + rb * 2 + gl) * Cin + row) * 4 + slot] = *reinterpret_cast<uint4 *>(&o);   // (rows_pad == Cin)
    }
}

// Tile epilogue of a 64 x 128 block held as 4 x 8 accumulator tiles of the 16 x 16 MFMA (C/D layout: column = lane & 15, row =
// 4 (lane >> 4) + r): 16 rows at a time through the wave's private LDS slab ([16][128 + 4] floats), read back as 8 consecutive
// columns per lane -- then exactly store_acc_tile's arithmetic (igemm_common.h: bias / ReLU / dropout / gate on 8 values, 16-byte
// gate loads and output stores), element for element.
//   NI x NJ = accumulator tiles of the wave's block: 4 x 8 (64 rows x 128 columns, the 256-row tile) or 7 x 4 (112 x 64, the 224-row tile)
template <bool YBF, int NI = 4, int NJ = 8>
__device__ __forceinline__ void store_acc_tile16(const IgemmArgs &a, const bt_f32x4 (&acc)[NI][NJ], float *stage, int lane, int row0, int col0,
                                                 float *slab) {
    constexpr int WN = NJ * 16, SP = WN + 4, CPR = WN / 8;
    const int l15 = lane & 15, lq = lane >> 4;
#pragma unroll
    for (int i = 0; i < NI; i++) {   // (fully unrolled: a dynamic index would move the accumulators to scratch)
#pragma unroll
        for (int j = 0; j < NJ; j++)
#pragma unroll
            for (int r = 0; r < 4; r++) stage[(4 * lq + r) * SP + j * 16 + l15] = acc[i][j][r];
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
        for (int it = 0; it < 16 * CPR / 64; it++) {
            const int idx = it * 64 + lane;
            const int row_l = idx / CPR, ch = idx % CPR;
            const int row = row0 + i * 16 + row_l;
            const int col = col0 + ch * 8;
            const float4 v0 = *reinterpret_cast<const float4 *>(stage + row_l * SP + ch * 8);
            const float4 v1 = *reinterpret_cast<const float4 *>(stage + row_l * SP + ch * 8 + 4);
            if (row >= a.m_end || col >= a.N) continue;
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
                if (a.keep < 1.0f) {
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

// ABLM (diagnostic library only, tune "ablate"): parts of the kernel COMPILED out -- 1 no DMAs (every stage filled once), 2 no
// fragment reads, 4 no epilogue, 8 no MFMAs; results are garbage, scripts/bf16t_ablate.py reads the time.
// M224 (round 6; plain launches only): 224 x 256 tiles -- the 8 MFMA waves as 2 (rows) x 4 (columns), a 112 x 64 block each = 7 x 4
// accumulator tiles, 7 A + 4 B fragments per K step for 28 MFMAs (the 256-row tile: 12 for 32).  For the launches whose 256-row tiles
// fill one partial round: M = 26,896 (batch 16 at 41 x 41) is 105.06 tile rows -- 212 tiles of 256 x 256 on 256 CUs at N = 512, but 121 x 2
// = 242 tiles of 224 rows, each 7 / 8 of the work (conv4_x / conv5_x forward and data gradient, fc6's data gradient).  The operand stage
// keeps its 256 A rows: rows 224 .. 255 are never requested (their DMA lanes carry the out-of-range bit) and never read.
template <bool SK, bool YBF, int ABLM = 0, bool M224 = false>
__global__ void __launch_bounds__(BT_THREADS, 3) igemm_b16t_kernel(IgemmArgs a) {
    constexpr int BM = M224 ? 224 : 256, BN = 256, BK = BT_BK, NBUF = BT_NBUF, AHEAD = BT_AHEAD, STAGE = BT_STAGE, OPER = BT_OPER, IPW = BT_IPW;
    constexpr int NI = M224 ? 7 : 4, NJ = M224 ? 4 : 8;   // accumulator tiles (16 x 16) of a wave's block: rows x columns
    static_assert(!(SK && M224), "the 224-row tile runs plain launches only");
    extern __shared__ __attribute__((aligned(16))) char ldsb[];

    // ---- work of this workgroup: one (tile, K range) piece, or with SK a sequence of them (every wave computes the same list)
    const int nk_all = a.K / BK;
    int tile_m = 0, tile_n = 0, kt0 = 0, nk = 0;
    int lvl = -1;
    int sk_u = 0, sk_end = 0, sk_units = 0, sk_v = 0;
    {
        const int ntile = a.tiles_m * a.tiles_n;
        const int nblk = SK ? (int)gridDim.x : ntile * a.splits;
        const int bid = blockIdx.x;
        const int xcd = bid & 7, q = nblk >> 3, r = nblk & 7;
        int t = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
        if constexpr (SK) {
            sk_v = t;
            sk_units = ntile * nk_all;
            sk_u = __builtin_amdgcn_readfirstlane((int)sk_start(t, sk_units, nblk));
            sk_end = __builtin_amdgcn_readfirstlane((int)sk_start(t + 1, sk_units, nblk));
        } else {
            const int split = t / ntile;
            t -= split * ntile;
            const int GM = a.gm;
            const int per_group = GM * a.tiles_n;
            const int g = t / per_group;
            const int first_m = g * GM;
            const int gsz = (a.tiles_m - first_m < GM) ? (a.tiles_m - first_m) : GM;
            const int in_g = t - g * per_group;
            tile_m = first_m + in_g % gsz;
            tile_n = in_g / gsz;
            kt0 = (int)((long long)nk_all * split / a.splits);
            nk = (int)((long long)nk_all * (split + 1) / a.splits) - kt0;
            lvl = a.splits > 1 ? split : -1;
        }
    }
    auto next_piece = [&]() -> bool {
        if (sk_u >= sk_end) return false;
        const int t = __builtin_amdgcn_readfirstlane(sk_u / nk_all);
        kt0 = sk_u - t * nk_all;
        tile_m = __builtin_amdgcn_readfirstlane(t / a.tiles_n);
        tile_n = t - tile_m * a.tiles_n;
        const int left = sk_end - sk_u;
        nk = (nk_all - kt0 < left) ? nk_all - kt0 : left;
        lvl = nk == nk_all ? -1 : sk_v - __builtin_amdgcn_readfirstlane(sk_owner((long long)(sk_u - kt0), sk_units, (int)gridDim.x));
        sk_u += nk;
        return true;
    };
    if constexpr (SK) {
        if (!next_piece()) return;
    }

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);

    if (wave >= BT_MWAVES) {
        // ======================================= DMA waves =======================================
        const int lw = wave - BT_MWAVES;
        // (the tap offset rides in the DMA's scalar operand, an UNSIGNED addend: the buffer starts `bias` bytes before the tensor)
        const int rowb = a.Cx * 2;             // bytes of a pixel
        const int n_pad = (int)((a.N + 255) / 256) * 256;   // filter rows per (tap, group of 32 k) in the tiled filter image
        const int cgroups = a.Cx / BK;
        int min_dh = a.sign * (0 - a.pad_h), min_dw = a.sign * (0 - a.pad_w);
        {
            const int KH = a.ntaps / a.KW;
            const int dh1 = a.sign * ((KH - 1) * a.dil - a.pad_h), dw1 = a.sign * ((a.KW - 1) * a.dil - a.pad_w);
            min_dh = dh1 < min_dh ? dh1 : min_dh;
            min_dw = dw1 < min_dw ? dw1 : min_dw;
        }
        const int min_sp = min_dh * a.W + min_dw;
        const unsigned bias = min_sp < 0 ? (unsigned)(-min_sp) * (unsigned)rowb : 0u;
        const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<char *>(reinterpret_cast<const char *>(a.x)) - bias, 0, (int)(a.x_bytes + bias), 0x00020000);
        const __amdgpu_buffer_rsrc_t wr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(a.w), 0, (int)a.w_bytes, 0x00020000);
        constexpr unsigned kOOB = 0x80000000u;
        // instruction j of this wave moves slots [(lw * IPW + j) * 64, +64) of an operand's image: lane -> (row, slot)
        int ld_row[IPW], ld_chunk[IPW];
        unsigned b_off[IPW];
#pragma unroll
        for (int j = 0; j < IPW; j++) {
            const int q = (lw * IPW + j) * 64 + lane;
            ld_row[j] = q >> 2;
            ld_chunk[j] = (q & 3) ^ bt_swz(ld_row[j]);
            b_off[j] = (unsigned)q * 16u;   // the filter image is stored in LDS order: plain copies
        }
        do {   // one piece of work (SK: several)
            const int m0 = a.m_begin + tile_m * BM, n0 = tile_n * BN;
            unsigned a_off[IPW], a_bad[IPW];
#pragma unroll
            for (int j = 0; j < IPW; j++) {
                const int r = ld_row[j];
                const int m = (r < BM && m0 + r < a.m_end) ? m0 + r : -1;
                if (m >= 0) {
                    const int t2 = m / a.W;
                    const int oh = t2 % a.H, ow = m - t2 * a.W;
                    a_off[j] = (unsigned)m * (unsigned)rowb + (unsigned)ld_chunk[j] * 16u;
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
            // DMAs of K step kt (of this piece's tile) into the ring stage at byte offset `off`: this wave's 4 KB of each operand.
            // K order = (channel group, tap).  A step past the end of the piece (live = false) carries the out-of-range bit on every
            // lane and deposits zeros: every step issues the same eight instructions, so the vmcnt bookkeeping is a constant.
            auto fill = [&](int off, int kt, bool live) {
                if constexpr (ABLM & 1) { if (kt - kt0 >= NBUF) return; }
                const unsigned chunk = __umulhi((unsigned)kt, a.magic_nt);
                const unsigned tap = a.ntaps > 1 ? (unsigned)kt - chunk * (unsigned)a.ntaps : 0u;
                const unsigned chunk_u = a.ntaps > 1 ? chunk : (unsigned)kt;
                const unsigned kh = div_magic(tap, a.magic_kw);
                const int kw = (int)(tap - kh * (unsigned)a.KW);
                const int dh = a.sign * ((int)kh * a.dil - a.pad_h);
                const int dw = a.sign * (kw * a.dil - a.pad_w);
                const unsigned s_x = (unsigned)__builtin_amdgcn_readfirstlane((int)((unsigned)((dh * a.W + dw) * rowb) + chunk_u * (unsigned)BT_ROWB + bias));
                const unsigned s_w = (unsigned)__builtin_amdgcn_readfirstlane((int)(((tap * (unsigned)cgroups + chunk_u) * (unsigned)n_pad + (unsigned)n0) * (unsigned)BT_ROWB));
                const unsigned sh = 31u - (tap & 31u);   // moves this tap's "outside the image" bit to bit 31
                const unsigned dead = live ? 0u : kOOB;
                char *as = ldsb + off + lw * (IPW * 1024);
                char *bs = ldsb + off + OPER + lw * (IPW * 1024);
#pragma unroll
                for (int j = 0; j < IPW; j++)
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(xr, (lds_void_t *)(as + j * 1024), 16, (int)(a_off[j] | ((a_bad[j] << sh) & kOOB) | dead), (int)s_x, 0, 0);
#pragma unroll
                for (int j = 0; j < IPW; j++)
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(wr, (lds_void_t *)(bs + j * 1024), 16, (int)(b_off[j] | dead), (int)s_w, 0, 0);
            };
            __syncthreads();   // the ring is free (SK: the previous piece's epilogue staged through it)
#pragma unroll
            for (int s = 0; s < AHEAD; s++) fill(s * STAGE, kt0 + s, s < nk);
            wait_vmcnt<(AHEAD - 1) * 2 * IPW>();
            __builtin_amdgcn_s_barrier();   // step 0 has landed
            int dst = AHEAD * STAGE;
            for (int kt = 0; kt < nk; kt++) {
                fill(dst, kt0 + kt + AHEAD, kt + AHEAD < nk);
                wait_vmcnt<(AHEAD - 1) * 2 * IPW>();   // step kt + 1 has landed (kt + 2, kt + 3 in flight)
                __builtin_amdgcn_s_barrier();
                dst = dst == (NBUF - 1) * STAGE ? 0 : dst + STAGE;
            }
            // the zero-depositing DMAs of the last steps are still in flight: drain them before the ring becomes the epilogue's
            // staging area
            wait_vmcnt<0>();
            __syncthreads();
            if constexpr (SK) __syncthreads();   // (pairs with the MFMA waves' barrier between a publisher's slab stores and its arrival word)
        } while (SK && next_piece());
        return;
    }

    // ======================================= MFMA waves =======================================
    const int wm = M224 ? (wave & 1) : (wave & 3), wn = M224 ? (wave >> 1) : (wave >> 2);
    // fragment addresses (bytes inside a stage): lane (l15, lq) reads k 8 lq .. 8 lq + 7 of row l15 (+ 16 i): chunk lq, stored at
    // slot lq ^ bt_swz(row); 16 rows on are four row quads: the same swizzle
    const int l15 = lane & 15, lq = lane >> 4;
    const int a_base = (wm * (NI * 16) + l15) * BT_ROWB + ((lq ^ bt_swz(l15)) << 4);
    const int b_base = OPER + (wn * (NJ * 16) + l15) * BT_ROWB + ((lq ^ bt_swz(l15)) << 4);
    bt_f32x4 acc[NI][NJ];
    bt_bf16x8 av[NI], bv[3];
    int sk_bad = 0;   // (scalar) a publisher of a tile this workgroup heads never arrived
#pragma unroll
    for (int n = 0; n < NI; n++)
#pragma unroll
        for (int e = 0; e < 8; e++) av[n][e] = (__bf16)0.f;
#pragma unroll
    for (int n = 0; n < 3; n++)
#pragma unroll
        for (int e = 0; e < 8; e++) bv[n][e] = (__bf16)0.f;

    do {   // one piece of work (SK: several)
    const int m0 = a.m_begin + tile_m * BM, n0 = tile_n * BN;
#pragma unroll
    for (int i = 0; i < NI; i++)
#pragma unroll
        for (int j = 0; j < NJ; j++)
#pragma unroll
            for (int r = 0; r < 4; r++) acc[i][j][r] = 0.f;
    __syncthreads();                // (pairs with the DMA waves' barrier at the top of a piece)
    __builtin_amdgcn_s_barrier();   // step 0 has landed
    asm volatile("" ::: "memory");
#define BT_SB __builtin_amdgcn_sched_barrier(0);
#define BT_RDA(I, BASE) if constexpr (!(ABLM & 2)) { av[I] = *reinterpret_cast<const bt_bf16x8 *>(ldsb + (BASE) + a_base + (I) * (16 * BT_ROWB)); BT_SB }
#define BT_RDB(S, J, BASE) if constexpr (!(ABLM & 2)) { bv[S] = *reinterpret_cast<const bt_bf16x8 *>(ldsb + (BASE) + b_base + (J) * (16 * BT_ROWB)); BT_SB }
#define BT_MF(I, J, S) if constexpr (!(ABLM & 8)) { acc[I][J] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av[I], bv[S], acc[I][J], 0, 0, 0); BT_SB }
    if constexpr (M224) {
    // ---- 7 x 4 block: column J = 7 MFMAs on B fragment J; fragment J + 2 is requested in front of them (from the next stage once J + 2 >= 4:
    // behind the step's barrier, which sits in front of column 2); the rotation slot of column J of step kt is (J + 4 kt) % 3 = (J + kt) % 3:
    // three bodies, O = 0, 1, 2; the next step's A fragments replace the current ones behind their last MFMAs (column 3)
    BT_RDA(0, 0) BT_RDA(1, 0) BT_RDA(2, 0) BT_RDA(3, 0) BT_RDA(4, 0) BT_RDA(5, 0) BT_RDA(6, 0) BT_RDB(0, 0, 0) BT_RDB(1, 1, 0)
#define BT_COL7(J, O)                                                                                                                  \
        if ((J) == 2) { __builtin_amdgcn_s_barrier(); asm volatile("" ::: "memory"); BT_SB }                                           \
        if ((J) + 2 < 4) { BT_RDB(((J) + 2 + (O)) % 3, (J) + 2, cur) } else { BT_RDB(((J) + 2 + (O)) % 3, (J) + 2 - 4, nx) }           \
        BT_MF(0, J, ((J) + (O)) % 3) if ((J) == 3) { BT_RDA(0, nx) }                                                                   \
        BT_MF(1, J, ((J) + (O)) % 3) if ((J) == 3) { BT_RDA(1, nx) }                                                                   \
        BT_MF(2, J, ((J) + (O)) % 3) if ((J) == 3) { BT_RDA(2, nx) }                                                                   \
        BT_MF(3, J, ((J) + (O)) % 3) if ((J) == 3) { BT_RDA(3, nx) }                                                                   \
        BT_MF(4, J, ((J) + (O)) % 3) if ((J) == 3) { BT_RDA(4, nx) }                                                                   \
        BT_MF(5, J, ((J) + (O)) % 3) if ((J) == 3) { BT_RDA(5, nx) }                                                                   \
        BT_MF(6, J, ((J) + (O)) % 3) if ((J) == 3) { BT_RDA(6, nx) }
#define BT_STEP7(O)                                                                \
        {                                                                          \
            const int nx = cur == (NBUF - 1) * STAGE ? 0 : cur + STAGE;            \
            BT_SB                                                                  \
            BT_COL7(0, O) BT_COL7(1, O) BT_COL7(2, O) BT_COL7(3, O)                \
            cur = nx;                                                              \
        }
        int cur = 0;
        for (int kt = 0; kt < nk; kt += 3) {
            BT_STEP7(0)
            if (kt + 1 < nk) BT_STEP7(1)
            if (kt + 2 < nk) BT_STEP7(2)
        }
#undef BT_STEP7
#undef BT_COL7
    } else {
    BT_RDA(0, 0) BT_RDA(1, 0) BT_RDA(2, 0) BT_RDA(3, 0) BT_RDB(0, 0, 0) BT_RDB(1, 1, 0)
    // column J of the wave's block: 4 MFMAs on B fragment J; fragment J + 2 is requested in front of them (from the next stage once
    // J + 2 >= 8: behind the step's barrier, which sits in front of column 4).  The rotation slot of column J of step kt is
    // (J + 2 kt) % 3 (8 columns advance it by 2 per step): three bodies, O = 0, 2, 1.  Every statement ends in a scheduling barrier:
    // the order written IS the schedule.
#define BT_COL(J, O)                                                                                                                   \
        if ((J) == 4) { __builtin_amdgcn_s_barrier(); asm volatile("" ::: "memory"); BT_SB }                                           \
        if ((J) + 2 < 8) { BT_RDB(((J) + 2 + (O)) % 3, (J) + 2, cur) } else { BT_RDB(((J) + 2 + (O)) % 3, (J) + 2 - 8, nx) }           \
        BT_MF(0, J, ((J) + (O)) % 3) if ((J) == 7) { BT_RDA(0, nx) }                                                                   \
        BT_MF(1, J, ((J) + (O)) % 3) if ((J) == 7) { BT_RDA(1, nx) }                                                                   \
        BT_MF(2, J, ((J) + (O)) % 3) if ((J) == 7) { BT_RDA(2, nx) }                                                                   \
        BT_MF(3, J, ((J) + (O)) % 3) if ((J) == 7) { BT_RDA(3, nx) }
#define BT_STEP(O)                                                                                                  \
        {                                                                                                           \
            const int nx = cur == (NBUF - 1) * STAGE ? 0 : cur + STAGE;                                             \
            BT_SB                                                                                                   \
            BT_COL(0, O) BT_COL(1, O) BT_COL(2, O) BT_COL(3, O) BT_COL(4, O) BT_COL(5, O) BT_COL(6, O) BT_COL(7, O) \
            cur = nx;                                                                                               \
        }
    {
        int cur = 0;
        for (int kt = 0; kt < nk; kt += 3) {
            BT_STEP(0)
            if (kt + 1 < nk) BT_STEP(2)
            if (kt + 2 < nk) BT_STEP(1)
        }
    }
    }   // (!M224)
#undef BT_STEP
#undef BT_COL
#undef BT_MF
#undef BT_RDB
#undef BT_RDA
#undef BT_SB
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __syncthreads();   // (pairs with the DMA waves' closing barrier: no DMA is in flight, nobody reads the ring)

    int lane_e = tid & 63;
    asm volatile("" : "+v"(lane_e));
    if constexpr (SK) {
        // Tiles shared by several workgroups are combined INSIDE the launch, exactly as in the fp32 kernel (igemm_v2.h, note at sk_start:
        // only a workgroup's FIRST piece can start inside a tile, so it publishes at most one partial tile, early, and a tile's head
        // is the LAST piece of the workgroup in front): the publisher stores its accumulators in register order -- register
        // (i * 8 + j) of wave w <-> bytes [(w * 32 + i * 8 + j) * 1024 + lane * 16, +16) of slab[workgroup] -- with write-through (sc1)
        // stores, every storing wave drains them, the workgroup's barrier follows, ONE lane raises the arrival word; the head polls the
        // words of the workgroups behind it in turn, reads their slabs with sc1 loads and adds them in ascending k.  The memory-model
        // contract is DESIGN.md 4.1's (MI355X_MICROARCH.md, "Valid forms", first row of the sc1 hand-off table).
        const unsigned lane_off = (unsigned)(lane_e * 16);
        constexpr int kWaveFloats = kSkSlabFloats / BT_MWAVES;
        if (lvl > 0) {
            const __amdgpu_buffer_rsrc_t sr = __builtin_amdgcn_make_buffer_rsrc(a.slab + (long long)sk_v * kSkSlabFloats + wave * kWaveFloats, 0, kWaveFloats * 4, 0x00020000);
#pragma unroll
            for (int i = 0; i < 4; i++)
#pragma unroll
                for (int j = 0; j < 8; j++)
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, acc[i][j]), sr, (int)lane_off, (i * 8 + j) * 1024, 16);   // aux 16 = sc1
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // every storing wave, ahead of the barrier that precedes the arrival word
        }
        if (lvl == 0) {
            // head of a shared tile (the piece list has moved on: sk_u - nk is this tile's first unit)
            const int sk_more = __builtin_amdgcn_readfirstlane(sk_owner((long long)(sk_u - nk) + nk_all - 1, sk_units, (int)gridDim.x)) - sk_v;
            for (int l = 1; l <= sk_more; l++) {
                const sk_flag_t *f = ((const sk_flag_t *)a.sk_flags) + (sk_v + l);
                bool ok = false;
                for (unsigned spins = 0; spins < (a.sk_drop_flag ? (1u << 12) : kSkSpinMax); spins++) {
                    if (__hip_atomic_load(f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) { ok = true; break; }
                    __builtin_amdgcn_s_sleep(8);
                }
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");   // (no instruction: keeps the loads below the poll)
                const __amdgpu_buffer_rsrc_t sr = __builtin_amdgcn_make_buffer_rsrc(a.slab + (long long)(sk_v + l) * kSkSlabFloats + wave * kWaveFloats, 0, ok ? kWaveFloats * 4 : 0, 0x00020000);
#pragma unroll
                for (int i = 0; i < 4; i++) {
                    bt_f32x4 v[8];
#pragma unroll
                    for (int j = 0; j < 8; j++)
                        v[j] = __builtin_bit_cast(bt_f32x4, __builtin_amdgcn_raw_buffer_load_b128(sr, (int)lane_off, (i * 8 + j) * 1024, 16));   // sc1
#pragma unroll
                    for (int j = 0; j < 8; j++) acc[i][j] += v[j];
                    __builtin_amdgcn_sched_barrier(0);
                }
                // (a publisher that never arrived -- a co-tenant kernel that never ends: a NaN in every 16 x 16 block of the tile, which
                // the epilogue propagates, and the failure in the device status word, raised when the kernel ends)
                sk_bad |= __builtin_amdgcn_readfirstlane(ok ? 0 : 1);
#pragma unroll
                for (int i = 0; i < 4; i++)
#pragma unroll
                    for (int j = 0; j < 8; j++) acc[i][j][0] = ok ? acc[i][j][0] : __builtin_nanf("");
            }
        }
    }
    if (!(SK && lvl > 0)) {
        float *slab_ = !SK && lvl >= 0 ? a.slab + ((long long)lvl * (a.m_end - a.m_begin) - a.m_begin) * a.N : nullptr;   // (split-K levels of the plain launches)
        float *stage = reinterpret_cast<float *>(ldsb) + wave * (16 * (NJ * 16 + 4));
        static_assert(BT_MWAVES * 16 * (128 + 4) * 4 <= BT_LDS, "epilogue slabs must fit in the ring");
        if (!(ABLM & 4) || a.m_end < -1)   // (ablation 4: a condition that is never true at run time keeps the accumulators alive)
            store_acc_tile16<YBF, NI, NJ>(a, acc, stage, lane_e, m0 + wm * (NI * 16), n0 + wn * (NJ * 16), slab_);
    }
    if constexpr (SK) {
        __syncthreads();   // every wave's slab stores were drained ahead of this barrier
        if (lvl > 0 && tid == 0 && !a.sk_drop_flag)
            __hip_atomic_store(((sk_flag_t *)a.sk_flags) + sk_v, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    } while (SK && next_piece());
    if constexpr (SK) {
        if (sk_bad && a.status && (threadIdx.x & 63) == 0)
            __hip_atomic_fetch_or(a.status, EMADAPT_STATUS_STREAMK_TIMEOUT, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

// a.splits: SK = false -> split-K factor (>= 1; > 1 needs a.slab); SK = true -> number of workgroups (CUs), a.slab for the levels
template <bool SK, bool YBF, int ABLM = 0, bool M224 = false>
int launch_igemm_b16t(IgemmArgs &a, hipStream_t stream) {
#ifdef EMADAPT_ABLATE_BUILD
    if constexpr (ABLM == 0 && YBF && !M224) {
        switch (tune(TUNE_ABLATE)) {
#define EMADAPT_ABL_CASE(m) case m: return launch_igemm_b16t<SK, YBF, m>(a, stream);
            EMADAPT_ABL_CASE(1) EMADAPT_ABL_CASE(2) EMADAPT_ABL_CASE(3) EMADAPT_ABL_CASE(4) EMADAPT_ABL_CASE(8) EMADAPT_ABL_CASE(9) EMADAPT_ABL_CASE(10)
#undef EMADAPT_ABL_CASE
            default: break;
        }
    }
#endif
    static DynLdsAttr attr;
    int rc = attr.ensure(reinterpret_cast<const void *>(&igemm_b16t_kernel<SK, YBF, ABLM, M224>), BT_LDS);
    if (rc) return rc;
    a.tiles_m = (int)ceil_div(a.m_end - a.m_begin, M224 ? 224 : 256);
    a.tiles_n = (int)ceil_div(a.N, 256);
    const int nk_all = a.K / BT_BK;
    const long long tiles = (long long)a.tiles_m * a.tiles_n;
    if constexpr (SK) {
        // a.splits = CUs to lay the pieces out for; a.sk_flags / a.slab = the workspace of streamk_workspace_bytes(G) when tiles are shared
        if (a.splits < 1 || tiles * nk_all > 0x7fffffffLL || tiles * 16 > 0x7fffffffLL) return EMADAPT_ERR_UNSUPPORTED;
        const int G = streamk_grid(tiles, nk_all, a.splits);
        const int levels = streamk_levels(tiles, nk_all, a.splits);
        if (levels > 0) {   // shared tiles: arrival words (zeroed here, ahead of every launch) + one accumulator image per workgroup
            if (!a.slab || !a.sk_flags) return EMADAPT_ERR_WORKSPACE;
            EMADAPT_HIP_TRY(hipMemsetAsync(a.sk_flags, 0, streamk_flag_bytes(G), stream));
        }
        hipLaunchKernelGGL((igemm_b16t_kernel<true, YBF, ABLM>), dim3((unsigned)G), dim3(BT_THREADS), BT_LDS, stream, a);
        return launch_status();
    } else {
        const long long nblk = tiles * a.splits;
        if (nblk > 0x7fffffffLL) return EMADAPT_ERR_UNSUPPORTED;
        hipLaunchKernelGGL((igemm_b16t_kernel<false, YBF, ABLM, M224>), dim3((unsigned)nblk), dim3(BT_THREADS), BT_LDS, stream, a);
        rc = launch_status();
        if (rc || a.splits <= 1) return rc;
        long long g = ((long long)(a.m_end - a.m_begin) * a.N / 4 + 255) / 256;
        if (g > 8192) g = 8192;
        hipLaunchKernelGGL(splitk_epilogue_t_kernel<YBF>, dim3((unsigned)g), dim3(256), 0, stream, a);
        return launch_status();
    }
}

}  // namespace

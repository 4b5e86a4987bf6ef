// Direct 3 x 3 convolution for 64 -> 64 channels on bf16-stored tensors (conv1_2's forward and data gradient), included by
// conv_bf16.hip (round 5).  The layer is the largest tensor of the net (16 x 321 x 321 x 64: 211 MB in, 211 MB out) with the smallest
// GEMM behind it (K = 576, N = 64): 0.085 ms of HBM time, 0.05 ms of matrix time -- and 0.26 / 0.29 ms on the implicit-GEMM kernels,
// which gather every activation nine times and stream a filter that never changes.  What the patch kernel's ablation named
// (profiles/r05_p3_patch_kernel.txt: a K step of 16 MFMAs cannot carry a filter stream, a barrier and a DMA burst) is taken away here:
//   * the filter -- 9 taps x 64 x 64 bf16 = 72 KB, 72 MFMA A-operand fragments of 16 output channels x 32 input channels -- is
//     loaded once per workgroup: the first D64_RTAPS taps live in the REGISTERS of every wave, the others in LDS as ready-made fragment
//     images (1 KB, one conflict-free ds_read_b128 each, shared by the wave's four pixel rows).  One wave per SIMD, four per CU, no
//     producer waves: every wave issues its share of the DMAs between its MFMAs.  (All nine taps in registers = 288 + accumulators +
//     the gate prefetch spilled; six taps made the compiler park the fragments in AGPRs and copy them out in front of every MFMA.)
//   * persistent workgroups walk 16 x 16-pixel output tiles; the 18 x 18-pixel input patch (128 B per pixel, 41.5 KB) comes in by
//     LDS-DMA into one of two buffers while the previous tile is computed: each activation crosses the vector memory path once per
//     tile (1.27x with the halo) instead of nine times, and there is ONE barrier per tile, none per K step;
//   * the MFMA's B operand (32 input channels x 16 pixels) is read from the patch with ds_read_b128 at register + immediate addresses:
//     a tap is a constant shift of (ty x 18 + tx) pixels; the row part is an immediate, the column part and the swizzle cost six
//     address registers per lane;
//   * one wave per SIMD has nobody to hide behind: an MFMA keeps the pipe for 16 cycles and the wave may issue three more instructions
//     in them, so the loop is written as that interleave -- every MFMA is followed by one small piece of the other work (a fragment
//     read for the next group, a step of the next DMA's address arithmetic, the DMA), pinned with scheduling barriers.  The first form
//     (the 16 MFMAs of a group, then everything else) ran the MFMA loop alone at 3x its matrix time (scripts/d64_ablate.py);
//   * accumulators are transposed (output channels along the MFMA's rows): a lane ends up with 4 consecutive channels of one pixel
//     = 8 bytes of bf16 after bias / ReLU.  Each wave turns an output row (16 pixels) through 2 KB of LDS (8-byte pieces at a swizzled
//     position, conflict-free both ways) and writes 1 KB of whole lines per instruction; the ReLU gradient's gate is fetched in that
//     same whole-line layout ahead of the MFMAs and applied to the bf16 value (a select: exact);
//   * bank conflicts: a pixel row is 128 B = half the banks, pixels 2 apart collide.  The 16-byte chunk c of patch column px is stored
//     at chunk position c ^ s(px) (the DMA lanes fetch their chunks in that order); s is a table found by search that makes every
//     ds_read_b128 lane group of the 16 x 16 x 32 operand hit 64 different banks at all three column shifts.
// Arithmetic: bf16 products, fp32 sums over (tap, channel) in one accumulator per output -- the same model as the implicit-GEMM kernels
// (the summation order differs: within the parity bar, not bit-identical to them); bit-reproducible run to run.
// Requirements: 3 x 3, rate 1, 64 gathered and 64 output channels, x / y (and the gate) stored bf16, no dropout in the epilogue.
#pragma once

namespace {

constexpr int D64_PW = 18;                       // patch width and height in pixels (16 + the halo)
constexpr int D64_PATCH = 44 * 1024;             // 324 pixels x 128 B = 40.5 KB; every wave issues 11 DMA instructions of 1 KB (the last three: padding)
constexpr int D64_RTAPS = 3;                     // filter taps held in registers; the others as fragment images in LDS
constexpr int D64_FLDS = 2 * D64_PATCH;          // [9 - D64_RTAPS taps][2][4] fragments of 1 KB
constexpr int D64_BIAS = D64_FLDS + (9 - D64_RTAPS) * 8 * 1024;   // 64 floats
constexpr int D64_STAGE = D64_BIAS + 256;        // [4 waves][16 pixels][128 B]: an output row on its way from the accumulator layout to whole lines
constexpr size_t D64_LDS = D64_STAGE + 4 * 2048;
static_assert(D64_LDS <= 160 * 1024, "LDS budget");
constexpr unsigned long long D64_SWZ = 0xd92dad912240ull;   // s(px), 3 bits each, px = 0 .. 17: 0 0 1 1 2 2 4 4 5 5 6 6 2 2 6 6 0 0
__device__ __forceinline__ int d64_swz(int px) { return (int)((D64_SWZ >> (3 * px)) & 7ull); }

// The ReLU-gradient gate on two packed bf16 values: out = gate > 0 ? v : (v is NaN ? v : 0) per half (gate_keep_nan on the rounded value),
// in packed 16-bit integer arithmetic -- gate > 0 as a bf16 <=> 1 <= bits <= 0x7f80 (not zero, not negative, not NaN), v is NaN <=>
// (bits & 0x7fff) > 0x7f80; checked over all 65,536 patterns of either (scripts-free: ten instructions per pair against seventeen in fp32).
typedef unsigned short d64_u16x2 __attribute__((ext_vector_type(2)));
typedef short d64_s16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned d64_gate_pk(unsigned v, unsigned gbits) {
    const d64_u16x2 gg = __builtin_bit_cast(d64_u16x2, gbits);
    const d64_u16x2 a = __builtin_elementwise_min(gg, (d64_u16x2)(0x7f81));     // negatives (as unsigned) and NaNs -> 0x7f81
    const d64_u16x2 b = a - (d64_u16x2)(1);                                      // 0 -> 0xffff
    const d64_u16x2 c = __builtin_elementwise_min(b, (d64_u16x2)(0x7f80));       // valid <=> c < 0x7f80
    const d64_s16x2 mg = (__builtin_bit_cast(d64_s16x2, c) - (d64_s16x2)(0x7f80)) >> (d64_s16x2)(15);
    const d64_s16x2 e = __builtin_bit_cast(d64_s16x2, v & 0x7fff7fffu);
    const d64_s16x2 mn = ((d64_s16x2)(0x7f80) - e) >> (d64_s16x2)(15);
    return v & __builtin_bit_cast(unsigned, mg | mn);
}

struct D64Geo {
    int tiles_x, tiles_y, tiles;
    unsigned magic_txy, magic_tx;
};

inline bool d64_shape_ok(const IgemmArgs &a) {   // (plus, at the launch: x, y and the gate stored bf16, no dropout)
    if (a.ntaps != 9 || a.KW != 3 || a.dil != 1 || a.Cx != 64 || a.N != 64) return false;
    if (a.M <= 0 || a.M % (a.H * a.W) != 0 || (long long)a.M * 128 >= (1LL << 31) || a.H > 16384 || a.W > 16384) return false;
    const long long per_img = (long long)((a.W + 15) / 16) * ((a.H + 15) / 16), tiles = (long long)(a.M / (a.H * a.W)) * per_img;
    return (tiles + 2 * 1024) * per_img < (1LL << 32);   // (the tile -> image division by multiply-high, tiles past the end included)
}

// ABLM (diagnostic library only, tune "ablate"): parts COMPILED out -- 1 no patch DMAs behind the first, 2 no MFMAs, 4 no output stores,
// 8 no gate loads, 16 no fragment reads, 32 no per-tile barrier, 128 nothing (the build that skips the stores at run time: a.ablate & 64);
// results are garbage, scripts/d64_ablate.py reads the time.
// DG: the data gradient (filter taps mirrored, gate epilogue instead of bias / ReLU) -- two kernels, so that neither carries the other's epilogue.
template <bool DG, int ABLM = 0>
__global__ void __launch_bounds__(256, 1) conv_d64_kernel(IgemmArgs a, D64Geo geo) {
    extern __shared__ __attribute__((aligned(16))) char ldsb[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int G = (int)gridDim.x, bid = (int)blockIdx.x;
    const int my_tiles = bid < geo.tiles ? (geo.tiles - 1 - bid) / G + 1 : 0;
    if (my_tiles == 0) return;
    const int r = lane & 15, g = lane >> 4;
    const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(a.x), 0, (int)a.x_bytes, 0x00020000);
    constexpr unsigned kOOB = 0x80000000u;

    int t_img, t_y0, t_x0;   // a tile of this workgroup: image, first row, first column
    auto tile_coords = [&](int ti) {
        const int t = bid + ti * G;
        const unsigned img = div_magic((unsigned)t, geo.magic_txy);
        const unsigned rem = (unsigned)t - img * (unsigned)(geo.tiles_x * geo.tiles_y);
        const unsigned tyi = div_magic(rem, geo.magic_tx);
        t_img = (int)img;
        t_y0 = (int)tyi * 16;
        t_x0 = (int)(rem - tyi * (unsigned)geo.tiles_x) * 16;
    };
    // The patch image in LDS: [18 x 18 pixels][8 chunk positions of 16 B]; DMA instruction n = wave + 4 s (s = 0 .. 10) covers its units
    // 64 n .. 64 n + 63: unit -> pixel (py, px), position cp holding chunk cp ^ s(px).  Per lane and slot, tile-independent: the source
    // offset relative to the patch's first pixel and (py, px); units past the patch (n = 40's upper half, n > 40) are never in range.
    int d_rel[11], d_pyx[11];
#pragma unroll
    for (int s = 0; s < 11; s++) {
        const int u = (wave + 4 * s) * 64 + lane;
        const int P = u >> 3;
        const int py = (P * 3641) >> 16, px = P - py * D64_PW;   // (P / 18, exact below 400)
        d_rel[s] = (py * a.W + px) * 128 + (((u & 7) ^ d64_swz(px < D64_PW ? px : 0)) << 4);
        d_pyx[s] = P < D64_PW * D64_PW ? (py | (px << 16)) : 0x7fff;   // (row + 32767: never inside an image, H <= 16384)
    }
    // one DMA instruction in three steps (so that each fits between two MFMAs); halo outside the image, tiles past the end: zeros
    int d_iy, d_ix, d_tb;        // scratch of the steps; d_tb = byte offset of the patch's first pixel (y0 - 1, x0 - 1), d_live = tile exists
    bool d_live;
    unsigned d_off;
    auto dma_a = [&](int s) { d_iy = t_y0 - 1 + (d_pyx[s] & 0xffff); d_ix = t_x0 - 1 + (d_pyx[s] >> 16); };
    auto dma_b = [&](int s) {
        const bool ok = d_live && (unsigned)d_iy < (unsigned)a.H && (unsigned)d_ix < (unsigned)a.W;
        d_off = (unsigned)(d_tb + d_rel[s]) | (ok ? 0u : kOOB);
    };
    auto dma_c = [&](int s, int buf) {
        __builtin_amdgcn_raw_ptr_buffer_load_lds(xr, (lds_void_t *)(ldsb + buf * D64_PATCH + (wave + 4 * s) * 1024), 16, (int)d_off, 0, 0, 0);
    };
    auto patch_tile = [&](int ti) {   // the tile whose patch the next dma_* calls fetch
        tile_coords(ti);
        d_live = ti < my_tiles;
        d_tb = ((t_img * a.H + t_y0 - 1) * a.W + t_x0 - 1) * 128;
    };
    patch_tile(0);
#pragma unroll
    for (int s = 0; s < 11; s++) { dma_a(s); dma_b(s); dma_c(s, 0); }

    // the filter: fragment (tap, h, j) = output channels 16 j + (lane & 15), input channels 32 h + 8 (lane >> 4) .. + 7 of the tap
    // (image [tap][n][64], the reduction channel contiguous -- both the forward and the data-gradient image have that form)
    bt_bf16x8 F[D64_RTAPS][2][4];
    {
        const char *wb = reinterpret_cast<const char *>(a.w);
#pragma unroll
        for (int tap = 0; tap < 9; tap++) {
            const int ti = DG ? 8 - tap : tap;
#pragma unroll
            for (int h = 0; h < 2; h++)
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    const char *src = wb + (((ti * 64 + j * 16 + r) * 64) + h * 32 + g * 8) * 2;
                    if (tap < D64_RTAPS) {
                        F[tap][h][j] = *reinterpret_cast<const bt_bf16x8 *>(src);
                    } else if (((tap * 2 + h) & 3) == wave) {   // (the LDS fragments: a quarter per wave)
                        *reinterpret_cast<bt_bf16x8 *>(ldsb + D64_FLDS + (((tap - D64_RTAPS) * 2 + h) * 4 + j) * 1024 + lane * 16) =
                            *reinterpret_cast<const bt_bf16x8 *>(src);
                    }
                }
        }
    }
    if (tid < 64) reinterpret_cast<float *>(ldsb + D64_BIAS)[tid] = (!DG && a.epi != EMADAPT_EPI_NONE && a.bias) ? a.bias[tid] : 0.f;

    // B-operand addresses: pixel (row 4 wave + i + ty, column r + tx) of the patch, chunk 4 h + g -> one register per (tx, h), the
    // row (i + ty) an immediate
    int xa[3][2];
#pragma unroll
    for (int tx = 0; tx < 3; tx++)
#pragma unroll
        for (int h = 0; h < 2; h++)
            xa[tx][h] = ((4 * wave) * D64_PW + r + tx) * 128 + (((4 * h + g) ^ d64_swz(r + tx)) << 4);

    wait_vmcnt<0>();
    // (the compiler's own wait-count bookkeeping does not see the inline wait: without a use of the register fragments HERE it waits for their
    // loads inside the tile loop, at vmcnt values that make the MFMAs wait for the gate loads and the previous tile's stores)
#pragma unroll
    for (int tap = 0; tap < D64_RTAPS; tap++)
#pragma unroll
        for (int h = 0; h < 2; h++)
#pragma unroll
            for (int j = 0; j < 4; j++) asm volatile("" : "+v"(F[tap][h][j]));
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // (this wave's filter fragments and bias are in LDS before it joins the barrier)
    __builtin_amdgcn_s_barrier();   // patch 0, the filter's LDS part and the bias are in place
    asm volatile("" ::: "memory");

    const bool relu = !DG && a.epi == EMADAPT_EPI_BIAS_RELU;
    const __bf16 *gate = DG ? reinterpret_cast<const __bf16 *>(a.gate) : nullptr;
    __bf16 *yb = reinterpret_cast<__bf16 *>(a.y);
    const int ocol = lane >> 3, och = lane & 7;   // whole-line layout: a lane = the 16-byte chunk och of pixel column 8 hh + ocol of an output row
    for (int ti = 0; ti < my_tiles; ti++) {
        const int buf = ti & 1;
        tile_coords(ti);
        const int oy0w = t_y0 + 4 * wave, t_x0c = t_x0, t_imgc = t_img;   // this wave's four output rows
        u32x4 gv[8];
        if (DG && !(ABLM & 8) && gate) {   // the stored activation behind the ReLU gradient, fetched ahead of the MFMAs
#pragma unroll
            for (int kk = 0; kk < 8; kk++) {
                const int oy = oy0w + (kk >> 1), oxx = t_x0c + 8 * (kk & 1) + ocol;
                const bool ok = oy < a.H && oxx < a.W;
                gv[kk] = ok ? *reinterpret_cast<const u32x4 *>(gate + (((long long)t_imgc * a.H + oy) * a.W + oxx) * 64 + och * 8) : u32x4{0u, 0u, 0u, 0u};
            }
        }
        patch_tile(ti + 1);   // (its patch is requested between the MFMAs below)
        bt_f32x4 acc[4][4];
#pragma unroll
        for (int i = 0; i < 4; i++)
#pragma unroll
            for (int j = 0; j < 4; j++)
#pragma unroll
                for (int e = 0; e < 4; e++) acc[i][j][e] = 0.f;
        const char *pb = ldsb + buf * D64_PATCH;
        // 18 groups (tap, channel half) of 4 B fragments (the wave's four pixel rows) x 4 A fragments (the filter) = 16 MFMAs; behind
        // MFMA (i, j) of group q comes one piece of: the next group's fragment reads (j = 0: B fragment i; j = 1: A fragment i where the
        // tap's filter is in LDS), DMA instruction q of the next patch (three steps behind (i, 2) / (i, 3) of i = 0, 1)
        bt_bf16x8 X[2][4], FL[2][4];
#define D64_SB __builtin_amdgcn_sched_barrier(0);
#define D64_RX(Q, I)                                                                                                                              \
        if constexpr (!(ABLM & 16)) {                                                                                                             \
            X[(Q) & 1][I] = *reinterpret_cast<const bt_bf16x8 *>(pb + xa[((Q) >> 1) % 3][(Q) & 1] + ((I) + ((Q) >> 1) / 3) * (D64_PW * 128)); D64_SB \
        }
#define D64_RF(Q, J)                                                                                                                              \
        if (!(ABLM & 16) && ((Q) >> 1) >= D64_RTAPS) {                                                                                            \
            FL[(Q) & 1][J] = *reinterpret_cast<const bt_bf16x8 *>(ldsb + D64_FLDS + (((((Q) >> 1) - D64_RTAPS) * 2 + ((Q) & 1)) * 4 + (J)) * 1024 + lane * 16); D64_SB \
        }
        D64_RX(0, 0) D64_RX(0, 1) D64_RX(0, 2) D64_RX(0, 3) D64_RF(0, 0) D64_RF(0, 1) D64_RF(0, 2) D64_RF(0, 3)
#pragma unroll
        for (int q = 0; q < 18; q++) {
            const int tap = q >> 1, h = q & 1;
#pragma unroll
            for (int i = 0; i < 4; i++)
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    if constexpr (!(ABLM & 2)) {
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(tap < D64_RTAPS ? F[tap < D64_RTAPS ? tap : 0][h][j] : FL[q & 1][j], X[q & 1][i],
                                                                            acc[i][j], 0, 0, 0);
                        D64_SB
                    }
                    if (q + 1 < 18) {
                        if (j == 0) { D64_RX(q + 1, i) }
                        if (j == 1) { D64_RF(q + 1, i) }
                    }
                    if (!(ABLM & 1) && q < 11) {
                        if (i == 0 && j == 2) { dma_a(q); D64_SB }
                        if (i == 0 && j == 3) { dma_b(q); D64_SB }
                        if (i == 1 && j == 2) { dma_c(q, buf ^ 1); D64_SB }
                    }
                }
        }
#undef D64_RF
#undef D64_RX
#undef D64_SB
        wait_vmcnt<0>();   // the next patch (this wave's part of it) and the gate have landed -- in front of the stores, which then drain behind
                           // the next tile's MFMAs instead of being waited for here
        // epilogue, one output row (16 pixels) at a time: acc[i][j][e] = pixel (row i, column r), channel 16 j + 4 g + e -> bias / ReLU / gate
        // scale in fp32, 4 bf16 = one 8-byte piece, parked at piece position (4 j + g) ^ r of pixel r's 128-byte row (the 16 lanes of a write
        // hit 16 different positions); read back as 16-byte chunks, 8 lanes per pixel: two instructions of 1 KB of whole lines per row
        if constexpr (ABLM & 256) {   // diagnostic: no epilogue -- one word that depends on every accumulator, stored only if the run-time flag allows
            float sum = 0.f;
#pragma unroll
            for (int i = 0; i < 4; i++)
#pragma unroll
                for (int j = 0; j < 4; j++) sum += (acc[i][j][0] + acc[i][j][1]) + (acc[i][j][2] + acc[i][j][3]);
            if (!(a.ablate & 64)) reinterpret_cast<float *>(a.y)[lane] = sum;
            if constexpr (!(ABLM & 32)) __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            continue;
        }
        char *stg = ldsb + D64_STAGE + wave * 2048;
        const float gscale = (DG && gate) ? a.scale : 1.f;
        float4 bj[4];
#pragma unroll
        for (int j = 0; j < 4; j++) bj[j] = DG ? make_float4(0.f, 0.f, 0.f, 0.f) : *reinterpret_cast<const float4 *>(ldsb + D64_BIAS + (j * 16 + 4 * g) * 4);
#pragma unroll
        for (int i = 0; i < 4; i++) {
#pragma unroll
            for (int j = 0; j < 4; j++) {
                float4 v = make_float4(acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]);
                if constexpr (DG) {
                    v.x *= gscale; v.y *= gscale; v.z *= gscale; v.w *= gscale;
                } else {
                    v.x += bj[j].x; v.y += bj[j].y; v.z += bj[j].z; v.w += bj[j].w;
                    if (relu) { v.x = relu_keep_nan(v.x); v.y = relu_keep_nan(v.y); v.z = relu_keep_nan(v.z); v.w = relu_keep_nan(v.w); }
                }
                *reinterpret_cast<uint2 *>(stg + r * 128 + (((4 * j + g) ^ r) << 3)) = cvt4(v);
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // (the wave's own pieces: LDS operations of one wave complete in order)
#pragma unroll
            for (int hh = 0; hh < 2; hh++) {
                const int kk = 2 * i + hh, col = 8 * hh + ocol;
                // chunk och = pieces 2 och, 2 och + 1 at positions (2 och) ^ col, (2 och + 1) ^ col: one aligned 16-byte unit, halves swapped for odd columns
                u32x4 o = *reinterpret_cast<const u32x4 *>(stg + col * 128 + ((och ^ (col >> 1)) << 4));
                if (col & 1) o = u32x4{o.z, o.w, o.x, o.y};
                if (DG && gate) {   // out = gate > 0 ? v : (v is NaN ? v : 0), on the rounded value (a select: the same bits as gating before the rounding)
                    const unsigned res[4] = {d64_gate_pk(o.x, gv[kk].x), d64_gate_pk(o.y, gv[kk].y), d64_gate_pk(o.z, gv[kk].z), d64_gate_pk(o.w, gv[kk].w)};
                    o = u32x4{res[0], res[1], res[2], res[3]};
                }
                const int oy = oy0w + i, oxx = t_x0c + col;
                if (!(ABLM & 4) && !(ABLM && (a.ablate & 64)) && oy < a.H && oxx < a.W)   // (64: skipped at run time -- the arithmetic stays alive)
                    *reinterpret_cast<u32x4 *>(yb + (((long long)t_imgc * a.H + oy) * a.W + oxx) * 64 + och * 8) = o;
            }
        }
        if constexpr (!(ABLM & 32)) __builtin_amdgcn_s_barrier();   // everybody's part of the next patch has landed, and nobody reads this tile's buffer any more
        asm volatile("" ::: "memory");
    }
    wait_vmcnt<0>();   // (the zero fill requested for the tile past the end lands in LDS this workgroup still owns)
}

template <int ABLM = 0>
int launch_d64(IgemmArgs &a, bool dgrad, hipStream_t stream) {
    if (dgrad ? (a.bias != nullptr || a.epi != EMADAPT_EPI_NONE) : a.gate != nullptr) return EMADAPT_ERR_INVALID;   // (neither entry point passes these)
#ifdef EMADAPT_ABLATE_BUILD
    if constexpr (ABLM == 0) {
        switch (tune(TUNE_ABLATE)) {
#define EMADAPT_ABL_CASE(m) case m: return launch_d64<m>(a, dgrad, stream);
            EMADAPT_ABL_CASE(1) EMADAPT_ABL_CASE(2) EMADAPT_ABL_CASE(8) EMADAPT_ABL_CASE(16) EMADAPT_ABL_CASE(32)
#undef EMADAPT_ABL_CASE
            case 64: a.ablate = 64; return launch_d64<128>(a, dgrad, stream);        // everything compiled in, the output stores skipped at run time
            case 65: a.ablate = 64; return launch_d64<129>(a, dgrad, stream);        // .. and no patch DMAs: the MFMA loop + epilogue arithmetic alone
            case 80: a.ablate = 64; return launch_d64<144>(a, dgrad, stream);        // .. no fragment reads
            case 81: a.ablate = 64; return launch_d64<145>(a, dgrad, stream);        // .. no fragment reads, no DMAs: MFMAs + epilogue arithmetic
            case 66: a.ablate = 64; return launch_d64<130>(a, dgrad, stream);        // .. no MFMAs: DMAs + reads + epilogue arithmetic
            case 82: a.ablate = 64; return launch_d64<146>(a, dgrad, stream);        // .. no MFMAs, no fragment reads: DMAs + epilogue arithmetic
            case 83: a.ablate = 64; return launch_d64<147>(a, dgrad, stream);        // .. no MFMAs, reads, DMAs: the epilogue arithmetic alone
            case 320: a.ablate = 64; return launch_d64<384>(a, dgrad, stream);       // no epilogue: DMAs + reads + MFMAs
            case 321: a.ablate = 64; return launch_d64<385>(a, dgrad, stream);       // no epilogue, no DMAs: reads + MFMAs
            case 337: a.ablate = 64; return launch_d64<401>(a, dgrad, stream);       // no epilogue, no DMAs, no reads: the MFMAs alone
            case 353: a.ablate = 64; return launch_d64<433>(a, dgrad, stream);       // .. and no barrier
            default: break;
        }
    }
#endif
    if constexpr (ABLM == 0) a.ablate = 0;
    static DynLdsAttr attr_f, attr_d;
    const int arc = dgrad ? attr_d.ensure(reinterpret_cast<const void *>(&conv_d64_kernel<true, ABLM>), D64_LDS)
                          : attr_f.ensure(reinterpret_cast<const void *>(&conv_d64_kernel<false, ABLM>), D64_LDS);
    if (arc) return arc;
    D64Geo geo;
    const int B = a.M / (a.H * a.W);
    geo.tiles_x = (a.W + 15) / 16;
    geo.tiles_y = (a.H + 15) / 16;
    geo.tiles = B * geo.tiles_x * geo.tiles_y;
    geo.magic_txy = magic_u32(geo.tiles_x * geo.tiles_y);
    geo.magic_tx = magic_u32(geo.tiles_x);
    if (geo.tiles <= 0) return EMADAPT_OK;
    const int G = geo.tiles < plan_cus() ? geo.tiles : plan_cus();
    if (dgrad) hipLaunchKernelGGL((conv_d64_kernel<true, ABLM>), dim3((unsigned)G), dim3(256), D64_LDS, stream, a, geo);
    else hipLaunchKernelGGL((conv_d64_kernel<false, ABLM>), dim3((unsigned)G), dim3(256), D64_LDS, stream, a, geo);
    return launch_status();
}

}  // namespace

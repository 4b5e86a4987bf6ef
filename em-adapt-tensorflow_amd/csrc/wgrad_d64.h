// Direct filter gradient for 3 x 3, 64 -> 64 channels on bf16-stored tensors (conv1_2's), included by conv_bf16.hip (round 5):
// dw[tap][ci][co] = sum_p x[p + tap][ci] dy[p][co] over 1.65 M pixels -- 422 MB of operands for a 147 KB result.  The split-M kernel
// (wgrad_bf16_kernel<128, 64, 64>: five filter-row tiles x 102 pixel ranges) streams dy five times and x through register-staged
// producers: 0.19 - 0.21 ms, 2 TB/s.  Here, in conv_d64.h's mould:
//   * persistent workgroups walk 16 x 16-pixel tiles; per tile the 18 x 18 patch of x and the 16 x 16 tile of dy come in ONCE by LDS-DMA
//     (two stages of 44 + 32 KB), one barrier per tile;
//   * the WHOLE filter gradient of the workgroup's tiles stays in registers: wave w owns input channels 16 w .. 16 w + 15 of all nine
//     taps = 36 accumulator tiles of the 16 x 16 MFMA (144 registers), plus one tile for the bias gradient (an all-ones A fragment
//     against the wave's own dy fragment: BiasAddGrad rides on the matrix pipe, 1 MFMA in 37);
//   * the reduction index is the pixel: a K step = 32 pixels (two tile rows), both operands read with the transposing
//     ds_read_b64_tr_b16 from images that lie as they lie in memory ([pixel][64 channels]); a tap is a constant pixel shift of the x
//     fragment.  Pixel <-> k-slot: the 16-lane group g, read h, lane quad q <-> tile row 2 s + (g >> 1), column 8 h + 4 (g & 1) + q, so
//     the 8 pixel rows of a half-wave read are 8 consecutive pixels of one row; the 32-byte unit u (16 channels) of pixel P is stored
//     at unit position u ^ ((P >> 1) & 3) -- 8 consecutive pixels then cover the 8 bank segments once, at any tap shift;
//   * every MFMA is followed by one piece of the other work (next step's fragment reads, the next tile's DMA instructions in three steps
//     each), pinned with scheduling barriers: one wave per SIMD has nobody to hide behind (conv_d64.h);
//   * at the end each workgroup writes its partial gradient (147 KB + 256 B) to its slab; the usual two-segment slab_sum adds the
//     slabs in workgroup order: bit-reproducible.
// Arithmetic: bf16 products, fp32 sums -- a workgroup's partial is summed in tile order, the partials in workgroup order (another
// order than the split-M kernel's: within the parity bar, not bit-identical to it).
// Requirements: 3 x 3, rate 1, 64 input and 64 output channels, x and dy stored bf16.
#pragma once

namespace {

constexpr int WD_XBYTES = 44 * 1024;            // the x patch: 324 pixels x 128 B = 40.5 KB, 11 DMA instructions per wave (the last three: padding)
constexpr int WD_YBYTES = 32 * 1024;            // the dy tile: 256 pixels x 128 B, 8 DMA instructions per wave
constexpr int WD_STAGE = WD_XBYTES + WD_YBYTES;
constexpr size_t WD_LDS = 2 * WD_STAGE;
static_assert(WD_LDS <= 160 * 1024, "LDS budget");

inline bool wgrad_d64_shape_ok(const emadapt_conv_desc *d, int x_bf16, int dy_bf16) {
    if (!x_bf16 || !dy_bf16 || d->KH != 3 || d->KW != 3 || d->dil != 1 || d->Cin != 64 || d->Cout != 64 || d->B <= 0) return false;
    const long long M = (long long)d->B * d->H * d->W;
    if (M * 128 >= (1LL << 31) || d->H > 16384 || d->W > 16384) return false;
    const long long per_img = (long long)((d->W + 15) / 16) * ((d->H + 15) / 16);
    return ((long long)d->B * per_img + 2 * 1024) * per_img < (1LL << 32);   // (the tile -> image division by multiply-high)
}

__global__ void __launch_bounds__(256, 1) wgrad_d64_kernel(WgradBf16Args a, D64Geo geo) {
    extern __shared__ __attribute__((aligned(16))) char ldsb[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int G = (int)gridDim.x, bid = (int)blockIdx.x;
    const int my_tiles = bid < geo.tiles ? (geo.tiles - 1 - bid) / G + 1 : 0;
    const int g = lane >> 4, q = (lane >> 2) & 3, p = lane & 3;
    const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(a.x), 0, (int)a.x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t yr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(a.dy), 0, (int)a.y_bytes, 0x00020000);
    constexpr unsigned kOOB = 0x80000000u;

    int t_img, t_y0, t_x0;
    auto tile_coords = [&](int ti) {
        const int t = bid + ti * G;
        const unsigned img = div_magic((unsigned)t, geo.magic_txy);
        const unsigned rem = (unsigned)t - img * (unsigned)(geo.tiles_x * geo.tiles_y);
        const unsigned tyi = div_magic(rem, geo.magic_tx);
        t_img = (int)img;
        t_y0 = (int)tyi * 16;
        t_x0 = (int)(rem - tyi * (unsigned)geo.tiles_x) * 16;
    };
    // DMA slots of this wave: 11 for the x patch (instruction n = wave + 4 s of 44, units 64 n .. 64 n + 63 of [324 pixels][8 chunk positions]),
    // 8 for the dy tile (n = wave + 4 s of 32, [256 pixels][8]).  Unit -> pixel P = unit >> 3 and chunk position cp = unit & 7, which holds
    // chunk cp ^ (((P >> 1) & 3) << 1).  Per lane and slot, tile-independent: the source offset relative to the tile's first pixel and the
    // pixel's (row, column) for the range check.
    int dx_rel[11], dx_pyx[11], dy_rel[8], dy_pyx[8];
#pragma unroll
    for (int s = 0; s < 11; s++) {
        const int u = (wave + 4 * s) * 64 + lane;
        const int P = u >> 3;
        const int py = (P * 3641) >> 16, px = P - py * 18;   // (P / 18, exact below 400)
        dx_rel[s] = (py * a.W + px) * 128 + (((u & 7) ^ (((P >> 1) & 3) << 1)) << 4);
        dx_pyx[s] = P < 324 ? (py | (px << 16)) : 0x7fff;      // (row + 32767: never inside an image)
    }
#pragma unroll
    for (int s = 0; s < 8; s++) {
        const int u = (wave + 4 * s) * 64 + lane;
        const int P = u >> 3;
        dy_rel[s] = ((P >> 4) * a.W + (P & 15)) * 128 + (((u & 7) ^ (((P >> 1) & 3) << 1)) << 4);
        dy_pyx[s] = (P >> 4) | ((P & 15) << 16);
    }
    int d_iy, d_ix, d_xb, d_yb;   // scratch of the three steps of a DMA instruction; d_xb / d_yb = byte offset of the patch's / tile's first pixel
    bool d_live;
    unsigned d_off;
    auto dma_a = [&](int s) {     // s < 11: x slot s; else dy slot s - 11
        if (s < 11) { d_iy = t_y0 - 1 + (dx_pyx[s] & 0xffff); d_ix = t_x0 - 1 + (dx_pyx[s] >> 16); }
        else { d_iy = t_y0 + (dy_pyx[s - 11] & 0xffff); d_ix = t_x0 + (dy_pyx[s - 11] >> 16); }
    };
    auto dma_b = [&](int s) {
        const bool ok = d_live && (unsigned)d_iy < (unsigned)a.H && (unsigned)d_ix < (unsigned)a.W;
        d_off = (unsigned)(s < 11 ? d_xb + dx_rel[s] : d_yb + dy_rel[s - 11]) | (ok ? 0u : kOOB);
    };
    auto dma_c = [&](int s, int buf) {
        if (s < 11) __builtin_amdgcn_raw_ptr_buffer_load_lds(xr, (lds_void_t *)(ldsb + buf * WD_STAGE + (wave + 4 * s) * 1024), 16, (int)d_off, 0, 0, 0);
        else __builtin_amdgcn_raw_ptr_buffer_load_lds(yr, (lds_void_t *)(ldsb + buf * WD_STAGE + WD_XBYTES + (wave + 4 * (s - 11)) * 1024), 16, (int)d_off, 0, 0, 0);
    };
    auto stage_tile = [&](int ti) {   // the tile whose operands the next dma_* calls fetch
        tile_coords(ti);
        d_live = ti < my_tiles;
        d_xb = ((t_img * a.H + t_y0 - 1) * a.W + t_x0 - 1) * 128;
        d_yb = ((t_img * a.H + t_y0) * a.W + t_x0) * 128;
    };

    // fragment addresses.  Lane (g, q, p) of a transposing read supplies pixel (tile row 2 s + (g >> 1), column 8 h + c4), c4 = 4 (g & 1) + q,
    // channels 4 p .. 4 p + 3 of the fragment's 16.  dy tile: pixel index 16 row + col, unit position j ^ ((c4 >> 1) & 3) -- one register
    // per co tile j, (s, h) in the immediate.  x patch: pixel (row + ty) 18 + col + tx, unit position w ^ ((row + ty + ((c4 + tx) >> 1)) & 3):
    // one register per (tx, m), m = (2 s + ty) & 3, the rest in the immediate.
    const int c4 = 4 * (g & 1) + q;
    int dya[4], xa[3][4];
#pragma unroll
    for (int j = 0; j < 4; j++) dya[j] = WD_XBYTES + ((g >> 1) * 16 + c4) * 128 + ((j ^ ((c4 >> 1) & 3)) << 5) + p * 8;
#pragma unroll
    for (int tx = 0; tx < 3; tx++)
#pragma unroll
        for (int m = 0; m < 4; m++)
            xa[tx][m] = ((g >> 1) * 18 + c4 + tx) * 128 + ((wave ^ (((g >> 1) + ((c4 + tx) >> 1) + m) & 3)) << 5) + p * 8;

    bt_f32x4 acc[9][4], accb;
#pragma unroll
    for (int t = 0; t < 9; t++)
#pragma unroll
        for (int j = 0; j < 4; j++)
#pragma unroll
            for (int e = 0; e < 4; e++) acc[t][j][e] = 0.f;
#pragma unroll
    for (int e = 0; e < 4; e++) accb[e] = 0.f;
    const u32x4 ones_u = u32x4{0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u};
    const bt_bf16x8 ones = __builtin_bit_cast(bt_bf16x8, ones_u);

    if (my_tiles > 0) {
        stage_tile(0);
#pragma unroll
        for (int s = 0; s < 19; s++) { dma_a(s); dma_b(s); dma_c(s, 0); }
        wait_vmcnt<0>();
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
    }
    for (int ti = 0; ti < my_tiles; ti++) {
        const int buf = ti & 1;
        stage_tile(ti + 1);
        const char *sb = ldsb + buf * WD_STAGE;
        // fragments of K step s: Df[j] = dy (co tile j), Xf[t] = x at tap t; two transposing reads each (pixels of read h = 0, 1)
        // (kept as 32-bit words: the compiler takes a shuffle of two 4 x bf16 vectors apart into sixteen-bit registers and back)
        u32x4 Df[2][4], Xf[2][9];
#define WD_SB __builtin_amdgcn_sched_barrier(0);
        // (inline: the compiler puts s_waitcnt vmcnt(0) in front of every LDS read it can see while this wave's LDS-DMA loads are in flight --
        // it cannot tell the stage being read from the stage being filled -- and the K loop then stops 19 times a tile for a whole memory round
        // trip: 1.27 ms.  Reads it does not see it does not wait for either: lgkmcnt(0) at the head of every K step, where the step's fragments,
        // requested during the step before, are due.)
        const unsigned sbase = (unsigned)(unsigned long long)sb;
#define WD_TR(DST, ADDR, IMM) asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(DST) : "v"(ADDR), "n"(IMM));
#define WD_RD(S, J)                                                                                                                     \
        {                                                                                                                               \
            const unsigned ad = sbase + (unsigned)dya[J];                                                                               \
            uint2 lo, hi;                                                                                                               \
            WD_TR(lo, ad, (S) * 4096) WD_TR(hi, ad, (S) * 4096 + 1024)                                                                  \
            Df[(S) & 1][J] = u32x4{lo.x, lo.y, hi.x, hi.y}; WD_SB                                                                        \
        }
#define WD_RX(S, T)                                                                                                                     \
        {                                                                                                                               \
            const unsigned ad = sbase + (unsigned)xa[(T) % 3][(2 * (S) + (T) / 3) & 3];                                                 \
            uint2 lo, hi;                                                                                                               \
            WD_TR(lo, ad, (2 * (S) + (T) / 3) * (18 * 128)) WD_TR(hi, ad, (2 * (S) + (T) / 3) * (18 * 128) + 1024)                      \
            Xf[(S) & 1][T] = u32x4{lo.x, lo.y, hi.x, hi.y}; WD_SB                                                                        \
        }
        WD_RD(0, 0) WD_RD(0, 1) WD_RD(0, 2) WD_RD(0, 3)
        WD_RX(0, 0) WD_RX(0, 1) WD_RX(0, 2) WD_RX(0, 3) WD_RX(0, 4) WD_RX(0, 5) WD_RX(0, 6) WD_RX(0, 7) WD_RX(0, 8)
#pragma unroll
        for (int s = 0; s < 8; s++) {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // this step's fragments have landed
            WD_SB
            // 37 MFMAs: (tap t, co tile j) in t-major order, the bias tile last; behind MFMA number n = 4 t + j comes one piece of:
            //   n = 0 .. 3: dy fragment n of step s + 1;  n = 4 + 3 t': x fragment t' of step s + 1 (t' = 0 .. 8: n = 4, 7, .., 28);
            //   DMA instruction d (of 19) of the next tile in steps at n = 5 + 3 k, 6 + 3 k, 8 + 3 k' ... : three instructions per K step
            //   (slots 3 s' .. of the step), the 19th in step 6
#pragma unroll
            for (int t = 0; t < 9; t++)
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    acc[t][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bt_bf16x8, Xf[s & 1][t]), __builtin_bit_cast(bt_bf16x8, Df[s & 1][j]), acc[t][j], 0, 0, 0);
                    WD_SB
                    const int n = 4 * t + j;
                    if (s + 1 < 8) {
                        if (n < 4) { WD_RD(s + 1, n) }
                        if (n >= 4 && n <= 28 && (n - 4) % 3 == 0) { WD_RX(s + 1, (n - 4) / 3) }
                    }
                    // DMA instruction d = 3 s + k (k = 0, 1, 2; d < 19) in three pieces behind MFMAs 5 + 10 k, 6 + 10 k, 8 + 10 k
                    if (n == 5 || n == 15 || n == 25) { const int d = 3 * s + (n - 5) / 10; if (d < 19) { dma_a(d); WD_SB } }
                    if (n == 6 || n == 16 || n == 26) { const int d = 3 * s + (n - 6) / 10; if (d < 19) { dma_b(d); WD_SB } }
                    if (n == 8 || n == 18 || n == 29) { const int d = 3 * s + (n == 29 ? 2 : (n - 8) / 10); if (d < 19) { dma_c(d, buf ^ 1); WD_SB } }
                }
            {   // (the wave's own dy fragment: a wave-uniform choice among four registers sets)
                const u32x4 dw_ = wave == 0 ? Df[s & 1][0] : wave == 1 ? Df[s & 1][1] : wave == 2 ? Df[s & 1][2] : Df[s & 1][3];
                accb = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ones, __builtin_bit_cast(bt_bf16x8, dw_), accb, 0, 0, 0);
            }
            WD_SB
        }
#undef WD_RX
#undef WD_RD
#undef WD_TR
#undef WD_SB
        wait_vmcnt<0>();                // the next tile's operands (this wave's part) have landed
        __builtin_amdgcn_s_barrier();   // .. everybody's have, and nobody reads this tile's stage any more
        asm volatile("" ::: "memory");
    }
    wait_vmcnt<0>();   // (the zero fill requested for the tile past the end lands in LDS this workgroup still owns)

    // ---- this workgroup's partial gradient -> its slab: acc[t][j][e] = (ci = 16 wave + 4 g + e, co = 16 j + (lane & 15)) of tap t
    float *out = a.out + (long long)bid * a.K * a.N;
#pragma unroll
    for (int t = 0; t < 9; t++)
#pragma unroll
        for (int j = 0; j < 4; j++)
#pragma unroll
            for (int e = 0; e < 4; e++) out[(t * 64 + 16 * wave + 4 * g + e) * 64 + 16 * j + (lane & 15)] = acc[t][j][e];
    if (a.bias_out && g == 0) a.bias_out[(long long)bid * a.N + 16 * wave + (lane & 15)] = accb[0];   // (every row of the tile holds the column sums)
}

inline int wgrad_d64_groups(const emadapt_conv_desc *d) {   // workgroups = slabs of the launch
    const long long tiles = (long long)d->B * ((d->W + 15) / 16) * ((d->H + 15) / 16);
    return (int)(tiles < plan_cus() ? tiles : plan_cus());
}

int launch_wgrad_d64(const WgradBf16Args &a, hipStream_t stream) {
    static DynLdsAttr attr;
    const int arc = attr.ensure(reinterpret_cast<const void *>(&wgrad_d64_kernel), WD_LDS);
    if (arc) return arc;
    D64Geo geo;
    const int B = a.M / (a.H * a.W);
    geo.tiles_x = (a.W + 15) / 16;
    geo.tiles_y = (a.H + 15) / 16;
    geo.tiles = B * geo.tiles_x * geo.tiles_y;
    geo.magic_txy = magic_u32(geo.tiles_x * geo.tiles_y);
    geo.magic_tx = magic_u32(geo.tiles_x);
    hipLaunchKernelGGL(wgrad_d64_kernel, dim3((unsigned)a.splits), dim3(256), WD_LDS, stream, a, geo);   // (a.splits = wgrad_d64_groups: one slab per workgroup)
    return launch_status();
}

}  // namespace

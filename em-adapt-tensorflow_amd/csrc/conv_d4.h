// Direct 3 x 3 convolution of the 4-channel f32 image into 64 bf16 channels (conv1_1's forward), included by conv_bf16.hip (round 5).
// The layer writes the net's largest tensor (16 x 321 x 321 x 64 bf16 = 211 MB) from its smallest (26 MB) through a GEMM of K = 36:
// 0.05 ms of HBM time -- and 0.17 ms on the implicit-GEMM kernel (12,881 workgroups whose 128 x 64 tiles are all epilogue).  Here:
//   * one workgroup (4 waves) per 16 x 16-pixel output tile, several workgroups per CU (22 KB of LDS each): the 18 x 18-pixel input patch
//     is read once with 16-byte loads (one pixel = one float4), rounded to bf16 and parked as [pixel][8 channels] (4 real + 4 zeros:
//     16 bytes, the layout of the filter image's k pitch);
//   * K = 9 taps x 8 channels = 3 MFMA steps of 32 (the last one a quarter full): the MFMA's 8-k lane group is ONE TAP, so a B fragment
//     (32 k x 16 pixels) is one ds_read_b128 per lane at pixel + the lane group's tap shift -- three address registers; the filter
//     (12 A fragments of 16 output channels x 32 k) sits in registers;
//   * transposed accumulators, bias / ReLU in fp32, an output row at a time through 2 KB of LDS per wave to whole 128-byte lines
//     (conv_d64.h's epilogue).
// Arithmetic: the image and the filter rounded to bf16, fp32 sums over (tap, channel) in one accumulator per output -- the model of the
// implicit-GEMM kernel with f32-stored input (another summation order: within the parity bar, not bit-identical); bit-reproducible.
// Requirements: 3 x 3, rate 1, 4 gathered channels stored f32 (the padded image), 64 output channels stored bf16, no dropout.
#pragma once

namespace {

constexpr int D4_PW = 18;
constexpr int D4_PATCH = D4_PW * D4_PW * 16;   // 5184 B
constexpr int D4_STAGE = D4_PATCH;             // [4 waves][16 pixels][128 B]
constexpr int D4_LDS = D4_STAGE + 4 * 2048;

inline bool d4_shape_ok(const IgemmArgs &a) {   // (plus, at the launch: x stored f32, y stored bf16, the forward entry, no dropout)
    if (a.ntaps != 9 || a.KW != 3 || a.dil != 1 || a.Cx != 4 || a.N != 64 || a.sign < 0) return false;
    if (a.M <= 0 || a.M % (a.H * a.W) != 0 || (long long)a.M * 128 >= (1LL << 40) || a.H > 16384 || a.W > 16384) return false;
    const long long per_img = (long long)((a.W + 15) / 16) * ((a.H + 15) / 16), tiles = (long long)(a.M / (a.H * a.W)) * per_img;
    return tiles <= 0x7fffffffLL && tiles * per_img < (1LL << 32);   // (grid size; the tile -> image division by multiply-high)
}

__global__ void __launch_bounds__(256) conv_d4_kernel(IgemmArgs a, D64Geo geo) {
    __shared__ __attribute__((aligned(16))) char ldsb[D4_LDS];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 15, g = lane >> 4;
    const int t = (int)blockIdx.x;
    const unsigned img = div_magic((unsigned)t, geo.magic_txy);
    const unsigned rem = (unsigned)t - img * (unsigned)(geo.tiles_x * geo.tiles_y);
    const unsigned tyi = div_magic(rem, geo.magic_tx);
    const int y0 = (int)tyi * 16, x0 = (int)(rem - tyi * (unsigned)geo.tiles_x) * 16;

    // ---- the patch: pixel (y0 - 1 + py, x0 - 1 + px) -> [4 bf16 channels, 4 zeros] at (py * 18 + px) * 16; outside the image: zeros
    for (int P = tid; P < D4_PW * D4_PW; P += 256) {
        const int py = (P * 3641) >> 16, px = P - py * D4_PW;   // (P / 18, exact below 400)
        const int iy = y0 - 1 + py, ix = x0 - 1 + px;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if ((unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W) v = *reinterpret_cast<const float4 *>(a.x + (((long long)img * a.H + iy) * a.W + ix) * 4);
        const uint2 c = cvt4(v);
        *reinterpret_cast<u32x4 *>(ldsb + P * 16) = u32x4{c.x, c.y, 0u, 0u};
    }
    // ---- the filter: fragment (s, j) = output channels 16 j + r, k group g = tap 4 s + g (its 8 channels; taps past the ninth: zeros)
    bt_bf16x8 F[3][4];
    {
        const char *wb = reinterpret_cast<const char *>(a.w);   // image [tap][64][8] bf16
#pragma unroll
        for (int s = 0; s < 3; s++) {
            const int tap = 4 * s + g;
#pragma unroll
            for (int j = 0; j < 4; j++) {
                u32x4 f = u32x4{0u, 0u, 0u, 0u};
                if (tap < 9) f = *reinterpret_cast<const u32x4 *>(wb + ((tap * 64 + j * 16 + r) * 8) * 2);
                F[s][j] = __builtin_bit_cast(bt_bf16x8, f);
            }
        }
    }
    float4 bj[4];
#pragma unroll
    for (int j = 0; j < 4; j++)
        bj[j] = (a.epi != EMADAPT_EPI_NONE && a.bias) ? make_float4(a.bias[j * 16 + 4 * g], a.bias[j * 16 + 4 * g + 1], a.bias[j * 16 + 4 * g + 2], a.bias[j * 16 + 4 * g + 3])
                                                      : make_float4(0.f, 0.f, 0.f, 0.f);
    // the lane group's tap shift inside the patch, per K step (taps past the ninth read tap 8's pixel: their filter is zero)
    int toff[3];
#pragma unroll
    for (int s = 0; s < 3; s++) {
        const int tap = 4 * s + g < 9 ? 4 * s + g : 8;
        toff[s] = ((tap / 3) * D4_PW + tap % 3) * 16;
    }
    __syncthreads();

    bt_f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; i++)
#pragma unroll
        for (int j = 0; j < 4; j++)
#pragma unroll
            for (int e = 0; e < 4; e++) acc[i][j][e] = 0.f;
#pragma unroll
    for (int s = 0; s < 3; s++) {
        bt_bf16x8 X[4];
#pragma unroll
        for (int i = 0; i < 4; i++) X[i] = *reinterpret_cast<const bt_bf16x8 *>(ldsb + ((4 * wave + i) * D4_PW + r) * 16 + toff[s]);
#pragma unroll
        for (int i = 0; i < 4; i++)
#pragma unroll
            for (int j = 0; j < 4; j++) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(F[s][j], X[i], acc[i][j], 0, 0, 0);
    }

    // ---- epilogue (conv_d64.h): acc[i][j][e] = pixel (row i, column r), channel 16 j + 4 g + e -> bias / ReLU, 4 bf16 = one 8-byte piece at
    // piece position (4 j + g) ^ r of pixel r's 128-byte row; read back as 16-byte chunks, 8 lanes per pixel: 1 KB of whole lines per store
    const bool relu = a.epi == EMADAPT_EPI_BIAS_RELU;
    __bf16 *yb = reinterpret_cast<__bf16 *>(a.y);
    char *stg = ldsb + D4_STAGE + wave * 2048;
    const int ocol = lane >> 3, och = lane & 7;
    const int oy0w = y0 + 4 * wave;
#pragma unroll
    for (int i = 0; i < 4; i++) {
#pragma unroll
        for (int j = 0; j < 4; j++) {
            float4 v = make_float4(acc[i][j][0] + bj[j].x, acc[i][j][1] + bj[j].y, acc[i][j][2] + bj[j].z, acc[i][j][3] + bj[j].w);
            if (relu) { v.x = relu_keep_nan(v.x); v.y = relu_keep_nan(v.y); v.z = relu_keep_nan(v.z); v.w = relu_keep_nan(v.w); }
            *reinterpret_cast<uint2 *>(stg + r * 128 + (((4 * j + g) ^ r) << 3)) = cvt4(v);
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
        for (int hh = 0; hh < 2; hh++) {
            const int col = 8 * hh + ocol;
            u32x4 o = *reinterpret_cast<const u32x4 *>(stg + col * 128 + ((och ^ (col >> 1)) << 4));
            if (col & 1) o = u32x4{o.z, o.w, o.x, o.y};
            const int oy = oy0w + i, oxx = x0 + col;
            if (oy < a.H && oxx < a.W) *reinterpret_cast<u32x4 *>(yb + (((long long)img * a.H + oy) * a.W + oxx) * 64 + och * 8) = o;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");   // (the row's reads are issued before the next row's pieces overwrite them)
        __builtin_amdgcn_wave_barrier();
    }
}

int launch_d4(IgemmArgs &a, hipStream_t stream) {
    D64Geo geo;
    const int B = a.M / (a.H * a.W);
    geo.tiles_x = (a.W + 15) / 16;
    geo.tiles_y = (a.H + 15) / 16;
    const long long tiles = (long long)B * geo.tiles_x * geo.tiles_y;
    if (tiles <= 0) return EMADAPT_OK;
    if (tiles > 0x7fffffffLL) return EMADAPT_ERR_UNSUPPORTED;
    geo.tiles = (int)tiles;
    geo.magic_txy = magic_u32(geo.tiles_x * geo.tiles_y);
    geo.magic_tx = magic_u32(geo.tiles_x);
    hipLaunchKernelGGL(conv_d4_kernel, dim3((unsigned)tiles), dim3(256), 0, stream, a, geo);
    return launch_status();
}

}  // namespace

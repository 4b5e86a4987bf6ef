// 1 x 1 data gradient with a SHORT reduction (at most 32 gathered channels) on bf16-stored tensors -- fc8's: dx[m][n] = gate(sum_k dy[m][k]
// w[n][k]), k = the 21 + 3 classes, n = fc7's 4096 (1024) channels -- included by conv_bf16.hip (round 5).  The launch is one MFMA step
// deep and moves 440 MB (the gate in, dx out): 0.09 ms of HBM time, 0.15 ms on the implicit-GEMM kernel, whose 256 x 256 tiles stage two
// operand tiles through LDS for a single K step.  Here nothing is staged: a wave's operands are 4 + 4 fragments that come straight from
// memory (16 bytes per lane each: dy rows of 48 bytes, filter rows of 48 bytes, both L2-resident), 16 MFMAs, and conv_d64.h's epilogue --
// transposed accumulators, gate scale in fp32, an output row of 16 pixels x 64 channels at a time through 2 KB of LDS to whole 128-byte
// lines, the gate fetched in that layout ahead of the MFMAs and applied to the rounded value (a select: exact).
// One workgroup = 64 pixels x 256 channels (4 waves x 64 channels).  Arithmetic: as the implicit-GEMM kernel -- one fp32 accumulator per
// output over the same products (the order inside an MFMA step is the hardware's: within the parity bar, bit-reproducible).
// Requirements: 1 x 1, gathered channels a multiple of 8 up to 32, output channels a multiple of 256, dy / dx / gate stored bf16, no bias.
#pragma once

namespace {

inline bool k32_shape_ok(const IgemmArgs &a) {   // (plus, at the launch: dy, dx and the gate stored bf16; the data-gradient entry)
    return a.ntaps == 1 && a.Cx % 8 == 0 && a.Cx <= 32 && a.N % 256 == 0 && a.M > 0 && (long long)((a.M + 63) / 64) * (a.N / 256) <= 0x7fffffffLL;
}

__global__ void __launch_bounds__(256) conv_k32_kernel(IgemmArgs a) {
    __shared__ __attribute__((aligned(16))) char ldsb[4 * 2048];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 15, g = lane >> 4;
    const int nb = a.N / 256;
    const int mt = (int)blockIdx.x / nb, nt = (int)blockIdx.x - mt * nb;   // (channel blocks of one pixel tile are neighbours: they share dy and the lines of dx)
    const int m0 = mt * 64, n0 = nt * 256 + wave * 64;
    const int Cxp = a.Cx;                       // k pitch of both operands (a multiple of 8: 16-byte pieces)
    const bool kvalid = g * 8 < Cxp;
    const char *xb = reinterpret_cast<const char *>(a.x), *wb = reinterpret_cast<const char *>(a.w);
    const __bf16 *gate = reinterpret_cast<const __bf16 *>(a.gate);
    __bf16 *yb = reinterpret_cast<__bf16 *>(a.y);
    const int ocol = lane >> 3, och = lane & 7;   // whole-line layout: a lane = the 16-byte chunk och of pixel 8 hh + ocol of a 16-pixel row

    u32x4 gv[8];
    if (gate) {
#pragma unroll
        for (int kk = 0; kk < 8; kk++) {
            const int m = m0 + (kk >> 1) * 16 + 8 * (kk & 1) + ocol;
            gv[kk] = m < a.M ? *reinterpret_cast<const u32x4 *>(gate + (long long)m * a.N + n0 + och * 8) : u32x4{0u, 0u, 0u, 0u};
        }
    }
    // A fragments: filter rows n0 + 16 j + r (image [N][Cxp], k contiguous); B fragments: pixels m0 + 16 i + r (dy [M][Cxp]); k group g = channels 8 g ..
    bt_bf16x8 F[4], X[4];
#pragma unroll
    for (int j = 0; j < 4; j++) {
        u32x4 f = u32x4{0u, 0u, 0u, 0u};
        if (kvalid) f = *reinterpret_cast<const u32x4 *>(wb + ((long long)(n0 + j * 16 + r) * Cxp + g * 8) * 2);
        F[j] = __builtin_bit_cast(bt_bf16x8, f);
    }
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const int m = m0 + i * 16 + r;
        u32x4 f = u32x4{0u, 0u, 0u, 0u};
        if (kvalid && m < a.M) f = *reinterpret_cast<const u32x4 *>(xb + ((long long)m * Cxp + g * 8) * 2);
        X[i] = __builtin_bit_cast(bt_bf16x8, f);
    }
    bt_f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; i++)
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const bt_f32x4 z = {0.f, 0.f, 0.f, 0.f};
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(F[j], X[i], z, 0, 0, 0);
        }
    // epilogue (conv_d64.h): acc[i][j][e] = pixel m0 + 16 i + r, channel n0 + 16 j + 4 g + e
    char *stg = ldsb + wave * 2048;
    const float gscale = gate ? a.scale : 1.f;
#pragma unroll
    for (int i = 0; i < 4; i++) {
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const float4 v = make_float4(acc[i][j][0] * gscale, acc[i][j][1] * gscale, acc[i][j][2] * gscale, acc[i][j][3] * gscale);
            *reinterpret_cast<uint2 *>(stg + r * 128 + (((4 * j + g) ^ r) << 3)) = cvt4(v);
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
        for (int hh = 0; hh < 2; hh++) {
            const int kk = 2 * i + hh, col = 8 * hh + ocol;
            u32x4 o = *reinterpret_cast<const u32x4 *>(stg + col * 128 + ((och ^ (col >> 1)) << 4));
            if (col & 1) o = u32x4{o.z, o.w, o.x, o.y};
            if (gate) o = u32x4{d64_gate_pk(o.x, gv[kk].x), d64_gate_pk(o.y, gv[kk].y), d64_gate_pk(o.z, gv[kk].z), d64_gate_pk(o.w, gv[kk].w)};
            const int m = m0 + i * 16 + col;
            if (m < a.M) *reinterpret_cast<u32x4 *>(yb + (long long)m * a.N + n0 + och * 8) = o;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");   // (the row's reads are issued before the next row's pieces overwrite them)
        __builtin_amdgcn_wave_barrier();
    }
}

int launch_k32(IgemmArgs &a, hipStream_t stream) {
    const long long g = (long long)((a.M + 63) / 64) * (a.N / 256);
    hipLaunchKernelGGL(conv_k32_kernel, dim3((unsigned)g), dim3(256), 0, stream, a);
    return launch_status();
}

}  // namespace

// bf16x6 on pre-split operands: filter-gradient kernel, included by conv_x6p.hip.
//
//   dw[k][n] = sum over pixels m of x[m + tap(k)][c(k)] * dy[m][n]      (k = tap * Cin + c)
// from the plane images of x and dy (igemm_x6p.h: [pixel][channel / 16][plane][16 bf16]).  Built like wgrad2_kernel
// (conv_wgrad.hip) and igemm_x6p_kernel: 256(k) x 256(n) filter tiles, one workgroup of 8 waves per CU, every wave a 64 x 128
// accumulator block and its share of the DMAs; 16 pixels per reduction step (one bf16 MFMA depth), ring of 3 stages; the tile's
// 256 k are one tap's channels (Cin % 256 == 0), so a pixel's validity and offsets are scalars; pixel skipping over the taps'
// boxes as in wgrad2_kernel.
//
// LDS image of a step, per operand: 16 pixel rows of 1536 bytes (256 channels x 6 B: 16 groups x 3 planes x 32 B), exactly as in
// memory.  Wave w fetches pixels 2w and 2w + 1 (3 DMA instructions: 3072 contiguous bytes in LDS) and a 64-byte gap follows every
// such pair.  Both operands have the reduction index (the pixel) as the slow one, so fragments are gathered with
// ds_read_b64_tr_b16: 16 lanes = 4 pixels x 32 bytes.  Which pixel feeds which k slot of the MFMA is free as long as both operands
// agree, and is chosen so that the 4 pixels of one read come from 4 DIFFERENT pairs (slot (half, e, q) = pixel 2 (4 half + q) + e):
// their rows then start 64 bytes apart modulo 256 and, with the neighbouring channel group 96 bytes on, the 32 lanes of a read
// cover all 64 banks once -- no swizzle, one address register per operand, everything else immediates.
#pragma once
#include "igemm_x6p.h"

namespace emadapt {
int launch_slab_sum(const float *slab, int nslab, long long n, float *out, hipStream_t stream);
}

namespace {

struct WgradXPArgs {
    const char *xp, *dyp;   // plane images of x [M][Cin] and dy [M][N]
    float *out;             // dw or slab base: [splits][K][N]
    int H, W, Cin, N, KW, dil, pad_h, pad_w;
    int M, K, ntaps;
    int tiles_k, tiles_n, splits;
    int rows_per_split;     // multiple of 16
    unsigned x_bytes, y_bytes;
    unsigned magic_w, magic_h;
    int skip;               // walk only the pixels at which the tile's tap is inside the image
    int lockstep;           // 1: every wave issues its DMA burst at the same point (tune "dma_stagger" = 1)
    unsigned char korder[256];
};

constexpr int WP_PAIR = 2 * 1536 + 64;      // two pixel rows + gap
constexpr int WP_OPER = 8 * WP_PAIR;        // 25088: one operand's step
constexpr int WP_STAGE = 2 * WP_OPER;
constexpr int WP_NBUF = 3;
constexpr size_t WP_LDS = (size_t)WP_NBUF * WP_STAGE;   // 150528

__device__ __forceinline__ xp_bf16x8 wp_tr_frag(const char *lo_p) {
    typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
    typedef __attribute__((address_space(3))) bf16x4 lds_bf16x4;
    const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4 *)(lo_p));
    const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4 *)(lo_p + 1536));   // the pair's second pixel
    return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
}

// COMPACT: walk only the pixels inside the tap's box (a.skip).  The loaders' scalar arithmetic is branch-free: it runs between
// the MFMAs of waves that have nobody else to issue them.
template <bool COMPACT>
__global__ void __launch_bounds__(512, 2) wgrad_x6p_kernel(WgradXPArgs a) {
    constexpr int BMR = 16, NBUF = WP_NBUF, TK = 2, TN = 4;
    extern __shared__ __attribute__((aligned(16))) char ldsb[];

    // block order: wgrad2_kernel's
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
    if constexpr (COMPACT) {   // tiles sorted by work, heaviest first (wgrad2_kernel)
        const int per_k = a.tiles_n * a.splits;
        const int kr = (int)blockIdx.x / per_k, r = (int)blockIdx.x - kr * per_k;
        tile_k = a.korder[kr];
        tile_n = r / a.splits;
        split = r - tile_n * a.splits;
    }
    const int k0 = tile_k * 256, n0 = tile_n * 256;
    const int tap_s = k0 / a.Cin;
    const int c0_s = k0 - tap_s * a.Cin;
    const int dh_s = (tap_s / a.KW) * a.dil - a.pad_h, dw_s = (tap_s % a.KW) * a.dil - a.pad_w;
    constexpr bool compact = COMPACT;
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
    const int rps = compact ? (int)((((long long)m_total + a.splits - 1) / a.splits + BMR - 1) / BMR) * BMR : a.rows_per_split;
    const int m_begin = __builtin_amdgcn_readfirstlane(split * rps);
    int m_end = m_begin + rps;
    if (m_end > m_total) m_end = m_total;
    m_end = __builtin_amdgcn_readfirstlane(m_end);
    const int nsteps = (m_end > m_begin) ? (m_end - m_begin + BMR - 1) / BMR : 0;

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wk = wave & 3, wn = wave >> 2;
    const bool late = wave >= 4 && !a.lockstep;   // (the SIMD partner of wave w is wave w + 4)

    // ---- loader side
    const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc(const_cast<char *>(a.xp), 0, (int)a.x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t yr = __builtin_amdgcn_make_buffer_rsrc(const_cast<char *>(a.dyp), 0, (int)a.y_bytes, 0x00020000);
    constexpr unsigned kOOB = 0x80000000u;
    const unsigned magic_pv = compact && pv > 1 ? (unsigned)__builtin_amdgcn_readfirstlane((int)((0x100000000ULL + (unsigned)pv - 1) / (unsigned)pv)) : 0u;
    const unsigned magic_vw = compact && vw > 1 ? (unsigned)__builtin_amdgcn_readfirstlane((int)((0x100000000ULL + (unsigned)vw - 1) / (unsigned)vw)) : 0u;
    const int xrow = a.Cin / 16 * XP_ROWB, yrow = a.N / 16 * XP_ROWB;   // bytes per pixel in the plane images
    // per-lane constants of the three instructions of a pixel pair: byte inside the pixel's 1536-byte block
    //   0: first pixel [0, 1024)   1: first pixel [1024, 1536) | second pixel [0, 512)   2: second pixel [512, 1536)
    const unsigned lc0 = (unsigned)lane * 16u, lc1 = lane < 32 ? 1024u + (unsigned)lane * 16u : (unsigned)(lane - 32) * 16u,
                   lc2 = 512u + (unsigned)lane * 16u;
    // dy: channels past N (a last, partial n tile) are out of range
    // (kept apart from the byte offsets: a dead pixel's offset carries the bit too, and the sum of two would wrap into range)
    auto ybad = [&](unsigned lc) { return n0 + (int)(lc / XP_ROWB) * 16 < a.N ? 0u : kOOB; };
    const unsigned yb0 = ybad(lc0), yb1 = ybad(lc1), yb2 = ybad(lc2);
    const bool second = lane >= 32;

    // n / d by multiply-high with d == 1 encoded as magic 0 -- as a select, not a branch
    auto udiv = [](unsigned n, unsigned magic) { return __umulhi(n, magic) + (magic ? 0u : n); };
    // pixel i of this split -> scalar byte offsets of its x block (at the tile's tap) and its dy block, each with the
    // out-of-range bit where the pixel is past the end (resp. the tap outside the image)
    auto pixel = [&](int i, unsigned &vx, unsigned &vy) {
        const unsigned ui = (unsigned)i;
        unsigned m, oh, ow;
        if constexpr (COMPACT) {
            const unsigned img = udiv(ui, magic_pv);
            const unsigned r = ui - img * (unsigned)pv;
            const unsigned rh = udiv(r, magic_vw);
            oh = (unsigned)h0 + rh;
            ow = (unsigned)w0 + (r - rh * (unsigned)vw);
            m = (img * (unsigned)a.H + oh) * (unsigned)a.W + ow;
        } else {
            m = ui;
            const unsigned t2 = udiv(m, a.magic_w);
            ow = m - t2 * (unsigned)a.W;
            oh = t2 - udiv(t2, a.magic_h) * (unsigned)a.H;
        }
        const bool live = i < m_end;
        const bool ok = live & ((unsigned)((int)oh + dh_s) < (unsigned)a.H) & ((unsigned)((int)ow + dw_s) < (unsigned)a.W);
        vx = ok ? (unsigned)((int)m + dh_s * a.W + dw_s) * (unsigned)xrow + (unsigned)(c0_s / 16 * XP_ROWB) : kOOB;
        vy = live ? m * (unsigned)yrow + (unsigned)(n0 / 16 * XP_ROWB) : kOOB;
    };
    // DMAs of reduction step st into the ring buffer at byte offset `off`: this wave's pixels 2 wave (first half) and 2 wave + 1
    // (second half; the middle instruction of each operand serves both pixels, so it goes with the second)
    unsigned vxa = 0, vya = 0;
    auto fill_first = [&](int off, int st) {
        pixel(__builtin_amdgcn_readfirstlane(m_begin + st * BMR + 2 * wave), vxa, vya);
        char *xs = ldsb + off + wave * WP_PAIR;
        xp_dma(xr, xs, vxa + lc0, 0);
        xp_dma(yr, xs + WP_OPER, (vya + lc0) | yb0, 0);
    };
    auto fill_second = [&](int off, int st) {
        unsigned vxb, vyb;
        pixel(__builtin_amdgcn_readfirstlane(m_begin + st * BMR + 2 * wave + 1), vxb, vyb);
        char *xs = ldsb + off + wave * WP_PAIR;
        xp_dma(xr, xs + 1024, (second ? vxb : vxa) + lc1, 0);
        xp_dma(xr, xs + 2048, vxb + lc2, 0);
        xp_dma(yr, xs + WP_OPER + 1024, ((second ? vyb : vya) + lc1) | yb1, 0);
        xp_dma(yr, xs + WP_OPER + 2048, (vyb + lc2) | yb2, 0);
    };
    auto fill = [&](int off, int st) { fill_first(off, st); fill_second(off, st); };

    // ---- consumer side: one fragment address per operand (bytes inside a stage)
    const int grp = lane >> 4, q = (lane >> 2) & 3, p = lane & 3;
    const int l31 = lane & 31, lh = lane >> 5;
    const int slot = (4 * (grp >> 1) + q) * WP_PAIR + (grp & 1) * XP_ROWB + p * 8;
    const int x_frag = slot + wk * 4 * XP_ROWB;
    const int y_frag = WP_OPER + slot + wn * 8 * XP_ROWB;
    f32x16 acc[TK][TN];
    xp_bf16x8 av[3][TK], bv[3][TN];
#pragma unroll
    for (int i = 0; i < TK; i++)
#pragma unroll
        for (int j = 0; j < TN; j++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[i][j][r] = 0.f;

#define WP_SB __builtin_amdgcn_sched_barrier(0);
#define WP_RA(P, I, BASE) av[P][I] = wp_tr_frag(ldsb + (BASE) + x_frag + (P) * 32 + (I) * (2 * XP_ROWB)); WP_SB
#define WP_RB(Q, J, BASE) bv[Q][J] = wp_tr_frag(ldsb + (BASE) + y_frag + (Q) * 32 + (J) * (2 * XP_ROWB)); WP_SB
#define WP_MF(P, Q, I, J) acc[I][J] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av[P][I], bv[Q][J], acc[I][J], 0, 0, 0); WP_SB
#define WP_GROUP(P, Q) WP_MF(P, Q, 0, 0) WP_MF(P, Q, 0, 1) WP_MF(P, Q, 0, 2) WP_MF(P, Q, 0, 3) WP_MF(P, Q, 1, 0) WP_MF(P, Q, 1, 1) WP_MF(P, Q, 1, 2) WP_MF(P, Q, 1, 3)
    // the software pipeline of igemm_x6p_kernel: six groups (0,2) (0,1) (0,0) | (1,0) (2,0) (1,1), barrier after the third, the next
    // step's fragments fetched into the registers this step has finished with, the DMAs of step st + 2 as one burst in front of
    // the barrier (into the buffer of step st - 1)
    auto k_step = [&](int NX, int PV, int st) {
        WP_SB
        WP_GROUP(0, 2) WP_GROUP(0, 1) WP_GROUP(0, 0)
        // (round 4: the SIMD partners' bursts staggered around the barrier as in igemm_x6p_kernel -- waves 0-3 in front, 4-7 behind)
        if (!late) fill(PV, st + 2);   // (steps past the end: both pixels dead, zeros)
        WP_SB
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        // this wave's DMAs of step st + 1 have landed (waves 0-3: those of st + 2 have just left)
        if (late) wait_vmcnt<0>(); else wait_vmcnt<6>();
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if (late) fill(PV, st + 2);
        WP_SB
        WP_MF(1, 0, 0, 0) WP_RA(0, 0, NX) WP_MF(1, 0, 0, 1) WP_RA(0, 1, NX) WP_MF(1, 0, 0, 2) WP_RB(2, 0, NX) WP_MF(1, 0, 0, 3) WP_RB(2, 1, NX)
        WP_MF(1, 0, 1, 0) WP_RB(2, 2, NX) WP_MF(1, 0, 1, 1) WP_RB(2, 3, NX) WP_MF(1, 0, 1, 2) WP_MF(1, 0, 1, 3)
        WP_GROUP(2, 0)
        WP_MF(1, 1, 0, 0) WP_RB(0, 0, NX) WP_MF(1, 1, 0, 1) WP_RB(0, 1, NX) WP_MF(1, 1, 0, 2) WP_RB(0, 2, NX) WP_MF(1, 1, 0, 3) WP_RB(0, 3, NX)
        WP_MF(1, 1, 1, 0) WP_RA(2, 0, NX) WP_MF(1, 1, 1, 1) WP_RA(2, 1, NX) WP_MF(1, 1, 1, 2) WP_MF(1, 1, 1, 3)
        WP_RB(1, 0, NX) WP_RB(1, 1, NX) WP_RB(1, 2, NX) WP_RB(1, 3, NX) WP_RA(1, 0, NX) WP_RA(1, 1, NX)
    };

    if (nsteps > 0) {
        fill(0, 0);
        fill(WP_STAGE, 1);
        wait_vmcnt<6>();
        __builtin_amdgcn_s_barrier();   // step 0 has landed
        asm volatile("" ::: "memory");
        WP_RA(0, 0, 0) WP_RA(0, 1, 0) WP_RB(2, 0, 0) WP_RB(2, 1, 0) WP_RB(2, 2, 0) WP_RB(2, 3, 0)
        WP_RB(1, 0, 0) WP_RB(1, 1, 0) WP_RB(1, 2, 0) WP_RB(1, 3, 0) WP_RB(0, 0, 0) WP_RB(0, 1, 0) WP_RB(0, 2, 0) WP_RB(0, 3, 0)
        WP_RA(1, 0, 0) WP_RA(1, 1, 0) WP_RA(2, 0, 0) WP_RA(2, 1, 0)
        int u_off = 0, pv = (NBUF - 1) * WP_STAGE;
        for (int st = 0; st < nsteps; st++) {
            const int nx = u_off == (NBUF - 1) * WP_STAGE ? 0 : u_off + WP_STAGE;
            k_step(nx, pv, st);
            pv = u_off;
            u_off = nx;
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        wait_vmcnt<0>();
    }
#undef WP_SB
#undef WP_RA
#undef WP_RB
#undef WP_MF
#undef WP_GROUP

    // tile row of the 32x32 MFMA = (r&3) + 8*(r>>2) + 4*lh -> filter row k0 + wk*64 + i*32 + row; column n0 + wn*128 + j*32 + l31
    float *out = a.out + (long long)split * a.K * a.N;
#pragma unroll
    for (int i = 0; i < TK; i++)
#pragma unroll
        for (int j = 0; j < TN; j++) {
            const int col = n0 + wn * 128 + j * 32 + l31;
            const bool col_ok = col < a.N;
#pragma unroll
            for (int r = 0; r < 16; r++) {
                const int row = k0 + wk * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                if (col_ok && row < a.K) out[(long long)row * a.N + col] = acc[i][j][r];
            }
        }
}

struct PlanWP {
    int tiles_k, tiles_n, splits, rows_per_split;
};

// Shapes the kernel takes: a 256-wide k tile inside one tap, whole groups of 16 output channels, plane images below 2 GiB.
bool wp_ok(const emadapt_conv_desc *d) {
    const long long M = (long long)d->B * d->H * d->W;
    return d->Cin % 256 == 0 && d->Cout % 16 == 0 && d->Cout >= 128 && d->KH * d->KW * (d->Cin / 256) <= 256 &&
           M * d->Cin * 6 < (1LL << 31) && M * d->Cout * 6 < (1LL << 31) && M * (d->W > d->H ? d->W : d->H) < (1LL << 32) &&
           M * d->H * d->W < (1LL << 32);
}

// Split-M plan (conv_wgrad.hip's, for one workgroup per CU).
PlanWP plan_wgrad_x6p(const emadapt_conv_desc *d) {
    PlanWP p;
    const long long M = (long long)d->B * d->H * d->W;
    const int K = d->KH * d->KW * d->Cin;
    p.tiles_k = (int)ceil_div(K, 256);
    p.tiles_n = (int)ceil_div(d->Cout, 256);
    const long long tiles = (long long)p.tiles_k * p.tiles_n;
    const double t_kernel = 2.0 * (double)M * (p.tiles_k * 256.0) * (p.tiles_n * 256.0) / 230e12;
    const double kn_bytes = (double)K * d->Cout * 4.0;
    long long best = 1;
    double best_t = 1e30;
    const long long max_s = ceil_div(M, 256) < 2048 ? ceil_div(M, 256) : 2048;
    const double slots = plan_cus();
    for (long long s = 1; s <= max_s; s++) {
        const double rounds = (double)tiles * s / slots;
        const double eff = rounds / (double)(long long)(rounds + 0.999999);
        double t = t_kernel / eff + 2e-6 * (double)(long long)(rounds + 0.999999);
        if (s > 1) t += (double)(s + 1) * kn_bytes / 3e12 + 3e-6;
        if (t < best_t * 0.995) { best_t = t; best = s; }
        if (rounds > 24.0) break;
    }
    long long rps = ceil_div(M, best);
    rps = ceil_div(rps, 16) * 16;
    p.rows_per_split = (int)rps;
    p.splits = (int)ceil_div(M, rps);
    return p;
}

void wp_order_k_tiles(WgradXPArgs &a) {
    const int nt = a.tiles_k;
    int work[256], idx[256];
    for (int t = 0; t < nt; t++) {
        const int tap = t * 256 / a.Cin;
        const int dh = (tap / a.KW) * a.dil - a.pad_h, dw = (tap % a.KW) * a.dil - a.pad_w;
        const int vh = a.H - (dh < 0 ? -dh : dh), vw = a.W - (dw < 0 ? -dw : dw);
        work[t] = (vh > 0 ? vh : 0) * (vw > 0 ? vw : 0);
        idx[t] = t;
    }
    for (int i = 1; i < nt; i++) {
        const int v = idx[i];
        int j = i - 1;
        while (j >= 0 && work[idx[j]] < work[v]) { idx[j + 1] = idx[j]; j--; }
        idx[j + 1] = v;
    }
    for (int i = 0; i < nt; i++) a.korder[i] = (unsigned char)idx[i];
}

int launch_wgrad_x6p(const WgradXPArgs &a, hipStream_t stream) {
    const long long nblk = (long long)a.tiles_k * a.tiles_n * a.splits;
    if (nblk > 0x7fffffffLL) return EMADAPT_ERR_UNSUPPORTED;
    if (a.skip) {
        static DynLdsAttr attr;
        int rc = attr.ensure(reinterpret_cast<const void *>(&wgrad_x6p_kernel<true>), WP_LDS);
        if (rc) return rc;
        hipLaunchKernelGGL(wgrad_x6p_kernel<true>, dim3((unsigned)nblk), dim3(512), WP_LDS, stream, a);
    } else {
        static DynLdsAttr attr;
        int rc = attr.ensure(reinterpret_cast<const void *>(&wgrad_x6p_kernel<false>), WP_LDS);
        if (rc) return rc;
        hipLaunchKernelGGL(wgrad_x6p_kernel<false>, dim3((unsigned)nblk), dim3(512), WP_LDS, stream, a);
    }
    return launch_status();
}

}  // namespace

// bf16 forward / data-gradient kernel on LDS-DMA ("b16t": bf16 operands, Tiled filter image), included by conv_bf16.hip.
//
// Round 1-3's bf16 kernel (igemm_bf16_kernel) moves its operands global -> registers -> LDS with four producer waves beside
// eight MFMA waves and sits at 43-55 % of the dense bf16 matrix rate: producers alone and consumers alone each need ~55 % of
// the kernel's time and overlap by a fifth (profiles/r03_bf16_ablation.txt) -- the ds_write_b128 path (13 cycles of
// VGPR -> LDS transfer per KB), the producers' issue slots and 192 registers of staging are all spent on bytes that need no
// arithmetic.  This kernel is the plane-image kernel of the bf16x6 path (igemm_x6p.h) with ONE plane: a bf16 NHWC activation
// already IS an image whose K-step slices are contiguous runs of a row, so the tiles land in LDS by LDS-DMA exactly as they lie
// in memory and no wave issues a vector instruction per element.
//
//   * 256 x 256 x 32 tiles, ONE workgroup of 8 waves per CU, every wave a 64 x 128 accumulator block (TM = 2, TN = 4: 16 MFMAs
//     32x32x16 per K step from 12 ds_read_b128) and 4 of the step's 32 DMA instructions; no loader waves.
//   * LDS image of an operand's K step: [256 rows][64 B] dense (16 KB); ring of 4 stages x 2 operands = 128 KB, fetched 3 steps
//     ahead.  (K steps of 64 -- 128-byte row pieces, a whole cache line -- allow only two 64 KB stages: round 2 measured that
//     form at the mercy of the DMA latency.)  A DMA instruction moves 64 consecutive 16-byte slots = 16 rows.
//   * bank conflicts: the 16 lanes of a ds_read_b128 group read one 16-byte column of 16 rows 64 bytes apart -- a quarter of the
//     banks -- so slot s of row r holds the row's chunk s ^ ((r >> 2) & 3) (the DMA lanes fetch the chunks in that order, the
//     fragment address applies the same XOR): the four row quads of a group land on four different slot columns.
//   * the FILTER is stored tile by tile in exactly that LDS order (filter_tiles_b16_kernel, once per optimizer step):
//         [tap][k / 32][row, zero rows up to a multiple of 256][4 slots of 16 B, swizzled]
//     so a K step's 256 filter rows are 16 contiguous KB and the filter DMAs are plain 1 KB copies.
//   * pipeline (k_step): two groups of 8 MFMAs per step (k 0..15, k 16..31).  The barrier sits between them; behind it the
//     fragments of the NEXT step's first group are fetched into the registers the first group has just finished with, one read
//     behind each MFMA of the second group, and the next step's second group is fetched behind the step's last MFMA (in flight
//     while the next first group runs).  The four DMAs of step kt + 3 leave as one burst in front of the barrier, into the
//     stage of step kt - 1 (free since the previous barrier).
//   * work split: plain tiles (XCD-aware order, optional split-K into slab levels) or stream-K (SK: the K steps of all tiles cut
//     into one piece per CU; partial tiles go to slab levels and streamk_fixup_t_kernel adds them in workgroup order).
// Requirements: activations stored bf16 (x_dtype = EMADAPT_DT_BF16), Cx % 32 == 0, KH*KW <= 32, N % 8 == 0 for bf16 outputs,
// images below 2 GiB.
#pragma once
#include "igemm_v2.h"

namespace {

typedef __bf16 bt_bf16x8 __attribute__((ext_vector_type(8)));

constexpr int BT_BK = 32;
constexpr int BT_ROWB = 2 * BT_BK;             // bytes of one row's K step, in memory and in LDS
constexpr int BT_OPER = 256 * BT_ROWB;         // one operand's K step in LDS (16 KB)
constexpr int BT_STAGE = 2 * BT_OPER;          // A + B
constexpr int BT_NBUF = 4;
constexpr int BT_WAVES = 8;
constexpr int BT_IPW = BT_OPER / 1024 / BT_WAVES;   // DMA instructions per wave, operand and K step (2)
constexpr size_t BT_LDS = (size_t)BT_NBUF * BT_STAGE;

// ------------------------------------------------------------------------------------------------------------------
// Tiled bf16 image of a filter: [tap][k / 32][row (zero rows up to a multiple of 256)][4 slots], slot s of row r = the row's
// k chunk s ^ ((r >> 2) & 3) (8 bf16 each).  DGRAD = false: rows = Cout, k = Cin (forward); true: rows = Cin, k = Cout (data
// gradient; k beyond the real channel count reads zero).  One thread = one 16-byte slot.
// ------------------------------------------------------------------------------------------------------------------
template <bool DGRAD>
__global__ void __launch_bounds__(256) filter_tiles_b16_kernel(const float *__restrict__ w, uint4 *__restrict__ out, int Cin, int Cout,
                                                               int kpad, int taps, int rows_pad) {
    const int rows = DGRAD ? Cin : Cout, kdim = DGRAD ? Cout : Cin;
    const int G = kpad / BT_BK;
    const long long total = (long long)taps * G * rows_pad * 4;
    for (long long idx = (long long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long long)gridDim.x * 256) {
        const int slot = (int)(idx & 3);
        long long t = idx >> 2;
        const int row = (int)(t % rows_pad);
        t /= rows_pad;
        const int g = (int)(t % G), tap = (int)(t / G);
        const int chunk = slot ^ ((row >> 2) & 3);
        bt_bf16x8 v;
#pragma unroll
        for (int e = 0; e < 8; e++) {
            const int k = g * BT_BK + chunk * 8 + e;
            const float f = (row < rows && k < kdim) ? (DGRAD ? w[((long long)tap * Cin + row) * Cout + k] : w[((long long)tap * Cin + k) * Cout + row]) : 0.f;
            v[e] = (__bf16)f;
        }
        out[idx] = *reinterpret_cast<uint4 *>(&v);
    }
}

// Stream-K second stage for bf16-storage outputs (the f32 kernels -- igemm_v2.h, igemm_x6p.h -- combine their shared tiles inside the launch since round 4): the tiles that more than one
// workgroup worked on.  One block = 16 rows of one 256 x 256 tile.
template <bool YBF, int BM = 256, int BN = 256>
__global__ void __launch_bounds__(256) streamk_fixup_t_kernel(IgemmArgs a, int G, int nk_all) {
    constexpr int PARTS = BM / 16, QN = BN / 4, ITS = 16 * QN / 256;   // one block = 16 rows of one BM x BN tile
    static_assert(ITS >= 1 && 16 * QN % 256 == 0, "tile shape");
    const int t = blockIdx.x / PARTS, part = blockIdx.x % PARTS;
    const int tile_m = t / a.tiles_n, tile_n = t - tile_m * a.tiles_n;
    const long long U = (long long)a.tiles_m * a.tiles_n * nk_all, u0 = (long long)t * nk_all;
    const int vf = sk_owner(u0, U, G), vl = sk_owner(u0 + nk_all - 1, U, G);
    const int cnt = vl - vf + 1;
    if (cnt <= 1) return;
    const long long mn = (long long)(a.m_end - a.m_begin) * a.N;
#pragma unroll
    for (int it = 0; it < ITS; it++) {
        const int idx = it * 256 + threadIdx.x;
        const int r = part * 16 + idx / QN;
        const int row = a.m_begin + tile_m * BM + r;
        const int col = tile_n * BN + (idx % QN) * 4;
        if (row >= a.m_end || col >= a.N) continue;
        const long long o = (long long)row * a.N + col;
        const float *sp = a.slab + (o - (long long)a.m_begin * a.N);
        float4 v = *reinterpret_cast<const float4 *>(sp);
        for (int l = 1; l < cnt; l++) {
            const float4 u = *reinterpret_cast<const float4 *>(sp + l * mn);
            v.x += u.x; v.y += u.y; v.z += u.z; v.w += u.w;
        }
        v.x = epilogue_t<YBF>(a, v.x, col + 0, o + 0);
        v.y = epilogue_t<YBF>(a, v.y, col + 1, o + 1);
        v.z = epilogue_t<YBF>(a, v.z, col + 2, o + 2);
        v.w = epilogue_t<YBF>(a, v.w, col + 3, o + 3);
        if constexpr (YBF) *reinterpret_cast<uint2 *>(reinterpret_cast<__bf16 *>(a.y) + o) = cvt4(v);
        else *reinterpret_cast<float4 *>(a.y + o) = v;
    }
}

// ABLM (diagnostic library only, tune "ablate"): parts of the K loop COMPILED out -- 1 no DMAs, 2 no fragment reads, 4 no waits /
// barrier, 8 no MFMAs, 16 no barrier (waits stay); results are garbage, scripts/bf16t_ablate.py reads the time.
// Tile shapes (round 4, second half): the 8 waves are WMW x (8 / WMW) blocks of 64 x (32 TN) accumulators:
//     WMW 4, TN 4: 256 x 256 (the default);  WMW 8, TN 4: 512 x 128;  WMW 8, TN 2: 512 x 64
// The tall tiles serve the layers with 128 or 64 output columns (conv1_2, conv2_x, the data gradients into 64 / 128 channels): the
// round-1 kernel runs them on 128 x 128 / 128 x 64 tiles with a two-stage pipeline, whose 9-18 K steps per tile each wait out a
// memory round trip (conv1_2: 0.30 ms against an HBM floor of 0.08); here the ring runs three steps ahead.  A stage is
// [BM rows][64 B] of activations + [BN rows][64 B] of filter; with BN = 64 only waves 0-3 have filter rows to fetch -- the others
// issue the same instruction with every lane out of range into a scratch KB behind the ring, so that all waves count alike.
template <int WMW, int TN>
struct BtShape {
    static constexpr int WNW = BT_WAVES / WMW, BM = WMW * 64, BN = WNW * TN * 32;
    static constexpr int A_OPER = BM * BT_ROWB, B_OPER = BN * BT_ROWB, STAGE = A_OPER + B_OPER;
    static constexpr int A_IPW = A_OPER / 1024 / BT_WAVES;                                   // DMA instructions per wave and step
    static constexpr int B_INS = B_OPER / 1024, B_IPW = B_INS >= BT_WAVES ? B_INS / BT_WAVES : 1;
    static constexpr bool B_SOME = B_INS < BT_WAVES;                                         // only waves < B_INS fetch filter rows
    static constexpr size_t LDS = (size_t)BT_NBUF * STAGE + (B_SOME ? 1024 : 0);
    static_assert(A_IPW >= 1 && BM % 128 == 0 && (B_INS % BT_WAVES == 0 || B_INS < BT_WAVES) && LDS <= 160 * 1024, "tile shape");
};

template <bool SK, bool YBF, int WMW = 4, int TN = 4, int ABLM = 0>
__global__ void __launch_bounds__(BT_WAVES * 64, 2) igemm_b16t_kernel(IgemmArgs a) {
    using SH = BtShape<WMW, TN>;
    constexpr int BM = SH::BM, BN = SH::BN, BK = BT_BK, NBUF = BT_NBUF, TM = 2;
    constexpr int A_IPW = SH::A_IPW, B_IPW = SH::B_IPW, STAGE = SH::STAGE, A_OPER = SH::A_OPER;
    extern __shared__ __attribute__((aligned(16))) char ldsb[];

    // ---- work of this workgroup: one (tile, K range) piece, or with SK a sequence of them
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
    const int wm = wave % WMW, wn = wave / WMW;
    const bool late = wave >= 4 && !a.lockstep;   // (the SIMD partner of wave w is wave w + 4)

    // ---- loader side: constants of the launch
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
    // instruction j of this wave moves slots [(wave * IPW + j) * 64, +64) of an operand's image: lane -> (row, slot)
    int ld_row[A_IPW], ld_chunk[A_IPW];
#pragma unroll
    for (int j = 0; j < A_IPW; j++) {
        const int q = (wave * A_IPW + j) * 64 + lane;
        ld_row[j] = q >> 2;
        ld_chunk[j] = (q & 3) ^ ((ld_row[j] >> 2) & 3);
    }

    // ---- consumer side: fragment addresses (bytes inside a stage).  Lane (l31, lh) reads k 8 lh .. 8 lh + 7 of group g from
    // row l31: chunk 2 g + lh, stored at slot (2 g + lh) ^ ((l31 >> 2) & 3); the wave's next 32 rows are 2048 bytes on
    const int l31 = lane & 31, lh = lane >> 5;
    const int sw = (l31 >> 2) & 3;
    int a_frag[2], b_frag[2];
#pragma unroll
    for (int g = 0; g < 2; g++) {
        a_frag[g] = (wm * 64 + l31) * BT_ROWB + (((2 * g + lh) ^ sw) << 4);
        b_frag[g] = A_OPER + (wn * (TN * 32) + l31) * BT_ROWB + (((2 * g + lh) ^ sw) << 4);
    }
    f32x16 acc[TM][TN];
    bt_bf16x8 av[2][TM], bv[2][TN];

    do {   // one piece of work (SK: several)
    const int m0 = a.m_begin + tile_m * BM, n0 = tile_n * BN;
    unsigned a_off[A_IPW], a_bad[A_IPW], b_off[B_IPW];
#pragma unroll
    for (int j = 0; j < A_IPW; j++) {
        const int r = ld_row[j];
        const int m = m0 + r < a.m_end ? m0 + r : -1;
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
    // the filter image is stored in LDS order: plain copies (a wave without filter rows of its own: every lane out of range)
    const bool b_mine = !SH::B_SOME || wave < SH::B_INS;
#pragma unroll
    for (int j = 0; j < B_IPW; j++) b_off[j] = b_mine ? (unsigned)((wave * B_IPW + j) * 64 + lane) * 16u : 0x80000000u;

    // DMAs of K step kt (of this piece's tile) into the ring buffer at byte offset `off`: this wave's 2 KB of each operand.
    // K order = (channel group, tap).  A step past the end of the piece (live = false) carries the out-of-range bit on every
    // lane and deposits zeros: every step issues the same four instructions, so the vmcnt bookkeeping is a constant.
    auto fill = [&](int off, int kt, bool live) {
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
        char *as = ldsb + off + wave * (A_IPW * 1024);
        char *bs = b_mine ? ldsb + off + A_OPER + wave * (B_IPW * 1024) : ldsb + NBUF * STAGE;   // (scratch KB: zeros nobody reads)
        if constexpr (ABLM & 1) return;
#pragma unroll
        for (int j = 0; j < A_IPW; j++)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(xr, (lds_void_t *)(as + j * 1024), 16, (int)(a_off[j] | ((a_bad[j] << sh) & kOOB) | dead), (int)s_x, 0, 0);
#pragma unroll
        for (int j = 0; j < B_IPW; j++)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(wr, (lds_void_t *)(bs + j * 1024), 16, (int)(b_off[j] | dead), (int)s_w, 0, 0);
    };
#define BT_SB __builtin_amdgcn_sched_barrier(0);
#define BT_RA0(G, I, BASE) av[G][I] = *reinterpret_cast<const bt_bf16x8 *>(ldsb + (BASE) + a_frag[G] + (I) * (32 * BT_ROWB)); BT_SB
#define BT_RB0(G, J, BASE) bv[G][J] = *reinterpret_cast<const bt_bf16x8 *>(ldsb + (BASE) + b_frag[G] + (J) * (32 * BT_ROWB)); BT_SB
#define BT_RA(G, I, BASE) if constexpr (!(ABLM & 2)) { BT_RA0(G, I, BASE) }
#define BT_RB(G, J, BASE) if constexpr (!(ABLM & 2)) { BT_RB0(G, J, BASE) }
#define BT_MF(G, I, J) if constexpr (!(ABLM & 8)) { acc[I][J] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av[G][I], bv[G][J], acc[I][J], 0, 0, 0); BT_SB }
    //   NX / PV = byte offsets of the ring buffers of steps kt + 1 and kt - 1 (= kt + 3)
    // (the loops below are fully unrolled -- TM, TN are template constants -- and every statement ends in a scheduling barrier: the
    // order written IS the software pipeline)
    auto read_frag = [&](int G, int n, int BASE) {   // fragment n of k group G: the TM A fragments, then the TN B fragments
        if (n < TM) { BT_RA(G, n, BASE) } else { BT_RB(G, n - TM, BASE) }
    };
    auto k_step = [&](int NX, int PV, int kt) {
        BT_SB
#pragma unroll
        for (int i = 0; i < TM; i++)
#pragma unroll
            for (int j = 0; j < TN; j++) { BT_MF(0, i, j) }
        // The two waves of a SIMD (w and w + 4) run the same program between the same barriers; issued at the same point, their
        // DMA bursts (~100 cycles of address-unit wait per instruction) would idle the SIMD's matrix pipe together.  Waves 0-3
        // issue theirs in FRONT of the barrier, waves 4-7 BEHIND it: each half's burst runs beside the other half's MFMAs.
        if (!late) fill(PV, kt0 + kt + 3, kt + 3 < nk);
        BT_SB
        if constexpr (!(ABLM & 4)) {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // every fragment of this step has arrived
            // this wave's DMAs of step kt + 1 have landed (in flight: kt + 2 and, for waves 0-3, kt + 3)
            if (late) wait_vmcnt<A_IPW + B_IPW>(); else wait_vmcnt<2 * (A_IPW + B_IPW)>();
            if constexpr (!(ABLM & 16)) __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
        }
        if (late) fill(PV, kt0 + kt + 3, kt + 3 < nk);
        BT_SB
        // the second k group's MFMAs, the next step's first-group fragments one behind each of them (TM + TN <= TM * TN reads) ...
#pragma unroll
        for (int i = 0; i < TM; i++)
#pragma unroll
            for (int j = 0; j < TN; j++) {
                BT_MF(1, i, j)
                if (i * TN + j < TM + TN) read_frag(0, i * TN + j, NX);
            }
        // ... and its second group behind the last one (B first: the next step's first MFMAs need A's group 0 only)
#pragma unroll
        for (int j = 0; j < TN; j++) { BT_RB(1, j, NX) }
#pragma unroll
        for (int i = 0; i < TM; i++) { BT_RA(1, i, NX) }
    };

#pragma unroll
    for (int i = 0; i < TM; i++)
#pragma unroll
        for (int j = 0; j < TN; j++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[i][j][r] = 0.f;
    __syncthreads();   // the ring is free (SK: the previous piece's epilogue staged through it)
    fill(0, kt0, true);
    fill(STAGE, kt0 + 1, 1 < nk);
    fill(2 * STAGE, kt0 + 2, 2 < nk);
    wait_vmcnt<2 * (A_IPW + B_IPW)>();
    __builtin_amdgcn_s_barrier();   // step 0 has landed
    asm volatile("" ::: "memory");
#pragma unroll
    for (int g = 0; g < 2; g++) {
#pragma unroll
        for (int i = 0; i < TM; i++) { BT_RA0(g, i, 0) }
#pragma unroll
        for (int j = 0; j < TN; j++) { BT_RB0(g, j, 0) }
    }
    {
        int u_off = 0, pv = (NBUF - 1) * STAGE;
        for (int kt = 0; kt < nk; kt++) {
            const int nx = u_off == (NBUF - 1) * STAGE ? 0 : u_off + STAGE;
            k_step(nx, pv, kt);
            pv = u_off;
            u_off = nx;
        }
    }
    // the zero-depositing DMAs of the last steps are still in flight, and so may be the last fetches: drain both before
    // the ring becomes the epilogue's staging area
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    wait_vmcnt<0>();
    __syncthreads();
#undef BT_SB
#undef BT_RA
#undef BT_RB
#undef BT_RA0
#undef BT_RB0
#undef BT_MF

    float *slab_ = lvl >= 0 ? a.slab + ((long long)lvl * (a.m_end - a.m_begin) - a.m_begin) * a.N : nullptr;
    int lane_e = tid & 63;
    asm volatile("" : "+v"(lane_e));
    float *stage = reinterpret_cast<float *>(ldsb) + wave * (16 * (TN * 32 + 4));
    static_assert(BT_WAVES * 16 * (TN * 32 + 4) * 4 <= SH::LDS, "epilogue slabs must fit in the ring");
    store_acc_tile<TM, TN, 16, YBF>(a, acc, stage, lane_e, m0 + wm * 64, n0 + wn * (TN * 32), slab_);
    } while (SK && next_piece());
}

// ------------------------------------------------------------------------------------------------------------------
// "Wide A" variant (igemm_b16w_kernel): the activation side fetched in 128-BYTE pieces.
//
// What caps igemm_b16t_kernel is the delivery rate of its operand stream (profiles/r04_bf16t_ablation.txt: the DMAs alone run at
// 47 GB/s per CU = 62 % of the matrix rate): a K step of 32 channels is a 64-byte piece of a pixel's row, half a cache line per
// lane group -- every line crosses L2 -> L1 twice, a K step apart, and a 64-lane instruction spans 16 lines instead of 8.  The
// filter side has no such problem (its image is stored in LDS order: contiguous KBs).  Here the ACTIVATION ring holds DOUBLE steps:
//     A slot = [256 rows][128 B] = 64 channels of one tap (32 KB; ring of 3)          B slot = [256 rows][64 B] as before (ring of 4)
// 96 + 64 = 160 KB, all of the LDS.  The K order becomes (channel pair-group of 64, tap, half); a unit of work is a double step
// (stream-K and split-K cut at multiples of 64 channels: Cx % 64 == 0), and inside it the two half steps run igemm_b16t_kernel's
// schedule unchanged (barrier per half step, fragments of the next half step prefetched behind it): the A fragments of half h come
// from the 64-byte half h of the 128-byte rows.  An A slot is filled every other half step (4 DMA instructions per wave, 8 whole
// rows each), a B slot every half step (2 instructions).  Swizzle of the 128-byte rows: slot s (of 8) of row r holds chunk
// s ^ f(r), f(r) = ((r >> 1) & 3) | (((r >> 4) & 1) << 2) -- the 16 rows of a ds_read_b128 lane group ({0-3, 12-15, 20-27} and
// {4-11, 16-19, 28-31} mod 32) then hit 16 different 16-byte columns of the two 128-byte halves of a bank row.
// ------------------------------------------------------------------------------------------------------------------
constexpr int BW_AROW = 128;
constexpr int BW_ASLOT = 256 * BW_AROW;          // 32 KB
constexpr int BW_NA = 3, BW_NB = 4;
constexpr int BW_BBASE = BW_NA * BW_ASLOT;       // the B ring behind the A ring
constexpr int BW_AIPW = BW_ASLOT / 1024 / BT_WAVES;   // 4 DMA instructions per wave and A slot
constexpr size_t BW_LDS = (size_t)BW_BBASE + (size_t)BW_NB * BT_OPER;
static_assert(BW_LDS == 160 * 1024, "the two rings fill the LDS");

template <bool SK, bool YBF, int ABLM = 0>
__global__ void __launch_bounds__(BT_WAVES * 64, 2) igemm_b16w_kernel(IgemmArgs a) {
    constexpr int BM = 256, BN = 256, TM = 2, TN = 4;
    extern __shared__ __attribute__((aligned(16))) char ldsb[];

    // ---- work of this workgroup, in DOUBLE steps (64 channels of one tap)
    const int nk_all = a.K / 64;
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
    const int wm = wave & 3, wn = wave >> 2;
    const bool late = wave >= 4 && !a.lockstep;

    const int rowb = a.Cx * 2;
    const int n_pad = a.tiles_n * 256;
    const int cgroups = a.Cx / 32;
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
    // A: instruction j of this wave moves slots [(wave * 4 + j) * 64, +64) of an A slot = 8 whole rows: lane -> (row, chunk)
    int ld_row[BW_AIPW], ld_chunk[BW_AIPW];
#pragma unroll
    for (int j = 0; j < BW_AIPW; j++) {
        const int q = (wave * BW_AIPW + j) * 64 + lane;
        const int r = q >> 3;
        ld_row[j] = r;
        ld_chunk[j] = (q & 7) ^ (((r >> 1) & 3) | (((r >> 4) & 1) << 2));
    }

    // fragment addresses: A in a slot of 128-byte rows (half h, k group g: chunk 4 h + 2 g + lh), B in a slot of 64-byte rows
    const int l31 = lane & 31, lh = lane >> 5;
    const int fa = ((l31 >> 1) & 3) | (((l31 >> 4) & 1) << 2);
    const int sw = (l31 >> 2) & 3;
    int a_frag[2][2], b_frag[2];
#pragma unroll
    for (int h = 0; h < 2; h++)
#pragma unroll
        for (int g = 0; g < 2; g++) a_frag[h][g] = (wm * 64 + l31) * BW_AROW + (((4 * h + 2 * g + lh) ^ fa) << 4);
#pragma unroll
    for (int g = 0; g < 2; g++) b_frag[g] = BW_BBASE + (wn * 128 + l31) * BT_ROWB + (((2 * g + lh) ^ sw) << 4);
    f32x16 acc[TM][TN];
    bt_bf16x8 av[2][TM], bv[2][TN];

    do {   // one piece of work (SK: several)
    const int m0 = a.m_begin + tile_m * BM, n0 = tile_n * BN;
    unsigned a_off[BW_AIPW], a_bad[BW_AIPW], b_off[BT_IPW];
#pragma unroll
    for (int j = 0; j < BW_AIPW; j++) {
        const int r = ld_row[j];
        const int m = m0 + r < a.m_end ? m0 + r : -1;
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
#pragma unroll
    for (int j = 0; j < BT_IPW; j++) b_off[j] = (unsigned)((wave * BT_IPW + j) * 64 + lane) * 16u;

    // double step D (of the tile, absolute) -> (channel pair-group, tap); the filter's K step = (tap, chunk 2 cp + half)
    auto fillA = [&](int slot, int D, bool live) {
        if constexpr (ABLM & 1) return;
        const unsigned cp = __umulhi((unsigned)D, a.magic_nt);
        const unsigned tap = a.ntaps > 1 ? (unsigned)D - cp * (unsigned)a.ntaps : 0u;
        const unsigned cp_u = a.ntaps > 1 ? cp : (unsigned)D;
        const unsigned kh = div_magic(tap, a.magic_kw);
        const int kw = (int)(tap - kh * (unsigned)a.KW);
        const int dh = a.sign * ((int)kh * a.dil - a.pad_h);
        const int dw = a.sign * (kw * a.dil - a.pad_w);
        const unsigned s_x = (unsigned)__builtin_amdgcn_readfirstlane((int)((unsigned)((dh * a.W + dw) * rowb) + cp_u * (unsigned)BW_AROW + bias));
        const unsigned sh = 31u - (tap & 31u);
        const unsigned dead = live ? 0u : kOOB;
        char *as = ldsb + slot * BW_ASLOT + wave * (BW_AIPW * 1024);
#pragma unroll
        for (int j = 0; j < BW_AIPW; j++)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(xr, (lds_void_t *)(as + j * 1024), 16, (int)(a_off[j] | ((a_bad[j] << sh) & kOOB) | dead), (int)s_x, 0, 0);
    };
    auto fillB = [&](int slot, int kt, bool live) {   // kt: half step of the tile, absolute
        if constexpr (ABLM & 1) return;
        const unsigned D = (unsigned)kt >> 1, half = (unsigned)kt & 1u;
        const unsigned cp = __umulhi(D, a.magic_nt);
        const unsigned tap = a.ntaps > 1 ? D - cp * (unsigned)a.ntaps : 0u;
        const unsigned cp_u = a.ntaps > 1 ? cp : D;
        const unsigned s_w = (unsigned)__builtin_amdgcn_readfirstlane((int)(((tap * (unsigned)cgroups + 2u * cp_u + half) * (unsigned)n_pad + (unsigned)n0) * (unsigned)BT_ROWB));
        const unsigned dead = live ? 0u : kOOB;
        char *bs = ldsb + BW_BBASE + slot * BT_OPER + wave * (BT_IPW * 1024);
#pragma unroll
        for (int j = 0; j < BT_IPW; j++)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(wr, (lds_void_t *)(bs + j * 1024), 16, (int)(b_off[j] | dead), (int)s_w, 0, 0);
    };
#define BW_SB __builtin_amdgcn_sched_barrier(0);
#define BW_RA0(H, G, I, AS) av[G][I] = *reinterpret_cast<const bt_bf16x8 *>(ldsb + (AS) * BW_ASLOT + a_frag[H][G] + (I) * (32 * BW_AROW)); BW_SB
#define BW_RB0(G, J, BS) bv[G][J] = *reinterpret_cast<const bt_bf16x8 *>(ldsb + (BS) * BT_OPER + b_frag[G] + (J) * (32 * BT_ROWB)); BW_SB
#define BW_RA(H, G, I, AS) if constexpr (!(ABLM & 2)) { BW_RA0(H, G, I, AS) }
#define BW_RB(G, J, BS) if constexpr (!(ABLM & 2)) { BW_RB0(G, J, BS) }
#define BW_MF(G, I, J) if constexpr (!(ABLM & 8)) { acc[I][J] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av[G][I], bv[G][J], acc[I][J], 0, 0, 0); BW_SB }
    // One half step (H = 0 / 1 static).  kt = half step of the piece; as_n / hn = A slot and half of the NEXT half step, bs_n its B
    // slot; as_f / bs_f = the A / B slots this step's DMAs fill (A[D + 2] when H == 0, B[kt + 3]).
#define BW_HALF(H, KT, AS_N, HN, BS_N, AS_F, BS_F, D)                                                                              \
    {                                                                                                                              \
        BW_SB                                                                                                                      \
        BW_MF(0, 0, 0) BW_MF(0, 0, 1) BW_MF(0, 0, 2) BW_MF(0, 0, 3) BW_MF(0, 1, 0) BW_MF(0, 1, 1) BW_MF(0, 1, 2) BW_MF(0, 1, 3)   \
        if (!late) {                                                                                                               \
            if (H == 0) fillA(AS_F, kt0 + (D) + 2, (D) + 2 < nk);                                                                  \
            fillB(BS_F, 2 * kt0 + (KT) + 3, (KT) + 3 < 2 * nk);                                                                    \
        }                                                                                                                          \
        BW_SB                                                                                                                      \
        if constexpr (!(ABLM & 4)) {                                                                                               \
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                                     \
            if (late) { if (H == 0) wait_vmcnt<BT_IPW>(); else wait_vmcnt<BW_AIPW + BT_IPW>(); }                                   \
            else wait_vmcnt<BW_AIPW + 2 * BT_IPW>();                                                                               \
            if constexpr (!(ABLM & 16)) __builtin_amdgcn_s_barrier();                                                              \
            asm volatile("" ::: "memory");                                                                                         \
        }                                                                                                                          \
        if (late) {                                                                                                                \
            if (H == 0) fillA(AS_F, kt0 + (D) + 2, (D) + 2 < nk);                                                                  \
            fillB(BS_F, 2 * kt0 + (KT) + 3, (KT) + 3 < 2 * nk);                                                                    \
        }                                                                                                                          \
        BW_SB                                                                                                                      \
        BW_MF(1, 0, 0) BW_RA(HN, 0, 0, AS_N) BW_MF(1, 0, 1) BW_RA(HN, 0, 1, AS_N) BW_MF(1, 0, 2) BW_RB(0, 0, BS_N) BW_MF(1, 0, 3) BW_RB(0, 1, BS_N) \
        BW_MF(1, 1, 0) BW_RB(0, 2, BS_N) BW_MF(1, 1, 1) BW_RB(0, 3, BS_N) BW_MF(1, 1, 2) BW_MF(1, 1, 3)                           \
        BW_RB(1, 0, BS_N) BW_RB(1, 1, BS_N) BW_RB(1, 2, BS_N) BW_RB(1, 3, BS_N) BW_RA(HN, 1, 0, AS_N) BW_RA(HN, 1, 1, AS_N)       \
    }

#pragma unroll
    for (int i = 0; i < TM; i++)
#pragma unroll
        for (int j = 0; j < TN; j++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[i][j][r] = 0.f;
    __syncthreads();   // the rings are free (SK: the previous piece's epilogue staged through them)
    // issue order A[0] B[0] A[1] B[1] B[2]: behind B[0] stand 4 + 2 + 2 instructions
    fillA(0, kt0, true);
    fillB(0, 2 * kt0, true);
    fillA(1, kt0 + 1, 1 < nk);
    fillB(1, 2 * kt0 + 1, true);
    fillB(2, 2 * kt0 + 2, 2 < 2 * nk);
    wait_vmcnt<BW_AIPW + 2 * BT_IPW>();
    __builtin_amdgcn_s_barrier();   // A[0], B[0] have landed
    asm volatile("" ::: "memory");
    BW_RA0(0, 0, 0, 0) BW_RA0(0, 0, 1, 0) BW_RB0(0, 0, 0) BW_RB0(0, 1, 0) BW_RB0(0, 2, 0) BW_RB0(0, 3, 0)
    BW_RB0(1, 0, 0) BW_RB0(1, 1, 0) BW_RB0(1, 2, 0) BW_RB0(1, 3, 0) BW_RA0(0, 1, 0, 0) BW_RA0(0, 1, 1, 0)
    {
        int as = 0, bs = 0;   // A slot of double step D, B slot of half step 2 D
        for (int D = 0; D < nk; D++) {
            const int as1 = as == BW_NA - 1 ? 0 : as + 1;            // A[D + 1]
            const int as2 = as1 == BW_NA - 1 ? 0 : as1 + 1;          // A[D + 2]'s slot (= A[D - 1]'s)
            const int b1 = (bs + 1) & 3, b2 = (bs + 2) & 3, b3 = (bs + 3) & 3;
            // half 0: next = (D, half 1): same A slot, B slot b1; fills A[D + 2] -> as2, B[2 D + 3] -> b3
            BW_HALF(0, 2 * D, as, 1, b1, as2, b3, D)
            // half 1: next = (D + 1, half 0): A slot as1, B slot b2; fills B[2 D + 4] -> slot of B[2 D] = bs
            BW_HALF(1, 2 * D + 1, as1, 0, b2, as2, bs, D)
            as = as1;
            bs = b2;
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    wait_vmcnt<0>();
    __syncthreads();
#undef BW_SB
#undef BW_RA0
#undef BW_RB0
#undef BW_RA
#undef BW_RB
#undef BW_MF
#undef BW_HALF

    float *slab_ = lvl >= 0 ? a.slab + ((long long)lvl * (a.m_end - a.m_begin) - a.m_begin) * a.N : nullptr;
    int lane_e = tid & 63;
    asm volatile("" : "+v"(lane_e));
    float *stage = reinterpret_cast<float *>(ldsb) + wave * (16 * (128 + 4));
    store_acc_tile<TM, TN, 16, YBF>(a, acc, stage, lane_e, m0 + wm * 64, n0 + wn * 128, slab_);
    } while (SK && next_piece());
}

// a.splits: SK = false -> split-K factor (>= 1; > 1 needs a.slab); SK = true -> number of workgroups (CUs), a.slab for the levels
template <bool SK, bool YBF, int WMW = 4, int TN = 4, int ABLM = 0>
int launch_igemm_b16t(IgemmArgs &a, hipStream_t stream) {
    using SH = BtShape<WMW, TN>;
#ifdef EMADAPT_ABLATE_BUILD
    if constexpr (ABLM == 0 && YBF) {
        switch (tune(TUNE_ABLATE)) {
#define EMADAPT_ABL_CASE(m) case m: return launch_igemm_b16t<SK, YBF, WMW, TN, m>(a, stream);
            EMADAPT_ABL_CASE(1) EMADAPT_ABL_CASE(2) EMADAPT_ABL_CASE(3) EMADAPT_ABL_CASE(4) EMADAPT_ABL_CASE(7) EMADAPT_ABL_CASE(8)
            EMADAPT_ABL_CASE(9) EMADAPT_ABL_CASE(10) EMADAPT_ABL_CASE(14) EMADAPT_ABL_CASE(16)
#undef EMADAPT_ABL_CASE
            default: break;
        }
    }
#endif
    static DynLdsAttr attr;
    int rc = attr.ensure(reinterpret_cast<const void *>(&igemm_b16t_kernel<SK, YBF, WMW, TN, ABLM>), SH::LDS);
    if (rc) return rc;
    a.tiles_m = (int)ceil_div(a.m_end - a.m_begin, SH::BM);
    a.tiles_n = (int)ceil_div(a.N, SH::BN);
    const int nk_all = a.K / BT_BK;
    const long long tiles = (long long)a.tiles_m * a.tiles_n;
    if constexpr (SK) {
        if (a.splits < 1 || tiles * nk_all > 0x7fffffffLL || tiles * (SH::BM / 16) > 0x7fffffffLL) return EMADAPT_ERR_UNSUPPORTED;
        const int G = streamk_grid(tiles, nk_all, a.splits);
        const int levels = streamk_levels(tiles, nk_all, a.splits);
        if (levels > 0 && !a.slab) return EMADAPT_ERR_WORKSPACE;
        hipLaunchKernelGGL((igemm_b16t_kernel<true, YBF, WMW, TN, ABLM>), dim3((unsigned)G), dim3(BT_WAVES * 64), SH::LDS, stream, a);
        rc = launch_status();
        if (rc || levels == 0) return rc;
        hipLaunchKernelGGL((streamk_fixup_t_kernel<YBF, SH::BM, SH::BN>), dim3((unsigned)(tiles * (SH::BM / 16))), dim3(256), 0, stream, a, G, nk_all);
        return launch_status();
    } else {
        const long long nblk = tiles * a.splits;
        if (nblk > 0x7fffffffLL) return EMADAPT_ERR_UNSUPPORTED;
        hipLaunchKernelGGL((igemm_b16t_kernel<false, YBF, WMW, TN, ABLM>), dim3((unsigned)nblk), dim3(BT_WAVES * 64), SH::LDS, stream, a);
        rc = launch_status();
        if (rc || a.splits <= 1) return rc;
        long long g = ((long long)(a.m_end - a.m_begin) * a.N / 4 + 255) / 256;
        if (g > 8192) g = 8192;
        hipLaunchKernelGGL(splitk_epilogue_t_kernel<YBF>, dim3((unsigned)g), dim3(256), 0, stream, a);
        return launch_status();
    }
}

template <bool SK, bool YBF, int ABLM = 0>
int launch_igemm_b16w(IgemmArgs &a, hipStream_t stream) {
#ifdef EMADAPT_ABLATE_BUILD
    if constexpr (ABLM == 0 && YBF) {
        switch (tune(TUNE_ABLATE)) {
#define EMADAPT_ABL_CASE(m) case m: return launch_igemm_b16w<SK, YBF, m>(a, stream);
            EMADAPT_ABL_CASE(1) EMADAPT_ABL_CASE(2) EMADAPT_ABL_CASE(3) EMADAPT_ABL_CASE(4) EMADAPT_ABL_CASE(7) EMADAPT_ABL_CASE(8)
            EMADAPT_ABL_CASE(9) EMADAPT_ABL_CASE(10) EMADAPT_ABL_CASE(14) EMADAPT_ABL_CASE(16)
#undef EMADAPT_ABL_CASE
            default: break;
        }
    }
#endif
    static DynLdsAttr attr;
    int rc = attr.ensure(reinterpret_cast<const void *>(&igemm_b16w_kernel<SK, YBF, ABLM>), BW_LDS);
    if (rc) return rc;
    a.tiles_m = (int)ceil_div(a.m_end - a.m_begin, 256);
    a.tiles_n = (int)ceil_div(a.N, 256);
    const int nk_all = a.K / 64;
    const long long tiles = (long long)a.tiles_m * a.tiles_n;
    if constexpr (SK) {
        if (a.splits < 1 || tiles * nk_all > 0x7fffffffLL || tiles * 16 > 0x7fffffffLL) return EMADAPT_ERR_UNSUPPORTED;
        const int G = streamk_grid(tiles, nk_all, a.splits);
        const int levels = streamk_levels(tiles, nk_all, a.splits);
        if (levels > 0 && !a.slab) return EMADAPT_ERR_WORKSPACE;
        hipLaunchKernelGGL((igemm_b16w_kernel<true, YBF, ABLM>), dim3((unsigned)G), dim3(BT_WAVES * 64), BW_LDS, stream, a);
        rc = launch_status();
        if (rc || levels == 0) return rc;
        hipLaunchKernelGGL(streamk_fixup_t_kernel<YBF>, dim3((unsigned)(tiles * 16)), dim3(256), 0, stream, a, G, nk_all);
        return launch_status();
    } else {
        const long long nblk = tiles * a.splits;
        if (nblk > 0x7fffffffLL) return EMADAPT_ERR_UNSUPPORTED;
        hipLaunchKernelGGL((igemm_b16w_kernel<false, YBF, ABLM>), dim3((unsigned)nblk), dim3(BT_WAVES * 64), BW_LDS, stream, a);
        rc = launch_status();
        if (rc || a.splits <= 1) return rc;
        long long g = ((long long)(a.m_end - a.m_begin) * a.N / 4 + 255) / 256;
        if (g > 8192) g = 8192;
        hipLaunchKernelGGL(splitk_epilogue_t_kernel<YBF>, dim3((unsigned)g), dim3(256), 0, stream, a);
        return launch_status();
    }
}

}  // namespace

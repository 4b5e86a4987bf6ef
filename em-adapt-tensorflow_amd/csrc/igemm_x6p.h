// bf16x6 on PRE-SPLIT operands ("plane images"): forward / data-gradient kernel, included by conv_x6p.hip.
//
// conv_x6.hip splits both fp32 operands into their three bf16 planes inside the loader waves -- once per tile, i.e. an
// activation is split again by every N tile and every filter tap that reads it (7 vector-ALU instructions per MFMA at
// 128x128 tiles) and stored to LDS through registers.  Here the split happens ONCE per tensor (split_planes_kernel) into a plane image
//     [row][k / 16][plane 0..2][16 x bf16]          (96 bytes per row and group of 16 k; a row is a pixel / a filter row)
// and the GEMM kernel is built like the second-generation fp32 kernel (igemm_v2.h): tiles land in LDS by LDS-DMA exactly as
// they lie in memory, nobody issues a vector-ALU instruction per element.
//
//   * 256 x 256 x 16 tiles, ONE workgroup of 8 waves per CU, every wave a 64 x 128 accumulator block (TM = 2, TN = 4: 48 MFMAs
//     32x32x16 per K step from 18 ds_read_b128) AND 6 of the step's 48 DMA instructions -- no loader waves: with 72 fragment
//     registers beside the 128 accumulators a wave needs more than the 168 registers a 12-wave workgroup would leave it.
//   * LDS image of an operand's K step: [256 rows][96 B], dense (24 KB; ring of 3 stages x 2 operands = 144 KB).  A DMA
//     instruction moves 64 consecutive 16-byte chunks of that image, i.e. 10 2/3 rows: lane -> (row, chunk) is a per-lane
//     constant of the instruction, the row's pixel and its tap validity bits are looked up once per tile.
//     (buffer_load_dwordx3 ... lds would give 8 whole rows per instruction, but it deposits lane l's 12 bytes at base + 16 l,
//     not 12 l: scripts/dma_x3_lab.hip.)
//   * bank conflicts: the 16 lanes of one ds_read_b128 group read the same 16-byte column of 16 rows 96 bytes apart -- only
//     half of the banks -- so rows with bit 3 set store the two 16-byte halves of every plane swapped (the DMA lanes fetch
//     them swapped, the fragment address XORs 16): conflict-free.
//   * pipeline (k_step below): the next step's fragments are fetched into the registers this step has finished with while its
//     MFMAs run, the barrier sits in the middle of the step's MFMA stream (work queued on both sides of it, no LDS round trip
//     behind it), and the DMAs of step kt + 2 leave as one burst in front of it.
//   * the FILTER operand does not need the gather at all: its plane image is stored tile by tile in exactly the LDS order
//         [tap][k / 16][row, padded with zero rows to a multiple of 256][6 chunks of 16 B, halves swapped where row bit 3 is set]
//     (filter_tiles_kernel, once per optimizer step), so a K step's 256 filter rows are 24 contiguous KB and the filter DMAs are
//     plain 1 KB copies -- half of the kernel's operand traffic at the DMA path's best rate (the activation side's 96-byte
//     pieces run at about half of it: scripts/x6p_ablate.py).
//   * work split, packed rows, epilogues, stream-K fix-up: igemm_v2.h's, unchanged (same accumulator layout).
// Requirements: Cx % 16 == 0, KH*KW <= 32, N % 4 == 0; plane images below 2 GiB.
#pragma once
#include "igemm_v2.h"

namespace {

typedef __bf16 xp_bf16x8 __attribute__((ext_vector_type(8)));

constexpr int XP_ROWB = 96;                  // bytes of one row's K step (3 planes x 16 bf16), in memory and in LDS
constexpr int XP_OPER = 256 * XP_ROWB;       // one operand's K step in LDS
constexpr int XP_STAGE = 2 * XP_OPER;        // A + B
constexpr int XP_NBUF = 3;
constexpr int XP_WAVES = 8;
constexpr int XP_IPW = XP_OPER / 1024 / XP_WAVES;   // DMA instructions per wave, operand and K step (3)
constexpr size_t XP_LDS = (size_t)XP_NBUF * XP_STAGE;

__device__ __forceinline__ void xp_dma(__amdgpu_buffer_rsrc_t r, char *lds_base, unsigned voffset, unsigned soffset) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (lds_void_t *)lds_base, 16, (int)voffset, (int)soffset, 0, 0);
}

// ABLM (diagnostic library only, tune "ablate"): parts of the K loop COMPILED out -- 1 no DMAs, 2 no fragment reads, 4 no waits /
// barrier, 8 no MFMAs, 16 no barrier (waits stay), 32 no wait for the DMAs (results are garbage; scripts/x6p_ablate.py reads the time).
template <bool SK, int ABLM = 0>
__global__ void __launch_bounds__(XP_WAVES * 64, 2) igemm_x6p_kernel(IgemmArgs a) {
    constexpr int BM = 256, BN = 256, BK = 16, NBUF = XP_NBUF, TM = 2, TN = 4;
    extern __shared__ __attribute__((aligned(16))) char ldsb[];

    // ---- work of this workgroup: igemm2_kernel's piece list (one (tile, K range) piece, or with SK a sequence of them)
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
            sk_units = a.pk.n > 0 ? a.pk.run_u0[a.pk.nruns] : ntile * nk_all;
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
    int run = 0;
    int nk_tile = nk_all;
    auto next_piece = [&]() -> bool {
        if (sk_u >= sk_end) return false;
        if (a.pk.n > 0) {
            while (run + 1 < a.pk.nruns && sk_u >= a.pk.run_u0[run + 1]) run++;
            nk_tile = (a.Cx / BK) * (int)a.pk.run_nkh[run] * (int)a.pk.run_nkw[run];
            const int rel = sk_u - a.pk.run_u0[run];
            const int t_in = __builtin_amdgcn_readfirstlane(rel / nk_tile);
            kt0 = rel - t_in * nk_tile;
            const int q = __builtin_amdgcn_readfirstlane(t_in / a.tiles_n);
            tile_m = a.pk.run_tile0[run] + q;
            tile_n = t_in - q * a.tiles_n;
        } else {
            const int t = __builtin_amdgcn_readfirstlane(sk_u / nk_tile);
            kt0 = sk_u - t * nk_tile;
            tile_m = __builtin_amdgcn_readfirstlane(t / a.tiles_n);
            tile_n = t - tile_m * a.tiles_n;
        }
        const int left = sk_end - sk_u;
        nk = (nk_tile - kt0 < left) ? nk_tile - kt0 : left;
        lvl = nk == nk_tile ? -1 : sk_v - __builtin_amdgcn_readfirstlane(sk_owner((long long)(sk_u - kt0), sk_units, (int)gridDim.x));
        sk_u += nk;
        return true;
    };
    if constexpr (SK) {
        if (!next_piece()) return;
    }

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave & 3, wn = wave >> 2;
    const bool late = wave >= 4 && !a.lockstep;   // (the SIMD partner of wave w is wave w + 4; see k_step)

    // ---- loader side: constants of the launch
    // (the tap offset rides in the DMA's scalar operand, an UNSIGNED addend: the buffer starts `bias` bytes before the tensor)
    const int rowb = a.Cx / 16 * XP_ROWB;   // bytes of a pixel in a plane image
    const int n_pad = a.tiles_n * 256;     // filter rows per (tap, group of 16 k) in the tiled filter image
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
    // instruction j of this wave moves chunks [(wave * IPW + j) * 64, +64) of an operand's image: lane -> (row, chunk position);
    // rows with bit 3 set hold the halves of every plane swapped
    int ld_row[XP_IPW], ld_chunk[XP_IPW];
#pragma unroll
    for (int j = 0; j < XP_IPW; j++) {
        const int q = (wave * XP_IPW + j) * 64 + lane;
        ld_row[j] = q / 6;
        ld_chunk[j] = (q - ld_row[j] * 6) ^ ((ld_row[j] >> 3) & 1);
    }

    // ---- consumer side: fragment addresses (bytes inside a stage); plane p is 32 bytes on, the wave's next 32 rows 3072
    const int l31 = lane & 31, lh = lane >> 5;
    const int half = (lh ^ ((l31 >> 3) & 1)) * 16;
    const int a_frag = (wm * 64 + l31) * XP_ROWB + half;
    const int b_frag = XP_OPER + (wn * 128 + l31) * XP_ROWB + half;
    f32x16 acc[TM][TN];
    xp_bf16x8 av[3][TM], bv[3][TN];

    int sk_bad = 0;   // (scalar) a publisher of a tile this workgroup heads never arrived
    do {   // one piece of work (SK: several)
    const bool packed = SK && a.pk.n > 0;
    const int m0 = a.m_begin + tile_m * BM, n0 = tile_n * BN;   // (packed: m0 is a GEMM row, a.pk.rowmap gives the pixel)
    const int kh_first = packed ? a.pk.run_kh0[run] : 0, kw_first = packed ? a.pk.run_kw0[run] : 0;
    const int nkw_t = packed ? a.pk.run_nkw[run] : a.KW;
    const int ntaps_t = packed ? a.pk.run_nkh[run] * nkw_t : a.ntaps;
    const unsigned magic_nt_t = packed ? (ntaps_t > 1 ? (unsigned)((0x100000000ULL + (unsigned)ntaps_t - 1) / (unsigned)ntaps_t) : 0u) : a.magic_nt;
    const unsigned magic_nkw_t = packed ? (nkw_t > 1 ? (unsigned)((0x100000000ULL + (unsigned)nkw_t - 1) / (unsigned)nkw_t) : 0u) : a.magic_kw;
    unsigned a_off[XP_IPW], a_bad[XP_IPW], b_off[XP_IPW];
#pragma unroll
    for (int j = 0; j < XP_IPW; j++) {
        const int r = ld_row[j];
        const int m = packed ? a.pk.rowmap[m0 + r] : (m0 + r < a.m_end ? m0 + r : -1);
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
        b_off[j] = (unsigned)((wave * XP_IPW + j) * 64 + lane) * 16u;   // the filter image is stored tile by tile in LDS order: plain copies
    }

    // DMAs of K step kt (of this piece's tile) into ring buffer `buf`: this wave's 3 KB of each operand
    // DMAs of K step kt (of this piece's tile) into the ring buffer at byte offset `off`: this wave's 3 KB of each operand.
    // A step past the end of the piece (live = false) carries the out-of-range bit on every lane and deposits zeros: every
    // step issues the same six instructions, so the vmcnt bookkeeping is a constant.
    auto fill = [&](int off, int kt, bool live) {
        const unsigned chunk = __umulhi((unsigned)kt, magic_nt_t);
        const unsigned tap_l = ntaps_t > 1 ? (unsigned)kt - chunk * (unsigned)ntaps_t : 0u;
        const unsigned chunk_u = ntaps_t > 1 ? chunk : (unsigned)kt;
        const unsigned khl = div_magic(tap_l, magic_nkw_t);
        const unsigned kh = (unsigned)kh_first + khl;
        const int kw = kw_first + (int)(tap_l - khl * (unsigned)nkw_t);
        const unsigned tap_u = kh * (unsigned)a.KW + (unsigned)kw;
        const int dh = a.sign * ((int)kh * a.dil - a.pad_h);
        const int dw = a.sign * (kw * a.dil - a.pad_w);
        const unsigned s_x = (unsigned)__builtin_amdgcn_readfirstlane((int)((unsigned)((dh * a.W + dw) * rowb) + chunk_u * (unsigned)XP_ROWB + bias));
        const unsigned s_w = (unsigned)__builtin_amdgcn_readfirstlane((int)(((tap_u * (unsigned)(a.Cx / 16) + chunk_u) * (unsigned)n_pad + (unsigned)n0) * (unsigned)XP_ROWB));
        const unsigned sh = 31u - (tap_u & 31u);   // moves this tap's "outside the image" bit to bit 31
        const unsigned dead = live ? 0u : kOOB;
        char *as = ldsb + off + wave * (XP_IPW * 1024);
        if constexpr (!(ABLM & 1)) {
#pragma unroll
            for (int j = 0; j < XP_IPW; j++) xp_dma(xr, as + j * 1024, a_off[j] | ((a_bad[j] << sh) & kOOB) | dead, s_x);
#pragma unroll
            for (int j = 0; j < XP_IPW; j++) xp_dma(wr, as + XP_OPER + j * 1024, b_off[j] | dead, s_w);
        }
    };
    // Fragment reads and MFMAs, one instruction each; XP_SB pins the order they are written in (the schedule below IS the
    // software pipeline: the compiler must not regroup it).
#define XP_SB __builtin_amdgcn_sched_barrier(0);
#define XP_RA0(P, I, BASE) av[P][I] = *reinterpret_cast<const xp_bf16x8 *>(ldsb + (BASE) + a_frag + (P) * 32 + (I) * (32 * XP_ROWB)); XP_SB
#define XP_RB0(Q, J, BASE) bv[Q][J] = *reinterpret_cast<const xp_bf16x8 *>(ldsb + (BASE) + b_frag + (Q) * 32 + (J) * (32 * XP_ROWB)); XP_SB
#define XP_RA(P, I, BASE) if constexpr (!(ABLM & 2)) { XP_RA0(P, I, BASE) }
#define XP_RB(Q, J, BASE) if constexpr (!(ABLM & 2)) { XP_RB0(Q, J, BASE) }
#define XP_MF(P, Q, I, J) if constexpr (!(ABLM & 8)) { acc[I][J] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av[P][I], bv[Q][J], acc[I][J], 0, 0, 0); XP_SB }
#define XP_GROUP(P, Q) XP_MF(P, Q, 0, 0) XP_MF(P, Q, 0, 1) XP_MF(P, Q, 0, 2) XP_MF(P, Q, 0, 3) XP_MF(P, Q, 1, 0) XP_MF(P, Q, 1, 1) XP_MF(P, Q, 1, 2) XP_MF(P, Q, 1, 3)
    // One K step = six groups of 8 MFMAs (plane pairs) in the cyclic order (0,2) (0,1) (0,0) | (1,0) (2,0) (1,1), the barrier
    // after the third.  In that order every fragment's last use lies at least one group before its first use in the next
    // step, so the next step's fragments are fetched INTO THE SAME REGISTERS while this step's MFMAs run: a0 / b2 right behind
    // the barrier (dead since groups 3 / 1), b0 / a2 during group 6, b1 / a1 behind group 6 -- no wave ever waits for LDS in
    // front of an MFMA.  The six DMAs of step kt + 2 go out as ONE burst behind group 3, in front of the waits and the barrier
    // (into the buffer of step kt - 1, free since the previous barrier): what a wave loses waiting for the address unit it would
    // have spent waiting at the barrier anyway.  Measured against the alternatives in one process (profiles/r03_x6p_ablation.txt:
    // one DMA per group, bursts behind the barrier at fetch distance 3, issue staggered over the waves): 4-11 % faster.
    // ONE branch-free body, the ring position a run-time scalar: with a body per case (or per ring position) the register allocator
    // moved the loop-carried accumulators between bodies through out-of-place MFMAs and spilled.
    //   u_off / NX / PV = byte offsets of the ring buffers of steps kt, kt + 1 and kt - 1 (= kt + 2)
    auto k_step = [&](int u_off, int NX, int PV, int kt) {
        XP_SB
        XP_GROUP(0, 2) XP_GROUP(0, 1) XP_GROUP(0, 0)
        // (round 4) The two waves of a SIMD, w and w + 4, run this program in lockstep between the same barriers: issued at the same
        // point, their DMA bursts wait for the address unit together while the SIMD's matrix pipe idles.  Waves 0-3 issue theirs in
        // FRONT of the barrier, waves 4-7 BEHIND it (into the same stage, free since the previous barrier either way): each half's
        // burst runs beside the other half's MFMAs.  Measured on the bf16 kernel of igemm_b16t.h (+4-8 %), and here:
        // profiles/r04_x6p_stagger.txt.  tune "dma_stagger" = 1 switches it off (A/B).
        if (!late) fill(PV, kt0 + kt + 2, kt + 2 < nk);
        XP_SB
        if constexpr (!(ABLM & 4)) {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // every fragment of this step has arrived (a1 was fetched last)
            // this wave's DMAs of step kt + 1 have landed (waves 0-3: those of kt + 2 have just left)
            if constexpr (!(ABLM & 32)) { if (late) wait_vmcnt<0>(); else wait_vmcnt<2 * XP_IPW>(); }
            if constexpr (!(ABLM & 16)) __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
        }
        if (late) fill(PV, kt0 + kt + 2, kt + 2 < nk);
        XP_SB
        XP_MF(1, 0, 0, 0) XP_RA(0, 0, NX) XP_MF(1, 0, 0, 1) XP_RA(0, 1, NX) XP_MF(1, 0, 0, 2) XP_RB(2, 0, NX) XP_MF(1, 0, 0, 3) XP_RB(2, 1, NX)
        XP_MF(1, 0, 1, 0) XP_RB(2, 2, NX) XP_MF(1, 0, 1, 1) XP_RB(2, 3, NX) XP_MF(1, 0, 1, 2) XP_MF(1, 0, 1, 3)
        XP_GROUP(2, 0)
        XP_MF(1, 1, 0, 0) XP_RB(0, 0, NX) XP_MF(1, 1, 0, 1) XP_RB(0, 1, NX) XP_MF(1, 1, 0, 2) XP_RB(0, 2, NX) XP_MF(1, 1, 0, 3) XP_RB(0, 3, NX)
        XP_MF(1, 1, 1, 0) XP_RA(2, 0, NX) XP_MF(1, 1, 1, 1) XP_RA(2, 1, NX) XP_MF(1, 1, 1, 2) XP_MF(1, 1, 1, 3)
        XP_RB(1, 0, NX) XP_RB(1, 1, NX) XP_RB(1, 2, NX) XP_RB(1, 3, NX) XP_RA(1, 0, NX) XP_RA(1, 1, NX)
    };

#pragma unroll
    for (int i = 0; i < TM; i++)
#pragma unroll
        for (int j = 0; j < TN; j++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[i][j][r] = 0.f;
    __syncthreads();   // the ring is free (SK: the previous piece's epilogue staged through it)
    fill(0, kt0, true);
    fill(XP_STAGE, kt0 + 1, 1 < nk);
    wait_vmcnt<2 * XP_IPW>();
    __builtin_amdgcn_s_barrier();   // step 0 has landed
    asm volatile("" ::: "memory");
    XP_RA0(0, 0, 0) XP_RA0(0, 1, 0) XP_RB0(2, 0, 0) XP_RB0(2, 1, 0) XP_RB0(2, 2, 0) XP_RB0(2, 3, 0)
    XP_RB0(1, 0, 0) XP_RB0(1, 1, 0) XP_RB0(1, 2, 0) XP_RB0(1, 3, 0) XP_RB0(0, 0, 0) XP_RB0(0, 1, 0) XP_RB0(0, 2, 0) XP_RB0(0, 3, 0)
    XP_RA0(1, 0, 0) XP_RA0(1, 1, 0) XP_RA0(2, 0, 0) XP_RA0(2, 1, 0)
    {
        int u_off = 0, pv = (NBUF - 1) * XP_STAGE;
        for (int kt = 0; kt < nk; kt++) {
            const int nx = u_off == (NBUF - 1) * XP_STAGE ? 0 : u_off + XP_STAGE;
            k_step(u_off, nx, pv, kt);
            pv = u_off;
            u_off = nx;
        }
    }
    // the zero-depositing DMAs of the last steps are still in flight, and so may be the last fetches: drain both before
    // the ring becomes the epilogue's staging area
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    wait_vmcnt<0>();
    __syncthreads();
#undef XP_SB
#undef XP_RA0
#undef XP_RB0
#undef XP_RA
#undef XP_RB
#undef XP_MF
#undef XP_GROUP

    float *slab_ = !SK && lvl >= 0 ? a.slab + ((long long)lvl * (a.m_end - a.m_begin) - a.m_begin) * a.N : nullptr;
    int lane_e = tid & 63;
    asm volatile("" : "+v"(lane_e));
    if constexpr (SK) {
        // Shared tiles are combined inside the launch, exactly as igemm2_kernel does it (igemm_v2.h, note at sk_start): a piece that
        // starts inside its tile (always the first of its workgroup) publishes its accumulators in register order to slab[workgroup]
        // with write-through stores and raises flag[workgroup]; the workgroup with the tile's head adds the following workgroups'
        // slabs in ascending k and runs the fused epilogue.  (This kernel has registers to spare: 8 loads in flight per lane.)
        const unsigned lane_off = (unsigned)(lane_e * 16);
        constexpr int kWaveFloats = kSkSlabFloats / XP_WAVES;
        if (lvl > 0) {
            const __amdgpu_buffer_rsrc_t sr = __builtin_amdgcn_make_buffer_rsrc(a.slab + (long long)sk_v * kSkSlabFloats + wave * kWaveFloats, 0, kWaveFloats * 4, 0x00020000);
#pragma unroll
            for (int i = 0; i < TM; i++)
#pragma unroll
                for (int j = 0; j < TN; j++)
#pragma unroll
                    for (int qd = 0; qd < 4; qd++) {
                        u32x4 v;
                        v.x = __float_as_uint(acc[i][j][qd * 4 + 0]); v.y = __float_as_uint(acc[i][j][qd * 4 + 1]);
                        v.z = __float_as_uint(acc[i][j][qd * 4 + 2]); v.w = __float_as_uint(acc[i][j][qd * 4 + 3]);
                        __builtin_amdgcn_raw_buffer_store_b128(v, sr, (int)lane_off, ((i * TN + j) * 4 + qd) * 1024, 16);   // aux 16 = sc1
                    }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // every storing wave, then the barrier, then ONE lane's flag
            __syncthreads();
            if (tid == 0 && !a.sk_drop_flag) __hip_atomic_store(((sk_flag_t *)a.sk_flags) + sk_v, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            continue;   // (nothing else to store; the next piece starts with a barrier)
        }
        if (lvl == 0) {
            const int sk_more = __builtin_amdgcn_readfirstlane(sk_owner((long long)(sk_u - nk) + nk_tile - 1, sk_units, (int)gridDim.x)) - sk_v;
            for (int l = 1; l <= sk_more; l++) {
                const sk_flag_t *f = ((const sk_flag_t *)a.sk_flags) + (sk_v + l);
                bool ok = false;
                for (unsigned spins = 0; spins < (a.sk_drop_flag ? (1u << 12) : kSkSpinMax); spins++) {
                    if (__hip_atomic_load(f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) { ok = true; break; }
                    __builtin_amdgcn_s_sleep(8);
                }
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");   // (no instruction: keeps the loads below the poll)
                const __amdgpu_buffer_rsrc_t sr = __builtin_amdgcn_make_buffer_rsrc(a.slab + (long long)(sk_v + l) * kSkSlabFloats + wave * kWaveFloats, 0, ok ? kWaveFloats * 4 : 0, 0x00020000);
                typedef float f32x4_ __attribute__((ext_vector_type(4)));
                typedef float f32x8_ __attribute__((ext_vector_type(8)));
#pragma unroll
                for (int t = 0; t < TM * TN; t += 2) {
                    f32x4_ v[8];
#pragma unroll
                    for (int e = 0; e < 8; e++)
                        v[e] = __builtin_bit_cast(f32x4_, __builtin_amdgcn_raw_buffer_load_b128(sr, (int)lane_off, (t * 4 + e) * 1024, 16));   // sc1
#pragma unroll
                    for (int u = 0; u < 2; u++) {
                        const f32x8_ lo = __builtin_shufflevector(v[4 * u], v[4 * u + 1], 0, 1, 2, 3, 4, 5, 6, 7);
                        const f32x8_ hi = __builtin_shufflevector(v[4 * u + 2], v[4 * u + 3], 0, 1, 2, 3, 4, 5, 6, 7);
                        acc[(t + u) / TN][(t + u) % TN] += __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
                // (a publisher that never arrived: a NaN in every 32 x 32 block, and the failure in the device status word -- igemm_v2.h)
                sk_bad |= __builtin_amdgcn_readfirstlane(ok ? 0 : 1);   // (raised at the end of the kernel)
#pragma unroll
                for (int i = 0; i < TM; i++)
#pragma unroll
                    for (int j = 0; j < TN; j++) acc[i][j][0] = ok ? acc[i][j][0] : __builtin_nanf("");
            }
        }
    }
    float *stage = reinterpret_cast<float *>(ldsb) + wave * (16 * (128 + 4));
    static_assert(XP_WAVES * 16 * (128 + 4) * 4 <= XP_LDS, "epilogue slabs must fit in the ring");
    // (after the last step's barrier nobody reads the ring and no DMA is in flight: the waves stage their accumulators through it)
    if (packed)
        store_acc_tile<TM, TN, 16, false, false, MapRows>(a, acc, stage, lane_e, m0 + wm * 64, n0 + wn * 128, slab_, MapRows{a.pk.rowmap});
    else
        store_acc_tile<TM, TN, 16, false>(a, acc, stage, lane_e, m0 + wm * 64, n0 + wn * 128, slab_);
    } while (SK && next_piece());
    if constexpr (SK) {
        if (sk_bad && a.status && (threadIdx.x & 63) == 0)
            __hip_atomic_fetch_or(a.status, EMADAPT_STATUS_STREAMK_TIMEOUT, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

template <bool SK, int ABLM = 0>
int launch_igemm_x6p(IgemmArgs &a, hipStream_t stream) {
#ifdef EMADAPT_ABLATE_BUILD
    if constexpr (ABLM == 0) {
        switch (tune(TUNE_ABLATE)) {
            case 1: return launch_igemm_x6p<SK, 1>(a, stream);
            case 2: return launch_igemm_x6p<SK, 2>(a, stream);
            case 3: return launch_igemm_x6p<SK, 3>(a, stream);
            case 4: return launch_igemm_x6p<SK, 4>(a, stream);
            case 7: return launch_igemm_x6p<SK, 7>(a, stream);
            case 8: return launch_igemm_x6p<SK, 8>(a, stream);
            case 9: return launch_igemm_x6p<SK, 9>(a, stream);
            case 10: return launch_igemm_x6p<SK, 10>(a, stream);
            case 12: return launch_igemm_x6p<SK, 12>(a, stream);
            case 14: return launch_igemm_x6p<SK, 14>(a, stream);
            case 16: return launch_igemm_x6p<SK, 16>(a, stream);
            case 32: return launch_igemm_x6p<SK, 32>(a, stream);
            case 48: return launch_igemm_x6p<SK, 48>(a, stream);
            default: break;
        }
    }
#endif
    static DynLdsAttr attr;
    int rc = attr.ensure(reinterpret_cast<const void *>(&igemm_x6p_kernel<SK, ABLM>), XP_LDS);
    if (rc) return rc;
    a.tiles_m = (int)ceil_div(a.m_end - a.m_begin, 256);
    a.tiles_n = (int)ceil_div(a.N, 256);
    if constexpr (SK) {
        const int nk_all = a.K / 16;
        const long long tiles = (long long)a.tiles_m * a.tiles_n;
        if (a.splits < 1 || tiles * nk_all > 0x7fffffffLL || tiles * 16 > 0x7fffffffLL) return EMADAPT_ERR_UNSUPPORTED;
        const int G = a.pk.n > 0 ? streamk_grid_units(a.pk.run_u0[a.pk.nruns], a.splits) : streamk_grid(tiles, nk_all, a.splits);
        const int levels = a.pk.n > 0 ? streamk_levels_packed(a, a.tiles_n, a.splits) : streamk_levels(tiles, nk_all, a.splits);
        if (a.pk.n > 0) {
            if (!a.pk.rowmap || a.m_begin != 0 || a.m_end != a.M) return EMADAPT_ERR_INVALID;
            hipLaunchKernelGGL(rowmap_kernel, dim3((unsigned)a.tiles_m), dim3(256), 0, stream, a, const_cast<int *>(a.pk.rowmap));
            rc = launch_status();
            if (rc) return rc;
        }
        if (levels > 0) {   // shared tiles: arrival words (zeroed ahead of every launch) + one slab per workgroup
            if (!a.slab || !a.sk_flags) return EMADAPT_ERR_WORKSPACE;
            EMADAPT_HIP_TRY(hipMemsetAsync(a.sk_flags, 0, streamk_flag_bytes(G), stream));
        }
        hipLaunchKernelGGL((igemm_x6p_kernel<true, ABLM>), dim3((unsigned)G), dim3(XP_WAVES * 64), XP_LDS, stream, a);
        return launch_status();
    } else {
        const long long nblk = (long long)a.tiles_m * a.tiles_n * a.splits;
        if (nblk > 0x7fffffffLL) return EMADAPT_ERR_UNSUPPORTED;
        hipLaunchKernelGGL((igemm_x6p_kernel<false, ABLM>), dim3((unsigned)nblk), dim3(XP_WAVES * 64), XP_LDS, stream, a);
        rc = launch_status();
        if (rc || a.splits <= 1) return rc;
        return launch_splitk_epilogue(a, stream);
    }
}

}  // namespace

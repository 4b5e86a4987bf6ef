// Second-generation fp32 implicit-GEMM kernel (forward and data gradient), included by conv_igemm.hip.
//
// Same arithmetic as igemm_kernel / igemm_ws_kernel (v_mfma_f32_32x32x2_f32, fp32 FMA chains, K order = (channel chunk,
// tap), bounds-checked gathers, fused epilogue, split-K slabs), re-laid-out around what the matrix pipe of gfx950
// tolerates (profiles/r02_mfma_loop_lab.txt):
//
//   * the fp32 MFMA runs on the vector ALU's datapath: every VALU / LDS instruction a MFMA wave issues costs ~10 cycles
//     of matrix time.  The round-1 loop (64x64 per wave, two ds_read2_b32 + two v_add per four MFMAs) tops out at 84-90 %;
//     a few wide LDS reads with immediate offsets per SIXTEEN MFMAs reach 97.5 %.  So every MFMA wave owns a 64 x 128
//     accumulator block (TM = 2, TN = 4 tiles of 32x32, 128 VGPRs) and its instruction stream is wide LDS reads and
//     MFMAs, nothing else;
//   * with two such dense MFMA waves per SIMD the LOADER waves become the critical path: ~116 vector instructions per
//     K step (loads, transposing LDS stores, address arithmetic) took longer to issue than the consumers' 256 MFMAs.
//     So the tiles are filled by LDS-DMA (buffer_load_dwordx4 ... lds): 16 instructions per K step and loader wave, no
//     data registers, no ds_write -- which means the tiles lie in LDS exactly as they lie in memory:
//         A  [row][k]     rows of BK floats (k contiguous), 16-byte chunks XOR-swizzled by the row
//         Bw [k][n]       forward: rows of BN floats (n contiguous)           -> one ds_read_b128 per k-pair
//            [n][k]       data gradient (filter read "transposed"): like A
//     A k-contiguous operand is consumed without a transpose by letting the MFMA's k slot follow the memory order: a
//     lane (l31, lh) reads the 8 bytes at columns 4q + 2*lh, 4q + 2*lh + 1 of its row and uses them as its operand of
//     k-pairs 2q and 2q + 1, i.e. k-pair 2q multiplies columns {4q, 4q+2}, k-pair 2q+1 columns {4q+1, 4q+3} -- the same
//     products, summed in a slightly different order than the round-1 kernels (both are valid fp32 sums; the parity
//     tests hold both to the same bar).  The forward filter tile is read at rows 4q + 2*lh + (kk & 1) accordingly.
//   * the loader waves issue their gathers with CONSTANT per-lane offsets (template parameter NT: one ready-made offset per
//     row group and tap, validity folded in as the out-of-range bit; what moves from K step to K step rides in the load's
//     scalar offset operand) and the MFMA waves' loop is unrolled over the ring (LDS addresses = register + immediate):
//     no wave issues a vector-ALU instruction in the steady state -- a VALU instruction of a LOADER wave waits for the
//     matrix pipe like anyone else's, and the 4-5 per gather of the first version made the loaders the slow side
//     (forward 90 -> 93.5 %, data gradient 89 -> 93 %, and 88 -> 98 % for the filter-gradient kernel built the same way);
//   * SK (stream-K): one workgroup per CU walks an equal share of all tiles' K steps (see sk_start below) -- for layers
//     whose tile count is not a whole number of rounds of the CUs;
//   * forward: the lane's four B tiles are four ADJACENT output columns (l31*4 + j, not l31 + 32*j): the epilogue is
//     direct 16-byte stores (512 contiguous bytes per row and wave), no staging through LDS.  The data gradient keeps the
//     l31 + 32*j columns (its filter tile is k-contiguous per column: adjacent columns would put a read's 32 lanes on 4
//     bank pairs) and the LDS-staged vector epilogue of the round-1 kernels.
//
// Configuration <WMW, WNW, BK, PW, NBUF, DIST>: consumer waves WMW x WNW (64 x 128 each), loader waves PW, LDS ring
//   <4, 2, 16, 4, 4, 3>  256 x 256 x 16, 12 waves, 128 KB LDS, one workgroup per CU
// (a 128 x 256 x 16 two-workgroups-per-CU configuration, <2, 2, 16, 2, 3, 2>, was built and measured 3 points slower
// than the round-1 wave-specialised kernel on conv4_x / conv5_x, and BK = 32 with two buffers ran at the mercy of the
// DMA latency: profiles/r02_fp32_tile_variants.txt.)
// Requirements (plan_launches falls back to the round-1 kernels otherwise): Cx % 16 == 0, KH*KW <= 32.
#pragma once
#include "igemm_common.h"

namespace {

typedef __attribute__((address_space(3))) void lds_void_t;

// LDS-DMA: 64 lanes x 16 bytes land at lds_base + 16 * lane (wave-uniform base), source = buffer + voffset per lane; an
// out-of-range offset (bit 31 set) deposits zeros like the register form of the load.
__device__ __forceinline__ void dma16(__amdgpu_buffer_rsrc_t r, float *lds_base, unsigned voffset) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (lds_void_t *)lds_base, 16, (int)voffset, 0, 0, 0);
}

// The same with a wave-uniform byte offset added by the address unit (SGPR operand of the instruction): the per-lane
// offset stays constant over the K loop, the loop only moves the scalar.
__device__ __forceinline__ void dma16s(__amdgpu_buffer_rsrc_t r, float *lds_base, unsigned voffset, unsigned soffset) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (lds_void_t *)lds_base, 16, (int)voffset, (int)soffset, 0, 0);
}

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// Stream-K work split (SK kernels): the K steps of all tiles, tile after tile, form one line of U = tiles * nk units that
// is cut into G equal pieces, one per workgroup (G = number of CUs): a workgroup runs the tail of one tile, possibly whole
// tiles, and the head of the next.  No CU idles in a partial last round, whatever the tile count.  A tile that several
// workgroups share is combined INSIDE the launch (round 4; rounds 2-3 wrote every piece to an [M][N] slab level and a second
// kernel, ~40 us per launch and 21 launches per step, added the levels and applied the epilogue):
//   * only the FIRST piece of a workgroup can start inside a tile (kt0 > 0: the tile's tail, or -- a share shorter than a
//     tile -- its middle); every later piece starts at k = 0.  So a workgroup publishes at most one partial tile, and it
//     does so early: its accumulators, in register order (64 lanes x 16 bytes per store: no staging), go to ITS slab
//     (slab[workgroup], 256 KB) with write-through (sc1) stores, every wave drains them, and after the piece's closing
//     barrier one lane raises flag[workgroup];
//   * the workgroup that runs the HEAD of a shared tile (k = 0: always the last piece of its share, finished when the
//     others' first pieces are long done) polls the flags of the next workgroups in turn, adds their slabs to its
//     accumulators -- head + next + next: ascending k, the order of the second kernel it replaces, so the sums are the
//     same bit for bit -- with sc1 loads (L1-bypassing; MI355X_MICROARCH.md, inter-workgroup visibility: sc1 payload,
//     drained, relaxed agent-scope flag, sc1 loads by the wave that polled) and runs the fused epilogue like a whole tile.
// The wait is bounded (kSkSpinMax polls of ~0.2 us: a publisher that never arrives -- it cannot, unless the launch shares the
// chip with a kernel that never ends -- poisons the tile with NaNs instead of hanging the queue); the flags are zeroed by a
// memset node ahead of every launch.
typedef __attribute__((address_space(1))) unsigned sk_flag_t;
constexpr unsigned kSkSpinMax = 1u << 23;
constexpr int kSkSlabFloats = 256 * 256;   // one workgroup's accumulators (8 waves x 64 lanes x 128 registers)
__host__ __device__ inline long long sk_start(long long v, long long U, int G) { return v * U / G; }
__host__ __device__ inline int sk_owner(long long u, long long U, int G) { return (int)(((u + 1) * G - 1) / U); }   // piece holding unit u

// NBUF LDS buffers, tiles are fetched DIST K steps ahead of their use (NBUF >= DIST + 1): a DMA has DIST - 1 whole K steps
// plus the rest of its own to land (with DIST = 1 the kernel ran at the mercy of the L2 / HBM latency: 66 % on conv4_x).
// NT > 0 (instantiated as 1; the value no longer matters): the loaders issue their gathers with a per-lane offset that is constant
// over the K loop -- what moves from step to step rides in the instruction's scalar offset -- plus the tap's validity bit (two
// vector instructions per gather; the loaders share their SIMD's issue port with two MFMA waves), and the MFMA waves' loop is
// unrolled over the ring.  NT = 0: per-step vector address arithmetic, run-time ring position (any shape the scalar form cannot
// address: specialised_taps() in conv_igemm.hip).
template <int WMW, int WNW, int BK, int PW, int NBUF, int DIST, bool BT, int NT, bool SK>
__global__ void __launch_bounds__((WMW * WNW + PW) * 64, 3) igemm2_kernel(IgemmArgs a) {
    constexpr int CONS = WMW * WNW;          // consumer (MFMA) waves
    constexpr int BM = WMW * 64, BN = WNW * 128;
    constexpr int TM = 2, TN = 4;
    constexpr int CPR = BK / 4;               // 16-byte chunks per k-contiguous row (8 or 4)
    constexpr int RPI = 64 / CPR;             // rows one DMA instruction fills (8 or 16)
    constexpr int A_BUF = BM * BK;            // floats per A buffer   [row][BK]
    constexpr int B_BUF = BN * BK;            // floats per B buffer   fwd [BK][BN], dgrad [col][BK]
    constexpr int NQ = BK / 4;                // 4-column groups (two k-pairs each) per K step
    constexpr int A_N = BM / RPI / PW;        // DMA instructions per loader wave and K step for A
    constexpr int B_N = BT ? BN / RPI / PW : BK * (BN / 256) / PW;   // fwd: one instruction = 256 floats of one k row
    static_assert(A_N >= 1 && B_N >= 1 && (BM / RPI) % PW == 0 && BN % 256 == 0, "tile / loader shape");
    static_assert(NBUF >= DIST + 1 && DIST >= 2 && (DIST - 2) * (A_N + B_N) <= 63 && (BK / 4) % 2 == 0, "ring depth / vmcnt range");
    // Swizzle of a k-contiguous row: chunk q of row r is stored at chunk position q ^ swz(r), swz in 0..3 (the low two
    // bits of the position: the addresses of chunks q and q + 4 differ by an immediate).  A rows: swz from the row bits
    // above those that share a 256-byte bank line; dgrad filter rows (read at rows 4*l31 + j): swz = l31 & 3.
    constexpr int SWA_SHIFT = CPR == 8 ? 1 : 2;

    extern __shared__ __attribute__((aligned(16))) float lds[];
    float *As = lds;                     // [NBUF][A_BUF]
    float *Bs = lds + NBUF * A_BUF;      // [NBUF][B_BUF]

    // Work of this workgroup: one (tile, K range) piece, or with SK a sequence of them (next_piece()).
    const int nk_all = a.K / BK;  // Cx % BK == 0: every K step is one tap x one BK-channel chunk
    int tile_m = 0, tile_n = 0, kt0 = 0, nk = 0;
    int lvl = -1;                 // slab level the piece's partial sums go to; -1: the whole reduction -> fused epilogue
    int sk_u = 0, sk_end = 0, sk_units = 0;   // (tiles * nk_all < 2^31: checked by the launcher)
    int sk_v = 0;
    {
        // XCD-aware order (see igemm_kernel): consecutive work items run on one XCD and share operands in its L2
        const int ntile = a.tiles_m * a.tiles_n;
        const int nblk = SK ? (int)gridDim.x : ntile * a.splits;
        const int bid = blockIdx.x;
        const int xcd = bid & 7, q = nblk >> 3, r = nblk & 7;
        int t = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
        if constexpr (SK) {
            // (wave-uniform values that come out of vector-ALU divisions are moved to scalar registers by hand: left in vector
            // registers, the piece list's state spilled the accumulators)
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
    int run = 0;            // run of the current tile (a.pk.n > 0: stream-K launches over packed rows, RowPack)
    int nk_tile = nk_all;   // K steps of the current tile (its run's tap box)
    auto next_piece = [&]() -> bool {   // SK: the next piece of [sk_u, sk_end); every wave of the workgroup walks the same list
        if (sk_u >= sk_end) return false;
        if (a.pk.n > 0) {
            while (run + 1 < a.pk.nruns && sk_u >= a.pk.run_u0[run + 1]) run++;
            nk_tile = (a.Cx / BK) * (int)a.pk.run_nkh[run] * (int)a.pk.run_nkw[run];
            const int rel = sk_u - a.pk.run_u0[run];
            const int t_in = __builtin_amdgcn_readfirstlane(rel / nk_tile);   // tile inside the run, N tiles fastest
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
        if (!next_piece()) return;   // (more workgroups than units: nothing to do, for every wave alike)
    }

    if (threadIdx.x >= CONS * 64) {
        // ================================ LOADER wavefronts ================================
        const int lane = threadIdx.x & 63;
        const int pw = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6) - CONS);   // loader wave index (uniform)
        // The tap offset travels in the instruction's scalar operand, which the address unit adds as an UNSIGNED number: the
        // buffer starts `bias` bytes before the tensor, so that offset + bias >= 0 for every tap (nothing below a.x is touched:
        // lanes whose tap falls outside the image carry the out-of-range bit).
        int min_dh = a.sign * (0 - a.pad_h), min_dw = a.sign * (0 - a.pad_w);
        {
            const int KH = a.ntaps / a.KW;
            const int dh1 = a.sign * ((KH - 1) * a.dil - a.pad_h), dw1 = a.sign * ((a.KW - 1) * a.dil - a.pad_w);
            min_dh = dh1 < min_dh ? dh1 : min_dh;
            min_dw = dw1 < min_dw ? dw1 : min_dw;
        }
        const int min_sp = min_dh * a.W + min_dw;
        const unsigned bias = NT > 0 && min_sp < 0 ? (unsigned)(-min_sp * a.Cx) * 4u : 0u;
        const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<char *>(reinterpret_cast<const char *>(a.x)) - bias, 0, (int)(a.x_bytes + bias), 0x00020000);
        const __amdgpu_buffer_rsrc_t wr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(a.w), 0, (int)a.w_bytes, 0x00020000);
        constexpr unsigned kOOB = 0x80000000u;
        // instruction j of this wave fills rows [(j*PW + pw)*RPI, +RPI) of the tile; lane -> (row in group, chunk position)
        const int rl = lane / CPR, pos = lane % CPR;
        do {   // one piece of work (SK: several)
        const bool packed = SK && a.pk.n > 0;
        const int m0 = a.m_begin + tile_m * BM, n0 = tile_n * BN;   // (packed: m0 is a GEMM row, a.pk.rowmap gives the pixel)
        // taps of this tile's K loop: filter rows [kh_first, kh_first + nkh) x columns [kw_first, kw_first + nkw), tap-inner order
        const int kh_first = packed ? a.pk.run_kh0[run] : 0, kw_first = packed ? a.pk.run_kw0[run] : 0;
        const int nkw_t = packed ? a.pk.run_nkw[run] : a.KW;
        const int ntaps_t = packed ? a.pk.run_nkh[run] * nkw_t : a.ntaps;
        // (the divisors are <= 25 and <= 5; one real division each per piece)
        const unsigned magic_nt_t = packed ? (ntaps_t > 1 ? (unsigned)((0x100000000ULL + (unsigned)ntaps_t - 1) / (unsigned)ntaps_t) : 0u) : a.magic_nt;
        const unsigned magic_nkw_t = packed ? (nkw_t > 1 ? (unsigned)((0x100000000ULL + (unsigned)nkw_t - 1) / (unsigned)nkw_t) : 0u) : a.magic_kw;
        unsigned a_off[A_N], a_bad[A_N];
#pragma unroll
        for (int j = 0; j < A_N; j++) {
            const int r = (j * PW + pw) * RPI + rl;
            const int m = packed ? a.pk.rowmap[m0 + r] : (m0 + r < a.m_end ? m0 + r : -1);
            const int q = pos ^ ((r >> SWA_SHIFT) & 3);   // the channel chunk this lane fetches
            if (m >= 0) {
                const int t2 = m / a.W;
                const int oh = t2 % a.H, ow = m - t2 * a.W;
                a_off[j] = (unsigned)(m * a.Cx + q * 4) * 4u;
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
        unsigned b_off[B_N];
        if constexpr (!BT) {
            // instruction j: k row (j*PW + pw) / (BN/256), 256-float segment (j*PW + pw) % (BN/256); lane = 4 floats
#pragma unroll
            for (int j = 0; j < B_N; j++) {
                const int idx = j * PW + pw;
                const int kr = idx / (BN / 256), seg = idx % (BN / 256);
                const int n = n0 + seg * 256 + lane * 4;
                b_off[j] = ((unsigned)(kr * a.N + n) * 4u) | (n < a.N ? 0u : kOOB);
            }
        } else {
#pragma unroll
            for (int j = 0; j < B_N; j++) {
                const int r = (j * PW + pw) * RPI + rl;
                const int n = n0 + r;
                const int q = pos ^ ((r >> SWA_SHIFT) & 3);
                b_off[j] = ((unsigned)(n * a.Cx + q * 4) * 4u) | (n < a.N ? 0u : kOOB);
            }
        }

        // Fill LDS buffer `buf` with the tile of K step kt (DMA; the caller awaits completion).
        auto fill = [&](int buf, int kt) {
            const unsigned chunk = __umulhi((unsigned)kt, magic_nt_t);
            const unsigned tap_l = ntaps_t > 1 ? (unsigned)kt - chunk * (unsigned)ntaps_t : 0u;   // tap inside the tile's tap box
            const unsigned chunk_u = ntaps_t > 1 ? chunk : (unsigned)kt;
            const unsigned khl = div_magic(tap_l, magic_nkw_t);
            const unsigned kh = (unsigned)kh_first + khl;
            const int kw = kw_first + (int)(tap_l - khl * (unsigned)nkw_t);
            const unsigned tap_u = kh * (unsigned)a.KW + (unsigned)kw;
            const int dh = a.sign * ((int)kh * a.dil - a.pad_h);
            const int dw = a.sign * (kw * a.dil - a.pad_w);
            const unsigned toff = (unsigned)(((dh * a.W + dw) * a.Cx + (int)chunk_u * BK) * 4);
            const unsigned sh = 31u - tap_u;   // moves this tap's "bad" bit to bit 31
            unsigned woff;
            if constexpr (!BT) woff = (unsigned)(((int)tap_u * a.Cx + (int)chunk_u * BK) * a.N) * 4u;
            else woff = (unsigned)((int)tap_u * a.N * a.Cx + (int)chunk_u * BK) * 4u;
            float *as = As + buf * A_BUF;
            float *bs = Bs + buf * B_BUF;
            if constexpr (NT > 0) {
                const unsigned s_x = (unsigned)__builtin_amdgcn_readfirstlane((int)(toff + bias));
                const unsigned s_w = (unsigned)__builtin_amdgcn_readfirstlane((int)woff);
#pragma unroll
                for (int j = 0; j < B_N; j++) {
                    if (ABL(16)) break;
                    if constexpr (!BT) dma16s(wr, bs + (j * PW + pw) * 256, b_off[j], s_w);
                    else dma16s(wr, bs + (j * PW + pw) * RPI * BK, b_off[j], s_w);
                }
                // the lane's offset is constant over the K loop (the tap's offset rides in the scalar operand); what changes is
                // whether the tap reads inside the image from this lane's pixel: its bit of a_bad moved to the out-of-range bit,
                // two vector instructions per gather.  (Rounds 2-3 kept one ready-made offset per (row group, tap) and picked it by
                // a scalar compare chain; the compiler turned that into an indexed array in SCRATCH, read back before every
                // step's gathers behind a vmcnt wait that also drained the fetches in flight: -0.3 ms per step, round 4.)
                if (!ABL(8)) {
#pragma unroll
                    for (int j = 0; j < A_N; j++) dma16s(xr, as + (j * PW + pw) * RPI * BK, a_off[j] | ((a_bad[j] << sh) & kOOB), s_x);
                }
                return;
            }
#pragma unroll
            for (int j = 0; j < (A_N > B_N ? A_N : B_N); j++) {
                if (j < A_N && !ABL(1)) dma16(xr, as + (j * PW + pw) * RPI * BK, (a_off[j] + toff) | ((a_bad[j] << sh) & kOOB));
                if (j < B_N && !ABL(1)) {
                    if constexpr (!BT) dma16(wr, bs + (j * PW + pw) * 256, b_off[j] + woff);
                    else dma16(wr, bs + (j * PW + pw) * RPI * BK, b_off[j] + woff);
                }
            }
        };
#pragma unroll
        for (int d = 0; d < DIST; d++)
            if (d < nk) fill(d, kt0 + d);
        // The consumers fetch the first fragments of tile kt + 1 while they still multiply tile kt (no LDS round trip in
        // front of the first MFMA after a barrier), so the barrier that ends step kt - 1 must already cover tile kt + 1:
        // the loaders wait for all but their DIST - 2 youngest fills.
        if (nk >= DIST) wait_vmcnt<(DIST - 2) * (A_N + B_N)>();   // tiles 0 and 1 have landed
        else wait_vmcnt<0>();
        __builtin_amdgcn_s_barrier();
        for (int kt = 0; kt < nk; kt++) {
            // Buffer (kt + DIST) % NBUF was last read for tile kt + DIST - NBUF <= kt - 1, whose reads were drained before the
            // barrier that ended step kt - 1: free.  Before the barrier that ends this step, tile kt + 2 must have landed.
            if (kt + DIST < nk) {
                fill((kt + DIST) % NBUF, kt0 + kt + DIST);
                wait_vmcnt<(DIST - 2) * (A_N + B_N)>();
            } else {
                wait_vmcnt<0>();
            }
            __builtin_amdgcn_s_barrier();
        }
        if constexpr (SK) __builtin_amdgcn_s_barrier();   // the MFMA waves are done with the ring (their epilogue may stage through it)
        } while (SK && next_piece());
        return;
    }

    // ================================ CONSUMER wavefronts ================================
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // (uniform: kept in a scalar register)
    const int wm = wave % WMW, wn = wave / WMW;
    const int l31 = lane & 31, lh = lane >> 5;
    f32x16 acc[TM][TN];
    // Fragment addresses (float indices inside a buffer) for the chunks q = 0..3; chunk q + 4 (BK = 32) is 16 floats on,
    // the wave's second row block 32 rows on, the dgrad filter's columns j one row on: all immediates.
    const int arow = wm * 64 + l31;
    int a_idx[4], b_idx[BT ? 4 : 1];
#pragma unroll
    for (int q = 0; q < 4; q++) a_idx[q] = arow * BK + ((q ^ ((arow >> SWA_SHIFT) & 3)) * 4) + lh * 2;
    if constexpr (BT) {
        // dgrad filter tile [n][k]: the lane's four columns are l31 + 32*j (consecutive rows across the lanes of a read:
        // 2-way bank conflicts like A; the adjacent-column mapping l31*4 + j puts all 32 lanes of a read on 4 bank
        // pairs -- 8-way, 2.4e9 conflict cycles per step in the first PMC pass)
#pragma unroll
        for (int q = 0; q < 4; q++) {
            // measured from the start of LDS and opaque to the optimiser: the rest of an address (ring position, column block)
            // is then below 64 KB and rides in the instruction's offset field; folded together with the B region's 64 KB start
            // it would not fit, and the compiler kept one address register per (chunk, ring position) -- and spilled
            b_idx[q] = NBUF * A_BUF + (wn * 128 + l31) * BK + ((q ^ ((l31 >> SWA_SHIFT) & 3)) * 4) + lh * 2;
            asm volatile("" : "+v"(b_idx[q]));
        }
    } else {
        b_idx[0] = (lh * 2) * BN + wn * 128 + l31 * TN;        // [k][n]: row 4q + 2*lh + (kk & 1)
    }
    // the operands of k-pairs 2q and 2q+1 come from one 8-byte read per k-contiguous row, one 16-byte read per [k][n] row
    // the K loop is unrolled over the ring wherever the loaders are the specialised (vector-ALU-free) ones; the generic
    // form keeps the round-2a loop (run-time ring position)
    constexpr bool UNROLL_RING = NT > 0;
    float2 av[2][TM];
    float2 bt[2][BT ? TN : 1];
    float4 bf[2][BT ? 1 : 2];
    auto read_q = [&](int buf, int q, int slot) {
        if (ABL(64)) return;
        const float *as = As + buf * A_BUF;
        const float *bs = (BT ? lds : Bs) + buf * B_BUF;   // (BT: b_idx already counts from the start of LDS)
        const int hi = (q >> 2) * 16;   // chunks 4..7 of a 32-float row
#pragma unroll
        for (int i = 0; i < TM; i++) av[slot][i] = *reinterpret_cast<const float2 *>(as + a_idx[q & 3] + hi + i * 32 * BK);
        if constexpr (BT) {
#pragma unroll
            for (int j = 0; j < TN; j++) bt[slot][j] = *reinterpret_cast<const float2 *>(bs + b_idx[q & 3] + hi + j * 32 * BK);
        } else {
            bf[slot][0] = *reinterpret_cast<const float4 *>(bs + b_idx[0] + (4 * q + 0) * BN);
            bf[slot][1] = *reinterpret_cast<const float4 *>(bs + b_idx[0] + (4 * q + 1) * BN);
        }
    };
    // one K step on buffer `cur`
    auto k_step = [&](int cur, int nxt, bool more) {
#pragma unroll
        for (int q = 0; q < NQ; q++) {
            // fragments one group ahead of the MFMAs; the last group of a step fetches the first one of the NEXT tile (it
            // landed before the previous barrier), so no LDS latency sits between a barrier and the MFMAs behind it
            if (q + 1 < NQ) read_q(cur, q + 1, (q + 1) & 1);
            else if (more) read_q(nxt, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            const int s = q & 1;
#pragma unroll
            for (int h = 0; h < 2; h++) {   // k-pair 2q + h
#pragma unroll
                for (int i = 0; i < TM; i++) {
                    const float av_ = h ? av[s][i].y : av[s][i].x;
#pragma unroll
                    for (int j = 0; j < TN; j++) {
                        float bv_;
                        if constexpr (BT) {
                            bv_ = h ? bt[s][j].y : bt[s][j].x;
                        } else {
                            const float4 b4 = bf[s][h];
                            bv_ = j == 0 ? b4.x : j == 1 ? b4.y : j == 2 ? b4.z : b4.w;
                        }
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av_, bv_, acc[i][j], 0, 0, 0);
                    }
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        __syncthreads();
    };
    int sk_bad = 0;   // (scalar) a publisher of a tile this workgroup heads never arrived
    do {   // one piece of work (SK: several)
    const bool packed = SK && a.pk.n > 0;
    const int m0 = a.m_begin + tile_m * BM, n0 = tile_n * BN;   // (packed: a GEMM row)
#pragma unroll
    for (int i = 0; i < TM; i++)
#pragma unroll
        for (int j = 0; j < TN; j++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[i][j][r] = 0.f;
    __syncthreads();
    if (nk > 0) read_q(0, 0, 0);
    if constexpr (UNROLL_RING) {
        // the ring position is a compile-time constant in the unrolled body: every LDS address is a register + immediate and
        // the MFMA waves issue no vector-ALU instruction at all between their reads and MFMAs
        for (int kt = 0; kt < nk; kt += NBUF) {
#pragma unroll
            for (int u = 0; u < NBUF; u++) {
                if (kt + u >= nk) break;
                k_step(u, (u + 1) % NBUF, kt + u + 1 < nk);
            }
        }
    } else {
        for (int kt = 0; kt < nk; kt++) k_step(kt % NBUF, (kt + 1) % NBUF, kt + 1 < nk);
    }

    float *slab_ = !SK && lvl >= 0 ? a.slab + ((long long)lvl * (a.m_end - a.m_begin) - a.m_begin) * a.N : nullptr;
    // the epilogue derives its lane coordinates afresh (opaque to the optimiser): hoisted out of the piece loop they would
    // stay live through the MFMA loop, which has no vector register to spare (the compiler then spills operand fragments)
    int lane_e = tid & 63;
    asm volatile("" : "+v"(lane_e));
    const int l31e = lane_e & 31, lhe = lane_e >> 5;
    if constexpr (SK) {
        // register v of the accumulator block (v = (i * TN + j) * 4 + quad) <-> bytes [(wave * 32 + v) * 1024 + lane * 16, +16) of a slab
        // (the wave's 32 KB are the buffer: wave and slab index ride in the descriptor's base, scalar)
        const unsigned lane_off = (unsigned)(lane_e * 16);
        constexpr int kWaveFloats = kSkSlabFloats / CONS;
        if (lvl > 0) {
            // a piece that starts inside its tile: publish the accumulators (see the note at sk_start)
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
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // every storing wave, ahead of the barrier that precedes the flag
        }
        if (lvl == 0) {
            // head of a shared tile: the rest arrives from the next sk_more workgroups
            // (the piece list has moved on: sk_u - nk is this tile's first unit)
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
#pragma unroll
                for (int i = 0; i < TM; i++)
#pragma unroll
                    for (int j = 0; j < TN; j++) {
                        typedef float f32x4_ __attribute__((ext_vector_type(4)));
                        typedef float f32x8_ __attribute__((ext_vector_type(8)));
                        f32x4_ v[4];
#pragma unroll
                        for (int qd = 0; qd < 4; qd++)
                            v[qd] = __builtin_bit_cast(f32x4_, __builtin_amdgcn_raw_buffer_load_b128(sr, (int)lane_off, ((i * TN + j) * 4 + qd) * 1024, 16));   // sc1
                        // (one 16-wide add: element-wise updates of the accumulator tuples cost the MFMA loop its registers)
                        const f32x8_ lo = __builtin_shufflevector(v[0], v[1], 0, 1, 2, 3, 4, 5, 6, 7), hi = __builtin_shufflevector(v[2], v[3], 0, 1, 2, 3, 4, 5, 6, 7);
                        acc[i][j] += __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15);
                        // (4 loads in flight per lane, 32 KB per workgroup; 8 -- two blocks at a time -- do not fit beside the accumulators:
                        //  the compiler then spills inside the MFMA loop)
                        __builtin_amdgcn_sched_barrier(0);
                    }
                // (the publisher never arrived -- see the note at sk_start: a NaN in every 32 x 32 block of the tile, which the epilogues
                // propagate, and the launch's failure in the device status word: the optimiser kernels then skip their update and
                // emadapt_status_poll reports it -- raised at the end of the kernel: an atomic HERE cost the MFMA loop 700 bytes of spills)
                sk_bad |= __builtin_amdgcn_readfirstlane(ok ? 0 : 1);
#pragma unroll
                for (int i = 0; i < TM; i++)
#pragma unroll
                    for (int j = 0; j < TN; j++) acc[i][j][0] = ok ? acc[i][j][0] : __builtin_nanf("");
            }
        }
    }
    if (ABL(32)) {   // (diagnostic builds: no epilogue -- one lane keeps the accumulators alive)
        if (lane_e == 77) a.y[0] = acc[0][0][0] + acc[1][3][15];
    } else if (SK && lvl > 0) {
        // (published above)
    } else if constexpr (BT) {
        // data gradient: columns l31 + 32*j -> the LDS-staged vector epilogue of the round-1 kernels (igemm_common.h); the
        // ring is dead after the last barrier
        static_assert(CONS * 16 * (128 + 4) <= NBUF * (A_BUF + B_BUF), "epilogue slabs must fit in the ring");
        if (packed)
            store_acc_tile<TM, TN, 16, false, false, MapRows, false>(a, acc, lds + wave * (16 * (128 + 4)), lane_e, m0 + wm * 64, n0 + wn * 128,
                                                                     slab_, MapRows{a.pk.rowmap});
        else
            store_acc_tile<TM, TN, 16, false, false, PixelRows, false>(a, acc, lds + wave * (16 * (128 + 4)), lane_e, m0 + wm * 64, n0 + wn * 128, slab_);
    } else if constexpr (SK) {
        // (forward, stream-K: the same staged epilogue with the adjacent-column mapping -- inside the piece loop the direct
        // form below made the compiler spill operand fragments in the MFMA loop)
        if (packed)
            store_acc_tile<TM, TN, 16, false, true, MapRows>(a, acc, lds + wave * (16 * (128 + 4)), lane_e, m0 + wm * 64, n0 + wn * 128,
                                                             slab_, MapRows{a.pk.rowmap});
        else
            store_acc_tile<TM, TN, 16, false, true>(a, acc, lds + wave * (16 * (128 + 4)), lane_e, m0 + wm * 64, n0 + wn * 128, slab_);
    } else {
        // ---- forward epilogue: a lane holds, for each of its 2 x 16 rows, four consecutive output columns -> 16-byte stores.
        // C/D layout of the 32x32 MFMA: row in tile = (r&3) + 8*(r>>2) + 4*lh, column = l31 (here: column l31*4 + j).
        const int col = n0 + wn * 128 + l31e * TN;
        float *slab = slab_;
        float4 bias4 = make_float4(0.f, 0.f, 0.f, 0.f);
        if (col < a.N && !slab && a.epi != EMADAPT_EPI_NONE && a.bias) bias4 = *reinterpret_cast<const float4 *>(a.bias + col);
        if (col < a.N) {   // N % 4 == 0: the quad is all in or all out
#pragma unroll
            for (int i = 0; i < TM; i++) {
#pragma unroll
                for (int r = 0; r < 16; r++) {
                    const int row = m0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhe;
                    if (row >= a.m_end) continue;
                    const long long o = (long long)row * a.N + col;
                    float v[4] = {acc[i][0][r], acc[i][1][r], acc[i][2][r], acc[i][3][r]};
                    if (slab) {
                        *reinterpret_cast<float4 *>(slab + o) = make_float4(v[0], v[1], v[2], v[3]);
                        continue;
                    }
                    v[0] += bias4.x; v[1] += bias4.y; v[2] += bias4.z; v[3] += bias4.w;   // (bias4 = 0 when there is no bias)
                    if (a.epi == EMADAPT_EPI_BIAS_RELU) {
#pragma unroll
                        for (int e = 0; e < 4; e++) v[e] = relu_keep_nan(v[e]);
                        if (a.keep < 1.0f) {
                            unsigned mk = 0;
                            if (a.mask) mk = *reinterpret_cast<const unsigned *>(a.mask + o);   // four keep bytes (o % 4 == 0)
                            {   // (o % 4 == 0: o .. o + 3 is a quad of the generator)
                                bool k[4];
                                if (a.mask) {
#pragma unroll
                                    for (int e = 0; e < 4; e++) k[e] = ((mk >> (8 * e)) & 0xFFu) != 0;
                                } else dropout_keep4(a, (unsigned long long)o, k);
#pragma unroll
                                for (int e = 0; e < 4; e++) v[e] = k[e] ? drop_scale(a, v[e]) : 0.f;
                            }
                        }
                    }
                    if (a.gate) {
                        const float4 g = *reinterpret_cast<const float4 *>(a.gate + o);
                        v[0] = gate_keep_nan(g.x, v[0] * a.scale);
                        v[1] = gate_keep_nan(g.y, v[1] * a.scale);
                        v[2] = gate_keep_nan(g.z, v[2] * a.scale);
                        v[3] = gate_keep_nan(g.w, v[3] * a.scale);
                    }
                    *reinterpret_cast<float4 *>(a.y + o) = make_float4(v[0], v[1], v[2], v[3]);
                }
            }
        }
    }
    if constexpr (SK) {
        __syncthreads();   // pairs with the loaders' barrier: the ring may be refilled
        if (lvl > 0 && tid == 0 && !a.sk_drop_flag)   // every wave's slab stores were drained ahead of the barrier
            __hip_atomic_store(((sk_flag_t *)a.sk_flags) + sk_v, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    } while (SK && next_piece());
    if constexpr (SK) {
        if (sk_bad && a.status && (threadIdx.x & 63) == 0)
            __hip_atomic_fetch_or(a.status, EMADAPT_STATUS_STREAMK_TIMEOUT, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

// Workgroups of a stream-K launch over U units: one per CU, but never more than units (every piece list is then non-empty, so
// the workgroups of a tile are consecutive and its slab levels dense).
inline int streamk_grid_units(long long U, int cus) { return (int)(U < cus ? (U > 0 ? U : 1) : cus); }
inline int streamk_grid(long long tiles, int nk_all, int cus) { return streamk_grid_units(tiles * nk_all, cus); }

// Workspace of a stream-K launch on G workgroups with shared tiles: [arrival words, padded][G slabs of 256 KB]
inline size_t streamk_flag_bytes(int G) { return ((size_t)G * sizeof(unsigned) + 255) / 256 * 256; }
inline size_t streamk_workspace_bytes(int G) { return streamk_flag_bytes(G) + (size_t)G * kSkSlabFloats * sizeof(float); }

// Slab levels a stream-K launch over `tiles` tiles of nk_all K steps on `cus` CUs needs (0: every tile is whole).
inline int streamk_levels(long long tiles, int nk_all, int cus) {
    const long long U = tiles * nk_all;
    const int G = streamk_grid(tiles, nk_all, cus);
    int levels = 0;
    for (long long t = 0; t < tiles; t++) {
        const int cnt = sk_owner((t + 1) * nk_all - 1, U, G) - sk_owner(t * nk_all, U, G) + 1;
        if (cnt > 1 && cnt > levels) levels = cnt;
    }
    return levels;
}

// The same for a launch over packed rows (RowPack: the tiles of run r have pack_run_nk(r) K steps).
inline int pack_run_nk(const IgemmArgs &a, int r) { return (a.Cx / 16) * a.pk.run_nkh[r] * a.pk.run_nkw[r]; }
inline int streamk_levels_packed(const IgemmArgs &a, int tiles_n, int cus) {
    const RowPack &pk = a.pk;
    const long long U = pk.run_u0[pk.nruns];
    const int G = streamk_grid_units(U, cus);
    int levels = 0;
    for (int r = 0; r < pk.nruns; r++) {
        const int nk = pack_run_nk(a, r);
        const long long nt = (long long)(pk.run_tile0[r + 1] - pk.run_tile0[r]) * tiles_n;
        for (long long t = 0; t < nt; t++) {
            const long long u0 = pk.run_u0[r] + t * nk;
            const int cnt = sk_owner(u0 + nk - 1, U, G) - sk_owner(u0, U, G) + 1;
            if (cnt > 1 && cnt > levels) levels = cnt;
        }
    }
    return levels;
}

// GEMM row -> pixel for a launch over packed rows: row R lies in the class c with row0[c] <= R < row0[c + 1]; inside the class
// the pixels go image by image, row by row of the class's box.  One thread per row; the table lives in the workspace.
__global__ void __launch_bounds__(256) rowmap_kernel(IgemmArgs a, int *map) {
    const int R = blockIdx.x * 256 + threadIdx.x;
    int pix = -1;
    if (R < a.M) {
        int c = 0;
        for (int i = 1; i < a.pk.n; i++) c = R >= a.pk.row0[i] ? i : c;   // (row0[] is a scalar load per step: <= 36 steps)
        const unsigned j = (unsigned)(R - a.pk.row0[c]);
        const unsigned now = a.pk.now[c];
        const unsigned img = div_magic(j, a.pk.magic_img[c]);
        const unsigned rem = j - img * ((unsigned)a.pk.noh[c] * now);
        const unsigned rh = div_magic(rem, a.pk.magic_now[c]);
        pix = ((int)img * a.H + a.pk.oh0[c] + (int)rh) * a.W + a.pk.ow0[c] + (int)(rem - rh * now);
    }
    map[R] = pix;
}

// One dimension of the classes: bands of output rows (resp. columns) that see the same filter rows (columns).
struct Band1D { int o0, no, k0, nk; };
inline int bands_1d(int size, int K, int dil, int pad, int sign, Band1D *out, int max_out) {
    int cuts[16], nc = 0;
    cuts[nc++] = 0;
    cuts[nc++] = size;
    for (int k = 0; k < K; k++) {
        const int d = sign * (k * dil - pad);   // input index = o + d: inside for o in [-d, size - d)
        const int lo = -d, hi = size - d;
        if (lo > 0 && lo < size) cuts[nc++] = lo;
        if (hi > 0 && hi < size) cuts[nc++] = hi;
    }
    for (int i = 1; i < nc; i++) {
        const int v = cuts[i];
        int j = i - 1;
        while (j >= 0 && cuts[j] > v) { cuts[j + 1] = cuts[j]; j--; }
        cuts[j + 1] = v;
    }
    int n = 0;
    for (int i = 0; i + 1 < nc; i++) {
        if (cuts[i + 1] == cuts[i]) continue;
        if (n == max_out) return -1;
        Band1D b = {cuts[i], cuts[i + 1] - cuts[i], -1, 0};
        for (int k = 0; k < K; k++) {
            const int in = b.o0 + sign * (k * dil - pad);
            if (in >= 0 && in < size) {
                if (b.k0 < 0) b.k0 = k;
                if (k != b.k0 + b.nk) return -1;   // (the inside filter rows of a band are consecutive)
                b.nk++;
            }
        }
        if (b.nk == 0) return -1;   // a band no filter row reaches: leave the layer to the plain kernels
        out[n++] = b;
    }
    return n;
}

// The row packing of a convolution (RowPack) for 256-row tiles and K steps of 16 channels; returns false (pk.n = 0) when the
// filter has nothing to skip, the classes / runs do not fit the struct, or fewer than 5 % of the products would be saved
// ("pack_pct"; measured, profiles/r03_fp32_experiments.txt: conv5_x, 3x3 rate 2 on 41x41, saves 6.2 % of its products and 3.8 /
// 2.1 % of its forward / data-gradient time; the rate-1 layers' 2.7 % buy 1.6 / 0.5 % -- inside the noise of the step).
// tiles_n = N tiles of the launch (a unit is one K step of one tile).
constexpr int kPackMinSavingPct = 5;
inline bool build_row_pack(IgemmArgs &a, int tiles_n, double *exec_frac = nullptr) {
    RowPack &pk = a.pk;
    pk.n = 0;
    pk.nruns = 0;
    pk.rowmap = nullptr;
    const int KH = a.ntaps / a.KW;
    if (KH > 5 || a.KW > 5 || a.ntaps < 2 || a.Cx % 16 != 0 || a.N % 4 != 0 || a.H > 65535 || a.W > 65535) return false;
    if ((long long)a.M * a.H * a.W >= (1LL << 32) || a.M % (a.H * a.W) != 0) return false;   // (div_magic ranges; whole images)
    const int nimg = a.M / (a.H * a.W);
    Band1D hb[6], wb[6];
    const int nh = bands_1d(a.H, KH, a.dil, a.pad_h, a.sign, hb, 6), nw = bands_1d(a.W, a.KW, a.dil, a.pad_w, a.sign, wb, 6);
    if (nh < 1 || nw < 1 || nh * nw < 2 || nh * nw > kMaxRowClasses) return false;
    // classes in boustrophedon order (row band outside; column bands left to right, then right to left, ...): consecutive
    // classes always differ in ONE band, so a tile that straddles a boundary pays for the smallest possible union (fc6 at
    // batch 16: 82.3 % of the dense products; plain row-major order, whose row changes jump across the map: 83.6 %)
    int n = 0;
    int ckh0[kMaxRowClasses], ckh1[kMaxRowClasses], ckw0[kMaxRowClasses], ckw1[kMaxRowClasses];
    long long rows = 0;
    for (int i = 0; i < nh; i++)
        for (int jj = 0; jj < nw; jj++) {
            const int j = (i & 1) ? nw - 1 - jj : jj;
            const Band1D &h = hb[i], &w = wb[j];
            pk.row0[n] = (int)rows;
            pk.oh0[n] = (unsigned short)h.o0; pk.noh[n] = (unsigned short)h.no;
            pk.ow0[n] = (unsigned short)w.o0; pk.now[n] = (unsigned short)w.no;
            pk.magic_img[n] = magic_u32(h.no * w.no);
            pk.magic_now[n] = magic_u32(w.no);
            ckh0[n] = h.k0; ckh1[n] = h.k0 + h.nk; ckw0[n] = w.k0; ckw1[n] = w.k0 + w.nk;
            rows += (long long)nimg * h.no * w.no;
            n++;
        }
    pk.row0[n] = (int)rows;   // == a.M: every pixel exactly once
    const long long tiles_m = ceil_div(a.M, 256);
    if (tiles_m * tiles_n * (long long)(a.K / 16) > 0x7fffffffLL) return false;
    // runs of tiles with the same union box; cost in (tile rows x taps)
    long long cost = 0, units = 0;
    int nruns = 0, c_lo = 0;
    for (long long t = 0; t < tiles_m; t++) {
        const long long lo = t * 256, hi = (t + 1) * 256 < a.M ? (t + 1) * 256 : a.M;
        while (pk.row0[c_lo + 1] <= lo) c_lo++;
        int kh0 = 99, kh1 = 0, kw0 = 99, kw1 = 0;
        for (int c = c_lo; c < n && pk.row0[c] < hi; c++) {
            kh0 = ckh0[c] < kh0 ? ckh0[c] : kh0; kh1 = ckh1[c] > kh1 ? ckh1[c] : kh1;
            kw0 = ckw0[c] < kw0 ? ckw0[c] : kw0; kw1 = ckw1[c] > kw1 ? ckw1[c] : kw1;
        }
        const int nkh = kh1 - kh0, nkw = kw1 - kw0;
        cost += 256LL * nkh * nkw;
        if (nruns == 0 || pk.run_kh0[nruns - 1] != kh0 || pk.run_nkh[nruns - 1] != nkh || pk.run_kw0[nruns - 1] != kw0 ||
            pk.run_nkw[nruns - 1] != nkw) {
            if (nruns == kMaxTileRuns) return false;
            pk.run_tile0[nruns] = (int)t;
            pk.run_u0[nruns] = (int)units;
            pk.run_kh0[nruns] = (unsigned char)kh0; pk.run_nkh[nruns] = (unsigned char)nkh;
            pk.run_kw0[nruns] = (unsigned char)kw0; pk.run_nkw[nruns] = (unsigned char)nkw;
            nruns++;
        }
        units += (long long)tiles_n * (a.Cx / 16) * nkh * nkw;
    }
    pk.run_tile0[nruns] = (int)tiles_m;
    pk.run_u0[nruns] = (int)units;
    const long long dense = tiles_m * 256 * (long long)a.ntaps;
    const int min_saving = tune(TUNE_PACK_PCT) > 0 ? tune(TUNE_PACK_PCT) : kPackMinSavingPct;
    if (cost * 100 > dense * (100 - min_saving)) return false;
    pk.nruns = nruns;
    pk.n = n;
    if (exec_frac) *exec_frac = (double)cost / (double)dense;   // (tile row, tap) pairs walked / all of them
    return true;
}

template <int WMW, int WNW, int BK, int PW, int NBUF, int DIST, bool BT, int NT, bool SK>
int launch_igemm2(IgemmArgs &a, hipStream_t stream) {
    constexpr int BM = WMW * 64, BN = WNW * 128;
    constexpr int THREADS = (WMW * WNW + PW) * 64;
    constexpr size_t lds = sizeof(float) * NBUF * (BM + BN) * BK;
    static DynLdsAttr attr;
    int rc = attr.ensure(reinterpret_cast<const void *>(&igemm2_kernel<WMW, WNW, BK, PW, NBUF, DIST, BT, NT, SK>), lds);
    if (rc) return rc;
    a.tiles_m = (int)ceil_div(a.m_end - a.m_begin, BM);
    a.tiles_n = (int)ceil_div(a.N, BN);
    if constexpr (SK) {
        // a.splits = CUs to lay the pieces out for; a.sk_flags / a.slab = the workspace of streamk_workspace_bytes(G) when tiles are shared
        const int nk_all = a.K / BK;
        const long long tiles = (long long)a.tiles_m * a.tiles_n;
        if (a.splits < 1 || tiles * nk_all > 0x7fffffffLL || tiles * 16 > 0x7fffffffLL) return EMADAPT_ERR_UNSUPPORTED;
        const int G = a.pk.n > 0 ? streamk_grid_units(a.pk.run_u0[a.pk.nruns], a.splits) : streamk_grid(tiles, nk_all, a.splits);
        const int levels = a.pk.n > 0 ? streamk_levels_packed(a, a.tiles_n, a.splits) : streamk_levels(tiles, nk_all, a.splits);
        if (a.pk.n > 0) {   // the row table first (a.pk.rowmap points into the workspace: dispatch_igemm)
            if (!a.pk.rowmap || a.m_begin != 0 || a.m_end != a.M) return EMADAPT_ERR_INVALID;
            hipLaunchKernelGGL(rowmap_kernel, dim3((unsigned)a.tiles_m), dim3(256), 0, stream, a, const_cast<int *>(a.pk.rowmap));
            rc = launch_status();
            if (rc) return rc;
        }
        if (levels > 0) {   // shared tiles: arrival words (zeroed here, ahead of every launch) + one slab per workgroup
            if (!a.slab || !a.sk_flags) return EMADAPT_ERR_WORKSPACE;
            EMADAPT_HIP_TRY(hipMemsetAsync(a.sk_flags, 0, streamk_flag_bytes(G), stream));
        }
        hipLaunchKernelGGL((igemm2_kernel<WMW, WNW, BK, PW, NBUF, DIST, BT, NT, true>), dim3((unsigned)G), dim3(THREADS), lds, stream, a);
        return launch_status();
    } else {
        const long long nblk = (long long)a.tiles_m * a.tiles_n * a.splits;
        if (nblk > 0x7fffffffLL) return EMADAPT_ERR_UNSUPPORTED;
        hipLaunchKernelGGL((igemm2_kernel<WMW, WNW, BK, PW, NBUF, DIST, BT, NT, false>), dim3((unsigned)nblk), dim3(THREADS), lds, stream, a);
        rc = launch_status();
        if (rc || a.splits <= 1) return rc;
        return launch_splitk_epilogue(a, stream);
    }
}

}  // namespace

// bf16 forward / data-gradient kernel for the 3 x 3 layers with 128 and fewer output columns (conv1_2, conv2_x, the data gradients
// into 64 / 128 channels), included by conv_bf16.hip (round 5).  "p3" = patch, 3 x 3.
//
// What these layers cost the implicit-GEMM kernels is not arithmetic: conv1_2's MFMAs need 0.05 ms of the pipe and the layer moves
// 422 MB (0.08 ms of HBM) but ran 0.27 ms, because the im2col gather fetches every activation NINE times -- once per tap, in 64-byte
// halves of a pixel's 128-byte row (profiles/r04_bf16t_ablation.txt section 4: the gather alone 0.33 ms).  Here the input PATCH of an
// output tile -- the tile's pixels plus a halo of `dil` -- is staged in LDS once per group of 32 channels and all nine taps read their
// A fragments out of it at a shifted row; only the filter streams per tap (it is a few hundred KB, L2-resident, shared by all tiles).
//
//   * tile = TH x TW output pixels of one image, TH TW <= 256 (host: the shape that wastes least of the map); patch = (TH + 2 dil) x
//     (TW + 2 dil) pixel rows of 64 B (32 channels), <= 352 rows; four | three patch stages (one being read, the next groups' / tiles'
//     being filled: a group is nine short steps, a DMA round trip longer than that).  Traffic per 256 pixels and channel group: <= 22 KB of patch + 9 x N x 64 B of filter, against 9 x 16 KB + the same.
//   * one workgroup of 8 waves per CU, PERSISTENT over its tiles (tile = blockIdx, + gridDim, ...): 4 MFMA waves (64 pixels x all N
//     columns each: 4 A fragments x NBF B fragments of the 16x16x32 MFMA, NBF = 4 | 8 for N <= 64 | 128) and 4 DMA waves -- two stream
//     the filter (ring of 10 | 7 stages of N x 64 B, 8 | 5 steps ahead), two fill the patch stages of the groups behind the current
//     one.  The streams do not stop at tile boundaries: while the MFMA waves run a tile's epilogue (through an LDS staging area of its
//     own) the next tile's patch and first filter stages are already in flight -- a tile is only 18 - 36 K steps long, and round 4's
//     tall-tile attempt lost exactly there (per-tile fill and drain in the open).
//   * A fragments: lane (l15, lq) reads k chunk lq of the patch row of ITS output pixel shifted by the tap: row = r0[pixel] + (kh dil)
//     PW + kw dil.  The rows of a fragment are consecutive wherever the fragment does not cross a tile row, at ANY offset, so the LDS
//     swizzle must be conflict-free at any row offset: slot s of row r holds chunk s ^ ((r >> 1) & 2) (searched exhaustively: the
//     swizzles of the form f((r >> 2) & 3) that are conflict-free for the 16x16x32 read at every offset are exactly (0,2,0,2) and its
//     seven relabelings; igemm_b16t.h's (0,3,2,1) is 2-way off alignment).  In address bits: bit 5 ^= bit 8 -- four VALU per read.
//   * the data gradient is the same kernel over dy with the taps mirrored (a.sign) and the [tap][ci][co] filter image.
// Requirements: 3 x 3, dil 1 | 2, x stored bf16, Cx % 32 == 0, N <= 128 and N % 8 == 0, patch <= 352 rows.  Reads the plain bf16 filter
// images of emadapt_weights_to_bf16 ([tap][row][Cxp]): no new image, no new ABI entry -- emadapt_conv_{fwd,dgrad}_bf16 pick it.
#pragma once

namespace {

constexpr int P3_PATCH_ROWS = 352;
constexpr int P3_PATCH_BYTES = P3_PATCH_ROWS * 64;   // one patch stage
constexpr int P3_THREADS = 8 * 64;
constexpr int P3_NIW = 11;   // DMA instructions per patch wave and patch stage (22 KB / 2 waves): always all of them, so that vmcnt counts are constants
// A K step of these layers is 16 (N <= 64) or 32 MFMAs of ONE wave per SIMD -- 0.12 / 0.25 us -- while a DMA round trip under load is
// 1 - 2 us: the streams must run MANY steps ahead (the first version, one group and three filter steps ahead like igemm_b16t.h, was
// slower than the im2col kernel it replaces: 0.63 us per step).  Patch stages / filter ring depth by NBF:
template <int NBF> struct P3Cfg {
    static constexpr int NPATCH = 4;                     // the current group's patch + 3 groups in flight (27 steps)
    static constexpr int NBUF = NBF == 4 ? 12 : 8;       // filter ring (4 | 8 KB stages), fetched NBUF - 2 steps ahead
    static constexpr int AHEAD = NBUF - 2;
};

struct P3Geo {
    int TH, TW, PW, PP;          // tile, patch width, patch rows
    int tiles_x, tiles_y, tiles;  // per image / total
    int nimg;
    unsigned magic_tw, magic_pw, magic_txy, magic_tx;
};

__host__ __device__ inline int p3_swz(int r) { return (r >> 1) & 2; }

// tile shape: TH x TW <= 256 pixels, patch <= 352 rows, least MFMA work wasted (tiles x 256 over H x W), then least halo
inline bool p3_geometry(int B, int H, int W, int dil, P3Geo &g) {
    long long best = -1;
    for (int tw = 4; tw <= 64; tw++)
        for (int th = 2; th * tw <= 256; th++) {
            const int pp = (th + 2 * dil) * (tw + 2 * dil);
            if (pp > P3_PATCH_ROWS) continue;
            const long long tiles = (long long)((H + th - 1) / th) * ((W + tw - 1) / tw);
            const long long cost = tiles * 256 * 1024 + tiles * pp;   // (MFMA work first -- a tile costs 256 rows whatever it holds -- halo traffic second)
            if (best < 0 || cost < best) {
                best = cost;
                g.TH = th; g.TW = tw;
            }
        }
    if (best < 0) return false;
    g.PW = g.TW + 2 * dil;
    g.PP = (g.TH + 2 * dil) * g.PW;
    g.tiles_x = (W + g.TW - 1) / g.TW;
    g.tiles_y = (H + g.TH - 1) / g.TH;
    g.nimg = B;
    const long long tiles = (long long)g.tiles_x * g.tiles_y * B;
    if (tiles > 0x3fffffffLL) return false;
    g.tiles = (int)tiles;
    g.magic_tw = magic_u32(g.TW);
    g.magic_pw = magic_u32(g.PW);
    g.magic_txy = magic_u32(g.tiles_x * g.tiles_y);
    g.magic_tx = magic_u32(g.tiles_x);
    return true;
}

inline bool p3_ok(const IgemmArgs &a, bool x_bf16) {
    if (!x_bf16 || a.ntaps != 9 || a.KW != 3 || (a.dil != 1 && a.dil != 2) || a.Cx % 32 != 0 || a.N > 128 || a.N % 8 != 0) return false;
    if (a.M % (a.H * a.W) != 0 || (long long)a.M * a.Cx * 2 >= (1LL << 31) || (long long)a.M * a.N * 4 >= (1LL << 40)) return false;
    P3Geo g;
    return p3_geometry(a.M / (a.H * a.W), a.H, a.W, a.dil, g);
}

template <int NBF>
constexpr size_t p3_lds_bytes() { return P3Cfg<NBF>::NPATCH * P3_PATCH_BYTES + (size_t)P3Cfg<NBF>::NBUF * NBF * 16 * 64 + 512; }   // (+ the bias)
static_assert(p3_lds_bytes<4>() <= 160 * 1024 && p3_lds_bytes<8>() <= 160 * 1024, "LDS budget");

// NBF = B fragments (16 filter rows each) per step: 4 (N <= 64) | 8 (N <= 128)
// ABLM (diagnostic library only, tune "ablate"): parts COMPILED out -- 1 no patch DMAs, 2 no filter DMAs, 4 no MFMAs, 8 no epilogue, 16 no A
// fragment reads, 32 no B fragment reads; results are garbage, scripts/p3_ablate.py reads the time.
template <int NBF, bool YBF, int ABLM = 0>
__global__ void __launch_bounds__(P3_THREADS, 2) igemm_p3_kernel(IgemmArgs a, int Cxp, P3Geo geo) {
    constexpr int BN = NBF * 16, BSTAGE = BN * 64, NBUF = P3Cfg<NBF>::NBUF, AHEAD = P3Cfg<NBF>::AHEAD, NPATCH = P3Cfg<NBF>::NPATCH;
    constexpr int B_BASE = NPATCH * P3_PATCH_BYTES, BIAS_BASE = B_BASE + NBUF * BSTAGE;
    extern __shared__ __attribute__((aligned(16))) char ldsb[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int G = (int)gridDim.x, bid = (int)blockIdx.x;
    const int my_tiles = bid < geo.tiles ? (geo.tiles - 1 - bid) / G + 1 : 0;
    if (my_tiles == 0) return;
    if (tid < 128)   // the bias (zeros without one) where every epilogue reads it: LDS, 512 bytes (visible behind the prologue's barrier)
        reinterpret_cast<float *>(ldsb + BIAS_BASE)[tid] = (a.epi != EMADAPT_EPI_NONE && a.bias && tid < a.N) ? a.bias[tid] : 0.f;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // (written before this wave joins the prologue's barrier)
    const int NG = a.Cx / 32;
    const int steps_per_tile = NG * 9;
    const int total_steps = my_tiles * steps_per_tile, total_groups = my_tiles * NG;
    const int d = a.dil;

    if (wave >= 6) {
        // ======================================= patch waves (2) =======================================
        const int pw = wave - 6;
        constexpr int NIW = P3_NIW, D = NPATCH - 1;   // D groups in flight behind the one being read
        const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(a.x), 0, (int)a.x_bytes, 0x00020000);
        constexpr unsigned kOOB = 0x80000000u;
        unsigned off[NIW];
        auto tile_offsets = [&](int ti) {   // per-lane source offsets of this wave's instructions for tile ti (channel group 0)
            const int t = bid + ti * G;
            const unsigned img = div_magic((unsigned)t, geo.magic_txy);
            const unsigned rem = (unsigned)t - img * (unsigned)(geo.tiles_x * geo.tiles_y);
            const unsigned tyi = div_magic(rem, geo.magic_tx);
            const int y0 = (int)tyi * geo.TH, x0 = (int)(rem - tyi * (unsigned)geo.tiles_x) * geo.TW;
#pragma unroll
            for (int n = 0; n < NIW; n++) {
                const int s = ((pw * NIW + n) << 6) + lane;
                const int r = s >> 2, chunk = (s & 3) ^ p3_swz(r);
                const unsigned py = div_magic((unsigned)r, geo.magic_pw);
                const int px = r - (int)py * geo.PW;
                const int iy = y0 - d + (int)py, ix = x0 - d + px;
                const bool ok = ti < my_tiles && r < geo.PP && (unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W;
                off[n] = ok ? (unsigned)((((int)img * a.H + iy) * a.W + ix) * a.Cx * 2 + chunk * 16) : kOOB;
            }
        };
        // instructions [n0, n1) of group qq's patch into its stage (groups past the end: every lane out of range -- the same count)
        auto issue = [&](int qq, int stage, int n0, int n1) {
            const unsigned s_g = (unsigned)(qq % NG) * 64u;
            const unsigned dead = qq < total_groups ? 0u : kOOB;
            char *dst = ldsb + stage * P3_PATCH_BYTES + pw * NIW * 1024;
            if constexpr (ABLM & 1) return;
#pragma unroll
            for (int n = 0; n < NIW; n++)
                if (n >= n0 && n < n1)
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(xr, (lds_void_t *)(dst + n * 1024), 16, (int)(off[n] | dead), (int)s_g, 0, 0);
        };
        for (int qq = 0; qq < D; qq++) {   // groups 0 .. D - 1 in front of the loop
            if (qq % NG == 0) tile_offsets(qq / NG);
            issue(qq, qq, 0, NIW);
        }
        wait_vmcnt<(D - 1) * NIW>();
        __builtin_amdgcn_s_barrier();   // group 0's patch (and the filter's step 0) have landed
        int q = 0, t = 0, fst = D;      // fst: the stage group q + D goes into (= the stage of group q - 1)
        for (int s = 0; s < total_steps; s++) {
            if (t == 0 && (q + D) % NG == 0) tile_offsets((q + D) / NG);   // the group fetched during this one opens a new tile
            if (t < 6) issue(q + D, fst, t * 2, t * 2 + 2);
            if (t == 8) wait_vmcnt<(D - 1) * NIW>();   // group q + 1 has landed: its first A fragments are read behind this step's barrier
            __builtin_amdgcn_s_barrier();
            if (++t == 9) { t = 0; q++; fst = fst == NPATCH - 1 ? 0 : fst + 1; }
        }
        wait_vmcnt<0>();
        return;
    }
    if (wave >= 4) {
        // ======================================= filter waves (2) =======================================
        constexpr int IPW = NBF / 2;   // DMA instructions per wave and step (16 filter rows each)
        const int lw = wave - 4;
        const __amdgpu_buffer_rsrc_t wr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(a.w), 0, (int)a.w_bytes, 0x00020000);
        constexpr unsigned kOOB = 0x80000000u;
        unsigned b_off[IPW];
#pragma unroll
        for (int j = 0; j < IPW; j++) {
            const int row = (lw * IPW + j) * 16 + (lane >> 2);
            const int chunk = (lane & 3) ^ p3_swz(row);
            b_off[j] = row < a.N ? (unsigned)(row * Cxp + chunk * 8) * 2u : kOOB;
        }
        auto fill = [&](int stage, int s) {   // step s of this workgroup's sequence: (tile, group, tap) -> the filter's (tap, group)
            const int st = s % steps_per_tile;
            const unsigned g = (unsigned)st / 9u, tap = (unsigned)st - g * 9u;
            const unsigned s_w = (unsigned)__builtin_amdgcn_readfirstlane((int)((tap * (unsigned)a.N * (unsigned)Cxp + g * 32u) * 2u));
            const unsigned dead = s < total_steps ? 0u : kOOB;
            char *bs = ldsb + B_BASE + stage * BSTAGE + lw * (IPW * 1024);
            if constexpr (ABLM & 2) return;
#pragma unroll
            for (int j = 0; j < IPW; j++)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(wr, (lds_void_t *)(bs + j * 1024), 16, (int)(b_off[j] | dead), (int)s_w, 0, 0);
        };
#pragma unroll
        for (int s = 0; s < AHEAD; s++) fill(s, s);
        wait_vmcnt<(AHEAD - 1) * IPW>();
        __builtin_amdgcn_s_barrier();   // step 0 has landed
        int dst = AHEAD;
        for (int s = 0; s < total_steps; s++) {
            fill(dst, s + AHEAD);
            wait_vmcnt<(AHEAD - 1) * IPW>();   // step s + 1 has landed
            __builtin_amdgcn_s_barrier();
            dst = dst == NBUF - 1 ? 0 : dst + 1;
        }
        wait_vmcnt<0>();
        return;
    }

    // ======================================= MFMA waves (4) =======================================
    const int l15 = lane & 15, lq = lane >> 4;
    // the lane's output pixel of A fragment i: p = wave * 64 + i * 16 + l15 -> (ty, tx) in the tile -> patch row of tap (0, 0)
    int a0[4];
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const int p = wave * 64 + i * 16 + l15;
        const unsigned ty = div_magic((unsigned)p, geo.magic_tw);
        const int tx = p - (int)ty * geo.TW;
        const int r0 = (int)ty < geo.TH ? (int)ty * geo.PW + tx : 0;   // (pixels past the tile read row 0: discarded in the epilogue)
        a0[i] = (r0 << 6) + (lq << 4);
    }
    const int b0 = B_BASE + l15 * 64 + ((lq ^ p3_swz(l15)) << 4);   // + j * 1024 per fragment, + stage * BSTAGE
    bt_f32x4 acc[4][NBF];
    bt_bf16x8 av[2][4], bv[3];
    if constexpr (ABLM != 0) {   // (defined operands for the builds that skip reads)
#pragma unroll
        for (int e = 0; e < 8; e++) {
            const __bf16 c = (__bf16)((float)((lane * 7 + e) & 63) * 0.03125f - 1.0f);
#pragma unroll
            for (int i = 0; i < 4; i++) { av[0][i][e] = c; av[1][i][e] = c; }
            bv[0][e] = c; bv[1][e] = c; bv[2][e] = c;
        }
    }
    auto zero_acc = [&]() {
#pragma unroll
        for (int i = 0; i < 4; i++)
#pragma unroll
            for (int j = 0; j < NBF; j++)
#pragma unroll
                for (int r = 0; r < 4; r++) acc[i][j][r] = 0.f;
    };
    zero_acc();
    // tap -> patch-row shift in bytes (the data gradient mirrors the taps)
    auto tap_shift = [&](int tap) -> int {
        const int kh = tap / 3, kw = tap - kh * 3;
        const int sh = a.sign > 0 ? kh : 2 - kh, sw = a.sign > 0 ? kw : 2 - kw;
        return (sh * d * geo.PW + sw * d) << 6;
    };
#define P3_SB __builtin_amdgcn_sched_barrier(0);
#define P3_RDA(P, I, PBASE, SHIFT) if constexpr (!(ABLM & 16)) { const int t1_ = a0[I] + (SHIFT); av[P][I] = *reinterpret_cast<const bt_bf16x8 *>(ldsb + (PBASE) + (t1_ ^ ((t1_ >> 3) & 32))); P3_SB }
#define P3_RDB(S, J, STG) if constexpr (!(ABLM & 32)) { bv[S] = *reinterpret_cast<const bt_bf16x8 *>(ldsb + (STG) * BSTAGE + b0 + (J) * 1024); P3_SB }
// (the FILTER fragment is the MFMA's A operand: D = [16 output channels][16 pixels], so a lane ends up with 4 CONSECUTIVE channels --
// rows 4 lq + r -- of ONE pixel -- column l15: the epilogue stores them straight from registers, 8 bytes per lane and block)
#define P3_MF(P, I, J, S) if constexpr (!(ABLM & 4)) { acc[I][J] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bv[S], av[P][I], acc[I][J], 0, 0, 0); P3_SB }
    __builtin_amdgcn_s_barrier();   // group 0's patch and the filter's step 0 have landed
    asm volatile("" ::: "memory");
    {
        const int sh0 = tap_shift(0);
        P3_RDA(0, 0, 0, sh0) P3_RDA(0, 1, 0, sh0) P3_RDA(0, 2, 0, sh0) P3_RDA(0, 3, 0, sh0)
        P3_RDB(0, 0, 0) P3_RDB(1, 1, 0)
    }
    int cur = 0;            // filter ring stage of this step
    int t = 0, q = 0;       // tap, group (of this workgroup's sequence)
    int pst = 0;            // patch stage of group q
    int done = 0;           // tiles finished
    // One K step, P = the A buffer it multiplies (the other one is filled for the next step), O = rotation offset of its B fragments.
    // Column J: 4 MFMAs on B fragment J; fragment J + 2 is requested in front of them (from the next stage once J + 2 >= NBF: behind the
    // step's barrier, which sits in front of column NBF / 2); the next step's A fragments behind the barrier as well (at a group's last
    // tap they come from the OTHER patch stage, which the patch waves have completed in front of that barrier).
#define P3_COL(J, O, P)                                                                                                  \
        if constexpr ((J) < NBF) {                                                                                       \
            if ((J) == NBF / 2) { __builtin_amdgcn_s_barrier(); asm volatile("" ::: "memory"); P3_SB }                    \
            if ((J) + 2 < NBF) { P3_RDB(((J) + 2 + (O)) % 3, (J) + 2, cur) } else { P3_RDB(((J) + 2 + (O)) % 3, (J) + 2 - NBF, nx) }   \
            P3_MF(P, 0, J, ((J) + (O)) % 3) if ((J) == NBF / 2) { P3_RDA(1 - (P), 0, npb, nsh) }                           \
            P3_MF(P, 1, J, ((J) + (O)) % 3) if ((J) == NBF / 2) { P3_RDA(1 - (P), 1, npb, nsh) }                           \
            P3_MF(P, 2, J, ((J) + (O)) % 3) if ((J) == NBF / 2) { P3_RDA(1 - (P), 2, npb, nsh) }                           \
            P3_MF(P, 3, J, ((J) + (O)) % 3) if ((J) == NBF / 2) { P3_RDA(1 - (P), 3, npb, nsh) }                           \
        }
#define P3_STEP(O, P)                                                                                                    \
        {                                                                                                                \
            const int nx = cur == NBUF - 1 ? 0 : cur + 1;                                                                \
            const int nt = t == 8 ? 0 : t + 1, nq = t == 8 ? q + 1 : q;                                                  \
            const int npst = t == 8 ? (pst == NPATCH - 1 ? 0 : pst + 1) : pst;                                           \
            const int nsh = tap_shift(nt), npb = npst * P3_PATCH_BYTES;                                                  \
            P3_SB                                                                                                        \
            P3_COL(0, O, P) P3_COL(1, O, P) P3_COL(2, O, P) P3_COL(3, O, P) P3_COL(4, O, P) P3_COL(5, O, P) P3_COL(6, O, P) P3_COL(7, O, P) \
            cur = nx; t = nt; q = nq; pst = npst;                                                                        \
            if (t == 0 && q % NG == 0) { if constexpr (!(ABLM & 8)) tile_epilogue(done); done++; zero_acc(); }          \
        }
    // ---- a finished tile, straight from the accumulators: block (I, J) of a lane = output channels 16 J + 4 lq .. + 3 of pixel
    // wave * 64 + 16 I + l15.  No LDS staging (the first version parked the accumulators in LDS 16 rows at a time and read them back
    // row-major like store_acc_tile: 3.9 us per tile against 2.2 us of MFMAs -- two thirds of the kernel), nothing that waits for memory
    // in a chain: the lane's bias values are loaded once per kernel, a tile's gate values all at once.
    const float4 *bias_lds = reinterpret_cast<const float4 *>(ldsb + BIAS_BASE) + lq;   // [4 j]: channels 16 j + 4 lq .. + 3
    auto tile_epilogue = [&](int ti) {
        const int tl = bid + ti * G;
        const unsigned img = div_magic((unsigned)tl, geo.magic_txy);
        const unsigned rem = (unsigned)tl - img * (unsigned)(geo.tiles_x * geo.tiles_y);
        const unsigned tyi = div_magic(rem, geo.magic_tx);
        const int y0 = (int)tyi * geo.TH, x0 = (int)(rem - tyi * (unsigned)geo.tiles_x) * geo.TW;
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const int p = wave * 64 + i * 16 + l15;
            const unsigned ty = div_magic((unsigned)p, geo.magic_tw);
            const int tx = p - (int)ty * geo.TW;
            const int oy = y0 + (int)ty, ox = x0 + tx;
            const bool pix_ok = (int)ty < geo.TH && oy < a.H && ox < a.W;
            const long long o0 = ((long long)((int)img * a.H + oy) * a.W + ox) * a.N + 4 * lq;
#pragma unroll
            for (int jh = 0; jh < NBF; jh += 4) {   // four channel blocks at a time (their gate values requested together)
            float gt[4][4];
            if (a.gate) {
#pragma unroll
                for (int jj = 0; jj < 4; jj++) {
                    const int j = jh + jj;
                    const bool ok = pix_ok && 16 * j + 4 * lq < a.N;
                    if constexpr (YBF) {
                        const uint2 g2 = ok ? *reinterpret_cast<const uint2 *>(reinterpret_cast<const __bf16 *>(a.gate) + o0 + 16 * j) : make_uint2(0u, 0u);
                        const bf16x4 gv = __builtin_bit_cast(bf16x4, g2);
                        gt[jj][0] = (float)gv[0]; gt[jj][1] = (float)gv[1]; gt[jj][2] = (float)gv[2]; gt[jj][3] = (float)gv[3];
                    } else {
                        const float4 g4 = ok ? *reinterpret_cast<const float4 *>(a.gate + o0 + 16 * j) : make_float4(0.f, 0.f, 0.f, 0.f);
                        gt[jj][0] = g4.x; gt[jj][1] = g4.y; gt[jj][2] = g4.z; gt[jj][3] = g4.w;
                    }
                }
            }
#pragma unroll
            for (int jj = 0; jj < 4; jj++) {
                const int j = jh + jj;
                if (!pix_ok || 16 * j + 4 * lq >= a.N) continue;
                const long long o = o0 + 16 * j;
                const float4 b4 = bias_lds[4 * j];
                float v[4] = {acc[i][j][0] + b4.x, acc[i][j][1] + b4.y, acc[i][j][2] + b4.z, acc[i][j][3] + b4.w};
                if (a.epi == EMADAPT_EPI_BIAS_RELU) {
#pragma unroll
                    for (int e = 0; e < 4; e++) v[e] = relu_keep_nan(v[e]);
                    if (a.keep < 1.0f) {
                        {   // (o is a multiple of 4: o .. o + 3 is a quad of the generator)
                            bool k[4];
                            if (a.mask) {
#pragma unroll
                                for (int q = 0; q < 4; q++) k[q] = a.mask[o + q] != 0;
                            } else dropout_keep4(a, (unsigned long long)o, k);
#pragma unroll
                            for (int q = 0; q < 4; q++) v[q] = k[q] ? drop_scale(a, v[q]) : 0.f;
                        }
                    }
                }
                if (a.gate) {
#pragma unroll
                    for (int e = 0; e < 4; e++) v[e] = gate_keep_nan(gt[jj][e], v[e] * a.scale);
                }
                if constexpr (YBF) *reinterpret_cast<uint2 *>(reinterpret_cast<__bf16 *>(a.y) + o) = cvt4(make_float4(v[0], v[1], v[2], v[3]));
                else *reinterpret_cast<float4 *>(a.y + o) = make_float4(v[0], v[1], v[2], v[3]);
            }
            }
        }
    };
    // rotation offsets of six consecutive steps: (k NBF) % 3; A buffer: k % 2
    constexpr int O1 = NBF % 3, O2 = (2 * NBF) % 3;
    for (int s = 0; s < total_steps; s += 6) {
        P3_STEP(0, 0)
        if (s + 1 < total_steps) P3_STEP(O1, 1)
        if (s + 2 < total_steps) P3_STEP(O2, 0)
        if (s + 3 < total_steps) P3_STEP(0, 1)
        if (s + 4 < total_steps) P3_STEP(O1, 0)
        if (s + 5 < total_steps) P3_STEP(O2, 1)
    }
#undef P3_STEP
#undef P3_COL
#undef P3_MF
#undef P3_RDB
#undef P3_RDA
#undef P3_SB
}

template <bool YBF, int ABLM = 0>
int launch_p3(IgemmArgs &a, int Cxp, hipStream_t stream) {
#ifdef EMADAPT_ABLATE_BUILD
    if constexpr (ABLM == 0 && YBF) {
        switch (tune(TUNE_ABLATE)) {
#define EMADAPT_ABL_CASE(m) case m: return launch_p3<YBF, m>(a, Cxp, stream);
            EMADAPT_ABL_CASE(1) EMADAPT_ABL_CASE(2) EMADAPT_ABL_CASE(3) EMADAPT_ABL_CASE(4) EMADAPT_ABL_CASE(8) EMADAPT_ABL_CASE(12)
            EMADAPT_ABL_CASE(15) EMADAPT_ABL_CASE(16) EMADAPT_ABL_CASE(32) EMADAPT_ABL_CASE(48) EMADAPT_ABL_CASE(52) EMADAPT_ABL_CASE(60) EMADAPT_ABL_CASE(63)
#undef EMADAPT_ABL_CASE
            default: break;
        }
    }
#endif
    P3Geo g;
    if (!p3_geometry(a.M / (a.H * a.W), a.H, a.W, a.dil, g)) return EMADAPT_ERR_UNSUPPORTED;
    const int cus = plan_cus();
    const int grid = g.tiles < cus ? g.tiles : cus;
    if (a.N <= 64) {
        static DynLdsAttr attr;
        const int rc = attr.ensure(reinterpret_cast<const void *>(&igemm_p3_kernel<4, YBF, ABLM>), p3_lds_bytes<4>());
        if (rc) return rc;
        hipLaunchKernelGGL((igemm_p3_kernel<4, YBF, ABLM>), dim3((unsigned)grid), dim3(P3_THREADS), p3_lds_bytes<4>(), stream, a, Cxp, g);
    } else {
        static DynLdsAttr attr;
        const int rc = attr.ensure(reinterpret_cast<const void *>(&igemm_p3_kernel<8, YBF, ABLM>), p3_lds_bytes<8>());
        if (rc) return rc;
        hipLaunchKernelGGL((igemm_p3_kernel<8, YBF, ABLM>), dim3((unsigned)grid), dim3(P3_THREADS), p3_lds_bytes<8>(), stream, a, Cxp, g);
    }
    return launch_status();
}

}  // namespace

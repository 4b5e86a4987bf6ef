// E-step (weak-label assignment) for gfx950.
//
// Replaces the reference's estep.py:33-84 (estep_py) + tf.argmax (deeplab.py:123) and the native
// entry estep/weaklabel.cpp:7-124.  One 512-thread workgroup per image (eight waves, two per SIMD: the kernel is a chain
// of dependent LDS round trips and short instruction runs, which a second wave per SIMD fills):
//
//   * every thread owns the pixels  i = tid + 512*e  (e < EPT): their running per-pixel maximum
//     stays in registers, the channels of the classes present in the image stay in LDS
//     ([slot][pixel], conflict-free for the pixel-strided access), so one (iteration, class) step
//     never touches HBM;
//   * when the whole [C][pixel] image fits in LDS (41x41x21 does: 141 KB) the logits are staged in ONCE with
//     coalesced 16-byte loads and every later pass -- suppress, maxima, sweeps, argmax -- runs out of LDS;
//   * the k-th smallest of (max - channel) -- np.partition in the reference -- is found exactly by a 3-pass
//     radix select on the raw IEEE bits (order-preserving for the non-negative differences): 11+10+10 bits,
//     each pass = LDS histogram (ds_add_u32, two 16-bit counts per word) + a two-level prefix scan every wave runs
//     for itself (packed adds, DPP wave scan, ballot); no sorting, 3 barriers per selection (radix_select);
//   * every loop over a thread's pixels is written classes-outside / pixels-inside with unconditional (clamped)
//     loads, so that EPT independent loads are in flight instead of one dependent chain;
//   * arithmetic is fp32 add / sub / compare only, in the reference's order, and the two np.mean()
//     calls follow NumPy's pairwise summation tree (its leaf / combine program is built on the host: it depends on
//     H*W alone), so label maps AND float maps are bit-identical;
//   * the class visiting order (random.shuffle in the reference) is an input; it is staged in LDS and compacted to
//     the classes present in the image before the sweeps; for the LDS-resident geometries (round 5) it is flattened into ONE
//     visit list, read two visits ahead, and the visited class image stays in registers from the key to the update while
//     the next one is fetched during the current selection (86.5 -> 78.8 us: profiles/r05_estep_bench.txt).
//
// Round 3 (profiles/r03_estep_bench.txt): 192 -> 90 us at 16 x 41x41x21, 685 -> 232 us at 10 x 65x65x41 (there the classes
// that do not fit in LDS live class-major in the workspace, not in the NHWC float map).  What the
// first version lost: 12 barriers + 18 ds_bpermute steps per selection, FLAT loads / stores where one pointer could be
// LDS or HBM, ~50 serial LDS-stack steps of thread 0 building the pairwise-sum program, one load in flight per thread.
//
// Traffic: logits are read from HBM once (plus once by the batch-maximum pre-kernel), the
// int64 label map is written once; the float map is written only when the caller asks for it.
#include <float.h>
#include <math.h>
#include <stdlib.h>

#include <mutex>
#include <vector>

#include "common.h"

namespace {

using namespace emadapt;

// Phase-timing exits (emadapt_tune_set("estep_stop", n)) exist only in diagnostic builds (-DEMADAPT_ABLATE_BUILD).
#ifdef EMADAPT_ABLATE_BUILD
#define ESTEP_STOP(n) if (a.stop_after == (n)) return
#else
#define ESTEP_STOP(n) do { } while (0)
#endif

constexpr int kThreads = 512;   // 8 waves: two per SIMD (one wave per SIMD leaves every LDS / issue latency of this
                                // dependency-bound kernel in the open)
constexpr int kWaves = kThreads / 64;
constexpr int kMaxC = 256;       // classes
constexpr int kMaxLeaves = 192;  // leaves of NumPy's pairwise-sum tree for HW <= 8192
constexpr int kMaxLevels = 12;   // depth of that tree
constexpr int kMaxEPT = 16;      // pixels per thread -> HW <= 8192
constexpr size_t kLdsBudget = 160 * 1024;
constexpr int kHistBins = 2048;  // 11-bit first radix pass
constexpr int kPermCap = 1024;   // visiting-order entries (num_iter * (C-1)) staged in LDS; beyond that: read from HBM

// Batch-global maximum of the logits (np.amax(feature_map), estep.py:50), first stage: one maximum per block into
// out[blockIdx.x] (kGmaxBlocks floats of workspace; the E-step kernel folds them -- no memset, no atomics, one launch less
// in front of the kernel).  A 16-lane group reads whole rows with 16-byte loads when the row pitch allows it.
constexpr int kGmaxBlocks = 64;
__global__ void __launch_bounds__(256) estep_gmax_kernel(const float *__restrict__ logits, long long rows,
                                                       int C, long long ldl, float *__restrict__ out) {
    float m = -INFINITY;
    if ((ldl & 3) == 0 && (reinterpret_cast<uintptr_t>(logits) & 15) == 0) {
        const long long q4 = ldl >> 2, n4 = rows * q4;
        for (long long g = (long long)blockIdx.x * blockDim.x + threadIdx.x; g < n4; g += (long long)gridDim.x * blockDim.x) {
            const float4 v = *reinterpret_cast<const float4 *>(logits + g * 4);
            const int c = (int)(g % q4) * 4;
            if (c + 0 < C) m = fmaxf(m, v.x);
            if (c + 1 < C) m = fmaxf(m, v.y);
            if (c + 2 < C) m = fmaxf(m, v.z);
            if (c + 3 < C) m = fmaxf(m, v.w);
        }
    } else {
        for (long long r = (long long)blockIdx.x * blockDim.x + threadIdx.x; r < rows; r += (long long)gridDim.x * blockDim.x) {
            const float *p = logits + r * ldl;
            for (int c = 0; c < C; c++) m = fmaxf(m, p[c]);
        }
    }
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
    __shared__ float s[4];
    if ((threadIdx.x & 63) == 0) s[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) out[blockIdx.x] = fmaxf(fmaxf(s[0], s[1]), fmaxf(s[2], s[3]));
}

struct EstepArgs {
    const float *logits;
    long long ldl;
    const void *label;
    int label_kind;
    long long label_len;
    const int *perms;
    int HW, C, suppress, num_iter;
    float margin;
    int k_bg, k_fg;
    const float *gmax_part;   // [kGmaxBlocks] per-block maxima of estep_gmax_kernel
    float *map_out;
    float *ovf;       // [B][C][hwp] class-major home of the channels that are not LDS-resident (workspace; used when p_res < C)
    float *mxbuf;     // [B][hwp] per-pixel maxima after the suppress step, written by estep_prep_kernel (p_res < C only)
    float *part_v;    // [B][hwp] / [B][hwp] / [B]: labels-only runs of an overflow geometry -- the E-step kernel leaves the argmax over
    int *part_i;      //   its LDS-resident classes (value + shift, class) and the image's shift; estep_finish_kernel completes it
    float *shiftbuf;  //   over the class images in the workspace, on the whole chip
    long long *labels_out;
    int p_res;  // present-channel slots resident in LDS
    int hwp;    // HW rounded up to a multiple of 4 (LDS row pitch)
    int full;   // 1: all C channels are staged into LDS, indexed by class (p_res == C)
    int stop_after;  // diagnostic builds (tune "estep_stop"): leave the kernel after phase n (0 = run everything)
    // NumPy's pairwise-summation tree for n = HW (it depends on HW alone): leaves and the post-order combine program, built
    // on the host (pairwise_program).  On the device it was ~50 dependent LDS-stack steps of thread 0: 20 us of the kernel.
    // (round 4: the combine tree by LEVELS -- node i of level L adds the values of two earlier nodes -- so that one wave walks it in
    // <= 8 dependent steps; thread 0 replaying the post-order program with an LDS stack was 6 us per np.mean at 65 x 65)
    int nleaf, nnode, nlev;
    unsigned short leaf_off[kMaxLeaves];
    unsigned char leaf_n[kMaxLeaves];        // <= 128
    unsigned short node_l[kMaxLeaves], node_r[kMaxLeaves];   // value slots of a node's operands: leaf k = slot k, node i = slot nleaf + i
    unsigned short lev_start[kMaxLevels + 1];               // nodes [lev_start[L], lev_start[L + 1]) depend only on earlier levels
};

// np.add.reduce of a contiguous float32 array (numpy/core/src/umath/loops_utils.h.src, pairwise_sum): blocks of <= 128 are
// summed with eight running sums, larger ones are split at n/2 rounded down to a multiple of 8.
void pairwise_program(EstepArgs &a, int HW) {
    int nleaf = 0, sp = 1;
    int s_off[32], s_n[32], s_st[32];
    // post-order walk of the recursion; a value stack of slot ids replays it: leaves push, a split node pops right then left
    int vstack[32], vsp = 0;
    int nl[kMaxLeaves], nr[kMaxLeaves], lev[2 * kMaxLeaves], nnode = 0;   // nodes in creation (post-) order; lev[] per value id
    s_off[0] = 0; s_n[0] = HW; s_st[0] = 0;
    while (sp > 0) {
        const int off = s_off[sp - 1], n = s_n[sp - 1], st = s_st[sp - 1];
        if (n <= 128) {
            a.leaf_off[nleaf] = (unsigned short)off; a.leaf_n[nleaf] = (unsigned char)n;
            lev[nleaf] = 0;
            vstack[vsp++] = nleaf;          // value id of leaf k = k
            nleaf++;
            sp--;
        } else {
            int n2 = n / 2;
            n2 -= n2 % 8;
            if (st == 0) {
                s_st[sp - 1] = 1;
                s_off[sp] = off; s_n[sp] = n2; s_st[sp] = 0; sp++;
            } else if (st == 1) {
                s_st[sp - 1] = 2;
                s_off[sp] = off + n2; s_n[sp] = n - n2; s_st[sp] = 0; sp++;
            } else {
                const int r = vstack[--vsp], l = vstack[--vsp];
                nl[nnode] = l; nr[nnode] = r;            // value ids: leaves < kMaxLeaves, node i = kMaxLeaves + i (remapped below)
                const int ll = lev[l < kMaxLeaves ? l : kMaxLeaves + (l - kMaxLeaves)], lr = lev[r < kMaxLeaves ? r : kMaxLeaves + (r - kMaxLeaves)];
                lev[kMaxLeaves + nnode] = (ll > lr ? ll : lr) + 1;
                vstack[vsp++] = kMaxLeaves + nnode;
                nnode++;
                sp--;
            }
        }
    }
    // nodes sorted by level (stable); operands renamed to value slots: leaf k -> k, node (sorted position p) -> nleaf + p
    int order[kMaxLeaves], pos[kMaxLeaves], nlev = 0;
    for (int i = 0; i < nnode; i++) nlev = lev[kMaxLeaves + i] > nlev ? lev[kMaxLeaves + i] : nlev;
    int p = 0;
    a.lev_start[0] = 0;
    for (int L = 1; L <= nlev; L++) {
        for (int i = 0; i < nnode; i++)
            if (lev[kMaxLeaves + i] == L) { order[p] = i; pos[i] = p; p++; }
        a.lev_start[L] = (unsigned short)p;
    }
    for (int q = 0; q < nnode; q++) {
        const int i = order[q];
        a.node_l[q] = (unsigned short)(nl[i] < kMaxLeaves ? nl[i] : nleaf + pos[nl[i] - kMaxLeaves]);
        a.node_r[q] = (unsigned short)(nr[i] < kMaxLeaves ? nr[i] : nleaf + pos[nr[i] - kMaxLeaves]);
    }
    a.nleaf = nleaf;
    a.nnode = nnode;
    a.nlev = nlev;
}

// One leaf (n <= 128) of NumPy's float32 pairwise summation.
__device__ float np_leaf_sum(const float *a, int n) {
    if (n < 8) {
        float r = 0.f;
        for (int i = 0; i < n; i++) r = r + a[i];
        return r;
    }
    float r0 = a[0], r1 = a[1], r2 = a[2], r3 = a[3], r4 = a[4], r5 = a[5], r6 = a[6], r7 = a[7];
    int i = 8;
    const int lim = n - (n % 8);
    for (; i < lim; i += 8) {
        r0 = r0 + a[i + 0];
        r1 = r1 + a[i + 1];
        r2 = r2 + a[i + 2];
        r3 = r3 + a[i + 3];
        r4 = r4 + a[i + 4];
        r5 = r5 + a[i + 5];
        r6 = r6 + a[i + 6];
        r7 = r7 + a[i + 7];
    }
    float res = ((r0 + r1) + (r2 + r3)) + ((r4 + r5) + (r6 + r7));
    for (; i < n; i++) res = res + a[i];
    return res;
}

struct Lds {
    float *mx;        // [hwp]
    float *chan;      // [p_res][hwp]
    int *slot;        // [kMaxC]  class -> slot (or -1)
    int *red;         // [2][kWaves]
    float *leafsum;   // [2 * kMaxLeaves]: leaf sums, then the tree's inner nodes
    int *leaf_off;    // [kMaxLeaves]
    int *leaf_n;      // [kMaxLeaves]
    int *stk;         // [3][16]
    unsigned short *node_l, *node_r;   // [kMaxLeaves] operand slots of the tree's inner nodes, level by level
    unsigned short *lev_start;         // [kMaxLevels + 1]
    int *misc;        // [0]=P [1]=nleaf [2]=nnode [3]=nlev
    float *scal;      // [0]=shift
    unsigned *hist;   // [kHistBins] radix-select histogram
    int *sel;         // [4] selected bin, remaining rank
    unsigned short *perm;  // [kPermCap] class visiting order, staged once (a global load per visit would serialise)
};

__device__ __forceinline__ Lds carve(char *base, int hwp, int p_res) {
    Lds l;
    l.mx = reinterpret_cast<float *>(base);
    l.chan = l.mx + hwp;
    l.slot = reinterpret_cast<int *>(l.chan + (size_t)p_res * hwp);
    l.red = l.slot + kMaxC;
    l.leafsum = reinterpret_cast<float *>(l.red + 2 * kWaves);
    l.leaf_off = reinterpret_cast<int *>(l.leafsum + 2 * kMaxLeaves);
    l.leaf_n = l.leaf_off + kMaxLeaves;
    l.stk = l.leaf_n + kMaxLeaves;
    l.misc = l.stk + 48;
    l.scal = reinterpret_cast<float *>(l.misc + 4);
    l.hist = reinterpret_cast<unsigned *>(l.scal + 4);
    l.sel = reinterpret_cast<int *>(l.hist + kHistBins);
    l.perm = reinterpret_cast<unsigned short *>(l.sel + 4);
    l.node_l = l.perm + kPermCap;
    l.node_r = l.node_l + kMaxLeaves;
    l.lev_start = l.node_r + kMaxLeaves;
    return l;
}

size_t lds_bytes(int hwp, int p_res) {
    return sizeof(float) * ((size_t)hwp + (size_t)p_res * hwp) + sizeof(int) * (kMaxC + 2 * kWaves) +
           sizeof(float) * 2 * kMaxLeaves + sizeof(int) * (2 * kMaxLeaves + 48 + 4) + sizeof(float) * 4 +
           sizeof(unsigned) * kHistBins + sizeof(int) * 4 + sizeof(unsigned short) * (kPermCap + 2 * kMaxLeaves + kMaxLevels + 4);
}

// np.mean over l.mx[0..HW): leaves in parallel, then the combine tree level by level on wave 0 (each inner node = left + right,
// the adds of NumPy's recursion in its order; a level only reads values of earlier levels).
// Every thread must call; the result is valid on thread 0 only.
__device__ float np_mean_lds(const Lds &l, int HW, int tid) {
    const int nleaf = l.misc[1];
    if (tid < nleaf) l.leafsum[tid] = np_leaf_sum(l.mx + l.leaf_off[tid], l.leaf_n[tid]);
    __syncthreads();
    float res = 0.f;
    if (tid < 64) {
        const int nnode = l.misc[2], nlev = l.misc[3];
        for (int L = 0; L < nlev; L++) {
            const int i1 = l.lev_start[L + 1];
            for (int i = l.lev_start[L] + tid; i < i1; i += 64) l.leafsum[nleaf + i] = l.leafsum[l.node_l[i]] + l.leafsum[l.node_r[i]];
            // (one wave: its LDS stores are performed before its later loads; keep the compiler from moving them)
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        }
        if (tid == 0) res = __fdiv_rn(l.leafsum[nleaf + nnode - 1 + (nnode == 0 ? 1 - nleaf : 0)], (float)HW);   // nnode == 0: the single leaf
    }
    return res;
}

// k-th smallest (0-indexed) of the workgroup's keys, exact.  Valid keys are < 2^31 (bit patterns of non-negative
// floats); unused slots hold 0xFFFFFFFF and never match a prefix.  Three passes over 11 + 10 + 10 bits from the top:
// histogram the digit of every key that still matches the selected prefix, scan the histogram to find the bin
// containing rank k, descend into it.  Every thread returns the same value.
//
// ONE barrier per pass (round 3; the first version needed four -- clear, count, scan, broadcast -- and the 5 x P dependent
// selections of an image, 12 barriers each, were two thirds of the kernel's time):
//   * each pass has a histogram of its own, so nothing is cleared between counting and scanning: a pass's buffer is zeroed
//     one barrier after its scan, by which time every wave is past it (h0 after the second barrier, h1 after the third, h2
//     after the first barrier of the NEXT selection; all three are zero when the sweeps start);
//   * after the barrier that ends the counting EVERY wave scans the whole histogram by itself (32 / 16 bins per lane, wave
//     prefix sum, ballot for the owning lane): no partial sums cross waves, no result is broadcast through LDS;
//   * counts are at most HW <= 8192, so two bins share a 32-bit word (bin b = half b & 1 of word b >> 1, incremented by
//     1 << 16 * (b & 1)): the three histograms fit the 8 KB the single 2048-bin one had -- LDS is full at 41 x 41 x 21.
constexpr int kH1 = 1024, kH2 = 1536;   // word offsets of the 1024-bin histograms behind the 2048-bin one

// Inclusive prefix sum over the 64 lanes with data-parallel-primitive moves (six dependent v_add ... dpp, a few cycles each;
// __shfl_up goes through ds_bpermute, ~100 cycles per step): Kogge-Stone inside the rows of 16 lanes (row_shr 1, 2, 4, 8;
// lanes without a source add 0), then lane 15 of rows 0 / 2 into rows 1 / 3 (row_bcast:15) and lane 31 into rows 2 and 3
// (row_bcast:31).
__device__ __forceinline__ unsigned wave_inclusive_sum(unsigned v) {
    v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, false);
    v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, false);
    v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xf, false);
    v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xf, false);
    v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xa, 0xf, false);
    v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xc, 0xf, false);
    return v;
}

__device__ __forceinline__ void hist_add(unsigned *h, unsigned bin) { atomicAdd(&h[bin >> 1], 1u << (16u * (bin & 1u))); }

// Locate rank kk in a histogram of 64 * WPL * 2 bins (WPL words per lane); returns the bin, updates kk to the rank inside it.
// Two levels, both a wave prefix sum + ballot: first over the lanes' segment totals (the segment's words are added PACKED, two
// 16-bit counts per add -- no sum exceeds HW <= 8192), then over the 2 * WPL bins of the owning segment, one bin per lane
// (a 16-bit LDS read).  ~45 instructions: eight waves run this redundantly, so its length is what a selection waits for.
template <int WPL>
__device__ __forceinline__ unsigned hist_locate(const unsigned *h, unsigned &kk, int lane) {
    static_assert(WPL % 4 == 0 && 2 * WPL <= 64, "words per lane");
    unsigned acc = 0;
#pragma unroll
    for (int j = 0; j < WPL; j += 4) {
        const uint4 v = *reinterpret_cast<const uint4 *>(h + lane * WPL + j);
        acc += v.x + v.y + v.z + v.w;
    }
    const unsigned local = (acc & 0xFFFFu) + (acc >> 16);
    const unsigned incl = wave_inclusive_sum(local);
    const bool mine = kk >= incl - local && kk < incl;   // exactly one lane (the histogram holds more than kk keys)
    const int owner = __ffsll((long long)__ballot(mine)) - 1;   // wave-uniform (a scalar register)
    const unsigned r = (unsigned)__builtin_amdgcn_readlane((int)(kk - (incl - local)), owner);   // rank inside the segment
    const unsigned short *seg = reinterpret_cast<const unsigned short *>(h + owner * WPL);   // bin b = half-word b
    const unsigned c = lane < 2 * WPL ? (unsigned)seg[lane] : 0u;
    const unsigned incl2 = wave_inclusive_sum(c);
    const bool mine2 = r >= incl2 - c && r < incl2;
    const int l2 = __ffsll((long long)__ballot(mine2)) - 1;
    kk = (unsigned)__builtin_amdgcn_readlane((int)(r - (incl2 - c)), l2);
    return (unsigned)(owner * WPL * 2 + l2);
}

// (abl: diagnostic builds only -- timing with parts of the selection left out; results are then garbage)
#ifdef EMADAPT_ABLATE_BUILD
#define SEL_ABL(bit) (abl & (bit))
#else
#define SEL_ABL(bit) 0
#endif
template <int EPT>
__device__ unsigned radix_select(const unsigned (&key)[EPT], int k, const Lds &l, int tid, int lane, int abl = 0) {
    unsigned *h0 = l.hist, *h1 = l.hist + kH1, *h2 = l.hist + kH2;
    const uint4 zero4 = make_uint4(0u, 0u, 0u, 0u);
    unsigned kk = (unsigned)k;
    // ---- bits 30..20
#pragma unroll
    for (int e = 0; e < EPT; e++)
        if (key[e] < 0x80000000u && !SEL_ABL(16)) hist_add(h0, key[e] >> 20);
    if (!SEL_ABL(64)) __syncthreads();
    if (tid < 128) *reinterpret_cast<uint4 *>(h2 + tid * 4) = zero4;   // the previous selection's last histogram
    unsigned prefix = SEL_ABL(32) ? 0u : hist_locate<16>(h0, kk, lane);
    if (SEL_ABL(128)) return prefix << 20;
    // ---- bits 19..10
#pragma unroll
    for (int e = 0; e < EPT; e++)
        if ((key[e] >> 20) == prefix && !SEL_ABL(16)) hist_add(h1, (key[e] >> 10) & 1023u);
    if (!SEL_ABL(64)) __syncthreads();
    if (tid < 256) *reinterpret_cast<uint4 *>(h0 + tid * 4) = zero4;
    prefix = (prefix << 10) | (SEL_ABL(32) ? 0u : hist_locate<8>(h1, kk, lane));
    // ---- bits 9..0
#pragma unroll
    for (int e = 0; e < EPT; e++)
        if ((key[e] >> 10) == prefix && !SEL_ABL(16)) hist_add(h2, key[e] & 1023u);
    if (!SEL_ABL(64)) __syncthreads();
    if (tid < 128) *reinterpret_cast<uint4 *>(h1 + tid * 4) = zero4;
    return (prefix << 10) | (SEL_ABL(32) ? 0u : hist_locate<8>(h2, kk, lane));
}

// Overflow geometries (65 x 65 x 41: only 7 of the 41 class images fit in LDS beside the bookkeeping), step (ii) of estep.py:46-55
// for the whole batch ON THE WHOLE CHIP (round 4): one thread per pixel reads its logits row, forms the row minimum over
// (present ? x : x + global max), replaces the absent classes above it, and writes EVERY class image class-major into the
// workspace plus the pixel's maximum.  Rounds 2-3 did this inside the E-step kernel, i.e. on B of the 256 CUs: a workgroup moved
// its 2.2 MB (two passes over the rows + the class-major images) at the ~30 GB/s one CU gets out of 4-byte-per-lane accesses --
// 65-87 of the kernel's 210-236 us at B = 10.  Spread over 17 workgroups per image it is a few microseconds, and the E-step
// kernel's phase 2 shrinks to copying the resident class images into LDS.
__global__ void __launch_bounds__(256) estep_prep_kernel(EstepArgs a) {
    __shared__ int present[kMaxC];
    const int b = blockIdx.y, tid = threadIdx.x;
    const int HW = a.HW, C = a.C, hwp = a.hwp;
    for (int c = tid; c < C; c += 256) present[c] = 0;
    __syncthreads();
    if (a.label_kind == EMADAPT_LABEL_MULTIHOT) {      // image-level label set (estep.py:35-44), as in estep_kernel
        const unsigned char *mh = static_cast<const unsigned char *>(a.label) + (size_t)b * C;
        for (int c = tid; c < C; c += 256)
            if (mh[c]) present[c] = 1;
    } else if (a.label_kind == EMADAPT_LABEL_F32) {
        const float *lab = static_cast<const float *>(a.label) + (size_t)b * a.label_len;
        for (long long i = tid; i < a.label_len; i += 256) {
            const int v = (int)(unsigned char)(int)lab[i];
            if (v < C) present[v] = 1;
        }
    } else {
        const int *lab = static_cast<const int *>(a.label) + (size_t)b * a.label_len;
        for (long long i = tid; i < a.label_len; i += 256) {
            const int v = (int)(unsigned char)lab[i];
            if (v < C) present[v] = 1;
        }
    }
    float gmax = 0.f;
    const bool suppress = a.suppress != 0;
    if (suppress) {
        float g = a.gmax_part[tid & 63];
        for (int o = 32; o > 0; o >>= 1) g = fmaxf(g, __shfl_xor(g, o));
        gmax = g;
    }
    __syncthreads();
    const int i = blockIdx.x * 256 + tid;
    const int ic = i < HW ? i : HW - 1;
    const float *row = a.logits + ((size_t)b * HW + ic) * a.ldl;
    float *out = a.ovf + (size_t)b * C * hwp + ic;
    float m = FLT_MAX, mxv = 0.f;
    if ((a.ldl & 3) == 0 && (reinterpret_cast<uintptr_t>(a.logits) & 15) == 0) {
        const int nq = (C + 3) >> 2;
        if (suppress)
            for (int q = 0; q < nq; q++) {
                const float4 v = *reinterpret_cast<const float4 *>(row + 4 * q);
                const float x[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                for (int k = 0; k < 4; k++)
                    if (4 * q + k < C) {
                        const float t = present[4 * q + k] ? x[k] : x[k] + gmax;
                        m = (t < m) ? t : m;
                    }
            }
        for (int q = 0; q < nq; q++) {
            const float4 v = *reinterpret_cast<const float4 *>(row + 4 * q);   // (second touch of a row the thread has just read)
            float x[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const int c = 4 * q + k;
                if (c < C) {
                    if (suppress && !present[c] && x[k] > m) x[k] = m - a.margin;
                    mxv = (c == 0 || x[k] > mxv) ? x[k] : mxv;
                    if (i < HW) out[(size_t)c * hwp] = x[k];    // consecutive lanes -> consecutive pixels: coalesced
                }
            }
        }
    } else {
        if (suppress)
            for (int c = 0; c < C; c++) {
                const float t = present[c] ? row[c] : row[c] + gmax;
                m = (t < m) ? t : m;
            }
        for (int c = 0; c < C; c++) {
            float x = row[c];
            if (suppress && !present[c] && x > m) x = m - a.margin;
            mxv = (c == 0 || x > mxv) ? x : mxv;
            if (i < HW) out[(size_t)c * hwp] = x;
        }
    }
    if (i < HW) a.mxbuf[(size_t)b * hwp + i] = mxv;
}

// Second half of the argmax of an overflow geometry (labels-only runs; see the end of estep_kernel): one thread per pixel walks the
// class images that were NOT LDS-resident in estep_kernel -- absent classes and the present classes beyond the p_res-th -- in class
// order, on value + shift (the fp32 sum the reference compares, estep.py:81-82 + deeplab.py:123), and merges with the partial
// result: the larger value, on a tie the lower class.  Which classes were resident is recomputed from the label set (slots are
// handed out in ascending class order, estep_kernel).
__global__ void __launch_bounds__(256) estep_finish_kernel(EstepArgs a) {
    __shared__ int present[kMaxC];
    __shared__ int nonres[kMaxC];
    __shared__ int n_nonres;
    const int b = blockIdx.y, tid = threadIdx.x;
    const int HW = a.HW, C = a.C, hwp = a.hwp;
    for (int c = tid; c < C; c += 256) present[c] = 0;
    __syncthreads();
    if (a.label_kind == EMADAPT_LABEL_MULTIHOT) {
        const unsigned char *mh = static_cast<const unsigned char *>(a.label) + (size_t)b * C;
        for (int c = tid; c < C; c += 256)
            if (mh[c]) present[c] = 1;
    } else if (a.label_kind == EMADAPT_LABEL_F32) {
        const float *lab = static_cast<const float *>(a.label) + (size_t)b * a.label_len;
        for (long long i = tid; i < a.label_len; i += 256) {
            const int v = (int)(unsigned char)(int)lab[i];
            if (v < C) present[v] = 1;
        }
    } else {
        const int *lab = static_cast<const int *>(a.label) + (size_t)b * a.label_len;
        for (long long i = tid; i < a.label_len; i += 256) {
            const int v = (int)(unsigned char)lab[i];
            if (v < C) present[v] = 1;
        }
    }
    __syncthreads();
    if (tid == 0) {
        int P = 0, n = 0;
        for (int c = 0; c < C; c++) {
            const bool res = present[c] && P < a.p_res;
            P += present[c] ? 1 : 0;
            if (!res) nonres[n++] = c;
        }
        n_nonres = n;
    }
    __syncthreads();
    const int i = blockIdx.x * 256 + tid;
    if (i >= HW) return;
    const float shift = a.shiftbuf[b];
    const float *img = a.ovf + (size_t)b * C * hwp + i;
    int best = -1;
    float bv = 0.f;
    const int n = n_nonres;
    for (int k = 0; k < n; k++) {     // consecutive lanes -> consecutive pixels of one class image: coalesced
        const int c = nonres[k];
        const float v = img[(size_t)c * hwp] + shift;
        if (best < 0 || v > bv) { bv = v; best = c; }
    }
    const float pv = a.part_v[(size_t)b * hwp + i];
    const int pi = a.part_i[(size_t)b * hwp + i];
    if (pi >= 0 && (best < 0 || pv > bv || (pv == bv && pi < best))) best = pi;
    a.labels_out[(size_t)b * HW + i] = best;
}

template <int EPT>
__global__ void __launch_bounds__(kThreads, 2) estep_kernel(EstepArgs a) {   // (LDS admits one workgroup per CU: 2 waves per SIMD, 256 registers)
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const Lds l = carve(smem, a.hwp, a.p_res);
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int b = blockIdx.x;
    const int HW = a.HW, C = a.C, hwp = a.hwp, p_res = a.p_res;
    const bool full = a.full != 0;  // all channels LDS-resident, slot of class c == c

    // ---- class visiting order -> LDS (one coalesced load instead of one dependent HBM load per visit) ----
    const int nperm = a.num_iter * (C - 1);
    const bool perm_lds = nperm <= kPermCap;
    if (perm_lds)
        for (int t = tid; t < nperm; t += kThreads) l.perm[t] = (unsigned short)a.perms[t];

    // ---- image-level label set (estep.py:35-44) ----
    for (int c = tid; c < kMaxC; c += kThreads) l.slot[c] = 0;
    for (int h = tid; h < kHistBins; h += kThreads) l.hist[h] = 0u;   // radix_select expects its histograms zeroed
    __syncthreads();
    if (a.label_kind == EMADAPT_LABEL_MULTIHOT) {
        const unsigned char *mh = static_cast<const unsigned char *>(a.label) + (size_t)b * C;
        for (int c = tid; c < C; c += kThreads)
            if (mh[c]) l.slot[c] = 1;
    } else if (a.label_kind == EMADAPT_LABEL_F32) {
        const float *lab = static_cast<const float *>(a.label) + (size_t)b * a.label_len;
        for (long long i = tid; i < a.label_len; i += kThreads) {
            int v = (int)(unsigned char)(int)lab[i];  // label.astype(np.uint8)
            if (v < C) l.slot[v] = 1;
        }
    } else {
        const int *lab = static_cast<const int *>(a.label) + (size_t)b * a.label_len;
        for (long long i = tid; i < a.label_len; i += kThreads) {
            int v = (int)(unsigned char)lab[i];
            if (v < C) l.slot[v] = 1;
        }
    }
    __syncthreads();
    if (full) {   // slot of class c == c: nothing to count
        for (int c = tid; c < C; c += kThreads) l.slot[c] = l.slot[c] != 0 ? c : -1;
    } else if (tid == 0) {
        int P = 0;
        for (int c = 0; c < C; c++) {
            const bool present = l.slot[c] != 0;
            l.slot[c] = present ? P : -1;
            P += present ? 1 : 0;
        }
        l.misc[0] = P;
    }
    if (tid == 0) {
        l.misc[1] = a.nleaf;
        l.misc[2] = a.nnode;
        l.misc[3] = a.nlev;
    }
    for (int t = tid; t < a.nleaf; t += kThreads) { l.leaf_off[t] = a.leaf_off[t]; l.leaf_n[t] = a.leaf_n[t]; }
    for (int t = tid; t < a.nnode; t += kThreads) { l.node_l[t] = a.node_l[t]; l.node_r[t] = a.node_r[t]; }
    if (tid <= a.nlev) l.lev_start[tid] = a.lev_start[tid];
    __syncthreads();

    ESTEP_STOP(1);
    float gmax = 0.f;
    if (a.suppress) {   // fold the per-block maxima (max is exact: any order gives the same float)
        float g = a.gmax_part[lane];
        for (int o = 32; o > 0; o >>= 1) g = fmaxf(g, __shfl_xor(g, o));
        gmax = g;
    }
    const bool suppress = a.suppress != 0;
    float mx[EPT];

    if (full) {
        // ---- stage the image's logits into LDS as [class][pixel]: coalesced HBM reads, read exactly once ----
        const float *img = a.logits + (size_t)b * HW * a.ldl;
        if ((a.ldl & 3) == 0 && ((reinterpret_cast<uintptr_t>(a.logits) & 15) == 0)) {
            const int q4 = (int)(a.ldl >> 2);
            const int n4 = HW * q4;
            constexpr int U = 8;   // loads in flight per thread (one at a time, every load's full latency was in the open)
            for (int g0 = tid; g0 < n4; g0 += kThreads * U) {
                float4 v[U];
#pragma unroll
                for (int u = 0; u < U; u++) {
                    const int g = g0 + u * kThreads;
                    v[u] = *reinterpret_cast<const float4 *>(img + (size_t)(g < n4 ? g : g0) * 4);
                }
#pragma unroll
                for (int u = 0; u < U; u++) {
                    const int g = g0 + u * kThreads;
                    if (g >= n4) break;
                    const int row = g / q4, q = g - row * q4;
                    const int c = q * 4;
                    if (c + 0 < C) l.chan[(size_t)(c + 0) * hwp + row] = v[u].x;
                    if (c + 1 < C) l.chan[(size_t)(c + 1) * hwp + row] = v[u].y;
                    if (c + 2 < C) l.chan[(size_t)(c + 2) * hwp + row] = v[u].z;
                    if (c + 3 < C) l.chan[(size_t)(c + 3) * hwp + row] = v[u].w;
                }
            }
        } else {
            for (int g = tid; g < HW * C; g += kThreads) {
                const int row = g / C, c = g - row * C;
                l.chan[(size_t)c * hwp + row] = img[(size_t)row * a.ldl + c];
            }
        }
        __syncthreads();
        // ---- suppress absent classes (estep.py:46-55) in place, per-pixel max ----
        // (class loop outside, the thread's pixels inside: EPT independent compare chains instead of one dependent LDS
        //  read after another; whether a class is present is wave-uniform)
        float m[EPT], mxv[EPT];
#pragma unroll
        for (int e = 0; e < EPT; e++) m[e] = FLT_MAX;
        // (pixels past the end re-read the last one: every load is unconditional and the EPT of a class are in flight together)
        int ii[EPT];
#pragma unroll
        for (int e = 0; e < EPT; e++) ii[e] = min(tid + e * kThreads, HW - 1);
        if (suppress) {
            for (int c = 0; c < C; c++) {
                const bool present = __builtin_amdgcn_readfirstlane(l.slot[c]) >= 0;
                const float *ch = l.chan + (size_t)c * hwp;
                float v[EPT];
#pragma unroll
                for (int e = 0; e < EPT; e++) v[e] = ch[ii[e]];
#pragma unroll
                for (int e = 0; e < EPT; e++) {
                    const float t = present ? v[e] : v[e] + gmax;
                    m[e] = (t < m[e]) ? t : m[e];
                }
            }
        }
        for (int c = 0; c < C; c++) {
            const bool absent = suppress && __builtin_amdgcn_readfirstlane(l.slot[c]) < 0;
            float *ch = l.chan + (size_t)c * hwp;
            float v[EPT];
#pragma unroll
            for (int e = 0; e < EPT; e++) v[e] = ch[ii[e]];
#pragma unroll
            for (int e = 0; e < EPT; e++) {
                if (absent && v[e] > m[e]) {
                    v[e] = m[e] - a.margin;
                    if (tid + e * kThreads < HW) ch[ii[e]] = v[e];
                }
                mxv[e] = (c == 0 || v[e] > mxv[e]) ? v[e] : mxv[e];
            }
        }
#pragma unroll
        for (int e = 0; e < EPT; e++) {
            const int i = tid + e * kThreads;
            if (i < HW) l.mx[i] = mxv[e];
        }
    } else {
        // ---- overflow geometry: estep_prep_kernel has suppressed the absent classes (estep.py:46-55) and left EVERY class image
        // class-major in the workspace plus the per-pixel maxima; what remains here is to bring the resident classes into LDS
        // (coalesced 16-byte copies) and the maxima into l.mx.
        const int n4 = hwp >> 2;
        for (int c = 0; c < C; c++) {
            const int sl = __builtin_amdgcn_readfirstlane(l.slot[c]);
            if (sl < 0 || sl >= p_res) continue;
            const float4 *src = reinterpret_cast<const float4 *>(a.ovf + ((size_t)b * C + c) * hwp);
            float4 *dst = reinterpret_cast<float4 *>(l.chan + (size_t)sl * hwp);
#pragma unroll 4
            for (int g = tid; g < n4; g += kThreads) dst[g] = src[g];
        }
        {
            const float4 *src = reinterpret_cast<const float4 *>(a.mxbuf + (size_t)b * hwp);
            float4 *dst = reinterpret_cast<float4 *>(l.mx);
            for (int g = tid; g < n4; g += kThreads) dst[g] = src[g];
        }
        __syncthreads();
    }
#pragma unroll
    for (int e = 0; e < EPT; e++) {  // running max of the thread's own pixels -> registers
        const int i = tid + e * kThreads;
        mx[e] = (i < HW) ? l.mx[i] : 0.f;
    }
    __syncthreads();
    ESTEP_STOP(2);
    const float mean0 = np_mean_lds(l, HW, tid);  // estep.py:56 (thread 0 holds it)
    ESTEP_STOP(3);

    // ---- num_iter sweeps over the classes in the given order (estep.py:62-79) ----
    // The staged order is first compacted to the PRESENT classes (thread `it` rewrites row `it` in place; the counts go where
    // the set-up kept its stack): the sweep then walks P entries per iteration instead of testing all C classes -- each test
    // two dependent LDS reads with nothing to overlap them.
    const bool compact = perm_lds && a.num_iter <= 48;
    if (compact) {
        if (tid < a.num_iter) {
            unsigned short *row = l.perm + tid * (C - 1);
            int n = 0;
            for (int q = 0; q < C - 1; q++) {
                const int j = row[q];
                if ((unsigned)j < (unsigned)C && l.slot[j] >= 0) row[n++] = (unsigned short)j;
            }
            l.stk[tid] = n;
        }
        __syncthreads();
    }
    // Round 5: the LDS-resident geometries walk a FLAT visit list (slot of the class | 0x100 for class 0, built once by wave 0 behind the
    // compacted rows) and keep the visited class image in registers.  What a visit used to put in front of its selection -- order entry
    // -> class -> slot -> channel values, then the values again for the update: five dependent LDS round trips, 1.2 us of the 3 us per
    // visit with nothing to overlap them (scripts/estep_ablate.py) -- is now one list entry and one class image fetched a visit ahead,
    // while the current selection runs (another class's image does not depend on it; the same class twice in a row takes the updated
    // registers).
    const int nper = a.num_iter * (C - 1);
    const bool fast = compact && full && nper + a.num_iter * C <= kPermCap;
    if (fast) {
        unsigned short *vis = l.perm + nper;
        if (tid < 64) {
            const int s0 = l.slot[0];
            int base = 0;
            for (int it = 0; it < a.num_iter; it++) {
                const int n = l.stk[it];
                const unsigned short *row = l.perm + it * (C - 1);
                if (s0 >= 0 && tid == 0) vis[base] = (unsigned short)(s0 | 0x100);   // class 0 first (estep.py:64)
                const int b1 = base + (s0 >= 0 ? 1 : 0);
                for (int q = tid; q < n; q += 64) {
                    const int j = row[q];
                    vis[b1 + q] = (unsigned short)(l.slot[j] | (j == 0 ? 0x100 : 0));
                }
                base = b1 + n;
            }
            if (tid == 0) l.sel[3] = base;
        }
        __syncthreads();
        const int nv = l.sel[3];
        int ii[EPT];
#pragma unroll
        for (int e = 0; e < EPT; e++) ii[e] = min(tid + e * kThreads, HW - 1);   // (loads unconditional; pixels past the end re-read the last one)
        float chv[EPT], chn[EPT];
        int cur = nv > 0 ? (int)vis[0] : 0;
        int nxt = nv > 1 ? (int)vis[1] : cur;
        if (nv > 0) {
            const float *ch = l.chan + (size_t)(cur & 0xff) * hwp;
#pragma unroll
            for (int e = 0; e < EPT; e++) chv[e] = ch[ii[e]];
        }
        for (int n = 0; n < nv; n++) {
            const int nxt2 = (n + 2 < nv) ? (int)vis[n + 2] : nxt;   // (the list entry two visits ahead: the class image one visit ahead needs no dependent read)
            const int k = (cur & 0x100) ? a.k_bg : a.k_fg;
            unsigned key[EPT];
#pragma unroll
            for (int e = 0; e < EPT; e++) key[e] = (tid + e * kThreads < HW) ? __float_as_uint(mx[e] - chv[e]) : 0xFFFFFFFFu;  // >= +0: raw bits are order-preserving
            {
                const float *chx = l.chan + (size_t)(nxt & 0xff) * hwp;   // the next visit's class image: in flight during this selection
#pragma unroll
                for (int e = 0; e < EPT; e++) chn[e] = chx[ii[e]];
            }
            const float t = __uint_as_float(radix_select<EPT>(key, k, l, tid, lane, a.stop_after));  // np.partition(d, k)[k]
            float *ch = l.chan + (size_t)(cur & 0xff) * hwp;
            const bool same = (nxt & 0xff) == (cur & 0xff);
#pragma unroll
            for (int e = 0; e < EPT; e++) {
                const float v = chv[e] + t;
                if (tid + e * kThreads < HW) {
                    ch[ii[e]] = v;
                    mx[e] = (v > mx[e]) ? v : mx[e];
                }
                chv[e] = same ? v : chn[e];
            }
            cur = nxt;
            nxt = nxt2;
        }
    } else
    for (int it = 0; it < a.num_iter; it++) {
        const int nvis = compact ? l.stk[it] : C - 1;
        for (int q = -1; q < nvis; q++) {   // class 0 first (estep.py:64), then the (compacted) order
            const int pi = it * (C - 1) + q;
            const int j = (q < 0) ? 0 : (perm_lds ? (int)l.perm[pi] : a.perms[pi]);
            if ((unsigned)j >= (unsigned)C) continue;
            const int s = l.slot[j];
            if (s < 0) continue;
            const int k = (j == 0) ? a.k_bg : a.k_fg;
            const bool in_lds = s < p_res;
            float *gch = a.ovf + ((size_t)b * C + j) * hwp;   // a non-resident channel lives in the overflow workspace, class-major

            // (two copies of the loads / stores, chosen by a wave-uniform branch: one pointer that may be LDS or HBM makes
            //  every access a FLAT instruction, and the read-modify-write below a chain of dependent FLAT round trips)
            unsigned key[EPT];
            if (in_lds) {
                const float *ch = l.chan + (size_t)s * hwp;
#pragma unroll
                for (int e = 0; e < EPT; e++) {
                    const int i = tid + e * kThreads;
                    key[e] = (i < HW) ? __float_as_uint(mx[e] - ch[i]) : 0xFFFFFFFFu;  // >= +0: raw bits are order-preserving
                }
            } else {
#pragma unroll
                for (int e = 0; e < EPT; e++) {
                    const int i = tid + e * kThreads;
                    key[e] = (i < HW) ? __float_as_uint(mx[e] - gch[i]) : 0xFFFFFFFFu;
                }
            }
            const float t = __uint_as_float(radix_select<EPT>(key, k, l, tid, lane, a.stop_after));  // np.partition(d, k)[k]
            if (in_lds) {
                float *ch = l.chan + (size_t)s * hwp;
#pragma unroll
                for (int e = 0; e < EPT; e++) {
                    const int i = tid + e * kThreads;
                    if (i < HW) {
                        const float v = ch[i] + t;
                        ch[i] = v;
                        mx[e] = (v > mx[e]) ? v : mx[e];
                    }
                }
            } else {
#pragma unroll
                for (int e = 0; e < EPT; e++) {
                    const int i = tid + e * kThreads;
                    if (i < HW) {
                        const float v = gch[i] + t;
                        gch[i] = v;
                        mx[e] = (v > mx[e]) ? v : mx[e];
                    }
                }
            }
        }
    }

    ESTEP_STOP(4);
    // ---- re-centre (estep.py:80-82) and argmax (deeplab.py:123) ----
#pragma unroll
    for (int e = 0; e < EPT; e++) {
        const int i = tid + e * kThreads;
        if (i < HW) l.mx[i] = mx[e];
    }
    __syncthreads();
    const float mean1 = np_mean_lds(l, HW, tid);
    if (tid == 0) l.scal[0] = mean0 - mean1;
    __syncthreads();
    const float shift = l.scal[0];

    if (full) {
        int best[EPT];
        float bv[EPT];
        for (int c = 0; c < C; c++) {
            const float *ch = l.chan + (size_t)c * hwp;
            float v[EPT];
#pragma unroll
            for (int e = 0; e < EPT; e++) v[e] = ch[min(tid + e * kThreads, HW - 1)] + shift;
#pragma unroll
            for (int e = 0; e < EPT; e++)
                if (c == 0 || v[e] > bv[e]) { bv[e] = v[e]; best[e] = c; }
        }
#pragma unroll
        for (int e = 0; e < EPT; e++) {
            const int i = tid + e * kThreads;
            if (i < HW) a.labels_out[(size_t)b * HW + i] = best[e];  // consecutive lanes -> consecutive int64: coalesced
        }
        if (a.map_out) {  // float map of estep.py:82, written with coalesced stores
            float *mo = a.map_out + (size_t)b * HW * C;
            for (int g = tid; g < HW * C; g += kThreads) {
                const int row = g / C, c = g - row * C;
                mo[g] = l.chan[(size_t)c * hwp + row] + shift;
            }
        }
        return;
    }
    {
        // Not every class fits in LDS: the others live class-major in the overflow workspace (round 2 kept them in the NHWC float
        // map, which the host then demanded: 4-byte accesses 4*C bytes apart, 350 of the kernel's 470 us at 65 x 65 x 41).
        // Classes outside, the thread's pixels inside (EPT loads in flight); the float map is written only when asked for.
        int best[EPT];
        float bv[EPT];
        int pi[EPT];
        bool ok[EPT];
#pragma unroll
        for (int e = 0; e < EPT; e++) {
            const int i = tid + e * kThreads;
            ok[e] = i < HW;
            pi[e] = ok[e] ? i : HW - 1;
        }
        if (!a.map_out) {
            // Labels only (the training step).  34 of the 41 class images live in the workspace, and ONE CU reads them at ~30 GB/s
            // (32 of the kernel's 172 us at 65 x 65 x 41).  So this kernel only takes the argmax over its LDS-resident classes
            // (deeplab.py:123 on value + shift; slots ascend with the class, so the first maximum is the lowest class) and leaves
            // (value, class) per pixel and the image's shift; estep_finish_kernel completes the argmax over the other classes on the
            // whole chip: max of the two partial maxima, the lower class on a tie -- the same label as one pass in class order.
            int best[EPT];
            float bv[EPT];
#pragma unroll
            for (int e = 0; e < EPT; e++) { best[e] = -1; bv[e] = 0.f; }
            for (int c = 0; c < C; c++) {
                const int sl = __builtin_amdgcn_readfirstlane(l.slot[c]);
                if (sl < 0 || sl >= p_res) continue;
                const float *ch = l.chan + (size_t)sl * hwp;
#pragma unroll
                for (int e = 0; e < EPT; e++) {
                    const float v = ch[min(tid + e * kThreads, HW - 1)] + shift;
                    if (best[e] < 0 || v > bv[e]) { bv[e] = v; best[e] = c; }
                }
            }
#pragma unroll
            for (int e = 0; e < EPT; e++) {
                const int i = tid + e * kThreads;
                if (i < HW) {
                    a.part_v[(size_t)b * hwp + i] = bv[e];
                    a.part_i[(size_t)b * hwp + i] = best[e];
                }
            }
            if (tid == 0) a.shiftbuf[b] = shift;
            return;
        }
        for (int c = 0; c < C; c++) {
            const int sl = __builtin_amdgcn_readfirstlane(l.slot[c]);
            const float *src = (sl >= 0 && sl < p_res) ? nullptr : a.ovf + ((size_t)b * C + c) * hwp;
            float v[EPT];
            if (src == nullptr) {
#pragma unroll
                for (int e = 0; e < EPT; e++) v[e] = l.chan[(size_t)sl * hwp + pi[e]];
            } else {
#pragma unroll
                for (int e = 0; e < EPT; e++) v[e] = src[pi[e]];
            }
#pragma unroll
            for (int e = 0; e < EPT; e++) {
                v[e] = v[e] + shift;
                if (a.map_out && ok[e]) a.map_out[((size_t)b * HW + pi[e]) * C + c] = v[e];
                if (c == 0 || v[e] > bv[e]) { bv[e] = v[e]; best[e] = c; }
            }
        }
#pragma unroll
        for (int e = 0; e < EPT; e++)
            if (ok[e]) a.labels_out[(size_t)b * HW + pi[e]] = best[e];
    }
}

template <int EPT>
int launch_estep(const EstepArgs &a, int B, size_t lds, hipStream_t stream) {
    static DynLdsAttr attr;
    {
        const int arc = attr.ensure(reinterpret_cast<const void *>(&estep_kernel<EPT>), kLdsBudget);
        if (arc) return arc;
    }
    hipLaunchKernelGGL(estep_kernel<EPT>, dim3(B), dim3(kThreads), lds, stream, a);
    return launch_status();
}

int choose_p_res(int hwp, int C) {
    const size_t fixed = lds_bytes(hwp, 0);
    if (fixed >= kLdsBudget) return -1;
    size_t p = (kLdsBudget - fixed) / (sizeof(float) * (size_t)hwp);
    if (p > (size_t)C) p = C;
    return (int)p;
}

__global__ void resize_nearest_kernel(const float *__restrict__ in, int B, int H, int W, float *__restrict__ out,
                                      int oh, int ow, float hs, float ws) {
    const long long n = (long long)B * oh * ow;
    for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < n;
         idx += (long long)gridDim.x * blockDim.x) {
        const int x = (int)(idx % ow);
        const int y = (int)((idx / ow) % oh);
        const int b = (int)(idx / ((long long)ow * oh));
        int sy = (int)floorf((float)y * hs);
        int sx = (int)floorf((float)x * ws);
        sy = sy < H - 1 ? sy : H - 1;
        sx = sx < W - 1 ? sx : W - 1;
        out[idx] = in[((long long)b * H + sy) * W + sx];
    }
}

// Inference head of network.py:39-41: tf.image.resize_bilinear(output,(h,w)) (TF1 legacy mapping, align_corners
// = False: src = dst * in/out, no half-pixel offset) followed by tf.argmax(axis=3).  One thread per output pixel;
// the C interpolated logits are formed on the fly (optionally stored) and reduced to the first maximum.
// fp contraction is switched off in the kernel body so that every multiply and add rounds separately, like the
// restatement (and TF's CPU kernel).
__global__ void __launch_bounds__(256) pred_bilinear_argmax_kernel(const float *__restrict__ logits, long long ldl, int B,
                                                                 int IH, int IW, int C, int OH, int OW, float hs,
                                                                 float ws, long long *__restrict__ pred,
                                                                 float *__restrict__ rescale) {
#pragma clang fp contract(off)  // no FMA: the restatement (and TF's CPU kernel) round the multiply and the add separately
    const long long n = (long long)B * OH * OW;
    for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < n;
         idx += (long long)gridDim.x * blockDim.x) {
        const int x = (int)(idx % OW);
        const int y = (int)((idx / OW) % OH);
        const int b = (int)(idx / ((long long)OW * OH));
        const float fy = (float)y * hs, fx = (float)x * ws;
        const int y0 = (int)floorf(fy), x0 = (int)floorf(fx);
        const int y1 = y0 + 1 < IH ? y0 + 1 : IH - 1, x1 = x0 + 1 < IW ? x0 + 1 : IW - 1;
        const float ly = fy - (float)y0, lx = fx - (float)x0;
        const float *tl = logits + (((long long)b * IH + y0) * IW + x0) * ldl;
        const float *tr = logits + (((long long)b * IH + y0) * IW + x1) * ldl;
        const float *bl = logits + (((long long)b * IH + y1) * IW + x0) * ldl;
        const float *br = logits + (((long long)b * IH + y1) * IW + x1) * ldl;
        int best = 0;
        float bv = 0.f;
        for (int c = 0; c < C; c++) {
            // plain operators under `fp contract(off)`: each multiply and add rounds on its own
            const float top = tl[c] + (tr[c] - tl[c]) * lx;
            const float bot = bl[c] + (br[c] - bl[c]) * lx;
            const float v = top + (bot - top) * ly;
            if (rescale) rescale[idx * C + c] = v;
            if (c == 0 || v > bv) { bv = v; best = c; }
        }
        if (pred) pred[idx] = best;
    }
}

// Tail of dataset.py:175-181: RGB -> BGR, subtract the per-channel mean (dataset.py:15-18), optional left-right flip
// of image and label (image_flip, dataset.py:187-199).  in: f32 [B,H,W,3] RGB (the resized image), label f32 [B,H,W].
__global__ void __launch_bounds__(256) bgr_mean_flip_kernel(const float *__restrict__ img, const float *__restrict__ lab,
                                                          int B, int H, int W, float m0, float m1, float m2,
                                                          const unsigned char *__restrict__ flip,
                                                          float *__restrict__ img_out, float *__restrict__ lab_out) {
    const long long n = (long long)B * H * W;
    for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < n;
         idx += (long long)gridDim.x * blockDim.x) {
        const int x = (int)(idx % W);
        const long long row = idx / W;
        const int b = (int)(row / H);
        const int xs = (flip && flip[b]) ? W - 1 - x : x;
        const long long src = row * W + xs;
        const float r = img[src * 3 + 0], g = img[src * 3 + 1], bl = img[src * 3 + 2];
        img_out[idx * 3 + 0] = bl - m0;
        img_out[idx * 3 + 1] = g - m1;
        img_out[idx * 3 + 2] = r - m2;
        if (lab && lab_out) lab_out[idx] = lab[src];
    }
}

std::mutex g_seed_mu;
uint64_t g_seed_state = 0x9E3779B97F4A7C15ull;

uint64_t splitmix64(uint64_t &s) {
    uint64_t z = (s += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

}  // namespace

extern "C" {

// Workspace: the per-block maxima of the batch-maximum pre-kernel, then -- when not every class of an image fits in LDS (65 x 65
// x 41) -- the class-major overflow [B][C][hwp] of the channels that do not.
static size_t estep_ovf_offset() { return 256; }
size_t emadapt_estep_workspace_bytes(int B, int H, int W, int C) {
    static_assert(kGmaxBlocks * sizeof(float) <= 256, "per-block maxima");
    if (B <= 0 || H <= 0 || W <= 0 || C <= 0) return estep_ovf_offset();
    const long long HW = (long long)H * W;
    if (HW > (long long)kThreads * kMaxEPT || C > kMaxC) return estep_ovf_offset();
    const int hwp = (int)((HW + 3) / 4 * 4);
    const int p_res = choose_p_res(hwp, C);
    if (p_res < 0 || p_res >= C) return estep_ovf_offset();
    // class images + per-pixel maxima + the partial argmax (value, class) + the images' shifts (round 4)
    return estep_ovf_offset() + (size_t)B * (C + 3) * hwp * sizeof(float) + align_up((size_t)B * sizeof(float), 16);
}

// Since round 3 always 0 for a supported geometry (< 0: unsupported): the channels that do not fit in LDS live in the workspace,
// the float map is only an output.  (Kept for callers written against the first version, where 1 meant "map_out required".)
int emadapt_estep_needs_map(int H, int W, int C) {
    if (H <= 0 || W <= 0 || C <= 0) return EMADAPT_ERR_INVALID;
    const long long HW = (long long)H * W;
    if (HW > (long long)kThreads * kMaxEPT || C > kMaxC) return EMADAPT_ERR_UNSUPPORTED;
    const int p_res = choose_p_res((int)((HW + 3) / 4 * 4), C);
    if (p_res < 0) return EMADAPT_ERR_UNSUPPORTED;
    return 0;
}

int emadapt_estep(const float *logits, int64_t ldl, const void *label, int label_kind, int64_t label_len,
                  const int32_t *perms, int B, int H, int W, int C, int suppress_others, int num_iter,
                  float margin_others, int k_bg, int k_fg, float *map_out, int64_t *labels_out,
                  void *workspace, size_t workspace_bytes, void *stream_) {
    hipStream_t stream = as_stream(stream_);
    if (B < 0 || H <= 0 || W <= 0 || C <= 0 || num_iter < 0) return EMADAPT_ERR_INVALID;
    if (B == 0) return EMADAPT_OK;
    if (!logits || !label || !labels_out || ldl < C) return EMADAPT_ERR_INVALID;
    if (num_iter > 0 && C > 1 && !perms) return EMADAPT_ERR_INVALID;
    if (label_kind < 0 || label_kind > 2) return EMADAPT_ERR_INVALID;
    if (label_kind != EMADAPT_LABEL_MULTIHOT && label_len < 0) return EMADAPT_ERR_INVALID;
    const long long HW = (long long)H * W;
    if (HW > (long long)kThreads * kMaxEPT || C > kMaxC) return EMADAPT_ERR_UNSUPPORTED;
    if (num_iter > 0 && (k_bg < 0 || k_bg >= HW || k_fg < 0 || k_fg >= HW)) return EMADAPT_ERR_INVALID;
    if (!workspace || workspace_bytes < emadapt_estep_workspace_bytes(B, H, W, C)) return EMADAPT_ERR_WORKSPACE;

    const int hwp = (int)((HW + 3) / 4 * 4);
    const int p_res = choose_p_res(hwp, C);
    if (p_res < 0) return EMADAPT_ERR_UNSUPPORTED;

    float *gpart = static_cast<float *>(workspace);
    const double bytes = (double)B * HW * (C * 4.0 + 8.0) + (map_out ? (double)B * HW * C * 4.0 : 0.0);
    ProfScope prof(EMADAPT_K_ESTEP, 0.0, bytes, stream);
    if (suppress_others) {
        static_assert(kGmaxBlocks == 64, "the E-step kernel folds one value per lane");
        hipLaunchKernelGGL(estep_gmax_kernel, dim3(kGmaxBlocks), dim3(256), 0, stream, logits, (long long)B * HW, C, (long long)ldl, gpart);
        int rc = launch_status();
        if (rc) return rc;
    }
    EstepArgs a;
    a.logits = logits; a.ldl = ldl; a.label = label; a.label_kind = label_kind; a.label_len = label_len;
    a.perms = perms; a.HW = (int)HW; a.C = C; a.suppress = suppress_others ? 1 : 0; a.num_iter = num_iter;
    a.margin = margin_others; a.k_bg = k_bg; a.k_fg = k_fg; a.gmax_part = gpart; a.map_out = map_out;
    a.ovf = p_res < C ? reinterpret_cast<float *>(static_cast<char *>(workspace) + estep_ovf_offset()) : nullptr;
    a.mxbuf = a.ovf ? a.ovf + (size_t)B * C * hwp : nullptr;
    a.part_v = a.ovf ? a.mxbuf + (size_t)B * hwp : nullptr;
    a.part_i = a.ovf ? reinterpret_cast<int *>(a.part_v + (size_t)B * hwp) : nullptr;
    a.shiftbuf = a.ovf ? a.part_v + 2 * (size_t)B * hwp : nullptr;
    a.labels_out = reinterpret_cast<long long *>(labels_out); a.p_res = p_res; a.hwp = hwp;
    a.full = (p_res >= C) ? 1 : 0;
    pairwise_program(a, (int)HW);
    if (!a.full) {   // overflow geometry: the suppress step for the whole batch, on the whole chip
        hipLaunchKernelGGL(estep_prep_kernel, dim3((unsigned)ceil_div(HW, 256), (unsigned)B), dim3(256), 0, stream, a);
        int rc = launch_status();
        if (rc) return rc;
    }
#ifdef EMADAPT_ABLATE_BUILD
    a.stop_after = tune(TUNE_ESTEP_STOP);
#else
    a.stop_after = 0;
#endif
    const size_t lds = align_up(lds_bytes(hwp, p_res), 16);
    const int ept = (int)ceil_div(HW, kThreads);
    int rc;
    if (ept <= 1) rc = launch_estep<1>(a, B, lds, stream);
    else if (ept <= 4) rc = launch_estep<4>(a, B, lds, stream);   // 41x41 = 1681 (321x321 input)
    else if (ept <= 9) rc = launch_estep<9>(a, B, lds, stream);   // 65x65 = 4225 (513x513 input)
    else rc = launch_estep<16>(a, B, lds, stream);
    if (rc || a.full || a.map_out || a.stop_after) return rc;
    // overflow geometry, labels only: the argmax over the class images in the workspace, on the whole chip
    hipLaunchKernelGGL(estep_finish_kernel, dim3((unsigned)ceil_div(HW, 256), (unsigned)B), dim3(256), 0, stream, a);
    return launch_status();
}

int emadapt_resize_nearest(const float *in, int B, int H, int W, float *out, int oh, int ow, void *stream_) {
    if (!in || !out || B < 0 || H <= 0 || W <= 0 || oh <= 0 || ow <= 0) return EMADAPT_ERR_INVALID;
    if (B == 0) return EMADAPT_OK;
    const long long n = (long long)B * oh * ow;
    int grid = (int)ceil_div(n, 256);
    if (grid > 2048) grid = 2048;
    const float hs = (float)H / (float)oh, ws = (float)W / (float)ow;
    hipStream_t stream = as_stream(stream_);
    ProfScope prof(EMADAPT_K_OTHER, 0.0, (double)n * 8.0, stream);
    hipLaunchKernelGGL(resize_nearest_kernel, dim3(grid), dim3(256), 0, stream, in, B, H, W, out, oh, ow, hs, ws);
    return launch_status();
}

int emadapt_pred_bilinear_argmax(const float *logits, int64_t ldl, int B, int IH, int IW, int C, int OH, int OW,
                                 int64_t *pred, float *rescale_out, void *stream_) {
    if (!logits || (!pred && !rescale_out) || B < 0 || IH <= 0 || IW <= 0 || C <= 0 || OH <= 0 || OW <= 0 || ldl < C)
        return EMADAPT_ERR_INVALID;
    if (B == 0) return EMADAPT_OK;
    const long long n = (long long)B * OH * OW;
    int grid = (int)ceil_div(n, 256);
    if (grid > 8192) grid = 8192;
    const float hs = (float)IH / (float)OH, ws = (float)IW / (float)OW;
    hipStream_t stream = as_stream(stream_);
    ProfScope prof(EMADAPT_K_OTHER, 0.0, (double)B * IH * IW * C * 4.0 + (double)n * 8.0, stream);
    hipLaunchKernelGGL(pred_bilinear_argmax_kernel, dim3(grid), dim3(256), 0, stream, logits, (long long)ldl, B, IH, IW,
                       C, OH, OW, hs, ws, reinterpret_cast<long long *>(pred), rescale_out);
    return launch_status();
}

int emadapt_bgr_mean_flip(const float *img_rgb, const float *label, int B, int H, int W, const float mean_bgr[3],
                          const uint8_t *flip, float *img_out, float *label_out, void *stream_) {
    if (!img_rgb || !img_out || !mean_bgr || B < 0 || H <= 0 || W <= 0) return EMADAPT_ERR_INVALID;
    if (B == 0) return EMADAPT_OK;
    const long long n = (long long)B * H * W;
    int grid = (int)ceil_div(n, 256);
    if (grid > 8192) grid = 8192;
    hipStream_t stream = as_stream(stream_);
    ProfScope prof(EMADAPT_K_OTHER, 0.0, (double)n * 32.0, stream);
    hipLaunchKernelGGL(bgr_mean_flip_kernel, dim3(grid), dim3(256), 0, stream, img_rgb, label, B, H, W, mean_bgr[0],
                       mean_bgr[1], mean_bgr[2], flip, img_out, label_out);
    return launch_status();
}

int emadapt_estep_host(float *map, int B, int H, int W, int C, const void *label, int label_kind,
                       int64_t label_len, const int32_t *perms, int suppress_others, int num_iter,
                       float margin_others, int k_bg, int k_fg, int64_t *labels_out) {
    if (B < 0 || H <= 0 || W <= 0 || C <= 0) return EMADAPT_ERR_INVALID;
    if (B == 0) return EMADAPT_OK;
    if (!map || !label) return EMADAPT_ERR_INVALID;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return EMADAPT_ERR_NODEVICE;
    const size_t HW = (size_t)H * W;
    const size_t n_map = (size_t)B * HW * C;
    const size_t lab_elems = (label_kind == EMADAPT_LABEL_MULTIHOT) ? (size_t)B * C : (size_t)B * label_len;
    const size_t lab_bytes = lab_elems * (label_kind == EMADAPT_LABEL_MULTIHOT ? 1 : 4);
    const size_t perm_elems = (size_t)(num_iter > 0 ? num_iter : 0) * (size_t)(C > 1 ? C - 1 : 0);
    const size_t ws_bytes = emadapt_estep_workspace_bytes(B, H, W, C);

    float *d_in = nullptr, *d_map = nullptr;
    void *d_lab = nullptr, *d_ws = nullptr;
    int32_t *d_perm = nullptr;
    int64_t *d_out = nullptr;
    int rc = EMADAPT_OK;
    auto cleanup = [&]() {
        (void)hipFree(d_in); (void)hipFree(d_map); (void)hipFree(d_lab);
        (void)hipFree(d_ws); (void)hipFree(d_perm); (void)hipFree(d_out);
    };
#define TRY_OR_CLEAN(expr)                          \
    do {                                            \
        hipError_t _e = (expr);                     \
        if (_e != hipSuccess) {                     \
            cleanup();                              \
            return ::emadapt::hip_fail(_e);         \
        }                                           \
    } while (0)
    TRY_OR_CLEAN(hipMalloc(&d_in, n_map * 4));
    TRY_OR_CLEAN(hipMalloc(&d_map, n_map * 4));
    TRY_OR_CLEAN(hipMalloc(&d_lab, lab_bytes ? lab_bytes : 4));
    TRY_OR_CLEAN(hipMalloc(&d_ws, ws_bytes));
    TRY_OR_CLEAN(hipMalloc(&d_perm, perm_elems ? perm_elems * 4 : 4));
    TRY_OR_CLEAN(hipMalloc(&d_out, (size_t)B * HW * 8));
    TRY_OR_CLEAN(hipMemcpy(d_in, map, n_map * 4, hipMemcpyHostToDevice));
    if (lab_bytes) TRY_OR_CLEAN(hipMemcpy(d_lab, label, lab_bytes, hipMemcpyHostToDevice));
    if (perm_elems) {
        if (!perms) { cleanup(); return EMADAPT_ERR_INVALID; }
        TRY_OR_CLEAN(hipMemcpy(d_perm, perms, perm_elems * 4, hipMemcpyHostToDevice));
    }
    rc = emadapt_estep(d_in, C, d_lab, label_kind, label_len, d_perm, B, H, W, C, suppress_others, num_iter,
                       margin_others, k_bg, k_fg, d_map, d_out, d_ws, ws_bytes, nullptr);
    if (rc == EMADAPT_OK) {
        TRY_OR_CLEAN(hipDeviceSynchronize());
        TRY_OR_CLEAN(hipMemcpy(map, d_map, n_map * 4, hipMemcpyDeviceToHost));
        if (labels_out) TRY_OR_CLEAN(hipMemcpy(labels_out, d_out, (size_t)B * HW * 8, hipMemcpyDeviceToHost));
    }
#undef TRY_OR_CLEAN
    cleanup();
    return rc;
}

void emadapt_estep_seed(uint64_t seed) {
    std::lock_guard<std::mutex> lk(g_seed_mu);
    g_seed_state = seed ^ 0x9E3779B97F4A7C15ull;
}

float *e_step_perm(float *map, const int *shape, const int *stride, const int *label, const int *perms,
                   const bool suppress_others, const int num_iter, const float margin_others, float bg, float fg) {
    if (!map || !shape || !stride || !label) return nullptr;
    const int B = shape[0], H = shape[1], W = shape[2], C = shape[3];
    if (B < 0 || H <= 0 || W <= 0 || C <= 0) return nullptr;
    const size_t HW = (size_t)H * W;
    // thresholds as the native reference computes them (weaklabel.cpp:88): float * int, truncated
    const int k_bg = (int)(bg * (float)(int)HW);
    const int k_fg = (int)(fg * (float)(int)HW);
    // gather to a contiguous [B,H,W,C] staging copy honouring the byte strides (estep.py:30)
    const bool contiguous = stride[3] == 4 && stride[2] == 4 * C && stride[1] == 4 * C * W &&
                            stride[0] == (long long)4 * C * W * H;
    std::vector<float> staging;
    float *buf = map;
    if (!contiguous) {
        staging.resize((size_t)B * HW * C);
        const char *base = reinterpret_cast<const char *>(map);
        for (int b = 0; b < B; b++)
            for (int h = 0; h < H; h++)
                for (int w = 0; w < W; w++)
                    for (int c = 0; c < C; c++)
                        staging[(((size_t)b * H + h) * W + w) * C + c] = *reinterpret_cast<const float *>(
                            base + (ptrdiff_t)b * stride[0] + (ptrdiff_t)h * stride[1] + (ptrdiff_t)w * stride[2] +
                            (ptrdiff_t)c * stride[3]);
        buf = staging.data();
    }
    int rc = emadapt_estep_host(buf, B, H, W, C, label, EMADAPT_LABEL_I32, (int64_t)HW, perms, suppress_others,
                                num_iter, margin_others, k_bg, k_fg, nullptr);
    if (rc != EMADAPT_OK) return nullptr;
    if (!contiguous) {
        char *base = reinterpret_cast<char *>(map);
        for (int b = 0; b < B; b++)
            for (int h = 0; h < H; h++)
                for (int w = 0; w < W; w++)
                    for (int c = 0; c < C; c++)
                        *reinterpret_cast<float *>(base + (ptrdiff_t)b * stride[0] + (ptrdiff_t)h * stride[1] +
                                                   (ptrdiff_t)w * stride[2] + (ptrdiff_t)c * stride[3]) =
                            staging[(((size_t)b * H + h) * W + w) * C + c];
    }
    return map;
}

float *e_step(float *map, const int *shape, const int *stride, const int *label, const bool suppress_others,
              const int num_iter, const float margin_others, float bg, float fg) {
    if (!shape) return nullptr;
    const int C = shape[3];
    std::vector<int> perms;
    if (num_iter > 0 && C > 1) {
        perms.resize((size_t)num_iter * (C - 1));
        std::lock_guard<std::mutex> lk(g_seed_mu);
        for (int it = 0; it < num_iter; it++) {
            int *p = perms.data() + (size_t)it * (C - 1);
            for (int c = 0; c < C - 1; c++) p[c] = c + 1;
            for (int i = C - 2; i > 0; i--) {  // Fisher-Yates
                const int j = (int)(splitmix64(g_seed_state) % (uint64_t)(i + 1));
                const int t = p[i]; p[i] = p[j]; p[j] = t;
            }
        }
    }
    return e_step_perm(map, shape, stride, label, perms.empty() ? nullptr : perms.data(), suppress_others, num_iter,
                       margin_others, bg, fg);
}

}  // extern "C"

// bf16x6 on pre-split operands (compute_dtype = "bf16x6"): plane images, the 256x256 LDS-DMA kernels that read them, and
// their C entry points.  Arithmetic and error bound are conv_x6.hip's (x = p0 + p1 + p2 exactly, six bf16 MFMAs per product
// accumulated in fp32); what changes is WHERE the split happens -- once per tensor instead of once per tile and tap -- and with
// it the kernel structure (igemm_x6p.h).  Shapes these kernels do not take stay on conv_x6.hip's kernels; the rules, per kind,
// exactly as xp_ok (below) and wp_ok (wgrad_x6p.h) implement them and emadapt_conv_x6p_supported() reports them:
//   forward / data gradient: reduction channels (Cin resp. Cout) % 16 == 0, N (Cout resp. Cin) >= 256 and % 4 == 0, <= 32 taps;
//   filter gradient:         Cin % 256 == 0 (a 256-wide k tile inside one tap), Cout >= 128 and % 16 == 0, taps x Cin / 256 <= 256;
//   all:                     plane images below 2 GiB.
//
// Plane image of an fp32 tensor [rows][C], C % 16 == 0:   [rows][C / 16][plane 0..2][16 x bf16]    (6 bytes per element)
//   forward:          activation planes [B*H*W][Cin]    x   tiled filter planes, rows = Cout, k = Cin   (emadapt_filter_tiles)
//   data gradient:    dy planes [B*H*W][Cout]           x   tiled filter planes, rows = Cin, k = Cout
//   filter gradient:  activation planes, dy planes -> fp32 HWIO gradient
#include <stdlib.h>

#include "igemm_x6p.h"
#include "wgrad_x6p.h"

namespace {

typedef __bf16 bf16x8v __attribute__((ext_vector_type(8)));

// ------------------------------------------------------------------------------------------------------------------
// fp32 [rows][C] -> plane image.  One thread = 8 consecutive channels (half a group of 16): 32 bytes in, 3 x 16 bytes out.
// ------------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ void split8(const float4 v0, const float4 v1, uint4 &p0, uint4 &p1, uint4 &p2) {
    const float x[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
    bf16x8v a, b, c;
#pragma unroll
    for (int e = 0; e < 8; e++) {
        a[e] = (__bf16)x[e];
        const float r1 = x[e] - (float)a[e];  // exact
        b[e] = (__bf16)r1;
        const float r2 = r1 - (float)b[e];    // exact
        c[e] = (__bf16)r2;
    }
    p0 = *reinterpret_cast<uint4 *>(&a);
    p1 = *reinterpret_cast<uint4 *>(&b);
    p2 = *reinterpret_cast<uint4 *>(&c);
}

__global__ void __launch_bounds__(256) split_planes_kernel(const float4 *__restrict__ x, uint4 *__restrict__ out, long long n8) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n8; i += (long long)gridDim.x * 256) {
        const float4 v0 = x[2 * i], v1 = x[2 * i + 1];
        uint4 p0, p1, p2;
        split8(v0, v1, p0, p1, p2);
        uint4 *o = out + (i >> 1) * 6 + (i & 1);   // group of 16 channels: 6 chunks of 16 bytes; this thread's half
        o[0] = p0;
        o[2] = p1;
        o[4] = p2;
    }
}

// ------------------------------------------------------------------------------------------------------------------
// Tiled plane image of a filter for igemm_x6p_kernel (see igemm_x6p.h): [tap][k / 16][row (zero rows up to a multiple of 256)]
// [6 chunks], chunk position = (2 plane + half) ^ (bit 3 of the row).  DGRAD = false: rows = Cout, k = Cin (forward);
// true: rows = Cin, k = Cout (data gradient).  One thread = 8 consecutive k of one row.
// ------------------------------------------------------------------------------------------------------------------
template <bool DGRAD>
__global__ void __launch_bounds__(256) filter_tiles_kernel(const float *__restrict__ w, uint4 *__restrict__ out, int Cin, int Cout,
                                                           int taps, int rows_pad) {
    const int rows = DGRAD ? Cin : Cout, kdim = DGRAD ? Cout : Cin;
    const int G = kdim / 16;
    const long long total = (long long)taps * G * rows_pad * 2;
    for (long long idx = (long long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long long)gridDim.x * 256) {
        const int h = (int)(idx & 1);
        long long t = idx >> 1;
        const int row = (int)(t % rows_pad);
        t /= rows_pad;
        const int g = (int)(t % G), tap = (int)(t / G);
        float v[8];
#pragma unroll
        for (int e = 0; e < 8; e++) {
            const int k = g * 16 + h * 8 + e;
            v[e] = row < rows ? (DGRAD ? w[((long long)tap * Cin + row) * Cout + k] : w[((long long)tap * Cin + k) * Cout + row]) : 0.f;
        }
        uint4 p0, p1, p2;
        split8(make_float4(v[0], v[1], v[2], v[3]), make_float4(v[4], v[5], v[6], v[7]), p0, p1, p2);
        const int par = (row >> 3) & 1;
        uint4 *o = out + (((long long)tap * G + g) * rows_pad + row) * 6;
        o[(0 + h) ^ par] = p0;
        o[(2 + h) ^ par] = p1;
        o[(4 + h) ^ par] = p2;
    }
}

int launch_split_planes(const float *x, void *planes, long long elems, hipStream_t stream) {
    const long long n8 = elems / 8;
    if (n8 == 0) return EMADAPT_OK;
    long long g = (n8 + 255) / 256;
    if (g > 16384) g = 16384;
    hipLaunchKernelGGL(split_planes_kernel, dim3((unsigned)g), dim3(256), 0, stream, reinterpret_cast<const float4 *>(x),
                       reinterpret_cast<uint4 *>(planes), n8);
    return launch_status();
}

// ------------------------------------------------------------------------------------------------------------------
// Launch plan of the forward / data-gradient kernel: one workgroup per CU.  Whole rounds of tiles run as plain tiles; whatever is
// not a whole number of rounds runs stream-K (over packed rows where the filter's padding is worth skipping: igemm_v2.h).
// ------------------------------------------------------------------------------------------------------------------
struct PlanXP {
    int main_rows;     // rows of the first launch
    int main_sk;       // 1: stream-K
    int main_levels;   // its slab levels
    int bands;         // packed rows: pixel classes (0 = off)
    int tail_levels;   // rows [main_rows, M): a stream-K launch with this many slab levels (-1: a plain launch)
};

bool xp_ok(const IgemmArgs &a) {
    const long long plane_x = (long long)a.M * a.Cx * 6, plane_w = (long long)a.K * ceil_div(a.N, 256) * 256 * 6;
    // (per-lane offset + scalar offset must stay below bit 31, the out-of-range flag: image + the taps' reach + one pixel row)
    const long long reach = ((long long)(a.ntaps / a.KW - 1) * a.dil * a.W + (long long)(a.KW - 1) * a.dil + 1) * a.Cx * 6;
    return a.Cx % 16 == 0 && a.ntaps <= 32 && a.N % 4 == 0 && a.N >= 256 && plane_x + reach < (1LL << 31) && plane_w < (1LL << 31) &&
           (long long)a.M * a.N < (1LL << 40);
}

PlanXP plan_x6p(const IgemmArgs &a) {
    const int cus = plan_cus();
    PlanXP p = {a.M, 0, 0, 0, 0};
    const long long tn = ceil_div(a.N, 256), tm = ceil_div(a.M, 256), t = tm * tn;
    const int nk16 = a.K / 16;
    if (tune(TUNE_FP32_LOADER) != 2) {
        IgemmArgs b = a;
        if (build_row_pack(b, (int)tn) && b.pk.run_u0[b.pk.nruns] / cus >= 32) {
            p.main_sk = 1;
            p.bands = b.pk.n;
            p.main_levels = streamk_levels_packed(b, (int)tn, cus);
            return p;
        }
    }
    if (t % cus == 0 || t * nk16 / cus < 32) return p;   // whole rounds, or too little work per CU to cut into pieces
    // fewer tiles than CUs: stream-K over everything.  Otherwise whole rounds run as plain tiles in the XCD-aware order -- the 32
    // workgroups of an XCD then walk the SAME K step of neighbouring tiles and share its operand slices in that L2, which these
    // kernels feel (fc7 as one stream-K launch: 16 % slower, profiles/r04_streamk_combine.txt) -- and only the rows of the partial
    // last round run stream-K (conv3_x: 256 + 155 tiles; round 3 ran all 411 stream-K because N <= 1024 bounded the slab levels)
    if (t < cus) {
        p.main_sk = 1;
        p.main_levels = streamk_levels(t, nk16, cus);
        return p;
    }
    const long long tm_full = (t / cus) * cus / tn;
    if (tm_full >= 1 && tm_full < tm) {
        p.main_rows = (int)(tm_full * 256);
        const long long tail_tiles = (tm - tm_full) * tn;
        p.tail_levels = tail_tiles * nk16 / cus >= 32 ? streamk_levels(tail_tiles, nk16, cus) : -1;
    }
    return p;
}

size_t xp_workspace_bytes(const IgemmArgs &a) {
    const PlanXP p = plan_x6p(a);
    // stream-K launches with shared tiles: arrival words + one 256 KB slab per workgroup (igemm_v2.h), then the row table
    size_t b = p.main_levels > 0 || (p.main_rows < a.M && p.tail_levels > 0) ? streamk_workspace_bytes(plan_cus()) : 0;
    if (p.bands) b += (size_t)ceil_div(a.M, 256) * 256 * sizeof(int);
    return b;
}

int dispatch_x6p(IgemmArgs &a, void *workspace, size_t workspace_bytes, hipStream_t stream, ProfScope &prof) {
    const PlanXP p = plan_x6p(a);
    if (xp_workspace_bytes(a) > 0 && (!workspace || workspace_bytes < xp_workspace_bytes(a))) return EMADAPT_ERR_WORKSPACE;
    a.ablate = 0;
    a.tap_inner = 1;
    a.pk.n = 0;
    a.pk.rowmap = nullptr;
    a.m_begin = 0;
    a.m_end = p.main_rows;
    if (p.main_sk) {
        if (p.bands) {
            double exec = 1.0;
            if (!build_row_pack(a, (int)ceil_div(a.N, 256), &exec)) return EMADAPT_ERR_INVALID;
            prof.executed_fraction(exec);
            a.pk.rowmap = reinterpret_cast<const int *>(static_cast<char *>(workspace) + (p.main_levels > 0 ? streamk_workspace_bytes(plan_cus()) : 0));
        }
        a.splits = plan_cus();
        a.sk_flags = p.main_levels > 0 ? static_cast<unsigned *>(workspace) : nullptr;
        a.slab = p.main_levels > 0 ? reinterpret_cast<float *>(static_cast<char *>(workspace) + streamk_flag_bytes(plan_cus())) : nullptr;
        return launch_igemm_x6p<true>(a, stream);
    }
    a.splits = 1;
    a.slab = nullptr;
    a.sk_flags = nullptr;
    int rc = launch_igemm_x6p<false>(a, stream);
    if (rc || p.main_rows >= a.M) return rc;
    a.m_begin = p.main_rows;
    a.m_end = a.M;
    if (p.tail_levels < 0) return launch_igemm_x6p<false>(a, stream);
    a.splits = plan_cus();
    a.sk_flags = p.tail_levels > 0 ? static_cast<unsigned *>(workspace) : nullptr;
    a.slab = p.tail_levels > 0 ? reinterpret_cast<float *>(static_cast<char *>(workspace) + streamk_flag_bytes(plan_cus())) : nullptr;
    return launch_igemm_x6p<true>(a, stream);
}

void fill_x6p(IgemmArgs &a, const emadapt_conv_desc *d, bool dgrad) {
    fill_geometry(a, d, dgrad);
    a.x_bytes = (unsigned)((long long)a.M * (a.Cx / 16) * XP_ROWB);
    a.w_bytes = (unsigned)((long long)a.K / 16 * ceil_div(a.N, 256) * 256 * XP_ROWB);   // tiled filter image: rows padded to whole tiles
}

}  // namespace

namespace emadapt {
// (for the layers the plane kernels take: emadapt_conv_x6p_supported / _pays; the others run elsewhere and skip nothing)
double exec_fraction_x6p(const emadapt_conv_desc *d, int kind) {
    if (check_desc(d) != EMADAPT_OK || d->B == 0) return 1.0;
    if (kind == 2) {
        if (!wp_ok(d) || d->KH * d->KW <= 1 || tune(TUNE_FP32_LOADER) == 2) return 1.0;
        return taps_inside_fraction(d->H, d->W, d->KH, d->KW, d->dil, conv_pad_before(d->KH, d->dil), conv_pad_before(d->KW, d->dil));
    }
    IgemmArgs a;
    fill_x6p(a, d, kind == 1);
    if (!xp_ok(a)) return 1.0;
    const PlanXP p = plan_x6p(a);
    double e = 1.0;
    if (p.main_sk && p.bands && build_row_pack(a, (int)ceil_div(a.N, 256), &e)) return e;
    return 1.0;
}
}  // namespace emadapt

extern "C" {

size_t emadapt_planes_bytes(long long rows, int C) { return rows > 0 && C > 0 ? (size_t)rows * (size_t)C * 6 : 0; }

int emadapt_split_planes(const float *x, void *planes, long long rows, int C, void *stream_) {
    if (rows < 0 || C <= 0 || C % 16 != 0) return EMADAPT_ERR_INVALID;
    if (rows == 0) return EMADAPT_OK;
    if (!x || !planes || !aligned16(x) || !aligned16(planes)) return EMADAPT_ERR_INVALID;
    return launch_split_planes(x, planes, rows * C, as_stream(stream_));
}

size_t emadapt_filter_tiles_bytes(const emadapt_conv_desc *d, int dgrad) {
    if (check_desc(d) != EMADAPT_OK) return 0;
    const int rows = dgrad ? d->Cin : d->Cout, kdim = dgrad ? d->Cout : d->Cin;
    if (kdim % 16 != 0) return 0;
    return (size_t)d->KH * d->KW * (kdim / 16) * (size_t)(ceil_div(rows, 256) * 256) * XP_ROWB;
}

int emadapt_filter_tiles(const emadapt_conv_desc *d, const float *w, void *tiles, int dgrad, void *stream_) {
    int rc = check_desc(d);
    if (rc) return rc;
    if (!w || !tiles || !aligned16(w) || !aligned16(tiles)) return EMADAPT_ERR_INVALID;
    const int rows = dgrad ? d->Cin : d->Cout, kdim = dgrad ? d->Cout : d->Cin;
    if (kdim % 16 != 0) return EMADAPT_ERR_UNSUPPORTED;
    const int rows_pad = (int)(ceil_div(rows, 256) * 256), taps = d->KH * d->KW;
    long long g = ((long long)taps * (kdim / 16) * rows_pad * 2 + 255) / 256;
    if (g > 16384) g = 16384;
    if (dgrad)
        hipLaunchKernelGGL(filter_tiles_kernel<true>, dim3((unsigned)g), dim3(256), 0, as_stream(stream_), w,
                           reinterpret_cast<uint4 *>(tiles), d->Cin, d->Cout, taps, rows_pad);
    else
        hipLaunchKernelGGL(filter_tiles_kernel<false>, dim3((unsigned)g), dim3(256), 0, as_stream(stream_), w,
                           reinterpret_cast<uint4 *>(tiles), d->Cin, d->Cout, taps, rows_pad);
    return launch_status();
}

int emadapt_conv_x6p_supported(const emadapt_conv_desc *d, int kind) {
    if (check_desc(d) != EMADAPT_OK || kind < 0 || kind > 2) return 0;
    if (kind == 2) return wp_ok(d) ? 1 : 0;
    IgemmArgs a;
    fill_x6p(a, d, kind == 1);
    return xp_ok(a) ? 1 : 0;
}

// Advice for a host that has both kernel families: does the plane kernel have enough work to fill the part?  One workgroup per
// CU and 256 x 256 tiles need either stream-K's >= 32 K steps per CU or whole rounds of tiles; a single image's conv4_x (14 tiles
// of 288 steps) would run on 14 CUs -- conv_x6.hip's 128 x 128 split-K launches are the better fp32-grade choice there.
int emadapt_conv_x6p_pays(const emadapt_conv_desc *d, int kind) {
    if (!emadapt_conv_x6p_supported(d, kind) || d->B == 0) return 0;
    const int cus = plan_cus();
    if (kind == 2) {
        const PlanWP p = plan_wgrad_x6p(d);
        return (long long)p.tiles_k * p.tiles_n * p.splits * 2 >= cus ? 1 : 0;
    }
    IgemmArgs a;
    fill_x6p(a, d, kind == 1);
    const long long t = ceil_div(a.M, 256) * ceil_div(a.N, 256);
    return t * (a.K / 16) >= 32LL * cus ? 1 : 0;
}

size_t emadapt_conv_fwd_x6p_workspace_bytes(const emadapt_conv_desc *d) {
    if (check_desc(d) != EMADAPT_OK || d->B == 0) return 0;
    IgemmArgs a;
    fill_x6p(a, d, false);
    return xp_ok(a) ? xp_workspace_bytes(a) : 0;
}

size_t emadapt_conv_dgrad_x6p_workspace_bytes(const emadapt_conv_desc *d) {
    if (check_desc(d) != EMADAPT_OK || d->B == 0) return 0;
    IgemmArgs a;
    fill_x6p(a, d, true);
    return xp_ok(a) ? xp_workspace_bytes(a) : 0;
}

int emadapt_conv_x6p_plan(const emadapt_conv_desc *d, int dgrad, int plan[5]) {
    int rc = check_desc(d);
    if (rc) return rc;
    if (!plan) return EMADAPT_ERR_INVALID;
    IgemmArgs a;
    fill_x6p(a, d, dgrad != 0);
    if (!xp_ok(a)) return EMADAPT_ERR_UNSUPPORTED;
    const PlanXP p = plan_x6p(a);
    plan[0] = p.main_rows; plan[1] = p.main_sk; plan[2] = p.main_levels; plan[3] = p.bands; plan[4] = p.tail_levels;
    return EMADAPT_OK;
}

int emadapt_conv_fwd_x6p(const emadapt_conv_desc *d, const void *x_planes, const void *wk_planes, const float *bias, float *y,
                         int epilogue, float keep_prob, const uint8_t *drop_mask, uint64_t drop_seed, void *workspace,
                         size_t workspace_bytes, void *stream_) {
    int rc = check_desc(d);
    if (rc) return rc;
    if (!x_planes || !wk_planes || !y) return EMADAPT_ERR_INVALID;
    if (epilogue < EMADAPT_EPI_NONE || epilogue > EMADAPT_EPI_BIAS_RELU) return EMADAPT_ERR_INVALID;
    if (epilogue != EMADAPT_EPI_NONE && !bias) return EMADAPT_ERR_INVALID;
    if (!(keep_prob > 0.f) || keep_prob > 1.f) return EMADAPT_ERR_INVALID;
    if (keep_prob < 1.f && epilogue != EMADAPT_EPI_BIAS_RELU) return EMADAPT_ERR_INVALID;
    if (!aligned16(x_planes) || !aligned16(wk_planes) || !aligned16(y) || !aligned16(workspace)) return EMADAPT_ERR_INVALID;
    if (d->B == 0) return EMADAPT_OK;
    IgemmArgs a;
    fill_x6p(a, d, false);
    if (!xp_ok(a)) return EMADAPT_ERR_UNSUPPORTED;
    a.x = static_cast<const float *>(x_planes); a.w = static_cast<const float *>(wk_planes); a.y = y; a.bias = bias;
    a.gate = nullptr; a.mask = drop_mask;
    a.epi = epilogue; a.scale = 1.f; set_keep(a, keep_prob); a.seed = drop_seed;
    hipStream_t stream = as_stream(stream_);
    ProfScope prof(EMADAPT_K_IGEMM_FWD, 2.0 * a.M * (double)a.K * a.N, 6.0 * ((double)a.M * a.Cx + (double)a.K * a.N) + 4.0 * a.M * a.N, stream);   // plane images 6 B / element
    return dispatch_x6p(a, workspace, workspace_bytes, stream, prof);
}

int emadapt_conv_dgrad_x6p(const emadapt_conv_desc *d, const void *dy_planes, const void *w_planes, float *dx, const float *gate,
                           float gate_scale, void *workspace, size_t workspace_bytes, void *stream_) {
    int rc = check_desc(d);
    if (rc) return rc;
    if (!dy_planes || !w_planes || !dx) return EMADAPT_ERR_INVALID;
    if (!aligned16(dy_planes) || !aligned16(w_planes) || !aligned16(dx) || !aligned16(workspace)) return EMADAPT_ERR_INVALID;
    if (d->B == 0) return EMADAPT_OK;
    IgemmArgs a;
    fill_x6p(a, d, true);
    if (!xp_ok(a)) return EMADAPT_ERR_UNSUPPORTED;
    a.x = static_cast<const float *>(dy_planes); a.w = static_cast<const float *>(w_planes); a.y = dx; a.bias = nullptr;
    a.gate = gate; a.mask = nullptr;
    a.epi = EMADAPT_EPI_NONE; a.scale = gate_scale; set_keep(a, 1.f); a.seed = 0;
    hipStream_t stream = as_stream(stream_);
    ProfScope prof(EMADAPT_K_IGEMM_DGRAD, 2.0 * a.M * (double)a.K * a.N, 6.0 * ((double)a.M * a.Cx + (double)a.K * a.N) + 4.0 * a.M * a.N * (gate ? 2.0 : 1.0), stream);
    return dispatch_x6p(a, workspace, workspace_bytes, stream, prof);
}

size_t emadapt_conv_wgrad_x6p_workspace_bytes(const emadapt_conv_desc *d) {
    if (check_desc(d) != EMADAPT_OK || d->B == 0 || !wp_ok(d)) return 0;
    const PlanWP p = plan_wgrad_x6p(d);
    if (p.splits <= 1) return 0;
    return (size_t)p.splits * d->KH * d->KW * d->Cin * d->Cout * sizeof(float);
}

int emadapt_conv_wgrad_x6p(const emadapt_conv_desc *d, const void *x_planes, const void *dy_planes, float *dw, void *workspace,
                           size_t workspace_bytes, void *stream_) {
    int rc = check_desc(d);
    if (rc) return rc;
    if (!wp_ok(d)) return EMADAPT_ERR_UNSUPPORTED;
    if (!x_planes || !dy_planes || !dw) return EMADAPT_ERR_INVALID;
    if (!aligned16(x_planes) || !aligned16(dy_planes) || !aligned16(dw) || !aligned16(workspace)) return EMADAPT_ERR_INVALID;
    hipStream_t stream = as_stream(stream_);
    const long long KN = (long long)d->KH * d->KW * d->Cin * d->Cout;
    if (d->B == 0) {
        EMADAPT_HIP_TRY(hipMemsetAsync(dw, 0, KN * sizeof(float), stream));
        return EMADAPT_OK;
    }
    const PlanWP p = plan_wgrad_x6p(d);
    if (p.splits > 1 && (!workspace || workspace_bytes < emadapt_conv_wgrad_x6p_workspace_bytes(d))) return EMADAPT_ERR_WORKSPACE;
    WgradXPArgs a;
    a.xp = static_cast<const char *>(x_planes); a.dyp = static_cast<const char *>(dy_planes);
    a.out = p.splits > 1 ? static_cast<float *>(workspace) : dw;
    a.H = d->H; a.W = d->W; a.Cin = d->Cin; a.N = d->Cout; a.KW = d->KW; a.dil = d->dil;
    a.pad_h = conv_pad_before(d->KH, d->dil); a.pad_w = conv_pad_before(d->KW, d->dil);
    a.M = d->B * d->H * d->W; a.ntaps = d->KH * d->KW; a.K = a.ntaps * d->Cin;
    a.tiles_k = p.tiles_k; a.tiles_n = p.tiles_n; a.splits = p.splits; a.rows_per_split = p.rows_per_split;
    a.x_bytes = (unsigned)((long long)a.M * d->Cin * 6); a.y_bytes = (unsigned)((long long)a.M * d->Cout * 6);
    a.magic_w = magic_u32(d->W); a.magic_h = magic_u32(d->H);
    // skip the pixels at which a tile's tap is outside the image (fp32_loader = 2 keeps the dense walk: tests, A/B)
    a.skip = a.ntaps > 1 && tune(TUNE_FP32_LOADER) != 2;
    a.lockstep = tune(TUNE_DMA_STAGGER) == 1 ? 1 : 0;
    if (a.skip) wp_order_k_tiles(a);
    ProfScope prof(EMADAPT_K_IGEMM_WGRAD, 2.0 * a.M * (double)a.K * a.N, 6.0 * ((double)a.M * d->Cin + (double)a.M * a.N) + 4.0 * a.K * a.N, stream);
    if (a.skip) prof.executed_fraction(taps_inside_fraction(d->H, d->W, d->KH, d->KW, d->dil, a.pad_h, a.pad_w));
    rc = launch_wgrad_x6p(a, stream);
    if (rc) return rc;
    if (p.splits > 1) rc = launch_slab_sum(static_cast<const float *>(workspace), p.splits, KN, dw, stream);
    return rc;
}

}  // extern "C"

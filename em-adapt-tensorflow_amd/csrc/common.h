// Shared host-side helpers for libemadapt_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/emadapt.h"

namespace emadapt {

extern thread_local int g_last_hip_error;

inline int hip_fail(hipError_t e) {
    g_last_hip_error = (int)e;
    return EMADAPT_ERR_HIP;
}

#define EMADAPT_HIP_TRY(expr)                                   \
    do {                                                        \
        hipError_t _e = (expr);                                 \
        if (_e != hipSuccess) return ::emadapt::hip_fail(_e);   \
    } while (0)

inline hipStream_t as_stream(void *s) { return reinterpret_cast<hipStream_t>(s); }

// After a kernel launch: surface launch-configuration errors without synchronising.
inline int launch_status() {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return hip_fail(e);
    return EMADAPT_OK;
}

// Device address of the status word of the current device (include/emadapt.h, emadapt_status_poll): a __device__ variable of
// the library's code object -- no allocation, no stream operation, safe under graph capture.  nullptr when HIP cannot resolve it.
unsigned *status_word();
// 1 while emadapt_status_collective(1) is in force (process-wide, like the tune keys): see "Collective mode" below
int status_collective();

// ---- optimiser arithmetic (deeplab.py:202-205), shared by elementwise.hip and conv_bf16.hip's fused update ----
// The optimiser's arithmetic, PINNED.  hipcc contracts `mu * v + a * s` into an fma around EITHER product by its own heuristics: two
// kernels with the same source expression (Momentum on a loaded accumulator / on one computed in registers) came out 1 ulp apart on
// 28 % of the elements.  These spell out what the round-1..4 vector kernels compiled to (the values every checkpoint and test since
// round 1 were produced with): the decay and accumulate steps are fmas, Momentum rounds mu * v and fuses the gradient term, the
// weight step is an fma.  (HIP's default -ffp-contract=fast-honor-pragmas honours the pragma in mul_rn.)
__device__ __forceinline__ float mul_rn(float a, float b) {
#pragma clang fp contract(off)
    return a * b;
}
__device__ __forceinline__ float opt_decay(float g, float wd, float w) { return __builtin_fmaf(wd, w, g); }                       // g + wd w
__device__ __forceinline__ float opt_accum(float a, float g, float scale) { return __builtin_fmaf(g, scale, a); }                  // a + g scale
__device__ __forceinline__ float opt_velocity(float mu, float v, float a, float s) { return __builtin_fmaf(a, s, mul_rn(mu, v)); }  // mu v + a s
__device__ __forceinline__ float opt_step(float w, float lr, float v) { return __builtin_fmaf(-lr, v, w); }                        // w - lr v


// The device status word (include/emadapt.h, emadapt_status_poll): while a kernel has a failure recorded there the optimiser
// kernels return at once, on the device -- a gradient with a poisoned tile never reaches weights, momentum or accumulators, and no
// host synchronisation is needed for that (the word is read once per wave: a scalar load).
__device__ __forceinline__ bool status_raised(const unsigned *status) { return status && __builtin_nontemporal_load(status) != 0u; }

// Collective mode of the status word (emadapt_status_collective(1): data-parallel jobs, where "this rank's word is raised" must not
// decide anything a peer decides differently).  The failure TRAVELS WITH THE GRADIENT instead: the kernels that feed the exchange
// (accumulate, accumulate + pack, pack) emit NaN for everything while the local word is raised, sums over ranks keep a NaN, and the
// optimiser kernels -- which then ignore the local word -- leave an element whose summed gradient is NaN alone (weight and momentum
// kept, accumulator still zeroed) and raise EMADAPT_STATUS_NAN_GRADIENT on their own device.  Every rank reads the same summed
// gradient, so every rank skips the same elements and every rank's next poll reports the failure.
__device__ __forceinline__ float poison_nan() { return __builtin_nanf(""); }
__device__ __forceinline__ void opt_update(float &w, float &v, float a, float lr, float mu, float ascale, bool coll, bool &bad) {
    const float vn = opt_velocity(mu, v, a, ascale);
    const float wn = opt_step(w, lr, vn);
    const bool nan = coll && a != a;
    v = nan ? v : vn;
    w = nan ? w : wn;
    bad = bad || nan;
}
__device__ __forceinline__ void status_raise(unsigned *status, unsigned bits) {
    if (status) __hip_atomic_fetch_or(status, bits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// ---- per-kernel event timing (bench.py roofline; see emadapt_prof_* in emadapt.h) ----
struct ProfScope {
    ProfScope(int kind, double flops, double bytes, hipStream_t stream);
    ~ProfScope();
    // FLOPs the launch really issues when that is less than the algorithmic count given to the constructor (kernels that
    // skip the products with a dilated filter's zero padding): fraction of the algorithmic FLOPs, 1.0 by default
    void executed_fraction(double f);
    int kind_;
    hipStream_t stream_;
    void *rec_;
};

constexpr int kWave = 64;

// Filter gradient with pixel skipping: (pixel, tap) pairs at which the tap reads inside the image / all of them.
inline double taps_inside_fraction(int H, int W, int KH, int KW, int dil, int pad_h, int pad_w) {
    double walked = 0.0;
    for (int tap = 0; tap < KH * KW; tap++) {
        const int dh = (tap / KW) * dil - pad_h, dw = (tap % KW) * dil - pad_w;
        const int vh = H - (dh < 0 ? -dh : dh), vw = W - (dw < 0 ? -dw : dw);
        walked += (double)(vh > 0 ? vh : 0) * (vw > 0 ? vw : 0);
    }
    return walked / ((double)KH * KW * H * W);
}
// executed / algorithmic FLOPs of the launches a conv entry makes for this shape (host arithmetic only)
double exec_fraction_fp32(const emadapt_conv_desc *d, int kind);   // conv_igemm.hip (kind 0 / 1), conv_wgrad.hip (kind 2)
double exec_fraction_fp32_wgrad(const emadapt_conv_desc *d);
double exec_fraction_x6p(const emadapt_conv_desc *d, int kind);    // conv_x6p.hip

// Division by a launch-time constant d as one v_mul_hi_u32: magic = ceil(2^32 / d) gives n / d exactly while
// n * d < 2^32 (checked on the host for every quantity divided this way); d == 1 is encoded as magic 0 (identity).
inline unsigned magic_u32(int d) { return d <= 1 ? 0u : (unsigned)((0x100000000ULL + (unsigned)d - 1) / (unsigned)d); }
__device__ __forceinline__ unsigned div_magic(unsigned n, unsigned magic) { return magic ? __umulhi(n, magic) : n; }

// ReLU and the ReLU-gradient gate as the fused epilogues apply them.  Both PROPAGATE a NaN like tf.nn.relu (deeplab.py:71,100: TF's
// max(x, 0) returns NaN for NaN) -- `v > 0 ? v : 0` maps NaN to 0 and would erase the poison a timed-out stream-K head leaves in a
// tile (round-4 verdict) or an overflowed activation.  relu(-0.0) is -0.0 (the comparison is false), which no consumer tells
// from +0.0.  The gate keeps ReluGrad's select (gate > 0 ? dy : 0) for numbers and lets a NaN gradient through either branch.
__device__ __forceinline__ float relu_keep_nan(float v) { return v < 0.f ? 0.f : v; }
__device__ __forceinline__ float gate_keep_nan(float gate, float v_scaled) { return gate > 0.f ? v_scaled : (v_scaled != v_scaled ? v_scaled : 0.f); }

// ---- launch-plan overrides (emadapt_tune_set in emadapt.h): 0 = automatic.  They select among kernel variants that
// all compute the same result (the parity tests run every variant on small awkward shapes through them).
enum TuneKey {
    TUNE_FP32_TILE,    // 1..4: tile variant of the fp32 forward / data-gradient launch (conv_igemm.hip: LaunchPlan)
    TUNE_WGRAD_GEN,    // 1 / 2: kernel generation of the fp32 filter-gradient launch (conv_wgrad.hip)
    TUNE_SPLITK,       // force the split-K factor of the forward / data-gradient launches (when K allows it)
    TUNE_BF16_TILE,    // 128 | 256
    TUNE_X6_TILE,      // 128 | 256
    TUNE_PLAN_CUS,     // CU count the launch plans are laid out for (default 256)
    TUNE_FP32_LOADER,  // 1: loaders of the 256x256 fp32 kernel compute their offsets per K step for any tap count (the
                       //    automatic choice only for filters other than 1x1 / 3x3 / 4x4); 0: tap-specialised where available,
                       //    filter gradient skips out-of-image pixel-tap pairs; 2: tap-specialised, no skipping
    TUNE_PACK_PCT,     // packed rows (igemm_common.h RowPack) are used when they save at least this many per cent of the dense
                       //    launch's products (0 = the default, 5)
    TUNE_ESTEP_STOP,   // diagnostic builds only (-DEMADAPT_ABLATE_BUILD): leave the E-step kernel after phase n
    TUNE_ABLATE,       // diagnostic builds only: ablation bit mask of the fp32 igemm kernel
    TUNE_DMA_STAGGER,  // 1: the LDS-DMA kernels without DMA/MFMA wave roles (igemm_x6p.h, igemm_b16t.h) issue every wave's DMA burst at the
                       //    same point again (round 3's form; A/B of the round-4 stagger of SIMD partners)
    TUNE_RASTER_GM,    // M tiles per group of the plain-tile launches' XCD-aware order (0 = the default, 8): the workgroups resident on
                       //    one XCD cover GM M-tiles x the N-tiles that fit (scripts/raster_sweep.py measures the L2 hit rate and the time)
    TUNE_WGRAD_BIAS,   // 1: emadapt_conv_wgrad_bias_fused() answers 0 everywhere (the separate bias-gradient kernel: A/B, tests)
    TUNE_SK_DROP_FLAG, // tests only: 1 = the stream-K publishers do not raise their arrival words and the heads give up after 2^12 polls
                       //    -- exercises the bounded wait (a shared tile then comes out as NaNs instead of hanging the queue)
    TUNE_BF16T_SK,     // 1: the LDS-DMA bf16 kernel's launches run stream-K over ALL rows whatever N (default: N > 1024 runs the rows of whole
                       //    rounds as plain tiles + a stream-K tail); A/B
    TUNE_POOL_ROWS,    // 1: the stride-2 max-pool gradient runs the row form (elementwise.hip maxpool_bwd_rows_kernel) instead of quads (A/B, tests)
    TUNE_COUNT
};
int tune(TuneKey key);
inline int plan_cus() { const int c = tune(TUNE_PLAN_CUS); return c > 0 ? c : 256; }

// hipFuncAttributeMaxDynamicSharedMemorySize is a per-device property of a kernel: set it once per (kernel, device).
struct DynLdsAttr {
    unsigned done_mask = 0;  // bit d: set on device d (devices >= 32 are set on every launch)
    int ensure(const void *kernel, size_t bytes);
};

inline int64_t ceil_div(int64_t a, int64_t b) { return (a + b - 1) / b; }
inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

}  // namespace emadapt

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

// ---- per-kernel event timing (bench.py roofline; see emadapt_prof_* in emadapt.h) ----
struct ProfScope {
    ProfScope(int kind, double flops, double bytes, hipStream_t stream);
    ~ProfScope();
    int kind_;
    hipStream_t stream_;
    void *rec_;
};

constexpr int kWave = 64;

inline int64_t ceil_div(int64_t a, int64_t b) { return (a + b - 1) / b; }
inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

}  // namespace emadapt

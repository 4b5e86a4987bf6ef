// Library-level entry points: version, error strings, event-based per-kernel timing.
#include <string.h>

#include <atomic>
#include <mutex>
#include <vector>

#include "common.h"

namespace emadapt {

thread_local int g_last_hip_error = 0;

namespace {
struct ProfRec {
    int kind;
    double flops, bytes, flops_exec;
    hipEvent_t start, stop;
};
std::mutex g_prof_mu;
bool g_prof_on = false;
std::vector<ProfRec *> g_prof_recs;
std::vector<ProfRec *> g_prof_free;
}  // namespace

namespace {
std::atomic<int> g_tune[TUNE_COUNT];
const char *const kTuneNames[TUNE_COUNT] = {"fp32_tile", "wgrad_gen", "splitk", "bf16_tile", "x6_tile", "plan_cus", "fp32_loader",
                                            "pack_pct", "estep_stop", "ablate", "dma_stagger", "raster_gm", "wgrad_bias", "sk_drop_flag", "bf16t_sk", "pool_rows"};
std::mutex g_attr_mu;
}  // namespace

int tune(TuneKey key) { return g_tune[key].load(std::memory_order_relaxed); }

namespace {
std::atomic<int> g_status_collective{0};
}
int status_collective() { return g_status_collective.load(std::memory_order_relaxed); }

// The status word: one per device, living in the code object (a module is loaded per device, so the symbol has an address per
// device); its address is looked up once per device and cached.
__device__ unsigned g_device_status[4];   // [0] the flags; the rest pads the word onto a line of its own use
namespace {
std::mutex g_status_mu;
unsigned *g_status_ptr[64];
}  // namespace
unsigned *status_word() {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0) return nullptr;
    std::lock_guard<std::mutex> lk(g_status_mu);
    if (dev < 64 && g_status_ptr[dev]) return g_status_ptr[dev];
    void *p = nullptr;
    if (hipGetSymbolAddress(&p, HIP_SYMBOL(g_device_status)) != hipSuccess) return nullptr;
    if (dev < 64) g_status_ptr[dev] = static_cast<unsigned *>(p);
    return static_cast<unsigned *>(p);
}

int DynLdsAttr::ensure(const void *kernel, size_t bytes) {
    int dev = 0;
    EMADAPT_HIP_TRY(hipGetDevice(&dev));
    const unsigned bit = (dev >= 0 && dev < 32) ? 1u << dev : 0u;
    std::lock_guard<std::mutex> lk(g_attr_mu);
    if (bit && (done_mask & bit)) return EMADAPT_OK;
    EMADAPT_HIP_TRY(hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
    done_mask |= bit;
    return EMADAPT_OK;
}

ProfScope::ProfScope(int kind, double flops, double bytes, hipStream_t stream)
    : kind_(kind), stream_(stream), rec_(nullptr) {
    if (!g_prof_on) return;
    std::lock_guard<std::mutex> lk(g_prof_mu);
    ProfRec *r;
    if (!g_prof_free.empty()) {
        r = g_prof_free.back();
        g_prof_free.pop_back();
    } else {
        r = new ProfRec();
        if (hipEventCreate(&r->start) != hipSuccess || hipEventCreate(&r->stop) != hipSuccess) {
            delete r;
            return;
        }
    }
    r->kind = kind;
    r->flops = flops;
    r->flops_exec = flops;
    r->bytes = bytes;
    (void)hipEventRecord(r->start, stream);
    rec_ = r;
}

void ProfScope::executed_fraction(double f) {
    if (rec_) static_cast<ProfRec *>(rec_)->flops_exec = static_cast<ProfRec *>(rec_)->flops * f;
}

ProfScope::~ProfScope() {
    if (!rec_) return;
    ProfRec *r = static_cast<ProfRec *>(rec_);
    (void)hipEventRecord(r->stop, stream_);
    std::lock_guard<std::mutex> lk(g_prof_mu);
    g_prof_recs.push_back(r);
}

}  // namespace emadapt

using namespace emadapt;

extern "C" {

int emadapt_version(void) { return EMADAPT_ABI_VERSION; }

const char *emadapt_strerror(int code) {
    switch (code) {
        case EMADAPT_OK: return "ok";
        case EMADAPT_ERR_INVALID: return "invalid argument";
        case EMADAPT_ERR_UNSUPPORTED: return "unsupported shape";
        case EMADAPT_ERR_HIP: return "HIP runtime error";
        case EMADAPT_ERR_WORKSPACE: return "workspace missing or too small";
        case EMADAPT_ERR_NODEVICE: return "no gfx950 device";
        case EMADAPT_ERR_DEVICE_STATUS: return "a kernel reported a failure in the device status word";
        default: return "unknown error";
    }
}

int emadapt_last_hip_error(void) { return g_last_hip_error; }

int emadapt_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

int emadapt_status_word(void **device_ptr) {
    if (!device_ptr) return EMADAPT_ERR_INVALID;
    *device_ptr = status_word();
    return *device_ptr ? EMADAPT_OK : EMADAPT_ERR_HIP;
}

int emadapt_status_poll(unsigned *flags, int clear, void *stream_) {
    unsigned *word = status_word();
    if (!word) return EMADAPT_ERR_HIP;
    hipStream_t stream = as_stream(stream_);
    unsigned host = 0;
    EMADAPT_HIP_TRY(hipMemcpyAsync(&host, word, sizeof(unsigned), hipMemcpyDeviceToHost, stream));
    if (clear) EMADAPT_HIP_TRY(hipMemsetAsync(word, 0, sizeof(unsigned), stream));
    EMADAPT_HIP_TRY(hipStreamSynchronize(stream));
    if (flags) *flags = host;
    return host ? EMADAPT_ERR_DEVICE_STATUS : EMADAPT_OK;
}

namespace {
__global__ void status_raise_kernel(unsigned *word, unsigned bits) { __hip_atomic_fetch_or(word, bits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
}  // namespace

int emadapt_status_raise(unsigned bits, void *stream_) {
    unsigned *word = status_word();
    if (!word) return EMADAPT_ERR_HIP;
    if (!bits) return EMADAPT_OK;
    hipLaunchKernelGGL(status_raise_kernel, dim3(1), dim3(1), 0, as_stream(stream_), word, bits);
    return launch_status();
}

// ---- pinned staging for the input pipeline (data.py): the decode workers are separate processes that write into a shared-memory
// block; registering it with the runtime makes the batch's one host-to-device copy a true asynchronous DMA
int emadapt_host_register(void *ptr, size_t bytes) {
    if (!ptr || !bytes) return EMADAPT_ERR_INVALID;
    EMADAPT_HIP_TRY(hipHostRegister(ptr, bytes, hipHostRegisterDefault));
    return EMADAPT_OK;
}

int emadapt_host_unregister(void *ptr) {
    if (!ptr) return EMADAPT_ERR_INVALID;
    EMADAPT_HIP_TRY(hipHostUnregister(ptr));
    return EMADAPT_OK;
}

int emadapt_memcpy_h2d_async(void *dst_device, const void *src_host, size_t bytes, void *stream_) {
    if (!bytes) return EMADAPT_OK;
    if (!dst_device || !src_host) return EMADAPT_ERR_INVALID;
    EMADAPT_HIP_TRY(hipMemcpyAsync(dst_device, src_host, bytes, hipMemcpyHostToDevice, as_stream(stream_)));
    return EMADAPT_OK;
}

int emadapt_status_collective(int on) {
    const int old = g_status_collective.exchange(on ? 1 : 0, std::memory_order_relaxed);
    return old;
}

int emadapt_tune_set(const char *key, int value) {
    if (!key) return EMADAPT_ERR_INVALID;
    for (int k = 0; k < TUNE_COUNT; k++)
        if (strcmp(key, kTuneNames[k]) == 0) {
            g_tune[k].store(value, std::memory_order_relaxed);
            return EMADAPT_OK;
        }
    return EMADAPT_ERR_INVALID;
}

int emadapt_tune_get(const char *key, int *value) {
    if (!key || !value) return EMADAPT_ERR_INVALID;
    for (int k = 0; k < TUNE_COUNT; k++)
        if (strcmp(key, kTuneNames[k]) == 0) {
            *value = g_tune[k].load(std::memory_order_relaxed);
            return EMADAPT_OK;
        }
    return EMADAPT_ERR_INVALID;
}

void emadapt_prof_enable(int on) {
    std::lock_guard<std::mutex> lk(g_prof_mu);
    g_prof_on = on != 0;
}

void emadapt_prof_reset(void) {
    std::lock_guard<std::mutex> lk(g_prof_mu);
    for (ProfRec *r : g_prof_recs) g_prof_free.push_back(r);
    g_prof_recs.clear();
}

int emadapt_prof_collect(int64_t launches[EMADAPT_K_COUNT], double ms[EMADAPT_K_COUNT],
                         double flops[EMADAPT_K_COUNT], double bytes[EMADAPT_K_COUNT]) {
    return emadapt_prof_collect2(launches, ms, flops, bytes, nullptr);
}

int emadapt_prof_collect2(int64_t launches[EMADAPT_K_COUNT], double ms[EMADAPT_K_COUNT], double flops[EMADAPT_K_COUNT],
                          double bytes[EMADAPT_K_COUNT], double flops_executed[EMADAPT_K_COUNT]) {
    if (!launches || !ms || !flops || !bytes) return EMADAPT_ERR_INVALID;
    std::lock_guard<std::mutex> lk(g_prof_mu);
    for (int k = 0; k < EMADAPT_K_COUNT; k++) {
        launches[k] = 0;
        ms[k] = flops[k] = bytes[k] = 0.0;
        if (flops_executed) flops_executed[k] = 0.0;
    }
    for (ProfRec *r : g_prof_recs) {
        EMADAPT_HIP_TRY(hipEventSynchronize(r->stop));
        float t = 0.f;
        EMADAPT_HIP_TRY(hipEventElapsedTime(&t, r->start, r->stop));
        int k = (r->kind >= 0 && r->kind < EMADAPT_K_COUNT) ? r->kind : EMADAPT_K_OTHER;
        launches[k] += 1;
        ms[k] += t;
        flops[k] += r->flops;
        bytes[k] += r->bytes;
        if (flops_executed) flops_executed[k] += r->flops_exec;
    }
    return EMADAPT_OK;
}

}  // extern "C"

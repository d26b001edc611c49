// Library identity + the optional launch profiler.
// The profiler is measurement infrastructure for bench.py's `roofline` object: the only mutable global
// state in the library, guarded by a mutex, and inert unless enabled.  Start/stop events are attached to
// the kernel dispatch itself (hipExtLaunchKernelGGL in LORA_LAUNCH), so a record is the kernel's own
// duration on the caller's stream — the same quantity rocprofv3 --kernel-trace reports.
#include <atomic>
#include <mutex>
#include <vector>

#include "common.h"

namespace {

struct ProfRecord {
    hipEvent_t e0, e1;
    int kind;
    double bytes, flops;
};
struct ProfState {
    std::mutex mu;
    std::vector<ProfRecord> rec;  // preallocated events
    int used = 0;
    bool on = false;
};
ProfState& prof() {
    static ProfState s;
    return s;
}
thread_local double tl_bytes = 0.0, tl_flops = 0.0;
std::atomic<bool> g_null_mode{false};

const char* const kKernelNames[LORA_PROF_KINDS] = {
    "lora_gemm_kernel<*, 128, 128|160, true>", "lora_gemm_kernel<*, 256, 128, true>", "lora_gemm_kernel<*, 64, 64|128|160, true>",
    "lora_gemm_kernel<*, 128, 64, false>", "lora_gemm_kernel<*, 64, 64, false>", "lora_grad_{mfma_,}kernel, ranks <= 4",
    "lora_grad_{mfma_,}kernel, ranks 5-8", "lora_grad_{mfma_,}kernel, ranks 9-16", "ddpm_mse_kernel",
    "other",
    "geglu_linear_bwd: lora_gemm_kernel<*, 128, 128, GATE=2> (frozen ff.net.2 backward GEMM + GEGLU gate backward)",
    "attn_flash_fwd_kernel", "attn_flash_dq_kernel", "attn_flash_dkdv_kernel", "attn_ctx_fwd_kernel",
    "attn_ctx_bwd_kernel (+ attn_ctx_reduce_kernel)", "lora_gemm_kernel<*, 64|128, 128, true> split-K (in-launch combine)",
    "lora_grad_mfma_planned_kernel (every rank class of a pass in one launch)"};

}  // namespace

void lora_prof_set_work(double bytes, double flops) {
    tl_bytes = bytes;
    tl_flops = flops;
}

bool lora_prof_acquire(int kernel_id, hipEvent_t* e0, hipEvent_t* e1) {
    ProfState& s = prof();
    if (!s.on) return false;
    std::lock_guard<std::mutex> lk(s.mu);
    if (!s.on || s.used >= (int)s.rec.size()) return false;
    ProfRecord& r = s.rec[s.used++];
    r.kind = kernel_id;
    r.bytes = tl_bytes;
    r.flops = tl_flops;
    *e0 = r.e0;
    *e1 = r.e1;
    return true;
}

bool lora_prof_null_on() { return g_null_mode.load(std::memory_order_relaxed); }

extern "C" int lora_prof_null_mode(int on) {
    g_null_mode.store(on != 0, std::memory_order_relaxed);
    return LORA_OK;
}

extern "C" int lora_version(void) { return LORA_HIP_ABI_VERSION; }

extern "C" const char* lora_status_string(int status) {
    switch (status) {
        case LORA_OK: return "ok";
        case LORA_E_BADARG: return "bad argument (null pointer, non-positive size or unknown dtype)";
        case LORA_E_RANK: return "LoRA rank must be >= 1 and <= min(in_features, out_features)";
        case LORA_E_ALIGN: return "pointer must be 16-byte aligned";
        case LORA_E_LAUNCH: return "HIP kernel launch failed";
        case LORA_E_UNSUPPORTED: return "unsupported combination";
        default: return "unknown status";
    }
}

extern "C" const char* lora_prof_kernel_name(int kind) {
    return (kind >= 0 && kind < LORA_PROF_KINDS) ? kKernelNames[kind] : "";
}

extern "C" int lora_prof_enable(int capacity) {
    ProfState& s = prof();
    std::lock_guard<std::mutex> lk(s.mu);
    for (auto& r : s.rec) {
        (void)hipEventDestroy(r.e0);
        (void)hipEventDestroy(r.e1);
    }
    s.rec.clear();
    s.used = 0;
    s.on = false;
    if (capacity <= 0) return LORA_OK;
    s.rec.resize(capacity);
    for (auto& r : s.rec) {
        if (hipEventCreate(&r.e0) != hipSuccess || hipEventCreate(&r.e1) != hipSuccess) return LORA_E_LAUNCH;
    }
    s.on = true;
    return LORA_OK;
}

extern "C" int lora_prof_collect(lora_prof_totals* out) {
    if (!out) return LORA_E_BADARG;
    ProfState& s = prof();
    std::lock_guard<std::mutex> lk(s.mu);
    for (int k = 0; k < LORA_PROF_KINDS; ++k) {
        out->launches[k] = 0;
        out->ms[k] = out->bytes[k] = out->flops[k] = 0.0;
    }
    for (int i = 0; i < s.used; ++i) {
        ProfRecord& r = s.rec[i];
        if (hipEventSynchronize(r.e1) != hipSuccess) return LORA_E_LAUNCH;
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, r.e0, r.e1) != hipSuccess) return LORA_E_LAUNCH;
        if (r.kind >= 0 && r.kind < LORA_PROF_KINDS) {
            out->launches[r.kind] += 1;
            out->ms[r.kind] += ms;
            out->bytes[r.kind] += r.bytes;
            out->flops[r.kind] += r.flops;
        }
    }
    s.used = 0;
    return LORA_OK;
}

__global__ void lora_zero_ticket_kernel(unsigned* ticket) {
    if (threadIdx.x < 4) ticket[threadIdx.x] = 0u;
}

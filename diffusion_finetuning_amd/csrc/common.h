// Shared device helpers for liblora_hip (gfx950 / CDNA4 only: wave64, MFMA, 160 KiB LDS).
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdint.h>

#include "../../include/lora_hip.h"

typedef _Float16 half_t;
typedef __bf16 bf16_t;
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <typename T> struct ElemTraits;
template <> struct ElemTraits<float> {
    static constexpr int kDtype = LORA_F32;
    static constexpr int kVec = 4;  // elements per 16-byte chunk
};
template <> struct ElemTraits<half_t> {
    static constexpr int kDtype = LORA_F16;
    static constexpr int kVec = 8;
};
template <> struct ElemTraits<bf16_t> {
    static constexpr int kDtype = LORA_BF16;
    static constexpr int kVec = 8;
};

template <typename T> __device__ __forceinline__ float to_f32(T v) { return static_cast<float>(v); }
template <typename T> __device__ __forceinline__ T from_f32(float v) { return static_cast<T>(v); }

// 16-byte chunk of VEC elements, usable as a register-resident vector.
template <typename T> struct alignas(16) Chunk {
    T v[ElemTraits<T>::kVec];
};

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

// GEGLU gate arithmetic of diffusers GEGLU.forward (exact-erf gelu), shared by sandwich.hip and the GATE epilogue of
// lora_gemm.hip.  fp32 tensors: erff / expf.  16-bit tensors: Φ(g) from the Abramowitz–Stegun 7.1.26 form
//     erfc(z) = t·(a1 + t·(a2 + t·(a3 + t·(a4 + t·a5))))·exp(−z²),  t = 1/(1 + p·z),  z = |g|/√2        (|ε| ≤ 1.5e-7)
// used on the erfc side for g < 0 (no cancellation), one v_rcp_f32 + one v_exp_f32 + 8 VALU instead of erff's ~55: its error
// is 3e-4 of a half-precision ulp for g > 0 and stays under a tenth of an ulp down to g = −3 (below that |gelu| < 4e-3·|g|),
// and exp(−g²/2) comes out of the same evaluation for the derivative.  The erff form cost 31 µs per 16384×1280 gate tile
// pass inside the GEMM epilogue (VALU-bound there), this one 12.
__device__ __forceinline__ void gelu_parts_fast(float g, float& half, float& e) {  // half = Φ(−|g|), e = exp(−g²/2)
    const float z = fabsf(g) * 0.70710678118654752440f;
    const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, z, 1.f));
    e = __expf(-z * z);
    // the A&S coefficients halved: Φ(−|g|) = erfc(z)/2
    const float poly = t * fmaf(t, fmaf(t, fmaf(t, fmaf(t, 0.5307027145f, -0.7265760135f), 0.7107068705f), -0.142248368f), 0.127414796f);
    half = poly * e;
}
template <typename T> __device__ __forceinline__ float gelu_f(float g) {
    if constexpr (sizeof(T) == 4) {
        return 0.5f * g * (1.f + erff(g * 0.70710678118654752440f));
    } else {
        // g·Φ(g) = max(g, 0) − |g|·Φ(−|g|) on both sides of zero: no select, no 1 − x
        float half, e;
        gelu_parts_fast(g, half, e);
        return fmaf(-fabsf(g), half, fmaxf(g, 0.f));
    }
}
template <typename T> __device__ __forceinline__ float gelu_grad_f(float g) {  // Φ(g) + g·φ(g)
    if constexpr (sizeof(T) == 4) {
        return 0.5f * (1.f + erff(g * 0.70710678118654752440f)) + g * 0.39894228040143267794f * expf(-0.5f * g * g);
    } else {
        float half, e;
        gelu_parts_fast(g, half, e);
        const float Phi = g < 0.f ? half : 1.f - half;
        return fmaf(g * 0.39894228040143267794f, e, Phi);
    }
}

static inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

// Zeroes the 16-byte ticket header of a reduction workspace with a KERNEL, not hipMemsetAsync: inside a captured
// hipGraph a memset node is not reliably ordered before the kernel node that follows it when the graph is launched
// into an idle queue (observed on ROCm 7.2 / gfx950: the last-arriver count of ddpm_mse_kernel was reset mid-kernel and
// the loss came out as a partial sum), while kernel→kernel ordering holds.
__global__ void lora_zero_ticket_kernel(unsigned* ticket);
static inline bool lora_zero_ticket(void* workspace, hipStream_t s) {
    hipLaunchKernelGGL(lora_zero_ticket_kernel, dim3(1), dim3(64), 0, s, static_cast<unsigned*>(workspace));
    return hipGetLastError() == hipSuccess;
}

// Launch-profiler hooks (prof.hip).  An entry point declares the algorithmic work of its next launch with a
// ProfWork object; LORA_LAUNCH attaches start/stop events to the dispatch itself (hipExtLaunchKernelGGL), so
// the recorded time is the kernel's own duration, not the gap between host-side event records.
enum ProfKernel {
    PK_GEMM_128x128 = 0, PK_GEMM_256x128, PK_GEMM_64x64, PK_SKINNY_128, PK_SKINNY_64,
    PK_GRAD_R4, PK_GRAD_R8, PK_GRAD_R16, PK_MSE, PK_OTHER,
    PK_GATED_BWD,                                  // frozen ff.net.2 backward-input GEMM with the GEGLU gate's backward (f-4)
    PK_FLASH_FWD, PK_FLASH_DQ, PK_FLASH_DKDV,      // long-context attention core (f-4)
    PK_CTX_FWD, PK_CTX_BWD,                        // short-context attention core (f-4; BWD includes its ordered chunk sum)
    PK_GEMM_SPLITK,                                // fused GEMM launches whose contraction is cut into K-slices (in-launch combine)
    PK_GRAD_PLANNED,                               // every factor-gradient problem of a pass in ONE launch, whatever its rank
    PK_COUNT
};
static_assert(PK_COUNT == LORA_PROF_KINDS, "lora_hip.h LORA_PROF_KINDS out of date");
void lora_prof_set_work(double bytes, double flops);
bool lora_prof_acquire(int kernel_id, hipEvent_t* e0, hipEvent_t* e1);
struct ProfWork {
    ProfWork(double bytes, double flops) { lora_prof_set_work(bytes, flops); }
    ~ProfWork() { lora_prof_set_work(0.0, 0.0); }
};

// Launch-floor mode (lora_prof_null_mode): the empty stand-in of a kernel — same parameter list, hence the same kernel-argument
// segment; it loads two argument words (so the segment is really fetched) and returns.
bool lora_prof_null_on();
template <typename... A> __global__ void lora_null_kernel(A...) {
#if defined(__HIP_DEVICE_COMPILE__)
    const auto* k = (const __attribute__((address_space(4))) unsigned*)__builtin_amdgcn_kernarg_segment_ptr();
    if (k[0] == 0x9e3779b9u && k[1] == 0x7f4a7c15u) __builtin_trap();
#endif
}
template <typename... A> constexpr auto lora_null_for(void (*)(A...)) -> void (*)(A...) { return &lora_null_kernel<A...>; }

#define LORA_LAUNCH(id, kern, grid, block, lds, stream, ...)                                                        \
    do {                                                                                                             \
        hipEvent_t e0_, e1_;                                                                                         \
        if (lora_prof_null_on()) {                                                                                   \
            auto nk_ = lora_null_for(kern);                                                                          \
            if ((lds) > 48 * 1024)                                                                                   \
                (void)hipFuncSetAttribute(reinterpret_cast<const void*>(nk_), hipFuncAttributeMaxDynamicSharedMemorySize, (lds)); \
            if (lora_prof_acquire(id, &e0_, &e1_))                                                                   \
                hipExtLaunchKernelGGL(nk_, grid, block, lds, stream, e0_, e1_, 0, __VA_ARGS__);                      \
            else                                                                                                     \
                hipLaunchKernelGGL(nk_, grid, block, lds, stream, __VA_ARGS__);                                      \
        } else if (lora_prof_acquire(id, &e0_, &e1_))                                                                \
            hipExtLaunchKernelGGL(kern, grid, block, lds, stream, e0_, e1_, 0, __VA_ARGS__);                         \
        else                                                                                                         \
            hipLaunchKernelGGL(kern, grid, block, lds, stream, __VA_ARGS__);                                         \
    } while (0)

#define LORA_LAUNCH_CHECK()                                   \
    do {                                                      \
        if (hipGetLastError() != hipSuccess) return LORA_E_LAUNCH; \
    } while (0)

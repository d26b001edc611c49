// Shared device helpers for liblora_hip (gfx950 / CDNA4 only: wave64, MFMA, 160 KiB LDS).
#pragma once
#include <hip/hip_runtime.h>
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

static inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

// Launch-profiler hooks (prof.hip).  begin returns a slot (<0: profiling off).
int lora_prof_begin(int kind, double bytes, double flops, hipStream_t stream);
void lora_prof_end(int slot, hipStream_t stream);

#define LORA_LAUNCH_CHECK()                                   \
    do {                                                      \
        if (hipGetLastError() != hipSuccess) return LORA_E_LAUNCH; \
    } while (0)

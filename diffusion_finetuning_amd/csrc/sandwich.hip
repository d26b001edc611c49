// The two ops sandwiched by the hot path inside a transformer block, as streaming HIP kernels:
//   * GEGLU gate   out = h · gelu(g),  [h | g] = proj(x)            (diffusers GEGLU.forward, the caller of the
//     `proj` LoraInjectedLinear — target class "GEGLU", lora_diffusion/lora.py:53).  The stock composite runs
//     chunk → gelu → mul as strided elementwise kernels over the largest activation of the model ([M, 8d]), and
//     three more in backward; here forward is one pass (read 2C, write C per row) and backward one pass producing
//     the contiguous dY[M, 2C] that the LoRA backward kernels consume directly.
//   * head split / merge around softmax(QKᵀ/√d)V: [B, N, H·d] ↔ [B, H, N, D] with D ≥ d zero-padded, so that the
//     attention core can run at a head size its kernels are tuned for (64 / 128) without generic strided copies.
// All HBM-bound, 16-byte accesses, no reductions.
#include "common.h"

namespace {

template <typename T>
__global__ __launch_bounds__(256) void geglu_fwd_kernel(const T* y, T* out, int64_t M, int C) {
    constexpr int VEC = ElemTraits<T>::kVec;
    const int cpr = C / VEC;  // chunks per output row
    const int64_t total = M * cpr;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int64_t m = i / cpr;
        const int c = (int)(i - m * cpr) * VEC;
        const Chunk<T> h = *reinterpret_cast<const Chunk<T>*>(y + m * 2 * C + c);
        const Chunk<T> g = *reinterpret_cast<const Chunk<T>*>(y + m * 2 * C + C + c);
        Chunk<T> o;
#pragma unroll
        for (int e = 0; e < VEC; ++e) o.v[e] = from_f32<T>(to_f32<T>(h.v[e]) * gelu_f<T>(to_f32<T>(g.v[e])));
        *reinterpret_cast<Chunk<T>*>(out + m * C + c) = o;
    }
}

template <typename T>
__global__ __launch_bounds__(256) void geglu_bwd_kernel(const T* y, const T* dout, T* dy, int64_t M, int C) {
    constexpr int VEC = ElemTraits<T>::kVec;
    const int cpr = C / VEC;
    const int64_t total = M * cpr;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int64_t m = i / cpr;
        const int c = (int)(i - m * cpr) * VEC;
        const Chunk<T> h = *reinterpret_cast<const Chunk<T>*>(y + m * 2 * C + c);
        const Chunk<T> g = *reinterpret_cast<const Chunk<T>*>(y + m * 2 * C + C + c);
        const Chunk<T> d = *reinterpret_cast<const Chunk<T>*>(dout + m * C + c);
        Chunk<T> dh, dg;
#pragma unroll
        for (int e = 0; e < VEC; ++e) {
            const float gv = to_f32<T>(g.v[e]), dv = to_f32<T>(d.v[e]);
            dh.v[e] = from_f32<T>(dv * gelu_f<T>(gv));
            dg.v[e] = from_f32<T>(dv * to_f32<T>(h.v[e]) * gelu_grad_f<T>(gv));
        }
        *reinterpret_cast<Chunk<T>*>(dy + m * 2 * C + c) = dh;
        *reinterpret_cast<Chunk<T>*>(dy + m * 2 * C + C + c) = dg;
    }
}

// split: src [B, N, H·d] → dst [B, H, N, D] (columns d..D-1 zero);  merge: src [B, H, N, D] → dst [B, N, H·d]
template <typename T, bool SPLIT>
__global__ __launch_bounds__(256) void heads_kernel(const T* src, T* dst, int B, int N, int H, int d, int D,
                                                    int64_t sB, int64_t sH, int64_t sN) {  // strides of the 4-D side
    constexpr int VEC = ElemTraits<T>::kVec;
    const int cpd = D / VEC;  // chunks per padded head row
    const int64_t total = (int64_t)B * H * N * cpd;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int ch = (int)(i % cpd);
        const int64_t row = i / cpd;
        // walk the rows in the order in which the 4-D side lies in memory: (b,h,n) for a contiguous [B,H,N,D] tensor,
        // (b,n,h) for a transposed view of [B,N,H,D] — then both sides are streamed linearly
        const bool n_major = sH < sN;
        const int n = (int)(n_major ? (row / H) % N : row % N);
        const int h = (int)(n_major ? row % H : (row / N) % H);
        const int64_t b = row / ((int64_t)N * H);
        const int col = ch * VEC;
        const int64_t flat = ((b * N + n) * H + h) * d + col;  // [B,N,H·d] side
        const int64_t off4 = b * sB + h * sH + n * sN + col;   // [B,H,N,D] side (rows may be strided)
        if (SPLIT) {
            Chunk<T> v;
            if (col < d) {
                v = *reinterpret_cast<const Chunk<T>*>(src + flat);
            } else {
#pragma unroll
                for (int e = 0; e < VEC; ++e) v.v[e] = from_f32<T>(0.f);
            }
            *reinterpret_cast<Chunk<T>*>(dst + off4) = v;
        } else if (col < d) {
            *reinterpret_cast<Chunk<T>*>(dst + flat) = *reinterpret_cast<const Chunk<T>*>(src + off4);
        }
    }
}

template <typename T>
int launch_geglu(const void* y, const void* dout, void* out, int64_t M, int C, bool bwd, hipStream_t s) {
    constexpr int VEC = ElemTraits<T>::kVec;
    if (C % VEC) return LORA_E_UNSUPPORTED;
    int64_t blocks = (M * (C / VEC) + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    if (blocks < 1) blocks = 1;
    if (bwd)
        hipLaunchKernelGGL(geglu_bwd_kernel<T>, dim3((unsigned)blocks), dim3(256), 0, s, static_cast<const T*>(y),
                           static_cast<const T*>(dout), static_cast<T*>(out), M, C);
    else
        hipLaunchKernelGGL(geglu_fwd_kernel<T>, dim3((unsigned)blocks), dim3(256), 0, s, static_cast<const T*>(y),
                           static_cast<T*>(out), M, C);
    LORA_LAUNCH_CHECK();
    return LORA_OK;
}

template <typename T>
int launch_heads(const void* src, void* dst, int B, int N, int H, int d, int D, int64_t sB, int64_t sH, int64_t sN,
                 bool split, hipStream_t s) {
    constexpr int VEC = ElemTraits<T>::kVec;
    if (d % VEC || D % VEC || D < d || sB % VEC || sH % VEC || sN % VEC) return LORA_E_UNSUPPORTED;
    int64_t blocks = ((int64_t)B * H * N * (D / VEC) + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    if (blocks < 1) blocks = 1;
    if (split)
        hipLaunchKernelGGL((heads_kernel<T, true>), dim3((unsigned)blocks), dim3(256), 0, s, static_cast<const T*>(src),
                           static_cast<T*>(dst), B, N, H, d, D, sB, sH, sN);
    else
        hipLaunchKernelGGL((heads_kernel<T, false>), dim3((unsigned)blocks), dim3(256), 0, s, static_cast<const T*>(src),
                           static_cast<T*>(dst), B, N, H, d, D, sB, sH, sN);
    LORA_LAUNCH_CHECK();
    return LORA_OK;
}

}  // namespace

extern "C" int geglu_gate_fwd(const void* y, void* out, int64_t M, int C, int dtype, void* stream) {
    if (!y || !out || M < 0 || C < 1) return LORA_E_BADARG;
    if (M == 0) return LORA_OK;
    if (!aligned16(y) || !aligned16(out)) return LORA_E_ALIGN;
    hipStream_t s = static_cast<hipStream_t>(stream);
    switch (dtype) {
        case LORA_F32: return launch_geglu<float>(y, nullptr, out, M, C, false, s);
        case LORA_F16: return launch_geglu<half_t>(y, nullptr, out, M, C, false, s);
        case LORA_BF16: return launch_geglu<bf16_t>(y, nullptr, out, M, C, false, s);
        default: return LORA_E_BADARG;
    }
}

extern "C" int geglu_gate_bwd(const void* y, const void* dout, void* dy, int64_t M, int C, int dtype, void* stream) {
    if (!y || !dout || !dy || M < 0 || C < 1) return LORA_E_BADARG;
    if (M == 0) return LORA_OK;
    if (!aligned16(y) || !aligned16(dout) || !aligned16(dy)) return LORA_E_ALIGN;
    hipStream_t s = static_cast<hipStream_t>(stream);
    switch (dtype) {
        case LORA_F32: return launch_geglu<float>(y, dout, dy, M, C, true, s);
        case LORA_F16: return launch_geglu<half_t>(y, dout, dy, M, C, true, s);
        case LORA_BF16: return launch_geglu<bf16_t>(y, dout, dy, M, C, true, s);
        default: return LORA_E_BADARG;
    }
}

namespace {
int run_heads(const void* src, void* dst, int B, int N, int H, int d, int D, int64_t sB, int64_t sH, int64_t sN,
              bool split, int dtype, void* stream) {
    if (!src || !dst || B < 1 || N < 1 || H < 1 || d < 1) return LORA_E_BADARG;
    if (!aligned16(src) || !aligned16(dst)) return LORA_E_ALIGN;
    hipStream_t s = static_cast<hipStream_t>(stream);
    switch (dtype) {
        case LORA_F32: return launch_heads<float>(src, dst, B, N, H, d, D, sB, sH, sN, split, s);
        case LORA_F16: return launch_heads<half_t>(src, dst, B, N, H, d, D, sB, sH, sN, split, s);
        case LORA_BF16: return launch_heads<bf16_t>(src, dst, B, N, H, d, D, sB, sH, sN, split, s);
        default: return LORA_E_BADARG;
    }
}
}  // namespace

extern "C" int attn_split_heads(const void* src, void* dst, int B, int N, int H, int d, int D, int dtype,
                                void* stream) {
    return run_heads(src, dst, B, N, H, d, D, (int64_t)H * N * D, (int64_t)N * D, D, true, dtype, stream);
}

extern "C" int attn_merge_heads(const void* src, void* dst, int B, int N, int H, int d, int D, int dtype,
                                void* stream) {
    return run_heads(src, dst, B, N, H, d, D, (int64_t)H * N * D, (int64_t)N * D, D, false, dtype, stream);
}

extern "C" int attn_merge_heads_strided(const void* src, void* dst, int B, int N, int H, int d, int D, int64_t sB,
                                        int64_t sH, int64_t sN, int dtype, void* stream) {
    if (sB < 0 || sH < 0 || sN < 0) return LORA_E_BADARG;
    return run_heads(src, dst, B, N, H, d, D, sB, sH, sN, false, dtype, stream);
}

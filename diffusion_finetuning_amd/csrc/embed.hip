// Token-embedding rows of a text encoder whose INPUT EMBEDDINGS train next to the UNet's LoRA factors — the tuning phase of
// lora_diffusion/cli_lora_pti.py with continue_inversion (the default, :528): `text_encoder.get_input_embeddings().parameters()`
// joins the optimizer (:706-722) and loss_step runs `text_encoder(batch["input_ids"])[0]` inside the step (:199-206), so the
// step needs the table's forward gather and, in backward, the gradient rows of the tokens that occurred (BASELINE config 5:
// "+ extended-latent TI").  torch's embedding backward scatters with atomics (the sum order of a token that occurs several
// times — every padding position — varies from run to run); here one workgroup OWNS a token: the first position of a token
// sums all its positions in index order.  Deterministic, and the same on every data-parallel rank when the positions of all
// ranks are concatenated in rank order (trainer.TokenTable).
#include "common.h"

namespace {

template <typename T>
__global__ __launch_bounds__(256) void embed_rows_fwd_kernel(const float* __restrict__ table, const int64_t* __restrict__ ids,
                                                              T* __restrict__ out, int D, int64_t V) {
    const int64_t id = ids[blockIdx.x];
    T* dst = out + (int64_t)blockIdx.x * D;
    if (id < 0 || id >= V) {
        // An id outside the table (torch.nn.Embedding raises): the row is POISONED with NaN — the loss of the step is NaN, the
        // optimizer's overflow check skips the update and the trainer warns — instead of silently training on another token's
        // row; the backward kernel drops the same positions.  The binding layer validates ids on the host whenever it can.
        const T nan = from_f32<T>(__builtin_nanf(""));
        for (int c = threadIdx.x; c < D; c += 256) dst[c] = nan;
        return;  // (block-uniform)
    }
    const float4* src = reinterpret_cast<const float4*>(table + id * D);
    for (int c = threadIdx.x; c < D / 4; c += 256) {  // (D % 4 == 0: checked by the entry point)
        const float4 v = src[c];
        dst[4 * c + 0] = from_f32<T>(v.x);
        dst[4 * c + 1] = from_f32<T>(v.y);
        dst[4 * c + 2] = from_f32<T>(v.z);
        dst[4 * c + 3] = from_f32<T>(v.w);
    }
}

// grid = positions.  Block p exits unless p is the FIRST position of its token; the owner adds the rows of every position
// of that token in ascending position order (fp32) and stores (or accumulates onto) the table-gradient row.
template <typename T>
__global__ __launch_bounds__(256) void embed_rows_bwd_kernel(const T* __restrict__ dE, const int64_t* __restrict__ ids,
                                                              float* __restrict__ grad, unsigned char* __restrict__ active,
                                                              int64_t n, int D, int64_t V, int accumulate) {
    const int64_t p = blockIdx.x;
    const int64_t id = ids[p];
    if (id < 0 || id >= V) return;  // (block-uniform)
    int earlier = 0;
    for (int64_t q = threadIdx.x; q < p; q += 256) earlier |= ids[q] == id ? 1 : 0;
    if (__syncthreads_or(earlier)) return;
    if (active != nullptr && threadIdx.x == 0) active[id] = 1;  // the row has (had) a gradient: lora_adamw_rows gives it the full update
    for (int c = threadIdx.x; c < D; c += 256) {
        float acc = accumulate ? grad[id * D + c] : 0.f;
        for (int64_t q = p; q < n; ++q)
            if (ids[q] == id) acc += to_f32<T>(dE[q * D + c]);  // (ids[q]: a scalar load, the branch is block-uniform)
        grad[id * D + c] = acc;
    }
}

}  // namespace

extern "C" int embed_rows_fwd(const float* table, const int64_t* ids, void* out, int64_t n, int D, int64_t V, int out_dtype,
                              void* stream) {
    if (n < 0 || D < 1 || V < 1) return LORA_E_BADARG;
    if (n == 0) return LORA_OK;
    if (!table || !ids || !out) return LORA_E_BADARG;
    if (!aligned16(table) || (D % 4) != 0) return LORA_E_ALIGN;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const dim3 grid((unsigned)n);
    switch (out_dtype) {
        case LORA_F32: hipLaunchKernelGGL(embed_rows_fwd_kernel<float>, grid, dim3(256), 0, s, table, ids, static_cast<float*>(out), D, V); break;
        case LORA_F16: hipLaunchKernelGGL(embed_rows_fwd_kernel<half_t>, grid, dim3(256), 0, s, table, ids, static_cast<half_t*>(out), D, V); break;
        case LORA_BF16: hipLaunchKernelGGL(embed_rows_fwd_kernel<bf16_t>, grid, dim3(256), 0, s, table, ids, static_cast<bf16_t*>(out), D, V); break;
        default: return LORA_E_BADARG;
    }
    LORA_LAUNCH_CHECK();
    return LORA_OK;
}

extern "C" int embed_rows_bwd(const void* dE, const int64_t* ids, float* grad_table, unsigned char* active, int64_t n, int D,
                              int64_t V, int dtype, int accumulate, void* stream) {
    if (n < 0 || D < 1 || V < 1) return LORA_E_BADARG;
    if (n == 0) return LORA_OK;
    if (!dE || !ids || !grad_table) return LORA_E_BADARG;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const dim3 grid((unsigned)n);
    switch (dtype) {
        case LORA_F32: hipLaunchKernelGGL(embed_rows_bwd_kernel<float>, grid, dim3(256), 0, s, static_cast<const float*>(dE), ids, grad_table, active, n, D, V, accumulate); break;
        case LORA_F16: hipLaunchKernelGGL(embed_rows_bwd_kernel<half_t>, grid, dim3(256), 0, s, static_cast<const half_t*>(dE), ids, grad_table, active, n, D, V, accumulate); break;
        case LORA_BF16: hipLaunchKernelGGL(embed_rows_bwd_kernel<bf16_t>, grid, dim3(256), 0, s, static_cast<const bf16_t*>(dE), ids, grad_table, active, n, D, V, accumulate); break;
        default: return LORA_E_BADARG;
    }
    LORA_LAUNCH_CHECK();
    return LORA_OK;
}

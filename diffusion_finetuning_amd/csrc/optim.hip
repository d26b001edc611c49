// Fused gradient clipping + AdamW over the flat LoRA slab, weight merge and operand cache builders.
//   clip + AdamW : training_scripts/train_lora_dreambooth.py:878-888, lora_diffusion/cli_lora_pti.py:448-451
//                  (torch.nn.utils.clip_grad_norm_ and torch.optim.AdamW semantics, restated)
//   merge        : lora_diffusion/lora.py:410-424 (weight_apply_lora)
// The 288 LoRA tensors of an SD1.5 UNet live in ONE fp32 slab (parameters, gradients and both Adam
// moments each), so the whole optimizer step is two HBM-streaming launches instead of hundreds.
#include "common.h"

namespace {

constexpr int kMaxBlocks = 1024;

__device__ __forceinline__ float block_sum_256(float v, float* s_wave) {
    v = wave_sum(v);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    if (lane == 0) s_wave[w] = v;
    __syncthreads();
    float t = 0.f;
    if (threadIdx.x == 0) t = (s_wave[0] + s_wave[1]) + (s_wave[2] + s_wave[3]);
    __syncthreads();
    return t;
}

__global__ __launch_bounds__(256) void grad_sqnorm_kernel(const float* g, int64_t n, float mul, float* norm_out,
                                                          float* partials, float* flags, unsigned* ticket) {
    __shared__ float s_wave[4];
    __shared__ bool s_last;
    float local = 0.f, bad = 0.f;
    const int64_t nvec = n >> 2;
    const float4* g4 = reinterpret_cast<const float4*>(g);
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < nvec; i += (int64_t)gridDim.x * 256) {
        const float4 v = g4[i];
        const float a = v.x * mul, b = v.y * mul, c = v.z * mul, d = v.w * mul;
        local = fmaf(a, a, local);
        local = fmaf(b, b, local);
        local = fmaf(c, c, local);
        local = fmaf(d, d, local);
        if (!(isfinite(a) && isfinite(b) && isfinite(c) && isfinite(d))) bad = 1.f;
    }
    if (blockIdx.x == 0) {
        for (int64_t i = (nvec << 2) + threadIdx.x; i < n; i += 256) {
            const float a = g[i] * mul;
            local = fmaf(a, a, local);
            if (!isfinite(a)) bad = 1.f;
        }
    }
    const float bsum = block_sum_256(local, s_wave);
    const float bbad = block_sum_256(bad, s_wave);
    if (threadIdx.x == 0) {
        partials[blockIdx.x] = bsum;
        flags[blockIdx.x] = bbad;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const unsigned t = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const bool last = (t == gridDim.x - 1);
        if (last) {
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        s_last = last;
    }
    __syncthreads();
    if (s_last) {
        float t = 0.f, f = 0.f;
        for (int i = threadIdx.x; i < (int)gridDim.x; i += 256) {
            t += partials[i];
            f += flags[i];
        }
        t = block_sum_256(t, s_wave);
        f = block_sum_256(f, s_wave);
        if (threadIdx.x == 0) {
            const bool overflow = f != 0.f || !isfinite(t);
            norm_out[0] = t;
            norm_out[1] = overflow ? 1.f : 0.f;
            // applied-step counter for the bias corrections (exact in fp32 up to 2^24 steps): a skipped step does not
            // count, exactly as GradScaler does not call optimizer.step() on overflow; [3] counts the skipped ones
            if (overflow) norm_out[3] += 1.f;
            else norm_out[2] += 1.f;
        }
    }
}

struct AdamParams {
    float* p;
    const float* g;
    float* m;
    float* v;
    int64_t n;
    const float* norm_in;
    float grad_mul, max_norm, lr, beta1, beta2, eps, wd;
    float bc1, bc2_sqrt;  // 1-β1^t, sqrt(1-β2^t); bc1 <= 0: compute them from the device step counter norm_in[2]
};

// torch.optim.AdamW (single-tensor path) per element, in this order:
//   p *= 1 - lr·wd ; m = lerp(m, g, 1-β1) ; v = β2·v + (1-β2)·g² ;
//   denom = sqrt(v)/sqrt(1-β2^t) + eps ; p -= (lr/(1-β1^t)) · m/denom
__global__ __launch_bounds__(256) void adamw_kernel(AdamParams a) {
    float clip = 1.f;
    if (a.norm_in) {
        if (a.norm_in[1] != 0.f) return;  // overflow: skip the step (GradScaler semantics)
        if (a.max_norm > 0.f) {
            const float c = a.max_norm / (sqrtf(a.norm_in[0]) + 1e-6f);
            clip = c < 1.f ? c : 1.f;
        }
    }
    float bc1 = a.bc1, bc2_sqrt = a.bc2_sqrt;
    if (bc1 <= 0.f) {  // device-resident step count (wave-uniform branch): same double arithmetic as the host path
        const double t = (double)a.norm_in[2];
        bc1 = (float)(1.0 - pow((double)a.beta1, t));
        bc2_sqrt = (float)sqrt(1.0 - pow((double)a.beta2, t));
    }
    const float gm = a.grad_mul * clip;
    const float step_size = a.lr / bc1;
    const float decay = 1.f - a.lr * a.wd;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < a.n; i += (int64_t)gridDim.x * 256) {
        const float g = a.g[i] * gm;
        float p = a.p[i] * decay;
        float m = a.m[i];
        m = m + (g - m) * (1.f - a.beta1);
        const float v = a.beta2 * a.v[i] + (1.f - a.beta2) * g * g;
        const float denom = sqrtf(v) / bc2_sqrt + a.eps;
        p = p - step_size * (m / denom);
        a.p[i] = p;
        a.m[i] = m;
        a.v[i] = v;
    }
}

// The same update for a ROW-structured parameter of which only a few rows ever receive a gradient — the token-embedding table of
// cli_lora_pti.py's continue_inversion (:706-722; 49408 × 1024, a caption touches a few dozen rows).  torch's dense AdamW still
// visits every element each step: decoupled weight decay on all rows, moment decay on the rows that were touched before.  For a
// row that has NEVER received a gradient g = m = v = 0, and the dense update reduces — bit for bit — to p ← p·(1 − lr·wd):
// m and v stay 0 and the Adam term is lr/bc1 · 0/(0 + eps) = 0.  `active[row]` (set by embed_rows_bwd for every token that
// occurred, never cleared) tells the two kinds of rows apart, so an untouched row costs one read and one write of p instead of
// four reads and three writes: 0.45 → 0.1 ms per step on the 202-MB table, with the result of the dense kernel.
__global__ __launch_bounds__(256) void adamw_rows_kernel(AdamParams a, const unsigned char* __restrict__ active, int64_t V, int D) {
    float clip = 1.f;
    if (a.norm_in) {
        if (a.norm_in[1] != 0.f) return;  // overflow: skip the step (GradScaler semantics)
        if (a.max_norm > 0.f) {
            const float c = a.max_norm / (sqrtf(a.norm_in[0]) + 1e-6f);
            clip = c < 1.f ? c : 1.f;
        }
    }
    float bc1 = a.bc1, bc2_sqrt = a.bc2_sqrt;
    if (bc1 <= 0.f) {
        const double t = (double)a.norm_in[2];
        bc1 = (float)(1.0 - pow((double)a.beta1, t));
        bc2_sqrt = (float)sqrt(1.0 - pow((double)a.beta2, t));
    }
    const float gm = a.grad_mul * clip;
    const float step_size = a.lr / bc1;
    const float decay = 1.f - a.lr * a.wd;
    for (int64_t row = blockIdx.x; row < V; row += gridDim.x) {
        const int64_t base = row * D;
        if (active[row] == 0) {  // (block-uniform)
            if ((D & 3) == 0) {  // rows of whole 16-byte chunks (the slab's tail is 16-byte aligned): one load and one store per lane
                float4* pr = reinterpret_cast<float4*>(a.p + base);
                for (int c = threadIdx.x; c < D / 4; c += 256) {
                    float4 q = pr[c];
                    q.x *= decay; q.y *= decay; q.z *= decay; q.w *= decay;
                    pr[c] = q;
                }
            } else {
                for (int c = threadIdx.x; c < D; c += 256) a.p[base + c] = a.p[base + c] * decay;
            }
            continue;
        }
        for (int c = threadIdx.x; c < D; c += 256) {
            const int64_t i = base + c;
            const float g = a.g[i] * gm;
            float p = a.p[i] * decay;
            float m = a.m[i];
            m = m + (g - m) * (1.f - a.beta1);
            const float v = a.beta2 * a.v[i] + (1.f - a.beta2) * g * g;
            const float denom = sqrtf(v) / bc2_sqrt + a.eps;
            p = p - step_size * (m / denom);
            a.p[i] = p;
            a.m[i] = m;
            a.v[i] = v;
        }
    }
}

// weight_apply_lora, op-by-op rounding as the reference: (B@A) in fp32 → .type(W.dtype) → ·α → + W.
template <typename T>
__global__ __launch_bounds__(256) void merge_kernel(T* W, const float* A, const float* B, int K, int N, int r,
                                                    float alpha, int factor_dtype) {
    const int64_t total = (int64_t)N * K;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int n = (int)(i / K), k = (int)(i - (int64_t)n * K);
        float d = 0.f;
        for (int j = 0; j < r; ++j) d = fmaf(B[(int64_t)n * r + j], A[(int64_t)j * K + k], d);
        if (factor_dtype == LORA_F16) d = to_f32<half_t>(from_f32<half_t>(d));
        if (factor_dtype == LORA_BF16) d = to_f32<bf16_t>(from_f32<bf16_t>(d));
        const T dt = from_f32<T>(d);
        const T upd = from_f32<T>(alpha * to_f32<T>(dt));
        W[i] = from_f32<T>(to_f32<T>(W[i]) + to_f32<T>(upd));
    }
}

// Every layer of a model in ONE launch: blockIdx.y = layer, table[l] = {W, A, B, K, N, r, dtype, factor_dtype}
// (device pointers as int64).  Same arithmetic as merge_kernel.
template <typename T>
__device__ __forceinline__ void merge_rows(T* W, const float* A, const float* B, int K, int N, int r, float alpha,
                                           int factor_dtype) {
    const int64_t total = (int64_t)N * K;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int n = (int)(i / K), k = (int)(i - (int64_t)n * K);
        float d = 0.f;
        for (int j = 0; j < r; ++j) d = fmaf(B[(int64_t)n * r + j], A[(int64_t)j * K + k], d);
        if (factor_dtype == LORA_F16) d = to_f32<half_t>(from_f32<half_t>(d));
        if (factor_dtype == LORA_BF16) d = to_f32<bf16_t>(from_f32<bf16_t>(d));
        const T dt = from_f32<T>(d);
        const T upd = from_f32<T>(alpha * to_f32<T>(dt));
        W[i] = from_f32<T>(to_f32<T>(W[i]) + to_f32<T>(upd));
    }
}
__global__ __launch_bounds__(256) void merge_batched_kernel(const int64_t* table, float alpha) {
    const int64_t* e = table + (int64_t)blockIdx.y * 8;
    void* W = reinterpret_cast<void*>(e[0]);
    const float* A = reinterpret_cast<const float*>(e[1]);
    const float* B = reinterpret_cast<const float*>(e[2]);
    const int K = (int)e[3], N = (int)e[4], r = (int)e[5], fd = (int)e[7];
    switch ((int)e[6]) {
        case LORA_F32: merge_rows<float>(static_cast<float*>(W), A, B, K, N, r, alpha, fd); break;
        case LORA_F16: merge_rows<half_t>(static_cast<half_t*>(W), A, B, K, N, r, alpha, fd); break;
        case LORA_BF16: merge_rows<bf16_t>(static_cast<bf16_t*>(W), A, B, K, N, r, alpha, fd); break;
        default: break;
    }
}

// LoRA (+) LoRA interpolation of lora_diffusion/cli_lora_add.py:52-55, op by op in the tensors' own dtype:
//   x1 = T( T(a·x1) + T(b·x2) )      a = alpha, b = 1 - alpha
template <typename T>
__global__ __launch_bounds__(256) void lerp_kernel(T* x1, const T* x2, int64_t n, float a, float b) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        // separately rounded products and sum, as three torch ops give them (no fma contraction for fp32 tensors)
        const T p = from_f32<T>(__fmul_rn(a, to_f32<T>(x1[i])));
        const T q = from_f32<T>(__fmul_rn(b, to_f32<T>(x2[i])));
        x1[i] = from_f32<T>(__fadd_rn(to_f32<T>(p), to_f32<T>(q)));
    }
}

template <typename S, typename D>
__global__ __launch_bounds__(256) void cast_matrix_kernel(const S* src, D* dst, int64_t rows, int64_t cols,
                                                          int transpose) {
    __shared__ float tile[32][33];
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
    const int64_t c0 = (int64_t)blockIdx.x * 32, r0 = (int64_t)blockIdx.y * 32;
#pragma unroll
    for (int i = 0; i < 32; i += 8) {
        const int64_t r = r0 + ty + i, c = c0 + tx;
        if (r < rows && c < cols) tile[ty + i][tx] = to_f32<S>(src[r * cols + c]);
    }
    __syncthreads();
    if (transpose) {
#pragma unroll
        for (int i = 0; i < 32; i += 8) {
            const int64_t c = c0 + ty + i, r = r0 + tx;  // dst[c, r]
            if (r < rows && c < cols) dst[c * rows + r] = from_f32<D>(tile[tx][ty + i]);
        }
    } else {
#pragma unroll
        for (int i = 0; i < 32; i += 8) {
            const int64_t r = r0 + ty + i, c = c0 + tx;
            if (r < rows && c < cols) dst[r * cols + c] = from_f32<D>(tile[ty + i][tx]);
        }
    }
}

template <typename S>
int cast_dispatch_dst(const void* src, void* dst, int64_t rows, int64_t cols, int dst_dtype, int transpose,
                      hipStream_t s) {
    dim3 grid((unsigned)((cols + 31) / 32), (unsigned)((rows + 31) / 32));
    switch (dst_dtype) {
        case LORA_F32:
            hipLaunchKernelGGL((cast_matrix_kernel<S, float>), grid, dim3(256), 0, s, static_cast<const S*>(src),
                               static_cast<float*>(dst), rows, cols, transpose);
            break;
        case LORA_F16:
            hipLaunchKernelGGL((cast_matrix_kernel<S, half_t>), grid, dim3(256), 0, s, static_cast<const S*>(src),
                               static_cast<half_t*>(dst), rows, cols, transpose);
            break;
        case LORA_BF16:
            hipLaunchKernelGGL((cast_matrix_kernel<S, bf16_t>), grid, dim3(256), 0, s, static_cast<const S*>(src),
                               static_cast<bf16_t*>(dst), rows, cols, transpose);
            break;
        default: return LORA_E_BADARG;
    }
    LORA_LAUNCH_CHECK();
    return LORA_OK;
}

}  // namespace

extern "C" int64_t lora_sqnorm_workspace_bytes(void) { return 16 + 2 * kMaxBlocks * 4; }

extern "C" int lora_grad_sqnorm(const float* grad, int64_t n, float grad_mul, float* norm_out, void* workspace,
                                void* stream) {
    if (!grad || !norm_out || !workspace || n < 1) return LORA_E_BADARG;
    if (!aligned16(workspace) || !aligned16(grad)) return LORA_E_ALIGN;
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (!lora_zero_ticket(workspace, s)) return LORA_E_LAUNCH;
    int64_t blocks = (n / 4 + 255) / 256;
    if (blocks > kMaxBlocks) blocks = kMaxBlocks;
    if (blocks < 1) blocks = 1;
    char* ws = static_cast<char*>(workspace);
    hipLaunchKernelGGL(grad_sqnorm_kernel, dim3((unsigned)blocks), dim3(256), 0, s, grad, n, grad_mul, norm_out,
                       reinterpret_cast<float*>(ws + 16), reinterpret_cast<float*>(ws + 16 + kMaxBlocks * 4),
                       reinterpret_cast<unsigned*>(ws));
    LORA_LAUNCH_CHECK();
    return LORA_OK;
}

extern "C" int lora_adamw_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t n,
                               const float* norm_in, float grad_mul, float max_norm, float lr, float beta1,
                               float beta2, float eps, float weight_decay, int step, void* stream) {
    if (!param || !grad || !exp_avg || !exp_avg_sq || n < 1 || step < 0) return LORA_E_BADARG;
    if (step == 0 && !norm_in) return LORA_E_BADARG;  // step 0 = "use the device counter norm_in[2]"
    AdamParams a{};
    a.p = param; a.g = grad; a.m = exp_avg; a.v = exp_avg_sq; a.n = n; a.norm_in = norm_in;
    a.grad_mul = grad_mul; a.max_norm = max_norm; a.lr = lr; a.beta1 = beta1; a.beta2 = beta2;
    a.eps = eps; a.wd = weight_decay;
    // bias corrections in double on the host, as torch does with python floats
    a.bc1 = step > 0 ? (float)(1.0 - pow((double)beta1, (double)step)) : 0.f;
    a.bc2_sqrt = step > 0 ? (float)sqrt(1.0 - pow((double)beta2, (double)step)) : 0.f;
    int64_t blocks = (n + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(adamw_kernel, dim3((unsigned)blocks), dim3(256), 0, static_cast<hipStream_t>(stream), a);
    LORA_LAUNCH_CHECK();
    return LORA_OK;
}

extern "C" int lora_adamw_rows(float* param, const float* grad, float* exp_avg, float* exp_avg_sq,
                               const unsigned char* active, int64_t V, int D, const float* norm_in, float grad_mul,
                               float max_norm, float lr, float beta1, float beta2, float eps, float weight_decay, int step,
                               void* stream) {
    if (!param || !grad || !exp_avg || !exp_avg_sq || !active || V < 1 || D < 1 || step < 0) return LORA_E_BADARG;
    if (step == 0 && !norm_in) return LORA_E_BADARG;
    AdamParams a{};
    a.p = param; a.g = grad; a.m = exp_avg; a.v = exp_avg_sq; a.n = V * D; a.norm_in = norm_in;
    a.grad_mul = grad_mul; a.max_norm = max_norm; a.lr = lr; a.beta1 = beta1; a.beta2 = beta2;
    a.eps = eps; a.wd = weight_decay;
    a.bc1 = step > 0 ? (float)(1.0 - pow((double)beta1, (double)step)) : 0.f;
    a.bc2_sqrt = step > 0 ? (float)sqrt(1.0 - pow((double)beta2, (double)step)) : 0.f;
    const int64_t blocks = V < 4096 ? V : 4096;
    hipLaunchKernelGGL(adamw_rows_kernel, dim3((unsigned)blocks), dim3(256), 0, static_cast<hipStream_t>(stream), a, active, V, D);
    LORA_LAUNCH_CHECK();
    return LORA_OK;
}

extern "C" int lora_merge_weight(void* W, const float* A, const float* B, int K, int N, int r, float alpha,
                                 int dtype, int factor_dtype, void* stream) {
    if (!W || !A || !B || K < 1 || N < 1) return LORA_E_BADARG;
    if (r < 1 || r > (K < N ? K : N)) return LORA_E_RANK;
    int64_t blocks = ((int64_t)N * K + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipStream_t s = static_cast<hipStream_t>(stream);
    switch (dtype) {
        case LORA_F32:
            hipLaunchKernelGGL(merge_kernel<float>, dim3((unsigned)blocks), dim3(256), 0, s, static_cast<float*>(W), A,
                               B, K, N, r, alpha, factor_dtype);
            break;
        case LORA_F16:
            hipLaunchKernelGGL(merge_kernel<half_t>, dim3((unsigned)blocks), dim3(256), 0, s, static_cast<half_t*>(W),
                               A, B, K, N, r, alpha, factor_dtype);
            break;
        case LORA_BF16:
            hipLaunchKernelGGL(merge_kernel<bf16_t>, dim3((unsigned)blocks), dim3(256), 0, s, static_cast<bf16_t*>(W),
                               A, B, K, N, r, alpha, factor_dtype);
            break;
        default: return LORA_E_BADARG;
    }
    LORA_LAUNCH_CHECK();
    return LORA_OK;
}

extern "C" int lora_merge_weight_batched(const int64_t* table, int n_layers, int64_t max_elems, float alpha,
                                         void* stream) {
    if (!table || n_layers < 1 || max_elems < 1) return LORA_E_BADARG;
    int64_t blocks = (max_elems + 1023) / 1024;  // ~4 elements per thread on the largest layer
    if (blocks > 1024) blocks = 1024;
    hipLaunchKernelGGL(merge_batched_kernel, dim3((unsigned)blocks, (unsigned)n_layers), dim3(256), 0,
                       static_cast<hipStream_t>(stream), table, alpha);
    LORA_LAUNCH_CHECK();
    return LORA_OK;
}

extern "C" int lora_lerp(void* x1, const void* x2, int64_t n, float a, float b, int dtype, void* stream) {
    if (!x1 || !x2 || n < 1) return LORA_E_BADARG;
    int64_t blocks = (n + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    hipStream_t s = static_cast<hipStream_t>(stream);
    switch (dtype) {
        case LORA_F32:
            hipLaunchKernelGGL(lerp_kernel<float>, dim3((unsigned)blocks), dim3(256), 0, s, static_cast<float*>(x1),
                               static_cast<const float*>(x2), n, a, b);
            break;
        case LORA_F16:
            hipLaunchKernelGGL(lerp_kernel<half_t>, dim3((unsigned)blocks), dim3(256), 0, s, static_cast<half_t*>(x1),
                               static_cast<const half_t*>(x2), n, a, b);
            break;
        case LORA_BF16:
            hipLaunchKernelGGL(lerp_kernel<bf16_t>, dim3((unsigned)blocks), dim3(256), 0, s, static_cast<bf16_t*>(x1),
                               static_cast<const bf16_t*>(x2), n, a, b);
            break;
        default: return LORA_E_BADARG;
    }
    LORA_LAUNCH_CHECK();
    return LORA_OK;
}

extern "C" int lora_cast_matrix(const void* src, void* dst, int64_t rows, int64_t cols, int src_dtype,
                                int dst_dtype, int transpose, void* stream) {
    if (!src || !dst || rows < 1 || cols < 1) return LORA_E_BADARG;
    hipStream_t s = static_cast<hipStream_t>(stream);
    switch (src_dtype) {
        case LORA_F32: return cast_dispatch_dst<float>(src, dst, rows, cols, dst_dtype, transpose, s);
        case LORA_F16: return cast_dispatch_dst<half_t>(src, dst, rows, cols, dst_dtype, transpose, s);
        case LORA_BF16: return cast_dispatch_dst<bf16_t>(src, dst, rows, cols, dst_dtype, transpose, s);
        default: return LORA_E_BADARG;
    }
}

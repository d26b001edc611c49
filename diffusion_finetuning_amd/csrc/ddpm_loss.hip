// DDPM noise-prediction loss (forward value + gradient in one pass), the PTI mask preparation and
// the add_noise / target prologue.
//   loss      : training_scripts/train_lora_dreambooth.py:855-875, lora_diffusion/cli_lora_pti.py:243-247
//   mask prep : lora_diffusion/cli_lora_pti.py:222-241
//   prologue  : training_scripts/train_lora_dreambooth.py:824-853 (DDPM add_noise / get_velocity)
// All HBM-bound elementwise/reduction work: 16-byte vector loads, wave-shuffle → LDS → one partial
// per workgroup, and a deterministic "last workgroup sums the partials in index order" finish
// (agent-scope release/acquire around an arrival ticket, so the result does not depend on which
// XCD a workgroup ran on).
#include "common.h"

namespace {

constexpr int kMaxBlocks = 1024;
constexpr int64_t kWsBytes = 16 + kMaxBlocks * 4;

// Block-level sum in a fixed order; result valid in thread 0.
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

// Publishes this workgroup's partial, returns true in the workgroup that arrived last, after an
// agent-scope acquire (so plain loads of every partial are fresh for all its waves).
__device__ __forceinline__ bool publish_partial_and_check_last(float partial, float* partials,
                                                               unsigned* ticket, bool* s_last) {
    if (threadIdx.x == 0) {
        partials[blockIdx.x] = partial;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const unsigned t = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const bool last = (t == gridDim.x - 1);
        if (last) {
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        *s_last = last;
    }
    __syncthreads();
    return *s_last;
}

struct MseParams {
    const void* pred;
    const void* target;
    const float* mask;
    void* dpred;
    float* loss_out;
    float* partials;
    unsigned* ticket;
    int64_t n_total;   // rows * per_row
    int64_t per_row;
    int64_t hw;
    int64_t inst_elems;  // n_inst * per_row
    float coef_inst;     // 1 / (n_inst * per_row)
    float coef_prior;    // prior_weight / (n_prior * per_row)
    float grad_scale;
};

template <typename T, int VEC>
__global__ __launch_bounds__(256) void ddpm_mse_kernel(MseParams p) {
    __shared__ float s_wave[4];
    __shared__ bool s_last;
    const T* pred = static_cast<const T*>(p.pred);
    const T* target = static_cast<const T*>(p.target);
    T* dpred = static_cast<T*>(p.dpred);
    float local = 0.f;
    const int64_t nvec = p.n_total / VEC;
    for (int64_t v = (int64_t)blockIdx.x * 256 + threadIdx.x; v < nvec; v += (int64_t)gridDim.x * 256) {
        const int64_t i0 = v * VEC;
        T pv[VEC], tv[VEC], gv[VEC];
        if constexpr (VEC > 1) {
            *reinterpret_cast<Chunk<T>*>(pv) = *reinterpret_cast<const Chunk<T>*>(pred + i0);
            *reinterpret_cast<Chunk<T>*>(tv) = *reinterpret_cast<const Chunk<T>*>(target + i0);
        } else {
            pv[0] = pred[i0];
            tv[0] = target[i0];
        }
        // a vector never straddles a row: per_row % VEC == 0 on this path
        const float coef = i0 < p.inst_elems ? p.coef_inst : p.coef_prior;
        const int64_t row = i0 / p.per_row;
        const int64_t in_row = i0 - row * p.per_row;
#pragma unroll
        for (int e = 0; e < VEC; ++e) {
            float mk = 1.f;
            if (p.mask) mk = p.mask[row * p.hw + (in_row + e) % p.hw];
            // pred·mask and target·mask promote to fp32 in the reference (fp32 mask tensor)
            const float a = to_f32<T>(pv[e]) * mk;
            const float b = to_f32<T>(tv[e]) * mk;
            const float d = a - b;
            local = fmaf(coef * d, d, local);
            gv[e] = from_f32<T>(p.grad_scale * 2.f * coef * d * mk);
        }
        if (dpred) {
            if constexpr (VEC > 1) {
                *reinterpret_cast<Chunk<T>*>(dpred + i0) = *reinterpret_cast<const Chunk<T>*>(gv);
            } else {
                dpred[i0] = gv[0];
            }
        }
    }
    const float bsum = block_sum_256(local, s_wave);
    if (publish_partial_and_check_last(bsum, p.partials, p.ticket, &s_last)) {
        float t = 0.f;
        for (int i = threadIdx.x; i < (int)gridDim.x; i += 256) t += p.partials[i];
        t = block_sum_256(t, s_wave);
        if (threadIdx.x == 0) p.loss_out[0] = t;
    }
}

__global__ __launch_bounds__(1024) void mask_prepare_kernel(const float* in, float* out, int B, int Hin,
                                                            int Win, int H, int W) {
    __shared__ float s_part[16];
    __shared__ float s_mean;
    const int64_t n = (int64_t)B * H * W;
    const float sh = (float)Hin / (float)H, sw = (float)Win / (float)W;
    float local = 0.f;
    for (int64_t i = threadIdx.x; i < n; i += 1024) {
        const int x = (int)(i % W);
        const int y = (int)((i / W) % H);
        const int b = (int)(i / ((int64_t)W * H));
        int sy = (int)floorf(y * sh), sx = (int)floorf(x * sw);
        if (sy > Hin - 1) sy = Hin - 1;
        if (sx > Win - 1) sx = Win - 1;
        const float v = in[((int64_t)b * Hin + sy) * Win + sx] + 0.05f;
        out[i] = v;
        local += v;
    }
    local = wave_sum(local);
    if ((threadIdx.x & 63) == 0) s_part[threadIdx.x >> 6] = local;
    __syncthreads();
    if (threadIdx.x == 0) {
        float t = 0.f;
        for (int i = 0; i < 16; ++i) t += s_part[i];
        s_mean = t / (float)n;
    }
    __syncthreads();
    const float mean = s_mean;
    for (int64_t i = threadIdx.x; i < n; i += 1024) out[i] = out[i] / mean;
}

template <typename T>
__global__ __launch_bounds__(256) void add_noise_kernel(const float* x0, const float* eps, const int64_t* t,
                                                        const float* sa, const float* sb, T* noisy, T* target,
                                                        int64_t per_row, int64_t n_total, int v_pred) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n_total; i += (int64_t)gridDim.x * 256) {
        const int64_t b = i / per_row;
        const int64_t ti = t[b];
        const float a = sa[ti], s = sb[ti];
        const float x = x0[i], e = eps[i];
        noisy[i] = from_f32<T>(a * x + s * e);
        if (target) target[i] = from_f32<T>(v_pred ? a * e - s * x : e);
    }
}

template <typename T>
int launch_mse(MseParams p, hipStream_t s) {
    constexpr int V = ElemTraits<T>::kVec;
    const bool vec = (p.per_row % V) == 0 && aligned16(p.pred) && aligned16(p.target) &&
                     (!p.dpred || aligned16(p.dpred));
    const int64_t work = vec ? p.n_total / V : p.n_total;
    int64_t blocks = (work + 255) / 256;
    if (blocks > kMaxBlocks) blocks = kMaxBlocks;
    if (blocks < 1) blocks = 1;
    if (vec)
        LORA_LAUNCH(PK_MSE, (ddpm_mse_kernel<T, V>), dim3((unsigned)blocks), dim3(256), 0, s, p);
    else
        LORA_LAUNCH(PK_MSE, (ddpm_mse_kernel<T, 1>), dim3((unsigned)blocks), dim3(256), 0, s, p);
    LORA_LAUNCH_CHECK();
    return LORA_OK;
}

}  // namespace

extern "C" int64_t lora_mse_workspace_bytes(void) { return kWsBytes; }

extern "C" int ddpm_mse_fwd_bwd(const void* pred, const void* target, const float* mask, int n_inst,
                                int n_prior, int64_t per_row, int64_t hw, float prior_weight, float grad_scale,
                                float* loss_out, void* dpred, void* workspace, int dtype, void* stream) {
    if (!pred || !target || !loss_out || !workspace) return LORA_E_BADARG;
    if (n_inst < 1 || n_prior < 0 || per_row < 1 || hw < 1 || (per_row % hw) != 0) return LORA_E_BADARG;
    if (!aligned16(workspace)) return LORA_E_ALIGN;
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (!lora_zero_ticket(workspace, s)) return LORA_E_LAUNCH;
    MseParams p{};
    p.pred = pred; p.target = target; p.mask = mask; p.dpred = dpred; p.loss_out = loss_out;
    p.ticket = static_cast<unsigned*>(workspace);
    p.partials = reinterpret_cast<float*>(static_cast<char*>(workspace) + 16);
    p.n_total = (int64_t)(n_inst + n_prior) * per_row;
    p.per_row = per_row; p.hw = hw;
    p.inst_elems = (int64_t)n_inst * per_row;
    p.coef_inst = 1.0f / ((float)n_inst * (float)per_row);
    p.coef_prior = n_prior > 0 ? prior_weight / ((float)n_prior * (float)per_row) : 0.f;
    p.grad_scale = grad_scale;
    const double e = dtype == LORA_F32 ? 4.0 : 2.0;
    ProfWork work(e * (double)p.n_total * (dpred ? 3.0 : 2.0), 4.0 * (double)p.n_total);
    int rc;
    switch (dtype) {
        case LORA_F32: rc = launch_mse<float>(p, s); break;
        case LORA_F16: rc = launch_mse<half_t>(p, s); break;
        case LORA_BF16: rc = launch_mse<bf16_t>(p, s); break;
        default: rc = LORA_E_BADARG;
    }
    return rc;
}

extern "C" int lora_mask_prepare(const float* mask_in, float* mask_out, int B, int Hin, int Win, int H, int W,
                                 void* stream) {
    if (!mask_in || !mask_out || B < 1 || Hin < 1 || Win < 1 || H < 1 || W < 1) return LORA_E_BADARG;
    hipLaunchKernelGGL(mask_prepare_kernel, dim3(1), dim3(1024), 0, static_cast<hipStream_t>(stream), mask_in,
                       mask_out, B, Hin, Win, H, W);
    LORA_LAUNCH_CHECK();
    return LORA_OK;
}

extern "C" int ddpm_add_noise(const float* x0, const float* eps, const int64_t* t, const float* sqrt_acp,
                              const float* sqrt_1macp, void* noisy, void* target, int B, int64_t per_row,
                              int v_prediction, int dtype, void* stream) {
    if (!x0 || !eps || !t || !sqrt_acp || !sqrt_1macp || !noisy || B < 1 || per_row < 1) return LORA_E_BADARG;
    const int64_t n = (int64_t)B * per_row;
    int64_t blocks = (n + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    hipStream_t s = static_cast<hipStream_t>(stream);
    switch (dtype) {
        case LORA_F32:
            hipLaunchKernelGGL(add_noise_kernel<float>, dim3((unsigned)blocks), dim3(256), 0, s, x0, eps, t, sqrt_acp,
                               sqrt_1macp, static_cast<float*>(noisy), static_cast<float*>(target), per_row, n,
                               v_prediction);
            break;
        case LORA_F16:
            hipLaunchKernelGGL(add_noise_kernel<half_t>, dim3((unsigned)blocks), dim3(256), 0, s, x0, eps, t, sqrt_acp,
                               sqrt_1macp, static_cast<half_t*>(noisy), static_cast<half_t*>(target), per_row, n,
                               v_prediction);
            break;
        case LORA_BF16:
            hipLaunchKernelGGL(add_noise_kernel<bf16_t>, dim3((unsigned)blocks), dim3(256), 0, s, x0, eps, t, sqrt_acp,
                               sqrt_1macp, static_cast<bf16_t*>(noisy), static_cast<bf16_t*>(target), per_row, n,
                               v_prediction);
            break;
        default: return LORA_E_BADARG;
    }
    LORA_LAUNCH_CHECK();
    return LORA_OK;
}

// ---------------------------------------------------------------------------------------------------------
// Step prologue with build-owned, counter-based randomness:
//   t_b  ~ U{0..T-1},  eps ~ N(0,1)  from Philox4x32-10 keyed by (seed, step) — the same stream on every rank
//   (set_seed semantics of train_lora_dreambooth.py:509-510) and on the CPU oracle (oracle/philox.py) —
//   then noisy = sqrt_acp[t]·x0 + sqrt_1macp[t]·eps and target = eps | velocity, in ONE elementwise launch that
//   replaces randn_like + randint + add_noise (+ get_velocity) of train_lora_dreambooth.py:824-853.
// Counter layout: element group g (4 consecutive elements) uses counter (g, 0, stream, 0) with stream 0 for
// eps and 1 for the timesteps (row b uses counter (b, 0, 1, 0), first word).  Normals: Box–Muller on the two
// word pairs, u = (x + 0.5)·2^-32 ∈ (0,1).
namespace {

__device__ __forceinline__ void philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0,
                                              uint32_t k1, uint32_t out[4]) {
#pragma unroll
    for (int i = 0; i < 10; ++i) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * c0;
        const uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
        const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
        const uint32_t n1 = (uint32_t)p1;
        const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
        const uint32_t n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

__device__ __forceinline__ float u01(uint32_t x) { return ((float)x + 0.5f) * 2.3283064365386963e-10f; }

template <typename T>
__global__ __launch_bounds__(256) void noise_prologue_kernel(const float* x0, const float* sa, const float* sb,
                                                             T* noisy, T* target, float* eps_out, int64_t* t_out,
                                                             int B, int64_t per_row, int n_timesteps, uint32_t seed,
                                                             uint32_t step, int v_pred) {
    const int64_t n_total = (int64_t)B * per_row;
    const int64_t groups = (n_total + 3) >> 2;
    for (int64_t g = (int64_t)blockIdx.x * 256 + threadIdx.x; g < groups; g += (int64_t)gridDim.x * 256) {
        uint32_t r[4];
        philox4x32_10((uint32_t)g, (uint32_t)(g >> 32), 0u, 0u, seed, step, r);
        float z[4];
        {
            const float r0 = sqrtf(-2.f * logf(u01(r[0]))), r1 = sqrtf(-2.f * logf(u01(r[2])));
            float s0, c0, s1, c1;
            sincosf(6.283185307179586f * u01(r[1]), &s0, &c0);
            sincosf(6.283185307179586f * u01(r[3]), &s1, &c1);
            z[0] = r0 * c0; z[1] = r0 * s0; z[2] = r1 * c1; z[3] = r1 * s1;
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int64_t i = g * 4 + e;
            if (i >= n_total) break;
            const int64_t b = i / per_row;
            uint32_t tr[4];
            philox4x32_10((uint32_t)b, 0u, 1u, 0u, seed, step, tr);
            const int64_t ti = (int64_t)(((uint64_t)tr[0] * (uint64_t)n_timesteps) >> 32);
            if (i == b * per_row && t_out) t_out[b] = ti;
            const float a = sa[ti], s = sb[ti];
            const float x = x0[i];
            noisy[i] = from_f32<T>(a * x + s * z[e]);
            if (target) target[i] = from_f32<T>(v_pred ? a * z[e] - s * x : z[e]);
            if (eps_out) eps_out[i] = z[e];
        }
    }
}

}  // namespace

extern "C" int ddpm_noise_prologue(const float* x0, const float* sqrt_acp, const float* sqrt_1macp, void* noisy,
                                   void* target, float* eps_out, int64_t* t_out, int B, int64_t per_row,
                                   int n_timesteps, uint64_t seed, uint64_t step, int v_prediction, int dtype,
                                   void* stream) {
    if (!x0 || !sqrt_acp || !sqrt_1macp || !noisy || B < 1 || per_row < 1 || n_timesteps < 1) return LORA_E_BADARG;
    const int64_t groups = ((int64_t)B * per_row + 3) / 4;
    int64_t blocks = (groups + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const uint32_t sd = (uint32_t)seed, st = (uint32_t)step;
    switch (dtype) {
        case LORA_F32:
            hipLaunchKernelGGL(noise_prologue_kernel<float>, dim3((unsigned)blocks), dim3(256), 0, s, x0, sqrt_acp,
                               sqrt_1macp, static_cast<float*>(noisy), static_cast<float*>(target), eps_out, t_out, B,
                               per_row, n_timesteps, sd, st, v_prediction);
            break;
        case LORA_F16:
            hipLaunchKernelGGL(noise_prologue_kernel<half_t>, dim3((unsigned)blocks), dim3(256), 0, s, x0, sqrt_acp,
                               sqrt_1macp, static_cast<half_t*>(noisy), static_cast<half_t*>(target), eps_out, t_out, B,
                               per_row, n_timesteps, sd, st, v_prediction);
            break;
        case LORA_BF16:
            hipLaunchKernelGGL(noise_prologue_kernel<bf16_t>, dim3((unsigned)blocks), dim3(256), 0, s, x0, sqrt_acp,
                               sqrt_1macp, static_cast<bf16_t*>(noisy), static_cast<bf16_t*>(target), eps_out, t_out, B,
                               per_row, n_timesteps, sd, st, v_prediction);
            break;
        default: return LORA_E_BADARG;
    }
    LORA_LAUNCH_CHECK();
    return LORA_OK;
}

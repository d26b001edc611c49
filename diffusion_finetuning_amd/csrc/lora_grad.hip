// Gradients of the rank-r factors (base W frozen): the two tall-skinny reductions over M
//     gB[N,r] += s·dYᵀ·T        gA[r,K] += s·Uᵀ·X          (T = X·Aᵀ, U = dY·B, both [M,r] fp32)
// — the autograd of lora_diffusion/lora.py:49-50 restricted to the parameters that
// lora.py:179-180 mark trainable.  Both are  G[c,j] += s·Σ_m S[m,c]·P[m,j]  with a streamed
// operand S ∈ {dY, X} read exactly once, so this is an HBM-streaming kernel:
//   - a thread owns one 16-byte column chunk (8 halfs / 4 floats) and walks rows; r×VEC fp32
//     accumulators stay in VGPRs; the P row (r floats) is a broadcast load;
//   - 256 threads cover ⌊256/CL⌋ rows per pass when the strip is narrower than the workgroup;
//   - partial sums of the row groups are combined with LDS float atomics, then written with
//     CONTIGUOUS global float atomics (256-B runs) into the flat gradient slab.
// Both problems of a layer go out in ONE launch (blockIdx.z selects dY→gB or X→gA).
#include "common.h"

namespace {

struct GradProblem {
    const void* S;    // [M, C]
    const float* P;   // [M, r]
    float* G;         // output
    int C;
    int out_kn;       // 1: G is [r, C] (gA layout: j*C + c); 0: G is [C, r] (gB layout: c*r + j)
    int CL;           // column chunks (threads) per row inside a strip
    int strips;
};
struct GradParams {
    GradProblem prob[2];
    int64_t M;
    int r;
    int rows_per_block;
    float scale;
};

template <typename T, int RP /* padded rank: 4, 8, 16 */>
__global__ __launch_bounds__(256) void lora_grad_kernel(GradParams p) {
    constexpr int VEC = ElemTraits<T>::kVec;
    extern __shared__ __attribute__((aligned(16))) float sred[];

    const GradProblem& q = p.prob[blockIdx.z];
    if ((int)blockIdx.x >= q.strips) return;
    const int tid = threadIdx.x;
    const int CL = q.CL;
    const int rows_pp = 256 / CL;  // rows per pass
    const int rsub = tid / CL;
    const int cg = tid - rsub * CL;
    const int c_local = cg * VEC;
    const int c0 = blockIdx.x * CL * VEC;
    const int stripW = min(CL * VEC, q.C - c0);
    const bool active = rsub < rows_pp && (c_local < stripW);

    float acc[RP][VEC];
#pragma unroll
    for (int j = 0; j < RP; ++j)
#pragma unroll
        for (int e = 0; e < VEC; ++e) acc[j][e] = 0.f;

    const int64_t m_begin = (int64_t)blockIdx.y * p.rows_per_block;
    int64_t m_end = m_begin + p.rows_per_block;
    if (m_end > p.M) m_end = p.M;

    if (active) {
        const T* S = static_cast<const T*>(q.S) + c0 + c_local;
        for (int64_t m = m_begin + rsub; m < m_end; m += rows_pp) {
            const Chunk<T> s = *reinterpret_cast<const Chunk<T>*>(S + m * q.C);
            float pv[RP];
#pragma unroll
            for (int j = 0; j < RP; ++j) pv[j] = j < p.r ? q.P[m * p.r + j] : 0.f;
            float sv[VEC];
#pragma unroll
            for (int e = 0; e < VEC; ++e) sv[e] = to_f32<T>(s.v[e]);
#pragma unroll
            for (int j = 0; j < RP; ++j)
#pragma unroll
                for (int e = 0; e < VEC; ++e) acc[j][e] = fmaf(sv[e], pv[j], acc[j][e]);
        }
    }

    // combine row groups: LDS image [j][c_local] (j-major), zero → ds_add → contiguous flush
    const int total = p.r * stripW;
    for (int i = tid; i < total; i += 256) sred[i] = 0.f;
    __syncthreads();
    if (active) {
#pragma unroll
        for (int j = 0; j < RP; ++j) {
            if (j < p.r) {
#pragma unroll
                for (int e = 0; e < VEC; ++e) atomicAdd(&sred[j * stripW + c_local + e], acc[j][e]);
            }
        }
    }
    __syncthreads();
    if (q.out_kn) {
        // gA[j, c0 + c]: every j row is a contiguous run of stripW floats
        for (int i = tid; i < total; i += 256) {
            const int j = i / stripW, c = i - j * stripW;
            atomicAdd(q.G + (int64_t)j * q.C + c0 + c, p.scale * sred[i]);
        }
    } else {
        // gB[(c0 + c), j]: the whole strip block is one contiguous run of stripW·r floats
        for (int i = tid; i < total; i += 256) {
            const int c = i / p.r, j = i - c * p.r;
            atomicAdd(q.G + (int64_t)c0 * p.r + i, p.scale * sred[j * stripW + c]);
        }
    }
}

// Unaligned / large-rank path: one thread per output element, serial over M.  Correct, not fast.
template <typename T>
__global__ void lora_grad_generic_kernel(GradParams p) {
    const GradProblem& q = p.prob[blockIdx.z];
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (int64_t)q.C * p.r) return;
    const int c = (int)(idx / p.r), j = (int)(idx % p.r);
    const T* S = static_cast<const T*>(q.S);
    float s = 0.f;
    for (int64_t m = 0; m < p.M; ++m) s = fmaf(to_f32<T>(S[m * q.C + c]), q.P[m * p.r + j], s);
    float* dst = q.out_kn ? q.G + (int64_t)j * q.C + c : q.G + idx;
    atomicAdd(dst, p.scale * s);
}

template <typename T>
int launch_grad(GradParams p, hipStream_t stream) {
    constexpr int VEC = ElemTraits<T>::kVec;
    bool fast = p.r <= 16;
    for (int i = 0; i < 2; ++i) fast = fast && (p.prob[i].C % VEC) == 0 && aligned16(p.prob[i].S);
    if (!fast) {
        int cmax = p.prob[0].C > p.prob[1].C ? p.prob[0].C : p.prob[1].C;
        const int64_t n = (int64_t)cmax * p.r;
        hipLaunchKernelGGL(lora_grad_generic_kernel<T>, dim3((unsigned)((n + 255) / 256), 1, 2), dim3(256), 0,
                           stream, p);
        LORA_LAUNCH_CHECK();
        return LORA_OK;
    }
    const int rp = p.r <= 4 ? 4 : (p.r <= 8 ? 8 : 16);
    // LDS image r·stripW floats ≤ 32 KiB  ⇒  CL ≤ 8192 / (VEC·rp)
    const int cl_cap = 8192 / (VEC * rp) < 256 ? 8192 / (VEC * rp) : 256;
    int max_strips = 1, max_lds = 0;
    for (int i = 0; i < 2; ++i) {
        GradProblem& q = p.prob[i];
        const int chunks = q.C / VEC;
        // balance strips: smallest strip count that respects the cap, then even widths
        const int strips = (chunks + cl_cap - 1) / cl_cap;
        q.CL = (chunks + strips - 1) / strips;
        q.strips = strips;
        if (strips > max_strips) max_strips = strips;
        const int lds = p.r * q.CL * VEC * 4;
        if (lds > max_lds) max_lds = lds;
    }
    // rows per block: aim at ~1024 workgroups over both problems, at least 32 rows each
    const int64_t want_blocks = 1024 / (2 * max_strips) > 0 ? 1024 / (2 * max_strips) : 1;
    int64_t rpb = (p.M + want_blocks - 1) / want_blocks;
    if (rpb < 32) rpb = 32;
    p.rows_per_block = (int)rpb;
    const unsigned gy = (unsigned)((p.M + rpb - 1) / rpb);
    dim3 grid(max_strips, gy, 2);
    switch (rp) {
        case 4: LORA_LAUNCH(PK_GRAD_R4, (lora_grad_kernel<T, 4>), grid, dim3(256), max_lds, stream, p); break;
        case 8: LORA_LAUNCH(PK_GRAD_R8, (lora_grad_kernel<T, 8>), grid, dim3(256), max_lds, stream, p); break;
        default: LORA_LAUNCH(PK_GRAD_R16, (lora_grad_kernel<T, 16>), grid, dim3(256), max_lds, stream, p); break;
    }
    LORA_LAUNCH_CHECK();
    return LORA_OK;
}

}  // namespace

extern "C" int lora_linear_bwd_params(const void* dY, const void* X, const float* T, const float* U,
                                      float* gA, float* gB, int64_t M, int K, int N, int r, float scale,
                                      int dtype, void* stream) {
    if (M < 0 || K <= 0 || N <= 0) return LORA_E_BADARG;
    if (r < 1 || r > (K < N ? K : N)) return LORA_E_RANK;
    if (M == 0) return LORA_OK;
    if (!dY || !X || !T || !U || !gA || !gB) return LORA_E_BADARG;
    GradParams p{};
    p.prob[0].S = dY; p.prob[0].P = T; p.prob[0].G = gB; p.prob[0].C = N; p.prob[0].out_kn = 0;
    p.prob[1].S = X;  p.prob[1].P = U; p.prob[1].G = gA; p.prob[1].C = K; p.prob[1].out_kn = 1;
    p.M = M; p.r = r; p.scale = scale;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const double e = dtype == LORA_F32 ? 4.0 : 2.0;
    ProfWork work(e * ((double)M * N + (double)M * K) + 4.0 * r * (double)(K + N) + 8.0 * M * r,
                  2.0 * M * r * (double)(K + N));
    int rc;
    switch (dtype) {
        case LORA_F32: rc = launch_grad<float>(p, s); break;
        case LORA_F16: rc = launch_grad<half_t>(p, s); break;
        case LORA_BF16: rc = launch_grad<bf16_t>(p, s); break;
        default: rc = LORA_E_BADARG;
    }
    return rc;
}

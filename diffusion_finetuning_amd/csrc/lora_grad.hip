// Gradients of the rank-r factors (base W frozen): the two tall-skinny reductions over M
//     gB[N,r] = s·dYᵀ·T        gA[r,K] = s·Uᵀ·X          (T = X·Aᵀ, U = dY·B, both [M,r] fp32)
// — the autograd of lora_diffusion/lora.py:49-50 restricted to the parameters that
// lora.py:179-180 mark trainable.  Both are  G[c,j] = s·Σ_m S[m,c]·P[m,j]  with a streamed
// operand S ∈ {dY, X} read exactly once, so this is an HBM-streaming kernel:
//   - a thread owns one 16-byte column chunk (8 halfs / 4 floats) and walks rows, UNROLL rows per trip so
//     several independent loads are in flight; r×VEC fp32 accumulators stay in VGPRs; the P row is a
//     broadcast load;
//   - 256 threads cover ⌊256/CL⌋ rows per pass when the strip is narrower than the workgroup; the row
//     groups are combined through LDS (16-B writes, one summing thread per output);
//   - the M range is cut into `n_blocks` row blocks; each block STORES its partial sums (plain 16-B-friendly
//     stores, no global atomics: the outputs are only a few KB wide, and atomics from hundreds of
//     workgroups onto so few cache lines serialise at the memory side);
//   - lora_reduce_partials sums the row blocks in index order (deterministic) — one launch for the whole
//     gradient slab of a model, or one per call in the plain autograd mode.
// Both problems of a layer go out in ONE launch (blockIdx.z selects dY→gB or X→gA).
#include "common.h"

namespace {

struct GradProblem {
    const void* S;    // [M, C]
    const float* P;   // [M, r]
    float* G;         // partial output of row block 0
    int C;
    int out_kn;       // 1: G is [r, C] (gA layout: j*C + c); 0: G is [C, r] (gB layout: c*r + j)
    int CL;           // column chunks (threads) per row inside a strip
    int strips;
};
struct GradParams {
    GradProblem prob[2];
    int64_t M;
    int64_t part_stride;  // floats between consecutive row blocks' partials
    int r;
    int rows_per_block;
    float scale;
};

constexpr int kUnroll = 8;

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

    if (active && m_begin < m_end) {
        const T* S = static_cast<const T*>(q.S) + c0 + c_local;
        const int64_t last = m_end - 1;
        for (int64_t m = m_begin + rsub; m < m_end; m += (int64_t)rows_pp * kUnroll) {
            // kUnroll independent rows per trip; loads are unconditional from clamped rows (a load under
            // a per-lane condition is branched around and waited for one by one), tails are zero-weighted
            Chunk<T> s[kUnroll];
            float pv[kUnroll][RP];
#pragma unroll
            for (int u = 0; u < kUnroll; ++u) {
                const int64_t mu = m + (int64_t)u * rows_pp;
                const int64_t ml = mu < m_end ? mu : last;
                s[u] = *reinterpret_cast<const Chunk<T>*>(S + ml * q.C);
#pragma unroll
                for (int j = 0; j < RP; ++j) pv[u][j] = q.P[ml * p.r + (j < p.r ? j : p.r - 1)];
            }
#pragma unroll
            for (int u = 0; u < kUnroll; ++u) {
                const bool ok = m + (int64_t)u * rows_pp < m_end;
#pragma unroll
                for (int j = 0; j < RP; ++j) {
                    const float w = (ok && j < p.r) ? pv[u][j] : 0.f;
#pragma unroll
                    for (int e = 0; e < VEC; ++e) acc[j][e] = fmaf(to_f32<T>(s[u].v[e]), w, acc[j][e]);
                }
            }
        }
    }

    // combine the row groups through LDS, four rank columns at a time: image [row group][jj][strip column],
    // 16-B writes, then every output is summed over the row groups by one thread and stored (plain stores)
    float* G = q.G + (int64_t)blockIdx.y * p.part_stride;
    const int WS = CL * VEC;
    const bool writer = rsub < rows_pp;
#pragma unroll
    for (int j0 = 0; j0 < RP; j0 += 4) {
        if (j0 < p.r) {  // wave-uniform
            __syncthreads();
            if (writer) {
#pragma unroll
                for (int jj = 0; jj < 4; ++jj)
#pragma unroll
                    for (int e = 0; e < VEC; e += 4)
                        *reinterpret_cast<float4*>(&sred[(rsub * 4 + jj) * WS + c_local + e]) =
                            float4{acc[j0 + jj][e], acc[j0 + jj][e + 1], acc[j0 + jj][e + 2], acc[j0 + jj][e + 3]};
            }
            __syncthreads();
            for (int i = tid; i < 4 * stripW; i += 256) {
                const int jj = i / stripW, c = i - jj * stripW;
                const int j = j0 + jj;
                if (j < p.r) {
                    float sum = 0.f;
                    for (int g = 0; g < rows_pp; ++g) sum += sred[(g * 4 + jj) * WS + c];
                    sum *= p.scale;
                    if (q.out_kn)
                        G[(int64_t)j * q.C + c0 + c] = sum;       // gA[j, c]: contiguous runs per j
                    else
                        G[(int64_t)(c0 + c) * p.r + j] = sum;     // gB[c, j]
                }
            }
        }
    }
}

// Unaligned / large-rank path: one thread per output element, serial over the row block.  Correct, not fast.
template <typename T>
__global__ void lora_grad_generic_kernel(GradParams p) {
    const GradProblem& q = p.prob[blockIdx.z];
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (int64_t)q.C * p.r) return;
    const int c = (int)(idx / p.r), j = (int)(idx % p.r);
    const T* S = static_cast<const T*>(q.S);
    const int64_t m_begin = (int64_t)blockIdx.y * p.rows_per_block;
    int64_t m_end = m_begin + p.rows_per_block;
    if (m_end > p.M) m_end = p.M;
    float s = 0.f;
    for (int64_t m = m_begin; m < m_end; ++m) s = fmaf(to_f32<T>(S[m * q.C + c]), q.P[m * p.r + j], s);
    float* G = q.G + (int64_t)blockIdx.y * p.part_stride;
    (q.out_kn ? G[(int64_t)j * q.C + c] : G[idx]) = p.scale * s;
}

// grads[i] (+)= Σ_b partials[b·stride + i], b ascending: deterministic.
__global__ __launch_bounds__(256) void reduce_partials_kernel(const float* partials, int64_t stride, int n_blocks,
                                                              float* grads, int64_t n, int accumulate) {
    const int64_t nvec = n >> 2;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < nvec; i += (int64_t)gridDim.x * 256) {
        float4 s = accumulate ? reinterpret_cast<const float4*>(grads)[i] : float4{0.f, 0.f, 0.f, 0.f};
        for (int b = 0; b < n_blocks; ++b) {
            const float4 v = *reinterpret_cast<const float4*>(partials + (int64_t)b * stride + 4 * i);
            s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
        }
        reinterpret_cast<float4*>(grads)[i] = s;
    }
    if (blockIdx.x == 0) {
        for (int64_t i = (nvec << 2) + threadIdx.x; i < n; i += 256) {
            float s = accumulate ? grads[i] : 0.f;
            for (int b = 0; b < n_blocks; ++b) s += partials[(int64_t)b * stride + i];
            grads[i] = s;
        }
    }
}

template <typename T>
int launch_grad(GradParams p, int n_blocks, hipStream_t stream) {
    constexpr int VEC = ElemTraits<T>::kVec;
    p.rows_per_block = (int)((p.M + n_blocks - 1) / n_blocks);
    if (p.rows_per_block < 1) p.rows_per_block = 1;
    bool fast = p.r <= 16;
    for (int i = 0; i < 2; ++i) fast = fast && (p.prob[i].C % VEC) == 0 && aligned16(p.prob[i].S);
    if (!fast) {
        int cmax = p.prob[0].C > p.prob[1].C ? p.prob[0].C : p.prob[1].C;
        const int64_t n = (int64_t)cmax * p.r;
        hipLaunchKernelGGL(lora_grad_generic_kernel<T>, dim3((unsigned)((n + 255) / 256), n_blocks, 2), dim3(256), 0,
                           stream, p);
        LORA_LAUNCH_CHECK();
        return LORA_OK;
    }
    const int rp = p.r <= 4 ? 4 : (p.r <= 8 ? 8 : 16);
    const int cl_cap = 256;  // a strip is at most one workgroup wide
    int max_strips = 1, max_lds = 0;
    for (int i = 0; i < 2; ++i) {
        GradProblem& q = p.prob[i];
        const int chunks = q.C / VEC;
        // balance strips: smallest strip count that respects the cap, then even widths
        const int strips = (chunks + cl_cap - 1) / cl_cap;
        q.CL = (chunks + strips - 1) / strips;
        q.strips = strips;
        if (strips > max_strips) max_strips = strips;
        const int lds = 256 * 4 * VEC * 4;  // [row groups][4][strip] floats, one 4-column slab at a time
        if (lds > max_lds) max_lds = lds;
    }
    dim3 grid(max_strips, n_blocks, 2);
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
                                      float* gA_part, float* gB_part, int64_t part_stride, int n_blocks,
                                      int64_t M, int K, int N, int r, float scale, int dtype, void* stream) {
    if (M < 0 || K <= 0 || N <= 0 || n_blocks < 1) return LORA_E_BADARG;
    if (r < 1 || r > (K < N ? K : N)) return LORA_E_RANK;
    if (!gA_part || !gB_part) return LORA_E_BADARG;
    if (M > 0 && (!dY || !X || !T || !U)) return LORA_E_BADARG;
    GradParams p{};
    p.prob[0].S = dY; p.prob[0].P = T; p.prob[0].G = gB_part; p.prob[0].C = N; p.prob[0].out_kn = 0;
    p.prob[1].S = X;  p.prob[1].P = U; p.prob[1].G = gA_part; p.prob[1].C = K; p.prob[1].out_kn = 1;
    p.M = M; p.r = r; p.scale = scale; p.part_stride = part_stride;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const double e = dtype == LORA_F32 ? 4.0 : 2.0;
    ProfWork work(e * ((double)M * N + (double)M * K) + 4.0 * r * (double)(K + N) + 8.0 * M * r,
                  2.0 * M * r * (double)(K + N));
    switch (dtype) {
        case LORA_F32: return launch_grad<float>(p, n_blocks, s);
        case LORA_F16: return launch_grad<half_t>(p, n_blocks, s);
        case LORA_BF16: return launch_grad<bf16_t>(p, n_blocks, s);
        default: return LORA_E_BADARG;
    }
}

extern "C" int lora_reduce_partials(const float* partials, int64_t part_stride, int n_blocks, float* grads,
                                    int64_t n, int accumulate, void* stream) {
    if (!partials || !grads || n < 1 || n_blocks < 1) return LORA_E_BADARG;
    if (!aligned16(partials) || !aligned16(grads) || (part_stride & 3)) return LORA_E_ALIGN;
    int64_t blocks = (n / 4 + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(reduce_partials_kernel, dim3((unsigned)blocks), dim3(256), 0, static_cast<hipStream_t>(stream),
                       partials, part_stride, n_blocks, grads, n, accumulate);
    LORA_LAUNCH_CHECK();
    return LORA_OK;
}

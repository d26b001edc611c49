// Gradients of the rank-r factors (base W frozen): the two tall-skinny reductions over M
//     gB[N,r] = s·dYᵀ·T        gA[r,K] = s·Uᵀ·X          (T = X·Aᵀ, U = dY·B, both [M,r] fp32)
// — the autograd of lora_diffusion/lora.py:49-50 restricted to the parameters that
// lora.py:179-180 mark trainable.  Both are  G[c,j] = s·Σ_m S[m,c]·P[m,j]  with a streamed
// operand S ∈ {dY, X} read exactly once, so this is an HBM-streaming kernel.  16-bit operands take the matrix-core form
// (lora_grad_mfma_kernel below); fp32 operands the VALU form:
//   - a thread owns one 16-byte column chunk (8 halfs / 4 floats) and walks rows, several rows per trip so
//     that independent loads are in flight; r×VEC fp32 accumulators stay in VGPRs; the P row is a broadcast load;
//   - 256 threads cover ⌊256/CL⌋ rows per pass when the strip is narrower than the workgroup; the row
//     groups are combined through LDS (16-B writes, one summing thread per output);
//   - the M range is cut into row blocks; each block STORES its partial sums (plain stores, no global
//     atomics: the outputs are only a few KB wide, and atomics from hundreds of workgroups onto so few
//     cache lines serialise at the memory side); lora_fold_partials / lora_reduce_partials sum the row
//     blocks in index order (deterministic).
// A launch covers a TABLE of such problems that lives in the kernel-argument segment (≤ 28 problems per
// launch, no device-side table to upload, safe to record into a hipGraph): a training step hands ALL its
// 2×144 problems to lora_grad_batched at the end of backward — ~10 chip-filling launches instead of 144
// latency-bound ones.  Operands may be strided views (a slice of a grouped projection's output) and the
// rank columns of one problem may belong to several layers (grouped q/k/v: U is [M, 3r], three gA outputs).
#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "attn_common.h"

namespace {

struct GradItem {       // 128 bytes
    const void* S;      // [M, C] elements, row stride s_stride
    const float* P;     // [M, r] fp32, row stride p_stride
    float* out[4];      // partial output of row block 0 for rank group g = j / rg
    int64_t s_stride;
    int64_t part_stride;  // floats between consecutive row blocks' partials
    int p_stride, C, r, rg;
    int out_kn;         // 1: out is [rg, C] (gA layout: jl*C + c); 0: out is [C, rg] (gB layout: c*rg + jl)
    int CL;             // column chunks (threads) per row inside a strip
    int strips, rows_per_block;
    int nb;
    float scale;
    int64_t M;
    int64_t pad_[2];
};
static_assert(sizeof(GradItem) == 128, "GradItem is laid out for the kernel-argument segment");

constexpr int kItemsPerLaunch = 28;
struct GradBatch {
    GradItem item[kItemsPerLaunch];
    int first_block[kItemsPerLaunch + 1];  // prefix sums of strips·nb
    int n;
};
static_assert(sizeof(GradBatch) <= 4096, "kernel arguments are limited to 4 KB");

constexpr int kChunkRows = 256;  // rows of P staged in LDS at a time

template <typename T, int RP /* padded rank: 4, 8, 12, 16 */>
__global__ __launch_bounds__(256) void lora_grad_kernel(const GradBatch p) {
    constexpr int VEC = ElemTraits<T>::kVec;
    constexpr int UNROLL = RP >= 12 ? 4 : 8;  // rows in flight per thread
    extern __shared__ __attribute__((aligned(16))) float smem_f[];
    float* sred = smem_f;                           // [row groups][4][strip] — the cross-row-group reduction image
    float* sP = smem_f + 256 * 4 * VEC;             // [kChunkRows][RP]      — the P rows of the current row chunk

    // which problem does this workgroup belong to: linear scan of ≤ 28 prefix sums held in SGPRs
    int it = 0;
    for (int i = 1; i < p.n; ++i) it += ((int)blockIdx.x >= p.first_block[i]) ? 1 : 0;
    it = __builtin_amdgcn_readfirstlane(it);
    const GradItem& q = p.item[it];
    const int local = blockIdx.x - p.first_block[it];
    const int strip = local % q.strips;
    const int rb = local / q.strips;

    const int tid = threadIdx.x;
    const int CL = q.CL;
    const int rows_pp = 256 / CL;  // rows per pass
    const int rsub = tid / CL;
    const int cg = tid - rsub * CL;
    const int c_local = cg * VEC;
    const int c0 = strip * CL * VEC;
    const int stripW = min(CL * VEC, q.C - c0);
    const bool active = rsub < rows_pp && (c_local < stripW);
    const int r = q.r;

    float acc[RP][VEC];
#pragma unroll
    for (int j = 0; j < RP; ++j)
#pragma unroll
        for (int e = 0; e < VEC; ++e) acc[j][e] = 0.f;

    const int64_t m_begin = (int64_t)rb * q.rows_per_block;
    int64_t m_end = m_begin + q.rows_per_block;
    if (m_end > q.M) m_end = q.M;
    const T* S = static_cast<const T*>(q.S) + c0 + c_local;

    for (int64_t mc = m_begin; mc < m_end; mc += kChunkRows) {  // wave-uniform trip count
        const int rows = (int)min((int64_t)kChunkRows, m_end - mc);
        // P rows of this chunk → LDS, zero-padded to RP columns: the row loop then reads them as 16-byte broadcasts
        // instead of RP scalar global loads per row (which out-numbered the S loads 4:1 … 16:1 in the issue stream)
        __syncthreads();
        for (int i = tid; i < rows * RP; i += 256) {
            const int row = i / RP, j = i - row * RP;
            sP[i] = j < r ? q.P[(mc + row) * q.p_stride + j] : 0.f;
        }
        __syncthreads();
        if (active) {
            const int last = rows - 1;
            for (int m = rsub; m < rows; m += rows_pp * UNROLL) {
                // UNROLL independent rows per trip; loads are unconditional from clamped rows (a load under
                // a per-lane condition is branched around and waited for one by one), tails are zero-weighted
                Chunk<T> s[UNROLL];
#pragma unroll
                for (int u = 0; u < UNROLL; ++u) {
                    const int mu = m + u * rows_pp;
                    const int ml = mu < rows ? mu : last;
                    s[u] = *reinterpret_cast<const Chunk<T>*>(S + (mc + ml) * q.s_stride);
                }
#pragma unroll
                for (int u = 0; u < UNROLL; ++u) {
                    const int mu = m + u * rows_pp;
                    const bool ok = mu < rows;
                    const float4* prow = reinterpret_cast<const float4*>(sP + (ok ? mu : last) * RP);
#pragma unroll
                    for (int j4 = 0; j4 < RP / 4; ++j4) {
                        const float4 pv = prow[j4];
                        const float w[4] = {ok ? pv.x : 0.f, ok ? pv.y : 0.f, ok ? pv.z : 0.f, ok ? pv.w : 0.f};
#pragma unroll
                        for (int jj = 0; jj < 4; ++jj)
#pragma unroll
                            for (int e = 0; e < VEC; ++e)
                                acc[j4 * 4 + jj][e] = fmaf(to_f32<T>(s[u].v[e]), w[jj], acc[j4 * 4 + jj][e]);
                    }
                }
            }
        }
    }

    // combine the row groups through LDS, four rank columns at a time: image [row group][jj][strip column],
    // 16-B writes, then every output is summed over the row groups by one thread and stored (plain stores)
    const int WS = CL * VEC;
    const bool writer = rsub < rows_pp;
    const int64_t part_off = (int64_t)rb * q.part_stride;
#pragma unroll
    for (int j0 = 0; j0 < RP; j0 += 4) {
        if (j0 < r) {  // wave-uniform
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
                if (j < r) {
                    float sum = 0.f;
                    for (int g = 0; g < rows_pp; ++g) sum += sred[(g * 4 + jj) * WS + c];
                    sum *= q.scale;
                    const int grp = j / q.rg, jl = j - grp * q.rg;
                    float* G = q.out[grp] + part_off;
                    if (q.out_kn)
                        G[(int64_t)jl * q.C + c0 + c] = sum;        // gA[j, c]: contiguous runs per j
                    else
                        G[(int64_t)(c0 + c) * q.rg + jl] = sum;     // gB[c, j]
                }
            }
        }
    }
}

// The same reduction on the matrix cores, for 16-bit operands:  G[j, c] = Σ_m P[m, j]·S[m, c]  is an MFMA with the ROW index as
// its contraction — first operand P (16 rank columns, zero-padded), second operand a 16-column fragment of S.  At rank 16
// the VALU form above spends 128 FMAs per 16-byte chunk and is instruction-bound at ≈ 2 TB/s; here a 32-row × 16-column
// fragment costs two MFMAs whatever the rank, and the kernel streams.
//   - P is fp32: staged per 256-row chunk as a hi + lo pair of 16-bit values (the split lora_gemm.hip's rank
//     epilogue uses), already in operand order, so a lane's eight contraction values are one 16-byte LDS read.  fp16 has
//     five exponent bits, so the pair is only exact in a window: below 2^-3 the lo part is subnormal, below 2^-14 the hi
//     part is, above 65504 it is inf — and U = dY·B shrinks with the loss scale while T = X·Aᵀ is not scaled at all.  So
//     for fp16 every rank column of P is multiplied by a power of two chosen per ROW BLOCK (its largest magnitude lands
//     in [2^13, 2^14): a prologue pass over the block's P rows, L2-resident, they are shared by all strips) and the
//     inverse goes into the final scale: 22 significant bits for the entries within 2^-17 of their column's maximum,
//     degrading gracefully below, whatever the magnitude of P.  bf16 has fp32's exponent range and needs none of this;
//   - S goes global → registers → a wave-private LDS tile (32 rows × 64 columns) and comes back through the transposing
//     read (ds_read_b64_tr_b16), which delivers exactly the second operand's layout; no workgroup barrier on that path —
//     LDS operations of one wave complete in order — and the next step's rows are in flight while this one is multiplied;
//   - a wave owns 64 columns of the strip for the whole row block: no cross-wave reduction, 16 accumulator registers.
// Work decomposition (items, strips, row blocks, partial layout) is the VALU kernel's, so the fold is unchanged.
// elements per row of a wave's S tile: 160 B.  A row stride ≡ 32 (mod 64) bytes is what keeps ds_read_b64_tr_b16 (two groups
// of 32 lanes over 64 banks) conflict-free on gfx950; 144 B made every transposing read a 2-way conflict (PMC, round 4)
constexpr int kTileLd = 64 + 16;
// the 16-byte slot of lane group lq inside rank column j's 64-byte record of the P image: ds_read_b128 is serviced in the
// lane groups {0–3,12–15,20–27}, {4–11,16–19,28–31}, …, i.e. columns j and j ± 4, ± 8, ± 12 — which share their 16 banks —
// meet in one group with two different lq; this XOR keeps their four slots distinct in every group
__device__ __forceinline__ int p_slot(int j, int lq) { return lq ^ ((4 - (j >> 2)) & 3); }
template <typename T>
__device__ __forceinline__ void grad_mfma_body(const GradItem& q, const int local) {
    using F8 = typename Mma<T>::F8;
    __shared__ __attribute__((aligned(16))) T sPh[kChunkRows * 16];  // [32-row group][j][lq][8 rows in operand order]
    __shared__ __attribute__((aligned(16))) T sPl[kChunkRows * 16];
    __shared__ __attribute__((aligned(16))) T sS[4][32 * kTileLd];
    __shared__ float sColMax[4][16];
    __shared__ float sScale[2][16];  // [0][j]: 2^k of rank column j for this row block, [1][j]: 2^-k
    constexpr bool kScaleP = sizeof(typename Mma<T>::F8) == 16 && std::is_same<T, half_t>::value;

    const int strip = local % q.strips;
    const int rb = local / q.strips;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l15 = lane & 15, lq = lane >> 4;
    const int c0 = strip * q.CL * 8;
    const int stripW = min(q.CL * 8, q.C - c0);
    const int cw = wave * 64;                      // this wave's first column inside the strip
    const bool wave_on = cw < stripW;              // wave-uniform
    const int r = q.r;

    const int64_t m_begin = (int64_t)rb * q.rows_per_block;
    int64_t m_end = m_begin + q.rows_per_block;
    if (m_end > q.M) m_end = q.M;
    const int n_rows = (int)(m_end - m_begin);
    const int n_steps = (n_rows + 31) / 32;

    // lane → four 16-byte pieces of the 32 × 64 tile: piece lane + 64·i is row (lane>>3) + 8·i, columns (lane&7)·8 …
    // columns past the strip are read from its last chunk (their products land in outputs nobody stores)
    const int prow = lane >> 3;
    const int pcol = min(cw + (lane & 7) * 8, stripW - 8);
    const T* S = static_cast<const T*>(q.S) + m_begin * q.s_stride + c0 + (wave_on ? pcol : 0);
    T* tile = sS[wave];
    T* tile_w = tile + prow * kTileLd + (lane & 7) * 8;

    f32x4 acc[4];
#pragma unroll
    for (int f = 0; f < 4; ++f) acc[f] = f32x4{0.f, 0.f, 0.f, 0.f};

    typedef unsigned raw4 __attribute__((ext_vector_type(4)));  // 16 bytes, register-resident
    // two steps of rows in flight (register sets a / b) while a third is multiplied
    raw4 a0, a1, a2, a3, b0, b1, b2, b3;
#define GRAD_LOAD_STEP(x, s_)                                                                                        \
    do {                                                                                                             \
        const int rbase_ = (s_) * 32 + prow; /* rows past the block: clamped, their P is zero */                     \
        x##0 = *reinterpret_cast<const raw4*>(S + (int64_t)min(rbase_, n_rows - 1) * q.s_stride);                    \
        x##1 = *reinterpret_cast<const raw4*>(S + (int64_t)min(rbase_ + 8, n_rows - 1) * q.s_stride);                \
        x##2 = *reinterpret_cast<const raw4*>(S + (int64_t)min(rbase_ + 16, n_rows - 1) * q.s_stride);               \
        x##3 = *reinterpret_cast<const raw4*>(S + (int64_t)min(rbase_ + 24, n_rows - 1) * q.s_stride);               \
    } while (0)
#define GRAD_STEP(x, s_)                                                                                             \
    do {                                                                                                             \
        const raw4 c0_ = x##0, c1_ = x##1, c2_ = x##2, c3_ = x##3;                                                   \
        if ((s_) + 2 < n_steps) GRAD_LOAD_STEP(x, (s_) + 2);                                                         \
        *reinterpret_cast<raw4*>(tile_w) = c0_;                                                                      \
        *reinterpret_cast<raw4*>(tile_w + 8 * kTileLd) = c1_;                                                        \
        *reinterpret_cast<raw4*>(tile_w + 16 * kTileLd) = c2_;                                                       \
        *reinterpret_cast<raw4*>(tile_w + 24 * kTileLd) = c3_;                                                       \
        const int at_ = ((((s_) & 7) * 16 + l15) * 4 + p_slot(l15, lq)) << 3;                                        \
        const F8 ph_ = *reinterpret_cast<const F8*>(sPh + at_);                                                      \
        const F8 pl_ = *reinterpret_cast<const F8*>(sPl + at_);                                                      \
        _Pragma("unroll") for (int f = 0; f < 4; ++f) {                                                              \
            const F8 sf_ = tr_pair<T>(lds_tr_block(tile + f * 16, kTileLd, lane),                                    \
                                      lds_tr_block(tile + 16 * kTileLd + f * 16, kTileLd, lane));                    \
            acc[f] = Mma<T>::k32(ph_, sf_, acc[f]);                                                                  \
            acc[f] = Mma<T>::k32(pl_, sf_, acc[f]);                                                                  \
        }                                                                                                            \
    } while (0)
    a0 = a1 = a2 = a3 = b0 = b1 = b2 = b3 = raw4{0u, 0u, 0u, 0u};
    if (wave_on && n_steps > 0) GRAD_LOAD_STEP(a, 0);
    if (wave_on && n_steps > 1) GRAD_LOAD_STEP(b, 1);

    if (kScaleP) {  // per rank column: the power of two that centres this row block's P in fp16's exact hi + lo window
        const int j = tid & 15;
        const float* src = q.P + m_begin * q.p_stride + (j < r ? j : 0);
        float mx = 0.f;
        for (int row = tid >> 4; row < n_rows; row += 16) mx = fmaxf(mx, fabsf(src[(int64_t)row * q.p_stride]));
        mx = fmaxf(mx, __shfl_xor(mx, 16));
        mx = fmaxf(mx, __shfl_xor(mx, 32));
        if (lane < 16) sColMax[wave][lane] = mx;
        __syncthreads();
        if (tid < 16) {
            const float m4 = fmaxf(fmaxf(sColMax[0][tid], sColMax[1][tid]), fmaxf(sColMax[2][tid], sColMax[3][tid]));
            int e = 0;
            if (m4 > 0.f && m4 < __builtin_inff()) (void)frexpf(m4, &e);  // m4 < 2^e
            const int k = max(-100, min(100, 14 - e));                    // (inf / nan in P: k = 14, the gradient comes out non-finite)
            sScale[0][tid] = m4 > 0.f ? ldexpf(1.f, k) : 1.f;
            sScale[1][tid] = m4 > 0.f ? ldexpf(1.f, -k) : 1.f;
        }
        // (published by the barrier that opens the first chunk's staging below)
    }

    for (int s = 0; s < n_steps; s += 2) {
        if ((s & 7) == 0) {  // the P rows of the next 256: hi/lo pairs in operand order
            __syncthreads();
            const int base = s * 32;
            // a thread builds two operand vectors: rank column j, the eight rows one lane group multiplies.  Loads are
            // unconditional from clamped addresses (a load under a per-lane condition is waited for one by one)
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                const int u = tid + 256 * k;
                const int j = u & 15, slot = u >> 4;
                const int row0 = base + (slot >> 2) * 32 + (slot & 3) * 4;
                const float* src = q.P + m_begin * q.p_stride + (j < r ? j : 0);
                float v[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const int row = row0 + (e & 3) + ((e >> 2) << 4);
                    v[e] = src[(int64_t)min(row, n_rows - 1) * q.p_stride];
                }
                const float pscale = kScaleP ? sScale[0][j] : 1.f;
                F8 hi, lo;
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const int row = row0 + (e & 3) + ((e >> 2) << 4);
                    const float x = (row < n_rows && j < r) ? v[e] * pscale : 0.f;
                    const T h = from_f32<T>(x);
                    hi[e] = h;
                    lo[e] = from_f32<T>(x - to_f32<T>(h));
                }
                const int at = (((slot >> 2) * 16 + j) * 4 + p_slot(j, slot & 3)) << 3;
                *reinterpret_cast<F8*>(sPh + at) = hi;
                *reinterpret_cast<F8*>(sPl + at) = lo;
            }
            __syncthreads();
        }
        if (wave_on) {
            GRAD_STEP(a, s);
            if (s + 1 < n_steps) GRAD_STEP(b, s + 1);
        }
    }

#undef GRAD_STEP
#undef GRAD_LOAD_STEP
    // lane (l15, lq) of fragment f holds G[j = 4·lq + e][c = cw + 16·f + l15]
    if (wave_on) {
        const int64_t part_off = (int64_t)rb * q.part_stride;
#pragma unroll
        for (int f = 0; f < 4; ++f) {
            const int c = cw + f * 16 + l15;
            if (c < stripW) {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int j = lq * 4 + e;
                    if (j < r) {
                        const int grp = j / q.rg, jl = j - grp * q.rg;
                        float* G = (grp == 0 ? q.out[0] : grp == 1 ? q.out[1] : grp == 2 ? q.out[2] : q.out[3]) + part_off;  // (no indexed read: the item may live in SGPRs)
                        const float v = acc[f][e] * (kScaleP ? q.scale * sScale[1][j] : q.scale);
                        if (q.out_kn)
                            G[(int64_t)jl * q.C + c0 + c] = v;
                        else
                            G[(int64_t)(c0 + c) * q.rg + jl] = v;
                    }
                }
            }
        }
    }
}

// One item as wave-uniform scalars, field by field — a plain struct copy went through vector loads and a scratch image, and
// the body must not index `out[]` dynamically on the copy.  `cp`: a reference into the kernel-argument struct (already scalar
// loads), or a CONSTANT-address-space pointer into the plan in device memory (s_load into SGPRs).
template <typename ItemRef> __device__ __forceinline__ GradItem copy_item(ItemRef cp, int* first_block) {
    GradItem q;
    q.S = cp->S; q.P = cp->P;
    q.out[0] = cp->out[0]; q.out[1] = cp->out[1]; q.out[2] = cp->out[2]; q.out[3] = cp->out[3];
    q.s_stride = cp->s_stride; q.part_stride = cp->part_stride;
    q.p_stride = cp->p_stride; q.C = cp->C; q.r = cp->r; q.rg = cp->rg; q.out_kn = cp->out_kn; q.CL = cp->CL;
    q.strips = cp->strips; q.rows_per_block = cp->rows_per_block; q.nb = cp->nb; q.scale = cp->scale; q.M = cp->M;
    *first_block = (int)cp->pad_[0];
    return q;
}
// (only for a REAL device pointer: a by-value kernel argument has no constant-address-space address unless the optimizer
// happens to elide its private copy — ADVICE r5)
__device__ __forceinline__ GradItem fetch_item(const GradItem* gp, int* first_block) {
    typedef const __attribute__((address_space(4))) GradItem* ItemPtr;
    return copy_item((ItemPtr)gp, first_block);
}

template <typename T>
__global__ __launch_bounds__(256) void lora_grad_mfma_kernel(const GradBatch p) {
    int it = 0;
    for (int i = 1; i < p.n; ++i) it += ((int)blockIdx.x >= p.first_block[i]) ? 1 : 0;
    it = __builtin_amdgcn_readfirstlane(it);
    int unused;
    const GradItem q = copy_item(&p.item[it], &unused);  // (the table travels in the kernel arguments: plain reads)
    grad_mfma_body<T>(q, (int)blockIdx.x - p.first_block[it]);
}

// The same kernel over a PLAN in device memory (lora_grad_plan / lora_grad_planned): every problem of a step in ONE launch —
// the ≤ 28-problem launches above end on a tail each, and the last few of a step hold a few dozen workgroups (15 µs apiece for
// next to no bytes).  plan = [n items of 128 bytes | one int per workgroup: its item]; an item carries its first workgroup id.
template <typename T>
__global__ __launch_bounds__(256) void lora_grad_mfma_planned_kernel(const GradItem* __restrict__ items,
                                                                     const int* __restrict__ block_item) {
    const int it = __builtin_amdgcn_readfirstlane(block_item[blockIdx.x]);
    int first;
    const GradItem q = fetch_item(items + it, &first);
    grad_mfma_body<T>(q, (int)blockIdx.x - first);
}

// Unaligned / large-rank path: one thread per output element, serial over the row block.  Correct, not fast.
struct GenericGrad {
    const void* S;
    const float* P;
    float* G;
    int64_t M, part_stride;
    int C, r, out_kn, rows_per_block;
    float scale;
};
template <typename T>
__global__ void lora_grad_generic_kernel(GenericGrad q) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (int64_t)q.C * q.r) return;
    const int c = (int)(idx / q.r), j = (int)(idx % q.r);
    const T* S = static_cast<const T*>(q.S);
    const int64_t m_begin = (int64_t)blockIdx.y * q.rows_per_block;
    int64_t m_end = m_begin + q.rows_per_block;
    if (m_end > q.M) m_end = q.M;
    float s = 0.f;
    for (int64_t m = m_begin; m < m_end; ++m) s = fmaf(to_f32<T>(S[m * q.C + c]), q.P[m * q.r + j], s);
    float* G = q.G + (int64_t)blockIdx.y * q.part_stride;
    (q.out_kn ? G[(int64_t)j * q.C + c] : G[idx]) = q.scale * s;
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

// The same fold for a table of slab ranges, each with its OWN block count (a layer with few rows wrote few
// partials; nothing else is read): ranges[k] = {offset, length, blocks, 0} in floats, device memory.
__global__ __launch_bounds__(256) void fold_partials_kernel(const int64_t* ranges, const float* partials, int64_t stride,
                                                            float* grads, int accumulate) {
    const int64_t* e = ranges + (int64_t)blockIdx.y * 4;
    const int64_t off = e[0], n = e[1];
    const int nb = (int)e[2];
    const float* src = partials + off;
    float* dst = grads + off;
    if (((off | n) & 3) == 0) {
        const int64_t nvec = n >> 2;
        for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < nvec; i += (int64_t)gridDim.x * 256) {
            float4 s = accumulate ? reinterpret_cast<const float4*>(dst)[i] : float4{0.f, 0.f, 0.f, 0.f};
            for (int b = 0; b < nb; ++b) {
                const float4 v = *reinterpret_cast<const float4*>(src + (int64_t)b * stride + 4 * i);
                s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
            }
            reinterpret_cast<float4*>(dst)[i] = s;
        }
    } else {
        for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
            float s = accumulate ? dst[i] : 0.f;
            for (int b = 0; b < nb; ++b) s += src[(int64_t)b * stride + i];
            dst[i] = s;
        }
    }
}

// Row blocks and strips of one problem, from a sweep over all 288 problems of an SD1.5 step (tools/gemm_bench.py --grads,
// profiles/README.md): 512 rows per block (at most 64 blocks) and strips of 32 chunks (256 columns in 16-bit types) —
// many narrow workgroups balance the chip better than few wide ones (541 µs at 256 rows × 256 chunks, 397 µs here, 5.0 TB/s)
// and halve the partial sums the fold has to read.
int plan_row_blocks(int64_t M) {
    static const int rows = [] { const char* e = getenv("LORA_GRAD_ROWS"); return e ? atoi(e) : 512; }();  // tuning knob
    int64_t nb = M / rows;
    if (nb < 1) nb = 1;
    if (nb > LORA_GRAD_MAX_BLOCKS) nb = LORA_GRAD_MAX_BLOCKS;
    return (int)nb;
}

template <typename T>
bool plan_item(GradItem& q, int nb) {
    constexpr int VEC = ElemTraits<T>::kVec;
    if (q.C % VEC != 0 || !aligned16(q.S) || (q.s_stride % VEC) != 0) return false;
    const int chunks = q.C / VEC;
    static const int cl_cap = [] { const char* e = getenv("LORA_GRAD_STRIP"); return e ? atoi(e) : 32; }();  // tuning knob, <= 256
    q.strips = (chunks + cl_cap - 1) / cl_cap;  // smallest strip count that respects the cap, then even widths
    q.CL = (chunks + q.strips - 1) / q.strips;
    q.nb = nb;
    q.rows_per_block = (int)((q.M + nb - 1) / nb);
    if (q.rows_per_block < 1) q.rows_per_block = 1;
    return true;
}

// smallest rank class that goes to the matrix-core kernel (16-bit operands only).  All of them by default: on the 288
// problems of an SD1.5 step (tools/gemm_bench.py --grads, GB_RANK) it takes 372 / 403 / 417 µs at rank 4 / 8 / 16 where
// the VALU kernel takes 396 / 657 / 1626.  LORA_GRAD_MFMA=99 sends everything back to the VALU kernel (A/B knob).
int mfma_min_rank() {
    static const int v = [] { const char* e = getenv("LORA_GRAD_MFMA"); return e ? atoi(e) : 4; }();
    return v;
}

template <typename T>
int launch_batch(const GradBatch& b, int rp, hipStream_t stream) {
    constexpr int VEC = ElemTraits<T>::kVec;
    const int lds = 256 * 4 * VEC * 4 + kChunkRows * rp * 4;  // reduction image + the staged P rows
    const dim3 grid((unsigned)b.first_block[b.n]);
    if constexpr (sizeof(T) == 2) {
        if (rp >= mfma_min_rank()) {
            const int id = rp == 4 ? PK_GRAD_R4 : (rp == 8 ? PK_GRAD_R8 : PK_GRAD_R16);
            LORA_LAUNCH(id, (lora_grad_mfma_kernel<T>), grid, dim3(256), 0, stream, b);
            LORA_LAUNCH_CHECK();
            return LORA_OK;
        }
    }
    switch (rp) {
        case 4: LORA_LAUNCH(PK_GRAD_R4, (lora_grad_kernel<T, 4>), grid, dim3(256), lds, stream, b); break;
        case 8: LORA_LAUNCH(PK_GRAD_R8, (lora_grad_kernel<T, 8>), grid, dim3(256), lds, stream, b); break;
        case 12: LORA_LAUNCH(PK_GRAD_R16, (lora_grad_kernel<T, 12>), grid, dim3(256), lds, stream, b); break;
        default: LORA_LAUNCH(PK_GRAD_R16, (lora_grad_kernel<T, 16>), grid, dim3(256), lds, stream, b); break;
    }
    LORA_LAUNCH_CHECK();
    return LORA_OK;
}

template <typename T>
int launch_generic(const lora_grad_problem& g, int nb, hipStream_t stream) {
    if (g.rg != g.r || g.s_stride != g.C || g.p_stride != g.r) return LORA_E_UNSUPPORTED;
    GenericGrad q{};
    q.S = g.S; q.P = g.P; q.G = g.out[0]; q.M = g.M; q.part_stride = g.part_stride; q.C = g.C; q.r = g.r;
    q.out_kn = g.out_kn; q.scale = g.scale;
    q.rows_per_block = (int)((g.M + nb - 1) / nb);
    if (q.rows_per_block < 1) q.rows_per_block = 1;
    const int64_t n = (int64_t)g.C * g.r;
    hipLaunchKernelGGL(lora_grad_generic_kernel<T>, dim3((unsigned)((n + 255) / 256), nb), dim3(256), 0, stream, q);
    LORA_LAUNCH_CHECK();
    return LORA_OK;
}

double problem_bytes(const lora_grad_problem& g, double e) {
    return e * (double)g.M * g.C + 4.0 * (double)g.M * g.r + 4.0 * (double)g.r * g.C;
}

// Plans and launches `n` problems: rank classes (4 / 8 / 16 accumulator columns) go out separately, each in
// chunks of ≤ 28 problems whose tables travel as kernel arguments.
template <typename T>
int run_problems(const lora_grad_problem* probs, int n, const int* n_blocks, hipStream_t stream) {
    const double e = sizeof(T);
    // biggest problems first: the launch then ends on small workgroups instead of on the tail of a 1-MB-per-block problem
    std::vector<int> order(n);
    for (int i = 0; i < n; ++i) order[i] = i;
    std::stable_sort(order.begin(), order.end(), [&](int a, int b) {
        return (double)probs[a].M * probs[a].C > (double)probs[b].M * probs[b].C;
    });
    for (int cls = 0; cls < 4; ++cls) {  // 12 = a grouped q/k/v gA at rank 4 (three rank groups in one pass over X)
        const int rp = 4 * (cls + 1);
        GradBatch b;
        b.n = 0;
        b.first_block[0] = 0;
        double bytes = 0.0, flops = 0.0;
        auto flush = [&]() -> int {
            if (b.n == 0) return LORA_OK;
            ProfWork work(bytes, flops);
            const int st = launch_batch<T>(b, rp, stream);
            b.n = 0;
            bytes = flops = 0.0;
            return st;
        };
        for (int oi = 0; oi < n; ++oi) {
            const int i = order[oi];
            const lora_grad_problem& g = probs[i];
            const int want = g.r <= 4 ? 4 : (g.r <= 8 ? 8 : (g.r <= 12 ? 12 : 16));
            if (g.r > 16) {
                if (cls == 0) {
                    const int st = launch_generic<T>(g, n_blocks[i], stream);
                    if (st != LORA_OK) return st;
                }
                continue;
            }
            if (want != rp) continue;
            GradItem q{};
            q.S = g.S; q.P = g.P;
            for (int k = 0; k < 4; ++k) q.out[k] = g.out[k];
            q.s_stride = g.s_stride; q.part_stride = g.part_stride; q.p_stride = g.p_stride; q.C = g.C; q.r = g.r;
            q.rg = g.rg; q.out_kn = g.out_kn; q.scale = g.scale; q.M = g.M;
            if (!plan_item<T>(q, n_blocks[i])) {  // unaligned operand: the shape-agnostic kernel
                const int st = launch_generic<T>(g, n_blocks[i], stream);
                if (st != LORA_OK) return st;
                continue;
            }
            b.item[b.n] = q;
            b.first_block[b.n + 1] = b.first_block[b.n] + q.strips * q.nb;
            ++b.n;
            bytes += problem_bytes(g, e);
            flops += 2.0 * (double)g.M * g.r * g.C;
            if (b.n == kItemsPerLaunch) {
                const int st = flush();
                if (st != LORA_OK) return st;
            }
        }
        const int st = flush();
        if (st != LORA_OK) return st;
    }
    return LORA_OK;
}

int check_problem(const lora_grad_problem& g) {
    if (g.M < 0 || g.C <= 0 || g.r < 1 || g.rg < 1 || g.rg > g.r || (g.r + g.rg - 1) / g.rg > 4) return LORA_E_BADARG;
    if (g.M > 0 && (!g.S || !g.P)) return LORA_E_BADARG;
    for (int k = 0; k < (g.r + g.rg - 1) / g.rg; ++k)
        if (!g.out[k]) return LORA_E_BADARG;
    return LORA_OK;
}

}  // namespace

extern "C" int lora_grad_row_blocks(int64_t M) { return plan_row_blocks(M); }

extern "C" int lora_grad_batched(const lora_grad_problem* problems, int n, int dtype, void* stream) {
    if (!problems || n < 1) return LORA_E_BADARG;
    if (n > 4096) return LORA_E_BADARG;
    int nb[4096];
    for (int i = 0; i < n; ++i) {
        const int st = check_problem(problems[i]);
        if (st != LORA_OK) return st;
        nb[i] = problems[i].n_blocks > 0 ? problems[i].n_blocks : plan_row_blocks(problems[i].M);
    }
    hipStream_t s = static_cast<hipStream_t>(stream);
    switch (dtype) {
        case LORA_F32: return run_problems<float>(problems, n, nb, s);
        case LORA_F16: return run_problems<half_t>(problems, n, nb, s);
        case LORA_BF16: return run_problems<bf16_t>(problems, n, nb, s);
        default: return LORA_E_BADARG;
    }
}

// ---- one launch for all problems of a step ------------------------------------------------------------------------------
namespace {
template <typename T>
int plan_problems(const lora_grad_problem* probs, int n, char* plan, int64_t plan_bytes, int* n_items, int* n_blocks) {
    std::vector<int> order(n);
    for (int i = 0; i < n; ++i) order[i] = i;
    // biggest problems first: the launch ends on small workgroups instead of on the tail of a 1-MB-per-block problem
    std::stable_sort(order.begin(), order.end(), [&](int a, int b) {
        return (double)probs[a].M * probs[a].C > (double)probs[b].M * probs[b].C;
    });
    std::vector<GradItem> items;
    items.reserve(n);
    int64_t blocks = 0;
    for (int oi = 0; oi < n; ++oi) {
        const lora_grad_problem& g = probs[order[oi]];
        if (g.M == 0) continue;
        if (g.r > 16) return LORA_E_UNSUPPORTED;
        GradItem q{};
        q.S = g.S; q.P = g.P;
        for (int k = 0; k < 4; ++k) q.out[k] = g.out[k];
        q.s_stride = g.s_stride; q.part_stride = g.part_stride; q.p_stride = g.p_stride; q.C = g.C; q.r = g.r;
        q.rg = g.rg; q.out_kn = g.out_kn; q.scale = g.scale; q.M = g.M;
        if (!plan_item<T>(q, g.n_blocks > 0 ? g.n_blocks : plan_row_blocks(g.M))) return LORA_E_UNSUPPORTED;
        q.pad_[0] = blocks;  // first workgroup of the item
        blocks += (int64_t)q.strips * q.nb;
        items.push_back(q);
    }
    if (blocks > (1 << 20)) return LORA_E_UNSUPPORTED;
    const int64_t need = (int64_t)items.size() * (int64_t)sizeof(GradItem) + blocks * 4;
    *n_items = (int)items.size();
    *n_blocks = (int)blocks;
    if (need > plan_bytes) return LORA_E_BADARG;
    if (!items.empty()) std::memcpy(plan, items.data(), items.size() * sizeof(GradItem));
    int* block_item = reinterpret_cast<int*>(plan + items.size() * sizeof(GradItem));
    for (size_t i = 0; i < items.size(); ++i)
        for (int b = 0; b < items[i].strips * items[i].nb; ++b) block_item[items[i].pad_[0] + b] = (int)i;
    return LORA_OK;
}

}  // namespace

extern "C" int64_t lora_grad_plan_bytes(const lora_grad_problem* problems, int n) {
    if (!problems || n < 1) return 0;
    int64_t blocks = 0;
    for (int i = 0; i < n; ++i) {
        const lora_grad_problem& g = problems[i];
        const int nb = g.n_blocks > 0 ? g.n_blocks : plan_row_blocks(g.M);
        blocks += (int64_t)nb * ((g.C + 7) / 8);  // (an upper bound: at least 8 columns per strip)
    }
    return (int64_t)n * (int64_t)sizeof(GradItem) + blocks * 4;
}

extern "C" int lora_grad_plan(const lora_grad_problem* problems, int n, int dtype, void* plan_host, int64_t plan_bytes,
                              int* n_items, int* n_blocks) {
    if (!problems || n < 1 || !plan_host || !n_items || !n_blocks) return LORA_E_BADARG;
    if (mfma_min_rank() > 4) return LORA_E_UNSUPPORTED;  // (A/B knob: the VALU kernels only exist in the table-in-arguments form)
    for (int i = 0; i < n; ++i) {
        const int st = check_problem(problems[i]);
        if (st != LORA_OK) return st;
    }
    switch (dtype) {
        case LORA_F16: return plan_problems<half_t>(problems, n, static_cast<char*>(plan_host), plan_bytes, n_items, n_blocks);
        case LORA_BF16: return plan_problems<bf16_t>(problems, n, static_cast<char*>(plan_host), plan_bytes, n_items, n_blocks);
        case LORA_F32: return LORA_E_UNSUPPORTED;
        default: return LORA_E_BADARG;
    }
}

extern "C" int lora_grad_planned(const void* plan_dev, int n_items, int n_blocks, int dtype, double bytes, double flops,
                                 void* stream) {
    if (n_items < 0 || n_blocks < 0) return LORA_E_BADARG;
    if (n_items == 0 || n_blocks == 0) return LORA_OK;
    if (!plan_dev || !aligned16(plan_dev)) return LORA_E_BADARG;
    const GradItem* items = static_cast<const GradItem*>(plan_dev);
    const int* block_item = reinterpret_cast<const int*>(items + n_items);
    hipStream_t s = static_cast<hipStream_t>(stream);
    ProfWork work(bytes, flops);
    switch (dtype) {
        case LORA_F16: LORA_LAUNCH(PK_GRAD_PLANNED, (lora_grad_mfma_planned_kernel<half_t>), dim3((unsigned)n_blocks), dim3(256), 0, s, items, block_item); break;
        case LORA_BF16: LORA_LAUNCH(PK_GRAD_PLANNED, (lora_grad_mfma_planned_kernel<bf16_t>), dim3((unsigned)n_blocks), dim3(256), 0, s, items, block_item); break;
        default: return LORA_E_UNSUPPORTED;
    }
    LORA_LAUNCH_CHECK();
    return LORA_OK;
}

extern "C" int lora_linear_bwd_params(const void* dY, const void* X, const float* T, const float* U,
                                      float* gA_part, float* gB_part, int64_t part_stride, int n_blocks,
                                      int64_t M, int K, int N, int r, float scale, int dtype, void* stream) {
    if (M < 0 || K <= 0 || N <= 0 || n_blocks < 1) return LORA_E_BADARG;
    if (r < 1 || r > (K < N ? K : N)) return LORA_E_RANK;
    if (!gA_part || !gB_part) return LORA_E_BADARG;
    if (M > 0 && (!dY || !X || !T || !U)) return LORA_E_BADARG;
    lora_grad_problem pr[2] = {};
    pr[0].S = dY; pr[0].P = T; pr[0].out[0] = gB_part; pr[0].s_stride = N; pr[0].p_stride = r; pr[0].C = N;
    pr[0].out_kn = 0;
    pr[1].S = X;  pr[1].P = U; pr[1].out[0] = gA_part; pr[1].s_stride = K; pr[1].p_stride = r; pr[1].C = K;
    pr[1].out_kn = 1;
    for (int i = 0; i < 2; ++i) {
        pr[i].part_stride = part_stride; pr[i].M = M; pr[i].r = r; pr[i].rg = r; pr[i].scale = scale;
        pr[i].n_blocks = n_blocks;
    }
    return lora_grad_batched(pr, 2, dtype, stream);
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

extern "C" int lora_fold_partials(const int64_t* ranges, int n_ranges, int64_t max_len, const float* partials,
                                  int64_t part_stride, float* grads, int accumulate, void* stream) {
    if (!ranges || !partials || !grads || n_ranges < 1 || max_len < 1) return LORA_E_BADARG;
    if (!aligned16(partials) || !aligned16(grads) || (part_stride & 3)) return LORA_E_ALIGN;
    int64_t bx = (max_len / 4 + 255) / 256;
    if (bx > 64) bx = 64;
    if (bx < 1) bx = 1;
    hipLaunchKernelGGL(fold_partials_kernel, dim3((unsigned)bx, (unsigned)n_ranges), dim3(256), 0,
                       static_cast<hipStream_t>(stream), ranges, partials, part_stride, grads, accumulate);
    LORA_LAUNCH_CHECK();
    return LORA_OK;
}

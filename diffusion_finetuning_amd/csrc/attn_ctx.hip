// Short-context attention core for gfx950:  O = softmax(Q·Kᵀ·scale)·V  per head, for key/value sequences of at
// most 128 tokens — the cross-attention (`attn2`) of every transformer block, whose keys/values are the 77 text
// tokens (the op sandwiched between the to_q/to_k/to_v and to_out LoRA linears; diffusers
// CrossAttention.forward, the caller of the layers wrapped at lora_diffusion/lora.py:137-183).
//
// Why a dedicated kernel: with so few keys the whole K and V of a head fit in LDS, there is ONE key tile (no online
// softmax, no saved log-sum-exp), and the op is a streaming pass over Q (and dO): HBM-bound.  The generic
// flash-attention backward parallelises over key blocks and has one or two of them here, which leaves the chip idle
// (MI355X, 4 × 4096 queries × 77 keys, 8 heads of 40: ~190 µs backward for ~30 MB of traffic).
//
// Layout: Q/O/dO/dQ are [B, Tq, H·d] and K/V/dK/dV [B, Tk, H·d] — exactly what the LoRA linears produce and
// consume, so no head split/merge copies.  Each workgroup (4 waves) owns one (batch, head) and a chunk of query
// rows; a wave works on blocks of 16 query rows with MFMA 16x16x32:
//   Sᵀ = K·Qᵀ  with the K fragment as the FIRST operand, so a lane owns one query row and 4 consecutive keys per
//   fragment: the softmax row reductions are in-lane plus two cross-lane steps, and the probability registers are
//   directly the second operand of the next product (contraction over keys) — P never goes through LDS.  K and V sit in
//   LDS row-major only; the Vᵀ (and, for dQ, Kᵀ) fragments come from the transposing LDS read (ds_read_b64_tr_b16), which
//   delivers a lane's keys in exactly that register order.
// Backward recomputes P from Q and K (cheap: one tile), uses Σ_key P·dP for the softmax correction (so O is not
// needed), writes dQ directly, and accumulates dK/dV over the chunk in registers (MFMA 16x16x16, contraction over the
// 16 query rows; P, dS, Q and dO are re-read transposed from small per-wave LDS tiles for that).  Per-workgroup fp32 partials are
// summed in a fixed order by a small second kernel: deterministic, no atomics.
#include "attn_common.h"

namespace {

constexpr int kCtxSumFrags = 32;  // accumulator fragments per wave and phase of the backward's closing sum (4 × 32 KB of LDS)

template <int KS, int DF, int NKF> struct CtxShape {
    static constexpr int DP = KS * 32;       // head dim padded for the Q·Kᵀ contraction
    static constexpr int DV = DF * 16;       // head dim padded as an MFMA output extent
    static constexpr int NK = NKF * 16;      // keys padded
    // LDS row strides (halfs).  K/V/Q/dO tiles: +32 B — a stride ≡ 32 (mod 64) bytes is what makes the fragment reads
    // (ds_read_b128, serviced in the lane groups {0–3,12–15,20–27}, …) and the transposing reads (two groups of 32 lanes)
    // conflict-free on gfx950's 64 banks; +16 B made every one of them a 2-way conflict (attn_flash.hip, FlashShape).  P/dS
    // tiles: +32 B halves their transposing reads' conflicts (3-way → 2-way: four 8-byte pieces 16 bytes apart per row).
    static constexpr int KROW = DP + 16;
    static constexpr int TROW = NK + 16;
};

// K (or V) of one (batch, head) → registers: thread owns chunks idx = tid + i*256 of the [NK][DP/8] chunk grid
template <typename T, int KS, int DF, int NKF> struct StageRegs {
    using S = CtxShape<KS, DF, NKF>;
    static constexpr int CPR = S::DP / 8;
    static constexpr int N = S::NK * CPR;
    static constexpr int IT = (N + 255) / 256;
    Chunk<T> v[IT];
    __device__ __forceinline__ void load(const T* src, int64_t row_stride, int Tk, int d) {
#pragma unroll
        for (int i = 0; i < IT; ++i) {
            const int idx = threadIdx.x + i * 256;
            const int key = idx / CPR, c = (idx - key * CPR) * 8;
            const bool ok = idx < N && key < Tk && c < d;
            v[i] = load_or_zero<T>(ok ? src + key * row_stride + c : src, ok);
        }
    }
    // row-major [NK][KROW] (operand rows = keys), zero beyond Tk and d
    __device__ __forceinline__ void store_rows(T* dst) const {
#pragma unroll
        for (int i = 0; i < IT; ++i) {
            const int idx = threadIdx.x + i * 256;
            const int key = idx / CPR, c = (idx - key * CPR) * 8;
            if (idx < N) *reinterpret_cast<Chunk<T>*>(dst + key * S::KROW + c) = v[i];
        }
    }
};

// softmax over the keys of one query row held as Sᵀ accumulators (lane = query l15; keys nf*16 + lq*4 + r).
// In: raw scores.  Out: normalised probabilities in place; returns the row max (log2 domain) and 1/sum.
// Round 5 (the kernels are bound by vector issue; this routine was 60 % of the forward loop's instructions): the scale rides in the
// exponent's fma — p = exp2(fma(s, c, −c·max s)), c = scale·log2 e > 0, no separate multiply and subtract pass; the exponent is the bare
// v_exp_f32 (arguments ≤ 0: the library exp2f's denormal-range fix-ups were a compare, an ldexp and two selects per score); keys
// past Tk arrive masked: their scores START at −inf — key_mask() is the initial accumulator of the score chains (a per-lane constant,
// built once per kernel: "row constants as the initial accumulator", here a key constant), so masking costs the loop nothing.
template <int NKF>
__device__ __forceinline__ void key_mask(f32x4 (&k0)[NKF], int lq, int Tk) {
#pragma unroll
    for (int nf = 0; nf < NKF; ++nf)
#pragma unroll
        for (int r = 0; r < 4; ++r) k0[nf][r] = nf * 16 + lq * 4 + r >= Tk ? -INFINITY : 0.f;
}
template <int NKF>
__device__ __forceinline__ void softmax_rows(f32x4 (&s)[NKF], float scale_log2e, float& m, float& inv_l) {
    float mr = -INFINITY;
#pragma unroll
    for (int nf = 0; nf < NKF; ++nf) mr = fmaxf(mr, fmaxf(fmaxf(s[nf][0], s[nf][1]), fmaxf(s[nf][2], s[nf][3])));
    mr = fmaxf(mr, __shfl_xor(mr, 16, 64));
    mr = fmaxf(mr, __shfl_xor(mr, 32, 64));
    m = mr * scale_log2e;
    const float nm = -m;
    float l = 0.f;
#pragma unroll
    for (int nf = 0; nf < NKF; ++nf)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float p = __builtin_amdgcn_exp2f(fmaf(s[nf][r], scale_log2e, nm));  // exp2(−inf) = 0 for masked keys
            s[nf][r] = p;
            l += p;
        }
    l += __shfl_xor(l, 16, 64);
    l += __shfl_xor(l, 32, 64);
    inv_l = 1.f / l;
#pragma unroll
    for (int nf = 0; nf < NKF; ++nf)
#pragma unroll
        for (int r = 0; r < 4; ++r) s[nf][r] *= inv_l;
}

template <typename T, int KS, int DF, int NKF>
__global__ __launch_bounds__(256) void attn_ctx_fwd_kernel(const T* __restrict__ Q, const T* __restrict__ K,
                                                            const T* __restrict__ V, T* __restrict__ O, int Tq, int Tk,
                                                            int H, int d, float scale_log2e, int rq, int chunks,
                                                            int64_t ldk) {
    using S = CtxShape<KS, DF, NKF>;
    using F8 = typename Mma<T>::F8;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    T* Ks = reinterpret_cast<T*>(smem);          // [NK][KROW]
    T* Vs = Ks + S::NK * S::KROW;                 // [NK][KROW]  row-major; Vᵀ fragments by transposing LDS reads

    const int chunk = blockIdx.x % chunks;
    const int bh = blockIdx.x / chunks;
    const int b = bh / H, h = bh - b * H;
    const int64_t HD = (int64_t)H * d;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int l15 = lane & 15, lq = lane >> 4;

    {
        StageRegs<T, KS, DF, NKF> kr, vr;  // both tensors in flight before the first LDS write
        kr.load(K + (int64_t)b * Tk * ldk + h * d, ldk, Tk, d);  // K/V may be column slices of a wider buffer
        vr.load(V + (int64_t)b * Tk * ldk + h * d, ldk, Tk, d);
        kr.store_rows(Ks);
        vr.store_rows(Vs);
    }
    __syncthreads();

    const int row_end = min(Tq, (chunk + 1) * rq);
    const T* Qh = Q + (int64_t)b * Tq * HD + h * d;
    F8 qf[KS];
    {
        const int t = chunk * rq + wave * 16 + l15;
        load_row_frags<T, KS>(Qh + (int64_t)t * HD, Qh, t < row_end, d, lq, qf);
    }
    f32x4 kmask[NKF];
    key_mask<NKF>(kmask, lq, Tk);
    for (int t0 = chunk * rq + wave * 16; t0 < row_end; t0 += 64) {
        const int t = t0 + l15;
        const bool valid = t < row_end;
        F8 qn[KS];  // next block's rows: in flight while this block is computed
        load_row_frags<T, KS>(Qh + (int64_t)(t + 64) * HD, Qh, t + 64 < row_end, d, lq, qn);

        f32x4 s[NKF];
#pragma unroll
        for (int nf = 0; nf < NKF; ++nf) {
            s[nf] = kmask[nf];  // 0, or −inf for keys past Tk
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                const F8 kf = *reinterpret_cast<const F8*>(Ks + (nf * 16 + l15) * S::KROW + ks * 32 + lq * 8);
                s[nf] = Mma<T>::k32(kf, qf[ks], s[nf]);
            }
        }
        float m, inv_l;
        softmax_rows<NKF>(s, scale_log2e, m, inv_l);

        F8 pf[NKF / 2];
#pragma unroll
        for (int kk = 0; kk < NKF / 2; ++kk) pf[kk] = pair_frag<T>(s[2 * kk], s[2 * kk + 1]);

        T* orow = O + ((int64_t)b * Tq + t) * HD + h * d;
#pragma unroll
        for (int df = 0; df < DF; ++df) {
            f32x4 o = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int kk = 0; kk < NKF / 2; ++kk) {
                // lane = head-dim column; slots = keys kk*32 + {0,16} + lq*4 + (0..3), the order of the P registers
                const F8 vf = tr_pair<T>(lds_tr_block(Vs + (kk * 32) * S::KROW + df * 16, S::KROW, lane),
                                         lds_tr_block(Vs + (kk * 32 + 16) * S::KROW + df * 16, S::KROW, lane));
                o = Mma<T>::k32(vf, pf[kk], o);
            }
            const int c = df * 16 + lq * 4;  // the lane owns head-dim values c..c+3 of query row t
            if (valid && c < d) {
                Quad4<T> out;
#pragma unroll
                for (int r = 0; r < 4; ++r) out.v[r] = from_f32<T>(o[r]);
                *reinterpret_cast<Quad4<T>*>(orow + c) = out;
            }
        }
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) qf[ks] = qn[ks];
    }
}

// two workgroups per CU where the accumulators leave room for it (heads of ≤ 48 and ≤ 96 keys: 36 fragments, 247 registers): the row loop is a chain of
// dependent LDS round trips that one wave per SIMD cannot hide
template <int KS, int DF, int NKF> constexpr int ctx_bwd_occupancy() { return (KS == 2 && DF == 3 && NKF == 6) ? 2 : 1; }

template <typename T, int KS, int DF, int NKF>
__global__ __launch_bounds__(256, (ctx_bwd_occupancy<KS, DF, NKF>())) void attn_ctx_bwd_kernel(const T* __restrict__ Q, const T* __restrict__ K,
                                                            const T* __restrict__ V, const T* __restrict__ dO,
                                                            T* __restrict__ dQ, float* __restrict__ part, int Tq,
                                                            int Tk, int H, int d, float scale, float scale_log2e,
                                                            int rq, int chunks, int64_t ldk) {
    using S = CtxShape<KS, DF, NKF>;
    using F8 = typename Mma<T>::F8;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    T* Ks = reinterpret_cast<T*>(smem);          // [NK][KROW]   rows = keys, for Sᵀ and S
    T* Vs = Ks + S::NK * S::KROW;                 // [NK][KROW]   rows = keys, for dPᵀ and dP
    T* Qw = Vs + S::NK * S::KROW + (threadIdx.x >> 6) * 2 * 16 * S::KROW;  // this wave's [16][KROW] copy of its Q rows
    T* Gw = Qw + 16 * S::KROW;                    //                            ... and of its dO rows
    T* Pw = Vs + S::NK * S::KROW + 4 * 2 * 16 * S::KROW + (threadIdx.x >> 6) * 2 * 16 * S::TROW;  // wave's P  [16][TROW]
    T* Sw = Pw + 16 * S::TROW;                                                                      // wave's dS [16][TROW]
    float* red = reinterpret_cast<float*>(smem);  // overlay after the main loop: the waves' accumulator images

    const int chunk = blockIdx.x % chunks;
    const int bh = blockIdx.x / chunks;
    const int b = bh / H, h = bh - b * H;
    const int64_t HD = (int64_t)H * d;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int l15 = lane & 15, lq = lane >> 4;
    const int c0 = blockIdx.y * S::DV;  // head-dim slice whose dQ / dK / dV this workgroup produces

    {
        StageRegs<T, KS, DF, NKF> kr, vr;
        kr.load(K + (int64_t)b * Tk * ldk + h * d, ldk, Tk, d);  // K/V may be column slices of a wider buffer
        vr.load(V + (int64_t)b * Tk * ldk + h * d, ldk, Tk, d);
        kr.store_rows(Ks);
        vr.store_rows(Vs);
    }
    // key mask behind the staging area (the closing sum's overlay may run over it: it is dead by then)
    float* kmask_s = reinterpret_cast<float*>(smem + (2 * S::NK * S::KROW + 4 * 2 * 16 * S::KROW + 4 * 2 * 16 * S::TROW) * 2);
    if (threadIdx.x < S::NK) kmask_s[threadIdx.x] = (int)threadIdx.x >= Tk ? -INFINITY : 0.f;
    __syncthreads();

    f32x4 dk[NKF][DF], dv[NKF][DF];  // lane = head-dim column l15 of fragment df; keys nf*16 + lq*4 + r
#pragma unroll
    for (int nf = 0; nf < NKF; ++nf)
#pragma unroll
        for (int df = 0; df < DF; ++df) dk[nf][df] = dv[nf][df] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int row_end = min(Tq, (chunk + 1) * rq);
    F8 qf[KS], gf[KS];
    {
        const int t = chunk * rq + wave * 16 + l15;
        const int64_t roff = ((int64_t)b * Tq + t) * HD + h * d;
        load_row_frags<T, KS>(Q + roff, Q, t < row_end, d, lq, qf);
        load_row_frags<T, KS>(dO + roff, dO, t < row_end, d, lq, gf);
    }
    for (int t0 = chunk * rq + wave * 16; t0 < row_end; t0 += 64) {
        const int t = t0 + l15;
        const bool valid = t < row_end;
        const int64_t roff = ((int64_t)b * Tq + t) * HD + h * d;
        // this block's rows into the wave's LDS tiles (source of the transposed operands below) ...
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            *reinterpret_cast<F8*>(Qw + l15 * S::KROW + ks * 32 + lq * 8) = qf[ks];
            *reinterpret_cast<F8*>(Gw + l15 * S::KROW + ks * 32 + lq * 8) = gf[ks];
        }
        // ... and the next block's rows into flight
        F8 qn[KS], gn[KS];
        load_row_frags<T, KS>(Q + roff + 64 * HD, Q, t + 64 < row_end, d, lq, qn);
        load_row_frags<T, KS>(dO + roff + 64 * HD, dO, t + 64 < row_end, d, lq, gn);

        // ---- query-per-lane layout: P, dP, the softmax correction, dS → dQ ----------------
        f32x4 s[NKF], dp[NKF];
#pragma unroll
        for (int nf = 0; nf < NKF; ++nf) {
            // 0, or −inf for keys past Tk: the key mask as the initial accumulator (here from LDS — this kernel has no 24
            // registers to spare at two workgroups per CU; one broadcast 16-byte read per fragment)
            s[nf] = *reinterpret_cast<const f32x4*>(kmask_s + nf * 16 + lq * 4);
            dp[nf] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                const int off = (nf * 16 + l15) * S::KROW + ks * 32 + lq * 8;
                s[nf] = Mma<T>::k32(*reinterpret_cast<const F8*>(Ks + off), qf[ks], s[nf]);
                dp[nf] = Mma<T>::k32(*reinterpret_cast<const F8*>(Vs + off), gf[ks], dp[nf]);
            }
        }
        float m, inv_l;
        softmax_rows<NKF>(s, scale_log2e, m, inv_l);
        // P in the permuted key order, one row per query: re-read below with the keys along the lanes
#pragma unroll
        for (int kk = 0; kk < NKF / 2; ++kk)
            *reinterpret_cast<F8*>(Pw + l15 * S::TROW + kk * 32 + lq * 8) = pair_frag<T>(s[2 * kk], s[2 * kk + 1]);
        float corr = 0.f;  // Σ_key P·dP  (= Σ_c dO·O of this query row)
#pragma unroll
        for (int nf = 0; nf < NKF; ++nf)
#pragma unroll
            for (int r = 0; r < 4; ++r) corr += s[nf][r] * dp[nf][r];
        corr += __shfl_xor(corr, 16, 64);
        corr += __shfl_xor(corr, 32, 64);
#pragma unroll
        for (int nf = 0; nf < NKF; ++nf)
#pragma unroll
            for (int r = 0; r < 4; ++r) s[nf][r] = s[nf][r] * (dp[nf][r] - corr) * scale;  // dS
        {
            F8 dsf[NKF / 2];
#pragma unroll
            for (int kk = 0; kk < NKF / 2; ++kk) {
                dsf[kk] = pair_frag<T>(s[2 * kk], s[2 * kk + 1]);
                *reinterpret_cast<F8*>(Sw + l15 * S::TROW + kk * 32 + lq * 8) = dsf[kk];
            }
#pragma unroll
            for (int df = 0; df < DF; ++df) {
                f32x4 g = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int kk = 0; kk < NKF / 2; ++kk) {
                    // Kᵀ fragment of this head-dim slice from the row-major K tile (two transposing block reads)
                    const F8 kf = tr_pair<T>(lds_tr_block(Ks + (kk * 32) * S::KROW + c0 + df * 16, S::KROW, lane),
                                             lds_tr_block(Ks + (kk * 32 + 16) * S::KROW + c0 + df * 16, S::KROW, lane));
                    g = Mma<T>::k32(kf, dsf[kk], g);
                }
                const int c = c0 + df * 16 + lq * 4;
                if (valid && c < d) {
                    Quad4<T> out;
#pragma unroll
                    for (int r = 0; r < 4; ++r) out.v[r] = from_f32<T>(g[r]);
                    *reinterpret_cast<Quad4<T>*>(dQ + roff + c) = out;
                }
            }
        }

        // ---- dK, dV: contraction over the 16 query rows.  Operands with the keys (resp. head-dim columns) along the
        // lanes and 4 query rows per lane, read back transposed from the wave's LDS tiles --------------------------
        // (transposing LDS reads: one ds_read_b64_tr_b16 per operand instead of four 2-byte reads and packing)
        tr4 qT[DF], gT[DF];  // lane = head-dim column c0 + df*16 + l15, the block's rows lq*4 .. +3
#pragma unroll
        for (int df = 0; df < DF; ++df) {
            qT[df] = lds_tr_block(Qw + c0 + df * 16, S::KROW, lane);
            gT[df] = lds_tr_block(Gw + c0 + df * 16, S::KROW, lane);
        }
#pragma unroll
        for (int nf = 0; nf < NKF; ++nf) {
            // lane = key nf*16 + l15, which sits at position key_pos(...) = (nf>>1)*32 + 8·(l15>>2) + 4·(nf&1) + (l15&3) of a P row:
            // lane 4q+p of a group supplies row lq*4 + q, the four positions of key group p
            const int off = (lq * 4 + (l15 >> 2)) * S::TROW + (nf >> 1) * 32 + (l15 & 3) * 8 + (nf & 1) * 4;
            const tr4 pa = lds_tr_at(Pw + off), dsa = lds_tr_at(Sw + off);
#pragma unroll
            for (int df = 0; df < DF; ++df) {
                dv[nf][df] = Mma<T>::k16rr(pa, gT[df], dv[nf][df]);
                dk[nf][df] = Mma<T>::k16rr(dsa, qT[df], dk[nf][df]);
            }
        }
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            qf[ks] = qn[ks];
            gf[ks] = gn[ks];
        }
    }

    // ---- sum the four waves' dK/dV in wave order, one fp32 partial per workgroup --------------------------------------
    // All four waves deposit a run of accumulator fragments in LDS at once (one image per wave, fragment order
    // [fragment][lane][r], whole 16-byte accumulators), then all 256 threads add the four images — ((w0 + w1) + w2) + w3, the
    // order of the former wave-after-wave rounds, so the partials are bit-identical — and store the sums straight to the
    // global partial.  The wave-after-wave form had ONE wave per round reading, adding and re-writing the whole image with
    // no free registers to batch the reads behind (the kernel holds ~500): 4.4 – 8.6 µs per launch, plus 2.3 – 5 µs for
    // copying the finished image out; this form: two or three short phases in which every lane works.
    __syncthreads();  // staging buffers are dead
    constexpr int F = 2 * NKF * DF;                  // fragments per wave: dk then dv
    constexpr int PH = (F + kCtxSumFrags - 1) / kCtxSumFrags;
    constexpr int CH = (F + PH - 1) / PH;            // fragments per phase (≤ kCtxSumFrags)
    f32x4* img = reinterpret_cast<f32x4*>(red);      // [4 waves][CH][64]
    float* out = part + ((int64_t)blockIdx.x * gridDim.y + blockIdx.y) * 2 * S::NK * S::DV;
#pragma unroll
    for (int ph = 0; ph < PH; ++ph) {
#pragma unroll
        for (int fl = 0; fl < CH; ++fl) {
            const int f = ph * CH + fl;  // compile-time after unrolling
            if (f < F) {
                const int ten = f / (NKF * DF), rem = f - ten * (NKF * DF);
                img[(wave * CH + fl) * 64 + lane] = ten == 0 ? dk[rem / DF][rem % DF] : dv[rem / DF][rem % DF];
            }
        }
        __syncthreads();
        const int n_here = (F - ph * CH < CH ? F - ph * CH : CH) * 64;
        for (int e = threadIdx.x; e < n_here; e += 256) {
            const f32x4 v = ((img[e] + img[CH * 64 + e]) + img[2 * CH * 64 + e]) + img[3 * CH * 64 + e];
            *reinterpret_cast<f32x4*>(out + ((int64_t)(ph * CH) * 64 + e) * 4) = v;
        }
        if (ph + 1 < PH) __syncthreads();
    }
}

// dK/dV [B, Tk, H·d] = Σ_chunk partials, summed in chunk order (deterministic), cast to T.  A thread owns one 16-byte
// accumulator of the partial image (fragment order [tensor][nf][df][lane][r]: four consecutive keys of one head-dim column),
// so the partials — the bulk of the traffic — are read as whole coalesced lines; the four outputs go out as 2-byte stores
// into the (small) [B, Tk, H·d] gradients.  (The first form walked the OUTPUT in memory order and gathered 4 bytes out of
// every 16 of the partials: 7 – 9.5 µs per launch against 4 – 6 now.)
template <typename T>
__global__ __launch_bounds__(256) void attn_ctx_reduce_kernel(const float* __restrict__ part, T* __restrict__ dK,
                                                               T* __restrict__ dV, int B, int Tk, int H, int d,
                                                               int chunks, int slices, int NK, int DV, int64_t ld_dk) {
    const int per_img = (NK * DV) >> 2;                 // f32x4 per tensor of one partial
    const int64_t per_grp = 2 * (int64_t)per_img;       // ... per (batch, head, slice)
    const int64_t total = (int64_t)B * H * slices * per_grp;
    const int dfs = DV >> 4;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int64_t grp = i / per_grp;                // (b·H + h)·slices + sl
        const int e = (int)(i - grp * per_grp);
        const int ten = e >= per_img ? 1 : 0, f = e - ten * per_img;
        const int lane = f & 63, frag = f >> 6;
        const int nf = frag / dfs, df = frag - nf * dfs;
        const int sl = (int)(grp % slices);
        const int64_t bh = grp / slices;
        const int h = (int)(bh % H), b = (int)(bh / H);
        const int c = sl * DV + df * 16 + (lane & 15);
        const int key0 = nf * 16 + (lane >> 4) * 4;
        if (c >= d || key0 >= Tk) continue;
        // partial of chunk ch of this group: ((bh·chunks + ch)·slices + sl)·2·NK·DV floats
        const f32x4* p = reinterpret_cast<const f32x4*>(part + ((bh * chunks) * slices + sl) * 2 * (int64_t)NK * DV) + e;
        f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
        for (int ch = 0; ch < chunks; ++ch) acc += p[(int64_t)ch * slices * per_grp];
        T* out = (ten ? dV : dK) + ((int64_t)b * Tk + key0) * ld_dk + (int64_t)h * d + c;
#pragma unroll
        for (int r = 0; r < 4; ++r)
            if (key0 + r < Tk) out[(int64_t)r * ld_dk] = from_f32<T>(acc[r]);
    }
}

struct CtxPlan {
    int ks, df, nkf;  // template selection (df: fragments per workgroup — the whole head forward, one slice backward)
    int slices;       // head-dim slices in backward (dK/dV accumulators of a whole wide head do not fit in registers)
    int chunks, rq;   // query rows per workgroup
};

bool plan_ctx(int B, int Tq, int Tk, int H, int d, bool backward, CtxPlan* pl) {
    if (B < 1 || Tq < 1 || Tk < 1 || H < 1 || d < 8 || (d % 8) != 0 || d > 160 || Tk > 128) return false;
    if (d > 96 && Tk > 96) return false;  // the wide-head backward with 128 keys does not fit in LDS
    pl->slices = 1;
    if (d <= 96) {
        pl->ks = d <= 64 ? 2 : 3;
        pl->df = (d + 15) / 16;
        if (pl->df < 3) pl->df = 3;
    } else {
        pl->ks = 5;
        pl->df = backward ? 5 : 10;
        pl->slices = backward ? 2 : 1;
    }
    pl->nkf = Tk <= 96 ? 6 : 8;
    // enough workgroups to fill 256 CUs, but few chunks: every chunk re-stages K/V (and writes a partial in backward)
    // backward: one workgroup per CU (the accumulators take most of the register file), two where the kernel is built for it
    // (ctx_bwd_occupancy).  More chunks than that lose: every chunk re-stages K/V and writes a dK/dV partial
    // (profiles/r04_ctx_attention_workgroup_sweep.log)
    const int bwd_wgs = (pl->ks == 2 && pl->df == 3 && pl->nkf == 6) ? 512 : 256;
    const int want = (backward ? bwd_wgs : 512) / (B * H * pl->slices);
    int chunks = want < 1 ? 1 : want;
    const int max_chunks = (Tq + 63) / 64;
    if (chunks > max_chunks) chunks = max_chunks;
    int rq = (Tq + chunks - 1) / chunks;
    rq = (rq + 63) / 64 * 64;  // whole rounds of the four waves
    pl->rq = rq;
    pl->chunks = (Tq + rq - 1) / rq;
    return true;
}

template <int KS, int DF, int NKF> constexpr int fwd_lds() {
    using S = CtxShape<KS, DF, NKF>;
    return 2 * S::NK * S::KROW * 2;
}
template <int KS, int DF, int NKF> constexpr int bwd_lds() {
    using S = CtxShape<KS, DF, NKF>;
    constexpr int stage = (2 * S::NK * S::KROW + 4 * 2 * 16 * S::KROW + 4 * 2 * 16 * S::TROW) * 2 + S::NK * 4;  // + key mask
    constexpr int F = 2 * NKF * DF, PH = (F + kCtxSumFrags - 1) / kCtxSumFrags, CH = (F + PH - 1) / PH;
    constexpr int red = 4 * CH * 64 * 16;  // four waves' images of one phase of the closing sum
    return stage > red ? stage : red;
}

struct CtxArgs {
    const void *Q, *K, *V, *dO;
    void *O, *dQ, *dK, *dV;
    float* part;
    int B, Tq, Tk, H, d;
    float scale;
    int64_t ldk, ld_dk;  // row strides (elements) of K/V and of dK/dV
};

template <typename T, int KS, int DF, int NKF, bool BWD>
int launch_ctx(const CtxArgs& a, const CtxPlan& pl, hipStream_t stream) {
    const float l2e = a.scale * 1.4426950408889634f;
    const dim3 grid((unsigned)(a.B * a.H * pl.chunks));
    if constexpr (!BWD) {
        constexpr int lds = fwd_lds<KS, DF, NKF>();
        auto kern = attn_ctx_fwd_kernel<T, KS, DF, NKF>;
        if (lds > 48 * 1024) {
            static const hipError_t attr = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                                               hipFuncAttributeMaxDynamicSharedMemorySize, lds);
            if (attr != hipSuccess) return LORA_E_LAUNCH;
        }
        {
            const double bh = (double)a.B * a.H, e = sizeof(T);
            lora_prof_set_work(e * bh * a.d * (2.0 * a.Tq + 2.0 * a.Tk), 4.0 * bh * a.Tq * (double)a.Tk * a.d);
        }
        LORA_LAUNCH(PK_CTX_FWD, kern, grid, dim3(256), lds, stream, static_cast<const T*>(a.Q), static_cast<const T*>(a.K),
                    static_cast<const T*>(a.V), static_cast<T*>(a.O), a.Tq, a.Tk, a.H, a.d, l2e, pl.rq, pl.chunks, a.ldk);
        lora_prof_set_work(0.0, 0.0);
        LORA_LAUNCH_CHECK();
        return LORA_OK;
    } else {
    constexpr int lds = bwd_lds<KS, DF, NKF>();
    auto kern = attn_ctx_bwd_kernel<T, KS, DF, NKF>;
    if (lds > 48 * 1024) {
        static const hipError_t attr = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                                           hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        if (attr != hipSuccess) return LORA_E_LAUNCH;
    }
    {   // Q, dO read, dQ written (3·Tq rows); K, V read, dK, dV written (4·Tk rows); S, dP, dQ, dK, dV: 10·B·H·Tq·Tk·d
        const double bh = (double)a.B * a.H, e = sizeof(T);
        lora_prof_set_work(e * bh * a.d * (3.0 * a.Tq + 4.0 * a.Tk), 10.0 * bh * a.Tq * (double)a.Tk * a.d);
    }
    LORA_LAUNCH(PK_CTX_BWD, kern, dim3(grid.x, (unsigned)pl.slices), dim3(256), lds, stream, static_cast<const T*>(a.Q),
                static_cast<const T*>(a.K), static_cast<const T*>(a.V), static_cast<const T*>(a.dO),
                static_cast<T*>(a.dQ), a.part, a.Tq,
                a.Tk, a.H, a.d, a.scale, l2e, pl.rq, pl.chunks, a.ldk);
    lora_prof_set_work(0.0, 0.0);
    LORA_LAUNCH_CHECK();
    const int64_t total = (int64_t)a.B * a.H * pl.slices * 2 * (NKF * 16) * (DF * 16) / 4;  // 16-byte accumulators of one partial set
    const unsigned blocks = (unsigned)((total + 255) / 256 > 2048 ? 2048 : (total + 255) / 256);
    // (the ordered sum of the chunk partials: its time is charged to the backward's kind, it carries no algorithmic bytes)
    LORA_LAUNCH(PK_CTX_BWD, attn_ctx_reduce_kernel<T>, dim3(blocks), dim3(256), 0, stream, a.part, static_cast<T*>(a.dK),
                static_cast<T*>(a.dV), a.B, a.Tk, a.H, a.d, pl.chunks, pl.slices, NKF * 16, DF * 16, a.ld_dk);
    LORA_LAUNCH_CHECK();
    return LORA_OK;
    }
}

template <typename T>
int dispatch_ctx(const CtxArgs& a, const CtxPlan& pl, bool backward, hipStream_t stream) {
#define CTX_CASE(KS_, DF_, NKF_)                                                                 \
    if (pl.ks == KS_ && pl.df == DF_ && pl.nkf == NKF_)                                              \
        return backward ? launch_ctx<T, KS_, DF_, NKF_, true>(a, pl, stream) : launch_ctx<T, KS_, DF_, NKF_, false>(a, pl, stream);
#define CTX_FWD_ONLY(KS_, DF_, NKF_) \
    if (!backward && pl.ks == KS_ && pl.df == DF_ && pl.nkf == NKF_) return launch_ctx<T, KS_, DF_, NKF_, false>(a, pl, stream);
#define CTX_BWD_ONLY(KS_, DF_, NKF_) \
    if (backward && pl.ks == KS_ && pl.df == DF_ && pl.nkf == NKF_) return launch_ctx<T, KS_, DF_, NKF_, true>(a, pl, stream);
    CTX_CASE(2, 3, 6) CTX_CASE(2, 4, 6) CTX_CASE(3, 5, 6) CTX_CASE(3, 6, 6)
    CTX_CASE(2, 3, 8) CTX_CASE(2, 4, 8) CTX_CASE(3, 5, 8) CTX_CASE(3, 6, 8)
    CTX_FWD_ONLY(5, 10, 6) CTX_BWD_ONLY(5, 5, 6)  // heads of 104 … 160 (SD: 160 at the two coarsest levels)
#undef CTX_CASE
#undef CTX_FWD_ONLY
#undef CTX_BWD_ONLY
    return LORA_E_BADARG;
}

int run_ctx(const CtxArgs& a, bool backward, int dtype, hipStream_t stream) {
    CtxPlan pl;
    if (!plan_ctx(a.B, a.Tq, a.Tk, a.H, a.d, backward, &pl)) return LORA_E_BADARG;
    switch (dtype) {
        case LORA_F16: return dispatch_ctx<half_t>(a, pl, backward, stream);
        case LORA_BF16: return dispatch_ctx<bf16_t>(a, pl, backward, stream);
        default: return LORA_E_BADARG;  // fp32 tensors stay on the caller's generic attention
    }
}

}  // namespace

extern "C" int attn_ctx_supported(int B, int Tq, int Tk, int H, int d, int dtype) {
    CtxPlan pl;
    return (dtype == LORA_F16 || dtype == LORA_BF16) && plan_ctx(B, Tq, Tk, H, d, false, &pl) ? 1 : 0;
}

extern "C" int64_t attn_ctx_bwd_workspace_bytes(int B, int Tq, int Tk, int H, int d) {
    CtxPlan pl;
    if (!plan_ctx(B, Tq, Tk, H, d, true, &pl)) return -1;
    return (int64_t)B * H * pl.chunks * pl.slices * 2 * (pl.nkf * 16) * (pl.df * 16) * 4;
}

extern "C" int attn_ctx_fwd_strided(const void* Q, const void* K, const void* V, void* O, int64_t ldk, int B, int Tq,
                                    int Tk, int H, int d, float scale, int dtype, void* stream) {
    if (!Q || !K || !V || !O) return LORA_E_BADARG;
    if (!aligned16(Q) || !aligned16(K) || !aligned16(V) || !aligned16(O)) return LORA_E_BADARG;
    if (ldk < (int64_t)H * d || (ldk % 8) != 0) return LORA_E_BADARG;
    CtxArgs a{};
    a.Q = Q; a.K = K; a.V = V; a.O = O; a.B = B; a.Tq = Tq; a.Tk = Tk; a.H = H; a.d = d; a.scale = scale;
    a.ldk = ldk; a.ld_dk = (int64_t)H * d;
    return run_ctx(a, false, dtype, static_cast<hipStream_t>(stream));
}

extern "C" int attn_ctx_fwd(const void* Q, const void* K, const void* V, void* O, int B, int Tq, int Tk, int H, int d,
                            float scale, int dtype, void* stream) {
    return attn_ctx_fwd_strided(Q, K, V, O, (int64_t)H * d, B, Tq, Tk, H, d, scale, dtype, stream);
}

extern "C" int attn_ctx_bwd_strided(const void* Q, const void* K, const void* V, const void* dO, void* dQ, void* dK,
                                    void* dV, void* workspace, int64_t ldk, int64_t ld_dk, int B, int Tq, int Tk, int H,
                                    int d, float scale, int dtype, void* stream) {
    if (!Q || !K || !V || !dO || !dQ || !dK || !dV || !workspace) return LORA_E_BADARG;
    if (!aligned16(Q) || !aligned16(K) || !aligned16(V) || !aligned16(dO) || !aligned16(dQ) || !aligned16(workspace))
        return LORA_E_BADARG;
    if (ldk < (int64_t)H * d || (ldk % 8) != 0 || ld_dk < (int64_t)H * d) return LORA_E_BADARG;
    CtxArgs a{};
    a.Q = Q; a.K = K; a.V = V; a.dO = dO; a.dQ = dQ; a.dK = dK; a.dV = dV; a.part = static_cast<float*>(workspace);
    a.B = B; a.Tq = Tq; a.Tk = Tk; a.H = H; a.d = d; a.scale = scale; a.ldk = ldk; a.ld_dk = ld_dk;
    return run_ctx(a, true, dtype, static_cast<hipStream_t>(stream));
}

extern "C" int attn_ctx_bwd(const void* Q, const void* K, const void* V, const void* dO, void* dQ, void* dK, void* dV,
                            void* workspace, int B, int Tq, int Tk, int H, int d, float scale, int dtype,
                            void* stream) {
    return attn_ctx_bwd_strided(Q, K, V, dO, dQ, dK, dV, workspace, (int64_t)H * d, (int64_t)H * d, B, Tq, Tk, H, d,
                                scale, dtype, stream);
}

// Long-context attention core for gfx950:  O = softmax(Q·Kᵀ·scale)·V  per head with an online softmax over key
// tiles — the self-attention (`attn1`) of every transformer block, 4096 / 1024 / 256 / 64 tokens in the SD UNets
// (the op between the to_q/to_k/to_v and to_out LoRA linears; diffusers CrossAttention.forward).
//
// Same conventions as attn_ctx.hip: tensors stay [B, T, H·d] (what the LoRA linears produce and consume), MFMA 16x16x32
// with the key fragment as the FIRST operand so that a lane owns one query row (softmax statistics are per lane, two
// cross-lane steps per reduction) and the probability registers feed the P·V product directly; K and V tiles are staged
// row-major and the products that contract over the keys fetch their Vᵀ / Kᵀ fragments with the transposing LDS read
// (ds_read_b64_tr_b16: two block reads deliver a lane's 8 keys in exactly that register order — no transposed copy, no
// 2-byte scatter writes: d = 160 forward 22 → 13 µs, d = 80 41 → 34 µs).  New here: 64-key tiles double-buffered in LDS (global → registers
// at the top of an iteration, registers → LDS after the tile's arithmetic), running max / sum with accumulator
// rescaling, several 16-row blocks per wave so that every K / V fragment read from LDS is used RB times, and the
// row-wise log-sum-exp (base 2, of the scaled scores) written out for the backward kernels.
#include "attn_flash_common.h"


namespace {

// ONES (head dims that leave padding rows in the Vᵀ tile, d % 16 == 8: the 40-wide heads of SD1.5's widest level): row d of
// Vᵀ is set to ones once, so the P·V product itself accumulates Σp — on the matrix pipe, in fp32, rescaled together with
// the output — and the 64 VALU adds and two cross-lane reductions per row block and tile disappear.
template <typename T, int KS, int DF, int RB, bool ONES>
__global__ __launch_bounds__(256, 2) void attn_flash_fwd_kernel(const T* __restrict__ Q, const T* __restrict__ K,
                                                              const T* __restrict__ V, T* __restrict__ O,
                                                              float* __restrict__ LSE, int Tq, int Tk, int H, int d,
                                                              float scale_log2e, int64_t ldq) {
    using S = FlashShape<KS, DF>;
    using F8 = typename Mma<T>::F8;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    T* Ks = reinterpret_cast<T*>(smem);            // [2][kTile][KROW]
    T* Vs = Ks + 2 * S::K_HALFS;                    // [2][kTile][KROW]

    int bx, bh;
    xcd_block(bx, bh);
    const int b = bh / H, h = bh - b * H;
    const int64_t HD = (int64_t)H * d;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int l15 = lane & 15, lq = lane >> 4;
    // Q, K, V share the row stride ldq (they may be the three column slices of one grouped projection's output)
    const T* Qh = Q + (int64_t)b * Tq * ldq + h * d;
    const T* Kh = K + (int64_t)b * Tk * ldq + h * d;
    const T* Vh = V + (int64_t)b * Tk * ldq + h * d;

    const int row0 = bx * (64 * RB) + wave * (16 * RB);  // first query row of this wave
    F8 qf[RB][KS];
#pragma unroll
    for (int rb = 0; rb < RB; ++rb) {
        const int t = row0 + rb * 16 + l15;
        load_row_frags<T, KS>(Qh + (int64_t)t * ldq, Qh, t < Tq, d, lq, qf[rb]);
        prescale_frags<T, KS>(qf[rb], scale_log2e);  // scores come out of the MFMA in the exp2 domain
    }
    f32x4 o[RB][DF];
    float m[RB], l[RB];  // m: reference maximum of the row (exp2 domain), −inf until the first tile has set it
#pragma unroll
    for (int rb = 0; rb < RB; ++rb) {
        m[rb] = -INFINITY;
        l[rb] = 0.f;
#pragma unroll
        for (int df = 0; df < DF; ++df) o[rb][df] = f32x4{0.f, 0.f, 0.f, 0.f};
    }

    lds_zero(smem, flash_fwd_lds_bytes<KS, DF>());
    __syncthreads();
    if constexpr (ONES) {  // never re-staged: the stage writes rows < d only
        if (threadIdx.x < 2 * kTile) Vs[(threadIdx.x >> 6) * S::K_HALFS + (threadIdx.x & 63) * S::KROW + d] = from_f32<T>(1.f);
    }
    TileStage<T, KS, DF> stage;
    stage.init(d, ldq, ldq);
    stage.load(Kh, Vh, Tk);
    stage.store_a_rows(Ks);
    stage.store_b_rows(Vs);
    __syncthreads();
    const int n_tiles = (Tk + kTile - 1) / kTile;
    // Vᵀ fragment (df, kk) for the P·V product: lane = head-dim column df*16 + l15, its 8 contraction slots are the keys
    // kk*32 + {0,16} + lq*4 + (0..3) — the order in which two neighbouring score fragments sit in a lane's registers —
    // i.e. two transposing block reads of the row-major V tile
    auto vt_frag = [&](const T* Vc, int df, int kk) {
        return tr_pair<T>(lds_tr_block(Vc + (kk * 32) * S::KROW + df * 16, S::KROW, lane),
                          lds_tr_block(Vc + (kk * 32 + 16) * S::KROW + df * 16, S::KROW, lane));
    };
    // one key tile; RAGGED (only ever the last tile) is a compile-time flag so that full tiles carry no masking code
    auto tile = [&](int kt, auto ragged_tag) {
        constexpr bool RAGGED = decltype(ragged_tag)::value;
        const int cur = kt & 1;
        const T* Kc = Ks + cur * S::K_HALFS;
        const T* Vc = Vs + cur * S::K_HALFS;
        if (kt + 1 < n_tiles) stage.load(Kh + (int64_t)(kt + 1) * kTile * ldq, Vh + (int64_t)(kt + 1) * kTile * ldq, Tk - (kt + 1) * kTile);  // in flight during this tile's work

        // ---- online softmax of one 16-row block, one query row per lane.  `sc` holds s·c − m (the MFMA chain started at −m:
        // see scores()), so a probability is ONE v_exp_f32 of an accumulator register.  Lazy rescaling: the reference maximum
        // moves only when a row outgrows it by more than 2^8 (or has none yet) — after the first tiles almost never — so the
        // correction exp2, the DF·4 accumulator multiplies and the per-score subtraction run on very few tiles.  Probabilities
        // reach at most 2^8 (exact in fp32, far inside the 16-bit operand's range); O = o / l and LSE = m + log2 l do not depend
        // on which m was used.  Only the rows that outgrew their reference move it (the others shift by 0): a row's result
        // never depends on which other rows share its wave.
        auto softmax_block = [&](int rb, f32x4 (&sc)[kNKF], int kt_) {
            if constexpr (RAGGED) {  // keys past Tk never win the max and get probability 0
                const int key_base = kt_ * kTile + lq * 4;
    #pragma unroll
                for (int nf = 0; nf < kNKF; ++nf)
    #pragma unroll
                    for (int r = 0; r < 4; ++r)
                        if (key_base + nf * 16 + r >= Tk) sc[nf][r] = -INFINITY;
            }
            const float mt = quad_max(max16(sc));  // relative to the reference maximum
            const bool fresh = m[rb] == -INFINITY;
            const bool move = fresh || mt > 8.f;
            float sum = 0.f;
            if (__any(move)) {
                const float base = fresh ? 0.f : m[rb];
                const float m_new = move ? base + mt : m[rb];
                const float shift = m_new - base;                       // 0 for the rows that keep their reference
                const float alpha = fresh ? 1.f : fast_exp2(-shift);    // (o and l of a fresh row are still zero)
                m[rb] = m_new;
                l[rb] *= alpha;
    #pragma unroll
                for (int df = 0; df < DF; ++df) o[rb][df] *= alpha;
    #pragma unroll
                for (int nf = 0; nf < kNKF; ++nf)
    #pragma unroll
                    for (int r = 0; r < 4; ++r) sc[nf][r] -= shift;  // (only on the tiles where some row of the wave moved)
            }
    #pragma unroll
            for (int nf = 0; nf < kNKF; ++nf)
    #pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float p = fast_exp2(sc[nf][r]);
                    sc[nf][r] = p;
                    if constexpr (!ONES) sum += p;
                }
            if constexpr (!ONES) l[rb] += quad_sum(sum);
        };

        // Narrow heads (the fragments of a whole tile fit the register file): software pipeline, below.  Wide heads keep the
        // phase-by-phase form — resident fragments and the look-ahead accumulators would spill there.
        constexpr bool RES = 4 * KS + 2 * DF <= 22 && RB * DF <= 12;
        if constexpr (RES) {
            // ---- software pipeline over the RB row blocks: the score MFMAs of block rb+1 are issued BEFORE the softmax of
            // block rb, so that block's exp2 / max / convert VALU work runs while the matrix pipe chews on the next block's
            // scores (and on this block's P·V behind them) instead of the two pipes taking turns.  The K and Vᵀ fragments of
            // the tile are read from LDS once and stay in registers for all row blocks.
            F8 kf[RES ? kNKF : 1][RES ? KS : 1], vf[RES ? DF : 1][RES ? kNKF / 2 : 1];
            auto kfrag = [&](int nf, int ks) {
                if constexpr (RES) return kf[nf][ks];
                else return *reinterpret_cast<const F8*>(Kc + (nf * 16 + l15) * S::KROW + ks * 32 + lq * 8);
            };
            auto vfrag = [&](int df, int kk) {
                if constexpr (RES) return vf[df][kk];
                else return vt_frag(Vc, df, kk);
            };
            if constexpr (RES) {
    #pragma unroll
                for (int nf = 0; nf < kNKF; ++nf)
    #pragma unroll
                    for (int ks = 0; ks < KS; ++ks)
                        kf[nf][ks] = *reinterpret_cast<const F8*>(Kc + (nf * 16 + l15) * S::KROW + ks * 32 + lq * 8);
    #pragma unroll
                for (int df = 0; df < DF; ++df)
    #pragma unroll
                    for (int kk = 0; kk < kNKF / 2; ++kk)
                        vf[df][kk] = vt_frag(Vc, df, kk);
            }
            // Sᵀ = K·Qᵀ − m for one 16-row block: the accumulators START at −m (0 while the row has no reference maximum yet)
            auto scores = [&](int rb, f32x4 (&sc)[kNKF]) {
                const float base = m[rb] == -INFINITY ? 0.f : -m[rb];
    #pragma unroll
                for (int nf = 0; nf < kNKF; ++nf) {
                    sc[nf] = f32x4{base, base, base, base};
    #pragma unroll
                    for (int ks = 0; ks < KS; ++ks) sc[nf] = Mma<T>::k32(kfrag(nf, ks), qf[rb][ks], sc[nf]);
                }
            };
            f32x4 sbuf[2][kNKF];  // ping-pong: block rb lives in sbuf[rb & 1] (rb is a compile-time constant below)
            scores(0, sbuf[0]);
    #pragma unroll
            for (int rb = 0; rb < RB; ++rb) {
                f32x4 (&s_cur)[kNKF] = sbuf[rb & 1];
                if (rb + 1 < RB) scores(rb + 1, sbuf[(rb + 1) & 1]);
                softmax_block(rb, s_cur, kt);
                // ---- Oᵀ += Vᵀ·Pᵀ for this block ----------------------------------------------------------------------
    #pragma unroll
                for (int kk = 0; kk < kNKF / 2; ++kk) {
                    const F8 pf = pair_frag<T>(s_cur[2 * kk], s_cur[2 * kk + 1]);
    #pragma unroll
                    for (int df = 0; df < DF; ++df) o[rb][df] = Mma<T>::k32(vfrag(df, kk), pf, o[rb][df]);
                }
            }
        } else {
            // ---- Sᵀ = K·Qᵀ − m: every K fragment read once, used for all RB row blocks; the chains start at −m -----------------
            f32x4 s[RB][kNKF];
    #pragma unroll
            for (int rb = 0; rb < RB; ++rb) {
                const float base = m[rb] == -INFINITY ? 0.f : -m[rb];
    #pragma unroll
                for (int nf = 0; nf < kNKF; ++nf) s[rb][nf] = f32x4{base, base, base, base};
            }
    #pragma unroll
            for (int nf = 0; nf < kNKF; ++nf)
    #pragma unroll
                for (int ks = 0; ks < KS; ++ks) {
                    const F8 kf = *reinterpret_cast<const F8*>(Kc + (nf * 16 + l15) * S::KROW + ks * 32 + lq * 8);
    #pragma unroll
                    for (int rb = 0; rb < RB; ++rb) s[rb][nf] = Mma<T>::k32(kf, qf[rb][ks], s[rb][nf]);
                }
            F8 pf[RB][kNKF / 2];
    #pragma unroll
            for (int rb = 0; rb < RB; ++rb) {
                softmax_block(rb, s[rb], kt);
    #pragma unroll
                for (int kk = 0; kk < kNKF / 2; ++kk) pf[rb][kk] = pair_frag<T>(s[rb][2 * kk], s[rb][2 * kk + 1]);
            }
            // ---- Oᵀ += Vᵀ·Pᵀ -------------------------------------------------------------------------------------
    #pragma unroll
            for (int df = 0; df < DF; ++df)
    #pragma unroll
                for (int kk = 0; kk < kNKF / 2; ++kk) {
                    const F8 vf = vt_frag(Vc, df, kk);
    #pragma unroll
                    for (int rb = 0; rb < RB; ++rb) o[rb][df] = Mma<T>::k32(vf, pf[rb][kk], o[rb][df]);
                }
        }
        if (kt + 1 < n_tiles) {
            stage.store_a_rows(Ks + (cur ^ 1) * S::K_HALFS);
            stage.store_b_rows(Vs + (cur ^ 1) * S::K_HALFS);
        }
        __syncthreads();
    };
    const int n_full = Tk / kTile;
    for (int kt = 0; kt < n_full; ++kt) tile(kt, std::false_type{});
    if (n_full < n_tiles) tile(n_full, std::true_type{});

    if constexpr (ONES) {  // Σp sits in output column d: fragment DF-1, lane group lq = 2, register 0 (d % 16 == 8)
#pragma unroll
        for (int rb = 0; rb < RB; ++rb) l[rb] = __shfl(o[rb][DF - 1][0], l15 + 32, 64);
    }
#pragma unroll
    for (int rb = 0; rb < RB; ++rb) {
        const int t = row0 + rb * 16 + l15;
        if (t >= Tq) continue;
        const float inv = 1.f / l[rb];
        T* orow = O + ((int64_t)b * Tq + t) * HD + h * d;
#pragma unroll
        for (int df = 0; df < DF; ++df) {
            const int c = df * 16 + lq * 4;
            if (c < d) {
                Quad4<T> out;
#pragma unroll
                for (int r = 0; r < 4; ++r) out.v[r] = from_f32<T>(o[rb][df][r] * inv);
                *reinterpret_cast<Quad4<T>*>(orow + c) = out;
            }
        }
        if (lq == 0 && LSE != nullptr) LSE[(int64_t)bh * Tq + t] = m[rb] + log2f(l[rb]);
    }
}

// ---------------------------------------------------------------------------------------------------------
// Backward.  Two launches, no atomics, every output element written by exactly one workgroup:
//   dQ     : query-owned like the forward — also computes Δ[b,h,q] = Σ_c dO·O (the softmax correction) for its rows from
//            the dO fragments it holds and writes it out for the second launch.  P is rebuilt from the saved log-sum-exp,
//            dPᵀ = V·dOᵀ has the layout of Sᵀ, dS = P∘(dP − Δ)·scale feeds dQᵀ += Kᵀ·dSᵀ straight from registers (K staged
//            a second time transposed)
//   dK, dV : key-owned — a wave keeps the K and V fragments of its keys in registers and walks over ALL query tiles
//            (Q, dO, LSE, Δ staged in LDS, shared by the four waves); S = Q·Kᵀ is taken with the query fragment FIRST,
//            so its accumulators already are MFMA 16x16x16 operands with the keys along the lanes and four query rows
//            per lane — exactly what the contractions over the query rows (dV = Pᵀ·dO, dK = dSᵀ·Q) need.

template <typename T, int KS, int DF, int RB>
__global__ __launch_bounds__(256, 2) void attn_flash_dq_kernel(const T* __restrict__ Q, const T* __restrict__ K,
                                                             const T* __restrict__ V, const T* __restrict__ O,
                                                             const T* __restrict__ dO, const float* __restrict__ LSE,
                                                             float* __restrict__ Delta, T* __restrict__ dQ, int Tq,
                                                             int Tk, int H, int d, float scale, float scale_log2e,
                                                             int64_t ldq, int64_t ld_dq) {
    using S = FlashShape<KS, DF>;
    using F8 = typename Mma<T>::F8;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    T* Ks = reinterpret_cast<T*>(smem);   // [2][kTile][KROW]
    T* Vs = Ks + 2 * S::K_HALFS;           // [2][kTile][KROW]

    int bx, bh;
    xcd_block(bx, bh);
    const int b = bh / H, h = bh - b * H;
    const int64_t HD = (int64_t)H * d;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int l15 = lane & 15, lq = lane >> 4;
    const T* Kh = K + (int64_t)b * Tk * ldq + h * d;
    const T* Vh = V + (int64_t)b * Tk * ldq + h * d;
    const int row0 = bx * (64 * RB) + wave * (16 * RB);

    F8 qf[RB][KS], gf[RB][KS];
    float lse[RB], delta[RB];
    f32x4 acc[RB][DF];
#pragma unroll
    for (int rb = 0; rb < RB; ++rb) {
        const int t = row0 + rb * 16 + l15;
        const bool valid = t < Tq;
        const int64_t roff = ((int64_t)b * Tq + t) * HD + h * d;
        load_row_frags<T, KS>(Q + ((int64_t)b * Tq + t) * ldq + h * d, Q, valid, d, lq, qf[rb]);
        prescale_frags<T, KS>(qf[rb], scale_log2e);  // (Q only feeds the scores here: S comes out in the exp2 domain)
        load_row_frags<T, KS>(dO + roff, dO, valid, d, lq, gf[rb]);
        lse[rb] = valid ? LSE[(int64_t)bh * Tq + t] : INFINITY;  // +inf: probability 0 for rows past the end
        // Δ = Σ_c dO·O of the row (the softmax correction), computed here from the dO fragments the kernel holds anyway and
        // written out for the dK/dV kernel, which is launched AFTER this one: no separate pass over O and dO
        F8 of[KS];
        load_row_frags<T, KS>(O + roff, O, valid, d, lq, of);
        float dsum = 0.f;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks)
#pragma unroll
            for (int e = 0; e < 8; ++e) dsum = fmaf(to_f32<T>(of[ks][e]), to_f32<T>(gf[rb][ks][e]), dsum);
        delta[rb] = quad_sum(dsum);
        if (valid && lq == 0) Delta[(int64_t)bh * Tq + t] = delta[rb];
#pragma unroll
        for (int df = 0; df < DF; ++df) acc[rb][df] = f32x4{0.f, 0.f, 0.f, 0.f};
    }

    lds_zero(smem, flash_dq_lds_bytes<KS, DF>());
    __syncthreads();
    TileStage<T, KS, DF> stage;
    stage.init(d, ldq, ldq);
    stage.load(Kh, Vh, Tk);
    stage.store_a_rows(Ks);
    stage.store_b_rows(Vs);
    __syncthreads();
    const int n_tiles = (Tk + kTile - 1) / kTile;
    auto tile = [&](int kt, auto ragged_tag) {
        constexpr bool RAGGED = decltype(ragged_tag)::value;
        const int cur = kt & 1;
        const T* Kc = Ks + cur * S::K_HALFS;
        const T* Vc = Vs + cur * S::K_HALFS;
        if (kt + 1 < n_tiles) stage.load(Kh + (int64_t)(kt + 1) * kTile * ldq, Vh + (int64_t)(kt + 1) * kTile * ldq, Tk - (kt + 1) * kTile);

        // row constants as the initial accumulators: S′ = c·q·k − LSE and dP′ = dO·v − Δ leave the matrix pipe ready, so that
        // p = exp2(S′) and dS = p·dP′ cost one v_exp_f32 and one multiply per score (no fma, no subtraction)
        f32x4 s[RB][kNKF], dp[RB][kNKF];
#pragma unroll
        for (int rb = 0; rb < RB; ++rb)
#pragma unroll
            for (int nf = 0; nf < kNKF; ++nf) {
                s[rb][nf] = f32x4{-lse[rb], -lse[rb], -lse[rb], -lse[rb]};
                dp[rb][nf] = f32x4{-delta[rb], -delta[rb], -delta[rb], -delta[rb]};
            }
#pragma unroll
        for (int nf = 0; nf < kNKF; ++nf)
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                const int off = (nf * 16 + l15) * S::KROW + ks * 32 + lq * 8;
                const F8 kf = *reinterpret_cast<const F8*>(Kc + off);
                const F8 vf = *reinterpret_cast<const F8*>(Vc + off);
#pragma unroll
                for (int rb = 0; rb < RB; ++rb) {
                    s[rb][nf] = Mma<T>::k32(kf, qf[rb][ks], s[rb][nf]);
                    dp[rb][nf] = Mma<T>::k32(vf, gf[rb][ks], dp[rb][nf]);
                }
            }
        const int key_base = kt * kTile + lq * 4;
        F8 dsf[RB][kNKF / 2];
#pragma unroll
        for (int rb = 0; rb < RB; ++rb) {
#pragma unroll
            for (int nf = 0; nf < kNKF; ++nf)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float p = fast_exp2(s[rb][nf][r]);
                    if constexpr (RAGGED) {
                        if (key_base + nf * 16 + r >= Tk) p = 0.f;
                    }
                    s[rb][nf][r] = p * dp[rb][nf][r];  // the 1/√d factor is applied once, on the way out
                }
#pragma unroll
            for (int kk = 0; kk < kNKF / 2; ++kk) dsf[rb][kk] = pair_frag<T>(s[rb][2 * kk], s[rb][2 * kk + 1]);
        }
#pragma unroll
        for (int df = 0; df < DF; ++df)
#pragma unroll
            for (int kk = 0; kk < kNKF / 2; ++kk) {
                // Kᵀ fragment: lane = head-dim column, slots = keys kk*32 + {0,16} + lq*4 + (0..3): two transposing block reads
                const F8 ktf = tr_pair<T>(lds_tr_block(Kc + (kk * 32) * S::KROW + df * 16, S::KROW, lane),
                                          lds_tr_block(Kc + (kk * 32 + 16) * S::KROW + df * 16, S::KROW, lane));
#pragma unroll
                for (int rb = 0; rb < RB; ++rb) acc[rb][df] = Mma<T>::k32(ktf, dsf[rb][kk], acc[rb][df]);
            }
        if (kt + 1 < n_tiles) {
            stage.store_a_rows(Ks + (cur ^ 1) * S::K_HALFS);
            stage.store_b_rows(Vs + (cur ^ 1) * S::K_HALFS);
        }
        __syncthreads();
    };
    const int n_full = Tk / kTile;
    for (int kt = 0; kt < n_full; ++kt) tile(kt, std::false_type{});
    if (n_full < n_tiles) tile(n_full, std::true_type{});
#pragma unroll
    for (int rb = 0; rb < RB; ++rb) {
        const int t = row0 + rb * 16 + l15;
        if (t >= Tq) continue;
        T* grow = dQ + ((int64_t)b * Tq + t) * ld_dq + h * d;
#pragma unroll
        for (int df = 0; df < DF; ++df) {
            const int c = df * 16 + lq * 4;
            if (c < d) {
                Quad4<T> out;
#pragma unroll
                for (int r = 0; r < 4; ++r) out.v[r] = from_f32<T>(acc[rb][df][r] * scale);
                *reinterpret_cast<Quad4<T>*>(grow + c) = out;
            }
        }
    }
}

// PAIR: contract over two 16-row query blocks per MFMA (16x16x32) instead of one (16x16x16).  Half the MFMA issues for
// dK/dV, but two blocks of operands live at once: a win for wide heads (few key fragments per wave), a loss for the
// 40/64-wide heads where four key fragments per wave already fill the register file (measured: tools/flash_check.py).
template <typename T, int KS, int DF, int NKW, bool PAIR>
__global__ __launch_bounds__(256, 2) void attn_flash_dkdv_kernel(const T* __restrict__ Q, const T* __restrict__ K,
                                                               const T* __restrict__ V, const T* __restrict__ dO,
                                                               const float* __restrict__ LSE,
                                                               const float* __restrict__ Delta, T* __restrict__ dK,
                                                               T* __restrict__ dV, int Tq, int Tk, int H, int d,
                                                               float scale, float scale_log2e, int64_t ldq,
                                                               int64_t ld_dq) {
    using S = FlashShape<KS, DF>;
    using F8 = typename Mma<T>::F8;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    T* Qs = reinterpret_cast<T*>(smem);          // [2][64][KROW]
    T* Gs = Qs + 2 * S::K_HALFS;                  // [2][64][KROW]
    float* lse_s = reinterpret_cast<float*>(Gs + 2 * S::K_HALFS);  // [2][64]
    float* delta_s = lse_s + 2 * 64;                               // [2][64]

    int bx, bh;
    xcd_block(bx, bh);
    const int b = bh / H, h = bh - b * H;
    const int64_t HD = (int64_t)H * d;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int l15 = lane & 15, lq = lane >> 4;
    const T* Qh = Q + (int64_t)b * Tq * ldq + h * d;
    const T* Gh = dO + (int64_t)b * Tq * HD + h * d;
    const T* Kh = K + (int64_t)b * Tk * ldq + h * d;
    const T* Vh = V + (int64_t)b * Tk * ldq + h * d;
    const float* lse_h = LSE + (int64_t)bh * Tq;
    const float* delta_h = Delta + (int64_t)bh * Tq;
    const int key0 = bx * (64 * NKW) + wave * (16 * NKW);  // first key of this wave

    // the wave's keys as second MFMA operands (lane = key row, 8 head-dim values), kept for the whole kernel
    F8 kfr[NKW][KS], vfr[NKW][KS];
#pragma unroll
    for (int nf = 0; nf < NKW; ++nf) {
        const int key = key0 + nf * 16 + l15;
        load_row_frags<T, KS>(Kh + (int64_t)key * ldq, Kh, key < Tk, d, lq, kfr[nf]);
        prescale_frags<T, KS>(kfr[nf], scale_log2e);  // (these K fragments only feed the scores: dK contracts dS with Q)
        load_row_frags<T, KS>(Vh + (int64_t)key * ldq, Vh, key < Tk, d, lq, vfr[nf]);
    }
    f32x4 dk[NKW][DF], dv[NKW][DF];  // lane = head-dim column l15 of fragment df; keys nf*16 + lq*4 + r
#pragma unroll
    for (int nf = 0; nf < NKW; ++nf)
#pragma unroll
        for (int df = 0; df < DF; ++df) dk[nf][df] = dv[nf][df] = f32x4{0.f, 0.f, 0.f, 0.f};

    lds_zero(smem, flash_dkdv_lds_bytes<KS, DF>());
    __syncthreads();
    TileStageS<T, S> stage;  // (raw 128-bit chunks + settle(): attn_flash_common.h)
    RowStats stats;
    stage.init(d, ldq, HD);
    stage.load(Qh, Gh, Tq);
    stats.load(lse_h, delta_h, 0, Tq);
    stage.store_a_rows(Qs);
    stage.store_b_rows(Gs);
    stage.settle();
    stats.store(lse_s, delta_s);
    __syncthreads();
    const int n_tiles = (Tq + 63) / 64;
    // every load of the prologue has landed before the loop: hipcc's wait bookkeeping otherwise carries the never-waited-for
    // loads of the resident V fragments into the loop as "pending" and puts an s_waitcnt vmcnt(0) in front of their first use in
    // EVERY query block — behind the tile loads, whose latency that wait then exposes on every tile (round 6, from the ISA)
    __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0)
    // Keys past Tk (the last waves of the last workgroup of a ragged length) need NO masking here: a key's probabilities only
    // enter ITS OWN rows of dK and dV (the key is the output row of both contractions), those rows are never stored, and their
    // K / V fragments were loaded as zeros, so nothing non-finite can arise.  (Through round 5 a second instantiation of the loop
    // body masked them — 32 v_cndmask per query block — behind a wave-uniform branch; ADVICE r5 flagged that branch.)
    auto query_tile = [&](int qt) {
        const int cur = qt & 1;
        const T* Qc = Qs + cur * S::K_HALFS;
        const T* Gc = Gs + cur * S::K_HALFS;
        const float* lc = lse_s + cur * 64;   // −LSE (RowStats::store)
        const float* dc = delta_s + cur * 64;  // −Δ
        if (qt + 1 < n_tiles) {
            // (nothing here may make hipcc wait for these loads before the tile's arithmetic: TileStageS::load and RowStats —
            // through round 5 an s_waitcnt vmcnt(0) sat right behind them and exposed their whole latency on every tile)
            stage.load(Qh + (int64_t)(qt + 1) * 64 * ldq, Gh + (int64_t)(qt + 1) * 64 * HD, Tq - (qt + 1) * 64);
            stats.load(lse_h, delta_h, (qt + 1) * 64, Tq);
        }
        if constexpr (PAIR) {
#pragma unroll 1
            for (int qg = 0; qg < 2; ++qg) {  // 32 query rows at a time: two 16-row blocks fill one 32-deep MFMA contraction
                const int r0 = qg * 32;
                F8 pa[NKW], dsa[NKW];  // rows = the wave's keys (lane l15); contraction slot (lq, e) = row hb*16 + lq*4 + (e & 3)
#pragma unroll
                for (int hb = 0; hb < 2; ++hb) {
                    F8 qa[KS], ga[KS];
#pragma unroll
                    for (int ks = 0; ks < KS; ++ks) {
                        const int off = (r0 + hb * 16 + l15) * S::KROW + ks * 32 + lq * 8;
                        qa[ks] = *reinterpret_cast<const F8*>(Qc + off);
                        ga[ks] = *reinterpret_cast<const F8*>(Gc + off);
                    }
                    const f32x4 lse4 = *reinterpret_cast<const f32x4*>(lc + r0 + hb * 16 + lq * 4);
                    const f32x4 del4 = *reinterpret_cast<const f32x4*>(dc + r0 + hb * 16 + lq * 4);
#pragma unroll
                    for (int nf = 0; nf < NKW; ++nf) {
                        f32x4 s2 = lse4, dp2 = del4;  // row constants (−LSE, −Δ) as the initial accumulators (see the dQ kernel)
#pragma unroll
                        for (int ks = 0; ks < KS; ++ks) {
                            s2 = Mma<T>::k32(qa[ks], kfr[nf][ks], s2);  // D[q][key]: lane = key, rows lq*4 + r
                            dp2 = Mma<T>::k32(ga[ks], vfr[nf][ks], dp2);
                        }
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const float p = fast_exp2(s2[r]);
                            pa[nf][hb * 4 + r] = from_f32<T>(p);
                            dsa[nf][hb * 4 + r] = from_f32<T>(p * dp2[r]);  // 1/√d goes onto dK at the end
                        }
                    }
                }
                // transposed operands (lane = head-dim column, same contraction slots), one fragment at a time
#pragma unroll
                for (int df = 0; df < DF; ++df) {
                    // rows r0 + hb*16 + lq*4 + (0..3), column df*16 + l15: two transposing block reads per operand
                    const F8 qT = tr_pair<T>(lds_tr_block(Qc + r0 * S::KROW + df * 16, S::KROW, lane),
                                             lds_tr_block(Qc + (r0 + 16) * S::KROW + df * 16, S::KROW, lane));
                    const F8 gT = tr_pair<T>(lds_tr_block(Gc + r0 * S::KROW + df * 16, S::KROW, lane),
                                             lds_tr_block(Gc + (r0 + 16) * S::KROW + df * 16, S::KROW, lane));
#pragma unroll
                    for (int nf = 0; nf < NKW; ++nf) {
                        dv[nf][df] = Mma<T>::k32(pa[nf], gT, dv[nf][df]);
                        dk[nf][df] = Mma<T>::k32(dsa[nf], qT, dk[nf][df]);
                    }
                }
            }
        } else {
#pragma unroll 1
            for (int qb = 0; qb < 4; ++qb) {
                F8 qa[KS], ga[KS];
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) {
                    const int off = (qb * 16 + l15) * S::KROW + ks * 32 + lq * 8;
                    qa[ks] = *reinterpret_cast<const F8*>(Qc + off);
                    ga[ks] = *reinterpret_cast<const F8*>(Gc + off);
                }
                const f32x4 lse4 = *reinterpret_cast<const f32x4*>(lc + qb * 16 + lq * 4);
                const f32x4 del4 = *reinterpret_cast<const f32x4*>(dc + qb * 16 + lq * 4);
                tr4 qT[DF], gT[DF];  // transposed operands (lane = head-dim column, 4 query rows): one transposing read each
#pragma unroll
                for (int df = 0; df < DF; ++df) {
                    qT[df] = lds_tr_block(Qc + qb * 16 * S::KROW + df * 16, S::KROW, lane);
                    gT[df] = lds_tr_block(Gc + qb * 16 * S::KROW + df * 16, S::KROW, lane);
                }
#pragma unroll
                for (int nf = 0; nf < NKW; ++nf) {
                    f32x4 s2 = lse4, dp2 = del4;  // row constants (−LSE, −Δ) as the initial accumulators (see the dQ kernel)
#pragma unroll
                    for (int ks = 0; ks < KS; ++ks) {
                        s2 = Mma<T>::k32(qa[ks], kfr[nf][ks], s2);    // D[q][key]: lane = key, rows lq*4 + r
                        dp2 = Mma<T>::k32(ga[ks], vfr[nf][ks], dp2);
                    }
                    T pa[4], dsa[4];
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float p = fast_exp2(s2[r]);
                        pa[r] = from_f32<T>(p);
                        dsa[r] = from_f32<T>(p * dp2[r]);  // 1/√d goes onto dK at the end
                    }
#pragma unroll
                    for (int df = 0; df < DF; ++df) {
                        dv[nf][df] = Mma<T>::k16r(pa, gT[df], dv[nf][df]);
                        dk[nf][df] = Mma<T>::k16r(dsa, qT[df], dk[nf][df]);
                    }
                }
            }
        }
        if (qt + 1 < n_tiles) {
            stage.store_a_rows(Qs + (cur ^ 1) * S::K_HALFS);
            stage.store_b_rows(Gs + (cur ^ 1) * S::K_HALFS);
            stage.settle();
            stats.store(lse_s + (cur ^ 1) * 64, delta_s + (cur ^ 1) * 64);
        }
        __syncthreads();
    };
    for (int qt = 0; qt < n_tiles; ++qt) query_tile(qt);
#pragma unroll
    for (int nf = 0; nf < NKW; ++nf)
#pragma unroll
        for (int df = 0; df < DF; ++df) {
            const int c = df * 16 + l15;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int key = key0 + nf * 16 + lq * 4 + r;
                if (key < Tk && c < d) {
                    const int64_t off = ((int64_t)b * Tk + key) * ld_dq + h * d + c;
                    dK[off] = from_f32<T>(dk[nf][df][r] * scale);
                    dV[off] = from_f32<T>(dv[nf][df][r]);
                }
            }
        }
}

struct FlashPlan {
    int ks, df;
    int rb;      // 16-row blocks per wave, forward
    int rb_dq;   //                    ... dQ kernel (it carries dO fragments and dP accumulators as well)
    int nkw;     // 16-key fragments per wave, dK/dV kernel
};

bool plan_flash(int B, int Tq, int Tk, int H, int d, FlashPlan* pl) {
    if (B < 1 || Tq < 1 || Tk < 1 || H < 1 || d < 8 || (d % 8) != 0 || d > 160) return false;
    if (d <= 48) { pl->ks = 2; pl->df = 3; pl->rb = 4; pl->rb_dq = 2; pl->nkw = 4; }
    else if (d <= 64) { pl->ks = 2; pl->df = 4; pl->rb = 4; pl->rb_dq = 2; pl->nkw = 2; }  // (4 fragments per wave spill at this width)
    else if (d <= 80) { pl->ks = 3; pl->df = 5; pl->rb = 2; pl->rb_dq = 2; pl->nkw = 2; }
    else if (d <= 96) { pl->ks = 3; pl->df = 6; pl->rb = 2; pl->rb_dq = 2; pl->nkw = 2; }
    else if (d <= 128) { pl->ks = 4; pl->df = 8; pl->rb = 2; pl->rb_dq = 1; pl->nkw = 1; }
    else { pl->ks = 5; pl->df = 10; pl->rb = 1; pl->rb_dq = 1; pl->nkw = 1; }
    static const int nkw_env = [] { const char* e = getenv("FLASH_NKW"); return e ? atoi(e) : 0; }();  // tools/flash_check.py
    if (nkw_env == 2 && pl->nkw == 4) pl->nkw = 2;
    static const int rb_env = [] { const char* e = getenv("FLASH_RB"); return e ? atoi(e) : 0; }();
    if (rb_env == 2 && pl->rb == 4) pl->rb = 2;  // tuning knob for tools/flash_check.py
    return true;
}

template <int KS, int DF> constexpr int flash_fwd_lds() { return flash_fwd_lds_bytes<KS, DF>(); }

struct FlashArgs {
    const void *Q, *K, *V;
    void* O;
    float* LSE;
    int B, Tq, Tk, H, d;
    float scale;
    int64_t ldq;  // row stride (elements) shared by Q, K and V
};

template <typename T, int KS, int DF, int RB, bool ONES>
int launch_flash_fwd_v(const FlashArgs& a, hipStream_t stream) {
    constexpr int lds = flash_fwd_lds<KS, DF>();
    auto kern = attn_flash_fwd_kernel<T, KS, DF, RB, ONES>;
    if (lds > 48 * 1024) {
        static const hipError_t attr = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                                           hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        if (attr != hipSuccess) return LORA_E_LAUNCH;
    }
    const dim3 grid((unsigned)((a.Tq + 64 * RB - 1) / (64 * RB)), (unsigned)(a.B * a.H));
    {   // algorithmic work (profiler only): QKᵀ and PV; Q, K, V read and O written once
        const double bh = (double)a.B * a.H, e = sizeof(T);
        lora_prof_set_work(e * bh * a.d * (2.0 * a.Tq + 2.0 * a.Tk), 4.0 * bh * a.Tq * (double)a.Tk * a.d);
    }
    LORA_LAUNCH(PK_FLASH_FWD, kern, grid, dim3(256), lds, stream, static_cast<const T*>(a.Q), static_cast<const T*>(a.K),
                static_cast<const T*>(a.V), static_cast<T*>(a.O), a.LSE, a.Tq, a.Tk, a.H, a.d,
                a.scale * 1.4426950408889634f, a.ldq);
    lora_prof_set_work(0.0, 0.0);
    LORA_LAUNCH_CHECK();
    return LORA_OK;
}

template <typename T, int KS, int DF, int RB>
int launch_flash_fwd(const FlashArgs& a, hipStream_t stream) {
    static const int ones_env = [] { const char* e = getenv("FLASH_ONES"); return e ? atoi(e) : 1; }();  // tools/flash_check.py
    if (ones_env && (a.d % 16) == 8 && a.d / 16 == DF - 1) return launch_flash_fwd_v<T, KS, DF, RB, true>(a, stream);
    return launch_flash_fwd_v<T, KS, DF, RB, false>(a, stream);
}

template <typename T>
int dispatch_flash_fwd(const FlashArgs& a, const FlashPlan& pl, hipStream_t stream) {
#define FLASH_CASE(KS_, DF_, RB_) \
    if (pl.ks == KS_ && pl.df == DF_ && pl.rb == RB_) return launch_flash_fwd<T, KS_, DF_, RB_>(a, stream);
    FLASH_CASE(2, 3, 4) FLASH_CASE(2, 4, 4) FLASH_CASE(2, 3, 2) FLASH_CASE(2, 4, 2) FLASH_CASE(3, 5, 2) FLASH_CASE(3, 6, 2) FLASH_CASE(4, 8, 2) FLASH_CASE(5, 10, 1)
#undef FLASH_CASE
    return LORA_E_BADARG;
}

template <int KS, int DF> constexpr int flash_dq_lds() { return flash_dq_lds_bytes<KS, DF>(); }
template <int KS, int DF> constexpr int flash_dkdv_lds() { return flash_dkdv_lds_bytes<KS, DF>(); }

using FlashBwdArgs = LoraFlashBwdArgs;

template <typename T, int KS, int DF, int RBQ, int NKW>
int launch_flash_bwd(const FlashBwdArgs& a, hipStream_t stream) {
    const float l2e = a.scale * 1.4426950408889634f;
    // dQ first: it also produces Δ = Σ dO·O, which the dK/dV kernel reads
    {
        constexpr int lds = flash_dq_lds<KS, DF>();
        auto kern = attn_flash_dq_kernel<T, KS, DF, RBQ>;
        if (lds > 48 * 1024) {
            static const hipError_t attr = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                                               hipFuncAttributeMaxDynamicSharedMemorySize, lds);
            if (attr != hipSuccess) return LORA_E_LAUNCH;
        }
        const dim3 grid((unsigned)((a.Tq + 64 * RBQ - 1) / (64 * RBQ)), (unsigned)(a.B * a.H));
        {   // algorithmic work of the backward = 10·B·H·Tq·Tk·d (S, dP, dQ, dK, dV once each); this launch is charged dQ and
            // half of the shared S / dP (4·), the dK/dV launch the rest (6·) — both kernels really recompute S and dP
            const double bh = (double)a.B * a.H, e = sizeof(T);
            lora_prof_set_work(e * bh * a.d * (4.0 * a.Tq + 2.0 * a.Tk), 4.0 * bh * a.Tq * (double)a.Tk * a.d);
        }
        LORA_LAUNCH(PK_FLASH_DQ, kern, grid, dim3(256), lds, stream, static_cast<const T*>(a.Q), static_cast<const T*>(a.K),
                    static_cast<const T*>(a.V), static_cast<const T*>(a.O), static_cast<const T*>(a.dO), a.LSE,
                    a.delta, static_cast<T*>(a.dQ), a.Tq, a.Tk, a.H, a.d, a.scale, l2e, a.ldq, a.ld_dq);
        lora_prof_set_work(0.0, 0.0);
        LORA_LAUNCH_CHECK();
    }
    {
        constexpr int lds = flash_dkdv_lds<KS, DF>();
        auto kern = attn_flash_dkdv_kernel<T, KS, DF, NKW, (KS >= 3 || NKW <= 2)>;
        if (lds > 48 * 1024) {
            static const hipError_t attr = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                                               hipFuncAttributeMaxDynamicSharedMemorySize, lds);
            if (attr != hipSuccess) return LORA_E_LAUNCH;
        }
        const dim3 grid((unsigned)((a.Tk + 64 * NKW - 1) / (64 * NKW)), (unsigned)(a.B * a.H));
        {
            const double bh = (double)a.B * a.H, e = sizeof(T);
            lora_prof_set_work(e * bh * a.d * (2.0 * a.Tq + 4.0 * a.Tk), 6.0 * bh * a.Tq * (double)a.Tk * a.d);
        }
        LORA_LAUNCH(PK_FLASH_DKDV, kern, grid, dim3(256), lds, stream, static_cast<const T*>(a.Q), static_cast<const T*>(a.K),
                    static_cast<const T*>(a.V), static_cast<const T*>(a.dO), a.LSE, a.delta,
                    static_cast<T*>(a.dK), static_cast<T*>(a.dV), a.Tq, a.Tk, a.H, a.d, a.scale, l2e, a.ldq,
                    a.ld_dq);
        lora_prof_set_work(0.0, 0.0);
        LORA_LAUNCH_CHECK();
    }
    return LORA_OK;
}

template <typename T>
int dispatch_flash_bwd(const FlashBwdArgs& a, const FlashPlan& pl, hipStream_t stream) {
#define FLASH_BCASE(KS_, DF_, RBQ_, NKW_) \
    if (pl.ks == KS_ && pl.df == DF_ && pl.nkw == NKW_) return launch_flash_bwd<T, KS_, DF_, RBQ_, NKW_>(a, stream);
    FLASH_BCASE(2, 3, 2, 4) FLASH_BCASE(2, 4, 2, 4) FLASH_BCASE(2, 3, 2, 2) FLASH_BCASE(2, 4, 2, 2) FLASH_BCASE(3, 5, 2, 2) FLASH_BCASE(3, 6, 2, 2)
    FLASH_BCASE(4, 8, 1, 1) FLASH_BCASE(5, 10, 1, 1)
#undef FLASH_BCASE
    return LORA_E_BADARG;
}

}  // namespace

extern "C" int64_t attn_flash_bwd_workspace_bytes(int B, int Tq, int H) { return (int64_t)B * H * Tq * 4; }

extern "C" int attn_flash_bwd_strided(const void* Q, const void* K, const void* V, const void* O, const void* dO,
                                      const float* LSE, void* dQ, void* dK, void* dV, void* workspace, int64_t ldq,
                                      int64_t ld_dq, int B, int Tq, int Tk, int H, int d, float scale, int dtype,
                                      void* stream) {
    if (!Q || !K || !V || !O || !dO || !LSE || !dQ || !dK || !dV || !workspace) return LORA_E_BADARG;
    if (!aligned16(Q) || !aligned16(K) || !aligned16(V) || !aligned16(O) || !aligned16(dO) || !aligned16(dQ) ||
        !aligned16(dK) || !aligned16(dV) || !aligned16(workspace))
        return LORA_E_BADARG;
    const int64_t HD = (int64_t)H * d;
    if (ldq < HD || ld_dq < HD || (ldq % 8) != 0 || (ld_dq % 8) != 0 || 64 * ldq > 0x7fffffffLL) return LORA_E_BADARG;
    FlashPlan pl;
    if (!plan_flash(B, Tq, Tk, H, d, &pl)) return LORA_E_BADARG;
    FlashBwdArgs a{Q, K, V, O, dO, LSE, dQ, dK, dV, static_cast<float*>(workspace), B, Tq, Tk, H, d, scale, ldq, ld_dq};
    hipStream_t s = static_cast<hipStream_t>(stream);
    switch (dtype) {
        case LORA_F16: return dispatch_flash_bwd<half_t>(a, pl, s);
        case LORA_BF16: return dispatch_flash_bwd<bf16_t>(a, pl, s);
        default: return LORA_E_BADARG;
    }
}

extern "C" int attn_flash_bwd(const void* Q, const void* K, const void* V, const void* O, const void* dO,
                              const float* LSE, void* dQ, void* dK, void* dV, void* workspace, int B, int Tq, int Tk,
                              int H, int d, float scale, int dtype, void* stream) {
    return attn_flash_bwd_strided(Q, K, V, O, dO, LSE, dQ, dK, dV, workspace, (int64_t)H * d, (int64_t)H * d, B, Tq,
                                  Tk, H, d, scale, dtype, stream);
}

extern "C" int attn_flash_supported(int B, int Tq, int Tk, int H, int d, int dtype) {
    FlashPlan pl;
    return (dtype == LORA_F16 || dtype == LORA_BF16) && plan_flash(B, Tq, Tk, H, d, &pl) ? 1 : 0;
}

extern "C" int attn_flash_fwd_strided(const void* Q, const void* K, const void* V, void* O, float* LSE, int64_t ldq,
                                      int B, int Tq, int Tk, int H, int d, float scale, int dtype, void* stream) {
    if (!Q || !K || !V || !O) return LORA_E_BADARG;
    if (!aligned16(Q) || !aligned16(K) || !aligned16(V) || !aligned16(O)) return LORA_E_BADARG;
    if (ldq < (int64_t)H * d || (ldq % 8) != 0 || 64 * ldq > 0x7fffffffLL) return LORA_E_BADARG;
    FlashPlan pl;
    if (!plan_flash(B, Tq, Tk, H, d, &pl)) return LORA_E_BADARG;
    FlashArgs a{Q, K, V, O, LSE, B, Tq, Tk, H, d, scale, ldq};
    hipStream_t s = static_cast<hipStream_t>(stream);
    switch (dtype) {
        case LORA_F16: return dispatch_flash_fwd<half_t>(a, pl, s);
        case LORA_BF16: return dispatch_flash_fwd<bf16_t>(a, pl, s);
        default: return LORA_E_BADARG;
    }
}

extern "C" int attn_flash_fwd(const void* Q, const void* K, const void* V, void* O, float* LSE, int B, int Tq, int Tk,
                              int H, int d, float scale, int dtype, void* stream) {
    return attn_flash_fwd_strided(Q, K, V, O, LSE, (int64_t)H * d, B, Tq, Tk, H, d, scale, dtype, stream);
}

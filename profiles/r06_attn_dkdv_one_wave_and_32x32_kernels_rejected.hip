// Long-context attention backward for NARROW heads (d <= 48: the 40-wide heads of SD1.5's 4096-token level), dK / dV half.
// SURVEY §8 f-4 — the op between the to_q/to_k/to_v and to_out LoRA linears (reference callers lora_diffusion/lora.py:49-50;
// switched on at training_scripts/train_lora_dreambooth.py:623-625).
//
// Same arithmetic, operand layouts and output owners as attn_flash_dkdv_kernel (attn_flash.hip) — key-owned, S = Q·Kᵀ with
// the query fragment first so that the score accumulators ARE the first operands of the contractions over the query rows —
// built around one wave per SIMD instead of two:
//   * the loop of the two-wave form is a chain LDS read → score MFMAs → exponent → convert → dV/dK MFMAs with nothing else
//     to issue (profiles/r05_flash_dkdv_ablation.log: matrix-pipe time and everything else add up, 137 + 137 of 285 µs),
//     and its 253 registers leave no room to start the next block early.  Here a wave has the whole 512-entry file: its 64
//     keys' K / V fragments and dK / dV accumulators (160 registers) plus a three-stage software pipeline over PAIRS of
//     16-row query blocks: [score MFMAs of block 0] → [exponents of block 0 under the score MFMAs of block 1] →
//     [exponents of block 1 under the dV / dK MFMAs], with the next pair's row fragments already in flight;
//   * two query blocks fill ONE 32-deep contraction of the dV / dK products (16x16x32 instead of two 16x16x16 at the same
//     issue cost): 112 instead of 160 MFMA slots per 64 x 64 scores;
//   * Q / dO tiles go through a ring of THREE LDS buffers with one workgroup barrier per tile, placed in the MIDDLE of a
//     tile's arithmetic: the tile after the current one is complete in LDS half a tile before it is needed, so the pipeline
//     runs across tile seams (with two buffers the barrier has to sit between two tiles, and the first fragments of every tile
//     wait for an LDS round trip behind it).
// Built with -mllvm -amdgpu-mfma-vgpr-form (build_native.py): at one wave per SIMD hipcc otherwise gives every MFMA an
// accumulation-register destination and copies each score out (v_accvgpr_read) before the exponent — 196 more vector
// instructions per tile.
#include "attn_flash_common.h"

namespace {

constexpr int kNB = 3;  // LDS ring depth (tiles)

#ifndef FLASH_STAMP
#define FLASH_STAMP 0  // dev builds only (tools/flash_stamps.py): shader-clock stamps of one tile of one wave; results unaffected
#endif
#if FLASH_STAMP
__device__ unsigned long long flash_stamps[64];
#define STAMP(i)                                                                            \
    do {                                                                                    \
        __builtin_amdgcn_sched_barrier(0);                                                  \
        if (stamp_on) st[i] = (unsigned)__builtin_amdgcn_s_memtime();                       \
        __builtin_amdgcn_sched_barrier(0);                                                  \
    } while (0)
#else
#define STAMP(i) do {} while (0)
#endif

// two fp32 values rounded to the storage type (to nearest even, like from_f32) and packed into one register: v_cvt_pk_*
template <typename T> __device__ __forceinline__ uint32_t pack2(float a, float b) {
    typedef T t2 __attribute__((ext_vector_type(2)));
    const t2 v = {from_f32<T>(a), from_f32<T>(b)};
    return __builtin_bit_cast(uint32_t, v);
}
// an 8-wide MFMA operand assembled from four packed pairs
template <typename T> struct PkOperand {
    uint32_t w[4];
    __device__ __forceinline__ typename Mma<T>::F8 f8() const {
        typedef uint32_t u4 __attribute__((ext_vector_type(4)));
        const u4 v = {w[0], w[1], w[2], w[3]};
        return __builtin_bit_cast(typename Mma<T>::F8, v);
    }
};

template <typename T> constexpr int narrow_lds_bytes() {
    return kNB * (2 * FlashShape<2, 3>::K_HALFS * (int)sizeof(T) + 2 * 64 * 4) + 256 * 16;  // ring + a 16-byte dump per thread
}

template <typename T>
__global__ __launch_bounds__(256, 1) void attn_flash_dkdv_narrow_kernel(const T* __restrict__ Q, const T* __restrict__ K,
                                                                      const T* __restrict__ V, const T* __restrict__ dO,
                                                                      const float* __restrict__ LSE,
                                                                      const float* __restrict__ Delta, T* __restrict__ dK,
                                                                      T* __restrict__ dV, int Tq, int Tk, int H, int d,
                                                                      float scale, float scale_log2e, int64_t ldq,
                                                                      int64_t ld_dq) {
    constexpr int KS = 2, DF = 3, NKW = 4;
    using S = FlashShape<KS, DF>;
    using F8 = typename Mma<T>::F8;
    using Pk4 = PkOperand<T>;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    T* Qs = reinterpret_cast<T*>(smem);                              // [kNB][64][KROW]
    T* Gs = Qs + kNB * S::K_HALFS;                                   // [kNB][64][KROW]
    float* lse_s = reinterpret_cast<float*>(Gs + kNB * S::K_HALFS);  // [kNB][64]   (−LSE)
    float* delta_s = lse_s + kNB * 64;                               // [kNB][64]   (−Δ)
    T* dump = reinterpret_cast<T*>(delta_s + kNB * 64) + threadIdx.x * 8;  // where a thread's chunk goes when it has none in the tile

    int bx, bh;
    xcd_block(bx, bh);
    const int b = bh / H, h = bh - b * H;
    const int64_t HD = (int64_t)H * d;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
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

    lds_zero(smem, narrow_lds_bytes<T>());
    __syncthreads();
    const int n_tiles = (Tq + 63) / 64;
    TileStage<T, KS, DF> stage;
    RowStats stats;
    stage.init(d, ldq, HD);
    // tile 0 → ring slot 0; tile 1 → registers (it goes to slot 1 in the middle of tile 0)
    stage.load(Qh, Gh, Tq);
    stats.load(lse_h, delta_h, 0, Tq);
    stage.store_a_rows(Qs);
    stage.store_b_rows(Gs);
    stats.store(lse_s, delta_s);
    auto fetch_tile = [&](int t) {  // global → registers; past the end: the last tile again (never used)
        const int tt = t < n_tiles ? t : n_tiles - 1;
        stage.load(Qh + (int64_t)tt * 64 * ldq, Gh + (int64_t)tt * 64 * HD, Tq - tt * 64);
        stats.load(lse_h, delta_h, tt * 64, Tq);
    };
    fetch_tile(1);
    __syncthreads();

    // ---- row fragments of one PAIR of 16-row query blocks (rows r0 + hb·16 + l15): first operands of the score / dP MFMAs,
    // and the row constants −LSE, −Δ that start those chains (rows r0 + hb·16 + lq·4 + 0..3 — the accumulator layout)
    struct RowFrags {
        F8 qa[2][KS], ga[2][KS];
        f32x4 lse4[2], del4[2];
    } rf;
    auto read_rows = [&](int slot, int r0) {
        const T* Qc = Qs + slot * S::K_HALFS;
        const T* Gc = Gs + slot * S::K_HALFS;
#pragma unroll
        for (int hb = 0; hb < 2; ++hb) {
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                const int off = (r0 + hb * 16 + l15) * S::KROW + ks * 32 + lq * 8;
                rf.qa[hb][ks] = *reinterpret_cast<const F8*>(Qc + off);
                rf.ga[hb][ks] = *reinterpret_cast<const F8*>(Gc + off);
            }
            rf.lse4[hb] = *reinterpret_cast<const f32x4*>(lse_s + slot * 64 + r0 + hb * 16 + lq * 4);
            rf.del4[hb] = *reinterpret_cast<const f32x4*>(delta_s + slot * 64 + r0 + hb * 16 + lq * 4);
        }
    };
    read_rows(0, 0);
    // every load of the prologue has landed before the loop: hipcc's wait bookkeeping otherwise carries the never-waited-for
    // loads of the resident V fragments into the loop as "pending" and puts an s_waitcnt vmcnt(0) in front of their first use
    // in EVERY iteration — behind the tile loads, whose latency that wait then exposes on every tile
    __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0)
    // (keys past Tk need no masking: attn_flash.hip, dK/dV kernel)

    // One pair: 32 query rows of ring slot `slot`; (nslot, nr0) = where the NEXT pair's row fragments are.  The instruction
    // ORDER below is the schedule: at one wave per SIMD nothing else fills an issue slot, so every MFMA is followed by the
    // two or three vector instructions that fit under its 16 cycles of matrix pipe, and a scheduling fence after each such
    // group keeps hipcc from clustering the MFMAs (which it does, exponents after them, when left alone).
#define FENCE() __builtin_amdgcn_sched_barrier(0)
#if FLASH_STAMP
    bool stamp_on = false;
    const bool stamp_wave = blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x < 64;
    if (stamp_wave) {
        flash_stamps[0] = __builtin_amdgcn_s_memtime();
        flash_stamps[1] = __builtin_amdgcn_s_memrealtime();
    }
    unsigned st[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) st[i] = 0;
#endif
    auto pair = [&](int slot, int r0, int nslot, int nr0, auto mid, auto stamp_off) {
        [[maybe_unused]] constexpr int stamp_base = decltype(stamp_off)::value;
        const T* Qc = Qs + slot * S::K_HALFS;
        const T* Gc = Gs + slot * S::K_HALFS;
        // transposed operands of the pair for dV = Pᵀ·dO and dK = dSᵀ·Q (lane = head-dim column; contraction slot (lq, e) =
        // row r0 + (e>>2)·16 + lq·4 + (e&3)): two transposing block reads each, in flight under the score MFMAs
        F8 qT[DF], gT[DF];
#pragma unroll
        for (int df = 0; df < DF; ++df) {
            qT[df] = tr_pair<T>(lds_tr_block(Qc + r0 * S::KROW + df * 16, S::KROW, lane),
                                lds_tr_block(Qc + (r0 + 16) * S::KROW + df * 16, S::KROW, lane));
            gT[df] = tr_pair<T>(lds_tr_block(Gc + r0 * S::KROW + df * 16, S::KROW, lane),
                                lds_tr_block(Gc + (r0 + 16) * S::KROW + df * 16, S::KROW, lane));
        }
        f32x4 s0[NKW], p0[NKW], s1[NKW], p1[NKW];  // scores′ / dP′ of block 0 and block 1: D[q][key], lane = key, rows lq*4 + r
        // the four MFMAs of one (block, key fragment): i = 0: S, first half of the contraction (starts at −LSE); 1: dP (−Δ); 2, 3: second halves
        auto score_mfma = [&](int hb, int nf, int i, f32x4& s2, f32x4& dp2) {
            if (i == 0) s2 = Mma<T>::k32(rf.qa[hb][0], kfr[nf][0], rf.lse4[hb]);
            if (i == 1) dp2 = Mma<T>::k32(rf.ga[hb][0], vfr[nf][0], rf.del4[hb]);
            if (i == 2) s2 = Mma<T>::k32(rf.qa[hb][1], kfr[nf][1], s2);
            if (i == 3) dp2 = Mma<T>::k32(rf.ga[hb][1], vfr[nf][1], dp2);
        };
        // P and dS of the pair as packed 16-bit pairs: word w of fragment nf = contraction slots 2w, 2w+1 (block hb = w>>1)
        Pk4 pa[NKW], dsa[NKW];
        float e[4];
        // the vector work of one (block, key fragment) in four pieces of 2–4 instructions: p = exp2(s′), dS = p·dP′
        auto prob_piece = [&](int hb, int nf, int i, const f32x4& s2, const f32x4& dp2) {
            if (i == 0) {
                e[0] = fast_exp2(s2[0]);
                e[1] = fast_exp2(s2[1]);
            }
            if (i == 1) {
                e[2] = fast_exp2(s2[2]);
                e[3] = fast_exp2(s2[3]);
            }
            if (i == 2) {
                pa[nf].w[hb * 2] = pack2<T>(e[0], e[1]);
                pa[nf].w[hb * 2 + 1] = pack2<T>(e[2], e[3]);
            }
            if (i == 3) {  // 1/√d goes onto dK at the end
                dsa[nf].w[hb * 2] = pack2<T>(e[0] * dp2[0], e[1] * dp2[1]);
                dsa[nf].w[hb * 2 + 1] = pack2<T>(e[2] * dp2[2], e[3] * dp2[3]);
            }
        };
        // phase A: scores of block 0 (pure matrix pipe; the transposing reads above land under it)
#pragma unroll
        for (int nf = 0; nf < NKW; ++nf)
#pragma unroll
            for (int i = 0; i < 4; ++i) score_mfma(0, nf, i, s0[nf], p0[nf]);
        FENCE();
        STAMP(stamp_base + 1);
        // phase B: exponents of block 0 under the score MFMAs of block 1
#pragma unroll
        for (int nf = 0; nf < NKW; ++nf)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                score_mfma(1, nf, i, s1[nf], p1[nf]);
                prob_piece(0, nf, i, s0[nf], p0[nf]);
                FENCE();
            }
        // the row fragments' last readers have been issued: fetch the next pair's (and, in a tile's first half, move the
        // tile after this one from registers into the ring)
        STAMP(stamp_base + 2);
        read_rows(nslot, nr0);
        mid();
        FENCE();
        STAMP(stamp_base + 3);
        // phase C: exponents of block 1 under the dV / dK MFMAs
#pragma unroll
        for (int i = 0; i < 4; ++i) prob_piece(1, 0, i, s1[0], p1[0]);
        FENCE();
        STAMP(stamp_base + 4);
#pragma unroll
        for (int nf = 0; nf < NKW; ++nf)
#pragma unroll
            for (int j = 0; j < 2 * DF; ++j) {
                const int df = j >> 1;
                if (j & 1) dk[nf][df] = Mma<T>::k32(dsa[nf].f8(), qT[df], dk[nf][df]);
                else dv[nf][df] = Mma<T>::k32(pa[nf].f8(), gT[df], dv[nf][df]);
                if (nf + 1 < NKW && j < 4) prob_piece(1, nf + 1, j, s1[nf + 1], p1[nf + 1]);
                FENCE();
            }
    };

    int slot = 0;
    for (int qt = 0; qt < n_tiles; ++qt) {
        const int nslot = slot + 1 == kNB ? 0 : slot + 1;
#if FLASH_STAMP
        stamp_on = __builtin_amdgcn_readfirstlane(stamp_wave && qt == 20);
        STAMP(0);
#endif
        // first half; in its last phase tile qt+1 goes registers → ring slot nslot (its previous tenant, tile qt−2, was last
        // read before the barrier of tile qt−1), then the barrier of this tile
        pair(slot, 0, slot, 32, [&] {
            stage.store_rows_unmasked(Qs + nslot * S::K_HALFS, Gs + nslot * S::K_HALFS, dump);
            stats.store_unmasked(lse_s + nslot * 64, delta_s + nslot * 64);
        }, std::integral_constant<int, 0>{});
        STAMP(5);
        __syncthreads();
        STAMP(6);
        fetch_tile(qt + 2);  // in flight for a whole tile
#if FLASH_STAMP
        STAMP(7);
#endif
        // second half; its last phase already reads the first row fragments of tile qt+1
        pair(slot, 32, nslot, 0, [] {}, std::integral_constant<int, 8>{});
        STAMP(13);
        slot = nslot;
    }
#if FLASH_STAMP
    if (stamp_wave) {
        flash_stamps[2] = __builtin_amdgcn_s_memtime();
        flash_stamps[3] = __builtin_amdgcn_s_memrealtime();
#pragma unroll
        for (int i = 0; i < 16; ++i) flash_stamps[4 + i] = st[i];
    }
#endif
#undef FENCE

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

// =====================================================================================================================
// 32x32x16 form (round 6, second kernel of this file).  What the issue-cost measurements (tools/micro/issue_cost.hip,
// profiles/r06_issue_cost_*.log) say about these loops: an MFMA costs a wave ~12 issue cycles whatever its shape, a 16x16x32
// leaves 4 of its 16 pipe cycles for other instructions and a 32x32x16 leaves 20 of its 32 — and v_exp_f32 (8.6 cycles) plus
// the multiply / convert per score do not fit under 16x16 MFMAs: the two-wave kernel runs at MFMA time PLUS vector time.
// Here every product is a 32x32x16:
//   S′[q, key] = Q·Kᵀ − LSE and dP′ = dO·Vᵀ − Δ with the QUERY rows as the first operand (row fragments straight from the
//     row-major LDS tile, ds_read_b128) and the wave's keys as the second (8 consecutive head-dim values of a key row: a plain
//     16-byte global load, resident): 3 K-steps cover 48 head channels — 6 MFMAs per 32x32 scores instead of 16;
//   the accumulator tile has the key on the lane and 16 query rows in its registers, which IS the first operand of the
//     contractions over the query rows (cdna_hip_programming.md §3, "an accumulator tile as the next MFMA's operand"):
//     registers 8t..8t+7, converted pairwise, are K-step t of dV[key, c] += Pᵀ·dO and dK[key, c] += dSᵀ·Q; the second operand
//     (lane = head channel c, same query rows) comes from the transposing LDS read of the same tiles;
//   a wave owns 32 keys (dK / dV: 2 x 2 accumulator tiles of 32 keys x 32 channels, 64 registers), a workgroup 128, two
//     workgroups per CU as before.
// LDS rows are 144 bytes here (72 halfs): the 32-row first-operand reads are then conflict-free under gfx950's ds_read_b128 lane
// groups (MI355X_MICROARCH.md, LDS); the transposing reads of a 32-channel block touch 4 rows x 16 dwords per half-wave and
// collide on 8 of the 64 banks.
struct Mx32Shape {
    static constexpr int KROW = 72;
    static constexpr int IT = 2;  // 64 rows x (d <= 48)/8 chunks over 256 threads
    static constexpr int K_HALFS = kTile * KROW;
};
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <typename T> struct Mma32;
template <> struct Mma32<half_t> {
    static __device__ __forceinline__ f32x16 run(f16x8 a, f16x8 b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0); }
};
template <> struct Mma32<bf16_t> {
    static __device__ __forceinline__ f32x16 run(bf16x8 a, bf16x8 b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0); }
};
template <typename T> constexpr int mx32_lds_bytes() { return 4 * Mx32Shape::K_HALFS * (int)sizeof(T) + 4 * 64 * 4; }

template <typename T>
__global__ __launch_bounds__(256, 2) void attn_flash_dkdv_mx32_kernel(const T* __restrict__ Q, const T* __restrict__ K,
                                                                    const T* __restrict__ V, const T* __restrict__ dO,
                                                                    const float* __restrict__ LSE,
                                                                    const float* __restrict__ Delta, T* __restrict__ dK,
                                                                    T* __restrict__ dV, int Tq, int Tk, int H, int d,
                                                                    float scale, float scale_log2e, int64_t ldq,
                                                                    int64_t ld_dq) {
    using S = Mx32Shape;
    using F8 = typename Mma<T>::F8;
    using Pk4 = PkOperand<T>;
    constexpr int KT = 3;   // 16-deep K-steps over the head channels (d <= 48)
    constexpr int CB = 2;   // 32-channel blocks of dK / dV (channels >= d are padding: zero columns of the LDS tiles)
    extern __shared__ __attribute__((aligned(16))) char smem[];
    T* Qs = reinterpret_cast<T*>(smem);          // [2][64][KROW]
    T* Gs = Qs + 2 * S::K_HALFS;                  // [2][64][KROW]
    float* lse_s = reinterpret_cast<float*>(Gs + 2 * S::K_HALFS);  // [2][64]  (−LSE)
    float* delta_s = lse_s + 2 * 64;                               // [2][64]  (−Δ)

    int bx, bh;
    xcd_block(bx, bh);
    const int b = bh / H, h = bh - b * H;
    const int64_t HD = (int64_t)H * d;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int c32 = lane & 31, hh = lane >> 5;
    const T* Qh = Q + (int64_t)b * Tq * ldq + h * d;
    const T* Gh = dO + (int64_t)b * Tq * HD + h * d;
    const T* Kh = K + (int64_t)b * Tk * ldq + h * d;
    const T* Vh = V + (int64_t)b * Tk * ldq + h * d;
    const float* lse_h = LSE + (int64_t)bh * Tq;
    const float* delta_h = Delta + (int64_t)bh * Tq;
    const int key0 = bx * 128 + wave * 32;  // first key of this wave
    const bool key_ok = key0 + c32 < Tk;  // (loads only: keys past Tk need no masking — attn_flash.hip, dK/dV kernel)

    // the wave's keys as SECOND operands of the score products: lane = key c32, K-step s holds head channels 16s + 8·hh + 0..7
    F8 kB[KT], vB[KT];
    {
        const int key = key0 + c32;
#pragma unroll
        for (int s = 0; s < KT; ++s) {
            const int col = 16 * s + 8 * hh;
            const bool ok = key_ok && col < d;
            const Chunk<T> kc = load_or_zero<T>(ok ? Kh + (int64_t)key * ldq + col : Kh, ok);
            const Chunk<T> vc = load_or_zero<T>(ok ? Vh + (int64_t)key * ldq + col : Vh, ok);
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                kB[s][e] = from_f32<T>(to_f32<T>(kc.v[e]) * scale_log2e);  // scores leave the matrix pipe in the exp2 domain
                vB[s][e] = vc.v[e];
            }
        }
    }
    f32x16 dk[CB], dv[CB];  // D[key, c]: lane = channel 32·cb + c32, register r = key (r&3) + 8(r>>2) + 4·hh of the wave's 32
#pragma unroll
    for (int cb = 0; cb < CB; ++cb)
#pragma unroll
        for (int r = 0; r < 16; ++r) dk[cb][r] = dv[cb][r] = 0.f;

    lds_zero(smem, mx32_lds_bytes<T>());
    __syncthreads();
    TileStageS<T, S> stage;
    RowStats stats;
    stage.init(d, ldq, HD);
    stage.load(Qh, Gh, Tq);
    stats.load(lse_h, delta_h, 0, Tq);
    stage.store_a_rows(Qs);
    stage.store_b_rows(Gs);
    stats.store(lse_s, delta_s);
    __syncthreads();
    const int n_tiles = (Tq + 63) / 64;
    __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0): no prologue load is carried into the loop as "pending" (attn_flash.hip)

    // transposing-read address of this lane inside a tile: 16-lane group gi reads 4 rows x 16 channels — rows 4·(gi>>1) + (i>>2)
    // of the block, channels 16·(gi&1) + 4·(i&3) of the 32-channel block (lane i of the group receives channel i of the rows)
    const int tr_lane_off = (4 * (lane >> 5) + ((lane & 15) >> 2)) * S::KROW + 16 * ((lane >> 4) & 1) + 4 * (lane & 3);

    for (int qt = 0; qt < n_tiles; ++qt) {
        const int cur = qt & 1;
        const T* Qc = Qs + cur * S::K_HALFS;
        const T* Gc = Gs + cur * S::K_HALFS;
        const float* lc = lse_s + cur * 64;
        const float* dc = delta_s + cur * 64;
        if (qt + 1 < n_tiles) {
            stage.load(Qh + (int64_t)(qt + 1) * 64 * ldq, Gh + (int64_t)(qt + 1) * 64 * HD, Tq - (qt + 1) * 64);
            stats.load(lse_h, delta_h, (qt + 1) * 64, Tq);
        }
#pragma unroll
        for (int qb = 0; qb < 2; ++qb) {  // 32 query rows at a time
            const int r0 = qb * 32;
            // row fragments (first operand: lane = query row r0 + c32, channels 16s + 8·hh + 0..7) and the row constants in the
            // accumulators' own layout (register 4g + i = row r0 + 8g + 4·hh + i)
            F8 qA[KT], gA[KT];
#pragma unroll
            for (int s = 0; s < KT; ++s) {
                const int off = (r0 + c32) * S::KROW + 16 * s + 8 * hh;
                qA[s] = *reinterpret_cast<const F8*>(Qc + off);
                gA[s] = *reinterpret_cast<const F8*>(Gc + off);
            }
            f32x16 sc, dp;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const f32x4 l4 = *reinterpret_cast<const f32x4*>(lc + r0 + 8 * g + 4 * hh);
                const f32x4 d4 = *reinterpret_cast<const f32x4*>(dc + r0 + 8 * g + 4 * hh);
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    sc[4 * g + i] = l4[i];
                    dp[4 * g + i] = d4[i];
                }
            }
            // (two batches of LDS reads per block, pinned: hipcc otherwise issues each read right in front of its MFMA — ten exposed
            //  LDS round trips per block with only two waves per SIMD to cover them.  The first batch feeds the score products;
            //  the second — the transposed second operands of dV / dK: K-step t = rows r0 + 16t + {0, 8} + 4·hh + 0..3 — is
            //  issued before those products start and lands under them and the exponents)
            __builtin_amdgcn_sched_barrier(0);
            F8 gT[2][CB], qT[2][CB];
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int cb = 0; cb < CB; ++cb) {
                    const int base = (r0 + 16 * t) * S::KROW + 32 * cb + tr_lane_off;
                    gT[t][cb] = tr_pair<T>(lds_tr_at(Gc + base), lds_tr_at(Gc + base + 8 * S::KROW));
                    qT[t][cb] = tr_pair<T>(lds_tr_at(Qc + base), lds_tr_at(Qc + base + 8 * S::KROW));
                }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int s = 0; s < KT; ++s) {
                sc = Mma32<T>::run(qA[s], kB[s], sc);
                dp = Mma32<T>::run(gA[s], vB[s], dp);
            }
            __builtin_amdgcn_sched_barrier(0);
            // p = exp2(s′), dS = p·dP′ (1/√d goes onto dK at the end); registers 8t..8t+7 → K-step t of the next products
            Pk4 pa[2], dsa[2];
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int w = 0; w < 4; ++w) {
                    const float p0 = fast_exp2(sc[8 * t + 2 * w]), p1 = fast_exp2(sc[8 * t + 2 * w + 1]);
                    pa[t].w[w] = pack2<T>(p0, p1);
                    dsa[t].w[w] = pack2<T>(p0 * dp[8 * t + 2 * w], p1 * dp[8 * t + 2 * w + 1]);
                }
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int cb = 0; cb < CB; ++cb) {
                    dv[cb] = Mma32<T>::run(pa[t].f8(), gT[t][cb], dv[cb]);
                    dk[cb] = Mma32<T>::run(dsa[t].f8(), qT[t][cb], dk[cb]);
                }
        }
        if (qt + 1 < n_tiles) {
            stage.store_a_rows(Qs + (cur ^ 1) * S::K_HALFS);
            stage.store_b_rows(Gs + (cur ^ 1) * S::K_HALFS);
            stats.store(lse_s + (cur ^ 1) * 64, delta_s + (cur ^ 1) * 64);
        }
        __syncthreads();
    }
#pragma unroll
    for (int cb = 0; cb < CB; ++cb) {
        const int c = 32 * cb + c32;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int key = key0 + (r & 3) + 8 * (r >> 2) + 4 * hh;
            if (key < Tk && c < d) {
                const int64_t off = ((int64_t)b * Tk + key) * ld_dq + h * d + c;
                dK[off] = from_f32<T>(dk[cb][r] * scale);
                dV[off] = from_f32<T>(dv[cb][r]);
            }
        }
    }
}

template <typename T> int launch_dkdv_mx32(const LoraFlashBwdArgs& a, hipStream_t stream) {
    constexpr int lds = mx32_lds_bytes<T>();
    auto k0 = attn_flash_dkdv_mx32_kernel<T>;
    const dim3 grid((unsigned)((a.Tk + 127) / 128), (unsigned)(a.B * a.H));
    {
        const double bh = (double)a.B * a.H, e = sizeof(T);
        lora_prof_set_work(e * bh * a.d * (2.0 * a.Tq + 4.0 * a.Tk), 6.0 * bh * a.Tq * (double)a.Tk * a.d);
    }
    const float l2e = a.scale * 1.4426950408889634f;
    LORA_LAUNCH(PK_FLASH_DKDV, k0, grid, dim3(256), lds, stream, static_cast<const T*>(a.Q), static_cast<const T*>(a.K),
                    static_cast<const T*>(a.V), static_cast<const T*>(a.dO), a.LSE, a.delta, static_cast<T*>(a.dK),
                    static_cast<T*>(a.dV), a.Tq, a.Tk, a.H, a.d, a.scale, l2e, a.ldq, a.ld_dq);
    lora_prof_set_work(0.0, 0.0);
    LORA_LAUNCH_CHECK();
    return LORA_OK;
}

template <typename T> int launch_dkdv_narrow(const LoraFlashBwdArgs& a, hipStream_t stream) {
    constexpr int lds = narrow_lds_bytes<T>();
    auto k0 = attn_flash_dkdv_narrow_kernel<T>;
    static const hipError_t attr0 = hipFuncSetAttribute(reinterpret_cast<const void*>(k0), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    if (attr0 != hipSuccess) return LORA_E_LAUNCH;
    const dim3 grid((unsigned)((a.Tk + 255) / 256), (unsigned)(a.B * a.H));
    {
        const double bh = (double)a.B * a.H, e = sizeof(T);
        lora_prof_set_work(e * bh * a.d * (2.0 * a.Tq + 4.0 * a.Tk), 6.0 * bh * a.Tq * (double)a.Tk * a.d);
    }
    const float l2e = a.scale * 1.4426950408889634f;
    LORA_LAUNCH(PK_FLASH_DKDV, k0, grid, dim3(256), lds, stream, static_cast<const T*>(a.Q), static_cast<const T*>(a.K),
                    static_cast<const T*>(a.V), static_cast<const T*>(a.dO), a.LSE, a.delta, static_cast<T*>(a.dK),
                    static_cast<T*>(a.dV), a.Tq, a.Tk, a.H, a.d, a.scale, l2e, a.ldq, a.ld_dq);
    lora_prof_set_work(0.0, 0.0);
    LORA_LAUNCH_CHECK();
    return LORA_OK;
}

}  // namespace

#if FLASH_STAMP
extern "C" int lora_flash_read_stamps(unsigned long long* host64) {
    return hipMemcpyFromSymbol(host64, HIP_SYMBOL(flash_stamps), 64 * sizeof(unsigned long long)) == hipSuccess ? 0 : 1;
}
#endif

int lora_flash_dkdv_narrow(const LoraFlashBwdArgs& a, int dtype, hipStream_t stream, int form) {
    if (a.d > 48 || a.d < 8 || (a.d % 8) != 0) return LORA_E_BADARG;
    if (form == 2) {  // 32x32x16 products, two workgroups per CU
        switch (dtype) {
            case LORA_F16: return launch_dkdv_mx32<half_t>(a, stream);
            case LORA_BF16: return launch_dkdv_mx32<bf16_t>(a, stream);
            default: return LORA_E_BADARG;
        }
    }
    switch (dtype) {  // one wave per SIMD, 16x16x32 products
        case LORA_F16: return launch_dkdv_narrow<half_t>(a, stream);
        case LORA_BF16: return launch_dkdv_narrow<bf16_t>(a, stream);
        default: return LORA_E_BADARG;
    }
}

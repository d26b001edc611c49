// Definitions shared by the long-context attention kernels (attn_flash.hip; round 6 also built two alternative dK / dV kernels on
// them, measured and kept under profiles/r06_attn_dkdv_one_wave_and_32x32_kernels_rejected.hip).
#pragma once
#include <cstdlib>
#include <type_traits>

#include "attn_common.h"

// Arguments of the backward launches
struct LoraFlashBwdArgs {
    const void *Q, *K, *V, *O, *dO;
    const float* LSE;
    void *dQ, *dK, *dV;
    float* delta;
    int B, Tq, Tk, H, d;
    float scale;
    int64_t ldq, ld_dq;  // row strides shared by Q/K/V and by dQ/dK/dV
};

namespace {

constexpr int kTile = 64;          // keys per tile
constexpr int kNKF = kTile / 16;   // key fragments per tile

template <int KS, int DF> struct FlashShape {
    static constexpr int DP = KS * 32;
    static constexpr int DV = DF * 16;
    // halfs per K row in LDS: +32 B.  gfx950 services a ds_read_b128 in the lane groups {0–3,12–15,20–27}, {4–11,16–19,28–31}, …
    // and a ds_read_b64_tr_b16 in two groups of 32 lanes over 64 banks (MI355X_MICROARCH.md, LDS): with the usual +16 B every
    // fragment read and every transposing read of these tiles is a 2-way conflict (SQ_LDS_BANK_CONFLICT = half of
    // SQ_LDS_IDX_ACTIVE on all three kernels, round 4); a row stride ≡ 32 (mod 64) bytes makes both conflict-free.
    static constexpr int KROW = DP + 16;
    static constexpr int CPR = DP / 8;      // 16-byte chunks per key row
    static constexpr int N = kTile * CPR;
    static constexpr int IT = (N + 255) / 256;
    static constexpr int K_HALFS = kTile * KROW;
};

// K and V tiles are both staged ROW-major ([64 keys][KROW]); the products that contract over the keys (P·V, dS·K) read them
// with the transposing LDS read (ds_read_b64_tr_b16, attn_common.h) instead of from a second, transposed copy
template <int KS, int DF> constexpr int flash_fwd_lds_bytes() { return 4 * FlashShape<KS, DF>::K_HALFS * 2; }
template <int KS, int DF> constexpr int flash_dq_lds_bytes() { return 4 * FlashShape<KS, DF>::K_HALFS * 2; }
template <int KS, int DF> constexpr int flash_dkdv_lds_bytes() { return 4 * FlashShape<KS, DF>::K_HALFS * 2 + 4 * 64 * 4; }

// One 64-row tile of two [rows, H·d] tensors (K and V, or Q and dO) on its way global → registers → LDS.
// The chunk map is fixed for the whole kernel and computed once: a thread owns up to IT 16-byte chunks (row, col) of
// the d/8 REAL chunks of each row — the padding columns of the LDS tiles are zeroed once, never re-staged — so a full
// tile costs one 64-bit add and two unpredicated loads per chunk, no selects; only a ragged last tile checks rows.
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
// 16-byte load that is never predicated (attn_common.h, load_or_zero) as one 128-bit register group
__device__ __forceinline__ u32x4 load16_or_zero(const void* p, bool ok) {
    const u32x4 v = *reinterpret_cast<const u32x4*>(p);
    return ok ? v : u32x4{0u, 0u, 0u, 0u};
}

template <typename T, typename S> struct TileStageS {
    // (raw 128-bit registers, not element arrays: with 16-bit ELEMENTS on the ragged path hipcc merges the two paths of
    // load() element-wise and re-packs every chunk with v_alignbit / v_perm right behind its load — an s_waitcnt vmcnt
    // that exposes the whole load latency on every tile)
    u32x4 a[S::IT], b[S::IT];
    int64_t src[S::IT];           // element offset of the chunk inside tile 0 of A
    int dld;                      // row stride of B minus row stride of A (uniform): B's offset = src + row·dld
    int row[S::IT], rowoff[S::IT];  // row; offset in a row-major [64][KROW] tile
    bool have[S::IT];
    // (round 6 also tried ONE packed register of map per chunk, offsets recomputed per tile: the dQ kernel gained 2 %, the
    //  4096-token dK/dV launch lost 6 % — profiles/r06_flash_dkdv_stage_state_bisect.log — and kept this form)
#ifdef FLASH_ABL_NOSTAGE
    bool staged_once = false;
#endif
    __device__ __forceinline__ void init(int d, int64_t lda, int64_t ldb) {
        const int cpr = d >> 3, n = kTile * cpr;
        dld = (int)(ldb - lda);
#pragma unroll
        for (int i = 0; i < S::IT; ++i) {
            const int idx = threadIdx.x + i * 256;
            have[i] = idx < n;
            const int r = have[i] ? idx / cpr : 0, c = have[i] ? (idx - r * cpr) * 8 : 0;
            row[i] = r;
            src[i] = (int64_t)r * lda + c;
            rowoff[i] = r * S::KROW + c;
        }
    }
    // A, B: first row of the tile in each tensor; rows_valid >= 64 for a full tile
    __device__ __forceinline__ void load(const T* A, const T* B, int rows_valid) {
#ifdef FLASH_ABL_NOSTAGE  // diagnostic builds only (timing; results are wrong): every tile after the first re-uses tile 0's rows
        if (staged_once) return;
        staged_once = true;
#endif
        if (rows_valid >= kTile) {
#pragma unroll
            for (int i = 0; i < S::IT; ++i) {
                const int64_t off = have[i] ? src[i] : 0;
                int rr = have[i] ? row[i] : 0;
                asm volatile("" : "+v"(rr));  // recompute rr·dld at every tile: hoisted out of the loop it costs registers
                a[i] = *reinterpret_cast<const u32x4*>(A + off);
                b[i] = *reinterpret_cast<const u32x4*>(B + off + rr * dld);
            }
        } else {
#pragma unroll
            for (int i = 0; i < S::IT; ++i) {
                const bool ok = have[i] && row[i] < rows_valid;
                const int64_t off = ok ? src[i] : 0;
                int rr = ok ? row[i] : 0;
                asm volatile("" : "+v"(rr));
                a[i] = load16_or_zero(A + off, ok);
                b[i] = load16_or_zero(B + off + rr * dld, ok);
            }
        }
    }
    // Call after the chunks of a tile have gone to LDS (store_*_rows).  The chunks are stored under `have[i]`, so on the other
    // lanes' path hipcc's wait bookkeeping still counts the tile's loads as pending, carries that into the next iteration and —
    // depending on the register allocation — puts an s_waitcnt vmcnt(0) BETWEEN the next tile's loads or right behind them,
    // which exposes their whole latency on every tile (round 6: +30 µs on the 4096-token dK/dV launch after a one-line change
    // elsewhere moved two address registers).  Here the loads are a tile old: the wait is free, and it settles the bookkeeping.
    // (In FRONT of the next tile's loads the same wait cost 2.5 % of that launch: it pins the order of everything around it.)
    __device__ __forceinline__ void settle() const { __builtin_amdgcn_s_waitcnt(0x0F70); }  // vmcnt(0)
    __device__ __forceinline__ void store_a_rows(T* dst) const {
#pragma unroll
        for (int i = 0; i < S::IT; ++i)
            if (have[i]) *reinterpret_cast<u32x4*>(dst + rowoff[i]) = a[i];
    }
    __device__ __forceinline__ void store_b_rows(T* dst) const {
#pragma unroll
        for (int i = 0; i < S::IT; ++i)
            if (have[i]) *reinterpret_cast<u32x4*>(dst + rowoff[i]) = b[i];
    }
    // both tiles without a branch: a thread's chunks beyond the tile go to `dump` (16 bytes of LDS of its own)
    __device__ __forceinline__ void store_rows_unmasked(T* dst_a, T* dst_b, T* dump) const {
#pragma unroll
        for (int i = 0; i < S::IT; ++i) {
            *reinterpret_cast<u32x4*>(have[i] ? dst_a + rowoff[i] : dump) = a[i];
            *reinterpret_cast<u32x4*>(have[i] ? dst_b + rowoff[i] : dump) = b[i];
        }
    }
};

// The forward and dQ kernels keep round 5's form of the stage (16-bit element chunks): they never had a wait behind their tile
// loads, and with the raw-register form above the dQ kernel measured 3 % slower on one box (profiles/r06_flash_final_ab.log).
// One 64-row tile of two [rows, H·d] tensors (K and V, or Q and dO) on its way global → registers → LDS.
// The chunk map is fixed for the whole kernel and computed once: a thread owns up to IT 16-byte chunks (row, col) of
// the d/8 REAL chunks of each row — the padding columns of the LDS tiles are zeroed once, never re-staged — so a full
// tile costs one 64-bit add and two unpredicated loads per chunk, no selects; only a ragged last tile checks rows.
template <typename T, int KS, int DF> struct TileStage {
    using S = FlashShape<KS, DF>;
    Chunk<T> a[S::IT], b[S::IT];
    int64_t src[S::IT];           // element offset of the chunk inside tile 0 of A
    int dld;                      // row stride of B minus row stride of A (uniform): B's offset = src + row·dld
    int row[S::IT], rowoff[S::IT];  // row; offset in a row-major [64][KROW] tile
    bool have[S::IT];
    __device__ __forceinline__ void init(int d, int64_t lda, int64_t ldb) {
        const int cpr = d >> 3, n = kTile * cpr;
        dld = (int)(ldb - lda);
#pragma unroll
        for (int i = 0; i < S::IT; ++i) {
            const int idx = threadIdx.x + i * 256;
            have[i] = idx < n;
            const int r = have[i] ? idx / cpr : 0, c = have[i] ? (idx - r * cpr) * 8 : 0;
            row[i] = r;
            src[i] = (int64_t)r * lda + c;
            rowoff[i] = r * S::KROW + c;
        }
    }
    // A, B: first row of the tile in each tensor; rows_valid >= 64 for a full tile
    __device__ __forceinline__ void load(const T* A, const T* B, int rows_valid) {
        if (rows_valid >= kTile) {
#pragma unroll
            for (int i = 0; i < S::IT; ++i) {
                const int64_t off = have[i] ? src[i] : 0;
                int rr = have[i] ? row[i] : 0;
                asm volatile("" : "+v"(rr));  // recompute rr·dld at every tile: hoisted out of the loop it costs registers
                a[i] = *reinterpret_cast<const Chunk<T>*>(A + off);
                b[i] = *reinterpret_cast<const Chunk<T>*>(B + off + rr * dld);
            }
        } else {
#pragma unroll
            for (int i = 0; i < S::IT; ++i) {
                const bool ok = have[i] && row[i] < rows_valid;
                const int64_t off = ok ? src[i] : 0;
                int rr = ok ? row[i] : 0;
                asm volatile("" : "+v"(rr));
                a[i] = load_or_zero<T>(A + off, ok);
                b[i] = load_or_zero<T>(B + off + rr * dld, ok);
            }
        }
    }
    __device__ __forceinline__ void store_a_rows(T* dst) const {
#pragma unroll
        for (int i = 0; i < S::IT; ++i)
            if (have[i]) *reinterpret_cast<Chunk<T>*>(dst + rowoff[i]) = a[i];
    }
    __device__ __forceinline__ void store_b_rows(T* dst) const {
#pragma unroll
        for (int i = 0; i < S::IT; ++i)
            if (have[i]) *reinterpret_cast<Chunk<T>*>(dst + rowoff[i]) = b[i];
    }
};


// zero `bytes` of LDS (multiple of 16) cooperatively: the padding columns / rows of the tiles stay zero for the kernel's life
__device__ __forceinline__ void lds_zero(char* base, int bytes) {
    for (int o = threadIdx.x * 16; o < bytes; o += 256 * 16) *reinterpret_cast<f32x4*>(base + o) = f32x4{0.f, 0.f, 0.f, 0.f};
}

// v_exp_f32 without the library's denormal-range fix-ups (arguments here are <= 0; tiny results may flush to zero)
__device__ __forceinline__ float fast_exp2(float x) { return __builtin_amdgcn_exp2f(x); }

// max / sum over the four lanes that share a query row (lane = l15 + 16·lq): two row swaps on the VALU (gfx950
// v_permlane32_swap / v_permlane16_swap) instead of two ds_bpermute round trips through the LDS crossbar
__device__ __forceinline__ float quad_max(float x) {
    const unsigned u = __float_as_uint(x);
    const auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false);
    const float a = fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
    const unsigned ua = __float_as_uint(a);
    const auto q = __builtin_amdgcn_permlane16_swap(ua, ua, false, false);
    return fmaxf(__uint_as_float(q[0]), __uint_as_float(q[1]));
}
__device__ __forceinline__ float quad_sum(float x) {
    const unsigned u = __float_as_uint(x);
    const auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false);
    const float a = __uint_as_float(r[0]) + __uint_as_float(r[1]);
    const unsigned ua = __float_as_uint(a);
    const auto q = __builtin_amdgcn_permlane16_swap(ua, ua, false, false);
    return __uint_as_float(q[0]) + __uint_as_float(q[1]);
}

// Workgroup → (block along x, batch·head) with the blocks of ONE (batch, head) kept on ONE XCD.  The hardware deals consecutive
// workgroup ids (x fastest) round-robin to the eight XCDs, so with the plain (blockIdx.x, blockIdx.y) mapping the 16 row blocks of
// a head are spread over all eight L2s and every one of them fetches that head's K and V (forward, dQ) or Q and dO (dK/dV): 4.5×
// the algorithmic bytes at 4096 tokens (PMC, round 4).  Here XCD x owns a contiguous run of (batch·head, block) pairs — the same
// bijective remap as the fused GEMM's tiles — so a head's streamed operands are fetched by one L2.
__device__ __forceinline__ void xcd_block(int& bx, int& bh) {
    const unsigned nx = gridDim.x, total = gridDim.x * gridDim.y;
    const unsigned id = blockIdx.x + nx * blockIdx.y;
    const unsigned q = total >> 3, rem = total & 7, xcd = id & 7, slot = id >> 3;
    const unsigned n = (xcd < rem ? xcd * (q + 1) : rem * (q + 1) + (xcd - rem) * q) + slot;
    bh = (int)(n / nx);
    bx = (int)(n - (unsigned)bh * nx);
}

// max of the 16 scores a lane holds for one row block, as a tree of three-input maxima (v_max3_f32): 8 instructions instead of
// the 16-long chain the scalar loop compiles to
__device__ __forceinline__ float max3f(float a, float b, float c) { return __builtin_fmaxf(__builtin_fmaxf(a, b), c); }
__device__ __forceinline__ float max16(const f32x4 (&s)[kNKF]) {
    const float a = max3f(s[0][0], s[0][1], s[0][2]), b = max3f(s[0][3], s[1][0], s[1][1]);
    const float c = max3f(s[1][2], s[1][3], s[2][0]), d = max3f(s[2][1], s[2][2], s[2][3]);
    const float e = max3f(s[3][0], s[3][1], s[3][2]);
    return __builtin_fmaxf(max3f(a, b, c), max3f(d, e, s[3][3]));
}

// Softmax scale folded into the operand: q·(scale·log2 e), rounded to the storage type ONCE per kernel.  Together with a row
// constant as the INITIAL accumulator of the score MFMAs (−m, −LSE, −Δ) the exponent's argument leaves the matrix pipe
// ready: p = exp2(acc) — no per-score fma / subtraction on the VALU (cdna_hip_programming.md, attention backward: "row
// constants as the initial accumulator").  The extra rounding of q (2^-11 relative in f16) moves the result by about as much
// as the storage rounding of q itself: 2e-4 → 3e-4 relative against float64 at the SD shapes (tolerance of the tests: 2e-3).
template <typename T, int KS>
__device__ __forceinline__ void prescale_frags(typename Mma<T>::F8 (&f)[KS], float c) {
#pragma unroll
    for (int ks = 0; ks < KS; ++ks)
#pragma unroll
        for (int e = 0; e < 8; ++e) f[ks][e] = from_f32<T>(to_f32<T>(f[ks][e]) * c);
}

// LSE and Δ of one 64-row query tile: threads 0..63 carry one row each (+inf / 0 past the end: probability 0)
struct RowStats {
    float lse, delta;  // as loaded (a valid row's, whatever `ok` says)
    bool ok;
    // No predicated load and no select behind the load (either would make hipcc wait for it on the spot): the row index is
    // clamped, and rows past the end get their +inf / 0 when the values go to LDS
    __device__ __forceinline__ void load(const float* lse_h, const float* delta_h, int row0, int Tq) {
        const int r = row0 + (int)(threadIdx.x & 63);
        ok = r < Tq;
        const int rr = ok ? r : Tq - 1;
        lse = lse_h[rr];
        delta = delta_h[rr];
    }
    // stored NEGATED: they are the initial accumulators −LSE / −Δ of the score and dP chains, and a negation per query block
    // and wave is eight VALU instructions in a loop that is bound by vector issue.  Every wave holds the same 64 rows and
    // writes the same words (no `threadIdx.x < 64` branch: a consumer under a branch leaves the loads "pending" on the other
    // waves' path in hipcc's wait bookkeeping — attn_flash.hip, dK/dV kernel)
    __device__ __forceinline__ void store(float* lse_s, float* delta_s) const {
        lse_s[threadIdx.x & 63] = ok ? -lse : -INFINITY;
        delta_s[threadIdx.x & 63] = ok ? -delta : 0.f;
    }
    __device__ __forceinline__ void store_unmasked(float* lse_s, float* delta_s) const { store(lse_s, delta_s); }
};

}  // namespace

// LoRA-fused GEMM for gfx950.
//
// One kernel template serves both big contractions of the hot path:
//   forward   (lora_diffusion/lora.py:49-50):   Y  = X·Wᵀ + b + s·(X·Aᵀ)·Bᵀ     and  T = X·Aᵀ
//   backward  (autograd of the same line):      dX = dY·W   + s·(dY·B)·A        and  U = dY·B
// written once as
//   C[M,Nc] = Am[M,Kc]·Bm[Nc,Kc]ᵀ + bias + s·P·Qᵀ ,   P[M,r] = Am·Fᵀ  (stored unscaled)
// with (Am,Bm,F,Q) = (X, W, A, B) forward and (dY, Wᵀ, Bᵀ, Aᵀ) backward.  Both big operands are
// contraction-contiguous ("NT" GEMM), which is why the frozen weight is cached in both
// orientations.  F arrives PACKED: [16, Kc] in the compute dtype, rows ≥ r zero
// (lora_pack_factors), so it is staged exactly like an operand tile.
//
// Structure per workgroup of NW waves (2 row waves × NW/2 column waves; 256 threads in every instantiated form), BM×BN output
// tile (128×128, 128×160, 64×160, 64×128, 64×64), 128-byte K-steps:
//   - PIPE main loop: an LDS ring of STG = 2–4 stages filled by LDS-DMA (global_load_lds, 16 B per lane, no VGPR staging),
//     STG−1 K-steps in flight behind a COUNTED s_waitcnt vmcnt and one raw s_barrier per step.  The XOR swizzle of the 16-B
//     chunks is applied to the per-lane SOURCE address (the DMA writes LDS in lane order) and again on the fragment reads,
//     which are then bank-conflict-free ds_read_b128.  Two workgroups per CU by construction (≤ 80 KB LDS, ≤ 256 registers);
//   - fallback main loop (contraction not a multiple of the K-step): register-staged, 2 barriers per step;
//   - base contraction on MFMA 16x16x32 (f16/bf16) or 16x16x4 (f32, exact), issued with the Bm fragment as the
//     first operand: a lane then owns 4 CONSECUTIVE COLUMNS of one output row (and 4 consecutive rank entries of
//     P), so the epilogue moves 8/16 bytes per LDS access instead of one element;
//   - the rank-r factor rides along as a 16-row tile: the X fragments already in VGPRs are multiplied
//     with it (one extra MFMA per row fragment, K-steps split between the two column waves), so T
//     costs no extra HBM or LDS traffic for X;
//   - epilogue on the ring's own buffers: the partial P tiles of the column waves meet in the buffer that is free after the
//     last K-step; thread (row, half) sums them, writes P out and leaves s·P as ONE packed record per row — a high and a low
//     16-bit part in MFMA operand order — so every wave fetches a fragment with one 16-byte read and s·P·Qᵀ enters the
//     accumulators as ONE extra MFMA K-step ([hi | lo] × [Q | Q]: P keeps ~fp32 precision).  The Q tile itself is fetched by
//     DMA during the last K-step into the same free buffer (no LDS of its own: what lets two 128×160 tiles share a CU);
//   - bias is added in fp32, the tile is transposed through the buffer of the last K-step and stored as whole 16-B row chunks;
//   - GATE instantiation (the `proj` layer of a GEGLU block): a column tile is 64 hidden columns plus the 64 gate columns behind
//     them, and the store pass writes h·gelu(g) next to (or instead of) the [h | g] tile — diffusers GEGLU.forward in the epilogue.
//   - SPLITK instantiations (long contractions on grids too small for the chip): the K-slices of a tile are workgroups of the
//     SAME launch; each stores its fp32 partials write-through (sc1), draws a ticket, and the last arriver adds the slices in
//     index order and runs the epilogue above — deterministic, no second launch (plan_splitk()).
// Workgroups are dealt to XCDs so that the column tiles of one row panel (or the row tiles of one column panel, whichever
// re-fetches the smaller operand) share an L2 — or, for square-ish problems, so that each XCD owns a rectangle of the tile grid.
// Tile / ring / split choice per shape: launch_pipe(), gate_tile_width(), plan_splitk() — all measured with cold weights.
#include <cstdlib>

#include "common.h"

namespace {

struct GemmParams {
    const void* Am;
    const void* Bm;
    const void* bias;
    const void* Fp;  // packed main-loop factor   [16, Kc] (dtype T), rows >= r zero
    const void* Qp;  // packed epilogue factor    [Nc, 16] (dtype T), columns >= r zero
    void* C;
    float* P;
    int64_t M;
    int Kc, Nc, r;
    float scale;
    int tiles_m, tiles_n;
    int col_major;  // 1: consecutive tiles walk down M inside a column tile (an XCD then owns a slice of Bm)
    int xcd_m;      // > 1: the 8 XCDs form an xcd_m × (8/xcd_m) grid over the tile grid, each XCD owns a RECTANGLE of tiles
                    // (both splits exact) — an XCD's L2 then fetches |Am|/xcd_m + |Bm|·xcd_m/8 instead of all of one operand
    int64_t lda;    // row stride of Am in elements (>= Kc: Am may be a column slice of a wider buffer)
    // Grouped launches (several LoRA layers that share Am in one launch; nullptr = one layer):
    //  MAIN:  tile_part[tn] = part | first << 16 for every column tile — the tile multiplies with the part's own
    //         factor rows Fp + part·16·Kc, and the part's first tile writes P + part·M·r (column tiles never
    //         straddle parts: the host picks BN accordingly);
    // !MAIN:  part_table[g] = {column offset into Am, contraction length, element offset into Fp, float offset
    //         into P} — blockIdx covers (row tile, part), every part contracts its own column range of Am.
    const int* tile_part;
    const int64_t* part_table;
    // Equal-width parts without a table (grouped q/k/v at any rank <= 16: lora_gemm_parts):
    //  n_parts > 0, parts_on_k == 0 (forward):  the Nc output columns are n_parts runs of part_n; a column tile lies inside one
    //         part (the host picks BN | part_n), multiplies with that part's factor rows Fp + part·16·Kc, and the part's first
    //         tile writes its P columns [part·r, part·r + r) of the [M, ldp] buffer;
    //  n_parts > 0, parts_on_k == 1 (backward-input, KP kernels): the CONTRACTION is n_parts runs of part_n — Fp [16, Kc]
    //         holds part g's factor rows in its own column run, P_g = Am[:, run g]·F_gᵀ accumulates per part, and the
    //         epilogue adds s·Σ_g P_g·Q_gᵀ with Qp = [n_parts][Nc, 16].
    int n_parts, part_n, parts_on_k;
    int ldp;  // row stride of P in floats (r for a single layer)
    // Split-K (contractions on grids too small for the chip): slice s of a tile contracts K-steps [s·steps_per_slice, …),
    // STORES its fp32 accumulators at ws_c[tile][s] (BM·BN floats in the lanes' own register order) and its partial P at
    // ws_p[tile][s][BM][16], and takes a ticket of the tile.  The workgroup that draws the LAST ticket adds the slices
    // 0..S-1 in index order — whoever arrives last, the sum has one order: deterministic — and runs the normal epilogue
    // (rank-r term, bias, store) on the totals; it leaves the ticket at zero for the next launch.  splitk <= 1: off.
    int splitk, steps_per_slice;
    int split_aff;      // 1: slice s of every tile runs on XCD s % S (S divides 8) — a byte of Am / Bm is fetched by ONE L2
    float* ws_c;
    float* ws_p;
    unsigned* tickets;  // one per output tile, zero on entry, zero again on exit
    int split_bm;       // (host only) row-tile height of a split launch
    // GEGLU gate in the epilogue (GATE kernels; diffusers GEGLU.forward behind the `proj` LoraInjectedLinear): Nc = 2·gateF,
    // a column tile owns BN/2 columns of h = C[:, :gateF] AND the matching BN/2 columns of g = C[:, gateF:], and the
    // epilogue writes C2[M, gateF] = h·gelu(g) next to C (C may be null: nothing is saved for a backward pass).
    void* C2;
    int gateF;
    int dbg;  // tools/gemm_bench.py ablations (LORA_GEMM_DBG, timing only — results are wrong): 1 = no DMA after the prologue, 2 = no MFMA work
};

constexpr int kRowBytes = 128;  // one K-step of one tile row
constexpr int kRP = 16;         // rank padded to one MFMA fragment
constexpr int kSPS = 20;        // fp32 row stride of the P image in LDS: 80 B keeps the 16-B row reads conflict-free
template <typename T> struct alignas(sizeof(T) * 4) Quad { T v[4]; };

__device__ __forceinline__ int lds_off(int row, int chunk) {
    return row * kRowBytes + ((chunk ^ (row & 7)) << 4);
}

__device__ __forceinline__ void glds16(const void* gsrc, char* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gsrc),
                                     (__attribute__((address_space(3))) void*)(lds_wave_base), 16, 0, 0);
}

template <typename T> struct Mfma;
template <> struct Mfma<half_t> {
    using Frag = f16x8;
    static __device__ __forceinline__ f32x4 run(Frag a, Frag b, f32x4 c) {
        return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
    }
};
template <> struct Mfma<bf16_t> {
    using Frag = bf16x8;
    static __device__ __forceinline__ f32x4 run(Frag a, Frag b, f32x4 c) {
        return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
    }
};

#ifdef LORA_STAMPS  // diagnostic build only (tools/wg_timeline.py): per-workgroup phase timestamps
__device__ unsigned long long g_stamps[8192 * 16];
#define STAMP(i)                                                                  \
    do {                                                                          \
        if (tid == 0 && blockIdx.x < 8192) g_stamps[blockIdx.x * 16 + (i)] = __builtin_readcyclecounter(); \
    } while (0)
#define STAMP_WALL(i)                                                             \
    do {                                                                          \
        if (tid == 0 && blockIdx.x < 8192) g_stamps[blockIdx.x * 16 + (i)] = wall_clock64(); \
    } while (0)
#else
#define STAMP(i)
#define STAMP_WALL(i)
#endif

// One stage of the ring / the single staging buffer: A rows, B rows, 16 factor rows.  Only the first two waves load the
// factor rows, so they carry one more DMA per stage than the others: the counted waits are picked per wave.
template <int BM, int BN, bool MAIN, int STG, int NW> constexpr int stage_bytes() {  // (same for every wave layout)
    return (BM + (MAIN ? BN : 0) + kRP) * kRowBytes;
}
template <int BM, int BN, typename T, bool MAIN, int STG, int NW, int WM> constexpr int gemm_lds_bytes() {
    constexpr int ring = (STG > 0 ? STG : 1) * stage_bytes<BM, BN, MAIN, STG, NW>();
    constexpr int sq = MAIN && STG == 0 ? BN * kRP * (int)sizeof(T) : 0;  // ring kernels stage Q in a free ring buffer
    constexpr int ep = sizeof(T) == 4 ? 2 : 1;
    constexpr int sc = (MAIN && STG == 0) || (MAIN && sizeof(T) == 4) ? (BM / ep) * (BN * (int)sizeof(T) + 16) : 0;
    constexpr int sp = (NW / WM) * BM * kSPS * 4;
    constexpr int a = ring + sq;
    constexpr int b = sc > sp ? sc : sp;
    return a > b ? a : b;
}

// STG: LDS ring depth of the DMA pipeline (2 or 3); 0 selects the register-staged fallback loop.
// NW waves as WM row waves × NW/WM column waves; every wave owns a (BM/WM) × (BN·WM/NW) piece of the tile.
// SPLITK: the launch cuts its contraction into K-slices (in-launch combine, below) — instantiations of their own, so that the
// unsplit kernels carry none of that code and profilers can tell the two apart by name
// KP > 1: the contraction consists of KP equal parts with a factor pair each (grouped q/k/v backward-input at 3r > 16)
template <typename T, int BM, int BN, bool MAIN, int STG, int NW, int WM, int GATE = 0, bool SPLITK = false, int KP = 1>  // GATE: 0 none, 1 GEGLU forward, 2 GEGLU backward
__global__ __launch_bounds__(NW * 64, NW == 4 ? 2 : 1) void lora_gemm_kernel(GemmParams p) {  // 4-wave tiles: two per CU
    constexpr bool PIPE = STG > 0;
    constexpr int kStages = PIPE ? STG : 1;
    constexpr int NT = NW * 64;        // threads
    constexpr int WN = NW / WM;        // column waves
    constexpr int WTN = BN / WN;       // wave tile width
    constexpr int RPP = NW * 8;        // tile rows covered by one staging pass
    constexpr int VEC = ElemTraits<T>::kVec;
    constexpr int BK = kRowBytes / (int)sizeof(T);
    constexpr int WTM = BM / WM;       // wave tile height
    constexpr int MI = WTM / 16;  // 16-row fragments per wave
    constexpr int NI = WTN / 16;
    constexpr int PA = BM / RPP;  // staging passes
    constexpr int PB = BN / RPP;
    constexpr bool F32 = sizeof(T) == 4;
    constexpr int STAGE = stage_bytes<BM, BN, MAIN, STG, NW>();
    constexpr int OFF_B = BM * kRowBytes;
    constexpr int OFF_F = (BM + (MAIN ? BN : 0)) * kRowBytes;

    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* sQ = smem + kStages * STAGE;  // (ring kernels: re-pointed into the ring when the Q tile is fetched, below)
    float* sP = reinterpret_cast<float*>(smem);  // epilogue overlay: [NW/2][BM][kSPS] (ring kernels: a free ring buffer)

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN;
    const int wn = wave % WN;
    const int l15 = lane & 15;
    const int lq = lane >> 4;

    STAMP_WALL(0);
    STAMP(1);
#ifdef LORA_STAMPS
    if (tid == 0 && blockIdx.x < 8192) {
        g_stamps[blockIdx.x * 16 + 10] = __builtin_amdgcn_s_getreg((31 << 11) | 4);   // HW_ID
        g_stamps[blockIdx.x * 16 + 11] = __builtin_amdgcn_s_getreg((31 << 11) | 20);  // XCC_ID
    }
#endif
    // XCD-aware tile assignment: blocks b and b+8 share an XCD (round-robin dispatch), so give
    // every XCD a contiguous run of tiles; column tiles of one row panel are consecutive.
    // Pull every kernel argument into SGPRs now: one scalar-load round trip instead of two dependent ones.
    asm volatile("" ::"s"(p.Am), "s"(p.Bm), "s"(p.bias), "s"(p.Fp), "s"(p.Qp), "s"(p.C), "s"(p.P), "s"(p.M), "s"(p.Kc),
                 "s"(p.Nc), "s"(p.scale), "s"(p.tiles_m), "s"(p.tiles_n), "s"(p.col_major), "s"(p.lda), "s"(p.tile_part),
                 "s"(p.part_table), "s"(p.splitk), "s"(p.steps_per_slice), "s"(p.ws_c), "s"(p.ws_p), "s"(p.tickets), "s"(p.xcd_m),
                 "s"(p.split_aff));
    int tile, slice = 0;
    {
        const int S = SPLITK ? p.splitk : 1;
        const int total = p.tiles_m * p.tiles_n * S;
        const int id = blockIdx.x;
        const int q = total >> 3, rem = total & 7;
        const int xcd = id & 7, slot = id >> 3;
        tile = (xcd < rem ? xcd * (q + 1) : rem * (q + 1) + (xcd - rem) * q) + slot;
        if (SPLITK && p.split_aff) {
            // Slice → XCD affinity (S divides 8): XCD x runs slice x % S of the tiles of sub-range g = x / S (8/S sub-ranges of
            // the tile list).  A K-range of Am and Bm is then streamed by 8/S L2s instead of by all eight — with S = 8 every
            // operand byte is fetched by exactly one XCD (round 3's run order made every XCD stream ALL of Bm: 3.5x the
            // algorithmic bytes).  The block counts match the round-robin dispatch: XCD x holds total/8 (+1 if x < total % 8)
            // blocks and total % 8 = S·(tiles % G), so exactly the XCDs of the sub-ranges g < tiles % G take one more tile.
            const int G = 8 / S, tiles = p.tiles_m * p.tiles_n;
            const int g = xcd / S, tq = tiles / G, tr = tiles % G;
            slice = xcd % S;
            tile = g * tq + (g < tr ? g : tr) + slot;
        } else if (S > 1) {  // the slices of one tile are neighbours: they read the same operand rows, a K-range each
            slice = tile % S;
            tile = tile / S;
        }
    }
    // Which operand should stay inside one XCD's L2?  Row-major tile order keeps a row panel of Am there and makes
    // every XCD stream all of Bm; column-major order keeps a slice of Bm there and streams Am instead.  The host
    // picks the order that re-fetches the SMALLER operand eight times (col_major when Bm is the bigger one).
    int tm, tn;
    if (p.xcd_m > 1) {
        // 2-D XCD ownership (square-ish problems: neither operand is small enough to be re-fetched by all eight L2s).
        // Block id → (XCD, slot) as above; the XCD's rectangle is rm × rn tiles, walked column by column.
        const int xn = 8 / p.xcd_m;
        const unsigned rm = p.tiles_m / p.xcd_m, rn = p.tiles_n / xn;
        const unsigned xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
        const unsigned ln = slot / rm, lm = slot - ln * rm;
        tm = (xcd / xn) * rm + lm;
        tn = (xcd % xn) * rn + ln;
        tile = tm * p.tiles_n + tn;
    } else {
        const unsigned inner = p.col_major ? p.tiles_m : p.tiles_n;  // tiles along the fast direction
        const unsigned t_slow = (unsigned)tile / inner, t_fast = (unsigned)tile - t_slow * inner;
        tm = p.col_major ? t_fast : t_slow;
        tn = p.col_major ? t_slow : t_fast;
    }
    STAMP(12);
    const int64_t m0 = (int64_t)tm * BM;
    const int n0 = tn * BN;
    // tile-local column → column of C / row of Bm.  GATE: the first half of the tile is a run of h columns, the second
    // half the run of g columns that gates them (both halves are whole: gateF % (BN/2) == 0).
    auto gcol = [&](int c) {
        if constexpr (GATE == 1) return c < BN / 2 ? tn * (BN / 2) + c : p.gateF + tn * (BN / 2) + (c - BN / 2);
        else return n0 + c;
    };

    const T* Ag = static_cast<const T*>(p.Am);
    const T* Bg = static_cast<const T*>(p.Bm);
    const T* Fg = static_cast<const T*>(p.Fp);
    float* Pout = p.P;
    int Kc = p.Kc;
    bool write_p = p.P != nullptr && tn == 0;
    if constexpr (MAIN) {
        if (p.tile_part != nullptr) {
            const int e = p.tile_part[tn];
            const int part = e & 0xffff;
            Fg += (int64_t)part * kRP * Kc;
            if (Pout != nullptr) Pout += (int64_t)part * p.M * p.r;
            write_p = p.P != nullptr && (e >> 16) != 0;
        } else if (KP == 1 && p.n_parts > 0) {  // equal parts over the output columns (lora_gemm_parts, forward form)
            const int part = n0 / p.part_n;
            Fg += (int64_t)part * kRP * Kc;
            if (Pout != nullptr) Pout += part * p.r;
            write_p = p.P != nullptr && n0 == part * p.part_n;
        }
    } else {
        if (p.part_table != nullptr) {
            const int64_t* e = p.part_table + (int64_t)tn * 4;
            Ag += e[0];
            Kc = (int)e[1];
            Fg += e[2];
            if (Pout != nullptr) Pout += e[3];
            write_p = p.P != nullptr;
        }
    }

    // ---- staging addresses: thread = (row ld_row (+32·pass), 16-B chunk ld_chunk) ------------
    const int ld_chunk = tid & 7;
    const int ld_row = tid >> 3;
    const T* a_ptr[PA];
    const T* b_ptr[MAIN ? PB : 1];
#pragma unroll
    for (int i = 0; i < PA; ++i) {
        int64_t m = m0 + ld_row + RPP * i;
        if (m > p.M - 1) m = p.M - 1;  // clamp: rows past M are loaded from a valid row, never stored
        a_ptr[i] = Ag + m * p.lda;
    }
    if constexpr (MAIN) {
#pragma unroll
        for (int i = 0; i < PB; ++i) {
            int n = gcol(ld_row + RPP * i);
            if (n > p.Nc - 1) n = p.Nc - 1;
            b_ptr[i] = Bg + (int64_t)n * Kc;
        }
    }
    const T* f_ptr = Fg + (int64_t)(ld_row & 15) * Kc;

    // ---- Q tile (epilogue factor): BN packed rows of 16 values, 16-B chunks, rows clamped at the edge -----
    // Ring kernels fetch it by DMA during the LAST K-step into the ring buffer that is free by then (issue_q below): it
    // costs no LDS of its own — which is what lets two 128×160 workgroups share a CU — and stays off the prologue.
    constexpr int CPRQ = kRP * (int)sizeof(T) / 16;  // chunks per packed row (2 for 16-bit, 4 for f32)
    constexpr int QOFF = WN * BM * kSPS * 4;         // behind the P image when both land in the same buffer
    auto issue_q = [&](char* dst, int part = 0) {
        const T* Qg = static_cast<const T*>(p.Qp) + (int64_t)part * p.Nc * kRP;
        for (int base = wave * 64; base < BN * CPRQ; base += NT) {
            const int idx = base + lane;
            const int n = idx / CPRQ, ch = idx - n * CPRQ;
            const int nn = gcol(n) < p.Nc ? gcol(n) : p.Nc - 1;
            glds16(Qg + (int64_t)nn * kRP + ch * VEC, dst + base * 16);
        }
    };
    if constexpr (MAIN && !PIPE) {
        const T* Qg = static_cast<const T*>(p.Qp);
        for (int idx = tid; idx < BN * CPRQ; idx += NT) {
            const int n = idx / CPRQ, ch = idx - n * CPRQ;
            const int nn = gcol(n) < p.Nc ? gcol(n) : p.Nc - 1;
            *reinterpret_cast<Chunk<T>*>(sQ + idx * 16) =
                *reinterpret_cast<const Chunk<T>*>(Qg + (int64_t)nn * kRP + ch * VEC);
        }
    }

    // bias of the lane's 4·NI output columns (4 consecutive columns per fragment: ONE 8-/16-byte load each), fetched
    // now so the epilogue never waits on it
    float bias_v[MAIN ? NI : 1][4];
    if constexpr (MAIN) {
        if (p.bias != nullptr) {
            const T* bias = static_cast<const T*>(p.bias);
#pragma unroll
            for (int ni = 0; ni < NI; ++ni) {
                int col = gcol(wn * WTN + ni * 16 + lq * 4);
                if (col > p.Nc - 4) col = p.Nc - 4;  // Nc % VEC == 0 here: a group of 4 is inside or past the edge
                const Quad<T> q = *reinterpret_cast<const Quad<T>*>(bias + col);
#pragma unroll
                for (int reg = 0; reg < 4; ++reg) bias_v[ni][reg] = to_f32<T>(q.v[reg]);
            }
        }
    }
    STAMP(13);

    f32x4 acc[MAIN ? MI : 1][MAIN ? NI : 1];
    f32x4 pacc[MI];
    f32x4 pacc_x[KP > 1 ? KP - 1 : 1][MI];  // KP kernels: parts 1.. (part 0 uses pacc)
#pragma unroll
    for (int i = 0; i < (MAIN ? MI : 1); ++i)
#pragma unroll
        for (int j = 0; j < (MAIN ? NI : 1); ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < MI; ++i) pacc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int g = 0; g < (KP > 1 ? KP - 1 : 1); ++g)
#pragma unroll
        for (int i = 0; i < MI; ++i) pacc_x[g][i] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int steps_per_part = KP > 1 ? p.part_n / BK : 1;  // (part_n % BK == 0: checked on the host)

    // ---- one K-step of MFMA work out of a staged buffer ------------------------------------
    auto compute = [&](const char* st, int kt) {
        const char* sA = st;
        const char* sB = st + OFF_B;
        const char* sF = st + OFF_F;
        if constexpr (!F32) {
            using Frag = typename Mfma<T>::Frag;
            // (round 3, measured: issuing ALL fragment reads of the K-step up front, both 32-wide halves, so that the MFMAs wait
            //  on counted lgkmcnt instead of the lgkmcnt(0) hipcc puts in front of every MFMA group here, changes nothing —
            //  1024×1280×1280 15.5 / 15.4 / 15.4 µs for none / small tiles / all tiles, every other shape within 1 %: the
            //  lone-workgroup main loop is not paced by exposed LDS latency)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                const int chunk = ks * 4 + lq;
                Frag af[MI];
#pragma unroll
                for (int mi = 0; mi < MI; ++mi)
                    af[mi] = *reinterpret_cast<const Frag*>(sA + lds_off(wm * WTM + mi * 16 + l15, chunk));
                if ((kt * 2 + ks) % WN == wn) {
                    const Frag ff = *reinterpret_cast<const Frag*>(sF + lds_off(l15, chunk));
                    if constexpr (KP > 1) {
                        const int part = kt / steps_per_part;  // wave-uniform: the K-step lies inside one part's column run
                        if (part == 0) {
#pragma unroll
                            for (int mi = 0; mi < MI; ++mi) pacc[mi] = Mfma<T>::run(ff, af[mi], pacc[mi]);
                        }
#pragma unroll
                        for (int g = 1; g < KP; ++g)
                            if (part == g) {
#pragma unroll
                                for (int mi = 0; mi < MI; ++mi) pacc_x[g - 1][mi] = Mfma<T>::run(ff, af[mi], pacc_x[g - 1][mi]);
                            }
                    } else {
#pragma unroll
                    for (int mi = 0; mi < MI; ++mi) pacc[mi] = Mfma<T>::run(ff, af[mi], pacc[mi]);
                    }
                }
                if constexpr (MAIN) {
                    Frag bf[NI];
#pragma unroll
                    for (int ni = 0; ni < NI; ++ni)
                        bf[ni] = *reinterpret_cast<const Frag*>(sB + lds_off(wn * WTN + ni * 16 + l15, chunk));
#pragma unroll
                    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                        for (int ni = 0; ni < NI; ++ni) acc[mi][ni] = Mfma<T>::run(bf[ni], af[mi], acc[mi][ni]);
                }
            }
        } else {
            // f32: lane (row l15, group lq) holds chunk lq+4h = 4 consecutive k; element e of every
            // lane feeds k-step e, so A and B agree on k per lane group (the sum order is free).
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int chunk = lq + 4 * h;
                f32x4 af[MI];
#pragma unroll
                for (int mi = 0; mi < MI; ++mi)
                    af[mi] = *reinterpret_cast<const f32x4*>(sA + lds_off(wm * WTM + mi * 16 + l15, chunk));
                if ((kt * 2 + h) % WN == wn) {
                    const f32x4 ff = *reinterpret_cast<const f32x4*>(sF + lds_off(l15, chunk));
#pragma unroll
                    for (int e = 0; e < 4; ++e)
#pragma unroll
                        for (int mi = 0; mi < MI; ++mi)
                            pacc[mi] = __builtin_amdgcn_mfma_f32_16x16x4f32(ff[e], af[mi][e], pacc[mi], 0, 0, 0);
                }
                if constexpr (MAIN) {
                    f32x4 bf[NI];
#pragma unroll
                    for (int ni = 0; ni < NI; ++ni)
                        bf[ni] = *reinterpret_cast<const f32x4*>(sB + lds_off(wn * WTN + ni * 16 + l15, chunk));
#pragma unroll
                    for (int e = 0; e < 4; ++e)
#pragma unroll
                        for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                            for (int ni = 0; ni < NI; ++ni)
                                acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x4f32(bf[ni][e], af[mi][e],
                                                                                  acc[mi][ni], 0, 0, 0);
                }
            }
        }
    };

    int nk = (Kc + BK - 1) / BK;
    const int kt0 = slice * p.steps_per_slice;  // first K-step of this workgroup (0 unless split-K)
    if constexpr (SPLITK) nk = nk - kt0 < p.steps_per_slice ? nk - kt0 : p.steps_per_slice;
    int buf = 0;  // ring position: after the loop, the buffer that would be filled next — i.e. a FREE one
    if constexpr (PIPE) {
        // ---- LDS-DMA ring.  Lane (row, physical chunk c') fetches logical chunk c' ^ (row & 7): the DMA
        // writes lane-linearly, so the swizzle lives on the source address and on the fragment reads.
        constexpr int L = PA + (MAIN ? PB : 0);  // DMA loads per lane per stage (+1 on the two factor-loading waves)
        static_assert(kStages <= 6 && 4 * (L + 1) < 64, "counted waits cover up to four newer stages; vmcnt is a 6-bit field");
        const int src_off = (ld_chunk ^ (ld_row & 7)) * VEC;
        const int wave_rows = wave * 8 * kRowBytes;
        auto issue = [&](int kt, int buf) {
            char* st = smem + buf * STAGE + wave_rows;
            const int k0 = (kt0 + kt) * BK + src_off;
#pragma unroll
            for (int i = 0; i < PA; ++i) glds16(a_ptr[i] + k0, st + RPP * i * kRowBytes);
            if constexpr (MAIN) {
#pragma unroll
                for (int i = 0; i < PB; ++i) glds16(b_ptr[i] + k0, st + OFF_B + RPP * i * kRowBytes);
            }
            if (wave < 2) glds16(f_ptr + k0, st + OFF_F);
        };
        constexpr int DIST = kStages - 1;  // K-steps in flight ahead of the one being multiplied
#pragma unroll
        for (int i = 0; i < DIST; ++i)
            if (i < nk) issue(i, i);
        STAMP(2);
        for (int kt = 0; kt < nk; ++kt) {
            // stage kt has landed for this wave once only the loads of the stages issued after it are outstanding:
            // `ahead` of them, L loads each (L + 1 on the two waves that also fetch the factor rows)
            const int left = nk - 1 - kt;
            const int ahead = left < DIST - 1 ? left : DIST - 1;
            const bool fw = wave < 2;
            switch (ahead) {
                case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
                case 1:
                    if (fw) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(L + 1) : "memory");
                    else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(L) : "memory");
                    break;
                case 2:
                    if (fw) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * (L + 1)) : "memory");
                    else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * L) : "memory");
                    break;
                case 3:
                    if (fw) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(3 * (L + 1)) : "memory");
                    else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(3 * L) : "memory");
                    break;
                default:
                    if (fw) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(4 * (L + 1)) : "memory");
                    else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(4 * L) : "memory");
                    break;
            }
            __builtin_amdgcn_s_barrier();  // every wave's part of stage kt is in; stage kt-1 is fully read
            if (kt == 0) STAMP(3);
            char* const refill = smem + (buf >= 1 ? buf - 1 : kStages - 1) * STAGE;  // the buffer read last step
            if (kt + DIST < nk) {
                if (!(p.dbg & 1)) issue(kt + DIST, buf >= 1 ? buf - 1 : kStages - 1);
            } else if (MAIN && kt == nk - 1) {  // (every K-slice of a split tile: any of them may turn out to be the last arriver)
                sQ = refill + QOFF;  // nothing left to prefetch: the epilogue's Q tile takes the free buffer
                issue_q(sQ);
            }
            // (round 3, measured: the refill's DMA instructions spread BETWEEN this step's MFMAs instead of issued as one burst
            //  here — same counted waits, same buffers — is slower on every shape: 4096×640×640 12.6 → 14.0 µs, 1024×1280×1280
            //  equal, and the 128-row tiles spill (address registers stay live across the MFMA stream): 35.7 → 51.6 µs)
            if (!(p.dbg & 2)) compute(smem + buf * STAGE, kt);
            buf = buf + 1 == kStages ? 0 : buf + 1;
        }
    } else {
        // ---- register-staged fallback (ragged contraction): unconditional clamped loads, zero on write ----
        Chunk<T> ra[PA];
        Chunk<T> rb[MAIN ? PB : 1];
        Chunk<T> rfc;
        int staged_k = 0;
        auto load_step = [&](int k0) {
            const int kc = k0 + ld_chunk * VEC;
            const int kld = kc < Kc ? kc : Kc - VEC;  // Kc % VEC == 0 on this path
            staged_k = kc;
#pragma unroll
            for (int i = 0; i < PA; ++i) ra[i] = *reinterpret_cast<const Chunk<T>*>(a_ptr[i] + kld);
            if constexpr (MAIN) {
#pragma unroll
                for (int i = 0; i < PB; ++i) rb[i] = *reinterpret_cast<const Chunk<T>*>(b_ptr[i] + kld);
            }
            rfc = *reinterpret_cast<const Chunk<T>*>(f_ptr + kld);
        };
        auto store_step = [&]() {
            if (staged_k >= Kc) {
#pragma unroll
                for (int e = 0; e < VEC; ++e) {
#pragma unroll
                    for (int i = 0; i < PA; ++i) ra[i].v[e] = from_f32<T>(0.f);
                    if constexpr (MAIN) {
#pragma unroll
                        for (int i = 0; i < PB; ++i) rb[i].v[e] = from_f32<T>(0.f);
                    }
                    rfc.v[e] = from_f32<T>(0.f);
                }
            }
#pragma unroll
            for (int i = 0; i < PA; ++i)
                *reinterpret_cast<Chunk<T>*>(smem + lds_off(ld_row + RPP * i, ld_chunk)) = ra[i];
            if constexpr (MAIN) {
#pragma unroll
                for (int i = 0; i < PB; ++i)
                    *reinterpret_cast<Chunk<T>*>(smem + OFF_B + lds_off(ld_row + RPP * i, ld_chunk)) = rb[i];
            }
            if (ld_row < kRP) *reinterpret_cast<Chunk<T>*>(smem + OFF_F + lds_off(ld_row, ld_chunk)) = rfc;
        };
        load_step(kt0 * BK);
        for (int kt = 0; kt < nk; ++kt) {
            if (kt > 0) __syncthreads();
            store_step();
            __syncthreads();
            if (kt + 1 < nk) load_step((kt0 + kt + 1) * BK);
            compute(smem, kt);
        }
    }

    STAMP(4);
    // ---- epilogue 1: combine the two partial P tiles through LDS --------------------------
    // Ring kernels put the P image into the ring buffer that is FREE after the last step (nothing in flight, last
    // read one step ago behind a barrier) and, for 16-bit types, the C tile into the buffer of the last step (free
    // once every wave has passed the barrier below): two workgroup barriers in the whole epilogue instead of four.
    constexpr bool FREEBUF = PIPE;
    constexpr bool FASTC = PIPE && MAIN && !F32;
    static_assert(!FREEBUF || QOFF + (MAIN ? BN * kRP * (int)sizeof(T) : 0) <= STAGE, "P image + Q tile must fit one ring buffer");
    char* const last_buf = smem + (buf == 0 ? kStages - 1 : buf - 1) * STAGE;
    if constexpr (FREEBUF) {
        sP = reinterpret_cast<float*>(smem + buf * STAGE);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's share of the Q tile has landed (barrier below: everyone's)
    } else {
        __syncthreads();
    }
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) {
        const int row = wm * WTM + mi * 16 + l15;
        *reinterpret_cast<f32x4*>(&sP[(wn * BM + row) * kSPS + lq * 4]) = pacc[mi];
    }
    __syncthreads();

    if constexpr (SPLITK) {
        static_assert(PIPE && GATE == 0, "split-K exists on the ungated ring kernels");
        {
            // In-launch split-K combine (cdna_hip_programming.md, "Projection GEMM" item 2, the sc1 form): every byte that
            // crosses workgroups is stored WRITE-THROUGH (sc1) and loaded sc1 — the XCDs' L2s are not coherent with each
            // other and a CU's L1 is never refreshed, so plain accesses would need an agent-scope release in every slice
            // (an L2 write-back: measured 4–10× the whole kernel) and an acquire in the reducer.  Order: sc1 stores → every
            // wave drains its stores (vmcnt(0)) → workgroup barrier → ONE lane draws the ticket (agent-scope atomic).
            typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
            const int S = p.splitk;
            const __amdgpu_buffer_rsrc_t rc = __builtin_amdgcn_make_buffer_rsrc(
                p.ws_c + (int64_t)tile * S * (BM * BN), 0, MAIN ? S * BM * BN * 4 : 0, 0x00020000);
            const __amdgpu_buffer_rsrc_t rp = __builtin_amdgcn_make_buffer_rsrc(
                p.ws_p + (int64_t)tile * S * (BM * kRP), 0, S * BM * kRP * 4, 0x00020000);
            // (a) this slice's partial sums: P (all 16 rank slots; every column tile keeps its own copy, its last arriver
            // needs it) and the fp32 accumulators in the lanes' own register order — perfectly coalesced 16-byte stores,
            // no edge handling (rows / columns past the edge are never stored to C)
            {
                const int half = tid & 1;
                for (int row = tid >> 1; row < BM; row += NT / 2) {
                    f32x4 a = {0.f, 0.f, 0.f, 0.f}, b = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int w = 0; w < WN; ++w) {
                        const f32x4* src = reinterpret_cast<const f32x4*>(&sP[(w * BM + row) * kSPS + half * 8]);
                        a += src[0];
                        b += src[1];
                    }
                    const int off = ((slice * BM + row) * kRP + half * 8) * 4;
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, a), rp, off, 0, 16);
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, b), rp, off + 16, 0, 16);
                }
            }
            if constexpr (MAIN) {
#pragma unroll
                for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                    for (int ni = 0; ni < NI; ++ni)
                        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, acc[mi][ni]), rc,
                                                               (slice * (BM * BN) + ((mi * NI + ni) * NT + tid) * 4) * 4, 0, 16);
            }
            // (b) ticket.  The flag lives in the buffer of the last K-step, free since the barrier behind the P exchange.
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // EVERY storing wave drains before the barrier
            __syncthreads();
            volatile unsigned* flag = reinterpret_cast<volatile unsigned*>(last_buf);
            if (tid == 0) {
                const unsigned old = __hip_atomic_fetch_add(p.tickets + tile, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                const bool last = old == (unsigned)(S - 1);
                if (last) __hip_atomic_store(p.tickets + tile, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                *flag = last ? 1u : 0u;
            }
            __syncthreads();
            STAMP(14);  // (diagnostic build: slab stored, ticket drawn)
            if (*flag == 0u) {
                STAMP(8);
                STAMP_WALL(9);
                return;
            }
            // (c) the last arriver: totals over the slices in index order (its own contribution included, from memory, so
            // that the order of the additions never depends on who came last).  EVERY load of a slab is an sc1 load.
            if constexpr (MAIN) {
#pragma unroll
                for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                    for (int ni = 0; ni < NI; ++ni) acc[mi][ni] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll 1
                for (int s = 0; s < S; ++s) {
                    u32x4 v[MI][NI];
#pragma unroll
                    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                        for (int ni = 0; ni < NI; ++ni)
                            v[mi][ni] = __builtin_amdgcn_raw_buffer_load_b128(
                                rc, (s * (BM * BN) + ((mi * NI + ni) * NT + tid) * 4) * 4, 0, 16);
#pragma unroll
                    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                        for (int ni = 0; ni < NI; ++ni) acc[mi][ni] += __builtin_bit_cast(f32x4, v[mi][ni]);
                }
            }
            {
                // total P of the tile's rows goes into wave slot 0 of the P image, zeros into the other slots: everything
                // below sums the slots exactly as it does for an unsplit tile
                const int half = tid & 1;
                for (int row = tid >> 1; row < BM; row += NT / 2) {
                    f32x4 a = {0.f, 0.f, 0.f, 0.f}, b = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 1
                    for (int s = 0; s < S; ++s) {
                        const int off = ((s * BM + row) * kRP + half * 8) * 4;
                        a += __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rp, off, 0, 16));
                        b += __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rp, off + 16, 0, 16));
                    }
                    f32x4* dst = reinterpret_cast<f32x4*>(&sP[row * kSPS + half * 8]);
                    dst[0] = a;
                    dst[1] = b;
#pragma unroll
                    for (int w = 1; w < WN; ++w) {
                        f32x4* z = reinterpret_cast<f32x4*>(&sP[(w * BM + row) * kSPS + half * 8]);
                        z[0] = f32x4{0.f, 0.f, 0.f, 0.f};
                        z[1] = f32x4{0.f, 0.f, 0.f, 0.f};
                    }
                }
            }
            __syncthreads();
        }
    }
    constexpr bool PACKED_P = MAIN && !F32 && PIPE;  // the rank-r term's operand is built ONCE per row, cooperatively
    constexpr int kPkStride = 80;                     // bytes per row of the packed image [hi j0..15 | lo j0..15] (+16 pad)
    char* const sPk = reinterpret_cast<char*>(sP) + QOFF + (MAIN ? BN * kRP * (int)sizeof(T) : 0);
    static_assert(KP == 1 || (PACKED_P && !SPLITK && GATE == 0), "part-wise contractions exist on the plain 16-bit ring kernels");
    // thread = (row, half of the 16 rank slots): sums the wave partials, writes P out, and leaves s·P split into a
    // high and a low 16-bit part in MFMA operand order — every wave then fetches a fragment with ONE 16-byte read
    // instead of rebuilding it from 2·WN fp32 reads and 40 conversions per lane (4× redundantly across the workgroup)
    auto build_packed = [&](int part) {
        if constexpr (PACKED_P) {
        static_assert(!PACKED_P || QOFF + BN * kRP * (int)sizeof(T) + BM * kPkStride <= STAGE, "P partials + Q tile + packed P fit one buffer");
        const int half = tid & 1;
        for (int row = tid >> 1; row < BM; row += NT / 2) {
            f32x4 a = {0.f, 0.f, 0.f, 0.f}, b = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int w = 0; w < WN; ++w) {
                const f32x4* src = reinterpret_cast<const f32x4*>(&sP[(w * BM + row) * kSPS + half * 8]);
                a += src[0];
                b += src[1];
            }
            const float v[8] = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
            if (write_p && m0 + row < p.M) {
#pragma unroll
                for (int e = 0; e < 8; ++e)
                    if (half * 8 + e < p.r) Pout[(m0 + row) * p.ldp + part * p.r + half * 8 + e] = v[e];
            }
            Chunk<T> hi, lo;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float sv = v[e] * p.scale;
                hi.v[e] = from_f32<T>(sv);
                lo.v[e] = from_f32<T>(sv - to_f32<T>(hi.v[e]));
            }
            *reinterpret_cast<Chunk<T>*>(sPk + row * kPkStride + half * 16) = hi;
            *reinterpret_cast<Chunk<T>*>(sPk + row * kPkStride + 32 + half * 16) = lo;
        }
        __syncthreads();
        }
    };
    // acc += s·P·Qᵀ as ONE extra MFMA K-step out of the packed record and the Q tile at `q_base`
    auto rank_step_packed = [&](const char* q_base) {
        if constexpr (PACKED_P) {
            using Frag = typename Mfma<T>::Frag;
            const int j0 = 8 * (lq & 1);
            Frag qf[NI];
#pragma unroll
            for (int ni = 0; ni < NI; ++ni)
                qf[ni] = *reinterpret_cast<const Frag*>(q_base + ((wn * WTN + ni * 16 + l15) * kRP + j0) * (int)sizeof(T));
            Frag pf[MI];
#pragma unroll
            for (int mi = 0; mi < MI; ++mi)
                pf[mi] = *reinterpret_cast<const Frag*>(sPk + (wm * WTM + mi * 16 + l15) * kPkStride + lq * 16);
#pragma unroll
            for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                for (int ni = 0; ni < NI; ++ni) acc[mi][ni] = Mfma<T>::run(qf[ni], pf[mi], acc[mi][ni]);
        }
    };
    if constexpr (PACKED_P) {
        if constexpr (KP > 1) {
            // Parts 1.. of the epilogue factor go into the buffer of the last K-step (free since the barrier behind the P
            // exchange; the C tile lands there only after the last part): fetched by DMA while part 0 is worked on.
            constexpr int QB = BN * kRP * (int)sizeof(T);
            static_assert((KP - 1) * QB <= STAGE, "the epilogue factors of parts 1.. fit the last step's buffer");
#pragma unroll
            for (int g = 1; g < KP; ++g) issue_q(last_buf + (g - 1) * QB, g);
            build_packed(0);
            rank_step_packed(sQ);
#pragma unroll
            for (int g = 1; g < KP; ++g) {
                __syncthreads();  // every wave is done with the previous part's partials and packed record
#pragma unroll
                for (int mi = 0; mi < MI; ++mi) {
                    const int row = wm * WTM + mi * 16 + l15;
                    *reinterpret_cast<f32x4*>(&sP[(wn * BM + row) * kSPS + lq * 4]) = pacc_x[g - 1][mi];
                }
                if (g == 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's share of Q_1.. has landed
                __syncthreads();
                build_packed(g);
                rank_step_packed(last_buf + (g - 1) * QB);
            }
            __syncthreads();  // the Q tiles in the last step's buffer are dead: the C tile may land there
        } else {
            build_packed(0);
        }
    } else if (write_p) {
        const int half = tid & 1;
        for (int row = tid >> 1; row < BM && m0 + row < p.M; row += NT / 2) {
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int j = half * 8 + e;
                if (j < p.r) {
                    float v = 0.f;
#pragma unroll
                    for (int w = 0; w < WN; ++w) v += sP[(w * BM + row) * kSPS + j];
                    Pout[(m0 + row) * p.ldp + j] = v;
                }
            }
        }
    }

    STAMP(5);
    if constexpr (MAIN) {
        // ---- epilogue 2: acc += s·P·Qᵀ as one extra MFMA K-step ---------------------------
        if constexpr (!F32) {
            using Frag = typename Mfma<T>::Frag;
            if constexpr (PACKED_P) {
                if constexpr (KP == 1) rank_step_packed(sQ);
            } else {
            const int j0 = 8 * (lq & 1);
            Frag qf[NI];
#pragma unroll
            for (int ni = 0; ni < NI; ++ni)
                qf[ni] = *reinterpret_cast<const Frag*>(
                    sQ + ((wn * WTN + ni * 16 + l15) * kRP + j0) * (int)sizeof(T));
            constexpr int MG = MI < 4 ? MI : 4;  // row fragments whose P image is fetched together
#pragma unroll
            for (int mg = 0; mg < MI; mg += MG) {
                f32x4 pimg[MG][WN][2];
#pragma unroll
                for (int i = 0; i < MG; ++i) {
                    const int row = wm * WTM + (mg + i) * 16 + l15;
#pragma unroll
                    for (int w = 0; w < WN; ++w) {
                        const f32x4* src = reinterpret_cast<const f32x4*>(&sP[(w * BM + row) * kSPS + j0]);
                        pimg[i][w][0] = src[0];
                        pimg[i][w][1] = src[1];
                    }
                }
#pragma unroll
                for (int i = 0; i < MG; ++i) {
                    Frag pf;
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        float v = 0.f;
#pragma unroll
                        for (int w = 0; w < WN; ++w) v += pimg[i][w][e >> 2][e & 3];
                        v *= p.scale;
                        const T hi = from_f32<T>(v);
                        const T lo = from_f32<T>(v - to_f32<T>(hi));
                        pf[e] = lq < 2 ? hi : lo;
                    }
#pragma unroll
                    for (int ni = 0; ni < NI; ++ni) acc[mg + i][ni] = Mfma<T>::run(qf[ni], pf, acc[mg + i][ni]);
                }
            }
            }
        } else {
            const float* q = reinterpret_cast<const float*>(sQ);
#pragma unroll
            for (int st = 0; st < 4; ++st) {
                const int j = 4 * st + lq;
                float qv[NI];
#pragma unroll
                for (int ni = 0; ni < NI; ++ni) qv[ni] = q[(wn * WTN + ni * 16 + l15) * kRP + j];
#pragma unroll
                for (int mi = 0; mi < MI; ++mi) {
                    const int row = wm * WTM + mi * 16 + l15;
                    float pv = 0.f;
#pragma unroll
                    for (int w = 0; w < WN; ++w) pv += sP[(w * BM + row) * kSPS + j];
                    pv *= p.scale;
#pragma unroll
                    for (int ni = 0; ni < NI; ++ni)
                        acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x4f32(qv[ni], pv, acc[mi][ni], 0, 0, 0);
                }
            }
        }

        STAMP(6);
        // ---- epilogue 3: bias (fp32), transpose through LDS, 16-B row stores -------------
        if (p.bias != nullptr) {
#pragma unroll
            for (int ni = 0; ni < NI; ++ni)
#pragma unroll
                for (int reg = 0; reg < 4; ++reg)
#pragma unroll
                    for (int mi = 0; mi < MI; ++mi) acc[mi][ni][reg] += bias_v[ni][reg];
        }
        constexpr int SC_STRIDE = BN * (int)sizeof(T) + 16;
        // the C tile leaves in EP passes of ROWS rows (one pass when it fits the LDS area it is staged in)
        constexpr int EP = F32 ? 2 : (FASTC && BM * SC_STRIDE > STAGE ? 2 : 1);
        static_assert(WM % EP == 0, "a pass covers whole row waves");
        constexpr int ROWS = BM / EP;
        constexpr int CPR = BN / VEC;  // 16-B chunks per tile row
        static_assert(!FASTC || ROWS * SC_STRIDE <= STAGE, "C tile must fit one ring buffer");
        char* const sC = FASTC ? last_buf : smem;
        T* Cg = static_cast<T*>(p.C);
#pragma unroll
        for (int ep = 0; ep < EP; ++ep) {
            if (!FASTC || ep > 0) __syncthreads();  // sP / sQ (or the previous pass) are dead
            if (EP == 1 || wm / (WM / EP) == ep) {
#pragma unroll
                for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                    for (int ni = 0; ni < NI; ++ni) {
                        // the lane owns 4 consecutive columns of one row: one 8-/16-byte LDS write
                        const int row = (EP == 1 ? wm : wm % (WM / EP)) * WTM + mi * 16 + l15;
                        const int col = wn * WTN + ni * 16 + lq * 4;
                        Quad<T> q;
#pragma unroll
                        for (int reg = 0; reg < 4; ++reg) q.v[reg] = from_f32<T>(acc[mi][ni][reg]);
                        *reinterpret_cast<Quad<T>*>(sC + row * SC_STRIDE + col * (int)sizeof(T)) = q;
                    }
            }
            __syncthreads();
            STAMP(7);
            if constexpr (GATE == 2) {
                // GEGLU BACKWARD in the epilogue of the GEMM that produces its incoming gradient: this launch is
                //     dout[M,F] = dZ[M,Nz]·W2        (backward-input of the linear layer behind the gate: ff.net.2)
                // and the tile in LDS is dout, rounded to the storage type exactly as the separate tensor would be.  Thread =
                // (row, chunk): it loads the h and g chunks of Y = [h | g] (p.C2, read-only) and writes the two halves of
                //     dY[:, :F] = dout·gelu(g),   dY[:, F:] = dout·h·gelu'(g)        (p.C)
                // — the arithmetic of geglu_bwd_kernel on the same values; dout itself never goes to memory.
                // (Per pass of ROWS rows; a chunk count that does not divide over the threads runs its last trip on clamped
                //  indices with the stores predicated — loads are never under a per-lane condition.)
                static_assert(FASTC, "the gate epilogue reads the C tile from a ring buffer");
                constexpr int TOT = ROWS * CPR;
                constexpr int NST = (TOT + NT - 1) / NT;
                constexpr int H0 = (NST + 1) / 2;  // two register batches
                const T* Yg = static_cast<const T*>(p.C2);
                const int F = p.gateF;
#pragma unroll
                for (int hb = 0; hb < 2; ++hb) {
                    const int first = hb == 0 ? 0 : H0, cnt = hb == 0 ? H0 : NST - H0;
                    Chunk<T> dv[H0], hv[H0], gv[H0];
#pragma unroll
                    for (int i = 0; i < H0; ++i) {
                        if (i < cnt) {
                            int idx = tid + (first + i) * NT;
                            if (TOT % NT != 0 && idx > TOT - 1) idx = TOT - 1;
                            const int row = idx / CPR, ch = idx - row * CPR;
                            int64_t m = m0 + ep * ROWS + row;
                            if (m > p.M - 1) m = p.M - 1;  // rows past the end: loaded from a valid row, never stored
                            const T* yrow = Yg + m * (2 * (int64_t)F) + n0 + ch * VEC;
                            hv[i] = *reinterpret_cast<const Chunk<T>*>(yrow);
                            gv[i] = *reinterpret_cast<const Chunk<T>*>(yrow + F);
                            dv[i] = *reinterpret_cast<const Chunk<T>*>(sC + row * SC_STRIDE + ch * 16);
                        }
                    }
#pragma unroll
                    for (int i = 0; i < H0; ++i) {
                        if (i < cnt) {
                            const int idx = tid + (first + i) * NT;
                            const int row = idx / CPR, ch = idx - row * CPR;
                            const int64_t m = m0 + ep * ROWS + row;
                            Chunk<T> dh, dg;
#pragma unroll
                            for (int e = 0; e < VEC; ++e) {
                                const float g = to_f32<T>(gv[i].v[e]), d = to_f32<T>(dv[i].v[e]);
                                dh.v[e] = from_f32<T>(d * gelu_f<T>(g));
                                dg.v[e] = from_f32<T>(d * to_f32<T>(hv[i].v[e]) * gelu_grad_f<T>(g));
                            }
                            if (idx < TOT && m < p.M) {
                                T* drow = Cg + m * (2 * (int64_t)F) + n0 + ch * VEC;
                                *reinterpret_cast<Chunk<T>*>(drow) = dh;
                                *reinterpret_cast<Chunk<T>*>(drow + F) = dg;
                            }
                        }
                    }
                }
                continue;
            }
            if constexpr (GATE == 1) {
                // thread = (row, 16-B chunk of the h half) and the chunk of g behind it: y leaves as it is (when a backward
                // pass will want it), out = h·gelu(g) from the SAME rounded values the separate gate kernel would read
                static_assert(FASTC, "the gate epilogue reads the C tile from a ring buffer");
                constexpr int HC = CPR / 2;                 // chunks per half row
                constexpr int TOT = ROWS * HC;
                constexpr int NG = (TOT + NT - 1) / NT;     // (row, chunk) pairs per thread (last trip clamped / predicated)
                T* Og = static_cast<T*>(p.C2);
                const int F = p.gateF;
                Chunk<T> hv[NG], gv[NG];
#pragma unroll
                for (int i = 0; i < NG; ++i) {
                    int idx = tid + i * NT;
                    if (TOT % NT != 0 && idx > TOT - 1) idx = TOT - 1;
                    const int row = idx / HC, ch = idx - row * HC;
                    hv[i] = *reinterpret_cast<const Chunk<T>*>(sC + row * SC_STRIDE + ch * 16);
                    gv[i] = *reinterpret_cast<const Chunk<T>*>(sC + row * SC_STRIDE + (HC + ch) * 16);
                }
                if (Cg != nullptr) {
#pragma unroll
                    for (int i = 0; i < NG; ++i) {
                        const int idx = tid + i * NT;
                        const int row = idx / HC, ch = idx - row * HC;
                        const int64_t m = m0 + ep * ROWS + row;
                        const int col = tn * (BN / 2) + ch * VEC;
                        if (idx < TOT && m < p.M) {
                            *reinterpret_cast<Chunk<T>*>(Cg + m * p.Nc + col) = hv[i];
                            *reinterpret_cast<Chunk<T>*>(Cg + m * p.Nc + F + col) = gv[i];
                        }
                    }
                }
#pragma unroll
                for (int i = 0; i < NG; ++i) {
                    const int idx = tid + i * NT;
                    const int row = idx / HC, ch = idx - row * HC;
                    const int64_t m = m0 + ep * ROWS + row;
                    Chunk<T> o;
#pragma unroll
                    for (int e = 0; e < VEC; ++e) o.v[e] = from_f32<T>(to_f32<T>(hv[i].v[e]) * gelu_f<T>(to_f32<T>(gv[i].v[e])));
                    if (idx < TOT && m < p.M) *reinterpret_cast<Chunk<T>*>(Og + m * F + tn * (BN / 2) + ch * VEC) = o;
                }
                continue;
            }
            constexpr int NST = ROWS * CPR / NT;  // 16-B chunks per thread
            static_assert(ROWS * CPR % NT == 0, "tile rows must split evenly over the threads");
            Chunk<T> out[NST];
#pragma unroll
            for (int i = 0; i < NST; ++i) {
                const int idx = tid + i * NT;
                const int row = idx / CPR, ch = idx - row * CPR;
                out[i] = *reinterpret_cast<const Chunk<T>*>(sC + row * SC_STRIDE + ch * 16);
            }
#pragma unroll
            for (int i = 0; i < NST; ++i) {
                const int idx = tid + i * NT;
                const int row = idx / CPR, ch = idx - row * CPR;
                const int64_t m = m0 + ep * ROWS + row;
                const int col = n0 + ch * VEC;
                if (m < p.M && col < p.Nc) *reinterpret_cast<Chunk<T>*>(Cg + m * p.Nc + col) = out[i];
            }
        }
    }
    STAMP(8);
    STAMP_WALL(9);
}

// Shape-agnostic path (unaligned sizes or r > 16): correct, not fast.  F/Q read as fp32 masters.
struct GenericParams {
    const void* Am;
    const void* Bm;
    const void* bias;
    const float* F;
    int64_t f_sr, f_sk;
    const float* Q;
    int64_t q_sn, q_sj;
    void* C;
    float* P;
    int64_t M;
    int Kc, Nc, r;
    float scale;
};
template <typename T>
__global__ void lora_skinny_generic_kernel(GenericParams p) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= p.M * p.r) return;
    const int64_t m = idx / p.r;
    const int j = (int)(idx - m * p.r);
    const T* a = static_cast<const T*>(p.Am) + m * p.Kc;
    float s = 0.f;
    for (int k = 0; k < p.Kc; ++k) s += to_f32<T>(a[k]) * to_f32<T>(from_f32<T>(p.F[j * p.f_sr + k * p.f_sk]));
    p.P[idx] = s;
}
template <typename T>
__global__ void lora_gemm_generic_kernel(GenericParams p) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= p.M * p.Nc) return;
    const int64_t m = idx / p.Nc;
    const int n = (int)(idx - m * p.Nc);
    const T* a = static_cast<const T*>(p.Am) + m * p.Kc;
    const T* b = static_cast<const T*>(p.Bm) + (int64_t)n * p.Kc;
    float s = 0.f;
    for (int k = 0; k < p.Kc; ++k) s += to_f32<T>(a[k]) * to_f32<T>(b[k]);
    float l = 0.f;
    for (int j = 0; j < p.r; ++j)
        l += p.P[m * p.r + j] * to_f32<T>(from_f32<T>(p.Q[(int64_t)n * p.q_sn + j * p.q_sj]));
    s += p.scale * l;
    if (p.bias) s += to_f32<T>(static_cast<const T*>(p.bias)[n]);
    static_cast<T*>(p.C)[idx] = from_f32<T>(s);
}

// Packed factors, both orientations per factor (dtype T, rank padded to 16 with zeros):
//   Apack = [ A16 [16,K] : A16[j,k] = A[j,k] | At16 [K,16] : At16[k,j] = A[j,k] ]      (32·K elements)
//   Bpack = [ Bt16[16,N] : Bt16[j,n] = B[n,j] | B16 [N,16] : B16[n,j] = B[n,j] ]       (32·N elements)
// The [16,len] halves are the main-loop factor tiles (F), the [len,16] halves the epilogue factors (Q).
template <typename T>
__device__ __forceinline__ void pack_one(const float* A, const float* B, T* Apack, T* Bpack, int K, int N, int r,
                                         int which, int start, int step) {
    const int len = which == 0 ? K : N;
    T* dst = which == 0 ? Apack : Bpack;
    for (int idx = start; idx < kRP * len; idx += step) {
        const int j = idx / len, c = idx - j * len;
        const int jj = j < r ? j : r - 1;
        const float v = which == 0 ? A[(int64_t)jj * K + c] : B[(int64_t)c * r + jj];
        const T t = from_f32<T>(j < r ? v : 0.f);
        dst[idx] = t;                                   // [16, len]
        dst[(int64_t)kRP * len + (int64_t)c * kRP + j] = t;  // [len, 16]
    }
}
template <typename T>
__global__ __launch_bounds__(256) void pack_factor_kernel(const float* A, const float* B, T* Apack, T* Bpack, int K,
                                                          int N, int r) {
    pack_one<T>(A, B, Apack, Bpack, K, N, r, blockIdx.y, blockIdx.x * 256 + threadIdx.x, gridDim.x * 256);
}

// All layers of a slab in one launch.  table[l] = {a_off, b_off, K, N, r, apack_off, bpack_off, 0}: element
// offsets of A[r,K] / B[N,r] inside `params` (fp32) and of Apack / Bpack inside `packed` (T).
template <typename T>
__global__ __launch_bounds__(256) void pack_factors_batched_kernel(const int64_t* table, const float* params,
                                                                   T* packed) {
    const int64_t* e = table + (int64_t)(blockIdx.y >> 1) * 8;
    const int K = (int)e[2], N = (int)e[3], r = (int)e[4];
    if (r > kRP) return;
    pack_one<T>(params + e[0], params + e[1], packed + e[5], packed + e[6], K, N, r, blockIdx.y & 1,
                blockIdx.x * 256 + threadIdx.x, gridDim.x * 256);
}

// Split-K plan.  Splitting buys chip occupancy with a combine that costs the tile ≈ 5–6 µs (write-through slab stores, the
// ticket's round trip, the last arriver's serial slab reads at the cross-XCD rate), so it pays only for LONG contractions on
// under-filled grids: measured (tools/splitk_sweep.sh, weights cold) the 20-K-step 1280-wide projections LOSE 2–3 µs at
// every slice count (1024×1280×1280: 15.7 → 18.5 µs), the 60-K-step grouped q/k/v backward gains 5–8 µs (35 → 30, 26 → 19),
// the 80/160-K-step GEGLU `proj` backward 18–30 µs (66 → 48, 84 → 54, 62 → 31).  Returns the slice count (1 = off) and the
// row-tile height of the split launch: ≈ 480 workgroups, 64-row tiles on the small grids (two per CU overlap each other's
// phases), at least 10 K-steps left per slice.
struct SplitPlan {
    int S, bm;
};
// Slice → XCD affinity (LORA_SPLIT_AFFINITY=1; off by default).  Measured in round 4 (profiles/r04_splitk_xcd_affinity_ab.log,
// weights cold): it cuts the operand fetch to |Am| + |Bm| as intended, but the S slices of a tile then sit on S different XCDs,
// so the last arriver's slab reads cross the fabric instead of hitting the L2 that holds the neighbours' write-through
// stores — 4096×5120→640 49.5 → 64.0 µs, 1024×10240→1280 57.7 → 71.4, 256×10240→1280 29.8 → 35.1, grouped q/k/v backward at
// 1024 rows 30.6 → 39.7: these launches are paced by the combine and their fixed phases, not by operand traffic.
bool split_affinity() {
    static const int env = [] { const char* e = getenv("LORA_SPLIT_AFFINITY"); return e ? atoi(e) : 0; }();
    return env != 0;
}
constexpr int kTicketBytes = LORA_GEMM_WS_TICKET_BYTES;  // ticket header of the workspace: one u32 per output tile
SplitPlan plan_splitk(int64_t M, int Kc, int Nc, int esize) {
    static const int env = [] { const char* e = getenv("LORA_SPLITK"); return e ? atoi(e) : -1; }();
    static const int env_bm = [] { const char* e = getenv("LORA_SPLIT_BM"); return e ? atoi(e) : 0; }();
    static const int env_min = [] { const char* e = getenv("LORA_SPLIT_MINSTEPS"); return e ? atoi(e) : 0; }();
    SplitPlan off{1, 128};
    if (env == 0) return off;
    const int nk = (Kc * esize + kRowBytes - 1) / kRowBytes;
    const int64_t tiles128 = ((M + 127) / 128) * ((Nc + 127) / 128);
    const int64_t tiles64 = ((M + 63) / 64) * ((Nc + 127) / 128);
    if ((Nc & 7) != 0 || (Kc * esize) % kRowBytes != 0 || tiles128 >= 192) return off;
    if (env <= 0 && nk < 48) return off;
    int bm = tiles128 <= 96 ? 64 : 128;
    // round 5 (profiles/r05_splitk_plan_sweep.log, weights cold): a split launch is paced by its MAIN LOOP — 74 % of the 48-µs
    // 1024×10240→1280 launch, 0.67 µs per K-step at two workgroups per CU: the L1-fill bound of the unsplit kernels — not by the
    // combine (slab store + ticket 2 µs, the last arriver's sum 3.7 µs).  So the longest contractions take the 128-row tile (0.65× the
    // L1 bytes per flop) with more slices where that still fills the chip: 80 tiles × 6 slices, 58.2 → 53.5 µs.
    if (tiles128 >= 64 && nk >= 128) bm = 128;
    if (env_bm == 64 || env_bm == 128) bm = env_bm;
    const int64_t tiles = bm == 64 ? tiles64 : tiles128;
    if (tiles > kTicketBytes / 4) return off;
    int S = (int)((480 + tiles / 2) / tiles);
    if (S > 8) S = 8;
    const int min_steps = env_min > 0 ? env_min : 10;
    while (S > 1 && nk / S < min_steps) --S;
    if (env > 1) S = env > 8 ? 8 : env;
    while (S > 1 && (S - 1) * ((nk + S - 1) / S) >= nk) --S;  // every slice owns at least one K-step
    if (split_affinity() && env <= 1) {
        // slice → XCD affinity needs S | 8: the nearest power of two that still leaves min_steps per slice (3 → 4, 6 → 8 / 4)
        int S2 = S >= 6 ? 8 : (S >= 3 ? 4 : S);
        while (S2 > 1 && nk / S2 < min_steps) S2 >>= 1;
        S = S2;
    }
    return SplitPlan{S, bm};
}
int64_t splitk_ws_bytes(int64_t M, int Nc, const SplitPlan& sp) {
    const int64_t tm = (M + sp.bm - 1) / sp.bm, tn = (Nc + 127) / 128;
    return kTicketBytes + tm * tn * sp.S * ((int64_t)sp.bm * 128 + (int64_t)sp.bm * kRP) * 4;
}

// Generalised packing for grouped layers: one table row per FACTOR,
//   {src_off, which (0: A [r,len] / 1: B [len,r]), len, r, d16_off, d16_ld, dT_off, rows}
// writes rows j < `rows` of the [16,len] form at packed[d16_off + j·d16_ld + c] and/or columns j < `rows` of the [len,16]
// form at packed[dT_off + c·16 + j] (an offset of -1 skips that form).  rows = 16 zero-fills the unused rank rows; a
// block-diagonal group passes rows = r and pre-offset destinations so that every member fills only its own rank slots of
// a buffer that was zeroed once.
template <typename T>
__global__ __launch_bounds__(256) void pack_items_kernel(const int64_t* table, const float* params, T* packed) {
    const int64_t* e = table + (int64_t)blockIdx.y * 8;
    const float* src = params + e[0];
    const int which = (int)e[1], len = (int)e[2], r = (int)e[3], rows = (int)e[7];
    const int64_t d16 = e[4], ld16 = e[5], dT = e[6];
    for (int idx = blockIdx.x * 256 + threadIdx.x; idx < rows * len; idx += gridDim.x * 256) {
        const int j = idx / len, c = idx - j * len;
        const int jj = j < r ? j : r - 1;
        const float v = which == 0 ? src[(int64_t)jj * len + c] : src[(int64_t)c * r + jj];
        const T t = from_f32<T>(j < r ? v : 0.f);
        if (d16 >= 0) packed[d16 + (int64_t)j * ld16 + c] = t;
        if (dT >= 0) packed[dT + (int64_t)c * kRP + j] = t;
    }
}

template <typename T, int BM, int BN, bool MAIN, int STG, int NW = 4, int WM = 2>
int launch_tile(GemmParams p, hipStream_t stream) {
    p.tiles_m = (int)((p.M + BM - 1) / BM);
    p.tiles_n = MAIN ? (p.Nc + BN - 1) / BN : (p.part_table ? p.tiles_n : 1);  // skinny grouped: tiles_n = parts
    if (p.ldp == 0) p.ldp = p.r;
    static const int order_env = [] { const char* e = getenv("LORA_FORCE_COLMAJOR"); return e ? atoi(e) : -1; }();
    p.col_major = order_env >= 0 ? order_env : (MAIN && (int64_t)p.Nc > p.M ? 1 : 0);
    p.xcd_m = 1;
    if (MAIN && p.splitk <= 1 && p.tile_part == nullptr && p.n_parts == 0) {
        // Which XCD grid re-fetches the fewest operand bytes?  An xm × xn grid (xm·xn = 8) makes the eight L2s fetch
        // xn·|Am| + xm·|Bm| in total; (8,1) and (1,8) are the row- / column-major runs above (any tile counts), the 2-D
        // grids need exact splits.  Only square-ish problems (the 1280-wide layers at 1024 / 256 rows) pick one.
        static const int x2d_env = [] { const char* e = getenv("LORA_XCD2D"); return e ? atoi(e) : 1; }();
        const double am = (double)p.M * p.Kc, bm = (double)p.Nc * p.Kc;
        double best = p.col_major ? 8.0 * am + bm : am + 8.0 * bm;
        for (int xm = 2; xm <= 4 && x2d_env; xm *= 2) {
            const int xn = 8 / xm;
            if (p.tiles_m % xm != 0 || p.tiles_n % xn != 0) continue;
            const double cost = xn * am + xm * bm;
            if (cost < 0.8 * best) {
                best = cost;
                p.xcd_m = xm;
            }
        }
    }
    constexpr int lds = gemm_lds_bytes<BM, BN, T, MAIN, STG, NW, WM>();
    constexpr bool CAN_SPLIT = MAIN && STG > 0 && NW == 4 && BN == 128 && (BM == 64 || BM == 128);  // what plan_splitk hands out
    if constexpr (CAN_SPLIT) {
        if (p.splitk > 1) {  // workspace = [tickets | fp32 tiles | P tiles] (splitk_ws_bytes)
            auto kern = lora_gemm_kernel<T, BM, BN, MAIN, STG, NW, WM, 0, true>;
            static const hipError_t attr = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                                               hipFuncAttributeMaxDynamicSharedMemorySize, lds);
            if (attr != hipSuccess) return LORA_E_LAUNCH;
            p.split_aff = split_affinity() && (8 % p.splitk) == 0 ? 1 : 0;
            p.ws_c = reinterpret_cast<float*>(reinterpret_cast<char*>(p.tickets) + kTicketBytes);
            p.ws_p = p.ws_c + (int64_t)p.tiles_m * p.tiles_n * p.splitk * (BM * BN);
            LORA_LAUNCH(PK_GEMM_SPLITK, kern, dim3(p.tiles_m * p.tiles_n * p.splitk), dim3(NW * 64), lds, stream, p);
            LORA_LAUNCH_CHECK();
            return LORA_OK;
        }
    }
    p.splitk = 0;
    auto kern = lora_gemm_kernel<T, BM, BN, MAIN, STG, NW, WM>;
    if (lds > 48 * 1024) {
        static const hipError_t attr = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                                           hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        if (attr != hipSuccess) return LORA_E_LAUNCH;
    }
    constexpr int prof_id = MAIN ? (BM >= 256 ? PK_GEMM_256x128 : (BM == 128 ? PK_GEMM_128x128 : PK_GEMM_64x64))
                                 : (BM == 128 ? PK_SKINNY_128 : PK_SKINNY_64);
    LORA_LAUNCH(prof_id, kern, dim3(p.tiles_m * p.tiles_n), dim3(NW * 64), lds, stream, p);
    LORA_LAUNCH_CHECK();
    return LORA_OK;
}

// Column-tile width of the big launches.  Every SD width is a multiple of 128 AND of 160, so both tiles are exact; a
// 128×160 tile costs ≈ 1.2× a 128×128 one and the chip holds 512 of either (two workgroups per CU), so the grid with fewer,
// fatter tiles wins wherever the 128-wide grid leaves its last round mostly empty: 1024×1280→2·5120 is 640 tiles of 128 (two
// rounds, the second a quarter full) or 512 tiles of 160 (ONE round).  `gated`: the measured rule of the gated launches
// (tools/gemm_bench.py --geglu --cold-read with LORA_GATE_BN=128|160, µs 128 → 160; forward: 16384 rows 65.1 → 67.1,
// 4096 rows 60.2 → 51.0, 1024 rows 57.0 → 39.1, 256 rows 27.1 → 29.6; backward: 53.1 → 53.6, 41.0 → 35.9, 40.4 → 36.8,
// 30.1 → 34.7): 160 for grids of 257..2047 128-wide tiles.  Ungated launches use the round-count model only.
int gate_tile_width(int64_t tiles_m, int cols, bool gated) {
    static const int env = [] { const char* e = getenv("LORA_GATE_BN"); return e ? atoi(e) : 0; }();
    if (cols % 160 != 0) return 128;
    if (env == 128 || env == 160) return env;
    const int64_t t128 = tiles_m * (cols / 128), t160 = tiles_m * (cols / 160);
    if (gated) return t128 > 256 && t128 < 2048 ? 160 : 128;
    const double c128 = (double)((t128 + 511) / 512), c160 = 1.25 * (double)((t160 + 511) / 512);
    return c160 < c128 ? 160 : 128;
}

// tools/gemm_bench.py / tools/skeleton_per_class.py ablations (GemmParams::dbg): read once, applied by every launcher
static int gemm_dbg_env() {
    static const int v = [] { const char* e = getenv("LORA_GEMM_DBG"); return e ? atoi(e) : 0; }();
    return v;
}

// GEGLU-gated forward: the two-stage ring kernel, a tile = BN/2 h columns + their BN/2 g columns.
template <typename T, int BN>
int launch_gate_bn(GemmParams p, hipStream_t stream) {
    constexpr int BM = 128;
    p.dbg = gemm_dbg_env();
    p.tiles_m = (int)((p.M + BM - 1) / BM);
    p.tiles_n = p.gateF / (BN / 2);
    p.ldp = p.r;
    p.col_major = (int64_t)p.Nc > p.M ? 1 : 0;
    constexpr int lds = gemm_lds_bytes<BM, BN, T, true, 2, 4, 2>();
    auto kern = lora_gemm_kernel<T, BM, BN, true, 2, 4, 2, 1>;
    static const hipError_t attr = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                                       hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    if (attr != hipSuccess) return LORA_E_LAUNCH;
    LORA_LAUNCH(PK_GEMM_128x128, kern, dim3(p.tiles_m * p.tiles_n), dim3(256), lds, stream, p);
    LORA_LAUNCH_CHECK();
    return LORA_OK;
}
template <typename T>
int launch_gate(GemmParams p, hipStream_t stream) {
    // (the h / g halves of a tile are column runs of 64 resp. 80: both divide every SD hidden width)
    if (gate_tile_width((p.M + 127) / 128, 2 * p.gateF, true) == 160 && p.gateF % 80 == 0) return launch_gate_bn<T, 160>(p, stream);
    return launch_gate_bn<T, 128>(p, stream);
}

// GEGLU backward in the epilogue of dout = dZ·W2: 128-row tiles over [M, F].
template <typename T, int BN>
int launch_gate_bwd_bn(GemmParams p, hipStream_t stream) {
    constexpr int BM = 128;
    p.dbg = gemm_dbg_env();
    p.tiles_m = (int)((p.M + BM - 1) / BM);
    p.tiles_n = p.Nc / BN;
    p.ldp = p.r;
    p.col_major = (int64_t)p.Nc > p.M ? 1 : 0;
    constexpr int lds = gemm_lds_bytes<BM, BN, T, true, 2, 4, 2>();
    auto kern = lora_gemm_kernel<T, BM, BN, true, 2, 4, 2, 2>;
    static const hipError_t attr = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                                       hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    if (attr != hipSuccess) return LORA_E_LAUNCH;
    LORA_LAUNCH(PK_GATED_BWD, kern, dim3(p.tiles_m * p.tiles_n), dim3(256), lds, stream, p);
    LORA_LAUNCH_CHECK();
    return LORA_OK;
}
template <typename T>
int launch_gate_bwd(GemmParams p, hipStream_t stream) {
    if (gate_tile_width((p.M + 127) / 128, p.Nc, true) == 160) return launch_gate_bwd_bn<T, 160>(p, stream);
    return launch_gate_bwd_bn<T, 128>(p, stream);
}

// Backward-input of a grouped projection whose contraction consists of KP equal parts (lora_gemm_parts, parts_on_k): the
// part-wise kernels exist on 64-row tiles — 64×160 where the output width wants it (320, 960), else 64×128, else 64×64.
template <typename T, int BM, int BN, int STG, int KP>
int launch_kparts_tile(GemmParams p, hipStream_t stream) {
    p.tiles_m = (int)((p.M + BM - 1) / BM);
    p.tiles_n = (p.Nc + BN - 1) / BN;
    p.col_major = (int64_t)p.Nc > p.M ? 1 : 0;
    p.xcd_m = 1;
    p.splitk = 0;
    constexpr int lds = gemm_lds_bytes<BM, BN, T, true, STG, 4, 2>();
    auto kern = lora_gemm_kernel<T, BM, BN, true, STG, 4, 2, 0, false, KP>;
    static const hipError_t attr = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                                       hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    if (attr != hipSuccess) return LORA_E_LAUNCH;
    LORA_LAUNCH(PK_GEMM_64x64, kern, dim3(p.tiles_m * p.tiles_n), dim3(256), lds, stream, p);
    LORA_LAUNCH_CHECK();
    return LORA_OK;
}
template <typename T, int KP>
int launch_kparts(const GemmParams& p, hipStream_t stream) {
    if constexpr (sizeof(T) == 2) {
        const int64_t tm = (p.M + 63) / 64;
        if ((p.Nc % 160) == 0 && (p.Nc % 128) != 0) return launch_kparts_tile<T, 64, 160, 2, KP>(p, stream);
        if ((p.Nc % 128) == 0) {
            if (tm * (p.Nc / 128) >= 256) return launch_kparts_tile<T, 64, 128, 2, KP>(p, stream);
            return launch_kparts_tile<T, 64, 128, 3, KP>(p, stream);
        }
        return launch_kparts_tile<T, 64, 64, 3, KP>(p, stream);
    } else {
        return LORA_E_UNSUPPORTED;
    }
}

int forced_tile() {  // tuning knob for tools/gemm_bench.py only
    static const int forced = [] {
        const char* e = getenv("LORA_FORCE_TILE");
        return e ? atoi(e) : -1;
    }();
    return forced;
}

// Tile and ring-depth choice, from tools/gemm_bench.py sweeps on MI355X — hot and with cold weights (--cold-read) — and from
// the in-model launch-class table (tools/summarize_profile.py shapes; profiles/README.md):
//  * occupancy beats prefetch depth: a 2-stage ring lets two 128-row workgroups share a CU (≤ 78 KB LDS each) and
//    is 25-35 % faster than the 3-stage ring at one workgroup per CU on every shape that fills the chip;
//  * 128×160 for widths that are multiples of 160 but not of 128 (320, 960): no padding columns;
//  * 128×128 once its grid has >= 256 tiles (and the last column tile is not mostly padding); 64×128 (2 stages) for
//    128..255-tile grids, 64×128 behind a 3-stage ring for 64..127-tile grids;
//  * 64×64 below that: 2 stages (3 workgroups per CU) when there are >= 512 tiles to overlap, else 3, or 4 stages on
//    contractions of >= 8 K-steps (the 6-stage one-workgroup form lost once the weights are cold).  128×64 never won;
//  * 256×128 / 128×256 (one 4- or 8-wave workgroup per CU, with or without a ping-pong DMA order) are correct but 5-100 %
//    slower than two independent 128-row workgroups per CU on every hot-path shape.  Not instantiated (the template
//    still supports them).
template <typename T, bool MAIN>
int launch_pipe(const GemmParams& p_in, hipStream_t stream) {
    GemmParams p = p_in;
    static const int dbg_env = [] { const char* e = getenv("LORA_GEMM_DBG"); return e ? atoi(e) : 0; }();
    p.dbg = dbg_env;
    if (!MAIN) return launch_tile<T, 64, 64, false, 3>(p, stream);
    static const int stg_env = [] { const char* e = getenv("LORA_FORCE_STAGES"); return e ? atoi(e) : 0; }();
    if (p.splitk > 1) {  // (tile height from plan_splitk: the workspace was sized for it)
        if (p.split_bm == 64) return launch_tile<T, 64, 128, true, 3, 4>(p, stream);
        return launch_tile<T, 128, 128, true, 2, 4>(p, stream);
    }
    if (p.tile_part != nullptr) {  // grouped: 64-wide column tiles never straddle two parts
        const int64_t t64 = ((p.M + 63) / 64) * ((p.Nc + 63) / 64);
        return t64 < 512 ? launch_tile<T, 64, 64, true, 3>(p, stream) : launch_tile<T, 64, 64, true, 2>(p, stream);
    }
    const int64_t tiles128 = ((p.M + 127) / 128) * ((p.Nc + 127) / 128);
    const int64_t tiles64 = ((p.M + 63) / 64) * ((p.Nc + 63) / 64);
    const int padded = (p.Nc + 127) / 128 * 128;
    // equal parts over the output columns: a column tile must lie inside one part
    const bool p128 = p.n_parts == 0 || (p.part_n % 128) == 0, p160 = p.n_parts == 0 || (p.part_n % 160) == 0;
    bool big = tiles128 >= 128 && (padded - p.Nc) * 4 <= p.Nc && p128;
    bool deep = tiles64 < 512;
    if (forced_tile() == 0) big = true;
    if (forced_tile() == 2) big = false;
    // (8-wave workgroups — 128×256 as 2×4 waves, 256×128 as 4×2 waves, one per CU, 26 % fewer L2→LDS bytes per flop —
    //  are supported by the template (WM parameter) and were measured: correct, 5–100 % slower on every hot-path shape,
    //  e.g. 16384×320×2560 54 vs 50 µs; not instantiated.)
    if constexpr (sizeof(T) == 2) {
        // widths that are whole multiples of 160 but not of 128 (320, 960: every q/k/v/out projection of the widest SD
        // level): 128×160 tiles waste no MFMA, LDS or L2→LDS traffic on padding columns (128-wide tiles pad 320 to 384)
        // and re-read the X panel twice instead of three times; two workgroups still share a CU (77.8 KB each)
        const int64_t tiles160 = ((p.M + 127) / 128) * (p.Nc / 160);
        bool w160 = (p.Nc % 160) == 0 && ((p.Nc % 128) != 0 || !p128) && tiles160 >= 128 && p160;
        if (forced_tile() == 7) w160 = (p.Nc % 160) == 0;
        if (forced_tile() == 0 || forced_tile() == 2 || forced_tile() == 1) w160 = false;  // 1: the 128|64-square rules only
        // ... and when that grid has fewer than 384 tiles (the 320-wide projections at 16384 rows: 256 tiles, one per CU),
        // 64×160 tiles double the count — two workgroups per CU overlap each other's fixed phases: 16384×320×320
        // 12.1 → 11.4 µs, 16384×1280→320 27.9 → 27.0 (round 3; wider outputs lose: 16384×320→1280 26.4 → 30.6)
        bool half160 = w160 && tiles160 < 384;
        if (forced_tile() == 8) half160 = (p.Nc % 160) == 0;
        if (forced_tile() == 7) half160 = false;
        if (half160) return launch_tile<T, 64, 160, true, 2, 4>(p, stream);
        if (w160) return launch_tile<T, 128, 160, true, 2, 4>(p, stream);
        // 128×128 grids of 128..255 tiles leave a third of the CUs idle (4096×640×640: 160 tiles): 64×128 tiles double the
        // count at 3/4 of the flops per staged byte — 12.5 → 11.0 µs there, 32.8 → 28.2 µs at K = 2560
        bool mid = big && tiles128 < 256;
        if (forced_tile() == 9) mid = true;
        if (forced_tile() == 0 || forced_tile() == 2 || forced_tile() == 1) mid = false;
        if (mid) return launch_tile<T, 64, 128, true, 2, 4>(p, stream);
        // grids too small for 128-row tiles, measured with the weights COLD (tools/gemm_bench.py --cold-read: in the model every
        // frozen weight comes from HBM): 64×128 tiles behind a 3-stage ring (two workgroups per CU, two K-steps in flight)
        // for 64..127-tile grids — 1024×1280×1280: 17.3 → 15.0 µs, the grouped q/k/v backward at 1024 rows 39 → 34 µs
        // (round 3: the same tile behind a 4- or 5-stage ring — one workgroup per CU, as these 160-workgroup grids have anyway —
        //  is SLOWER, 15.4 → 16.0 → 16.3 µs: a lone workgroup is not waiting on prefetch depth)
        // (also measured on this grid, round 3: the same tile as ONE 8-wave workgroup — 4×2 waves, two per SIMD instead of
        //  one — 15.1 vs 15.2 µs behind 3 stages, 19.2 behind 2: neither the wave count nor a ring deeper than 3 moves it)
        if (!big && tiles128 >= 64 && (p.Nc % 128) == 0 && p128 && stg_env == 0 && forced_tile() < 0)
            return launch_tile<T, 64, 128, true, 3, 4>(p, stream);
    }
    if constexpr (sizeof(T) == 2) {
        // chip-filling grids whose width divides by 128 AND 160: the tile whose grid wastes less of its last round
        if (big && p160 && tiles128 >= 256 && forced_tile() < 0 && stg_env == 0 && gate_tile_width((p.M + 127) / 128, p.Nc, false) == 160)
            return launch_tile<T, 128, 160, true, 2, 4>(p, stream);
    }
    if (big) {
        // (a 3-stage ring on grids of <= 256 tiles — one workgroup per CU anyway — was measured: no gain, 12.8 → 13.6 µs on
        //  4096×640×640; a lone workgroup is paced by the CU's vector-memory path issuing its own DMAs, not by latency)
        if (stg_env == 3) return launch_tile<T, 128, 128, true, 3>(p, stream);
        return launch_tile<T, 128, 128, true, 2, 4>(p, stream);
    }
    if (stg_env == 2) deep = false;
    if (stg_env == 3) deep = true;
    // long contractions on grids that leave at most ~1 workgroup per CU: only prefetch depth hides the L2/HBM latency
    // of each K-step there (4 stages = 2 workgroups per CU, 6 stages = 1)
    const int nk = (p.Kc * (int)sizeof(T) + kRowBytes - 1) / kRowBytes;
    int ring = deep ? 3 : 2;
    if (deep && nk >= 8) ring = 4;  // (6 stages at one workgroup per CU lost to 4 stages at two once the weights are cold)
    if (stg_env == 4) ring = stg_env;
    switch (ring) {
        case 4: return launch_tile<T, 64, 64, true, 4>(p, stream);
        case 3: return launch_tile<T, 64, 64, true, 3>(p, stream);
        default: return launch_tile<T, 64, 64, true, 2>(p, stream);
    }
}

struct CallArgs {  // what an entry point knows
    const void* Am;
    const void* Bm;
    const void* bias;
    const void* Fp;      // packed main-loop factor [16,Kc] (T)
    const void* Qp;      // packed epilogue factor [Nc,16] (T)
    const float* F;      // fp32 master of the main-loop factor, strided (generic path)
    int64_t f_sr, f_sk;
    const float* Q;
    int64_t q_sn, q_sj;
    void* C;
    float* P;
    int64_t M;
    int Kc, Nc, r;
    float scale;
    int64_t lda;               // 0 = Kc
    const int* tile_part;      // grouped MAIN launch
    const int64_t* part_table; // grouped skinny launch
    int n_parts;
    bool packed_only;          // no fp32 masters behind Fp/Qp: the shape-agnostic kernels cannot run
    void* workspace;           // optional fp32 scratch for split-K (lora_gemm_workspace_bytes)
    int64_t ws_bytes;
};

template <typename T>
int launch_typed(const CallArgs& c, bool main_part, hipStream_t stream) {
    constexpr int VEC = ElemTraits<T>::kVec;
    constexpr int BK = kRowBytes / (int)sizeof(T);
    const int64_t lda = c.lda > 0 ? c.lda : c.Kc;
    const bool grouped = c.tile_part != nullptr || c.part_table != nullptr;
    const bool fast = c.r <= kRP && c.Fp != nullptr && (c.Qp != nullptr || !main_part) && (c.Kc % VEC) == 0 &&
                      (lda % VEC) == 0 && aligned16(c.Am) && aligned16(c.Fp) && aligned16(c.Qp) &&
                      (!main_part || ((c.Nc % VEC) == 0 && aligned16(c.Bm) && aligned16(c.C)));
    if (fast) {
        GemmParams p{};
        p.Am = c.Am; p.Bm = c.Bm; p.bias = c.bias; p.Fp = c.Fp; p.Qp = c.Qp;
        p.C = c.C; p.P = c.P; p.M = c.M; p.Kc = c.Kc; p.Nc = c.Nc; p.r = c.r; p.scale = c.scale;
        p.lda = lda; p.tile_part = c.tile_part; p.part_table = c.part_table; p.tiles_n = c.n_parts;
        if (main_part && !grouped && c.workspace != nullptr && aligned16(c.workspace) && (c.Kc % BK) == 0) {
            const SplitPlan sp = plan_splitk(c.M, c.Kc, c.Nc, (int)sizeof(T));
            if (sp.S > 1 && c.ws_bytes >= splitk_ws_bytes(c.M, c.Nc, sp)) {
                const int nk = c.Kc / BK;
                p.splitk = sp.S;
                p.split_bm = sp.bm;
                p.steps_per_slice = (nk + sp.S - 1) / sp.S;
                p.tickets = static_cast<unsigned*>(c.workspace);
            }
        }
        if ((c.Kc % BK) == 0) return main_part ? launch_pipe<T, true>(p, stream) : launch_pipe<T, false>(p, stream);
        if (grouped) return LORA_E_UNSUPPORTED;  // grouped launches exist only on the LDS-DMA path
        return main_part ? launch_tile<T, 64, 64, true, 0>(p, stream) : launch_tile<T, 64, 64, false, 0>(p, stream);
    }
    if (grouped || c.packed_only || lda != c.Kc) return LORA_E_UNSUPPORTED;
    GenericParams g{};
    g.Am = c.Am; g.Bm = c.Bm; g.bias = c.bias; g.F = c.F; g.f_sr = c.f_sr; g.f_sk = c.f_sk; g.Q = c.Q;
    g.q_sn = c.q_sn; g.q_sj = c.q_sj; g.C = c.C; g.P = c.P; g.M = c.M; g.Kc = c.Kc; g.Nc = c.Nc; g.r = c.r;
    g.scale = c.scale;
    {
        const int64_t n = g.M * g.r;
        hipLaunchKernelGGL(lora_skinny_generic_kernel<T>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, g);
        LORA_LAUNCH_CHECK();
    }
    if (main_part) {
        const int64_t n = g.M * g.Nc;
        hipLaunchKernelGGL(lora_gemm_generic_kernel<T>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, g);
        LORA_LAUNCH_CHECK();
    }
    return LORA_OK;
}

int launch_gemm(const CallArgs& c, bool main_part, int dtype, hipStream_t stream) {
    switch (dtype) {
        case LORA_F32: return launch_typed<float>(c, main_part, stream);
        case LORA_F16: return launch_typed<half_t>(c, main_part, stream);
        case LORA_BF16: return launch_typed<bf16_t>(c, main_part, stream);
        default: return LORA_E_BADARG;
    }
}

int check_common(int64_t M, int K, int N, int r, int dtype) {
    if (M < 0 || K <= 0 || N <= 0) return LORA_E_BADARG;
    if (dtype != LORA_F32 && dtype != LORA_F16 && dtype != LORA_BF16) return LORA_E_BADARG;
    if (r < 1 || r > (K < N ? K : N)) return LORA_E_RANK;
    return LORA_OK;
}

double esize(int dtype) { return dtype == LORA_F32 ? 4.0 : 2.0; }

}  // namespace

#ifdef LORA_STAMPS
extern "C" int lora_debug_stamps(unsigned long long* host_out, int n_words) {
    return hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_stamps), (size_t)n_words * 8) == hipSuccess ? 0 : -1;
}
extern "C" int lora_debug_stamps_reset() {
    void* d = nullptr;
    if (hipGetSymbolAddress(&d, HIP_SYMBOL(g_stamps)) != hipSuccess) return -1;
    return hipMemset(d, 0, sizeof(g_stamps)) == hipSuccess ? 0 : -1;
}
#endif

extern "C" int lora_pack_factors(const float* A, const float* B, void* Apack, void* Bpack, int K, int N, int r,
                                 int dtype, void* stream) {
    void* A16 = Apack;
    void* Bt16 = Bpack;
    if (!A || !B || !A16 || !Bt16 || K < 1 || N < 1) return LORA_E_BADARG;
    if (r < 1 || r > (K < N ? K : N)) return LORA_E_RANK;
    if (r > kRP) return LORA_OK;  // large ranks run on the generic kernels, which read the fp32 masters
    const int len = K > N ? K : N;
    dim3 grid((unsigned)((kRP * len + 255) / 256), 2);
    hipStream_t s = static_cast<hipStream_t>(stream);
    switch (dtype) {
        case LORA_F32:
            hipLaunchKernelGGL(pack_factor_kernel<float>, grid, dim3(256), 0, s, A, B, static_cast<float*>(A16),
                               static_cast<float*>(Bt16), K, N, r);
            break;
        case LORA_F16:
            hipLaunchKernelGGL(pack_factor_kernel<half_t>, grid, dim3(256), 0, s, A, B, static_cast<half_t*>(A16),
                               static_cast<half_t*>(Bt16), K, N, r);
            break;
        case LORA_BF16:
            hipLaunchKernelGGL(pack_factor_kernel<bf16_t>, grid, dim3(256), 0, s, A, B, static_cast<bf16_t*>(A16),
                               static_cast<bf16_t*>(Bt16), K, N, r);
            break;
        default: return LORA_E_BADARG;
    }
    LORA_LAUNCH_CHECK();
    return LORA_OK;
}

extern "C" int lora_pack_factors_batched(const int64_t* table, int n_layers, int max_len, const float* params,
                                         void* packed, int dtype, void* stream) {
    if (!table || !params || !packed || n_layers < 1 || max_len < 1) return LORA_E_BADARG;
    dim3 grid((unsigned)((kRP * max_len + 255) / 256 > 64 ? 64 : (kRP * max_len + 255) / 256), 2 * n_layers);
    hipStream_t s = static_cast<hipStream_t>(stream);
    switch (dtype) {
        case LORA_F32:
            hipLaunchKernelGGL(pack_factors_batched_kernel<float>, grid, dim3(256), 0, s, table, params,
                               static_cast<float*>(packed));
            break;
        case LORA_F16:
            hipLaunchKernelGGL(pack_factors_batched_kernel<half_t>, grid, dim3(256), 0, s, table, params,
                               static_cast<half_t*>(packed));
            break;
        case LORA_BF16:
            hipLaunchKernelGGL(pack_factors_batched_kernel<bf16_t>, grid, dim3(256), 0, s, table, params,
                               static_cast<bf16_t*>(packed));
            break;
        default: return LORA_E_BADARG;
    }
    LORA_LAUNCH_CHECK();
    return LORA_OK;
}

extern "C" int lora_pack_items(const int64_t* table, int n_items, int max_len, const float* params, void* packed,
                               int dtype, void* stream) {
    if (!table || !params || !packed || n_items < 1 || max_len < 1) return LORA_E_BADARG;
    dim3 grid((unsigned)((kRP * max_len + 255) / 256 > 64 ? 64 : (kRP * max_len + 255) / 256), n_items);
    hipStream_t s = static_cast<hipStream_t>(stream);
    switch (dtype) {
        case LORA_F32:
            hipLaunchKernelGGL(pack_items_kernel<float>, grid, dim3(256), 0, s, table, params, static_cast<float*>(packed));
            break;
        case LORA_F16:
            hipLaunchKernelGGL(pack_items_kernel<half_t>, grid, dim3(256), 0, s, table, params,
                               static_cast<half_t*>(packed));
            break;
        case LORA_BF16:
            hipLaunchKernelGGL(pack_items_kernel<bf16_t>, grid, dim3(256), 0, s, table, params,
                               static_cast<bf16_t*>(packed));
            break;
        default: return LORA_E_BADARG;
    }
    LORA_LAUNCH_CHECK();
    return LORA_OK;
}

extern "C" int lora_linear_fwd(const void* X, const void* W, const void* bias, const float* A, const float* B,
                               const void* Apack, const void* Bpack, void* Y, float* T_out, int64_t M, int K, int N,
                               int r, float scale, int dtype, void* stream) {
    return lora_linear_fwd_ws(X, W, bias, A, B, Apack, Bpack, Y, T_out, M, K, N, r, scale, dtype, nullptr, 0, stream);
}

extern "C" int lora_linear_fwd_ws(const void* X, const void* W, const void* bias, const float* A, const float* B,
                                  const void* Apack, const void* Bpack, void* Y, float* T_out, int64_t M, int K, int N,
                                  int r, float scale, int dtype, void* workspace, int64_t ws_bytes, void* stream) {
    const int st = check_common(M, K, N, r, dtype);
    if (st != LORA_OK) return st;
    if (M == 0) return LORA_OK;  // empty batch: nothing to do (pointers may be null)
    if (!X || !W || !A || !B || !Y || !T_out) return LORA_E_BADARG;
    CallArgs c{};
    c.Am = X; c.Bm = W; c.bias = bias;
    const size_t es = dtype == LORA_F32 ? 4 : 2;
    c.Fp = Apack;                                                              // A16  [16,K]
    c.Qp = Bpack ? static_cast<const char*>(Bpack) + (size_t)kRP * N * es : nullptr;  // B16  [N,16]
    c.F = A; c.f_sr = K; c.f_sk = 1;               // F[j,k] = A[j,k]
    c.Q = B; c.q_sn = r; c.q_sj = 1;               // Q[n,j] = B[n,j]
    c.C = Y; c.P = T_out;
    c.M = M; c.Kc = K; c.Nc = N; c.r = r; c.scale = scale;
    c.workspace = workspace; c.ws_bytes = ws_bytes;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const double e = esize(dtype);
    ProfWork work(e * ((double)M * K + (double)N * K + (double)M * N) + e * r * (K + N) + (bias ? e * N : 0.0),
                  2.0 * M * K * N + 2.0 * M * r * (double)(K + N));
    return launch_gemm(c, true, dtype, s);
}

extern "C" int lora_linear_geglu_fwd(const void* X, const void* W, const void* bias, const void* Apack, const void* Bpack,
                                     void* Y, void* Out, float* T_out, int64_t M, int K, int N, int r, float scale,
                                     int dtype, void* stream) {
    const int st = check_common(M, K, N, r, dtype);
    if (st != LORA_OK) return st;
    if ((N & 1) != 0) return LORA_E_BADARG;
    if (M == 0) return LORA_OK;
    if (!X || !W || !Out || !T_out) return LORA_E_BADARG;
    // the fused epilogue exists on the LDS-DMA ring kernel for 16-bit types only; everything else: LORA_E_UNSUPPORTED, and
    // the caller runs lora_linear_fwd + geglu_gate_fwd
    const int F = N / 2;
    if (dtype == LORA_F32 || r > kRP || !Apack || !Bpack || (K % 64) != 0 || (F % 64) != 0) return LORA_E_UNSUPPORTED;
    if (!aligned16(X) || !aligned16(W) || !aligned16(Apack) || !aligned16(Bpack) || !aligned16(Out) || (Y && !aligned16(Y)) ||
        (bias && (reinterpret_cast<uintptr_t>(bias) & 7u)))
        return LORA_E_UNSUPPORTED;
    GemmParams p{};
    p.Am = X; p.Bm = W; p.bias = bias;
    p.Fp = Apack;                                                   // A16 [16,K]
    p.Qp = static_cast<const char*>(Bpack) + (size_t)kRP * N * 2;   // B16 [N,16]
    p.C = Y; p.C2 = Out; p.gateF = F; p.P = T_out;
    p.M = M; p.Kc = K; p.Nc = N; p.r = r; p.scale = scale; p.lda = K;
    const double e = 2.0;
    ProfWork work(e * ((double)M * K + (double)N * K + (Y ? (double)M * N : 0.0) + (double)M * F) + e * r * (K + N) +
                      (bias ? e * N : 0.0),
                  2.0 * M * K * N + 2.0 * M * r * (double)(K + N));
    hipStream_t s = static_cast<hipStream_t>(stream);
    return dtype == LORA_F16 ? launch_gate<half_t>(p, s) : launch_gate<bf16_t>(p, s);
}

extern "C" int geglu_linear_bwd(const void* dZ, const void* W2t, const void* Y, void* dY, const void* zeros, int64_t M,
                                int Nz, int F, int dtype, void* stream) {
    if (M < 0 || Nz <= 0 || F <= 0) return LORA_E_BADARG;
    if (dtype != LORA_F32 && dtype != LORA_F16 && dtype != LORA_BF16) return LORA_E_BADARG;
    if (M == 0) return LORA_OK;
    if (!dZ || !W2t || !Y || !dY || !zeros) return LORA_E_BADARG;
    if (dtype == LORA_F32 || (Nz % 64) != 0 || (F % 128) != 0) return LORA_E_UNSUPPORTED;
    if (!aligned16(dZ) || !aligned16(W2t) || !aligned16(Y) || !aligned16(dY) || !aligned16(zeros)) return LORA_E_UNSUPPORTED;
    GemmParams p{};
    p.Am = dZ; p.Bm = W2t; p.bias = nullptr;
    p.Fp = zeros;  // no rank-r term in this launch: a zero factor tile [16, Nz] and a zero epilogue factor [F, 16]
    p.Qp = zeros;
    p.C = dY; p.C2 = const_cast<void*>(Y); p.gateF = F; p.P = nullptr;
    p.M = M; p.Kc = Nz; p.Nc = F; p.r = 1; p.scale = 0.f; p.lda = Nz;
    const double e = 2.0;
    ProfWork work(e * ((double)M * Nz + (double)F * Nz + 4.0 * (double)M * F), 2.0 * M * (double)Nz * F);
    hipStream_t s = static_cast<hipStream_t>(stream);
    return dtype == LORA_F16 ? launch_gate_bwd<half_t>(p, s) : launch_gate_bwd<bf16_t>(p, s);
}

extern "C" int64_t lora_gemm_workspace_bytes(int64_t M, int Kc, int Nc, int dtype) {
    if (M < 1 || Kc < 1 || Nc < 1) return 0;
    const SplitPlan sp = plan_splitk(M, Kc, Nc, dtype == LORA_F32 ? 4 : 2);
    return sp.S > 1 ? splitk_ws_bytes(M, Nc, sp) : 0;
}

extern "C" int lora_linear_bwd_input(const void* dY, const void* Wt, const float* A, const float* B,
                                     const void* Apack, const void* Bpack, void* dX, float* U_out, int64_t M, int K,
                                     int N, int r, float scale, int dtype, void* stream) {
    return lora_linear_bwd_input_ws(dY, Wt, A, B, Apack, Bpack, dX, U_out, M, K, N, r, scale, dtype, nullptr, 0, stream);
}

extern "C" int lora_linear_bwd_input_ws(const void* dY, const void* Wt, const float* A, const float* B,
                                        const void* Apack, const void* Bpack, void* dX, float* U_out, int64_t M, int K,
                                        int N, int r, float scale, int dtype, void* workspace, int64_t ws_bytes,
                                        void* stream) {
    const int st = check_common(M, K, N, r, dtype);
    if (st != LORA_OK) return st;
    if (M == 0) return LORA_OK;
    if (!dY || !A || !B || !U_out) return LORA_E_BADARG;
    if (dX && !Wt) return LORA_E_BADARG;
    CallArgs c{};
    c.Am = dY; c.Bm = Wt; c.bias = nullptr;
    const size_t es = dtype == LORA_F32 ? 4 : 2;
    c.Fp = Bpack;                                                              // Bt16 [16,N]
    c.Qp = Apack ? static_cast<const char*>(Apack) + (size_t)kRP * K * es : nullptr;  // At16 [K,16]
    c.F = B; c.f_sr = 1; c.f_sk = r;               // F[j,n] = B[n,j]
    c.Q = A; c.q_sn = 1; c.q_sj = K;               // Q[k,j] = A[j,k]
    c.C = dX; c.P = U_out;
    c.M = M; c.Kc = N; c.Nc = K; c.r = r; c.scale = scale;
    c.workspace = workspace; c.ws_bytes = ws_bytes;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const double e = esize(dtype);
    const double bytes = dX ? e * ((double)M * N + (double)N * K + (double)M * K) + e * r * (K + N)
                            : e * (double)M * N + e * r * N;
    const double flops = dX ? 2.0 * M * K * N + 2.0 * M * r * (double)(K + N) : 2.0 * M * r * (double)N;
    ProfWork work(bytes, flops);
    return launch_gemm(c, dX != nullptr, dtype, s);
}

extern "C" int lora_gemm_packed(const void* Am, int64_t lda, const void* Bm, const void* bias, const void* Fp,
                                const void* Qp, const int* tile_part, const int64_t* part_table, int n_parts, void* C,
                                float* P_out, int64_t M, int Kc, int Nc, int r, float scale, int64_t work_cols,
                                void* workspace, int64_t ws_bytes, int dtype, void* stream) {
    if (M < 0 || Kc <= 0 || r < 1 || r > kRP) return LORA_E_BADARG;
    if (dtype != LORA_F32 && dtype != LORA_F16 && dtype != LORA_BF16) return LORA_E_BADARG;
    if (M == 0) return LORA_OK;
    const bool main_part = C != nullptr;
    if (!Am || !Fp || (main_part && (!Bm || !Qp || Nc <= 0))) return LORA_E_BADARG;
    if (tile_part && !main_part) return LORA_E_BADARG;
    if (part_table && (main_part || n_parts < 1)) return LORA_E_BADARG;
    CallArgs c{};
    c.Am = Am; c.Bm = Bm; c.bias = bias; c.Fp = Fp; c.Qp = Qp; c.C = C; c.P = P_out;
    c.M = M; c.Kc = Kc; c.Nc = main_part ? Nc : 1; c.r = r; c.scale = scale; c.lda = lda;
    c.tile_part = tile_part; c.part_table = part_table; c.n_parts = n_parts; c.packed_only = true;
    c.workspace = workspace; c.ws_bytes = ws_bytes;
    const double e = esize(dtype);
    const double kc = work_cols > 0 ? (double)work_cols : (double)Kc;
    ProfWork work(main_part ? e * ((double)M * Kc + (double)Nc * Kc + (double)M * Nc) + e * r * (double)(Kc + Nc) +
                                  (bias ? e * Nc : 0.0)
                            : e * (double)M * kc + e * r * kc,
                  main_part ? 2.0 * M * (double)Kc * Nc + 2.0 * M * r * (double)(Kc + Nc) : 2.0 * M * r * kc);
    return launch_gemm(c, main_part, dtype, static_cast<hipStream_t>(stream));
}

extern "C" int lora_gemm_parts(const void* Am, const void* Bm, const void* bias, const void* Fp, const void* Qp, void* C,
                               float* P_out, int64_t ldp, int64_t M, int Kc, int Nc, int r, int n_parts, int parts_on_k,
                               float scale, int dtype, void* stream) {
    if (M < 0 || Kc <= 0 || Nc <= 0 || r < 1 || r > kRP || n_parts < 2 || n_parts > 4) return LORA_E_BADARG;
    if (dtype != LORA_F32 && dtype != LORA_F16 && dtype != LORA_BF16) return LORA_E_BADARG;
    if (M == 0) return LORA_OK;
    if (!Am || !Bm || !Fp || !Qp || !C || !P_out || ldp < (int64_t)n_parts * r) return LORA_E_BADARG;
    const int split = parts_on_k ? Kc : Nc;  // the axis that is cut into parts
    if (split % n_parts != 0) return LORA_E_BADARG;
    const int part_n = split / n_parts;
    if (dtype == LORA_F32 || (part_n % 64) != 0 || (Kc % 64) != 0 || (Nc % 8) != 0) return LORA_E_UNSUPPORTED;
    if (!aligned16(Am) || !aligned16(Bm) || !aligned16(Fp) || !aligned16(Qp) || !aligned16(C) ||
        (bias && (reinterpret_cast<uintptr_t>(bias) & 7u)))
        return LORA_E_UNSUPPORTED;
    GemmParams p{};
    p.Am = Am; p.Bm = Bm; p.bias = bias; p.Fp = Fp; p.Qp = Qp; p.C = C; p.P = P_out;
    p.M = M; p.Kc = Kc; p.Nc = Nc; p.r = r; p.scale = scale; p.lda = Kc; p.ldp = (int)ldp;
    p.n_parts = n_parts; p.part_n = part_n; p.parts_on_k = parts_on_k ? 1 : 0;
    const double e = 2.0;
    ProfWork work(e * ((double)M * Kc + (double)Nc * Kc + (double)M * Nc) + e * r * (double)(Kc + Nc) * 1.0 + (bias ? e * Nc : 0.0),
                  2.0 * M * (double)Kc * Nc + 2.0 * M * r * (double)(Kc + Nc));
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (!parts_on_k) return dtype == LORA_F16 ? launch_pipe<half_t, true>(p, s) : launch_pipe<bf16_t, true>(p, s);
    if (n_parts != 3) return LORA_E_UNSUPPORTED;  // (the part-wise backward is instantiated for q / k / v groups)
    return dtype == LORA_F16 ? launch_kparts<half_t, 3>(p, s) : launch_kparts<bf16_t, 3>(p, s);
}

// LoRA-fused GEMM for gfx950.
//
// One kernel template serves both big contractions of the hot path:
//   forward   (lora_diffusion/lora.py:49-50):   Y  = X·Wᵀ + b + s·(X·Aᵀ)·Bᵀ     and  T = X·Aᵀ
//   backward  (autograd of the same line):      dX = dY·W   + s·(dY·B)·A        and  U = dY·B
// written once as
//   C[M,Nc] = Am[M,Kc]·Bm[Nc,Kc]ᵀ + bias + s·P·Qᵀ ,   P[M,r] = Am·Fᵀ  (stored unscaled)
// with (Am,Bm,F,Q) = (X, W, A, B) forward and (dY, Wᵀ, Bᵀ, Aᵀ) backward.  Both big operands are
// contraction-contiguous ("NT" GEMM), which is why the frozen weight is cached in both
// orientations.
//
// Structure per 256-thread workgroup (4 waves as 2×2), BM×BN output tile, 128-byte K-steps:
//   - register-staged global→LDS copies (16 B per lane, XOR-swizzled 16-B chunks), the loads of
//     K-step t+1 in flight while step t is multiplied;
//   - base contraction on MFMA 16x16x32 (f16/bf16) or 16x16x4 (f32, exact);
//   - the rank-r factor F rides along as a 16-row LDS tile: the X fragments already in VGPRs are
//     multiplied with it (one extra MFMA per row fragment, K-steps split between the two
//     column waves), so T costs no extra HBM or LDS traffic for X;
//   - epilogue: the two partial P tiles are summed through LDS, P is written once, and
//     s·P·Qᵀ is added to the accumulators as ONE extra MFMA K-step: for 16-bit types the 32-wide
//     step carries [hi(sP) | lo(sP)] × [Q | Q], keeping P at ~fp32 precision;
//   - bias is added in fp32, the tile is transposed through LDS and stored as whole 16-B row chunks.
// Workgroups are dealt to XCDs so that the column tiles of one row panel share an L2.
#include "common.h"

namespace {

struct GemmParams {
    const void* Am;
    const void* Bm;
    const void* bias;
    const float* F;
    int64_t f_sr, f_sk;  // F[j,k]  at F[j*f_sr + k*f_sk]
    const float* Q;
    int64_t q_sn, q_sj;  // Q[n,j]  at Q[n*q_sn + j*q_sj]
    void* C;
    float* P;
    int64_t M;
    int Kc, Nc, r;
    float scale;
    int tiles_m, tiles_n;
};

constexpr int kRowBytes = 128;  // one K-step of one tile row
constexpr int kRP = 16;         // rank padded to one MFMA fragment

__device__ __forceinline__ int lds_off(int row, int chunk) {
    return row * kRowBytes + ((chunk ^ (row & 7)) << 4);
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

template <int BM, int BN, typename T> constexpr int gemm_lds_bytes(bool main_part) {
    constexpr int tiles = (BM + BN) * kRowBytes;
    constexpr int sf = kRP * kRowBytes;
    constexpr int sq = BN * kRP * (int)sizeof(T);
    constexpr int ep = sizeof(T) == 4 ? 2 : 1;
    constexpr int sc = (BM / ep) * (BN * (int)sizeof(T) + 16);
    const int a = main_part ? tiles + sf + sq : BM * kRowBytes + sf;
    const int b = main_part ? sc : 0;
    return a > b ? a : b;
}

template <typename T, int BM, int BN, bool MAIN>
__global__ __launch_bounds__(256) void lora_gemm_kernel(GemmParams p) {
    constexpr int VEC = ElemTraits<T>::kVec;
    constexpr int BK = kRowBytes / (int)sizeof(T);
    constexpr int MI = BM / 32;  // 16-row fragments per wave
    constexpr int NI = BN / 32;
    constexpr int PA = BM / 32;  // staging passes (32 rows per pass)
    constexpr int PB = BN / 32;
    constexpr bool F32 = sizeof(T) == 4;

    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* sA = smem;
    char* sB = smem + BM * kRowBytes;
    char* sF = MAIN ? sB + BN * kRowBytes : sB;
    char* sQ = sF + kRP * kRowBytes;
    float* sP = reinterpret_cast<float*>(smem);  // epilogue overlay on sA: [2][BM][16]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1;
    const int wn = wave & 1;
    const int l15 = lane & 15;
    const int lq = lane >> 4;

    // XCD-aware tile assignment: blocks b and b+8 share an XCD (round-robin dispatch), so give
    // every XCD a contiguous run of tiles; column tiles of one row panel are consecutive.
    int tile;
    {
        const int total = p.tiles_m * p.tiles_n;
        const int id = blockIdx.x;
        const int q = total >> 3, rem = total & 7;
        const int xcd = id & 7, slot = id >> 3;
        tile = (xcd < rem ? xcd * (q + 1) : rem * (q + 1) + (xcd - rem) * q) + slot;
    }
    const int tm = tile / p.tiles_n;
    const int tn = tile - tm * p.tiles_n;
    const int64_t m0 = (int64_t)tm * BM;
    const int n0 = tn * BN;

    const T* Ag = static_cast<const T*>(p.Am);
    const T* Bg = static_cast<const T*>(p.Bm);

    // ---- staging state -------------------------------------------------------------------
    const int ld_chunk = tid & 7;
    const int ld_row = tid >> 3;
    const T* a_ptr[PA];
    const T* b_ptr[MAIN ? PB : 1];
#pragma unroll
    for (int i = 0; i < PA; ++i) {
        int64_t m = m0 + ld_row + 32 * i;
        if (m > p.M - 1) m = p.M - 1;
        a_ptr[i] = Ag + m * p.Kc + ld_chunk * VEC;
    }
    if constexpr (MAIN) {
#pragma unroll
        for (int i = 0; i < PB; ++i) {
            int n = n0 + ld_row + 32 * i;
            if (n > p.Nc - 1) n = p.Nc - 1;
            b_ptr[i] = Bg + (int64_t)n * p.Kc + ld_chunk * VEC;
        }
    }
    const int f_row = tid >> 3;  // valid for tid < 128
    const bool f_thread = tid < kRP * 8;

    Chunk<T> ra[PA];
    Chunk<T> rb[MAIN ? PB : 1];
    Chunk<T> rf;

    auto load_step = [&](int k0) {
        const bool kin = (k0 + ld_chunk * VEC) < p.Kc;
#pragma unroll
        for (int i = 0; i < PA; ++i) {
            if (kin) {
                ra[i] = *reinterpret_cast<const Chunk<T>*>(a_ptr[i] + k0);
            } else {
#pragma unroll
                for (int e = 0; e < VEC; ++e) ra[i].v[e] = from_f32<T>(0.f);
            }
        }
        if constexpr (MAIN) {
#pragma unroll
            for (int i = 0; i < PB; ++i) {
                if (kin) {
                    rb[i] = *reinterpret_cast<const Chunk<T>*>(b_ptr[i] + k0);
                } else {
#pragma unroll
                    for (int e = 0; e < VEC; ++e) rb[i].v[e] = from_f32<T>(0.f);
                }
            }
        }
        if (f_thread) {
#pragma unroll
            for (int e = 0; e < VEC; ++e) {
                const int k = k0 + ld_chunk * VEC + e;
                float v = 0.f;
                if (f_row < p.r && k < p.Kc) v = p.F[f_row * p.f_sr + k * p.f_sk];
                rf.v[e] = from_f32<T>(v);
            }
        }
    };
    auto store_step = [&]() {
#pragma unroll
        for (int i = 0; i < PA; ++i)
            *reinterpret_cast<Chunk<T>*>(sA + lds_off(ld_row + 32 * i, ld_chunk)) = ra[i];
        if constexpr (MAIN) {
#pragma unroll
            for (int i = 0; i < PB; ++i)
                *reinterpret_cast<Chunk<T>*>(sB + lds_off(ld_row + 32 * i, ld_chunk)) = rb[i];
        }
        if (f_thread) *reinterpret_cast<Chunk<T>*>(sF + lds_off(f_row, ld_chunk)) = rf;
    };

    // ---- Q tile (epilogue factor), staged once -------------------------------------------
    if constexpr (MAIN) {
        T* q = reinterpret_cast<T*>(sQ);
        for (int idx = tid; idx < BN * kRP; idx += 256) {
            const int n = idx >> 4, j = idx & 15;
            float v = 0.f;
            if (j < p.r && n0 + n < p.Nc) v = p.Q[(int64_t)(n0 + n) * p.q_sn + j * p.q_sj];
            q[idx] = from_f32<T>(v);
        }
    }

    f32x4 acc[MAIN ? MI : 1][MAIN ? NI : 1];
    f32x4 pacc[MI];
#pragma unroll
    for (int i = 0; i < (MAIN ? MI : 1); ++i)
#pragma unroll
        for (int j = 0; j < (MAIN ? NI : 1); ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < MI; ++i) pacc[i] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int nk = (p.Kc + BK - 1) / BK;
    load_step(0);
    for (int kt = 0; kt < nk; ++kt) {
        if (kt > 0) __syncthreads();
        store_step();
        __syncthreads();
        if (kt + 1 < nk) load_step((kt + 1) * BK);

        if constexpr (!F32) {
            using Frag = typename Mfma<T>::Frag;
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                const int chunk = ks * 4 + lq;
                Frag af[MI];
#pragma unroll
                for (int mi = 0; mi < MI; ++mi)
                    af[mi] = *reinterpret_cast<const Frag*>(
                        sA + lds_off(wm * (BM / 2) + mi * 16 + l15, chunk));
                if (ks == wn) {
                    const Frag ff = *reinterpret_cast<const Frag*>(sF + lds_off(l15, chunk));
#pragma unroll
                    for (int mi = 0; mi < MI; ++mi) pacc[mi] = Mfma<T>::run(af[mi], ff, pacc[mi]);
                }
                if constexpr (MAIN) {
                    Frag bf[NI];
#pragma unroll
                    for (int ni = 0; ni < NI; ++ni)
                        bf[ni] = *reinterpret_cast<const Frag*>(
                            sB + lds_off(wn * (BN / 2) + ni * 16 + l15, chunk));
#pragma unroll
                    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                        for (int ni = 0; ni < NI; ++ni)
                            acc[mi][ni] = Mfma<T>::run(af[mi], bf[ni], acc[mi][ni]);
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
                    af[mi] = *reinterpret_cast<const f32x4*>(
                        sA + lds_off(wm * (BM / 2) + mi * 16 + l15, chunk));
                if (h == wn) {
                    const f32x4 ff = *reinterpret_cast<const f32x4*>(sF + lds_off(l15, chunk));
#pragma unroll
                    for (int e = 0; e < 4; ++e)
#pragma unroll
                        for (int mi = 0; mi < MI; ++mi)
                            pacc[mi] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[mi][e], ff[e],
                                                                            pacc[mi], 0, 0, 0);
                }
                if constexpr (MAIN) {
                    f32x4 bf[NI];
#pragma unroll
                    for (int ni = 0; ni < NI; ++ni)
                        bf[ni] = *reinterpret_cast<const f32x4*>(
                            sB + lds_off(wn * (BN / 2) + ni * 16 + l15, chunk));
#pragma unroll
                    for (int e = 0; e < 4; ++e)
#pragma unroll
                        for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                            for (int ni = 0; ni < NI; ++ni)
                                acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x4f32(
                                    af[mi][e], bf[ni][e], acc[mi][ni], 0, 0, 0);
                }
            }
        }
    }

    // ---- epilogue 1: combine the two partial P tiles through LDS --------------------------
    __syncthreads();
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {
            const int row = wm * (BM / 2) + mi * 16 + lq * 4 + reg;
            sP[(wn * BM + row) * kRP + l15] = pacc[mi][reg];
        }
    __syncthreads();

    if (p.P != nullptr && tn == 0) {
        const int row = tid >> 1, half = tid & 1;
        if (row < BM && m0 + row < p.M) {
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int j = half * 8 + e;
                if (j < p.r)
                    p.P[(m0 + row) * p.r + j] = sP[row * kRP + j] + sP[(BM + row) * kRP + j];
            }
        }
    }

    if constexpr (MAIN) {
        // ---- epilogue 2: acc += s·P·Qᵀ as one extra MFMA K-step ---------------------------
        if constexpr (!F32) {
            using Frag = typename Mfma<T>::Frag;
            const int j0 = 8 * (lq & 1);
            Frag qf[NI];
#pragma unroll
            for (int ni = 0; ni < NI; ++ni)
                qf[ni] = *reinterpret_cast<const Frag*>(
                    sQ + ((wn * (BN / 2) + ni * 16 + l15) * kRP + j0) * (int)sizeof(T));
#pragma unroll
            for (int mi = 0; mi < MI; ++mi) {
                const int row = wm * (BM / 2) + mi * 16 + l15;
                Frag pf;
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float v =
                        (sP[row * kRP + j0 + e] + sP[(BM + row) * kRP + j0 + e]) * p.scale;
                    const T hi = from_f32<T>(v);
                    const T lo = from_f32<T>(v - to_f32<T>(hi));
                    pf[e] = lq < 2 ? hi : lo;
                }
#pragma unroll
                for (int ni = 0; ni < NI; ++ni) acc[mi][ni] = Mfma<T>::run(pf, qf[ni], acc[mi][ni]);
            }
        } else {
            const float* q = reinterpret_cast<const float*>(sQ);
#pragma unroll
            for (int st = 0; st < 4; ++st) {
                const int j = 4 * st + lq;
                float qv[NI];
#pragma unroll
                for (int ni = 0; ni < NI; ++ni) qv[ni] = q[(wn * (BN / 2) + ni * 16 + l15) * kRP + j];
#pragma unroll
                for (int mi = 0; mi < MI; ++mi) {
                    const int row = wm * (BM / 2) + mi * 16 + l15;
                    const float pv = (sP[row * kRP + j] + sP[(BM + row) * kRP + j]) * p.scale;
#pragma unroll
                    for (int ni = 0; ni < NI; ++ni)
                        acc[mi][ni] =
                            __builtin_amdgcn_mfma_f32_16x16x4f32(pv, qv[ni], acc[mi][ni], 0, 0, 0);
                }
            }
        }

        // ---- epilogue 3: bias (fp32), transpose through LDS, 16-B row stores -------------
        if (p.bias != nullptr) {
            const T* bias = static_cast<const T*>(p.bias);
#pragma unroll
            for (int ni = 0; ni < NI; ++ni) {
                int col = n0 + wn * (BN / 2) + ni * 16 + l15;
                if (col > p.Nc - 1) col = p.Nc - 1;
                const float bv = to_f32<T>(bias[col]);
#pragma unroll
                for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                    for (int reg = 0; reg < 4; ++reg) acc[mi][ni][reg] += bv;
            }
        }
        constexpr int EP = F32 ? 2 : 1;
        constexpr int ROWS = BM / EP;
        constexpr int SC_STRIDE = BN * (int)sizeof(T) + 16;
        constexpr int CPR = BN / VEC;  // 16-B chunks per tile row
        T* Cg = static_cast<T*>(p.C);
#pragma unroll
        for (int ep = 0; ep < EP; ++ep) {
            __syncthreads();  // sP / sQ (or the previous pass) are dead
            if (EP == 1 || wm == ep) {
#pragma unroll
                for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                    for (int ni = 0; ni < NI; ++ni)
#pragma unroll
                        for (int reg = 0; reg < 4; ++reg) {
                            const int row = (EP == 1 ? wm * (BM / 2) : 0) + mi * 16 + lq * 4 + reg;
                            const int col = wn * (BN / 2) + ni * 16 + l15;
                            *reinterpret_cast<T*>(smem + row * SC_STRIDE + col * (int)sizeof(T)) =
                                from_f32<T>(acc[mi][ni][reg]);
                        }
            }
            __syncthreads();
            for (int idx = tid; idx < ROWS * CPR; idx += 256) {
                const int row = idx / CPR, ch = idx - row * CPR;
                const int64_t m = m0 + ep * ROWS + row;
                const int col = n0 + ch * VEC;
                if (m < p.M && col < p.Nc)
                    *reinterpret_cast<Chunk<T>*>(Cg + m * p.Nc + col) =
                        *reinterpret_cast<const Chunk<T>*>(smem + row * SC_STRIDE + ch * 16);
            }
        }
    }
}

// Shape-agnostic path (unaligned sizes or r > 16): correct, not fast.
template <typename T>
__global__ void lora_skinny_generic_kernel(GemmParams p) {
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
__global__ void lora_gemm_generic_kernel(GemmParams p) {
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

template <typename T, int BM, int BN, bool MAIN>
int launch_tile(GemmParams p, hipStream_t stream) {
    p.tiles_m = (int)((p.M + BM - 1) / BM);
    p.tiles_n = MAIN ? (p.Nc + BN - 1) / BN : 1;
    const int lds = gemm_lds_bytes<BM, BN, T>(MAIN);
    auto kern = lora_gemm_kernel<T, BM, BN, MAIN>;
    if (lds > 48 * 1024) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess)
            return LORA_E_LAUNCH;
    }
    constexpr int prof_id = MAIN ? (BM == 128 && BN == 128 ? PK_GEMM_128x128 : (BM == 128 ? PK_GEMM_128x64 : PK_GEMM_64x64))
                                 : (BM == 128 ? PK_SKINNY_128 : PK_SKINNY_64);
    LORA_LAUNCH(prof_id, kern, dim3(p.tiles_m * p.tiles_n), dim3(256), lds, stream, p);
    LORA_LAUNCH_CHECK();
    return LORA_OK;
}

// Tile choice: fill 256 CUs × ~3 resident workgroups; prefer the tile with least padding waste.
template <typename T, bool MAIN>
int launch_fast(const GemmParams& p, hipStream_t stream) {
    if (!MAIN) {
        return p.M >= 128 * 192 ? launch_tile<T, 128, 64, false>(p, stream)
                                : launch_tile<T, 64, 64, false>(p, stream);
    }
    struct Cand {
        int bm, bn;
    };
    const Cand cands[3] = {{128, 128}, {128, 64}, {64, 64}};
    double best = 1e30;
    int pick = 2;
    for (int i = 0; i < 3; ++i) {
        const double tm = (double)((p.M + cands[i].bm - 1) / cands[i].bm);
        const double tn = (double)((p.Nc + cands[i].bn - 1) / cands[i].bn);
        const double tiles = tm * tn;
        const double slots = 256.0 * (cands[i].bm * cands[i].bn >= 128 * 128 ? 2.0 : 4.0);
        const double rounds = tiles <= slots ? 1.0 : tiles / slots;
        // per-tile time model: MFMA work + operand traffic (smaller tiles re-read more)
        const double work = (double)cands[i].bm * cands[i].bn + 24.0 * (cands[i].bm + cands[i].bn) + 3000.0;
        const double cost = rounds * work;
        if (cost < best) {
            best = cost;
            pick = i;
        }
    }
    switch (pick) {
        case 0: return launch_tile<T, 128, 128, true>(p, stream);
        case 1: return launch_tile<T, 128, 64, true>(p, stream);
        default: return launch_tile<T, 64, 64, true>(p, stream);
    }
}

template <typename T>
int launch_typed(const GemmParams& p, bool main_part, hipStream_t stream) {
    constexpr int VEC = ElemTraits<T>::kVec;
    const bool fast = p.r <= kRP && (p.Kc % VEC) == 0 && aligned16(p.Am) &&
                      (!main_part || ((p.Nc % VEC) == 0 && aligned16(p.Bm) && aligned16(p.C)));
    if (fast) return main_part ? launch_fast<T, true>(p, stream) : launch_fast<T, false>(p, stream);
    {
        const int64_t n = p.M * p.r;
        hipLaunchKernelGGL(lora_skinny_generic_kernel<T>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0,
                           stream, p);
        LORA_LAUNCH_CHECK();
    }
    if (main_part) {
        const int64_t n = p.M * p.Nc;
        hipLaunchKernelGGL(lora_gemm_generic_kernel<T>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0,
                           stream, p);
        LORA_LAUNCH_CHECK();
    }
    return LORA_OK;
}

int launch_gemm(const GemmParams& p, bool main_part, int dtype, hipStream_t stream) {
    switch (dtype) {
        case LORA_F32: return launch_typed<float>(p, main_part, stream);
        case LORA_F16: return launch_typed<half_t>(p, main_part, stream);
        case LORA_BF16: return launch_typed<bf16_t>(p, main_part, stream);
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

extern "C" int lora_linear_fwd(const void* X, const void* W, const void* bias, const float* A,
                               const float* B, void* Y, float* T_out, int64_t M, int K, int N, int r,
                               float scale, int dtype, void* stream) {
    const int st = check_common(M, K, N, r, dtype);
    if (st != LORA_OK) return st;
    if (M == 0) return LORA_OK;  // empty batch: nothing to do (pointers may be null)
    if (!X || !W || !A || !B || !Y || !T_out) return LORA_E_BADARG;
    GemmParams p{};
    p.Am = X; p.Bm = W; p.bias = bias;
    p.F = A; p.f_sr = K; p.f_sk = 1;   // F[j,k] = A[j,k]
    p.Q = B; p.q_sn = r; p.q_sj = 1;   // Q[n,j] = B[n,j]
    p.C = Y; p.P = T_out;
    p.M = M; p.Kc = K; p.Nc = N; p.r = r; p.scale = scale;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const double e = esize(dtype);
    ProfWork work(e * ((double)M * K + (double)N * K + (double)M * N) + e * r * (K + N) + (bias ? e * N : 0.0),
                  2.0 * M * K * N + 2.0 * M * r * (double)(K + N));
    return launch_gemm(p, true, dtype, s);
}

extern "C" int lora_linear_bwd_input(const void* dY, const void* Wt, const float* A, const float* B,
                                     void* dX, float* U_out, int64_t M, int K, int N, int r, float scale,
                                     int dtype, void* stream) {
    const int st = check_common(M, K, N, r, dtype);
    if (st != LORA_OK) return st;
    if (M == 0) return LORA_OK;
    if (!dY || !A || !B || !U_out) return LORA_E_BADARG;
    if (dX && !Wt) return LORA_E_BADARG;
    GemmParams p{};
    p.Am = dY; p.Bm = Wt; p.bias = nullptr;
    p.F = B; p.f_sr = 1; p.f_sk = r;   // F[j,n] = B[n,j]
    p.Q = A; p.q_sn = 1; p.q_sj = K;   // Q[k,j] = A[j,k]
    p.C = dX; p.P = U_out;
    p.M = M; p.Kc = N; p.Nc = K; p.r = r; p.scale = scale;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const double e = esize(dtype);
    const double bytes = dX ? e * ((double)M * N + (double)N * K + (double)M * K) + e * r * (K + N)
                            : e * (double)M * N + e * r * N;
    const double flops = dX ? 2.0 * M * K * N + 2.0 * M * r * (double)(K + N) : 2.0 * M * r * (double)N;
    ProfWork work(bytes, flops);
    return launch_gemm(p, dX != nullptr, dtype, s);
}

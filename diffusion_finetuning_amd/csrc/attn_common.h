// Device helpers shared by the attention kernels (attn_ctx.hip: one key tile; attn_flash.hip: many).
#pragma once
#include "common.h"

namespace {

typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef short s16x4 __attribute__((ext_vector_type(4)));

template <typename T> struct Mma;
template <> struct Mma<half_t> {
    using F8 = f16x8;
    static __device__ __forceinline__ f32x4 k32(F8 a, F8 b, f32x4 c) {
        return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
    }
    static __device__ __forceinline__ f32x4 k16(const half_t* a, const half_t* b, f32x4 c) {
        const f16x4 av = {a[0], a[1], a[2], a[3]}, bv = {b[0], b[1], b[2], b[3]};
        return __builtin_amdgcn_mfma_f32_16x16x16f16(av, bv, c, 0, 0, 0);
    }
    static __device__ __forceinline__ f32x4 k16rr(s16x4 a, s16x4 b, f32x4 c) {  // both operands as raw 16-bit quads
        return __builtin_amdgcn_mfma_f32_16x16x16f16(__builtin_bit_cast(f16x4, a), __builtin_bit_cast(f16x4, b), c, 0, 0, 0);
    }
    static __device__ __forceinline__ f32x4 k16r(const half_t* a, s16x4 b, f32x4 c) {  // b: four raw 16-bit values
        const f16x4 av = {a[0], a[1], a[2], a[3]};
        return __builtin_amdgcn_mfma_f32_16x16x16f16(av, __builtin_bit_cast(f16x4, b), c, 0, 0, 0);
    }
};
template <> struct Mma<bf16_t> {
    using F8 = bf16x8;
    static __device__ __forceinline__ f32x4 k32(F8 a, F8 b, f32x4 c) {
        return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
    }
    static __device__ __forceinline__ f32x4 k16(const bf16_t* a, const bf16_t* b, f32x4 c) {
        s16x4 av, bv;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            av[e] = __builtin_bit_cast(short, a[e]);
            bv[e] = __builtin_bit_cast(short, b[e]);
        }
        return __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(av, bv, c, 0, 0, 0);
    }
    static __device__ __forceinline__ f32x4 k16rr(s16x4 a, s16x4 b, f32x4 c) {
        return __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a, b, c, 0, 0, 0);
    }
    static __device__ __forceinline__ f32x4 k16r(const bf16_t* a, s16x4 b, f32x4 c) {
        s16x4 av;
#pragma unroll
        for (int e = 0; e < 4; ++e) av[e] = __builtin_bit_cast(short, a[e]);
        return __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(av, b, c, 0, 0, 0);
    }
};

template <typename T> struct alignas(8) Quad4 { T v[4]; };

// gfx950 transposing LDS read (ds_read_b64_tr_b16): every group of 16 consecutive lanes reads a block of 4 rows × 16
// columns of 16-bit elements and receives it column-major — lane i of the group gets column i of the four rows.  Lane
// 4q+p of the group supplies the address of row q, columns 4p..4p+3 (8-byte aligned); EXEC must be all ones.  `rows` points
// at element [first row of the WAVE's blocks][first column]; the group g = lane>>4 takes rows 4g..4g+3 from there.
typedef short tr4 __attribute__((__vector_size__(4 * sizeof(short))));
template <typename T>
__device__ __forceinline__ tr4 lds_tr_block(const T* rows, int row_stride, int lane) {
    const int l15 = lane & 15, g = lane >> 4;
    const T* p = rows + (g * 4 + (l15 >> 2)) * row_stride + (l15 & 3) * 4;
    return __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) tr4*)(p));
}
// the same read with the lane's own address (4 contiguous elements of its row; 8-byte aligned): the 16 "columns" a group
// delivers need not be contiguous — column 4p+j is element j at the address lane 4q+p supplies
template <typename T> __device__ __forceinline__ tr4 lds_tr_at(const T* p) {
    return __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) tr4*)(p));
}
template <typename T> __device__ __forceinline__ typename Mma<T>::F8 tr_pair(tr4 lo, tr4 hi) {  // two blocks → one 8-wide operand
    typedef short s8 __attribute__((__vector_size__(8 * sizeof(short))));
    const s8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    return __builtin_bit_cast(typename Mma<T>::F8, v);
}

// position of a key inside the permuted key axis: fragments 2k and 2k+1 interleave in groups of four, which is the
// order in which a lane's accumulator registers of two neighbouring S fragments form one 8-wide MFMA operand
__device__ __forceinline__ int key_pos(int key) {
    return (key & ~31) | (((key >> 2) & 3) << 3) | (((key >> 4) & 1) << 2) | (key & 3);
}

// 16-byte load that is never predicated (a predicated load costs a branch and a full wait per load on this
// compiler): the caller passes an address that is valid either way, the value is zeroed afterwards when !ok
template <typename T> __device__ __forceinline__ Chunk<T> load_or_zero(const T* p, bool ok) {
    Chunk<T> v = *reinterpret_cast<const Chunk<T>*>(p);
#pragma unroll
    for (int e = 0; e < 8; ++e) v.v[e] = ok ? v.v[e] : from_f32<T>(0.f);
    return v;
}

// row fragments (second MFMA operand: lane = (row l15, 8 consecutive head-dim values at ks*32 + lq*8)).
// `safe` is any valid address of the tensor: rows past the end are read from there and zeroed.
template <typename T, int KS>
__device__ __forceinline__ void load_row_frags(const T* base, const T* safe, bool valid, int d, int lq,
                                               typename Mma<T>::F8 (&f)[KS]) {
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
        const int c = ks * 32 + lq * 8;
        const bool ok = valid && c < d;
        const Chunk<T> v = load_or_zero<T>(ok ? base + c : safe, ok);
#pragma unroll
        for (int e = 0; e < 8; ++e) f[ks][e] = v.v[e];
    }
}

// two neighbouring accumulator fragments → one 8-wide operand over 32 (permuted) keys
template <typename T>
__device__ __forceinline__ typename Mma<T>::F8 pair_frag(const f32x4& a, const f32x4& b) {
    typename Mma<T>::F8 f;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        f[e] = from_f32<T>(a[e]);
        f[4 + e] = from_f32<T>(b[e]);
    }
    return f;
}


}  // namespace

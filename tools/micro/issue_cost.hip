// Dev micro-benchmark (gfx950): what a slot "one 16x16x32 MFMA + a few vector instructions" costs ONE wave per SIMD, in
// shader cycles (s_memtime around 2000 repetitions of an unrolled group of 8 slots).
// hipcc --offload-arch=gfx950 -O3 tools/micro/issue_cost.hip -o /tmp/issue_cost && /tmp/issue_cost
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

// one asm statement per slot: hipcc pads every statement with an s_nop, which would double the cost of a one-instruction statement
#define MF "v_mfma_f32_16x16x32_f16 %0, %1, %2, %0\n\t"
#define MB "v_mfma_f32_32x32x16_f16 %0, %1, %2, %0\n\t"
#define E0 "v_exp_f32 %3, %7\n\t"
#define E1 "v_exp_f32 %4, %7\n\t"
#define X0 "v_mul_f32 %3, %7, %8\n\t"
#define X1 "v_mul_f32 %4, %7, %8\n\t"
#define X2 "v_mul_f32 %5, %7, %8\n\t"
#define X3 "v_mul_f32 %6, %7, %8\n\t"
#define C0 "v_cvt_pk_f16_f32 %3, %7, %8\n\t"
#define C1 "v_cvt_pk_f16_f32 %4, %7, %8\n\t"
#define C2 "v_cvt_pk_f16_f32 %5, %7, %8\n\t"
#define C3 "v_cvt_pk_f16_f32 %6, %7, %8\n\t"
#define PK "v_pk_mul_f32 %9, %10, %11\n\t"
#define FM0 "v_fma_mixlo_f16 %3, %7, %8, 0\n\t"
#define FM1 "v_fma_mixhi_f16 %3, %7, %8, 0\n\t"
#define SLOT(str, i) asm volatile(str : "+v"(acc[i]), "+v"(a8), "+v"(b8), "=v"(t0), "=v"(t1), "=v"(t2), "=v"(t3) : "v"(src), "v"(src2), "v"(r2), "v"(p2), "v"(q2))
#define SLOTB(str, i) asm volatile(str : "+v"(big[i & 3]), "+v"(a8), "+v"(b8), "=v"(t0), "=v"(t1), "=v"(t2), "=v"(t3) : "v"(src), "v"(src2), "v"(r2), "v"(p2), "v"(q2))
#define SLOTA(str, i) asm volatile(str : "+v"(acc[i]), "+v"(a8), "+a"(bA), "=v"(t0), "=v"(t1), "=v"(t2), "=v"(t3) : "v"(src), "v"(src2), "v"(r2), "v"(p2), "v"(q2))
#define SLOTC(str, i) asm volatile(str : "+a"(accA[i]), "+v"(a8), "+v"(b8), "=v"(t0), "=v"(t1), "=v"(t2), "=v"(t3) : "v"(src), "v"(src2), "v"(r2), "v"(p2), "v"(q2))

template <int MODE, int OCC> __global__ __launch_bounds__(256 * OCC) void k(unsigned long long* out, int iters) {
    __shared__ float lds[4096];
    // (OCC = 2: ONE workgroup of 8 waves per CU = two waves per SIMD, certainly co-resident)
    f32x4 acc[8], accA[8];
    f32x16 big[4];
    for (int i = 0; i < 4; ++i) for (int e = 0; e < 16; ++e) big[i][e] = 0.f;
    for (int i = 0; i < 8; ++i) acc[i] = accA[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    f16x8 a8, b8, bA;
    for (int e = 0; e < 8; ++e) a8[e] = b8[e] = bA[e] = (_Float16)(threadIdx.x * 0.001f + e);
    float src = threadIdx.x * 0.01f, src2 = 1.5f, t0, t1, t2, t3;
    f32x2 p2 = {src, src2}, q2 = {src2, src}, r2 = {0.f, 0.f};
    lds[threadIdx.x & 1023] = src;
    __syncthreads();
    const unsigned long long c0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if (MODE == 0) SLOT(MF, i);
            if (MODE == 1) SLOTA(MF, i);
            if (MODE == 2) SLOTC(MF, i);
            if (MODE == 3) SLOT(MF E0 E1, i);
            if (MODE == 4) SLOT(MF C0 C1 C2 C3, i);
            if (MODE == 5) SLOT(MF X0 X1 C2 PK C3, i);
            if (MODE == 6) SLOT(MF X0 X1 C2 X2 X3 C3, i);
            if (MODE == 7) SLOT(MF E0, i);
            if (MODE == 8) SLOT(MF X0 X1, i);
            if (MODE == 9) SLOT(MF X0 X1 X2 X3, i);
            if (MODE == 10) SLOT(MF FM0 FM1, i);
            if (MODE == 11) SLOT(E0 E1 E0 E1, i);
            if (MODE == 12) SLOT(X0 X1 X2 X3, i);
            if (MODE == 13) SLOT(PK PK, i);
            if (MODE == 14) SLOT(C0 C1 C2 C3, i);
            if (MODE == 15) SLOT(MF E0 E1 X2 X3, i);
            if (MODE == 16) SLOT(MF E0 X2, i);
            if (MODE == 17) SLOT(MF E0 X2 X3, i);
            if (MODE == 18) SLOT(MF C0 C1, i);
            if (MODE == 19) SLOT(MF X0 X1 X2, i);
            if (MODE == 20) SLOT(FM0 FM1 FM0 FM1, i);
            if (MODE == 21) SLOT(MF E0 E1 X2 X3 C0, i);
            if (MODE == 22) SLOT(MF MF MF MF E0 E1 X2 X3 C0 E0 E1 X2 X3 C0 E0 E1 X2 X3 C0 E0 E1 X2 X3 C0, i);
            if (MODE == 23) SLOT(E0 E1 X2 X3 C0, i);
            if (MODE == 24) SLOTB(MB, i);
            if (MODE == 25) SLOTB(MB E0 E1, i);
            if (MODE == 26) SLOTB(MB X0 X1 X2 X3, i);
            if (MODE == 27) SLOTB(MB E0 E1 X2 X3 C0, i);
            if (MODE == 28) SLOTB(MB E0 E1 X2 X3 C0 C1 X0, i);
            if (MODE == 29) SLOTB(MB E0 E1 E0 E1 X2 X3 C0 C1 X0 X1, i);
        }
    }
    __syncthreads();  // every wave of the workgroup has finished its slots (an older wave wins the arbitration and ends early)
    const unsigned long long c1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
    for (int i = 0; i < 8; ++i) s += acc[i][0] + accA[i][1] + big[i & 3][i];
    if (threadIdx.x == 0 && blockIdx.x == 0) out[MODE + 32 * (OCC - 1)] = c1 - c0;
    if (s == 12345.678f) out[63] = (unsigned long long)s;  // keep everything alive
}
int main() {
    unsigned long long* d; hipMalloc(&d, 64 * 8); hipMemset(d, 0, 64 * 8);
    const int iters = 2000;
    const char* names[] = {"M (all VGPR)", "M, B operand in AGPR", "M, accumulator in AGPR", "M e e", "M c c c c", "M x x c X c", "M x x c x x c",
                           "M e", "M x x", "M x x x x", "M fma_mixlo fma_mixhi", "e e e e (no MFMA)", "x x x x (no MFMA)", "X X (no MFMA)", "c c c c (no MFMA)",
                           "M e e x x", "M e x", "M e x x", "M c c", "M x x x", "mixlo mixhi mixlo mixhi (no MFMA)", "M e e x x c", "4M then 4x(e e x x c)  [per 4 slots]", "e e x x c (no MFMA)", "M32 (32x32x16)", "M32 e e", "M32 x x x x", "M32 e e x x c", "M32 e e x x c c x", "M32 e e e e x x c c x x"};
#define RUN(m) hipLaunchKernelGGL((k<m, 1>), dim3(256), dim3(256), 0, 0, d, iters); hipLaunchKernelGGL((k<m, 2>), dim3(256), dim3(512), 0, 0, d, iters);
    RUN(0) RUN(1) RUN(2) RUN(3) RUN(4) RUN(5) RUN(6) RUN(7) RUN(8) RUN(9) RUN(10) RUN(11) RUN(12) RUN(13) RUN(14) RUN(15) RUN(16) RUN(17) RUN(18) RUN(19) RUN(20) RUN(21) RUN(22) RUN(23) RUN(24) RUN(25) RUN(26) RUN(27) RUN(28) RUN(29)
    hipDeviceSynchronize();
    unsigned long long h[64]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    printf("%-44s %10s %28s\n", "slot", "1 wave/SIMD", "2 waves/SIMD: per slot PAIR (one slot of each wave)");
    for (int m = 0; m < 30; ++m) printf("%-44s %7.1f %14.1f\n", names[m], (double)h[m] / (iters * 8.0), (double)h[m + 32] / (iters * 8.0));
    return 0;
}

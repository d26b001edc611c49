// Dev micro-benchmark: issue rate of the 16x16x16 (CDNA3-era) against the 16x16x32 (CDNA4) f16 MFMA on gfx950.
// hipcc --offload-arch=gfx950 -O3 tools/micro/mfma_rate.hip -o /tmp/mfma_rate && /tmp/mfma_rate
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int MODE> __global__ __launch_bounds__(256) void k(float* out, int iters) {
    f32x4 acc[8];
    for (int i = 0; i < 8; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    f16x8 a8; f16x4 a4;
    for (int e = 0; e < 8; ++e) a8[e] = (_Float16)(threadIdx.x * 0.001f + e);
    for (int e = 0; e < 4; ++e) a4[e] = a8[e];
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            // inline asm on fixed VGPR accumulators: the builtin form drags accumulator moves into this loop
            if (MODE == 0) asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %1, %0" : "+v"(acc[i]) : "v"(a8));
            else asm volatile("v_mfma_f32_16x16x16_f16 %0, %1, %1, %0" : "+v"(acc[i]) : "v"(a4));
        }
    }
    float s = 0.f;
    for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
int main() {
    float* d; hipMalloc(&d, 1024 * 256 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 20000;
    for (int mode = 0; mode < 2; ++mode) {
        for (int rep = 0; rep < 2; ++rep) {
            hipEventRecord(e0);
            if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(256), dim3(256), 0, 0, d, iters);   // one wave per SIMD
            else hipLaunchKernelGGL(k<1>, dim3(256), dim3(256), 0, 0, d, iters);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            const double n = (double)iters * 8;  // MFMAs per wave
            printf("%s: %.3f ms, %.2f ns per MFMA per wave (one wave per SIMD) -> %.1f cycles at 2.1 GHz, %.0f TFLOP/s chip\n",
                   mode == 0 ? "16x16x32" : "16x16x16", ms, ms * 1e6 / n, ms * 1e6 / n * 2.1,
                   n * 1024 * (mode == 0 ? 16384.0 : 8192.0) / (ms * 1e-3) / 1e12);
        }
    }
    return 0;
}

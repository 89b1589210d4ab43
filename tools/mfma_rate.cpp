// Developer tool: raw v_mfma_f32_32x32x2_f32 issue rate (dependent chain vs independent accumulators).
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int NACC>
__global__ void __launch_bounds__(256) k(float* out, int iters, float a0, float b0) {
    f32x16 acc[NACC];
    for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    float a = a0 + threadIdx.x, b = b0 - threadIdx.x;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
            for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
    }
    float s = 0; for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int NACC> void run(int blocks_per_cu) {
    float* out; hipMalloc(&out, 256 * 256 * 16 * 4);
    int iters = 2000 / NACC; hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    int grid = 256 * blocks_per_cu;
    hipLaunchKernelGGL(k<NACC>, dim3(grid), dim3(256), 0, 0, out, 10, 1.f, 2.f);
    hipDeviceSynchronize();
    hipEventRecord(e0); hipLaunchKernelGGL(k<NACC>, dim3(grid), dim3(256), 0, 0, out, iters, 1.f, 2.f); hipEventRecord(e1);
    hipEventSynchronize(e1); float ms; hipEventElapsedTime(&ms, e0, e1);
    double n = (double)grid * 4 * iters * 8 * NACC;
    printf("NACC=%d blocks/CU=%d: %.3f ms, %.1f TFLOP/s, %.1f cycles/MFMA/SIMD @2.4GHz\n", NACC, blocks_per_cu, ms,
           n * 4096 / ms * 1e-9, ms * 1e-3 * 2.4e9 / (n / 1024));
}
int main() { for (int b : {1, 2, 4}) { run<1>(b); run<2>(b); run<4>(b); } return 0; }

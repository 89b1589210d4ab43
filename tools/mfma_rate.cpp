// Developer tool: raw fp32 MFMA issue rate on gfx950 as a function of waves per SIMD and independent accumulators,
// with and without the LDS fragment reads a GEMM loop needs (2 ds_read_b32 per 32x32x2 MFMA for a 1x1-fragment wave
// tile, 1 per MFMA for a 2x2 one), on random operands.
//   hipcc --offload-arch=gfx950 -O3 -x hip tools/mfma_rate.cpp -o tools/mfma_rate && tools/mfma_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// TD x TP fragments per wave (TD*TP accumulators); LDS: 0 = operands stay in registers, 1 = fragments re-read from LDS
// for every k-step (TD + TP ds_read_b32 per TD*TP MFMAs), software-pipelined one step ahead
template <int TD, int TP, int LDS, int M16>
__global__ void __launch_bounds__(256) k(float* out, const float* in, int iters) {
    __shared__ float sm[16 * 256];
    for (int i = threadIdx.x; i < 16 * 256; i += 256) sm[i] = in[i];
    __syncthreads();
    typedef typename std::conditional<M16 != 0, f32x4, f32x16>::type acc_t;
    constexpr int NR = M16 ? 4 : 16;
    acc_t acc[TD][TP];
    for (int i = 0; i < TD; ++i) for (int j = 0; j < TP; ++j) for (int r = 0; r < NR; ++r) acc[i][j][r] = 0.f;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float fa[2][TD], fb[2][TP];
    for (int i = 0; i < TD; ++i) fa[0][i] = fa[1][i] = in[lane + 64 * i];
    for (int j = 0; j < TP; ++j) fb[0][j] = fb[1][j] = in[1024 + lane + 64 * j];
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int s = 0; s < 8; ++s) {
            const int set = s & 1;
            if (LDS) {
#pragma unroll
                for (int i = 0; i < TD; ++i) fa[set ^ 1][i] = sm[((s + 1) & 7) * 256 + i * 64 + lane];
#pragma unroll
                for (int j = 0; j < TP; ++j) fb[set ^ 1][j] = sm[(8 + ((s + 1) & 7)) * 256 + ((j + wave) & 3) * 64 + lane];
                __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int i = 0; i < TD; ++i)
#pragma unroll
                for (int j = 0; j < TP; ++j) {
                    if constexpr (M16 != 0) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[set][i], fb[set][j], acc[i][j], 0, 0, 0);
                    else acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[set][i], fb[set][j], acc[i][j], 0, 0, 0);
                }
        }
    }
    float s = 0;
    for (int i = 0; i < TD; ++i) for (int j = 0; j < TP; ++j) for (int r = 0; r < NR; ++r) s += acc[i][j][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

static float *g_out, *g_in;
template <int TD, int TP, int LDS, int M16> void run(int blocks_per_cu) {
    const int per = TD * TP * 8;                       // MFMAs per wave per iteration
    int iters = 16000 / per; if (M16) iters *= 2;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int grid = 256 * blocks_per_cu;
    hipLaunchKernelGGL((k<TD, TP, LDS, M16>), dim3(grid), dim3(256), 0, 0, g_out, g_in, 10);
    hipDeviceSynchronize();
    hipEventRecord(e0); hipLaunchKernelGGL((k<TD, TP, LDS, M16>), dim3(grid), dim3(256), 0, 0, g_out, g_in, iters); hipEventRecord(e1);
    hipEventSynchronize(e1); float ms; hipEventElapsedTime(&ms, e0, e1);
    const double n = (double)grid * 4 * iters * per;
    const double fl = M16 ? 2048.0 : 4096.0;
    printf("%s frags %dx%d lds=%d waves/SIMD=%d: %7.1f TFLOP/s  %.1f cycles/MFMA/SIMD @2.4GHz\n", M16 ? "16x16x4" : "32x32x2", TD, TP, LDS,
           blocks_per_cu, n * fl / ms * 1e-9, ms * 1e-3 * 2.4e9 / (n / 1024));
}
int main() {
    hipMalloc(&g_out, 256 * 8 * 256 * 4); hipMalloc(&g_in, 16 * 256 * 4);
    float h[16 * 256]; for (float& v : h) v = (rand() % 2001 - 1000) * 1e-3f;
    hipMemcpy(g_in, h, sizeof h, hipMemcpyHostToDevice);
    for (int b : {1, 2, 4, 7}) {
        run<1, 1, 0, 0>(b); run<1, 1, 1, 0>(b); run<2, 2, 0, 0>(b); run<2, 2, 1, 0>(b);
        run<1, 4, 0, 1>(b); run<1, 4, 1, 1>(b); run<2, 2, 1, 1>(b);
    }
    return 0;
}

// Developer probe: what a ONE-block-per-CU, big-accumulator fp32 MFMA tile sustains with LDS-DMA staging on gfx950.
// C[BD x BP] += A[K x BD]^T B[K x BP] per block, BD = 128 output channels x BP = 208 pixels (13 fragments of 16: one 14x14
// frame = 196 pixels fits with 6 % padding), 16x16x4 MFMA, 4 waves (one per SIMD, 2 x 13 accumulator fragments each),
// K consumed in chunks of 16 through NST LDS stages filled by 16-byte buffer LDS-DMA, counted vmcnt, raw s_barrier.
//   hipcc --offload-arch=gfx950 -O3 -std=c++20 -x hip tools/bigtile_probe.cpp -o tools/bigtile_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void* lds_ptr_t;

template <int BD, int NPF, int NST>      // NPF pixel fragments of 16
__global__ void __launch_bounds__(256) k(const float* __restrict__ A, const float* __restrict__ B, float* __restrict__ C, int K) {
    constexpr int BP = NPF * 16, KC = 16;
    constexpr int AF = KC * BD, BF = KC * BP;                    // floats per stage
    constexpr int NA = AF / 256, NB = (BF + 255) / 256;          // 1 KiB DMA pieces per stage
    __shared__ __attribute__((aligned(16))) float smem[NST * (AF + NB * 256)];
    const int t = threadIdx.x, lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int l15 = lane & 15, lk = lane >> 4;
    const float* Ab = A;                                          // weights shared by every block (L2 resident)
    const float* Bb = B + (size_t)(blockIdx.x & 255) * K * BP;           // this block's activation panel
    const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc((void*)Ab, 0, K * BD * 4, 0x00020000);
    const __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc((void*)Bb, 0, K * BP * 4, 0x00020000);
    f32x4 acc[2][NPF];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < NPF; ++j) acc[i][j] = f32x4{0, 0, 0, 0};
    const int nchunks = K / KC;
    constexpr int NPW = (NA + NB + 3) / 4;                        // pieces per wave per stage
    auto issue = [&](int c, int st) {
        float* as = smem + st * (AF + NB * 256);
        float* bs = as + AF;
#pragma unroll
        for (int q = 0; q < NPW; ++q) {
            const int ins = wave + 4 * q;
            if (ins < NA)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(ra, (lds_ptr_t)(as + ins * 256), 16, lane * 16 + ins * 1024, c * AF * 4, 0, 0);
            else if (ins < NA + NB) {
                const int ib = ins - NA;
                unsigned off = lane * 16 + ib * 1024;
                if (off >= (unsigned)(BF * 4)) off = 0x80000000u;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rb, (lds_ptr_t)(bs + ib * 256), 16, off, c * BF * 4, 0, 0);
            }
        }
    };
    for (int c = 0; c < NST - 1 && c < nchunks; ++c) issue(c, c);
    int st = 0;
    for (int c = 0; c < nchunks; ++c) {
        // own DMA of chunk c landed (the NST-2 younger chunks may still be in flight), then everyone's
        if (NST >= 3 && c + 1 < nchunks) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NPW * (NST - 2)) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (c + NST - 1 < nchunks) issue(c + NST - 1, (st + NST - 1) % NST);
        const float* as = smem + st * (AF + NB * 256);
        const float* bs = as + AF;
        float fa[2][2], fb[2][NPF];
        auto rd = [&](int s, int set) {
#pragma unroll
            for (int i = 0; i < 2; ++i) fa[set][i] = as[(4 * s + lk) * BD + wave * 32 + i * 16 + l15];
#pragma unroll
            for (int j = 0; j < NPF; ++j) fb[set][j] = bs[(4 * s + lk) * BP + j * 16 + l15];
        };
        rd(0, 0);
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            if (s + 1 < 4) rd(s + 1, (s + 1) & 1);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < NPF; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[s & 1][i], fb[s & 1][j], acc[i][j], 0, 0, 0);
        }
        st = (st + 1) % NST;
    }
    // store (row = channel, 16 consecutive pixels per lane group)
    float* Cb = C + (size_t)(blockIdx.x & 255) * BD * BP;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < NPF; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) Cb[(size_t)(wave * 32 + i * 16 + 4 * lk + r) * BP + j * 16 + l15] = acc[i][j][r];
}

template <int BD, int NPF, int NST> void run(int K, int blocks, const float* dA, const float* dB, float* dC, std::vector<float>& hA, std::vector<float>& hB) {
    constexpr int BP = NPF * 16;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k<BD, NPF, NST>), dim3(blocks), dim3(256), 0, 0, dA, dB, dC, K);
    hipDeviceSynchronize();
    // check block 0, a few entries
    std::vector<float> c0(BD * BP); hipMemcpy(c0.data(), dC, c0.size() * 4, hipMemcpyDeviceToHost);
    double maxerr = 0;
    for (int cd : {0, 17, 127 % BD}) for (int px : {0, 15, 100 % BP, BP - 1}) {
        double ref = 0; for (int kk = 0; kk < K; ++kk) ref += (double)hA[(size_t)kk * BD + cd] * hB[(size_t)kk * BP + px];
        maxerr = fmax(maxerr, fabs(ref - c0[cd * BP + px]));
    }
    const int iters = 10;
    hipEventRecord(e0);
    for (int i = 0; i < iters; ++i) hipLaunchKernelGGL((k<BD, NPF, NST>), dim3(blocks), dim3(256), 0, 0, dA, dB, dC, K);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= iters;
    printf("BD=%d BP=%d (%d frags) stages=%d K=%d blocks=%d: %.3f ms  %.1f TFLOP/s useful  (maxerr %.2e)\n", BD, BP, NPF, NST, K, blocks, ms,
           2.0 * blocks * BD * BP * (double)K / ms * 1e-9, maxerr);
}

int main() {
    const int K = 2304, blocks = 256;
    std::vector<float> hA((size_t)K * 128), hB((size_t)blocks * K * 256);
    for (auto& v : hA) v = (rand() % 2001 - 1000) * 1e-3f;
    for (auto& v : hB) v = (rand() % 2001 - 1000) * 1e-3f;
    float *dA, *dB, *dC;
    hipMalloc(&dA, hA.size() * 4); hipMalloc(&dB, hB.size() * 4); hipMalloc(&dC, (size_t)blocks * 128 * 256 * 4);
    hipMemcpy(dA, hA.data(), hA.size() * 4, hipMemcpyHostToDevice); hipMemcpy(dB, hB.data(), hB.size() * 4, hipMemcpyHostToDevice);
    run<128, 13, 2>(K, blocks, dA, dB, dC, hA, hB);
    run<128, 13, 3>(K, blocks, dA, dB, dC, hA, hB);
    run<128, 13, 4>(K, blocks, dA, dB, dC, hA, hB);
    run<128, 8, 3>(K, blocks, dA, dB, dC, hA, hB);
    run<128, 16, 3>(K, blocks, dA, dB, dC, hA, hB);
    run<128, 13, 3>(1024, blocks, dA, dB, dC, hA, hB);
    run<128, 13, 3>(K, 512, dA, dB, dC, hA, hB);
    return 0;
}

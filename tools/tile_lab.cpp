// Developer lab (round 3): main-loop schedules of the 64x64 fp32-MFMA tile, isolated from the convolution's address logic.
// The per-block timeline of conv_igemm (tools/conv_microbench.cpp -DX_CLOCK) showed that the co-resident blocks of a CU do NOT
// finish together (the hardware arbitrates oldest wave first): the last block of a CU runs alone, and a lone block -- one wave
// per SIMD -- reaches only about half of the matrix pipe.  This lab measures what a lone wave per SIMD can sustain under
// different instruction schedules, with the product kernel's staging (buffer LDS-DMA, 16 K rows per chunk):
//   SCHED 0  product schedule: vmcnt(0) + barrier at the top of a chunk, fragments read ONE k-step ahead, 2 LDS stages
//   SCHED 1  as 0, but every fragment of the chunk is requested right after the barrier (counted lgkmcnt waits)
//   SCHED 2  3 LDS stages, DMA two chunks ahead, barrier in the shadow of MFMA 5, fragments TWO k-steps ahead across the chunk seam
// C[cd][px] = sum_k A[k][cd] * B[k][px];  A = K x 256 (weights, shared), B = K x P.  Block tile 64 x 64, 4 waves (2 x 2) of 32 x 32.
//   hipcc --offload-arch=gfx950 -O3 -std=c++20 -x hip tools/tile_lab.cpp -o tools/tile_lab
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <algorithm>
#include <utility>
#include <vector>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __attribute__((address_space(3))) void* lds_ptr_t;
constexpr int KC = 16, BD = 64, BP = 64, KS = 8;

// BW = bytes per lane of the B-tile DMA pieces: 16 (pointwise layers) or 4 (3x3 layers: one 64-pixel row per instruction)
template <int SCHED, int BW>
__global__ void __launch_bounds__(256) lab(const float* __restrict__ A, const float* __restrict__ B, float* __restrict__ C, int K, int P, int CD,
                                           unsigned long long* __restrict__ cyc) {
    constexpr int NST = SCHED == 2 ? 3 : 2;
    __shared__ __attribute__((aligned(16))) float smem[NST * KC * (BD + BP)];
    float (*As)[KC][BD] = reinterpret_cast<float (*)[KC][BD]>(smem);
    float (*Bs)[KC][BP] = reinterpret_cast<float (*)[KC][BP]>(smem + NST * KC * BD);
    const int t = threadIdx.x, lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int wd = wave >> 1, wp = wave & 1, l31 = lane & 31, lk = lane >> 5;
    const int n_cd = CD / BD, cd0 = (blockIdx.x % n_cd) * BD, px0 = (blockIdx.x / n_cd) * BP;
    const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc((void*)A, 0, K * CD * 4, 0x00020000);
    const __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc((void*)B, 0, (unsigned)((size_t)K * P * 4), 0x00020000);
    // A piece of wave w: floats [256 w, 256 w + 256) of the [16][64] tile: row 4 w + lane / 16, columns 4 (lane % 16)
    const unsigned aoff = (unsigned)(((4 * wave + (lane >> 4)) * CD + cd0 + 4 * (lane & 15)) * 4);
    const unsigned boff16 = (unsigned)((((4 * wave + (lane >> 4)) * (size_t)P) + px0 + 4 * (lane & 15)) * 4);
    const unsigned boff4 = (unsigned)((px0 + lane) * 4);                 // row given by the scalar offset
    constexpr int NL = BW == 16 ? 2 : 5;                                  // DMA instructions per wave per chunk
    auto piece = [&](auto jtag, int c, int buf) {
        constexpr int j = decltype(jtag)::value;
        if constexpr (j == 0) __builtin_amdgcn_raw_ptr_buffer_load_lds(ra, (lds_ptr_t)(&As[buf][0][0] + wave * 256), 16, aoff, c * KC * CD * 4, 0, 0);
        else if constexpr (BW == 16) __builtin_amdgcn_raw_ptr_buffer_load_lds(rb, (lds_ptr_t)(&Bs[buf][0][0] + wave * 256), 16, boff16, (unsigned)((size_t)c * KC * P * 4), 0, 0);
        else {
            const int row = wave + 4 * (j - 1);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rb, (lds_ptr_t)(&Bs[buf][row][0]), 4, boff4, (unsigned)(((size_t)c * KC + row) * P * 4), 0, 0);
        }
    };
    auto issue_all = [&](int c, int buf) {
        [&]<int... J>(std::integer_sequence<int, J...>) { (piece(std::integral_constant<int, J>{}, c, buf), ...); }(std::make_integer_sequence<int, NL>{});
    };
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    const int nchunks = K / KC;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    if constexpr (SCHED == 0 || SCHED == 1) {
        issue_all(0, 0);
        int buf = 0;
        for (int c = 0; c < nchunks; ++c) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            const bool more = c + 1 < nchunks;
            if constexpr (SCHED == 0) {
                float fa[2], fb[2];
                auto rd = [&](int s, int set) { fa[set] = As[buf][2 * s + lk][wd * 32 + l31]; fb[set] = Bs[buf][2 * s + lk][wp * 32 + l31]; };
                rd(0, 0);
                [&]<int... S>(std::integer_sequence<int, S...>) {
                    (([&] {
                        constexpr int s = S, set = S & 1;
                        if constexpr (s + 1 < KS) rd(s + 1, set ^ 1);
                        __builtin_amdgcn_sched_barrier(0);
                        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[set], fb[set], acc, 0, 0, 0);
                        if constexpr (s < NL) { if (more) piece(std::integral_constant<int, s>{}, c + 1, buf ^ 1); }
                    }()), ...);
                }(std::make_integer_sequence<int, KS>{});
            } else {
                float fa[KS], fb[KS];
#pragma unroll
                for (int s = 0; s < KS; ++s) { fa[s] = As[buf][2 * s + lk][wd * 32 + l31]; fb[s] = Bs[buf][2 * s + lk][wp * 32 + l31]; }
                __builtin_amdgcn_sched_barrier(0);
                [&]<int... S>(std::integer_sequence<int, S...>) {
                    (([&] {
                        constexpr int s = S;
                        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[s], fb[s], acc, 0, 0, 0);
                        if constexpr (s < NL) { if (more) piece(std::integral_constant<int, s>{}, c + 1, buf ^ 1); }
                        __builtin_amdgcn_sched_barrier(0);
                    }()), ...);
                }(std::make_integer_sequence<int, KS>{});
            }
            buf ^= 1;
        }
    } else {
        // 3 stages.  Chunk c lives in buffer c % 3.  Iteration c: MFMAs of chunk c; DMA of chunk c + 2 behind MFMAs 0..NL-1 (its
        // buffer held chunk c - 1, which every wave finished reading before the barrier of iteration c - 1); behind MFMA 5 the wave
        // waits for ITS pieces of chunk c + 1 (all but the NL youngest DMA instructions) and joins the barrier; the fragments of
        // k-steps 0, 1 of chunk c + 1 are requested behind MFMAs 6, 7.  Fragments are always two k-steps ahead of their MFMA.
        issue_all(0, 0);
        if (nchunks > 1) issue_all(1, 1);
        if (nchunks > 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NL) : "memory"); else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        float fa[4], fb[4];
        auto rd = [&](int buf, int s, int set) { fa[set] = As[buf][2 * s + lk][wd * 32 + l31]; fb[set] = Bs[buf][2 * s + lk][wp * 32 + l31]; };
        rd(0, 0, 0); rd(0, 1, 1);
        int b0 = 0, b1 = 1, b2 = 2;                                       // buffers of chunks c, c + 1, c + 2
        for (int c = 0; c < nchunks; ++c) {
            const bool more1 = c + 1 < nchunks, more2 = c + 2 < nchunks;
            // the two fragment sets carried across the seam are "used" here: hipcc places its wait for them HERE (they were requested
            // two MFMAs ago: free) instead of behind the next ds_reads with lgkmcnt(0), which would serialise read -> MFMA again
            asm volatile("" :: "v"(fa[0]), "v"(fb[0]), "v"(fa[1]), "v"(fb[1]));
            [&]<int... S>(std::integer_sequence<int, S...>) {
                (([&] {
                    constexpr int s = S, set = S % 4, nset = (S + 2) % 4;      // ring of 4: KS % 4 == 0, so k-step 0 of every chunk is set 0
                    if constexpr (s + 2 < KS) rd(b0, s + 2, nset);
                    else rd(b1, s + 2 - KS, nset);                        // (last chunk: reads a stale buffer, never used)
                    __builtin_amdgcn_sched_barrier(0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[set], fb[set], acc, 0, 0, 0);
                    if constexpr (s < NL) { if (more2) piece(std::integral_constant<int, s>{}, c + 2, b2); }
                    if constexpr (s == 5) {
                        if (more1) {
                            if (more2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NL) : "memory"); else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                            __builtin_amdgcn_s_barrier();
                        }
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }()), ...);
            }(std::make_integer_sequence<int, KS>{});
            const int tb = b0; b0 = b1; b1 = b2; b2 = tb;
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (t == 0) cyc[blockIdx.x] = t1 - t0;
    // store: lane l, register r -> pixel l31, channel (r & 3) + 8 (r >> 2) + 4 lk
    float* Cb = C + (size_t)(cd0 + wd * 32) * P + px0 + wp * 32 + l31;
#pragma unroll
    for (int r = 0; r < 16; ++r) Cb[(size_t)((r & 3) + 8 * (r >> 2) + 4 * lk) * P] = acc[r];
}

template <int SCHED, int BW>
static void run(const char* name, int K, int per_cu, const float* dA, const float* dB, float* dC, unsigned long long* dcyc,
                const std::vector<float>& hA, const std::vector<float>& hB, int Pmax) {
    const int CD = 256, blocks = 256 * per_cu, P = blocks / (CD / BD) * BP;
    if (P > Pmax) { printf("skip\n"); return; }
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    auto launch = [&] { hipLaunchKernelGGL((lab<SCHED, BW>), dim3(blocks), dim3(256), 0, 0, dA, dB, dC, K, P, CD, dcyc); };
    launch(); hipDeviceSynchronize();
    std::vector<float> c((size_t)CD * P); hipMemcpy(c.data(), dC, c.size() * 4, hipMemcpyDeviceToHost);
    double maxerr = 0;
    for (int cd : {0, 37, 255}) for (int px : {0, 31, 64 + 5, P - 1}) {
        double ref = 0; for (int kk = 0; kk < K; ++kk) ref += (double)hA[(size_t)kk * CD + cd] * hB[(size_t)kk * P + px];
        maxerr = std::max(maxerr, fabs(ref - c[(size_t)cd * P + px]));
    }
    hipEventRecord(e0); launch(); hipEventRecord(e1); hipEventSynchronize(e1);
    float one; hipEventElapsedTime(&one, e0, e1);
    const int warm = (int)(60.f / std::max(one, 1e-3f)) + 1, iters = std::max(10, (int)(30.f / std::max(one, 1e-3f)) + 1);
    for (int i = 0; i < warm; ++i) launch();
    hipEventRecord(e0);
    for (int i = 0; i < iters; ++i) launch();
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= iters;
    std::vector<unsigned long long> cy(blocks); hipMemcpy(cy.data(), dcyc, blocks * 8, hipMemcpyDeviceToHost);
    std::sort(cy.begin(), cy.end());
    const double per_chunk = (double)cy[blocks / 2] / (K / KC);
    printf("%-34s K=%4d blocks/CU=%d: %7.1f us %6.1f TFLOP/s | K-loop cycles/chunk med %.0f max %.0f (ideal %d) | maxerr %.1e\n", name, K, per_cu, ms * 1e3,
           2.0 * CD * (double)P * K / ms * 1e-9, per_chunk, (double)cy.back() / (K / KC), 512 * per_cu, maxerr);
    fflush(stdout);
}

int main(int argc, char** argv) {
    const int Kmax = 2304, Pmax = 256 * 8 / 4 * 64;
    std::vector<float> hA((size_t)Kmax * 256), hB((size_t)Kmax * Pmax);
    for (auto& v : hA) v = (rand() % 2001 - 1000) * 1e-3f;
    for (auto& v : hB) v = (rand() % 2001 - 1000) * 1e-3f;
    float *dA, *dB, *dC; unsigned long long* dcyc;
    hipMalloc(&dA, hA.size() * 4); hipMalloc(&dB, hB.size() * 4); hipMalloc(&dC, (size_t)256 * Pmax * 4); hipMalloc(&dcyc, 8 * 256 * 8);
    hipMemcpy(dA, hA.data(), hA.size() * 4, hipMemcpyHostToDevice);
    for (int K : {2304, 256}) for (int per_cu : {1, 2, 4, 7}) {
        // B is laid out [K][P] for the P of this run
        const int P = 256 * per_cu / 4 * 64;
        std::vector<float> hb((size_t)K * P);
        for (size_t i = 0; i < hb.size(); ++i) hb[i] = hB[i];
        hipMemcpy(dB, hb.data(), hb.size() * 4, hipMemcpyHostToDevice);
        run<0, 16>("sched0 product, 16-byte B pieces", K, per_cu, dA, dB, dC, dcyc, hA, hb, Pmax);
        run<1, 16>("sched1 read-all, 16-byte B", K, per_cu, dA, dB, dC, dcyc, hA, hb, Pmax);
        run<2, 16>("sched2 3-stage seam, 16-byte B", K, per_cu, dA, dB, dC, dcyc, hA, hb, Pmax);
        run<0, 4>("sched0 product, 4-byte B pieces", K, per_cu, dA, dB, dC, dcyc, hA, hb, Pmax);
        run<1, 4>("sched1 read-all, 4-byte B", K, per_cu, dA, dB, dC, dcyc, hA, hb, Pmax);
        run<2, 4>("sched2 3-stage seam, 4-byte B", K, per_cu, dA, dB, dC, dcyc, hA, hb, Pmax);
    }
    return 0;
}

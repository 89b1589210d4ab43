// Developer tool (not part of the product or the tests): times conv_igemm on one layer shape with
// synthetic operands so that kernel experiments do not need the whole attack loop.
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -std=c++17 -x hip tools/conv_microbench.cpp -o /tmp/cmb
//   /tmp/cmb <frames> <Cin> <Cout> <H> <k> [iters]
#include "../image-to-video-i2v-attack_amd/csrc/i2v_kernels.hip"

#include <stdlib.h>
#include <string.h>
#include <vector>

int main(int argc, char** argv) {
    int N = argc > 1 ? atoi(argv[1]) : 128, Cin = argc > 2 ? atoi(argv[2]) : 256, Cout = argc > 3 ? atoi(argv[3]) : 256;
    int H = argc > 4 ? atoi(argv[4]) : 14, k = argc > 5 ? atoi(argv[5]) : 3, iters = argc > 6 ? atoi(argv[6]) : 20;
    int pad = k / 2, K = k * k * Cin, Kpad = (K + I2V_KC - 1) / I2V_KC * I2V_KC, Cdpad = (Cout + 127) / 128 * 128;
    std::vector<float> wp((size_t)Kpad * Cdpad), src((size_t)N * Cin * H * H);
    std::vector<I2VKEntry> kt(Kpad, I2VKEntry{0, 0, 0, 0});
    for (auto& v : wp) v = (rand() % 2001 - 1000) * 1e-4f;
    for (auto& v : src) v = (rand() % 2001 - 1000) * 1e-3f;
    for (int r = 0; r < k; ++r) for (int s = 0; s < k; ++s) for (int c = 0; c < Cin; ++c)
        kt[(r * k + s) * Cin + c] = I2VKEntry{c * H * H, r - pad, s - pad, 1};
    float *dw, *ds, *dd; I2VKEntry* dk;
    hipMalloc(&dw, wp.size() * 4); hipMalloc(&ds, src.size() * 4); hipMalloc(&dd, (size_t)N * Cout * H * H * 4);
    hipMalloc(&dk, kt.size() * sizeof(I2VKEntry));
    hipMemcpy(dw, wp.data(), wp.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(ds, src.data(), src.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(dk, kt.data(), kt.size() * sizeof(I2VKEntry), hipMemcpyHostToDevice);
    I2VConvParams p; memset((void*)&p, 0, sizeof p);
    p.src = ds; p.src_nstride = (int64_t)Cin * H * H; p.Hs = p.Ws = H; p.Cs = Cin;
    p.src_span_bytes = (int32_t)((int64_t)N * Cin * H * H * 4);
    p.wp = dw; p.ktab = dk; p.K = K; p.Kpad = Kpad; p.Cd = Cout; p.Cdpad = Cdpad;
    p.N = N; p.Hg = p.Wg = H; p.sh = p.sw = 1;
    p.dst = dd; p.dst_nstride = (int64_t)Cout * H * H; p.Ho = p.Wo = H; p.osh = p.osw = 1;
    p.add0_stride = 1; p.relu = 1;
    p.pointwise = (k == 1 && (H * H) % 4 == 0); p.tap_uniform = (Cin % I2V_KC == 0);
    p.vec_epilogue = ((H * H) % 4 == 0);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int i = 0; i < 3; ++i) k_conv(p, nullptr);
    hipDeviceSynchronize();
    hipEventRecord(a, nullptr);
    for (int i = 0; i < iters; ++i) k_conv(p, nullptr);
    hipEventRecord(b, nullptr); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b); ms /= iters;
    double fl = 2.0 * N * H * H * (double)Cout * K;
    printf("N=%d Cin=%d Cout=%d H=%d k=%d cfg=%d: %.3f ms  %.1f TFLOP/s  (err=%s)\n", N, Cin, Cout, H, k, conv_pick(p), ms,
           fl / ms * 1e-9, be_error() ? be_error() : "none");
    return 0;
}

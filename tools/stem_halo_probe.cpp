// Developer tool (round 5): conv_stem_halo -- the narrow forward stem (SlowFast's fast pathway: 3 -> 8 channels, 5x7x7, spatial stride 2,
// frame pairs) on a 2-D halo tile -- against the conv_tile launch it replaces (MODE 4 "quad rows" on the 16x256 tile), on the packing
// pack_fwd builds (K order (channel, frame tap, row tap, column quad x 4), rows (frame class, channel)).
//   * correctness: every output word bit for bit against the conv_tile launch on the same parameter block;
//   * speed: microseconds per launch and TFLOP/s of ALGORITHMIC flops for both.
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -std=c++20 -x hip -DI2V_NO_CONV_DISPATCH tools/stem_halo_probe.cpp -o tools/sth_probe
//   tools/sth_probe [clips] [T] [H] [W] [cout] [iters]        (T input frames per clip; the stem reads every 2nd: stride_t = dil_t = 2, pad_t = 4)
//   tools/sth_probe [frames] 1 [H] [W] 64 [iters]             the WIDE image stem (conv_stem64_halo)
#ifndef I2V_NO_CONV_DISPATCH
#define I2V_NO_CONV_DISPATCH
#endif
#include "../image-to-video-i2v-attack_amd/csrc/i2v_kernels.hip"

#include <string.h>

#include <algorithm>
#include <vector>

// The wide image stem (3 -> 64, 7x7 / 2, K = (tap, channel)): conv_stem64_halo against conv_tile's MODE 0 on the 64x64 tile with the dense
// epilogue (shift, ReLU, 1-bit gates): values AND gate words bit for bit.
static int wide(int N, int H, int W, int iters) {
    const int cin = 3, cout = 64, k = 7, st = 2, pad = 3, Ho = (H + 2 * pad - k) / st + 1, Wo = (W + 2 * pad - k) / st + 1;
    const int K = 147, Kpad = 160, Cdpad = 128;
    std::vector<float> wp((size_t)Kpad * Cdpad, 0.f), x((size_t)N * cin * H * W), shift(cout);
    std::vector<I2VKEntry> kt(Kpad, I2VKEntry{0, 0, 0, 0});
    for (auto& v : x) v = (rand() % 2001 - 1000) * 1e-3f;
    for (auto& v : shift) v = (rand() % 2001 - 1000) * 1e-3f;
    for (int r = 0; r < k; ++r) for (int s = 0; s < k; ++s) for (int ci = 0; ci < cin; ++ci) {
        const int kk = (r * k + s) * cin + ci;
        kt[kk] = I2VKEntry{ci * H * W, r - pad, s - pad, 1};
        for (int co = 0; co < cout; ++co) wp[(size_t)kk * Cdpad + co] = (rand() % 2001 - 1000) * 1e-4f;
    }
    const size_t outn = (size_t)N * cout * Ho * Wo; const int64_t P = (int64_t)N * Ho * Wo; const int gstride = (int)((P + 31) / 32) + 4;
    float *dw, *ds, *d0, *d1, *dsh; I2VKEntry* dk; unsigned *g0, *g1;
    hipMalloc(&dw, wp.size() * 4); hipMalloc(&ds, x.size() * 4); hipMalloc(&d0, outn * 4); hipMalloc(&d1, outn * 4); hipMalloc(&dk, kt.size() * sizeof(I2VKEntry)); hipMalloc(&dsh, cout * 4);
    hipMalloc(&g0, (size_t)cout * gstride * 4); hipMalloc(&g1, (size_t)cout * gstride * 4);
    hipMemcpy(dw, wp.data(), wp.size() * 4, hipMemcpyHostToDevice); hipMemcpy(ds, x.data(), x.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(dk, kt.data(), kt.size() * sizeof(I2VKEntry), hipMemcpyHostToDevice); hipMemcpy(dsh, shift.data(), cout * 4, hipMemcpyHostToDevice);
    hipMemset(d0, 0xff, outn * 4); hipMemset(d1, 0xee, outn * 4); hipMemset(g0, 0, (size_t)cout * gstride * 4); hipMemset(g1, 0, (size_t)cout * gstride * 4);
    I2VConvParams p; memset((void*)&p, 0, sizeof p);
    p.src = ds; p.src_nstride = (int64_t)cin * H * W; p.Hs = H; p.Ws = W; p.Cs = cin; p.src_span_bytes = (int32_t)((int64_t)N * cin * H * W * 4);
    p.wp = dw; p.ktab = dk; p.K = K; p.Kpad = Kpad; p.Cd = cout; p.Cdpad = Cdpad; p.halo = 49;
    p.N = N; p.Hg = Ho; p.Wg = Wo; p.sh = p.sw = st; p.dst_nstride = (int64_t)cout * Ho * Wo; p.Ho = Ho; p.Wo = Wo; p.osh = p.osw = 1; p.blk = 1; p.blkt = 1;
    p.Tg = p.Ts = p.To = p.st = p.ost = 1; p.oct = 1; p.add0_stride = 1; p.shift = dsh; p.relu = 1; p.gate_out_stride = gstride;
    p.vec_epilogue = ((Ho * Wo) % 4 == 0) ? 1 : 0;
    conv_magics(p);
    p.gate_out = g0;
    printf("wide stem 7x7/2, 3 -> 64, %d frames of %d x %d: output %d x %d, eligible %d\n", N, H, W, Ho, Wo, (int)conv_stem64_ok(p));
    auto old_launch = [&](float* dst, unsigned* g) { I2VConvParams q = p; q.dst = dst; q.gate_out = g; q.cfg = 4;
        hipLaunchKernelGGL((conv_igemm<64, 64, 2, 2, 0, false>), dim3((unsigned)((P + 63) / 64)), dim3(256), 0, 0, q, 1); };
    auto new_launch = [&](float* dst, unsigned* g) { I2VConvParams q = p; q.dst = dst; q.gate_out = g; return launch_conv_stem64(q, 0); };
    old_launch(d0, g0);
    if (!conv_stem64_ok(p)) { printf("not eligible\n"); return 1; }
    if (new_launch(d1, g1)) { printf("launch failed: %s\n", be_error()); return 1; }
    if (hipDeviceSynchronize() != hipSuccess) { printf("device error\n"); return 1; }
    std::vector<uint32_t> a(outn), b(outn), ga((size_t)cout * gstride), gb((size_t)cout * gstride);
    hipMemcpy(a.data(), d0, outn * 4, hipMemcpyDeviceToHost); hipMemcpy(b.data(), d1, outn * 4, hipMemcpyDeviceToHost);
    hipMemcpy(ga.data(), g0, ga.size() * 4, hipMemcpyDeviceToHost); hipMemcpy(gb.data(), g1, gb.size() * 4, hipMemcpyDeviceToHost);
    size_t bad = 0, first = 0, gbad = 0;
    for (size_t i = 0; i < outn; ++i) if (a[i] != b[i]) { if (!bad) first = i; ++bad; }
    for (size_t i = 0; i < ga.size(); ++i) if (ga[i] != gb[i]) ++gbad;
    if (getenv("STH_CPU") && outn <= (1u << 22)) {      // scalar restatement on the host: which of the two launches is off
        size_t wa = 0, wb = 0;
        for (int n = 0; n < N; ++n) for (int co = 0; co < cout; ++co) for (int oy = 0; oy < Ho; ++oy) for (int ox = 0; ox < Wo; ++ox) {
            float acc = 0.f;
            for (int kk = 0; kk < K; ++kk) {
                const int tap = kk / cin, ci = kk % cin, r = tap / k, sx = tap % k, iy = oy * st + r - pad, ix = ox * st + sx - pad;
                const float xv = (iy >= 0 && iy < H && ix >= 0 && ix < W) ? x[((size_t)n * cin + ci) * H * W + (size_t)iy * W + ix] : 0.f;
                acc = fmaf(wp[(size_t)kk * Cdpad + co], xv, acc);
            }
            float v = acc + shift[co]; v = v > 0.f ? v : 0.f;
            uint32_t u; memcpy(&u, &v, 4);
            const size_t i = (((size_t)n * cout + co) * Ho + oy) * Wo + ox;
            wa += a[i] != u; wb += b[i] != u;
        }
        printf("against the host chain: conv_tile %zu, conv_stem64_halo %zu words differ\n", wa, wb);
    }
    if (bad && getenv("STH_PATTERN")) {      // where the differing words sit: by channel, by row / column inside the 16 x 16 tile
        std::vector<size_t> bc(cout, 0), br(16, 0), bx(16, 0);
        for (size_t i = 0; i < outn; ++i) if (a[i] != b[i]) { const size_t px = i % ((size_t)Ho * Wo); bc[(i / ((size_t)Ho * Wo)) % cout]++; br[(px / Wo) % 16]++; bx[(px % Wo) % 16]++; }
        printf("by channel:"); for (int c = 0; c < cout; ++c) printf(" %zu", bc[c]); printf("\nby tile row:"); for (int r = 0; r < 16; ++r) printf(" %zu", br[r]);
        printf("\nby tile column:"); for (int r = 0; r < 16; ++r) printf(" %zu", bx[r]); printf("\n");
    }
    printf("bitwise: %zu of %zu words differ, %zu of %zu gate words differ", bad, outn, gbad, ga.size());
    if (bad) { float u, v; memcpy(&u, &a[first], 4); memcpy(&v, &b[first], 4); printf(" (first at %zu: %g vs %g)", first, u, v); }
    printf("\n");
    const double flop = 2.0 * N * Ho * Wo * cout * cin * k * k;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int which = 0; which < 2; ++which) {
        for (int i = 0; i < 5; ++i) which ? (void)new_launch(d1, g1) : old_launch(d0, g0);
        hipEventRecord(e0, 0);
        for (int i = 0; i < iters; ++i) which ? (void)new_launch(d1, g1) : old_launch(d0, g0);
        hipEventRecord(e1, 0); hipEventSynchronize(e1);
        float ms = 0.f; hipEventElapsedTime(&ms, e0, e1);
        printf("%s: %.1f us per launch, %.1f TFLOP/s algorithmic\n", which ? "conv_stem64_halo" : "conv_tile MODE 0", ms / iters * 1e3, flop / (ms / iters * 1e-3) * 1e-12);
    }
    return (bad || gbad) ? 2 : 0;
}

int main(int argc, char** argv) {
    if (argc > 5 && atoi(argv[5]) == 64) return wide(atoi(argv[1]), atoi(argv[3]), atoi(argv[4]), argc > 6 ? atoi(argv[6]) : 20);
    const int N = argc > 1 ? atoi(argv[1]) : 8, T = argc > 2 ? atoi(argv[2]) : 32, H = argc > 3 ? atoi(argv[3]) : 224, W = argc > 4 ? atoi(argv[4]) : 224;
    const int cout = argc > 5 ? atoi(argv[5]) : 8, iters = argc > 6 ? atoi(argv[6]) : 20;
    const int cin = 3, kt = 5, kh = 7, kw = 7, st = 2, pad = 3, stt = 2, dilt = 2, padt = 4, kwq = 2;
    const int To = (T + 2 * padt - dilt * (kt - 1) - 1) / stt + 1, Ho = (H + 2 * pad - kh) / st + 1, Wo = (W + 2 * pad - kw) / st + 1;
    std::vector<int> taps;
    for (int ct = 0; ct < 2; ++ct) for (int q = 0; q < kt; ++q) { const int u = ct * stt + q * dilt - padt; if (std::find(taps.begin(), taps.end(), u) == taps.end()) taps.push_back(u); }
    std::sort(taps.begin(), taps.end());
    const int NU = (int)taps.size(), K = cin * NU * kh * kwq * 4, Cd = 2 * cout, Cdpad = 128;
    std::vector<float> wq((size_t)K * Cdpad, 0.f), w((size_t)cout * cin * kt * kh * kw), x((size_t)N * T * cin * H * W), shift(cout);
    std::vector<I2VKEntry> kq(K);
    for (auto& v : w) v = (rand() % 2001 - 1000) * 1e-4f;
    for (auto& v : x) v = (rand() % 2001 - 1000) * 1e-3f;
    for (auto& v : shift) v = (rand() % 2001 - 1000) * 1e-3f;
    for (int ci = 0; ci < cin; ++ci) for (int ui = 0; ui < NU; ++ui) for (int r = 0; r < kh; ++r) for (int s4 = 0; s4 < kwq * 4; ++s4) {
        const int k = (((ci * NU + ui) * kh + r) * kwq) * 4 + s4;
        kq[k] = I2VKEntry{ci * H * W, r - pad, s4 - pad, (s4 < kw ? 1 : 0) + 2 * taps[ui]};
        if (s4 >= kw) continue;
        for (int ct = 0; ct < 2; ++ct) {
            const int num = taps[ui] - ct * stt + padt;
            if (num < 0 || num % dilt || num / dilt >= kt) continue;
            const int q = num / dilt;
            for (int co = 0; co < cout; ++co) wq[(size_t)k * Cdpad + ct * cout + co] = w[((((size_t)co * cin + ci) * kt + q) * kh + r) * kw + s4];
        }
    }
    float *dw, *ds, *d0, *d1, *dsh; I2VKEntry* dk;
    const size_t outn = (size_t)N * To * cout * Ho * Wo;
    hipMalloc(&dw, wq.size() * 4); hipMalloc(&ds, x.size() * 4 + 1024); ds += 64; hipMalloc(&d0, outn * 4); hipMalloc(&d1, outn * 4); hipMalloc(&dk, kq.size() * sizeof(I2VKEntry)); hipMalloc(&dsh, cout * 4);
    hipMemcpy(dw, wq.data(), wq.size() * 4, hipMemcpyHostToDevice); hipMemcpy(ds, x.data(), x.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(dk, kq.data(), kq.size() * sizeof(I2VKEntry), hipMemcpyHostToDevice); hipMemcpy(dsh, shift.data(), cout * 4, hipMemcpyHostToDevice);
    hipMemset(d0, 0xff, outn * 4); hipMemset(d1, 0xee, outn * 4);
    I2VConvParams p; memset((void*)&p, 0, sizeof p);
    p.src = ds; p.src_nstride = (int64_t)cin * H * W; p.Hs = H; p.Ws = W; p.Cs = cin; p.src_span_bytes = (int32_t)((int64_t)N * T * cin * H * W * 4);
    p.wp = dw; p.ktab = dk; p.K = p.Kpad = K; p.Cd = Cd; p.Cdpad = Cdpad; p.quad = kwq; p.quad_kw = kw; p.quad_dw0 = -pad;
    p.Tg = (To + 1) / 2; p.N = N * p.Tg; p.Hg = Ho; p.Wg = Wo; p.sh = p.sw = st;
    p.dst_nstride = (int64_t)cout * Ho * Wo; p.Ho = Ho; p.Wo = Wo; p.osh = p.osw = 1; p.blk = 1; p.blkt = 2;
    p.Ts = T; p.To = To; p.st = 2 * stt; p.ost = 2; p.ot0 = 0; p.oct = 1; p.add0_stride = 1; p.temporal = 1;
    p.shift = dsh; p.relu = 1;
    conv_magics(p);
    printf("fast stem %dx%dx%d, 3 -> %d, %d clips of %d frames of %d x %d: %d output frames of %d x %d, %d class rows, K %d (%d frame taps), eligible %d\n", kt, kh, kw, cout, N, T, H, W, To, Ho,
           Wo, Cd, K, NU, (int)conv_stemhalo_ok(p));
    const int64_t P = (int64_t)p.N * Ho * Wo;
    auto old_launch = [&](float* dst) { I2VConvParams q = p; q.dst = dst; q.cfg = 6;
        hipLaunchKernelGGL((conv_igemm<16, 256, 1, 4, 4, false, false, true, true>), dim3((unsigned)((P + 255) / 256)), dim3(256), 0, 0, q, 1); };
    auto new_launch = [&](float* dst) { I2VConvParams q = p; q.dst = dst; return launch_conv_stemhalo(q, 0); };
    old_launch(d0);
    if (!conv_stemhalo_ok(p)) { printf("not eligible\n"); return 1; }
    if (new_launch(d1)) { printf("launch failed: %s\n", be_error()); return 1; }
    if (hipDeviceSynchronize() != hipSuccess) { printf("device error\n"); return 1; }
    std::vector<uint32_t> a(outn), b(outn);
    hipMemcpy(a.data(), d0, outn * 4, hipMemcpyDeviceToHost); hipMemcpy(b.data(), d1, outn * 4, hipMemcpyDeviceToHost);
    size_t bad = 0, first = 0;
    for (size_t i = 0; i < outn; ++i) if (a[i] != b[i]) { if (!bad) first = i; ++bad; }
    printf("bitwise: %zu of %zu words differ", bad, outn);
    if (bad) { float u, v; memcpy(&u, &a[first], 4); memcpy(&v, &b[first], 4); printf(" (first at %zu: %g vs %g)", first, u, v); }
    printf("\n");
    const double flop = 2.0 * N * To * Ho * Wo * cout * cin * kt * kh * kw;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int which = 0; which < 2; ++which) {
        for (int i = 0; i < 5; ++i) which ? (void)new_launch(d1) : old_launch(d0);
        hipEventRecord(e0, 0);
        for (int i = 0; i < iters; ++i) which ? (void)new_launch(d1) : old_launch(d0);
        hipEventRecord(e1, 0); hipEventSynchronize(e1);
        float ms = 0.f; hipEventElapsedTime(&ms, e0, e1);
        printf("%s: %.1f us per launch, %.1f TFLOP/s algorithmic\n", which ? "conv_stem_halo  " : "conv_tile MODE 4", ms / iters * 1e3, flop / (ms / iters * 1e-3) * 1e-12);
    }
    return bad ? 2 : 0;
}

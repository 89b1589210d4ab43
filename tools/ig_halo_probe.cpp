// Developer tool (round 5): conv_imggrad_halo -- the class-packed image gradient on a 2-D halo tile -- against the conv_tile launch
// it replaces (conv_igemm<16,256,1,4,2,...,MF16>), on the packing pack_img builds for a stride-s stem (tap-uniform K order).
//   * correctness: every output word bit for bit against the conv_tile launch on the same parameter block;
//   * speed: microseconds per launch and TFLOP/s of ALGORITHMIC flops (2 * Hs*Ws * cout * 3 * k*k per frame) for both.
// Built WITHOUT the product's dispatch (only the kernels launched here are instantiated):
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -std=c++20 -x hip -DI2V_NO_CONV_DISPATCH tools/ig_halo_probe.cpp -o tools/igh_probe
//   tools/igh_probe [frames] [H] [cout] [k] [stride] [pad] [iters] [kt] [stride_t] [pad_t] [T]
//   (kt > 1: a video stem -- `frames` counts clips of T frames, temporal classes in front of the spatial ones: the I3D's is 8 224 64 7 2 3 10 5 2 2 32)
#ifndef I2V_NO_CONV_DISPATCH
#define I2V_NO_CONV_DISPATCH
#endif
#include "../image-to-video-i2v-attack_amd/csrc/i2v_kernels.hip"

#include <string.h>

#include <vector>

static int posmod(int a, int b) { int r = a % b; return r < 0 ? r + b : r; }
static int floordiv(int a, int b) { return (a - posmod(a, b)) / b; }

int main(int argc, char** argv) {
    const int N = argc > 1 ? atoi(argv[1]) : 128, H = argc > 2 ? atoi(argv[2]) : 224, cout = argc > 3 ? atoi(argv[3]) : 64;
    const int k = argc > 4 ? atoi(argv[4]) : 7, st = argc > 5 ? atoi(argv[5]) : 2, pad = argc > 6 ? atoi(argv[6]) : 3, iters = argc > 7 ? atoi(argv[7]) : 20;
    const int kt = argc > 8 ? atoi(argv[8]) : 1, stt = argc > 9 ? atoi(argv[9]) : 1, padt = argc > 10 ? atoi(argv[10]) : 0, T = argc > 11 ? atoi(argv[11]) : 1;
    const int Ts = (T + 2 * padt - kt) / stt + 1, Bt = stt, Tg = (T + Bt - 1) / Bt;      // dy frames per clip, temporal classes, grid frames per clip
    int tlo = 1 << 30, thi = -(1 << 30);
    for (int ct = 0; ct < Bt; ++ct) for (int q = 0; q < kt; ++q) if (posmod(ct + padt - q, stt) == 0) { const int d = floordiv(ct + padt - q, stt); tlo = d < tlo ? d : tlo; thi = d > thi ? d : thi; }
    const int TT = thi - tlo + 1;
    const int cin = 3, Hs = (H + 2 * pad - k) / st + 1, B = st == 1 ? 2 : st, m = B / st, Hg = (H + B - 1) / B;
    int dlo = 1 << 30, dhi = -(1 << 30);
    for (int ph = 0; ph < B; ++ph) for (int r = 0; r < k; ++r) if (posmod(ph + pad - r, st) == 0) { const int d = floordiv(ph + pad - r, st); dlo = d < dlo ? d : dlo; dhi = d > dhi ? d : dhi; }
    const int TH = dhi - dlo + 1, TW = TH, NT = TT * TH * TW;
    const int K = NT * cout, Cd = Bt * B * B * cin, Cdpad = 128;
    const bool quad = cout % 16 != 0;         // fewer than 16 channels: the quad-row order (channel, frame tap, row tap, column tap x 4)
    if (quad && (TH != 4 || TW != 4)) { printf("quad-row order: 4 x 4 union taps only\n"); return 1; }
    std::vector<float> wp((size_t)K * Cdpad, 0.f), w((size_t)cout * cin * kt * k * k), dy((size_t)N * Ts * cout * Hs * Hs);
    std::vector<I2VKEntry> ktab(K);
    for (auto& v : w) v = (rand() % 2001 - 1000) * 1e-4f;
    for (auto& v : dy) v = (rand() % 2001 - 1000) * 1e-3f;
    auto krow = [&](int tt, int th, int tw, int co) { return quad ? ((co * TT + tt) * TH + th) * TW + tw : ((co / 16) * NT + (tt * TH + th) * TW + tw) * 16 + co % 16; };
    for (int tt = 0; tt < TT; ++tt) for (int th = 0; th < TH; ++th) for (int tw = 0; tw < TW; ++tw) for (int co = 0; co < cout; ++co)
        ktab[krow(tt, th, tw, co)] = I2VKEntry{co * Hs * Hs, th + dlo, tw + dlo, 1 + 2 * (tt + tlo)};
    for (int ct = 0; ct < Bt; ++ct) for (int q = 0; q < kt; ++q) {
        if (posmod(ct + padt - q, stt)) continue;
        const int tt = floordiv(ct + padt - q, stt) - tlo;
        for (int ph = 0; ph < B; ++ph) for (int pw = 0; pw < B; ++pw) for (int r = 0; r < k; ++r) {
            if (posmod(ph + pad - r, st)) continue;
            const int th = floordiv(ph + pad - r, st) - dlo;
            for (int s = 0; s < k; ++s) {
                if (posmod(pw + pad - s, st)) continue;
                const int tw = floordiv(pw + pad - s, st) - dlo;
                for (int co = 0; co < cout; ++co) for (int ci = 0; ci < cin; ++ci)
                    wp[(size_t)krow(tt, th, tw, co) * Cdpad + ((ct * B + ph) * B + pw) * cin + ci] = w[((((size_t)co * cin + ci) * kt + q) * k + r) * k + s];
            }
        }
    }
    float *dw, *ds, *d0, *d1; I2VKEntry* dk;
    const size_t outn = (size_t)N * T * cin * H * H;
    hipMalloc(&dw, wp.size() * 4); hipMalloc(&ds, dy.size() * 4 + 1024); ds += 64;      /* (slack around the source: conv_tile's quad-row staging reads 64 bytes either side) */ hipMalloc(&d0, outn * 4); hipMalloc(&d1, outn * 4); hipMalloc(&dk, ktab.size() * sizeof(I2VKEntry));
    hipMemcpy(dw, wp.data(), wp.size() * 4, hipMemcpyHostToDevice); hipMemcpy(ds, dy.data(), dy.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(dk, ktab.data(), ktab.size() * sizeof(I2VKEntry), hipMemcpyHostToDevice);
    hipMemset(d0, 0xff, outn * 4); hipMemset(d1, 0xee, outn * 4);
    I2VConvParams p; memset((void*)&p, 0, sizeof p);
    p.src = ds; p.src_nstride = (int64_t)cout * Hs * Hs; p.Hs = p.Ws = Hs; p.Cs = cout; p.src_span_bytes = (int32_t)((int64_t)N * Ts * cout * Hs * Hs * 4);
    p.wp = dw; p.ktab = dk; p.K = p.Kpad = K; p.Cd = Cd; p.Cdpad = Cdpad; p.tap_uniform = quad ? 0 : 1;
    if (quad) { p.quad = 1; p.quad_kw = TW; p.quad_dw0 = dlo; }
    p.N = N * Tg; p.Hg = p.Wg = Hg; p.sh = p.sw = m;
    p.dst_nstride = (int64_t)cin * H * H; p.Ho = p.Wo = H; p.osh = p.osw = B; p.blk = B; p.blkt = Bt;
    p.Tg = Tg; p.Ts = Ts; p.To = T; p.st = 1; p.ost = Bt; p.ot0 = 0; p.oct = 1; p.add0_stride = 1;
    p.temporal = (kt > 1 || stt > 1 || quad) ? 1 : 0;
    p.ig_tt = TT; p.ig_th = TH; p.ig_tw = TW;
    conv_magics(p);
    printf("stem %dx%dx%d/(%d,%d) pad (%d,%d), %d -> 3, %d x %d frames of %d^2: class grid %d x %d^2, Cd %d, K %d (%d x %d x %d union taps), eligible %d\n", kt, k, k, stt, st,
           padt, pad, cout, N, T, H, Tg, Hg, Cd, K, TT, TH, TW, (int)conv_ighalo_ok(p));
    const int64_t P = (int64_t)N * Tg * Hg * Hg;
    auto old_launch = [&](float* dst) { I2VConvParams q = p; q.dst = dst; q.cfg = 6;
        if (quad && Cd <= 16) hipLaunchKernelGGL((conv_igemm<16, 256, 1, 4, 4, false, false, true, true>), dim3((unsigned)((P + 255) / 256)), dim3(256), 0, 0, q, 1);
        else if (quad) hipLaunchKernelGGL((conv_igemm<32, 256, 1, 4, 4, false, false, true, false>), dim3((unsigned)((P + 255) / 256)), dim3(256), 0, 0, q, 1);
        else if (Cd <= 16 && !p.temporal) hipLaunchKernelGGL((conv_igemm<16, 256, 1, 4, 2, false, false, false, true>), dim3((unsigned)((P + 255) / 256)), dim3(256), 0, 0, q, 1);
        else if (Cd <= 16) hipLaunchKernelGGL((conv_igemm<16, 256, 1, 4, 2, false, false, true, true>), dim3((unsigned)((P + 255) / 256)), dim3(256), 0, 0, q, 1);
        else if (!p.temporal) hipLaunchKernelGGL((conv_igemm<32, 256, 1, 4, 2, false>), dim3((unsigned)((P + 255) / 256)), dim3(256), 0, 0, q, 1);
        else hipLaunchKernelGGL((conv_igemm<32, 256, 1, 4, 2, false, false, true>), dim3((unsigned)((P + 255) / 256)), dim3(256), 0, 0, q, 1); };
    auto new_launch = [&](float* dst) { I2VConvParams q = p; q.dst = dst; return launch_conv_ighalo(q, 0); };
    old_launch(d0);
    if (!conv_ighalo_ok(p)) { printf("not eligible\n"); return 1; }
    if (new_launch(d1)) { printf("launch failed: %s\n", be_error()); return 1; }
    if (hipDeviceSynchronize() != hipSuccess) { printf("device error\n"); return 1; }
    std::vector<uint32_t> a(outn), b(outn);
    hipMemcpy(a.data(), d0, outn * 4, hipMemcpyDeviceToHost); hipMemcpy(b.data(), d1, outn * 4, hipMemcpyDeviceToHost);
    size_t bad = 0, first = 0;
    for (size_t i = 0; i < outn; ++i) if (a[i] != b[i]) { if (!bad) first = i; ++bad; }
    printf("bitwise: %zu of %zu words differ", bad, outn);
    if (bad) { float x, y; memcpy(&x, &a[first], 4); memcpy(&y, &b[first], 4); printf(" (first at %zu: %g vs %g)", first, x, y); }
    printf("\n");
    const double flop = 2.0 * N * Ts * Hs * Hs * cout * cin * kt * k * k;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int which = 0; which < 2; ++which) {
        for (int i = 0; i < 5; ++i) which ? (void)new_launch(d1) : old_launch(d0);
        hipEventRecord(e0, 0);
        for (int i = 0; i < iters; ++i) which ? (void)new_launch(d1) : old_launch(d0);
        hipEventRecord(e1, 0); hipEventSynchronize(e1);
        float ms = 0.f; hipEventElapsedTime(&ms, e0, e1);
        printf("%s: %.1f us per launch, %.1f TFLOP/s algorithmic\n", which ? "conv_imggrad_halo" : "conv_tile 16x256 ", ms / iters * 1e3, flop / (ms / iters * 1e-3) * 1e-12);
    }
    return bad ? 2 : 0;
}

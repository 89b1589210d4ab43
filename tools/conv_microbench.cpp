// Developer tool (not part of the product or the tests): times conv_igemm on the ResNet-50 layer shapes with
// synthetic operands (random data, chunk-major K order as the planner packs it) so that kernel experiments do not
// need the whole attack loop.  Built against whichever kernel source KSRC names, so two builds (old / new kernel)
// can be run back to back on the same device:
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -std=c++20 -x hip -DKSRC='"../image-to-video-i2v-attack_amd/csrc/i2v_kernels.hip"' tools/conv_microbench.cpp -o tools/cmb
//   tools/cmb [frames] [iters]            all shapes x all valid tile configurations
//   tools/cmb <frames> <Cin> <Cout> <H> <k> [iters]     one shape (I2V_FORCE_CFG picks the configuration)
// -DCMB_PROBE: per-block time stamps through the kernel source's probe hook (I2V_PROBE_T): 8 words per block {K-loop shader cycles,
// K-loop 100 MHz ticks, entry, loop start, loop end, exit (100 MHz wall clock), HW_ID, XCC_ID}; the timeline of the last launch is
// printed with every measurement (when the blocks of a CU start and finish, blocks per CU, in-kernel clock).
#ifndef KSRC
#define KSRC "../image-to-video-i2v-attack_amd/csrc/i2v_kernels.hip"
#endif
#ifdef CMB_PROBE
#include <hip/hip_runtime.h>
__device__ unsigned long long g_xclk[1 << 20];
struct CmbProbe {
    unsigned long long e, t0, r0;
    __device__ __forceinline__ void entry() { e = __builtin_amdgcn_s_memrealtime(); }
    __device__ __forceinline__ void loop_begin() { t0 = __builtin_amdgcn_s_memtime(); r0 = __builtin_amdgcn_s_memrealtime(); }
    __device__ __forceinline__ void loop_end(int slot) {
        if (threadIdx.x == 0 && slot < (1 << 17)) {
            const unsigned long long r1 = __builtin_amdgcn_s_memrealtime();
            g_xclk[8 * slot] = __builtin_amdgcn_s_memtime() - t0; g_xclk[8 * slot + 1] = r1 - r0; g_xclk[8 * slot + 3] = r0; g_xclk[8 * slot + 4] = r1;
        }
    }
    __device__ __forceinline__ void exit(int slot) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (threadIdx.x == 0 && slot < (1 << 17)) {
            g_xclk[8 * slot + 2] = e; g_xclk[8 * slot + 5] = __builtin_amdgcn_s_memrealtime();
            g_xclk[8 * slot + 6] = __builtin_amdgcn_s_getreg((32 - 1) << 11 | 0 << 6 | 4);       // HW_REG_HW_ID
            g_xclk[8 * slot + 7] = __builtin_amdgcn_s_getreg((4 - 1) << 11 | 0 << 6 | 20);      // HW_REG_XCC_ID
        }
    }
};
#define I2V_PROBE_T CmbProbe
#endif
#include KSRC

#include <stdlib.h>
#include <string.h>
#include <vector>
#include <algorithm>
#include <map>

struct Shape { int Cin, Cout, H, k; const char* what; int stride = 1; };

static double run(int N, const Shape& sh, int cfg, int iters, int with_epi) {
    const int Cin = sh.Cin, Cout = sh.Cout, H = sh.H, k = sh.k, st = sh.stride, Ho = (H + 2 * (k / 2) - k) / st + 1;
    int pad = k / 2; int K = k * k * Cin, Kpad = (K + I2V_KC - 1) / I2V_KC * I2V_KC, Cdpad = (Cout + 127) / 128 * 128;
    std::vector<float> wp((size_t)Kpad * Cdpad), src((size_t)N * Cin * H * H);
    std::vector<I2VKEntry> kt(Kpad, I2VKEntry{0, 0, 0, 0});
    for (auto& v : wp) v = (rand() % 2001 - 1000) * 1e-4f;
    for (auto& v : src) v = (rand() % 2001 - 1000) * 1e-3f;
    if (const char* dm = getenv("CMB_DATA")) {      // relu: post-ReLU-like activations (half zeros); zero: all zeros
        if (!strcmp(dm, "relu")) for (auto& v : src) v = v > 0.f ? v : 0.f;
        if (!strcmp(dm, "zero")) for (auto& v : src) v = 0.f;
    }
    const bool tu = Cin % I2V_KC == 0;
    const int NT = k * k;
    const bool quad = Cin < 16 && k >= 2 && k <= 8 && !getenv("CMB_NOQUAD");     // stems: rows (c, r, s-quad x 4)
    const int kwq = (k + 3) / 4;
    if (quad) {
        K = Cin * k * kwq * 4; Kpad = (K + I2V_KC - 1) / I2V_KC * I2V_KC;
        wp.assign((size_t)Kpad * Cdpad, 0.f); kt.assign(Kpad, I2VKEntry{0, 0, 0, 0});
        for (int c = 0; c < Cin; ++c) for (int r = 0; r < k; ++r) for (int sq = 0; sq < kwq; ++sq) for (int e = 0; e < 4; ++e) {
            const int s2 = sq * 4 + e, kk = ((c * k + r) * kwq + sq) * 4 + e;
            kt[kk] = I2VKEntry{c * H * H, r - pad, s2 - pad, s2 < k ? 1 : 0};
            if (s2 < k) for (int co = 0; co < Cout; ++co) wp[(size_t)kk * Cdpad + co] = (rand() % 2001 - 1000) * 1e-4f;
        }
    } else
    for (int r = 0; r < k; ++r) for (int s = 0; s < k; ++s) for (int c = 0; c < Cin; ++c) {
        const int tap = r * k + s;
        const int kk = tu ? ((c / I2V_KC) * NT + tap) * I2V_KC + c % I2V_KC : tap * Cin + c;
        kt[kk] = I2VKEntry{c * H * H, r - pad, s - pad, 1};
    }
    float *dw, *ds, *dd, *da; I2VKEntry* dk;
    const size_t outn = (size_t)N * Cout * Ho * Ho;
    hipMalloc(&dw, wp.size() * 4); hipMalloc(&ds, src.size() * 4 + 1024); ds += 64;      /* slack around the source for the quad-row staging */ hipMalloc(&dd, outn * 4); hipMalloc(&da, outn * 4);
    hipMalloc(&dk, kt.size() * sizeof(I2VKEntry));
    hipMemcpy(dw, wp.data(), wp.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(ds, src.data(), src.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(dk, kt.data(), kt.size() * sizeof(I2VKEntry), hipMemcpyHostToDevice);
    hipMemset(da, 0, outn * 4);
    I2VConvParams p; memset((void*)&p, 0, sizeof p);
    p.src = ds; p.src_nstride = (int64_t)Cin * H * H; p.Hs = p.Ws = H; p.Cs = Cin;
    p.src_span_bytes = (int32_t)((int64_t)N * Cin * H * H * 4);
    p.wp = dw; p.ktab = dk; p.K = K; p.Kpad = Kpad; p.Cd = Cout; p.Cdpad = Cdpad;
    if (quad) { p.quad = kwq; p.quad_kw = k; p.quad_dw0 = -pad; }
    p.N = N; p.Hg = p.Wg = Ho; p.sh = p.sw = st;
    p.dst = dd; p.dst_nstride = (int64_t)Cout * Ho * Ho; p.Ho = p.Wo = Ho; p.osh = p.osw = 1;
    p.add0_stride = 1; p.relu = 1;
    p.blkt = 1; p.Tg = p.Ts = p.To = p.st = p.ost = 1; p.oct = 1;
    if (with_epi) { p.add0 = da; p.add0_nstride = p.dst_nstride; }      // residual-style addend (the expand convolutions)
    p.tap_uniform = tu;
    p.pointwise = (k == 1 && st == 1 && (H * H) % 4 == 0 && !getenv("CMB_NOPW")); p.tap_uniform = tu;
    p.vec_epilogue = ((Ho * Ho) % 4 == 0);
    p.cfg = cfg + 1;
    void* dw3 = nullptr;
    if (getenv("CMB_BF3") && !quad) {      // split-bf16 arithmetic: three bf16 terms per weight in the 32x32x16 MFMA's fragment order
        auto bf = [](float x) { unsigned u; memcpy(&u, &x, 4); u = (u + 0x7fffu + ((u >> 16) & 1u)) & 0xffff0000u; float y; memcpy(&y, &u, 4); return y; };
        std::vector<uint16_t> w3((size_t)(Kpad / 16) * (Cdpad / 32) * 3 * 64 * 8);
        for (int c = 0; c < Kpad / 16; ++c) for (int t = 0; t < Cdpad / 32; ++t) for (int l = 0; l < 64; ++l) for (int j = 0; j < 8; ++j) {
            float w = wp[(size_t)(16 * c + 8 * (l >> 5) + j) * Cdpad + 32 * t + (l & 31)];
            for (int term = 0; term < 3; ++term) {
                const float b = bf(w); unsigned u; memcpy(&u, &b, 4);
                w3[((((size_t)c * (Cdpad / 32) + t) * 3 + term) * 64 + l) * 8 + j] = (uint16_t)(u >> 16);
                w -= b;
            }
        }
        hipMalloc(&dw3, w3.size() * 2); hipMemcpy(dw3, w3.data(), w3.size() * 2, hipMemcpyHostToDevice);
        p.wp3 = dw3; p.bf3 = 1;
    }
    if (tu && k == 3 && st == 1) p.halo = 9;      // the K order above is (16-channel group, tap, channel): I2V_FORCE_CFG=19 (3 | 16) runs MODE 5
    if (getenv("CMB_CHECK") && p.bf3) {    // split-bf16 result against the fp32-MFMA result of the same launch: max |diff| / max |value|
        std::vector<float> r3(outn), r1(outn);
        k_conv(p, nullptr); hipDeviceSynchronize(); hipMemcpy(r3.data(), dd, outn * 4, hipMemcpyDeviceToHost);
        I2VConvParams q = p; q.bf3 = 0; k_conv(q, nullptr); hipDeviceSynchronize(); hipMemcpy(r1.data(), dd, outn * 4, hipMemcpyDeviceToHost);
        double md = 0, mv = 0, sd = 0; for (size_t i = 0; i < outn; ++i) { md = std::max(md, (double)fabsf(r3[i] - r1[i])); mv = std::max(mv, (double)fabsf(r1[i])); sd += fabs((double)r3[i] - r1[i]); }
        printf("   [split-bf16 vs fp32 MFMA: max|diff| %.3e, mean|diff| %.3e, max|value| %.3e -> relative %.2e]\n", md, sd / outn, mv, md / mv);
    }
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    // steady state: the chip's clock takes milliseconds to settle after an idle period (a 3 ms measurement right after the host
    // prepared the operands read 1.8-1.9 GHz where a long run holds 2.2): >= CMB_WARM_MS (default 80) ms of back-to-back
    // launches first, then a timed region of at least 30 ms
    for (int i = 0; i < 2; ++i) k_conv(p, nullptr);
    hipDeviceSynchronize();
    {
        hipEventRecord(a, nullptr); k_conv(p, nullptr); hipEventRecord(b, nullptr); hipEventSynchronize(b);
        float one; hipEventElapsedTime(&one, a, b); if (one < 1e-3f) one = 1e-3f;
        const char* wm = getenv("CMB_WARM_MS"); const float warm_ms = wm ? (float)atof(wm) : 80.f;
        const int nw = (int)(warm_ms / one) + 1; for (int i = 0; i < nw; ++i) k_conv(p, nullptr);
        if (iters * one < 30.f) iters = (int)(30.f / one) + 1;
    }
    hipEventRecord(a, nullptr);
    for (int i = 0; i < iters; ++i) k_conv(p, nullptr);
    hipEventRecord(b, nullptr); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b); ms /= iters;
#ifdef CMB_PROBE
    {   // diagnostic build: timeline of the LAST launch from per-block stamps (100 MHz wall clock, shader cycles over the K loop)
        const size_t NW = (size_t)8 << 17;
        std::vector<unsigned long long> h(NW);
        hipMemcpyFromSymbol(h.data(), HIP_SYMBOL(g_xclk), NW * 8);
        std::vector<double> clk, cyc, ent, pro, loop, epi, ext, loopq;
        std::map<unsigned, std::vector<int>> per_cu;
        unsigned long long T0 = ~0ull;
        for (int i = 0; i < (1 << 17); ++i) if (h[8 * i + 5]) T0 = std::min(T0, h[8 * i + 2]);
        for (int i = 0; i < (1 << 17); ++i) {
            const unsigned long long* w = &h[8 * i];
            if (!w[5]) continue;
            const bool quarter = i >= 65536;
            if (!quarter) { clk.push_back((double)w[0] / w[1] * 0.1); cyc.push_back((double)w[0]); loop.push_back((w[4] - w[3]) * 0.01); } else loopq.push_back((w[4] - w[3]) * 0.01);
            ent.push_back((w[2] - T0) * 0.01); pro.push_back((w[3] - w[2]) * 0.01); epi.push_back((w[5] - w[4]) * 0.01); ext.push_back((w[5] - T0) * 0.01);
            const unsigned hw = (unsigned)w[6], xcc = (unsigned)w[7] & 15;
            per_cu[(xcc << 12) | (((hw >> 13) & 7) << 8) | (((hw >> 12) & 1) << 4) | ((hw >> 8) & 15)].push_back(i);
        }
        auto st = [](std::vector<double>& v, const char* nm) { if (v.empty()) return; std::sort(v.begin(), v.end()); printf(" %s[min %.1f med %.1f p90 %.1f max %.1f]", nm, v.front(), v[v.size() / 2], v[v.size() * 9 / 10], v.back()); };
        if (!clk.empty()) {
            std::sort(clk.begin(), clk.end()); std::sort(cyc.begin(), cyc.end());
            printf("   [launch %.1f us | clock GHz med %.3f (%.3f..%.3f) | K-loop cycles med %.0f max %.0f = %.1f / %.1f per chunk | us:", ms * 1e3, clk[clk.size() / 2], clk.front(), clk.back(),
                   cyc[cyc.size() / 2], cyc.back(), cyc[cyc.size() / 2] / (Kpad / I2V_KC), cyc.back() / (Kpad / I2V_KC));
            st(ent, "entry"); st(pro, "prologue"); st(loop, "loop"); st(loopq, "loop(quarter)"); st(epi, "epilogue"); st(ext, "exit");
            std::map<int, int> hist; std::map<int, double> exit_by_n; for (auto& kv : per_cu) { hist[(int)kv.second.size()]++; double mx = 0; for (int b : kv.second) mx = std::max(mx, (h[8 * b + 5] - T0) * 0.01); exit_by_n[(int)kv.second.size()] = std::max(exit_by_n[(int)kv.second.size()], mx); }
            printf(" | CUs %zu, blocks/CU:", per_cu.size()); for (auto& kv : hist) printf(" %dx%d(last exit %.1f)", kv.first, kv.second, exit_by_n[kv.first]);
            printf("]\n");
        }
        
        static std::vector<unsigned long long> z(NW, 0); hipMemcpyToSymbol(HIP_SYMBOL(g_xclk), z.data(), NW * 8);
    }
#endif
    if (dw3) hipFree(dw3); hipFree(dw); hipFree(ds - 64); hipFree(dd); hipFree(da); hipFree(dk); hipEventDestroy(a); hipEventDestroy(b);
    if (be_error()) { printf("ERROR %s\n", be_error()); exit(1); }
    return 2.0 * N * Ho * Ho * (double)Cout * (k * k * Cin) / ms * 1e-9;      // algorithmic flops (not the padded quad rows)
}

int main(int argc, char** argv) {
    if (argc > 5) {
        Shape sh{atoi(argv[2]), atoi(argv[3]), atoi(argv[4]), atoi(argv[5]), "cli"};
        const char* f = getenv("I2V_FORCE_CFG");
        printf("%.1f TFLOP/s\n", run(atoi(argv[1]), sh, f ? atoi(f) : 3, argc > 6 ? atoi(argv[6]) : 20, getenv("CMB_EPI") ? 1 : 0));      // CMB_EPI: with a residual addend
        return 0;
    }
    const int N = argc > 1 ? atoi(argv[1]) : 128, iters = argc > 2 ? atoi(argv[2]) : 10;
    static const Shape S[] = {
        {256, 256, 14, 3, "layer3 3x3"}, {128, 128, 28, 3, "layer2 3x3"}, {64, 64, 56, 3, "layer1 3x3"},
        {256, 1024, 14, 1, "layer3 expand"}, {1024, 256, 14, 1, "layer3 reduce"}, {128, 512, 28, 1, "layer2 expand"},
        {512, 128, 28, 1, "layer2 reduce"}, {64, 256, 56, 1, "layer1 expand"}, {256, 64, 56, 1, "layer1 reduce"},
        {512, 1024, 14, 1, "layer3 down(K512)"}, {256, 512, 28, 1, "layer2 down(K256)"}, {64, 64, 56, 1, "layer1 first"},
        {3, 64, 224, 7, "stem 7x7/2", 2}, {3, 64, 224, 3, "vgg first 3x3", 1},
        {15, 8, 224, 7, "fast stem 5x7x7 (as 15 ch)", 2}};
    static const char* CN[5] = {"128x128", "64x128", "128x64", "64x64", "32x256"};
    printf("%-20s %5s %5s %3s %2s |", "shape", "Cin", "Cout", "H", "k");
    for (int c = 0; c < 4; ++c) printf(" %8s", CN[c]);
    printf(" | +addend(64x64)\n");
    for (const Shape& sh : S) {
        printf("%-20s %5d %5d %3d %2d |", sh.what, sh.Cin, sh.Cout, sh.H, sh.k);
        for (int c = 0; c < 4; ++c) {
            const int BD = (c == 0 || c == 2) ? 128 : 64;
            if (sh.Cout <= BD / 2) { printf(" %8s", "-"); continue; }
            printf(" %8.1f", run(N, sh, c, iters, 0));
        }
        if (sh.Cout <= 16) printf(" | cfg5(16x256 MF16) %8.1f\n", run(N, sh, 5, iters, 0));
        else printf(" | %8.1f\n", run(N, sh, 3, iters, 1));
        fflush(stdout);
    }
    return 0;
}

// gfx950 (MI355X / CDNA4) kernel backend of the I2V engine.  Wave64, fp32-input MFMA.
//
//   conv_igemm      implicit-GEMM convolution, forward and input-gradient (I2VConvParams):
//                   D[cd][pixel] = Wp[k][cd]^T * im2col[k][pixel] on v_mfma_f32_32x32x2_f32, with the
//                   pixel axis on the MFMA column (lane) index so that NCHW loads AND stores are
//                   coalesced along W; im2col is formed while staging into LDS.
//                   The gradient w.r.t. the 3-channel image reuses it with the Cd axis packing
//                   (position-class, channel) pairs (I2VConvParams::blk), since GEMM-N = 3 alone is
//                   not an MFMA shape.
//   pool / addmask / cosine / std / compose / Adam / sign-step: HBM-bound streaming kernels.
#include "i2v_be.h"

thread_local char g_be_err[256];
thread_local bool g_be_has_err = false;

int hip_fail(hipError_t e, const char* what) {
    snprintf(g_be_err, sizeof g_be_err, "%s: %s", what, hipGetErrorString(e));
    g_be_has_err = true;
    return 1;
}

const char* be_name() { return "hip:gfx950"; }
long long g_stat_conv = 0, g_stat_pws = 0, g_stat_bf3 = 0, g_stat_igh = 0, g_stat_sth = 0;      // (relaxed counters: diagnostics only)
long long be_stat(const char* name) {
    if (!strcmp(name, "conv_launches")) return __atomic_load_n(&g_stat_conv, __ATOMIC_RELAXED);
    if (!strcmp(name, "pws_launches")) return __atomic_load_n(&g_stat_pws, __ATOMIC_RELAXED);
    if (!strcmp(name, "bf3_launches")) return __atomic_load_n(&g_stat_bf3, __ATOMIC_RELAXED);
    if (!strcmp(name, "ighalo_launches")) return __atomic_load_n(&g_stat_igh, __ATOMIC_RELAXED);
    if (!strcmp(name, "stemhalo_launches")) return __atomic_load_n(&g_stat_sth, __ATOMIC_RELAXED);
    if (!strcmp(name, "fastblock_launches")) return __atomic_load_n(&g_stat_fastblock, __ATOMIC_RELAXED);
    if (!strcmp(name, "vfma_launches")) return __atomic_load_n(&g_stat_vfma, __ATOMIC_RELAXED);
    if (!strcmp(name, "igvfma_launches")) return __atomic_load_n(&g_stat_igv, __ATOMIC_RELAXED);
#ifdef I2V_EXPERIMENTAL      // 1: the library carries the experimental kernels (fused pair, split-bf16 loop, conv_pw_stream, conv_stem64_halo)
    if (!strcmp(name, "experimental")) return 1;
#else
    if (!strcmp(name, "experimental")) return 0;
#endif
    return -1;
}
const char* be_error() { return g_be_has_err ? g_be_err : nullptr; }
int be_set_device(int device) { HIPCHK(hipSetDevice(device)); return 0; }
void* be_malloc(size_t bytes) { void* p = nullptr; if (hipMalloc(&p, bytes) != hipSuccess) return nullptr; return p; }
void be_free(void* p) { (void)hipFree(p); }
int be_h2d(void* dst, const void* src, size_t bytes) { HIPCHK(hipMemcpy(dst, src, bytes, hipMemcpyHostToDevice)); return 0; }
int be_d2d_2d(void* dst, size_t dpitch, const void* src, size_t spitch, size_t width, size_t rows, i2v_stream_t s) {
    HIPCHK(hipMemcpy2DAsync(dst, dpitch, src, spitch, width, rows, hipMemcpyDeviceToDevice, (hipStream_t)s));
    return 0;
}
int be_memset0(void* p, size_t bytes, i2v_stream_t s) { HIPCHK(hipMemsetAsync(p, 0, bytes, (hipStream_t)s)); return 0; }

void* be_event_create() { hipEvent_t e = nullptr; if (hipEventCreate(&e) != hipSuccess) return nullptr; return (void*)e; }
void be_event_destroy(void* ev) { (void)hipEventDestroy((hipEvent_t)ev); }
int be_event_record(void* ev, i2v_stream_t s) { HIPCHK(hipEventRecord((hipEvent_t)ev, (hipStream_t)s)); return 0; }
int be_event_elapsed_ms(void* a, void* b, float* ms) { HIPCHK(hipEventElapsedTime(ms, (hipEvent_t)a, (hipEvent_t)b)); return 0; }
int be_stream_sync(i2v_stream_t s) { HIPCHK(hipStreamSynchronize((hipStream_t)s)); return 0; }
int be_device_sync() { HIPCHK(hipDeviceSynchronize()); return 0; }
i2v_stream_t be_stream_create() { hipStream_t s = nullptr; if (hipStreamCreateWithFlags(&s, hipStreamNonBlocking) != hipSuccess) return nullptr; return (i2v_stream_t)s; }
void be_stream_destroy(i2v_stream_t s) { if (s) (void)hipStreamDestroy((hipStream_t)s); }
int be_stream_wait_event(i2v_stream_t s, void* ev) { HIPCHK(hipStreamWaitEvent((hipStream_t)s, (hipEvent_t)ev, 0)); return 0; }

__constant__ float c_mean[3] = {0.485f, 0.456f, 0.406f};
__constant__ float c_std[3] = {0.229f, 0.224f, 0.225f};

// Tile choice per launch.  A 32x32x2 fp32 MFMA occupies its SIMD for 64 cycles, so a block's matrix
// time is fixed by its tile; what varies is how evenly the grid covers the 256 CUs (the 14x14 layers
// have only a few hundred 128x128 tiles) against the extra operand traffic of small tiles.
static int conv_pick(const I2VConvParams& p) {
    const char* force = getenv("I2V_FORCE_CFG");     // developer / test knob (only consulted for launches the autotuner did not pin)
    if (force && *force) { const int f = atoi(force); return (p.pre_scale && f != 0 && f != 3) ? 3 : f; }
    if (p.Cd <= 16 && !p.pre_scale) return 5;        // 16-row fragments: no padding rows to speak of
    if (conv_bf3_ok(p)) return p.Cd > 64 ? 2 : 3;    // (without the autotuner)
    if (conv_wants_prefetch(p)) return 3;
    if (p.pre_scale) {              // pre-activation variants exist for the 128x128 and 64x64 tiles only
        const double blocks128 = ceil(p.Cd / 128.0) * ceil((double)p.N * p.Hg * p.Wg / 128.0);
        return (p.Cd > 64 && blocks128 >= 256.0 * 6) ? 0 : 3;
    }
    // ineff: relative cost per unit of tile area measured with tools/conv_microbench.cpp (small tiles pay
    // more operand traffic per MFMA); a launch that cannot fill the CUs' block slots also loses the overlap
    // between co-resident blocks.
    static const struct { int BD, BP, occ; double ineff; } C[5] = {
        {128, 128, 3, 1.00}, {64, 128, 5, 1.05}, {128, 64, 5, 1.04}, {64, 64, 7, 1.10}, {32, 256, 4, 1.08}};      // occ = resident blocks per CU (conv_waves_per_simd)
    const double P = (double)p.N * p.Hg * p.Wg;
    const int nchunks = p.Kpad / I2V_KC;
    int best = 0; double best_t = 1e300;
    for (int i = 0; i < 5; ++i) {
        if (C[i].BD > 32 && p.Cd <= C[i].BD / 2) continue;        // more than half the rows would be padding
        if (C[i].BD == 32 && p.Cd > 32) continue;
        const double blocks = ceil(p.Cd / (double)C[i].BD) * ceil(P / C[i].BP);
        const double rounds = ceil(blocks / 256.0);
        const double fill = blocks / (256.0 * C[i].occ);
        const double mfma = (double)nchunks * (C[i].BD / 32) * (C[i].BP / 32) / 4 * 8 * 64 * C[i].ineff *
                            (1.0 + 0.15 * (fill < 1.0 ? 1.0 - fill : 0.0));
        const double overhead = 1500.0 + (C[i].BD * C[i].BP / 256) * 10.0;
        const double t = rounds * mfma + ceil(rounds / C[i].occ) * overhead;
        if (t < best_t) { best_t = t; best = i; }
    }
    return best;
}

// Tile configurations a launch may use (the engine's plan-time autotuner times each of them on the real
// shapes and pins the fastest through I2VConvParams::cfg; `conv_pick` is the model used without it).
int k_conv_candidates(const I2VConvParams& p, int* out) {
    int n = 0;
    if (p.pre_scale) { if (p.Cd > 64) out[n++] = 0; out[n++] = 3; return n; }
#ifdef I2V_EXPERIMENTAL
    if (conv_bf3_ok(p)) {          // split-bf16 K loop: the four square-ish tiles, each with one or two chunks per barrier
        static const int BD3[4] = {128, 64, 128, 64};
        for (int i = 0; i < 4; ++i) {
            if (p.Cd <= BD3[i] / 2) continue;
            out[n++] = i;
            if (I2V_BF3_VARIANT == 2 && (p.Kpad / I2V_KC) % 2 == 0) out[n++] = i | 64;
            if (I2V_BF3_VARIANT == 1 && i == 0 && !p.temporal) out[n++] = 0 | 64;      // 128x128: the software-pipelined loop
        }
        return n;
    }
#endif
    static const int BD[5] = {128, 64, 128, 64, 32};
    for (int i = 0; i < 5; ++i) {
        if (BD[i] > 32 && p.Cd <= BD[i] / 2) continue;
        if (BD[i] == 32 && p.Cd > 32) continue;
        out[n++] = i;
    }
    if (conv_wants_prefetch(p)) out[n++] = 3 | 8;       // 64x64 WITHOUT the epilogue-operand prefetch
    else if (conv_tail_px_tiles(p) > 0 && p.Cd > 32) out[n++] = 3 | 32;      // 64x64 with the remainder tiles cut into quarter tiles
    static const bool no_halo = [] { const char* e = getenv("I2V_HALO"); return e && e[0] == '0'; }();
    if (conv_halo_ok(p) && p.Cd > 32 && !no_halo) out[n++] = 3 | 16;         // 64x64 with halo staging (MODE 5)
    static const bool no_dc = [] { const char* e = getenv("I2V_DC"); return e && e[0] == '0'; }();
    if (conv_dc_ok(p) && !no_dc) out[n++] = 3 | 64;                          // 64x64 with two chunks per barrier (32-row LDS buffers)
    if (p.Cd <= 16) out[n++] = 5;                        // 16x256 tile on 16x16x4 MFMA fragments
    static const bool no_igh = [] { const char* e = getenv("I2V_IGHALO"); return e && e[0] == '0'; }();
    if (conv_ighalo_ok(p) && !no_igh) out[n++] = (p.Cd <= 16 ? 5 : 4) | 512;      // the class-packed image gradient on a 2-D halo tile (conv_imggrad_halo)
    if (conv_stemhalo_ok(p) && !no_igh) out[n++] = 5 | 1024;                       // the narrow forward stem on a 2-D halo tile (conv_stem_halo)
    static const bool no_vfma = [] { const char* e = getenv("I2V_VFMA"); return e && e[0] == '0'; }();
    if (conv_vfma_ok(p) && !no_vfma) out[n++] = (p.Cd <= 16 ? 5 : 4) | 2048;
    if (conv_igvfma_ok(p) && !no_vfma) out[n++] = 5 | 4096;                        // the quad-row image gradient of a narrow stem on packed-fp32 vector FMAs (conv_igvfma_kernel)      // a narrow (<= 32 x 32) launch with taps at (0, 0) on packed-fp32 vector FMAs (conv_vfma_kernel)
#ifdef I2V_EXPERIMENTAL
    // conv_stem64_halo (the wide 7x7 / 2 forward stem on a 2-D halo tile) is built, bit-identical (values and gate words) and faster in
    // isolation (tools/stem_halo_probe.cpp, 128 frames, random operands: 435 -> 363 us), but in the attack the conv_tile launch it would
    // replace runs at 351 us and the forward pass was 0.3 % SLOWER with it (same box, alternated, gpurun_out r5u): its scalar epilogue
    // -- 64 four-byte stores and 64 ballots per lane -- costs what the staging saves.  Offered to the autotuner only on request (I2V_STEM64=1).
    static const bool want_stem64 = [] { const char* e = getenv("I2V_STEM64"); return e && e[0] == '1'; }();
    if (conv_stem64_ok(p) && want_stem64) out[n++] = 3 | 1024;
    // conv_pw_stream (one persistent role-split workgroup per CU) is built, bit-identical and SLOWER than conv_igemm on every shape it
    // admits (round 5, tools/pw_stream_probe.cpp, profiles/r5_pw_stream_probe.txt: 56 / 80 / 91 TFLOP/s on 64 -> 256 / 128 -> 512 /
    // 256 -> 1024 at 128 frames against 69 / 108 / 117): offered to the autotuner only on request (I2V_PWS=1), like the fused pair
    static const bool want_pws = [] { const char* e = getenv("I2V_PWS"); return e && e[0] == '1'; }();
    if (want_pws && conv_pws_grid(p)) out[n++] = 3 | 256;
#endif
    return n;
}

#ifndef I2V_EXPERIMENTAL
// The fused 3x3 -> pointwise pair (conv_fused_kernel, i2v_conv_exp.hip) is not part of the product library: no pair is fusable, the
// planner's mark_fusable therefore marks none, and a fused launch cannot be reached.
int k_conv_fusable(const I2VConvParams&, const I2VConvParams&) { return 0; }
int k_conv_fused(const I2VConvParams&, const I2VConvParams&, int, i2v_stream_t) {
    snprintf(g_be_err, sizeof g_be_err, "fused conv launch: this library was built without -DI2V_EXPERIMENTAL"); g_be_has_err = true;
    return 1;
}
#endif

int k_conv(const I2VConvParams& p_in, i2v_stream_t s) {
    hipStream_t st = (hipStream_t)s;
    I2VConvParams p = p_in;
    if ((int64_t)p.N * p.Hg * p.Wg + 1024 >= (1ll << 31)) { snprintf(g_be_err, sizeof g_be_err, "conv launch of more than 2^31 grid pixels"); g_be_has_err = true; return 1; }
    conv_magics(p);
    if (p.cfg <= 0) p.cfg = conv_pick(p) + 1;          // the model's pick -- or $I2V_FORCE_CFG, which may carry the variant bits too
    __atomic_fetch_add(&g_stat_conv, 1, __ATOMIC_RELAXED);
#ifdef I2V_EXPERIMENTAL
    if (((p.cfg - 1) & 1024) && conv_stem64_ok(p) && !((uintptr_t)p.src & 15)) {       // wide 7x7 / 2 forward stem on a 2-D halo tile (autotuner bit 10)
        __atomic_fetch_add(&g_stat_sth, 1, __ATOMIC_RELAXED);
        return launch_conv_stem64(p, st);
    }
#endif
    if (((p.cfg - 1) & 1024) && conv_stemhalo_ok(p) && !((uintptr_t)p.src & 15)) {     // narrow forward stem on a 2-D halo tile (autotuner bit 10)
        __atomic_fetch_add(&g_stat_sth, 1, __ATOMIC_RELAXED);
        return launch_conv_stemhalo(p, st);
    }
    if (((p.cfg - 1) & 2048) && conv_vfma_ok(p)) {      // narrow launch on packed-fp32 vector FMAs (autotuner bit 11)
        __atomic_fetch_add(&g_stat_vfma, 1, __ATOMIC_RELAXED);
        return launch_conv_vfma(p, st);
    }
    if (((p.cfg - 1) & 4096) && conv_igvfma_ok(p)) {    // quad-row image gradient on packed-fp32 vector FMAs (autotuner bit 12)
        __atomic_fetch_add(&g_stat_igv, 1, __ATOMIC_RELAXED);
        return launch_conv_igvfma(p, st);
    }
    if (((p.cfg - 1) & 512) && conv_ighalo_ok(p)) {     // image gradient on a 2-D halo tile (autotuner bit 9)
        __atomic_fetch_add(&g_stat_igh, 1, __ATOMIC_RELAXED);
        return launch_conv_ighalo(p, st);
    }
    // Bit 10 pins a halo-tile stem kernel, and the executor then hands the caller's frames over WITHOUT the staging copy whose slack
    // conv_tile's quad-row staging (MODE 4) reads: a quad-row stem that carries the bit and is not eligible must not fall through to it.
    if (((p.cfg - 1) & 1024) && p.quad && !p.ig_th && p.sh > 1) {      // (a STRIDED forward quad-row launch is a stem reading the caller's frames; other launches ignore the bit, as under I2V_FORCE_CFG)
        snprintf(g_be_err, sizeof g_be_err, "conv launch pinned to a halo-tile stem kernel (configuration bit 10) is not eligible for one (source %s16-byte aligned)",
                 ((uintptr_t)p.src & 15) ? "not " : "");
        g_be_has_err = true;
        return 1;
    }
#ifdef I2V_EXPERIMENTAL
    if (((p.cfg - 1) & 256) && conv_pws_grid(p)) {     // persistent role-split pointwise kernel (autotuner bit 8)
        __atomic_fetch_add(&g_stat_pws, 1, __ATOMIC_RELAXED);
        return launch_conv_pws(p, st);
    }
    if (conv_bf3_ok(p) && ((p.cfg - 1) & 7) <= 3) {    // split-bf16 K loop (I2V_MATH=bf16x3; bit 6 of the configuration: its pipelined variant)
        __atomic_fetch_add(&g_stat_bf3, 1, __ATOMIC_RELAXED);
        return launch_conv_bf3(p, st);
    }
#endif
    switch ((p.cfg - 1) & 7) {
        case 0: return launch_conv_cfg0(p, st);
        case 1: return launch_conv_cfg1(p, st);
        case 2: return launch_conv_cfg2(p, st);
        case 3: return launch_conv_cfg3(p, st);
        case 5: return launch_conv_cfg5(p, st);
        default: return launch_conv_cfg4(p, st);
    }
}

// =============================================================================================
// max pooling (window-relative arg-max byte saved by forward: first maximum in scan order, as ATen)
// =============================================================================================
// Both pooling kernels: one block per (frame, channel) plane and band of output rows; the band's input
// rows (fwd) / arg-max bytes and upstream gradients (bwd) are staged in LDS with coalesced loads, so the
// kernels stream at HBM rate instead of issuing k*k strided global loads per element.  32-bit index math.
#define POOL_LDS_FLOATS 8192
#define POOL_FWD_LDS_FLOATS 4096    // forward band: 16 KB -> 10 blocks per CU in flight (8192: 2.7 TB/s, 4096: 4.0, 2048: 3.6)
__global__ void __launch_bounds__(256) pool_fwd_kernel(const I2VPoolParams p, const int band_rows) {
    __shared__ float xs[POOL_FWD_LDS_FLOATS];
    const int plane = blockIdx.x, n = plane / p.C, c = plane - n * p.C;
    const int ho0 = blockIdx.y * band_rows, ho1 = min(ho0 + band_rows, p.Ho);
    const int h_lo = max(ho0 * p.stride - p.pad, 0), h_hi = min((ho1 - 1) * p.stride - p.pad + p.k, p.Hs);   // [h_lo, h_hi)
    const float* x = p.x + (int64_t)n * p.x_nstride + (int64_t)c * p.Hs * p.Ws;
    const int cnt = (h_hi - h_lo) * p.Ws;
    const float* xb = x + h_lo * p.Ws;
    if (((p.Ws & 3) == 0) && ((((uintptr_t)xb) & 15) == 0)) {
        for (int e = threadIdx.x * 4; e < cnt; e += 1024) *reinterpret_cast<float4*>(&xs[e]) = *reinterpret_cast<const float4*>(xb + e);
    } else {
        for (int e = threadIdx.x; e < cnt; e += 256) xs[e] = xb[e];
    }
    __syncthreads();
    float* y = p.y + (int64_t)n * p.y_nstride + (int64_t)c * p.Ho * p.Wo;
    uint8_t* ix = p.idx + (int64_t)plane * p.Ho * p.Wo;
    const int nout = (ho1 - ho0) * p.Wo;
    for (int e = threadIdx.x; e < nout; e += 256) {
        const int ho = ho0 + e / p.Wo, wo = e % p.Wo;
        int best = -1; float bv = 0.f;
        for (int kr = 0; kr < p.k; ++kr) {
            const int h = ho * p.stride - p.pad + kr; if (h < 0 || h >= p.Hs) continue;
            for (int ks = 0; ks < p.k; ++ks) {
                const int w = wo * p.stride - p.pad + ks; if (w < 0 || w >= p.Ws) continue;
                const float v = xs[(h - h_lo) * p.Ws + w];
                if (best < 0 || v > bv || v != v) { bv = v; best = kr * p.k + ks; }    // first maximum wins (ATen)
            }
        }
        y[ho * p.Wo + wo] = bv;
        ix[ho * p.Wo + wo] = (uint8_t)best;
    }
}

// gather form (no atomics): an input element collects from the <= ceil(k/stride)^2 windows holding it whose
// stored arg-max points back at it.  The ReLU gate of the pooled tensor (x > 0) is taken from the pooled OUTPUT:
// an element only receives gradient from a window whose maximum it is, and then x equals that window's y -- so the
// full-resolution activation (4x the bytes of y) is not read at all.
// KK/SS/PP: window, stride and padding as compile-time constants for the common geometries (the index divisions become
// shifts); KK == 0 reads them from the parameters.
template <int KK, int SS, int PP>
__global__ void __launch_bounds__(256) pool_bwd_kernel(const I2VPoolParams p, const int band_rows) {
    const int pk = KK ? KK : p.k, pstride = KK ? SS : p.stride, ppad = KK ? PP : p.pad;
    __shared__ float gs[POOL_LDS_FLOATS / 2];
    __shared__ uint8_t is[POOL_LDS_FLOATS / 2];
    const int plane = blockIdx.x, n = plane / p.C, c = plane - n * p.C;
    const int h0 = blockIdx.y * band_rows, h1 = min(h0 + band_rows, p.Hs);           // input rows of this band
    int ho_lo = h0 + ppad - pk + 1; ho_lo = ho_lo <= 0 ? 0 : (ho_lo + pstride - 1) / pstride;
    const int ho_hi = min((h1 - 1 + ppad) / pstride, p.Ho - 1);                    // output rows [ho_lo, ho_hi]
    const float* gy = p.y + (int64_t)n * p.y_nstride + (int64_t)c * p.Ho * p.Wo;
    const float* yv = p.yact ? p.yact + (int64_t)n * p.yact_nstride + (int64_t)c * p.Ho * p.Wo : nullptr;
    const uint8_t* ix = p.idx + (int64_t)plane * p.Ho * p.Wo;
    const int cnt = (ho_hi - ho_lo + 1) * p.Wo;
    for (int e = threadIdx.x; e < cnt; e += 256) {
        float g = gy[ho_lo * p.Wo + e];
        if (yv && !(yv[ho_lo * p.Wo + e] > 0.f)) g = 0.f;       // gate folded into the staged upstream gradient
        gs[e] = g; is[e] = ix[ho_lo * p.Wo + e];
    }
    __syncthreads();
    const float* x = p.x + (int64_t)n * p.x_nstride + (int64_t)c * p.Hs * p.Ws;
    float* gx = p.gx + (int64_t)n * p.gx_nstride + (int64_t)c * p.Hs * p.Ws;
    const int nin = (h1 - h0) * p.Ws;
    const bool gate_x = p.mask_relu && !yv;                      // no pooled activation supplied: gate on x itself
    const bool vec = ((p.Ws & 3) == 0) && ((((uintptr_t)(x + h0 * p.Ws) | (uintptr_t)(gx + h0 * p.Ws)) & 15) == 0);
    for (int e4 = threadIdx.x * (vec ? 4 : 1); e4 < nin; e4 += 256 * (vec ? 4 : 1)) {
        float xv[4] = {1.f, 1.f, 1.f, 1.f}, gv[4];
        if (gate_x) {
            if (vec) { const float4 t4 = *reinterpret_cast<const float4*>(x + h0 * p.Ws + e4); xv[0] = t4.x; xv[1] = t4.y; xv[2] = t4.z; xv[3] = t4.w; }
            else xv[0] = x[h0 * p.Ws + e4];
        }
        const int h = h0 + e4 / p.Ws, wb = e4 % p.Ws;
        int a_lo = h + ppad - pk + 1; a_lo = a_lo <= 0 ? 0 : (a_lo + pstride - 1) / pstride;
        const int a_hi = min((h + ppad) / pstride, p.Ho - 1);
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            if (!vec && u > 0) break;
            const int w = wb + u;
            float g = 0.f;
            if (!gate_x || xv[u] > 0.f) {
                int b_lo = w + ppad - pk + 1; b_lo = b_lo <= 0 ? 0 : (b_lo + pstride - 1) / pstride;
                const int b_hi = min((w + ppad) / pstride, p.Wo - 1);
                for (int ho = a_lo; ho <= a_hi; ++ho)
                    for (int wo = b_lo; wo <= b_hi; ++wo) {
                        const int me = (h - (ho * pstride - ppad)) * pk + (w - (wo * pstride - ppad));
                        const int li = (ho - ho_lo) * p.Wo + wo;
                        if (is[li] == me) g += gs[li];
                    }
            }
            gv[u] = g;
        }
        if (vec) *reinterpret_cast<float4*>(gx + h0 * p.Ws + e4) = make_float4(gv[0], gv[1], gv[2], gv[3]);
        else gx[h0 * p.Ws + e4] = gv[0];
    }
}

// Round 4: the 3 / 2 / 1 window of the ResNet stems (112^2 -> 56^2, 540 MB per launch at 128 frames -- three quarters of them the
// gradient it WRITES) as a patch kernel.  The generic gather above spends ~150 instructions per float4 (two runtime divisions, a
// window loop with byte compares per element: 2.2 TB/s of algorithmic bytes, 0.28 of the HBM peak).  Here a thread owns a 2 x 4
// input patch (rows 2a, 2a+1; columns 4b .. 4b+3): the windows that can point into it are the 2 x 3 outputs (a .. a+1, 2b .. 2b+2),
// read from LDS once; which of them covers which element, and with which window-relative index, is a compile-time table (an even
// row / column is the centre of one window, an odd one the edge of two), so an element is <= 4 compare-select-adds -- in the
// generic kernel's order (output row, then output column), hence bit-identical to it and to the scalar restatement.
__global__ void __launch_bounds__(256) pool_bwd_321_kernel(const I2VPoolParams p, const int band_pairs, const unsigned w4_m, const unsigned w4_s) {
    __shared__ float gs[POOL_LDS_FLOATS / 2];
    __shared__ uint8_t is[POOL_LDS_FLOATS / 2];
    const int plane = blockIdx.x, n = plane / p.C, c = plane - n * p.C;
    const int a0 = blockIdx.y * band_pairs, a1 = min(a0 + band_pairs, p.Hs >> 1);      // pair rows [a0, a1) = input rows [2 a0, 2 a1)
    const int ho_lo = a0, ho_hi = min(a1, p.Ho - 1);                                     // output rows [a0, min(a1, Ho - 1)]
    const float* gy = p.y + (int64_t)n * p.y_nstride + (int64_t)c * p.Ho * p.Wo;
    const float* yv = p.yact ? p.yact + (int64_t)n * p.yact_nstride + (int64_t)c * p.Ho * p.Wo : nullptr;
    const uint8_t* ix = p.idx + (int64_t)plane * p.Ho * p.Wo;
    const int cnt = (ho_hi - ho_lo + 1) * p.Wo;
    for (int e = threadIdx.x; e < cnt; e += 256) {
        float g = gy[ho_lo * p.Wo + e];
        if (yv && !(yv[ho_lo * p.Wo + e] > 0.f)) g = 0.f;       // gate folded into the staged upstream gradient
        gs[e] = g; is[e] = ix[ho_lo * p.Wo + e];
    }
    __syncthreads();
    float* gx = p.gx + (int64_t)n * p.gx_nstride + (int64_t)c * p.Hs * p.Ws;
    const int W4 = p.Ws >> 2, items = (a1 - a0) * W4;
    for (int e = threadIdx.x; e < items; e += 256) {
        const int al = (int)fastdiv((unsigned)e, w4_m, w4_s), b = e - al * W4, a = a0 + al;
        // the 2 x 3 windows (output rows a, a+1; columns 2b, 2b+1, 2b+2); rows / columns past the pooled plane contribute nothing
        float g[2][3]; int ix6[2][3];
#pragma unroll
        for (int r = 0; r < 2; ++r)
#pragma unroll
            for (int q = 0; q < 3; ++q) {
                const bool in = (a + r) <= ho_hi && (2 * b + q) < p.Wo;
                const int li = (al + r) * p.Wo + 2 * b + q;
                g[r][q] = in ? gs[li] : 0.f;
                ix6[r][q] = in ? (int)is[li] : 255;
            }
        // element (row parity rp, column u): windows in (output row, output column) order with their window-relative index kr * 3 + ks.
        // row 2a: window row a with kr = 1; row 2a+1: window rows a (kr = 2) and a+1 (kr = 0).
        // column 4b: window column 2b, ks = 1; 4b+1: 2b (ks = 2), 2b+1 (ks = 0); 4b+2: 2b+1, ks = 1; 4b+3: 2b+1 (ks = 2), 2b+2 (ks = 0).
        float o[2][4];
#pragma unroll
        for (int rp = 0; rp < 2; ++rp)
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                float acc = 0.f;
#pragma unroll
                for (int wr = 0; wr < (rp ? 2 : 1); ++wr) {
                    const int kr = rp ? (wr ? 0 : 2) : 1;
#pragma unroll
                    for (int wc = 0; wc < ((u & 1) ? 2 : 1); ++wc) {
                        const int q = (u >> 1) + ((u & 1) ? wc : 0) + ((u == 2) ? 0 : 0);       // window column 2b + q
                        const int ks = (u & 1) ? (wc ? 0 : 2) : 1;
                        if (ix6[wr][q] == kr * 3 + ks) acc += g[wr][q];
                    }
                }
                o[rp][u] = acc;
            }
        float* row = gx + (int64_t)(2 * a) * p.Ws + 4 * b;
        *reinterpret_cast<float4*>(row) = make_float4(o[0][0], o[0][1], o[0][2], o[0][3]);
        *reinterpret_cast<float4*>(row + p.Ws) = make_float4(o[1][0], o[1][1], o[1][2], o[1][3]);
    }
}

__global__ void avgpool_fwd_kernel(const I2VPoolParams p) {
    const int64_t total = (int64_t)p.N * p.C * p.Ho * p.Wo;
    const float inv = 1.f / (float)(p.k * p.k);
    for (int64_t idx = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
        const int wo = idx % p.Wo; int64_t r = idx / p.Wo;
        const int ho = r % p.Ho; r /= p.Ho;
        const int c = r % p.C; const int64_t n = r / p.C;
        const float* pl = p.x + n * p.x_nstride + (int64_t)c * p.Hs * p.Ws;
        float s = 0.f;
        for (int kr = 0; kr < p.k; ++kr)
            for (int ks = 0; ks < p.k; ++ks) s += pl[(ho * p.stride + kr) * p.Ws + wo * p.stride + ks];
        p.y[n * p.y_nstride + ((int64_t)c * p.Ho + ho) * p.Wo + wo] = s * inv;
    }
}

__global__ void avgpool_bwd_kernel(const I2VPoolParams p) {
    const int64_t total = (int64_t)p.N * p.C * p.Hs * p.Ws;
    const float inv = 1.f / (float)(p.k * p.k);
    for (int64_t idx = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
        const int w = idx % p.Ws; int64_t r = idx / p.Ws;
        const int h = r % p.Hs; r /= p.Hs;
        const int c = r % p.C; const int64_t n = r / p.C;
        const int ho = h / p.stride, wo = w / p.stride;
        const bool in = (h - ho * p.stride) < p.k && (w - wo * p.stride) < p.k && ho < p.Ho && wo < p.Wo;
        p.gx[n * p.gx_nstride + ((int64_t)c * p.Hs + h) * p.Ws + w] =
            in ? p.y[n * p.y_nstride + ((int64_t)c * p.Ho + ho) * p.Wo + wo] * inv : 0.f;
    }
}

static unsigned stream_grid(int64_t total, int per_block) {
    int64_t b = (total + per_block - 1) / per_block;
    if (b < 1) b = 1;
    if (b > 256 * 16) b = 256 * 16;
    return (unsigned)b;
}

static int pool_fail(const char* m) { snprintf(g_be_err, sizeof g_be_err, "%s", m); g_be_has_err = true; return 1; }

int k_pool_fwd(const I2VPoolParams& p, i2v_stream_t s) {
    // band of output rows whose input rows fit the LDS buffer
    if (POOL_FWD_LDS_FLOATS / p.Ws < p.k) return pool_fail("max-pool row too wide for the LDS band");     // (a negative numerator would truncate towards 0 below)
    int band = (POOL_FWD_LDS_FLOATS / p.Ws - p.k) / p.stride + 1;
    if (band < 1) return pool_fail("max-pool row too wide for the LDS band");
    if (band > p.Ho) band = p.Ho;
    dim3 grid((unsigned)(p.N * p.C), (unsigned)((p.Ho + band - 1) / band));
    hipLaunchKernelGGL(pool_fwd_kernel, grid, dim3(256), 0, (hipStream_t)s, p, band);
    LAUNCH_CHECK("pool_fwd"); return 0;
}
int k_pool_bwd(const I2VPoolParams& p, i2v_stream_t s) {
    // band of input rows whose covering output rows fit the LDS buffers
    int out_rows = (POOL_LDS_FLOATS / 2) / p.Wo;
    if (out_rows < 1) return pool_fail("max-pool row too wide for the LDS band");
    int band = (out_rows - 1) * p.stride - p.k + 1; if (out_rows >= p.Ho) band = p.Hs;
    if (band < 1) return pool_fail("max-pool row too wide for the LDS band");
    if (band > p.Hs) band = p.Hs;
    dim3 grid((unsigned)(p.N * p.C), (unsigned)((p.Hs + band - 1) / band));
    if (p.k == 3 && p.stride == 2 && p.pad == 1 && (p.yact || !p.mask_relu) && (p.Ws & 3) == 0 && (p.Hs & 1) == 0 && p.Ho * 2 == p.Hs && p.Wo * 2 == p.Ws &&
        (((uintptr_t)p.gx | (uintptr_t)(p.gx_nstride * 4)) & 15) == 0) {
        // patch kernel: bands of pair rows whose output rows (one more than the pairs) fit the LDS buffers
        int pairs = (POOL_LDS_FLOATS / 2) / p.Wo - 1;
        if (pairs < 1) return pool_fail("max-pool row too wide for the LDS band");
        if (pairs > p.Hs / 2) pairs = p.Hs / 2;
        uint32_t m, sh; fastdiv_magic((unsigned)(p.Ws >> 2), &m, &sh);
        dim3 g2((unsigned)(p.N * p.C), (unsigned)((p.Hs / 2 + pairs - 1) / pairs));
        hipLaunchKernelGGL(pool_bwd_321_kernel, g2, dim3(256), 0, (hipStream_t)s, p, pairs, m, sh);
    } else if (p.k == 3 && p.stride == 2 && p.pad == 1) hipLaunchKernelGGL((pool_bwd_kernel<3, 2, 1>), grid, dim3(256), 0, (hipStream_t)s, p, band);
    else if (p.k == 2 && p.stride == 2 && p.pad == 0) hipLaunchKernelGGL((pool_bwd_kernel<2, 2, 0>), grid, dim3(256), 0, (hipStream_t)s, p, band);
    else if (p.k == 3 && p.stride == 2 && p.pad == 0) hipLaunchKernelGGL((pool_bwd_kernel<3, 2, 0>), grid, dim3(256), 0, (hipStream_t)s, p, band);
    else hipLaunchKernelGGL((pool_bwd_kernel<0, 0, 0>), grid, dim3(256), 0, (hipStream_t)s, p, band);
    LAUNCH_CHECK("pool_bwd"); return 0;
}

// ---- video max pooling (frame-major clips; window kt x k x k).  Small tensors on this path (the pools of an
// inflated ResNet stem): one thread per element, arg-max byte (q*k + r)*k + s, gather-form backward. ----
__global__ void pool3d_fwd_kernel(const I2VPoolParams p) {
    const int64_t total = (int64_t)p.N * p.C * p.Ho * p.Wo;
    for (int64_t idx = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
        const int wo = idx % p.Wo; int64_t r = idx / p.Wo;
        const int ho = r % p.Ho; r /= p.Ho;
        const int c = r % p.C; const int64_t n = r / p.C;
        const int64_t clip = n / p.To; const int to = (int)(n - clip * p.To);
        int best = -1; float bv = 0.f;
        for (int q = 0; q < p.kt; ++q) {
            const int ts = to * p.stride_t - p.pad_t + q; if (ts < 0 || ts >= p.Ts) continue;
            const float* pl = p.x + (clip * p.Ts + ts) * p.x_nstride + (int64_t)c * p.Hs * p.Ws;
            for (int kr = 0; kr < p.k; ++kr) {
                const int h = ho * p.stride - p.pad + kr; if (h < 0 || h >= p.Hs) continue;
                for (int ks = 0; ks < p.k; ++ks) {
                    const int w = wo * p.stride - p.pad + ks; if (w < 0 || w >= p.Ws) continue;
                    const float v = pl[h * p.Ws + w];
                    if (best < 0 || v > bv || v != v) { bv = v; best = (q * p.k + kr) * p.k + ks; }   // first maximum wins (ATen)
                }
            }
        }
        p.y[n * p.y_nstride + ((int64_t)c * p.Ho + ho) * p.Wo + wo] = bv;
        p.idx[idx] = (uint8_t)best;
    }
}

__global__ void pool3d_bwd_kernel(const I2VPoolParams p) {
    const int64_t clips = p.N / p.To, total = clips * p.Ts * p.C * p.Hs * p.Ws;
    for (int64_t idx = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
        const int w = idx % p.Ws; int64_t r = idx / p.Ws;
        const int h = r % p.Hs; r /= p.Hs;
        const int c = r % p.C; const int64_t ns = r / p.C;
        const int64_t clip = ns / p.Ts; const int ts = (int)(ns - clip * p.Ts);
        const int64_t xo = ns * p.x_nstride + ((int64_t)c * p.Hs + h) * p.Ws + w;
        float g = 0.f;
        if (!p.mask_relu || p.x[xo] > 0.f) {
            int t_lo = ts + p.pad_t - p.kt + 1; t_lo = t_lo <= 0 ? 0 : (t_lo + p.stride_t - 1) / p.stride_t;
            const int t_hi = min((ts + p.pad_t) / p.stride_t, p.To - 1);
            int a_lo = h + p.pad - p.k + 1; a_lo = a_lo <= 0 ? 0 : (a_lo + p.stride - 1) / p.stride;
            const int a_hi = min((h + p.pad) / p.stride, p.Ho - 1);
            int b_lo = w + p.pad - p.k + 1; b_lo = b_lo <= 0 ? 0 : (b_lo + p.stride - 1) / p.stride;
            const int b_hi = min((w + p.pad) / p.stride, p.Wo - 1);
            for (int to = t_lo; to <= t_hi; ++to)
                for (int ho = a_lo; ho <= a_hi; ++ho)
                    for (int wo = b_lo; wo <= b_hi; ++wo) {
                        const int me = ((ts - (to * p.stride_t - p.pad_t)) * p.k + (h - (ho * p.stride - p.pad))) * p.k +
                                       (w - (wo * p.stride - p.pad));
                        const int64_t no = clip * p.To + to;
                        if (p.idx[((no * p.C + c) * p.Ho + ho) * p.Wo + wo] == me)
                            g += p.y[no * p.y_nstride + ((int64_t)c * p.Ho + ho) * p.Wo + wo];
                    }
        }
        p.gx[ns * p.gx_nstride + ((int64_t)c * p.Hs + h) * p.Ws + w] = g;
    }
}

int k_pool3d_fwd(const I2VPoolParams& p, i2v_stream_t s) {
    if (p.kt * p.k * p.k > 256) return pool_fail("video max-pool window larger than 256 taps");
    hipLaunchKernelGGL(pool3d_fwd_kernel, dim3(stream_grid((int64_t)p.N * p.C * p.Ho * p.Wo, 256)), dim3(256), 0, (hipStream_t)s, p);
    LAUNCH_CHECK("pool3d_fwd"); return 0;
}
int k_pool3d_bwd(const I2VPoolParams& p, i2v_stream_t s) {
    hipLaunchKernelGGL(pool3d_bwd_kernel, dim3(stream_grid((int64_t)(p.N / p.To) * p.Ts * p.C * p.Hs * p.Ws, 256)), dim3(256), 0, (hipStream_t)s, p);
    LAUNCH_CHECK("pool3d_bwd"); return 0;
}

int k_avgpool_fwd(const I2VPoolParams& p, i2v_stream_t s) {
    hipLaunchKernelGGL(avgpool_fwd_kernel, dim3(stream_grid((int64_t)p.N * p.C * p.Ho * p.Wo, 256)), dim3(256), 0, (hipStream_t)s, p);
    LAUNCH_CHECK("avgpool_fwd"); return 0;
}
int k_avgpool_bwd(const I2VPoolParams& p, i2v_stream_t s) {
    hipLaunchKernelGGL(avgpool_bwd_kernel, dim3(stream_grid((int64_t)p.N * p.C * p.Hs * p.Ws, 256)), dim3(256), 0, (hipStream_t)s, p);
    LAUNCH_CHECK("avgpool_bwd"); return 0;
}

// =============================================================================================
// out = (a0 + a1 + a2) gated by mask > 0
// =============================================================================================
__global__ void addmask_kernel(const I2VAddMaskParams p) {
    const int64_t plane = (int64_t)p.C * p.HW, total = plane * p.N;
    for (int64_t idx = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
        const int64_t n = idx / plane, i = idx - n * plane;
        float v = 0.f;
        if (p.a[0]) v += p.a[0][n * p.a_nstride[0] + i];
        if (p.a[1]) v += p.a[1][n * p.a_nstride[1] + i];
        if (p.a[2]) v += p.a[2][n * p.a_nstride[2] + i];
        if (p.gate) {
            const int64_t c = i / p.HW, bit = n * p.HW + (i - c * p.HW);
            if (!((p.gate[c * p.gate_stride + (bit >> 5)] >> (bit & 31)) & 1u)) v = 0.f;
        } else if (p.mask && !(p.mask[n * p.mask_nstride + i] > 0.f)) v = 0.f;
        if (p.gain != 0.f) v = __fmul_rn(p.gain, v);
        p.out[n * p.out_nstride + i] = v;
    }
}
int k_addmask(const I2VAddMaskParams& p, i2v_stream_t s) {
    hipLaunchKernelGGL(addmask_kernel, dim3(stream_grid((int64_t)p.N * p.C * p.HW, 256)), dim3(256), 0, (hipStream_t)s, p);
    LAUNCH_CHECK("addmask"); return 0;
}

// =============================================================================================
// cosine similarity forward + gradient (two launches, deterministic: no atomics)
// =============================================================================================
int cos_nblk(int64_t D) { int64_t b = D / 4096; if (b < 1) b = 1; if (b > 64) b = 64; return (int)b; }

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    return v;
}
__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    return v;
}

// grid (nblk, N): partial[n][blk] = (dot, aa, bb) over this block's slice of the feature
__global__ void __launch_bounds__(256) cos_reduce_kernel(const I2VCosParams p) {
    const int blk = blockIdx.x, n = blockIdx.y;
    const int64_t chunk = ((p.D + p.nblk - 1) / p.nblk + 3) & ~(int64_t)3;
    const int64_t lo = blk * chunk, hi = (lo + chunk < p.D) ? lo + chunk : p.D;
    const float* a = p.a + (int64_t)n * p.a_nstride;
    const float* b = p.b + (int64_t)n * p.b_nstride;
    float dot = 0.f, aa = 0.f, bb = 0.f;
    const bool vec = ((p.a_nstride | p.b_nstride) & 3) == 0 && (((uintptr_t)p.a | (uintptr_t)p.b) & 15) == 0;
    if (vec) {
        const int64_t hi4 = lo + ((hi - lo) & ~(int64_t)3);
        for (int64_t i = lo + threadIdx.x * 4; i < hi4; i += 1024) {
            const float4 x = *reinterpret_cast<const float4*>(a + i), y = *reinterpret_cast<const float4*>(b + i);
            dot += x.x * y.x + x.y * y.y + x.z * y.z + x.w * y.w;
            aa += x.x * x.x + x.y * x.y + x.z * x.z + x.w * x.w;
            bb += y.x * y.x + y.y * y.y + y.z * y.z + y.w * y.w;
        }
        for (int64_t i = hi4 + threadIdx.x; i < hi; i += 256) { dot += a[i] * b[i]; aa += a[i] * a[i]; bb += b[i] * b[i]; }
    } else {
        for (int64_t i = lo + threadIdx.x; i < hi; i += 256) { dot += a[i] * b[i]; aa += a[i] * a[i]; bb += b[i] * b[i]; }
    }
    __shared__ float red[3][4];
    dot = wave_sum(dot); aa = wave_sum(aa); bb = wave_sum(bb);
    if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = dot; red[1][threadIdx.x >> 6] = aa; red[2][threadIdx.x >> 6] = bb; }
    __syncthreads();
    if (threadIdx.x == 0) {
        float* o = p.partial + ((int64_t)n * p.nblk + blk) * 4;
        o[0] = (red[0][0] + red[0][1]) + (red[0][2] + red[0][3]);
        o[1] = (red[1][0] + red[1][1]) + (red[1][2] + red[1][3]);
        o[2] = (red[2][0] + red[2][1]) + (red[2][2] + red[2][3]);
    }
}

// grid (gblk, N): finish the reduction (double), write cos, then the gradient elementwise
__global__ void __launch_bounds__(256) cos_grad_kernel(const I2VCosParams p) {
    const int n = blockIdx.y;
    __shared__ double fin[3];
    if (threadIdx.x < 64) {
        double d = 0, x = 0, y = 0;
        if ((int)threadIdx.x < p.nblk) {
            const float* o = p.partial + ((int64_t)n * p.nblk + threadIdx.x) * 4;
            d = o[0]; x = o[1]; y = o[2];
        }
        d = wave_sum_d(d); x = wave_sum_d(x); y = wave_sum_d(y);
        if (threadIdx.x == 0) { fin[0] = d; fin[1] = x; fin[2] = y; }
    }
    __syncthreads();
    const double n1 = fmax(sqrt(fin[1]), 1e-8), n2 = fmax(sqrt(fin[2]), 1e-8);
    const double cs = fin[0] / (n1 * n2);
    if (blockIdx.x == 0 && threadIdx.x == 0) p.cos_out[n] = (float)cs;
    double coef = (double)p.coef_host;
    if (p.coef_dev) coef *= (double)p.coef_dev[p.coef_index];
    const double c1 = coef / (n1 * n2), c2 = coef * cs / (n1 * n1);
    const float* a = p.a + (int64_t)n * p.a_nstride;
    const float* b = p.b + (int64_t)n * p.b_nstride;
    float* g = p.grad + (int64_t)n * p.grad_nstride;
    for (int64_t i = blockIdx.x * 256 + threadIdx.x; i < p.D; i += (int64_t)gridDim.x * 256) {
        const float av = a[i];
        float v = (float)(c1 * (double)b[i] - c2 * (double)av);
        if (p.mask_relu && !(av > 0.f)) v = 0.f;
        g[i] = p.accumulate ? g[i] + v : v;
    }
}

int k_cos(const I2VCosParams& p, i2v_stream_t s) {
    hipLaunchKernelGGL(cos_reduce_kernel, dim3(p.nblk, p.N), dim3(256), 0, (hipStream_t)s, p);
    LAUNCH_CHECK("cos_reduce");
    int gblk = (int)((p.D + 2047) / 2048); if (gblk > 64) gblk = 64;
    hipLaunchKernelGGL(cos_grad_kernel, dim3(gblk, p.N), dim3(256), 0, (hipStream_t)s, p);
    LAUNCH_CHECK("cos_grad");
    return 0;
}

// =============================================================================================
// Dispersion-Reduction loss: unbiased std over the whole tensor
// =============================================================================================
__global__ void __launch_bounds__(256) std_reduce_kernel(const I2VStdParams p) {
    const int blk = blockIdx.x, n = blockIdx.y;
    const int64_t chunk = (p.D + p.nblk - 1) / p.nblk;
    const int64_t lo = blk * chunk, hi = (lo + chunk < p.D) ? lo + chunk : p.D;
    const float* a = p.a + (int64_t)n * p.a_nstride;
    double s = 0, ss = 0;
    for (int64_t i = lo + threadIdx.x; i < hi; i += 256) { const double v = a[i]; s += v; ss += v * v; }
    __shared__ double red[2][4];
    s = wave_sum_d(s); ss = wave_sum_d(ss);
    if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = s; red[1][threadIdx.x >> 6] = ss; }
    __syncthreads();
    if (threadIdx.x == 0) {
        double* o = p.partial + ((int64_t)n * p.nblk + blk) * 2;
        o[0] = (red[0][0] + red[0][1]) + (red[0][2] + red[0][3]);
        o[1] = (red[1][0] + red[1][1]) + (red[1][2] + red[1][3]);
    }
}

__global__ void __launch_bounds__(256) std_finish_kernel(const I2VStdParams p) {
    __shared__ double red[2][4];
    double s = 0, ss = 0;
    const int np = p.N * p.nblk;
    for (int i = threadIdx.x; i < np; i += 256) { s += p.partial[2 * i]; ss += p.partial[2 * i + 1]; }
    s = wave_sum_d(s); ss = wave_sum_d(ss);
    if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = s; red[1][threadIdx.x >> 6] = ss; }
    __syncthreads();
    if (threadIdx.x == 0) {
        p.sums[0] = (red[0][0] + red[0][1]) + (red[0][2] + red[0][3]);
        p.sums[1] = (red[1][0] + red[1][1]) + (red[1][2] + red[1][3]);
    }
}

__global__ void __launch_bounds__(256) std_grad_kernel(const I2VStdParams p) {
    const int n = blockIdx.y;
    const double s = p.sums[0], ss = p.sums[1];
    const double cnt = p.total_count;
    const double mu = s / cnt;
    const double var = fmax((ss - cnt * mu * mu) / (cnt - 1.0), 0.0);
    const double sd = sqrt(var);
    if (blockIdx.x == 0 && n == 0 && threadIdx.x == 0) p.std_out[0] = (float)sd;
    const double inv = 1.0 / ((cnt - 1.0) * sd);
    const float* a = p.a + (int64_t)n * p.a_nstride;
    float* g = p.grad + (int64_t)n * p.grad_nstride;
    for (int64_t i = blockIdx.x * 256 + threadIdx.x; i < p.D; i += (int64_t)gridDim.x * 256) {
        const float av = a[i];
        float v = (float)(((double)av - mu) * inv);
        if (p.mask_relu && !(av > 0.f)) v = 0.f;
        g[i] = p.accumulate ? g[i] + v : v;
    }
}

int k_std_reduce(const I2VStdParams& p, i2v_stream_t s) {
    hipLaunchKernelGGL(std_reduce_kernel, dim3(p.nblk, p.N), dim3(256), 0, (hipStream_t)s, p);
    LAUNCH_CHECK("std_reduce");
    hipLaunchKernelGGL(std_finish_kernel, dim3(1), dim3(256), 0, (hipStream_t)s, p);
    LAUNCH_CHECK("std_finish");
    return 0;
}

int k_std_grad(const I2VStdParams& p, i2v_stream_t s) {
    int gblk = (int)((p.D + 2047) / 2048); if (gblk > 64) gblk = 64;
    hipLaunchKernelGGL(std_grad_kernel, dim3(gblk, p.N), dim3(256), 0, (hipStream_t)s, p);
    LAUNCH_CHECK("std_grad");
    return 0;
}

// =============================================================================================
// ILAF loss (image_attacks.py:579-611) over one hooked tensor: whole-tensor norms, so reduce -> finish -> grad
// =============================================================================================
__device__ __forceinline__ float tap_root(float x) { return x > 0.f ? sqrtf(x) : (x < 0.f ? -sqrtf(-x) : 0.f); }
__global__ void __launch_bounds__(256) ilaf_reduce_kernel(const I2VIlafParams p) {
    const int blk = blockIdx.x, n = blockIdx.y;
    const int64_t chunk = (p.D + p.nblk - 1) / p.nblk;
    const int64_t lo = blk * chunk, hi = (lo + chunk < p.D) ? lo + chunk : p.D;
    const float* a = p.a + (int64_t)n * p.a_nstride;
    const float* o = p.ori + (int64_t)n * p.D;
    const float* a0 = p.adv0 + (int64_t)n * p.D;
    double dd = 0, dq = 0;
    if (p.mode == 1) {                                       // TAP: r(a) - r(ori) with r(x) = sign(x) sqrt|x| in fp32, as torch
        for (int64_t i = lo + threadIdx.x; i < hi; i += 256) {
            const double d = (double)__fsub_rn(tap_root(a[i]), tap_root(o[i]));
            dd += d * d;
        }
    } else
    for (int64_t i = lo + threadIdx.x; i < hi; i += 256) {
        const float ov = o[i];
        const double d = (double)__fsub_rn(a[i], ov), d0 = (double)__fsub_rn(a0[i], ov);    // fp32 differences, as torch
        dd += d * d; dq += d * d0;
    }
    __shared__ double red[2][4];
    dd = wave_sum_d(dd); dq = wave_sum_d(dq);
    if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = dd; red[1][threadIdx.x >> 6] = dq; }
    __syncthreads();
    if (threadIdx.x == 0) {
        double* out = p.partial + ((int64_t)n * p.nblk + blk) * 2;
        out[0] = (red[0][0] + red[0][1]) + (red[0][2] + red[0][3]);
        out[1] = (red[1][0] + red[1][1]) + (red[1][2] + red[1][3]);
    }
}

// one block per segment (I2VIlafParams::fps): the partials of its frames in the order a one-clip call sums them
__global__ void __launch_bounds__(256) ilaf_finish_kernel(const I2VIlafParams p) {
    __shared__ double red[2][4];
    double s = 0, q = 0;
    const int fps = p.fps > 0 ? p.fps : p.N, seg = blockIdx.x;
    const int np = fps * p.nblk;
    const double* part = p.partial + (int64_t)seg * np * 2;
    for (int i = threadIdx.x; i < np; i += 256) { s += part[2 * i]; q += part[2 * i + 1]; }
    s = wave_sum_d(s); q = wave_sum_d(q);
    if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = s; red[1][threadIdx.x >> 6] = q; }
    __syncthreads();
    if (threadIdx.x == 0) {
        p.sums[2 * seg] = (red[0][0] + red[0][1]) + (red[0][2] + red[0][3]);
        p.sums[2 * seg + 1] = (red[1][0] + red[1][1]) + (red[1][2] + red[1][3]);
    }
}

// loss = -(0.5 s/n0 + q/(n0 s)),  s = |d|, q = <d0, d>;   d loss/d a = -((0.5/s - q/s^3) d + d0/s) / n0
__global__ void __launch_bounds__(256) ilaf_grad_kernel(const I2VIlafParams p) {
    const int n = blockIdx.y;
    const int fps = p.fps > 0 ? p.fps : p.N, seg = n / fps;
    if (p.mode == 1) {                                       // TAP feature distance (I2VIlafParams::mode)
        const double dist = sqrt(p.sums[2 * seg]);
        if (blockIdx.x == 0 && n == seg * fps && threadIdx.x == 0) p.loss_out[seg] = (float)dist;
        const double c = dist > 0.0 ? p.coef / dist : 0.0;
        const float* a = p.a + (int64_t)n * p.a_nstride;
        const float* o = p.ori + (int64_t)n * p.D;
        float* g = p.grad + (int64_t)n * p.grad_nstride;
        for (int64_t i = blockIdx.x * 256 + threadIdx.x; i < p.D; i += (int64_t)gridDim.x * 256) {
            const float av = a[i];
            float v = 0.f;
            if (av != 0.f && !(p.mask_relu && !(av > 0.f)))
                v = (float)(c * (double)__fsub_rn(tap_root(av), tap_root(o[i])) * 0.5 / (double)sqrtf(fabsf(av)));
            g[i] = p.accumulate ? g[i] + v : v;
        }
        return;
    }
    const double s = sqrt(p.sums[2 * seg]), q = p.sums[2 * seg + 1], n0 = p.init_sq ? sqrt(p.init_sq[seg]) : p.init_norm;
    if (blockIdx.x == 0 && n == seg * fps && threadIdx.x == 0) p.loss_out[seg] = (float)(-(0.5 * s / n0 + q / (n0 * s)));
    const double cd = -(0.5 / s - q / (s * s * s)) / n0, c0 = -1.0 / (s * n0);
    const float* a = p.a + (int64_t)n * p.a_nstride;
    const float* o = p.ori + (int64_t)n * p.D;
    const float* a0 = p.adv0 + (int64_t)n * p.D;
    float* g = p.grad + (int64_t)n * p.grad_nstride;
    for (int64_t i = blockIdx.x * 256 + threadIdx.x; i < p.D; i += (int64_t)gridDim.x * 256) {
        const float av = a[i], ov = o[i];
        float v = (float)(cd * (double)__fsub_rn(av, ov) + c0 * (double)__fsub_rn(a0[i], ov));
        if (p.mask_relu && !(av > 0.f)) v = 0.f;
        g[i] = p.accumulate ? g[i] + v : v;
    }
}

int k_ilaf_reduce(const I2VIlafParams& p, i2v_stream_t s) {
    hipLaunchKernelGGL(ilaf_reduce_kernel, dim3(p.nblk, p.N), dim3(256), 0, (hipStream_t)s, p);
    LAUNCH_CHECK("ilaf_reduce");
    hipLaunchKernelGGL(ilaf_finish_kernel, dim3(p.fps > 0 ? p.N / p.fps : 1), dim3(256), 0, (hipStream_t)s, p);
    LAUNCH_CHECK("ilaf_finish");
    return 0;
}

int k_ilaf_grad(const I2VIlafParams& p, i2v_stream_t s) {
    int gblk = (int)((p.D + 2047) / 2048); if (gblk > 64) gblk = 64;
    hipLaunchKernelGGL(ilaf_grad_kernel, dim3(gblk, p.N), dim3(256), 0, (hipStream_t)s, p);
    LAUNCH_CHECK("ilaf_grad");
    return 0;
}

// =============================================================================================
// classifier head: global average pool -> Linear -> softmax cross-entropy and its gradient (I2VHeadParams)
// =============================================================================================
// grid (C, clips): mean over the clip's T frames and HW pixels of one channel (double accumulation, fixed tree)
__global__ void __launch_bounds__(256) head_pool_kernel(const I2VHeadParams p) {
    const int c = blockIdx.x, clip = blockIdx.y;
    const int per = p.T * p.HW;
    double s = 0;
    for (int i = threadIdx.x; i < per; i += 256) {
        const int t = i / p.HW, px = i - t * p.HW;
        s += (double)p.a[((int64_t)clip * p.T + t) * p.a_nstride + (int64_t)c * p.HW + px];
    }
    __shared__ double red[4];
    s = wave_sum_d(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) p.pooled[(int64_t)clip * p.Ctot + p.c_off + c] = (float)(((red[0] + red[1]) + (red[2] + red[3])) / (double)per);
}

// grid (clips): logits, softmax, loss, d loss / d pooled
__global__ void __launch_bounds__(256) head_logits_kernel(const I2VHeadParams p) {
    const int clip = blockIdx.x;
    const float* x = p.pooled + (int64_t)clip * p.Ctot;
    float* lg = p.logits + (int64_t)clip * p.K;
    for (int k = threadIdx.x; k < p.K; k += 256) {
        double acc = p.bias ? (double)p.bias[k] : 0.0;
        const float* w = p.W + (int64_t)k * p.Ctot;
        for (int c = 0; c < p.Ctot; ++c) acc += (double)w[c] * (double)x[c];
        lg[k] = (float)acc;
    }
    __syncthreads();
    __shared__ double red[4]; __shared__ double mx_s, sum_s;
    double mx = -1e300;
    for (int k = threadIdx.x; k < p.K; k += 256) mx = fmax(mx, (double)lg[k]);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mx = fmax(mx, __shfl_down(mx, o, 64));
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = mx;
    __syncthreads();
    if (threadIdx.x == 0) mx_s = fmax(fmax(red[0], red[1]), fmax(red[2], red[3]));
    __syncthreads();
    double se = 0;
    for (int k = threadIdx.x; k < p.K; k += 256) se += exp((double)lg[k] - mx_s);
    se = wave_sum_d(se);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = se;
    __syncthreads();
    if (threadIdx.x == 0) {
        sum_s = (red[0] + red[1]) + (red[2] + red[3]);
        const int lab = p.labels[clip];
        p.loss_each[clip] = (float)(-((double)lg[lab] - mx_s - log(sum_s)));
    }
    __syncthreads();
    // d(scale * mean_clips loss) / d pooled[c] = scale/clips * sum_k W[k][c] (softmax_k - [k == label]); the division by the
    // feature's T*HW (the average pool's backward) happens where the gradient is spread, per feature
    const int lab = p.labels[clip];
    const double f = (double)p.scale / (double)p.clips;
    for (int c = threadIdx.x; c < p.Ctot; c += 256) {
        double acc = 0;
        for (int k = 0; k < p.K; ++k) {
            const double pk = exp((double)lg[k] - mx_s) / sum_s - (k == lab ? 1.0 : 0.0);
            acc += (double)p.W[(int64_t)k * p.Ctot + c] * pk;
        }
        p.dpooled[(int64_t)clip * p.Ctot + c] = (float)(f * acc);
    }
}

// grid (blocks, frames): the pooled gradient spread back over the frame's positions, gated by the feature's ReLU
__global__ void __launch_bounds__(256) head_grad_kernel(const I2VHeadParams p) {
    const int n = blockIdx.y, clip = n / p.T;
    const int64_t D = (int64_t)p.C * p.HW;
    const float* a = p.a + (int64_t)n * p.a_nstride;
    float* g = p.grad + (int64_t)n * p.grad_nstride;
    const float* dp = p.dpooled + (int64_t)clip * p.Ctot + p.c_off;
    const float cnt = (float)(p.T * p.HW);
    for (int64_t i = blockIdx.x * 256 + threadIdx.x; i < D; i += (int64_t)gridDim.x * 256) {
        float v = __fdiv_rn(dp[i / p.HW], cnt);
        if (p.mask_relu && !(a[i] > 0.f)) v = 0.f;
        g[i] = p.accumulate ? g[i] + v : v;
    }
}

int k_head_ce(const I2VHeadParams& p, i2v_stream_t s) {
    if (p.phase & 1) {
        hipLaunchKernelGGL(head_pool_kernel, dim3(p.C, p.clips), dim3(256), 0, (hipStream_t)s, p);
        LAUNCH_CHECK("head_pool");
    }
    if (p.phase & 2) {
        hipLaunchKernelGGL(head_logits_kernel, dim3(p.clips), dim3(256), 0, (hipStream_t)s, p);
        LAUNCH_CHECK("head_logits");
    }
    if (p.phase & 4) {
        const int64_t D = (int64_t)p.C * p.HW;
        int gblk = (int)((D + 2047) / 2048); if (gblk > 64) gblk = 64;
        hipLaunchKernelGGL(head_grad_kernel, dim3(gblk, p.clips * p.T), dim3(256), 0, (hipStream_t)s, p);
        LAUNCH_CHECK("head_grad");
    }
    return 0;
}

// =============================================================================================
// frame flatten + un-normalise, compose, Adam (+ compose backward), sign steps, AENS weights
// =============================================================================================
// decoded uint8 frames (b, t, h, w, 3) -> normalised clip (b, 3, t, h, w): ClipToTensor (/255) + Normalize
// ((x - mean)/std), the tail of the reference's loader (datasets.py:88-93), fused with the layout change
__global__ void clip_from_u8_kernel(const uint8_t* __restrict__ frames, float* __restrict__ video, int b, int t, int hw) {
    const int64_t total = (int64_t)b * 3 * t * hw;
    for (int64_t o = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; o < total; o += (int64_t)gridDim.x * blockDim.x) {
        const int i = o % hw; int64_t r = o / hw;
        const int ti = r % t; r /= t;
        const int c = r % 3; const int64_t bi = r / 3;
        const float v = __fdiv_rn((float)frames[((bi * t + ti) * hw + i) * 3 + c], 255.f);
        video[o] = __fdiv_rn(__fsub_rn(v, c_mean[c]), c_std[c]);
    }
}

// Decoded uint8 frames (b, t, H, W, 3) -> bilinear resize to (rh, rw) in OpenCV's 8-bit fixed-point arithmetic (what gluoncv's
// `video_transforms.Resize` runs on decord's numpy frames, datasets.py:88) -> centre crop (oh, ow) -> /255 -> (x - mean)/std ->
// clip layout (b, 3, t, oh, ow): the whole validation transform of the reference's loader (datasets.py:86-93) in one pass over the
// pixels that survive the crop.  xtab / ytab hold (source index, weight of it, weight of the next one; weights in 1/2048) per
// RESIZED column / row, built on the host exactly as cv::resize builds them.
//   horizontal: S = src[sx]*a0 + src[sx+1]*a1                     (int, <= 255*2048)
//   vertical:   d = (((b0 * (S0 >> 4)) >> 16) + ((b1 * (S1 >> 4)) >> 16) + 2) >> 2          (cv::VResizeLinear<uchar, int, short>)
__global__ void clip_resize_crop_kernel(const uint8_t* __restrict__ frames, float* __restrict__ video, const int32_t* __restrict__ xtab,
                                        const int32_t* __restrict__ ytab, int b, int t, int H, int W, int cy, int cx, int oh, int ow) {
    const int64_t total = (int64_t)b * t * oh * ow;
    for (int64_t o = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; o < total; o += (int64_t)gridDim.x * blockDim.x) {
        const int x = o % ow; int64_t r = o / ow;
        const int y = r % oh; r /= oh;
        const int ti = r % t; const int64_t bi = r / t;
        const int32_t* xe = xtab + 3 * (x + cx); const int32_t* ye = ytab + 3 * (y + cy);
        const int sx0 = xe[0], a0 = xe[1], a1 = xe[2], sy0 = ye[0], b0 = ye[1], b1 = ye[2];
        const int sx1 = min(sx0 + 1, W - 1), sy1 = min(sy0 + 1, H - 1);
        const uint8_t* f = frames + (bi * t + ti) * (int64_t)H * W * 3;
        const uint8_t* r0 = f + (int64_t)sy0 * W * 3; const uint8_t* r1 = f + (int64_t)sy1 * W * 3;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const int S0 = r0[sx0 * 3 + c] * a0 + r0[sx1 * 3 + c] * a1;
            const int S1 = r1[sx0 * 3 + c] * a0 + r1[sx1 * 3 + c] * a1;
            const int d = (((b0 * (S0 >> 4)) >> 16) + ((b1 * (S1 >> 4)) >> 16) + 2) >> 2;
            const float v = __fdiv_rn((float)d, 255.f);
            video[(((bi * 3 + c) * t + ti) * oh + y) * (int64_t)ow + x] = __fdiv_rn(__fsub_rn(v, c_mean[c]), c_std[c]);
        }
    }
}

// The UCF-101 loader's validation transform (dataset_ucf101.py:113-126) for one output element: Pillow's antialiased BILINEAR
// resample (libImaging/Resample.c, 8-bit path) restricted to the crop window -- horizontal pass over the rows the vertical pass
// needs, each result rounded and clipped to 8 bits as Pillow stores its intermediate image, then the vertical pass, rounded
// and clipped again -- followed by ToTensor (/255), Normalize and the (b,3,t,h,w) layout.  Taps and 22-bit fixed-point
// coefficients per RESIZED column / row are built on the host as `precompute_coeffs` / `normalize_coeffs_8bpc` build them.
__global__ void clip_resample_crop_kernel(const uint8_t* __restrict__ frames, float* __restrict__ video, const int32_t* __restrict__ xb,
                                          const int32_t* __restrict__ xk, int kx, const int32_t* __restrict__ yb, const int32_t* __restrict__ yk,
                                          int ky, int b, int t, int H, int W, int cy, int cx, int oh, int ow) {
    const int64_t total = (int64_t)b * t * oh * ow;
    for (int64_t o = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; o < total; o += (int64_t)gridDim.x * blockDim.x) {
        const int x = o % ow; int64_t r = o / ow;
        const int y = r % oh; r /= oh;
        const int ti = r % t; const int64_t bi = r / t;
        const int x0 = xb[2 * (x + cx)], nx = xb[2 * (x + cx) + 1], y0 = yb[2 * (y + cy)], ny = yb[2 * (y + cy) + 1];
        const int32_t* kxr = xk + (int64_t)(x + cx) * kx; const int32_t* kyr = yk + (int64_t)(y + cy) * ky;
        const uint8_t* f = frames + (bi * t + ti) * (int64_t)H * W * 3;
        int v0 = 1 << 21, v1 = 1 << 21, v2 = 1 << 21;
        for (int j = 0; j < ny; ++j) {
            const uint8_t* row = f + ((int64_t)(y0 + j) * W + x0) * 3;
            int h0 = 1 << 21, h1 = 1 << 21, h2 = 1 << 21;
            for (int i = 0; i < nx; ++i) { const int k = kxr[i]; h0 += row[3 * i] * k; h1 += row[3 * i + 1] * k; h2 += row[3 * i + 2] * k; }
            const int k = kyr[j];
            v0 += min(max(h0 >> 22, 0), 255) * k; v1 += min(max(h1 >> 22, 0), 255) * k; v2 += min(max(h2 >> 22, 0), 255) * k;
        }
        const int d[3] = {min(max(v0 >> 22, 0), 255), min(max(v1 >> 22, 0), 255), min(max(v2 >> 22, 0), 255)};
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const float v = __fdiv_rn((float)d[c], 255.f);
            video[(((bi * 3 + c) * t + ti) * oh + y) * (int64_t)ow + x] = __fdiv_rn(__fsub_rn(v, c_mean[c]), c_std[c]);
        }
    }
}

__global__ void frames_from_video_kernel(const float* __restrict__ video, float* __restrict__ x, float* __restrict__ u,
                                         int b, int f, int hw) {
    const int64_t total = (int64_t)b * 3 * f * hw;
    for (int64_t o = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; o < total; o += (int64_t)gridDim.x * blockDim.x) {
        // o indexes the FRAME layout (b, f, 3, hw)
        const int i = o % hw; int64_t r = o / hw;
        const int c = r % 3; r /= 3;
        const int fi = r % f; const int64_t bi = r / f;
        const float v = video[((bi * 3 + c) * f + fi) * hw + i];
        x[o] = v;
        u[o] = __fadd_rn(__fmul_rn(v, c_std[c]), c_mean[c]);       // mul_ then add_: two roundings
    }
}

__global__ void compose_kernel(const float* __restrict__ u, const float* __restrict__ d, float* __restrict__ x,
                               int b, int f, int hw, float eps, int video_layout) {
    const int64_t total = (int64_t)b * 3 * f * hw;
    for (int64_t o = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; o < total; o += (int64_t)gridDim.x * blockDim.x) {
        const int i = o % hw; int64_t r = o / hw;
        const int c = r % 3; r /= 3;
        const int fi = r % f; const int64_t bi = r / f;
        const float dc = fminf(fmaxf(d[o], -eps), eps);
        const float s = u[o] + dc;
        const float xi = fminf(fmaxf(s, 0.f), 1.f);
        const float v = __fdiv_rn(__fsub_rn(xi, c_mean[c]), c_std[c]);
        const int64_t oo = video_layout ? ((bi * 3 + c) * f + fi) * hw + i : o;
        x[oo] = v;
    }
}

__global__ void adam_kernel(float* __restrict__ delta, float* __restrict__ m, float* __restrict__ v,
                            const float* __restrict__ gx, const float* __restrict__ u, int64_t n, int hw, float eps,
                            float step_size, float bc2_sqrt, float w1, float beta2, float w2, float adam_eps) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const int c = (int)((i / hw) % 3);
        const float d = delta[i];
        const float dc = fminf(fmaxf(d, -eps), eps);
        const float s = u[i] + dc;
        const bool pass = d >= -eps && d <= eps && s >= 0.f && s <= 1.f;     // inclusive clamp masks
        const float g = pass ? __fdiv_rn(gx[i], c_std[c]) : 0.f;
        const float mm = fmaf(w1, __fsub_rn(g, m[i]), m[i]);                  // lerp_(g, 1-b1)
        const float vv = __fadd_rn(__fmul_rn(v[i], beta2), __fmul_rn(__fmul_rn(w2, g), g));   // mul_, addcmul_
        // sqrtf, not __fsqrt_rn: on ROCm 7 the intrinsic is NOT correctly rounded for small arguments (166 290 of 2^20
        // values in [1e-13, 1e-11] differ from the IEEE result), sqrtf is (hipcc's default correctly-rounded divide/sqrt)
        const float den = __fadd_rn(__fdiv_rn(sqrtf(vv), bc2_sqrt), adam_eps);
        delta[i] = __fadd_rn(d, __fmul_rn(-step_size, __fdiv_rn(mm, den)));   // addcdiv_
        m[i] = mm; v[i] = vv;
    }
}

__global__ void sign_bim_kernel(float* __restrict__ adv, const float* __restrict__ u, const float* __restrict__ grad,
                                int64_t n, int64_t cs, float step, float eps) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const int c = (int)((i / cs) % 3);
        float a = __fadd_rn(__fmul_rn(adv[i], c_std[c]), c_mean[c]);
        const float g = grad[i];
        const float sg = g > 0.f ? 1.f : (g < 0.f ? -1.f : 0.f);
        a = __fadd_rn(a, __fmul_rn(step, sg));
        const float d = fminf(fmaxf(__fsub_rn(a, u[i]), -eps), eps);
        const float r = fminf(fmaxf(__fadd_rn(u[i], d), 0.f), 1.f);
        adv[i] = __fdiv_rn(__fsub_rn(r, c_mean[c]), c_std[c]);
    }
}

__global__ void sign_delta_kernel(float* __restrict__ delta, const float* __restrict__ grad, int64_t n, float step) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const float g = grad[i];
        delta[i] = __fsub_rn(delta[i], __fmul_rn(step, g > 0.f ? 1.f : (g < 0.f ? -1.f : 0.f)));
    }
}

// ILAF update from the gradient w.r.t. the composed frames: the compose backward only gates (inclusive clamp
// masks) and scales by 1/std > 0, so sign(d cost / d delta) = pass ? sign(gx) : 0   (image_attacks.py:589-617)
__global__ void sign_delta_gx_kernel(float* __restrict__ delta, const float* __restrict__ gx, const float* __restrict__ u,
                                     int64_t n, float eps, float step) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const float d = delta[i];
        const float s = u[i] + fminf(fmaxf(d, -eps), eps);
        const bool pass = d >= -eps && d <= eps && s >= 0.f && s <= 1.f;
        const float g = pass ? gx[i] : 0.f;
        delta[i] = __fsub_rn(d, __fmul_rn(step, g > 0.f ? 1.f : (g < 0.f ? -1.f : 0.f)));
    }
}

__global__ void aens_coeffs_kernel(const float* prev, float* coeffs, float momentum, int L) {
    // one wave: softmax(softmax(prev) + momentum*coeffs)
    const int l = threadIdx.x;
    float pv = l < L ? prev[l] : -INFINITY;
    float mx = pv;
    for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
    float e = l < L ? expf(pv - mx) : 0.f;
    float sum = e;
    for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o, 64);
    float b = l < L ? e / sum + momentum * coeffs[l] : -INFINITY;
    mx = b;
    for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
    e = l < L ? expf(b - mx) : 0.f;
    sum = e;
    for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o, 64);
    if (l < L) coeffs[l] = e / sum;
}

__global__ void aens_reduce_kernel(const float* cosv, const float* coeffs, int L, int frames, float* feat_sum, float* weighted) {
    const int l = blockIdx.x;
    double s = 0;
    for (int n = threadIdx.x; n < frames; n += 64) s += cosv[(int64_t)l * frames + n];
    s = wave_sum_d(s);
    if (threadIdx.x == 0) { feat_sum[l] = (float)s; weighted[l] = coeffs[l] * (float)s; }
}

int k_clip_from_u8(const uint8_t* frames, float* video, int b, int t, int h, int w, i2v_stream_t s) {
    const int64_t total = (int64_t)b * 3 * t * h * w;
    hipLaunchKernelGGL(clip_from_u8_kernel, dim3(stream_grid(total, 1024)), dim3(256), 0, (hipStream_t)s, frames, video, b, t, h * w);
    LAUNCH_CHECK("clip_from_u8"); return 0;
}
int k_clip_resize_crop(const uint8_t* frames, float* video, const int32_t* xtab, const int32_t* ytab, int b, int t, int H, int W,
                       int cy, int cx, int oh, int ow, i2v_stream_t s) {
    const int64_t total = (int64_t)b * t * oh * ow;
    hipLaunchKernelGGL(clip_resize_crop_kernel, dim3(stream_grid(total, 256)), dim3(256), 0, (hipStream_t)s, frames, video, xtab, ytab,
                       b, t, H, W, cy, cx, oh, ow);
    LAUNCH_CHECK("clip_resize_crop"); return 0;
}
int k_clip_resample_crop(const uint8_t* frames, float* video, const int32_t* xb, const int32_t* xk, int kx, const int32_t* yb, const int32_t* yk,
                         int ky, int b, int t, int H, int W, int cy, int cx, int oh, int ow, i2v_stream_t s) {
    const int64_t total = (int64_t)b * t * oh * ow;
    hipLaunchKernelGGL(clip_resample_crop_kernel, dim3(stream_grid(total, 256)), dim3(256), 0, (hipStream_t)s, frames, video, xb, xk, kx, yb, yk, ky,
                       b, t, H, W, cy, cx, oh, ow);
    LAUNCH_CHECK("clip_resample_crop"); return 0;
}
int k_frames_from_video(const float* video, float* x, float* u, int b, int f, int h, int w, i2v_stream_t s) {
    const int64_t total = (int64_t)b * 3 * f * h * w;
    hipLaunchKernelGGL(frames_from_video_kernel, dim3(stream_grid(total, 1024)), dim3(256), 0, (hipStream_t)s, video, x, u, b, f, h * w);
    LAUNCH_CHECK("frames_from_video"); return 0;
}
int k_compose(const float* u, const float* delta, float* x, int b, int f, int h, int w, float eps, int video_layout, i2v_stream_t s) {
    const int64_t total = (int64_t)b * 3 * f * h * w;
    hipLaunchKernelGGL(compose_kernel, dim3(stream_grid(total, 1024)), dim3(256), 0, (hipStream_t)s, u, delta, x, b, f, h * w, eps, video_layout);
    LAUNCH_CHECK("compose"); return 0;
}
int k_adam(float* delta, float* m, float* v, const float* gx, const float* u, int64_t n, int hw, float eps,
           float step_size, float bc2_sqrt, float w1, float beta2, float w2, float adam_eps, i2v_stream_t s) {
    hipLaunchKernelGGL(adam_kernel, dim3(stream_grid(n, 1024)), dim3(256), 0, (hipStream_t)s, delta, m, v, gx, u, n, hw, eps,
                       step_size, bc2_sqrt, w1, beta2, w2, adam_eps);
    LAUNCH_CHECK("adam"); return 0;
}
int k_sign_bim(float* adv, const float* u, const float* grad, int64_t n, int64_t cs, float step, float eps, i2v_stream_t s) {
    hipLaunchKernelGGL(sign_bim_kernel, dim3(stream_grid(n, 1024)), dim3(256), 0, (hipStream_t)s, adv, u, grad, n, cs, step, eps);
    LAUNCH_CHECK("sign_bim"); return 0;
}
int k_sign_delta(float* delta, const float* grad, int64_t n, float step, i2v_stream_t s) {
    hipLaunchKernelGGL(sign_delta_kernel, dim3(stream_grid(n, 1024)), dim3(256), 0, (hipStream_t)s, delta, grad, n, step);
    LAUNCH_CHECK("sign_delta"); return 0;
}
int k_sign_delta_gx(float* delta, const float* gx, const float* u, int64_t n, float eps, float step, i2v_stream_t s) {
    hipLaunchKernelGGL(sign_delta_gx_kernel, dim3(stream_grid(n, 1024)), dim3(256), 0, (hipStream_t)s, delta, gx, u, n, eps, step);
    LAUNCH_CHECK("sign_delta_gx"); return 0;
}
// Temporal-translation gradient augmentation (video_attacks.py:160-175): grads (D, NC, T, HW) are the input gradients of D
// cyclically frame-shifted copies of a clip; out = (1-w) * sum_d k[d] g_d  +  w * sum_d k[d] roll(g_d, -move_d along T), the
// two sums as fmaf chains over d in order (the reference's 1 x D matmul), then two products and one addition as torch forms them.
struct TTMix { float k[64]; int move[64]; };
__global__ void __launch_bounds__(256) tt_grad_mix_kernel(const float* __restrict__ g, float* __restrict__ out, const TTMix m, const int D,
                                                          const int64_t M, const int T, const int HW, const float w1, const float w) {
    const int64_t per = (int64_t)T * HW;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < M; i += (int64_t)gridDim.x * 256) {
        const int64_t nc = i / per; const int r = (int)(i - nc * per); const int t = r / HW, x = r - t * HW;
        float s = 0.f, d = 0.f;
        for (int k = 0; k < D; ++k) {
            const float* gk = g + (int64_t)k * M + nc * per;
            int ts = (t + m.move[k]) % T; if (ts < 0) ts += T;
            s = fmaf(m.k[k], gk[r], s);
            d = fmaf(m.k[k], gk[(int64_t)ts * HW + x], d);
        }
        out[i] = __fadd_rn(__fmul_rn(w1, s), __fmul_rn(w, d));
    }
}
int k_tt_grad_mix(const float* grads, float* out, const float* kern, const int* moves, int D, int64_t NC, int T, int HW, float w1, float w,
                  i2v_stream_t s) {
    TTMix m; for (int k = 0; k < D; ++k) { m.k[k] = kern[k]; m.move[k] = moves[k]; }
    const int64_t M = NC * T * HW;
    hipLaunchKernelGGL(tt_grad_mix_kernel, dim3(stream_grid(M, 1024)), dim3(256), 0, (hipStream_t)s, grads, out, m, D, M, T, HW, w1, w);
    LAUNCH_CHECK("tt_grad_mix"); return 0;
}
// =============================================================================================
// Non-local block core (gluoncv `i3d_nl5_*`; I2VAttnGemm / I2VSoftmaxRows): three product forms between frame-major activation
// views and a dense per-clip matrix, and the row softmax / its backward.  64 x 64 output tiles, 4 waves of 32 x 32 on
// v_mfma_f32_32x32x2_f32, K in chunks of 32 through double-buffered LDS in the canonical [k][m] image (operands whose K axis is
// the contiguous one are transposed while they are written), register-staged prefetch.  Every output element is ONE k-ordered fmaf
// chain computed by one block: no split K, no atomics.  The blocks cost ~15 % of the FLOPs of the stage they sit in, so the
// kernel is kept simple (plain loads, no DMA staging).
// =============================================================================================
__device__ __forceinline__ const float* act_addr(const I2VActMat& a, int clip, int c, int pos) {
    const int t = pos / a.HW, r = pos - t * a.HW;
    return a.p + ((int64_t)clip * a.T + t) * a.nstride + (int64_t)c * a.HW + r;
}
// 4 consecutive elements along the contiguous axis of an operand, zero beyond `lim` (elements left on that axis)
__device__ __forceinline__ float4 load4_guard(const float* p, int lim, bool vec_ok) {
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (lim >= 4 && vec_ok) return *reinterpret_cast<const float4*>(p);
    if (lim > 0) v.x = p[0];
    if (lim > 1) v.y = p[1];
    if (lim > 2) v.z = p[2];
    if (lim > 3) v.w = p[3];
    return v;
}
// the same along the POSITION axis of a frame-major activation view: positions are contiguous inside a frame only, so without
// the vector path (HW % 4 == 0 keeps an aligned group of four inside one frame) every element takes its own address
__device__ __forceinline__ float4 load4_act(const I2VActMat& a, int clip, int c, int pos, int lim, bool vec_ok) {
    if (lim >= 4 && vec_ok) return *reinterpret_cast<const float4*>(act_addr(a, clip, c, pos));
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (lim > 0) v.x = *act_addr(a, clip, c, pos);
    if (lim > 1) v.y = *act_addr(a, clip, c, pos + 1);
    if (lim > 2) v.z = *act_addr(a, clip, c, pos + 2);
    if (lim > 3) v.w = *act_addr(a, clip, c, pos + 3);
    return v;
}
template <int FORM>
__global__ void __launch_bounds__(256) attn_gemm_kernel(const I2VAttnGemm p) {
    // LDS images [k][m]; an operand whose K axis is the contiguous one in memory is transposed while it is written: row stride 66
    // (66 % 32 = 2: the four k-quads x eight rows of a 32-lane write group land on 32 different banks; with 64 they were 4-way
    // conflicts that kept the LDS busier than the matrix pipe), 64 (16-byte rows for ds_write_b128) for the K-major ones
    constexpr int KC = 32, LS = FORM == 1 ? 64 : 66, RS = FORM == 2 ? 66 : 64;
    __shared__ __attribute__((aligned(16))) float Ls[2][KC][LS], Rs[2][KC][RS];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6, wd = wave >> 1, wp = wave & 1, l31 = lane & 31, lk = lane >> 5;
    // output tile: rows m0.. (form 1: i; forms 2, 3: channel), columns n0.. (form 1: j; form 2: i; form 3: j), reduction K
    const int ROWS = FORM == 1 ? p.M : p.Cc, COLS = FORM == 2 ? p.M : p.N, KFULL = FORM == 1 ? p.Cc : (FORM == 2 ? p.N : p.M);
    // this block's K segment [KBEG, K)
    const int split = (FORM != 1 && p.ksplit > 1) ? p.ksplit : 1, kseg = attn_kseg(KFULL, split);
    const int KBEG = min((int)blockIdx.z * kseg, KFULL), K = min(KBEG + kseg, KFULL);
    const int tiles_n = (COLS + 63) / 64;
    const int clip = blockIdx.y, m0 = (blockIdx.x / tiles_n) * 64, n0 = (blockIdx.x % tiles_n) * 64;
    const float* Dn = p.Din ? p.Din + (int64_t)clip * p.M * p.N : nullptr;
    const bool a_vec = (p.A.HW % 4 == 0) && (p.A.nstride % 4 == 0) && (((uintptr_t)p.A.p & 15) == 0);
    const bool b_vec = FORM == 1 && (p.B.HW % 4 == 0) && (p.B.nstride % 4 == 0) && (((uintptr_t)p.B.p & 15) == 0);
    const bool d_vec = FORM != 1 && (p.N % 4 == 0) && (((uintptr_t)p.Din & 15) == 0);
    // thread's share of a chunk, two pieces h = 0, 1.  K-major operands: row k = t / 16 + 16 h, 4 columns from (t % 16) * 4;
    // M-major: row m = t / 4, 4 k from (t % 4) * 4 + 16 h
    const int kk = t >> 4, c4 = (t & 15) * 4, mm = t >> 2, k4 = (t & 3) * 4;
    const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
    float4 ra[2], rb[2];
    auto fetch = [&](const int k0) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            if constexpr (FORM == 1) {                 // Lhs[k=c][m=i] = A(c, i), Rhs[k=c][n=j] = B(c, j): both K-major
                const int c = k0 + kk + 16 * h;
                ra[h] = (c < K && m0 + c4 < ROWS) ? load4_act(p.A, clip, c, m0 + c4, ROWS - (m0 + c4), a_vec) : zero4;
                rb[h] = (c < K && n0 + c4 < COLS) ? load4_act(p.B, clip, c, n0 + c4, COLS - (n0 + c4), b_vec) : zero4;
            } else if constexpr (FORM == 2) {          // Lhs[k=j][m=c] = A(c, j) (M-major), Rhs[k=j][n=i] = D[i][j] (M-major)
                const int c = m0 + mm, j = k0 + k4 + 16 * h, i = n0 + mm;
                ra[h] = (c < ROWS && j < K) ? load4_act(p.A, clip, c, j, K - j, a_vec) : zero4;
                rb[h] = (i < COLS && j < K) ? load4_guard(Dn + (int64_t)i * p.N + j, K - j, d_vec) : zero4;
            } else {                                   // Lhs[k=i][m=c] = A(c, i) (M-major), Rhs[k=i][n=j] = D[i][j] (K-major)
                const int c = m0 + mm, i = k0 + k4 + 16 * h, ik = k0 + kk + 16 * h;
                ra[h] = (c < ROWS && i < K) ? load4_act(p.A, clip, c, i, K - i, a_vec) : zero4;
                rb[h] = (ik < K && n0 + c4 < COLS) ? load4_guard(Dn + (int64_t)ik * p.N + n0 + c4, COLS - (n0 + c4), d_vec) : zero4;
            }
        }
    };
    auto stash = [&](const int buf) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int kr = kk + 16 * h, kq = k4 + 16 * h;
            if constexpr (FORM == 1) *reinterpret_cast<float4*>(&Ls[buf][kr][c4]) = ra[h];
            else { Ls[buf][kq][mm] = ra[h].x; Ls[buf][kq + 1][mm] = ra[h].y; Ls[buf][kq + 2][mm] = ra[h].z; Ls[buf][kq + 3][mm] = ra[h].w; }
            if constexpr (FORM == 2) { Rs[buf][kq][mm] = rb[h].x; Rs[buf][kq + 1][mm] = rb[h].y; Rs[buf][kq + 2][mm] = rb[h].z; Rs[buf][kq + 3][mm] = rb[h].w; }
            else *reinterpret_cast<float4*>(&Rs[buf][kr][c4]) = rb[h];
        }
    };
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    const int nchunks = (K - KBEG + KC - 1) / KC;
    fetch(KBEG); stash(0);
    __syncthreads();
    for (int c = 0; c < nchunks; ++c) {
        const int buf = c & 1;
        if (c + 1 < nchunks) fetch(KBEG + (c + 1) * KC);
#pragma unroll
        for (int s2 = 0; s2 < KC / 2; ++s2)
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(Ls[buf][2 * s2 + lk][wd * 32 + l31], Rs[buf][2 * s2 + lk][wp * 32 + l31], acc, 0, 0, 0);
        if (c + 1 < nchunks) stash(buf ^ 1);
        __syncthreads();
    }
    // D layout: lane -> column wp*32 + l31, register r -> row wd*32 + (r & 3) + 8 (r >> 2) + 4 lk
    const int col = n0 + wp * 32 + l31;
    if (col >= COLS) return;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int row = m0 + wd * 32 + (r & 3) + 8 * (r >> 2) + 4 * lk;
        if (row >= ROWS) continue;
        if constexpr (FORM == 1) p.D[((int64_t)clip * p.M + row) * p.N + col] = p.scale == 1.f ? acc[r] : __fmul_rn(p.scale, acc[r]);
        else if (split > 1) p.part[(((int64_t)clip * split + blockIdx.z) * ROWS + row) * COLS + col] = acc[r];
        else {
            const int tt = col / p.C_HW, rr = col - tt * p.C_HW;
            float* o = p.Cact + ((int64_t)clip * p.C_T + tt) * p.C_nstride + (int64_t)row * p.C_HW + rr;
            *o = p.accumulate ? __fadd_rn(*o, acc[r]) : acc[r];
        }
    }
}
// the segment sums of a K-split launch, added in segment order
__global__ void __launch_bounds__(256) attn_split_reduce_kernel(const I2VAttnGemm p, const int COLS) {
    const int64_t per = (int64_t)p.Cc * COLS, total = per * p.clips;
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
        const int clip = (int)(e / per); const int64_t w = e - (int64_t)clip * per;
        const int row = (int)(w / COLS), col = (int)(w - (int64_t)row * COLS);
        const float* q = p.part + (int64_t)clip * p.ksplit * per + w;
        float sum = q[0];
        for (int z = 1; z < p.ksplit; ++z) sum = __fadd_rn(sum, q[(int64_t)z * per]);
        const int tt = col / p.C_HW, rr = col - tt * p.C_HW;
        float* o = p.Cact + ((int64_t)clip * p.C_T + tt) * p.C_nstride + (int64_t)row * p.C_HW + rr;
        *o = p.accumulate ? __fadd_rn(*o, sum) : sum;
    }
}
int k_attn_gemm(const I2VAttnGemm& p, i2v_stream_t s) {
    const int rows = p.form == 1 ? p.M : p.Cc, cols = p.form == 2 ? p.M : p.N;
    if (rows <= 0 || cols <= 0 || p.clips <= 0) return 0;
    const int split = (p.form != 1 && p.ksplit > 1) ? p.ksplit : 1;
    if (split > 1 && !p.part) return pool_fail("attn_gemm: a K-split launch needs its scratch");
    const dim3 grid((unsigned)(((rows + 63) / 64) * ((cols + 63) / 64)), (unsigned)p.clips, (unsigned)split);
    if (p.form == 1) hipLaunchKernelGGL((attn_gemm_kernel<1>), grid, dim3(256), 0, (hipStream_t)s, p);
    else if (p.form == 2) hipLaunchKernelGGL((attn_gemm_kernel<2>), grid, dim3(256), 0, (hipStream_t)s, p);
    else hipLaunchKernelGGL((attn_gemm_kernel<3>), grid, dim3(256), 0, (hipStream_t)s, p);
    LAUNCH_CHECK("attn_gemm");
    if (split > 1) {
        hipLaunchKernelGGL(attn_split_reduce_kernel, dim3(stream_grid((int64_t)p.clips * p.Cc * cols, 1024)), dim3(256), 0, (hipStream_t)s, p, cols);
        LAUNCH_CHECK("attn_split_reduce");
    }
    return 0;
}
// one block per row; thread t owns columns t, t + 256, ...; reductions: wave shuffles, then the four wave values in order
__global__ void __launch_bounds__(256) softmax_rows_kernel(const I2VSoftmaxRows p) {
    __shared__ float red[4];
    __shared__ float bc;
    float* x = p.X + (int64_t)blockIdx.x * p.N;
    const int t = threadIdx.x;
    if (p.mode == 0) {
        float m = -INFINITY;
        for (int j = t; j < p.N; j += 256) m = fmaxf(m, x[j]);
        for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_down(m, o));
        if ((t & 63) == 0) red[t >> 6] = m;
        __syncthreads();
        if (t == 0) bc = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
        __syncthreads();
        m = bc;
        float sum = 0.f;
        for (int j = t; j < p.N; j += 256) { const float e = expf(__fsub_rn(x[j], m)); x[j] = e; sum = __fadd_rn(sum, e); }
        sum = wave_sum(sum);
        __syncthreads();
        if ((t & 63) == 0) red[t >> 6] = sum;
        __syncthreads();
        if (t == 0) bc = __fadd_rn(__fadd_rn(red[0], red[1]), __fadd_rn(red[2], red[3]));
        __syncthreads();
        const float tot = bc;
        for (int j = t; j < p.N; j += 256) x[j] = __fdiv_rn(x[j], tot);
    } else {
        const float* P = p.P + (int64_t)blockIdx.x * p.N;
        float dot = 0.f;
        for (int j = t; j < p.N; j += 256) dot = __fadd_rn(dot, __fmul_rn(x[j], P[j]));
        dot = wave_sum(dot);
        if ((t & 63) == 0) red[t >> 6] = dot;
        __syncthreads();
        if (t == 0) bc = __fadd_rn(__fadd_rn(red[0], red[1]), __fadd_rn(red[2], red[3]));
        __syncthreads();
        dot = bc;
        for (int j = t; j < p.N; j += 256) x[j] = __fmul_rn(P[j], __fsub_rn(x[j], dot));
    }
}
int k_softmax_rows(const I2VSoftmaxRows& p, i2v_stream_t s) {
    if (p.rows <= 0) return 0;
    hipLaunchKernelGGL(softmax_rows_kernel, dim3((unsigned)p.rows), dim3(256), 0, (hipStream_t)s, p);
    LAUNCH_CHECK("softmax_rows"); return 0;
}
// =============================================================================================
// base_attacks.py transforms: DI-FGSM's input diversity (:357-376) = nearest resize -> zero pad -> nearest resize, which composes
// into ONE index map per axis (map < 0: padding); its gradient gathers over the (contiguous: the maps are monotone) ranges of output
// positions that read a source position.  TI-FGSM / TI-FGSM-3D (:412-441, :613-651) smooth the gradient with a Gaussian that is
// an outer product of one 1-D kernel, i.e. one depthwise 1-D pass per axis.  HBM-bound streaming kernels.
// =============================================================================================
__global__ void __launch_bounds__(256) resample_nearest_kernel(const float* __restrict__ src, float* __restrict__ dst, const int64_t total, const int Hs,
                                                               const int Ws, const int Hd, const int Wd, const int* __restrict__ my, const int* __restrict__ mx) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int x = (int)(i % Wd); const int64_t r = i / Wd; const int y = (int)(r % Hd); const int64_t pl = r / Hd;
        const int sy = my[y], sx = mx[x];
        dst[i] = (sy >= 0 && sx >= 0) ? src[(pl * Hs + sy) * Ws + sx] : 0.f;
    }
}
__global__ void __launch_bounds__(256) resample_nearest_bwd_kernel(const float* __restrict__ g, float* __restrict__ gs, const int64_t total, const int Hd,
                                                                   const int Wd, const int Hs, const int Ws, const int* __restrict__ ylo, const int* __restrict__ yhi,
                                                                   const int* __restrict__ xlo, const int* __restrict__ xhi) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int sx = (int)(i % Ws); const int64_t r = i / Ws; const int sy = (int)(r % Hs); const int64_t pl = r / Hs;
        float acc = 0.f;
        for (int y = ylo[sy]; y < yhi[sy]; ++y)
            for (int x = xlo[sx]; x < xhi[sx]; ++x) acc = __fadd_rn(acc, g[(pl * Hd + y) * Wd + x]);
        gs[i] = acc;
    }
}
struct DwTaps { float t[64]; };
__global__ void __launch_bounds__(256) dwconv1d_kernel(const float* __restrict__ src, float* __restrict__ dst, const int64_t total, const int len,
                                                       const int64_t inner, const DwTaps taps, const int k) {
    const int half = k / 2;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int64_t q = i / inner; const int pos = (int)(q % len);
        const float* base = src + (i - (int64_t)pos * inner);
        float acc = 0.f;
        for (int t = 0; t < k; ++t) {           // zero padding: taps outside the axis contribute nothing
            const int pp = pos + t - half;
            if (pp >= 0 && pp < len) acc = __fadd_rn(acc, __fmul_rn(taps.t[t], base[(int64_t)pp * inner]));
        }
        dst[i] = acc;
    }
}
int k_resample_nearest(const float* src, float* dst, int64_t planes, int Hs, int Ws, int Hd, int Wd, const int32_t* map_y, const int32_t* map_x,
                       i2v_stream_t s) {
    const int64_t total = planes * Hd * Wd;
    hipLaunchKernelGGL(resample_nearest_kernel, dim3(stream_grid(total, 1024)), dim3(256), 0, (hipStream_t)s, src, dst, total, Hs, Ws, Hd, Wd, map_y, map_x);
    LAUNCH_CHECK("resample_nearest"); return 0;
}
int k_resample_nearest_bwd(const float* g, float* gsrc, int64_t planes, int Hd, int Wd, int Hs, int Ws, const int32_t* ylo, const int32_t* yhi,
                           const int32_t* xlo, const int32_t* xhi, i2v_stream_t s) {
    const int64_t total = planes * Hs * Ws;
    hipLaunchKernelGGL(resample_nearest_bwd_kernel, dim3(stream_grid(total, 1024)), dim3(256), 0, (hipStream_t)s, g, gsrc, total, Hd, Wd, Hs, Ws, ylo, yhi, xlo, xhi);
    LAUNCH_CHECK("resample_nearest_bwd"); return 0;
}
int k_dwconv1d(const float* src, float* dst, int64_t outer, int len, int64_t inner, const float* taps, int k, i2v_stream_t s) {
    DwTaps t; for (int i = 0; i < 64; ++i) t.t[i] = i < k ? taps[i] : 0.f;
    const int64_t total = outer * len * inner;
    hipLaunchKernelGGL(dwconv1d_kernel, dim3(stream_grid(total, 1024)), dim3(256), 0, (hipStream_t)s, src, dst, total, len, inner, t, k);
    LAUNCH_CHECK("dwconv1d"); return 0;
}
// ---- gradient post-processing of the sign-step family (i2v_grad_post_f32): mean-abs / L1 normalisation, momentum, layout -------------
// Element e of group q sits at base(q) + (e / inner) * outer_stride + (e % inner) in the clip layout; `fm` maps a clip-layout offset to
// the frame-major gradient the backbone wrote.
struct GradPost {
    const float* g; float* mom; float* out; double* partial;
    int B, C, F, H, W, fm, mode, splits; float decay;
    int64_t ge;
};
__device__ __forceinline__ int64_t gp_src(const GradPost& p, int64_t o) {           // clip-layout offset -> offset in `g`
    if (!p.fm) return o;
    const int HW = p.H * p.W; const int i = (int)(o % HW); int64_t r = o / HW;
    const int f = (int)(r % p.F); r /= p.F; const int c = (int)(r % p.C); const int64_t b = r / p.C;
    return ((b * p.F + f) * p.C + c) * (int64_t)HW + i;
}
__device__ __forceinline__ int64_t gp_elem(const GradPost& p, int q, int64_t e) {   // element e of group q -> clip-layout offset
    const int64_t HW = (int64_t)p.H * p.W, CFHW = (int64_t)p.C * p.F * HW;
    switch (p.mode) {
        case 1: { const int b = q / p.F, f = q - b * p.F; return b * CFHW + (e / HW) * (p.F * HW) + f * HW + e % HW; }
        case 2: return (int64_t)q * CFHW + e;
        case 3: { const int b = q / p.W, wcol = q - b * p.W; return b * CFHW + e * p.W + wcol; }
        default: return e;
    }
}
__global__ void __launch_bounds__(256) grad_post_reduce_kernel(const GradPost p) {
    __shared__ double red[256];
    const int q = blockIdx.x, sp = blockIdx.y;
    const int64_t per = (p.ge + p.splits - 1) / p.splits, e0 = sp * per, e1 = min(e0 + per, p.ge);
    double acc = 0.0;
    for (int64_t e = e0 + threadIdx.x; e < e1; e += 256) acc += (double)fabsf(p.g[gp_src(p, gp_elem(p, q, e))]);
    red[threadIdx.x] = acc;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) { if ((int)threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s]; __syncthreads(); }
    if (threadIdx.x == 0) p.partial[(int64_t)q * p.splits + sp] = red[0];
}
__global__ void __launch_bounds__(256) grad_post_apply_kernel(const GradPost p, const int64_t total) {
    const int64_t HW = (int64_t)p.H * p.W, CFHW = (int64_t)p.C * p.F * HW;
    for (int64_t o = blockIdx.x * 256ll + threadIdx.x; o < total; o += (int64_t)gridDim.x * 256) {
        float v = p.g[gp_src(p, o)];
        if (p.mode) {
            const int b = (int)(o / CFHW);
            const int q = p.mode == 1 ? b * p.F + (int)((o / HW) % p.F) : p.mode == 2 ? b : p.mode == 3 ? b * p.W + (int)(o % p.W) : 0;
            double sum = 0.0;
            for (int sidx = 0; sidx < p.splits; ++sidx) sum += p.partial[(int64_t)q * p.splits + sidx];
            const float den = p.mode == 4 ? (float)sum : (float)sum / (float)p.ge;          // ||g||_1, or mean|g| (fp32 quotient as torch.mean)
            v = v / den;
        }
        if (p.mom) { v = v + p.mom[o] * p.decay; p.mom[o] = v; }
        p.out[o] = v;
    }
}
int k_grad_post_groups(int b, int c, int f, int h, int w, int mode, int64_t* ge) {
    const int64_t HW = (int64_t)h * w;
    switch (mode) {
        case 1: *ge = c * HW; return b * f;
        case 2: *ge = (int64_t)c * f * HW; return b;
        case 3: *ge = (int64_t)c * f * h; return b * w;
        case 4: *ge = (int64_t)b * c * f * HW; return 1;
        default: *ge = 0; return 0;
    }
}
int k_grad_post_splits(int64_t ge) { const int64_t s = (ge + 65535) / 65536; return (int)(s < 1 ? 1 : (s > 256 ? 256 : s)); }
int k_grad_post(const float* g, float* momentum, float* out, int b, int c, int f, int h, int w, int frame_major, int mode, float decay,
                double* partial, i2v_stream_t s) {
    GradPost p{g, momentum, out, partial, b, c, f, h, w, frame_major, mode, 1, decay, 0};
    const int G = k_grad_post_groups(b, c, f, h, w, mode, &p.ge);
    p.splits = k_grad_post_splits(p.ge);
    if (mode) { hipLaunchKernelGGL(grad_post_reduce_kernel, dim3((unsigned)G, (unsigned)p.splits), dim3(256), 0, (hipStream_t)s, p); LAUNCH_CHECK("grad_post_reduce"); }
    const int64_t total = (int64_t)b * c * f * h * w;
    hipLaunchKernelGGL(grad_post_apply_kernel, dim3(stream_grid(total, 1024)), dim3(256), 0, (hipStream_t)s, p, total);
    LAUNCH_CHECK("grad_post_apply"); return 0;
}

// ---- TAP's elementwise steps (i2v_tap_*_f32) ----
__global__ void __launch_bounds__(256) tap_perts_kernel(const float* __restrict__ adv, const float* __restrict__ vid, float* __restrict__ out, const int64_t n, const int64_t fhw) {
    for (int64_t i = blockIdx.x * 256ll + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) out[i] = (adv[i] - vid[i]) / c_std[(i / fhw) % 3];
}
__global__ void __launch_bounds__(256) tap_sign_abs_kernel(const float* __restrict__ sm, float* __restrict__ sg, double* __restrict__ partial, const int64_t n) {
    __shared__ double red[256];
    double acc = 0.0;
    for (int64_t i = blockIdx.x * 256ll + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const float v = sm[i];
        sg[i] = v > 0.f ? 1.f : (v < 0.f ? -1.f : v);                   // torch.sign: +-0 and NaN pass through
        acc += (double)fabsf(v);
    }
    red[threadIdx.x] = acc; __syncthreads();
    for (int s = 128; s > 0; s >>= 1) { if ((int)threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s]; __syncthreads(); }
    if (threadIdx.x == 0) partial[blockIdx.x] = red[0];
}
__global__ void tap_sum_finish_kernel(const double* __restrict__ partial, const int nblk, float* __restrict__ reg) {
    double s = 0.0; for (int i = 0; i < nblk; ++i) s += partial[i];
    *reg = (float)s;
}
__global__ void __launch_bounds__(256) tap_grad_kernel(const float* __restrict__ gx, const float* __restrict__ bs, float* __restrict__ out, const int64_t n,
                                                       const int C, const int F, const int HW, const float weight) {
    for (int64_t o = blockIdx.x * 256ll + threadIdx.x; o < n; o += (int64_t)gridDim.x * 256) {
        const int i = (int)(o % HW); int64_t r = o / HW; const int f = (int)(r % F); r /= F; const int c = (int)(r % C); const int64_t b = r / C;
        out[o] = gx[((b * F + f) * C + c) * (int64_t)HW + i] + weight * bs[o] / c_std[c];
    }
}
int k_tap_perts(const float* adv, const float* videos, float* out, int b, int c, int f, int h, int w, i2v_stream_t s) {
    const int64_t n = (int64_t)b * c * f * h * w;
    hipLaunchKernelGGL(tap_perts_kernel, dim3(stream_grid(n, 1024)), dim3(256), 0, (hipStream_t)s, adv, videos, out, n, (int64_t)f * h * w);
    LAUNCH_CHECK("tap_perts"); return 0;
}
int k_tap_sign_abs(const float* smooth, float* sign_out, float* reg, int64_t n, double* partial, i2v_stream_t s) {
    unsigned nblk = stream_grid(n, 4096); if (nblk > 1024) nblk = 1024;
    hipLaunchKernelGGL(tap_sign_abs_kernel, dim3(nblk), dim3(256), 0, (hipStream_t)s, smooth, sign_out, partial, n);
    hipLaunchKernelGGL(tap_sum_finish_kernel, dim3(1), dim3(1), 0, (hipStream_t)s, partial, (int)nblk, reg);
    LAUNCH_CHECK("tap_sign_abs"); return 0;
}
int k_tap_grad(const float* gx, const float* boxsign, float* out, int b, int c, int f, int h, int w, float weight, i2v_stream_t s) {
    const int64_t n = (int64_t)b * c * f * h * w;
    hipLaunchKernelGGL(tap_grad_kernel, dim3(stream_grid(n, 1024)), dim3(256), 0, (hipStream_t)s, gx, boxsign, out, n, c, f, h * w, weight);
    LAUNCH_CHECK("tap_grad"); return 0;
}

int k_aens_coeffs(const float* prev, float* coeffs, float momentum, int L, i2v_stream_t s) {
    hipLaunchKernelGGL(aens_coeffs_kernel, dim3(1), dim3(64), 0, (hipStream_t)s, prev, coeffs, momentum, L);
    LAUNCH_CHECK("aens_coeffs"); return 0;
}
int k_aens_reduce(const float* cosv, const float* coeffs, int L, int frames, float* feat_sum, float* weighted, i2v_stream_t s) {
    hipLaunchKernelGGL(aens_reduce_kernel, dim3(L), dim3(64), 0, (hipStream_t)s, cosv, coeffs, L, frames, feat_sum, weighted);
    LAUNCH_CHECK("aens_reduce"); return 0;
}


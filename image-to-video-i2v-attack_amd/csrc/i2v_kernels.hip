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
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <type_traits>
#include <utility>

#include "i2v_kernels.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

static thread_local char g_be_err[256];
static thread_local bool g_be_has_err = false;

static int hip_fail(hipError_t e, const char* what) {
    snprintf(g_be_err, sizeof g_be_err, "%s: %s", what, hipGetErrorString(e));
    g_be_has_err = true;
    return 1;
}
#define HIPCHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) return hip_fail(e_, #x); } while (0)
#define LAUNCH_CHECK(name) do { hipError_t e_ = hipGetLastError(); if (e_ != hipSuccess) return hip_fail(e_, name); } while (0)

const char* be_name() { return "hip:gfx950"; }
static long long g_stat_conv = 0, g_stat_pws = 0, g_stat_bf3 = 0, g_stat_igh = 0, g_stat_sth = 0;      // (relaxed counters: diagnostics only)
long long be_stat(const char* name) {
    if (!strcmp(name, "conv_launches")) return __atomic_load_n(&g_stat_conv, __ATOMIC_RELAXED);
    if (!strcmp(name, "pws_launches")) return __atomic_load_n(&g_stat_pws, __ATOMIC_RELAXED);
    if (!strcmp(name, "bf3_launches")) return __atomic_load_n(&g_stat_bf3, __ATOMIC_RELAXED);
    if (!strcmp(name, "ighalo_launches")) return __atomic_load_n(&g_stat_igh, __ATOMIC_RELAXED);
    if (!strcmp(name, "stemhalo_launches")) return __atomic_load_n(&g_stat_sth, __ATOMIC_RELAXED);
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

// Developer hook: tools/conv_microbench.cpp defines I2V_PROBE_T as a type that records per-block time stamps (kernel entry, K-loop
// start / end, exit) before it includes this file.  The product compiles the empty probe below: every call is an inline no-op.
#ifndef I2V_PROBE_T
struct I2VNoProbe {
    __device__ __forceinline__ void entry() {}
    __device__ __forceinline__ void loop_begin() {}
    __device__ __forceinline__ void loop_end(int /*block slot*/) {}
    __device__ __forceinline__ void exit(int /*block slot*/) {}
};
#define I2V_PROBE_T I2VNoProbe
#endif
__constant__ float c_mean[3] = {0.485f, 0.456f, 0.406f};
__constant__ float c_std[3] = {0.229f, 0.224f, 0.225f};

// =============================================================================================
// implicit-GEMM convolution on fp32 MFMA
// =============================================================================================
// Block: 256 threads = 4 waves arranged WD x WP; block tile BD (output channels) x BP (pixels),
// K consumed in chunks of I2V_KC=16 through double-buffered LDS (register-staged prefetch).
// MFMA operand roles: A = weights (row i = channel), B = activations (column j = pixel):
//   A: lane l holds Wp[k = kk + (l>>5)][cd = l&31]      B: lane l holds X[k = kk + (l>>5)][px = l&31]
//   D: lane l, register r  ->  pixel l&31, channel (r&3) + 8*(r>>2) + 4*(l>>5)
// so every global store instruction writes 32 consecutive pixels of one channel plane.
// n / d for 0 <= n < 2^31 with the precomputed (m, s) of fastdiv_magic: exact
__device__ __forceinline__ unsigned fastdiv(unsigned n, unsigned m, unsigned s) {
    return (unsigned)(((unsigned long long)n * m) >> s);
}
static void fastdiv_magic(unsigned d, uint32_t* m, uint32_t* s) {
    // s = 31 + ceil(log2 d), m = floor(2^s / d) + 1 (< 2^32): floor(n*m / 2^s) == floor(n / d) for every n < 2^31
    unsigned l = 0; while ((1ull << l) < d) ++l;
    *s = 31 + l;
    *m = (uint32_t)(((1ull << (31 + l)) / d) + 1);
}
// k-table row through the constant address space: stays a scalar (SMEM) load next to the LDS-DMA traffic;
// an ordinary VGPR-destination load there would make hipcc drain vmcnt(0) inside the pipeline.
typedef int i2v_v4i __attribute__((ext_vector_type(4)));
__device__ __forceinline__ I2VKEntry load_kentry(const I2VKEntry* tab, int k) {
#if defined(__HIP_DEVICE_COMPILE__)
    const i2v_v4i v = ((const __attribute__((address_space(4))) i2v_v4i*)tab)[k];
    return I2VKEntry{v.x, v.y, v.z, v.w};
#else
    return tab[k];
#endif
}

// MODE 0: per-row k-table gather (any geometry); 1: pointwise float4 (1x1, stride 1, planes 16-B aligned);
// 2: tap-uniform chunks (every 16-row K chunk shares one spatial tap: channel count % 16 == 0)
// 4: "quad rows" (I2VConvParams::quad): the 3-channel stems.  K rows come in groups of four adjacent taps (dw0 .. dw0+3) of one
//    (channel, frame, row) tap: ONE 16-byte DMA per lane stages four K rows of its pixel -- the LDS image of a chunk is
//    [quad][pixel][4] -- instead of four 4-byte pieces with a k-table row each (MODE 0 spent its time issuing DMA
//    instructions: 17 TFLOP/s on SlowFast's 5x7x7 stem).  Elements whose tap falls outside the row, or beyond the kernel
//    width (zero weights), hold a neighbour's pixel and are replaced by 0 when the fragment is read.
// PREF (single-pass tiles only): the epilogue's addend / gate tiles are fetched into registers BEFORE the
// K loop, so for the low-K, HBM-bound layers the read traffic overlaps the matrix work instead of following it.
// PRE: the B operand is relu(x * pre_scale[k] + pre_shift[k]) (DenseNet norm->relu->1x1 conv), applied when the
// fragment is read from LDS; k == input channel for the 1x1 convolutions this is used on.
// VID: the launch has temporal taps or a non-identity frame mapping (video networks, I2VConvParams::temporal);
// image launches -- and the spatial / pointwise convolutions of video networks -- compile without any of it.
// MF16: 16x16x4 MFMA fragments instead of 32x32x2 (same peak rate): for launches with <= 16 output rows -- the
// class-packed image gradient (12 rows), 8/16-channel layers -- a 32-row tile would be mostly padding.
// Residency: every tile is compiled for a stated number of waves per SIMD (= resident 256-thread blocks per CU), which
// makes the register allocator count the MFMA accumulators in the unified VGPR file and stop at the matching budget:
//   64x64   7  (49 registers since the tile body became a device function -- 61 before --; the allocator then also keeps <= 96
//               SGPRs -- MI355X_MICROARCH.md "Residency": 98+ SGPRs admit only 6 blocks per CU.  The launches that matter have
//               1568*k tiles = 6.125*k per CU: at 6 resident blocks the last 32 tiles waited for a second round.  8 blocks fit
//               as well (78 SGPRs) and measured 0.5 % slower, twice)
//   64x64 with epilogue prefetch  6  (78 registers; 7 would spill)
//   128x64  6  (70 registers, 24 KB of LDS; +0.5 % over 5)     64x128  5  (83 registers, 32 KB)
//   128x128  3  (147-150; left alone the allocator used 147 + 64 AGPRs = 2 blocks)
// The 256-pixel tiles are bounded by LDS: 32x256 is compiled for the 4 blocks it gets, 16x256 is left alone.  All without spills (-Rpass-analysis).
#ifndef I2V_PRIO_LEVELS      // progress-ordered wave priority in the K loop (conv_tile, chunk_body): highest level; 0 = off
#define I2V_PRIO_LEVELS 3
#endif
#ifndef I2V_DEEP             // deeper-prefetch K loop for the short-K HBM-bound pointwise launches (conv_tile, DEEP)
#define I2V_DEEP 1
#endif
#ifndef I2V_DEEP_STAGES      // LDS buffers of that loop: 3 = two chunks ahead at the two-buffer loop's residency (24 KB, 6 blocks)
#define I2V_DEEP_STAGES 3
#endif
#ifndef I2V_BF3_STAGES       // LDS buffers of the split-bf16 loop (conv_tile, BF3 == 1): chunks in flight = stages - 1.  Measured (tools/bf3_sweep.sh,
#define I2V_BF3_STAGES 2     // profiles/r5_split_bf16.txt): 3 and 4 buffers change nothing on the 128x128 tile and cost the smaller tiles a resident block
#endif
#ifndef I2V_SMALL_WPE
#define I2V_SMALL_WPE 7
#endif
#ifndef I2V_PREF_WPE
#define I2V_PREF_WPE 6
#endif
#ifndef I2V_MID_WPE
#define I2V_MID_WPE 5
#endif
#ifndef I2V_TALL_WPE
#define I2V_TALL_WPE 6
#endif
#ifndef I2V_BIG_WPE
#define I2V_BIG_WPE 3
#endif

static constexpr int conv_waves_per_simd(int BD, int BP, bool PREF, bool hi, int MODE = 0) {
    return (BD == 64 && BP == 64) ? (PREF ? (I2V_DEEP && MODE == 1 && I2V_DEEP_STAGES > 3 ? 4 /* 32 KB of LDS: the 5th block does not fit beside the runtime's own */ : I2V_PREF_WPE) : I2V_SMALL_WPE) : (BD == 128 && BP == 64) ? I2V_TALL_WPE : (BD == 32 && BP == 256) ? 4 /* LDS-bound: what the allocator delivers anyway */ : BD * BP == 8192 ? I2V_MID_WPE : BD * BP == 16384 ? I2V_BIG_WPE : (hi ? 8 : 1);
}
#define I2V_CONV_WPE __attribute__((amdgpu_waves_per_eu(conv_waves_per_simd(BD, BP, PREF, false, MODE), conv_waves_per_simd(BD, BP, PREF, true, MODE))))
// The pointwise variant with prefetched epilogue operands (short K, HBM-bound) stages four chunks instead of two (conv_tile, DEEP)
static constexpr bool conv_deep(int MODE, bool PREF) { return I2V_DEEP && PREF && MODE == 1; }
// LDS floats one tile needs: operand staging [NST][KC][BD] + [NST][KC][BP], re-used by the epilogue as a [WD*FR][BP] transpose buffer
template <int BD, int BP, int WD, bool MF16, int NST = 2, int CPB = 1, int BF3 = 0>
constexpr int conv_lds_floats() {
    // (BF3: the weight tile of a chunk is 3 bf16 planes in MFMA-fragment order, 3 KB per 32 rows instead of fp32's 2 KB)
    // (BF3 == 1: (BD / 32) * 3 one-KB pieces per chunk, rounded up to a multiple of 4 so that every wave issues the same number; BF3 == 2: weights never enter LDS)
    constexpr int stage = NST * CPB * (I2V_KC * BP + (BF3 == 2 ? 0 : BF3 ? ((BD / 32) * 3 + 3) / 4 * 4 * 256 : I2V_KC * BD)), epi = WD * (MF16 ? 16 : 32) * BP;      // (BF3 == 3: NST = 3)
    return stage > epi ? stage : epi;
}

// MODE 5 ("halo") of the 64x64 tile: weights [2][KC][64] + halo rows [2][KC][64 + 2 W + 2]
template <int HWM>
constexpr int conv_halo_lds_floats() {
    constexpr int stage = 2 * I2V_KC * (64 + 64 + 2 * HWM + 2), epi = 64 * 64;
    return stage > epi ? stage : epi;
}

// The dense ("vector") epilogue of a tile, shared by conv_tile and by the second phase of the fused pair kernel (conv_fused_kernel):
// accumulators -> LDS transpose -> per lane 4 consecutive pixels of one channel -> gate_scale / shift / addends / ReLU / gates ->
// 16-byte store (+ this tensor's own 1-bit gates).  FUSE: the result is deposited in the LDS tile `mid` ([BD][BP], zeros where the
// tile sticks out of the launch) instead of `p.dst` -- the intermediate of a fused pair never goes to memory.
// ... its second half, the ROW pass: thread `t` of 256 (lane = t & 63) takes 4 consecutive pixels of NQ channel rows of the transposed
// tile `Cs` ([rows][BP], pass `i` of the tile's TD passes) through gate_scale / shift / addends / ReLU / gates to the 16-byte store.  A
// function of its own since round 5: the persistent pointwise kernel (conv_pw_stream) runs it on dedicated epilogue waves while the
// matrix waves are already in the next tile -- one implementation, the same expressions in the same order.
template <int BD, int BP, int WD, bool PREF, bool MF16, bool FUSE, typename PT>
__device__ __forceinline__ void conv_vec_rows(const PT& p, const int i, const int cd0, const int64_t px0, const float (*const Cs)[BP], const int t,
                                              const float4* const pre0, const unsigned* const pregw, float* const mid) {
    constexpr int FR = MF16 ? 16 : 32;
    const int lane = t & 63;
    const int HWg = p.Hg * p.Wg, HoWo = p.Ho * p.Wo;
    const int64_t P = (int64_t)p.N * HWg;
    (void)pre0; (void)pregw; (void)mid;
#ifdef I2V_NT_ALL
    const bool nt_store = true;
#else
    const bool nt_store = p.cfg > 0 && ((p.cfg - 1) & 128);
#endif
    constexpr int C4 = BP / 4, RSTEP = 1024 / BP, NQ = WD * FR / RSTEP;
    const int c4 = t % C4, rbase = t / C4;
    const int64_t pp = px0 + (int64_t)c4 * 4;
    const bool pok = pp < P;
    const int64_t n = pok ? fastdiv((unsigned)pp, p.dv_hw_m, p.dv_hw_s) : 0;
    const int64_t poff = pp - n * HWg;
    #pragma unroll
    for (int q = 0; q < NQ; ++q) {
        const int row = rbase + q * RSTEP;
        const int cd = cd0 + (row / FR) * (BD / WD) + i * FR + (row % FR);
        const bool valid = pok && cd < p.Cd;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (valid) {
        v = *reinterpret_cast<const float4*>(&Cs[row][c4 * 4]);
        const int64_t o = (int64_t)cd * HoWo + poff;
        if (p.gate_scale) {      // pre-activation gate on THIS contribution, before the (accumulating) adds
            const float4 m = *reinterpret_cast<const float4*>(p.mask + n * p.mask_nstride + o);
            const float gs = p.gate_scale[cd], gt = p.gate_shift[cd];
            if (!(fmaf(m.x, gs, gt) > 0.f)) v.x = 0.f;
            if (!(fmaf(m.y, gs, gt) > 0.f)) v.y = 0.f;
            if (!(fmaf(m.z, gs, gt) > 0.f)) v.z = 0.f;
            if (!(fmaf(m.w, gs, gt) > 0.f)) v.w = 0.f;
        }
        if (p.shift) { const float sh = p.shift[cd]; v.x += sh; v.y += sh; v.z += sh; v.w += sh; }
        if (PREF) {
            v.x += pre0[q].x; v.y += pre0[q].y; v.z += pre0[q].z; v.w += pre0[q].w;
        } else {
            if (p.add0 && p.add0_stride == 1) {
                const float4 a = *reinterpret_cast<const float4*>(p.add0 + n * p.add0_nstride + o);
                v.x += a.x; v.y += a.y; v.z += a.z; v.w += a.w;
            } else if (p.add0) {
                // compact stride-2 addend (input-gradient of a 1x1/2 shortcut): defined at even
                // (h, w) only; the 4 pixels start at a multiple of 4, so elements 0 and 2 receive
                const int oh = (int)fastdiv((unsigned)poff, p.dv_wo_m, p.dv_wo_s), ow = (int)(poff - (int64_t)oh * p.Wo);
                if (!(oh & 1) && (oh >> 1) < p.add0_H) {
                    const float2 a = *reinterpret_cast<const float2*>(
                        p.add0 + n * p.add0_nstride + (int64_t)cd * p.add0_H * p.add0_W + (oh >> 1) * p.add0_W + (ow >> 1));
                    v.x += a.x; v.z += a.y;
                }
            }
        }
        if (p.add1) {
            const float4 a = *reinterpret_cast<const float4*>(p.add1 + n * p.add1_nstride + o);
            v.x += a.x; v.y += a.y; v.z += a.z; v.w += a.w;
        }
        if (p.relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
        if (p.gate) {            // 1-bit gates of the tensor whose gradient this is: 4 bits of one word
            const unsigned w = PREF ? pregw[q] : p.gate[(int64_t)cd * p.gate_stride + ((p.gate_pix0 + pp) >> 5)];
            const unsigned g = w >> ((unsigned)(p.gate_pix0 + pp) & 31u);
            if (!(g & 1u)) v.x = 0.f;
            if (!(g & 2u)) v.y = 0.f;
            if (!(g & 4u)) v.z = 0.f;
            if (!(g & 8u)) v.w = 0.f;
        } else if (p.mask && !p.gate_scale) {
            const float4 m = *reinterpret_cast<const float4*>(p.mask + n * p.mask_nstride + o);
            if (!(m.x > 0.f)) v.x = 0.f;
            if (!(m.y > 0.f)) v.y = 0.f;
            if (!(m.z > 0.f)) v.z = 0.f;
            if (!(m.w > 0.f)) v.w = 0.f;
        }
        if constexpr (!FUSE) {
            // Streaming (non-temporal) store, autotuner bit 7 (round 4): the tile's 16-byte stores go past the L2 instead of
            // allocating lines in it.  Isolated (tools/pw_sweep.sh, configurations | 128): +13 % on 128 -> 512 @28^2, +13...20 % on
            // 64 -> 64 @56^2, +3 % on 64 -> 256 @56^2, -4 % on 256 -> 1024 @14^2: shape- and epilogue-dependent, so it is timed per launch
            // (second stage of the plan-time autotuner); in the attack it is worth 0.3-0.45 % (64 -> 256 forward -5 %).  Streaming LOADS of
            // the addend were measured too (+1.5...7 % alone, worse than the stores alone when combined) and not kept.
            if (nt_store) { typedef float nt4 __attribute__((ext_vector_type(4))); const nt4 w4 = {v.x, v.y, v.z, v.w};
                            __builtin_nontemporal_store(w4, reinterpret_cast<nt4*>(p.dst + n * p.dst_nstride + o)); }
            else *reinterpret_cast<float4*>(p.dst + n * p.dst_nstride + o) = v;
        }
        }
        if constexpr (FUSE)       // first phase of a fused pair: the finished tile stays in LDS, [channel][pixel], zeros outside
            *reinterpret_cast<float4*>(mid + (cd - cd0) * BP + c4 * 4) = v;
        if (p.gate_out) {
            // this tensor's own gates: 8 consecutive lanes hold 32 consecutive pixels of one channel row
            // (BP/4 lanes per row, a multiple of 8); every lane takes part in the exchange, invalid ones with 0
            unsigned nib = valid ? ((v.x > 0.f ? 1u : 0u) | (v.y > 0.f ? 2u : 0u) | (v.z > 0.f ? 4u : 0u) | (v.w > 0.f ? 8u : 0u)) : 0u;
            nib <<= 4 * (lane & 7);
            // OR over the 8 lanes with DPP moves (VALU only; __shfl_xor would go through the LDS crossbar):
            // quad_perm [1,0,3,2], quad_perm [2,3,0,1], then row_half_mirror (lane i <-> 7-i of each 8)
            nib |= (unsigned)__builtin_amdgcn_update_dpp(0, (int)nib, 0xB1, 0xF, 0xF, true);
            nib |= (unsigned)__builtin_amdgcn_update_dpp(0, (int)nib, 0x4E, 0xF, 0xF, true);
            nib |= (unsigned)__builtin_amdgcn_update_dpp(0, (int)nib, 0x141, 0xF, 0xF, true);
            if (valid && (lane & 7) == 0)
                p.gate_out[(int64_t)cd * p.gate_out_stride + ((p.gate_out_pix0 + pp) >> 5)] = nib;
        }
    }
}

#define I2V_FROW(r) (MF16 ? 4 * lk + (r) : ((r) & 3) + 8 * ((r) >> 2) + 4 * lk)
template <int BD, int BP, int WD, int WP, bool PREF, bool MF16, bool FUSE, typename ACC, typename PT>
__device__ __forceinline__ void conv_vec_epilogue(const PT& p, ACC (&acc)[BD / WD / (MF16 ? 16 : 32)][BP / WP / (MF16 ? 16 : 32)], const int cd0,
                                                  const int64_t px0, float* const smem, const float4* const pre0, const unsigned* const pregw,
                                                  float* const mid) {
    constexpr int FR = MF16 ? 16 : 32, NR = MF16 ? 4 : 16, TD = BD / WD / FR, TP = BP / WP / FR;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int wd = wave / WP, wpx = wave % WP;
    const int l31 = MF16 ? (lane & 15) : (lane & 31), lk = MF16 ? (lane >> 4) : (lane >> 5);
    // Dense output (grid == output plane, plane % 4 == 0): transpose the accumulators through LDS so
    // that each lane owns 4 consecutive pixels of one channel; addends, gate and result then move as
    // 16-byte accesses, 512 contiguous bytes per channel row.
    float (*Cs)[BP] = reinterpret_cast<float (*)[BP]>(smem);
#pragma unroll
    for (int i = 0; i < TD; ++i) {
        __syncthreads();
#pragma unroll
        for (int j = 0; j < TP; ++j)
#pragma unroll
            for (int r = 0; r < NR; ++r)
                Cs[wd * FR + I2V_FROW(r)][wpx * (BP / WP) + j * FR + l31] = acc[i][j][r];
        __syncthreads();
        conv_vec_rows<BD, BP, WD, PREF, MF16, FUSE>(p, i, cd0, px0, Cs, t, pre0, pregw, mid);
    }
}
#undef I2V_FROW

// One tile of the implicit GEMM.  `bid` of `nwg` blocks share `n_cd_tiles` channel tiles per pixel tile, the first pixel tile
// starting at pixel `px_base` (a launch may be cut into regions with different tile shapes, conv_igemm_tail below).
// CPB ("chunks per barrier", round 4): an LDS buffer holds CPB consecutive K chunks and the loop synchronises once per CPB chunks --
// the same packing, k-table and k order (results are bit-identical), half the vmcnt(0) / barrier / first-fragment round trips per
// MFMA.  Those are what a block that is alone on its CU (an under-filled launch: a single 32-frame clip leaves the 14x14 layers
// with 1.5 tiles per CU) cannot hide behind a neighbour.  Costs LDS (64x64: 32 KB, 5 resident blocks), so it is one more
// configuration of the autotuner (bit 6), for launches whose chunk count is a multiple of CPB.
template <int BD, int BP, int WD, int WP, int MODE, bool PREF, bool PRE = false, bool VID = false, bool MF16 = false, int HWM = 0, int CPB = 1, int FUSE = 0, int BF3 = 0>
__device__ __forceinline__ void conv_tile(const I2VConvParams& p, const int n_cd_tiles, const int bid, const int nwg, const int64_t px_base,
                                          float* const smem, I2V_PROBE_T& probe, const int probe_slot, const int prio_arg = I2V_PRIO_LEVELS,
                                          float* const mid = nullptr) {
#if defined(__HIP_DEVICE_COMPILE__)      // buffer-resource types and LDS-DMA builtins exist only in the device pass
    constexpr int KC = I2V_KC;
    constexpr int FR = MF16 ? 16 : 32;                       // fragment edge
    constexpr int NR = MF16 ? 4 : 16;                        // accumulator registers per fragment
    constexpr int TD = BD / WD / FR, TP = BP / WP / FR;
    static_assert(!(MF16 && (PRE || PREF)), "no pre-activation / prefetch variants of the 16x16 tile");
    // one LDS array: operand staging [NST][KC][BD] + [NST][KC][BP], re-used by the epilogue as a
    // [WD*32][BP] transpose buffer
    // LDS operand buffers: chunk c is consumed while chunk c+1 is in flight -- or, for the short-K HBM-bound pointwise launches
    // (DEEP, see the main loop), while chunks c+1 .. c+3 are
    // ... and the split-bf16 loop with staged weights (BF3 == 1), whose chunks last 6 x 32 cycles per fragment pair instead of 8 x 64: one
    // chunk of look-ahead no longer covers an L2 round trip
    // BF3 == 3: variant 1 with the ACTIVATION fragments of chunk c + 1 read and split under the MFMAs of chunk c (software pipelining across the
    // barrier): three staging buffers -- chunk c + 2 is in flight, chunk c + 1 is being read, chunk c's weights are being read
    constexpr bool DEEP = conv_deep(MODE, PREF) || (BF3 == 1 && I2V_BF3_STAGES > 2) || BF3 == 3;
    constexpr int NST = BF3 == 3 ? 3 : BF3 == 1 ? I2V_BF3_STAGES : DEEP ? I2V_DEEP_STAGES : 2, AHEAD = NST - 1;
    // MODE 5 ("halo"): a 3x3 / stride-1 / pad-1 launch on planes exactly HWM wide stages, per 16-channel group, ONE halo row per
    // channel -- the tile's 64 pixels plus a source row and a pixel on either side -- instead of nine shifted copies of the tile
    constexpr bool HALO = MODE == 5;
    constexpr int HS = HALO ? BP + 2 * HWM + 2 : 1, HQ = (HS + 63) / 64;
    static_assert(!HALO || (HWM > 0 && BD == 64 && BP == 64 && !PREF && !PRE && !VID && !MF16), "halo staging: the plain 64x64 image tile only");
    static_assert(CPB == 1 || (!HALO && !DEEP && !PREF && !PRE && MODE != 4 && MODE != 0), "several chunks per barrier: the plain pointwise / tap-uniform loops only");
    constexpr int KB = CPB * KC;                              // K rows per LDS buffer
    // BF3 (round 5, "split-bf16" arithmetic): the weights arrive pre-split into three bf16 planes in the 32x32x16 MFMA's own fragment
    // order (I2VConvParams::wp3: per K chunk and 32-row tile 3 x 64 lanes x 16 bytes), the activations stay fp32 in LDS and are split
    // in registers when a fragment is read; six bf16 MFMAs per 16 K rows replace eight fp32 ones at twice the cycles each.
    static_assert(!BF3 || (!MF16 && !PRE && !PREF && !HALO && (MODE == 1 || MODE == 2) && FUSE == 0 && BD % 32 == 0), "split-bf16 K loop: the plain pointwise / tap-uniform tiles");
    // BF3 == 2: the weight fragments do not go through LDS at all -- they are already in fragment order in memory, so every wave loads
    // its own (16 bytes per lane and term, coalesced 1 KB per load, served by L1 / L2 for the waves that share rows) one chunk ahead into
    // registers.  The LDS-DMA instruction stream of a chunk then carries only the activations: with A staged (BF3 == 1) a 128x64 tile
    // issued 12 weight pieces + 4 activation pieces per 48 MFMAs, and a bf16 MFMA lasts 32 cycles where a DMA piece costs its wave
    // 60-185 to issue -- the loop was bound by DMA issue (matrix pipe 41 % busy on the layer3 3x3 shape).
    constexpr int AF = BF3 == 2 ? 0 : BF3 ? CPB * (((BD / 32) * 3 + 3) / 4 * 4) * 256 : KB * BD;    // floats of weight staging per LDS buffer
    float (*As)[KB][BD] = reinterpret_cast<float (*)[KB][BD]>(smem);         // (fp32 path)
    float* const As3 = smem;                                                   // (BF3 path: [NST][CPB][BD / 32][3][64 lanes][4 floats])
    float (*Bs)[KB][BP] = reinterpret_cast<float (*)[KB][BP]>(smem + NST * AF);

    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int wd = wave / WP, wpx = wave % WP;

    // XCD-aware remap: consecutive logical tiles (same pixel tile, neighbouring channel tiles) share
    // one XCD's L2 instead of being dealt round-robin over the 8 XCDs (bijective form).
    const int q8 = nwg >> 3, r8 = nwg & 7, xcd = bid & 7;
    const int lid = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (bid >> 3);
    const int cd_tile = lid % n_cd_tiles;
    const int64_t px0 = px_base + (int64_t)(lid / n_cd_tiles) * BP;
    const int cd0 = cd_tile * BD;

    const int HWg = p.Hg * p.Wg;
    const int64_t P = (int64_t)p.N * HWg;

    constexpr bool PW = MODE == 1;
    constexpr bool QUAD = MODE == 4;
    typedef __attribute__((address_space(3))) void* lds_ptr_t;
    // ---- operand staging: global -> LDS by buffer DMA (`buffer_load ... lds`) -------------------------
    // No VGPR round trip and no ds_write.  A wave-instruction deposits 64 lanes x {16,4} bytes at a
    // wave-uniform LDS base + lane*size, so the LDS images stay linear ([k][BD] / [k][BP]) and the im2col
    // gather lives in the per-lane 32-bit buffer offset.  Lanes that must contribute zeros (padding taps,
    // K tail, pixel tail) get an out-of-range offset: the buffer range check makes the DMA write 0.0 for
    // them (probed on gfx950: tools/bufdma_test.cpp), so the steady-state cost per chunk is a handful of
    // VALU instructions instead of 64-bit pointer arithmetic and pointer selects per load.
    // Wave w issues instructions w, w+4, ...; with that assignment a lane always serves ONE pixel column.
    constexpr unsigned OOB = 0x80000000u;                     // >= num_records (spans are kept < 2 GiB)
    const int wv = __builtin_amdgcn_readfirstlane(wave);      // scalar copy: LDS bases / M0 stay in SGPRs
    const __amdgpu_buffer_rsrc_t rs_w = BF3 ? __builtin_amdgcn_make_buffer_rsrc((void*)p.wp3, 0, (p.Kpad / KC) * (p.Cdpad / 32) * 3072, 0x00020000)
                                            : __builtin_amdgcn_make_buffer_rsrc((void*)p.wp, 0, p.Kpad * p.Cdpad * 4, 0x00020000);
    // MODE 4 reads up to 3 pixels before / 6 behind a row (masked afterwards): the resource starts 64 bytes early and ends 64
    // late -- a lane whose 16 bytes START out of range is zero-filled as a whole, its in-range pixels included -- and every
    // offset carries +64 (the executor keeps that slack around the staged input: Net::in_stage)
    constexpr unsigned XB = QUAD ? 64u : 0u;
    const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc((void*)((const char*)p.src - XB), 0, p.src_span_bytes + 2 * XB, 0x00020000);
    constexpr int NA = BF3 == 2 ? 0 : BF3 ? (BD / 32) * 3 : KC * BD / 256, NAQ = (NA + 3) / 4;     // weights: instructions of 256 floats (BF3 == 1: 1 KB = one plane of a 32-row tile)
    constexpr int BPER = (PW || QUAD) ? 256 : 64;             // activations: 16-byte or 4-byte pieces (floats per instruction)
    constexpr int NB = HALO ? 0 : KC * BP / BPER, NBQ = (NB + 3) / 4;      // MODE 5 stages its activations as halo rows
    const int bcol = PW ? (lane * 4) % BP : (BP >= 64 ? ((wave * 64) % BP) + lane : lane % BP);
    const int64_t ppix = px0 + bcol;
    const bool pvalid = ppix < P;
    const int64_t pn = pvalid ? fastdiv((unsigned)ppix, p.dv_hw_m, p.dv_hw_s) : 0;        // P < 2^31 (checked by k_conv)
    const int prem = (int)(ppix - pn * HWg);
    const int HWs = p.Hs * p.Ws;
    int h0 = 0, w0 = 0, t0 = 0;
    int64_t pns = pn;                                         // source frame of this lane's pixel
    if (VID) {                                                // grid frame (clip, tg) reads source frames tg*st + dt
        const int64_t clip = fastdiv((unsigned)pn, p.dv_t_m, p.dv_t_s);
        t0 = (int)(pn - clip * p.Tg) * p.st;
        pns = clip * p.Ts + t0;
    }
    const int nstr = (int)p.src_nstride;                      // a launch's source span is < 2 GiB
    unsigned xoff;                                            // byte offset of this lane's pixel in `src`
    if (PW) xoff = (unsigned)((pns * p.src_nstride + prem) * 4);
    else {
        const int gi = (int)fastdiv((unsigned)prem, p.dv_w_m, p.dv_w_s), gj = prem - gi * p.Wg;
        h0 = gi * p.sh; w0 = gj * p.sw;
        xoff = (unsigned)((pns * p.src_nstride + (int64_t)h0 * p.Ws + w0) * 4) + XB;
    }
    if (!pvalid) xoff = OOB;
    unsigned aoff[NAQ ? NAQ : 1];
#pragma unroll
    for (int q = 0; q < NAQ; ++q) {
        const int f = (wave + 4 * q) * 256 + lane * 4;
        aoff[q] = BF3 ? (unsigned)(((cd0 / 32) * 3 + wave + 4 * q) * 1024 + lane * 16) : (unsigned)(((f / BD) * p.Cdpad + f % BD + cd0) * 4);
    }
    unsigned boff[PW ? NBQ : 1];                              // PW: + row inside the chunk (lane dependent)
    if (PW) {
#pragma unroll
        for (int q = 0; q < NBQ; ++q) boff[q] = pvalid ? xoff + (unsigned)((((wave + 4 * q) * 256 + lane * 4) / BP) * HWs * 4) : OOB;
    }

    // One DMA instruction of this wave's share of a K chunk.  Piece j (compile-time) of the NL = NAQ + NBQ pieces a wave
    // issues per chunk: j < NAQ is a 16-byte piece of the weight tile, the others are pieces of the activation tile.
    // `vb_` is the per-lane byte offset of the chunk's tap (MODE 2; computed once per chunk by I2V_CHUNK_VB).
#define I2V_ISSUE_PIECE(j_, k0_, buf_, vb_) I2V_ISSUE_PIECE_SUB(j_, k0_, buf_, vb_, 0)
    // ... `sub_`: which of the buffer's CPB chunks the piece belongs to (its rows start at sub_ * KC)
#define I2V_ISSUE_PIECE_SUB(j_, k0_, buf_, vb_, sub_)                                                     \
    {                                                                                                     \
        constexpr int jj = (j_);                                                                          \
        const int k0 = (k0_);                                                                             \
        if constexpr (jj < NAQ) {                                                                         \
            const int ins = wv + 4 * jj;                                                                  \
            if constexpr (BF3) {   /* every wave issues NAQ pieces (the wait at the top of a chunk counts them): a piece beyond the tile's reads nothing */ \
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_w, (lds_ptr_t)(As3 + (buf_) * AF + (sub_) * (AF / CPB) + ins * 256), 16, ins < NA ? aoff[jj] : OOB, \
                                                         (k0 / KC) * (p.Cdpad / 32) * 3072, 0, 0);       \
            } else if (NA % 4 == 0 || ins < NA) {                                                         \
                if constexpr (false)                                                                      \
                    ;                                                                                     \
                else                                                                                      \
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_w, (lds_ptr_t)(&As[buf_][(sub_) * KC][0] + ins * 256), 16, aoff[jj],  \
                                                             k0 * p.Cdpad * 4, 0, 0);                     \
            }                                                                                             \
        } else {                                                                                          \
            constexpr int q = jj - NAQ;                                                                   \
            const int ins = wv + 4 * q;                                                                   \
            float* const bbuf = &Bs[buf_][(sub_) * KC][0];                                                \
            if (NB % 4 == 0 || ins < NB) {                                                                \
                if constexpr (PW) {                                                                       \
                    unsigned v = boff[q];                                                                 \
                    if (k0 + KC > p.K)       /* K tail (uniform test): rows >= K contribute zeros */      \
                        v = (k0 + ((wave + 4 * q) * 256 + lane * 4) / BP < p.K) ? v : OOB;                \
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_x, (lds_ptr_t)(bbuf + ins * 256), 16, v,         \
                                                             k0 * HWs * 4, 0, 0);                         \
                } else if constexpr (QUAD) {                                                              \
                    /* piece = quad (ins*64)/BP of the chunk x 64 pixels; its first row's k-table entry gives   \
                       channel plane, row / frame tap and dw0; the row run's quads alternate (quad = 1 or 2) */ \
                    const I2VKEntry e = load_kentry(p.ktab, k0 + 4 * ((ins * 64) / BP));                  \
                    const int hs = h0 + e.dh, dtk = VID ? (e.valid >> 1) : 0;                             \
                    const bool ok = pvalid && (unsigned)hs < (unsigned)p.Hs && (!VID || (unsigned)(t0 + dtk) < (unsigned)p.Ts); \
                    const unsigned v = ok ? xoff + (unsigned)((e.chan_off + dtk * nstr + e.dh * p.Ws + e.dw) * 4) : OOB; \
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_x, (lds_ptr_t)(bbuf + ins * 256), 16, v, 0, 0, 0); \
                } else if constexpr (MODE == 2) {                                                         \
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_x, (lds_ptr_t)(bbuf + ins * 64), 4, vb_,         \
                                                             ((ins * 64) / BP) * HWs * 4, 0, 0);          \
                } else {                                                                                  \
                    const I2VKEntry e = load_kentry(p.ktab, k0 + (ins * 64) / BP);                        \
                    const int hs = h0 + e.dh, ws = w0 + e.dw, dtk = VID ? (e.valid >> 1) : 0;             \
                    const bool ok = pvalid && (e.valid & 1) && (unsigned)hs < (unsigned)p.Hs &&           \
                                    (unsigned)ws < (unsigned)p.Ws && (!VID || (unsigned)(t0 + dtk) < (unsigned)p.Ts); \
                    const unsigned v = ok ? xoff + (unsigned)((e.chan_off + dtk * nstr + e.dh * p.Ws + e.dw) * 4) : OOB; \
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_x, (lds_ptr_t)(bbuf + ins * 64), 4, v, 0, 0, 0); \
                }                                                                                         \
            }                                                                                             \
        }                                                                                                 \
    }
    // MODE 2: every K row of a chunk shares ONE tap, described by the chunk's first k-table row
#define I2V_CHUNK_VB(e_)                                                                                  \
    ([&]() -> unsigned {                                                                                  \
        const int hs = h0 + (e_).dh, ws = w0 + (e_).dw, dtk = VID ? ((e_).valid >> 1) : 0;                \
        const bool ok = pvalid && (unsigned)hs < (unsigned)p.Hs && (unsigned)ws < (unsigned)p.Ws &&       \
                        (!VID || (unsigned)(t0 + dtk) < (unsigned)p.Ts);                                  \
        return ok ? xoff + (unsigned)(((e_).chan_off + dtk * nstr + (e_).dh * p.Ws + (e_).dw) * 4) : OOB; \
    }())

    // ---- epilogue operand prefetch ----
    constexpr int E_C4 = BP / 4, E_RSTEP = 1024 / BP, E_NQ = WD * FR / E_RSTEP;
    static_assert(!PREF || TD == 1, "PREF needs a single epilogue pass");
    // Only the first addend and the 1-bit gate word are prefetched (20 registers): a second addend or an fp32 mask
    // (I2V_GATES=0) is read in the epilogue itself.  Prefetching all four cost 48 more registers and one third of the
    // resident blocks -- on launches that are HBM-bound and live on bytes in flight.
    float4 pre0[PREF ? E_NQ : 1];
    unsigned pregw[PREF ? E_NQ : 1];                       // 1-bit gates: the word holding this lane's 4 bits
    if (PREF) {
        const int e_c4 = t % E_C4, e_rbase = t / E_C4;
        const int64_t e_pp = px0 + (int64_t)e_c4 * 4;
        const bool e_ok = e_pp < P;
        const int64_t e_n = e_ok ? fastdiv((unsigned)e_pp, p.dv_hw_m, p.dv_hw_s) : 0;
        const int64_t e_poff = e_pp - e_n * HWg;
        const int e_HoWo = p.Ho * p.Wo;
#pragma unroll
        for (int q = 0; q < E_NQ; ++q) {
            const int row = e_rbase + q * E_RSTEP;
            const int cd = cd0 + (row >> 5) * (BD / WD) + (row & 31);
            const bool ok = e_ok && cd < p.Cd;
            const int64_t o = (int64_t)cd * e_HoWo + e_poff;
            pre0[q] = (ok && p.add0) ? *reinterpret_cast<const float4*>(p.add0 + e_n * p.add0_nstride + o) : make_float4(0.f, 0.f, 0.f, 0.f);
            pregw[q] = (ok && p.gate) ? p.gate[(int64_t)cd * p.gate_stride + ((p.gate_pix0 + e_pp) >> 5)] : 0xffffffffu;
        }
    }

    probe.loop_begin();
    typedef short bf8 __attribute__((ext_vector_type(8)));
    typedef float f4 __attribute__((ext_vector_type(4)));
    constexpr int TD_ = BD / WD / (MF16 ? 16 : 32);
    bf8 wcur[BF3 == 2 ? CPB : 1][BF3 == 2 ? TD_ : 1][3];      // BF3 == 2: this wave's weight fragments of the current buffer fill
    const char* const w3lane = BF3 ? (const char*)p.wp3 + (size_t)lane * 16 + (size_t)(cd0 / 32 + (wave / WP) * TD_) * 3072 : nullptr;
    const size_t w3chunk = BF3 ? (size_t)(p.Cdpad / 32) * 3072 : 0;          // bytes of one K chunk of wp3
    auto load_w3 = [&](const int chunk, const int i, const int pl) {
        return __builtin_bit_cast(bf8, *reinterpret_cast<const f4*>(w3lane + (size_t)chunk * w3chunk + (i * 3 + pl) * 1024));
    };
    if constexpr (BF3 == 2) {
#pragma unroll
        for (int sb = 0; sb < CPB; ++sb)
#pragma unroll
            for (int i = 0; i < TD_; ++i)
#pragma unroll
                for (int pl = 0; pl < 3; ++pl) wcur[sb][i][pl] = load_w3(sb, i, pl);
    }
    (void)wcur; (void)w3lane; (void)w3chunk;
    constexpr int TP_ = BP / WP / (MF16 ? 16 : 32);
    bf8 xcur[BF3 == 3 ? TP_ : 1][3];                           // BF3 == 3: the split activation fragments of the CURRENT chunk
    (void)xcur;
    auto bf3_split2 = [](const float lo, const float hi, unsigned& p1, unsigned& p2, unsigned& p3) {
        // (plain casts, not inline asm: hipcc emits v_cvt_pk_bf16_f32 for them on gfx950 -- round to nearest even -- and, unlike asm
        //  statements, the instruction scheduler may interleave them with the MFMAs)
        typedef __bf16 bf2 __attribute__((ext_vector_type(2)));
        typedef float f2 __attribute__((ext_vector_type(2)));
        const unsigned a = __builtin_bit_cast(unsigned, __builtin_convertvector((f2){lo, hi}, bf2));
        const float rl = lo - __builtin_bit_cast(float, a << 16), rh = hi - __builtin_bit_cast(float, a & 0xffff0000u);
        const unsigned b = __builtin_bit_cast(unsigned, __builtin_convertvector((f2){rl, rh}, bf2));
        const float sl = rl - __builtin_bit_cast(float, b << 16), sh = rh - __builtin_bit_cast(float, b & 0xffff0000u);
        const unsigned c = __builtin_bit_cast(unsigned, __builtin_convertvector((f2){sl, sh}, bf2));
        p1 = a; p2 = b; p3 = c;
    };
    auto bf3_split_frag = [&](const float (&x)[8], bf8 (&out)[3]) {
        typedef unsigned u4 __attribute__((ext_vector_type(4)));
        unsigned q1[4], q2[4], q3[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) bf3_split2(x[2 * e], x[2 * e + 1], q1[e], q2[e], q3[e]);
        out[0] = __builtin_bit_cast(bf8, (u4){q1[0], q1[1], q1[2], q1[3]});
        out[1] = __builtin_bit_cast(bf8, (u4){q2[0], q2[1], q2[2], q2[3]});
        out[2] = __builtin_bit_cast(bf8, (u4){q3[0], q3[1], q3[2], q3[3]});
    };
    (void)bf3_split_frag;
    typedef typename std::conditional<MF16, f32x4, f32x16>::type acc_t;
    acc_t acc[TD][TP];
#pragma unroll
    for (int a = 0; a < TD; ++a)
#pragma unroll
        for (int b = 0; b < TP; ++b)
#pragma unroll
            for (int r = 0; r < NR; ++r) acc[a][b][r] = 0.f;

    const int nchunks = p.Kpad / KC;
    // fragment coordinates of this lane: column (pixel) inside a fragment, K row inside a k-step, and the
    // accumulator register -> fragment row map  (32x32x2: row = (r&3) + 8(r>>2) + 4(l>>5);  16x16x4: row = 4(l>>4) + r)
    const int l31 = MF16 ? (lane & 15) : (lane & 31), lk = MF16 ? (lane >> 4) : (lane >> 5);
#define I2V_FROW(r) (MF16 ? 4 * lk + (r) : ((r) & 3) + 8 * ((r) >> 2) + 4 * lk)
    // ---- main loop: two LDS buffers, ONE barrier per K chunk, software-pipelined inside the wave -------------
    // An fp32 MFMA holds its SIMD for 64 (32x32x2) / 32 (16x16x4) cycles, so everything else a wave has to do
    // for a chunk is issued in the shadow of its own MFMAs instead of in front of them:
    //   top of iteration c:  s_waitcnt vmcnt(0) (this wave's DMA of chunk c, issued a whole iteration ago),
    //                        raw s_barrier (every wave's DMA landed AND every wave finished reading the other buffer);
    //   then the fragments of k-step 0 are read, and -- k-step by k-step -- the fragments of step s+1 are requested
    //   before the MFMAs of step s, and the DMA instructions of chunk c+1 follow the MFMAs of steps 0, 1, ... one or
    //   two at a time (an LDS-DMA instruction costs its wave tens of issue cycles: behind an MFMA they are free, in
    //   front of the chunk's first MFMA they were a bubble on the matrix pipe).  The k-table row of chunk c+2 (MODE 2)
    //   is fetched (SMEM) an iteration before it is needed, so its latency is off the path as well.
    // `__syncthreads()` is avoided on purpose: its fence would add waits the pipeline does not need.
    constexpr int NL = NAQ + NBQ;                                       // DMA instructions per wave per chunk
    constexpr int KR = MF16 ? 4 : 2;                                    // K rows per MFMA (32x32x2 / 16x16x4)
    constexpr int KS = KC / KR;                                         // k-steps per chunk
    constexpr int PPS = (NL + KS - 1) / KS;                             // DMA pieces issued behind each k-step
    I2VKEntry e_next[CPB];                                              // MODE 2: k-table rows of the chunks of the next-but-one buffer fill
#pragma unroll
    for (int h = 0; h < CPB; ++h) e_next[h] = I2VKEntry{0, 0, 0, 0};
    const int nsuper = nchunks / CPB;                                   // loop iterations (k_conv offers CPB > 1 only when it divides)
    // MODE 4: which of this lane's B-fragment elements are real taps.  Element e of run-quad qi is dw = quad_dw0 + 4 qi + e; it
    // counts if it lies inside the kernel (4 qi + e < quad_kw) and inside the row.  16x16x4: a lane's element is lk, k-step s
    // is quad s of the chunk; 32x32x2: k-step s is half (s & 1) of quad s >> 1, element 2 (s & 1) + lk.  A chunk holds 4
    // quads and the run length (1 or 2) divides 4, so run-quad = quad & (p.quad - 1): the masks do not depend on the chunk.
    bool qok[QUAD ? TP : 1][2][MF16 ? 1 : 2];
    if constexpr (QUAD) {
#pragma unroll
        for (int j = 0; j < TP; ++j) {
            const int64_t fp = px0 + wpx * (BP / WP) + j * FR + l31;
            const unsigned fr = fp < P ? (unsigned)fp - fastdiv((unsigned)fp, p.dv_hw_m, p.dv_hw_s) * (unsigned)HWg : 0u;
            const int fj = (int)(fr - fastdiv(fr, p.dv_w_m, p.dv_w_s) * (unsigned)p.Wg);
#pragma unroll
            for (int qi = 0; qi < 2; ++qi)
#pragma unroll
                for (int hf = 0; hf < (MF16 ? 1 : 2); ++hf) {
                    const int el = MF16 ? lk : 2 * hf + lk;
                    qok[j][qi][hf] = 4 * qi + el < p.quad_kw && (unsigned)(fj * p.sw + p.quad_dw0 + 4 * qi + el) < (unsigned)p.Ws;
                }
        }
    }
    if constexpr (HALO) {
        // ---- MODE 5 main loop: groups of 16 channels x 9 taps; the nine chunks of a group are unrolled (tap index compile-time) ----
        constexpr int NT = 9, HPW = 4 * HQ, PPC = (HPW + NT - 1) / NT;       // halo DMA pieces per wave per group / per chunk
        float* const Hb = smem + NST * KC * BD;                                // [2][KC][HS]
        const int W_ = HWM;
        int tsh[NT], tdh[NT], tdw[NT];                                         // per tap: shift inside a halo row, row / column offset
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const I2VKEntry e = load_kentry(p.ktab, t * KC);
            tdh[t] = e.dh; tdw[t] = e.dw; tsh[t] = e.dh * W_ + e.dw + W_ + 1;
        }
        const int chan0 = load_kentry(p.ktab, 0).chan_off;
        const int ngroups = nchunks / NT;
        const int gstride = ngroups > 1 ? load_kentry(p.ktab, NT * KC).chan_off - chan0 : 0;
        // this lane's fragment pixel: validity of each tap as one bit
        const int64_t fp = px0 + wpx * (BP / WP) + l31;
        const bool fpv = fp < P;
        const unsigned fr = fpv ? (unsigned)fp - fastdiv((unsigned)fp, p.dv_hw_m, p.dv_hw_s) * (unsigned)HWg : 0u;
        const int fh = (int)fastdiv(fr, p.dv_w_m, p.dv_w_s), fw = (int)fr - fh * p.Wg;
        unsigned tmask = 0;
#pragma unroll
        for (int t = 0; t < NT; ++t)
            tmask |= (fpv && (unsigned)(fh + tdh[t]) < (unsigned)p.Hs && (unsigned)(fw + tdw[t]) < (unsigned)p.Ws) ? (1u << t) : 0u;
        const int lbase = lk * HS + wpx * (BP / WP) + l31;                    // float index of this lane's element in row (k = lk), shift 0
        // halo element 64 q + lane of a row = flattened pixel px0 - (W + 1) + 64 q + lane, in whichever frame it lies
        unsigned hoff[HQ];
#pragma unroll
        for (int q = 0; q < HQ; ++q) {
            const int64_t vp = px0 - (W_ + 1) + 64 * q + lane;
            const bool ok = vp >= 0 && vp < P;
            const int64_t vn = ok ? fastdiv((unsigned)vp, p.dv_hw_m, p.dv_hw_s) : 0;
            hoff[q] = ok ? (unsigned)((vn * p.src_nstride + (vp - vn * HWg)) * 4) : OOB;
        }
        // piece idx (compile-time) of a group: channel 4 wave + idx / HQ, 64-lane piece idx % HQ of its row.  A wave's rows are
        // written in order, so a row's last piece may run into the next row (overwritten by that row's own pieces, issued
        // later by the same wave); only the LAST row of a wave must not overrun: that piece is cut by EXEC.
        auto halo_piece = [&]<int IDX>(std::integral_constant<int, IDX>, const int gb, const int chan_off) {
            constexpr int chl = IDX / HQ, q = IDX % HQ;
            float* const dst = Hb + ((gb * KC + wv * 4 + chl) * HS + 64 * q);
            const int so = (chan_off + (wv * 4 + chl) * HWs) * 4;
            if constexpr (chl == 3 && q == HQ - 1) {
                if (lane < HS - 64 * q) __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_x, (lds_ptr_t)dst, 4, hoff[q], so, 0, 0);
            } else
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_x, (lds_ptr_t)dst, 4, hoff[q], so, 0, 0);
        };
        if (ngroups > 0) {      // prologue: the halo rows of group 0, the weight tile of chunk 0
            [&]<int... I>(std::integer_sequence<int, I...>) { ((halo_piece(std::integral_constant<int, I>{}, 0, chan0)), ...); }
            (std::make_integer_sequence<int, HPW>{});
            I2V_ISSUE_PIECE(0, 0, 0, OOB);
        }
        int gbuf = 0;
        for (int g = 0; g < ngroups; ++g) {
            const bool more_g = g + 1 < ngroups;
            const int chan_next = chan0 + (g + 1) * gstride;
            [&]<int... T>(std::integer_sequence<int, T...>) {
                (([&] {
                    constexpr int t = T;
                    const int abuf = (g + t) & 1;                              // chunk g * 9 + t: 9 is odd
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    __builtin_amdgcn_s_barrier();
                    const int hidx = lbase + gbuf * KC * HS + tsh[t];
                    const bool ok = (tmask >> t) & 1u;
                    float fa[2], fb[2];
                    auto rd = [&](const int s_, const int set) {
                        fa[set] = As[abuf][KR * s_ + lk][wd * (BD / WD) + l31];
                        fb[set] = Hb[hidx + KR * s_ * HS];
                    };
                    rd(0, 0);
                    [&]<int... S>(std::integer_sequence<int, S...>) {
                        (([&] {
                            constexpr int s_ = S, set = S & 1;
                            if constexpr (s_ + 1 < KS) rd(s_ + 1, set ^ 1);
                            __builtin_amdgcn_sched_barrier(0);
                            fb[set] = ok ? fb[set] : 0.f;
                            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[set], fb[set], acc[0][0], 0, 0, 0);
                            if constexpr (s_ == 0) {                           // the weight tile of the next chunk
                                if (t < NT - 1 || more_g) I2V_ISSUE_PIECE(0, (g * NT + t + 1) * KC, abuf ^ 1, OOB);
                            }
                            if constexpr (s_ >= 1 && s_ <= PPC) {              // the next group's halo rows, PPC pieces per chunk, in order
                                constexpr int idx = t * PPC + (s_ - 1);
                                if constexpr (idx < HPW) { if (more_g) halo_piece(std::integral_constant<int, idx>{}, gbuf ^ 1, chan_next); }
                            }
                        }()), ...);
                    }(std::make_integer_sequence<int, KS>{});
                }()), ...);
            }(std::make_integer_sequence<int, NT>{});
            gbuf ^= 1;
        }
    } else {   // prologue: the first buffer fill(s) (and the k-table rows of the next one)
        unsigned vb0[CPB];
#pragma unroll
        for (int h = 0; h < CPB; ++h) vb0[h] = OOB;
        if constexpr (MODE == 2) {
#pragma unroll
            for (int h = 0; h < CPB; ++h) { const I2VKEntry e0 = load_kentry(p.ktab, h * KC); vb0[h] = I2V_CHUNK_VB(e0); }
#pragma unroll
            for (int h = 0; h < CPB; ++h) e_next[h] = load_kentry(p.ktab, (nsuper > AHEAD ? AHEAD * CPB + h : h) * KC);      // the rows of fill AHEAD (issued in iteration 0)
        }
        (void)vb0;
        for (int c0 = 0; c0 < AHEAD && c0 < nsuper; ++c0) {
            if constexpr (MODE == 2 && AHEAD > 1) {      // (deeper look-ahead: every prologue fill has its own tap)
                if (c0 > 0) {
#pragma unroll
                    for (int h = 0; h < CPB; ++h) { const I2VKEntry e0 = load_kentry(p.ktab, (c0 * CPB + h) * KC); vb0[h] = I2V_CHUNK_VB(e0); }
                }
            }
            [&]<int... J>(std::integer_sequence<int, J...>) {
                (([&] { constexpr int sub = J / NL; I2V_ISSUE_PIECE_SUB(J % NL, (c0 * CPB + sub) * KC, c0, vb0[sub], sub); }()), ...);
            }(std::make_integer_sequence<int, CPB * NL>{});
        }
    }
    // Progress-ordered priority (round 3).  The per-block timeline of a launch (tools/conv_microbench.cpp -DCMB_PROBE) shows that
    // the co-resident blocks of a CU do NOT finish together: the hardware serves the oldest wave first, so on the layer3 3x3
    // shape the first of a CU's six blocks leaves its K loop after 137 us and the last after 227 -- every CU ends a launch with
    // one or two blocks left, which cannot fill the matrix pipe on their own (a lone 64x64 block is issue-bound at ~55 % of it).
    // A block therefore starts at priority `prio_hi` and steps down each time it completes another 1 / (prio_hi + 1) of its K
    // loop: blocks that are behind outrank blocks that are ahead, they advance and finish together (first block out at 181 us,
    // last at 223).  Only for loops of >= 16 chunks that are not the HBM-bound prefetching variant (those measured -3..-11 %:
    // their time is the epilogue's memory traffic, and four steps over 4-8 chunks only reorder it).  Arbitration only: the
    // arithmetic is untouched.  Measured per shape (same binary otherwise): +1..2 %; with the tail split, whose quarter tiles
    // run ABOVE these levels (conv_igemm_tail), layer3 3x3 117.5 -> 124.1 TFLOP/s, layer3 reduce 120.4 -> 125.1.
    // ... and only for launches that fill the chip several times over (>= 3 blocks per CU): an under-filled launch has nothing to
    // keep together, and when two clip lanes share the GPU a nearly finished block (level 0) would starve behind the other lane's
    // fresh ones (single clip, two frame lanes: 495-504 frames/s with the levels everywhere, 507-517 without)
    const int prio_hi = (nchunks >= 16 && !PREF && nwg >= 3 * 256) ? prio_arg : 0;
    int prio_lvl = prio_hi, prio_next = 0, prio_step = 0;
    if (prio_hi > 0) {
        prio_step = (nsuper + prio_hi) / (prio_hi + 1); prio_next = prio_step;
        if (prio_hi >= 3) __builtin_amdgcn_s_setprio(3); else if (prio_hi == 2) __builtin_amdgcn_s_setprio(2); else __builtin_amdgcn_s_setprio(1);
    }
    // DEEP (round 3; the pointwise variant with prefetched epilogue operands, i.e. K <= 256 and HBM-bound): four LDS buffers, the
    // DMA of chunk c+3 issued during chunk c.  These launches spend their time waiting for memory, not in the matrix pipe: a
    // 64 -> 256 expand convolution has FOUR chunks of 8 MFMAs (0.2 us) each, and with one chunk in flight every one of them
    // exposed a full round trip of the saturated memory system (per-block timeline: K loop 4.6 us of a 13 us block).  With three
    // chunks in flight the loop pays about one round trip in all.  The wait at the top of a chunk counts the DMA instructions of
    // the YOUNGER chunks that may stay in flight (every wave issues the same NL per chunk; the epilogue prefetch loads are older).
    auto chunk_body = [&](const int c, const int buf, auto more_tag) {
        constexpr bool MORE = decltype(more_tag)::value;                // a chunk c+AHEAD exists: its DMA is issued here
        if constexpr (BF3 == 3) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // chunk c + 1 has landed too (c + 2 is issued below)
        else if constexpr (DEEP) {
            const int younger = nsuper - 1 - c < AHEAD - 1 ? nsuper - 1 - c : AHEAD - 1;      // chunks behind c already issued
            if (NST > 3 && younger >= 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * NL) : "memory");
            else if (younger >= 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NL) : "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        } else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (prio_hi > 0 && c == prio_next) {
            prio_next += prio_step; --prio_lvl;
            if (prio_lvl == 2) __builtin_amdgcn_s_setprio(2); else if (prio_lvl == 1) __builtin_amdgcn_s_setprio(1); else __builtin_amdgcn_s_setprio(0);
        }
        unsigned vb[CPB];
#pragma unroll
        for (int h = 0; h < CPB; ++h) vb[h] = OOB;
        if constexpr (MORE && MODE == 2) {
#pragma unroll
            for (int h = 0; h < CPB; ++h) vb[h] = I2V_CHUNK_VB(e_next[h]);      // taps of the fill issued now, c + AHEAD (rows fetched last iteration)
#pragma unroll
            for (int h = 0; h < CPB; ++h) {
                const int c2 = (c + AHEAD + 1) * CPB + h < nchunks ? (c + AHEAD + 1) * CPB + h : nchunks - 1;
                e_next[h] = load_kentry(p.ktab, c2 * KC);                       // prefetch the rows of the fill after that
            }
        }
        (void)vb;
        if constexpr (BF3 == 3) {
            // ---- split-bf16 chunk, software-pipelined: weight fragments of chunk c and RAW activation values of chunk c + 1 are requested
            // first, the MFMAs of chunk c run on the activation fragments split during chunk c - 1 (with the DMA pieces of chunk c + 2 and the
            // split of chunk c + 1 interleaved behind them by the scheduling hints below), so the matrix pipe does not wait for LDS latency
            // and the 44-instruction split at the top of every chunk.
            static_assert(CPB == 1, "one chunk per barrier");
            bf8 wa[TD][3];
            const float* const abase = As3 + buf * AF + lane * 4;
#pragma unroll
            for (int i = 0; i < TD; ++i)
#pragma unroll
                for (int pl = 0; pl < 3; ++pl)
                    wa[i][pl] = __builtin_bit_cast(bf8, *reinterpret_cast<const f4*>(abase + ((wd * TD + i) * 3 + pl) * 256));
            // (branch-free on purpose: after the last chunk this reads and splits whatever the ring's next buffer holds and nothing uses it --
            //  a branch would end the scheduling region and put the split back behind the MFMAs)
            const int nb = buf + 1 == NST ? 0 : buf + 1;
            float xr[TP][8];
#pragma unroll
            for (int j = 0; j < TP; ++j)
#pragma unroll
                for (int e = 0; e < 8; ++e) xr[j][e] = Bs[nb][8 * lk + e][wpx * (BP / WP) + j * FR + l31];
            constexpr int TW[6] = {2, 1, 0, 1, 0, 0}, TX[6] = {0, 1, 2, 0, 1, 0};
            constexpr int NM = 6 * TD * TP, NS = 4 * TP;                  // MFMAs of the chunk; slices of the split (one value pair each)
            unsigned nq[TP][3][4];                                        // the next chunk's fragments, pair by pair
            [&]<int... M>(std::integer_sequence<int, M...>) {
                (([&] {
                    constexpr int m = M, term = m / (TD * TP), ij = m % (TD * TP), i = ij / TP, j = ij % TP;
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wa[i][TW[term]], xcur[j][TX[term]], acc[i][j], 0, 0, 0);
                    if constexpr (MORE) {
                        if constexpr (m < NL) { I2V_ISSUE_PIECE_SUB(m, (c + AHEAD) * KC, (buf + AHEAD) % NST, vb[0], 0); }
                    }
                    // one slice of the split behind every (NM / NS)-th MFMA, pinned there: left to itself the scheduler issues all MFMAs first
                    // and the 44 vector instructions after them, where nothing overlaps them
                    [&]<int... KK>(std::integer_sequence<int, KK...>) {      // slice k sits behind MFMA (k + 1) NM / NS - 1
                        (([&] {
                            constexpr int k = KK, jj = k / 4, e = k % 4;
                            if constexpr ((k + 1) * NM / NS - 1 == m)
                                bf3_split2(xr[jj][2 * e], xr[jj][2 * e + 1], nq[jj][0][e], nq[jj][1][e], nq[jj][2][e]);
                        }()), ...);
                    }(std::make_integer_sequence<int, NS>{});
                    __builtin_amdgcn_sched_barrier(0);
                }()), ...);
            }(std::make_integer_sequence<int, NM>{});
            typedef unsigned u4 __attribute__((ext_vector_type(4)));
#pragma unroll
            for (int j = 0; j < TP; ++j)
#pragma unroll
                for (int pl = 0; pl < 3; ++pl) xcur[j][pl] = __builtin_bit_cast(bf8, (u4){nq[j][pl][0], nq[j][pl][1], nq[j][pl][2], nq[j][pl][3]});
            return;
        }
        if constexpr (BF3) {
            // ---- split-bf16 chunk: per 16 K rows, 3 x TD weight fragments (ds_read_b128, pre-split) and TP activation fragments read as
            // fp32 (8 values per lane: K rows 8 lk .. 8 lk + 7 of this lane's pixel) and split into three bf16 terms x = x1 + x2 + x3
            // (x1 = bf16(x), x2 = bf16(x - x1), x3 = bf16(x - x1 - x2): the residuals are exact in fp32, what is left after x3 is below
            // 2^-26 |x|).  Products kept: w1 x1, w1 x2, w2 x1, w1 x3, w2 x2, w3 x1 -- everything down to 2^-26 of |w||x|, i.e. below an
            // fp32 product's own rounding; each bf16 x bf16 product is exact in the MFMA's fp32 accumulation.  Fixed order, small terms
            // first.  The DMA pieces of the next buffer fill follow the MFMAs one at a time, as in the fp32 loop.
            auto split2 = [](const float lo, const float hi, unsigned& p1, unsigned& p2, unsigned& p3) {
                unsigned a, b, c;
                asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(a) : "v"(lo), "v"(hi));
                const float rl = lo - __builtin_bit_cast(float, a << 16), rh = hi - __builtin_bit_cast(float, a & 0xffff0000u);
                asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(b) : "v"(rl), "v"(rh));
                const float sl = rl - __builtin_bit_cast(float, b << 16), sh = rh - __builtin_bit_cast(float, b & 0xffff0000u);
                asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(c) : "v"(sl), "v"(sh));
                p1 = a; p2 = b; p3 = c;
            };
            bf8 wnxt[BF3 == 2 ? CPB : 1][BF3 == 2 ? TD : 1][3];      // the next buffer fill's weight fragments, in flight during this one's MFMAs
            if constexpr (BF3 == 2 && MORE) {
#pragma unroll
                for (int sb = 0; sb < CPB; ++sb)
#pragma unroll
                    for (int i = 0; i < TD; ++i)
#pragma unroll
                        for (int pl = 0; pl < 3; ++pl) wnxt[sb][i][pl] = load_w3((c + 1) * CPB + sb, i, pl);
            }
            (void)wnxt;
            [&]<int... SB>(std::integer_sequence<int, SB...>) {
                (([&] {
                    constexpr int sub = SB;
                    bf8 wa[TD][3], xb[TP][3];
                    if constexpr (BF3 == 2) {
#pragma unroll
                        for (int i = 0; i < TD; ++i)
#pragma unroll
                            for (int pl = 0; pl < 3; ++pl) wa[i][pl] = wcur[sub][i][pl];
                    } else {
                        const float* const abase = As3 + buf * AF + sub * (AF / CPB) + lane * 4;
#pragma unroll
                        for (int i = 0; i < TD; ++i)
#pragma unroll
                            for (int pl = 0; pl < 3; ++pl)
                                wa[i][pl] = __builtin_bit_cast(bf8, *reinterpret_cast<const f4*>(abase + ((wd * TD + i) * 3 + pl) * 256));
                    }
#pragma unroll
                    for (int j = 0; j < TP; ++j) {
                        float x[8];
#pragma unroll
                        for (int e = 0; e < 8; ++e) x[e] = Bs[buf][sub * KC + 8 * lk + e][wpx * (BP / WP) + j * FR + l31];
                        unsigned q1[4], q2[4], q3[4];
#pragma unroll
                        for (int e = 0; e < 4; ++e) split2(x[2 * e], x[2 * e + 1], q1[e], q2[e], q3[e]);
                        typedef unsigned u4 __attribute__((ext_vector_type(4)));
                        xb[j][0] = __builtin_bit_cast(bf8, (u4){q1[0], q1[1], q1[2], q1[3]});
                        xb[j][1] = __builtin_bit_cast(bf8, (u4){q2[0], q2[1], q2[2], q2[3]});
                        xb[j][2] = __builtin_bit_cast(bf8, (u4){q3[0], q3[1], q3[2], q3[3]});
                    }
                    // (weight term, activation term) pairs, smallest products first
                    constexpr int TW[6] = {2, 1, 0, 1, 0, 0}, TX[6] = {0, 1, 2, 0, 1, 0};
                    [&]<int... M>(std::integer_sequence<int, M...>) {
                        (([&] {
                            constexpr int m = M, term = m / (TD * TP), ij = m % (TD * TP), i = ij / TP, j = ij % TP;
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wa[i][TW[term]], xb[j][TX[term]], acc[i][j], 0, 0, 0);
                            if constexpr (MORE) {
                                constexpr int jp = sub * 6 * TD * TP + m;          // one DMA piece behind each of the first CPB * NL MFMAs
                                if constexpr (jp < CPB * NL) {
                                    constexpr int sb2 = jp / NL;
                                    I2V_ISSUE_PIECE_SUB(jp % NL, ((c + AHEAD) * CPB + sb2) * KC, (buf + AHEAD) % NST, vb[sb2], sb2);
                                }
                            }
                        }()), ...);
                    }(std::make_integer_sequence<int, 6 * TD * TP>{});
                }()), ...);
            }(std::make_integer_sequence<int, CPB>{});
            if constexpr (BF3 == 2 && MORE) {
#pragma unroll
                for (int sb = 0; sb < CPB; ++sb)
#pragma unroll
                    for (int i = 0; i < TD; ++i)
#pragma unroll
                        for (int pl = 0; pl < 3; ++pl) wcur[sb][i][pl] = wnxt[sb][i][pl];
            }
            return;
        }
        float fa[2][TD], fb[2][TP];
        auto read_frags = [&](const int s, const int set) {
#pragma unroll
            for (int i = 0; i < TD; ++i) fa[set][i] = As[buf][KR * s + lk][wd * (BD / WD) + i * FR + l31];
            if constexpr (QUAD) {       // [quad][pixel][4] image: 16x16x4 reads element lk of quad s, 32x32x2 element 2(s&1)+lk of quad s>>1
                const float* const bq = &Bs[buf][0][0];
#pragma unroll
                for (int j = 0; j < TP; ++j) {
                    const int px = wpx * (BP / WP) + j * FR + l31;
                    const int qd = MF16 ? s : (s >> 1), el = MF16 ? lk : 2 * (s & 1) + lk;
                    const float v = bq[(qd * BP + px) * 4 + el];
                    const bool m = ((qd & 1) && p.quad == 2) ? qok[j][1][MF16 ? 0 : (s & 1)] : qok[j][0][MF16 ? 0 : (s & 1)];
                    fb[set][j] = m ? v : 0.f;
                }
            } else {
#pragma unroll
                for (int j = 0; j < TP; ++j) fb[set][j] = Bs[buf][KR * s + lk][wpx * (BP / WP) + j * FR + l31];
            }
        };
        read_frags(0, 0);
        [&]<int... S>(std::integer_sequence<int, S...>) {
            (([&] {
                constexpr int s = S, set = S & 1;
                if constexpr (s + 1 < CPB * KS) read_frags(s + 1, set ^ 1);
                __builtin_amdgcn_sched_barrier(0);          // keep the NEXT step's LDS reads in front of this step's MFMAs
                if constexpr (PRE) {
                    typedef const __attribute__((address_space(4))) float* cfp;       // scalar (SMEM) loads
                    const int kr = c * KB + 2 * s;
                    const float sc = lk ? ((cfp)p.pre_scale)[kr + 1] : ((cfp)p.pre_scale)[kr];
                    const float sh = lk ? ((cfp)p.pre_shift)[kr + 1] : ((cfp)p.pre_shift)[kr];
#pragma unroll
                    for (int j = 0; j < TP; ++j) fb[set][j] = fmaxf(fmaf(fb[set][j], sc, sh), 0.f);
                }
#pragma unroll
                for (int i = 0; i < TD; ++i)
#pragma unroll
                    for (int j = 0; j < TP; ++j) {
                        if constexpr (MF16) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[set][i], fb[set][j], acc[i][j], 0, 0, 0);
                        else acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[set][i], fb[set][j], acc[i][j], 0, 0, 0);
                    }
                if constexpr (MORE) {
                    [&]<int... Q>(std::integer_sequence<int, Q...>) {
                        (([&] {
                            constexpr int jp = s * PPS + Q;
                            if constexpr (jp < CPB * NL) {
                                constexpr int sub = jp / NL;
                                I2V_ISSUE_PIECE_SUB(jp % NL, ((c + AHEAD) * CPB + sub) * KC, DEEP ? (buf + AHEAD) % NST : (buf ^ 1), vb[sub], sub);
                            }
                        }()), ...);
                    }(std::make_integer_sequence<int, PPS>{});
                }
            }()), ...);
        }(std::make_integer_sequence<int, CPB * KS>{});
    };
    if constexpr (BF3 == 3) {      // the first chunk's activation fragments, before the loop
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        const int l31_ = lane & 31, lk_ = lane >> 5;
#pragma unroll
        for (int j = 0; j < TP_; ++j) {
            float x[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) x[e] = Bs[0][8 * lk_ + e][(wave % WP) * (BP / WP) + j * 32 + l31_];
            bf3_split_frag(x, xcur[j]);
        }
    }
    if constexpr (!HALO) {
        int c = 0, buf = 0;
        for (; c + AHEAD < nsuper; ++c) { chunk_body(c, buf, std::true_type{}); buf = buf + 1 == NST ? 0 : buf + 1; }
        for (; c < nsuper; ++c) { chunk_body(c, buf, std::false_type{}); buf = buf + 1 == NST ? 0 : buf + 1; }
    }
#undef I2V_ISSUE_PIECE
#undef I2V_ISSUE_PIECE_SUB
#undef I2V_CHUNK_VB
    if (prio_hi > 0) __builtin_amdgcn_s_setprio(0);
    probe.loop_end(probe_slot);

    // ---- epilogue: shift, addends, ReLU, gradient gate, NCHW store ----
    const int HoWo = p.Ho * p.Wo;
    if (p.vec_epilogue) {
        conv_vec_epilogue<BD, BP, WD, WP, PREF, MF16, FUSE != 0>(p, acc, cd0, px0, smem, pre0, pregw, mid);
        return;
    }
    if (p.blk > 1 || (VID && p.blkt > 1)) {
        // class-packed Cd (image gradient; frame-paired forward stems with blk = 1): cd = ((ct*blk + ph)*blk + pw)*Creal + c -> channel c at
        // (gi*osh+ph, gj*osw+pw) of frame tau*ost + ot0 + ct
        const int bb = p.blk * p.blk, Creal = p.Cd / ((VID ? p.blkt : 1) * bb);
#pragma unroll
        for (int j = 0; j < TP; ++j) {
            const int64_t pp = px0 + wpx * (BP / WP) + j * FR + l31;
            if (pp >= P) continue;
            const int64_t ng = fastdiv((unsigned)pp, p.dv_hw_m, p.dv_hw_s);
            const int rem = (int)(pp - ng * HWg);
            const int gi = (int)fastdiv((unsigned)rem, p.dv_w_m, p.dv_w_s), gj = rem - gi * p.Wg;
            const int64_t clip = VID ? fastdiv((unsigned)ng, p.dv_t_m, p.dv_t_s) : ng;
            const int otb = VID ? (int)(ng - clip * p.Tg) * p.ost + p.ot0 : 0;
#pragma unroll
            for (int i = 0; i < TD; ++i)
#pragma unroll
                for (int r = 0; r < NR; ++r) {
                    const int cd = cd0 + wd * (BD / WD) + i * FR + I2V_FROW(r);
                    if (cd >= p.Cd) continue;
                    const int cls3 = cd / Creal, c = cd - cls3 * Creal;
                    const int ct = VID ? cls3 / bb : 0, cls = cls3 - ct * bb;
                    const int oh = gi * p.osh + cls / p.blk + p.oh0, ow = gj * p.osw + cls % p.blk + p.ow0;
                    if (oh >= p.Ho || ow >= p.Wo || (VID && otb + ct * p.oct >= p.To)) continue;
                    const int64_t n = VID ? clip * p.To + otb + ct * p.oct : ng;
                    const int64_t o = (int64_t)c * HoWo + oh * p.Wo + ow;
                    float v = acc[i][j][r];
                    if (p.shift) v += p.shift[c];
                    if (p.add1) v += p.add1[n * p.add1_nstride + o];
                    if (p.relu) v = fmaxf(v, 0.f);
                    if (p.mask && !(p.mask[n * p.mask_nstride + o] > 0.f)) v = 0.f;
                    p.dst[n * p.dst_nstride + o] = v;
                }
        }
        return;
    }
#pragma unroll
    for (int j = 0; j < TP; ++j) {
        // (no early `continue`s: when the launch emits gate words, every lane of the wave takes part in the ballots)
        const int64_t pp = px0 + wpx * (BP / WP) + j * FR + l31;
        const bool pin = pp < P;
        const int64_t ng = pin ? fastdiv((unsigned)pp, p.dv_hw_m, p.dv_hw_s) : 0;
        const int rem = (int)(pin ? pp - ng * HWg : 0);
        const int gi = (int)fastdiv((unsigned)rem, p.dv_w_m, p.dv_w_s), gj = rem - gi * p.Wg;
        const int oh = gi * p.osh + p.oh0, ow = gj * p.osw + p.ow0;
        bool ok = pin && oh < p.Ho && ow < p.Wo;
        int64_t n = ng;                                          // destination frame
        if (VID) {
            const int64_t clip = fastdiv((unsigned)ng, p.dv_t_m, p.dv_t_s);
            const int ot = (int)(ng - clip * p.Tg) * p.ost + p.ot0;
            if (ot >= p.To) ok = false;
            n = clip * p.To + ot;
        }
        const int opix = oh * p.Wo + ow;
        float* dstn = p.dst + n * p.dst_nstride + opix;
        const float* a0 = nullptr; int a0_plane = HoWo;
        if (p.add0 && ok) {
            if (p.add0_stride == 1) a0 = p.add0 + n * p.add0_nstride + opix;
            else {
                const int s = p.add0_stride, qh = oh / s, qw = ow / s;
                if (qh * s == oh && qw * s == ow && qh < p.add0_H && qw < p.add0_W) {
                    a0 = p.add0 + n * p.add0_nstride + qh * p.add0_W + qw;
                    a0_plane = p.add0_H * p.add0_W;
                }
            }
        }
        const float* a1 = p.add1 ? p.add1 + n * p.add1_nstride + opix : nullptr;
        const float* mk = p.mask ? p.mask + n * p.mask_nstride + opix : nullptr;
        const int64_t gidx = (int64_t)p.gate_pix0 + n * HoWo + opix;           // this element's bit in a gate row
#pragma unroll
        for (int i = 0; i < TD; ++i) {
#pragma unroll
            for (int r = 0; r < NR; ++r) {
                const int cd = cd0 + wd * (BD / WD) + i * FR + I2V_FROW(r);
                const bool okc = ok && cd < p.Cd;
                float v = 0.f;
                if (okc) {
                    v = acc[i][j][r];
                    if (p.gate_scale && !(fmaf(mk[(int64_t)cd * HoWo], p.gate_scale[cd], p.gate_shift[cd]) > 0.f)) v = 0.f;
                    if (p.shift) v += p.shift[cd];
                    if (a0) v += a0[(int64_t)cd * a0_plane];
                    if (a1) v += a1[(int64_t)cd * HoWo];
                    if (p.relu) v = fmaxf(v, 0.f);
                    if (p.gate) { if (!((p.gate[(int64_t)cd * p.gate_stride + (gidx >> 5)] >> (gidx & 31)) & 1u)) v = 0.f; }
                    else if (mk && !p.gate_scale && !(mk[(int64_t)cd * HoWo] > 0.f)) v = 0.f;
                    dstn[(int64_t)cd * HoWo] = v;
                }
                if (p.gate_out) {
                    // dense forward output: a fragment's FR lanes are FR consecutive pixels (aligned to FR) of channel cd,
                    // so the ballot's FR-bit field IS that stretch of the gate row
                    const unsigned long long bal = __ballot(okc && v > 0.f);
                    if (l31 == 0 && pin && cd < p.Cd) {
                        const int64_t bit0 = (int64_t)p.gate_out_pix0 + pp;
                        if constexpr (MF16)
                            reinterpret_cast<uint16_t*>(p.gate_out + (int64_t)cd * p.gate_out_stride)[bit0 >> 4] = (uint16_t)(bal >> (16 * lk));
                        else
                            p.gate_out[(int64_t)cd * p.gate_out_stride + (bit0 >> 5)] = (unsigned)(bal >> (32 * lk));
                    }
                }
            }
        }
    }
#undef I2V_FROW
#endif
}

template <int BD, int BP, int WD, int WP, int MODE, bool PREF, bool PRE = false, bool VID = false, bool MF16 = false>
__global__ void __launch_bounds__(256) I2V_CONV_WPE conv_igemm(const I2VConvParams p, const int n_cd_tiles) {
    __shared__ __attribute__((aligned(16))) float smem[conv_lds_floats<BD, BP, WD, MF16, conv_deep(MODE, PREF) ? I2V_DEEP_STAGES : 2>()];
    I2V_PROBE_T probe;
    probe.entry();
    conv_tile<BD, BP, WD, WP, MODE, PREF, PRE, VID, MF16>(p, n_cd_tiles, blockIdx.x, gridDim.x, 0, smem, probe, blockIdx.x);
    probe.exit(blockIdx.x);
}

// MODE 5 launches (halo staging of 3x3 / stride-1 convolutions on planes HWM wide)
template <int HWM>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(HWM <= 14 ? I2V_SMALL_WPE : (HWM <= 28 ? 6 : 5), HWM <= 14 ? I2V_SMALL_WPE : (HWM <= 28 ? 6 : 5))))
conv_igemm_halo(const I2VConvParams p, const int n_cd_tiles) {
    __shared__ __attribute__((aligned(16))) float smem[conv_halo_lds_floats<HWM>()];
    I2V_PROBE_T probe;
    probe.entry();
    conv_tile<64, 64, 2, 2, 5, false, false, false, false, HWM>(p, n_cd_tiles, blockIdx.x, gridDim.x, 0, smem, probe, blockIdx.x);
    probe.exit(blockIdx.x);
}

// =============================================================================================
// Image gradient on a 2-D halo tile (round 5, autotuner bit 9)
// =============================================================================================
// The class-packed gradient w.r.t. the 3-channel input (I2VConvParams::blk / blkt; K order (16-channel group, frame tap, row tap,
// column tap, channel) -- the tap-uniform packing of pack_img) through conv_tile stages, per K chunk, one SHIFTED copy of its
// pixel tile: sixteen 4-byte LDS-DMA instructions per wave beside sixteen 32-cycle MFMAs, and the PMC shows the matrix pipe 0.59 busy --
// the launch is bound by DMA issue.  The TH x TW taps of a (group, frame tap) read the SAME 16 channel planes, so this kernel gives a
// block a 16 x 16 tile of the class grid of ONE grid frame, stages the tile plus its halo once per (group, frame tap) --
// [16 channels][16 + TH - 1 rows][16 + TW - 1 columns], 6 DMA instructions per plane instead of 16 per tap -- and reads every tap's B
// fragments from it at a shifted LDS address.  The weight fragments never enter LDS: a lane loads its own A values (L1 / L2 hits: 64 KB
// shared by every block) four chunks ahead into a register ring.  Same products in the same k order as the conv_tile launch (chunk =
// one tap of 16 channels, 16x16x4 fragments, rows 4s + lk of k-step s): bit-identical, which is what lets the autotuner choose
// between them.  Eligible (conv_ighalo_ok): tap-uniform packing, grid stride 1 (stride-2 stems: B = 2, m = 1), TH, TW <= 4 and
// TH * TW % 4 == 0, at most 32 class rows (TD = 1: 12 of 16 -- image stems, SlowFast's slow stem; TD = 2: 24 of 32 -- I3D's stem).
// Measured (tools/ig_halo_probe.cpp, ResNet's 7x7/2 stem, 128 frames of 224^2, random operands; conv_tile 16x256: 640 us = 47 TFLOP/s
// of algorithmic flops): two halo buffers with the next stage's burst under this stage's MFMAs, 3 blocks per CU: 486-490 us (whatever
// the look-ahead of the B fragments: 1, 2 or 3 k-steps); ONE buffer, two barriers per stage, 6 blocks per CU: 451 us = 67 TFLOP/s --
// what the loop needs is waves per SIMD, not depth per wave (with stores, DMA, weight traffic and LDS reads all removed the
// two-buffer version still took 458 us).  Shipped: one buffer.  The ceiling of this formulation is 157 x 12/16 rows x 49/64 taps x
// ~0.9 (raw fp32 MFMA issue on this part, tools/mfma_rate.cpp) = 80.  The I3D's stem (TD = 2, 24 of 32 rows, 5 of 6 frame taps; 4 blocks per
// CU by registers -- compiled for 5 it spills and gains nothing): 58.5 -> 64.4 TFLOP/s, against a ceiling of 157 x 24/32 x 49/64 x 5/6 x 0.9 = 68.
// (Since pack_img packs a dense temporal stride as ONE LAUNCH PER TEMPORAL CLASS when this kernel is a candidate -- 12 of 16 rows, each class
// its own frame taps, TD = 1 at six blocks per CU: 74 TFLOP/s on that stem -- TD = 2 runs only under I2V_IMG_SPLIT=0.)
static constexpr int IGH_RS = 20, IGH_PL = 400, IGH_NPC = 6;      // LDS row / plane stride in floats (400 % 32 == 16: the four K rows of a
                                                                  // fragment read land on disjoint bank halves), DMA pieces per plane
// QUAD: the "quad rows" packing of a stem with fewer than 16 output channels (SlowFast's fast pathway: 8), K order (channel, frame tap, row
// tap, column tap x 4) with 4 x 4 taps: a K chunk is ONE channel plane of one frame tap, k-step s is row tap s and a lane's K row lk is column
// tap lk.  A stage then holds the TT frame-tap planes of 1, 2 or 4 channels (whichever makes a whole number of four-chunk groups)
// and a chunk moves on by a plane instead of by a tap shift (<= 20 planes: the fast stem's gradient UNPAIRED has five frame taps of four channels).  The zero-weight taps that pad a 7-wide kernel to two quads read real (finite)
// pixels here where conv_tile's MODE 4 substitutes zeros: the product is a zero either way and the chain's value the same.
template <int TD, bool VID, bool QUAD = false>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(TD == 2 ? 4 : QUAD ? 5 : 6, TD == 2 ? 4 : QUAD ? 5 : 6)))      // (QUAD: 32 KB of LDS)
conv_imggrad_halo(const I2VConvParams p, const int tiles_x, const int tiles_xy) {
#if defined(__HIP_DEVICE_COMPILE__)
    constexpr int KC = I2V_KC, RS = IGH_RS, PL = IGH_PL, NPC = IGH_NPC;
    constexpr int NPLMAX = QUAD ? 20 : KC;                                 // planes of a stage (QUAD: five frame taps of four channels)
    __shared__ __attribute__((aligned(16))) float Hb[NPLMAX * PL];         // 25 600 bytes: six blocks per CU (TD = 2: four, by registers); QUAD: 32 000, five
    typedef __attribute__((address_space(3))) float* lds_fp_t;
    constexpr unsigned OOB = 0x80000000u;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int wv = __builtin_amdgcn_readfirstlane(wave);
    const int n16 = lane & 15, lk = lane >> 4;
    // block -> (grid frame, tile): consecutive tiles of a frame on one XCD (their halos overlap in its L2)
    const int nwg = gridDim.x, bid = blockIdx.x;
    const int q8 = nwg >> 3, r8 = nwg & 7, xcd = bid & 7;
    const int lid = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (bid >> 3);
    // (readfirstlane: the quotients are uniform but computed on the vector unit; inline asm takes an "s" operand as it finds it)
    const int ng = __builtin_amdgcn_readfirstlane(lid / tiles_xy), tl = lid - ng * tiles_xy;
    const int ty = __builtin_amdgcn_readfirstlane(tl / tiles_x), tx = tl - ty * tiles_x;
    const int y0 = ty * 16, x0 = tx * 16;
    const int TT = VID ? p.ig_tt : 1, TH = p.ig_th, TW = p.ig_tw, NTs = TH * TW;
    const I2VKEntry e0 = load_kentry(p.ktab, 0);
    const int dh_lo = e0.dh, dw_lo = e0.dw, dt_lo = VID ? (e0.valid >> 1) : 0;
    const int HWs = p.Hs * p.Ws;
    const int cps = !QUAD ? 0 : TT % 4 == 0 ? 1 : TT % 2 == 0 ? 2 : 4;       // QUAD: channels per stage
    const int npl = QUAD ? cps * TT : KC;                                     // planes per stage
    const int nstages = QUAD ? p.Cs / cps : (p.Cs / KC) * TT;
    const int ngroups = QUAD ? npl / 4 : NTs / 4;                             // four-chunk groups per stage
    int clip = ng, ts0 = 0;
    if (VID) { clip = __builtin_amdgcn_readfirstlane(ng / p.Tg); ts0 = (ng - clip * p.Tg) * p.st; }
    const int sframe0 = VID ? clip * p.Ts + ts0 : ng;                      // source frame of frame tap dt = 0
    const int nstr4 = (int)p.src_nstride * 4;
    // Every VMEM instruction of the main loop is inline asm and every vmcnt wait is written by hand: the wave's VMEM queue is a fixed
    // sequence (per tap 4 TD weight loads, per stage one burst of 24 LDS-DMA pieces), so the count that lets exactly the OLDEST ring slot
    // through is a compile-time number.  Left to the compiler (builtins for both), its wait-count pass put `s_waitcnt vmcnt(0)` in front
    // of the first LDS read behind a DMA burst (LDS-DMA may alias any LDS read) and at the head of the tap loop (loop-carried loads),
    // i.e. it drained the queue every four taps.
    // buffer resources as plain 4-dword scalars (what __builtin_amdgcn_make_buffer_rsrc builds: base, stride 0, bytes, raw dword access)
    auto make_rsrc = [](const void* base, const unsigned bytes) {
        const unsigned long long b = (unsigned long long)base;
        return (i2v_v4i){(int)(unsigned)b, (int)(unsigned)((b >> 32) & 0xffffu), (int)bytes, 0x00020000};
    };
    const i2v_v4i rs_w = make_rsrc(p.wp, (unsigned)(p.Kpad * p.Cdpad * 4));
    const i2v_v4i rs_x = make_rsrc(p.src, (unsigned)p.src_span_bytes);
    // halo element e = 64 q + lane of a plane: row e / RS, column e % RS of the staged window, whose corner is source pixel
    // (y0 + dh_lo, x0 + dw_lo); elements outside the window or outside the plane are zero-filled by the range check
    unsigned hoff[NPC];
#pragma unroll
    for (int q = 0; q < NPC; ++q) {
        const int e = 64 * q + lane, r = e / RS, c = e - r * RS;
        const int ys = y0 + dh_lo + r, xs = x0 + dw_lo + c;
        const bool ok = r < 16 + TH - 1 && c < 16 + TW - 1 && (unsigned)ys < (unsigned)p.Hs && (unsigned)xs < (unsigned)p.Ws;
        hoff[q] = ok ? (unsigned)((ys * p.Ws + xs) * 4) : OOB;
    }
    // stage q = (group g, frame tap tt): 16 planes of source frame sframe0 + dt_lo + tt (QUAD: channels q cps .. + cps - 1, plane = (channel,
    // frame tap)); wave w moves planes w, w + 4, w + 8, w + 12 as six pieces each.  A frame tap outside the clip reads nothing: every lane
    // out of range, zeros into the buffer.
    const unsigned ld0 = __builtin_bit_cast(unsigned, (lds_fp_t)Hb) + (unsigned)(wv * PL * 4);
    auto issue_stage = [&](const int q) {
        const int g = QUAD ? 0 : VID ? __builtin_amdgcn_readfirstlane(q / TT) : q, tt0 = QUAD ? 0 : VID ? q - g * TT : 0;
#pragma unroll
        for (int pl = 0; pl < NPLMAX / 4; ++pl) {
            const int plane = wv + 4 * pl;
            if (QUAD && plane >= npl) break;
            const int cl = QUAD ? (plane >= TT) + (plane >= 2 * TT) + (plane >= 3 * TT) : 0;      // QUAD: plane = cl TT + tt
            const int tt = QUAD ? plane - cl * TT : tt0, chan = QUAD ? q * cps + cl : g * KC + plane;
            const bool fok = !VID || (unsigned)(ts0 + dt_lo + tt) < (unsigned)p.Ts;
            const unsigned so = fok ? (unsigned)((sframe0 + dt_lo + tt) * nstr4 + chan * HWs * 4) : 0u;
#pragma unroll
            for (int h = 0; h < NPC; ++h) {
                const unsigned vo = fok ? hoff[h] : OOB;
                asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dword %1, %2, %3 offen lds"
                             :: "s"(__builtin_amdgcn_readfirstlane((int)(ld0 + (unsigned)((4 * pl * PL + 64 * h) * 4)))), "v"(vo), "s"(rs_x),
                                "s"(__builtin_amdgcn_readfirstlane((int)so)) : "memory");
                // (M0 is not on the clobber list -- the compiler rejects reserved registers there -- and need not be: it never keeps a value in
                //  M0 across statements, it sets it immediately in front of each instruction of its own that reads it)
            }
        }
    };
    // A fragments of chunk c: lane (row n16 of fragment i, K row 4 s + lk of k-step s) -> wp[(16 c + 4 s + lk)][16 i + n16]; chunks beyond
    // the last read zeros (range check) -- the ring runs four chunks ahead of the MFMAs to the very end
    const unsigned aoff = (unsigned)((lk * p.Cdpad + n16) * 4);
    auto load_a = [&](const int c, float (&a)[4][TD]) {
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
            for (int i = 0; i < TD; ++i)
                asm volatile("buffer_load_dword %0, %1, %2, %3 offen" : "=v"(a[s][i]) : "v"(aoff), "s"(rs_w), "s"(__builtin_amdgcn_readfirstlane(((c * KC + 4 * s) * p.Cdpad + 16 * i) * 4)) : "memory");
    };
    // `s_waitcnt vmcnt(N)` that the ring slot's registers pass THROUGH: the MFMAs reading them cannot be scheduled in front of it
    auto wait_a = [&]<int N>(std::integral_constant<int, N>, float (&a)[4][TD]) {
        if constexpr (TD == 1) asm volatile("s_waitcnt vmcnt(%4)" : "+v"(a[0][0]), "+v"(a[1][0]), "+v"(a[2][0]), "+v"(a[3][0]) : "n"(N) : "memory");
        else asm volatile("s_waitcnt vmcnt(%8)" : "+v"(a[0][0]), "+v"(a[1][0]), "+v"(a[2][0]), "+v"(a[3][0]), "+v"(a[0][1]), "+v"(a[1][1]), "+v"(a[2][1]), "+v"(a[3][1]) : "n"(N) : "memory");
    };
    f32x4 acc[TD][4];
#pragma unroll
    for (int i = 0; i < TD; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    float ring[4][4][TD];
    issue_stage(0);
#pragma unroll
    for (int u = 0; u < 4; ++u) load_a(u, ring[u]);
    // this lane's B element of fragment j (grid row 4 wave + j, column n16), K row lk (a channel plane; QUAD: a column tap), tap (0, 0)
    const float* const hb = Hb + (QUAD ? lk : lk * PL) + (4 * wave) * RS + n16;
    int chunk = 0;
    // The wave's VMEM queue: a stage ends with the reloads of ring slots 0 .. 3 and, behind the second barrier, the next stage's burst.
    //   stage top   vmcnt(0): the burst -- the youngest thing in the queue -- has landed, and with it all four slots; barrier
    //   taps 0-3    no wait;   taps >= 4: slot u is followed by the three reloads behind it: vmcnt(3 x 4 TD)
    //   stage end   every wave has read its last fragment (lgkmcnt(0), barrier) before the next stage's planes overwrite the buffer
    constexpr int W_IN = 3 * 4 * TD;
    for (int q = 0; q < nstages; ++q) {
#pragma unroll
        for (int u = 0; u < 4; ++u) wait_a(std::integral_constant<int, 0>{}, ring[u]);
        __builtin_amdgcn_s_barrier();
        int th = 0, tw = 0, pli = 0;
        // Four taps = sixteen k-steps as one software pipeline: the B fragments of k-step ks + 1 are requested before the MFMAs of k-step ks
        // (two register sets), across the tap boundaries.
        auto four_taps = [&](auto first_tag) {
            constexpr bool FIRST = decltype(first_tag)::value;
            const float* hp[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                if constexpr (QUAD) hp[u] = hb + (pli++) * PL;
                else { hp[u] = hb + th * RS + tw; if (++tw == TW) { tw = 0; ++th; } }
            }
            float fb[2][4];
            auto rd = [&]<int KS>(std::integral_constant<int, KS>) {
                if constexpr (KS < 16) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) fb[KS % 2][j] = hp[KS / 4][(QUAD ? (KS % 4) * RS : 4 * (KS % 4) * PL) + j * RS];
                }
            };
            rd(std::integral_constant<int, 0>{});
            [&]<int... KS>(std::integer_sequence<int, KS...>) {
                (([&] {
                    constexpr int ks = KS, u = KS / 4, s = KS % 4;
                    rd(std::integral_constant<int, ks + 1>{});
                    if constexpr (s == 0 && !FIRST) wait_a(std::integral_constant<int, W_IN>{}, ring[u]);
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int i = 0; i < TD; ++i)
#pragma unroll
                        for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(ring[u][s][i], fb[ks % 2][j], acc[i][j], 0, 0, 0);
                    if constexpr (s == 3) {
                        // the slot's reload BEHIND the tap's MFMAs, its last readers: the new values may land in the same registers
                        // (issued in front of them, the compiler copied the whole ring at the top of every iteration)
                        __builtin_amdgcn_sched_barrier(0);
                        load_a(chunk + 4, ring[u]);
                        ++chunk;
                    }
                }()), ...);
            }(std::make_integer_sequence<int, 16>{});
        };
        four_taps(std::true_type{});
        for (int gq = 1; gq < ngroups; ++gq) four_taps(std::false_type{});
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (q + 1 < nstages) issue_stage(q + 1);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // (the ring's run-out: nothing may still target a register ...
#pragma unroll
    for (int u = 0; u < 4; ++u)                             //  ... and the run-out loads' registers stay reserved until here)
#pragma unroll
        for (int i = 0; i < TD; ++i) asm volatile("" :: "v"(ring[u][0][i]), "v"(ring[u][1][i]), "v"(ring[u][2][i]), "v"(ring[u][3][i]));
    // ---- epilogue: the class-packed store of conv_tile, element for element ----
    const int HoWo = p.Ho * p.Wo;
    const int bb = p.blk * p.blk, Creal = p.Cd / ((VID ? p.blkt : 1) * bb);
    const int otb = VID ? (ng - clip * p.Tg) * p.ost + p.ot0 : 0;
    const int gj = x0 + n16;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int gi = y0 + 4 * wave + j;
        if (gi >= p.Hg || gj >= p.Wg) continue;
#pragma unroll
        for (int i = 0; i < TD; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int cd = 16 * i + 4 * lk + r;
                if (cd >= p.Cd) continue;
                const int cls3 = cd / Creal, c = cd - cls3 * Creal;
                const int ct = VID ? cls3 / bb : 0, cls = cls3 - ct * bb;
                const int oh = gi * p.osh + cls / p.blk + p.oh0, ow = gj * p.osw + cls % p.blk + p.ow0;
                if (oh >= p.Ho || ow >= p.Wo || (VID && otb + ct * p.oct >= p.To)) continue;
                const int64_t n = VID ? (int64_t)clip * p.To + otb + ct * p.oct : ng;
                const int64_t o = (int64_t)c * HoWo + oh * p.Wo + ow;
                float v = acc[i][j][r];
                if (p.shift) v += p.shift[c];
                if (p.add1) v += p.add1[n * p.add1_nstride + o];
                if (p.relu) v = fmaxf(v, 0.f);
                if (p.mask && !(p.mask[n * p.mask_nstride + o] > 0.f)) v = 0.f;
                p.dst[n * p.dst_nstride + o] = v;
            }
    }
#endif
}
static bool conv_ighalo_ok(const I2VConvParams& p) {
    if (p.ig_th <= 0 || p.ig_tw <= 0 || p.pre_scale || p.sh != 1 || p.sw != 1 || p.blk <= 1 || p.Cd > 32 || p.Kpad != p.K || p.gate || p.gate_out || p.add0 || p.gate_scale) return false;
    const int TT = p.ig_tt > 0 ? p.ig_tt : 1;
    if (p.quad) {       // quad-row order: a chunk is the 4 x 4 taps of one (channel, frame tap) plane
        if (p.quad != 1 || p.ig_th != 4 || p.ig_tw != 4 || p.K != p.Cs * TT * 16) return false;
        const int cps = TT % 4 == 0 ? 1 : TT % 2 == 0 ? 2 : 4;
        return cps * TT <= 20 && p.Cs % cps == 0;
    }
    return p.tap_uniform && p.ig_th <= 4 && p.ig_tw <= 4 && (p.ig_th * p.ig_tw) % 4 == 0 && p.Cs % I2V_KC == 0 && p.K == TT * p.ig_th * p.ig_tw * p.Cs;
}
template <int TD, bool VID, bool QUAD>
static void launch_conv_ighalo_t(const I2VConvParams& p, const int64_t grid, const int tiles_x, const int tiles_xy, hipStream_t s) {
    hipLaunchKernelGGL((conv_imggrad_halo<TD, VID, QUAD>), dim3((unsigned)grid), dim3(256), 0, s, p, tiles_x, tiles_xy);
}
static int launch_conv_ighalo(const I2VConvParams& p, hipStream_t s) {
    const int tiles_x = (p.Wg + 15) / 16, tiles_y = (p.Hg + 15) / 16, txy = tiles_x * tiles_y;
    const int64_t grid = (int64_t)p.N * txy;
    if (grid <= 0) return 0;
    if (grid > 0x7fffffff) { snprintf(g_be_err, sizeof g_be_err, "conv grid too large"); g_be_has_err = true; return 1; }
    const bool two = p.Cd > 16;
    if (p.quad) {       // (quad-row stems exist in video networks only)
        if (two) launch_conv_ighalo_t<2, true, true>(p, grid, tiles_x, txy, s); else launch_conv_ighalo_t<1, true, true>(p, grid, tiles_x, txy, s);
    } else if (p.temporal) {
        if (two) launch_conv_ighalo_t<2, true, false>(p, grid, tiles_x, txy, s); else launch_conv_ighalo_t<1, true, false>(p, grid, tiles_x, txy, s);
    } else {
        if (two) launch_conv_ighalo_t<2, false, false>(p, grid, tiles_x, txy, s); else launch_conv_ighalo_t<1, false, false>(p, grid, tiles_x, txy, s);
    }
    LAUNCH_CHECK("conv_imggrad_halo");
    return 0;
}

// =============================================================================================
// Narrow forward stem on a 2-D halo tile (round 5, autotuner bit 10)
// =============================================================================================
// SlowFast's fast stem (3 -> 8 channels, 5x7x7, spatial stride 2; frame pairs: 16 class rows, K = (channel, frame tap, row tap, column
// quad x 4) = 1008) through conv_tile's MODE 4 re-stages, for every 16-row K chunk, a 256-pixel B tile that only 16 output rows use:
// 16 KB of L2 -> LDS traffic per 64 MFMAs, 41 TFLOP/s of algorithmic flops where the zeros of the packing allow 100 -- the launch is
// bound by the operand fetch, not by the matrix pipe.  The 56 K rows of a (channel, frame tap) read ONE source plane, so this kernel
// gives a block a 16 x 16 tile of output pixels of one grid frame and stages the tile's source window -- 37 rows x 40 columns, its
// left edge moved one pixel out so that rows start 16-byte aligned: six 16-byte DMA instructions per plane -- once per plane; two
// planes (112 K rows = 7 chunks) form a stage, every B fragment address is a compile-time offset from the lane's base, and the seven
// chunks' weight fragments sit in a seven-slot register ring that is reloaded a whole stage ahead (no wait inside a stage).  One
// buffer, two barriers per stage, six blocks per CU, as conv_imggrad_halo.  Same k order, same products (a padded column tap reads a
// real pixel against a zero weight where MODE 4 substitutes a zero): bit-identical to the conv_tile launch.
// Eligible (conv_stemhalo_ok): quad-row packing of a 7 x 7 / stride-2 / pad-3 kernel, <= 16 class rows, plane width a multiple of 4.
static constexpr int SH_RS = 40, SH_WR = 37, SH_NPC = 6, SH_PL = SH_NPC * 256, SH_RPP = 56, SH_CPS = 7;     // window row stride / rows, DMA pieces per plane, plane floats (whole
                                                                                                            // pieces: the last one's zero-filled tail must not land in the next plane),
                                                                                                            // K rows per plane, chunks per stage
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(6, 6)))
conv_stem_halo(const I2VConvParams p, const int tiles_x, const int tiles_xy) {
#if defined(__HIP_DEVICE_COMPILE__)
    constexpr int KC = I2V_KC, RS = SH_RS, PL = SH_PL, NPC = SH_NPC, CPS = SH_CPS;
    __shared__ __attribute__((aligned(16))) float Hb[2 * PL];               // two planes, 12 288 bytes
    typedef __attribute__((address_space(3))) float* lds_fp_t;
    constexpr unsigned OOB = 0x80000000u;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int wv = __builtin_amdgcn_readfirstlane(wave);
    const int n16 = lane & 15, lk = lane >> 4;
    const int nwg = gridDim.x, bid = blockIdx.x;
    const int q8 = nwg >> 3, r8 = nwg & 7, xcd = bid & 7;
    const int lid = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (bid >> 3);
    const int ng = __builtin_amdgcn_readfirstlane(lid / tiles_xy), tl = lid - ng * tiles_xy;
    const int ty = __builtin_amdgcn_readfirstlane(tl / tiles_x), tx = tl - ty * tiles_x;
    const int y0 = ty * 16, x0 = tx * 16;
    const int HWs = p.Hs * p.Ws;
    const int nstages = p.K / (2 * SH_RPP);
    const int clip = __builtin_amdgcn_readfirstlane(ng / p.Tg), ts0 = (ng - clip * p.Tg) * p.st;      // source frame (in the clip) of frame tap 0
    const int nstr4 = (int)p.src_nstride * 4;
    auto make_rsrc = [](const void* base, const unsigned bytes) {
        const unsigned long long b = (unsigned long long)base;
        return (i2v_v4i){(int)(unsigned)b, (int)(unsigned)((b >> 32) & 0xffffu), (int)bytes, 0x00020000};
    };
    const i2v_v4i rs_w = make_rsrc(p.wp, (unsigned)(p.Kpad * p.Cdpad * 4));
    const i2v_v4i rs_x = make_rsrc(p.src, (unsigned)p.src_span_bytes);
    // window piece e = 64 h + lane: row e / 10, columns 4 (e % 10) .. + 3; the window's corner is source pixel (2 y0 - 3, 2 x0 - 4)
    unsigned hoff[NPC];
#pragma unroll
    for (int h = 0; h < NPC; ++h) {
        const int e = 64 * h + lane, r = e / 10, c4 = e - r * 10;
        const int ys = 2 * y0 - 3 + r, xs = 2 * x0 - 4 + 4 * c4;
        const bool ok = r < SH_WR && (unsigned)ys < (unsigned)p.Hs && (unsigned)xs < (unsigned)p.Ws;
        hoff[h] = ok ? (unsigned)((ys * p.Ws + xs) * 4) : OOB;
    }
    // stage q: planes 2 q and 2 q + 1 (plane = (channel, frame tap): k-table row 56 plane); pieces h = wave, wave + 4, wave + 8 of the 12
    const unsigned ld0 = __builtin_bit_cast(unsigned, (lds_fp_t)Hb);
    auto issue_stage = [&](const int q) {
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const int pc = wv + 4 * i, pl = pc >= NPC ? 1 : 0, h = pc - pl * NPC;      // (uniform)
            const I2VKEntry e = load_kentry(p.ktab, (2 * q + pl) * SH_RPP);
            const int dt = e.valid >> 1;
            const bool fok = (unsigned)(ts0 + dt) < (unsigned)p.Ts;
            const unsigned so = fok ? (unsigned)((clip * p.Ts + ts0 + dt) * nstr4 + e.chan_off * 4) : 0u;
            unsigned vo = hoff[0];
#pragma unroll
            for (int hh = 1; hh < NPC; ++hh) vo = h == hh ? hoff[hh] : vo;
            vo = fok ? vo : OOB;
            asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds"
                         :: "s"(__builtin_amdgcn_readfirstlane((int)(ld0 + (unsigned)((pl * PL + 256 * h) * 4)))), "v"(vo), "s"(rs_x),
                            "s"(__builtin_amdgcn_readfirstlane((int)so)) : "memory");
        }
    };
    const unsigned aoff = (unsigned)((lk * p.Cdpad + n16) * 4);
    auto load_a = [&](const int c, float (&a)[4]) {
#pragma unroll
        for (int s = 0; s < 4; ++s)
            asm volatile("buffer_load_dword %0, %1, %2, %3 offen" : "=v"(a[s]) : "v"(aoff), "s"(rs_w), "s"(__builtin_amdgcn_readfirstlane(((c * KC + 4 * s) * p.Cdpad) * 4)) : "memory");
    };
    f32x4 acc[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    float ring[CPS][4];
    issue_stage(0);
#pragma unroll
    for (int u = 0; u < CPS; ++u) load_a(u, ring[u]);
    // this lane's B element: output pixel (4 wave + j, n16) reads window (2 (4 wave + j) + r, 2 n16 + 1 + s4), column tap s4 = 4 quad + lk
    const float* const hb = Hb + (8 * wave) * RS + 2 * n16 + 1 + lk;
    int chunk = 0;
    for (int q = 0; q < nstages; ++q) {
        // every slot was reloaded a stage ago and the burst is the youngest thing in the queue: one wait for all
        asm volatile("s_waitcnt vmcnt(0)" : "+v"(ring[0][0]), "+v"(ring[0][1]), "+v"(ring[0][2]), "+v"(ring[0][3]) :: "memory");
#pragma unroll
        for (int u = 1; u < CPS; ++u)      // (the other slots pass through an empty statement behind the wait: volatile statements keep their order)
            asm volatile("" : "+v"(ring[u][0]), "+v"(ring[u][1]), "+v"(ring[u][2]), "+v"(ring[u][3]) :: "memory");
        __builtin_amdgcn_s_barrier();
        float fb[2][4];
        // k-step ks of the stage: K row 4 ks + lk of the stage's 112 = plane (4 ks) / 56, kernel row r, column quad: all compile-time
        auto rd = [&]<int KS>(std::integral_constant<int, KS>) {
            if constexpr (KS < 4 * CPS) {
                constexpr int row = 4 * KS, pl = row / SH_RPP, rr = row % SH_RPP, r = rr / 8, c0 = rr % 8;
#pragma unroll
                for (int j = 0; j < 4; ++j) fb[KS % 2][j] = hb[pl * PL + (2 * j + r) * RS + c0];
            }
        };
        rd(std::integral_constant<int, 0>{});
        [&]<int... KS>(std::integer_sequence<int, KS...>) {
            (([&] {
                constexpr int ks = KS, u = KS / 4, s = KS % 4;
                rd(std::integral_constant<int, ks + 1>{});
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(ring[u][s], fb[ks % 2][j], acc[j], 0, 0, 0);
                if constexpr (s == 3) {      // the slot's reload (the chunk a stage ahead) behind its last readers
                    __builtin_amdgcn_sched_barrier(0);
                    load_a(chunk + CPS, ring[u]);
                    ++chunk;
                }
            }()), ...);
        }(std::make_integer_sequence<int, 4 * CPS>{});
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (q + 1 < nstages) issue_stage(q + 1);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
    for (int u = 0; u < CPS; ++u) asm volatile("" :: "v"(ring[u][0]), "v"(ring[u][1]), "v"(ring[u][2]), "v"(ring[u][3]));      // (the run-out loads' registers stay reserved until here)
    // ---- epilogue: conv_tile's class-packed store with blk = 1 (row = (frame class, channel)), element for element ----
    const int HoWo = p.Ho * p.Wo;
    const int Creal = p.Cd / p.blkt;
    const int otb = (ng - clip * p.Tg) * p.ost + p.ot0;
    const int gj = x0 + n16;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int gi = y0 + 4 * wave + j;
        if (gi >= p.Hg || gj >= p.Wg) continue;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int cd = 4 * lk + r;
            if (cd >= p.Cd) continue;
            const int ct = cd / Creal, c = cd - ct * Creal;
            const int oh = gi * p.osh + p.oh0, ow = gj * p.osw + p.ow0;
            if (oh >= p.Ho || ow >= p.Wo || otb + ct * p.oct >= p.To) continue;
            const int64_t n = (int64_t)clip * p.To + otb + ct * p.oct;
            const int64_t o = (int64_t)c * HoWo + oh * p.Wo + ow;
            float v = acc[j][r];
            if (p.shift) v += p.shift[c];
            if (p.add1) v += p.add1[n * p.add1_nstride + o];
            if (p.relu) v = fmaxf(v, 0.f);
            if (p.mask && !(p.mask[n * p.mask_nstride + o] > 0.f)) v = 0.f;
            p.dst[n * p.dst_nstride + o] = v;
        }
    }
#endif
}
static bool conv_stemhalo_ok(const I2VConvParams& p) {
    return p.quad == 2 && p.quad_kw == 7 && p.quad_dw0 == -3 && p.sh == 2 && p.sw == 2 && p.blk == 1 && p.blkt == 2 && p.Cd <= 16 && p.Kpad == p.K &&
           p.K % (2 * SH_RPP) == 0 && p.Ws % 4 == 0 && p.src_nstride % 4 == 0 && p.osh == 1 && p.osw == 1 && p.oct == 1 && !p.pre_scale && !p.gate && !p.gate_out &&
           !p.add0 && !p.gate_scale && !p.ig_th;
}
static int launch_conv_stemhalo(const I2VConvParams& p, hipStream_t s) {
    const int tiles_x = (p.Wg + 15) / 16, tiles_y = (p.Hg + 15) / 16, txy = tiles_x * tiles_y;
    const int64_t grid = (int64_t)p.N * txy;
    if (grid <= 0) return 0;
    if (grid > 0x7fffffff || ((uintptr_t)p.src & 15)) { snprintf(g_be_err, sizeof g_be_err, "stem halo launch: grid too large or source not 16-byte aligned"); g_be_has_err = true; return 1; }
    hipLaunchKernelGGL(conv_stem_halo, dim3((unsigned)grid), dim3(256), 0, s, p, tiles_x, txy);
    LAUNCH_CHECK("conv_stem_halo");
    return 0;
}

// ... and the WIDE 7x7 / stride-2 stem of the image backbones (3 -> 64 channels: ResNet, DenseNet, SlowFast's slow pathway; K = (tap,
// channel) = 147; autotuner bit 10 as well).  conv_tile's MODE 0 stages it row by row -- one 4-byte DMA instruction and one k-table row
// per K row and 64 pixels -- and reaches 86 TFLOP/s at 128 frames.  Here a block takes a 16 x 16 tile of output pixels and ALL 64
// channels (sixteen 16x16x4 accumulators per wave), stages the three source planes' windows once (18 DMA instructions of 16 bytes per
// lane), and walks the 147 K rows as 37 fully unrolled k-steps: K row 4 ks + lk is (tap, channel) -> a compile-time LDS offset per
// quarter-wave, no k-table, no barrier in the loop; the weight fragments come through a three-slot register ring.  The epilogue is
// conv_tile's scalar one for 16-pixel fragments (shift, ReLU, 16-bit halves of the 1-bit gate words).  Same chain: bit-identical.
static constexpr int SW_PL = 6 * 256, SW_K = 147, SW_KS = 37, SW_D = 3;      // plane floats (6 pieces), K rows, k-steps, ring depth
static constexpr int stem64_off(int k) {       // LDS offset of K row k = (7 r + s) 3 + c relative to the lane's window corner
    k = k > SW_K - 1 ? SW_K - 1 : k;           // (the 148th row has zero weights: any finite element will do)
    const int tap = k / 3, ci = k % 3, r = tap / 7, sx = tap % 7;
    return ci * SW_PL + r * SH_RS + sx;
}
template <bool VID>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4, 4)))
conv_stem64_halo(const I2VConvParams p, const int tiles_x, const int tiles_xy) {
#if defined(__HIP_DEVICE_COMPILE__)
    constexpr int RS = SH_RS, PL = SW_PL, NPC = 6, D = SW_D;
    __shared__ __attribute__((aligned(16))) float Hb[3 * PL];               // three channel planes, 18 432 bytes
    typedef __attribute__((address_space(3))) float* lds_fp_t;
    constexpr unsigned OOB = 0x80000000u;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int wv = __builtin_amdgcn_readfirstlane(wave);
    const int n16 = lane & 15, lk = lane >> 4;
    const int nwg = gridDim.x, bid = blockIdx.x;
    const int q8 = nwg >> 3, r8 = nwg & 7, xcd = bid & 7;
    const int lid = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (bid >> 3);
    const int ng = __builtin_amdgcn_readfirstlane(lid / tiles_xy), tl = lid - ng * tiles_xy;
    const int ty = __builtin_amdgcn_readfirstlane(tl / tiles_x), tx = tl - ty * tiles_x;
    const int y0 = ty * 16, x0 = tx * 16;
    const int HWs = p.Hs * p.Ws;
    int clip = ng, tg = 0;
    if (VID) { clip = __builtin_amdgcn_readfirstlane(ng / p.Tg); tg = ng - clip * p.Tg; }
    const int sframe = VID ? clip * p.Ts + tg * p.st : ng;
    const bool fok = !VID || tg * p.st < p.Ts;
    auto make_rsrc = [](const void* base, const unsigned bytes) {
        const unsigned long long b = (unsigned long long)base;
        return (i2v_v4i){(int)(unsigned)b, (int)(unsigned)((b >> 32) & 0xffffu), (int)bytes, 0x00020000};
    };
    const i2v_v4i rs_w = make_rsrc(p.wp, (unsigned)(p.Kpad * p.Cdpad * 4));
    const i2v_v4i rs_x = make_rsrc(p.src, (unsigned)p.src_span_bytes);
    // the 18 window pieces (plane pc / 6, piece pc % 6), wave w issuing pc = w, w + 4, ...: piece e = 64 h + lane is row e / 10, columns
    // 4 (e % 10) .. + 3 of the window whose corner is source pixel (2 y0 - 3, 2 x0 - 4)
    const unsigned ld0 = __builtin_bit_cast(unsigned, (lds_fp_t)Hb);
    const unsigned sbase = (unsigned)(sframe * (int)p.src_nstride * 4);
    unsigned vo[5];
#pragma unroll
    for (int i = 0; i < 5; ++i) {
        const int pc = wv + 4 * i, pl = pc / NPC, h = pc - pl * NPC;
        const int e = 64 * h + lane, r = e / 10, c4 = e - r * 10;
        const int ys = 2 * y0 - 3 + r, xs = 2 * x0 - 4 + 4 * c4;
        const bool ok = fok && pc < 3 * NPC && r < SH_WR && (unsigned)ys < (unsigned)p.Hs && (unsigned)xs < (unsigned)p.Ws;
        vo[i] = ok ? (unsigned)((ys * p.Ws + xs) * 4) : OOB;
    }
#pragma unroll
    for (int i = 0; i < 5; ++i) {
        const int pc = wv + 4 * i;
        if (pc >= 3 * NPC) break;
        const int pl = pc / NPC, h = pc - pl * NPC;
        asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds"
                     :: "s"(__builtin_amdgcn_readfirstlane((int)(ld0 + (unsigned)((pl * PL + 256 * h) * 4)))), "v"(vo[i]), "s"(rs_x),
                        "s"(__builtin_amdgcn_readfirstlane((int)(sbase + (unsigned)(pl * HWs * 4)))) : "memory");
    }
    // A fragments of k-step ks: lane (row n16 of fragment i, K row 4 ks + lk) -> wp[4 ks + lk][16 i + n16]
    const unsigned aoff = (unsigned)((lk * p.Cdpad + n16) * 4);
    auto load_a = [&](const int ks, float (&a)[4]) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
            asm volatile("buffer_load_dword %0, %1, %2, %3 offen" : "=v"(a[i]) : "v"(aoff), "s"(rs_w), "s"(__builtin_amdgcn_readfirstlane((4 * ks * p.Cdpad + 16 * i) * 4)) : "memory");
    };
    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
        }
    float ring[D][4];
#pragma unroll
    for (int u = 0; u < D; ++u) load_a(u, ring[u]);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                        // the window pieces have landed (and the ring's first loads)
    // (the pieces' offset registers and the source descriptor stay allocated until here)
    asm volatile("" :: "v"(vo[0]), "v"(vo[1]), "v"(vo[2]), "v"(vo[3]), "v"(vo[4]), "s"(rs_x) : "memory");
    __builtin_amdgcn_s_barrier();
    // this lane's window corner: output pixel (4 wave + j, n16) reads window (2 (4 wave + j) + r, 2 n16 + 1 + s)
    const float* const hb = Hb + (8 * wave) * RS + 2 * n16 + 1;
    float fb[2][4];
    auto rd = [&]<int KS>(std::integral_constant<int, KS>) {
        if constexpr (KS < SW_KS) {
            constexpr int o0 = stem64_off(4 * KS), o1 = stem64_off(4 * KS + 1), o2 = stem64_off(4 * KS + 2), o3 = stem64_off(4 * KS + 3);
            const int off = lk == 0 ? o0 : lk == 1 ? o1 : lk == 2 ? o2 : o3;
            const float* const q = hb + off;
#pragma unroll
            for (int j = 0; j < 4; ++j) fb[KS % 2][j] = q[2 * j * RS];
        }
    };
    rd(std::integral_constant<int, 0>{});
    [&]<int... KS>(std::integer_sequence<int, KS...>) {
        (([&] {
            constexpr int ks = KS, u = KS % D;
            rd(std::integral_constant<int, ks + 1>{});
            // slot u is followed in the queue by the D - 1 younger slots' loads (the ring runs D k-steps ahead to the very end: rows up
            // to 4 (SW_KS + D) - 1 < Kpad = 160 hold zero weights)
            asm volatile("s_waitcnt vmcnt(%4)" : "+v"(ring[u][0]), "+v"(ring[u][1]), "+v"(ring[u][2]), "+v"(ring[u][3]) : "n"(4 * (D - 1)) : "memory");
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(ring[u][i], fb[ks % 2][j], acc[i][j], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            load_a(ks + D, ring[u]);
        }()), ...);
    }(std::make_integer_sequence<int, SW_KS>{});
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    // The ring's run-out loads are never read; named here, BEHIND the wait, their destination registers stay reserved until the data has
    // landed.  (In this fully unrolled loop the compiler sees that they are dead: without this it pointed all of them at one scratch
    // register and handed that register to an accumulator while the loads were still in flight -- a load landing in a live accumulator.)
#pragma unroll
    for (int u = 0; u < D; ++u) asm volatile("" :: "v"(ring[u][0]), "v"(ring[u][1]), "v"(ring[u][2]), "v"(ring[u][3]));
    // ---- epilogue: conv_tile's scalar one on 16-pixel fragments ----
    const int HoWo = p.Ho * p.Wo, HWg = p.Hg * p.Wg;
    const int64_t n = VID ? (int64_t)clip * p.To + tg * p.ost + p.ot0 : ng;
    const bool nok = !VID || tg * p.ost + p.ot0 < p.To;
    const int gj = x0 + n16;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int gi = y0 + 4 * wave + j;
        const bool ok = nok && gi < p.Hg && gj < p.Wg;
        const int64_t pp = (int64_t)ng * HWg + gi * p.Wg + gj;               // grid pixel: this element's bit in a gate row
        float* const dstn = p.dst + n * p.dst_nstride + gi * p.Wo + gj;
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int cd = 16 * i + 4 * lk + r;
                float v = 0.f;
                if (ok) {
                    v = acc[i][j][r];
                    if (p.shift) v += p.shift[cd];
                    if (p.relu) v = fmaxf(v, 0.f);
                    dstn[(int64_t)cd * HoWo] = v;
                }
                if (p.gate_out) {
                    const unsigned long long bal = __ballot(ok && v > 0.f);
                    if (n16 == 0 && nok && gi < p.Hg)
                        reinterpret_cast<uint16_t*>(p.gate_out + (int64_t)cd * p.gate_out_stride)[((int64_t)p.gate_out_pix0 + pp) >> 4] = (uint16_t)(bal >> (16 * lk));
                }
            }
    }
#endif
}
static bool conv_stem64_ok(const I2VConvParams& p) {
    return p.halo == 49 && !p.quad && !p.tap_uniform && !p.pointwise && p.K == SW_K && p.Kpad >= 4 * (SW_KS + SW_D) && p.Cd == 64 && p.sh == 2 && p.sw == 2 &&
           p.blk <= 1 && p.blkt <= 1 && p.Ws % 4 == 0 && p.src_nstride % 4 == 0 && p.osh == 1 && p.osw == 1 && p.oh0 == 0 && p.ow0 == 0 && p.Hg == p.Ho && p.Wg == p.Wo &&
           !p.pre_scale && !p.gate && !p.add0 && !p.add1 && !p.mask && !p.gate_scale && (!p.gate_out || p.Wg % 16 == 0);
}
static int launch_conv_stem64(const I2VConvParams& p, hipStream_t s) {
    const int tiles_x = (p.Wg + 15) / 16, tiles_y = (p.Hg + 15) / 16, txy = tiles_x * tiles_y;
    const int64_t grid = (int64_t)p.N * txy;
    if (grid <= 0) return 0;
    if (grid > 0x7fffffff) { snprintf(g_be_err, sizeof g_be_err, "conv grid too large"); g_be_has_err = true; return 1; }
    if (p.temporal) hipLaunchKernelGGL(conv_stem64_halo<true>, dim3((unsigned)grid), dim3(256), 0, s, p, tiles_x, txy);
    else hipLaunchKernelGGL(conv_stem64_halo<false>, dim3((unsigned)grid), dim3(256), 0, s, p, tiles_x, txy);
    LAUNCH_CHECK("conv_stem64_halo");
    return 0;
}

// =============================================================================================
// Fused pair (round 4): 3x3 convolution -> pointwise convolution over its channels, one launch
// =============================================================================================
// A bottleneck's conv2 (3x3, Cmid -> Cmid) and conv3 (1x1, Cmid -> 4 Cmid, + residual, ReLU) -- and, in the backward pass, the input
// gradient of conv2 (a 3x3 convolution with the flipped filter) followed by the input gradient of conv1 (1x1, Cmid -> 4 Cmid, + the
// residual path's gradient, gate) -- are a matrix-bound launch followed by an HBM-bound one whose only product is the other's operand.
// Unfused, the Cmid-channel intermediate is written and read back, and the two launches cannot overlap: the expand convolution streams
// at ~4.3 TB/s of algorithmic bytes (elementwise kernels reach 4.7-4.9 on this part) with the matrix pipe half idle, then the 3x3 runs
// with HBM idle.  Here a block computes its 64-pixel tile of ALL Cmid intermediate channels (phase 1: conv_tile's own main loop,
// MODE 2 or halo staging; its epilogue -- shift / ReLU / gates -- deposits the tile in LDS as [channel][pixel], which IS the B-operand
// image of a pointwise K loop), then runs the pointwise convolution over its Cout / 64 channel tiles with only the weights staged by
// DMA (phase 2), each through the ordinary dense epilogue.  The intermediate never goes to memory (only its 1-bit gates do), and the
// blocks of a CU are in different phases, so one block's streaming overlaps another's matrix work.  Every output element is the same
// k-ordered chain over the same fp32 values as in the two separate launches: bit-identical (tests/test_gpu_video.py).
// Phase 2 of the fused pair.  The epilogue operands of a channel tile (first addend -- the residual --, and the 1-bit gate words) are
// fetched into registers ONE TILE AHEAD (`prefetch`), tile 0's before phase 1 even starts (conv_fused_kernel): a channel tile's K loop
// is 4-8 chunks, far too short to cover a memory round trip issued at its start.
template <int BD1>
struct PwPre { float4 a0[2][4]; unsigned gw[2][4]; };
// (PT: the parameter block is read through a pointer into the kernel-argument segment -- constant address space, scalar loads -- that
// conv_fused_kernel launders per use: the fields are then loaded where they are needed and die there.  Named as a by-value argument
// next to phase 1's block, its ~40 live scalars pushed the kernel over the 102-SGPR file and the spills into VGPRs cost two blocks
// per CU.)
template <int BD1, typename PT>
__device__ __forceinline__ void conv_pw_prefetch(const PT& p, const int64_t px0, const int ct, float4 (&a0)[4], unsigned (&gw)[4]) {
    const int t = threadIdx.x;
    const int HWg = p.Hg * p.Wg;
    const int64_t P = (int64_t)p.N * HWg;
    const int e_c4 = t % 16, e_rbase = t / 16;
    const int64_t e_pp = px0 + (int64_t)e_c4 * 4;
    const bool e_ok = e_pp < P;
    const int64_t e_n = e_ok ? fastdiv((unsigned)e_pp, p.dv_hw_m, p.dv_hw_s) : 0;
    const int64_t e_poff = e_pp - e_n * HWg;
    const int e_HoWo = p.Ho * p.Wo;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int row = e_rbase + q * 16;
        const int cd = ct * 64 + (row >> 5) * 32 + (row & 31);
        const bool ok = e_ok && cd < p.Cd;
        const int64_t o = (int64_t)cd * e_HoWo + e_poff;
        // Every wave issues the SAME number of loads (a lane outside the launch reads element 0 and discards it): the K loop's first
        // wait counts them (conv_pw_from_lds), and a wave whose lanes are all outside must not come up short.
        if (p.add0) { const float4 v = *reinterpret_cast<const float4*>(p.add0 + (ok ? e_n * p.add0_nstride + o : 0)); a0[q] = ok ? v : make_float4(0.f, 0.f, 0.f, 0.f); }
        else a0[q] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (p.gate) { const unsigned v = p.gate[ok ? (int64_t)cd * p.gate_stride + ((p.gate_pix0 + e_pp) >> 5) : 0]; gw[q] = ok ? v : 0xffffffffu; }
        else gw[q] = 0xffffffffu;
    }
}
typedef const __attribute__((address_space(4))) I2VConvParams I2VConvParamsK;
__device__ __forceinline__ I2VConvParamsK* conv_second_kernarg() {           // the SECOND I2VConvParams of conv_fused_kernel's argument list
#if defined(__HIP_DEVICE_COMPILE__)
    unsigned long long v = (unsigned long long)__builtin_amdgcn_kernarg_segment_ptr() + sizeof(I2VConvParams);
    asm volatile("" : "+s"(v));                                               // no load is hoisted or shared across this point
    return (I2VConvParamsK*)v;
#else
    return nullptr;
#endif
}
template <int BD1>
__device__ __forceinline__ void conv_pw_from_lds(const int64_t px0, const float* const mid, float* const smem,
                                                 float4 (&pa)[2][4], unsigned (&pg)[2][4]) {
#if defined(__HIP_DEVICE_COMPILE__)
    constexpr int KC = I2V_KC, NCH = BD1 / KC, KS = KC / 2;
    typedef __attribute__((address_space(3))) void* lds_ptr_t;
    float (*As)[KC][64] = reinterpret_cast<float (*)[KC][64]>(smem);          // [2][16][64], under the epilogue's transpose buffer
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6, wd = wave >> 1, wpx = wave & 1;
    const int l31 = lane & 31, lk = lane >> 5;
    const int wv = __builtin_amdgcn_readfirstlane(wave);
    const int f = wave * 256 + lane * 4;                                       // this wave's quarter of a [16][64] chunk image
    const int n_ct = (conv_second_kernarg()->Cd + 63) / 64;
    const float* const mb = mid + lk * 64 + wpx * 32 + l31;                   // this lane's B element of k-step 0, chunk 0
    auto tile = [&](const int ct, auto set_tag) {
        constexpr int set = decltype(set_tag)::value;
        const I2VConvParamsK& p = *conv_second_kernarg();
        const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc((void*)p.wp, 0, p.Kpad * p.Cdpad * 4, 0x00020000);
        const unsigned aoff0 = (unsigned)(((f / 64) * p.Cdpad + f % 64) * 4);
        const int cd0 = ct * 64;
        f32x16 acc[1][1];
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[0][0][r] = 0.f;
        __syncthreads();                                                      // the previous tile's epilogue has left the transpose buffer
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_w, (lds_ptr_t)(&As[0][0][0] + wv * 256), 16, aoff0 + (unsigned)(cd0 * 4), 0, 0, 0);
        const bool more = ct + 1 < n_ct;
        if (more) conv_pw_prefetch<BD1>(p, px0, ct + 1, pa[set ^ 1], pg[set ^ 1]);              // the NEXT tile's addend / gates
        const int younger = more ? (p.add0 ? 4 : 0) + (p.gate ? 4 : 0) : 0;                       // loads issued behind the chunk-0 DMA
        [&]<int... CC>(std::integer_sequence<int, CC...>) {
            (([&] {
                constexpr int c = CC, buf = CC & 1;
                // the weight chunk is the wave's oldest-but-(prefetch) load: the prefetched operands may stay in flight
                if (c == 0 && younger == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
                else if (c == 0 && younger == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
                else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
                float fa[2], fb[2];
                fa[0] = As[buf][lk][wd * 32 + l31]; fb[0] = mb[c * KC * 64];
                [&]<int... S>(std::integer_sequence<int, S...>) {
                    (([&] {
                        constexpr int st = S, set2 = S & 1;
                        if constexpr (st + 1 < KS) { fa[set2 ^ 1] = As[buf][2 * (st + 1) + lk][wd * 32 + l31]; fb[set2 ^ 1] = mb[(c * KC + 2 * (st + 1)) * 64]; }
                        __builtin_amdgcn_sched_barrier(0);
                        acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[set2], fb[set2], acc[0][0], 0, 0, 0);
                        if constexpr (st == 0 && c + 1 < NCH)                  // the next chunk's weights, behind the first MFMA
                            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_w, (lds_ptr_t)(&As[buf ^ 1][0][0] + wv * 256), 16, aoff0 + (unsigned)(cd0 * 4),
                                                                     (c + 1) * KC * p.Cdpad * 4, 0, 0);
                    }()), ...);
                }(std::make_integer_sequence<int, KS>{});
            }()), ...);
        }(std::make_integer_sequence<int, NCH>{});
        conv_vec_epilogue<64, 64, 2, 2, true, false, false>(p, acc, cd0, px0, smem, pa[set], pg[set], nullptr);
    };
    for (int ct = 0; ct < n_ct; ct += 2) {
        tile(ct, std::integral_constant<int, 0>{});
        if (ct + 1 < n_ct) tile(ct + 1, std::integral_constant<int, 1>{});
    }
#endif
}

// LDS of the fused kernel: [phase-1 staging | phase-2 weight staging + transpose buffer] + the intermediate tile [BD1][64]
template <int BD1, int HWM>
constexpr int conv_fused_stage_floats() {
    constexpr int st1 = HWM ? conv_halo_lds_floats<HWM>() : conv_lds_floats<BD1, 64, 2, false>();
    return st1 > 64 * 64 ? st1 : 64 * 64;
}
template <int BD1, int HWM> constexpr int conv_fused_wpe() { return BD1 == 128 ? 2 : HWM == 56 ? 3 : 4; }      // bounded by LDS (two parameter blocks cost ~100 VGPRs: 5 would spill)
template <int BD1, int HWM>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(conv_fused_wpe<BD1, HWM>(), conv_fused_wpe<BD1, HWM>())))
conv_fused_kernel(const I2VConvParams p1, const I2VConvParams /* p2: read through conv_second_kernarg() */) {
    __shared__ __attribute__((aligned(16))) float smem[conv_fused_stage_floats<BD1, HWM>() + BD1 * 64];
    float* const mid = smem + conv_fused_stage_floats<BD1, HWM>();
    I2V_PROBE_T probe;
    probe.entry();
    // the pixel tile conv_tile takes (its XCD-aware remap with one channel tile per pixel tile)
    const int nwg = gridDim.x, bid = blockIdx.x, q8 = nwg >> 3, r8 = nwg & 7, xcd = bid & 7;
    const int lid = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (bid >> 3);
    float4 pa[2][4]; unsigned pg[2][4];
    conv_pw_prefetch<BD1>(*conv_second_kernarg(), (int64_t)lid * 64, 0, pa[0], pg[0]);      // phase 2's first addend / gate tile rides under all of phase 1
    conv_tile<BD1, 64, 2, 2, HWM ? 5 : 2, false, false, false, false, HWM, 1, 1>(p1, 1, blockIdx.x, gridDim.x, 0, smem, probe, blockIdx.x, I2V_PRIO_LEVELS, mid);
    __syncthreads();                                                          // the whole intermediate tile is in LDS
    conv_pw_from_lds<BD1>((int64_t)lid * 64, mid, smem, pa, pg);
    probe.exit(blockIdx.x);
}

// Several chunks per barrier (conv_tile, CPB): the plain 64x64 image tile with 32-row LDS buffers -- 32 KB, 5 resident blocks.
#ifndef I2V_DC_WPE
#define I2V_DC_WPE 5
#endif
template <int MODE, int CPB>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(I2V_DC_WPE, I2V_DC_WPE)))
conv_igemm_dc(const I2VConvParams p, const int n_cd_tiles) {
    __shared__ __attribute__((aligned(16))) float smem[conv_lds_floats<64, 64, 2, false, 2, CPB>()];
    I2V_PROBE_T probe;
    probe.entry();
    conv_tile<64, 64, 2, 2, MODE, false, false, false, false, 0, CPB>(p, n_cd_tiles, blockIdx.x, gridDim.x, 0, smem, probe, blockIdx.x);
    probe.exit(blockIdx.x);
}

// Split-bf16 arithmetic (conv_tile, BF3): the plain pointwise / tap-uniform image tiles on three-term bf16 operands.
#ifndef I2V_BF3_VARIANT
#define I2V_BF3_VARIANT 1        // 1: weight fragments staged through LDS by DMA, I2V_BF3_STAGES buffers; 2: loaded straight into registers, one chunk ahead
#endif
template <int BD, int BP, int WD, int WP, int MODE, int CPB, bool VID = false, int VAR = I2V_BF3_VARIANT>
__global__ void __launch_bounds__(256) conv_igemm_bf3(const I2VConvParams p, const int n_cd_tiles) {
    __shared__ __attribute__((aligned(16))) float smem[conv_lds_floats<BD, BP, WD, false, VAR == 3 ? 3 : VAR == 1 ? I2V_BF3_STAGES : 2, CPB, VAR>()];
    I2V_PROBE_T probe;
    probe.entry();
    conv_tile<BD, BP, WD, WP, MODE, false, false, VID, false, 0, CPB, 0, VAR>(p, n_cd_tiles, blockIdx.x, gridDim.x, 0, smem, probe, blockIdx.x);
    probe.exit(blockIdx.x);
}
static bool conv_bf3_ok(const I2VConvParams& p) {      // (temporal launches -- video networks' k x 1 x 1 and strided convolutions -- only as tap-uniform ones: the staging of MODE 2, VID)
    return p.bf3 && p.wp3 && (p.pointwise || p.tap_uniform) && (!p.temporal || p.tap_uniform) && !p.pre_scale && !p.quad && p.blk <= 1 && p.blkt <= 1 && p.Cd > 32;
}

// "Tail split": the first `nA` blocks compute 64x64 tiles over the pixel tiles [0, px_base_b / 64); the remaining blocks cover the
// rest of the pixels with 16x64 tiles on 16x16x4 fragments (a quarter of the work each).  A launch of 6.125 tiles per CU leaves
// 32 CUs with 7 tiles and 224 with 6; cut this way it is 6 tiles everywhere plus 128 quarter tiles on 128 CUs.  Every output
// element is still the same k-ordered fmaf chain (fragment shape does not enter): results are bit-identical.
template <int MODE>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(I2V_SMALL_WPE, I2V_SMALL_WPE)))
conv_igemm_tail(const I2VConvParams p, const int n_cd_a, const int nA, const int n_cd_b, const int64_t px_base_b) {
    constexpr int LA = conv_lds_floats<64, 64, 2, false>(), LB = conv_lds_floats<16, 64, 1, true>();
    __shared__ __attribute__((aligned(16))) float smem[LA > LB ? LA : LB];
    I2V_PROBE_T probe;
    probe.entry();
    const int slot = (int)blockIdx.x < nA ? (int)blockIdx.x : 65536 + (int)blockIdx.x - nA;      // (probe builds: quarter tiles from slot 65536 on)
    // The quarter tiles are dispatched last, i.e. they are the youngest waves of their CU: served last, they used to finish last
    // and alone (timeline: K loops of 238-248 us next to full tiles done at 229).  They run ABOVE every level the full tiles
    // use instead, are done in a quarter of a tile time and leave the CU to its six full tiles.
    if ((int)blockIdx.x < nA) conv_tile<64, 64, 2, 2, MODE, false, false, false, false>(p, n_cd_a, blockIdx.x, nA, 0, smem, probe, slot, I2V_PRIO_LEVELS > 2 ? 2 : I2V_PRIO_LEVELS);
    else {
        if (I2V_PRIO_LEVELS > 0) __builtin_amdgcn_s_setprio(3);
        conv_tile<16, 64, 1, 4, MODE, false, false, false, true>(p, n_cd_b, (int)blockIdx.x - nA, (int)gridDim.x - nA, px_base_b, smem, probe, slot, 0);
        if (I2V_PRIO_LEVELS > 0) __builtin_amdgcn_s_setprio(0);
    }
    probe.exit(slot);
}

// =============================================================================================
// Persistent, role-split pointwise kernel (round 5): conv_pw_stream
// =============================================================================================
// The short-K pointwise launches (64 -> 256 @56^2, 128 -> 512 @28^2 and their input gradients) run a 4-8 chunk K loop and then a
// byte-heavy epilogue, serially inside every block of conv_igemm; seven co-resident blocks overlap the two only statistically
// (PMC: matrix pipe 0.58 busy, HBM at 0.43 of its peak -- on neither roof).  Here ONE 768-thread workgroup per CU owns a 64-channel
// tile for the whole launch and walks its share of the pixel tiles:
//   * the [K][64] weight panel is staged into LDS ONCE and stays;
//   * waves 0-3 ("matrix waves", one per SIMD) only read LDS and issue MFMAs -- K / 2 of them back to back per tile, no wait inside a
//     tile; at the end of a tile they deposit the accumulators transposed into one of two [64][64] hand-off buffers;
//   * waves 8-11 ("loader waves") issue the LDS-DMA of the activation tiles into a ring, NBUF - 1 slabs ahead of the matrix waves (the
//     first version had the matrix waves issue them behind their MFMAs, as conv_tile does: with ONE matrix wave per SIMD every DMA issue
//     stall -- 60-185 cycles against an MFMA's 64 -- idled the pipe, and a ring one tile deep exposed the memory latency every tile:
//     45 / 70 / 89 TFLOP/s on 64 -> 256 / 128 -> 512 / 256 -> 1024 against conv_igemm's 69 / 108 / 117; tools/pw_stream_probe.cpp);
//   * waves 4-7 ("epilogue waves") meanwhile drain the PREVIOUS tile's hand-off buffer through conv_vec_rows -- addend / gate words
//     prefetched into registers up to three tiles ahead, shift, ReLU, gates, 16-byte stores;
//   * ONE s_barrier per tile (per 128-row slab) hands the buffers over: ring slot full / free, hand-off buffer full / free.
// So a CU's matrix pipe, its HBM reads (activations, addend) and its stores run concurrently by construction, not by luck of block
// phases.  Every output element is the same k-ordered fmaf chain over the same values as in conv_igemm (a 32x32x2 fp32 MFMA is a
// sequential chain along K whatever feeds it), and the row pass IS conv_igemm's: bit-identical.
// Blocks b, b + 8, ... share an XCD: the n_cd channel-tile blocks of one pixel-tile stream are neighbours there, so a pixel tile is
// fetched from HBM once and served to the other n_cd - 1 blocks by that XCD's L2.
// K = 256 (256 -> 1024 @14^2): the activations of a tile arrive as two 128-row SLABS through the same two-slot ring -- one barrier per
// slab --, the 64 KB weight panel stays whole: 160 KB of LDS, all a workgroup may have.
template <int K> constexpr int pws_slab() { return K < 128 ? K : 128; }                   // K rows per ring slot
template <int K> constexpr int pws_nbuf() { return K <= 64 ? 4 : K <= 128 ? 3 : 2; }       // activation ring slots ([slab][64] floats each): all of the 160 KB
template <int K> constexpr int pws_lds_floats() { return K * 64 + pws_nbuf<K>() * pws_slab<K>() * 64 + 2 * 64 * 64; }
#ifndef I2V_PWS_NSET
#define I2V_PWS_NSET 2                // epilogue-operand register sets (tiles of addend / gate words in flight per epilogue wave)
#endif
#define I2V_PWS_THREADS 768           // 4 matrix waves + 4 epilogue waves + 4 loader waves: three waves per SIMD
#ifdef I2V_PWS_STAMPS      // diagnostic build (tools/pw_stream_probe.cpp -DI2V_PWS_STAMPS): per block, 100 MHz ticks summed over its tiles
__device__ unsigned long long g_pws_stamps[256 * 8];
#define PWS_NOW() __builtin_amdgcn_s_memrealtime()
#define PWS_ACC(var, t0_) (var) += PWS_NOW() - (t0_)
#else
#define PWS_NOW() 0ull
#define PWS_ACC(var, t0_) ((void)(t0_))
#endif
template <int K>
__global__ void __launch_bounds__(I2V_PWS_THREADS) conv_pw_stream(const I2VConvParams p, const int n_cd, const int n_streams, const int n_px_tiles) {
#if defined(__HIP_DEVICE_COMPILE__)
    constexpr int KSL = pws_slab<K>(), S = K / KSL, NBUF = pws_nbuf<K>(), NSET = I2V_PWS_NSET, KS = KSL / 2, NPW = KSL / 16;     // NPW: DMA pieces per loader wave per slab
    static_assert(K % KSL == 0, "whole slabs");
    __shared__ __attribute__((aligned(16))) float smem[pws_lds_floats<K>()];
    float* const Wl = smem;                                   // [K][64]   weight panel of this block's channel tile
    float* const Bl = smem + K * 64;                          // [NBUF][KSL][64] activation ring
    float (*const Cl)[64][64] = reinterpret_cast<float (*)[64][64]>(smem + K * 64 + NBUF * KSL * 64);      // [2][64][64] hand-off buffers
    typedef __attribute__((address_space(3))) void* lds_ptr_t;
    const int t = threadIdx.x, lane = t & 63;
    const int wv = __builtin_amdgcn_readfirstlane(t >> 6);
    const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
    const int cd0 = (j % n_cd) * 64;
    const int stream = xcd * ((int)(gridDim.x >> 3) / n_cd) + j / n_cd;
    const int n_mine = stream < n_px_tiles ? (n_px_tiles - stream + n_streams - 1) / n_streams : 0;
    const int HWg = p.Hg * p.Wg;
    const int64_t P = (int64_t)p.N * HWg;
    auto px_of = [&](const int i) { return ((int64_t)stream + (int64_t)i * n_streams) * 64; };
    // Barrier #0 follows the weight panel and the ring's first slab; barrier #(g + 1) ends slab-phase g (g = i S + h: slab h of tile i):
    // by then the matrix waves are done with slab g (its ring slot is free) and, at a tile's last slab, have deposited the tile; the loader
    // waves have seen slab g + 1 land; the epilogue waves have finished reading the hand-off buffer of tile i - 1.
    if (wv >= 8) {
        // ------------------------------------------------------------------ loader waves: LDS-DMA only, so their vmcnt is exact
        constexpr unsigned OOB = 0x80000000u;
        const int lw = wv - 8;
        const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc((void*)p.wp, 0, p.Kpad * p.Cdpad * 4, 0x00020000);
        const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc((void*)p.src, 0, p.src_span_bytes, 0x00020000);
        const int HWs = p.Hs * p.Ws;
        // piece `ins` of a [rows][64] image = its rows 4 ins .. 4 ins + 3; lane l moves 16 bytes: row 4 ins + l / 16, columns 4 (l % 16) ..
        const unsigned aoff = (unsigned)(((lane >> 4) * p.Cdpad + cd0 + (lane & 15) * 4) * 4);
        auto issue_slab = [&](const int g, const int slot_) {                             // this wave's NPW pieces of slab g % S of tile g / S
            const int i = g / S, h = g - i * S;
            const int64_t pp = px_of(i) + (lane & 15) * 4;
            unsigned bo = OOB;                                                            // beyond this block's tiles / the launch: zero fill, same counts
            if (i < n_mine && pp < P) {
                const int64_t n = fastdiv((unsigned)pp, p.dv_hw_m, p.dv_hw_s);
                bo = (unsigned)((n * p.src_nstride + (pp - n * HWg) + (int64_t)(lane >> 4) * HWs) * 4);
            }
#pragma unroll
            for (int q = 0; q < NPW; ++q) {
                const int ins = lw + 4 * q;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_x, (lds_ptr_t)(Bl + slot_ * (KSL * 64) + ins * 256), 16, bo, (h * KSL + ins * 4) * HWs * 4, 0, 0);
            }
        };
#pragma unroll
        for (int q = 0; q < K / 16; ++q) {
            const int ins = lw + 4 * q;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_w, (lds_ptr_t)(Wl + ins * 256), 16, aoff, ins * 4 * p.Cdpad * 4, 0, 0);
        }
#pragma unroll
        for (int g = 0; g < NBUF - 1; ++g) issue_slab(g, g);
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NBUF - 2) * NPW) : "memory");           // the panel and slab 0
        __builtin_amdgcn_s_barrier();                                                     // #0
        int slot = NBUF - 1;                                                              // ring slot of slab g + NBUF - 1
        const int n_slabs = n_mine * S;
        unsigned long long l_issue = 0, l_wait = 0, l_bar = 0; (void)l_issue; (void)l_wait; (void)l_bar;
        for (int g = 0; g < n_slabs; ++g) {
            unsigned long long ts = PWS_NOW();
            issue_slab(g + NBUF - 1, slot);                                               // its slot held slab g - 1: free since barrier #g
            PWS_ACC(l_issue, ts); ts = PWS_NOW();
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NBUF - 2) * NPW) : "memory");       // slab g + 1 has landed (g + 2 .. may still fly)
            PWS_ACC(l_wait, ts); ts = PWS_NOW();
            __builtin_amdgcn_s_barrier();                                                 // #(g + 1)
            PWS_ACC(l_bar, ts);
            slot = slot + 1 == NBUF ? 0 : slot + 1;
        }
#ifdef I2V_PWS_STAMPS
        if (t == 512 && blockIdx.x < 256) { g_pws_stamps[8 * blockIdx.x + 5] = l_issue; g_pws_stamps[8 * blockIdx.x + 6] = l_wait; g_pws_stamps[8 * blockIdx.x + 7] = l_bar; }
#endif
    } else if (wv < 4) {
        // ------------------------------------------------------------------ matrix waves: LDS reads and MFMAs, nothing else
        const int wd = wv >> 1, wpx = wv & 1, l31 = lane & 31, lk = lane >> 5;
        __builtin_amdgcn_s_barrier();                                                     // #0
        int slot = 0;
        unsigned long long m_loop = 0, m_bar = 0; (void)m_loop; (void)m_bar;
        for (int i = 0; i < n_mine; ++i) {
            f32x16 acc;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = 0.f;
            [&]<int... H>(std::integer_sequence<int, H...>) {
                (([&] {
                    constexpr int h = H;
                    unsigned long long ts = PWS_NOW();
                    const float* const wbase = Wl + (h * KSL + lk) * 64 + wd * 32 + l31;
                    const float* const bbase = Bl + slot * (KSL * 64) + lk * 64 + wpx * 32 + l31;
                    float fa[3], fb[3];
                    fa[0] = wbase[0]; fb[0] = bbase[0];
                    fa[1] = wbase[128]; fb[1] = bbase[128];
                    [&]<int... SS>(std::integer_sequence<int, SS...>) {
                        (([&] {
                            constexpr int s_ = SS, cur = SS % 3, nx2 = (SS + 2) % 3;
                            if constexpr (s_ + 2 < KS) { fa[nx2] = wbase[(s_ + 2) * 128]; fb[nx2] = bbase[(s_ + 2) * 128]; }
                            __builtin_amdgcn_sched_barrier(0);
                            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[cur], fb[cur], acc, 0, 0, 0);
                        }()), ...);
                    }(std::make_integer_sequence<int, KS>{});
                    if constexpr (h == S - 1) {      // hand the tile over: accumulators transposed into the hand-off buffer (conv_vec_epilogue's deposit)
                        float (*const Cs)[64] = Cl[i & 1];
#pragma unroll
                        for (int r = 0; r < 16; ++r) Cs[wd * 32 + (r & 3) + 8 * (r >> 2) + 4 * lk][wpx * 32 + l31] = acc[r];
                    }
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    PWS_ACC(m_loop, ts); ts = PWS_NOW();
                    __builtin_amdgcn_s_barrier();                                         // #(i S + h + 1)
                    PWS_ACC(m_bar, ts);
                    slot = slot + 1 == NBUF ? 0 : slot + 1;
                }()), ...);
            }(std::make_integer_sequence<int, S>{});
        }
#ifdef I2V_PWS_STAMPS
        if (t == 0 && blockIdx.x < 256) { g_pws_stamps[8 * blockIdx.x + 0] = m_loop; g_pws_stamps[8 * blockIdx.x + 1] = m_bar; }
#endif
    } else {
        // ------------------------------------------------------------------ epilogue waves
        const int te = t - 256;
        float4 pa[NSET][4]; unsigned pg[NSET][4];
        auto prefetch = [&](const int i, float4 (&a0)[4], unsigned (&gw)[4]) {           // tile i's first addend and gate words (conv_tile's PREF)
            const int e_c4 = te & 15, e_rbase = te >> 4;
            const int64_t e_pp = px_of(i) + (int64_t)e_c4 * 4;
            const bool e_ok = i < n_mine && e_pp < P;
            const int64_t e_n = e_ok ? fastdiv((unsigned)e_pp, p.dv_hw_m, p.dv_hw_s) : 0;
            const int64_t e_poff = e_pp - e_n * HWg;
            const int e_HoWo = p.Ho * p.Wo;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int cd = cd0 + e_rbase + q * 16;
                const bool ok = e_ok && cd < p.Cd;
                const int64_t o = (int64_t)cd * e_HoWo + e_poff;
                a0[q] = (ok && p.add0) ? *reinterpret_cast<const float4*>(p.add0 + e_n * p.add0_nstride + o) : make_float4(0.f, 0.f, 0.f, 0.f);
                gw[q] = (ok && p.gate) ? p.gate[(int64_t)cd * p.gate_stride + ((p.gate_pix0 + e_pp) >> 5)] : 0xffffffffu;
            }
        };
        // The row pass: conv_vec_rows' expressions in conv_vec_rows' order (shift, addend, ReLU, gate bits, store, own gate word) for the
        // launches this kernel admits (no second addend, no fp32 mask, no pre-activation gate: conv_pws_grid), with every operand
        // already in a register.  conv_vec_rows itself reads `shift[cd]` inside its row loop behind an `s_waitcnt vmcnt(0)` -- harmless
        // among seven co-resident blocks, but here ONE epilogue wave per SIMD is the critical path: each of its four rows then waited
        // for the previous row's store to be acknowledged (3-4 us per tile against 1.9 us of MFMAs: the probe's first two versions).  A
        // thread's four channel rows are the same for every tile, so their shifts are loaded once.
        const bool has_shift = p.shift != nullptr, has_gate = p.gate != nullptr, has_gout = p.gate_out != nullptr, relu = p.relu != 0;
        const bool nt_store = p.cfg > 0 && ((p.cfg - 1) & 128);
        float shv[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) { const int cd = cd0 + (te >> 4) + q * 16; shv[q] = (has_shift && cd < p.Cd) ? p.shift[cd] : 0.f; }
        const int HoWo_ = p.Ho * p.Wo;
        auto rows = [&](const int64_t px0, const float (*const Cs)[64], const float4 (&a0)[4], const unsigned (&gw)[4]) {
            const int c4 = te & 15, rbase = te >> 4;
            const int64_t pp = px0 + (int64_t)c4 * 4;
            const bool pok = pp < P;
            const int64_t n = pok ? fastdiv((unsigned)pp, p.dv_hw_m, p.dv_hw_s) : 0;
            const int64_t poff = pp - n * HWg;
            float* const drow = p.dst + n * p.dst_nstride + poff;
            const unsigned gsh = (unsigned)(p.gate_pix0 + pp) & 31u;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int row = rbase + q * 16, cd = cd0 + row;
                const bool valid = pok && cd < p.Cd;
                float4 v = *reinterpret_cast<const float4*>(&Cs[row][c4 * 4]);
                if (has_shift) { const float sh = shv[q]; v.x += sh; v.y += sh; v.z += sh; v.w += sh; }
                v.x += a0[q].x; v.y += a0[q].y; v.z += a0[q].z; v.w += a0[q].w;
                if (relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
                if (has_gate) {
                    const unsigned g = gw[q] >> gsh;
                    if (!(g & 1u)) v.x = 0.f;
                    if (!(g & 2u)) v.y = 0.f;
                    if (!(g & 4u)) v.z = 0.f;
                    if (!(g & 8u)) v.w = 0.f;
                }
                if (valid) {
                    float* const d = drow + (int64_t)cd * HoWo_;
                    if (nt_store) { typedef float nt4 __attribute__((ext_vector_type(4))); const nt4 w4 = {v.x, v.y, v.z, v.w};
                                    __builtin_nontemporal_store(w4, reinterpret_cast<nt4*>(d)); }
                    else *reinterpret_cast<float4*>(d) = v;
                }
                if (has_gout) {      // this tensor's own gates: 8 consecutive lanes hold 32 consecutive pixels of one channel row (conv_vec_rows)
                    unsigned nib = valid ? ((v.x > 0.f ? 1u : 0u) | (v.y > 0.f ? 2u : 0u) | (v.z > 0.f ? 4u : 0u) | (v.w > 0.f ? 8u : 0u)) : 0u;
                    nib <<= 4 * (lane & 7);
                    nib |= (unsigned)__builtin_amdgcn_update_dpp(0, (int)nib, 0xB1, 0xF, 0xF, true);
                    nib |= (unsigned)__builtin_amdgcn_update_dpp(0, (int)nib, 0x4E, 0xF, 0xF, true);
                    nib |= (unsigned)__builtin_amdgcn_update_dpp(0, (int)nib, 0x141, 0xF, 0xF, true);
                    if (valid && (lane & 7) == 0) p.gate_out[(int64_t)cd * p.gate_out_stride + ((p.gate_out_pix0 + pp) >> 5)] = nib;
                }
            }
        };
        [&]<int... T>(std::integer_sequence<int, T...>) { ((prefetch(T, pa[T], pg[T])), ...); }(std::make_integer_sequence<int, NSET>{});
        __builtin_amdgcn_s_barrier();                                                     // #0
        unsigned long long e_rows = 0, e_pref = 0, e_bar = 0; (void)e_rows; (void)e_pref; (void)e_bar;
        // phase i: the rows of tile i - 1 (deposited before barrier #(i S)), then the prefetch of tile i - 1 + NSET into the set just freed
        for (int i0 = 0; i0 <= n_mine; i0 += NSET) {
            [&]<int... U>(std::integer_sequence<int, U...>) {
                (([&] {
                    constexpr int u = U, set = (U + NSET - 1) % NSET;                     // tile i - 1 uses set (i - 1) % NSET; i0 % NSET == 0
                    const int i = i0 + u;
                    if (i <= n_mine) {
                        if (i >= 1) {
                            unsigned long long ts = PWS_NOW();
                            rows(px_of(i - 1), Cl[(i - 1) & 1], pa[set], pg[set]);
                            PWS_ACC(e_rows, ts); ts = PWS_NOW();
                            prefetch(i - 1 + NSET, pa[set], pg[set]);
                            PWS_ACC(e_pref, ts);
                        }
                        if (i < n_mine) {
                            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");           // this wave's reads of the hand-off buffer are done
                            unsigned long long ts = PWS_NOW();
#pragma unroll
                            for (int h = 0; h < S; ++h) __builtin_amdgcn_s_barrier();     // #(i S + 1) .. #(i S + S)
                            PWS_ACC(e_bar, ts);
                        }
                    }
                }()), ...);
            }(std::make_integer_sequence<int, NSET>{});
        }
#ifdef I2V_PWS_STAMPS
        if (te == 0 && blockIdx.x < 256) { g_pws_stamps[8 * blockIdx.x + 2] = e_rows; g_pws_stamps[8 * blockIdx.x + 3] = e_pref; g_pws_stamps[8 * blockIdx.x + 4] = e_bar; }
#endif
    }
#endif
}

// conv_pw_stream applies (autotuner bit 8): a plain dense pointwise image launch with K = 64, 128 or 256, whole 64-channel tiles whose
// count divides the 32 blocks of an XCD, and enough pixel tiles to give every stream a few
static int conv_pws_grid(const I2VConvParams& p) {           // blocks (one per CU), 0 = not applicable
    if (!p.pointwise || !p.vec_epilogue || p.temporal || p.quad || p.pre_scale || p.gate_scale || p.blk > 1 || p.blkt > 1) return 0;
    if (p.add1 || p.mask) return 0;                          // (a second addend / an fp32 mask are read inside conv_vec_rows' row loop: not on this kernel's critical path)
    if (p.K != p.Kpad || (p.K != 64 && p.K != 128 && p.K != 256) || p.Cd % 64 != 0 || p.add0_stride > 1 || p.Hs != p.Hg || p.Ws != p.Wg) return 0;
    const int n_cd = p.Cd / 64;
    if (n_cd > 32 || 32 % n_cd != 0) return 0;
    const int64_t n_px = ((int64_t)p.N * p.Hg * p.Wg + 63) / 64;
    // fewer than four tiles per stream: the prologue (the weight panel, the ring's first slabs) would not amortise.  (I2V_PWS_MIN_TILES:
    // developer / test knob -- 0 admits launches that leave streams with one tile or none.)
    static const int min_tiles = [] { const char* e = getenv("I2V_PWS_MIN_TILES"); return e ? atoi(e) : 4; }();
    if (n_px < (int64_t)min_tiles * (256 / n_cd)) return 0;
    return 256;
}
static int launch_conv_pws(const I2VConvParams& p, hipStream_t s) {
    const int grid = conv_pws_grid(p), n_cd = p.Cd / 64, n_streams = grid / n_cd;
    const int n_px = (int)(((int64_t)p.N * p.Hg * p.Wg + 63) / 64);
    if (p.K == 64) hipLaunchKernelGGL((conv_pw_stream<64>), dim3(grid), dim3(I2V_PWS_THREADS), 0, s, p, n_cd, n_streams, n_px);
    else if (p.K == 128) hipLaunchKernelGGL((conv_pw_stream<128>), dim3(grid), dim3(I2V_PWS_THREADS), 0, s, p, n_cd, n_streams, n_px);
    else hipLaunchKernelGGL((conv_pw_stream<256>), dim3(grid), dim3(I2V_PWS_THREADS), 0, s, p, n_cd, n_streams, n_px);
    LAUNCH_CHECK("conv_pw_stream");
    return 0;
}

// Low-K layers with epilogue operands are HBM-bound (their FLOP/byte is below the machine balance): they
// run on 64x64 tiles with the epilogue operands prefetched under the K loop.
static bool conv_wants_prefetch(const I2VConvParams& p) {
    return p.vec_epilogue && !p.gate_scale && !p.pre_scale && p.add0_stride == 1 && (p.add0 || p.add1 || p.mask || p.gate) && p.Kpad <= 256 && (p.pointwise || p.tap_uniform) && p.Cd > 32;
}

// Tail split (conv_igemm_tail) applies to plain 64x64 image launches whose tile count leaves a small remainder over the 256 CUs:
// returns the number of trailing PIXEL tiles to hand to quarter tiles, 0 for none.  Chosen by the autotuner (bit 5 of the
// configuration), never by default.
static int conv_tail_px_tiles(const I2VConvParams& p) {
    if (p.quad || p.pre_scale || p.temporal || !(p.pointwise || p.tap_uniform) || p.Cd % 16 != 0 || p.blk > 1) return 0;
    const int64_t P = (int64_t)p.N * p.Hg * p.Wg;
    const int64_t n_px = (P + 63) / 64; const int n_cd = (p.Cd + 63) / 64;
    const int64_t tiles = n_px * n_cd;
    if (tiles < 2 * 256) return 0;
    const int r = (int)(tiles % 256);
    // a remainder beyond ~0.4 tiles per CU is better left as whole tiles.  (Round 4: ONE round plus a remainder -- a single 32-frame clip
    // leaves the 14x14 layers with 392 tiles, 1.53 per CU -- was tried with the whole remainder as quarter tiles: layer3 3x3 81.9 ->
    // 80.3 TFLOP/s, the K = 1024 reduce 78.1 -> 85.8 where two chunks per barrier reach 90.1: sixteen quarter tiles per pixel tile
    // re-stage the activations four times as often.  Not offered.)
    if (r == 0 || r > 104) return 0;
    return r / n_cd;
}

// MODE 5 applies: the planner marked the packing (K order (16-channel group, tap, channel), 3x3 / stride 1 / pad 1), the launch is a
// plain same-size image launch on a plane width the kernel is instantiated for, and the autotuner chose it (bit 4)
static bool conv_halo_ok(const I2VConvParams& p) {
    return p.halo == 9 && p.tap_uniform && !p.temporal && !p.pre_scale && !p.quad && p.blk <= 1 && p.sh == 1 && p.sw == 1 && p.Hs == p.Hg &&
           p.Ws == p.Wg && (p.Ws == 14 || p.Ws == 28 || p.Ws == 56) && p.Kpad == p.K && (p.Kpad / I2V_KC) % 9 == 0;
}

// CPB = 2 applies (autotuner bit 6): a plain pointwise / tap-uniform image launch with an even chunk count and a full 64-row tile
static bool conv_dc_ok(const I2VConvParams& p) {
    return (p.pointwise || p.tap_uniform) && !p.temporal && !p.pre_scale && !p.quad && p.Cd > 32 && (p.Kpad / I2V_KC) % 2 == 0 && p.Kpad >= 4 * I2V_KC;
}

#ifndef I2V_NO_CONV_DISPATCH      // (tools/pw_stream_probe.cpp compiles the kernels it launches itself, not the whole dispatch)
template <int BD, int BP, int WD, int WP, bool MF16 = false>
static int launch_conv_cfg(const I2VConvParams& p, hipStream_t s) {
    const int64_t P = (int64_t)p.N * p.Hg * p.Wg;
    const int n_cd = (p.Cd + BD - 1) / BD;
    const int64_t n_px = (P + BP - 1) / BP;
    const int64_t grid = n_px * n_cd;
    if (grid <= 0) return 0;
    if (grid > 0x7fffffff) { snprintf(g_be_err, sizeof g_be_err, "conv grid too large"); g_be_has_err = true; return 1; }
    if constexpr (!MF16) {
        if (conv_bf3_ok(p)) {     // split-bf16 K loop (bit 6 of the configuration: two chunks per barrier)
            __atomic_fetch_add(&g_stat_bf3, 1, __ATOMIC_RELAXED);
            const bool two = I2V_BF3_VARIANT == 2 && p.cfg > 0 && ((p.cfg - 1) & 64) && (p.Kpad / I2V_KC) % 2 == 0;      // (the deep-staged variant synchronises per chunk)
            // The loop is bound by VALU issue -- splitting an activation fragment costs 44 vector instructions, and a bf16 MFMA hides about
            // five --, so the 128x128 tile puts its four waves SIDE BY SIDE along the pixels (each 128 rows x 32 pixels): one activation
            // fragment split per 24 MFMAs instead of two, the four weight fragments are plain 16-byte LDS reads.
            if constexpr (BD == 128 && BP == 128 && WD == 2) {
                if (!p.temporal && !getenv("I2V_BF3_SQUARE")) {
                    // bit 6 of the configuration on this tile: the software-pipelined loop (BF3 == 3: the next chunk's activation fragments read and
                    // split under this chunk's MFMAs, three staging buffers).  Measured on the wide tile (tools/bf3_sweep.sh): layer3 3x3 172 -> 187
                    // TFLOP/s, layer2 3x3 174 -> 177, the pointwise shapes 0 ... -8 %; on the smaller tiles the third buffer costs a resident
                    // block and 10-25 % -- so it is one more candidate of the autotuner for this tile only.  Same arithmetic in the same order.
                    if (I2V_BF3_VARIANT == 1 && p.cfg > 0 && ((p.cfg - 1) & 64)) {
                        if (p.pointwise) hipLaunchKernelGGL((conv_igemm_bf3<128, 128, 1, 4, 1, 1, false, 3>), dim3((unsigned)grid), dim3(256), 0, s, p, n_cd);
                        else hipLaunchKernelGGL((conv_igemm_bf3<128, 128, 1, 4, 2, 1, false, 3>), dim3((unsigned)grid), dim3(256), 0, s, p, n_cd);
                    } else if (p.pointwise) hipLaunchKernelGGL((conv_igemm_bf3<128, 128, 1, 4, 1, 1>), dim3((unsigned)grid), dim3(256), 0, s, p, n_cd);
                    else hipLaunchKernelGGL((conv_igemm_bf3<128, 128, 1, 4, 2, 1>), dim3((unsigned)grid), dim3(256), 0, s, p, n_cd);
                    LAUNCH_CHECK("conv_igemm_bf3");
                    return 0;
                }
            }
            if (p.temporal) {
                hipLaunchKernelGGL((conv_igemm_bf3<BD, BP, WD, WP, 2, 1, true>), dim3((unsigned)grid), dim3(256), 0, s, p, n_cd);
                LAUNCH_CHECK("conv_igemm_bf3");
                return 0;
            }
            if (p.pointwise) {
                if constexpr (I2V_BF3_VARIANT == 2) { if (two) { hipLaunchKernelGGL((conv_igemm_bf3<BD, BP, WD, WP, 1, I2V_BF3_VARIANT == 2 ? 2 : 1>), dim3((unsigned)grid), dim3(256), 0, s, p, n_cd); LAUNCH_CHECK("conv_igemm_bf3"); return 0; } }
                hipLaunchKernelGGL((conv_igemm_bf3<BD, BP, WD, WP, 1, 1>), dim3((unsigned)grid), dim3(256), 0, s, p, n_cd);
            } else {
                if constexpr (I2V_BF3_VARIANT == 2) { if (two) { hipLaunchKernelGGL((conv_igemm_bf3<BD, BP, WD, WP, 2, I2V_BF3_VARIANT == 2 ? 2 : 1>), dim3((unsigned)grid), dim3(256), 0, s, p, n_cd); LAUNCH_CHECK("conv_igemm_bf3"); return 0; } }
                hipLaunchKernelGGL((conv_igemm_bf3<BD, BP, WD, WP, 2, 1>), dim3((unsigned)grid), dim3(256), 0, s, p, n_cd);
            }
            LAUNCH_CHECK("conv_igemm_bf3");
            return 0;
        }
    }
    if (p.quad) {           // "quad rows" stems (MODE 4)
        if (p.pre_scale || (p.quad != 1 && p.quad != 2)) { snprintf(g_be_err, sizeof g_be_err, "bad quad-row launch"); g_be_has_err = true; return 1; }
        if (p.temporal) hipLaunchKernelGGL((conv_igemm<BD, BP, WD, WP, 4, false, false, true, MF16>), dim3((unsigned)grid), dim3(256), 0, s, p, n_cd);
        else hipLaunchKernelGGL((conv_igemm<BD, BP, WD, WP, 4, false, false, false, MF16>), dim3((unsigned)grid), dim3(256), 0, s, p, n_cd);
        LAUNCH_CHECK("conv_igemm");
        return 0;
    }
    if constexpr (BD == 64 && BP == 64 && !MF16) {
        if (p.cfg > 0 && ((p.cfg - 1) & 16) && conv_halo_ok(p)) {
            if (p.Ws == 14) hipLaunchKernelGGL((conv_igemm_halo<14>), dim3((unsigned)grid), dim3(256), 0, s, p, n_cd);
            else if (p.Ws == 28) hipLaunchKernelGGL((conv_igemm_halo<28>), dim3((unsigned)grid), dim3(256), 0, s, p, n_cd);
            else hipLaunchKernelGGL((conv_igemm_halo<56>), dim3((unsigned)grid), dim3(256), 0, s, p, n_cd);
            LAUNCH_CHECK("conv_igemm_halo");
            return 0;
        }
        if (p.cfg > 0 && ((p.cfg - 1) & 64) && conv_dc_ok(p)) {
            if (p.pointwise) hipLaunchKernelGGL((conv_igemm_dc<1, 2>), dim3((unsigned)grid), dim3(256), 0, s, p, n_cd);
            else hipLaunchKernelGGL((conv_igemm_dc<2, 2>), dim3((unsigned)grid), dim3(256), 0, s, p, n_cd);
            LAUNCH_CHECK("conv_igemm_dc");
            return 0;
        }
        if (p.cfg > 0 && ((p.cfg - 1) & 32)) {
            const int tail = conv_tail_px_tiles(p);
            if (tail > 0 && !conv_wants_prefetch(p)) {
                const int nA = (int)((n_px - tail) * n_cd), n_cd_b = (p.Cd + 15) / 16;
                const int64_t nB = (int64_t)tail * n_cd_b;
                const dim3 g((unsigned)(nA + nB));
                if (p.pointwise) hipLaunchKernelGGL((conv_igemm_tail<1>), g, dim3(256), 0, s, p, n_cd, nA, n_cd_b, (n_px - tail) * 64);
                else hipLaunchKernelGGL((conv_igemm_tail<2>), g, dim3(256), 0, s, p, n_cd, nA, n_cd_b, (n_px - tail) * 64);
                LAUNCH_CHECK("conv_igemm_tail");
                return 0;
            }
        }
    }
    if constexpr (MF16) {
        if (p.pre_scale) { snprintf(g_be_err, sizeof g_be_err, "pre-activation convolutions have no 16-row variant"); g_be_has_err = true; return 1; }
        if (p.pointwise) hipLaunchKernelGGL((conv_igemm<BD, BP, WD, WP, 1, false, false, false, true>), dim3((unsigned)grid), dim3(256), 0, s, p, n_cd);
        else if (p.tap_uniform && p.temporal) hipLaunchKernelGGL((conv_igemm<BD, BP, WD, WP, 2, false, false, true, true>), dim3((unsigned)grid), dim3(256), 0, s, p, n_cd);
        else if (p.tap_uniform) hipLaunchKernelGGL((conv_igemm<BD, BP, WD, WP, 2, false, false, false, true>), dim3((unsigned)grid), dim3(256), 0, s, p, n_cd);
        else if (p.temporal) hipLaunchKernelGGL((conv_igemm<BD, BP, WD, WP, 0, false, false, true, true>), dim3((unsigned)grid), dim3(256), 0, s, p, n_cd);
        else hipLaunchKernelGGL((conv_igemm<BD, BP, WD, WP, 0, false, false, false, true>), dim3((unsigned)grid), dim3(256), 0, s, p, n_cd);
    } else if (p.pre_scale) {
        if (p.temporal) { snprintf(g_be_err, sizeof g_be_err, "pre-activation convolutions have no temporal variant"); g_be_has_err = true; return 1; }
        if constexpr ((BD == 64 && BP == 64) || (BD == 128 && BP == 128)) {
            if (p.pointwise) hipLaunchKernelGGL((conv_igemm<BD, BP, WD, WP, 1, false, true>), dim3((unsigned)grid), dim3(256), 0, s, p, n_cd);
            else if (p.tap_uniform) hipLaunchKernelGGL((conv_igemm<BD, BP, WD, WP, 2, false, true>), dim3((unsigned)grid), dim3(256), 0, s, p, n_cd);
            else hipLaunchKernelGGL((conv_igemm<BD, BP, WD, WP, 0, false, true>), dim3((unsigned)grid), dim3(256), 0, s, p, n_cd);
        }
    } else if (BD == 64 && BP == 64 && conv_wants_prefetch(p) && !(p.cfg > 0 && ((p.cfg - 1) & 8))) {
        if constexpr (BD == 64 && BP == 64) {
            if (p.pointwise) hipLaunchKernelGGL((conv_igemm<BD, BP, WD, WP, 1, true>), dim3((unsigned)grid), dim3(256), 0, s, p, n_cd);
            else if (p.temporal) hipLaunchKernelGGL((conv_igemm<BD, BP, WD, WP, 2, true, false, true>), dim3((unsigned)grid), dim3(256), 0, s, p, n_cd);
            else hipLaunchKernelGGL((conv_igemm<BD, BP, WD, WP, 2, true>), dim3((unsigned)grid), dim3(256), 0, s, p, n_cd);
        }
    } else if (p.pointwise) hipLaunchKernelGGL((conv_igemm<BD, BP, WD, WP, 1, false>), dim3((unsigned)grid), dim3(256), 0, s, p, n_cd);
    else if (p.tap_uniform && p.temporal) hipLaunchKernelGGL((conv_igemm<BD, BP, WD, WP, 2, false, false, true>), dim3((unsigned)grid), dim3(256), 0, s, p, n_cd);
    else if (p.tap_uniform) hipLaunchKernelGGL((conv_igemm<BD, BP, WD, WP, 2, false>), dim3((unsigned)grid), dim3(256), 0, s, p, n_cd);
    else if (p.temporal) hipLaunchKernelGGL((conv_igemm<BD, BP, WD, WP, 0, false, false, true>), dim3((unsigned)grid), dim3(256), 0, s, p, n_cd);
    else hipLaunchKernelGGL((conv_igemm<BD, BP, WD, WP, 0, false>), dim3((unsigned)grid), dim3(256), 0, s, p, n_cd);
    LAUNCH_CHECK("conv_igemm");
    return 0;
}

#endif

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
    return n;
}

static void conv_magics(I2VConvParams& p) {
    fastdiv_magic((unsigned)(p.Hg * p.Wg), &p.dv_hw_m, &p.dv_hw_s);
    fastdiv_magic((unsigned)p.Wg, &p.dv_w_m, &p.dv_w_s);
    fastdiv_magic((unsigned)(p.Tg > 0 ? p.Tg : 1), &p.dv_t_m, &p.dv_t_s);
    fastdiv_magic((unsigned)(p.Wo > 0 ? p.Wo : 1), &p.dv_wo_m, &p.dv_wo_s);
}

// Fused pair (conv_fused_kernel): the structural rule is i2v_conv_pair_fusable (i2v_kernels.h, shared with the host simulation).
// Returns 0 (no), 1 (plain staging only) or 3 (halo staging available too).  Whether a's output has OTHER readers is the planner's
// business (i2v_engine.cpp: mark_fusable).
int k_conv_fusable(const I2VConvParams& a, const I2VConvParams& b) {
    if (!i2v_conv_pair_fusable(a, b)) return 0;
    return (a.Cd == 64 && conv_halo_ok(a)) ? 3 : 1;
}

#ifndef I2V_NO_CONV_DISPATCH
int k_conv_fused(const I2VConvParams& a_in, const I2VConvParams& b_in, int halo, i2v_stream_t s) {
    I2VConvParams a = a_in, b = b_in;
    const int64_t P = (int64_t)a.N * a.Hg * a.Wg;
    if (P + 1024 >= (1ll << 31) || b.N != a.N) { snprintf(g_be_err, sizeof g_be_err, "fused conv launch: bad grid"); g_be_has_err = true; return 1; }
    const int ok = k_conv_fusable(a, b);
    if (!ok || !a.vec_epilogue || !b.vec_epilogue || (halo && !(ok & 2))) { snprintf(g_be_err, sizeof g_be_err, "fused conv launch: pair not eligible"); g_be_has_err = true; return 1; }
    conv_magics(a); conv_magics(b);
    a.cfg = b.cfg = 0;                    // (the variant bits of the separate launches -- streaming stores among them -- do not apply)
    const dim3 grid((unsigned)((P + 63) / 64));
    hipStream_t st = (hipStream_t)s;
    if (a.Cd == 128) hipLaunchKernelGGL((conv_fused_kernel<128, 0>), grid, dim3(256), 0, st, a, b);
    else if (!halo) hipLaunchKernelGGL((conv_fused_kernel<64, 0>), grid, dim3(256), 0, st, a, b);
    else if (a.Ws == 14) hipLaunchKernelGGL((conv_fused_kernel<64, 14>), grid, dim3(256), 0, st, a, b);
    else if (a.Ws == 28) hipLaunchKernelGGL((conv_fused_kernel<64, 28>), grid, dim3(256), 0, st, a, b);
    else hipLaunchKernelGGL((conv_fused_kernel<64, 56>), grid, dim3(256), 0, st, a, b);
    LAUNCH_CHECK("conv_fused");
    return 0;
}

int k_conv(const I2VConvParams& p_in, i2v_stream_t s) {
    hipStream_t st = (hipStream_t)s;
    I2VConvParams p = p_in;
    if ((int64_t)p.N * p.Hg * p.Wg + 1024 >= (1ll << 31)) { snprintf(g_be_err, sizeof g_be_err, "conv launch of more than 2^31 grid pixels"); g_be_has_err = true; return 1; }
    conv_magics(p);
    if (p.cfg <= 0) p.cfg = conv_pick(p) + 1;          // the model's pick -- or $I2V_FORCE_CFG, which may carry the variant bits too
    __atomic_fetch_add(&g_stat_conv, 1, __ATOMIC_RELAXED);
    if (((p.cfg - 1) & 1024) && conv_stem64_ok(p) && !((uintptr_t)p.src & 15)) {       // wide 7x7 / 2 forward stem on a 2-D halo tile (autotuner bit 10)
        __atomic_fetch_add(&g_stat_sth, 1, __ATOMIC_RELAXED);
        return launch_conv_stem64(p, st);
    }
    if (((p.cfg - 1) & 1024) && conv_stemhalo_ok(p) && !((uintptr_t)p.src & 15)) {     // narrow forward stem on a 2-D halo tile (autotuner bit 10)
        __atomic_fetch_add(&g_stat_sth, 1, __ATOMIC_RELAXED);
        return launch_conv_stemhalo(p, st);
    }
    if (((p.cfg - 1) & 512) && conv_ighalo_ok(p)) {     // image gradient on a 2-D halo tile (autotuner bit 9)
        __atomic_fetch_add(&g_stat_igh, 1, __ATOMIC_RELAXED);
        return launch_conv_ighalo(p, st);
    }
    if (((p.cfg - 1) & 256) && conv_pws_grid(p)) {     // persistent role-split pointwise kernel (autotuner bit 8)
        __atomic_fetch_add(&g_stat_pws, 1, __ATOMIC_RELAXED);
        return launch_conv_pws(p, st);
    }
    switch ((p.cfg - 1) & 7) {
        case 0: return launch_conv_cfg<128, 128, 2, 2>(p, st);
        case 1: return launch_conv_cfg<64, 128, 2, 2>(p, st);
        case 2: return launch_conv_cfg<128, 64, 2, 2>(p, st);
        case 3: return launch_conv_cfg<64, 64, 2, 2>(p, st);
        case 5: return launch_conv_cfg<16, 256, 1, 4, true>(p, st);
        default: return launch_conv_cfg<32, 256, 1, 4>(p, st);
    }
}

#endif

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

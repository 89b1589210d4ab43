// Internal header of the gfx950 kernel backend (csrc/i2v_*.hip): what its translation units share -- the error slot, launch
// counters, exact integer division, the host-side predicates that decide which convolution kernel a launch may use, and the launch
// entry points each translation unit exports to the dispatch (i2v_kernels.hip).  One family of kernels per translation unit so that
// the library builds in parallel (round 6: one 255 KB file took five minutes).
#pragma once
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

extern thread_local char g_be_err[256];
extern thread_local bool g_be_has_err;
int hip_fail(hipError_t e, const char* what);
#define HIPCHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) return hip_fail(e_, #x); } while (0)
#define LAUNCH_CHECK(name) do { hipError_t e_ = hipGetLastError(); if (e_ != hipSuccess) return hip_fail(e_, name); } while (0)
extern long long g_stat_conv, g_stat_pws, g_stat_bf3, g_stat_igh, g_stat_sth, g_stat_fastblock, g_stat_vfma, g_stat_igv;      // (relaxed counters: diagnostics only)

// n / d for 0 <= n < 2^31 with the precomputed (m, s) of fastdiv_magic: exact
__device__ __forceinline__ unsigned fastdiv(unsigned n, unsigned m, unsigned s) {
    return (unsigned)(((unsigned long long)n * m) >> s);
}
static inline void fastdiv_magic(unsigned d, uint32_t* m, uint32_t* s) {
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

static inline void conv_magics(I2VConvParams& p) {
    fastdiv_magic((unsigned)(p.Hg * p.Wg), &p.dv_hw_m, &p.dv_hw_s);
    fastdiv_magic((unsigned)p.Wg, &p.dv_w_m, &p.dv_w_s);
    fastdiv_magic((unsigned)(p.Tg > 0 ? p.Tg : 1), &p.dv_t_m, &p.dv_t_s);
    fastdiv_magic((unsigned)(p.Wo > 0 ? p.Wo : 1), &p.dv_wo_m, &p.dv_wo_s);
}

// ---- which kernel a convolution launch may use (host side; shared by the launchers and by the autotuner's candidate list) ----
// Low-K layers with epilogue operands are HBM-bound (their FLOP/byte is below the machine balance): they
// run on 64x64 tiles with the epilogue operands prefetched under the K loop.
static inline bool conv_wants_prefetch(const I2VConvParams& p) {
    return p.vec_epilogue && !p.gate_scale && !p.pre_scale && p.add0_stride == 1 && (p.add0 || p.add1 || p.mask || p.gate) && p.Kpad <= 256 && (p.pointwise || p.tap_uniform) && p.Cd > 32;
}

// Tail split (conv_igemm_tail) applies to plain 64x64 image launches whose tile count leaves a small remainder over the 256 CUs:
// returns the number of trailing PIXEL tiles to hand to quarter tiles, 0 for none.  Chosen by the autotuner (bit 5 of the
// configuration), never by default.
static inline int conv_tail_px_tiles(const I2VConvParams& p) {
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
static inline bool conv_halo_ok(const I2VConvParams& p) {
    return p.halo == 9 && p.tap_uniform && !p.temporal && !p.pre_scale && !p.quad && p.blk <= 1 && p.sh == 1 && p.sw == 1 && p.Hs == p.Hg &&
           p.Ws == p.Wg && (p.Ws == 14 || p.Ws == 28 || p.Ws == 56) && p.Kpad == p.K && (p.Kpad / I2V_KC) % 9 == 0;
}

// CPB = 2 applies (autotuner bit 6): a plain pointwise / tap-uniform image launch with an even chunk count and a full 64-row tile
static inline bool conv_dc_ok(const I2VConvParams& p) {
    return (p.pointwise || p.tap_uniform) && !p.temporal && !p.pre_scale && !p.quad && p.Cd > 32 && (p.Kpad / I2V_KC) % 2 == 0 && p.Kpad >= 4 * I2V_KC;
}

// Split-bf16 K loop (I2V_MATH=bf16x3): an EXPERIMENTAL build only (-DI2V_EXPERIMENTAL, i2v_conv_exp.hip); the default library has no
// such kernels and the planner refuses the mode (i2v_engine.cpp: math_bf16x3).
static inline bool conv_bf3_ok(const I2VConvParams& p) {      // (temporal launches -- video networks' k x 1 x 1 and strided convolutions -- only as tap-uniform ones: the staging of MODE 2, VID)
#ifdef I2V_EXPERIMENTAL
    return p.bf3 && p.wp3 && (p.pointwise || p.tap_uniform) && (!p.temporal || p.tap_uniform) && !p.pre_scale && !p.quad && p.blk <= 1 && p.blkt <= 1 && p.Cd > 32;
#else
    (void)p; return false;
#endif
}

// ---- launch entry points, one translation unit each ----
int launch_conv_cfg0(const I2VConvParams& p, hipStream_t s);      // 128 x 128 tile            i2v_conv_cfg0.hip
int launch_conv_cfg1(const I2VConvParams& p, hipStream_t s);      //  64 x 128                 i2v_conv_cfg1.hip
int launch_conv_cfg2(const I2VConvParams& p, hipStream_t s);      // 128 x  64                 i2v_conv_cfg2.hip
int launch_conv_cfg3(const I2VConvParams& p, hipStream_t s);      //  64 x  64 and its variants (halo, two chunks per barrier, tail split, prefetch)   i2v_conv_cfg3.hip
int launch_conv_cfg4(const I2VConvParams& p, hipStream_t s);      //  32 x 256                 i2v_conv_cfg4.hip
int launch_conv_cfg5(const I2VConvParams& p, hipStream_t s);      //  16 x 256 on 16x16x4 fragments   i2v_conv_cfg5.hip
bool conv_ighalo_ok(const I2VConvParams& p);                      // the 3-channel stems on 2-D halo tiles   i2v_conv_stems.hip
int launch_conv_ighalo(const I2VConvParams& p, hipStream_t s);
bool conv_igvfma_ok(const I2VConvParams& p);                      // the quad-row image gradient of a narrow stem on packed-fp32 vector FMAs
int launch_conv_igvfma(const I2VConvParams& p, hipStream_t s);
bool conv_stemhalo_ok(const I2VConvParams& p);
int launch_conv_stemhalo(const I2VConvParams& p, hipStream_t s);
bool conv_vfma_ok(const I2VConvParams& p);                        // one narrow launch on packed-fp32 vector FMAs   i2v_fastblock.hip
int launch_conv_vfma(const I2VConvParams& p, hipStream_t s);
#ifdef I2V_EXPERIMENTAL
#ifndef I2V_BF3_VARIANT
#define I2V_BF3_VARIANT 1        // 1: weight fragments staged through LDS by DMA, I2V_BF3_STAGES buffers; 2: loaded straight into registers, one chunk ahead
#endif
#endif
#ifdef I2V_EXPERIMENTAL                                           // built, measured, not part of the product: i2v_conv_exp.hip, i2v_conv_stems.hip
bool conv_stem64_ok(const I2VConvParams& p);
int launch_conv_stem64(const I2VConvParams& p, hipStream_t s);
int conv_pws_grid(const I2VConvParams& p);
int launch_conv_pws(const I2VConvParams& p, hipStream_t s);
int launch_conv_bf3(const I2VConvParams& p, hipStream_t s);       // tile = (p.cfg - 1) & 7 in 0..3
#endif

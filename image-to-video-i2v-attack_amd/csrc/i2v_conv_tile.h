// conv_tile: the implicit-GEMM convolution tile on fp32 MFMA (forward and input gradient), its dense epilogue, and the kernels that
// are nothing but a tile with a launch shape (conv_igemm, conv_igemm_halo, conv_igemm_dc, conv_igemm_tail).  Included by every
// translation unit that instantiates a tile shape (i2v_conv_cfg*.hip, i2v_conv_exp.hip); templates only -- nothing here has linkage.
#pragma once
#include "i2v_be.h"

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

// =============================================================================================
// implicit-GEMM convolution on fp32 MFMA
// =============================================================================================
// Block: 256 threads = 4 waves arranged WD x WP; block tile BD (output channels) x BP (pixels),
// K consumed in chunks of I2V_KC=16 through double-buffered LDS (register-staged prefetch).
// MFMA operand roles: A = weights (row i = channel), B = activations (column j = pixel):
//   A: lane l holds Wp[k = kk + (l>>5)][cd = l&31]      B: lane l holds X[k = kk + (l>>5)][px = l&31]
//   D: lane l, register r  ->  pixel l&31, channel (r&3) + 8*(r>>2) + 4*(l>>5)
// so every global store instruction writes 32 consecutive pixels of one channel plane.
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

// Fused bottleneck of SlowFast's fast pathway (round 6): I2VFastBlockParams / k_fastblock.
//
// The fast pathway's res-stage blocks have 8 mid channels -- half of a 16-row MFMA fragment -- on 128 frames of 56 x 56: as separate
// conv_igemm launches (conv1 3x1x1 32 -> 8, conv2 1x3x3 8 -> 8, conv3 1x1x1 8 -> 32 + residual + ReLU; backward: the input gradients
// in reverse) they take 17-45 us each at 1-17 TFLOP/s and 0.7-3.8 TB/s: on neither roof, per-launch latency and half-empty fragments
// (profiles/r5_ilaf_breakdown_slowfast.txt; a quarter of an ILAF step).  Here a block of 128 threads takes R whole rows of ONE frame and
// runs the block's convolutions back to back:
//   stage A  (conv1, or conv3's input gradient) on the R + 2 rows the 3 x 3 stage needs, one position per lane, operands straight
//            from global memory (consecutive lanes = consecutive pixels: 256-byte requests), results into LDS as [channel][row][col + 2]
//            with zero columns / rows where the 3 x 3 stage pads;
//   stage B  (conv2 or its input gradient) on the R rows, operands from LDS at k-table offsets;
//   stage C  (forward: conv3 + the residual -- a tensor, or the projection shortcut computed here -- + ReLU) on the lane's own pixel.
// The arithmetic is packed fp32 FMA on the vector unit: a lane holds the channel PAIRS of its pixel (v_pk_fma_f32: weight pair from
// SGPRs, the operand broadcast to both halves), 2 FMAs per lane and issue slot -- the fp32 MFMA's own rate without its empty rows.
// An fp32 MFMA is a k-ordered fmaf chain per output element; this kernel walks the SAME packed K order (the launches' own weights and
// k-tables) with fmaf, applies the same epilogue operations in the same order, and skips only what adds an exact zero (zero-weight
// padding rows, taps outside the clip or the plane): bit-identical to the launches it replaces (tests/test_gpu_video.py), which is
// what lets the plan-time autotuner choose between the two.
#include "i2v_be.h"

typedef float f2 __attribute__((ext_vector_type(2)));
typedef const __attribute__((address_space(4))) f2* wptr_t;       // packed weights through the constant address space: scalar loads, SGPR operands

long long g_stat_fastblock = 0, g_stat_vfma = 0, g_stat_igv = 0;

__device__ __forceinline__ unsigned fb_ballot_lo(unsigned long long b) { return (unsigned)b; }

typedef const __attribute__((address_space(4))) float* cfptr_t;   // per-channel vectors (shifts) through the constant address space: scalar loads

// 1-bit gates of NC channels for this wave's 64 consecutive pixels (first one at bit index `bit0`, a multiple of 32; the strip ends at
// `bit_end`, a multiple of 32): lane c keeps channel c's two words (`fb_gate_collect`, one ballot per channel), then lanes 0 .. NC - 1
// store them (`fb_gate_store`: two store instructions for all channels)
__device__ __forceinline__ void fb_gate_collect(unsigned& lo, unsigned& hi, const int c, const bool on, const int lane) {
    const unsigned long long m = __ballot(on);
    if (lane == c) { lo = (unsigned)m; hi = (unsigned)(m >> 32); }
}
__device__ __forceinline__ void fb_gate_store(uint32_t* rows, const int stride, const int NC, const unsigned lo, const unsigned hi, const int64_t bit0, const int64_t bit_end,
                                              const int lane) {
    if (lane < NC) {
        uint32_t* const w = rows + (int64_t)lane * stride + (bit0 >> 5);
        if (bit0 < bit_end) w[0] = lo;
        if (bit0 + 32 < bit_end) w[1] = hi;
    }
}

// LDS image of a block: the decoded k-tables of stages A and B (int per K row: element offset, or FB_SKIP for a row that adds an exact
// zero), then stage A's result [CM][R + 2][W + 2].
#define FB_SKIP (-0x40000000)
// CM: mid channels (8: SlowFast's fast pathway; 4: the tiny test graphs).  FWD: forward block (stages A, B, C: shift + ReLU + own gates each)
// or backward (stages A, B: the gates of the tensors whose gradients they are).  PROJ: the forward block's shortcut is a projection.
template <int CM, bool FWD, bool PROJ>
__global__ void __launch_bounds__(128) __attribute__((amdgpu_waves_per_eu(4, 4)))      // (7 blocks of 2 waves per CU at 128 frames of 56 x 56: all resident at once)
fast_block_kernel(const I2VFastBlockParams p) {
    constexpr int CP = CM / 2, C3 = 4 * CM, C3P = C3 / 2, PA = 3;      // PA: stage-A positions a lane works on at once (their loads in flight together)
    extern __shared__ __attribute__((aligned(16))) float lds_f[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int H = p.H, W = p.W, R = p.R, HW = H * W, AR = R + 2, AW = W + 2;
    int* const tabA = reinterpret_cast<int*>(lds_f);                    // [a.Kpad]
    int* const tabB = tabA + p.a.Kpad;                                   // [b.Kpad]
    float* const A1 = lds_f + ((p.a.Kpad + p.b.Kpad + 3) & ~3);          // [CM][AR][AW]
    // Block -> (clip, frame, strip).  Workgroups go to the 8 XCDs round robin, each XCD with its own L2: a UNIT -- one clip's frames over
    // a group of strips -- is given to ONE XCD and walked frame by frame there, so that a frame read as tap t - 1, t, t + 1 (and a halo
    // row read by two strips) comes from HBM once and from that L2 afterwards.  All of it in scalar registers (uniform base addresses).
    const unsigned b = blockIdx.x, xcd = b & 7u, slot = b >> 3;
    const unsigned uix = fastdiv(slot, p.dv_bu_m, p.dv_bu_s), i = slot - uix * p.BU, u = xcd + 8u * uix;      // unit, block within the unit
    if (u >= (unsigned)p.U) return;
    const int clip = (int)fastdiv(u, p.dv_s_m, p.dv_s_s), sg = (int)u - clip * p.S;
    const int t = (int)fastdiv(i, p.dv_g_m, p.dv_g_s), strip = sg * p.G + ((int)i - t * p.G);
    const int n = clip * p.T + t, r0 = strip * R;
    // ---- decode the k-tables once per block: what a K row reads, as an offset; rows that add an exact zero (padding rows, taps outside
    // the clip) are marked and skipped ----
    for (int k = tid; k < p.a.Kpad; k += 128) {
        const I2VKEntry e = p.a.ktab[k];
        const int tf = t + (e.valid >> 1);
        tabA[k] = ((e.valid & 1) && tf >= 0 && tf < p.T) ? (int)((int64_t)tf * p.src_nstride) + e.chan_off : FB_SKIP;      // (elements from the clip's first frame: < 2^30, k_fastblock checks)
    }
    for (int k = tid; k < p.b.Kpad; k += 128) {
        const I2VKEntry e = p.b.ktab[k];
        tabB[k] = (e.valid & 1) ? ((int)fastdiv((unsigned)e.chan_off, p.dv_hw_m, p.dv_hw_s) * AR + e.dh) * AW + e.dw : FB_SKIP;
    }
    // zero borders of stage A's image: columns 0 and W + 1 of every (channel, row)
    for (int i = tid; i < CM * AR * 2; i += 128) A1[(i >> 1) * AW + ((i & 1) ? AW - 1 : 0)] = 0.f;
    __syncthreads();
    // ---------------- stage A: R + 2 rows, PA positions per lane at once ----------------
    {
        const wptr_t w = (wptr_t)p.a.wp;
        const int wrow = p.a.Cdpad / 2;
        const float* const fbase = p.src + (int64_t)(clip * p.T) * p.src_nstride;          // the clip's first frame
        for (int q0 = tid; q0 < AR * W; q0 += 128 * PA) {
            // position q of the (R + 2) x W grid is plane pixel (r0 - 1) W + q: rows are contiguous (only the LDS image, two columns wider, needs the row)
            int pix[PA]; unsigned pxu[PA];
#pragma unroll
            for (int u = 0; u < PA; ++u) {
                const int px = (r0 - 1) * W + q0 + 128 * u;
                pix[u] = (q0 + 128 * u < AR * W && px >= 0 && px < HW) ? px : -1;           // -1: outside the plane (zero padding of the 3 x 3 stage) or no such position
                pxu[u] = pix[u] < 0 ? 0u : (unsigned)pix[u];
            }
            unsigned gw[PA][FWD ? 1 : CM];            // backward: the gate words of this stage's tensor at the lane's positions, requested with the operands
            if (!FWD) {
#pragma unroll
                for (int u = 0; u < PA; ++u)
#pragma unroll
                    for (int c = 0; c < CM; ++c) gw[u][c] = p.a.gate[(int64_t)c * p.a.gate_stride + (((int64_t)n * HW + pxu[u]) >> 5)];
            }
            f2 acc[PA][CP];
#pragma unroll
            for (int u = 0; u < PA; ++u)
#pragma unroll
                for (int c = 0; c < CP; ++c) acc[u][c] = f2{0.f, 0.f};
            float xa[PA][8], xb[PA][8];
            // software pipeline over chunks of 8 K rows: the next chunk's operands are requested before this chunk's FMAs
#define FB_LOAD(K0, X)                                                                                     \
            {                                                                                              \
                const int4 o0 = *reinterpret_cast<const int4*>(tabA + (K0)), o1 = *reinterpret_cast<const int4*>(tabA + (K0) + 4);      \
                const int off[8] = {o0.x, o0.y, o0.z, o0.w, o1.x, o1.y, o1.z, o1.w};                         \
                _Pragma("unroll") for (int j = 0; j < 8; ++j) {                                            \
                    const int ou = __builtin_amdgcn_readfirstlane(off[j]);       /* (uniform: a scalar base, the lane's pixel as the 32-bit offset) */ \
                    const float* const bk = fbase + (ou == FB_SKIP ? 0 : ou);                              \
                    _Pragma("unroll") for (int u = 0; u < PA; ++u) X[u][j] = bk[pxu[u]];                    \
                    if (ou == FB_SKIP) { _Pragma("unroll") for (int u = 0; u < PA; ++u) X[u][j] = 0.f; }     \
                }                                                                                          \
            }
#define FB_FMA(K0, X)                                                                                      \
            _Pragma("unroll") for (int j = 0; j < 8; ++j) {                                                \
                f2 wv[CP];                                                                                 \
                _Pragma("unroll") for (int c = 0; c < CP; ++c) wv[c] = w[(int64_t)((K0) + j) * wrow + c];   \
                _Pragma("unroll") for (int u = 0; u < PA; ++u) {                                           \
                    const f2 xx = f2{X[u][j], X[u][j]};                                                    \
                    _Pragma("unroll") for (int c = 0; c < CP; ++c) acc[u][c] = __builtin_elementwise_fma(wv[c], xx, acc[u][c]);      \
                }                                                                                          \
            }
            FB_LOAD(0, xa)
            for (int k0 = 0; k0 < p.a.Kpad; k0 += 16) {           // (Kpad is a multiple of 16)
                FB_LOAD(k0 + 8, xb)
                FB_FMA(k0, xa)
                if (k0 + 16 < p.a.Kpad) FB_LOAD(k0 + 16, xa)
                FB_FMA(k0 + 8, xb)
            }
#undef FB_LOAD
#undef FB_FMA
            float sh[CM];
#pragma unroll
            for (int c = 0; c < CM; ++c) sh[c] = FWD ? ((cfptr_t)p.a.shift)[c] : 0.f;
#pragma unroll
            for (int u = 0; u < PA; ++u) {
                const int q = q0 + 128 * u;
                if (q >= AR * W) continue;
                const int row = (int)fastdiv((unsigned)q, p.dv_w_m, p.dv_w_s), col = q - row * W;
                const bool inside = pix[u] >= 0;
                const unsigned bitn = (unsigned)(((int64_t)n * HW + pxu[u]) & 31);
#pragma unroll
                for (int c = 0; c < CM; ++c) {
                    float v = (c & 1) ? acc[u][c >> 1].y : acc[u][c >> 1].x;
                    if (FWD) v = fmaxf(v + sh[c], 0.f);
                    else if (!((gw[u][c] >> bitn) & 1u)) v = 0.f;
                    A1[(c * AR + row) * AW + col + 1] = inside ? v : 0.f;                   // rows outside the plane: the 3 x 3 stage's zero padding
                }
            }
        }
    }
    __syncthreads();
    // ---------------- stages B (and C): the R rows, one pixel per lane ----------------
    const int RW = R * W;
    const int64_t bit_strip = (int64_t)n * HW + (int64_t)r0 * W, bit_end = bit_strip + RW;
    for (int pp0 = tid - lane; pp0 < RW; pp0 += 128) {           // (every lane of a wave runs the round: the gate words come from ballots)
        const int pp = pp0 + lane;
        const bool act = pp < RW;
        const int ppc = act ? pp : 0;
        const int row = (int)fastdiv((unsigned)ppc, p.dv_w_m, p.dv_w_s), col = ppc - row * W;
        const unsigned upix = (unsigned)(r0 * W + ppc);          // (uniform base + this 32-bit offset: one address register per access)
        const int64_t bit0 = bit_strip + pp0;
        const float* const ctr = A1 + (row + 1) * AW + col + 1;     // the lane's own position in channel 0
        // operands of the later stages, requested now: the residual (an identity shortcut) or the projection's inputs; backward: gate words
        float xs_[PROJ ? CM : 1];
        unsigned gw[FWD ? 1 : CM];
        if (PROJ) {
            const float* const xs = p.src + (int64_t)n * p.src_nstride;
#pragma unroll
            for (int k = 0; k < CM; ++k) xs_[k] = (xs + (int64_t)k * HW)[upix];
        }
        if (!FWD) {
#pragma unroll
            for (int c = 0; c < CM; ++c) gw[c] = p.b.gate[(int64_t)c * p.b.gate_stride + (((int64_t)n * HW + upix) >> 5)];
        }
        if (FWD) {                   // stage A's own gates, from the finished values in LDS (post-ReLU: > 0 is the bit)
            unsigned glo = 0, ghi = 0;
#pragma unroll
            for (int c = 0; c < CM; ++c) fb_gate_collect(glo, ghi, c, act && ctr[c * AR * AW] > 0.f, lane);
            fb_gate_store(p.a.gate_out, p.a.gate_out_stride, CM, glo, ghi, bit0, bit_end, lane);
        }
        f2 acc[CP];
#pragma unroll
        for (int c = 0; c < CP; ++c) acc[c] = f2{0.f, 0.f};
        {
            const wptr_t w = (wptr_t)p.b.wp;
            const int wrow = p.b.Cdpad / 2;
            for (int k0 = 0; k0 < p.b.Kpad; k0 += 8) {
                const int4 o0 = *reinterpret_cast<const int4*>(tabB + k0), o1 = *reinterpret_cast<const int4*>(tabB + k0 + 4);
                const int off[8] = {o0.x, o0.y, o0.z, o0.w, o1.x, o1.y, o1.z, o1.w};
                float x[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) { const float v = ctr[off[j] == FB_SKIP ? 0 : off[j]]; x[j] = off[j] == FB_SKIP ? 0.f : v; }
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const f2 xx = f2{x[j], x[j]};
#pragma unroll
                    for (int c = 0; c < CP; ++c) acc[c] = __builtin_elementwise_fma(w[(int64_t)(k0 + j) * wrow + c], xx, acc[c]);
                }
            }
        }
        float a2[CM];
        {
            const unsigned bitn = (unsigned)(((int64_t)n * HW + upix) & 31);
#pragma unroll
            for (int c = 0; c < CM; ++c) {
                float v = (c & 1) ? acc[c >> 1].y : acc[c >> 1].x;
                if (FWD) v = fmaxf(v + ((cfptr_t)p.b.shift)[c], 0.f);
                else if (!((gw[c] >> bitn) & 1u)) v = 0.f;
                a2[c] = v;
            }
        }
        if (!FWD) {                  // backward: stage B's result is the output
            if (act) {
                float* const o = p.dst + (int64_t)n * p.dst_nstride;
#pragma unroll
                for (int c = 0; c < CM; ++c) (o + (int64_t)c * HW)[upix] = a2[c];
            }
            continue;
        }
        {
            unsigned glo = 0, ghi = 0;
#pragma unroll
            for (int c = 0; c < CM; ++c) fb_gate_collect(glo, ghi, c, act && a2[c] > 0.f, lane);
            fb_gate_store(p.b.gate_out, p.b.gate_out_stride, CM, glo, ghi, bit0, bit_end, lane);
        }
        // ---------------- stage C: conv3 (pointwise over the lane's own a2: K row k is channel k) + residual + ReLU ----------------
        // One output-channel PAIR per iteration of a ROLLED loop: its CM weight pairs (and the projection's) are scalar loads that live for
        // one iteration -- unrolled over the 4 CM channels the allocator kept 8 CM^2 weights in SGPRs and spilled whole 16-register tuples to
        // VGPR lanes around every FMA (3300 v_readlane / v_writelane in the first version).  Each output element still sums its K rows in order.
        float* const o = p.dst + (int64_t)n * p.dst_nstride;
        const float* const rs = PROJ ? nullptr : p.add0 + (int64_t)n * p.add0_nstride;
        unsigned glo = 0, ghi = 0;
        f2 rq[4] = {f2{0.f, 0.f}, f2{0.f, 0.f}, f2{0.f, 0.f}, f2{0.f, 0.f}};      // the residual of the next four pairs, in flight (requested four iterations ahead)
        if (!PROJ) {
#pragma unroll
            for (int j = 0; j < 4; ++j) rq[j] = f2{(rs + (int64_t)(2 * j) * HW)[upix], (rs + (int64_t)(2 * j + 1) * HW)[upix]};
        }
#pragma unroll 1
        for (int c = 0; c < C3P; ++c) {
            f2 r = rq[0];
            rq[0] = rq[1]; rq[1] = rq[2]; rq[2] = rq[3];
            if (!PROJ && c + 4 < C3P) rq[3] = f2{(rs + (int64_t)(2 * c + 8) * HW)[upix], (rs + (int64_t)(2 * c + 9) * HW)[upix]};
            if (PROJ) {              // projection shortcut: pointwise over x[t] (K row k is channel k), + its shift: the value the separate launch stores
                r = f2{0.f, 0.f};
                const wptr_t wd = (wptr_t)p.d.wp + c;
                const int wrow = p.d.Cdpad / 2;
#pragma unroll
                for (int k = 0; k < CM; ++k) r = __builtin_elementwise_fma(wd[(int64_t)k * wrow], f2{xs_[k], xs_[k]}, r);
                r = f2{r.x + ((cfptr_t)p.d.shift)[2 * c], r.y + ((cfptr_t)p.d.shift)[2 * c + 1]};
            }
            f2 ov = f2{0.f, 0.f};
            {
                const wptr_t w = (wptr_t)p.c.wp + c;
                const int wrow = p.c.Cdpad / 2;
#pragma unroll
                for (int k = 0; k < CM; ++k) ov = __builtin_elementwise_fma(w[(int64_t)k * wrow], f2{a2[k], a2[k]}, ov);
            }
            const float v0 = fmaxf((ov.x + ((cfptr_t)p.c.shift)[2 * c]) + r.x, 0.f), v1 = fmaxf((ov.y + ((cfptr_t)p.c.shift)[2 * c + 1]) + r.y, 0.f);
            if (act) { (o + (int64_t)(2 * c) * HW)[upix] = v0; (o + (int64_t)(2 * c + 1) * HW)[upix] = v1; }
            fb_gate_collect(glo, ghi, 2 * c, act && v0 > 0.f, lane);
            fb_gate_collect(glo, ghi, 2 * c + 1, act && v1 > 0.f, lane);
        }
        fb_gate_store(p.c.gate_out, p.c.gate_out_stride, C3, glo, ghi, bit0, bit_end, lane);
    }
}

// =============================================================================================
// fast_block2_kernel: the same fused block, re-blocked (round 6, second version)
// =============================================================================================
// What the first version measured (profiles/r6_fastblock_*.txt): 64-80 us per forward block at 128 frames where its HBM bytes take 35 and
// its FMAs 10 -- every block of the launch is resident at once and walks the same phases in step (load-bound, issue-bound,
// store-bound: the phases add up instead of overlapping), with one pixel per lane per stage, so that every scalar weight row feeds ONE
// packed FMA per lane, a 64-bit address is computed for every load, and R = 4 rows per block recompute half of stage A for the halo.
// Here a block is FB2_NW = 4 waves on a strip of R = 8 rows (halo 10 / 8), and a wave owns up to PA = 3 chunks of 64 consecutive positions
// in stage A and PB = 2 chunks of 64 consecutive pixels in stages B / C (the first build of this version ran 2 waves with 5 / 4 chunks:
// 1.6 waves per SIMD, parked at s_waitcnt for half of their cycles with one group of K rows in flight -- PMC in profiles/r6_fastblock_pmc.txt):
//   * a weight row (one s_load_dwordx8) feeds PA (PB) packed FMAs per channel pair; the rows that contribute are COMPACTED once per
//     block (k-table rows that add an exact zero -- padding rows, taps outside the clip -- are not walked at all);
//   * global operands move through buffer instructions: one descriptor per tensor, the lane's position as a 32-bit offset computed once
//     per chunk, the row / channel / frame part as the scalar offset -- no per-access address arithmetic; positions outside the plane
//     carry an out-of-range offset (loads return 0, stores are dropped);
//   * a block is a CHAIN of round trips (tools/fb_phase_probe.py, profiles/r6_fastblock_phase_probe.txt: at 64 frames a block's stages cost
//     6 + 8 + 10 + 10 us one after the other, and de-phasing the blocks only adds the delay): the compacted table therefore lives in
//     REGISTERS (an entry is a v_readlane away), weight rows are requested EIGHT at a time (one scalar round trip, ~0.35 us under this
//     load, for 64 SGPRs -- behind a scheduling barrier: left alone the scheduler spreads them between the FMAs), stage A's operands run
//     in steps of four K rows with three steps in flight (36-48 loads per wave, ~3.3 waves per SIMD);
//   * the residual of the first eight output channels (or the projection's inputs) and the gate words of stage B are requested before
//     stage B starts and arrive under it, the residual of the next eight channels under each stage-C group.
// At 128 frames the block is then HBM-bound in its own way: 13 us of chain + the marginal 16.4 us per 64 frames = 4.7 TB/s over the
// 153 MB it moves (x in, the residual read again 20 us later -- evicted from the 4 MB L2 by then --, the output).
// Same k-ordered fmaf chain per output element, same epilogue operations in the same order as the first version: bit-identical to the
// separate launches (tests/test_gpu_video.py runs both versions against them).
typedef float f8 __attribute__((ext_vector_type(8)));
constexpr int FB2_NW = 4, FB2_PA = 3, FB2_PB = 2;          // waves per block, chunks of 64 positions / pixels per wave
constexpr unsigned FB2_OOB = 0x80000000u;
template <int CM> struct fb2_row;                                    // one packed weight row of CM output channels, as scalar loads see it
template <> struct fb2_row<8> { typedef f8 type; };
template <> struct fb2_row<4> { typedef f32x4 type; };
#define FB2_RFL(x) __builtin_amdgcn_readfirstlane(x)
__device__ __forceinline__ float fb2_ld(const __amdgpu_buffer_rsrc_t rs, const unsigned vo, const int so) {
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, (int)vo, so, 0));
}
// gate words of channel c for a chunk of 64 pixels: the ballot's two words written straight into lane c (no per-channel lane masks: the
// first build kept 32 of them in SGPR pairs and spilled).  Pixels past the strip's end contribute bits to words fb_gate_store drops
// (a strip is a whole number of words), so the caller need not mask them.
extern "C" __device__ int fb2_writelane(int value, int lane, int old) __asm("llvm.amdgcn.writelane.i32");      // v_writelane_b32 (clang has no builtin for it)
__device__ __forceinline__ void fb2_gate_collect(unsigned& lo, unsigned& hi, const int c, const bool on) {
    const unsigned long long m = __ballot(on);
    lo = (unsigned)fb2_writelane((int)(unsigned)m, c, (int)lo);
    hi = (unsigned)fb2_writelane((int)(unsigned)(m >> 32), c, (int)hi);
}
__device__ __forceinline__ void fb2_st(const float v, const __amdgpu_buffer_rsrc_t rs, const unsigned vo, const int so) {
    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), rs, (int)vo, so, 0);
}

// Backward stages read the 1-bit gates of CM channels at a chunk of 64 consecutive pixels that starts at bit index `bit0` of the gate
// rows (any alignment: stage A's chunks start a row above the strip).  Lane l < 3 CM requests word (bit0 >> 5) + l % 3 of channel l / 3 --
// ONE load instruction per chunk, requested with the stage's operands -- and fb2_gate_bits turns it into the lane's own bits (bit c =
// the gate of channel c at this lane's pixel): three v_readlane and a scalar funnel shift per channel.  (The first build loaded CM words
// per lane and chunk: 24 registers in flight and spills at four waves per SIMD.)
__device__ __forceinline__ unsigned fb2_gate_request(const __amdgpu_buffer_rsrc_t rg, const int stride, const int64_t bit0, const int CM, const int lane) {
    const int64_t w = (bit0 >> 5) + (lane % 3);                         // (floor: bit0 may be negative in front of the first frame -- such words read as 0)
    const unsigned vo = (lane < 3 * CM && w >= 0) ? (unsigned)((w + (int64_t)(lane / 3) * stride) * 4) : FB2_OOB;
    return __builtin_amdgcn_raw_buffer_load_b32(rg, (int)vo, 0, 0);
}
template <int CM>
__device__ __forceinline__ unsigned fb2_gate_bits(const unsigned words, const int64_t bit0, const int lane) {
    const unsigned sh = (unsigned)FB2_RFL((int)(bit0 & 31));
    unsigned bits = 0u;
#pragma unroll
    for (int c = 0; c < CM; ++c) {
        const unsigned long long lo = (unsigned)__builtin_amdgcn_readlane((int)words, 3 * c), mid = (unsigned)__builtin_amdgcn_readlane((int)words, 3 * c + 1),
                                 hi = (unsigned)__builtin_amdgcn_readlane((int)words, 3 * c + 2);
        const unsigned long long m = (((mid << 32) | lo) >> sh) | ((hi << (63u - sh)) << 1);      // bits bit0 .. bit0 + 63 of channel c's row
        bits |= (unsigned)((m >> lane) & 1ull) << c;
    }
    return bits;
}

template <int CM, int MODE>       // MODE 0: forward, identity shortcut; 1: forward, projection shortcut; 2: backward
__global__ void __launch_bounds__(64 * FB2_NW) __attribute__((amdgpu_waves_per_eu(4, 4)))      // (3.5 blocks per CU at 128 frames of 56 x 56: all resident at once)
fast_block2_kernel(const I2VFastBlockParams p) {
    constexpr bool FWD = MODE != 2, PROJ = MODE == 1;
    constexpr int CP = CM / 2, C3 = 4 * CM, NG = C3 / 8, NW = FB2_NW, NT = 64 * FB2_NW, PA = FB2_PA, PB = FB2_PB;
    typedef typename fb2_row<CM>::type wrow_t;
    typedef const __attribute__((address_space(4))) wrow_t* wrowp_t;
    typedef const __attribute__((address_space(4))) f8* w8p_t;
    extern __shared__ __attribute__((aligned(16))) float lds_f[];
    const int tid = threadIdx.x, lane = tid & 63, wave = FB2_RFL(tid >> 6);
    const int H = p.H, W = p.W, R = p.R, HW = H * W, AR = R + 2, AW = W + 2, CH = AR * AW;      // CH: channel stride of stage A's LDS image
    int* const tabA = reinterpret_cast<int*>(lds_f);                     // [Kpad_a] (byte offset of the row's plane in the source tensor, K row): the rows that contribute
    int* const tabB = tabA + 2 * p.a.Kpad;                               // [Kpad_b] (float offset in the LDS image, K row)
    int* const cnts = tabB + 2 * p.b.Kpad;                               // rows in tabA, tabB
    float* const A1 = lds_f + ((2 * p.a.Kpad + 2 * p.b.Kpad + 4 + 3) & ~3);      // [CM][AR][AW]
    // block -> (clip, frame, strip): as in the first version (a clip's frames over a group of strips on ONE XCD)
    const unsigned b = blockIdx.x, xcd = b & 7u, slot = b >> 3;
    const unsigned uix = fastdiv(slot, p.dv_bu_m, p.dv_bu_s), bi = slot - uix * p.BU, un = xcd + 8u * uix;
    if (un >= (unsigned)p.U) return;
    const int clip = (int)fastdiv(un, p.dv_s_m, p.dv_s_s), sg = (int)un - clip * p.S;
    const int t = (int)fastdiv(bi, p.dv_g_m, p.dv_g_s), strip = sg * p.G + ((int)bi - t * p.G);
    const int n = clip * p.T + t, r0 = strip * R;
#ifdef I2V_EXPERIMENTAL          // probe hooks (tools/fb_phase_probe.py, experimental build only): I2V_FB_DELAY low byte = every other block sleeps first; bits 8 .. 10 = a stage's rows skipped
    const int probe = p.delay;
    if ((probe & 0xff) > 0 && (bi & 1u)) { for (int d = 0; d < (probe & 0xff); ++d) __builtin_amdgcn_s_sleep(127); }
#else
    constexpr int probe = 0;
#endif
    // ---- the K rows that contribute, compacted (wave 0: stage A's, wave 1: stage B's) ----
    if (wave < 2) {
        const I2VFastStage& st = wave == 0 ? p.a : p.b;
        int* const tab = wave == 0 ? tabA : tabB;
        int cnt = 0;
        for (int k0 = 0; k0 < st.Kpad; k0 += 64) {
            const int k = k0 + lane;
            bool ok = false; int off = 0;
            if (k < st.Kpad) {
                const I2VKEntry e = st.ktab[k];
                if (wave == 0) {
                    const int tf = t + (e.valid >> 1);
                    ok = (e.valid & 1) && tf >= 0 && tf < p.T;
                    off = (int)(((int64_t)(clip * p.T + tf) * p.src_nstride + e.chan_off) * 4);          // (< 2^31: k_fastblock checks)
                } else {
                    ok = (e.valid & 1) != 0;
                    off = ((int)fastdiv((unsigned)e.chan_off, p.dv_hw_m, p.dv_hw_s) * AR + e.dh) * AW + e.dw;
                }
            }
            const unsigned long long m = __ballot(ok);
            const int pos = cnt + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u));
            if (ok) { tab[2 * pos] = off; tab[2 * pos + 1] = k; }
            cnt += __popcll(m);
        }
        if (lane == 0) cnts[wave] = cnt;
    }
    for (int i = tid; i < CM * AR * 2; i += NT) A1[(i >> 1) * AW + ((i & 1) ? AW - 1 : 0)] = 0.f;      // zero columns 0 and W + 1 of every (channel, row)
    __syncthreads();
    const int NPOS = AR * W, NPIX = R * W;
    const int64_t pix_n = (int64_t)n * HW;                               // bit index of the frame's first pixel in a gate row
    // ---------------- stage A: the R + 2 rows, chunks wave, wave + NW, ... of 64 consecutive positions ----------------
    {
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)p.src, 0, (int)((int64_t)p.N * p.src_nstride * 4), 0x00020000);
        unsigned voff[PA]; int la[PA];
#pragma unroll
        for (int u = 0; u < PA; ++u) {
            const int q = (wave + NW * u) * 64 + lane;                    // position of the (R + 2) x W grid: plane pixel (r0 - 1) W + q (rows are contiguous)
            const int rowq = (int)fastdiv((unsigned)q, p.dv_w_m, p.dv_w_s), colq = q - rowq * W, prow = r0 - 1 + rowq;
            const bool ok = q < NPOS && prow >= 0 && prow < H;           // (outside the plane: the 3 x 3 stage's zero padding)
            voff[u] = ok ? (unsigned)(((r0 - 1) * W + q) * 4) : FB2_OOB;
            la[u] = q < NPOS ? rowq * AW + colq + 1 : -1;
        }
        unsigned gb[PA];                                                  // backward: the gate words of this stage's tensor for the lane's chunks (fb2_gate_request), bits after the K loop
#pragma unroll
        for (int u = 0; u < PA; ++u) gb[u] = 0u;
        if (!FWD) {
            const __amdgpu_buffer_rsrc_t rg = __builtin_amdgcn_make_buffer_rsrc((void*)p.a.gate, 0, CM * p.a.gate_stride * 4, 0x00020000);
#pragma unroll
            for (int u = 0; u < PA; ++u) gb[u] = fb2_gate_request(rg, p.a.gate_stride, pix_n + (int64_t)(r0 - 1) * W + (wave + NW * u) * 64, CM, lane);
        }
        f2 acc[PA][CP];
#pragma unroll
        for (int u = 0; u < PA; ++u)
#pragma unroll
            for (int c = 0; c < CP; ++c) acc[u][c] = f2{0.f, 0.f};
        const int nvA = (probe & 0x100) ? 0 : FB2_RFL(cnts[0]);
        const int wstride = p.a.Cdpad;
        // the compacted table in registers: lane l holds rows l and 64 + l (a stage has at most 128 rows: fb2_rows)
        const int tS0 = tabA[2 * lane], tK0 = tabA[2 * lane + 1], tS1 = tabA[2 * (64 + lane)], tK1 = tabA[2 * (64 + lane) + 1];
        // Operands in steps of FOUR K rows, three steps in flight (a ring of four register buffers: 36-48 loads per wave); weight rows in
        // batches of EIGHT (two steps), one scalar round trip each.  (Eight-row operand steps one ahead waited a memory round trip per
        // step -- 11.8 us of a block's 32 at 64 frames; three eight-row steps in flight do not fit 128 registers.)
        f2 xr[4][2][PA];                                                 // (two K rows per register PAIR: v_pk_fma_f32 takes its broadcast operand from either half of an aligned pair)
        wrow_t wv[8];
#define FB2_A_LOAD(G, B)                                                                                       \
        {                                                                                                      \
            const int r0_ = 4 * (G);                                                                           \
            const int tv = r0_ >= 64 ? tS1 : tS0;                                                              \
            _Pragma("unroll") for (int j = 0; j < 4; ++j) {                                                    \
                const int so = __builtin_amdgcn_readlane(tv, (r0_ + j) & 63);                                  \
                _Pragma("unroll") for (int u = 0; u < PA; ++u) {                                               \
                    const float xv = fb2_ld(rs, voff[u], so);                                                  \
                    if (j & 1) xr[B][j >> 1][u].y = xv; else xr[B][j >> 1][u].x = xv;                          \
                }                                                                                              \
            }                                                                                                  \
            __builtin_amdgcn_sched_barrier(0);                                                                 \
        }
#define FB2_A_W(G, NR)      /* the weight rows of steps G and G + 1 (NR = 8), or of step G alone (NR = 4): requested back to back, ONE round trip */ \
        {                                                                                                      \
            const int r0_ = 4 * (G);                                                                           \
            const int tv = r0_ >= 64 ? tK1 : tK0;                                                              \
            _Pragma("unroll") for (int j = 0; j < (NR); ++j)                                                   \
                wv[j] = *(wrowp_t)(p.a.wp + (int64_t)__builtin_amdgcn_readlane(tv, (r0_ + j) & 63) * wstride);      \
            __builtin_amdgcn_sched_barrier(0);                                                                 \
        }
#define FB2_A_FMA(B, J0)                                                                                       \
        {                                                                                                      \
            _Pragma("unroll") for (int j = 0; j < 4; ++j)                                                      \
                _Pragma("unroll") for (int u = 0; u < PA; ++u)                                                 \
                    _Pragma("unroll") for (int c = 0; c < CP; ++c)                                             \
                        acc[u][c] = __builtin_elementwise_fma(f2{wv[(J0) + j][2 * c], wv[(J0) + j][2 * c + 1]},                   \
                                                              (j & 1) ? f2{xr[B][j >> 1][u].y, xr[B][j >> 1][u].y} : f2{xr[B][j >> 1][u].x, xr[B][j >> 1][u].x}, acc[u][c]);      \
            __builtin_amdgcn_sched_barrier(0);                                                                 \
        }
        const int ng = nvA >> 2;
        if (ng > 0) FB2_A_LOAD(0, 0)
        if (ng > 1) FB2_A_LOAD(1, 1)
        if (ng > 2) FB2_A_LOAD(2, 2)
        int g = 0;
        for (; g + 7 <= ng; g += 4) {                                    // steady state: every load unconditional (the compiler's vmcnt then counts exactly)
            FB2_A_LOAD(g + 3, 3) FB2_A_W(g, 8) FB2_A_FMA(0, 0)
            FB2_A_LOAD(g + 4, 0) FB2_A_FMA(1, 4)
            FB2_A_LOAD(g + 5, 1) FB2_A_W(g + 2, 8) FB2_A_FMA(2, 0)
            FB2_A_LOAD(g + 6, 2) FB2_A_FMA(3, 4)
        }
        for (; g + 4 <= ng; g += 4) {                                    // the last steps: buffers 0 .. 2 hold steps g .. g + 2
            if (g + 3 < ng) FB2_A_LOAD(g + 3, 3)
            FB2_A_W(g, 8) FB2_A_FMA(0, 0)
            if (g + 4 < ng) FB2_A_LOAD(g + 4, 0)
            FB2_A_FMA(1, 4)
            if (g + 5 < ng) FB2_A_LOAD(g + 5, 1)
            FB2_A_W(g + 2, 8) FB2_A_FMA(2, 0)
            if (g + 6 < ng) FB2_A_LOAD(g + 6, 2)
            FB2_A_FMA(3, 4)
        }
        if (g + 2 <= ng) { FB2_A_W(g, 8) FB2_A_FMA(0, 0) FB2_A_FMA(1, 4) if (g + 2 < ng) { FB2_A_W(g + 2, 4) FB2_A_FMA(2, 0) } }
        else if (g < ng) { FB2_A_W(g, 4) FB2_A_FMA(0, 0) }
        for (int i = 4 * ng; i < nvA; ++i) {                             // (a row count that is not a multiple of 4: the last rows one by one)
            const int so = __builtin_amdgcn_readlane(i >= 64 ? tS1 : tS0, i & 63), kk = __builtin_amdgcn_readlane(i >= 64 ? tK1 : tK0, i & 63);
            const wrow_t wv = *(wrowp_t)(p.a.wp + (int64_t)kk * wstride);
#pragma unroll
            for (int u = 0; u < PA; ++u) {
                const float x = fb2_ld(rs, voff[u], so);
#pragma unroll
                for (int c = 0; c < CP; ++c) acc[u][c] = __builtin_elementwise_fma(f2{wv[2 * c], wv[2 * c + 1]}, f2{x, x}, acc[u][c]);
            }
        }
#undef FB2_A_LOAD
#undef FB2_A_W
#undef FB2_A_FMA
        float sh[CM];
#pragma unroll
        for (int c = 0; c < CM; ++c) sh[c] = FWD ? ((cfptr_t)p.a.shift)[c] : 0.f;
        if (!FWD) {
#pragma unroll
            for (int u = 0; u < PA; ++u) gb[u] = fb2_gate_bits<CM>(gb[u], pix_n + (int64_t)(r0 - 1) * W + (wave + NW * u) * 64, lane);
        }
#pragma unroll
        for (int u = 0; u < PA; ++u) {
            if (la[u] < 0) continue;
            const bool inside = voff[u] != FB2_OOB;
#pragma unroll
            for (int c = 0; c < CM; ++c) {
                float v = (c & 1) ? acc[u][c >> 1].y : acc[u][c >> 1].x;
                if (FWD) v = fmaxf(v + sh[c], 0.f);
                else if (!((gb[u] >> c) & 1u)) v = 0.f;
                A1[c * CH + la[u]] = inside ? v : 0.f;
            }
        }
    }
    __syncthreads();
    // ---------------- stages B (and C): the R rows, chunks NW - 1 - wave, 2 NW - 1 - wave, ... of 64 consecutive pixels ----------------
    const int64_t bit_strip = pix_n + (int64_t)r0 * W, bit_end = bit_strip + NPIX;
    const __amdgpu_buffer_rsrc_t rd = __builtin_amdgcn_make_buffer_rsrc((void*)p.dst, 0, (int)((int64_t)p.N * p.dst_nstride * 4), 0x00020000);
    const int dst_n = (int)((int64_t)n * p.dst_nstride * 4);
    int ctr[PB]; unsigned vo[PB]; bool act[PB];
#pragma unroll
    for (int u = 0; u < PB; ++u) {
        const int pp = ((NW - 1 - wave) + NW * u) * 64 + lane;
        act[u] = pp < NPIX;
        const int ppc = act[u] ? pp : 0;
        const int row = (int)fastdiv((unsigned)ppc, p.dv_w_m, p.dv_w_s), col = ppc - row * W;
        ctr[u] = (row + 1) * AW + col + 1;                               // the lane's own position in channel 0 of the LDS image
        vo[u] = act[u] ? (unsigned)((r0 * W + pp) * 4) : FB2_OOB;        // ... and in a channel plane of a global tensor (bytes)
    }
    // operands of the later stages, requested now: the residual (identity shortcut) or the projection's inputs; backward: gate words
    float res[2][(FWD && !PROJ) ? PB : 1][(FWD && !PROJ) ? 8 : 1];       // the residual of a group of 8 output channels, two groups (this one, the next one in flight)
    f2 xs[PROJ ? PB : 1][PROJ ? CP : 1];                                   // the projection's inputs, two channels per register pair
    unsigned gbb[PB];                                                     // backward: bit c = the gate of channel c of stage B's tensor at the lane's pixel
    const __amdgpu_buffer_rsrc_t rr = __builtin_amdgcn_make_buffer_rsrc((void*)((FWD && !PROJ) ? p.add0 : p.src), 0,
                                                                         (int)((int64_t)p.N * ((FWD && !PROJ) ? p.add0_nstride : p.src_nstride) * 4), 0x00020000);
    const int add_n = (int)((int64_t)n * p.add0_nstride * 4);
#define FB2_RES_LOAD(GQ, B)                                                                                    \
    _Pragma("unroll") for (int c = 0; c < 8; ++c)                                                              \
        _Pragma("unroll") for (int u = 0; u < PB; ++u) res[B][u][c] = fb2_ld(rr, vo[u], add_n + (8 * (GQ) + c) * HW * 4);
    if (FWD && !PROJ) { FB2_RES_LOAD(0, 0) }
    if (PROJ) {
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)p.src, 0, (int)((int64_t)p.N * p.src_nstride * 4), 0x00020000);
        const int src_n = (int)((int64_t)n * p.src_nstride * 4);
#pragma unroll
        for (int k = 0; k < CM; ++k)
#pragma unroll
            for (int u = 0; u < PB; ++u) { const float xv = fb2_ld(rs, vo[u], src_n + k * HW * 4); if (k & 1) xs[u][k >> 1].y = xv; else xs[u][k >> 1].x = xv; }
    }
#pragma unroll
    for (int u = 0; u < PB; ++u) gbb[u] = 0u;
    if (!FWD) {
        const __amdgpu_buffer_rsrc_t rg = __builtin_amdgcn_make_buffer_rsrc((void*)p.b.gate, 0, CM * p.b.gate_stride * 4, 0x00020000);
#pragma unroll
        for (int u = 0; u < PB; ++u) gbb[u] = fb2_gate_request(rg, p.b.gate_stride, bit_strip + ((NW - 1 - wave) + NW * u) * 64, CM, lane);
    }
    if (FWD) {                       // stage A's own gates, from the finished values in LDS (post-ReLU: > 0 is the bit)
#pragma unroll
        for (int u = 0; u < PB; ++u) {
            const int64_t bit0 = bit_strip + ((NW - 1 - wave) + NW * u) * 64;
            if (bit0 >= bit_end) continue;                               // (wave-uniform: this chunk does not exist)
            unsigned glo = 0, ghi = 0;
#pragma unroll
            for (int c = 0; c < CM; ++c) fb2_gate_collect(glo, ghi, c, A1[c * CH + ctr[u]] > 0.f);
            fb_gate_store(p.a.gate_out, p.a.gate_out_stride, CM, glo, ghi, bit0, bit_end, lane);
        }
    }
    f2 a2[PB][CP];                                                        // stage B's result, two channels per register pair
    {
        f2 acc[PB][CP];
#pragma unroll
        for (int u = 0; u < PB; ++u)
#pragma unroll
            for (int c = 0; c < CP; ++c) acc[u][c] = f2{0.f, 0.f};
        const int nvB = (probe & 0x200) ? 0 : FB2_RFL(cnts[1]);
        const int wstride = p.b.Cdpad;
        const int tS0 = tabB[2 * lane], tK0 = tabB[2 * lane + 1], tS1 = tabB[2 * (64 + lane)], tK1 = tabB[2 * (64 + lane) + 1];      // the table in registers, as in stage A
        int i = 0;
        for (; i + 8 <= nvB; i += 8) {                                   // eight K rows per step: operands (LDS) and weight rows (one scalar round trip) requested together
            const int tvs = i >= 64 ? tS1 : tS0, tvk = i >= 64 ? tK1 : tK0;
            f2 x[4][PB]; wrow_t wv[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int so = __builtin_amdgcn_readlane(tvs, (i + j) & 63);
#pragma unroll
                for (int u = 0; u < PB; ++u) { const float xv = A1[ctr[u] + so]; if (j & 1) x[j >> 1][u].y = xv; else x[j >> 1][u].x = xv; }
                wv[j] = *(wrowp_t)(p.b.wp + (int64_t)__builtin_amdgcn_readlane(tvk, (i + j) & 63) * wstride);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int j = 0; j < 8; ++j)
#pragma unroll
                for (int u = 0; u < PB; ++u)
#pragma unroll
                    for (int c = 0; c < CP; ++c)
                        acc[u][c] = __builtin_elementwise_fma(f2{wv[j][2 * c], wv[j][2 * c + 1]}, (j & 1) ? f2{x[j >> 1][u].y, x[j >> 1][u].y} : f2{x[j >> 1][u].x, x[j >> 1][u].x}, acc[u][c]);
            asm volatile("" ::: "memory");
        }
        for (; i < nvB; ++i) {
            const int so = __builtin_amdgcn_readlane(i >= 64 ? tS1 : tS0, i & 63), kk = __builtin_amdgcn_readlane(i >= 64 ? tK1 : tK0, i & 63);
            const wrow_t wv = *(wrowp_t)(p.b.wp + (int64_t)kk * wstride);
#pragma unroll
            for (int u = 0; u < PB; ++u) {
                const float x = A1[ctr[u] + so];
#pragma unroll
                for (int c = 0; c < CP; ++c) acc[u][c] = __builtin_elementwise_fma(f2{wv[2 * c], wv[2 * c + 1]}, f2{x, x}, acc[u][c]);
            }
        }
        if (!FWD) {
#pragma unroll
            for (int u = 0; u < PB; ++u) gbb[u] = fb2_gate_bits<CM>(gbb[u], bit_strip + ((NW - 1 - wave) + NW * u) * 64, lane);
        }
#pragma unroll
        for (int u = 0; u < PB; ++u) {
#pragma unroll
            for (int c = 0; c < CM; ++c) {
                float v = (c & 1) ? acc[u][c >> 1].y : acc[u][c >> 1].x;
                if (FWD) v = fmaxf(v + ((cfptr_t)p.b.shift)[c], 0.f);
                else if (!((gbb[u] >> c) & 1u)) v = 0.f;
                if (c & 1) a2[u][c >> 1].y = v; else a2[u][c >> 1].x = v;
            }
        }
    }
    if (!FWD) {                      // backward: stage B's result is the output
#pragma unroll
        for (int c = 0; c < CM; ++c)
#pragma unroll
            for (int u = 0; u < PB; ++u) fb2_st((c & 1) ? a2[u][c >> 1].y : a2[u][c >> 1].x, rd, vo[u], dst_n + c * HW * 4);
        return;
    }
#pragma unroll
    for (int u = 0; u < PB; ++u) {
        const int64_t bit0 = bit_strip + ((NW - 1 - wave) + NW * u) * 64;
        if (bit0 >= bit_end) continue;
        unsigned glo = 0, ghi = 0;
#pragma unroll
        for (int c = 0; c < CM; ++c) fb2_gate_collect(glo, ghi, c, ((c & 1) ? a2[u][c >> 1].y : a2[u][c >> 1].x) > 0.f);
        fb_gate_store(p.b.gate_out, p.b.gate_out_stride, CM, glo, ghi, bit0, bit_end, lane);
    }
    if (probe & 0x400) return;
    // ---------------- stage C: conv3 (pointwise over the lane's own a2: K row k is channel k) + residual + ReLU, 8 output channels at a time ----------------
    unsigned glo[PB], ghi[PB];
#pragma unroll
    for (int u = 0; u < PB; ++u) glo[u] = ghi[u] = 0u;
#pragma unroll
    for (int gq = 0; gq < NG; ++gq) {
        asm volatile("" ::: "memory");                                   // (one group's weight rows in SGPRs at a time)
        f2 ov[PB][4];
#pragma unroll
        for (int u = 0; u < PB; ++u)
#pragma unroll
            for (int j = 0; j < 4; ++j) ov[u][j] = f2{0.f, 0.f};
        {
            f8 w8[CM];                                                   // all CM rows of this group's weights: ONE scalar round trip
#pragma unroll
            for (int k = 0; k < CM; ++k) w8[k] = *(w8p_t)(p.c.wp + (int64_t)k * p.c.Cdpad + 8 * gq);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int k = 0; k < CM; ++k)
#pragma unroll
                for (int u = 0; u < PB; ++u)
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        ov[u][j] = __builtin_elementwise_fma(f2{w8[k][2 * j], w8[k][2 * j + 1]}, (k & 1) ? f2{a2[u][k >> 1].y, a2[u][k >> 1].y} : f2{a2[u][k >> 1].x, a2[u][k >> 1].x}, ov[u][j]);
            asm volatile("" ::: "memory");
        }
        f2 rv[PB][4];
        if (PROJ) {                  // projection shortcut: pointwise over x[t] (K row k is channel k), + its shift: the value the separate launch stores
#pragma unroll
            for (int u = 0; u < PB; ++u)
#pragma unroll
                for (int j = 0; j < 4; ++j) rv[u][j] = f2{0.f, 0.f};
            {
                f8 w8[CM];
#pragma unroll
                for (int k = 0; k < CM; ++k) w8[k] = *(w8p_t)(p.d.wp + (int64_t)k * p.d.Cdpad + 8 * gq);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int k = 0; k < CM; ++k)
#pragma unroll
                    for (int u = 0; u < PB; ++u)
#pragma unroll
                        for (int j = 0; j < 4; ++j)
                            rv[u][j] = __builtin_elementwise_fma(f2{w8[k][2 * j], w8[k][2 * j + 1]}, (k & 1) ? f2{xs[u][k >> 1].y, xs[u][k >> 1].y} : f2{xs[u][k >> 1].x, xs[u][k >> 1].x}, rv[u][j]);
                asm volatile("" ::: "memory");
            }
            const f8 sd = *(w8p_t)(p.d.shift + 8 * gq);
#pragma unroll
            for (int u = 0; u < PB; ++u)
#pragma unroll
                for (int j = 0; j < 4; ++j) rv[u][j] = f2{rv[u][j].x + sd[2 * j], rv[u][j].y + sd[2 * j + 1]};
        } else {
            if (gq + 1 < NG) { FB2_RES_LOAD(gq + 1, (gq + 1) & 1) }
#pragma unroll
            for (int u = 0; u < PB; ++u)
#pragma unroll
                for (int j = 0; j < 4; ++j) rv[u][j] = f2{res[gq & 1][u][2 * j], res[gq & 1][u][2 * j + 1]};
        }
        const f8 sc = *(w8p_t)(p.c.shift + 8 * gq);
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int ch = 8 * gq + 2 * j + h;
#pragma unroll
                for (int u = 0; u < PB; ++u) {
                    const float v = fmaxf(((h ? ov[u][j].y : ov[u][j].x) + sc[2 * j + h]) + (h ? rv[u][j].y : rv[u][j].x), 0.f);
                    fb2_st(v, rd, vo[u], dst_n + ch * HW * 4);
                    fb2_gate_collect(glo[u], ghi[u], ch, v > 0.f);
                }
            }
    }
#pragma unroll
    for (int u = 0; u < PB; ++u) {
        const int64_t bit0 = bit_strip + ((NW - 1 - wave) + NW * u) * 64;
        if (bit0 >= bit_end) continue;
        fb_gate_store(p.c.gate_out, p.c.gate_out_stride, C3, glo[u], ghi[u], bit0, bit_end, lane);
    }
}

static void fb_stage(I2VFastStage& st, const I2VConvParams& q) {
    st.wp = q.wp; st.ktab = q.ktab; st.Kpad = q.Kpad; st.Cdpad = q.Cdpad; st.shift = q.shift; st.relu = q.relu;
    st.gate = q.gate; st.gate_stride = q.gate_stride; st.gate_pix0 = q.gate_pix0;
    st.gate_out = q.gate_out; st.gate_out_stride = q.gate_out_stride; st.gate_out_pix0 = q.gate_out_pix0;
}

// Rows per block for the second version: the largest R (whole gate words per strip, a divisor of H) whose R + 2 rows fit FB2_NW x FB2_PA chunks
// of 64 positions and whose R rows fit FB2_NW x FB2_PB chunks of 64 pixels; 0: the first version runs the block.
static int fb2_rows(const I2VConvParams& a, const I2VConvParams& b, const I2VConvParams* c, const I2VConvParams* d) {
    const char* const e1 = getenv("I2V_FB_V1");                       // (read per launch: the GPU test runs both versions in one process)
    if (e1 && e1[0] == '1') return 0;
    const int H = a.Hg, W = a.Wg;
    int g = W, r32 = 32; while (r32) { const int t = g % r32; g = r32; r32 = t; }      // gcd(W, 32)
    const int Rq = 32 / g;
    int best = 0;
    for (int R = Rq; R <= H; R += Rq)
        if (H % R == 0 && (R + 2) * W <= 64 * FB2_NW * FB2_PA && R * W <= 64 * FB2_NW * FB2_PB && (size_t)a.Cd * (R + 2) * (W + 2) * 4 <= 56 * 1024) best = R;
    if (!best) return 0;
    auto small = [&](int64_t nstride) { return (int64_t)a.N * nstride * 4 < (1ll << 31); };       // every tensor through one buffer descriptor, offsets in 32 bits
    if (!small(a.src_nstride) || !small(c ? c->dst_nstride : b.dst_nstride) || (c && !d && !small(c->add0_nstride))) return 0;
    if (c && (c->Cd % 8 != 0 || c->Cdpad < c->Cd || (d && d->Cdpad < d->Cd))) return 0;
    if (a.Kpad > 128 || b.Kpad > 128) return 0;                       // (a stage's compacted table lives in two registers per lane)
    return best;
}

int k_fastblock(const I2VConvParams& a, const I2VConvParams& b, const I2VConvParams* c, const I2VConvParams* d, i2v_stream_t s) {
    int R = i2v_fastblock_rows(a, b, c, d);
    if (R <= 0) { snprintf(g_be_err, sizeof g_be_err, "fast-block launch: the convolutions are not eligible"); g_be_has_err = true; return 1; }
    const int R2 = fb2_rows(a, b, c, d);
    if (R2 > 0) R = R2;
    I2VFastBlockParams p;
    memset(&p, 0, sizeof p);
    p.mode = c ? 0 : 1; p.CM = a.Cd;
    p.src = a.src; p.src_nstride = a.src_nstride; p.Cs = a.Cs;
    p.N = a.N; p.T = a.Tg; p.H = a.Hg; p.W = a.Wg; p.R = R;
    fb_stage(p.a, a); fb_stage(p.b, b);
    if (c) { fb_stage(p.c, *c); p.dst = c->dst; p.dst_nstride = c->dst_nstride; if (!d) { p.add0 = c->add0; p.add0_nstride = c->add0_nstride; } }
    else { p.dst = b.dst; p.dst_nstride = b.dst_nstride; }
    if (d) fb_stage(p.d, *d);
    fastdiv_magic((unsigned)p.W, &p.dv_w_m, &p.dv_w_s);
    fastdiv_magic((unsigned)(p.H * p.W), &p.dv_hw_m, &p.dv_hw_s);
    if (p.N <= 0) return 0;
    // units for the XCD mapping: a clip's frames over a group of G strips, S groups per frame -- the fewest groups that give every XCD a unit
    const int strips = p.H / R, clips = p.N / p.T;
    int S = strips;
    for (int dv = 1; dv <= strips; ++dv) if (strips % dv == 0 && (int64_t)clips * dv >= 8) { S = dv; break; }
    p.S = S; p.G = strips / S; p.U = clips * S; p.BU = p.T * p.G;
    fastdiv_magic((unsigned)p.BU, &p.dv_bu_m, &p.dv_bu_s); fastdiv_magic((unsigned)p.S, &p.dv_s_m, &p.dv_s_s); fastdiv_magic((unsigned)p.G, &p.dv_g_m, &p.dv_g_s);
    const int64_t nblk = 8ll * ((p.U + 7) / 8) * p.BU;
    if (nblk > 0x7fffffff) { snprintf(g_be_err, sizeof g_be_err, "fast-block grid too large"); g_be_has_err = true; return 1; }
    const dim3 grid((unsigned)nblk);
    if ((int64_t)p.T * p.src_nstride >= (1ll << 30)) { snprintf(g_be_err, sizeof g_be_err, "fast-block launch: a clip's frames span more than 2^30 elements"); g_be_has_err = true; return 1; }
    __atomic_fetch_add(&g_stat_fastblock, 1, __ATOMIC_RELAXED);
    if (R2 > 0) {
        { const char* e = getenv("I2V_FB_DELAY"); p.delay = e ? atoi(e) : 0; }
        const size_t lds2 = (((size_t)2 * a.Kpad + 2 * b.Kpad + 4 + 3) & ~(size_t)3) * sizeof(int) + (size_t)p.CM * (R + 2) * (p.W + 2) * sizeof(float);
#define FB2_GO(CMV, MD) hipLaunchKernelGGL((fast_block2_kernel<CMV, MD>), grid, dim3(64 * FB2_NW), lds2, (hipStream_t)s, p)
        if (p.CM == 8) { if (!c) FB2_GO(8, 2); else if (d) FB2_GO(8, 1); else FB2_GO(8, 0); }
        else { if (!c) FB2_GO(4, 2); else if (d) FB2_GO(4, 1); else FB2_GO(4, 0); }
#undef FB2_GO
        LAUNCH_CHECK("fast_block2_kernel");
        return 0;
    }
    const size_t lds = (((size_t)a.Kpad + b.Kpad + 3) & ~(size_t)3) * sizeof(int) + (size_t)p.CM * (R + 2) * (p.W + 2) * sizeof(float);
#define FB_GO(CMV, FW, PR) hipLaunchKernelGGL((fast_block_kernel<CMV, FW, PR>), grid, dim3(128), lds, (hipStream_t)s, p)
    if (p.CM == 8) { if (!c) FB_GO(8, false, false); else if (d) FB_GO(8, true, true); else FB_GO(8, true, false); }
    else { if (!c) FB_GO(4, false, false); else if (d) FB_GO(4, true, true); else FB_GO(4, true, false); }
#undef FB_GO
    LAUNCH_CHECK("fast_block_kernel");
    return 0;
}

// =============================================================================================
// conv_vfma: ONE narrow convolution launch on packed-fp32 vector FMAs (autotuner bit 11)
// =============================================================================================
// The launches of the fast pathway that the fused block leaves alone -- conv1's input gradient (3 x 1 x 1 transposed: K = 24 -> 32 channels,
// + the shortcut's gradient, gated), a first block's projection gradient and conv1 gradient (K = 32 / 24 -> 8) -- are <= 32 K rows deep and
// <= 32 channels wide, with taps at (0, 0) only: through conv_tile they run at 5-14 TFLOP/s in 28-45 us (16-row fragments half empty, a
// 16-row K chunk per barrier, 64 x 64-pixel tiles for 8 channels).  Here a lane owns TWO pixels of one grid frame, requests all their K
// operands at once (<= 64 registers, every load in flight together), then walks the output channel PAIRS in a rolled loop: K packed FMAs
// per pixel (the weight pair a scalar load, alive for one iteration), the dense epilogue of conv_vec_rows for the two channels (shift,
// addends, ReLU, gate bit, store, own gate words by ballot), the next pair's addends requested an iteration ahead.  Every output element is
// the same k-ordered fmaf chain (rows that add an exact zero skipped): bit-identical to the conv_tile launch, so the autotuner may choose.
// One pixel per lane; EVERYTHING the lane will read is requested up front -- its K operands, its addends, its gate words (<= 96 loads in
// flight per lane) -- so that a block is ONE memory round trip, the arithmetic, the stores.  (The first version fetched the addends of
// channel pair c + 1 during pair c: sixteen dependent round trips per block, 108 us where conv_tile takes 38.)  The channel loop is
// unrolled (register-resident addends need compile-time indices) with a compiler memory fence per channel pair: without it the
// scheduler requests every pair's weight rows at once and spills SGPR tuples to VGPR lanes.
template <int KP, int CDP>       // K rows held in registers (16 / 32), output channel pairs (4 / 8 / 16)
__global__ void __launch_bounds__(128) __attribute__((amdgpu_waves_per_eu(4, 4)))
conv_vfma_kernel(const I2VConvParams p) {
    const int tid = threadIdx.x, lane = tid & 63;
    const int HW = p.Hg * p.Wg;
    const int n = blockIdx.y;                                                       // grid frame (scalar)
    const int clip = (int)fastdiv((unsigned)n, p.dv_t_m, p.dv_t_s), tg = n - clip * p.Tg;
    const int nf = clip * p.To + tg;                                                // destination frame (dense: To == Tg, ost == 1, ot0 == 0)
    // lane k decodes K row k: element offset from the clip's first source frame, or FB_SKIP (padding row / tap outside the clip: an exact zero)
    int myoff = FB_SKIP;
    if (lane < KP && lane < p.Kpad) {
        const I2VKEntry e = p.ktab[lane];
        const int tf = tg * p.st + (e.valid >> 1);
        if ((e.valid & 1) && tf >= 0 && tf < p.Ts) myoff = (int)((int64_t)tf * p.src_nstride) + e.chan_off;
    }
    const float* const fbase = p.src + (int64_t)(clip * p.Ts) * p.src_nstride;
    const int pix = blockIdx.x * 128 + tid;
    const bool act = pix < HW;
    const unsigned upx = act ? (unsigned)pix : 0u;
    const float* const a0 = p.add0 ? p.add0 + (int64_t)nf * p.add0_nstride : nullptr;
    const float* const a1 = p.add1 ? p.add1 + (int64_t)nf * p.add1_nstride : nullptr;
    const int64_t bitp = p.gate_pix0 + (int64_t)nf * HW + upx;
    // ---- every load of the block, at once ----
    float x[KP];
#pragma unroll
    for (int k = 0; k < KP; ++k) {
        const int o = __builtin_amdgcn_readlane(myoff, k);                          // (uniform: a scalar base per K row)
        x[k] = (fbase + (o == FB_SKIP ? 0 : o))[upx];
        if (o == FB_SKIP) x[k] = 0.f;
    }
    float ad[2 * CDP];
#pragma unroll
    for (int ch = 0; ch < 2 * CDP; ++ch) {
        ad[ch] = (a0 && ch < p.Cd) ? (a0 + (int64_t)ch * HW)[upx] : 0.f;          // (add1 -- a second shortcut, rare -- is read where it is added: the order v + add0 + add1 stays)
    }
    unsigned gbits = ~0u;                                                           // bit ch: the gate of channel ch at this pixel
    if (p.gate) {
        unsigned gw[2 * CDP];
#pragma unroll
        for (int ch = 0; ch < 2 * CDP; ++ch) gw[ch] = ch < p.Cd ? p.gate[(int64_t)ch * p.gate_stride + (bitp >> 5)] : ~0u;
        gbits = 0u;
#pragma unroll
        for (int ch = 0; ch < 2 * CDP; ++ch) gbits |= ((gw[ch] >> ((unsigned)bitp & 31u)) & 1u) << ch;
    }
    // ---- the arithmetic: one output-channel pair at a time, K packed FMAs, the dense epilogue of conv_vec_rows ----
    const int wrow = p.Cdpad / 2;
    float* const ob = p.dst + (int64_t)nf * p.dst_nstride;
    const int64_t bitw0 = p.gate_out_pix0 + (int64_t)nf * HW + blockIdx.x * 128 + (tid - lane);      // this wave's first pixel as a bit index: a multiple of 32
    unsigned glo = 0, ghi = 0;
#pragma unroll
    for (int c = 0; c < CDP; ++c) {
        asm volatile("" ::: "memory");                                              // (one pair's weight rows in SGPRs at a time)
        const wptr_t w = (wptr_t)p.wp + c;
        f2 acc = f2{0.f, 0.f};
#pragma unroll
        for (int k = 0; k < KP; ++k) acc = __builtin_elementwise_fma(w[(int64_t)k * wrow], f2{x[k], x[k]}, acc);
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int ch = 2 * c + h;
            const bool on = ch < p.Cd;
            float v = h ? acc.y : acc.x;
            if (p.shift) v += on ? ((cfptr_t)p.shift)[ch] : 0.f;
            if (a0) v += ad[ch];
            if (a1 && on) v += (a1 + (int64_t)ch * HW)[upx];
            if (p.relu) v = fmaxf(v, 0.f);
            if (p.gate && !((gbits >> ch) & 1u)) v = 0.f;
            if (act && on) (ob + (int64_t)ch * HW)[upx] = v;
            if (p.gate_out) fb_gate_collect(glo, ghi, ch, act && on && v > 0.f, lane);
        }
    }
    if (p.gate_out) fb_gate_store(p.gate_out, p.gate_out_stride, p.Cd, glo, ghi, bitw0, p.gate_out_pix0 + (int64_t)(nf + 1) * HW, lane);
}

// eligibility on the planned parameters (no field the executor fills in later is read): a dense same-size launch with taps at (0, 0) only
// (the planner's mark: halo == 1), at most 32 K rows and 32 output channels
bool conv_vfma_ok(const I2VConvParams& p) {
    if (p.halo != 1 || p.quad || p.pre_scale || p.gate_scale || p.mask || p.blk > 1 || p.blkt > 1 || p.Kpad > 32 || p.Cd > 32 || p.Cd < 2) return false;
    if (p.sh != 1 || p.sw != 1 || p.osh != 1 || p.osw != 1 || p.oh0 || p.ow0 || p.Hs != p.Hg || p.Hg != p.Ho || p.Ws != p.Wg || p.Wg != p.Wo) return false;
    if (p.Tg != p.To || p.ost != 1 || p.ot0 != 0 || (p.add0 && p.add0_stride != 1)) return false;
    if ((p.gate || p.gate_out) && (p.Hg * p.Wg) % 32 != 0) return false;
    return (int64_t)p.Ts * p.src_nstride < (1ll << 30);
}
int launch_conv_vfma(const I2VConvParams& p, hipStream_t s) {
    const int HW = p.Hg * p.Wg;
    if (p.N <= 0 || HW <= 0) return 0;
    if (p.N > 65535) { snprintf(g_be_err, sizeof g_be_err, "conv_vfma launch: more than 65535 frames"); g_be_has_err = true; return 1; }
    const dim3 grid((unsigned)((HW + 127) / 128), (unsigned)p.N);
#define VF_GO(KPV, CDV) hipLaunchKernelGGL((conv_vfma_kernel<KPV, CDV>), grid, dim3(128), 0, s, p)
    const int cdp = (p.Cd + 1) / 2;
    if (p.Kpad <= 16) { if (cdp <= 4) VF_GO(16, 4); else if (cdp <= 8) VF_GO(16, 8); else VF_GO(16, 16); }
    else { if (cdp <= 4) VF_GO(32, 4); else if (cdp <= 8) VF_GO(32, 8); else VF_GO(32, 16); }
#undef VF_GO
    LAUNCH_CHECK("conv_vfma_kernel");
    return 0;
}

// EXPERIMENTAL kernels (-DI2V_EXPERIMENTAL; not compiled into the product library, `__graft_entry__.build()` leaves the flag off):
// built, tested bit for bit, measured -- and not what the product runs.
//   conv_fused_kernel   3x3 -> pointwise pair in one launch (round 4: loses 7-35 % at the headline size)
//   conv_igemm_bf3      split-bf16 K loop, I2V_MATH=bf16x3 (round 5: 1.19x end to end; ruled "narrower arithmetic", uncredited)
//   conv_pw_stream      persistent role-split pointwise kernel (round 5: slower than conv_igemm on every shape)
// Each keeps its opt-in switch (I2V_FUSE=1, I2V_MATH=bf16x3, I2V_PWS=1) and its tests, which skip on a default build.
#ifdef I2V_EXPERIMENTAL
#include "i2v_conv_tile.h"

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
int conv_pws_grid(const I2VConvParams& p) {           // blocks (one per CU), 0 = not applicable
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
int launch_conv_pws(const I2VConvParams& p, hipStream_t s) {
    const int grid = conv_pws_grid(p), n_cd = p.Cd / 64, n_streams = grid / n_cd;
    const int n_px = (int)(((int64_t)p.N * p.Hg * p.Wg + 63) / 64);
    if (p.K == 64) hipLaunchKernelGGL((conv_pw_stream<64>), dim3(grid), dim3(I2V_PWS_THREADS), 0, s, p, n_cd, n_streams, n_px);
    else if (p.K == 128) hipLaunchKernelGGL((conv_pw_stream<128>), dim3(grid), dim3(I2V_PWS_THREADS), 0, s, p, n_cd, n_streams, n_px);
    else hipLaunchKernelGGL((conv_pw_stream<256>), dim3(grid), dim3(I2V_PWS_THREADS), 0, s, p, n_cd, n_streams, n_px);
    LAUNCH_CHECK("conv_pw_stream");
    return 0;
}

template <int BD, int BP, int WD, int WP>
static int launch_conv_bf3_cfg(const I2VConvParams& p, hipStream_t s) {
    const int64_t P = (int64_t)p.N * p.Hg * p.Wg;
    const int n_cd = (p.Cd + BD - 1) / BD;
    const int64_t n_px = (P + BP - 1) / BP;
    const int64_t grid = n_px * n_cd;
    if (grid <= 0) return 0;
    if (grid > 0x7fffffff) { snprintf(g_be_err, sizeof g_be_err, "conv grid too large"); g_be_has_err = true; return 1; }
        if (conv_bf3_ok(p)) {     // split-bf16 K loop (bit 6 of the configuration: two chunks per barrier)
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
    snprintf(g_be_err, sizeof g_be_err, "split-bf16 launch: not eligible"); g_be_has_err = true;
    return 1;
}
int launch_conv_bf3(const I2VConvParams& p, hipStream_t s) {
    switch ((p.cfg - 1) & 7) {
        case 0: return launch_conv_bf3_cfg<128, 128, 2, 2>(p, s);
        case 1: return launch_conv_bf3_cfg<64, 128, 2, 2>(p, s);
        case 2: return launch_conv_bf3_cfg<128, 64, 2, 2>(p, s);
        default: return launch_conv_bf3_cfg<64, 64, 2, 2>(p, s);
    }
}

// Fused pair (conv_fused_kernel): the structural rule is i2v_conv_pair_fusable (i2v_kernels.h, shared with the host simulation).
// Returns 0 (no), 1 (plain staging only) or 3 (halo staging available too).  Whether a's output has OTHER readers is the planner's
// business (i2v_engine.cpp: mark_fusable).
int k_conv_fusable(const I2VConvParams& a, const I2VConvParams& b) {
    if (!i2v_conv_pair_fusable(a, b)) return 0;
    return (a.Cd == 64 && conv_halo_ok(a)) ? 3 : 1;
}

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
#endif

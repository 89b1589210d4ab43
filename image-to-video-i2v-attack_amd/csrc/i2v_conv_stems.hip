// The 3-channel stems on 2-D halo tiles (round 5): conv_imggrad_halo (the class-packed image gradient, autotuner bit 9) and
// conv_stem_halo (SlowFast's narrow fast stem forward, bit 10); under -DI2V_EXPERIMENTAL also conv_stem64_halo (the wide 7x7 / 2 stem:
// built, bit-identical, no gain in the attack -- DESIGN.md section 4).
#include "i2v_conv_tile.h"

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
bool conv_ighalo_ok(const I2VConvParams& p) {
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
int launch_conv_ighalo(const I2VConvParams& p, hipStream_t s) {
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
bool conv_stemhalo_ok(const I2VConvParams& p) {
    return p.quad == 2 && p.halo == 77 /* kh = 7, rows from -3 (pack_fwd) */ && p.quad_kw == 7 && p.quad_dw0 == -3 && p.sh == 2 && p.sw == 2 && p.blk == 1 && p.blkt == 2 && p.Cd <= 16 && p.Kpad == p.K &&
           p.K % (2 * SH_RPP) == 0 && p.Ws % 4 == 0 && p.src_nstride % 4 == 0 && p.osh == 1 && p.osw == 1 && p.oct == 1 && !p.pre_scale && !p.gate && !p.gate_out &&
           !p.add0 && !p.gate_scale && !p.ig_th;
}
int launch_conv_stemhalo(const I2VConvParams& p, hipStream_t s) {
    const int tiles_x = (p.Wg + 15) / 16, tiles_y = (p.Hg + 15) / 16, txy = tiles_x * tiles_y;
    const int64_t grid = (int64_t)p.N * txy;
    if (grid <= 0) return 0;
    if (grid > 0x7fffffff || ((uintptr_t)p.src & 15)) { snprintf(g_be_err, sizeof g_be_err, "stem halo launch: grid too large or source not 16-byte aligned"); g_be_has_err = true; return 1; }
    hipLaunchKernelGGL(conv_stem_halo, dim3((unsigned)grid), dim3(256), 0, s, p, tiles_x, txy);
    LAUNCH_CHECK("conv_stem_halo");
    return 0;
}

#ifdef I2V_EXPERIMENTAL
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
bool conv_stem64_ok(const I2VConvParams& p) {
    return p.halo == 49 && !p.quad && !p.tap_uniform && !p.pointwise && p.K == SW_K && p.Kpad >= 4 * (SW_KS + SW_D) && p.Cd == 64 && p.sh == 2 && p.sw == 2 &&
           p.blk <= 1 && p.blkt <= 1 && p.Ws % 4 == 0 && p.src_nstride % 4 == 0 && p.osh == 1 && p.osw == 1 && p.oh0 == 0 && p.ow0 == 0 && p.Hg == p.Ho && p.Wg == p.Wo &&
           !p.pre_scale && !p.gate && !p.add0 && !p.add1 && !p.mask && !p.gate_scale && (!p.gate_out || p.Wg % 16 == 0);
}
int launch_conv_stem64(const I2VConvParams& p, hipStream_t s) {
    const int tiles_x = (p.Wg + 15) / 16, tiles_y = (p.Hg + 15) / 16, txy = tiles_x * tiles_y;
    const int64_t grid = (int64_t)p.N * txy;
    if (grid <= 0) return 0;
    if (grid > 0x7fffffff) { snprintf(g_be_err, sizeof g_be_err, "conv grid too large"); g_be_has_err = true; return 1; }
    if (p.temporal) hipLaunchKernelGGL(conv_stem64_halo<true>, dim3((unsigned)grid), dim3(256), 0, s, p, tiles_x, txy);
    else hipLaunchKernelGGL(conv_stem64_halo<false>, dim3((unsigned)grid), dim3(256), 0, s, p, tiles_x, txy);
    LAUNCH_CHECK("conv_stem64_halo");
    return 0;
}
#endif

// =============================================================================================
// Image gradient of a NARROW stem on packed-fp32 vector FMAs (round 6, autotuner bit 12)
// =============================================================================================
// SlowFast's fast stem (5 x 7 x 7 / (1, 2, 2), 3 -> 8 channels) is 13 % of an ILAF step in its input gradient alone, at 56 TFLOP/s: the
// class-packed launch fills 12 of a 16-row fragment, 49 of its 64 tap slots carry weights, and a fp32 MFMA runs at the fp32 VECTOR rate
// anyway (64 FLOP/clk/SIMD) -- so, as in the fused fast-pathway block (i2v_fastblock.hip), nothing is lost by leaving the matrix
// pipe, and the empty rows and tap slots can then be skipped.  Quad-row packing (pack_img: K order (channel, frame tap, row tap, column
// tap x 4), one 4 x 4 tap window per (channel, frame tap) PLANE of dz; 12 class rows = 2 x 2 positions x 3 input channels):
//   * a wave takes a tile of 64 columns x PV = 2 rows of the class grid; a lane owns one column, i.e. 2 grid positions = 2 x (2 x 2
//     pixels x 3 channels) = 24 accumulators in 12 register pairs;
//   * per plane it loads the 5 rows x 4 column shifts its positions' windows cover (20 coalesced loads through a buffer descriptor:
//     column shifts outside the plane carry an out-of-range offset per lane, rows outside it are skipped as a whole -- the row of a
//     load is uniform over the wave), the next plane's loads in flight under this plane's FMAs;
//   * and walks the plane's 16 K rows in order: weight row k (12 floats, one s_load_dwordx16; four rows = one row tap per round trip)
//     times the window value, packed FMAs into the six class-row pairs -- except the pairs whose two weights are STRUCTURALLY zero for
//     that tap (a stride-2 7 x 7 kernel: the last row tap belongs to the odd row class only, the last column tap to the odd column
//     class: 77 of 96 pairs per plane remain; pack_img verifies the zeros and sets I2VConvParams::ig_p77).
// Every output element is the k-ordered fmaf chain of the conv_tile launch minus terms that are exact zeros, then `+ add1` (a second
// stem accumulating onto the first) and the store: bit-identical (tests/test_gpu_video.py::test_stem_halo_kernels_are_bit_identical).
// Measured (ILAF on SlowFast, 128 fast frames of 224^2, inside the attack): conv_imggrad_halo 334 us (56.6 TFLOP/s of algorithmic flops) ->
// 307 us (61.4) with two grid rows per wave at five waves per SIMD (96 registers, no spill) -> 288 us (65.5) with the weight rows read
// from a compact copy (64-byte rows instead of the packing's 512: I2VConvParams::wpc); three rows per wave at four waves per SIMD:
// 337 us -- slower (I2V_IGV_PV=3).  PMC (profiles/r6_fastblock_pmc.txt): the vector unit issues 51 % of the time, the waves are parked on
// scalar / vector memory for 41 % of their cycles: four scalar round trips per plane (a row tap's four weight rows each) with 1.25 ready
// waves per SIMD.  The probe build (tools/igv_probe.sh, profiles/r6_igvfma_probe.txt) prices the waits: with every weight row the SAME row
// (always a scalar-cache hit) a launch takes 224 us instead of 305, without operand loads 281, without either 218 -- the 30 KB of
// weight rows (one 64-byte line each: the packing's rows are 512 bytes apart) do not stay in the 16 KB scalar cache while the CU's
// twenty waves drift apart.  Sixteen-wave blocks with a barrier per plane (the CU's waves on the same rows at the same time) were
// slower still (330 us; eight-wave blocks 394); a variant that stages the window once per block in LDS (four rows per wave) ran into
// the same scalar-register ceiling (48 SGPRs of weights in flight) and was not finished.  An autotuner candidate; the tap-uniform packings of the 64-channel stems stay on
// conv_imggrad_halo (66-74 TFLOP/s).
static constexpr bool igv_pair_needed(const int th, const int tw, const int q) {       // does tap (th, tw) of a stride-2 7 x 7 kernel feed class-row pair q?
    for (int cd = 2 * q; cd < 2 * q + 2; ++cd) {
        const int cls = cd / 3, ph = cls / 2, pw = cls % 2;
        if ((ph == 1 || th < 3) && (pw == 1 || tw < 3)) return true;
    }
    return false;
}
typedef float igv_f2 __attribute__((ext_vector_type(2)));
typedef float igv_f8 __attribute__((ext_vector_type(8)));
static constexpr int IGV_CDPAD = 16;    // row stride of the compact weight copy (I2VConvParams::wpc: 12 class rows in 64 bytes -- at the packing's own 512-byte stride the
                                        // 640 rows fall into one scalar-cache set in eight)
template <bool VID, bool P77, int PVT>     // PVT: grid rows per wave -- 2 at five waves per SIMD (96 registers), 3 at four (128); 4 takes 160
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(PVT == 2 ? 5 : 4, PVT == 2 ? 5 : 4)))
conv_igvfma_kernel(const I2VConvParams p, const int tiles_x, const int tiles_xy) {
#if defined(__HIP_DEVICE_COMPILE__)
    constexpr int PV = PVT, NR = PV + 3;
    constexpr unsigned OOB = 0x80000000u;
    typedef const __attribute__((address_space(4))) igv_f8* w8p_t;      // a weight row is 12 floats: one s_load_dwordx8 + one s_load_dwordx4 (a dwordx16 would hold 16 more
    typedef const __attribute__((address_space(4))) f32x4* w4p_t;       // SGPRs per four-row batch, and the allocator then parks rows in VGPR lanes: 400 v_readlane / v_writelane per 616 FMAs)
    const int t = threadIdx.x, lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6);
    // block -> (grid frame, tile): consecutive tiles of a frame on one XCD (their windows overlap in its L2)
    const int nwg = gridDim.x, bid = blockIdx.x;
    const int q8 = nwg >> 3, r8 = nwg & 7, xcd = bid & 7;
    const int lid = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (bid >> 3);
    const int ng = __builtin_amdgcn_readfirstlane(lid / tiles_xy), tl = lid - ng * tiles_xy;
    const int ty = __builtin_amdgcn_readfirstlane(tl / tiles_x), tx = tl - ty * tiles_x;
    const int y0 = ty * (4 * PV) + wave * PV, x = tx * 64 + lane;        // the wave's first grid row, the lane's grid column
    const int TT = VID ? p.ig_tt : 1;
    const I2VKEntry e0 = load_kentry(p.ktab, 0);
    const int dh_lo = e0.dh, dw_lo = e0.dw, dt_lo = VID ? (e0.valid >> 1) : 0;
    const int HWs = p.Hs * p.Ws;
    int clip = ng, ts0 = 0, tgr = 0;
    if (VID) { clip = __builtin_amdgcn_readfirstlane(ng / p.Tg); tgr = ng - clip * p.Tg; ts0 = tgr * p.st; }
    const int sframe0 = VID ? clip * p.Ts + ts0 : ng;                    // source frame of frame tap dt = 0
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)p.src, 0, p.src_span_bytes, 0x00020000);
    // the lane's four column shifts as byte offsets within a plane row (out of range where the shifted column leaves the plane)
    unsigned voffc[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int xx = x + dw_lo + j;
        voffc[j] = (x < p.Wg && xx >= 0 && xx < p.Ws) ? (unsigned)(xx * 4) : OOB;
    }
#ifdef I2V_EXPERIMENTAL          // probe hooks (tools/igv_probe.sh, timing only): I2V_IGV_PROBE bit 0 = every weight row is row 0 (always a scalar-cache hit), bit 1 = no operand loads
    const int probe = p.add0_H;
#else
    constexpr int probe = 0;
#endif
    igv_f2 acc[PV][6];
#pragma unroll
    for (int i = 0; i < PV; ++i)
#pragma unroll
        for (int q = 0; q < 6; ++q) acc[i][q] = igv_f2{0.f, 0.f};
    if (y0 >= p.Hg) return;                                              // (a wave past the grid's last row: nothing to do)
    const int nplanes = p.Cs * TT;
    // The window rows of a plane: [row][column-shift pair].  Two buffers, A and B, alternate; the plane ahead is requested row by row
    // as this plane's rows die (IGV_PLANE).
    igv_f2 xa[NR][2], xb[NR][2];
    // plane pl = (channel co, frame tap tt), walked with two counters: byte offset of its first row in the source view, or -1 when its
    // frame lies outside the clip.  off_cur belongs to the plane whose FMAs run, off_nxt to the one ahead.
    // (the frame part of the offset is tabulated once, lane l = frame tap l: one v_readlane per plane instead of 64-bit scalar arithmetic,
    //  which cost 190 spilled SGPRs in the loop)
    int ttoff_v = -1;
    if (lane < TT) {
        const int tf_ = ts0 + lane + dt_lo;
        if (!VID || (tf_ >= 0 && tf_ < p.Ts)) ttoff_v = (int)((int64_t)(sframe0 + (VID ? lane + dt_lo : 0)) * p.src_nstride * 4);
    }
    const int plane_bytes = HWs * 4;
    int pco = 0, ptt = 0;
    auto plane_off = [&](const int co_, const int tt_) {
        const int fo = __builtin_amdgcn_readlane(ttoff_v, tt_);
        return fo < 0 ? -1 : fo + co_ * plane_bytes;
    };
    int off_cur = plane_off(0, 0), off_nxt = -1;
#define IGV_ADVANCE() { ptt += 1; if (ptt == TT) { ptt = 0; pco += 1; } off_nxt = plane_off(pco, ptt); }
    // rows R0 .. R1 - 1 of a plane into buffer X: a row outside the plane (or a frame outside the clip) is requested with every lane's
    // offset out of range -- no branch, the load returns 0.  The row of a load is uniform over the wave; its kill word sits in a VECTOR
    // register all the same (as 64-bit scalar conditions the seven of them were spilled and re-read around every load).
    unsigned killv[NR];
    const int rowstride = p.Ws * 4;
    const int rowbase = (y0 + dh_lo) * rowstride;                       // (negative for a row above the plane: that row's loads are killed and its scalar offset is clamped)
#pragma unroll
    for (int r = 0; r < NR; ++r) {
        const int yy = y0 + dh_lo + r;
        killv[r] = (yy >= 0 && yy < p.Hs) ? 0u : OOB;
    }
#define IGV_LOAD(OFF_, X, R0, R1)                                                                              \
    {                                                                                                          \
        const int off_ = (OFF_);                                                                               \
        const unsigned pkill = (off_ < 0 || (probe & 2)) ? OOB : 0u;                                           \
        const int so0 = (off_ < 0 ? 0 : off_) + rowbase;                                                       \
        _Pragma("unroll") for (int r = (R0); r < (R1); ++r) {                                                  \
            const int so_ = so0 + r * rowstride < 0 ? 0 : so0 + r * rowstride;                                 \
            _Pragma("unroll") for (int j = 0; j < 4; ++j) {                                                    \
                const float v = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, (int)(voffc[j] | killv[r] | pkill), so_, 0));      \
                if (j & 1) X[r][j >> 1].y = v; else X[r][j >> 1].x = v;                                         \
            }                                                                                                  \
        }                                                                                                      \
        __builtin_amdgcn_sched_barrier(0);                                                                     \
    }
    // which class-row pairs a tap (th, tw) feeds: P77 -- row class ph owns row tap th iff ph == 1 || th < 3, column class pw owns tw
    // iff pw == 1 || tw < 3; class row cd = (2 ph + pw) 3 + ci, pair q = cd / 2.  (A column tap that only pads the quad has zero
    // weights and a finite operand: computed like any other.)
#define IGV_NEED(TH_, TW_, Q) (!P77 || igv_pair_needed(TH_, TW_, Q))
#define IGV_FMA_TH(PL, X, TH_)                                                                                 \
    {                                                                                                          \
        igv_f8 wa[4]; f32x4 wb[4];                                                                             \
        const float* const wth_ = p.wpc + ((probe & 1) ? 0 : (int64_t)((PL) * 4 + (TH_)) * (4 * IGV_CDPAD));     \
        _Pragma("unroll") for (int tw = 0; tw < 4; ++tw) {                                                     \
            const float* const wr_ = wth_ + tw * IGV_CDPAD;          /* (constant offsets from one base: immediates of the scalar loads) */ \
            wa[tw] = *(w8p_t)wr_; wb[tw] = *(w4p_t)(wr_ + 8);                                                  \
        }                                                                                                      \
        __builtin_amdgcn_sched_barrier(0);                                                                     \
        _Pragma("unroll") for (int tw = 0; tw < 4; ++tw)                                                       \
            _Pragma("unroll") for (int i = 0; i < PV; ++i) {                                                   \
                const igv_f2 xp = X[i + (TH_)][tw >> 1];                                                        \
                const igv_f2 xx = (tw & 1) ? igv_f2{xp.y, xp.y} : igv_f2{xp.x, xp.x};                           \
                _Pragma("unroll") for (int q = 0; q < 6; ++q)                                                  \
                    if (IGV_NEED(TH_, tw, q))                                                                  \
                        acc[i][q] = __builtin_elementwise_fma(q < 4 ? igv_f2{wa[tw][(2 * q) & 7], wa[tw][(2 * q + 1) & 7]} : igv_f2{wb[tw][(2 * q) & 3], wb[tw][(2 * q + 1) & 3]}, xx, acc[i][q]);      \
            }                                                                                                  \
        __builtin_amdgcn_sched_barrier(0);                                                                     \
    }
    // plane PL from buffer X; HASNEXT: plane PL + 1 goes to buffer Y under it, row by row as X's rows die -- its first PV rows after row
    // tap 0, then one row behind each further row tap: every row is requested at least two row taps before its first use, and at most
    // PV + 3 + PV rows are alive at a time.  (Requesting a plane's last three rows at the head of its own FMAs left each wave waiting
    // for them after its first row tap.)
#define IGV_PLANE(PL, X, Y, HASNEXT)                                                                           \
    {                                                                                                          \
        IGV_FMA_TH(PL, X, 0)                                                                                   \
        if (HASNEXT) { IGV_ADVANCE() IGV_LOAD(off_nxt, Y, 0, PV) }                                             \
        IGV_FMA_TH(PL, X, 1)                                                                                   \
        if (HASNEXT) IGV_LOAD(off_nxt, Y, PV, PV + 1)                                                          \
        IGV_FMA_TH(PL, X, 2)                                                                                   \
        if (HASNEXT) IGV_LOAD(off_nxt, Y, PV + 1, PV + 2)                                                      \
        IGV_FMA_TH(PL, X, 3)                                                                                   \
        if (HASNEXT) IGV_LOAD(off_nxt, Y, PV + 2, PV + 3)                                                      \
        off_cur = off_nxt;                                                                                     \
    }
    IGV_LOAD(off_cur, xa, 0, NR)
    int pl = 0;
    for (; pl + 2 < nplanes; pl += 2) {
        IGV_PLANE(pl, xa, xb, true)
        IGV_PLANE(pl + 1, xb, xa, true)
    }
    if (pl + 2 == nplanes) { IGV_PLANE(pl, xa, xb, true) IGV_PLANE(pl + 1, xb, xa, false) }
    else { IGV_PLANE(pl, xa, xb, false) }
#undef IGV_PLANE
#undef IGV_FMA_TH
#undef IGV_LOAD
#undef IGV_NEED
#undef IGV_ADVANCE
    // ---- epilogue: class row cd = (2 ph + pw) 3 + ci -> channel ci at pixel (2 gi + ph, 2 gj + pw) of the destination frame; + add1; store ----
    const int otb = VID ? tgr * p.ost + p.ot0 : 0;
    if (VID && otb >= p.To) return;
    const int64_t n = VID ? (int64_t)clip * p.To + otb : ng;
    const int HoWo = p.Ho * p.Wo;
    float* const dst = p.dst + n * p.dst_nstride;
    const float* const ad1 = p.add1 ? p.add1 + n * p.add1_nstride : nullptr;
    if (x >= p.Wg) return;
#pragma unroll
    for (int i = 0; i < PV; ++i) {
        const int gi = y0 + i;
        if (gi >= p.Hg) continue;
#pragma unroll
        for (int ph = 0; ph < 2; ++ph) {
            const int oh = gi * 2 + ph;
            if (oh >= p.Ho) continue;
#pragma unroll
            for (int ci = 0; ci < 3; ++ci) {
                const int cd0 = (2 * ph + 0) * 3 + ci, cd1 = (2 * ph + 1) * 3 + ci;
                float v0 = (cd0 & 1) ? acc[i][cd0 >> 1].y : acc[i][cd0 >> 1].x;
                float v1 = (cd1 & 1) ? acc[i][cd1 >> 1].y : acc[i][cd1 >> 1].x;
                const int64_t o = (int64_t)ci * HoWo + (int64_t)oh * p.Wo + 2 * x;
                const bool two = 2 * x + 1 < p.Wo;
                if (ad1) { v0 += ad1[o]; if (two) v1 += ad1[o + 1]; }
                dst[o] = v0;
                if (two) dst[o + 1] = v1;
            }
        }
    }
#endif
}

// Eligible: the quad-row packing of a narrow stride-2 stem's image gradient (one quad per row run, 4 x 4 union taps), 2 x 2 position classes
// of 3 channels, no temporal classes, a plain epilogue (+ add1)
bool conv_igvfma_ok(const I2VConvParams& p) {
    const int TT = p.ig_tt > 0 ? p.ig_tt : 1;
    return p.quad == 1 && p.ig_th == 4 && p.ig_tw == 4 && p.quad_kw >= 1 && p.quad_kw <= 4 && p.blk == 2 && p.blkt <= 1 && p.Cd == 12 && p.wpc != nullptr &&
           p.sh == 1 && p.sw == 1 && p.osh == 2 && p.osw == 2 && p.oh0 == 0 && p.ow0 == 0 && p.K == p.Cs * TT * 16 && p.Kpad == p.K &&
           !p.shift && !p.relu && !p.mask && !p.gate && !p.gate_out && !p.add0 && !p.pre_scale && !p.gate_scale && p.Hg * 2 >= p.Ho && p.Wg * 2 >= p.Wo;
}
int launch_conv_igvfma(const I2VConvParams& p, hipStream_t s) {
    static const int pv = [] { const char* e = getenv("I2V_IGV_PV"); return e && e[0] == '3' ? 3 : 2; }();      // (developer A/B knob)
#ifdef I2V_EXPERIMENTAL
    I2VConvParams pp = p; { const char* e = getenv("I2V_IGV_PROBE"); pp.add0_H = e ? atoi(e) : 0; }
#define p pp
#endif
    const int tiles_x = (p.Wg + 63) / 64, tiles_y = (p.Hg + 4 * pv - 1) / (4 * pv), txy = tiles_x * tiles_y;
    const int64_t grid = (int64_t)p.N * txy;
    if (grid <= 0) return 0;
    if (grid > 0x7fffffff) { snprintf(g_be_err, sizeof g_be_err, "image-gradient launch (vector FMAs): grid too large"); g_be_has_err = true; return 1; }
#define IGV_GO(V, P7, PV_) hipLaunchKernelGGL((conv_igvfma_kernel<V, P7, PV_>), dim3((unsigned)grid), dim3(256), 0, s, p, tiles_x, txy)
#define IGV_GO2(V, P7) { if (pv == 3) IGV_GO(V, P7, 3); else IGV_GO(V, P7, 2); }
    if (p.temporal) { if (p.ig_p77) IGV_GO2(true, true) else IGV_GO2(true, false) }
    else { if (p.ig_p77) IGV_GO2(false, true) else IGV_GO2(false, false) }
#undef IGV_GO2
#undef IGV_GO
#ifdef I2V_EXPERIMENTAL
#undef p
#endif
    LAUNCH_CHECK("conv_igvfma_kernel");
    return 0;
}

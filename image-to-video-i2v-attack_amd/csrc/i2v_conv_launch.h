// launch_conv_cfg<BD, BP, WD, WP, MF16>: picks the conv_tile instantiation of ONE tile shape for a launch (staging mode, video /
// pre-activation / prefetch variants, and for the 64 x 64 tile the halo, two-chunk and tail-split kernels) and launches it.  Each
// i2v_conv_cfgN.hip instantiates it for one tile shape: that is what makes the build parallel.
#pragma once
#include "i2v_conv_tile.h"

template <int BD, int BP, int WD, int WP, bool MF16 = false>
static int launch_conv_cfg(const I2VConvParams& p, hipStream_t s) {
    const int64_t P = (int64_t)p.N * p.Hg * p.Wg;
    const int n_cd = (p.Cd + BD - 1) / BD;
    const int64_t n_px = (P + BP - 1) / BP;
    const int64_t grid = n_px * n_cd;
    if (grid <= 0) return 0;
    if (grid > 0x7fffffff) { snprintf(g_be_err, sizeof g_be_err, "conv grid too large"); g_be_has_err = true; return 1; }
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

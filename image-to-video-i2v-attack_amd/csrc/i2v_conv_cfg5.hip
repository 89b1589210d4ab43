// conv_igemm on the 16 x 256 on 16x16x4 MFMA fragments tile: autotuner configuration 5 (k_conv, i2v_kernels.hip).
#include "i2v_conv_launch.h"

int launch_conv_cfg5(const I2VConvParams& p, hipStream_t s) { return launch_conv_cfg<16, 256, 1, 4, true>(p, s); }

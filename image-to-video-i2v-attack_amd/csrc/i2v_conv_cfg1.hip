// conv_igemm on the 64 x 128 tile: autotuner configuration 1 (k_conv, i2v_kernels.hip).
#include "i2v_conv_launch.h"

int launch_conv_cfg1(const I2VConvParams& p, hipStream_t s) { return launch_conv_cfg<64, 128, 2, 2>(p, s); }

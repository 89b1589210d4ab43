// conv_igemm on the 128 x 64 tile: autotuner configuration 2 (k_conv, i2v_kernels.hip).
#include "i2v_conv_launch.h"

int launch_conv_cfg2(const I2VConvParams& p, hipStream_t s) { return launch_conv_cfg<128, 64, 2, 2>(p, s); }

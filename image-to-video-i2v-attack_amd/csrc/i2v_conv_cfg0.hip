// conv_igemm on the 128 x 128 (channels x pixels) tile: autotuner configuration 0 (k_conv, i2v_kernels.hip).
#include "i2v_conv_launch.h"

int launch_conv_cfg0(const I2VConvParams& p, hipStream_t s) { return launch_conv_cfg<128, 128, 2, 2>(p, s); }

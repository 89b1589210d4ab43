// conv_igemm on the 32 x 256 tile: autotuner configuration 4 (k_conv, i2v_kernels.hip).
#include "i2v_conv_launch.h"

int launch_conv_cfg4(const I2VConvParams& p, hipStream_t s) { return launch_conv_cfg<32, 256, 1, 4>(p, s); }

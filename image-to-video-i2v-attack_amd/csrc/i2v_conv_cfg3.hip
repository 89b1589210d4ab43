// conv_igemm on the 64 x 64, with its halo / two-chunk / tail-split / prefetch variants tile: autotuner configuration 3 (k_conv, i2v_kernels.hip).
#include "i2v_conv_launch.h"

int launch_conv_cfg3(const I2VConvParams& p, hipStream_t s) { return launch_conv_cfg<64, 64, 2, 2>(p, s); }

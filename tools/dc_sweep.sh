#!/bin/bash
# Round 4 (GPU box): the 64x64 tile's variants -- plain (3), halo (19), tail split (35), two chunks per barrier (67) -- and the two
# rectangular tiles on the ResNet-50 shapes at 32 and 128 frames.  `bash tools/dc_sweep.sh tools/cmb_r4 > gpurun_out/dc_sweep.log`
BIN=${1:-tools/cmb_r4}
for N in ${FRAMES:-32 128}; do
  echo "== $N frames: Cin Cout H k | cfg 3 / 19 halo / 35 tail / 67 dc / 2 (128x64) / 1 (64x128)   [TFLOP/s]"
  while read -r cin cout h k; do
    line="$cin $cout $h $k |"
    for cfg in ${CFGS:-3 19 35 67 2 1}; do
      v=$(I2V_FORCE_CFG=$cfg $BIN $N $cin $cout $h $k 20 2>&1 | tail -1 | awk '{print $1}')
      line="$line $v"
    done
    echo "$line"
  done <<'SH'
256 256 14 3
1024 256 14 1
256 1024 14 1
512 1024 14 1
128 128 28 3
512 128 28 1
128 512 28 1
64 64 56 3
256 64 56 1
64 256 56 1
SH
done

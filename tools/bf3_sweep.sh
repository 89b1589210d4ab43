#!/bin/bash
# Round 5 (GPU box): the split-bf16 K loop (conv_tile BF3; CMB_BF3=1) against the fp32-MFMA loop on the ResNet-50 shapes, per tile
# configuration (0 = 128x128, 1 = 64x128, 2 = 128x64, 3 = 64x64; +64 = two chunks per barrier), TFLOP/s of ALGORITHMIC fp32 flops;
# the first shape of every frame count also prints the numerical distance between the two results.   bash tools/bf3_sweep.sh tools/cmb_bf3
BIN=${1:-tools/cmb_bf3}
for N in ${FRAMES:-128 32}; do
  echo "== $N frames: Cin Cout H k | fp32 cfg 0 1 2 3 | bf16x3 cfg 0 1 2 3 64 65 66 67   [TFLOP/s, algorithmic]"
  first=1
  while read -r cin cout h k; do
    line="$cin $cout $h $k |"
    for cfg in 0 1 2 3; do
      v=$(I2V_FORCE_CFG=$cfg $BIN $N $cin $cout $h $k 10 2>&1 | tail -1 | awk '{print $1}'); line="$line $v"
    done
    line="$line |"
    for cfg in 0 1 2 3 64 65 66 67; do
      v=$(CMB_BF3=1 I2V_FORCE_CFG=$cfg $BIN $N $cin $cout $h $k 10 2>&1 | tail -1 | awk '{print $1}'); line="$line $v"
    done
    echo "$line"
    if [ $first = 1 ]; then CMB_BF3=1 CMB_CHECK=1 I2V_FORCE_CFG=2 $BIN $N $cin $cout $h $k 3 2>&1 | grep "split-bf16"; first=0; fi
  done <<'SH'
256 256 14 3
128 128 28 3
64 64 56 3
1024 256 14 1
256 1024 14 1
512 128 28 1
128 512 28 1
256 64 56 1
64 256 56 1
SH
done

#!/bin/bash
# Developer tool (GPU box): round-3 diagnosis -- TFLOP/s and in-kernel clock of conv_igemm per tile configuration and operand data
for data in rand relu zero; do
  echo "=== data=$data"
  for shape in "128 256 256 14 3" "128 128 128 28 3" "128 64 64 56 3" "128 1024 256 14 1" "128 256 1024 14 1" "1024 256 256 14 3"; do
    for c in 3 35 19 0 2 1; do
      case "$shape" in *" 1") [ $c = 19 ] && continue;; esac
      echo -n "shape=[$shape] cfg=$c : "
      CMB_DATA=$data I2V_FORCE_CFG=$c tools/cmb_clk $shape 10
    done
  done
done
echo "=== un-stamped build, rand / relu"
for data in rand relu; do
for shape in "128 256 256 14 3" "128 1024 256 14 1"; do for c in 3 35 0; do echo -n "data=$data shape=[$shape] cfg=$c : "; CMB_DATA=$data I2V_FORCE_CFG=$c tools/cmb_plain $shape 10; done; done; done

#!/bin/bash
# Developer tool (GPU box): deep-prefetch K loop (tools/cmb_d1) vs two-buffer loop (tools/cmb_d0) on the short-K pointwise shapes with a residual addend
run() { echo "bin=$1 shape=[$2] cfg=$3 epi=${4:-0}"; if [ "${4:-0}" = 1 ]; then CMB_EPI=1 I2V_FORCE_CFG=$3 tools/$1 $2 10; else I2V_FORCE_CFG=$3 tools/$1 $2 10; fi; }
for b in cmb_d0 cmb_d1; do
  run $b "128 64 256 56 1" 3 1
  run $b "128 128 512 28 1" 3 1
  run $b "128 256 1024 14 1" 3 1
  run $b "128 256 64 56 1" 3 1
  run $b "128 512 128 28 1" 3 1
  run $b "32 64 256 56 1" 3 1
  run $b "128 64 256 56 1" 11 1
done

#!/bin/bash
# Developer tool (GPU box): priority experiments (tools/cmb_p0..p3 = microbench probe builds: baseline, progress priority, tail priority, both)
run() { echo "bin=$1 shape=[$2] cfg=$3 epi=${4:-0}"; if [ "${4:-0}" = 1 ]; then CMB_EPI=1 I2V_FORCE_CFG=$3 tools/$1 $2 10; else I2V_FORCE_CFG=$3 tools/$1 $2 10; fi; }
for b in cmb_p0 cmb_p1 cmb_p2 cmb_p3; do
  for c in 3 35; do run $b "128 256 256 14 3" $c; done
  for c in 3 35; do run $b "128 1024 256 14 1" $c; done
  run $b "128 256 1024 14 1" 3
  run $b "128 128 128 28 3" 3
  run $b "128 64 64 56 3" 3
  run $b "128 64 256 56 1" 3 1
  run $b "128 128 512 28 1" 3 1
  run $b "32 256 256 14 3" 3
  run $b "256 256 256 14 3" 3
  run $b "128 256 256 14 3" 0
  run $b "128 256 256 14 3" 2
done

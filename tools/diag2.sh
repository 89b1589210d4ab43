#!/bin/bash
# Developer tool (GPU box): per-block timeline of conv_igemm launches in steady state (tools/cmb_clk = microbench built with -DX_CLOCK)
run() { echo "shape=[$1] cfg=$2 data=${3:-rand} epi=${4:-0}"; if [ "${4:-0}" = 1 ]; then CMB_EPI=1 CMB_DATA=${3:-rand} I2V_FORCE_CFG=$2 tools/cmb_clk $1 10; else CMB_DATA=${3:-rand} I2V_FORCE_CFG=$2 tools/cmb_clk $1 10; fi; }
for c in 3 35 19 0 2; do run "128 256 256 14 3" $c; done
run "128 256 256 14 3" 3 relu
run "128 256 256 14 3" 3 zero
for c in 3 19 1; do run "128 128 128 28 3" $c; done
for c in 3 19 1; do run "128 64 64 56 3" $c; done
for c in 3 35 2; do run "128 1024 256 14 1" $c; done
for c in 3 2 0; do run "128 256 1024 14 1" $c; done
for c in 3 11; do run "128 64 256 56 1" $c rand 1; done
for c in 3 11 2; do run "128 128 512 28 1" $c rand 1; done
for c in 3 19; do run "32 256 256 14 3" $c; done
run "256 256 256 14 3" 3
run "1024 256 256 14 3" 3

#!/bin/bash
# Round 4 (GPU box): the HBM-bound pointwise launches WITH an epilogue addend (CMB_EPI=1): 64x64 + prefetch (3), 64x64 without (11),
# 128x64 (2), 128x64 waves-stacked + prefetch (130), 64x128 (1).  `bash tools/pw_sweep.sh tools/cmb_r4`
BIN=${1:-tools/cmb_r4}
export CMB_EPI=1
for N in ${FRAMES:-128 32}; do
  echo "== $N frames, with addend: Cin Cout H k | cfg 3 / 11 / 2 / 130 / 1   [TFLOP/s]"
  while read -r cin cout h k; do
    line="$cin $cout $h $k |"
    for cfg in ${CFGS:-3 11 2 130 1}; do
      v=$(I2V_FORCE_CFG=$cfg $BIN $N $cin $cout $h $k 20 2>&1 | tail -1 | awk '{print $1}')
      line="$line $v"
    done
    echo "$line"
  done <<'SH'
64 256 56 1
128 512 28 1
256 1024 14 1
64 64 56 1
SH
done

#!/bin/bash
# Round 4 (GPU box): instruction mix of the 64x64 tile's variants on the layer3 3x3 shape -- plain MODE 2 (cfg 3), halo staging (19),
# two chunks per barrier (67) -- at 32 and 128 frames: SALU / VALU / LDS / VMEM instructions per MFMA, MFMA-busy, parked wave cycles.
# `bash tools/pmc_variants.sh tools/cmb_r4 > gpurun_out/pmc_variants.log`
BIN=${1:-tools/cmb_r4}
for c in 3 19 67; do
  CFG=$c SHAPES=$'32 256 256 14 3 4\n128 256 256 14 3 4\n32 1024 256 14 1 4' bash tools/pmc_cmb.sh $BIN 2>/dev/null | sed "s/^/cfg$c /"
done

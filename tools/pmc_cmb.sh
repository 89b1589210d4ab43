#!/bin/bash
# Developer tool (GPU box): PMC comparison of two microbench shapes.
R=$PWD; OUT=$R/gpurun_out/pmc_cmb; rm -rf $OUT; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
i=0
for args in "1024 256 1024 14 1 4" "128 256 256 56 3 4" "1024 256 256 14 3 4"; do
  i=$((i+1))
  I2V_FORCE_CFG=${CFG:-0} rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES -d $OUT/a$i -o p --output-format csv -- $R/tools/cmb $args > $OUT/a$i.log 2>&1
  I2V_FORCE_CFG=${CFG:-0} rocprofv3 --kernel-trace --pmc SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM SQ_ACTIVE_INST_VMEM SQ_INSTS_MFMA SQ_ACTIVE_INST_SCA -d $OUT/b$i -o p --output-format csv -- $R/tools/cmb $args > $OUT/b$i.log 2>&1
  grep "^N=" $OUT/a$i.log
done

#!/bin/bash
# Developer tool (GPU box): PMC comparison of microbench builds on a few shapes: `bash tools/pmc_cmb.sh tools/cmb_a tools/cmb_b`.
# Two counter groups per (binary, shape), each its own run (--pmc never combined with other tracing); prints per-dispatch averages.
R=$PWD; OUT=$R/gpurun_out/pmc_cmb; rm -rf $OUT; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
export CMB_WARM_MS=5
G1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES"
G2="SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM SQ_ACTIVE_INST_VMEM SQ_INSTS_MFMA SQ_ACTIVE_INST_SCA"
i=0
for bin in "$@"; do
while read -r args; do
  i=$((i+1)); tag=$(basename $bin)_$(echo $args | tr ' ' '_')
  I2V_FORCE_CFG=${CFG:-3} rocprofv3 --kernel-trace --pmc $G1 -d $OUT/a_$tag -o p --output-format csv -- $R/$bin $args > $OUT/a_$tag.log 2>&1
  I2V_FORCE_CFG=${CFG:-3} rocprofv3 --kernel-trace --pmc $G2 -d $OUT/b_$tag -o p --output-format csv -- $R/$bin $args > $OUT/b_$tag.log 2>&1
  python3 - "$OUT" "$tag" <<'PY'
import csv, glob, sys, collections
out, tag = sys.argv[1], sys.argv[2]
acc = collections.defaultdict(lambda: [0.0, 0])
for grp in "ab":
    for f in glob.glob(f"{out}/{grp}_{tag}/**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(f)):
            if "conv_igemm" not in row.get("Kernel_Name", ""): continue
            a = acc[row["Counter_Name"]]; a[0] += float(row["Counter_Value"]); a[1] += 1
print(tag, " ".join(f"{k}={v[0] / max(v[1], 1):.3g}" for k, v in sorted(acc.items())))
PY
done <<< "${SHAPES:-$'32 256 256 14 3 4\n32 1024 256 14 1 4\n128 256 256 14 3 4'}"
done

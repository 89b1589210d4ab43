#!/bin/bash
# Developer tool (GPU box): PMC totals of the attention-product kernels (attn_gemm_kernel<1|2|3>) over one ILAF call on the non-local I3D.
# Two counter groups, each its own run (--pmc never combined with other tracing).
R=$PWD; OUT=$R/gpurun_out/pmc_attn; rm -rf $OUT; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
G1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES"
G2="SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM SQ_ACTIVE_INST_VMEM SQ_INSTS_MFMA SQ_LDS_BANK_CONFLICT"
ARGS="$R/bench.py --workload ilaf --white_model i3d_resnet50 --ilaf_clips 4 --streams 1 --steps 1 --warmup 0 --no-cpu-baseline"
rocprofv3 --kernel-trace --pmc $G1 -d $OUT/a -o p --output-format csv -- python3 $ARGS > $OUT/a.log 2>&1
rocprofv3 --kernel-trace --pmc $G2 -d $OUT/b -o p --output-format csv -- python3 $ARGS > $OUT/b.log 2>&1
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(float))
for grp in "ab":
    for f in glob.glob(f"{out}/{grp}/**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(f)):
            k = row.get("Kernel_Name", "")
            if "attn_" not in k and "softmax_rows" not in k: continue
            k = k.split("(")[0].replace("void ", "")
            acc[k][row["Counter_Name"]] += float(row["Counter_Value"]); acc[k]["n_" + grp] += 1
for k, v in sorted(acc.items()):
    print(k, " ".join(f"{c}={x:.4g}" for c, x in sorted(v.items())))
PY

#!/bin/bash
# Round 6 (GPU box): device-side durations and PMC of the fused fast-pathway block kernel and of conv_vfma inside the ILAF bench
# ("if the gate fails, record the PMC and stop").  Kernel trace / stats and each counter group are separate runs
# (--pmc never combined with other tracing); one clip stream so that a kernel's duration is its own.  Per template instantiation:
# launches, mean duration, share of wave cycles parked at s_waitcnt / s_barrier (SQ_WAIT_ANY), issue-stalled (SQ_WAIT_INST_ANY), issuing
# (SQ_ACTIVE_INST_ANY); vector / scalar / memory instruction counts per wave; HBM bytes (2 x FETCH_SIZE + WRITE_SIZE, the guide's gfx950
# correction).     bash tools/pmc_fastblock.sh [extra env assignments for the profiled run, e.g. I2V_FORCE_FASTBLOCK=1]
R=$PWD; OUT=$R/gpurun_out/pmc_fastblock; rm -rf $OUT; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
for kv in "$@"; do export "$kv"; done
CMD="python3 $R/bench.py --workload ilaf --streams 1 --steps 1 --warmup 1 --no-kernel-timing"
rocprofv3 --kernel-trace --stats -d $OUT/stats -o s --output-format csv -- $CMD > $OUT/stats.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES GRBM_GUI_ACTIVE -d $OUT/a -o p --output-format csv -- $CMD > $OUT/a.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VALU -d $OUT/b -o p --output-format csv -- $CMD > $OUT/b.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/f -o p --output-format csv -- $CMD > $OUT/f.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $OUT/w -o p --output-format csv -- $CMD > $OUT/w.log 2>&1
python3 - "$OUT" > $OUT/summary.txt <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
WANT = ("fast_block", "conv_vfma_kernel", "conv_igvfma", "conv_imggrad_halo")
def key(name): return name.split("(")[0].replace("void ", "")
# durations from the stats run (no counters attached)
dur = collections.defaultdict(list)
for f in glob.glob(f"{out}/stats/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if any(w in r["Kernel_Name"] for w in WANT):
            dur[key(r["Kernel_Name"])].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.defaultdict(lambda: collections.Counter())
for d in "abfw":
    for f in glob.glob(f"{out}/{d}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if any(w in r["Kernel_Name"] for w in WANT):
                k = key(r["Kernel_Name"]); acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[k][r["Counter_Name"]] += 1
for k in sorted(set(dur) | set(acc)):
    ds = sorted(dur.get(k, [0])); m = acc[k]; c = cnt[k]
    per = lambda name: m.get(name, 0.0) / max(c.get(name, 0), 1)            # per launch
    wc = per("SQ_WAVE_CYCLES") or 1.0; waves = per("SQ_WAVES") or 1.0
    print(f"{k}: launches {len(ds)}, duration us median {ds[len(ds) // 2] / 1e3:.1f} min {ds[0] / 1e3:.1f} max {ds[-1] / 1e3:.1f}")
    print(f"    waves per launch {waves:.0f}; wave cycles parked {per('SQ_WAIT_ANY') / wc:.3f}, issue-stalled {per('SQ_WAIT_INST_ANY') / wc:.3f}, issuing {per('SQ_ACTIVE_INST_ANY') / wc:.3f}")
    print(f"    per wave: VALU {per('SQ_INSTS_VALU') / waves:.0f}, SALU {per('SQ_INSTS_SALU') / waves:.0f}, SMEM {per('SQ_INSTS_SMEM') / waves:.0f}, LDS {per('SQ_INSTS_LDS') / waves:.0f}, "
          f"VMEM rd {per('SQ_INSTS_VMEM_RD') / waves:.0f}, VMEM wr {per('SQ_INSTS_VMEM_WR') / waves:.0f}")
    print(f"    HBM bytes per launch: 2 x FETCH_SIZE {2 * per('FETCH_SIZE') * 1024 / 1e6:.1f} MB (FETCH_SIZE is in KB), WRITE_SIZE {per('WRITE_SIZE') * 1024 / 1e6:.1f} MB")
PY
cd $R; cat $OUT/summary.txt; grep -i "error\|invalid" $OUT/*.log | head -5
find $OUT -name "*.csv" -size +20M -delete

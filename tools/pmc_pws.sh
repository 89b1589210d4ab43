#!/bin/bash
# Round 5 (GPU box): PMC of the conv_pw_stream probe ("if the gate fails, record the PMC of the probe and stop").
# One counter group, its own run (--pmc never combined with other tracing); per template instantiation (K = 64 / 128 / 256): launches,
# mean duration, matrix-pipe busy fraction (SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x t x 2.4 GHz), as tools/summarise_profile.py), share
# of wave cycles parked at s_waitcnt / s_barrier (SQ_WAIT_ANY) and issue-stalled (SQ_WAIT_INST_ANY).   bash tools/pmc_pws.sh [frames]
R=$PWD; OUT=$R/gpurun_out/pmc_pws; rm -rf $OUT; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES -d $OUT/a -o p --output-format csv -- $R/tools/pws_probe ${1:-128} > $OUT/a.log 2>&1
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
dur = {}
for f in glob.glob(f"{out}/a/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        dur[r["Dispatch_Id"]] = (r["Kernel_Name"], int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter(); t = collections.Counter()
seen = set()
for f in glob.glob(f"{out}/a/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"]
        if "conv_pw_stream" not in name: continue
        key = name.split("(")[0]
        acc[key][r["Counter_Name"]] += float(r["Counter_Value"])
        if r["Dispatch_Id"] not in seen and r["Dispatch_Id"] in dur:
            seen.add(r["Dispatch_Id"]); n[key] += 1; t[key] += dur[r["Dispatch_Id"]][1]
for k in sorted(acc):
    m, secs = acc[k], t[k] * 1e-9
    wc = m.get("SQ_WAVE_CYCLES", 0) or 1
    print(f"{k}: launches {n[k]}, mean {t[k] / max(n[k], 1) / 1e3:.1f} us (all shapes / modes of this K), matrix pipe busy {m.get('SQ_VALU_MFMA_BUSY_CYCLES', 0) / (1024 * secs * 2.4e9):.3f}, "
          f"wave cycles parked {m.get('SQ_WAIT_ANY', 0) / wc:.3f}, issue-stalled {m.get('SQ_WAIT_INST_ANY', 0) / wc:.3f}, issuing {m.get('SQ_ACTIVE_INST_ANY', 0) / wc:.3f}")
PY

#!/bin/bash
# Round 5 (GPU box): everything profiles/r5_* is made from -- rocprofv3 passes over the bench command (summarised on the box: the raw traces
# are too large to travel back), bench lines of every workload AFTER the profile (so that `roofline.traffic` finds the PMC file of this very
# build), per-shape breakdowns in both math modes.   bash tools/run_r5final.sh
O=gpurun_out/r5final; mkdir -p $O/profiles
bash tools/profile.sh r5 ilaf > $O/profile.log 2>&1
python tools/summarise_profile.py r5 > $O/summarise.log 2>&1
python tools/gap_probe.py gpurun_out/prof_r5/stats > profiles/r5_gap_probe.txt 2>&1
cp profiles/r5_* $O/profiles/
find gpurun_out/prof_r5 -name "*kernel_trace.csv" -delete; find gpurun_out/prof_r5 -name "*counter_collection.csv" -delete; find gpurun_out/prof_r5 -name "*agent_info.csv" -delete
python bench.py --steps 10 --warmup 1 --parity-f64 > $O/bench_default.json 2> $O/bench_default.err
python bench.py > $O/bench_default_driver_flags.json 2>/dev/null
python bench.py --clips 1 --steps 10 --warmup 1 --no-cpu-baseline --no-split-bf16 2>/dev/null | tail -1 > $O/bench_single_clip.json
for w in ens aens config2; do python bench.py --workload $w --steps 3 --warmup 1 2>/dev/null | tail -1 > $O/bench_$w.json; done
python bench.py --workload ilaf --steps 3 --warmup 1 2>/dev/null | tail -1 > $O/bench_ilaf_slowfast.json
python bench.py --workload ilaf --white_model i3d_resnet50 --steps 3 --warmup 1 2>/dev/null | tail -1 > $O/bench_ilaf_i3d.json
I2V_MATH=bf16x3 python bench.py --workload ilaf --steps 3 --warmup 1 2>/dev/null | tail -1 > $O/bench_ilaf_slowfast_bf16x3.json
I2V_MATH=bf16x3 python bench.py --workload ilaf --white_model i3d_resnet50 --steps 3 --warmup 1 2>/dev/null | tail -1 > $O/bench_ilaf_i3d_bf16x3.json
I2V_CLIP_LANES=1 I2V_TIMING_DUMP=/tmp/d1 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-split-bf16 > /dev/null 2>&1; python tools/timing_dump_agg.py /tmp/d1 40 > $O/layer_breakdown.txt
I2V_MATH=bf16x3 I2V_CLIP_LANES=1 I2V_TIMING_DUMP=/tmp/d5 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-split-bf16 > /dev/null 2>&1; python tools/timing_dump_agg.py /tmp/d5 40 > $O/layer_breakdown_bf16x3.txt
du -sh gpurun_out

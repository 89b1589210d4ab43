#!/bin/bash
# Developer tool (GPU box): everything profiles/r4_* is made from -- GPU suite, rocprofv3 passes (summarised on the box: the raw traces
# are too large to travel back), bench lines of every workload (after the profile, so that `roofline.traffic` finds the PMC file of
# this very build), per-shape breakdowns, the inter-kernel gap probe.  `bash tools/run_r4final.sh [notests]`
O=gpurun_out/r4final; mkdir -p $O/profiles
if [ "$1" != "notests" ]; then python -m pytest tests -m gpu -q 2>&1 | tail -12 > $O/tests.log; fi
bash tools/profile.sh r4 ilaf > $O/profile.log 2>&1
python tools/summarise_profile.py r4 > $O/summarise.log 2>&1
python tools/gap_probe.py gpurun_out/prof_r4/stats > $O/profiles/r4_gap_probe.txt 2>&1
cp profiles/r4_* $O/profiles/
find gpurun_out/prof_r4 -name "*kernel_trace.csv" -delete; find gpurun_out/prof_r4 -name "*counter_collection.csv" -delete; find gpurun_out/prof_r4 -name "*agent_info.csv" -delete
python bench.py --steps 10 --warmup 1 --parity-f64 > $O/bench_default.json 2> $O/bench_default.err
python bench.py --clips 1 --steps 10 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_single_clip.json
for w in ens aens config2; do python bench.py --workload $w --steps 3 --warmup 1 $( [ $w = config2 ] && echo --clips 8 ) 2>/dev/null | tail -1 > $O/bench_$w.json; done
python bench.py --workload ilaf --steps 3 --warmup 1 2>/dev/null | tail -1 > $O/bench_ilaf_slowfast.json
python bench.py --workload ilaf --white_model i3d_resnet50 --steps 3 --warmup 1 2>/dev/null | tail -1 > $O/bench_ilaf_i3d.json
I2V_CLIP_LANES=1 I2V_TIMING_DUMP=/tmp/d1 python bench.py --steps 2 --warmup 1 --no-cpu-baseline > /dev/null 2>&1; python tools/timing_dump_agg.py /tmp/d1 40 > $O/layer_breakdown.txt
I2V_CLIP_LANES=1 I2V_TIMING_DUMP=/tmp/d4 python bench.py --clips 1 --steps 2 --warmup 1 --no-cpu-baseline > /dev/null 2>&1; python tools/timing_dump_agg.py /tmp/d4 40 > $O/layer_breakdown_single_clip.txt
I2V_TIMING_DUMP=/tmp/d2 python bench.py --workload ilaf --clips 4 --streams 1 --steps 1 --warmup 1 > /dev/null 2>&1; python tools/timing_dump_agg.py /tmp/d2 30 > $O/ilaf_breakdown_slowfast.txt
I2V_TIMING_DUMP=/tmp/d3 python bench.py --workload ilaf --white_model i3d_resnet50 --clips 4 --streams 1 --steps 1 --warmup 1 > /dev/null 2>&1; python tools/timing_dump_agg.py /tmp/d3 30 > $O/ilaf_breakdown_i3d.txt
du -sh gpurun_out

#!/bin/bash
# Round 4 (GPU box): the rocprofv3 passes + summaries + bench lines only (after a source change that moves the build id)
O=gpurun_out/r4final; mkdir -p $O/profiles
bash tools/profile.sh r4 > $O/profile.log 2>&1
python tools/summarise_profile.py r4 > $O/summarise.log 2>&1
python tools/gap_probe.py gpurun_out/prof_r4/stats > profiles/r4_gap_probe.txt 2>&1
cp profiles/r4_* $O/profiles/
find gpurun_out/prof_r4 -name "*kernel_trace.csv" -delete; find gpurun_out/prof_r4 -name "*counter_collection.csv" -delete; find gpurun_out/prof_r4 -name "*agent_info.csv" -delete
python bench.py --steps 10 --warmup 1 --parity-f64 > $O/bench_default.json 2> $O/bench_default.err
python bench.py > $O/bench_default_driver_flags.json 2>/dev/null
for w in ens aens config2; do python bench.py --workload $w --steps 3 --warmup 1 $( [ $w = config2 ] && echo --clips 8 ) 2>/dev/null | tail -1 > $O/bench_$w.json; done
I2V_CLIP_LANES=1 I2V_TIMING_DUMP=/tmp/d1 python bench.py --steps 2 --warmup 1 --no-cpu-baseline > /dev/null 2>&1; python tools/timing_dump_agg.py /tmp/d1 40 > $O/layer_breakdown.txt

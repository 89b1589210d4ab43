#!/bin/bash
# Round 6 (GPU box): everything profiles/r6_* is made from -- rocprofv3 passes over the bench command (summarised on the box: the raw traces
# are too large to travel back), then ONE bench line per BASELINE.json config at its own size, AFTER the profile (so that `roofline.traffic`
# finds the PMC file of this very build).  BUILDER-RUN lines: the driver's own is BENCH_r06.json.     bash tools/run_r6final.sh
O=gpurun_out/r6final; mkdir -p $O/profiles
bash tools/profile.sh r6 ilaf > $O/profile.log 2>&1
python tools/summarise_profile.py r6 > $O/summarise.log 2>&1
python tools/gap_probe.py gpurun_out/prof_r6/stats > profiles/r6_gap_probe.txt 2>&1
cp profiles/r6_* $O/profiles/
find gpurun_out/prof_r6 -name "*kernel_trace.csv" -delete; find gpurun_out/prof_r6 -name "*counter_collection.csv" -delete; find gpurun_out/prof_r6 -name "*agent_info.csv" -delete
# configs[1] (the headline), twice: with longer timing + the live float64 yardstick + a generous budget for the same-GPU framework baseline
# (MIOpen on a fresh box), and exactly as the driver runs it
python bench.py --steps 10 --warmup 1 --parity-f64 --framework-budget 900 > $O/bench_default.json 2> $O/bench_default.err
python bench.py > $O/bench_default_driver_flags.json 2>/dev/null
python bench.py --clips 1 --steps 10 --warmup 1 --no-cpu-baseline --no-split-bf16 2>/dev/null | tail -1 > $O/bench_single_clip.json
# configs[2] at batch 8, configs[3] at this GPU's share (8 clips), the reference CLI's ENS list as an extra
for w in config2 aens ens; do python bench.py --workload $w --steps 3 --warmup 1 2>/dev/null | tail -1 > $O/bench_$w.json; done
# configs[4] with both feature guides
I2V_FUSE_DEBUG=1 python bench.py --workload ilaf --steps 3 --warmup 1 2> $O/ilaf_slowfast.err | tail -1 > $O/bench_ilaf_slowfast.json
grep "i2v fastblock" $O/ilaf_slowfast.err | sort | uniq > $O/profiles/r6_fastblock_autotune.txt
python bench.py --workload ilaf --white_model i3d_resnet50 --steps 3 --warmup 1 2>/dev/null | tail -1 > $O/bench_ilaf_i3d.json
I2V_FASTBLOCK=0 python bench.py --workload ilaf --steps 3 --warmup 1 2>/dev/null | tail -1 > $O/bench_ilaf_slowfast_no_fastblock.json
for st in 1 2 4; do python bench.py --workload ilaf --streams $st --steps 3 --warmup 1 2>/dev/null | tail -1 > $O/bench_ilaf_slowfast_streams$st.json; done
I2V_CLIP_LANES=1 I2V_TIMING_DUMP=/tmp/d1 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-split-bf16 --no-framework-baseline > /dev/null 2>&1; python tools/timing_dump_agg.py /tmp/d1 40 > $O/layer_breakdown.txt
I2V_TIMING_DUMP=/tmp/d2 python bench.py --workload ilaf --streams 1 --steps 1 --warmup 1 > /dev/null 2>&1; python tools/timing_dump_agg.py /tmp/d2 40 > $O/ilaf_breakdown_slowfast.txt
du -sh gpurun_out

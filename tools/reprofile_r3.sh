O=gpurun_out/r3final; mkdir -p $O/profiles
bash tools/profile.sh r3 ilaf > $O/profile.log 2>&1
python tools/summarise_profile.py r3 > $O/summarise.log 2>&1
cp profiles/r3_* $O/profiles/
find gpurun_out/prof_r3 -name "*kernel_trace.csv" -delete; find gpurun_out/prof_r3 -name "*counter_collection.csv" -delete; find gpurun_out/prof_r3 -name "*agent_info.csv" -delete
python bench.py --steps 10 --warmup 1 > $O/bench_default.json 2> $O/bench_default.err
python -m pytest tests/test_gpu_video.py -m gpu -q -k "full_size_against_oracle or independent_clips" 2>&1 | tail -3 > $O/tests_extra.log
cat $O/tests_extra.log

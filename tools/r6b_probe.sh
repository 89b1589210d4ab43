O=gpurun_out/r6b; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_video.py -m gpu -x -q -k "fast or vfma" > $O/tests_fb.log 2>&1; echo "rc $?" >> $O/tests_fb.log
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "launch_overlap or clip_lanes_full" > $O/tests_ov.log 2>&1; echo "rc $?" >> $O/tests_ov.log
timeout 600 python tools/overlap_probe.py > $O/overlap_probe.txt 2>&1
I2V_FUSE_DEBUG=1 timeout 600 python bench.py --workload ilaf --streams 1 --steps 2 --warmup 1 2> $O/ilaf_v2.err | tail -1 > $O/bench_ilaf_v2_s1.json
grep "i2v fastblock\|i2v vfma" $O/ilaf_v2.err | sort | uniq > $O/fastblock_autotune_v2.txt
I2V_FB_V1=1 I2V_FUSE_DEBUG=1 timeout 600 python bench.py --workload ilaf --streams 1 --steps 2 --warmup 1 2> $O/ilaf_v1.err | tail -1 > $O/bench_ilaf_v1_s1.json
grep "i2v fastblock\|i2v vfma" $O/ilaf_v1.err | sort | uniq > $O/fastblock_autotune_v1.txt
timeout 600 python bench.py --workload ilaf --steps 3 --warmup 1 2>/dev/null | tail -1 > $O/bench_ilaf_v2.json
bash tools/pmc_fastblock.sh > $O/pmc.log 2>&1; cp gpurun_out/pmc_fastblock/summary.txt $O/pmc_summary_v2.txt

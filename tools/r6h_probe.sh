O=gpurun_out/r6h; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_video.py -m gpu -x -q -k "fast or vfma or slowfast" > $O/tests_fb.log 2>&1; echo "rc $?" >> $O/tests_fb.log
python tools/fb_phase_probe.py > $O/phase_probe.txt 2>&1
I2V_FUSE_DEBUG=1 timeout 600 python bench.py --workload ilaf --streams 1 --steps 2 --warmup 1 2> $O/ilaf_s1.err | tail -1 > $O/bench_ilaf_s1.json
grep "i2v fastblock" $O/ilaf_s1.err | grep -v "launch [0-9]*:" | sort | uniq > $O/fastblock_autotune.txt
timeout 600 python bench.py --workload ilaf --steps 3 --warmup 1 2>/dev/null | tail -1 > $O/bench_ilaf.json

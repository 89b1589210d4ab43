O=gpurun_out/r6a; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_video.py -m gpu -x -q -k "fast or vfma or stem_halo or slowfast" > $O/tests.log 2>&1; echo "tests rc $?" >> $O/tests.log
I2V_FUSE_DEBUG=1 timeout 600 python bench.py --workload ilaf --steps 3 --warmup 1 2> $O/ilaf_slowfast.err | tail -1 > $O/bench_ilaf_slowfast.json
grep "i2v fastblock\|i2v vfma" $O/ilaf_slowfast.err | sort | uniq > $O/fastblock_autotune.txt
I2V_FASTBLOCK=0 I2V_VFMA=0 timeout 600 python bench.py --workload ilaf --steps 3 --warmup 1 2>/dev/null | tail -1 > $O/bench_ilaf_slowfast_nofb.json
I2V_TIMING_DUMP=/tmp/d2 timeout 600 python bench.py --workload ilaf --streams 1 --steps 1 --warmup 1 > /dev/null 2>&1; python tools/timing_dump_agg.py /tmp/d2 60 > $O/ilaf_breakdown_slowfast.txt
I2V_FASTBLOCK=0 I2V_VFMA=0 I2V_TIMING_DUMP=/tmp/d3 timeout 600 python bench.py --workload ilaf --streams 1 --steps 1 --warmup 1 > /dev/null 2>&1; python tools/timing_dump_agg.py /tmp/d3 60 > $O/ilaf_breakdown_slowfast_nofb.txt

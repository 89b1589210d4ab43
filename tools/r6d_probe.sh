O=gpurun_out/r6d; mkdir -p $O
I2V_TIMING_DUMP=/tmp/d2 timeout 600 python bench.py --workload ilaf --streams 1 --steps 1 --warmup 1 > /dev/null 2>&1; python tools/timing_dump_agg.py /tmp/d2 60 > $O/ilaf_breakdown_slowfast.txt
I2V_TIMING_DUMP=/tmp/d3 timeout 600 python bench.py --workload ilaf --white_model i3d_resnet50 --streams 1 --steps 1 --warmup 1 > /dev/null 2>&1; python tools/timing_dump_agg.py /tmp/d3 40 > $O/ilaf_breakdown_i3d.txt
timeout 1500 python -m pytest tests -m gpu -x -q > $O/gpu_suite.log 2>&1; echo "rc $?" >> $O/gpu_suite.log

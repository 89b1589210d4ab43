#!/bin/bash
I2V_CLIP_LANES=1 I2V_TIMING_DUMP=/tmp/d1 python bench.py --steps 2 --warmup 1 --no-cpu-baseline > /dev/null 2>&1; python tools/timing_dump_agg.py /tmp/d1 > gpurun_out/r3b_layer_breakdown.txt
I2V_TIMING_DUMP=/tmp/d2 python bench.py --workload ilaf --ilaf_clips 8 --streams 1 --steps 1 --warmup 1 > /dev/null 2>&1; python tools/timing_dump_agg.py /tmp/d2 > gpurun_out/r3b_ilaf_sf8.txt
I2V_TIMING_DUMP=/tmp/d3 python bench.py --workload ilaf --white_model i3d_resnet50 --ilaf_clips 8 --streams 1 --steps 1 --warmup 1 > /dev/null 2>&1; python tools/timing_dump_agg.py /tmp/d3 > gpurun_out/r3b_ilaf_i3d8.txt
python -m pytest tests/test_gpu_zconfigs.py -x -q -s -m gpu -k config2 2>&1 | grep -E "passed|failed|mid-trajectory|float32|Error|assert" | tail -20 > gpurun_out/r3b_tests.log

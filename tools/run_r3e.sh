#!/bin/bash
python -m pytest tests -m gpu -x -q 2>&1 | tail -8 > gpurun_out/r3e_tests.log
python bench.py --steps 10 --warmup 1 > gpurun_out/r3e_bench.json 2> gpurun_out/r3e_bench.err
python bench.py --workload ilaf --steps 3 --warmup 1 > gpurun_out/r3e_ilaf_slowfast.json 2>/dev/null
python bench.py --workload ilaf --white_model i3d_resnet50 --steps 3 --warmup 1 > gpurun_out/r3e_ilaf_i3d.json 2>/dev/null
python bench.py --workload ilaf --white_model i3d_plain_resnet50 --steps 3 --warmup 1 > gpurun_out/r3e_ilaf_i3dplain.json 2>/dev/null

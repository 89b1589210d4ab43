#!/bin/bash
python -m pytest tests -m gpu -x -q 2>&1 | tail -15 > gpurun_out/r3d_tests.log
python bench.py --workload ilaf --white_model i3d_resnet50 --steps 2 --warmup 1 --no-kernel-timing 2>/dev/null | tail -1 | cut -c1-400 > gpurun_out/r3d_ilaf_i3d.json

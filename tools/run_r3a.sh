#!/bin/bash
# Developer tool (GPU box): GPU suite + headline bench + ILAF clip/stream combinations
python -m pytest tests -m gpu -x -q -s 2>&1 | grep -E "passed|failed|error|mid-trajectory|Error" | tail -30 > gpurun_out/r3a_tests.log
python bench.py --steps 10 --warmup 1 > gpurun_out/r3a_bench.json 2> gpurun_out/r3a_bench.err
for combo in "1 3" "4 1" "4 2" "8 1" "2 2"; do set -- $combo
  python bench.py --workload ilaf --ilaf_clips $1 --streams $2 --steps 3 --warmup 1 --no-kernel-timing 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('ilaf slowfast clips=$1 streams=$2', d['value'], d['ms_per_step'])" >> gpurun_out/r3a_ilaf.log
  python bench.py --workload ilaf --white_model i3d_resnet50 --ilaf_clips $1 --streams $2 --steps 3 --warmup 1 --no-kernel-timing 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('ilaf i3d clips=$1 streams=$2', d['value'], d['ms_per_step'])" >> gpurun_out/r3a_ilaf.log
done

#!/bin/bash
python -m pytest "tests/test_gpu_video.py::test_video_backbone_tiny" -x -q -m gpu 2>&1 | grep -E "^E  |passed|failed|Error" | head -12 > gpurun_out/r3f_tests.log
for rep in 1 2; do for v in default noprio; do
  if [ $v = noprio ]; then export I2V_LIB=$PWD/tools/_libi2v_noprio.so; else unset I2V_LIB; fi
  python bench.py --steps 6 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', d['value'], d['roofline']['achieved'], d['product_default']['value'], d['single_clip']['value'])" >> gpurun_out/r3f_ab.log
done; done

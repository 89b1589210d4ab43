#!/bin/bash
# Round 4 (GPU box): the headline bench on library BUILDS (compile-time knobs) alternated on one box: `bash tools/ab_libs.sh tools/_lib_A.so tools/_lib_B.so`
# ("" = the product library).  Prints value / roofline frac / single clip per run.
for round in 1 2; do
  for lib in "" "$@"; do
    I2V_LIB=$lib python bench.py --no-cpu-baseline --steps 5 --warmup 1 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('${lib:-product}', d['value'], d['roofline']['frac'], d['product_default']['value'], d['single_clip']['value'])"
  done
done

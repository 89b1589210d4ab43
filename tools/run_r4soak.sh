#!/bin/bash
# Developer tool (GPU box): soak runs over the round-4 kernels (two chunks per barrier, the fused pair, the patch-form pool gradient)
# and the standing ones.  `bash tools/run_r4soak.sh [seconds per soak]`
T=${1:-150}; O=gpurun_out/r4soak; mkdir -p $O
python tools/soak_tail.py $T 41 dc > $O/dc.log 2>&1; tail -1 $O/dc.log
python tools/soak_tail.py $T 42 fuse > $O/fuse.log 2>&1; tail -1 $O/fuse.log
python tools/soak_tail.py $T 43 fusehalo > $O/fusehalo.log 2>&1; tail -1 $O/fusehalo.log
python tools/soak_tail.py $T 48 nt > $O/nt.log 2>&1; tail -1 $O/nt.log
python tools/soak_tail.py $T 44 tail > $O/tail.log 2>&1; tail -1 $O/tail.log
python tools/soak_tail.py $T 45 halo > $O/halo.log 2>&1; tail -1 $O/halo.log
python tools/soak_pool.py $T 46 > $O/pool.log 2>&1; tail -2 $O/pool.log
python tools/soak.py $T 47 > $O/conv.log 2>&1; tail -2 $O/conv.log

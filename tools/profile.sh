#!/bin/bash
# Run ON THE GPU BOX (gpurun -- 'bash tools/profile.sh r2'): rocprofv3 passes over the bench command.
# Kernel-trace/stats and each PMC group are separate runs (never combine --pmc with sys/hip traces); the program
# itself follows `--` (no env/bash hop).  The sources' build id (bench.build_id) is stored next to the results so that
# bench.py only quotes traffic measured on the build it is running.
TAG=${1:-r6}
R=$PWD
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
python3 -c "import bench; print(bench.build_id())" > $OUT/build_id.txt
cd /tmp && export TMPDIR=/tmp
export I2V_CLIP_LANES=1     # one clip lane: a kernel's duration is then its own (bench.py's timed region does the same)
export I2V_FUSE=0           # the plan-time autotuner runs INSIDE the profiled process, where every dispatch carries the tool's overhead: one
                            # fused launch then "beats" two that win in a plain run.  The plain bench fuses nothing at 128 frames
                            # (`fused_pairs` in its JSON line), so the profiled plan is made the same.
CMD="python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-timing"
rocprofv3 --kernel-trace --stats -d $OUT/stats -o s --output-format csv -- $CMD > $OUT/stats.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/fetch -o f --output-format csv -- $CMD > $OUT/fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $OUT/write -o w --output-format csv -- $CMD > $OUT/write.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE -d $OUT/mfma -o m --output-format csv -- $CMD > $OUT/mfma.log 2>&1
if [ "$2" = "ilaf" ]; then   # second workload: ILAF on the SlowFast graph (BASELINE.json configs[4]) -- kernel stats only
rocprofv3 --kernel-trace --stats -d $OUT/ilaf -o i --output-format csv -- python3 $R/bench.py --workload ilaf --steps 2 --warmup 1 --no-kernel-timing > $OUT/ilaf.log 2>&1
fi
cd $R
tail -1 $OUT/stats.log | cut -c1-200
ls $OUT/*/

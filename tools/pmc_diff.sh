#!/bin/bash
# usage: tools/pmc_diff.sh <outdir> : PMC counters for the substep kernel at profiling stops 2 and 3
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/$1; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
ARGS="1024 100 PHYSICS.SLEEP_STEPS=0 MIN_MOVABLE_BODIES=1 MAX_MOVABLE_BODIES=1 PHYSICS.NARROWPHASE_MAX_AGE=0"
for st in 2 3; do
  RV_DEBUG_STOP=$st rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_FLAT -d $OUT/a$st -o a --output-format csv -- python $R/tools/prof_sub.py $ARGS > $OUT/a$st.log 2>&1
  RV_DEBUG_STOP=$st rocprofv3 --kernel-trace --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_BRANCH SQ_IFETCH SQ_INSTS_SMEM -d $OUT/b$st -o b --output-format csv -- python $R/tools/prof_sub.py $ARGS > $OUT/b$st.log 2>&1
done
ls $OUT

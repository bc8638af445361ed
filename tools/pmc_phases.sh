#!/bin/bash
# usage (GPU box): tools/pmc_phases.sh <outdir> "<stops>" [prof_sub args...]
# SQ instruction / cycle counters of the substep kernel cut off at each RV_DEBUG_STOP.
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/$1; STOPS=$2; shift; shift; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
ARGS=${@:-"1024 200 link"}
for st in $STOPS; do
  RV_DEBUG_STOP=$st timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_FLAT -d $OUT/a$st -o a --output-format csv -- python $R/tools/prof_sub.py $ARGS > $OUT/a$st.log 2>&1
  RV_DEBUG_STOP=$st timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_BRANCH SQ_IFETCH SQ_INSTS_SMEM -d $OUT/b$st -o b --output-format csv -- python $R/tools/prof_sub.py $ARGS > $OUT/b$st.log 2>&1
done
python $R/tools/pmc_table.py $OUT $STOPS | tee $OUT/table.txt

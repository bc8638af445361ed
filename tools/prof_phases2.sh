#!/bin/bash
# usage (on the GPU box): tools/prof_phases2.sh <outname> [prof_sub args...]
# Substep kernel cut off after each phase group (RV_DEBUG_STOP), heavy phases included;
# per-phase costs are the differences between consecutive lines (last of 3 runs printed).
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/$1; shift; mkdir -p $(dirname $OUT)
ARGS=${@:-"1024 200 link"}
: > $OUT.txt
for st in 12 1 16 17 18 2 3 4 5 0; do
  echo -n "stop $st: " >> $OUT.txt
  RV_DEBUG_STOP=$st timeout 300 python $R/tools/prof_sub.py $ARGS 2>&1 | grep "^sub" | tail -1 >> $OUT.txt
done
cat $OUT.txt

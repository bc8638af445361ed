#!/bin/bash
# usage (on the GPU box, via gpurun): bash tools/profile_bench.sh <tag> [bench.py arguments]
# kernel-trace/stats pass + separate PMC passes (FETCH_SIZE, WRITE_SIZE, two SQ sets) of the headline bench command
R=$GRAFT_REPO_ROOT; TAG=$1; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
# optional further arguments go to bench.py (e.g. --workload config3 --steps 10)
shift; EXTRA="$@"; [ -z "$EXTRA" ] && EXTRA="--steps 20 --warmup 5"
CMD="python $R/bench.py $EXTRA --no-cpu-baseline --no-extra-legs"   # one k_env<MODE_ROLLOUT> launch is timed
echo "$CMD" > $OUT/command.txt
rocprofv3 --kernel-trace --stats -d $OUT/trace -- $CMD > $OUT/trace.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/fetch -o f --output-format csv -- $CMD > $OUT/fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $OUT/write -o w --output-format csv -- $CMD > $OUT/write.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_FLAT -d $OUT/sqa -o a --output-format csv -- $CMD > $OUT/sqa.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_BRANCH SQ_IFETCH SQ_INSTS_SMEM -d $OUT/sqb -o b --output-format csv -- $CMD > $OUT/sqb.log 2>&1
$CMD > $OUT/bench.json 2> $OUT/bench.err
tail -1 $OUT/bench.json | cut -c1-200; ls $OUT

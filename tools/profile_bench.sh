#!/bin/bash
# usage (on the GPU box, via gpurun): bash tools/profile_bench.sh <tag>
# kernel-trace/stats pass + two separate PMC passes (FETCH_SIZE, WRITE_SIZE) of the default bench command
R=$GRAFT_REPO_ROOT; TAG=$1; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
CMD="python $R/bench.py --no-cpu-baseline --no-async"   # default K / W; one k_env<MODE_ROLLOUT> launch is timed
rocprofv3 --kernel-trace --stats -d $OUT/trace -- $CMD > $OUT/trace.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/fetch -o f --output-format csv -- $CMD > $OUT/fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $OUT/write -o w --output-format csv -- $CMD > $OUT/write.log 2>&1
python $R/bench.py > $OUT/bench.json 2> $OUT/bench.err
tail -1 $OUT/trace.log | cut -c1-300; ls $OUT

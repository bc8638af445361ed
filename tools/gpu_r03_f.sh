#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03_f; mkdir -p $O; cd $R
timeout 900 python -m pytest tests -m gpu -x -q > $O/tests.log 2>&1; echo "tests rc=$?" >> $O/tests.log
timeout 900 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench.json 2> $O/bench.err
tail -12 $O/tests.log; tail -5 $O/bench.err

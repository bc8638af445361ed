#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03_c; mkdir -p $O; cd $R
timeout 900 python -m pytest tests -m gpu -x -q > $O/tests.log 2>&1; echo "tests rc=$?" >> $O/tests.log
timeout 300 python tools/prof_rollout.py --warm 1 --warm-steps 5 --top 10 > $O/parts_bench_launch.txt 2>&1
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench.json 2> $O/bench.err
tail -4 $O/tests.log; sed -n 2,5p $O/parts_bench_launch.txt; sed -n 46,60p $O/parts_bench_launch.txt; cut -c1-300 $O/bench.json

#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03_e; mkdir -p $O; cd $R
timeout 900 python -m pytest tests -m gpu -x -q > $O/tests.log 2>&1; echo "tests rc=$?" >> $O/tests.log
timeout 300 python tools/prof_rollout.py --warm 1 --warm-steps 5 --top 8 > $O/parts_bench_launch.txt 2>&1
timeout 600 python tools/prof_rollout.py --warm 0 --envs 4096 --steps 10 --top 6 --over TASK_NAME=crossing LAYOUT_ID=0 MOVABLE_NAME=CONCAVE MAX_STEPS=10 > $O/parts_config3.txt 2>&1
timeout 900 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench.json 2> $O/bench.err
tail -3 $O/tests.log; sed -n 2,5p $O/parts_bench_launch.txt; sed -n 43,52p $O/parts_bench_launch.txt; sed -n 2,5p $O/parts_config3.txt;  sed -n 43,50p $O/parts_config3.txt

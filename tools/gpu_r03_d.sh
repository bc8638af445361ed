#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03_d; mkdir -p $O; cd $R
bash tools/profile_bench.sh r03_d > $O/profile_bench.log 2>&1
timeout 300 python tools/prof_rollout.py --warm 1 --warm-steps 5 --top 10 > $O/parts_bench_launch.txt 2>&1
timeout 300 python tools/prof_rollout.py --warm 1 --envs 8192 --steps 10 --top 6 > $O/parts_8192.txt 2>&1
timeout 600 python tools/prof_rollout.py --warm 0 --envs 4096 --steps 10 --top 6 --over TASK_NAME=crossing LAYOUT_ID=0 MOVABLE_NAME=CONCAVE MAX_STEPS=10 > $O/parts_config3.txt 2>&1
tail -3 $O/profile_bench.log; head -8 $O/parts_config3.txt

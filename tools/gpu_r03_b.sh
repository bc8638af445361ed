#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03_b; mkdir -p $O; cd $R
timeout 300 python tools/prof_rollout.py --warm 1 --warm-steps 5 --top 10 > $O/parts_bench_launch.txt 2>&1
timeout 300 python tools/prof_rollout.py --warm 1 --top 10 > $O/parts_launch1.txt 2>&1
timeout 300 python tools/prof_rollout.py --warm 1 --envs 8192 --steps 10 --top 10 > $O/parts_8192.txt 2>&1
cat $O/parts_bench_launch.txt

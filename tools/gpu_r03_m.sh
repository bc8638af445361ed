#!/bin/bash
# end-of-round profile pass: rocprofv3 kernel stats + PMC for the headline and for configs 3, 4, 5; per-part tables
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03_m; mkdir -p $O; cd $R
bash tools/profile_bench.sh r03_m > $O/profile_bench.log 2>&1
bash tools/profile_bench.sh r03_m_c3 --workload config3 --steps 10 --warmup 2 > $O/profile_c3.log 2>&1
bash tools/profile_bench.sh r03_m_c4 --workload config4 --steps 10 --warmup 2 > $O/profile_c4.log 2>&1
bash tools/profile_bench.sh r03_m_c5 --workload config5 --steps 10 --warmup 2 > $O/profile_c5.log 2>&1
cd $R
timeout 300 python tools/prof_rollout.py --warm 1 --warm-steps 5 --top 10 > $O/parts_bench_launch.txt 2>&1
timeout 300 python tools/prof_rollout.py --warm 1 --envs 8192 --steps 10 --top 6 > $O/parts_config5_8192.txt 2>&1
timeout 600 python tools/prof_rollout.py --warm 0 --envs 4096 --steps 10 --top 6 --over TASK_NAME=crossing LAYOUT_ID=0 MOVABLE_NAME=CONCAVE MAX_STEPS=10 > $O/parts_config3_4096.txt 2>&1
timeout 600 python tools/prof_rollout.py --warm 0 --envs 2048 --steps 10 --top 6 --grasp > $O/parts_config4_grasp_2048.txt 2>&1
tail -2 $O/profile_bench.log $O/profile_c3.log $O/profile_c4.log $O/profile_c5.log; head -6 $O/parts_config4_grasp_2048.txt

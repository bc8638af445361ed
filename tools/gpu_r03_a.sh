#!/bin/bash
# round 3, call a: baseline of the tree + per-part profiles (several seeds, no-deactivation) + full bench
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03_a; mkdir -p $O; cd $R
timeout 900 python -m pytest tests -m gpu -x -q > $O/tests.log 2>&1; echo "tests rc=$?" >> $O/tests.log
for s in 1234 7 99; do timeout 300 python tools/prof_rollout.py --seed $s --warm 1 > $O/parts_seed$s.txt 2>&1; done
timeout 600 python tools/prof_rollout.py --seed 1234 --warm 0 --steps 10 --over PHYSICS.SLEEP_STEPS=0 > $O/parts_nodeact.txt 2>&1
timeout 1500 python bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err
tail -3 $O/tests.log; head -30 $O/parts_seed1234.txt; cut -c1-600 $O/bench.json

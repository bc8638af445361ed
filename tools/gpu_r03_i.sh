#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03_i; mkdir -p $O; cd $R
timeout 1500 python -m pytest tests -m gpu -x -q > $O/tests.log 2>&1; echo "tests rc=$?" >> $O/tests.log
timeout 900 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra-legs > $O/bench.json 2> $O/bench.err
tail -6 $O/tests.log; tail -c 600 $O/bench.json | head -c 300; python -c "
import json; d=json.loads(open('gpurun_out/r03_i/bench.json').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['roofline']['avg_kernel_ms'])"

#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03_j; mkdir -p $O; cd $R
timeout 900 python -m pytest tests/test_gpu_scale.py tests/test_gpu_parity.py -m gpu -x -q > $O/tests.log 2>&1; echo "tests rc=$?" >> $O/tests.log
for i in 1 2 3; do timeout 900 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra-legs 2> $O/bench.err | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step']*20, d['roofline']['avg_kernel_ms'])"; done
tail -3 $O/tests.log

#!/bin/bash
# the driver's own command, on a fresh box: full default bench (CPU legs included) + the whole GPU test-suite + smoke
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03_final; mkdir -p $O; cd $R
T0=$(date +%s); timeout 1500 python bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err; echo "bench.py: $(( $(date +%s) - T0 )) s" > $O/time.txt
T0=$(date +%s); timeout 1500 python -m pytest tests -m gpu -x -q > $O/tests.log 2>&1; echo "tests rc=$? in $(( $(date +%s) - T0 )) s" >> $O/time.txt
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?" >> $O/time.txt
cat $O/time.txt; tail -2 $O/tests.log; tail -1 $O/smoke.log
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r03_final/bench.json').read().strip().splitlines()[-1])
print('value', d['value'], 'lockstep', d['lockstep_env_step']['value'], 'partial', d['lockstep_partial']['value'], 'async', d['async_rollout']['value'])
print('c3', d['config3_4096']['value'], 'c5', d['config5_8192']['value'], d['config5_8192'].get('async_value'), 'c4', d['config4_grasp_2048']['value'], d['config4_grasp_2048']['grasp_success_rate'])
print('cpu', d['cpu_baseline']['value'], 'limb', d['limb_dynamics']['push_1024']['value'], d['limb_dynamics']['grasp_2048']['value'], d['limb_dynamics']['grasp_2048']['grasp_success_rate'])
print('roofline', d['roofline']['frac'], d['roofline']['avg_kernel_ms'], 'deact', {k: round(v['value']) for k, v in d['deactivation'].items() if isinstance(v, dict) and 'value' in v}, d['deactivation']['shipped_vs_reference_semantics'])
print('pose', d['pose_err']['substeps_100'] if d.get('pose_err') else None)
PY

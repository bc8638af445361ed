#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03_l; mkdir -p $O; cd $R
T0=$(date +%s); timeout 1500 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench.json 2> $O/bench.err; echo "Elapsed $(( $(date +%s) - T0 )) s" > $O/time.txt
grep -E "Elapsed|Maximum resident" $O/time.txt; python - <<'PY'
import json
d=json.loads(open('gpurun_out/r03_l/bench.json').read().strip().splitlines()[-1])
print('value', d['value'], 'lockstep', d['lockstep_env_step']['value'], 'partial', d['lockstep_partial']['value'], 'async', d['async_rollout']['value'])
print('partial', {k: d['lockstep_partial'][k] for k in ('polls','ms_per_poll_call','kernel_ms_per_poll','ms_per_poll_loop','steps_per_env_min_max')}); print('lockstep kernel ms/step', d['lockstep_env_step'].get('kernel_ms_per_step'), 'ms/step', d['lockstep_env_step']['ms_per_step']); print('c3', d['config3_4096']['value'], 'c5', d['config5_8192']['value'], d['config5_8192'].get('async_value'), 'c4', d['config4_grasp_2048']['value'])
print('limb', d['limb_dynamics']['push_1024']['value'], d['limb_dynamics']['grasp_2048']['value'])
print('pose_err', json.dumps(d.get('pose_err'))[:600])
PY

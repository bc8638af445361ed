"""Print the numbers of a bench.py JSON line that the round's targets are stated in."""
import json
import sys

d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
if 'legs' in d:       # the compact stdout line of round 6: the legs are scalar groups under `legs`; the full record is the side file
    for k, v in d['legs'].items():
        d.setdefault(k, v)


def g(*path, default=None):
    x = d
    for p in path:
        if not isinstance(x, dict) or p not in x:
            return default
        x = x[p]
    return x


def r(x):
    return None if x is None else round(x, 1)


print('value', r(d['value']), 'ms/step', r(d['ms_per_step']), 'frac', round(d['roofline']['frac'], 4), 'kernel ms', r(d['roofline']['avg_kernel_ms']))
for leg in ('config2_k50', 'lockstep_env_step', 'lockstep_partial', 'async_rollout', 'config3_4096', 'config5_8192', 'config4_grasp_2048'):
    print(leg, r(g(leg, 'value')), r(g(leg, 'async_value')))
print('limb', r(g('limb_dynamics', 'push_1024', 'value')), r(g('limb_dynamics', 'grasp_2048', 'value')))
for k, v in (g('deactivation') or {}).items():
    if isinstance(v, dict) and 'value' in v:
        print('deactivation.' + k, r(v['value']), 'disp', r(v.get('disp_mean_mm')), 'p50', v.get('disp_p50_mm'), 'useful/unsafe/ineff',
              round(v['useful'], 3), round(v['unsafe'], 3), round(v['ineffective'], 3))
for k, v in (g('reference_semantics') or {}).items():
    if isinstance(v, dict):
        print('reference_semantics.' + k, json.dumps({a: (round(b, 3) if isinstance(b, float) else b) for a, b in v.items() if not isinstance(b, (dict, str))}))
print('cpu_baseline', json.dumps(g('cpu_baseline')))
print('pose_err', json.dumps(g('pose_err', 'substeps_100')), json.dumps(g('pose_err', 'end_of_push')))

"""Rates of the two-waves-per-SIMD build of the env kernel (RV_ENV_OCC=2) on the headline workload and the big-world configs:
what a change of that build buys.   python tools/occ2_check.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from robovat_amd import configs, scenes, lib
scene, names = scenes.make_scene()
genv = configs.grasp_env_config(); gscene, gnames = scenes.make_scene(env_cfg=genv)
CASES = [('headline workload on the 256-register build', 1024, {}, 20, 5, '2'),
         ('c5', 8192, {}, 20, 5, None), ('c3', 4096, dict(TASK_NAME='crossing', LAYOUT_ID=0, MOVABLE_NAME='CONCAVE', MAX_STEPS=10), 10, 0, None),
         ('c4 grasp', 2048, 'grasp', 10, 0, None), ('nd 8192', 8192, {'PHYSICS.SLEEP_STEPS': 0}, 8, 0, None)]
want = sys.argv[1:]
for name, n, over, k, warm, occ in CASES:
    if want and not any(name.startswith(x) for x in want):
        continue
    os.environ.pop('RV_ENV_OCC', None)
    if occ:
        os.environ['RV_ENV_OCC'] = occ
    if over == 'grasp':
        cfg = configs.make_rv_config(env_cfg=genv, n_envs=n, seed=1234, shape_names=gnames); sc = gscene
    else:
        cfg = configs.make_rv_config(env_cfg=configs.push_env_config(**over), n_envs=n, seed=1234, shape_names=names); sc = scene
    rates = []
    for rep in range(2):
        w = lib.World(cfg, sc, device=0)
        w.reset()
        if warm:
            w.rollout(warm, first_macro_index=0, auto_reset=True, record=True)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        w.rollout(k, first_macro_index=warm, auto_reset=True, record=True)
        st = w.stats(); el = time.perf_counter() - t0
        rates.append(st['env_steps'] / el); km = w.last_kernel_ms()
        w.close()
    print('%-46s env-steps/s %s   kernel %.1f ms' % (name, ' '.join('%.0f' % r for r in rates), km), flush=True)

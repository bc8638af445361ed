"""Stress of the task-queue hand-over (rv_env_kernel.h): rollouts of a big world through the queue against the plain launch, many
times with changing sizes / step counts; prints the number of mismatching envs (0 expected) per run."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from robovat_amd import configs, scenes, lib
scene, names = scenes.make_scene()
bad = 0
for it, (n, k) in enumerate([(8192, 10), (4608, 8), (6000, 12), (8192, 8), (3000, 9), (8192, 16), (5000, 8), (8192, 10)]):
    outs = []
    for q in ('0', '1'):
        os.environ['RV_QUEUE'] = q
        cfg = configs.make_rv_config(env_cfg=configs.push_env_config(), n_envs=n, seed=100 + it, shape_names=names)
        w = lib.World(cfg, scene, device=0)
        w.reset()
        r, d = w.rollout(k, first_macro_index=0, auto_reset=True, record=True)
        taken = None
        if q == '1':
            taken = w.rollout_async(3 * n, first_macro_index=k).cpu().numpy()
        outs.append((w.body_state().cpu().numpy() if q == '0' else None, r.cpu().numpy(), d.cpu().numpy(), w.stats()['env_steps'], taken))
        if q == '1':
            ok_async = int(taken.sum()) == 3 * n
        w.close()
    m = int((outs[0][1] != outs[1][1]).any(0).sum() + (outs[0][2] != outs[1][2]).any(0).sum())
    bad += m + (not ok_async)
    print('run %d: %d envs x %d steps: envs with a differing reward / done row %d; async pool fully taken: %s' % (it, n, k, m, ok_async), flush=True)
print('BAD', bad)
sys.exit(1 if bad else 0)

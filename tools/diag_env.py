#!/usr/bin/env python
"""Follow ONE env of the bench workload step by step (diagnostic; MI355X):
    python tools/diag_env.py <env> [first_step=100] [n_steps=20]
prints, per env.step(), the substeps / awake substeps / convex queries of the env and what its bodies did."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
from robovat_amd import configs, scenes, lib  # noqa: E402

env = int(sys.argv[1]); first = int(sys.argv[2]) if len(sys.argv) > 2 else 100; n = int(sys.argv[3]) if len(sys.argv) > 3 else 20
scene, names = scenes.make_scene()
cfg = configs.make_rv_config(n_envs=1024, seed=1234, shape_names=names)
w = lib.World(cfg, scene, device=0)
w.reset()
k = 0
while k < first:
    m = min(20, first - k)
    w.rollout(m, first_macro_index=k, auto_reset=True, record=False); w.synchronize(); k += m
for s in range(n):
    b0 = w.body_state().cpu().numpy()[env]
    w.rollout(1, first_macro_index=k, auto_reset=True, record=False); w.synchronize(); k += 1
    c = w.env_counters().cpu().numpy()[env]
    b1 = w.body_state().cpu().numpy()[env]
    prm = w.body_params().cpu().numpy()[env]
    mv = np.linalg.norm(b1[:, :3] - b0[:, :3], axis=1)
    print('step %3d: substeps %5d awake %5d queries %5d | moved mm %s | z %s | shape %s asleep %s' % (
        k - 1, c[7], c[8], c[9], np.round(1e3 * mv, 1), np.round(b1[:, 2], 3), prm[:, 1].astype(int), prm[:, 7].astype(int)))

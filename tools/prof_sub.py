"""Profiling target: reset N envs, then K substeps.  Extra args: key=value overrides of PUSH_ENV_CONFIG."""
import sys, os, time, ast
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from robovat_amd import configs, scenes, lib
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
k = int(sys.argv[2]) if len(sys.argv) > 2 else 200
over = {}
link = False
for a in sys.argv[3:]:
    if a == 'link':
        link = True
        continue
    key, val = a.split('=')
    over[key] = ast.literal_eval(val)
scene, names = scenes.make_scene()
cfg = configs.make_rv_config(env_cfg=configs.push_env_config(**over), n_envs=n, seed=1234, shape_names=names)
w = lib.World(cfg, scene, 0)
w.reset(); w.synchronize()
print('reset ms %.1f' % w.last_kernel_ms(), over)
if link:
    import numpy as np
    w.step_sub(300)
    pose = np.tile(np.array([[0.6, 0.1, 0.3, 1.0, 0.0, 0.0, 0.0]], np.float32), (n, 1))
    w.set_link_target(pose)
for _ in range(3):
    w.step_sub(k); w.synchronize()
    st = w.stats()
    print('sub %d ms %.3f -> %.2f us/substep (awake frac %.2f)' % (k, w.last_kernel_ms(), 1e3 * w.last_kernel_ms() / k, st['awake_substeps'] / max(st['substeps'], 1)))

"""Profiling target: reset N envs, then K substeps (arm enabled, bodies resting)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from robovat_amd import configs, scenes, lib
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
k = int(sys.argv[2]) if len(sys.argv) > 2 else 200
scene, names = scenes.make_scene()
cfg = configs.make_rv_config(n_envs=n, seed=1234, shape_names=names)
w = lib.World(cfg, scene, 0)
w.reset(); w.synchronize()
print('reset ms', w.last_kernel_ms(), w.stats())
for _ in range(3):
    w.step_sub(k); w.synchronize()
    print('sub %d ms %.3f -> %.2f us/substep' % (k, w.last_kernel_ms(), 1e3 * w.last_kernel_ms() / k))

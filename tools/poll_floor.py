import sys, time
sys.path.insert(0, '.')
import torch
from robovat_amd import configs, scenes, lib
scene, names = scenes.make_scene()
n = 1024
cfg = configs.make_rv_config(n_envs=n, seed=1234, shape_names=names)
w = lib.World(cfg, scene, 0)
w.reset()
A = w.policy_random(0)
w.step_begin(A)
for label, kw in (('max_substeps=1', dict(max_substeps=1)), ('max_substeps=50', dict(max_substeps=50)), ('max_usec=1', dict(max_usec=1)), ('max_usec=200', dict(max_usec=200)), ('max_usec=800', dict(max_usec=800))):
    ks = []
    for i in range(12):
        if kw is None:
            w.step_sub(1)
        else:
            w.step_poll(**kw)
        w.synchronize(); ks.append(w.last_kernel_ms())
    print('%-16s kernel ms: %s' % (label, ' '.join('%.3f' % k for k in ks)))

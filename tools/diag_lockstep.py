#!/usr/bin/env python
"""Which env sets the pace of a lock-step env.step()?  (diagnostic; MI355X)

Builds a -DRV_DIAG_AWAKE_BODIES variant of the library (awake_last = awake-body count * 65536 +
awake substeps), runs K lock-step steps of BASELINE config 2 and prints, per step, the kernel
time and the slowest envs' substeps / awake substeps / mean awake bodies / convex queries.
"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
out = os.path.join(ROOT, 'robovat_amd', 'librovat_hip_diag.so')
from robovat_amd import lib as _l  # noqa: E402
subprocess.run([os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')] + _l.HIPCC_FLAGS + ['-DRV_DIAG_AWAKE_BODIES',
               os.path.join(_l.CSRC, 'rv_kernels.hip'), '-o', out], check=True, stderr=subprocess.DEVNULL)
os.environ['RV_LIB'] = out
_l.LIB_PATH = out
import numpy as np  # noqa: E402
from robovat_amd import configs, scenes, lib  # noqa: E402

scene, names = scenes.make_scene()
cfg = configs.make_rv_config(n_envs=1024, seed=1234, shape_names=names)
w = lib.World(cfg, scene, device=0)
w.reset()
K = int(sys.argv[1]) if len(sys.argv) > 1 else 12
for k in range(K):
    w.set_actions(w.policy_random(k)); w.step_macro(); w.synchronize()
    ms = w.last_kernel_ms()
    c = w.env_counters().cpu().numpy()
    sub, aw, pairs = c[:, 7], c[:, 8] & 0xffff, c[:, 9]
    bodies = (c[:, 8] >> 16) / np.maximum(aw, 1)
    order = np.argsort(-aw)[:3]
    print('step %2d kernel %7.1f ms | mean awake %5.0f | slowest by awake substeps: ' % (k, ms, aw.mean()) +
          ' ; '.join('env %4d sub %5d awake %5d bodies %.2f pairs %5d' % (i, sub[i], aw[i], bodies[i], pairs[i]) for i in order))

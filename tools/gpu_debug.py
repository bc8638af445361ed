"""GPU bisecting aid: compares tiny scenarios with the float oracle and prints the first divergence."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from robovat_amd import abi, configs, scenes, lib
from oracle import orc

scene, names = scenes.make_scene()
N = 2
cfg = configs.make_rv_config(n_envs=N, seed=5, shape_names=names)
w = lib.World(cfg, scene, 0); r = orc.OracleWorld(cfg, scene)

def show(tag):
    a = w.body_state().cpu().numpy(); b = r.body_state().astype(np.float32)
    print(tag, 'maxdiff', np.abs(a - b).max(), 'nan', np.isnan(a).any(), flush=True)
    if not np.array_equal(a, b):
        print(' gpu', a[0, 0]); print(' orc', b[0, 0])
    print('  man', w.manifold_counts().cpu().numpy()[0], r.manifold_counts()[0], 'cnt', w.env_counters().cpu().numpy()[0], r.env_counters()[0])

p = np.zeros((N, 4, 8), np.float32); p[:, 0] = [1, 0, 1.0, 0.2, 0.5, 0, 0.0, 0]
s = np.zeros((N, 4, 13), np.float32); s[..., 6] = 1; s[:, 0, :3] = [0.6, 0.0, 0.5]
for x in (w, r):
    x.set_body_params(p); x.set_body_state(s)
print('params', w.body_params().cpu().numpy()[0, 0], r.body_params()[0, 0])
show('init')
for k in range(3):
    w.step_sub(1); r.step_sub(1); show('freefall %d' % k)
s[:, 0, 2] = 0.035
for x in (w, r):
    x.set_body_state(s)
for k in range(5):
    w.step_sub(1); r.step_sub(1); show('contact %d' % k)
w.step_sub(200); r.step_sub(200); show('contact 205')
w.reset(); r.reset(); show('reset')
print('params', w.body_params().cpu().numpy()[0], '\n', r.body_params()[0])
print('stats', w.stats(), r.stats())

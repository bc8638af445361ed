import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from robovat_amd import abi, configs, scenes, lib
from oracle import orc
scene, names = scenes.make_scene()
N = 2
cfg = configs.make_rv_config(n_envs=N, seed=5, shape_names=names)
w = lib.World(cfg, scene, 0); r = orc.OracleWorld(cfg, scene)
def show(tag):
    a = w.body_state().cpu().numpy(); b = r.body_state().astype(np.float32)
    ja = w.joint_state().cpu().numpy(); jb = r.joint_state().astype(np.float32)
    print(tag, 'body maxdiff', np.abs(a - b).max(), 'joint maxdiff', np.abs(ja - jb).max(), 'cnt', w.env_counters().cpu().numpy()[0], r.env_counters()[0], flush=True)
w.reset(); r.reset(); show('reset')
for n in (1, 1, 8, 10, 100, 1000):
    t = time.time(); w.step_sub(n); w.synchronize(); dt = time.time() - t
    r.step_sub(n); show('sub %d (%.4fs)' % (n, dt))
a = r.policy_random(0); w.set_actions(a); r.set_actions(a)
t = time.time(); w.step_macro(); w.synchronize(); dt = time.time() - t
r.step_macro(); show('macro (%.3fs)' % dt)
print(w.stats(), r.stats())

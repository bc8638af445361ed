"""Where the non-kernel time of the timed rollout_record goes (allocation of the output buffers, zero fill,
k_point_cloud, stats)."""
import sys, time
sys.path.insert(0, '.')
import torch
from robovat_amd import configs, scenes, lib
scene, names = scenes.make_scene()
cfg = configs.make_rv_config(n_envs=1024, seed=1234, shape_names=names)
w = lib.World(cfg, scene, 0)
w.reset()
for k in range(5):
    w.set_actions(w.policy_random(k)); w.step_macro(); w.observe(point_cloud=True)
def sync(): torch.cuda.synchronize()
for rep in range(3):
    sync(); t0 = time.perf_counter()
    obs, b = w._obs_buffers((20, 1024), True, False)
    sync(); t1 = time.perf_counter()
    del obs, b
    sync(); t2 = time.perf_counter()
    o, r, d = w.rollout_record(20, first_macro_index=5 + 20 * rep, auto_reset=True, point_cloud=True)
    sync(); t3 = time.perf_counter()
    st = w.stats(); sync(); t4 = time.perf_counter()
    print('rep %d: alloc+zero %.2f ms | rollout_record %.2f ms (kernel %.2f) | stats %.2f ms' % (rep, 1e3 * (t1 - t0), 1e3 * (t3 - t2), w.last_kernel_ms(), 1e3 * (t4 - t3)))
    del o, r, d

import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from robovat_amd import configs, scenes, lib
scene, names = scenes.make_scene()
cfg = configs.make_rv_config(env_cfg=configs.push_env_config(), n_envs=8192, seed=1234, shape_names=names)
for mode in ('rollout', 'record', 'record+pc', 'rollout', 'record+pc'):
    w = lib.World(cfg, scene, device=0)
    w.reset()
    w.rollout(5, first_macro_index=0, auto_reset=True, record=True)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    if mode == 'rollout':
        w.rollout(20, first_macro_index=5, auto_reset=True, record=True)
    else:
        w.rollout_record(20, first_macro_index=5, auto_reset=True, point_cloud=(mode == 'record+pc'))
    st = w.stats(); el = time.perf_counter() - t0
    print('%-10s env-steps/s %.0f  kernel %.1f ms  elapsed %.1f ms' % (mode, st['env_steps'] / el, w.last_kernel_ms(), 1e3 * el), flush=True)
    w.close()

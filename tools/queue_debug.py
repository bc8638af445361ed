import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ['RV_QUEUE_DEBUG'] = '1'
import torch
from robovat_amd import configs, scenes, lib
scene, names = scenes.make_scene()
cfg = configs.make_rv_config(env_cfg=configs.push_env_config(), n_envs=8192, seed=1234, shape_names=names)
for rep in range(2):
    w = lib.World(cfg, scene, device=0)
    w.reset(); w.rollout(5, first_macro_index=0, auto_reset=True, record=True); w.stats()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    w.rollout(20, first_macro_index=5, auto_reset=True, record=True)
    st = w.stats(); el = time.perf_counter() - t0
    print('c5 %.0f env-steps/s kernel %.1f ms' % (st['env_steps'] / el, w.last_kernel_ms()), flush=True)
    w.close()

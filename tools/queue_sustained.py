"""config 5 (8192 envs) under sustained load: four 20-step rollouts back to back on one world, per queue mode.
    python tools/queue_sustained.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from robovat_amd import configs, scenes, lib
scene, names = scenes.make_scene()
cfg = configs.make_rv_config(env_cfg=configs.push_env_config(), n_envs=8192, seed=1234, shape_names=names)
for name, env in (('queues per XCD', {}), ('one queue', {'RV_QUEUE_GLOBAL': '1'}), ('queues per XCD', {}), ('one queue', {'RV_QUEUE_GLOBAL': '1'}), ('plain launch', {'RV_QUEUE': '0'})):
    for k in ('RV_QUEUE_GLOBAL', 'RV_QUEUE'):
        os.environ.pop(k, None)
    os.environ.update(env)
    w = lib.World(cfg, scene, device=0)
    w.reset()
    w.rollout(5, first_macro_index=0, auto_reset=True, record=True)
    out = []
    idx = 5
    for rep in range(4):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        w.rollout(20, first_macro_index=idx, auto_reset=True, record=True)
        st = w.stats(); el = time.perf_counter() - t0
        out.append('%.0f (%.0f ms, %.2f M substeps/step)' % (st['env_steps'] / el, w.last_kernel_ms(), st['substeps'] / 20 / 1e6))
        idx += 20
    print('%-16s %s' % (name, '  '.join(out)), flush=True)
    w.close()

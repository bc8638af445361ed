"""Rollout rate of the big-world configurations under the scheduling variants of the task queues (RV_QUEUE, RV_QUEUE_WT):
one process, one world per (config, variant), a warm-up launch and two timed ones.

    python tools/queue_variants.py [c5] [c3] [c4] [nd]
"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from robovat_amd import configs, scenes, lib

which = sys.argv[1:] or ['c5', 'c3']
scene, names = scenes.make_scene()
CONFIGS = {
    'c5': (8192, {}, 20, 5),
    'c5k10': (8192, {}, 10, 5),
    'c5k5': (8192, {}, 5, 5),
    'c5k2': (8192, {}, 2, 5),
    'c3k5': (4096, dict(TASK_NAME='crossing', LAYOUT_ID=0, MOVABLE_NAME='CONCAVE', MAX_STEPS=10), 5, 0),
    'nd50': (8192, {'PHYSICS.SLEEP_STEPS': 0, 'PHYSICS.SOLVER_TOL': 0.0, 'PHYSICS.SOLVER_STALL': 0}, 4, 0),
    'c3': (4096, dict(TASK_NAME='crossing', LAYOUT_ID=0, MOVABLE_NAME='CONCAVE', MAX_STEPS=10), 10, 0),
    'c4': (2048, 'grasp', 10, 0),
    'c2k': (2048, {}, 20, 5),
    'c3k': (3072, {}, 20, 5),
    'c4k': (4096, {}, 20, 5),
    'c6k': (6144, {}, 20, 5),
    'nd2k': (2048, {'PHYSICS.SLEEP_STEPS': 0}, 8, 0),
    'nd': (8192, {'PHYSICS.SLEEP_STEPS': 0}, 8, 0),
}
VARIANTS = [('plain launch', {'RV_QUEUE': '0'}), ('queues, sticky', {'RV_QUEUE': '1', 'RV_QUEUE_STICKY': '1'}),
            ('queues, fifo', {'RV_QUEUE': '1', 'RV_QUEUE_STICKY': '0'})]
if os.environ.get('VARIANTS') == 'occ':      # which build of the env kernel: all registers / one wave per SIMD vs 256 registers / two
    VARIANTS = [('default', {}), ('1 wave per SIMD, plain', {'RV_ENV_OCC': '1', 'RV_QUEUE': '0'}), ('1 wave per SIMD, queues', {'RV_ENV_OCC': '1', 'RV_QUEUE': '1'}),
                ('2 waves per SIMD, plain', {'RV_ENV_OCC': '2', 'RV_QUEUE': '0'}), ('2 waves per SIMD, queues', {'RV_ENV_OCC': '2', 'RV_QUEUE': '1'})]
for name in which:
    n, over, k, warm = CONFIGS[name]
    for vname, env in VARIANTS:
        for key in ('RV_QUEUE', 'RV_QUEUE_WT', 'RV_QUEUE_STICKY', 'RV_ENV_OCC'):
            os.environ.pop(key, None)
        os.environ.update(env)
        if over == 'grasp':
            genv = configs.grasp_env_config()
            gscene, gnames = scenes.make_scene(env_cfg=genv)
            cfg = configs.make_rv_config(env_cfg=genv, n_envs=n, seed=1234, shape_names=gnames)
            scene_ = gscene
        else:
            cfg = configs.make_rv_config(env_cfg=configs.push_env_config(**over), n_envs=n, seed=1234, shape_names=names)
            scene_ = scene
        rates = []
        for rep in range(2):
            w = lib.World(cfg, scene_, device=0)
            w.reset()
            if warm:
                w.rollout(warm, first_macro_index=0, auto_reset=True, record=True)
            torch.cuda.synchronize(); t0 = time.perf_counter()
            w.rollout(k, first_macro_index=warm, auto_reset=True, record=True)
            st = w.stats()
            el = time.perf_counter() - t0
            rates.append(st['env_steps'] / el)
            km = w.last_kernel_ms()
            w.close()
        print('%-6s %-24s env-steps/s %s   kernel %.1f ms' % (name, vname, ' '.join('%.0f' % r for r in rates), km), flush=True)

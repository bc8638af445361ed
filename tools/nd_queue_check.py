"""The no-deactivation same-steps leg of bench.py (8192 envs, K steps per env from a reset, records kept) as a plain launch
(one workgroup per env) and through the task queue (RV_QUEUE=1: one env.step() per task), the queue's threshold being K >= 12
for the shipped semantics.  Without deactivation a task is long (thousands of awake substeps) and the envs differ 3 - 4 x.
    python tools/nd_queue_check.py [K=8] [50sweeps]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from robovat_amd import configs, scenes, lib
K = int(sys.argv[1]) if len(sys.argv) > 1 else 8
over = {'PHYSICS.SLEEP_STEPS': 0}
if len(sys.argv) > 2: over.update({'PHYSICS.SOLVER_TOL': 0.0, 'PHYSICS.SOLVER_STALL': 0})
scene, names = scenes.make_scene()
cfg = configs.make_rv_config(env_cfg=configs.push_env_config(**over), n_envs=8192, seed=0, shape_names=names)
res = {}
for q in ('0', '1', '0', '1'):
    os.environ['RV_QUEUE'] = q
    w = lib.World(cfg, scene, device=0); w.reset(); w.synchronize()
    t = time.time(); obs, r, d = w.rollout_record(K, first_macro_index=0, auto_reset=False, point_cloud=False); w.synchronize(); el = time.time() - t
    st = w.stats(); bs = w.body_state().cpu().numpy(); rr = r.cpu().numpy()
    print('RV_QUEUE=%s K=%d: kernel %.0f ms, %.0f env-steps/s, substeps/env-step %.0f' % (q, K, w.last_kernel_ms(), st['env_steps'] / el, st['substeps'] / max(st['env_steps'], 1)), flush=True)
    if q in res: assert np.array_equal(res[q][0], bs) and np.array_equal(res[q][1], rr)
    res[q] = (bs, rr); w.close()
print('queue == plain launch:', np.array_equal(res['0'][0], res['1'][0]) and np.array_equal(res['0'][1], res['1'][1]))

"""Stress of the task-queue hand-over (rv_env_kernel.h: one queue per XCD, plain stores -> vmcnt(0) -> slot; poll -> acquire
fence -> plain loads): rollouts of a big world through the queues against the plain launch, many times with changing sizes,
step counts and LOADS -- the shipped semantics (most tasks short, a few long: uneven), no deactivation (every task long)
and no deactivation + 50 plain sweeps --, comparing EVERY word the host can read back of every env (body states, joint
states, env counters, per-step rewards / dones) and the queue's own assertion word (a block that arrives with the wrong
step / launch number makes rv_get_stats fail).  Prints the number of mismatching envs per run; 0 expected, exit 1 otherwise.

    python tools/queue_stress.py [runs=16]
"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from robovat_amd import configs, scenes, lib

runs = int(sys.argv[1]) if len(sys.argv) > 1 else 16
scene, names = scenes.make_scene()
SIZES = [(8192, 10), (4608, 8), (6000, 12), (8192, 8), (3000, 9), (8192, 16), (5000, 8), (8192, 20)]
LOADS = [({}, 1.0), ({}, 1.0), ({'PHYSICS.SLEEP_STEPS': 0}, 0.25), ({}, 1.0),
         ({'PHYSICS.SLEEP_STEPS': 0, 'PHYSICS.SOLVER_TOL': 0.0, 'PHYSICS.SOLVER_STALL': 0}, 0.2)]
bad = errors = 0
for it in range(runs):
    n, k = SIZES[it % len(SIZES)]
    over, shrink = LOADS[it % len(LOADS)]
    k = max(2, int(k * shrink)) if shrink < 1.0 else k          # (the slow semantics: fewer steps, same number of hand-overs per env-step)
    outs = []
    for q in ('0', '1'):
        os.environ['RV_QUEUE'] = q
        cfg = configs.make_rv_config(env_cfg=configs.push_env_config(**over), n_envs=n, seed=100 + it, shape_names=names)
        w = lib.World(cfg, scene, device=0)
        w.reset()
        r, d = w.rollout(k, first_macro_index=0, auto_reset=True, record=True)
        try:
            st = w.stats()
        except RuntimeError as ex:
            print('run %d: queue assertion: %s' % (it, ex), flush=True); errors += 1; st = {'env_steps': -1}
        state = (w.body_state().cpu().numpy(), w.joint_state().cpu().numpy(), w.env_counters().cpu().numpy())
        taken, ok_async = None, True
        if q == '1':
            taken = w.rollout_async(3 * n, first_macro_index=k).cpu().numpy()
            try:
                w.stats()
            except RuntimeError as ex:
                print('run %d: queue assertion (pool): %s' % (it, ex), flush=True); errors += 1
            ok_async = int(taken.sum()) == 3 * n
        outs.append((state, r.cpu().numpy(), d.cpu().numpy(), st['env_steps']))
        w.close()
    (sa, ra, da, ea), (sb, rb, db, eb) = outs
    m = np.zeros(n, bool)
    for x, y in zip(sa, sb):
        m |= (x.reshape(n, -1) != y.reshape(n, -1)).any(1)
    m |= (ra != rb).any(0) | (da != db).any(0)
    bad += int(m.sum()) + (not ok_async) + (ea != eb)
    print('run %d: %d envs x %d steps %s: envs with any differing word %d; env_steps %d / %d; async pool fully taken: %s'
          % (it, n, k, over or 'shipped', int(m.sum()), ea, eb, ok_async), flush=True)
print('BAD', bad, 'QUEUE_ASSERTIONS', errors)
sys.exit(1 if (bad or errors) else 0)

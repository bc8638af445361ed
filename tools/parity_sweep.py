#!/usr/bin/env python
"""HIP kernel vs float oracle, bit for bit, over many seeds / configs (MI355X; a wider net than the test-suite):
    python tools/parity_sweep.py [n_seeds=8] [n_envs=128] [steps=8] [split]
split: the steps of a run as several launches of 1, 2, 1, 3, ... steps (launches that begin with envs whose episode has just
ended, i.e. with a reset; the oracle's result does not depend on the split).  RV_LIB=build/librovat_poison_*.so: the same with
the LDS scratch block starting as garbage in every launch.
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
from robovat_amd import configs, scenes, lib  # noqa: E402
from oracle import orc  # noqa: E402

n_seeds = int(sys.argv[1]) if len(sys.argv) > 1 else 8
n = int(sys.argv[2]) if len(sys.argv) > 2 else 128
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 8
split = len(sys.argv) > 4 and sys.argv[4] == 'split'


def launches(total):
    if not split:
        return [(0, total)]
    out, k, i = [], 0, 0
    while k < total:
        c = min((1, 2, 1, 3)[i % 4], total - k); out.append((k, c)); k += c; i += 1
    return out


CASES = [('config 2', {}), ('crossing / concave', dict(TASK_NAME='crossing', LAYOUT_ID=0, MOVABLE_NAME='CONCAVE', MAX_STEPS=10)),
         ('crowded: 4 bodies in a 20 cm square', {'MOVABLE.CONVEX.POSE.X': [0.5, 0.7], 'MOVABLE.CONVEX.POSE.Y': [-0.1, 0.1], 'MOVABLE.CONVEX.MARGIN': 0.07}),
         ('dynamic limb, 1-4 bodies', {'PHYSICS.LIMB_DYNAMICS': 1, 'MIN_MOVABLE_BODIES': 1, 'MAX_MOVABLE_BODIES': 4}),
         ('no deactivation', {'PHYSICS.SLEEP_STEPS': 0, 'MAX_STEPS': 2}),
         ('no deactivation, crowded, rest tol 1e-6', {'PHYSICS.SLEEP_STEPS': 0, 'PHYSICS.SOLVER_TOL_REST': 1e-6, 'MAX_STEPS': 2, 'MOVABLE.CONVEX.POSE.X': [0.5, 0.7],
                                                      'MOVABLE.CONVEX.POSE.Y': [-0.1, 0.1], 'MOVABLE.CONVEX.MARGIN': 0.07}),
         ('arm effort limit + tilted gravity', {'PHYSICS.ARM_EFFORT_LIMIT': 1, 'PHYSICS.GRAVITY_XY': (0.3, -0.2)}),
         ('dynamic limb, crowded', {'PHYSICS.LIMB_DYNAMICS': 1, 'MOVABLE.CONVEX.POSE.X': [0.5, 0.7], 'MOVABLE.CONVEX.POSE.Y': [-0.1, 0.1], 'MOVABLE.CONVEX.MARGIN': 0.07}),
         ('user constraints (p2p and prismatic to the world, body - body fixed)', {'CONSTRAINTS': 1}),
         ('user constraints (revolute to the world, fixed to the hand link, body - body prismatic)', {'CONSTRAINTS': 2}),
         ('static wall across the workspace, 3 bodies', {'SIM.WALL.USE': True, 'SIM.WALL.POSE': [[0.66, 0.0, 0.4], [0, 0, 0]], 'MAX_MOVABLE_BODIES': 3, 'MIN_MOVABLE_BODIES': 3}),
         ('static wall, bodies dropped next to / into it, no deactivation', {'SIM.WALL.USE': True, 'SIM.WALL.POSE': [[0.72, 0.0, 0.4], [0, 0, 0.3]], 'MAX_MOVABLE_BODIES': 3, 'MIN_MOVABLE_BODIES': 2,
                                                                'PHYSICS.SLEEP_STEPS': 0, 'MAX_STEPS': 2, 'MOVABLE.CONVEX.POSE.X': [0.5, 0.7], 'MOVABLE.CONVEX.POSE.Y': [-0.1, 0.1], 'MOVABLE.CONVEX.MARGIN': 0.07}),
         ('static wall + dynamic limb', {'SIM.WALL.USE': True, 'SIM.WALL.POSE': [[0.6, 0.12, 0.4], [0, 0, 1.5708]], 'MAX_MOVABLE_BODIES': 3, 'MIN_MOVABLE_BODIES': 1, 'PHYSICS.LIMB_DYNAMICS': 1})]
bad = 0
for name, over in CASES:
    scene, names = scenes.make_scene()
    for seed in range(n_seeds):
        over = dict(over); cons = over.pop('CONSTRAINTS', 0)
        cfg = configs.make_rv_config(env_cfg=configs.push_env_config(**over), n_envs=n, seed=1000 + seed, shape_names=names)
        w = lib.World(cfg, scene, device=0); o = orc.OracleWorld(cfg, scene, double=False)
        w.reset(); o.reset()
        if cons == 2:
            from robovat_amd import abi
            qz = [0, 0, float(np.sin(0.2)), float(np.cos(0.2))]
            for x in (w, o):
                x.set_constraint(1, [0.6, 0.05 * (seed % 3), 0.1] + qz, frame7=[0.02, 0.0, 0.0] + qz, max_force=40.0, joint_type='revolute')
                x.set_constraint(2, [0.0, 0.0, -0.22, 0, 0, 0, 1], max_force=120.0, child=abi.RV_CHILD_LINK(7))
                x.set_constraint(3, [0.0, 0.0, 0.08, 0, 0, 0, 1], max_force=40.0, child=0, joint_type='prismatic')
        elif cons:      # (the constraints are per world: every env gets them; a reset drops them, so no auto-reset below)
            for x in (w, o):
                x.set_constraint(1, [0.6, 0.05 * (seed % 3), 0.12, 0, 0, 0, 1], frame7=[0.02, 0.01, 0.0, 0, 0, 0, 1], max_force=30.0, joint_type='point2point')
                x.set_constraint(2, [0.0, 0.0, 0.07, 0, 0, 0, 1], max_force=40.0, child=0)
                x.set_constraint(3, [0.55, -0.1, 0.1, 0, 0, np.sin(0.3), np.cos(0.3)], frame7=[0, 0, 0, 0, 0, np.sin(0.3), np.cos(0.3)], max_force=40.0, joint_type='prismatic')
        for k0, c in launches(steps if not cons else 2):
            w.rollout(c, first_macro_index=k0, auto_reset=not cons, record=False); w.synchronize()
            o.rollout(c, k0, not cons)
        eq_b = np.array_equal(w.body_state().cpu().numpy(), o.body_state().astype(np.float32))
        eq_j = np.array_equal(w.joint_state().cpu().numpy(), o.joint_state().astype(np.float32))
        eq_c = np.array_equal(w.env_counters().cpu().numpy()[:, :8], o.env_counters()[:, :8])
        st = w.stats()
        print('%-38s seed %d: bodies %s joints %s counters %s | awake %.3f' % (name, 1000 + seed, eq_b, eq_j, eq_c, st['awake_substeps'] / max(st['substeps'], 1)), flush=True)
        bad += not (eq_b and eq_j)
        w.close()
# Grasp4DofEnv (force-limited gripper in the solver, phase machine ticking every substep), kinematic and dynamic limb
for seed in range(max(n_seeds // 2, 1) * 2):
    genv = configs.grasp_env_config(**({'PHYSICS.LIMB_DYNAMICS': 1} if seed % 2 else {}))
    gscene, gnames = scenes.make_scene(env_cfg=genv)
    cfg = configs.make_rv_config(env_cfg=genv, n_envs=n, seed=2000 + seed, shape_names=gnames)
    w = lib.World(cfg, gscene, device=0); o = orc.OracleWorld(cfg, gscene, double=False)
    w.reset(); o.reset()
    for k0, c in launches(min(steps, 4)):
        w.rollout(c, first_macro_index=k0, auto_reset=True, record=False); w.synchronize()
        o.rollout(c, k0, True)
    eq_b = np.array_equal(w.body_state().cpu().numpy(), o.body_state().astype(np.float32))
    eq_j = np.array_equal(w.joint_state().cpu().numpy(), o.joint_state().astype(np.float32))
    st = w.stats()
    print('%-38s seed %d: bodies %s joints %s | successes %d / %d' % ('Grasp4DofEnv' + (', dynamic limb' if seed % 2 else ''), 2000 + seed, eq_b, eq_j, st['successes'], st['env_steps']), flush=True)
    bad += not (eq_b and eq_j)
    w.close()
print('MISMATCHES: %d' % bad)
sys.exit(1 if bad else 0)

#!/usr/bin/env python
"""Physics-parameter sensitivity of BASELINE config 2 (runs on the MI355X).

PyBullet is not available, so the shipped contact / deactivation knobs cannot be
checked against the reference's physics directly.  This script shows how much the
OUTCOMES of the benchmark workload depend on them: the same 1024-env, K-step random-
policy rollout is run with the shipped knobs and with Bullet's published defaults
(rolling/spinning friction 0.001 from
tools/templates/urdf_template.xml:11-16 when the build has those rows), one knob
group at a time and all together, and the table reports push outcomes, the
distribution of the per-step body displacement, substeps per env.step() and
env-steps/s.

    python tools/sensitivity.py [--envs 1024] [--steps 20] > profiles/r03_sensitivity.txt
"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import numpy as np  # noqa: E402

VARIANTS = [
    ('shipped', {}),
    ('rolling / spinning friction 0 (round-1 value; urdf_template: 0.001 = shipped)', {'PHYSICS.ROLLING_FRICTION': 0.0}),
    ('solver: at most 8 iterations (round-1 value; Bullet: 50 = shipped)', {'PHYSICS.SOLVER_ITERS': 8}),
    ('solver: 50 iterations, no early exit (Bullet)', {'PHYSICS.SOLVER_TOL': 0.0, 'PHYSICS.SOLVER_STALL': 0}),
    ('solver: residual exit only, no stall exit (round-2 behaviour)', {'PHYSICS.SOLVER_STALL': 0}),
    ('solver: stall exit after 6 sweeps (shipped: 12)', {'PHYSICS.SOLVER_STALL': 6}),
    ('limb dynamics (joint-space inertia + effort-limited motors while the arm touches a body)', {'PHYSICS.LIMB_DYNAMICS': 1}),
    ('limb joints without an acceleration limit (PyBullet\'s motors: commanded velocity within a step; shipped: 8-20 rad/s^2)', {'PHYSICS.ARM_ACCEL_SCALE': 1000.0}),
    ('limb acceleration limits x 4', {'PHYSICS.ARM_ACCEL_SCALE': 4.0}),
    ('sleep: Bullet\'s rule alone (0.8 m/s, 1 rad/s, 2 s)',
     {'PHYSICS.SLEEP_LINEAR': 0.8, 'PHYSICS.SLEEP_ANGULAR': 1.0, 'PHYSICS.SLEEP_STEPS': 2000, 'PHYSICS.SLEEP_POSITION_WINDOW': 0.0, 'PHYSICS.DEACTIVATION_STEPS': 0}),
    ('sleep: without Bullet\'s rule (the strict thresholds and the pose window only)', {'PHYSICS.DEACTIVATION_STEPS': 0}),
    ('sleep: strict velocity thresholds only (no pose window, no Bullet rule)', {'PHYSICS.SLEEP_POSITION_WINDOW': 0.0, 'PHYSICS.DEACTIVATION_STEPS': 0}),
    ('no deactivation at all (resting islands converge to 1e-7 N s: SOLVER_TOL_REST auto)', {'PHYSICS.SLEEP_STEPS': 0}),
    ('no deactivation, one residual (1e-5 N s) for every island (round 3: resting bodies creep)', {'PHYSICS.SLEEP_STEPS': 0, 'PHYSICS.SOLVER_TOL_REST': 0.0}),
    ('shipped + resting islands converge to 1e-7 N s', {'PHYSICS.SOLVER_TOL_REST': 1e-7}),
    ('no deactivation, residual exit at 1e-7 N s for every island', {'PHYSICS.SLEEP_STEPS': 0, 'PHYSICS.SOLVER_TOL': 1e-7}),
    ('no deactivation, 50 iterations, no early exit', {'PHYSICS.SLEEP_STEPS': 0, 'PHYSICS.SOLVER_TOL': 0.0, 'PHYSICS.SOLVER_STALL': 0}),
    ('contact breaking factor 0.04 (Bullet: 0.02 x the smaller shape\'s disc = shipped)', {'PHYSICS.BREAKING': 0.04}),
    ('narrow phase every substep (no gating)', {'PHYSICS.NARROWPHASE_MAX_AGE': 0}),
    ('all Bullet defaults (solver + sleep, no gating)',
     {'PHYSICS.SOLVER_ITERS': 50, 'PHYSICS.SOLVER_TOL': 0.0, 'PHYSICS.SOLVER_STALL': 0, 'PHYSICS.SLEEP_LINEAR': 0.8, 'PHYSICS.SLEEP_ANGULAR': 1.0,
      'PHYSICS.SLEEP_STEPS': 2000, 'PHYSICS.SLEEP_POSITION_WINDOW': 0.0, 'PHYSICS.NARROWPHASE_MAX_AGE': 0, 'PHYSICS.WAKE_GAP': 1.0, 'PHYSICS.DEACTIVATION_STEPS': 0}),
]


def run(over, n_envs, steps, seed):
    import torch
    from robovat_amd import configs, scenes, lib
    env_cfg = configs.push_env_config(**over)
    scene, names = scenes.make_scene(env_cfg=env_cfg)
    cfg = configs.make_rv_config(env_cfg=env_cfg, n_envs=n_envs, seed=seed, shape_names=names)
    w = lib.World(cfg, scene, device=0)
    w.reset()
    pos0 = w.observe()['position']
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    obs, r, d = w.rollout_record(steps, first_macro_index=0, auto_reset=False, point_cloud=False)
    st = w.stats()
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    pos = torch.cat([pos0[None], obs['position']], 0).cpu().numpy()          # [K+1, N, B, 3]
    disp = np.linalg.norm(pos[1:, ..., :2] - pos[:-1, ..., :2], axis=-1)      # per step, env, body
    moved = disp.sum(-1)                                                       # per step, env
    off = (pos[-1, ..., 2] < -0.05).mean()
    w.close()
    es = max(st['env_steps'], 1)
    return {
        'useful': st['useful'] / es, 'unsafe': st['unsafe'] / es, 'ineffective': st['ineffective'] / es,
        'disp_mean_mm': 1e3 * float(moved.mean()), 'disp_p50_mm': 1e3 * float(np.percentile(moved, 50)),
        'disp_p90_mm': 1e3 * float(np.percentile(moved, 90)), 'disp_p99_mm': 1e3 * float(np.percentile(moved, 99)),
        'bodies_off_table': float(off),
        'substeps_per_env_step': st['substeps'] / es, 'awake_frac': st['awake_substeps'] / max(st['substeps'], 1),
        'env_steps_per_s': st['env_steps'] / el,
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--envs', type=int, default=1024)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--seed', type=int, default=1234)
    ap.add_argument('--json', default=None)
    ap.add_argument('--only', default=None, help='run the variants whose name contains this text')
    args = ap.parse_args()
    rows = []
    for name, over in VARIANTS:
        if args.only and args.only not in name:
            continue
        res = run(over, args.envs, args.steps, args.seed)
        rows.append((name, over, res))
    print('# tools/sensitivity.py: BASELINE config 2, %d envs x %d env.step() (random policy, seed %d, one rv_rollout_record launch)'
          % (args.envs, args.steps, args.seed))
    print('# displacement = sum over the bodies of an env of the xy distance moved by one env.step(), mm')
    hdr = '%-78s %7s %7s %7s | %8s %7s %7s %7s | %8s %8s %7s | %9s' % (
        'variant', 'useful', 'unsafe', 'ineff', 'mean', 'p50', 'p90', 'p99', 'off-tbl', 'sub/step', 'awake', 'steps/s')
    print(hdr)
    for name, over, r in rows:
        print('%-78s %7.3f %7.3f %7.3f | %8.2f %7.2f %7.2f %7.2f | %8.4f %8.0f %7.3f | %9.0f' % (
            name, r['useful'], r['unsafe'], r['ineffective'], r['disp_mean_mm'], r['disp_p50_mm'], r['disp_p90_mm'],
            r['disp_p99_mm'], r['bodies_off_table'], r['substeps_per_env_step'], r['awake_frac'], r['env_steps_per_s']))
    if args.json:
        with open(args.json, 'w') as f:
            json.dump([{'variant': n, 'overrides': o, **r} for n, o, r in rows], f, indent=1)


if __name__ == '__main__':
    main()

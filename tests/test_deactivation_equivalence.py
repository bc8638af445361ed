"""Is the shipped island deactivation an optimisation WITHIN the stated tolerance, or a change of semantics?

The reference loads its bodies without URDF_ENABLE_SLEEPING (bullet_physics.py:173-181), so PyBullet most likely never
deactivates the movables and runs its 50 solver sweeps without an early exit (bullet_physics.py:106-109).  The shipped
default deactivates islands at rest (configs.py PHYSICS.SLEEP_STEPS) and leaves the sweeps on a residual / stall exit.
This test compares the two at POSE level on the FP64 oracle: >= 256 whole env.step() calls from identical settled states
and identical actions, once with the shipped semantics and once with `SLEEP_STEPS = 0, SOLVER_TOL = 0, SOLVER_STALL = 0`
(nothing ever sleeps, 50 plain sweeps), against the SAME bounds `test_fp32_tolerance_at_the_end_of_a_push` states for
FP32 vs FP64: median body position difference <= 20 um, 90th percentile <= 0.3 mm, outcome flags (is_safe, is_effective)
agreeing on >= 97 % of the env steps.  (The tail is a few bodies that tumble one way or the other: contact add / remove
decisions are discontinuous.)  bench.py reports the same comparison as `deactivation.pose_equivalence`."""
import numpy as np
import pytest

from robovat_amd import configs, scenes

REFERENCE_LIKE = {'PHYSICS.SLEEP_STEPS': 0, 'PHYSICS.SOLVER_TOL': 0.0, 'PHYSICS.SOLVER_STALL': 0}
NO_DEACTIVATION = {'PHYSICS.SLEEP_STEPS': 0}


def _world(over, n, seed):
    from oracle import orc
    env_cfg = configs.push_env_config(**over)
    scene, names = scenes.make_scene(env_cfg=env_cfg)
    cfg = configs.make_rv_config(env_cfg=env_cfg, n_envs=n, seed=seed, shape_names=names)
    return orc.OracleWorld(cfg, scene, double=True)


def pose_equivalence(over, n=256, seed=21):
    """Shipped semantics vs `over` on the FP64 oracle: one env.step() per env from identical states / actions.
    Returns the per-body position differences (metres, active bodies) and the share of env steps whose flags agree."""
    a = _world({}, n, seed)
    a.reset()
    state, params = a.body_state(), a.body_params()
    a.set_body_state(state)                        # (both start from the same cleared manifolds, everything awake)
    b = _world(over, n, seed)
    b.reset(); b.set_body_params(params); b.set_body_state(state)
    act = a.policy_random(0)
    a.set_actions(act); b.set_actions(act)
    a.step_macro(); b.step_macro()
    on = params[:, :, 0] > 0
    perr = np.linalg.norm(a.body_state()[..., :3] - b.body_state()[..., :3], axis=-1)[on]
    moved = np.linalg.norm(a.body_state()[..., :3] - state[..., :3], axis=-1)[on]
    ca, cb = a.env_counters(), b.env_counters()
    agree = float(((ca[:, 5] == cb[:, 5]) & (ca[:, 6] == cb[:, 6])).mean())
    return {'median_m': float(np.median(perr)), 'p90_m': float(np.percentile(perr, 90)), 'p99_m': float(np.percentile(perr, 99)),
            'max_m': float(perr.max()), 'flags_agree': agree, 'env_steps': int(n), 'bodies_moved_share': float((moved > 1e-3).mean()),
            'awake_share_shipped': a.stats()['awake_substeps'] / max(1, a.stats()['substeps']),
            'awake_share_other': b.stats()['awake_substeps'] / max(1, b.stats()['substeps'])}


@pytest.mark.parametrize('name,over', [('no deactivation, 50 plain sweeps', REFERENCE_LIKE), ('no deactivation, shipped exits', NO_DEACTIVATION)])
def test_deactivation_is_within_the_stated_pose_tolerance(name, over):
    r = pose_equivalence(over)
    print('shipped vs %s (FP64 oracle, %d env.step()): median %.2e m, p90 %.2e m, p99 %.2e m, max %.2e m; flags agree %.3f; '
          'awake substeps %.3f vs %.3f' % (name, r['env_steps'], r['median_m'], r['p90_m'], r['p99_m'], r['max_m'], r['flags_agree'],
                                          r['awake_share_shipped'], r['awake_share_other']))
    assert r['awake_share_other'] == 1.0 and r['awake_share_shipped'] < 0.2      # (the two runs do differ in what they compute)
    assert r['bodies_moved_share'] > 0.05                                        # (and the pushes do move things)
    assert r['median_m'] <= 2e-5 and r['p90_m'] <= 3e-4 and r['flags_agree'] >= 0.97, r


def _episode_divergence(make_b, n, seed, steps):
    """FP64 oracle, shipped semantics, against world `make_b()` from identical settled states and identical actions over
    `steps` consecutive env.step() calls: per-horizon (median, p90, p99, max) body position difference, flag agreement."""
    a = _world({}, n, seed)
    a.reset()
    state, params = a.body_state(), a.body_params()
    a.set_body_state(state)
    b = make_b()
    b.reset(); b.set_body_params(params); b.set_body_state(state)
    on = params[:, :, 0] > 0
    rows, agree = [], []
    for k in range(steps):
        act = a.policy_random(k)
        a.set_actions(act); b.set_actions(act); a.step_macro(); b.step_macro()
        perr = np.linalg.norm(a.body_state()[..., :3] - b.body_state()[..., :3], axis=-1)[on]
        rows.append((float(np.median(perr)), float(np.percentile(perr, 90)), float(np.percentile(perr, 99)), float(perr.max())))
        ca, cb = a.env_counters(), b.env_counters()
        agree.append(float(((ca[:, 5] == cb[:, 5]) & (ca[:, 6] == cb[:, 6])).mean()))
    return rows, agree


def test_the_tail_and_the_horizon_of_the_deactivation_claim():
    """The one-step test above bounds median and p90; this one bounds the TAIL (p99) and the HORIZON (a 6-step episode).
    Contact add / remove decisions are discontinuous, so ANY perturbation of a push -- the last bit of a float included --
    grows over an episode: the yardstick is what FP32-vs-FP64 rounding alone does to the same 128 envs under the same
    actions (the tolerance `north_star` asks to state).  At every horizon the difference between the shipped deactivation
    and no deactivation + 50 plain sweeps stays below 3 x that yardstick at p99 and at the median (measured: 0.7 - 1.0 x at
    p99: 14 mm vs 14 mm after one step, 64 mm vs 76 mm after six), and the outcome flags agree on >= 95 % of the
    env steps at every horizon."""
    from oracle import orc
    n, seed, steps = 128, 21, 6

    def f32_world():
        env_cfg = configs.push_env_config()
        scene, names = scenes.make_scene(env_cfg=env_cfg)
        return orc.OracleWorld(configs.make_rv_config(env_cfg=env_cfg, n_envs=n, seed=seed, shape_names=names), scene, double=False)
    rounding, _ = _episode_divergence(f32_world, n, seed, steps)
    deact, agree = _episode_divergence(lambda: _world(REFERENCE_LIKE, n, seed), n, seed, steps)
    for k in range(steps):
        print('horizon %d env.step(): deactivation median %.2e p90 %.2e p99 %.2e max %.2e | FP32 rounding median %.2e p90 %.2e p99 %.2e max %.2e | '
              'flags agree %.3f' % ((k + 1,) + deact[k] + rounding[k] + (agree[k],)))
        assert deact[k][2] <= 3.0 * max(rounding[k][2], 1e-3), (k, deact[k], rounding[k])       # p99
        assert deact[k][0] <= 3.0 * max(rounding[k][0], 1e-5), (k, deact[k], rounding[k])       # median
        assert agree[k] >= 0.95, (k, agree[k])

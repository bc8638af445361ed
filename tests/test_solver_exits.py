"""The solver's early exits (residual 1e-5 N s, stall after 12 sweeps: configs.py) against Bullet's plain 50 sweeps
(`SOLVER_TOL` = 0, `SOLVER_STALL` = 0) on the float oracle: after one env.step() from the same reset state with the
same action the body poses must agree closely for most envs (contact-rich pushes diverge chaotically: the tail is
bounded loosely), and the outcome flags must agree for nearly all.  The step goldens are regenerated with the
solver they test, so a regression of the exits would be invisible to them -- this test is the bound."""
import numpy as np
import pytest

from robovat_amd import configs, scenes


def _run(over, n=96, seed=7):
    from oracle import orc
    env_cfg = configs.push_env_config(**over)
    scene, names = scenes.make_scene(env_cfg=env_cfg)
    cfg = configs.make_rv_config(env_cfg=env_cfg, n_envs=n, seed=seed, shape_names=names)
    w = orc.OracleWorld(cfg, scene, double=False)
    w.reset()
    p0 = w.body_state().copy()
    a = w.policy_random(0)
    w.set_actions(a); w.step_macro()
    return p0, w.body_state().copy(), w.env_flags() if hasattr(w, 'env_flags') else None, w.stats()


@pytest.mark.parametrize('name,over,med,p75', [('no exits at all', {'PHYSICS.SOLVER_STALL': 0, 'PHYSICS.SOLVER_TOL': 0.0}, 1.5e-4, 1e-3),
                                              ('no stall exit', {'PHYSICS.SOLVER_STALL': 0}, 5e-5, 3e-4)])
def test_early_exits_stay_close_to_plain_sweeps(name, over, med, p75):
    p0, a, _, sa = _run({})
    q0, b, _, sb = _run(over)
    assert np.median(np.abs(p0[..., :3] - q0[..., :3]).max((-1, -2))) < 5e-5          # same episode (the drop-and-settle of reset runs the solver too)
    dev = np.linalg.norm(a[..., :3] - b[..., :3], axis=-1).max(-1)  # per env: the body that deviates most, metres
    moved = np.linalg.norm(a[..., :2] - p0[..., :2], axis=-1).max(-1)
    assert (moved > 1e-3).mean() > 0.15                             # (the pushes do move things)
    assert np.median(dev) < med and np.percentile(dev, 75) < p75, (name, np.median(dev), np.percentile(dev, 75))
    for k in ('useful', 'unsafe', 'ineffective'):
        assert abs(sa[k] - sb[k]) <= 3, (name, k, sa[k], sb[k])     # of 96 env.step() calls

"""Host-side mirrors and the oracle against golden vectors generated from the
reference itself (tests/golden/gen_golden.py)."""
import json
import os

import numpy as np
import pytest

from robovat_amd import abi, configs
from robovat_amd.math import Pose, get_transform, rotations as R

GOLD = os.path.join(os.path.dirname(__file__), 'golden')


def _load(name):
    with open(os.path.join(GOLD, name)) as f:
        return json.load(f)


def test_rotation_kats_and_round_trips():
    g = _load('math_golden.json')
    kat = {k['fn']: k['out'] for k in g['kat']}
    # transformations.py doctest value (xyzw), re-verified in SURVEY.md §4
    assert np.allclose(kat['quaternion_from_euler_ryxz_1_2_3'], [0.310622, -0.718287, 0.444435, 0.435953], atol=1e-6)
    assert np.allclose(np.asarray(Pose([[.5, .1, .2], [np.pi, 0, .3]]).quaternion),
                       kat['pose_pi_0_0.3_quaternion'], atol=1e-6)
    for c in g['euler_quat']:
        q = R.quaternion_from_euler(*c['euler'])
        assert np.allclose(q, c['quat'], atol=1e-12)
        assert np.allclose(R.matrix3_from_quaternion(q), c['matrix3'], atol=1e-12)
        assert np.allclose(R.euler_from_matrix3(c['matrix3']), c['euler_back'], atol=1e-9)
        q2 = R.quaternion_from_matrix3(c['matrix3'])
        assert min(np.abs(q2 - q).max(), np.abs(q2 + q).max()) < 1e-9   # sign-free


def test_pose_inverse_transform_get_transform():
    for c in _load('math_golden.json')['pose_ops']:
        pa, pb = Pose(c['a']), Pose(c['b'])
        assert np.allclose(np.asarray(pa.quaternion), c['a_quat'], atol=2e-6)
        assert np.allclose(pa.matrix3, c['a_matrix3'], atol=2e-6)
        inv, tr, gt = pa.inverse(), pa.transform(pb), get_transform(source=pa, target=pb)
        assert np.allclose(np.asarray(inv.position), c['inv_pos'], atol=2e-6) and np.allclose(inv.matrix3, c['inv_m'], atol=2e-6)
        assert np.allclose(np.asarray(tr.position), c['tr_pos'], atol=2e-6) and np.allclose(tr.matrix3, c['tr_m'], atol=2e-6)
        assert np.allclose(np.asarray(gt.position), c['gt_pos'], atol=2e-6) and np.allclose(gt.matrix3, c['gt_m'], atol=2e-6)


def test_robust_matrix_to_quaternion_near_pi():
    """The reference's trace-only conversion collapses here (Appendix B-3)."""
    m = R.matrix3_from_euler(np.pi, 0.0, 0.3)
    q = R.quaternion_from_matrix3(m)
    assert abs(np.linalg.norm(q) - 1.0) < 1e-12
    assert np.allclose(R.matrix3_from_quaternion(q), m, atol=1e-12)


@pytest.mark.parametrize('which', ['numpy', 'oracle_f32', 'oracle_f64'])
def test_push_reward_all_layouts(which):
    from robovat_amd.reward_fns import push_reward
    from oracle import orc
    n = 0
    for entry in _load('reward_golden.json'):
        task, lid = entry['task'], entry['layout_id']
        fn = push_reward.get_reward_fn(task, lid)
        env_cfg = configs.push_env_config(TASK_NAME=task, LAYOUT_ID=lid)
        cfg = configs.make_rv_config(env_cfg=env_cfg, n_envs=1)
        for c in entry['cases']:
            s = np.asarray(c['state'], np.float32)[None]; nx = np.asarray(c['next_state'], np.float32)[None]
            if which == 'numpy':
                r, t = fn(s, nx); r, t = float(r[0]), bool(t[0])
            else:
                r, t = orc.eval_reward(cfg, s[0], nx[0], double=(which == 'oracle_f64'))
            assert t == c['termination'], (task, lid, c)
            assert abs(r - c['reward']) < 2e-5, (task, lid, r, c['reward'])   # float32 arithmetic in the reference
            n += 1
    assert n > 700


def test_compute_waypoints():
    from oracle import orc
    g = _load('waypoints_golden.json')
    cfg = configs.make_rv_config(n_envs=1)
    assert np.allclose(list(cfg.cspace_low), g['config']['cspace_low'])
    for c in g['cases']:
        s, e = orc.eval_waypoints(cfg, c['action'])
        assert np.allclose(s[:3], c['start_pos'], atol=1e-6) and np.allclose(e[:3], c['end_pos'], atol=1e-6)
        for got, want in ((s[3:], c['start_quat']), (e[3:], c['end_quat'])):
            want = np.asarray(want)
            assert min(np.abs(got - want).max(), np.abs(got + want).max()) < 1e-6   # euler [pi, 0, 0]


def test_heuristic_sampler_reproduces_reference_draws():
    from robovat_amd.envs.push.heuristic_push_sampler import HeuristicPushSampler
    c0 = configs.push_env_config()
    for c in _load('heuristic_golden.json'):
        sampler = HeuristicPushSampler(c0.ACTION.CSPACE.LOW, c0.ACTION.CSPACE.HIGH, 0.2, 0.2)
        np.random.seed(c['seed'])
        act = sampler.sample(np.asarray(c['position']), np.asarray(c['mask']), c['num_episodes'], c['num_steps'])
        assert np.allclose(act, c['action'], atol=1e-6)
        wp = sampler.get_waypoints(act[0, :2], act[0, 2:])
        assert np.allclose(np.asarray(wp, dtype=np.float64), c['waypoints'], atol=1e-6)


def test_wait_until_stable_step_counts():
    from oracle import orc
    scripts = {'always_still': lambda n: True, 'never_still': lambda n: False,
               'still_after_300': lambda n: n >= 300, 'ang_only_until_150': lambda n: n >= 150}
    for c in _load('wait_until_stable_golden.json'):
        mx = c['kwargs'].get('max_steps', 2000)
        stable = np.array([scripts[c['script']](k) for k in range(1, mx + 1)], dtype=np.uint8)
        assert orc.eval_wait_until_stable(stable, 100, 100, mx) == c['steps'], c


def test_layout_tables_loaded_into_config():
    env_cfg = configs.push_env_config(TASK_NAME='crossing', LAYOUT_ID=0)
    cfg = configs.make_rv_config(env_cfg=env_cfg, n_envs=1)
    assert cfg.task == abi.RV_TASK_CROSSING and cfg.n_region == 12 and cfg.n_goal == 1 and cfg.n_obstacle == 18
    assert [list(cfg.goal[0])] == [[1.0, 2.0]] and [list(cfg.target[0])] == [[2.0, 5.0]]

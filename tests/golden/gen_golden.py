"""Generate golden vectors by IMPORTING the reference (build container only).

    python tests/golden/gen_golden.py

Follows SURVEY.md Appendix E: stub `pybullet`, `cv2`, `gym`, `easydict`,
`matplotlib`; alias the numpy-1 names the reference still uses.  Only DATA
(inputs + the reference's outputs) is written to tests/golden/*.json — the
reference sources never travel.
"""
import json
import os
import sys
import types

import numpy as np

REF = '/root/reference'
OUT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REF)
for a, t in (('bool', bool), ('int', int), ('float', float)):
    if not hasattr(np, a):
        setattr(np, a, t)


class _Any(types.ModuleType):
    def __getattr__(self, k):
        if k.startswith('__'):
            raise AttributeError(k)
        return 0


for name in ('pybullet', 'cv2'):
    sys.modules[name] = _Any(name)
gym = types.ModuleType('gym'); gym.Env = object
spaces = types.ModuleType('gym.spaces')


class _Box(object):
    def __init__(self, low, high, shape=None, dtype=np.float32):
        self.low, self.high = np.asarray(low), np.asarray(high)


spaces.Box = _Box; spaces.Discrete = lambda n: n; spaces.Dict = dict
gym.spaces = spaces
sys.modules['gym'] = gym; sys.modules['gym.spaces'] = spaces
ed = types.ModuleType('easydict')


class EasyDict(dict):
    def __init__(self, d=None, **kw):
        super().__init__()
        for k, v in dict(d or {}, **kw).items():
            self[k] = EasyDict(v) if isinstance(v, dict) else v
    __getattr__ = dict.__getitem__
    __setattr__ = dict.__setitem__


ed.EasyDict = EasyDict
sys.modules['easydict'] = ed

# numpy>=2: transformations.py:1154,1348 call numpy.array(..., copy=False)
import third_party.transformations as T  # noqa: E402
_orig_array = np.array


def _array(*a, **k):
    if k.get('copy') is False:
        k.pop('copy')
        return np.asarray(*a, **k)
    return _orig_array(*a, **k)


T.numpy.array = _array

from robovat.math import Pose, get_transform  # noqa: E402
from robovat.reward_fns import push_reward  # noqa: E402
from robovat.envs.push import heuristic_push_sampler as hps  # noqa: E402
from robovat.envs.push import layouts  # noqa: E402


def main():
    rng = np.random.RandomState(0)
    # ---- (1) transformations KATs (doctest values) + random round trips
    math_g = {'kat': [], 'euler_quat': [], 'pose_ops': []}
    q = T.quaternion_from_euler(1, 2, 3, 'ryxz')
    math_g['kat'].append({'fn': 'quaternion_from_euler_ryxz_1_2_3', 'out': q.tolist()})
    for _ in range(200):
        e = rng.uniform([-3.1, -1.5, -3.1], [3.1, 1.5, 3.1])
        qq = T.quaternion_from_euler(*e)            # default axes sxyz
        m = T.matrix3_from_quaternion(qq)
        e2 = T.euler_from_matrix3(np.asarray(m, dtype=np.float64))
        math_g['euler_quat'].append({'euler': e.tolist(), 'quat': np.asarray(qq).tolist(),
                                     'matrix3': np.asarray(m).tolist(), 'euler_back': np.asarray(e2).tolist()})
    # ---- (2) Pose ops (avoid roll ~ pi through matrix->quat, Appendix B-3)
    for _ in range(100):
        a = [rng.uniform(-1, 1, 3).tolist(), rng.uniform([-1.2, -1.2, -3], [1.2, 1.2, 3]).tolist()]
        b = [rng.uniform(-1, 1, 3).tolist(), rng.uniform([-1.2, -1.2, -3], [1.2, 1.2, 3]).tolist()]
        pa, pb = Pose(a), Pose(b)
        inv = pa.inverse(); tr = pa.transform(pb); gt = get_transform(source=pa, target=pb)
        math_g['pose_ops'].append({
            'a': a, 'b': b,
            'a_quat': np.asarray(pa.quaternion).tolist(), 'a_matrix3': np.asarray(pa.matrix3).tolist(),
            'inv_pos': np.asarray(inv.position).tolist(), 'inv_m': np.asarray(inv.matrix3).tolist(),
            'tr_pos': np.asarray(tr.position).tolist(), 'tr_m': np.asarray(tr.matrix3).tolist(),
            'gt_pos': np.asarray(gt.position).tolist(), 'gt_m': np.asarray(gt.matrix3).tolist()})
    p = Pose([[.5, .1, .2], [np.pi, 0, .3]])
    math_g['kat'].append({'fn': 'pose_pi_0_0.3_quaternion', 'out': np.asarray(p.quaternion).tolist()})
    json.dump(math_g, open(os.path.join(OUT, 'math_golden.json'), 'w'))

    # ---- (4) push_reward for all tasks x layouts over random + edge states
    rew = []
    for task in ('clearing', 'insertion', 'crossing'):
        for lid in range(3):
            fn = push_reward.get_reward_fn(task, lid)
            layout = layouts.TASK_NAME_TO_LAYOUTS[task][lid]
            cases = []
            for k in range(80):
                s = np.stack([rng.uniform(0.2, 1.0, 4), rng.uniform(-0.6, 0.7, 4)], axis=-1)[None].astype(np.float32)
                n = (s + rng.normal(0, 0.08, s.shape)).astype(np.float32)
                if k % 5 == 0 and layout.goal is not None:      # put the target on / near the goal tile
                    g = np.array(layout.offset) + np.array(layout.goal[0]) * layout.size
                    n[0, 0] = g + rng.uniform(-0.09, 0.09, 2)
                if k % 7 == 0:                                   # tile-edge cases
                    t = np.array(layout.offset) + np.array(layout.region[k % len(layout.region)]) * layout.size
                    n[0, 0] = t + np.array([0.5 * layout.size, 0.0]) + rng.uniform(-1e-3, 1e-3, 2)
                if k % 11 == 0:                                  # padded (absent) bodies are zeros
                    s[0, 3] = 0; n[0, 3] = 0
                r, term = fn(s, n)
                cases.append({'state': s[0].tolist(), 'next_state': n[0].tolist(),
                              'reward': float(r[0]), 'termination': bool(term[0])})
            rew.append({'task': task, 'layout_id': lid, 'cases': cases})
    r, t = push_reward.get_reward_fn(None, 0)(np.zeros((1, 4, 2)), np.zeros((1, 4, 2)))
    rew.append({'task': None, 'layout_id': 0, 'cases': [{'state': np.zeros((4, 2)).tolist(), 'next_state': np.zeros((4, 2)).tolist(),
                                                         'reward': float(r[0]), 'termination': bool(t[0])}]})
    json.dump(rew, open(os.path.join(OUT, 'reward_golden.json'), 'w'))

    # ---- (3) PushEnv._compute_waypoints via the unbound reference function
    from robovat.envs.push import push_env
    cfg = EasyDict({'ACTION': {'CSPACE': {'LOW': [0.35, -0.35, 0.02], 'HIGH': [0.85, 0.35, 0.02]},
                               'MOTION': {'TRANSLATION_X': 0.2, 'TRANSLATION_Y': 0.2}},
                    'ARM': {'FINGER_TIP_OFFSET': 0.14}})
    low = np.array(cfg.ACTION.CSPACE.LOW, dtype=np.float32); high = np.array(cfg.ACTION.CSPACE.HIGH, dtype=np.float32)
    fake = types.SimpleNamespace(config=cfg, cspace=_Box(low, high),
                                 start_offset=0.5 * (high + low), start_range=0.5 * (high - low),
                                 start_z=cfg.ARM.FINGER_TIP_OFFSET + 0.5 * (high + low)[2])
    way = {'config': {'cspace_low': low.tolist(), 'cspace_high': high.tolist(), 'translation_x': 0.2,
                      'translation_y': 0.2, 'finger_tip_offset': 0.14}, 'cases': []}
    grid = [rng.uniform(-1, 1, 4) for _ in range(60)] + [np.array([1, 1, 1, 1.]), np.array([-1, -1, -1, -1.]), np.zeros(4)]
    for a in grid:
        s, e = push_env.PushEnv._compute_waypoints(fake, a.astype(np.float32))
        way['cases'].append({'action': a.tolist(), 'start_pos': np.asarray(s.position).tolist(), 'start_quat': np.asarray(s.quaternion).tolist(),
                             'end_pos': np.asarray(e.position).tolist(), 'end_quat': np.asarray(e.quaternion).tolist()})
    json.dump(way, open(os.path.join(OUT, 'waypoints_golden.json'), 'w'))

    # ---- (5) HeuristicPushSampler under np.random.seed(k)
    heur = []
    for k in range(12):
        sampler = hps.HeuristicPushSampler(low, high, 0.2, 0.2, max_attemps=20000)
        pos = np.stack([rng.uniform(0.4, 0.8, 4), rng.uniform(-0.3, 0.3, 4), np.full(4, 0.03)], axis=-1)
        mask = np.array([1, 1, 1, 1 if k % 2 else 0], dtype=np.float32)
        np.random.seed(k)
        act = sampler.sample(pos, mask, num_episodes=k, num_steps=k % 3)
        wp = sampler.get_waypoints(act[0, :2], act[0, 2:])
        heur.append({'seed': k, 'position': pos.tolist(), 'mask': mask.tolist(), 'num_episodes': k, 'num_steps': k % 3,
                     'action': act.tolist(), 'waypoints': np.asarray(wp, dtype=np.float64).tolist(),
                     'clear_start': bool(sampler.is_waypoint_clear(wp[0], None, pos[:int(mask.sum())], 0.05))})
    json.dump(heur, open(os.path.join(OUT, 'heuristic_golden.json'), 'w'))

    # ---- (9) wait_until_stable substep counts under scripted velocities
    from robovat.simulation.simulator import Simulator
    wus = []

    def run(script, **kw):
        sim = Simulator.__new__(Simulator)
        state = {'n': 0}

        class B(object):
            linear_velocity = property(lambda s: np.array([script(state['n'])[0], 0, 0]))
            angular_velocity = property(lambda s: np.array([script(state['n'])[1], 0, 0]))
        sim.step = lambda: state.__setitem__('n', state['n'] + 1)
        Simulator.wait_until_stable(sim, B(), **kw)
        return state['n']
    scripts = {'always_still': lambda n: (0.0, 0.0), 'never_still': lambda n: (1.0, 0.0),
               'still_after_300': lambda n: (1.0, 0.0) if n < 300 else (0.0, 0.0),
               'ang_only_until_150': lambda n: (0.0, 1.0) if n < 150 else (0.0, 0.0)}
    for name, sc in scripts.items():
        wus.append({'script': name, 'kwargs': {}, 'steps': run(sc)})
        wus.append({'script': name, 'kwargs': {'linear_velocity_threshold': 0.1, 'angular_velocity_threshold': 0.1, 'max_steps': 500},
                    'steps': run(sc, linear_velocity_threshold=0.1, angular_velocity_threshold=0.1, max_steps=500)})
    json.dump(wus, open(os.path.join(OUT, 'wait_until_stable_golden.json'), 'w'))
    print('golden vectors written to', OUT)


if __name__ == '__main__':
    main()

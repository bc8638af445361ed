"""Push-task reward functions, host (numpy) edition.

Same call signature as ``robovat/reward_fns/push_reward.py:272-405``:
``get_reward_fn(task_name, layout_id, ...)(state, next_state) -> (reward,
termination)`` on ``[batch, bodies, 2]`` arrays (or observation dicts).  The
device computes the identical quantity inside ``rv_step_macro``; this module is
for API users and for the golden-vector tests.  The behaviour pinned by the
goldens includes the reference's quirks that downstream users see
(SURVEY.md Appendix B-8): the clearing score's overwritten minimum, the
all-False insertion termination outside planning mode, and tile centres that
move with the ``size`` argument of ``check_on_tiles``.
"""
import numpy as np

from robovat_amd.envs.push import push_layouts

CLEARING_GOAL_X = 0.7
CLEARING_GOAL_Y = 0.9


def process_state(state):
    if isinstance(state, dict):
        if 'position' in state:
            state = np.asarray(state['position'])[..., :2]
        else:
            state = np.mean(np.asarray(state['point_cloud']), axis=-2)[..., :2]
        if state.ndim == 2:
            state = state[np.newaxis]
    state = np.asarray(state)
    assert state.shape[-1] == 2
    return state


def _tile_centres(centers, size, offset):
    return np.asarray(offset, dtype=np.float64) + np.asarray(centers, dtype=np.float64) * size


def check_on_tiles(position, centers, size, offset, max_dist=None):
    max_dist = size if max_dist is None else max_dist
    d = np.abs(position[:, None, :] - _tile_centres(centers, size, offset)[None])
    return np.any((d[..., 0] <= 0.5 * max_dist) & (d[..., 1] <= 0.5 * max_dist), axis=1)


def get_tile_dists(position, centers, size, offset):
    d = position[:, None, :] - _tile_centres(centers, size, offset)[None]
    return np.linalg.norm(d, axis=-1).min(axis=1)


def _score(task, state, layout):
    if task == 'clearing':
        d1 = np.mean(np.abs(state[:, :, 0] - CLEARING_GOAL_X), axis=1)
        d3 = np.mean(np.abs(state[:, :, 1] + CLEARING_GOAL_Y), axis=1)
        return -np.minimum(d1, d3)
    return -get_tile_dists(state[:, 0, :], layout.goal, layout.size, layout.offset)


def _goal(task, state, layout):
    if task == 'clearing':
        ok = np.ones(state.shape[0], dtype=bool)
        for i in range(state.shape[1]):
            ok &= ~check_on_tiles(state[:, i, :], layout.region, layout.size * 1.25, layout.offset)
        return ok
    return check_on_tiles(state[:, 0, :], layout.goal, layout.size, layout.offset)


def _termination(task, next_state, layout):
    if task == 'crossing':
        return ~check_on_tiles(next_state[:, 0, :], layout.region, layout.size, layout.offset,
                               max_dist=layout.size * 1.5)
    return np.zeros(next_state.shape[0], dtype=bool)


def dummy_reward_fn(state, next_state):
    state = process_state(state)
    shape = () if state.ndim == 2 else (state.shape[0],)
    return np.ones(shape, dtype=np.float32), np.zeros(shape, dtype=bool)


def get_reward_fn(task_name, layout_id, goal_reward=100.0, termination_reward=-100.0,
                  dense_reward=1.0, time_reward=-1.0, use_dense_reward=True,
                  use_time_penalty=True, is_planning=False, is_high_level=False):
    if task_name is None or task_name == 'data_collection':
        return dummy_reward_fn
    if task_name not in push_layouts.TASK_NAME_TO_LAYOUTS:
        raise ValueError('Unrecognized manipulation task: %r' % task_name)
    if is_planning:
        raise NotImplementedError('planning-mode checks are outside the env.step() path')
    layout = push_layouts.TASK_NAME_TO_LAYOUTS[task_name][layout_id]

    def reward_fn(state, next_state):
        state, next_state = process_state(state), process_state(next_state)
        termination = _termination(task_name, next_state, layout)
        goal = _goal(task_name, next_state, layout) & ~termination
        reward = np.zeros(state.shape[0], dtype=np.float32)
        reward += goal_reward * goal.astype(np.float32)
        reward += termination_reward * (termination & ~goal).astype(np.float32)
        if use_dense_reward:
            reward += (np.abs(_score(task_name, next_state, layout) - _score(task_name, state, layout))
                       * dense_reward).astype(np.float32)
        if use_time_penalty:
            reward += time_reward
        return reward, termination | goal

    return reward_fn


class PushReward(object):
    """Reward function object with the reference's interface (push_reward.py:377-405)."""

    def __init__(self, name, task_name, layout_id, is_planning=False):
        self.name = name
        self.env = None
        self.reward_fn = get_reward_fn(task_name=task_name, layout_id=layout_id, is_planning=is_planning)

    def initialize(self, env):
        self.env = env

    def on_episode_start(self):
        pass

    def on_episode_end(self):
        pass

    def get_reward(self):
        assert self.env.prev_obs_data is not None and self.env.obs_data is not None
        reward, termination = self.reward_fn(self.env.prev_obs_data, self.env.obs_data)
        return float(np.asarray(reward).reshape(-1)[0]), bool(np.asarray(termination).reshape(-1)[0])

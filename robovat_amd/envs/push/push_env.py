"""PushEnv on the MI355X backend.

``VecPushEnv`` is the fast path: N envs advance together, ``step(actions)`` is
one ``rv_step_macro`` launch that runs the whole ``_execute_action`` phase
machine, the settle, the observations and the reward on the device
(reference: ``robovat/envs/push/push_env.py:599-937``,
``robovat/envs/robot_env.py:204-312``).  ``PushEnv`` keeps the reference's
single-env gym-style API (``reset() -> obs``, ``step(a) -> obs, reward, done,
None``, same observation keys / dtypes / shapes, ``push_env.py:169-267``) as a
batch of one.
"""
import collections

import numpy as np

from robovat_amd import abi, configs, scenes


class Box(object):
    """Stand-in for ``gym.spaces.Box`` (gym is not a dependency)."""

    def __init__(self, low, high, shape=None, dtype=np.float32):
        self.low = np.broadcast_to(np.asarray(low, dtype=dtype), shape if shape is not None else np.shape(low)).copy()
        self.high = np.broadcast_to(np.asarray(high, dtype=dtype), self.low.shape).copy()
        self.shape, self.dtype = self.low.shape, dtype

    def sample(self):
        return np.random.uniform(self.low, self.high).astype(self.dtype)

    def contains(self, x):
        x = np.asarray(x)
        return x.shape == self.shape and bool(np.all(x >= self.low) and np.all(x <= self.high))


class VecPushEnv(object):
    """N PushEnv instances on one GPU (env shards [offset, offset + N))."""

    def __init__(self, num_envs, config=None, robot_config=None, device=0, seed=0,
                 env_id_offset=0, use_point_cloud=False, physics=None):
        from robovat_amd import lib
        self.config = config or configs.push_env_config()
        self.robot_config = robot_config or configs.sawyer_config()
        self._owns_world = physics is None
        self._physics = physics
        if physics is None:
            self.scene, self.shape_names = scenes.make_scene()
            self.rv_config = configs.make_rv_config(self.config, self.robot_config, self.shape_names,
                                                    n_envs=num_envs, env_id_offset=env_id_offset, seed=seed)
            self.world = lib.World(self.rv_config, self.scene, device=device)
        else:
            # the env runs on the world of a Simulator's physics backend (HipPhysics): what the env
            # does is visible through the Simulator / Body API and the other way round
            assert int(num_envs) == 1, 'a Simulator holds one env'
            self._physics = physics
            physics.configure(self.config, self.robot_config, seed=seed, worker_id=env_id_offset)
            self.scene, self.shape_names = physics.scene, physics.shape_names
            self.world, self.rv_config = physics.world, physics.rv_config
        self.num_envs = int(num_envs)
        self.max_movable_bodies = abi.RV_MAXB
        self.use_point_cloud = bool(use_point_cloud)
        g = self.rv_config.num_goal_steps
        self.action_shape = (4,) if g == 0 else (g, 4)
        self.action_space = Box(-1.0, 1.0, self.action_shape)
        self._macro_index = 0

    @property
    def device(self):
        return self.world.device

    def get_observation(self):
        return self.world.observe(point_cloud=self.use_point_cloud)

    def camera_calibration(self):
        """(intrinsics [N, 3, 3], translation [N, 3], rotation [N, 3, 3]) of the simulated Kinect2 of every env: the
        configured calibration plus the uniform noise ArmEnv._reset_camera draws at each reset (arm_env.py:109-152;
        KINECT2.DEPTH.INTRINSICS_NOISE / TRANSLATION_NOISE / ROTATION_NOISE, push_env.py:273-280)."""
        cam = self.world.camera().cpu().numpy()
        k = np.zeros((self.num_envs, 3, 3), np.float32)
        k[:, 0, 0], k[:, 1, 1], k[:, 0, 2], k[:, 1, 2], k[:, 0, 1], k[:, 2, 2] = cam[:, 0], cam[:, 1], cam[:, 2], cam[:, 3], cam[:, 4], 1.0
        return k, cam[:, 14:17].copy(), cam[:, 5:14].reshape(-1, 3, 3).copy()

    def reset(self, mask=None):
        """RobotEnv.reset for every env (or the masked ones)."""
        self.world.reset(mask)
        if self._physics is not None:
            self._physics.on_env_reset()
        return self.get_observation()

    def step(self, actions):
        """RobotEnv.step for every env whose episode is not done."""
        self.world.set_actions(actions)
        self.world.step_macro()
        self._macro_index += 1
        obs = self.get_observation()
        reward, done = self.world.reward()
        return obs, reward, done.bool(), None

    def sample_random_actions(self):
        """RandomPolicy on the device (Philox keyed by global env id and step)."""
        a = self.world.policy_random(self._macro_index)
        return a.reshape((self.num_envs,) + self.action_shape)

    def sample_heuristic_actions(self, max_attempts=20000):
        a = self.world.policy_heuristic(max_attempts)
        return a.reshape((self.num_envs,) + self.action_shape)

    def rollout(self, n_steps, auto_reset=True, record=True):
        before = self.world.env_counters().cpu().numpy()[:, [2, 4]] if (auto_reset and self._physics is not None) else None
        out = self.world.rollout(n_steps, self._macro_index, auto_reset, record)
        if before is not None:
            # the host mirror of the user constraints is dropped only when an env really was reset
            # -- every episode end is followed by a reset at the env's NEXT step, so resets = (done before) + (episodes
            # ended) - (done after): an episode that merely ends on the last step of the rollout has not been reset yet,
            # its constraints are still active on the device and the mirror must keep them
            after = self.world.env_counters().cpu().numpy()[:, [2, 4]]
            resets = (before[:, 1] != 0).astype(int) + (after[:, 0] - before[:, 0]) - (after[:, 1] != 0).astype(int)
            if (resets > 0).any():
                self._physics.on_env_reset()
        self._macro_index += int(n_steps)
        return out

    def stats(self):
        return self.world.stats()

    def close(self):
        if self._owns_world:
            self.world.close()


class PushEnv(object):
    """Single-env PushEnv with the reference's API (a VecPushEnv of one)."""

    def __init__(self, simulator=None, config=None, debug=False, robot_config=None, device=0, seed=0,
                 worker_id=0):
        self._config = config or configs.push_env_config()
        self._debug = debug
        # PushEnv(simulator, config) as in the reference (push_env.py:43-48): the env lives in the
        # simulator's physics backend -- bodies, constraints and getters of that Simulator see it
        self._simulator = simulator
        physics = None
        if simulator is not None:
            physics = simulator.physics
            if not hasattr(physics, 'configure'):
                raise ValueError('PushEnv needs a Simulator whose physics backend is HipPhysics')
            if simulator.bodies or simulator.constraints:
                # configure() recreates the world: bodies / constraints added before would silently vanish
                raise ValueError('PushEnv(simulator): the Simulator already holds %d bodies / %d constraints; give the env '
                                 'an empty Simulator' % (len(simulator.bodies), len(simulator.constraints)))
        self._vec = VecPushEnv(1, self._config, robot_config, device=device, seed=seed, env_id_offset=worker_id,
                               use_point_cloud=True, physics=physics)
        self.max_movable_bodies = abi.RV_MAXB
        self.task_name = self._config.TASK_NAME
        self.layout_id = self._config.LAYOUT_ID
        self.num_goal_steps = self._config.NUM_GOAL_STEPS
        self.action_space = self._vec.action_space
        self.phase_list = list(abi.PHASES)
        self._obs_data = self._prev_obs_data = None
        self._done = True
        self._episode_reward = self._total_reward = 0.0

    config = property(lambda s: s._config)
    debug = property(lambda s: s._debug)
    simulator = property(lambda s: s._simulator)
    is_simulation = property(lambda s: True)
    obs_data = property(lambda s: s._obs_data)
    prev_obs_data = property(lambda s: s._prev_obs_data)
    done = property(lambda s: s._done)
    episode_reward = property(lambda s: s._episode_reward)
    total_reward = property(lambda s: s._total_reward)
    info = property(lambda s: {'name': 'PushEnv'})

    def _counters(self):
        return self._vec.world.env_counters().cpu().numpy()[0]

    num_steps = property(lambda s: int(s._counters()[1]))
    num_episodes = property(lambda s: int(s._counters()[2]))

    def _convert(self, obs):
        out = collections.OrderedDict()
        for key in ('num_episodes', 'num_steps', 'layout_id'):
            out[key] = np.array(obs[key][0].item(), dtype=np.int64)
        out['body_mask'] = obs['body_mask'][0].cpu().numpy().astype(np.float32)
        out['point_cloud'] = obs['point_cloud'][0].cpu().numpy().astype(np.float32)
        if self._config.USE_PRESTIGE_OBS:
            out['position'] = obs['position'][0].cpu().numpy().astype(np.float32)
            out['is_safe'] = np.array(obs['is_safe'][0].item(), dtype=np.int64)
            out['is_effective'] = np.array(obs['is_effective'][0].item(), dtype=np.int64)
        return out

    def reset(self):
        self._prev_obs_data = None
        if self._simulator is not None:
            # RobotEnv.reset -> simulator.reset() (robot_env.py:204-237): the episode's Body / Constraint wrappers go
            self._simulator._bodies.clear(); self._simulator._constraints.clear()
        self._obs_data = self._convert(self._vec.reset())
        self._done = False
        self._episode_reward = 0.0
        return self._obs_data

    def step(self, action):
        if self._done:
            raise ValueError('The environment is done. Forget to reset?')
        action = np.asarray(action, dtype=np.float32).reshape((1,) + self._vec.action_shape)
        obs, reward, done, _ = self._vec.step(action)
        self._prev_obs_data, self._obs_data = self._obs_data, self._convert(obs)
        reward = float(reward[0].item())
        self._done = bool(done[0].item())
        self._episode_reward += reward
        if self._done:
            self._total_reward += self._episode_reward
        return self._obs_data, reward, self._done, None

    def get_observation(self):
        return self._obs_data

    def close(self):
        self._vec.close()

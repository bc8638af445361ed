"""Grasp4DofEnv on the MI355X backend (BASELINE.json configs[3]).

``VecGrasp4DofEnv``: N envs, ``step(actions)`` is one ``rv_step_macro`` launch running the whole
``_execute_action`` phase machine (overhead -> prestart -> straight-line descent -> close -> lift;
``robovat/envs/grasp/grasp_4dof_env.py:213-345``) with the force-limited gripper in the contact
solver, the depth observation and ``GraspReward`` (``reward_fns/grasp_reward.py:49-68``).
``Grasp4DofEnv`` keeps the reference's single-env API as a batch of one.  Actions are
``[x, y, z, angle]`` in the world (``ACTION.TYPE == 'CUBOID'``) or ``[x1, y1, x2, y2, depth]`` in
the depth image (``'IMAGE'``, converted on the host by ``Grasp2D``).
"""
import collections

import numpy as np

from robovat_amd import abi, configs, scenes
from robovat_amd.envs.grasp.grasp_2d import Grasp2D
from robovat_amd.envs.push.push_env import Box
from robovat_amd.perception import Camera


class VecGrasp4DofEnv(object):

    def __init__(self, num_envs, config=None, robot_config=None, device=0, seed=0, env_id_offset=0):
        from robovat_amd import lib
        self.config = config or configs.grasp_env_config()
        self.robot_config = robot_config or configs.sawyer_config()
        self.scene, self.shape_names = scenes.make_scene(env_cfg=self.config)
        self.rv_config = configs.make_rv_config(self.config, self.robot_config, self.shape_names,
                                                n_envs=num_envs, env_id_offset=env_id_offset, seed=seed)
        self.world = lib.World(self.rv_config, self.scene, device=device)
        self.num_envs = int(num_envs)
        c = self.rv_config
        fx, fy, cx, cy, sk = list(c.cam_intrinsics)
        self.camera = Camera(height=c.cam_height, width=c.cam_width, intrinsics=[[fx, sk, cx], [0, fy, cy], [0, 0, 1]],
                             translation=list(c.cam_translation), rotation=np.array(list(c.cam_rotation)).reshape(3, 3))
        if self.config.ACTION.TYPE == 'CUBOID':
            self.action_space = Box(np.array(list(self.config.ACTION.CUBOID.LOW) + [0.0], np.float32),
                                    np.array(list(self.config.ACTION.CUBOID.HIGH) + [2 * np.pi], np.float32))
        elif self.config.ACTION.TYPE == 'IMAGE':
            w, h = c.cam_width, c.cam_height
            self.action_space = Box(np.array([0, 0, 0, 0, -(2 * 24 - 1)], np.float32), np.array([w, h, w, h, 2 * 24 - 1], np.float32))
        else:
            raise ValueError('Unrecognized action type: %r' % (self.config.ACTION.TYPE,))
        self._macro_index = 0

    device = property(lambda s: s.world.device)

    def get_observation(self):
        """CameraObs(OBSERVATION.TYPE = 'depth') + the camera calibration (grasp_4dof_env.py:97-115)."""
        depth, _ = self.world.render()
        k, t, r = self.camera_calibration()
        return {'depth': depth, 'intrinsics': k, 'translation': t, 'rotation': r}

    def camera_calibration(self):
        """(intrinsics [N, 3, 3], translation [N, 3], rotation [N, 3, 3]) each env is rendered with: the configured
        calibration plus the noise of its last reset (ArmEnv._reset_camera, arm_env.py:109-152; rv_get_camera)."""
        cam = self.world.camera().cpu().numpy()
        k = np.zeros((self.num_envs, 3, 3), np.float32)
        k[:, 0, 0], k[:, 1, 1], k[:, 0, 2], k[:, 1, 2], k[:, 0, 1], k[:, 2, 2] = cam[:, 0], cam[:, 1], cam[:, 2], cam[:, 3], cam[:, 4], 1.0
        return k, cam[:, 14:17].copy(), cam[:, 5:14].reshape(-1, 3, 3).copy()

    def reset(self, mask=None):
        self.world.reset(mask)
        return self.get_observation()

    def _to_4dof(self, actions):
        if self.config.ACTION.TYPE == 'CUBOID':
            return actions
        a = np.asarray(actions.cpu() if hasattr(actions, 'cpu') else actions, np.float64).reshape(self.num_envs, 5)
        return np.array([Grasp2D.from_vector(v, camera=self.camera).as_4dof() for v in a], np.float32)

    def step(self, actions):
        self.world.set_actions(self._to_4dof(actions))
        self.world.step_macro()
        self._macro_index += 1
        obs = self.get_observation()
        reward, done = self.world.reward()
        return obs, reward, done.bool(), None

    def sample_random_actions(self):
        """RandomPolicy = action_space.sample() on the device (uniform in ACTION.CUBOID x [0, 2 pi))."""
        return self.world.policy_random(self._macro_index).reshape(self.num_envs, 4)

    def rollout(self, n_steps, auto_reset=True, record=True):
        out = self.world.rollout(n_steps, self._macro_index, auto_reset, record)
        self._macro_index += int(n_steps)
        return out

    def stats(self):
        return self.world.stats()

    def close(self):
        self.world.close()


class Grasp4DofEnv(object):
    """Single-env Grasp4DofEnv with the reference's API (a VecGrasp4DofEnv of one)."""

    def __init__(self, simulator=None, config=None, debug=False, robot_config=None, device=0, seed=0, worker_id=0):
        self._config = config or configs.grasp_env_config()
        self._debug, self._simulator = debug, simulator
        self._vec = VecGrasp4DofEnv(1, self._config, robot_config, device=device, seed=seed, env_id_offset=worker_id)
        self.camera = self._vec.camera
        self.action_space = self._vec.action_space
        self._obs_data = None
        self._done = True
        self._episode_reward = self._total_reward = 0.0

    config = property(lambda s: s._config)
    debug = property(lambda s: s._debug)
    simulator = property(lambda s: s._simulator)
    is_simulation = property(lambda s: True)
    done = property(lambda s: s._done)
    episode_reward = property(lambda s: s._episode_reward)
    total_reward = property(lambda s: s._total_reward)
    obs_data = property(lambda s: s._obs_data)
    num_steps = property(lambda s: int(s._vec.world.env_counters().cpu().numpy()[0, 1]))
    num_episodes = property(lambda s: int(s._vec.world.env_counters().cpu().numpy()[0, 2]))

    def _convert(self, obs):
        out = collections.OrderedDict()
        out[self._config.OBSERVATION.TYPE] = obs['depth'][0].cpu().numpy()
        for key in ('intrinsics', 'translation', 'rotation'):
            out[key] = obs[key][0]
        return out

    def reset(self):
        self._obs_data = self._convert(self._vec.reset())
        self._done = False
        self._episode_reward = 0.0
        return self._obs_data

    def step(self, action):
        if self._done:
            raise ValueError('The environment is done. Forget to reset?')
        n = 4 if self._config.ACTION.TYPE == 'CUBOID' else 5
        obs, reward, done, _ = self._vec.step(np.asarray(action, np.float32).reshape(1, n))
        self._obs_data = self._convert(obs)
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

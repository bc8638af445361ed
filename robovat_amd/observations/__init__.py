"""Observation classes with the reference's interface (``robovat/observations/*.py``): an
``Observation`` is initialised with the env, exposes ``get_gym_space()`` and
``get_observation()``.  On this backend every observation is a read of device buffers the
kernels fill (``rv_observe`` / ``rv_render`` / ``rv_render_rgb``); the classes below are the
name-compatible handles for code written against the reference.

* ``CameraObs(modality='rgb' | 'depth' | 'segmask')``      camera_obs.py:33-88
* ``SegmentedPointCloudObs``                              camera_obs.py:182-238
* ``PoseObs(modality='position' | 'pose' | 'pose2d' | 'yaw_cossin')``   pose_obs.py:17-73
"""
import numpy as np


class _Box(object):
    def __init__(self, low, high, shape, dtype):
        self.low, self.high, self.shape, self.dtype = low, high, tuple(shape), dtype


class Observation(object):
    """observation.py:12-49"""

    def __init__(self, name=None):
        self.name = name
        self.env = None

    def initialize(self, env):
        self.env = env

    def on_episode_start(self):
        pass

    @property
    def world(self):
        return self.env.world


class CameraObs(Observation):
    """Image of the simulated Kinect2 (camera_obs.py:33-88): 'rgb' uint8 [H, W, 3], 'depth' float32
    [H, W, 1] (metres, 0 where nothing is hit), 'segmask' uint8 [H, W, 1] (body index, RV_MAXB + 1 = arm, RV_MAXB =
    table, 255 = nothing).  ``env_index``: which env of a vectorised env (images are per env)."""

    def __init__(self, camera=None, modality='rgb', max_visible_distance_m=None, name=None, env_index=0):
        Observation.__init__(self, name=name or modality)
        if modality not in ('rgb', 'depth', 'segmask'):
            raise ValueError('Unrecognized modality: %r.' % (modality,))
        self.camera, self.modality, self.env_index = camera, modality, int(env_index)
        self.max_visible_distance_m = max_visible_distance_m or 10.0

    def _hw(self):
        c = self.world.cfg
        return int(c.cam_height), int(c.cam_width)

    def get_gym_space(self):
        h, w = self._hw()
        if self.modality == 'rgb':
            return _Box(0, 255, (h, w, 3), np.uint8)
        if self.modality == 'depth':
            return _Box(0.0, self.max_visible_distance_m, (h, w, 1), np.float32)
        return _Box(0, 255, (h, w, 1), np.uint8)

    def get_observation(self):
        i = self.env_index
        if self.modality == 'rgb':
            return self.world.render_rgb()[i].cpu().numpy()
        depth, seg = self.world.render(segmask=True)
        if self.modality == 'depth':
            return depth[i].cpu().numpy()[..., None]
        return seg[i].cpu().numpy()[..., None]


class SegmentedPointCloudObs(Observation):
    """camera_obs.py:182-238: [num_bodies, num_points, 3] per env, from ``rv_observe``."""

    def __init__(self, camera=None, num_points=None, num_bodies=None, crop_min=None, crop_max=None, name=None, env_index=0):
        Observation.__init__(self, name=name or 'point_cloud')
        self.env_index = int(env_index)

    def get_gym_space(self):
        from robovat_amd import abi
        return _Box(-np.inf, np.inf, (abi.RV_MAXB, int(self.world.cfg.num_points), 3), np.float32)

    def get_observation(self):
        return self.world.observe(point_cloud=True)['point_cloud'][self.env_index].cpu().numpy()


class PoseObs(Observation):
    """pose_obs.py:17-73."""

    def __init__(self, num_bodies=None, modality='position', name=None, env_index=0):
        Observation.__init__(self, name=name or modality)
        if modality not in ('position', 'pose', 'pose2d', 'yaw_cossin'):
            raise ValueError('Unrecognized modality: %r.' % (modality,))
        self.modality, self.env_index = modality, int(env_index)

    def get_gym_space(self):
        from robovat_amd import abi
        k = {'position': 3, 'pose': 6, 'pose2d': 3, 'yaw_cossin': 2}[self.modality]
        return _Box(-np.inf, np.inf, (abi.RV_MAXB, k), np.float32)

    def get_observation(self):
        return self.world.observe(pose_modes=True)[self.modality][self.env_index].cpu().numpy()

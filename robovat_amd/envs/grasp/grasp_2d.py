"""Parallel-jaw grasp in image space (``ACTION.TYPE == 'IMAGE'`` of Grasp4DofEnv).

Same quantities as ``robovat/envs/grasp/grasp_2d.py:14-159`` -- centre (x, y) in the image,
angle of the grasp axis with the camera x axis, depth of the centre, opening width in metres --
and the same conversions: ``from_vector([x1, y1, x2, y2, depth])`` (jaw end points) and
``as_4dof()`` -> ``[x, y, z, angle]`` in the world.  Pinned by
``tests/golden/camera_golden.json`` (section ``grasp2d``, generated from the reference).
"""
import numpy as np

from robovat_amd.math import Pose, get_transform


class Grasp2D(object):

    def __init__(self, center, angle, depth, width=0.0, camera=None):
        self.center, self.angle, self.depth, self.width, self.camera = center, angle, depth, width, camera

    @property
    def axis(self):
        return np.array([np.cos(self.angle), np.sin(self.angle)])

    @property
    def width_pixel(self):
        """Opening width in pixels at the grasp depth."""
        self._need_camera()
        u1 = self.camera.project_point(np.array([0, 0, self.depth]), is_world_frame=False)
        u2 = self.camera.project_point(np.array([self.width, 0, self.depth]), is_world_frame=False)
        return np.linalg.norm(u1 - u2)

    @property
    def endpoints(self):
        half = 0.5 * float(self.width_pixel) * self.axis
        return self.center - half, self.center + half

    @property
    def vector(self):
        p1, p2 = self.endpoints
        return np.r_[p1, p2, self.depth]

    def _need_camera(self):
        if self.camera is None:
            raise ValueError('Must specify camera intrinsics.')

    @property
    def pose(self):
        """Grasp frame in the camera frame: y along the grasp axis, x along the optical axis."""
        self._need_camera()
        centre = self.camera.deproject_pixel(self.center, self.depth, is_world_frame=False)
        y = np.array([self.axis[0], self.axis[1], 0.0])
        y /= np.linalg.norm(y)
        z = np.cross(np.array([0.0, 0.0, 1.0]), y)
        x = np.cross(z, y)
        rot = np.array([x, y, z]).T
        if np.linalg.det(rot) < 0:
            rot[:, 0] = -rot[:, 0]
        return Pose([centre, rot])

    @staticmethod
    def from_vector(value, camera=None):
        value = np.asarray(value, dtype=np.float64)
        p1, p2, depth = value[:2], value[2:4], value[4]
        # (the reference swaps the coordinates of the end points when it measures the width)
        q1 = camera.deproject_pixel(np.array([p1[1], p1[0]]), depth, is_world_frame=False)
        q2 = camera.deproject_pixel(np.array([p2[1], p2[0]]), depth, is_world_frame=False)
        axis = p2 - p1
        return Grasp2D((p1 + p2) / 2, np.arctan2(axis[1], axis[0]), depth, np.linalg.norm(q1 - q2), camera)

    def as_4dof(self):
        """[x, y, z, angle] of the grasp in the world frame."""
        in_camera = Pose([self.pose.position, [0, 0, self.angle + np.pi / 2]])
        in_world = get_transform(source=self.camera.pose).transform(in_camera)
        x, y, z = in_world.position
        return [x, y, z, in_world.euler[2]]

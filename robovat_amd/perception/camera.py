"""Pinhole camera with the reference's conventions.

``Camera`` mirrors ``robovat/perception/camera/camera.py:19-244`` (extrinsics are
the world origin in the camera frame: ``x_cam = rotation @ x_world +
translation``; ``pose`` is the camera in the world) and
``intrinsic_to_projection_matrix`` mirrors
``robovat/simulation/camera/bullet_camera.py:28-83``.  Pinned by
``tests/golden/camera_golden.json`` (generated from the reference).
"""
import numpy as np

from robovat_amd.math import Orientation, Pose


def intrinsic_to_projection_matrix(intrinsics, height, width, near, far, upside_down=True):
    """Hartley-Zisserman intrinsics -> OpenGL/Bullet projection matrix, returned the way
    Bullet wants it: the 16 entries of the column-major (transposed) matrix, float32."""
    k = np.asarray(intrinsics)
    f_x, f_y, x_0, y_0, s = k[0, 0], k[1, 1], k[0, 2], k[1, 2], k[0, 1]
    if upside_down:
        x_0, y_0 = width - x_0, height - y_0
    m = np.zeros((4, 4), dtype=np.float32)
    m[0, 0] = 2 * f_x / width
    m[0, 1] = -2 * s / width
    m[0, 2] = (width - 2 * x_0) / width
    m[1, 1] = 2 * f_y / height
    m[1, 2] = (-height + 2 * y_0) / height
    m[2, 2] = (-far - near) / (far - near)
    m[2, 3] = -2 * far * near / (far - near)
    m[3, 2] = -1
    return list(m.T.flatten())


class Camera(object):

    def __init__(self, height=None, width=None, intrinsics=None, translation=None, rotation=None, crop=None):
        if crop is None:
            self._height, self._width = height, width
        else:
            self._height, self._width = crop[2] - crop[0], crop[3] - crop[1]
        self._crop = crop
        self._intrinsics = self._translation = self._rotation = None
        self.set_calibration(intrinsics, translation, rotation)

    height = property(lambda s: s._height)
    width = property(lambda s: s._width)
    intrinsics = property(lambda s: s._intrinsics)
    translation = property(lambda s: s._translation)
    rotation = property(lambda s: s._rotation)
    cx = property(lambda s: s._intrinsics[0, 2])
    cy = property(lambda s: s._intrinsics[1, 2])

    @property
    def pose(self):
        """The camera in the world frame (camera.py:76-79)."""
        return Pose([self._translation, self._rotation]).inverse()

    def set_calibration(self, intrinsics, translation, rotation):
        if intrinsics is not None:
            self._intrinsics = np.array(intrinsics, dtype=np.float64).reshape((3, 3))
            if self._crop is not None:
                self._intrinsics[0, 2] -= self._crop[1]
                self._intrinsics[1, 2] -= self._crop[0]
        if translation is not None:
            self._translation = np.array(translation, dtype=np.float64).reshape((3,))
        if rotation is not None:
            self._rotation = Orientation(rotation).matrix3

    def project_point(self, point, is_world_frame=True):
        """3D point(s) -> integer pixel(s) (camera.py:170-193)."""
        point = np.array(point, dtype=np.float64)
        if is_world_frame:
            pose = self.pose
            point = np.dot(point - pose.position, pose.matrix3)
        projected = np.dot(point, self.intrinsics.T)
        projected = projected / projected[..., 2:3]
        return np.round(projected)[..., :2].astype(np.int16)

    def deproject_pixel(self, pixel, depth, is_world_frame=True):
        """One pixel at a depth -> 3D point (camera.py:195-211)."""
        point = depth * np.linalg.inv(self.intrinsics).dot(np.r_[pixel, 1.0])
        if is_world_frame:
            pose = self.pose
            point = pose.position + np.dot(point, pose.matrix3.T)
        return point

    def deproject_depth_image(self, image, crop=None, is_world_frame=True):
        """Whole depth image -> [H * W, 3] points, row-major pixel order (camera.py:213-244)."""
        image = np.asarray(image)
        h, w = image.shape
        v, u = np.mgrid[0:h, 0:w]
        pixels = np.stack([u.ravel(), v.ravel(), np.ones(h * w)], axis=0) * image.reshape(1, -1)
        points = np.linalg.inv(self.intrinsics) @ pixels
        if crop is not None:
            # the reference indexes its [2, H*W] index array along the wrong axis here
            # (camera.py:231-235); no caller on the hot path passes a crop
            raise NotImplementedError('crop is not supported by deproject_depth_image')
        if is_world_frame:
            pose = self.pose
            points = np.asarray(pose.position).reshape(3, 1) + pose.matrix3 @ points
        return np.array(points.T)

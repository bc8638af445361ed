"""Pose / Orientation / Euler / Quaternion / Point with the reference's API
(``robovat/math/pose.py:16-279``, ``orientation.py:18-122``, ``euler.py``,
``quaternion.py``, ``point.py``): lazily converted orientation representations,
float32 ingestion (orientation.py:49), xyzw quaternions, static-xyz Euler.
"""
import numpy as np

from robovat_amd.math import rotations as R


class Point(np.ndarray):
    """3D point with x/y/z properties (robovat/math/point.py)."""

    def __new__(cls, value=(0, 0, 0)):
        obj = np.asarray(value, dtype=np.float32).reshape(3).copy().view(cls)
        return obj

    x = property(lambda s: float(s[0]), lambda s, v: s.__setitem__(0, v))
    y = property(lambda s: float(s[1]), lambda s, v: s.__setitem__(1, v))
    z = property(lambda s: float(s[2]), lambda s, v: s.__setitem__(2, v))


class Euler(np.ndarray):
    def __new__(cls, value=(0, 0, 0)):
        return np.asarray(value, dtype=np.float32).reshape(3).copy().view(cls)

    roll = property(lambda s: float(s[0]))
    pitch = property(lambda s: float(s[1]))
    yaw = property(lambda s: float(s[2]))

    @property
    def quaternion(self):
        return Quaternion(R.quaternion_from_euler(*[float(v) for v in self]))

    @property
    def matrix3(self):
        return R.matrix3_from_euler(*[float(v) for v in self]).astype(np.float32)


class Quaternion(np.ndarray):
    def __new__(cls, value=(0, 0, 0, 1)):
        return np.asarray(value, dtype=np.float32).reshape(4).copy().view(cls)

    @property
    def euler(self):
        return Euler(R.euler_from_quaternion(np.asarray(self, dtype=np.float64)))

    @property
    def matrix3(self):
        return R.matrix3_from_quaternion(np.asarray(self, dtype=np.float64)).astype(np.float32)


class Orientation(object):
    """3D orientation with lazily cached euler / quaternion / matrix3 views."""

    def __init__(self, value):
        self._euler = self._quaternion = self._matrix3 = None
        if value is None:
            return
        if isinstance(value, Orientation):
            self._euler = None if value._euler is None else value._euler.copy()
            self._quaternion = None if value._quaternion is None else value._quaternion.copy()
            self._matrix3 = None if value._matrix3 is None else value._matrix3.copy()
        elif isinstance(value, Euler):
            self._euler = value.copy()
        elif isinstance(value, Quaternion):
            self._quaternion = value.copy()
        else:
            value = np.array(value, dtype=np.float32)
            if value.size == 3:
                self._euler = Euler(value)
            elif value.size == 4:
                self._quaternion = Quaternion(value)
            elif value.size == 9:
                self._matrix3 = value.reshape([3, 3])
            else:
                raise ValueError('orientation must have 3, 4 or 9 elements')

    def copy(self):
        return Orientation(self)

    def __str__(self):
        return str(self.euler)

    @property
    def euler(self):
        if self._euler is None:
            if self._quaternion is not None:
                self._euler = self._quaternion.euler
            elif self._matrix3 is not None:
                self._euler = Euler(R.euler_from_matrix3(self._matrix3))
        return self._euler

    @property
    def quaternion(self):
        if self._quaternion is None:
            if self._euler is not None:
                self._quaternion = self._euler.quaternion
            elif self._matrix3 is not None:
                self._quaternion = Quaternion(R.quaternion_from_matrix3(self._matrix3))
        return self._quaternion

    @property
    def matrix3(self):
        if self._matrix3 is None:
            if self._quaternion is not None:
                self._matrix3 = self._quaternion.matrix3
            elif self._euler is not None:
                self._matrix3 = self._euler.matrix3
        return self._matrix3


class Pose(object):
    """3D pose: ``Pose([[x, y, z], orientation])`` or a 4x4 matrix."""

    def __init__(self, value=((0, 0, 0), (0, 0, 0))):
        if isinstance(value, Pose):
            self._position = value.position.copy()
            self._orientation = value.orientation.copy()
        elif isinstance(value, np.ndarray) and value.size == 16:
            m = value.reshape(4, 4)
            self._position = Point(m[:3, 3])
            self._orientation = Orientation(m[:3, :3])
        else:
            self._position = Point(value[0])
            self._orientation = Orientation(value[1])

    def __str__(self):
        e = self.euler
        return '[position: %g, %g, %g, euler: %g, %g, %g]' % (self.x, self.y, self.z, e[0], e[1], e[2])

    def __getitem__(self, index):
        if index == 0:
            return self.position
        if index == 1:
            return self.orientation
        raise ValueError('The index of a Pose instance can only be 0 or 1.')

    pose = property(lambda s: s)
    position = property(lambda s: s._position)
    orientation = property(lambda s: s._orientation)
    euler = property(lambda s: s._orientation.euler)
    quaternion = property(lambda s: s._orientation.quaternion)
    matrix3 = property(lambda s: s._orientation.matrix3)
    roll = property(lambda s: s.euler[0])
    pitch = property(lambda s: s.euler[1])
    yaw = property(lambda s: s.euler[2])

    @position.setter
    def position(self, value):
        self._position = Point(value)

    @orientation.setter
    def orientation(self, value):
        self._orientation = Orientation(value)

    x = property(lambda s: s._position.x, lambda s, v: s._position.__setitem__(0, v))
    y = property(lambda s: s._position.y, lambda s, v: s._position.__setitem__(1, v))
    z = property(lambda s: s._position.z, lambda s, v: s._position.__setitem__(2, v))

    @property
    def matrix4(self):
        m = np.eye(4, dtype=np.float32)
        m[:3, :3] = self.matrix3
        m[:3, 3] = self.position
        return m

    def inverse(self):
        """World origin expressed in this pose's frame (pose.py:161-172)."""
        position = np.dot(-np.asarray(self.position), self.matrix3)
        return Pose([position, self.matrix3.T])

    def transform(self, pose):
        """Map a pose from this (source) frame to the target frame (pose.py:174-189)."""
        pose = Pose(pose)
        position = np.asarray(self.position) + np.dot(np.asarray(pose.position), self.matrix3.T)
        return Pose([position, np.dot(self.matrix3, pose.matrix3)])

    def copy(self):
        return Pose(self)

    def to_array(self):
        return np.array(np.r_[np.asarray(self.position), np.asarray(self.euler)], dtype=np.float64)

    @staticmethod
    def uniform(x, y, z, roll=0.0, pitch=0.0, yaw=0.0):
        """Uniformly sampled pose; scalars are used as is (pose.py:208-246)."""
        def draw(v):
            return np.random.uniform(v[0], v[1]) if isinstance(v, (list, tuple)) else v
        return Pose([[draw(x), draw(y), draw(z)], [draw(roll), draw(pitch), draw(yaw)]])


def get_transform(source=None, target=None):
    """Rigid transform from the source frame to the target frame (pose.py:249-279)."""
    if source is not None and not isinstance(source, Pose):
        source = Pose(source)
    if target is not None and not isinstance(target, Pose):
        target = Pose(target)
    if source is not None and target is not None:
        orientation = np.dot(target.matrix3.T, source.matrix3)
        position = np.dot(np.asarray(source.position) - np.asarray(target.position), target.matrix3)
    elif source is not None:
        orientation, position = source.matrix3, source.position
    elif target is not None:
        orientation = target.matrix3.T
        position = np.dot(-np.asarray(target.position), target.matrix3)
    else:
        # the reference returns position ONES here;
        # the identity transform is what the name promises.
        orientation, position = np.eye(3, dtype=np.float32), np.zeros(3, dtype=np.float32)
    return Pose([position, orientation])

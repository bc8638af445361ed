"""``Constraint`` / ``ControllableConstraint`` with the reference's API
(``robovat/simulation/constraint.py:12-120``, ``controllable_constraint.py:21-170``): a joint between
a parent entity and a child entity, created through the physics backend
(``physics.add_constraint`` -> ``rv_set_constraint``).  The HIP backend builds the case the
reference's controllable constraint is made for: a FIXED joint between a movable body and a frame of
the world, limited to ``max_force``, whose world frame ``ControllableConstraint`` servoes towards a
target pose at bounded linear / angular speed."""
import numpy as np

from robovat_amd.math import Pose

NUM_STEPS_CHECK = 100            # controllable_constraint.py:14
TIMEOUT = 1.0                    # :15
POSITION_THRESHOLD = 0.01        # :16
EULER_THRESHOLD = np.pi / 36     # :17


class Constraint(object):

    def __init__(self, parent, child=None, joint_type='fixed', joint_axis=[0, 0, 0], parent_frame_pose=None,
                 child_frame_pose=None, max_force=None, name=None):
        self._simulator = parent.simulator
        self._uid = self.physics.add_constraint(parent.uid, None if child is None else child.uid, joint_type, joint_axis,
                                                parent_frame_pose, child_frame_pose)
        self._parent, self._child, self._joint_type = parent, child, joint_type
        if max_force is not None:
            self.max_force = max_force
        self.name = name or '%s_constraint_(%s)_(%s)_%s' % (joint_type, parent.name, getattr(child, 'name', None), self._uid)

    simulator = property(lambda s: s._simulator)
    physics = property(lambda s: s._simulator.physics)
    uid = property(lambda s: s._uid)
    parent = property(lambda s: s._parent)
    child = property(lambda s: s._child)
    joint_type = property(lambda s: s._joint_type)

    pose = property(lambda s: s.physics.get_constraint_pose(s._uid),
                    lambda s, v: s.physics.set_constraint_pose(s._uid, v))
    position = property(lambda s: s.physics.get_constraint_position(s._uid),
                        lambda s, v: s.physics.set_constraint_position(s._uid, v))
    orientation = property(lambda s: s.physics.get_constraint_orientation(s._uid),
                           lambda s, v: s.physics.set_constraint_orientation(s._uid, v))
    max_force = property(lambda s: s.physics.get_constraint_max_force(s._uid),
                         lambda s, v: s.physics.set_constraint_max_force(s._uid, v))

    def update(self):
        pass


class ControllableConstraint(Constraint):

    def __init__(self, parent, child=None, joint_type='fixed', joint_axis=[0, 0, 0], parent_frame_pose=None,
                 child_frame_pose=None, max_linear_velocity=None, max_angular_velocity=None, max_force=None, name=None):
        Constraint.__init__(self, parent, child, joint_type, joint_axis, parent_frame_pose, child_frame_pose,
                            max_force=max_force, name=name)
        self._max_linear_velocity = max_linear_velocity
        self._max_angular_velocity = max_angular_velocity
        self.reset_targets()

    def reset_targets(self):
        self._target_pose = None
        self._target_linear_velocity = self._target_angular_velocity = None
        self._start_time = self._stop_time = None

    def is_ready(self):
        return self._target_pose is None

    def set_target_pose(self, pose, linear_velocity=None, angular_velocity=None, timeout=TIMEOUT):
        self._target_pose = Pose(pose)
        self._target_linear_velocity = self.physics.time_step * (linear_velocity or self._max_linear_velocity)
        self._target_angular_velocity = self.physics.time_step * (angular_velocity or self._max_angular_velocity)
        self._start_time = self.physics.time()
        self._stop_time = self._start_time + timeout
        self._position_threshold, self._euler_threshold = POSITION_THRESHOLD, EULER_THRESHOLD

    def update(self):
        if self._target_pose is not None:
            self._update_pose_control()
            if self.physics.num_steps % NUM_STEPS_CHECK == 0:
                if self.check_reached() or self.check_timeout():
                    self.reset_targets()

    def _update_pose_control(self):
        """controllable_constraint.py:112-133: one step of bounded length towards the target."""
        cur = self.pose
        delta = np.asarray(self._target_pose.position, np.float64) - np.asarray(cur.position, np.float64)
        n = np.linalg.norm(delta)
        if n > 0:
            delta = delta / n * self._target_linear_velocity
        new_position = np.asarray(cur.position, np.float64) + delta
        de = (np.asarray(self._target_pose.euler) - np.asarray(cur.euler) + np.pi) % (2 * np.pi) - np.pi
        de = np.minimum(np.maximum(de, -self._target_angular_velocity), self._target_angular_velocity)
        new_euler = np.asarray(cur.euler, np.float64) + de
        new_euler[0] = (new_euler[0] + np.pi) % (2 * np.pi) - np.pi
        new_euler[1] = (new_euler[1] + 0.5 * np.pi) % np.pi - 0.5 * np.pi
        new_euler[2] = (new_euler[2] + np.pi) % (2 * np.pi) - np.pi
        self.pose = Pose((new_position, new_euler))

    def check_reached(self):
        dp = np.abs(np.asarray(self._target_pose.position) - np.asarray(self.pose.position))
        de = np.abs(np.asarray(self._target_pose.euler) - np.asarray(self.pose.euler)) % (2 * np.pi)
        return bool((dp < self._position_threshold).all() and (de < self._euler_threshold).all())

    def check_timeout(self):
        return self._stop_time is not None and self.physics.time() >= self._stop_time

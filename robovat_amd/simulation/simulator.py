"""``Simulator`` / ``Body`` with the reference's object-model API
(``robovat/simulation/simulator.py:20-376``, ``body.py:14-240``): the backend is
resolved by name with ``getattr(physics, physics_backend)`` exactly like the
reference (simulator.py:45-49); ``HipPhysics`` is the MI355X plugin."""
import os.path

import numpy as np

from robovat_amd.math import Pose
from robovat_amd.simulation import physics
from robovat_amd.simulation.constraint import Constraint, ControllableConstraint


class Body(object):
    """A body handle (body.py:14-240): getters read the backend state."""

    def __init__(self, simulator, filename, pose, scale=1.0, is_static=False, name=None):
        self._simulator = simulator
        self._uid = simulator.physics.add_body(filename, pose, scale=scale, is_static=is_static)
        self.name = name or 'body_%s' % self._uid
        self._is_static = is_static

    simulator = property(lambda s: s._simulator)
    physics = property(lambda s: s._simulator.physics)
    uid = property(lambda s: s._uid)
    is_static = property(lambda s: s._is_static)
    pose = property(lambda s: s.physics.get_body_pose(s._uid))
    position = property(lambda s: s.pose.position)
    orientation = property(lambda s: s.pose.orientation)
    euler = property(lambda s: s.pose.euler)
    quaternion = property(lambda s: s.pose.quaternion)
    linear_velocity = property(lambda s: s.physics.get_body_linear_velocity(s._uid))
    angular_velocity = property(lambda s: s.physics.get_body_angular_velocity(s._uid))

    matrix3 = property(lambda s: s.pose.matrix3)
    dynamics = property(lambda s: s.physics.get_body_dynamics(s._uid))
    contacts = property(lambda s: s.physics.get_body_contacts(s._uid))
    links = property(lambda s: [])
    joints = property(lambda s: [])
    # (body.py:84-116: one list per joint property)
    joint_velocities = property(lambda s: [j.velocity for j in s.joints])
    joint_lower_limits = property(lambda s: [j.lower_limit for j in s.joints])
    joint_upper_limits = property(lambda s: [j.upper_limit for j in s.joints])
    joint_max_efforts = property(lambda s: [j.max_effort for j in s.joints])
    joint_max_velocities = property(lambda s: [j.max_velocity for j in s.joints])
    joint_dampings = property(lambda s: [j.damping for j in s.joints])
    joint_frictions = property(lambda s: [j.friction for j in s.joints])
    joint_ranges = property(lambda s: [j.range for j in s.joints])

    @pose.setter
    def pose(self, value):
        self.physics.set_body_pose(self._uid, value)

    @position.setter
    def position(self, value):
        self.physics.set_body_position(self._uid, value)

    @orientation.setter
    def orientation(self, value):
        self.physics.set_body_orientation(self._uid, value)

    @linear_velocity.setter
    def linear_velocity(self, value):
        self.physics.set_body_linear_velocity(self._uid, value)

    @angular_velocity.setter
    def angular_velocity(self, value):
        self.physics.set_body_angular_velocity(self._uid, value)

    @property
    def mass(self):
        return self.physics.get_body_mass(self._uid)

    @mass.setter
    def mass(self, value):
        self.physics.set_body_mass(self._uid, value)

    def update(self):
        pass

    def set_dynamics(self, mass=None, lateral_friction=None, rolling_friction=None, spinning_friction=None):
        self.physics.set_body_dynamics(self._uid, mass=mass, lateral_friction=lateral_friction,
                                       rolling_friction=rolling_friction, spinning_friction=rolling_friction)

    def set_color(self, rgba=None, specular=None):
        self.physics.set_body_color(self._uid, rgba, specular)


class Link(object):
    """A link of the arm (link.py): index, name, world pose."""

    def __init__(self, body, index):
        self._body, self.index = body, int(index)

    uid = property(lambda s: (s._body.uid, s.index))
    name = property(lambda s: s._body.physics.get_link_name(s.uid))
    pose = property(lambda s: s._body.physics.get_link_pose(s.uid))
    position = property(lambda s: s.pose.position)
    orientation = property(lambda s: s.pose.orientation)
    parent = property(lambda s: s._body)
    center_of_mass = property(lambda s: s._body.physics.get_link_center_of_mass(s.uid))
    mass = property(lambda s: s._body.physics.get_link_mass(s.uid))
    dynamics = property(lambda s: s._body.physics.get_link_dynamics(s.uid))

    def set_dynamics(self, mass=None, lateral_friction=None, rolling_friction=None, spinning_friction=None):
        self._body.physics.set_link_dynamics(self.uid, mass=mass, lateral_friction=lateral_friction,
                                             rolling_friction=rolling_friction, spinning_friction=rolling_friction)


class Joint(object):
    """A joint of the arm (joint.py:41-92): index, name, limits, position / velocity."""

    def __init__(self, body, index):
        self._body, self.index = body, int(index)

    uid = property(lambda s: (s._body.uid, s.index))
    name = property(lambda s: s._body.physics.get_joint_name(s.uid))
    limit = property(lambda s: s._body.physics.get_joint_limit(s.uid))
    lower_limit = property(lambda s: s.limit['lower'])
    upper_limit = property(lambda s: s.limit['upper'])
    max_effort = property(lambda s: s.limit['effort'])
    max_velocity = property(lambda s: s.limit['velocity'])
    range = property(lambda s: s.upper_limit - s.lower_limit)
    velocity = property(lambda s: s._body.physics.get_joint_velocity(s.uid))
    parent = property(lambda s: s._body)
    dynamics = property(lambda s: s._body.physics.get_joint_dynamics(s.uid))
    damping = property(lambda s: s.dynamics['damping'])
    friction = property(lambda s: s.dynamics['friction'])
    reaction_force = property(lambda s: s._body.physics.get_joint_reaction_force(s.uid))

    def enable_sensor(self):
        self._body.physics.enable_joint_sensor(self.uid)

    def position_control(self, target_position, target_velocity=None, max_velocity=None, max_force=None,
                         position_gain=None, velocity_gain=None):
        """joint.py:129-155."""
        self._body.physics.position_control(self.uid, target_position, target_velocity=target_velocity, max_velocity=max_velocity,
                                            max_force=max_force, position_gain=position_gain, velocity_gain=velocity_gain)

    def velocity_control(self, target_velocity, max_force=None, position_gain=None, velocity_gain=None):
        self._body.physics.velocity_control(self.uid, target_velocity, max_force=max_force, position_gain=position_gain, velocity_gain=velocity_gain)

    def torque_control(self, target_torque):
        self._body.physics.torque_control(self.uid, target_torque)

    @property
    def position(self):
        return self._body.physics.get_joint_position(self.uid)

    @position.setter
    def position(self, value):
        self._body.physics.set_joint_position(self.uid, value)


class ControllableBody(Body):
    """The arm: targets are handed to the device-side ControllableBody state
    machine (controllable_body.py:263-345 -> rv_set_joint_targets / rv_set_link_target / rv_set_link_path)."""

    def __init__(self, *args, **kwargs):
        super(ControllableBody, self).__init__(*args, **kwargs)
        self._links = [Link(self, i) for i in self.physics.get_body_link_indices(self.uid)]
        self._joints = [Joint(self, i) for i in self.physics.get_body_joint_indices(self.uid)]

    links = property(lambda s: s._links)
    joints = property(lambda s: s._joints)

    def get_link_by_name(self, name):
        for link in self._links:
            if link.name == name:
                return link
        raise ValueError('The link %s is not found in body %s.' % (name, self.name))

    def get_joint_by_name(self, name):
        for joint in self._joints:
            if joint.name == name:
                return joint
        raise ValueError('The joint %s is not found in body %s.' % (name, self.name))

    def set_target_joint_positions(self, joint_positions, timeout=15.0, threshold=0.008726640):
        if isinstance(joint_positions, dict):
            joint_positions = [joint_positions[j.name] for j in self._joints[:7]]
        self.physics.world.set_joint_targets(np.asarray(joint_positions, np.float32)[None, :7],
                                             timeout=timeout, threshold=threshold)
        self._apply_vmax()

    def _pose7(self, link_pose):
        pose = Pose(link_pose)
        return np.concatenate([np.asarray(pose.position), np.asarray(pose.quaternion)]).astype(np.float32)

    def set_target_link_pose(self, link_ind, link_pose, timeout=15.0, threshold=0.008726640):
        self.physics.world.set_link_target(self._pose7(link_pose)[None], timeout=timeout, threshold=threshold)
        self._apply_vmax()

    def set_target_link_poses(self, link_ind, link_poses, timeout=15.0, threshold=0.008726640):
        """controllable_body.py:322-345: a path of gripper poses, followed one after the other."""
        poses = np.stack([self._pose7(p) for p in link_poses])
        self.physics.world.set_link_path(poses[None], timeout=timeout, threshold=threshold)
        self._apply_vmax()

    def set_max_joint_velocities(self, joint_velocities):
        """controllable_body.py:357-372.  The device applies LIMB_MAX_VELOCITY_RATIO x the URDF limits with every target it is
        given; limits that differ from those (SawyerSim.move_to_*(speed=...)) are sent after the target they belong to."""
        limb = self._joints[:7]
        if isinstance(joint_velocities, dict):
            v = [joint_velocities.get(j.name, None) for j in limb]
        else:
            v = list(joint_velocities)[:7]
        ratio = np.float32(self.physics.rv_config.limb_max_velocity_ratio)
        default = [float(ratio * np.float32(j.max_velocity)) for j in limb]
        v = [d if x is None else float(x) for x, d in zip(v, default)]
        if any(x <= 0 for x in v):
            raise ValueError('joint velocities must be positive: %r' % (v,))
        self._vmax_pending = None if np.allclose(v, default, rtol=1e-6, atol=0) else np.asarray(v, np.float32)

    def _apply_vmax(self):
        if getattr(self, '_vmax_pending', None) is not None:
            self.physics.world.set_max_joint_velocities(self._vmax_pending)

    def grip(self, value):
        """SawyerSim.grip (sawyer_sim.py:362-392): the two finger joints' target, replacing the limb's joint target."""
        self.physics.world.grip(float(value))

    def reset_targets(self):
        """controllable_body.py:347-350: drop the link / joint targets."""
        self.physics.world.reset_targets()

    def is_ready(self, joint_inds=None):
        """controllable_body.py:565-595 for the limb joints: no link target and no joint target on them is pending
        (targets that are done are retired by the query, as in the reference)."""
        return bool(self.physics.world.robot_ready().cpu().numpy()[0, 0])

    def is_gripper_ready(self):
        return bool(self.physics.world.robot_ready().cpu().numpy()[0, 1])

    @property
    def joint_positions(self):
        return list(self.physics.world.joint_state().cpu().numpy()[0, :, 0])


class Simulator(object):

    def __init__(self, assets_dir=None, physics_backend='HipPhysics', time_step=1e-3, gravity=[0, 0, -9.8],
                 worker_id=0, use_visualizer=False, **backend_kwargs):
        self._assets_dir = os.path.abspath(assets_dir or './')
        self._gravity = gravity
        physics_class = getattr(physics, physics_backend)
        self._physics = physics_class(time_step=time_step, use_visualizer=use_visualizer, worker_id=worker_id,
                                      **backend_kwargs)
        self._num_steps = 0
        self._bodies, self._constraints = dict(), dict()

    assets_dir = property(lambda s: s._assets_dir)
    physics = property(lambda s: s._physics)
    bodies = property(lambda s: s._bodies)
    constraints = property(lambda s: s._constraints)
    num_steps = property(lambda s: s._num_steps)
    time_step = property(lambda s: s._physics.time_step)

    def reset(self):
        self.physics.reset()
        self.physics.set_gravity(self._gravity)
        self._bodies, self._constraints = dict(), dict()
        self._num_steps = 0

    def start(self):
        self.physics.start()
        self._num_steps = 0

    def step(self):
        for body in self.bodies.values():
            body.update()
        for constraint in self.constraints.values():       # simulator.py:94-103
            constraint.update()
        self.physics.step()
        self._num_steps += 1

    def add_body(self, filename, pose=None, scale=1.0, is_static=False, is_controllable=False, name=None):
        if pose is None:
            pose = [[0, 0, 0], [0, 0, 0]]
        cls = ControllableBody if is_controllable else Body
        body = cls(simulator=self, filename=filename, pose=pose, scale=scale, is_static=is_static, name=name)
        self._bodies[body.name] = body
        return body

    def remove_body(self, name):
        self.physics.remove_body(self._bodies[name].uid)
        del self._bodies[name]

    def add_constraint(self, parent, child=None, joint_type='fixed', joint_axis=[0, 0, 0], parent_frame_pose=None,
                       child_frame_pose=None, max_force=None, max_linear_velocity=None, max_angular_velocity=None,
                       is_controllable=False, name=None):
        """simulator.py:166-224."""
        if is_controllable:
            constraint = ControllableConstraint(parent, child, joint_type, joint_axis, parent_frame_pose, child_frame_pose,
                                                max_force=max_force, max_linear_velocity=max_linear_velocity,
                                                max_angular_velocity=max_angular_velocity, name=name)
        else:
            assert max_linear_velocity is None and max_angular_velocity is None
            constraint = Constraint(parent, child, joint_type, joint_axis, parent_frame_pose, child_frame_pose,
                                    max_force=max_force, name=name)
        self._constraints[constraint.name] = constraint
        return constraint

    def remove_constraint(self, name):
        self.physics.remove_constraint(self._constraints[name].uid)
        del self._constraints[name]

    def receive_robot_commands(self, robot_command, component_type='body'):
        if component_type != 'body':
            raise ValueError('Unrecognized component type: %r' % component_type)
        component = self._bodies[robot_command.component]
        getattr(component, robot_command.command_type)(**robot_command.arguments)

    def check_contact(self, entity_a, entity_b=None):
        entities_a = entity_a if isinstance(entity_a, (list, tuple)) else [entity_a]
        entities_b = entity_b if isinstance(entity_b, (list, tuple)) else [entity_b]
        for a in entities_a:
            for b in entities_b:
                if len(self._physics.get_contact_points(a.uid, None if b is None else b.uid)) > 0:
                    return True
        return False

    def check_stable(self, body, linear_velocity_threshold, angular_velocity_threshold):
        lin = np.linalg.norm(body.linear_velocity)
        ang = np.linalg.norm(body.angular_velocity)
        moving_lin = linear_velocity_threshold is not None and lin >= linear_velocity_threshold
        moving_ang = angular_velocity_threshold is not None and ang >= angular_velocity_threshold
        return (not moving_lin) and (not moving_ang)

    def wait_until_stable(self, body, linear_velocity_threshold=0.005, angular_velocity_threshold=0.005,
                          check_after_steps=100, min_stable_steps=100, max_steps=2000):
        """simulator.py:325-376, one Python-level step at a time (API
        compatibility; VecPushEnv settles on the device)."""
        body_list = body if isinstance(body, (list, tuple)) else [body]
        num_steps = num_stable_steps = 0
        while True:
            self.step()
            num_steps += 1
            if num_steps < check_after_steps:
                continue
            if all(self.check_stable(b, linear_velocity_threshold, angular_velocity_threshold) for b in body_list):
                num_stable_steps += 1
            if num_stable_steps >= min_stable_steps or num_steps >= max_steps:
                break
        return num_steps

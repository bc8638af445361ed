"""``HipPhysics`` — the drop-in physics plugin backed by ``librovat_hip.so``.

It has the method names / arguments of ``BulletPhysics`` that the PushEnv and
grasp paths call (``robovat/simulation/physics/bullet_physics.py:89-137,
143-438, 444-729, 1061-1104, 1203-1304``; list in SURVEY.md §8b) for ONE env:
each getter reads the device state of env 0, each setter writes it through the
C ABI.  The batched fast path is ``robovat_amd.envs.VecPushEnv``; this class
exists so code written against ``Simulator`` / ``Body`` keeps working.

uids: movable bodies are their slot 0..RV_MAXB-1; the table is ``TABLE_UID``,
the arm ``ARM_UID``; links / joints are ``(ARM_UID, index)`` tuples.
"""
import os

import numpy as np

from robovat_amd import abi, configs, scenes
from robovat_amd.math import Pose, rotations
from robovat_amd.simulation.physics.physics import Physics

TABLE_UID = 100
ARM_UID = 200
STATIC_UID = 300     # ground / wall / tiles: visual only


class HipPhysics(Physics):

    def __init__(self, time_step=1e-3, use_visualizer=False, worker_id=0, config=None, robot_config=None,
                 device=0, seed=0):
        if use_visualizer:
            raise NotImplementedError('the HIP backend has no debug visualizer')
        self._env_config = config or configs.push_env_config()
        self._robot_config = robot_config or configs.sawyer_config()
        self._env_config.PHYSICS.TIME_STEP = time_step
        self._time_step = time_step
        self._worker_id, self._device, self._seed = worker_id, device, seed
        self._world = None
        self._gravity = None
        self._static = {}
        self._constraints = {}
        self._wrenches = {}          # apply_force_to_body / apply_torque_to_body: what the next step() applies
        self._velocity_targets = {}  # velocity_control(_array): joint index -> commanded velocity, refreshed every step()
        self.scene, self.shape_names = scenes.make_scene()
        self._num_steps = None

    # ---- lifecycle (bullet_physics.py:89-137)
    time_step = property(lambda s: s._time_step)
    num_steps = property(lambda s: s._num_steps)
    gravity = property(lambda s: s._gravity)
    uid = property(lambda s: s._worker_id)
    world = property(lambda s: s._world)

    rv_config = property(lambda s: s._cfg)

    def configure(self, env_config, robot_config=None, seed=None, worker_id=None):
        """(Re)create the world for an env's configuration: what an env handed this backend through
        its Simulator calls first (envs/push/push_env.py)."""
        if self._constraints:
            raise ValueError('HipPhysics.configure: the world would be recreated under %d user constraint(s) -- hand the '
                             'Simulator to the env before adding bodies or constraints' % len(self._constraints))
        self._env_config = env_config
        if robot_config is not None:
            self._robot_config = robot_config
        self._env_config.PHYSICS.TIME_STEP = self._time_step
        if seed is not None:
            self._seed = seed
        if worker_id is not None:
            self._worker_id = worker_id
        self.reset()
        if self._gravity is not None:
            self._world.set_gravity(self._gravity)
        self._num_steps = 0

    def reset(self):
        from robovat_amd import lib
        if self._world is not None:
            self._world.close()
        cfg = configs.make_rv_config(self._env_config, self._robot_config, self.shape_names, n_envs=1,
                                     env_id_offset=self._worker_id, seed=self._seed)
        self._cfg = cfg
        self._world = lib.World(cfg, self.scene, device=self._device)
        self._num_steps = None
        self._static = {}
        self._constraints = {}
        self._velocity_targets = {}

    def on_env_reset(self):
        """The env that runs on this world was reset (RobotEnv.reset -> simulator.reset() in the reference,
        robot_env.py:204-237): the device dropped every user constraint with the old episode's bodies, so
        the host mirror is dropped too -- a stale entry would re-attach the old constraint to whatever body
        takes that slot in the new episode."""
        self._constraints = {}

    def start(self):
        self._num_steps = 0

    def step(self):
        if self._wrenches:
            self._apply_wrenches()
        if self._velocity_targets:
            self._apply_velocity_targets()
        self._world.step_sub(1)
        self._num_steps += 1

    def is_real_time(self):
        return False

    def time(self):
        return self._time_step * self._num_steps

    def set_gravity(self, gravity):
        # bullet_physics.py:129-137
        self._gravity = list(gravity)
        if self._world is not None:
            self._world.set_gravity(gravity)

    # ---- helpers
    def _np(self, t):
        return t.cpu().numpy()

    def _slot(self, uid):
        if not isinstance(uid, (int, np.integer)) or not 0 <= uid < abi.RV_MAXB:
            raise ValueError('not a movable body uid: %r' % (uid,))
        return int(uid)

    # ---- bodies (bullet_physics.py:143-438)
    def add_body(self, filename, pose, scale=1.0, is_static=False):
        name = os.path.splitext(os.path.basename(filename))[0]
        pose = Pose(pose)
        if name in self.shape_names:
            params = self._np(self._world.body_params())
            free = [b for b in range(abi.RV_MAXB) if params[0, b, 0] == 0]
            if not free:
                raise ValueError('all %d movable body slots are in use' % abi.RV_MAXB)
            b = free[0]
            table_z = params[0, 0, 6]
            # URDF template defaults: mass 0.1, lateral friction 1.0 (tools/templates/urdf_template.xml:11-22);
            # is_static (bullet_physics.py:173-181 useFixedBase): mass 0 -- the body collides with the movable bodies,
            # nothing moves it (rv_dev_env.h: body_static)
            params[0, b] = [1, self.shape_names.index(name), scale, 0.0 if is_static else 0.1, 1.0, 0, table_z, 0]
            params[0, 0, 6] = table_z
            state = self._np(self._world.body_state())
            state[0, b, :3] = pose.position
            state[0, b, 3:7] = pose.quaternion
            state[0, b, 7:] = 0
            self._world.set_body_params(params)
            self._world.set_body_state(state)
            return b
        if 'table' in name:
            params = self._np(self._world.body_params())
            params[0, 0, 6] = pose.position[2]
            self._world.set_body_params(params)
            return TABLE_UID
        if 'sawyer' in name or 'arm' in name:
            js = np.zeros((1, abi.RV_NJ, 2), np.float32)
            js[0, :7, 0] = list(self._cfg.neutral_positions)
            js[0, 7, 0] = self.scene.arm.q_hi[7]
            js[0, 8, 0] = self.scene.arm.q_lo[8]
            self._world.set_joint_state(js)
            return ARM_UID
        uid = STATIC_UID + len(self._static)
        self._static[uid] = pose
        return uid

    def remove_body(self, body_uid):
        # bullet_physics.py:188-195.  The arm and the table are parts of every env of the device world: removing them
        # forgets nothing on the device -- add_body('sawyer...') puts the arm back at its neutral joint positions,
        # add_body('table...') sets the table height again (SawyerSim.reboot removes and re-adds the arm, sawyer_sim.py:90-97)
        if body_uid in (ARM_UID, TABLE_UID):
            return
        if body_uid in self._static:
            del self._static[body_uid]
            return
        b = self._slot(body_uid)
        params = self._np(self._world.body_params())
        params[0, b, 0] = 0
        self._world.set_body_params(params)

    def _state(self, body_uid):
        return self._np(self._world.body_state())[0, self._slot(body_uid)]

    def get_body_pose(self, body_uid):
        if body_uid == TABLE_UID:
            tz = float(self._np(self._world.body_params())[0, 0, 6])
            return Pose([[self._cfg.table_center[0], self._cfg.table_center[1], tz], [0, 0, 0]])
        if body_uid == ARM_UID:
            return Pose([list(self.scene.arm.base_pos), list(self.scene.arm.base_quat)])
        if body_uid in self._static:
            return self._static[body_uid]
        s = self._state(body_uid)
        return Pose([s[:3], s[3:7]])

    def get_body_position(self, body_uid):
        return np.asarray(self.get_body_pose(body_uid).position)

    def get_body_linear_velocity(self, body_uid):
        return np.array(self._state(body_uid)[7:10], dtype=np.float32)

    def get_body_angular_velocity(self, body_uid):
        return np.array(self._state(body_uid)[10:13], dtype=np.float32)

    def get_body_mass(self, body_uid):
        return float(self._np(self._world.body_params())[0, self._slot(body_uid), 3])

    def get_body_dynamics(self, body_uid):
        p = self._np(self._world.body_params())[0, self._slot(body_uid)]
        return {'mass': float(p[3]), 'lateral_friction': float(p[4]), 'rolling_friction': 0.0,
                'spinning_friction': 0.0}

    def get_body_link_indices(self, body_uid):
        return list(range(abi.RV_NFRAME)) if body_uid == ARM_UID else []

    def get_body_joint_indices(self, body_uid):
        return list(range(abi.RV_NJ)) if body_uid == ARM_UID else []

    def _set_state(self, body_uid, sl, value):
        b = self._slot(body_uid)
        state = self._np(self._world.body_state())
        state[0, b, sl] = value
        self._world.set_body_state(state)

    def set_body_pose(self, body_uid, pose):
        pose = Pose(pose)
        self.set_body_position(body_uid, pose.position)
        self.set_body_orientation(body_uid, pose.orientation)

    def set_body_position(self, body_uid, position):
        self._set_state(body_uid, slice(0, 3), np.asarray(position, np.float32))

    def set_body_orientation(self, body_uid, orientation):
        from robovat_amd.math import Orientation
        self._set_state(body_uid, slice(3, 7), np.asarray(Orientation(orientation).quaternion, np.float32))

    def set_body_linear_velocity(self, body_uid, linear_velocity):
        self._set_state(body_uid, slice(7, 10), np.asarray(linear_velocity, np.float32))

    def set_body_angular_velocity(self, body_uid, angular_velocity):
        self._set_state(body_uid, slice(10, 13), np.asarray(angular_velocity, np.float32))

    def set_body_mass(self, body_uid, mass):
        self.set_body_dynamics(body_uid, mass=mass)

    def set_body_dynamics(self, body_uid, mass=None, lateral_friction=None, rolling_friction=None,
                          spinning_friction=None):
        if body_uid == TABLE_UID:      # Body.set_dynamics on the table (grasp_4dof_env.py:262-293): its friction
            if lateral_friction is not None:
                self._world.set_friction(mu_table=lateral_friction)
            return
        b = self._slot(body_uid)
        params = self._np(self._world.body_params())
        if mass is not None:
            params[0, b, 3] = mass
        if lateral_friction is not None:
            params[0, b, 4] = lateral_friction
        state = self._np(self._world.body_state())
        self._world.set_body_params(params)
        self._world.set_body_state(state)

    def set_body_color(self, body_uid, rgba, specular):
        pass   # no renderer

    def get_body_contacts(self, body_uid):
        """Body.contacts (body.py:142-144, bullet_physics.py:1268-1286): the contact points the body takes part in."""
        return self.get_contact_points(body_uid)

    # ---- external forces (bullet_physics.py:1161-1197: applyExternalForce / applyExternalTorque with LINK_FRAME, which
    # act during ONE stepSimulation).  Kept on the host until the next step(), then applied as the velocity change
    # F dt / m, I^-1 (r x F + T) dt of that step (force, position and torque in the BODY frame, as LINK_FRAME says)
    def apply_force_to_body(self, uid, force, position):
        b = self._slot(uid)
        w = self._wrenches.setdefault(b, [np.zeros(3), np.zeros(3)])
        f = np.asarray(force, np.float64); r = np.asarray(position, np.float64)
        w[0] += f; w[1] += np.cross(r, f)

    def apply_torque_to_body(self, uid, force, position):
        b = self._slot(uid)
        w = self._wrenches.setdefault(b, [np.zeros(3), np.zeros(3)])
        w[1] += np.asarray(force, np.float64)

    def apply_force_to_link(self, uid, force, position):
        raise NotImplementedError('the arm is a kinematic pusher (its joints follow the position controller): an external force on a link moves nothing')

    def apply_torque_to_link(self, uid, force, position):
        raise NotImplementedError('the arm is a kinematic pusher (its joints follow the position controller): an external torque on a link moves nothing')

    def _apply_wrenches(self):
        params = self._np(self._world.body_params())
        state = self._np(self._world.body_state())
        dt = self._time_step
        for b, (f, t) in self._wrenches.items():
            active, shape, scale, mass = params[0, b, 0], int(params[0, b, 1]), float(params[0, b, 2]), float(params[0, b, 3])
            if not active or not mass > 0.0:
                continue                         # (absent, or a static body)
            R = np.asarray(rotations.matrix3_from_quaternion(np.asarray(state[0, b, 3:7], np.float64)), np.float64)
            ik = np.asarray(self.scene.shapes[shape].inertia_k[:], np.float64)      # principal inertia per unit mass at unit scale
            inv_i = 1.0 / (mass * scale * scale * ik)
            state[0, b, 7:10] += (R @ f) * (dt / mass)
            state[0, b, 10:13] += R @ (inv_i * t) * dt
        self._wrenches = {}
        self._world.set_body_state(state)

    # ---- links / joints (bullet_physics.py:444-729)
    def get_link_name(self, link_uid):
        return scenes.LINK_NAMES[link_uid[1]]

    def get_link_pose(self, link_uid):
        p = self._np(self._world.link_poses())[0, link_uid[1]]
        return Pose([p[:3], p[3:7]])

    def get_link_center_of_mass(self, link_uid):
        return self.get_link_pose(link_uid)

    def get_link_dynamics(self, link_uid):
        """bullet_physics.py:506-525: mass and the friction coefficients of a link (the limb links: PHYSICS.ARM_FRICTION,
        the finger tips: what Link.set_dynamics last gave them)."""
        mu = float(self._cfg.arm_friction)
        return {'mass': self.get_link_mass(link_uid), 'lateral_friction': mu, 'rolling_friction': 0.0, 'spinning_friction': 0.0}

    def set_link_mass(self, link_uid, mass):
        raise NotImplementedError('This is still buggy in PyBullet.')      # (bullet_physics.py:527-534 raises the same)

    def get_link_mass(self, link_uid):
        """<inertial> mass of a limb link / the hand (rv_arm.link_mass); the finger links are massless pads."""
        i = link_uid[1]
        return float(self.scene.arm.link_mass[i]) if i <= abi.RV_NLIMB else 0.0

    def set_link_dynamics(self, link_uid, mass=None, lateral_friction=None, rolling_friction=None,
                          spinning_friction=None):
        """Link.set_dynamics (bullet_physics.py:560-600).  What the reference uses it for is the lateral friction
        of the two finger tips (grasp_4dof_env.py:262-293): rv_set_friction.  The friction of the other links is
        the world-creation constant PHYSICS.ARM_FRICTION, masses come from the URDF."""
        if lateral_friction is None:
            return
        if link_uid[1] >= abi.RV_NLIMB + 1:         # the finger-tip links
            self._world.set_friction(mu_finger=lateral_friction)
        else:
            raise NotImplementedError('per-link friction of the limb links is a world-creation constant (PHYSICS.ARM_FRICTION)')

    def get_joint_name(self, joint_uid):
        names = scenes.LIMB_JOINT_NAMES + scenes.FINGER_JOINT_NAMES
        return names[joint_uid[1]]

    def get_joint_dynamics(self, joint_uid):
        return {'damping': 0.0, 'friction': 0.0}

    def get_joint_limit(self, joint_uid):
        j = joint_uid[1]
        a = self.scene.arm
        effort = 1.0 / a.inv_tau_max[j] if a.inv_tau_max[j] > 0 else float('inf')
        return {'lower': a.q_lo[j], 'upper': a.q_hi[j], 'effort': effort, 'velocity': a.v_max[j]}

    def get_joint_position(self, joint_uid):
        return float(self._np(self._world.joint_state())[0, joint_uid[1], 0])

    def get_joint_velocity(self, joint_uid):
        return float(self._np(self._world.joint_state())[0, joint_uid[1], 1])

    def set_joint_position(self, joint_uid, position):
        js = self._np(self._world.joint_state())
        js[0, joint_uid[1], 0] = position
        js[0, joint_uid[1], 1] = 0.0
        self._world.set_joint_state(js)

    def set_joint_velocity(self, joint_uid, velocity):
        """bullet_physics.py:713-729: resetJointState at the current position with this velocity."""
        js = self._np(self._world.joint_state())
        js[0, joint_uid[1], 1] = velocity
        self._world.set_joint_state(js)

    def enable_joint_sensor(self, joint_uid):
        pass      # (bullet_physics.py:731-742; nothing to switch on: see get_joint_reaction_force)

    def get_joint_reaction_force(self, joint_uid):
        raise NotImplementedError('the kinematic arm has no reaction forces')

    def get_joint_torque(self, joint_uid):
        raise NotImplementedError('the kinematic arm has no motor torques (PHYSICS.LIMB_DYNAMICS keeps its motor impulses on the device)')

    def position_control(self, joint_uid, target_position, target_velocity=None, max_velocity=None, max_force=None,
                         position_gain=None, velocity_gain=None):
        """bullet_physics.py:959-1006: POSITION_CONTROL of ONE joint -- position_control_array with one entry."""
        if max_velocity is not None:
            raise NotImplementedError('This is not implemented in pybullet.')
        self.position_control_array(joint_uid[0], [joint_uid[1]], [target_position])

    def velocity_control(self, joint_uid, target_velocity, max_force=None, position_gain=None, velocity_gain=None):
        """bullet_physics.py:1008-1031: VELOCITY_CONTROL of ONE joint -- velocity_control_array with one entry."""
        self.velocity_control_array(joint_uid[0], [joint_uid[1]], [target_velocity])

    def velocity_control_array(self, body_uid, joint_inds, target_velocities, max_forces=None, position_gains=None, velocity_gains=None):
        """bullet_physics.py:1106-1150: VELOCITY_CONTROL of joints of the arm.  The device's motor law is the POSITION_CONTROL
        one -- commanded velocity kp (q* - q) / dt, limited to the joint's speed limit and, per substep, to its
        effort-equivalent acceleration limit.  A velocity command v is that law with the target kept dt v / kp ahead of the
        joint: the commanded velocity is v exactly, the acceleration limit plays the role of the motor's maximum force
        (`max_forces` cannot change it: the kinematic arm has no force, only that limit).  The target is refreshed before every
        step() until the joint gets a position command (as PyBullet keeps a motor command until it is replaced)."""
        for j, v in zip(joint_inds, target_velocities):
            if not 0 <= int(j) < abi.RV_NJ:
                raise ValueError('joint index %r outside the arm (%d joints)' % (j, abi.RV_NJ))
            self._velocity_targets[int(j)] = float(v)

    def _apply_velocity_targets(self):
        js = self._np(self._world.joint_state())[0, :, 0]
        kp, dt = np.float32(self._cfg.kp), np.float32(self._cfg.dt)
        q = np.zeros((1, abi.RV_NJ), np.float32)
        mask = np.zeros((1, abi.RV_NJ), np.uint8)
        for j, v in self._velocity_targets.items():
            q[0, j] = np.float32(js[j]) + np.float32(v) * dt / kp
            mask[0, j] = 1
        self._world.set_motor_targets(q, mask)

    def torque_control(self, joint_uid, target_torque):
        raise NotImplementedError('the device-side motor law is POSITION_CONTROL (controllable_body.py:458-466 uses nothing else): no TORQUE_CONTROL')

    def torque_control_array(self, body_uid, joint_inds, target_torques):
        raise NotImplementedError('the device-side motor law is POSITION_CONTROL (controllable_body.py:458-466 uses nothing else): no TORQUE_CONTROL')

    # ---- control (bullet_physics.py:1061-1104) and IK (:1203-1262)
    def position_control_array(self, body_uid, joint_inds, target_positions, target_velocities=None,
                               max_velocities=None, max_forces=None, position_gains=None, velocity_gains=None):
        if max_velocities is not None:
            raise NotImplementedError('This is not implemented in pybullet.')
        q = np.zeros((1, abi.RV_NJ), np.float32)
        mask = np.zeros((1, abi.RV_NJ), np.uint8)
        for j, v in zip(joint_inds, target_positions):
            if not 0 <= int(j) < abi.RV_NJ:
                raise ValueError('joint index %r outside the arm (%d joints)' % (j, abi.RV_NJ))
            q[0, int(j)] = v
            mask[0, int(j)] = 1
            self._velocity_targets.pop(int(j), None)          # (a position command replaces a velocity command)
        self._world.set_motor_targets(q, mask)

    def compute_inverse_kinematics(self, link_uid, link_pose, upper_limits=None, lower_limits=None, ranges=None,
                                   damping=None, neutral_positions=None):
        pose = Pose(link_pose)
        p = np.concatenate([np.asarray(pose.position), np.asarray(pose.quaternion)]).astype(np.float32)
        q = self._np(self._world.compute_ik(p[None]))[0]
        js = self._np(self._world.joint_state())[0, :, 0]
        return list(q) + list(js[7:])

    # ---- contacts (bullet_physics.py:1268-1304): the LENGTH of the list is what callers use
    def get_contact_points(self, a_uid, b_uid=None):
        def body(uid):
            if isinstance(uid, (int, np.integer)):
                return int(uid)
            if isinstance(uid, (tuple, list)):
                return int(uid[0])
            raise ValueError
        a = body(a_uid)
        b = None if b_uid is None else body(b_uid)
        flags = self._np(self._world.query_contacts())[0]
        counts = self._np(self._world.manifold_counts())[0]
        if b is not None and a != ARM_UID and b == ARM_UID:
            a, b = b, a
        hit = False
        if a == ARM_UID:
            if b is None:
                hit = bool(flags[0] or flags[1])
            elif b == TABLE_UID:
                hit = bool(flags[0])
            elif 0 <= b < abi.RV_MAXB:
                hit = bool(flags[2 + b])
        elif 0 <= a < abi.RV_MAXB:
            if b is None or b == TABLE_UID:
                hit = counts[a] > 0
            if (b is None or (0 <= b < abi.RV_MAXB)) and not hit:
                for k, (x, y) in enumerate([(0, 1), (0, 2), (0, 3), (1, 2), (1, 3), (2, 3)]):
                    if a in (x, y) and (b is None or b in (x, y)) and counts[abi.RV_MAXB + k] > 0:
                        hit = True
        return [0.0] if hit else []

    # ---- not on the PushEnv / grasp paths ("implement last / stub")
    # ---- user constraints (bullet_physics.py:748-957: createConstraint / changeConstraint / removeConstraint)
    def add_constraint(self, parent_uid, child_uid, joint_type='fixed', joint_axis=[0, 0, 0],
                       parent_frame_pose=None, child_frame_pose=None):
        """A 'fixed', 'point2point', 'prismatic' (along joint_axis) or 'revolute' (about joint_axis) joint -- the four types of the
        reference's JOINT_TYPES_MAPPING (bullet_physics.py:20-25) -- between a frame of a movable body (the parent) and a frame of the world
        (child None: the constraint ControllableConstraint servoes) or of another movable body (the child).
        One party may be a LINK of the arm -- an `(ARM_UID, link_index)` entity as in the reference (bullet_physics.py:773-790),
        e.g. an object attached to the hand: 'fixed' and 'point2point' joints; the link moves kinematically and takes no
        impulse.  (The library's parent is always the movable body: with the link given as the parent the two frames swap
        roles, which a fixed / point-to-point joint does not notice.)  Returns the constraint uid."""
        if joint_type not in ('fixed', 'point2point', 'prismatic', 'revolute'):
            raise NotImplementedError("joint types built: 'fixed', 'point2point', 'prismatic', 'revolute' (not %r)" % (joint_type,))

        def is_link(uid):
            return isinstance(uid, (tuple, list)) and len(uid) == 2 and uid[0] == ARM_UID
        if is_link(parent_uid) and not is_link(child_uid) and child_uid is not None:
            parent_uid, child_uid = child_uid, parent_uid
            parent_frame_pose, child_frame_pose = child_frame_pose, parent_frame_pose
        link = None
        if is_link(child_uid):
            link = int(child_uid[1])
            if not 0 <= link < abi.RV_NFRAME:
                raise ValueError('no such link of the arm: %r' % (child_uid,))
            if joint_type not in ('fixed', 'point2point'):
                raise NotImplementedError('a link of the arm as a party: fixed and point2point joints (not %r)' % (joint_type,))
        b = self._slot(parent_uid)
        child = -1 if child_uid is None else (abi.RV_CHILD_LINK(link) if link is not None else self._slot(child_uid))
        if child == b:
            raise ValueError('a body cannot be constrained to itself')
        if b in self._constraints:
            raise ValueError('body %d already has a constraint' % b)
        frame = Pose(parent_frame_pose if parent_frame_pose is not None else [[0, 0, 0], [0, 0, 0]])
        if child_frame_pose is None:        # where the joint frame of the parent is now (seen from the child)
            child_frame_pose = self.get_body_pose(b).transform(frame)
            if link is not None:
                child_frame_pose = self.get_link_pose((ARM_UID, link)).inverse().transform(child_frame_pose)
            elif child >= 0:
                child_frame_pose = self.get_body_pose(child).inverse().transform(child_frame_pose)
        pose = Pose(child_frame_pose)
        entry = {'frame': frame, 'pose': pose, 'max_force': 500.0, 'child': child, 'joint_type': joint_type}     # pybullet's default maxForce
        if joint_type in ('prismatic', 'revolute'):
            # the library slides along (turns about) the x axis of the joint frame: a joint_axis (given in the child's joint frame,
            # pybullet's jointAxis) other than x is the same rotation applied to both joint frames.  Validated BEFORE the
            # mirror entry exists: a rejected call must leave the body free to be constrained again
            a = np.asarray(joint_axis, np.float64)
            if not np.linalg.norm(a) > 0.0:
                raise ValueError('a %s joint needs a joint_axis' % joint_type)
            a = a / np.linalg.norm(a)
            n = np.cross([1.0, 0.0, 0.0], a)
            if np.linalg.norm(n) < 1e-12:
                qr = np.array([0.0, 0.0, 0.0, 1.0]) if a[0] > 0 else np.array([0.0, 0.0, 1.0, 0.0])
            else:
                th = np.arctan2(np.linalg.norm(n), a[0])
                qr = np.concatenate([np.sin(0.5 * th) * n / np.linalg.norm(n), [np.cos(0.5 * th)]])
            entry['axis_quat'] = qr
        self._constraints[b] = entry
        try:
            self._push_constraint(b)
        except Exception:
            del self._constraints[b]          # nothing reached the device: no mirror entry either
            raise
        return b

    def _push_constraint(self, b):
        c = self._constraints[b]
        f = np.concatenate([np.asarray(c['frame'].position, np.float64), np.asarray(c['frame'].quaternion, np.float64)])
        t = np.concatenate([np.asarray(c['pose'].position, np.float64), np.asarray(c['pose'].quaternion, np.float64)])
        if 'axis_quat' in c:
            f[3:] = rotations.quaternion_multiply(f[3:], c['axis_quat']); t[3:] = rotations.quaternion_multiply(t[3:], c['axis_quat'])
        self._world.set_constraint(b, t, frame7=f, max_force=c['max_force'], child=c.get('child', -1), joint_type=c.get('joint_type', 'fixed'))

    def _con(self, uid):
        if uid not in self._constraints:
            raise ValueError('no such constraint: %r' % (uid,))
        return self._constraints[uid]

    def remove_constraint(self, constraint_uid):
        self._con(constraint_uid)
        self._world.remove_constraint(constraint_uid)
        del self._constraints[constraint_uid]

    def get_constraint_pose(self, constraint_uid):
        return self._con(constraint_uid)['pose']

    def get_constraint_position(self, constraint_uid):
        return self._con(constraint_uid)['pose'].position

    def get_constraint_orientation(self, constraint_uid):
        return self._con(constraint_uid)['pose'].orientation

    def get_constraint_max_force(self, constraint_uid):
        return self._con(constraint_uid)['max_force']

    def set_constraint_pose(self, constraint_uid, pose):
        self._con(constraint_uid)['pose'] = Pose(pose)
        self._push_constraint(constraint_uid)

    def set_constraint_position(self, constraint_uid, position):
        c = self._con(constraint_uid)
        c['pose'] = Pose([position, c['pose'].orientation])
        self._push_constraint(constraint_uid)

    def set_constraint_orientation(self, constraint_uid, orientation):
        c = self._con(constraint_uid)
        c['pose'] = Pose([c['pose'].position, orientation])
        self._push_constraint(constraint_uid)

    def set_constraint_max_force(self, constraint_uid, max_force):
        self._con(constraint_uid)['max_force'] = float(max_force)
        self._push_constraint(constraint_uid)

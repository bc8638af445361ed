"""``SawyerSim`` -- the robot facade env code drives (``robovat/robots/sawyer/sawyer_sim.py:15-416``), over the HIP backend.

The reference's class turns ``move_to_joint_positions / move_to_gripper_pose / move_along_gripper_path / grip`` into
``RobotCommand``s that ``Simulator.receive_robot_commands`` (simulator.py:226-244) dispatches to the arm's
``ControllableBody``; the control state machine then runs inside ``Simulator.step``.  Here the verbs and the dispatch
are the same, the state machine lives on the device: a command ends in ``rv_set_joint_targets``, ``rv_set_link_target``,
``rv_set_link_path`` or ``rv_grip`` (include/rovat.h), the readiness queries in ``rv_get_robot_ready``.  Reference-style
env code -- a phase machine that calls ``self.robot.*`` and ``simulator.step()`` -- therefore runs unchanged on a
``Simulator(physics_backend='HipPhysics')``; the batched envs (``VecPushEnv``) never come here, their phase machine is
on the device as well.

Speed arguments are accepted and must be the configured ``LIMB_MAX_VELOCITY_RATIO`` (or None): the device applies that
ratio to the URDF limits when a target is set (the reference computes per-call limits but has no way to hand them to
PyBullet, controllable_body.py:366)."""
import numpy as np

from robovat_amd import abi, configs
from robovat_amd.math import Pose
from robovat_amd.robots.robot_command import RobotCommand


class SawyerSim(object):
    ARM_NAME = 'sawyer_arm'
    BASE_NAME = 'sawyer_base'
    HEAD_NAME = 'sawyer_head'

    def __init__(self, simulator, pose=[[0, 0, 0], [0, 0, 0]], joint_positions=None, config=None):
        self.config = config or configs.sawyer_config()
        self._simulator = simulator
        self._arm_pose = Pose(pose)
        self._initial_joint_positions = list(joint_positions if joint_positions is not None
                                             else self.config.LIMB_NEUTRAL_POSITIONS)
        if len(self._initial_joint_positions) != len(self.config.LIMB_JOINT_NAMES):
            raise ValueError('%d joint positions for %d limb joints' % (len(self._initial_joint_positions), len(self.config.LIMB_JOINT_NAMES)))
        self._arm = None
        self.reboot()

    # ---- parts
    arm = property(lambda s: s._arm)
    base = property(lambda s: None)          # (base and head are visual bodies of the reference; not simulated)
    head = property(lambda s: None)
    end_effector = property(lambda s: s._end_effector)
    l_finger_tip = property(lambda s: s._l_finger_tip)
    r_finger_tip = property(lambda s: s._r_finger_tip)
    pose = property(lambda s: s._arm_pose)

    @property
    def joint_positions(self):
        return dict((joint.name, joint.position) for joint in self._limb_joints)

    def reboot(self):
        """(Re)create the arm in the simulator and put it at the initial joint positions, gripper open
        (sawyer_sim.py:86-171)."""
        sim = self._simulator
        if self._arm is not None and self.ARM_NAME in sim.bodies:
            sim.remove_body(self.ARM_NAME)
        self._arm = sim.add_body('sawyer.urdf', self._arm_pose, is_static=True, is_controllable=True, name=self.ARM_NAME)
        arm = self._arm
        self._limb_joints = [arm.get_joint_by_name(n) for n in self.config.LIMB_JOINT_NAMES]
        self._limb_inds = [j.index for j in self._limb_joints]
        self._end_effector = arm.get_link_by_name(self.config.END_EFFCTOR_NAME)
        self._l_finger_joint = arm.get_joint_by_name(self.config.L_FINGER_NAME)
        self._r_finger_joint = arm.get_joint_by_name(self.config.R_FINGER_NAME)
        self._l_finger_tip = arm.get_link_by_name(self.config.L_FINGER_TIP_NAME)
        self._r_finger_tip = arm.get_link_by_name(self.config.R_FINGER_TIP_NAME)
        for joint, q in zip(self._limb_joints, self._initial_joint_positions):
            joint.position = q
        self._l_finger_joint.position = self._l_finger_joint.upper_limit
        self._r_finger_joint.position = self._r_finger_joint.lower_limit
        arm.reset_targets()
        if self.config.OPEN_GRIPPER_WHEN_RESET:
            self.grip(0)

    def reset(self, positions=None):
        """Move the limb to the neutral (or the given) joint positions and open the gripper (sawyer_sim.py:173-184)."""
        self.move_to_joint_positions(self.config.LIMB_NEUTRAL_POSITIONS if positions is None else positions)
        self.grip(0)

    # ---- commands
    def _limits(self, speed, timeout, threshold):
        """sawyer_sim.py:196-206: the defaults of the three optional arguments of a motion command."""
        self._speed = self.config.LIMB_MAX_VELOCITY_RATIO if speed is None else float(speed)
        if not self._speed > 0:
            raise ValueError('speed must be a positive ratio of the joint velocity limits: %r' % (speed,))
        return (self.config.LIMB_TIMEOUT if timeout is None else timeout,
                self.config.LIMB_POSITION_THRESHOLD if threshold is None else threshold)

    def _command(self, command_type, **arguments):
        self._simulator.receive_robot_commands(RobotCommand(component=self._arm.name, command_type=command_type, arguments=arguments))

    def _max_velocity_command(self):
        # sawyer_sim.py:212-220: speed x the URDF velocity limit of every limb joint, with every motion command
        ratio = self._speed
        self._command('set_max_joint_velocities', joint_velocities=dict((j.name, ratio * j.max_velocity) for j in self._limb_joints))

    def move_to_joint_positions(self, positions, speed=None, timeout=None, threshold=None):
        timeout, threshold = self._limits(speed, timeout, threshold)
        if isinstance(positions, dict):
            positions = [positions[j.name] for j in self._limb_joints]
        self._arm.reset_targets()
        self._max_velocity_command()
        self._command('set_target_joint_positions', joint_positions=list(positions), timeout=timeout, threshold=threshold)

    def move_to_gripper_pose(self, pose, speed=None, timeout=None, threshold=None, straight_line=False):
        timeout_, threshold_ = self._limits(speed, timeout, threshold)
        pose = Pose(pose)
        if straight_line:
            # way points every END_EFFECTOR_STEP on the segment from where the end effector is, all with the orientation of
            # the target (sawyer_sim.py:259-276); the device's pose queue holds RV_MAXQ poses
            p0 = np.asarray(self.end_effector.pose.position, np.float64)
            delta = np.asarray(pose.position, np.float64) - p0
            num = min(int(np.linalg.norm(delta) / self.config.END_EFFECTOR_STEP), abi.RV_MAXQ - 1)
            waypoints = [Pose([p0 + delta * (float(i) / float(num)), pose.quaternion]) for i in range(num)]
            self.move_along_gripper_path(waypoints + [pose], speed=speed, timeout=timeout, threshold=threshold)
            return
        self._arm.reset_targets()
        self._max_velocity_command()
        self._command('set_target_link_pose', link_ind=self._end_effector.index, link_pose=pose, timeout=timeout_, threshold=threshold_)

    def move_along_gripper_path(self, poses, speed=None, timeout=None, threshold=None):
        timeout, threshold = self._limits(speed, timeout, threshold)
        self._arm.reset_targets()
        self._max_velocity_command()
        self._command('set_target_link_poses', link_ind=self._end_effector.index, link_poses=[Pose(p) for p in poses],
                      timeout=timeout, threshold=threshold)

    def grip(self, value=1):
        """0 = open, 1 = closed (sawyer_sim.py:362-392: the value is clipped to [0.01, 0.99] on the device as well)."""
        self._command('grip', value=float(value))

    def stop_limb(self):
        self._arm.reset_targets()

    # ---- queries
    def is_limb_ready(self):
        return self._arm.is_ready(joint_inds=self._limb_inds)

    def is_gripper_ready(self):
        return self._arm.is_gripper_ready()


def factory(simulator, pose=[[0, 0, 0], [0, 0, 0]], joint_positions=None, config=None):
    """``sawyer.factory`` (robots/sawyer/__init__.py): the simulated Sawyer for a simulator (the real robot -- ROS /
    Intera -- is out of scope)."""
    if simulator is None:
        raise NotImplementedError('SawyerReal (ROS / Intera) is not part of this build')
    return SawyerSim(simulator, pose=pose, joint_positions=joint_positions, config=config)

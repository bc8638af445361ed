"""Pin the arm-control logic against the reference's OWN classes (build container only).

    python tests/golden/gen_control_golden.py

The reference's unmodified `Simulator`, `SawyerSim` and `ControllableBody`
(robovat/simulation/simulator.py, robots/sawyer/sawyer_sim.py,
simulation/controllable_body.py) are imported from /root/reference and run
against a `Physics` plugin -- registered through the reference's own seam
`getattr(physics, physics_backend)` (simulator.py:45-49) -- whose joint motors,
forward kinematics and IK are the CPU oracle's (double build, control hooks of
oracle/orc.py).  Every decision of the control layer (when IK is recomputed,
when a link target pops, when a joint target is done or times out, what
`is_limb_ready` reports, how `grip` replaces the limb target) is therefore the
REFERENCE's; the joint trajectory that results is written to
tests/golden/control_golden.json.  tests/test_control_golden.py replays the same
commands through the oracle's own restatement of that layer (control_update(),
robot_move_to_*(), robot_grip()) and must reproduce the trajectory exactly.

Only DATA travels: commands, sampled joint states and ready flags.
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)

import gen_golden as _stubs  # noqa: E402,F401  (installs the pybullet/cv2/gym/easydict stubs, numpy aliases)

from robovat.math import Pose  # noqa: E402
from robovat.simulation import physics as ref_physics  # noqa: E402
from robovat.simulation.simulator import Simulator  # noqa: E402
from robovat.robots.sawyer.sawyer_sim import SawyerSim  # noqa: E402

from robovat_amd import abi, configs, scenes  # noqa: E402
from oracle import orc  # noqa: E402

EasyDict = sys.modules['easydict'].EasyDict
DT = float(np.float32(1e-3))


class OraclePhysics(object):
    """Physics plugin for the reference's Simulator: one Sawyer-like arm whose
    dynamics are the oracle's.  Method names/arguments follow
    robovat/simulation/physics/bullet_physics.py."""

    ARM_UID = 0

    def __init__(self, time_step=1e-3, use_visualizer=False, worker_id=0):
        self._time_step = DT
        self._num_steps = None
        scene, names = scenes.make_scene()
        cfg = configs.make_rv_config(n_envs=1, shape_names=names, seed=1)
        self.w = orc.OracleWorld(cfg, scene, double=True)
        self.w.set_external_control(True)
        self.joint_names = scenes.LIMB_JOINT_NAMES + scenes.FINGER_JOINT_NAMES
        self.link_names = scenes.LINK_NAMES
        self.arm = scene.arm
        self._bodies = 0
        self.ik_calls = 0
        # IK seed bookkeeping: the previous solution while it is
        # the one being position-controlled, else the current joint state
        self._last_ik = None
        self._tracking = False
        self._controlled = False

    # -- lifecycle (bullet_physics.py:89-109, 122-127)
    time_step = property(lambda self: self._time_step)

    def reset(self):
        self._num_steps = None

    def start(self):
        self._num_steps = 0

    def set_gravity(self, g):
        pass

    def step(self):
        if not self._controlled:
            self._tracking = False
        self._controlled = False
        self.w.step_sub(1)
        self._num_steps += 1

    def time(self):
        return self._time_step * self._num_steps

    # -- bodies: only the arm has joints; base / head are inert
    def add_body(self, filename, pose, scale=1.0, is_static=False):
        uid = self._bodies
        self._bodies += 1
        return uid

    def remove_body(self, uid):
        pass

    def get_body_link_indices(self, uid):
        return list(range(len(self.link_names))) if uid == self.ARM_UID else []

    def get_body_joint_indices(self, uid):
        return list(range(len(self.joint_names))) if uid == self.ARM_UID else []

    def get_link_name(self, uid):
        return self.link_names[uid[1]]

    def get_joint_name(self, uid):
        return self.joint_names[uid[1]]

    def get_joint_limit(self, uid):
        j = uid[1]
        return {'lower': float(self.arm.q_lo[j]), 'upper': float(self.arm.q_hi[j]),
                'effort': 0.0, 'velocity': float(self.arm.v_max[j])}

    def get_joint_position(self, uid):
        return float(self.w.joint_state()[0, uid[1], 0])

    def get_joint_velocity(self, uid):
        return float(self.w.joint_state()[0, uid[1], 1])

    def set_joint_position(self, uid, position):
        s = self.w.joint_state()
        s[0, uid[1], 0] = position
        s[0, uid[1], 1] = 0.0
        self.w.set_joint_state(s)

    def get_link_pose(self, uid):
        p = self.w.link_poses()[0, uid[1]]
        return Pose([p[:3], p[3:7]])

    def get_body_pose(self, uid):
        return Pose([[0, 0, 0], [0, 0, 0]])

    # -- control (bullet_physics.py:1061-1104, 1203-1262)
    def position_control_array(self, body_uid, joint_inds, target_positions, target_velocities=None,
                               max_velocities=None, max_forces=None, position_gains=None, velocity_gains=None):
        idx = [int(i) for i in joint_inds]
        pos = [float(p) for p in target_positions]
        self.w.motor_targets(idx, pos)
        self._controlled = True
        self._tracking = (self._last_ik is not None and idx == list(range(7)) and pos == self._last_ik)

    def compute_inverse_kinematics(self, link_uid, link_pose, upper_limits=None, lower_limits=None,
                                   ranges=None, damping=None, neutral_positions=None):
        pose = Pose(link_pose)
        p7 = np.concatenate([pose.position, pose.quaternion]).astype(np.float32)
        seed = self._last_ik if self._tracking else None
        q = self.w.compute_ik_seeded(seed, p7)
        self.ik_calls += 1
        self._last_ik = [float(x) for x in q]
        # Bullet returns every movable joint (controllable_body.py:480-482 truncates)
        return self._last_ik + [0.0, 0.0]

    def invalidate_ik_seed(self):
        self._tracking = False


ref_physics.OraclePhysics = OraclePhysics


def f32(x):
    return [float(np.float32(v)) for v in x]


def pose7(position, euler):
    p = Pose([f32(position), f32(euler)])
    return f32(np.concatenate([p.position, p.quaternion]))


def main():
    rc = configs.SAWYER_SIM_CONFIG
    cfg = EasyDict(dict(rc, ARM_URDF='arm.urdf', BASE_URDF='base.urdf', HEAD_URDF='head.urdf'))
    sim = Simulator(physics_backend='OraclePhysics', time_step=DT)
    sim.reset()
    initial = f32([0.1, -1.0, 0.05, 1.9, 0.0, 0.7, 3.2])
    sim.start()                      # robot_env.py:204-237: reset(), start(), then the robot is (re)booted
    robot = SawyerSim(simulator=sim, joint_positions=initial, config=cfg)
    phys = sim.physics

    # commands: (substep at which issued, kind, args).  Poses are float32-exact.
    ee0 = pose7([0.55, 0.10, 0.30], [0.0, np.pi, 0.0])
    ee1 = pose7([0.60, -0.15, 0.16], [0.0, np.pi, 0.6])
    ee2 = pose7([0.72, -0.05, 0.16], [0.0, np.pi, 0.6])
    path = [pose7([0.50 + 0.05 * k, 0.05 * k, 0.25], [0.0, np.pi, -0.3]) for k in range(4)]
    far = pose7([1.60, 0.0, 0.30], [0.0, np.pi, 0.0])          # unreachable -> times out
    commands = [
        (0, 'reset', None),                                   # SawyerSim.reset(): neutral + grip(0)
        (1500, 'move_to_gripper_pose', ee0),
        (3205, 'grip', 1.0),                                   # replaces the limb joint target mid-move
        (3700, 'move_to_gripper_pose', ee1),
        (5003, 'move_to_gripper_pose', ee2),                   # re-target while the first is still tracked
        (7000, 'move_along_gripper_path', path),
        (10500, 'move_to_joint_positions', f32(rc['LIMB_NEUTRAL_POSITIONS'])),
        (12000, 'move_to_gripper_pose_timeout', far),          # timeout=1.5 s
        (14100, 'grip', 0.0),
    ]
    total = 15000
    sample_every = 50
    samples, ready, events = [], [], []
    prev_ready = None
    ci = 0
    for k in range(total):
        while ci < len(commands) and commands[ci][0] == k:
            _, kind, arg = commands[ci]
            phys.invalidate_ik_seed()
            if kind == 'reset':
                robot.reset()
            elif kind == 'move_to_gripper_pose':
                robot.move_to_gripper_pose([arg[:3], arg[3:]])
            elif kind == 'move_to_gripper_pose_timeout':
                robot.move_to_gripper_pose([arg[:3], arg[3:]], timeout=1.5)
            elif kind == 'move_along_gripper_path':
                robot.move_along_gripper_path([Pose([p[:3], p[3:]]) for p in arg])
            elif kind == 'move_to_joint_positions':
                robot.move_to_joint_positions(list(arg))
            elif kind == 'grip':
                robot.grip(arg)
            ci += 1
        # RobotEnv-style polling every substep (push_env.py:631-937 polls is_limb_ready in its phase loop)
        r = (bool(robot.is_limb_ready()), bool(robot.is_gripper_ready()))
        if r != prev_ready:
            events.append([k, int(r[0]), int(r[1])])
            prev_ready = r
        if k % sample_every == 0:
            samples.append(phys.w.joint_state()[0].tolist())
        sim.step()
    samples.append(phys.w.joint_state()[0].tolist())
    out = {
        'about': 'reference Simulator+SawyerSim+ControllableBody driving the oracle arm (double build); '
                 'see tests/golden/gen_control_golden.py',
        'dt': DT, 'initial': initial, 'commands': commands, 'total_substeps': total,
        'sample_every': sample_every, 'joint_state': samples, 'ready_events': events,
        'ik_calls': phys.ik_calls,
        'final_link_poses': phys.w.link_poses()[0].tolist(),
    }
    with open(os.path.join(HERE, 'control_golden.json'), 'w') as f:
        json.dump(out, f)
    print('wrote control_golden.json: %d samples, %d ready events, %d IK calls' % (len(samples), len(events), phys.ik_calls))
    print(events)


if __name__ == '__main__':
    main()

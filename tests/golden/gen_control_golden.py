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
sys.path.insert(0, HERE)

from ref_harness import DT, Pose, f32, pose7, robot_config  # noqa: E402
from robovat.simulation.simulator import Simulator  # noqa: E402
from robovat.robots.sawyer.sawyer_sim import SawyerSim  # noqa: E402


def main():
    cfg = robot_config()
    rc = cfg
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

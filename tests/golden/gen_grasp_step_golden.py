"""Pin the Grasp4DofEnv macro-step against the reference's OWN methods (build container only).

    python tests/golden/gen_grasp_step_golden.py

The reference's unmodified `Grasp4DofEnv._execute_action` (grasp_4dof_env.py:213-293) with
`_is_phase_ready`, `_get_next_phase`, `SawyerSim.move_to_joint_positions / move_to_gripper_pose
(straight_line) / move_along_gripper_path / grip / reset`, `ControllableBody.update`,
`Link.set_dynamics`, `Simulator.step / check_contact / wait_until_stable` and
`GraspReward.get_reward` (grasp_reward.py:49-68) run on top of the oracle's physics (force-limited
gripper) through the `OraclePhysics` plugin (tests/golden/ref_harness.py).  The env instance is
created WITHOUT its constructor (absent configs / assets / camera); the attributes it would have
derived from the config are set from the BUILD-CHOSEN config the oracle uses.  The scene is the
oracle's own reset for the same seed on both sides.  tests/test_grasp_step_golden.py replays the
recorded actions with orc_step_macro().
"""
import json
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)

import ref_harness as H  # noqa: E402
from ref_harness import DT, EasyDict, configs  # noqa: E402

for name in ('matplotlib', 'matplotlib.pyplot'):
    if name not in sys.modules:
        sys.modules[name] = types.ModuleType(name)
sys.modules['matplotlib'].pyplot = sys.modules['matplotlib.pyplot']

from robovat.simulation.simulator import Simulator  # noqa: E402
from robovat.robots.sawyer.sawyer_sim import SawyerSim  # noqa: E402
from robovat.envs.grasp.grasp_4dof_env import Grasp4DofEnv  # noqa: E402
from robovat.reward_fns.grasp_reward import GraspReward  # noqa: E402


def build_env(seed, env_id, overrides):
    H.OraclePhysics.SEED = seed
    H.OraclePhysics.ENV_ID = env_id
    H.OraclePhysics.CFG_OVERRIDES = overrides
    H.OraclePhysics.ENV_KIND = 'grasp'
    sim = Simulator(physics_backend='OraclePhysics', time_step=DT)
    sim.reset()
    sim.start()
    phys = sim.physics
    phys.w.reset()                                  # the oracle's own reset builds the scene
    cnt = phys.w.env_counters()[0]
    sim._num_steps = int(cnt[0]); phys._num_steps = int(cnt[0])

    env = object.__new__(Grasp4DofEnv)
    env._simulator = sim
    env._debug = False
    env._num_episodes = 0
    env._num_steps = 0
    env._done = False
    c = phys.cfg
    env._config = EasyDict({
        'ACTION': {'TYPE': 'CUBOID'},
        'ARM': {'GRIPPER_SAFE_HEIGHT': float(c.gripper_safe_height), 'FINGER_TIP_OFFSET': float(c.finger_tip_offset),
                'OVERHEAD_POSITIONS': [float(x) for x in c.overhead_positions],
                'OFFSTAGE_POSITIONS': [float(x) for x in c.offstage_positions]},
        'SIM': {'MAX_ACTION_STEPS': int(c.max_action_steps)},
    })
    # ArmEnv._reset_scene / _reset_robot (arm_env.py:78-107), Grasp4DofEnv._reset_robot (:206-211)
    env.table = sim.add_body('table.urdf', is_static=True, name='table')
    env.robot = SawyerSim(simulator=sim, config=H.robot_config())
    env.robot.move_to_joint_positions(env.config.ARM.OFFSTAGE_POSITIONS)
    env.robot.reset(env.config.ARM.OFFSTAGE_POSITIONS)
    env.graspable = sim.add_body('movable_0.urdf', name='graspable')
    reward = GraspReward(name='grasp_reward', end_effector_name=env.robot.arm.name, graspable_name='graspable')
    reward.env = env
    reward.on_episode_start()
    return env, sim, phys, reward


def run(seed, env_id, aim, overrides=None):
    env, sim, phys, reward = build_env(seed, env_id, overrides or {})
    trace = []
    robot = env.robot
    for name in ('move_to_gripper_pose', 'move_to_joint_positions', 'move_along_gripper_path', 'grip'):
        def wrap(fn, name=name):
            def f(*a, **k):
                trace.append([int(sim.num_steps), name])
                phys.invalidate_ik_seed()
                return fn(*a, **k)
            return f
        setattr(robot, name, wrap(getattr(robot, name)))
    if aim:      # a grasp aimed at the object: its position, fingers across its yaw
        st = phys.w.body_state()[0, 0]
        from robovat_amd.math import rotations
        yaw = float(rotations.euler_from_quaternion(st[3:7])[2])
        action = np.array([st[0], st[1], 0.5 * (float(phys.cfg.grasp_cuboid_low[2]) + float(phys.cfg.grasp_cuboid_high[2])), yaw + aim[0]], np.float32)
        action[:2] += np.asarray(aim[1:3], np.float32)
    else:
        action = phys.w.policy_random(0)[0, 0]
    n0 = sim.num_steps
    env._execute_action(action.astype(np.float64))
    n_exec = sim.num_steps - n0
    body_exec = phys.w.body_state()[0].tolist()
    success, term = reward.get_reward()
    out = {'seed': seed, 'env_id': env_id, 'overrides': overrides or {}, 'action': [float(x) for x in action],
           'substeps_execute': int(n_exec), 'substeps_total': int(sim.num_steps - n0),
           'success': bool(success), 'termination': bool(term),
           'body_state_after_execute': body_exec,
           'body_state': phys.w.body_state()[0].tolist(), 'joint_state': phys.w.joint_state()[0].tolist(),
           'trace': [[t[0] - n0, t[1]] for t in trace]}
    print(seed, env_id, out['substeps_execute'], out['substeps_total'], out['success'], out['trace'])
    return out


def main():
    cases = [run(3, 0, (0.0, 0.0, 0.0)), run(3, 1, (0.0, 0.0, 0.0)), run(5, 2, (1.5708, 0.0, 0.0)),
             run(7, 0, None), run(7, 3, (0.3, 0.004, -0.003)), run(9, 5, (0.0, 0.03, 0.0))]
    with open(os.path.join(HERE, 'grasp_step_golden.json'), 'w') as f:
        json.dump({'about': 'reference Grasp4DofEnv._execute_action + GraspReward on the oracle physics; see gen_grasp_step_golden.py',
                   'cases': cases}, f)
    print('wrote grasp_step_golden.json')


if __name__ == '__main__':
    main()

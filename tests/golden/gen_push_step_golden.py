"""Pin the PushEnv macro-step (phase machine) against the reference's OWN methods
(build container only).

    python tests/golden/gen_push_step_golden.py

The reference's unmodified `PushEnv._execute_action` (push_env.py:631-733) and
everything it calls -- `_compute_all_waypoints`, `_is_phase_ready`,
`_get_next_phase`, `_check_singularity`, `_check_safety`,
`_check_effectiveness`, `_get_movable_status`, `Simulator.step/check_contact/
wait_until_stable`, `SawyerSim.move_to_*`, `ControllableBody.update` -- run on
top of the oracle's physics through the `OraclePhysics` plugin
(tests/golden/ref_harness.py).  The PushEnv instance is created WITHOUT running
its constructor (which needs the absent configs/assets/cameras); the attributes
the constructor would have derived from the config are set here from the same
BUILD-CHOSEN config the oracle uses (robovat_amd/configs.py).

The scene (table height, bodies, settled poses) is the oracle's own reset for the
same seed on both sides, so the only thing under test is the macro-step logic.
Outputs per macro step (actions are the oracle's Philox random policy):
substep count, is_safe / is_effective, body states, joint states, phase trace.
tests/test_push_step_golden.py replays them with orc_step_macro().

Note on exactness: the reference stores positions and orientations of `Pose` in
float32 (robovat/math/orientation.py:49, point.py) so IK targets carry 1e-7
rounding the double oracle does not have; the comparison is therefore by
tolerance, not bit-for-bit.
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)

import ref_harness as H  # noqa: E402
from ref_harness import DT, EasyDict, configs  # noqa: E402

import types  # noqa: E402
for name in ('matplotlib', 'matplotlib.pyplot'):
    if name not in sys.modules:
        sys.modules[name] = types.ModuleType(name)
sys.modules['matplotlib'].pyplot = sys.modules['matplotlib.pyplot']

from robovat.simulation.simulator import Simulator  # noqa: E402
from robovat.robots.sawyer.sawyer_sim import SawyerSim  # noqa: E402
from robovat.envs.push.push_env import PushEnv  # noqa: E402

import gym  # noqa: E402  (the stub installed by gen_golden)


def f32v(v):
    return float(np.float32(v))


def build_env(seed, env_id, overrides):
    H.OraclePhysics.SEED = seed
    H.OraclePhysics.ENV_ID = env_id
    H.OraclePhysics.CFG_OVERRIDES = overrides
    env_cfg = configs.push_env_config(**overrides)
    sim = Simulator(physics_backend='OraclePhysics', time_step=DT)
    sim.reset()
    sim.start()
    phys = sim.physics
    # the oracle's own reset builds the scene (same seed => same scene as the replay)
    phys.w.reset()
    cnt = phys.w.env_counters()[0]
    sim._num_steps = int(cnt[0]); phys._num_steps = int(cnt[0])      # steps taken while the bodies were dropped

    env = object.__new__(PushEnv)
    env._simulator = sim
    env._debug = False
    env._num_episodes = 0
    env._num_steps = 0
    env._done = False
    c = phys.cfg                                   # float32-rounded values, as the oracle holds them
    env._config = EasyDict({
        'DEBUG': False,
        'NUM_GOAL_STEPS': (int(c.num_goal_steps) or None),
        'ACTION': {'MOTION': {'TRANSLATION_X': float(c.translation_x), 'TRANSLATION_Y': float(c.translation_y)},
                   'MIN_DELTA_POSITION': float(c.min_delta_position), 'MIN_DELTA_ANGLE': float(c.min_delta_angle)},
        'ARM': {'GRIPPER_SAFE_HEIGHT': float(c.gripper_safe_height), 'FINGER_TIP_OFFSET': float(c.finger_tip_offset),
                'OFFSTAGE_POSITIONS': [float(x) for x in c.offstage_positions]},
        'SIM': {'STEPS_CHECK': int(c.steps_check), 'MAX_PHASE_STEPS': int(c.max_phase_steps),
                'MAX_MOTION_STEPS': int(c.max_motion_steps), 'MAX_OFFSTAGE_STEPS': int(c.max_offstage_steps)},
        'RECORDING': {'USE': False},
    })
    # what PushEnv.__init__ (push_env.py:70-90) derives from the config
    env.num_goal_steps = env._config.NUM_GOAL_STEPS
    low = np.array([float(x) for x in c.cspace_low]); high = np.array([float(x) for x in c.cspace_high])
    env.cspace = gym.spaces.Box(low=low, high=high)
    env.start_offset = 0.5 * (high + low)
    env.start_range = 0.5 * (high - low)
    env.start_z = float(c.finger_tip_offset) + env.start_offset[2]
    tx, ty = float(c.table_center[0]), float(c.table_center[1])
    env.table_workspace = gym.spaces.Box(
        low=np.array([tx - 0.5 * float(c.workspace_x_range), ty - 0.5 * float(c.workspace_y_range)]),
        high=np.array([tx + 0.5 * float(c.workspace_x_range), ty + 0.5 * float(c.workspace_y_range)]))
    env.phase_list = ['initial', 'pre', 'start', 'motion', 'post', 'offstage', 'done']
    env.use_recording = False
    env.layout_id = 0
    env.attributes = None
    env.max_phase_steps = None
    env.num_total_steps = env.num_unsafe = env.num_ineffective = env.num_useful = 0

    # ArmEnv._reset_scene / _reset_robot (arm_env.py:78-107)
    env.table = sim.add_body('table.urdf', is_static=True, name='table')
    robot_cfg = H.robot_config()
    env.robot = SawyerSim(simulator=sim, config=robot_cfg)
    env.robot.move_to_joint_positions(env.config.ARM.OFFSTAGE_POSITIONS)
    mask = phys.w.observe()[1][0]
    env.movable_bodies = [sim.add_body('movable_%d.urdf' % b, name='movable_%d' % b)
                          for b in range(len(mask)) if mask[b] > 0]
    env.movable_body_mask = mask
    return env, sim, phys


def run(seed, env_id, n_macro, overrides):
    env, sim, phys = build_env(seed, env_id, overrides)
    trace = []
    robot = env.robot
    for name in ('move_to_gripper_pose', 'move_to_joint_positions'):
        def wrap(fn, name=name):
            def f(*a, **k):
                trace.append([int(sim.num_steps), name, env.phase])
                phys.invalidate_ik_seed()          # a new command: IK restarts from the joint state
                return fn(*a, **k)
            return f
        setattr(robot, name, wrap(getattr(robot, name)))     # observation + plugin bookkeeping; logic untouched
    steps = []
    for k in range(n_macro):
        action = phys.w.policy_random(k)[0]                  # float32[G][4]
        if env.num_goal_steps is None:
            action = action[0]
        n0 = sim.num_steps
        del trace[:]
        # float64 copy of the float32 action: keeps `motion * TRANSLATION` in double under
        # numpy>=2 scalar promotion, as it was under the numpy 1.x the reference targets
        env._execute_action(action.astype(np.float64))
        steps.append({
            'action': np.asarray(action, np.float64).reshape(-1).tolist(),
            'substeps': int(sim.num_steps - n0),
            'is_safe': bool(env.attributes['is_safe']), 'is_effective': bool(env.attributes['is_effective']),
            'done_flag': bool(env._done),
            'body_state': phys.w.body_state()[0].tolist(),
            'joint_state': phys.w.joint_state()[0].tolist(),
            'trace': [[t[0] - n0, t[1], t[2]] for t in trace],
        })
        if env._done:
            break
    for st in steps:
        print(seed, env_id, st['substeps'], st['is_safe'], st['is_effective'], st['done_flag'],
              [(t[0], t[2]) for t in st['trace']])
    return {'seed': seed, 'env_id': env_id, 'overrides': overrides, 'n_bodies': len(env.movable_bodies), 'steps': steps}


def main():
    cases = [run(3, 0, 4, {}), run(11, 0, 4, {}), run(5, 0, 3, {'MOVABLE_NAME': 'CONCAVE'}),
             # unsafe / interrupted pushes (found by scanning the oracle): bodies stuck on the
             # gripper, arm-table contact ("singularity"), body pushed off the table => done
             run(100, 15, 2, {}), run(100, 46, 3, {}), run(100, 40, 2, {}),
             run(7, 0, 2, {'NUM_GOAL_STEPS': 2})]
    out = {'about': 'reference PushEnv._execute_action on the oracle physics; see gen_push_step_golden.py',
           'cases': cases}
    with open(os.path.join(HERE, 'push_step_golden.json'), 'w') as f:
        json.dump(out, f)
    print('wrote push_step_golden.json')


if __name__ == '__main__':
    main()

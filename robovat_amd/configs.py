"""BUILD-CHOSEN configuration values for the push / arm environments.

The reference reads these keys from ``configs/envs/push_env.yaml``,
``configs/robots/sawyer_sim.yaml`` and ``configs/policies/*.yaml``, none of
which is distributed with its source (README.md:48-59; key list in SURVEY.md
Appendix A).  The key *names* below follow the reference; every *value* is this
build's choice and is documented in DESIGN.md §6.  ``make_rv_config`` flattens
the nested dictionary into the C ``rv_config`` struct of ``include/rovat.h``.
"""
import copy
import math

import numpy as np

from robovat_amd import abi
from robovat_amd.envs.push import push_layouts


def _look_at(eye, target, up=(0.0, 0.0, 1.0)):
    """Extrinsics (R, t) with x_cam = R x_world + t of a camera at ``eye`` looking at
    ``target`` (camera z forward, x right, y down -- the Hartley-Zisserman frame the
    reference's intrinsics use, bullet_camera.py:28-83)."""
    eye, target, up = (np.asarray(v, dtype=np.float64) for v in (eye, target, up))
    z = target - eye
    z /= np.linalg.norm(z)
    x = np.cross(z, up)
    x /= np.linalg.norm(x)
    y = np.cross(z, x)
    rot = np.stack([x, y, z])
    return rot.tolist(), (-rot @ eye).tolist()


class AttrDict(dict):
    """Minimal EasyDict stand-in (attribute access, recursive)."""

    def __init__(self, d=None, **kw):
        super().__init__()
        d = dict(d or {}, **kw)
        for k, v in d.items():
            self[k] = AttrDict(v) if isinstance(v, dict) else v

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError:
            raise AttributeError(k)

    def __setattr__(self, k, v):
        self[k] = v


PUSH_ENV_CONFIG = {
    'TASK_NAME': None,              # None | 'clearing' | 'insertion' | 'crossing'
    'LAYOUT_ID': 0,
    'NUM_GOAL_STEPS': None,
    'MAX_STEPS': None,
    'SUCCESS_THRESH': 50.0,
    'DEBUG': False,
    'ACTION': {
        'CSPACE': {'LOW': [0.35, -0.35, 0.02], 'HIGH': [0.85, 0.35, 0.02]},
        'MOTION': {'TRANSLATION_X': 0.2, 'TRANSLATION_Y': 0.2},
        'MIN_DELTA_POSITION': 0.01,
        'MIN_DELTA_ANGLE': 0.05,
    },
    'ARM': {
        'FINGER_TIP_OFFSET': 0.14,
        'GRIPPER_SAFE_HEIGHT': 0.30,
        'OFFSTAGE_POSITIONS': [-1.5, -1.26, 0.00, 1.98, 0.00, 0.85, 3.3161],
    },
    'TABLE': {'X_RANGE': 0.76, 'Y_RANGE': 1.22, 'HEIGHT_RANGE': [-0.005, 0.005]},
    'SIM': {
        'TABLE': {'POSE': [[0.6, 0.0, 0.0], [0, 0, 0]], 'HALF_EXTENTS': [0.38, 0.61],
                  'THICKNESS': 0.05, 'FRICTION': 1.0},
        # the ground the table stands on (arm_env.py:85-88); z relative to the nominal table top
        'GROUND': {'Z': -0.75, 'FRICTION': 1.0},
        # arm_env.py:94-99: an optional static body behind the table (the reference's YAML with SIM.WALL.PATH / POSE is
        # not distributed: BUILD-CHOSEN slab 'wall' of scenes.default_shape_hulls at the far edge of the table).  With
        # USE set the wall takes body slot RV_MAXB - 1: MAX_MOVABLE_BODIES <= 3
        'WALL': {'USE': False, 'SHAPE': 'wall', 'SCALE': 1.0, 'POSE': [[1.0, 0.0, 0.4], [0, 0, 0]]},
        'STEPS_CHECK': 10,
        'MAX_PHASE_STEPS': 2000,
        'MAX_MOTION_STEPS': 3000,
        'MAX_OFFSTAGE_STEPS': 3000,
        'FALL_DEPTH': 0.3,
    },
    'MIN_MOVABLE_BODIES': 4,
    'MAX_MOVABLE_BODIES': 4,
    'MOVABLE_NAME': 'CONVEX',
    'MOVABLE': {
        'CONVEX': {
            'PATHS': ['box', 'cylinder16', 'hull_a', 'hull_b'],
            'TARGET_PATHS': ['box'],
            'SCALE': [0.8, 1.2], 'MASS': [0.1, 0.5], 'FRICTION': [0.3, 1.0],
            'MARGIN': 0.12,
            'POSE': {'X': [0.40, 0.80], 'Y': [-0.30, 0.30], 'Z': [0.08, 0.08],
                     'ROLL': [0.0, 0.0], 'PITCH': [0.0, 0.0],
                     'YAW': [-math.pi, math.pi]},
        },
        'CONCAVE': {
            'PATHS': ['concave_L', 'concave_T', 'concave_U', 'concave_X'],
            'TARGET_PATHS': ['concave_L', 'concave_T'],
            'SCALE': [0.8, 1.0], 'MASS': [0.1, 0.5], 'FRICTION': [0.3, 1.0],
            'MARGIN': 0.15,
            'POSE': {'X': [0.40, 0.80], 'Y': [-0.30, 0.30], 'Z': [0.08, 0.08],
                     'ROLL': [0.0, 0.0], 'PITCH': [0.0, 0.0],
                     'YAW': [-math.pi, math.pi]},
        },
    },
    'DROP': {'MASS': 0.1, 'FRICTION': 1.0, 'SAFE_HEIGHT': 0.2},
    'OBS': {'NUM_POINTS': 256, 'CROP_MIN': None, 'CROP_MAX': None},
    'USE_PRESTIGE_OBS': True,
    'USE_VISUALIZATION_OBS': False,
    # simulated Kinect2 depth camera (push_env.py:50-55).  Extrinsics in the reference's form
    # x_cam = ROTATION x_world + TRANSLATION (camera.py:76-79); BUILD-CHOSEN: the camera stands
    # across the table from the robot, 0.95 m above the table top, looking at the table centre
    'KINECT2': {'DEPTH': dict(zip(('ROTATION', 'TRANSLATION'), _look_at((1.45, 0.0, 0.95), (0.6, 0.0, 0.0))),
                              HEIGHT=424, WIDTH=512,
                              INTRINSICS=[[365.0, 0.0, 256.0], [0.0, 365.0, 212.0], [0.0, 0.0, 1.0]],
                              INTRINSICS_NOISE=None, TRANSLATION_NOISE=None, ROTATION_NOISE=None)},
    'PHYSICS': {
        'TIME_STEP': 1e-3, 'GRAVITY_Z': -9.8,
        'SOLVER_ITERS': 50, 'ERP': 0.2, 'SLOP': 0.0005, 'MARGIN': 0.001,
        'BREAKING': 0.02, 'WARMSTART': 0.85, 'MAX_PUSHOUT': 0.5,    # BREAKING = Bullet's gContactBreakingThreshold
        'LINEAR_DAMPING': 0.04, 'ANGULAR_DAMPING': 0.04,
        'CONTACT_QUERY_DIST': 0.001, 'ARM_FRICTION': 0.8,
        'SOLVER_TOL': 1e-5,      # sweeps stop early on this residual (Bullet: 50 iterations, no early exit)
        # ... an island all of whose bodies are below the sleep thresholds sweeps down to THIS residual (0: no such rule).  With
        # the 1e-5 exit a resting body creeps at ~4e-5 m/s: 8 um in the 200 substeps before it is put to sleep -- nothing --
        # but 1.3 mm per env.step() when nothing is ever put to sleep.  -1 = auto: 1e-7 without deactivation
        # (SLEEP_STEPS = 0, the reference's most likely semantics), 0 with it (the rule costs 4.5 x the sweeps at rest).
        'SOLVER_TOL_REST': -1.0,
        'SOLVER_STALL': 12,      # ... and when no new smallest residual has been seen for this many sweeps (a cycling Gauss-Seidel)
        'SLEEP_LINEAR': 0.02, 'SLEEP_ANGULAR': 0.5, 'SLEEP_STEPS': 200,
        'SLEEP_POSITION_WINDOW': 1e-3, 'SLEEP_ROTATION_WINDOW': 0.01,
        'NARROWPHASE_GATE': 1e-3, 'NARROWPHASE_MAX_AGE': 8,
        # rolling = spinning friction of the movables (tools/templates/urdf_template.xml:11-16; body.py:229)
        'ROLLING_FRICTION': 0.001,
        # the moving arm wakes a sleeping body when a collider box is this close to its hulls
        'WAKE_GAP': 0.003,
        # Bullet's deactivation rule as well: below 0.8 m/s and 1 rad/s for 2 s, touching nothing awake but the table
        'DEACTIVATION_LINEAR': 0.8, 'DEACTIVATION_ANGULAR': 1.0, 'DEACTIVATION_STEPS': 2000,
    },
}

def _grasp_env_config():
    """Grasp4DofEnv (configs/envs/grasp_4dof_env.yaml is not distributed; keys from
    grasp_4dof_env.py:63-345; values BUILD-CHOSEN)."""
    cfg = copy.deepcopy(PUSH_ENV_CONFIG)
    cfg['ENV_NAME'] = 'Grasp4DofEnv'
    cfg['MAX_STEPS'] = 1
    cfg['ACTION'] = {'TYPE': 'CUBOID',
                     # grasp centre x, y, fingertip height z (world frame); the angle is U[0, 2 pi)
                     'CUBOID': {'LOW': [0.56, -0.04, 0.012], 'HIGH': [0.64, 0.04, 0.012]},
                     'CSPACE': PUSH_ENV_CONFIG['ACTION']['CSPACE'], 'MOTION': PUSH_ENV_CONFIG['ACTION']['MOTION'],
                     'MIN_DELTA_POSITION': 0.01, 'MIN_DELTA_ANGLE': 0.05}
    cfg['ARM']['OVERHEAD_POSITIONS'] = list(SAWYER_SIM_CONFIG['LIMB_NEUTRAL_POSITIONS'])
    cfg['ARM']['GRIPPER_SAFE_HEIGHT'] = 0.30
    cfg['SIM']['MAX_ACTION_STEPS'] = 4000
    cfg['SIM']['GRASPABLE'] = {
        'PATHS': ['grasp_cube', 'grasp_bar', 'grasp_cyl'],
        'POSE': {'X': [0.58, 0.62], 'Y': [-0.02, 0.02], 'Z': [0.04, 0.04],
                 'ROLL': [0.0, 0.0], 'PITCH': [0.0, 0.0], 'YAW': [-math.pi, math.pi]},
        'SCALE': [0.9, 1.1], 'MASS': [0.05, 0.2], 'FRICTION': [0.5, 1.0],
        'RESAMPLE_N_EPISODES': 1, 'USE_RANDOM_SAMPLE': True,
        # lateral friction the env gives (finger tips, table) while it descends / lifts
        # (grasp_4dof_env.py:262-270, 282-293)
        'FRICTION_DESCEND': [0.001, 100.0], 'FRICTION_LIFT': [100.0, 1.0],
    }
    cfg['MIN_MOVABLE_BODIES'] = cfg['MAX_MOVABLE_BODIES'] = 1
    cfg['OBSERVATION'] = {'TYPE': 'depth'}
    cfg['PHYSICS'].update({'FINGER_DYNAMICS': True, 'FINGER_MASS': 0.1, 'FINGER_MAX_FORCE': 20.0})
    return cfg


SAWYER_SIM_CONFIG = {
    'LIMB_JOINT_NAMES': ['right_j%d' % i for i in range(7)],
    # hand 0.40 m above the table top (finger tips at 0.26 m): clear of the tallest movable standing on end
    'LIMB_NEUTRAL_POSITIONS': [0.03, -1.526, -0.036, 2.0, 0.0, 1.096, 3.308],
    'END_EFFCTOR_NAME': 'right_hand',
    'L_FINGER_NAME': 'right_gripper_l_finger_joint',
    'R_FINGER_NAME': 'right_gripper_r_finger_joint',
    'L_FINGER_TIP_NAME': 'right_gripper_l_finger_tip',
    'R_FINGER_TIP_NAME': 'right_gripper_r_finger_tip',
    'OPEN_GRIPPER_WHEN_RESET': True,
    'LIMB_MAX_VELOCITY_RATIO': 0.5,
    'LIMB_TIMEOUT': 15.0,
    'LIMB_POSITION_THRESHOLD': 0.008726640,
    'END_EFFECTOR_STEP': 0.1,
    'POSITION_GAIN': 0.05, 'VELOCITY_GAIN': 1.0, 'VELOCITY_THRESHOLD': 0.05,
    'IK': {'ITERS': 8, 'DAMPING': 0.05, 'RESIDUAL': 1e-4, 'MAX_STEP': 0.2},
}

HEURISTIC_PUSH_POLICY_CONFIG = {
    'ACTION': PUSH_ENV_CONFIG['ACTION'],
    'HEURISTICS': {'MAX_ATTEMPS': 20000},
}


def push_env_config(**overrides):
    cfg = AttrDict(copy.deepcopy(PUSH_ENV_CONFIG))
    for k, v in overrides.items():
        node = cfg
        parts = k.split('.')
        for p in parts[:-1]:
            node = node[p]
        node[parts[-1]] = AttrDict(v) if isinstance(v, dict) else v
    return cfg


def grasp_env_config(**overrides):
    cfg = AttrDict(_grasp_env_config())
    for k, v in overrides.items():
        node = cfg
        parts = k.split('.')
        for p in parts[:-1]:
            node = node[p]
        node[parts[-1]] = AttrDict(v) if isinstance(v, dict) else v
    return cfg


def sawyer_config(**overrides):
    cfg = AttrDict(copy.deepcopy(SAWYER_SIM_CONFIG))
    cfg.update(overrides)
    return cfg


def make_rv_config(env_cfg=None, robot_cfg=None, shape_names=None, n_envs=1,
                   env_id_offset=0, seed=0):
    """Flatten the nested configs into the C ``rv_config`` struct."""
    env_cfg = env_cfg or push_env_config()
    robot_cfg = robot_cfg or sawyer_config()
    if shape_names is None:
        from robovat_amd import scenes
        shape_names = [n for n, _ in scenes.default_shape_hulls()]
    c = abi.rv_config()
    c.n_envs = int(n_envs)
    c.env_id_offset = int(env_id_offset)
    c.seed_lo = int(seed) & 0xFFFFFFFF
    c.seed_hi = (int(seed) >> 32) & 0xFFFFFFFF
    ph = env_cfg.PHYSICS
    c.dt = ph.TIME_STEP
    c.gravity_z = ph.GRAVITY_Z
    c.arm_effort_limit = int(ph.get('ARM_EFFORT_LIMIT', 0))
    c.limb_dynamics = int(ph.get('LIMB_DYNAMICS', 0))
    gxy = ph.get('GRAVITY_XY', (0.0, 0.0))
    c.gravity_xy[0], c.gravity_xy[1] = float(gxy[0]), float(gxy[1])
    c.solver_iters = ph.SOLVER_ITERS
    c.erp, c.slop, c.margin = ph.ERP, ph.SLOP, ph.MARGIN
    c.breaking, c.warmstart, c.max_pushout = ph.BREAKING, ph.WARMSTART, ph.MAX_PUSHOUT
    c.lin_damp = float(np.float32((1.0 - ph.LINEAR_DAMPING) ** ph.TIME_STEP))
    c.ang_damp = float(np.float32((1.0 - ph.ANGULAR_DAMPING) ** ph.TIME_STEP))
    c.contact_query_dist = ph.CONTACT_QUERY_DIST
    c.solver_tol = ph.SOLVER_TOL
    c.solver_stall = int(ph.get('SOLVER_STALL', 0))
    tol_rest = float(ph.get('SOLVER_TOL_REST', -1.0))
    c.solver_tol_rest = tol_rest if tol_rest >= 0.0 else (1e-7 if int(ph.SLEEP_STEPS) == 0 else 0.0)
    c.sleep_lin, c.sleep_ang, c.sleep_steps = ph.SLEEP_LINEAR, ph.SLEEP_ANGULAR, int(ph.SLEEP_STEPS)
    c.sleep_pos_win, c.sleep_rot_win = ph.SLEEP_POSITION_WINDOW, ph.SLEEP_ROTATION_WINDOW
    c.np_gate, c.np_max_age = ph.NARROWPHASE_GATE, int(ph.NARROWPHASE_MAX_AGE)
    tb = env_cfg.SIM.TABLE
    abi.assign(c.table_center, tb.POSE[0][:2])
    abi.assign(c.table_half, tb.HALF_EXTENTS)
    c.table_thickness = tb.THICKNESS
    c.table_z = tb.POSE[0][2]
    abi.assign(c.table_height_range, env_cfg.TABLE.HEIGHT_RANGE)
    c.table_friction = tb.FRICTION
    c.arm_friction = ph.ARM_FRICTION
    c.fall_depth = env_cfg.SIM.FALL_DEPTH
    c.ground_z = tb.POSE[0][2] + env_cfg.SIM.GROUND.Z
    c.ground_friction = env_cfg.SIM.GROUND.FRICTION
    c.rolling_friction = float(ph.get('ROLLING_FRICTION', 0.0))
    c.wake_gap = float(ph.get('WAKE_GAP', 1.0))
    c.deact_lin, c.deact_ang = float(ph.get('DEACTIVATION_LINEAR', 0.8)), float(ph.get('DEACTIVATION_ANGULAR', 1.0))
    c.deact_steps = int(ph.get('DEACTIVATION_STEPS', 0))
    c.n_bodies_min = env_cfg.MIN_MOVABLE_BODIES
    c.n_bodies_max = env_cfg.MAX_MOVABLE_BODIES
    assert 1 <= c.n_bodies_min <= c.n_bodies_max <= abi.RV_MAXB
    grasp = env_cfg.get('ENV_NAME') == 'Grasp4DofEnv'
    if grasp:
        gr = env_cfg.SIM.GRASPABLE
        mv = AttrDict({'SCALE': gr.SCALE, 'MASS': gr.MASS, 'FRICTION': gr.FRICTION, 'MARGIN': 0.0, 'POSE': gr.POSE,
                       'PATHS': gr.PATHS, 'TARGET_PATHS': gr.PATHS})
        c.env_type = abi.RV_ENV_GRASP
        abi.assign(c.grasp_cuboid_low, env_cfg.ACTION.CUBOID.LOW)
        abi.assign(c.grasp_cuboid_high, env_cfg.ACTION.CUBOID.HIGH)
        abi.assign(c.overhead_positions, env_cfg.ARM.OVERHEAD_POSITIONS)
        c.max_action_steps = int(env_cfg.SIM.MAX_ACTION_STEPS)
        abi.assign(c.grasp_mu_descend, gr.FRICTION_DESCEND)
        abi.assign(c.grasp_mu_lift, gr.FRICTION_LIFT)
    else:
        mv = env_cfg.MOVABLE[env_cfg.MOVABLE_NAME.upper()]
    c.finger_dynamics = int(bool(ph.get('FINGER_DYNAMICS', False)))
    c.finger_mass = float(ph.get('FINGER_MASS', 0.1))
    c.finger_max_force = float(ph.get('FINGER_MAX_FORCE', 20.0))
    c.end_effector_step = robot_cfg.END_EFFECTOR_STEP
    abi.assign(c.scale_range, _rng(mv.SCALE))
    abi.assign(c.mass_range, _rng(mv.MASS))
    abi.assign(c.friction_range, _rng(mv.FRICTION))
    c.margin_xy = mv.MARGIN
    keys = ['X', 'Y', 'Z', 'ROLL', 'PITCH', 'YAW']
    abi.assign(c.pose_lo, [_rng(mv.POSE[k])[0] for k in keys])
    abi.assign(c.pose_hi, [_rng(mv.POSE[k])[1] for k in keys])
    c.drop_mass = env_cfg.DROP.MASS
    c.drop_friction = env_cfg.DROP.FRICTION
    c.safe_drop_height = env_cfg.DROP.SAFE_HEIGHT
    mov = [shape_names.index(p) for p in mv.PATHS]
    tgt = [shape_names.index(p) for p in mv.TARGET_PATHS]
    c.n_movable_shapes = len(mov)
    abi.assign(c.movable_shapes, mov)
    c.n_target_shapes = len(tgt)
    abi.assign(c.target_shapes, tgt)
    # layout
    task = env_cfg.TASK_NAME
    c.task = abi.TASK_IDS[task]
    c.layout_id = int(env_cfg.LAYOUT_ID)
    c.tile_size = push_layouts.SIZE
    abi.assign(c.tile_offset, push_layouts.OFFSET)
    if c.task != abi.RV_TASK_NONE:
        layout = push_layouts.TASK_NAME_TO_LAYOUTS[task][c.layout_id]
        c.use_tiles = 1
        c.tile_size = layout.size
        abi.assign(c.tile_offset, layout.offset)
        for name in ('region', 'goal', 'target', 'obstacle'):
            tiles = getattr(layout, name) or []
            assert len(tiles) <= abi.RV_MAXTILES
            setattr(c, 'n_' + name, len(tiles))
            abi.assign(getattr(c, name), tiles)
    # arm control
    c.kp, c.kd = robot_cfg.POSITION_GAIN, robot_cfg.VELOCITY_GAIN
    c.velocity_threshold = robot_cfg.VELOCITY_THRESHOLD
    c.limb_max_velocity_ratio = robot_cfg.LIMB_MAX_VELOCITY_RATIO
    c.limb_timeout = robot_cfg.LIMB_TIMEOUT
    c.limb_position_threshold = robot_cfg.LIMB_POSITION_THRESHOLD
    c.ik_iters = robot_cfg.IK.ITERS
    c.ik_damping, c.ik_residual, c.ik_max_step = (
        robot_cfg.IK.DAMPING, robot_cfg.IK.RESIDUAL, robot_cfg.IK.MAX_STEP)
    abi.assign(c.neutral_positions, robot_cfg.LIMB_NEUTRAL_POSITIONS)
    abi.assign(c.offstage_positions, env_cfg.ARM.OFFSTAGE_POSITIONS)
    c.open_gripper_when_reset = int(bool(robot_cfg.OPEN_GRIPPER_WHEN_RESET))
    # push env
    abi.assign(c.cspace_low, env_cfg.ACTION.CSPACE.LOW)
    abi.assign(c.cspace_high, env_cfg.ACTION.CSPACE.HIGH)
    c.translation_x = env_cfg.ACTION.MOTION.TRANSLATION_X
    c.translation_y = env_cfg.ACTION.MOTION.TRANSLATION_Y
    c.finger_tip_offset = env_cfg.ARM.FINGER_TIP_OFFSET
    c.gripper_safe_height = env_cfg.ARM.GRIPPER_SAFE_HEIGHT
    c.min_delta_position = env_cfg.ACTION.MIN_DELTA_POSITION
    c.min_delta_angle = env_cfg.ACTION.MIN_DELTA_ANGLE
    c.workspace_x_range = env_cfg.TABLE.X_RANGE
    c.workspace_y_range = env_cfg.TABLE.Y_RANGE
    c.steps_check = env_cfg.SIM.STEPS_CHECK
    c.max_phase_steps = env_cfg.SIM.MAX_PHASE_STEPS
    c.max_motion_steps = env_cfg.SIM.MAX_MOTION_STEPS
    c.max_offstage_steps = env_cfg.SIM.MAX_OFFSTAGE_STEPS
    c.num_goal_steps = int(env_cfg.NUM_GOAL_STEPS or 0)
    assert c.num_goal_steps <= abi.RV_MAXG
    c.max_steps = int(env_cfg.MAX_STEPS or 0)
    c.success_thresh = env_cfg.SUCCESS_THRESH
    c.num_points = env_cfg.OBS.NUM_POINTS
    cam = env_cfg.KINECT2.DEPTH
    c.cam_height, c.cam_width = int(cam.HEIGHT), int(cam.WIDTH)
    k = np.asarray(cam.INTRINSICS, dtype=np.float64).reshape(3, 3)
    abi.assign(c.cam_intrinsics, [k[0, 0], k[1, 1], k[0, 2], k[1, 2], k[0, 1]])
    abi.assign(c.cam_rotation, np.asarray(cam.ROTATION, dtype=np.float64).reshape(9).tolist())
    abi.assign(c.cam_translation, np.asarray(cam.TRANSLATION, dtype=np.float64).reshape(3).tolist())
    c.cam_near = 0.02                                    # bullet_camera.py:18
    # ArmEnv._reset_camera (arm_env.py:109-152): uniform noise on the calibration at every env.reset(), element by element;
    # each *_NOISE has the shape of what it perturbs (None = none)
    noise = np.zeros(17)
    if cam.get('INTRINSICS_NOISE') is not None:
        kn = np.abs(np.asarray(cam.INTRINSICS_NOISE, dtype=np.float64)).reshape(3, 3)
        noise[0:5] = [kn[0, 0], kn[1, 1], kn[0, 2], kn[1, 2], kn[0, 1]]
    if cam.get('ROTATION_NOISE') is not None:
        rn = np.abs(np.asarray(cam.ROTATION_NOISE, dtype=np.float64))
        if rn.size != 9:
            raise ValueError('KINECT2.DEPTH.ROTATION_NOISE must have the shape of ROTATION (3 x 3): arm_env.py:146-149 adds it element by element')
        noise[5:14] = rn.reshape(9)
    if cam.get('TRANSLATION_NOISE') is not None:
        noise[14:17] = np.abs(np.asarray(cam.TRANSLATION_NOISE, dtype=np.float64)).reshape(3)
    abi.assign(c.cam_noise, noise.tolist())
    # ArmEnv._reset_scene's wall (arm_env.py:94-99): a static body in the last body slot
    wall = env_cfg.SIM.get('WALL')
    if wall is not None and wall.get('USE'):
        from robovat_amd.math import Pose
        if c.n_bodies_max > abi.RV_MAXB - 1:
            raise ValueError('SIM.WALL.USE: the wall takes body slot %d, MAX_MOVABLE_BODIES must be <= %d' % (abi.RV_MAXB - 1, abi.RV_MAXB - 1))
        if wall.SHAPE not in shape_names:
            raise ValueError('SIM.WALL.SHAPE %r is not a shape template of the scene' % (wall.SHAPE,))
        wp = Pose(wall.POSE)
        c.wall_use = 1
        c.wall_shape = list(shape_names).index(wall.SHAPE)
        c.wall_scale = float(wall.get('SCALE', 1.0))
        abi.assign(c.wall_pose, [float(v) for v in list(wp.position) + list(wp.quaternion)])
    if env_cfg.OBS.get('CROP_MIN') is not None and env_cfg.OBS.get('CROP_MAX') is not None:
        c.use_crop = 1
        abi.assign(c.crop_min, list(env_cfg.OBS.CROP_MIN))
        abi.assign(c.crop_max, list(env_cfg.OBS.CROP_MAX))
    return c


def _rng(v):
    if isinstance(v, (int, float)):
        return [float(v), float(v)]
    return [float(v[0]), float(v[1])]

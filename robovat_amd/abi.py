"""ctypes mirror of ``include/rovat.h`` (struct layouts and capacities).

Both the product library (``librovat_hip.so``) and the test oracle take these
structures by pointer, so one Python object configures either side.
"""
import ctypes as C

RV_MAXB = 4
RV_MAXH = 4
RV_MAXV = 16
RV_MAXP = 28
RV_PC_MAXPIX = 2048
RV_MAX_SHAPES = 16
RV_NJ = 9
RV_NLIMB = 7
RV_NFRAME = 10
RV_NCOL = 10
RV_MAXTILES = 24
RV_MAXG = 4
RV_MAXQ = 8


def RV_CHILD_LINK(f):
    """child argument of rv_set_constraint_ex for frame f of the arm (include/rovat.h)"""
    return RV_MAXB + int(f)

RV_NBB = RV_MAXB * (RV_MAXB - 1) // 2
RV_NMAN = 2 * RV_MAXB + RV_NBB
RV_BODY_STRIDE = 13
RV_NCOUNTERS = 10

RV_OK, RV_ERR_VALUE, RV_ERR_STATE, RV_ERR_HIP, RV_ERR_NOTIMPL = 0, 1, 2, 3, 4
RV_TASK_NONE, RV_TASK_CLEARING, RV_TASK_INSERTION, RV_TASK_CROSSING = 0, 1, 2, 3
TASK_IDS = {None: 0, 'data_collection': 0, 'clearing': 1, 'insertion': 2,
            'crossing': 3}
PHASES = ['initial', 'pre', 'start', 'motion', 'post', 'offstage', 'done']
GRASP_PHASES = ['initial', 'overhead', 'prestart', 'start', 'end', 'postend', 'done']
RV_ENV_PUSH, RV_ENV_GRASP = 0, 1

f32, i32, u32, i64 = C.c_float, C.c_int32, C.c_uint32, C.c_int64


class rv_shape(C.Structure):
    _fields_ = [
        ('n_hulls', i32),
        ('n_verts', i32 * RV_MAXH),
        ('verts', ((f32 * 3) * RV_MAXV) * RV_MAXH),
        ('inertia_k', f32 * 3),
        ('radius', f32),
        ('n_planes', i32 * RV_MAXH),
        ('planes', ((f32 * 4) * RV_MAXP) * RV_MAXH),
    ]


class rv_arm(C.Structure):
    _fields_ = [
        ('base_pos', f32 * 3),
        ('base_quat', f32 * 4),
        ('jpos', (f32 * 3) * (RV_NLIMB + 1)),
        ('jquat', (f32 * 4) * (RV_NLIMB + 1)),
        ('q_lo', f32 * RV_NJ), ('q_hi', f32 * RV_NJ),
        ('v_max', f32 * RV_NJ),
        ('a_max', f32 * RV_NJ),
        ('finger_y0', f32 * 2),
        ('col_frame', i32 * RV_NCOL),
        ('col_center', (f32 * 3) * RV_NCOL),
        ('col_half', (f32 * 3) * RV_NCOL),
        ('inv_tau_max', f32 * RV_NJ),
        ('link_mass', f32 * (RV_NLIMB + 1)), ('link_com', (f32 * 3) * (RV_NLIMB + 1)), ('link_inertia', (f32 * 3) * (RV_NLIMB + 1)),
    ]


class rv_scene(C.Structure):
    _fields_ = [
        ('n_shapes', i32),
        ('shapes', rv_shape * RV_MAX_SHAPES),
        ('arm', rv_arm),
    ]


class rv_config(C.Structure):
    _fields_ = [
        ('n_envs', i32), ('env_id_offset', i32),
        ('seed_lo', u32), ('seed_hi', u32),
        ('dt', f32), ('gravity_z', f32),
        ('solver_iters', i32),
        ('erp', f32), ('slop', f32), ('margin', f32), ('breaking', f32),
        ('warmstart', f32), ('max_pushout', f32),
        ('lin_damp', f32), ('ang_damp', f32),
        ('contact_query_dist', f32),
        ('solver_tol', f32), ('sleep_lin', f32), ('sleep_ang', f32), ('sleep_steps', i32),
        ('sleep_pos_win', f32), ('sleep_rot_win', f32),
        ('np_gate', f32), ('np_max_age', i32),
        ('table_center', f32 * 2), ('table_half', f32 * 2),
        ('table_thickness', f32), ('table_z', f32),
        ('table_height_range', f32 * 2),
        ('table_friction', f32), ('arm_friction', f32), ('fall_depth', f32),
        ('n_bodies_min', i32), ('n_bodies_max', i32),
        ('scale_range', f32 * 2), ('mass_range', f32 * 2),
        ('friction_range', f32 * 2),
        ('margin_xy', f32),
        ('pose_lo', f32 * 6), ('pose_hi', f32 * 6),
        ('drop_mass', f32), ('drop_friction', f32), ('safe_drop_height', f32),
        ('n_movable_shapes', i32), ('movable_shapes', i32 * RV_MAX_SHAPES),
        ('n_target_shapes', i32), ('target_shapes', i32 * RV_MAX_SHAPES),
        ('task', i32), ('layout_id', i32), ('use_tiles', i32),
        ('tile_size', f32), ('tile_offset', f32 * 2),
        ('n_region', i32), ('n_goal', i32), ('n_target', i32),
        ('n_obstacle', i32),
        ('region', (f32 * 2) * RV_MAXTILES),
        ('goal', (f32 * 2) * RV_MAXTILES),
        ('target', (f32 * 2) * RV_MAXTILES),
        ('obstacle', (f32 * 2) * RV_MAXTILES),
        ('kp', f32), ('kd', f32), ('velocity_threshold', f32),
        ('limb_max_velocity_ratio', f32), ('limb_timeout', f32),
        ('limb_position_threshold', f32),
        ('ik_iters', i32),
        ('ik_damping', f32), ('ik_residual', f32), ('ik_max_step', f32),
        ('neutral_positions', f32 * RV_NLIMB),
        ('offstage_positions', f32 * RV_NLIMB),
        ('open_gripper_when_reset', i32),
        ('cspace_low', f32 * 3), ('cspace_high', f32 * 3),
        ('translation_x', f32), ('translation_y', f32),
        ('finger_tip_offset', f32), ('gripper_safe_height', f32),
        ('min_delta_position', f32), ('min_delta_angle', f32),
        ('workspace_x_range', f32), ('workspace_y_range', f32),
        ('steps_check', i32), ('max_phase_steps', i32),
        ('max_motion_steps', i32), ('max_offstage_steps', i32),
        ('num_goal_steps', i32), ('max_steps', i32),
        ('success_thresh', f32),
        ('num_points', i32),
        ('cam_height', i32), ('cam_width', i32),
        ('cam_intrinsics', f32 * 5),
        ('cam_rotation', f32 * 9),
        ('cam_translation', f32 * 3),
        ('cam_near', f32),
        ('use_crop', i32),
        ('crop_min', f32 * 3), ('crop_max', f32 * 3),
        ('env_type', i32), ('finger_dynamics', i32),
        ('finger_mass', f32), ('finger_max_force', f32),
        ('grasp_cuboid_low', f32 * 3), ('grasp_cuboid_high', f32 * 3),
        ('overhead_positions', f32 * RV_NLIMB),
        ('max_action_steps', i32), ('end_effector_step', f32),
        ('grasp_mu_descend', f32 * 2), ('grasp_mu_lift', f32 * 2),
        ('ground_z', f32), ('ground_friction', f32), ('rolling_friction', f32), ('wake_gap', f32), ('deact_lin', f32), ('deact_ang', f32), ('deact_steps', i32),
        ('gravity_xy', f32 * 2), ('arm_effort_limit', i32), ('limb_dynamics', i32), ('solver_stall', i32), ('solver_tol_rest', f32),
        ('cam_noise', f32 * 17),
        ('wall_use', i32), ('wall_shape', i32), ('wall_scale', f32), ('wall_pose', f32 * 7),
    ]


class rv_macro_stats(C.Structure):
    _fields_ = [
        ('substeps', i64), ('env_steps', i64), ('unsafe', i64),
        ('ineffective', i64), ('useful', i64), ('successes', i64),
        ('episodes_done', i64), ('max_substeps', i64), ('awake_substeps', i64),
    ]


class rv_obs_buffers(C.Structure):
    _fields_ = [
        ('d_position', C.c_void_p), ('d_body_mask', C.c_void_p),
        ('d_num_episodes', C.c_void_p), ('d_num_steps', C.c_void_p),
        ('d_layout_id', C.c_void_p), ('d_is_safe', C.c_void_p),
        ('d_is_effective', C.c_void_p), ('d_point_cloud', C.c_void_p),
        ('d_pose', C.c_void_p), ('d_pose2d', C.c_void_p), ('d_yaw_cossin', C.c_void_p),
    ]


class rv_rollout_extra(C.Structure):
    _fields_ = [('d_actions', C.c_void_p), ('d_reset', C.c_void_p), ('reset_obs', rv_obs_buffers)]


class rv_state_view(C.Structure):
    _fields_ = [
        ('d_envs', C.c_void_p), ('env_stride_bytes', i64),
        ('off_body', i64), ('off_active', i64), ('off_joint_q', i64), ('off_joint_qd', i64),
        ('off_link_pos', i64), ('off_link_quat', i64), ('off_obs_pos', i64), ('off_table_z', i64),
    ]


def assign(arr, values):
    """Copy a (nested) python/numpy sequence into a ctypes array."""
    for i, v in enumerate(values):
        if hasattr(v, '__len__'):
            assign(arr[i], v)
        else:
            arr[i] = v

/*
 * rovat.h — C ABI of librovat_hip.so, the MI355X-native batched rigid-body
 * backend behind RoboVat's `env.step()` hot path.
 *
 * This header is the drop-in boundary.  Each entry point
 * replaces the reference-side interface cited next to it (file:line relative
 * to the StanfordVL/robovat tree).  Everything is `extern "C"`, plain pointers
 * and sizes; no torch / HIP types appear in the signatures (a HIP stream is
 * passed as `void*`).  Bulk buffers named `d_*` are DEVICE pointers, buffers
 * named `h_*` are HOST pointers.  The caller owns every buffer it passes in;
 * the library never frees caller memory.
 *
 * Conventions (must match robovat/math, third_party/transformations.py:1034-1359):
 *   quaternions are xyzw, Euler angles are static-xyz ("sxyz"), poses are
 *   world-frame, SI units, float32 everywhere on the device.
 *
 * Error convention (reference: Python exceptions, bullet_physics.py:162,184,
 * 779,1286): every call returns an int status, 0 = RV_OK; rv_last_error()
 * returns a thread-local message.  The Python shim maps codes to the same
 * exception types the reference raises.
 */
#ifndef ROVAT_H_
#define ROVAT_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- compile-time capacities (one env = one wave64; state lives in LDS) ---- */
#define RV_MAXB        4   /* max movable bodies per env (MAX_MOVABLE_BODIES)      */
#define RV_MAXH        4   /* convex hulls per shape (V-HACD parts)                */
#define RV_MAXV       16   /* vertices per convex hull                             */
#define RV_MAXP       28   /* faces per convex hull (2 V - 4 for V = 16)           */
#define RV_PC_MAXPIX 2048  /* visible pixels of one body kept for point-cloud sampling (more: every stride-th one) */
#define RV_MAX_SHAPES 16   /* shape templates per scene                            */
#define RV_NJ          9   /* arm joints: 7 limb (right_j0..j6) + 2 finger         */
#define RV_NLIMB       7
#define RV_NFRAME     10   /* link frames: 7 limb links, hand, l_finger, r_finger  */
#define RV_NCOL       10   /* arm collider boxes                                   */
#define RV_MAXTILES   24   /* tiles per list of a push layout (layouts.py:25-245)  */
#define RV_MAXG        4   /* NUM_GOAL_STEPS upper bound (push_env.py:72)          */
#define RV_MAXQ        8   /* link-target pose queue (controllable_body.py:133)    */
#define RV_NBB   (RV_MAXB * (RV_MAXB - 1) / 2)
#define RV_NMAN  (2 * RV_MAXB + RV_NBB) /* manifolds: body-table, body-body, arm-body */
#define RV_BODY_STRIDE 13  /* pos3, quat4(xyzw), lin3, ang3                        */

/* status codes -> Python exceptions in robovat_amd/lib.py */
#define RV_OK            0
#define RV_ERR_VALUE     1   /* ValueError   */
#define RV_ERR_STATE     2   /* RuntimeError */
#define RV_ERR_HIP       3   /* RuntimeError (HIP runtime failure) */
#define RV_ERR_NOTIMPL   4   /* NotImplementedError */

/* task ids (push_reward.py:282-299) */
#define RV_TASK_NONE      0
#define RV_TASK_CLEARING  1
#define RV_TASK_INSERTION 2
#define RV_TASK_CROSSING  3

#define RV_ENV_PUSH  0
#define RV_ENV_GRASP 1
/* phases of Grasp4DofEnv._execute_action (grasp_4dof_env.py:300-308) */
#define RV_GPHASE_INITIAL  0
#define RV_GPHASE_OVERHEAD 1
#define RV_GPHASE_PRESTART 2
#define RV_GPHASE_START    3
#define RV_GPHASE_END      4
#define RV_GPHASE_POSTEND  5
#define RV_GPHASE_DONE     6
/* phases of PushEnv._execute_action (push_env.py:121-127) */
#define RV_PHASE_INITIAL  0
#define RV_PHASE_PRE      1
#define RV_PHASE_START    2
#define RV_PHASE_MOTION   3
#define RV_PHASE_POST     4
#define RV_PHASE_OFFSTAGE 5
#define RV_PHASE_DONE     6

/* One convex-decomposed shape template, expressed in its centre-of-mass /
 * principal-axes frame at unit scale (replaces URDF+OBJ ingest,
 * bullet_physics.py:143-186; hull format of tools/convert_obj_to_urdf.py). */
typedef struct rv_shape {
  int32_t n_hulls;
  int32_t n_verts[RV_MAXH];
  float   verts[RV_MAXH][RV_MAXV][3];
  float   inertia_k[3]; /* principal inertia per unit mass at unit scale        */
  float   radius;       /* bounding radius about the COM at unit scale          */
  /* face planes n.x <= d of every hull (same frame, unit scale), for the depth /
   * segmentation render behind SegmentedPointCloudObs (bullet_camera.py:188-235) */
  int32_t n_planes[RV_MAXH];
  float   planes[RV_MAXH][RV_MAXP][4];
} rv_shape;

/* Kinematic description of the Sawyer-like arm (replaces the URDF tree that
 * sawyer_sim.py:86-171 loads; numbers are BUILD-CHOSEN). */
typedef struct rv_arm {
  float base_pos[3];
  float base_quat[4];
  float jpos[RV_NLIMB + 1][3];   /* parent->child origin; entry 7 = fixed hand frame */
  float jquat[RV_NLIMB + 1][4];
  float q_lo[RV_NJ], q_hi[RV_NJ];
  float v_max[RV_NJ];            /* URDF velocity limits (joint.py:57)          */
  float a_max[RV_NJ];            /* effort-equivalent acceleration limits       */
  float finger_y0[2];            /* finger frame y offset in the hand frame     */
  int32_t col_frame[RV_NCOL];    /* which link frame each collider box rides on */
  float col_center[RV_NCOL][3];
  float col_half[RV_NCOL][3];
  /* 1 / (URDF effort limit) of every joint (1 / N m; fingers 1 / N), 0 = unlimited.  PyBullet's
   * POSITION_CONTROL motors are limited to the joint effort (bullet_physics.py:1061-1104: default
   * max force); with the kinematic limb the limit acts on what the arm can push with: the normal
   * impulse of an arm - body contact row is capped at dt x min_j tau_j / |J_j . n| over the joints
   * upstream of the collider (rv_config.arm_effort_limit) */
  float inv_tau_max[RV_NJ];
  /* inertial parameters of the limb (URDF <inertial>; controllable_body.py:458-466 drives PyBullet's
   * btMultiBody, which has them from the URDF): link i rides on frame i (i < 7), entry 7 = the hand
   * with the gripper; mass, centre of mass in the frame, principal moments (taken diagonal in the
   * frame).  Read by rv_config.limb_dynamics only */
  float link_mass[RV_NLIMB + 1];
  float link_com[RV_NLIMB + 1][3];
  float link_inertia[RV_NLIMB + 1][3];
} rv_arm;

typedef struct rv_scene {
  int32_t  n_shapes;
  rv_shape shapes[RV_MAX_SHAPES];
  rv_arm   arm;
} rv_scene;

/* Every config key the hot path reads, flattened.
 * Values are BUILD-CHOSEN defaults in robovat_amd/configs.py because the
 * reference's YAML files are not distributed with its source. */
typedef struct rv_config {
  /* world */
  int32_t  n_envs;
  int32_t  env_id_offset;        /* global id of env 0 on this rank            */
  uint32_t seed_lo, seed_hi;
  float    dt;                   /* simulator.py:26                            */
  float    gravity_z;            /* simulator.py:27                            */
  /* contact solver.  breaking: Bullet's gContactBreakingThreshold, a FACTOR -- the threshold of a
   * manifold is breaking x the smaller angular-motion disc of its two shapes (bounding radius of
   * a movable, half diagonal of a collider box; btCollisionShape::getContactBreakingThreshold) */
  int32_t  solver_iters;
  float    erp, slop, margin, breaking, warmstart, max_pushout;
  float    lin_damp, ang_damp;   /* per-substep velocity multipliers           */
  float    contact_query_dist;   /* check_contact threshold (simulator.py:246) */
  float    solver_tol;           /* PGS stops when max |d lambda| < tol (0 = run all iterations) */
  /* body deactivation ("sleeping", Bullet default behaviour): a body whose
   * speeds stay below the thresholds for sleep_steps substeps is put to
   * sleep until something comes near it (0 steps = never sleep) */
  float    sleep_lin, sleep_ang;
  int32_t  sleep_steps;
  /* a body that only oscillates in place (rocking on an edge) is at rest too: it
   * also goes to sleep when its pose stayed within sleep_pos_win metres and
   * sleep_rot_win (largest quaternion component change) of where it was when the
   * window opened, for sleep_steps substeps (0 = velocity test only) */
  float    sleep_pos_win, sleep_rot_win;
  /* narrow-phase gating: a pair's full GJK/feature pass is re-run only after
   * its bodies moved np_gate metres (linear + angular*radius) since the last
   * pass, when a cached point was lost, or every np_max_age-th substep; cached
   * points are refreshed every substep (np_max_age = 0: every substep) */
  float    np_gate;
  int32_t  np_max_age;
  /* table (arm_env.py:78-99; layouts.py:30) */
  float    table_center[2];
  float    table_half[2];
  float    table_thickness;
  float    table_z;
  float    table_height_range[2];
  float    table_friction;
  float    arm_friction;
  float    fall_depth;           /* bodies this far below the table are frozen */
  /* movable bodies (push_env.py:399-597) */
  int32_t  n_bodies_min, n_bodies_max;
  float    scale_range[2], mass_range[2], friction_range[2];
  float    margin_xy;
  float    pose_lo[6], pose_hi[6];
  float    drop_mass, drop_friction;
  float    safe_drop_height;
  int32_t  n_movable_shapes;
  int32_t  movable_shapes[RV_MAX_SHAPES];
  int32_t  n_target_shapes;
  int32_t  target_shapes[RV_MAX_SHAPES];
  /* layout (layouts.py:14-24) */
  int32_t  task;
  int32_t  layout_id;
  int32_t  use_tiles;
  float    tile_size;
  float    tile_offset[2];
  int32_t  n_region, n_goal, n_target, n_obstacle;
  float    region[RV_MAXTILES][2];
  float    goal[RV_MAXTILES][2];
  float    target[RV_MAXTILES][2];
  float    obstacle[RV_MAXTILES][2];
  /* arm control (controllable_body.py:14-25; sawyer_sim.py:186-308) */
  float    kp, kd;
  float    velocity_threshold;
  float    limb_max_velocity_ratio;
  float    limb_timeout;
  float    limb_position_threshold;
  int32_t  ik_iters;
  float    ik_damping, ik_residual, ik_max_step;
  float    neutral_positions[RV_NLIMB];
  float    offstage_positions[RV_NLIMB];
  int32_t  open_gripper_when_reset;
  /* push env (push_env.py:58-94, 631-937) */
  float    cspace_low[3], cspace_high[3];
  float    translation_x, translation_y;
  float    finger_tip_offset;
  float    gripper_safe_height;
  float    min_delta_position, min_delta_angle;
  float    workspace_x_range, workspace_y_range;
  int32_t  steps_check, max_phase_steps, max_motion_steps, max_offstage_steps;
  int32_t  num_goal_steps;       /* 0 = None                                   */
  int32_t  max_steps;            /* 0 = None (robot_env.py:214)                */
  float    success_thresh;
  /* observation (camera_obs.py:127-238): SegmentedPointCloudObs over the simulated
   * Kinect2 depth camera (push_env.py:50-55; bullet_camera.py:18-23).  Extrinsics as in
   * the reference: x_cam = cam_rotation * x_world + cam_translation (camera.py:76-79),
   * intrinsics fx, fy, cx, cy, skew (bullet_camera.py:47-51). */
  int32_t  num_points;
  int32_t  cam_height, cam_width;
  float    cam_intrinsics[5];
  float    cam_rotation[9];
  float    cam_translation[3];
  float    cam_near;
  int32_t  use_crop;             /* OBS.CROP_MIN / CROP_MAX (camera_obs.py:187-192)  */
  float    crop_min[3], crop_max[3];
  /* Grasp4DofEnv (grasp_4dof_env.py:63-345) and the force-limited gripper it needs */
  int32_t  env_type;             /* RV_ENV_PUSH / RV_ENV_GRASP                        */
  int32_t  finger_dynamics;      /* 1: the two finger joints are dynamic DOFs of the
                                  * contact solver, driven by POSITION_CONTROL motor rows
                                  * limited to finger_max_force (bullet_physics.py:1061-1104:
                                  * default max force = joint effort)                 */
  float    finger_mass, finger_max_force;
  float    grasp_cuboid_low[3], grasp_cuboid_high[3];   /* ACTION.CUBOID (:144-150)   */
  float    overhead_positions[RV_NLIMB];                /* ARM.OVERHEAD_POSITIONS     */
  int32_t  max_action_steps;                            /* SIM.MAX_ACTION_STEPS (:333)*/
  float    end_effector_step;                           /* sawyer_sim.py:264          */
  /* lateral friction the env gives the finger tips / the table while it descends and
   * while it lifts (grasp_4dof_env.py:262-270, 282-293)                              */
  float    grasp_mu_descend[2], grasp_mu_lift[2];
  /* the ground the table stands on (arm_env.py:85-88): a body that leaves the table lands on
   * it; below ground_z - fall_depth a body is frozen (safety net) */
  float    ground_z, ground_friction;
  /* rolling / spinning friction of a body on its support (urdf_template.xml:11-16: 0.001;
   * Body.set_dynamics passes spinning = rolling, body.py:229): a resisting angular impulse of at
   * most rolling_friction x (normal impulse of the body - table manifold) per substep */
  float    rolling_friction;
  /* a sleeping body is woken by the moving arm when a collider box comes within wake_gap of its
   * hulls (contact imminent); min(breaking threshold of the pair, wake_gap) is used.  Contact points of an AWAKE body
   * are still created at the contact-breaking distance; at larger gaps they carry no impulse, so
   * waking later changes the work, not the motion */
  float    wake_gap;
  /* Bullet's own deactivation rule (btRigidBody::updateDeactivation, gDeactivationTime): a body
   * slower than deact_lin / deact_ang for deact_steps substeps in a row that touches nothing awake
   * but the table goes to sleep as well (0 steps: rule off) */
  float    deact_lin, deact_ang;
  int32_t  deact_steps;
  /* horizontal components of the gravity vector (simulator.py:27 takes a 3-vector;
   * bullet_physics.py:129-137 set_gravity); gravity_z above is the third */
  float    gravity_xy[2];
  /* 1: arm - body contact forces are limited by the joint efforts (rv_arm.inv_tau_max) */
  int32_t  arm_effort_limit;
  /* 1: the seven limb joints are dynamic while the arm touches a body (SURVEY.md 8 f1;
   * controllable_body.py:458-466, bullet_physics.py:1061-1104): the solver of such a substep has the
   * joint velocities as unknowns next to the body velocities -- joint-space inertia M(q) of the chain
   * (composite bodies, rv_arm.link_*), contact rows with their joint-space Jacobians, and one
   * POSITION_CONTROL motor row per joint that pulls the joint back to the commanded velocity with at most
   * the joint effort (1 / rv_arm.inv_tau_max) minus what the free motion and holding the arm against
   * gravity already take.  A pad that lands on an object then stalls instead of crushing it.  0: the
   * limb is a kinematic pusher (its trajectory does not depend on contacts) */
  int32_t  limb_dynamics;
  /* > 0: an island also stops when its residual has not fallen below the smallest one seen so far for this
   * many sweeps in a row -- a Gauss-Seidel that cycles (friction rows dithering at the cone under a body the
   * arm pins to the table) instead of converging; 0: only solver_tol / solver_iters end a solve */
  int32_t  solver_stall;
  /* > 0: an island all of whose bodies were below the sleep thresholds (sleep_lin / sleep_ang) after the last
   * substep sweeps until its residual is below solver_tol_rest instead of solver_tol.  (With the plain 1e-5 N s
   * exit resting bodies creep at ~4e-5 m/s -- visible only without deactivation; Bullet runs its 50 sweeps without
   * an early exit.)  Islands that hold finger / limb motor rows keep solver_tol.  0 (or >= solver_tol): one tolerance */
  float    solver_tol_rest;
  /* ArmEnv._reset_camera (arm_env.py:109-152; push_env.py:273-280): on every env.reset() the camera of that env gets the
   * calibration above plus uniform noise in [-cam_noise, +cam_noise], element by element: [0..4] the five intrinsics
   * (fx, fy, cx, cy, skew), [5..13] the rotation matrix, [14..16] the translation (KINECT2.DEPTH.*_NOISE; 0 = none) */
  float    cam_noise[17];
  /* ArmEnv._reset_scene (arm_env.py:94-99): `if SIM.WALL.USE: self.wall = simulator.add_body(SIM.WALL.PATH, SIM.WALL.POSE,
   * is_static=True)`.  wall_use = 1: every env.reset() puts a STATIC body (mass 0: it collides with the movable bodies and
   * is seen by the cameras, nothing moves it) of shape template wall_shape, scaled by wall_scale, at wall_pose (x, y, z,
   * quaternion) into body slot RV_MAXB - 1; n_bodies_max must then leave that slot free.  0: no wall */
  int32_t  wall_use, wall_shape;
  float    wall_scale;
  float    wall_pose[7];
} rv_config;

/* Per-launch statistics of rv_step_macro / rv_reset (device-side reductions of
 * the counters push_env.py:136-141,729-733 keeps). */
typedef struct rv_macro_stats {
  int64_t substeps;      /* sum over envs of Simulator.step() calls           */
  int64_t env_steps;     /* envs that completed an env.step()                 */
  int64_t unsafe, ineffective, useful, successes, episodes_done;
  int64_t max_substeps;  /* slowest env of the launch                         */
  int64_t awake_substeps;/* substeps in which at least one body was awake     */
} rv_macro_stats;

typedef struct rv_world rv_world;

/* ---- lifecycle: Simulator.__init__/reset/start (simulator.py:23-92),
 *      BulletPhysics.__init__/reset/start (bullet_physics.py:31-104) ---- */
int  rv_create(const rv_config* cfg, const rv_scene* scene, int device, rv_world** out);
int  rv_destroy(rv_world* w);
const char* rv_last_error(void);
int  rv_set_stream(rv_world* w, void* hip_stream);
int  rv_synchronize(rv_world* w);
int  rv_num_envs(const rv_world* w);

/* ---- RobotEnv.reset (robot_env.py:204-237) = PushEnv._reset_scene +
 *      _load_movable_bodies + _sample_body_poses(_on_tiles) (push_env.py:331-597)
 *      + ArmEnv._reset_robot (arm_env.py:101-107).  d_env_mask: uint8[N], NULL = all. */
int  rv_reset(rv_world* w, const uint8_t* d_env_mask);

/* ---- RobotEnv.step (robot_env.py:239-275) = PushEnv._execute_action
 *      (push_env.py:631-733) + get_observation + get_reward, for all envs whose
 *      episode is not done.  d_actions: float[N][G][4] in [-1,1]. ---- */
int  rv_set_actions(rv_world* w, const float* d_actions);
int  rv_step_macro(rv_world* w);

/* ---- generate_episode's inner loop (episode_generation.py:44-46) for the
 *      on-device RandomPolicy: n_steps x { action = policy(obs); env.step(action) }
 *      per env inside ONE launch, so an env never waits for the slowest env of
 *      the batch between steps.  Actions are the rv_policy_random() draws for
 *      macro indices first_macro_index .. +n_steps-1; with auto_reset != 0 an
 *      env whose episode ended is reset (as rv_reset would) before its next
 *      step, otherwise it stops.  d_rewards / d_dones: optional [n_steps][N]. */
int  rv_rollout(rv_world* w, int32_t n_steps, int32_t first_macro_index, int32_t auto_reset,
                float* d_rewards, uint8_t* d_dones);
/* Partial batches (EnvPool style), for policies that run on the host and must not wait for the
 * slowest env of every batched env.step() (robot_env.py:239-275; the reference's worker
 * processes are just as independent, tools/parallel_run.py:54-90).  rv_step_begin gives the envs
 * flagged in d_mask (NULL: all) their next action ([N][G][4]); rv_step_poll advances every env
 * that is in the middle of a step by at most max_substeps Simulator.step() calls and / or about
 * max_usec microseconds of GPU time (0 = no limit) and sets d_finished[i] = 1 for the envs whose
 * env.step() completed in this launch; for those envs -- and only those -- row i of the optional
 * obs / d_reward / d_done ([N] rows, as rv_observe / rv_reward lay them out) receives what
 * env.step() returned (rows of the other envs are unspecified; their point-cloud rows are
 * zeroed).  An env.step() may take several polls; WHAT an env computes does not depend
 * on how its step is cut into launches: its trajectory is rv_step_macro's, bit for bit.  A step
 * begun on an env whose episode is over is reported finished at once (reward 0, done).
 * Mixing with the lock-step entry points: rv_step_macro, rv_rollout*, rv_step_sub and
 * rv_wait_until_stable CANCEL the pending partial step of every env they run on (the env is no
 * longer "stepping": a later poll does not resume or repeat it; the physics the cancelled step
 * already did stays done); rv_reset cancels it as well.  Begin a new step to continue.
 * Both env types: PushEnv (push_env.py:631-733) and Grasp4DofEnv (grasp_4dof_env.py:213-293: its phase loop ticks after
 * every substep; the reward's wait_until_stable is resumable like the closing wait of a push). */
int  rv_step_begin(rv_world* w, const float* d_actions /* [N][G][4] */, const uint8_t* d_mask /* [N] or NULL */);
/* on != 0: a step begun on an env whose episode is over RESETS it instead (RobotEnv.reset, robot_env.py:204-237,
 * as the loop of generate_episodes does between episodes, episode_generation.py:36-46): the next poll reports the
 * env finished with what env.reset() returns -- its observation, reward 0, done 0 -- and the action is not taken.
 * The reset is not cut by the poll's budget (a poll that resets envs lasts as long as their drop-and-settle).
 * Default: off (the env is reported finished at once with reward 0, done). */
int  rv_set_auto_reset(rv_world* w, int32_t on);
struct rv_obs_buffers;
int  rv_step_poll(rv_world* w, int32_t max_substeps, int32_t max_usec, uint8_t* d_finished /* [N] */,
                  const struct rv_obs_buffers* obs /* or NULL */, float* d_reward /* [N] or NULL */, uint8_t* d_done /* [N] or NULL */);
/* The same rollout returning what every env.step() of the loop returns
 * (robot_env.py:275: observation, reward, done): per-step observation rows
 * [n_steps][N]... in the layouts of rv_obs_buffers (NULL members are skipped;
 * steps not taken are zero rows).  The segmented point clouds of all steps are
 * rendered from per-step pose snapshots right after the stepping kernel. */
int  rv_rollout_record(rv_world* w, int32_t n_steps, int32_t first_macro_index, int32_t auto_reset,
                       float* d_rewards, uint8_t* d_dones, const struct rv_obs_buffers* step_obs /* host struct of device pointers */);
/* ---- the same loop run the way the reference runs it at scale: every env is an
 *      independent worker (tools/parallel_run.py:54-90 starts one process per
 *      env, none waits for another).  The N envs share a pool of
 *      total_env_steps env.step() calls; each env takes its next step (auto-reset
 *      when its episode ended) while the pool lasts.  The k-th step an env takes
 *      uses macro index first_macro_index + k, so every env's trajectory is the
 *      prefix of the rv_rollout trajectory of the same length.
 *      d_steps_taken: optional [N], the number of steps each env took. ---- */
int  rv_rollout_async(rv_world* w, int32_t total_env_steps, int32_t first_macro_index, int32_t* d_steps_taken);

/* ---- RandomPolicy._action (random_policy.py:14-23): U(-1,1)^(G*4) from
 *      Philox keyed by (seed, global env id, macro_index). ---- */
int  rv_policy_random(rv_world* w, int32_t macro_index, float* d_actions);
/* ---- HeuristicPushPolicy._action / HeuristicPushSampler._sample
 *      (push_policy.py:33-52, heuristic_push_sampler.py:66-123). ---- */
int  rv_policy_heuristic(rv_world* w, int32_t max_attempts, float* d_actions);

/* ---- Simulator.step x n (simulator.py:94-103): ControllableBody.update +
 *      BulletPhysics.step (bullet_physics.py:106-109), no phase machine. ---- */
int  rv_step_sub(rv_world* w, int32_t n_substeps);
/* ---- Simulator.wait_until_stable (simulator.py:325-376) over all movables. */
int  rv_wait_until_stable(rv_world* w, float lin_thresh, float ang_thresh,
                          int32_t check_after, int32_t min_stable, int32_t max_steps);

/* ---- state getters: Body.pose/linear_velocity/angular_velocity
 *      (body.py:72-125, bullet_physics.py:197-249); Joint.position/velocity
 *      (bullet_physics.py:635-663); Link.pose (bullet_physics.py:460-473). ---- */
int  rv_get_body_state(rv_world* w, float* d_out /* [N][RV_MAXB][13] */);
int  rv_set_body_state(rv_world* w, const float* d_in /* [N][RV_MAXB][13] */);
int  rv_get_body_params(rv_world* w, float* d_out /* [N][RV_MAXB][8]: active,shape,scale,mass,friction,frozen,0,0 */);
int  rv_set_body_params(rv_world* w, const float* d_in);
int  rv_get_joint_state(rv_world* w, float* d_out /* [N][RV_NJ][2] */);
int  rv_set_joint_state(rv_world* w, const float* d_in /* [N][RV_NJ][2] */);
int  rv_get_link_poses(rv_world* w, float* d_out /* [N][RV_NFRAME][7] */);
#define RV_NCOUNTERS 10
int  rv_get_env_counters(rv_world* w, int32_t* d_out /* [N][RV_NCOUNTERS]: sim_steps, num_steps, num_episodes, phase, done, is_safe, is_effective, substeps_last, awake_substeps_last, narrowphase_pairs_last */);

/* ---- ControllableBody.set_target_joint_positions / set_target_link_pose
 *      (controllable_body.py:263-345) via RobotCommand (simulator.py:226-244). */
/*      timeout (s) / threshold (rad): <= 0 selects the robot config's LIMB_TIMEOUT /
 *      LIMB_POSITION_THRESHOLD (sawyer_sim.py:201-204). */
int  rv_set_joint_targets(rv_world* w, const float* d_q /* [N][RV_NLIMB] */, float timeout, float threshold);
int  rv_set_link_target(rv_world* w, const float* d_pose /* [N][7] pos+xyzw */, float timeout, float threshold);
/* ---- SawyerSim.move_along_gripper_path (sawyer_sim.py:310-360) -> ControllableBody.set_target_link_poses
 *      (controllable_body.py:322-345): 1 <= n_poses <= RV_MAXQ gripper poses per env, followed one after the other
 *      (the next pose is taken when the IK solution of the current one is reached). */
int  rv_set_link_path(rv_world* w, const float* d_poses /* [N][n_poses][7] pos+xyzw */, int32_t n_poses, float timeout, float threshold);
/* ---- SawyerSim.is_limb_ready / is_gripper_ready (sawyer_sim.py:394-408; ControllableBody.is_ready,
 *      controllable_body.py:565-595): like the reference's query it retires link / joint targets that are done
 *      (reached, timed out, path exhausted).  The gripper is ready 0.5 s of simulated time after rv_grip. */
int  rv_get_robot_ready(rv_world* w, uint8_t* d_out /* [N][2]: limb ready, gripper ready */);
/* ControllableBody.set_max_joint_velocities (controllable_body.py:357-372), the robot command SawyerSim.move_to_joint_positions /
 * move_to_gripper_pose / move_along_gripper_path send with every motion (sawyer_sim.py:212-220, 285-293, 336-344: `speed` x
 * joint.max_velocity, default speed = LIMB_MAX_VELOCITY_RATIO): the speed limit (rad/s, > 0) of each of the seven limb joints
 * for the targets that are being followed.  rv_set_joint_targets / rv_set_link_target / rv_set_link_path put the configured
 * ratio back, so a per-call speed is set AFTER the target it belongs to (both before the next Simulator.step). */
int  rv_set_max_joint_velocities(rv_world* w, const float* d_vmax /* [N][RV_NLIMB] */);
/* ---- BulletPhysics.position_control_array (bullet_physics.py:1061-1104):
 *      POSITION_CONTROL motor targets (gains POSITION_GAIN / VELOCITY_GAIN,
 *      controllable_body.py:17-18) for the joints whose mask byte is non-zero
 *      (d_mask == NULL: all RV_NJ joints).  Bypasses the JointTarget layer. */
int  rv_set_motor_targets(rv_world* w, const float* d_q /* [N][RV_NJ] */, const uint8_t* d_mask /* [N][RV_NJ] or NULL */);
/* ---- SawyerSim.grip (sawyer_sim.py:362-392): value in [0, 1], 0 = open. */
int  rv_grip(rv_world* w, float value);
/* Simulator.add_constraint / Constraint.pose, max_force setters / remove_constraint
 * (simulator.py:166-224; bullet_physics.py:748-957 createConstraint / changeConstraint /
 * removeConstraint), for the constraint ControllableConstraint servoes
 * (controllable_constraint.py:21-170): a FIXED joint between the frame frame7 (host float[7]:
 * position + xyzw quaternion in the body frame; NULL = the body frame itself) of movable body
 * `body` and the world frame target7, applying at most max_force newtons per row.  Every env of
 * the world gets it; calling again moves the world frame (the servo does that every substep);
 * max_force < 0 removes the constraint.  (A movable child / a point-to-point joint: rv_set_constraint_ex.) */
int  rv_set_constraint(rv_world* w, int32_t body, const float* frame7, const float* target7, float max_force);
/* The same with the other arguments of BulletPhysics.add_constraint (bullet_physics.py:748-806: createConstraint(parent,
 * parentLink, child, childLink, jointType, jointAxis, parentFramePosition, childFramePosition, ...)): `child` = -1 (the
 * world) or another movable body slot -- child_frame7 is then given in the CHILD's frame and every row acts on both
 * bodies, which stay awake together; joint_type RV_JOINT_FIXED (six rows), RV_JOINT_POINT2POINT (pybullet
 * JOINT_POINT2POINT: the three linear rows, the bodies turn freely about the pivot) or RV_JOINT_PRISMATIC (the body
 * slides along the X AXIS of the frame it is tied to -- child_frame7's orientation: two linear rows across that axis
 * and the three angular rows; a jointAxis other than x is a rotation of both frames, which the HipPhysics mirror
 * applies) or RV_JOINT_REVOLUTE (the fourth type of the reference's JOINT_TYPES_MAPPING, bullet_physics.py:20-25: a hinge about
 * the X AXIS of the frame -- the three linear rows at the pivot and two angular rows across the axis).  Other pybullet joint
 * types (gear ...) are not in the reference's mapping: RV_ERR_NOTIMPL.  `child` = RV_CHILD_LINK(f): frame f of the ARM is the
 * other party (createConstraint with a (body, link) entity, bullet_physics.py:773-790: e.g. an object attached to the hand) --
 * child_frame7 is given in that link frame, which moves kinematically (the rows see the link's twist, the link takes no
 * impulse); fixed and point2point joints. */
#define RV_CHILD_LINK(f)     (RV_MAXB + (f))
#define RV_JOINT_REVOLUTE    0   /* pybullet.JOINT_REVOLUTE */
#define RV_JOINT_PRISMATIC   1   /* pybullet.JOINT_PRISMATIC */
#define RV_JOINT_FIXED       4   /* pybullet.JOINT_FIXED */
#define RV_JOINT_POINT2POINT 5   /* pybullet.JOINT_POINT2POINT */
int  rv_set_constraint_ex(rv_world* w, int32_t body, int32_t child, int32_t joint_type, const float* frame7,
                          const float* child_frame7, float max_force);
/* BulletPhysics.set_gravity (bullet_physics.py:129-137): the gravity vector of every env of
 * the world from now on (host float[3]) */
int  rv_set_gravity(rv_world* w, const float* gravity);
/* BulletPhysics.set_link_dynamics / set_body_dynamics, lateral friction only (bullet_physics.py:560-600,
 * 318-350; grasp_4dof_env.py:262-293 switches the finger-tip and table friction between the phases of a
 * grasp): the lateral friction of the two finger-tip pads and of the table top of every env of the world;
 * a negative value leaves that coefficient as it is */
int  rv_set_friction(rv_world* w, float mu_finger, float mu_table);
/* ---- ControllableBody.reset_targets (controllable_body.py:347-350): drop the
 *      link / joint targets; the motors keep their last commanded positions. */
int  rv_reset_targets(rv_world* w);
/* ---- BulletPhysics.compute_inverse_kinematics (bullet_physics.py:1203-1262). */
int  rv_compute_ik(rv_world* w, const float* d_pose /* [N][7] */, float* d_q /* [N][RV_NLIMB] */);

/* ---- Simulator.check_contact / BulletPhysics.get_contact_points
 *      (simulator.py:246-287, bullet_physics.py:1268-1304).
 *      d_out: uint8[N][2+RV_MAXB]: arm-table, arm-any-movable, arm-movable[b]. */
int  rv_query_contacts(rv_world* w, uint8_t* d_out);
/* ---- the camera calibration an env's observations are rendered with (Camera.intrinsics / translation / rotation,
 *      camera.py:150-168; the camera-calibration observations of camera_obs.py:241-320): rv_config's values plus the
 *      noise drawn at that env's last reset (cam_noise). */
int  rv_get_camera(rv_world* w, float* d_out /* [N][17]: fx, fy, cx, cy, skew, rotation[9] row-major, translation[3] */);
/* Number of manifold points per manifold slot, for parity tests. */
int  rv_get_manifold_counts(rv_world* w, int32_t* d_out /* [N][RV_NMAN] */);

/* ---- observations (push_env.py:169-236): PoseObs('position') pose_obs.py:53-73,
 *      attribute obs attribute_obs.py:16-115, SegmentedPointCloudObs
 *      camera_obs.py:182-238: ray-cast depth + segmentation, deprojection, P pixels per body
 *      emitted in the order of their Philox keys -- a random permutation, so the row order within
 *      a body's cloud carries no meaning; bodies with more than RV_PC_MAXPIX = 2048 visible pixels
 *      are sampled with a stride).  NULL pointers are skipped. */
typedef struct rv_obs_buffers {
  float*   d_position;     /* [N][RV_MAXB][3]                                   */
  float*   d_body_mask;    /* [N][RV_MAXB]                                      */
  int64_t* d_num_episodes; /* [N]                                               */
  int64_t* d_num_steps;    /* [N]                                               */
  int64_t* d_layout_id;    /* [N]                                               */
  int64_t* d_is_safe;      /* [N]                                               */
  int64_t* d_is_effective; /* [N]                                               */
  float*   d_point_cloud;  /* [N][RV_MAXB][num_points][3]                       */
  /* the other PoseObs modalities (pose_obs.py:53-73), zero rows for absent bodies */
  float*   d_pose;         /* [N][RV_MAXB][6]  position + static-xyz Euler      */
  float*   d_pose2d;       /* [N][RV_MAXB][3]  x, y, yaw                        */
  float*   d_yaw_cossin;   /* [N][RV_MAXB][2]  cos(yaw), sin(yaw)               */
} rv_obs_buffers;
int  rv_observe(rv_world* w, const rv_obs_buffers* obs);

/* ---- the complete record of a rollout: what generate_episode writes per transition
 *      (episode_generation.py:47-67: state, ACTION, reward, info, where the state of an episode's
 *      first transition is the observation env.reset() returned).  Beyond rv_rollout_record: the
 *      action every env.step() executed, and for every step that an auto-reset preceded, the
 *      observation of the freshly reset state (rows of the other steps are zero).  With them the
 *      [n_steps][N] buffers split into the reference's episodes without losing a transition. */
typedef struct rv_rollout_extra {
  float*   d_actions;        /* [n_steps][N][G][4]  (G = max(num_goal_steps, 1)); NULL: skipped */
  uint8_t* d_reset;          /* [n_steps][N]  1 = the env was reset right before this step      */
  rv_obs_buffers reset_obs;  /* [n_steps][N]...  observation env.reset() returned; NULL members skipped */
} rv_rollout_extra;
int  rv_rollout_record_full(rv_world* w, int32_t n_steps, int32_t first_macro_index, int32_t auto_reset,
                            float* d_rewards, uint8_t* d_dones, const rv_obs_buffers* step_obs,
                            const rv_rollout_extra* extra);

/* ---- CameraObs 'depth' / 'segmask' (camera_obs.py:33-88; BulletCamera._frames,
 *      bullet_camera.py:188-235) of the simulated depth camera: eye-space depth (0 where
 *      nothing is hit) and segmentation (body index, RV_MAXB = table, RV_MAXB + 1 = the arm's link boxes, 255 = nothing).
 *      The arm is drawn as its ten link collider boxes and occludes bodies.  Either pointer may be NULL. */
int  rv_render(rv_world* w, float* d_depth /* [N][cam_height][cam_width] */, uint8_t* d_segmask /* same shape */);
/* CameraObs 'rgb' (camera_obs.py:33-88; bullet_camera.py:188-235): the same ray cast, flat colours
 * per body slot / table / background, Lambert-shaded with the normal of the face that is hit under
 * one fixed directional light (BUILD-CHOSEN colours: the reference draws random rgba per body,
 * push_env.py:436).  The arm's link boxes are drawn in a flat grey. */
int  rv_render_rgb(rv_world* w, uint8_t* d_rgb /* [N][cam_height][cam_width][3] */);

/* ---- PushReward.get_reward (push_reward.py:377-405, 272-374) of the last
 *      macro step, and RobotEnv done flag (robot_env.py:257-259). ---- */
int  rv_reward(rv_world* w, float* d_reward /* [N] */, uint8_t* d_done /* [N] */);
int  rv_get_episode_returns(rv_world* w, float* d_returns /* [N] */);

/* ---- stats of the last rv_step_macro / rv_reset launch (host struct). ---- */
int  rv_get_stats(rv_world* w, rv_macro_stats* h_stats);
/* HIP-event duration of the last rv_step_macro / rv_step_sub / rv_reset kernel
 * on the world's stream, in milliseconds (used by bench.py's roofline). */
int  rv_last_kernel_ms(rv_world* w, float* h_ms);

/* ---- zero-copy, READ-ONLY views of the resident state (the
 *      scalar getters of body.py:72-125 / joint.py:41-92 batched).  The env
 *      blocks are env-major: field f of env i lives at
 *      d_envs + i * env_stride_bytes + off_f.  Valid until rv_destroy; contents
 *      change with every stepping call on the world's stream. ---- */
typedef struct rv_state_view {
  const void* d_envs;
  int64_t env_stride_bytes;
  int64_t off_body;        /* float[RV_MAXB][13]  pos3 quat4(xyzw) lin3 ang3     */
  int64_t off_active;      /* int32[RV_MAXB]                                     */
  int64_t off_joint_q;     /* float[RV_NJ]                                       */
  int64_t off_joint_qd;    /* float[RV_NJ]                                       */
  int64_t off_link_pos;    /* float[RV_NFRAME][3]                                */
  int64_t off_link_quat;   /* float[RV_NFRAME][4]                                */
  int64_t off_obs_pos;     /* float[RV_MAXB][3]   PoseObs('position') snapshot   */
  int64_t off_table_z;     /* float                                              */
} rv_state_view;
int  rv_get_state_ptrs(rv_world* w, rv_state_view* h_out);

/* sha256 (hex) of the sources the loaded binary was compiled from, baked in at
 * build time (robovat_amd/lib.py: source_hash()); __graft_entry__.smoke()
 * compares it with the hash of the sources that travelled with the binary. */
const char* rv_source_hash(void);

#ifdef __cplusplus
}
#endif
#endif /* ROVAT_H_ */

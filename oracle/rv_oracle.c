/*
 * rv_oracle.c — CPU restatement of RoboVat's `env.step()` hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Only tests/, __graft_entry__.smoke() and
 * bench.py's `cpu_baseline` leg may build, load or call this file.  The
 * product (librovat_hip.so) never links it and fails loudly without a GPU.
 *
 * PARITY STATUS
 *   - control / task logic (ControllableBody, SawyerSim, PushEnv phase machine,
 *     wait_until_stable, rewards): restated from the reference Python, pinned
 *     by golden vectors generated from the reference itself
 *     (tests/golden/gen_golden.py) and by trajectories the reference's own
 *     unmodified Simulator / SawyerSim / ControllableBody / PushEnv produce
 *     when run on this file's physics (tests/golden/gen_control_golden.py,
 *     gen_push_step_golden.py): reproduced with zero difference.
 *   - physics arithmetic (`pybullet.stepSimulation`, IK): PARITY UNPINNED.
 *     It lives in the third-party wheel pybullet==2.6.5 (requirements.txt:9),
 *     which is neither vendored under the reference tree nor installed here.
 *     The restated algorithm (GJK/EPA + persistent manifolds + PGS sequential
 *     impulses + kinematic articulated pusher + DLS IK) is specified in
 *     DESIGN.md §3 and anchored on the reference's call sites only.  What
 *     stands in for the missing reference run: analytic known answers
 *     (tests/test_kat_contact.py, test_oracle_kat.py) and an INDEPENDENT numerical
 *     pin that reads none of this file's solver or collision data
 *     (tests/test_independent_pin.py: float64 complementarity certificate of the
 *     contact solve, closed-form GJK / EPA cases, impulse-momentum bookkeeping of a
 *     whole push, FP32-vs-FP64 tolerance at the end of an env.step()).
 *     That pins the arithmetic to the published model, not to Bullet's binary.
 *
 * Build: see oracle/Makefile (float build = bit-for-bit target of the HIP
 * kernels; -DORC_DOUBLE build = pose-error oracle of record).
 */
#include <stdlib.h>
#include <stdio.h>
#ifdef _OPENMP
#include <omp.h>
#endif
#include "../include/rovat.h"
#include "orc_math.h"
#include "orc_collide.h"

#define STREAM_RESET  1u
#define STREAM_RANDOM 2u
#define STREAM_HEUR   3u

/* controllable_body.py:14-25 */
#define STEPS_TO_CHECK_DONE 100
#define STEPS_TO_UPDATE_IK  10

typedef struct { real p[3], q[4], v[3], w[3]; } orc_body;
typedef struct {
  int active, frozen, shape;
  int asleep, sleep_count, deact_count;
  int still_count; real still_ref[7];   /* pose window of the in-place oscillation test */
  int undisturbed;                      /* woken, but has not left the pose window it was sleeping in */
  real aabb[6];                         /* world box (lo, hi) of the hulls + margin, taken when the body fell asleep */
  real scale, mass, inv_mass, inv_inertia[3], friction, radius;
  /* user constraint (Simulator.add_constraint, simulator.py:166-224; bullet_physics.py:748-957):
   * a fixed joint between the frame (con_lpos, con_lquat) of this body and a frame of the world
   * (con_tpos, con_tquat), which can apply at most con_fmax N per row */
  int con_on; real con_lpos[3], con_lquat[4], con_tpos[3], con_tquat[4], con_fmax;
} orc_bparam;

/* JointTarget (controllable_body.py:28-129) */
typedef struct {
  int active, n_idx, idx[RV_NJ], has_vel, has_stop;
  int from_ik;   /* the target is the IK solution of the active link target: 1 = yes, 2 = yes and
                  * the solve ended on the residual test (a re-solve from it returns it unchanged) */
  real pos[RV_NJ];
  real start_t, stop_t, pos_thr, vel_thr;
} orc_jtarget;
/* LinkTarget (controllable_body.py:132-233) */
typedef struct {
  int active, has_pose, nq, has_stop;
  real pose[7];
  real queue[RV_MAXQ][7];
  real start_t, stop_t, pos_thr, vel_thr;
} orc_ltarget;

typedef struct {
  /* rigid bodies */
  orc_body body[RV_MAXB];
  orc_bparam bp[RV_MAXB];
  real table_z;
  int n_bodies;
  /* arm */
  int arm_enabled;
  real q[RV_NJ], qd[RV_NJ];
  int motor_on[RV_NJ];
  real motor_q[RV_NJ], motor_kp[RV_NJ], motor_kd[RV_NJ];
  real vmax_cmd[RV_NJ];          /* ControllableBody._max_joint_velocities */
  orc_jtarget jt; orc_ltarget lt;
  real gripper_ready_time;
  real fpos[RV_NFRAME][3], fquat[RV_NFRAME][4], frot[RV_NFRAME][9];
  real fv[RV_NFRAME][3], fw[RV_NFRAME][3];
  real arm_mot;                  /* largest travel of an arm collider vertex in this substep */
  real axis[RV_NLIMB][3];
  real colv[RV_NCOL][8][3], colc[RV_NCOL][3], colr[RV_NCOL];
  real colmin[RV_NCOL][3], colmax[RV_NCOL][3];   /* world AABB of each collider box */
  int arm_moving;
  /* per-substep caches */
  real rot[RV_MAXB][9], iinv[RV_MAXB][9];
  real mot[RV_MAXB];
  real wv[RV_MAXB][RV_MAXH][RV_MAXV][3];
  real tablev[8][3], groundv[8][3];
  /* contacts */
  orc_manifold man[RV_NMAN];
  int flag_arm_table, flag_arm_body[RV_MAXB];
  /* counters */
  int sim_steps;                 /* Simulator.num_steps            */
  int num_steps, num_episodes;   /* RobotEnv counters              */
  int obs_num_steps, obs_num_episodes;   /* env.attributes snapshot (push_env.py:368-375, 637-644) */
  /* lateral friction of the finger tips / the table (Link.set_dynamics, grasp_4dof_env.py:262-293) */
  real mu_finger, mu_table;
  /* the camera this env is observed with: rv_config's calibration + the noise of its last reset (arm_env.py:109-152) */
  real cam_intrinsics[5], cam_rotation[9], cam_translation[3];
  int num_action_steps;                  /* Grasp4DofEnv: substeps spent in the 'start' phase */
  real fing_dv[2], fing_vt[2], fing_qd0[2];   /* finger motors this substep: velocity step, commanded velocity, velocity after it */
  real limb_lam[RV_NLIMB];   /* impulses of the limb motor rows of the last solve that had them (diagnostic) */
  real limb_dv[RV_NLIMB], limb_vt[RV_NLIMB], limb_qd0[RV_NLIMB];   /* the same of the limb joints (rv_config.limb_dynamics) */
  int l_unsafe, l_ineffective, l_useful, l_episodes, l_successes;   /* per-launch sums (stats) */
  int done, phase, is_safe, is_effective;
  int reset_count;
  int substeps_last, awake_last, pairs_last;
  real episode_reward, last_reward;
  real action[RV_MAXG][4];
  real obs_pos[RV_MAXB][3], prev_obs_pos[RV_MAXB][3];
  int has_prev;
  /* stats */
  long num_total_steps, num_unsafe, num_ineffective, num_useful, num_successes;
  long cnt_islands, cnt_sweeps, cnt_rowsteps;   /* diagnostic (orc_debug_solver_counts): island solves, their sweeps, row steps */
} orc_env;

typedef struct orc_world {
  rv_config cfg;
  rv_scene scene;
  int n;
  orc_env* env;
  rv_macro_stats stats;
  int ext_control;   /* tests only: ControllableBody.update() is run by the caller */
  int pose_f32;      /* tests only: emulate the reference's float32 Orientation storage (orientation.py:49) */
} orc_world;

#define TIDX(b) (b)
#define BBIDX(k) (RV_MAXB + (k))
#define AIDX(b) (RV_MAXB + RV_NBB + (b))

static const int BB_A[RV_NBB] = {0, 0, 0, 1, 1, 2};
static const int BB_B[RV_NBB] = {1, 2, 3, 2, 3, 3};
/* round-robin colouring: pairs of one round touch disjoint bodies */
static const int BB_ROUND[3][2] = {{0, 5}, {1, 4}, {2, 3}};

static int body_on(const orc_env* e, int b) { return e->bp[b].active && !e->bp[b].frozen && !e->bp[b].asleep; }
/* A STATIC body (Simulator.add_body(..., is_static=True), simulator.py:195-224 -> bullet_physics.py:143-181 useFixedBase;
 * the wall of ArmEnv._reset_scene, arm_env.py:94-99): mass 0 as in Bullet -- inverse mass and inverse inertia 0.  It keeps
 * its slot among the bodies and its pair manifolds with the other bodies (its rows see a party that no impulse moves); it has
 * no manifold with the table or the arm, takes no gravity and is not integrated.  The env logic (observations, reward,
 * safety, policies, stability waits) looks at MOVABLE bodies only. */
static int body_static(const orc_env* e, int b) { return e->bp[b].inv_mass == R(0.0); }
static int body_movable(const orc_env* e, int b) { return e->bp[b].active && !body_static(e, b); }
/* Contact-breaking threshold of a manifold = rv_config.breaking x the smaller "angular motion disc" of the
 * two shapes (btCollisionShape::getContactBreakingThreshold, btPersistentManifold): a movable's disc is
 * its bounding radius about the body origin, a collider box's its half diagonal; the table's and the
 * ground's are larger than any of them. */
static real brk_body(const orc_env* e, const rv_config* c, int b) { return (real)c->breaking * (e->bp[b].radius - (real)c->margin); }
static real brk_col(const orc_world* w, int col) {
  const float* h = w->scene.arm.col_half[col];
  return (real)w->cfg.breaking * rsqrt_((real)h[0] * (real)h[0] + (real)h[1] * (real)h[1] + (real)h[2] * (real)h[2]);
}
static real brk_bb(const orc_env* e, const rv_config* c, int a, int b) { real x = brk_body(e, c, a), y = brk_body(e, c, b); return x < y ? x : y; }
static real brk_ab(const orc_world* w, const orc_env* e, int a, int col) { real x = brk_body(e, &w->cfg, a), y = brk_col(w, col); return x < y ? x : y; }
/* kind: 0 body-table, 1 body-body, 2 arm-body (col = the collider box) */
static real brk_of(const orc_world* w, const orc_env* e, int kind, int a, int b, int col) {
  return kind == 0 ? brk_body(e, &w->cfg, a) : (kind == 1 ? brk_bb(e, &w->cfg, a, b) : brk_ab(w, e, a, col));
}
/* how close a moving collider box must come to the hulls of a sleeping body to wake it (rv_config.wake_gap) */
static real wake_range(const orc_world* w, const orc_env* e, int b, int col) { real x = brk_ab(w, e, b, col), y = (real)w->cfg.wake_gap; return x < y ? x : y; }
static real sim_time(const orc_world* w, const orc_env* e) { return (real)w->cfg.dt * (real)e->sim_steps; }

/* a body entirely below the table slab can only touch the ground: its "table" manifold then
 * holds its contacts with the ground */
static int body_below_table(const orc_world* w, const orc_env* e, int b) {
  return e->body[b].p[2] + e->bp[b].radius < e->table_z - (real)w->cfg.table_thickness;
}

/* ------------------------------------------------------------------ arm -- */

/* forward kinematics of the limb for joint vector q; fills frames 0..7.
 * frame_i = frame_{i-1} o (jpos_i, jquat_i o Rz(q_i)); positions are advanced
 * with a quaternion rotation so the chain carries no matrices */
static void arm_fk_limb(const rv_arm* a, const real* q, real fpos[][3], real fquat[][4], real frot[][9], real axis[][3]) {
  real pp[3] = {(real)a->base_pos[0], (real)a->base_pos[1], (real)a->base_pos[2]};
  real pq[4] = {(real)a->base_quat[0], (real)a->base_quat[1], (real)a->base_quat[2], (real)a->base_quat[3]};
  for (int i = 0; i < RV_NLIMB; ++i) {
    real lp[3] = {(real)a->jpos[i][0], (real)a->jpos[i][1], (real)a->jpos[i][2]};
    real jq[4] = {(real)a->jquat[i][0], (real)a->jquat[i][1], (real)a->jquat[i][2], (real)a->jquat[i][3]};
    real s, c; rsincos(q[i] * R(0.5), &s, &c);
    real qz[4] = {R(0.0), R(0.0), s, c};
    real lq[4]; qmul(lq, jq, qz);
    real t[3], po[3], qf[4];
    qrotv(t, pq, lp); v3add(po, pp, t);
    qmul(qf, pq, lq);
    v3cpy(fpos[i], po); fquat[i][0] = qf[0]; fquat[i][1] = qf[1]; fquat[i][2] = qf[2]; fquat[i][3] = qf[3];
    qmat(frot[i], qf);
    axis[i][0] = frot[i][2]; axis[i][1] = frot[i][5]; axis[i][2] = frot[i][8];
    v3cpy(pp, po); pq[0] = qf[0]; pq[1] = qf[1]; pq[2] = qf[2]; pq[3] = qf[3];
  }
  real lp[3] = {(real)a->jpos[7][0], (real)a->jpos[7][1], (real)a->jpos[7][2]};
  real lq[4] = {(real)a->jquat[7][0], (real)a->jquat[7][1], (real)a->jquat[7][2], (real)a->jquat[7][3]};
  real t[3]; qrotv(t, pq, lp); v3add(fpos[7], pp, t);
  qmul(fquat[7], pq, lq);
  qmat(frot[7], fquat[7]);
}

/* full kinematic update: frames, twists, collider vertices */
static void arm_update_kinematics(const orc_world* w, orc_env* e) {
  const rv_arm* a = &w->scene.arm;
  arm_fk_limb(a, e->q, e->fpos, e->fquat, e->frot, e->axis);
  real yax[3] = {e->frot[7][1], e->frot[7][4], e->frot[7][7]};
  for (int k = 0; k < 2; ++k) {
    int f = 8 + k;
    real off = (real)a->finger_y0[k] + e->q[7 + k];
    v3madd(e->fpos[f], e->fpos[7], yax, off);
    memcpy(e->fquat[f], e->fquat[7], sizeof(real) * 4);
    memcpy(e->frot[f], e->frot[7], sizeof(real) * 9);
  }
  /* twists (base is static), frame by frame: w_f = sum_k axis_k qd_k,
   * v_f = sum_k (axis_k qd_k) x (p_f - p_k) over the joints k upstream of f;
   * the fingers add their slide along the hand's y axis */
  e->arm_mot = R(0.0);
  for (int f = 0; f < RV_NFRAME; ++f) {
    int kmax = f < RV_NLIMB ? f : RV_NLIMB - 1;
    real fw[3] = {R(0.0), R(0.0), R(0.0)}, fv[3] = {R(0.0), R(0.0), R(0.0)};
    for (int k = 0; k <= kmax; ++k) {
      real u[3], d[3], c[3];
      v3scale(u, e->axis[k], e->qd[k]);
      v3add(fw, fw, u);
      v3sub(d, e->fpos[f], e->fpos[k]); v3cross(c, u, d);
      v3add(fv, fv, c);
    }
    /* the slide of a finger along the hand's y axis (a solver DOF of its own in finger_dynamics mode) */
    if (f >= 8 && !w->cfg.finger_dynamics) v3madd(fv, fv, yax, e->qd[f - 1]);
    v3cpy(e->fv[f], fv); v3cpy(e->fw[f], fw);
    /* how far can a collider vertex riding on this frame travel in one substep */
    real ext = R(0.0);
    for (int c = 0; c < RV_NCOL; ++c) {
      if (a->col_frame[c] != f) continue;
      real cc[3] = {(real)a->col_center[c][0], (real)a->col_center[c][1], (real)a->col_center[c][2]};
      real hh[3] = {(real)a->col_half[c][0], (real)a->col_half[c][1], (real)a->col_half[c][2]};
      ext = rmax(ext, v3len(cc) + v3len(hh));
    }
    real fm = (v3len(fv) + v3len(fw) * ext) * (real)w->cfg.dt;
    if (f >= 8 && w->cfg.finger_dynamics) fm += rabs(e->qd[f - 1]) * (real)w->cfg.dt;
    e->arm_mot = rmax(e->arm_mot, fm);
  }
  for (int c = 0; c < RV_NCOL; ++c) {
    int f = a->col_frame[c];
    real cc[3] = {(real)a->col_center[c][0], (real)a->col_center[c][1], (real)a->col_center[c][2]};
    real hh[3] = {(real)a->col_half[c][0], (real)a->col_half[c][1], (real)a->col_half[c][2]};
    real t[3]; m3mulv(t, e->frot[f], cc); v3add(e->colc[c], e->fpos[f], t);
    e->colr[c] = rsqrt_(hh[0] * hh[0] + hh[1] * hh[1] + hh[2] * hh[2]) + (real)w->cfg.margin;
    for (int k = 0; k < 8; ++k) {
      real l[3] = {cc[0] + ((k & 1) ? hh[0] : -hh[0]), cc[1] + ((k & 2) ? hh[1] : -hh[1]), cc[2] + ((k & 4) ? hh[2] : -hh[2])};
      m3mulv(t, e->frot[f], l); v3add(e->colv[c][k], e->fpos[f], t);
    }
    /* world AABB of the oriented box: centre +- |R| half (no vertices needed) */
    for (int x = 0; x < 3; ++x) {
      const real ext = rabs(e->frot[f][3 * x]) * hh[0] + rabs(e->frot[f][3 * x + 1]) * hh[1] + rabs(e->frot[f][3 * x + 2]) * hh[2];
      e->colmin[c][x] = e->colc[c][x] - ext; e->colmax[c][x] = e->colc[c][x] + ext;
    }
  }
  e->arm_moving = 0;
  for (int j = 0; j < RV_NJ; ++j) if (rabs(e->qd[j]) > R(1e-3)) e->arm_moving = 1;
}

/* Damped-least-squares IK from the current joint state; restates the call
 * bullet_physics.py:1203-1262 makes (target pose of the end-effector link,
 * restPoses only => no null-space term).  out: 7 limb joint positions. */
static int arm_ik(const orc_world* w, const real* seed, const real* pose, real* out) {
  const rv_arm* a = &w->scene.arm;
  const rv_config* c = &w->cfg;
  real q[RV_NLIMB];
  int conv = 0;
  for (int i = 0; i < RV_NLIMB; ++i) q[i] = seed[i];
#ifdef ORC_TRACE_IK
  fprintf(stderr, "ik pose %.17g %.17g %.17g %.17g %.17g %.17g %.17g seed %.17g %.17g\n", (double)pose[0], (double)pose[1], (double)pose[2], (double)pose[3], (double)pose[4], (double)pose[5], (double)pose[6], (double)seed[0], (double)seed[3]);
#endif
  real fpos[RV_NFRAME][3], fquat[RV_NFRAME][4], frot[RV_NFRAME][9], axis[RV_NLIMB][3];
  for (int it = 0; it < c->ik_iters; ++it) {
    arm_fk_limb(a, q, fpos, fquat, frot, axis);
    real err[6];
    v3sub(err, pose, fpos[7]);
    real qc[4] = {-fquat[7][0], -fquat[7][1], -fquat[7][2], fquat[7][3]};
    real qe[4]; qmul(qe, pose + 3, qc);
    real sg = qe[3] < R(0.0) ? R(-2.0) : R(2.0);
    err[3] = qe[0] * sg; err[4] = qe[1] * sg; err[5] = qe[2] * sg;
    real e2 = R(0.0);
    for (int k = 0; k < 6; ++k) e2 += err[k] * err[k];
    if (e2 < (real)c->ik_residual * (real)c->ik_residual) { conv = 1; break; }
    real J[6][RV_NLIMB];
    for (int j = 0; j < RV_NLIMB; ++j) {
      real d[3], cr[3];
      v3sub(d, fpos[7], fpos[j]); v3cross(cr, axis[j], d);
      J[0][j] = cr[0]; J[1][j] = cr[1]; J[2][j] = cr[2];
      J[3][j] = axis[j][0]; J[4][j] = axis[j][1]; J[5][j] = axis[j][2];
    }
    real A[6][6];
    real lam2 = (real)c->ik_damping * (real)c->ik_damping;
    for (int r = 0; r < 6; ++r)
      for (int s = 0; s < 6; ++s) {
        real acc = R(0.0);
        for (int j = 0; j < RV_NLIMB; ++j) acc += J[r][j] * J[s][j];
        A[r][s] = acc + (r == s ? lam2 : R(0.0));
      }
    /* Cholesky A = L L^T, solve A y = err */
    real L[6][6];
    for (int r = 0; r < 6; ++r)
      for (int s = 0; s <= r; ++s) {
        real acc = A[r][s];
        for (int k = 0; k < s; ++k) acc -= L[r][k] * L[s][k];
        if (r == s) L[r][r] = rsqrt_(rmax(acc, R(1e-12)));
        else L[r][s] = acc / L[s][s];
      }
    real y[6];
    for (int r = 0; r < 6; ++r) {
      real acc = err[r];
      for (int k = 0; k < r; ++k) acc -= L[r][k] * y[k];
      y[r] = acc / L[r][r];
    }
    for (int r = 5; r >= 0; --r) {
      real acc = y[r];
      for (int k = r + 1; k < 6; ++k) acc -= L[k][r] * y[k];
      y[r] = acc / L[r][r];
    }
    real dq[RV_NLIMB], mx = R(0.0);
    for (int j = 0; j < RV_NLIMB; ++j) {
      real acc = R(0.0);
      for (int r = 0; r < 6; ++r) acc += J[r][j] * y[r];
      dq[j] = acc;
      mx = rmax(mx, rabs(acc));
    }
    real sc = mx > (real)c->ik_max_step ? (real)c->ik_max_step / mx : R(1.0);
    for (int j = 0; j < RV_NLIMB; ++j) q[j] = rclamp(q[j] + dq[j] * sc, (real)a->q_lo[j], (real)a->q_hi[j]);
  }
  for (int i = 0; i < RV_NLIMB; ++i) out[i] = q[i];
  return conv;
}

/* ------------------------------------------- ControllableBody restated -- */
static void jt_reset(orc_jtarget* t) { t->active = 0; t->n_idx = 0; t->has_stop = 0; t->from_ik = 0; }
static void lt_reset(orc_ltarget* t) { t->active = 0; t->has_pose = 0; t->nq = 0; t->has_stop = 0; }

/* JointTarget.set (controllable_body.py:91-129) */
static void jt_set(const orc_world* w, orc_env* e, int n, const int* idx, const real* pos, int has_vel,
                   int use_times, real start_t, real stop_t, real pos_thr, real vel_thr, real timeout) {
  orc_jtarget* t = &e->jt;
  t->active = 1; t->n_idx = n; t->has_vel = has_vel; t->from_ik = 0;
  for (int i = 0; i < n; ++i) { t->idx[i] = idx[i]; t->pos[i] = pos[i]; }
  if (use_times) { t->start_t = start_t; t->stop_t = stop_t; }
  else { t->start_t = sim_time(w, e); t->stop_t = t->start_t + timeout; }
  t->has_stop = 1;
  t->pos_thr = pos_thr; t->vel_thr = vel_thr;
}

/* ControllableBody.check_joints_reached (controllable_body.py:501-537) */
static int check_joints_reached(const orc_env* e) {
  const orc_jtarget* t = &e->jt;
  if (!t->active) return 1;
  for (int i = 0; i < t->n_idx; ++i) {
    int j = t->idx[i];
    real dp = t->pos[i] - e->q[j];
    int pr = rabs(dp) < t->pos_thr;
    int vr = 1;
    if (t->has_vel) { real dv = R(0.0) - e->qd[j]; vr = rabs(dv) < t->vel_thr; }
    if (!(pr && vr)) return 0;
  }
  return 1;
}
/* _check_link_target_done (controllable_body.py:415-432) */
static int check_link_target_done(const orc_world* w, const orc_env* e) {
  const orc_ltarget* t = &e->lt;
  if (!t->has_stop) return 1;
  if (sim_time(w, e) >= t->stop_t) return 1;
  if (!t->has_pose && t->nq == 0) return 1;
  return 0;
}
/* _check_joint_target_done (controllable_body.py:434-456) */
static int check_joint_target_done(const orc_world* w, const orc_env* e) {
  const orc_jtarget* t = &e->jt;
  if (!t->has_stop) return 1;
  if (sim_time(w, e) >= t->stop_t) return 1;
  if (check_joints_reached(e)) return 1;
  return 0;
}
/* LinkTarget.pop (controllable_body.py:226-233) */
static void lt_pop(orc_ltarget* t) {
  if (t->nq == 0) { lt_reset(t); return; }
  memcpy(t->pose, t->queue[0], sizeof(real) * 7);
  for (int i = 1; i < t->nq; ++i) memcpy(t->queue[i - 1], t->queue[i], sizeof(real) * 7);
  t->nq--; t->has_pose = 1;
}
/* _update_ik (controllable_body.py:468-499) */
static void update_ik(const orc_world* w, orc_env* e) {
  real qik[RV_NLIMB];
  /* the tracked target is the converged solution of this very pose: solving again
   * from it passes the residual test at once and returns it unchanged, and the
   * link target's times / thresholds have not changed either -> nothing to do */
  if (e->jt.active && e->jt.from_ik == 2) return;
  /* seed: the previous IK solution while it is still being tracked, else the
   * current joint state */
  int conv = arm_ik(w, (e->jt.active && e->jt.from_ik) ? e->jt.pos : e->q, e->lt.pose, qik);
  int idx[RV_NLIMB];
  for (int i = 0; i < RV_NLIMB; ++i) idx[i] = i;
  jt_set(w, e, RV_NLIMB, idx, qik, e->lt.nq == 0, 1, e->lt.start_t, e->lt.stop_t, e->lt.pos_thr, e->lt.vel_thr, R(0.0));
  e->jt.from_ik = conv ? 2 : 1;
}
/* _update_position_control (controllable_body.py:458-466) ->
 * setJointMotorControlArray(POSITION_CONTROL) (bullet_physics.py:1061-1104) */
static void update_position_control(const orc_world* w, orc_env* e) {
  for (int i = 0; i < e->jt.n_idx; ++i) {
    int j = e->jt.idx[i];
    e->motor_on[j] = 1; e->motor_q[j] = e->jt.pos[i];
    e->motor_kp[j] = (real)w->cfg.kp; e->motor_kd[j] = (real)w->cfg.kd;
  }
}
/* ControllableBody.update (controllable_body.py:387-413) */
static void control_update(const orc_world* w, orc_env* e) {
  int ik_updated = 0;
  if (e->lt.active) {
    if (e->sim_steps % STEPS_TO_CHECK_DONE == 0)
      if (check_link_target_done(w, e)) lt_reset(&e->lt);
  }
  if (e->lt.active) {
    if (e->sim_steps % STEPS_TO_UPDATE_IK == 0 || !e->jt.active) {
      update_ik(w, e);
      ik_updated = 1;
      if (check_joints_reached(e)) {
        lt_pop(&e->lt);                               /* next pose of the path: solve again */
        if (e->jt.from_ik == 2) e->jt.from_ik = 1;
      }
    }
  }
  if (e->jt.active) {
    if (e->sim_steps % STEPS_TO_CHECK_DONE == 0 || ik_updated)
      if (check_joint_target_done(w, e)) jt_reset(&e->jt);
  }
  if (e->jt.active) update_position_control(w, e);
}
/* ControllableBody.is_ready(joint_inds=limb) (controllable_body.py:565-595) */
static int arm_is_ready_limb(const orc_world* w, orc_env* e) {
  if (check_link_target_done(w, e)) lt_reset(&e->lt);
  if (check_joint_target_done(w, e)) jt_reset(&e->jt);
  if (e->lt.active) return 0; /* every limb joint index < end-effector link index */
  if (e->jt.active) {
    for (int i = 0; i < e->jt.n_idx; ++i) if (e->jt.idx[i] < RV_NLIMB) return 0;
  }
  return 1;
}
static void arm_reset_targets(orc_env* e) { lt_reset(&e->lt); jt_reset(&e->jt); }

/* SawyerSim.move_to_joint_positions (sawyer_sim.py:186-234) */
static void robot_move_to_joint_positions(const orc_world* w, orc_env* e, const real* pos) {
  const rv_config* c = &w->cfg;
  arm_reset_targets(e);
  for (int j = 0; j < RV_NLIMB; ++j) e->vmax_cmd[j] = (real)c->limb_max_velocity_ratio * (real)w->scene.arm.v_max[j];
  int idx[RV_NLIMB];
  for (int i = 0; i < RV_NLIMB; ++i) idx[i] = i;
  jt_set(w, e, RV_NLIMB, idx, pos, 1, 0, R(0.0), R(0.0), (real)c->limb_position_threshold, (real)c->velocity_threshold, (real)c->limb_timeout);
}
/* SawyerSim.move_to_gripper_pose, straight_line=False (sawyer_sim.py:236-308) */
static void robot_move_to_gripper_pose(const orc_world* w, orc_env* e, const real* pose) {
  const rv_config* c = &w->cfg;
  arm_reset_targets(e);
  for (int j = 0; j < RV_NLIMB; ++j) e->vmax_cmd[j] = (real)c->limb_max_velocity_ratio * (real)w->scene.arm.v_max[j];
  orc_ltarget* t = &e->lt;
  t->active = 1; t->has_pose = 1; t->nq = 0;
  memcpy(t->pose, pose, sizeof(real) * 7);
  t->start_t = sim_time(w, e); t->stop_t = t->start_t + (real)c->limb_timeout; t->has_stop = 1;
  t->pos_thr = (real)c->limb_position_threshold; t->vel_thr = (real)c->velocity_threshold;
}
/* SawyerSim.grip (sawyer_sim.py:362-392) */
static void robot_grip(const orc_world* w, orc_env* e, real value) {
  const rv_arm* a = &w->scene.arm;
  value = rclamp(value, R(0.01), R(0.99));
  real lpos = (real)a->q_hi[7] - value * ((real)a->q_hi[7] - (real)a->q_lo[7]);
  real rpos = (real)a->q_lo[8] + value * ((real)a->q_hi[8] - (real)a->q_lo[8]);
  int idx[2] = {7, 8}; real pos[2] = {lpos, rpos};
  jt_set(w, e, 2, idx, pos, 1, 0, R(0.0), R(0.0), R(0.008726640), (real)w->cfg.velocity_threshold, R(10000.0));
  e->gripper_ready_time = sim_time(w, e) + R(0.5);
}
static int robot_is_gripper_ready(const orc_world* w, const orc_env* e) { return sim_time(w, e) >= e->gripper_ready_time; }

/* joint motors of the kinematic arm: velocity-level position
 * motor, command-velocity and acceleration limited, joint limits clamped */
static void arm_motor_step(const orc_world* w, orc_env* e) {
  const rv_arm* a = &w->scene.arm;
  real dt = (real)w->cfg.dt;
  const real inv_dt = R(1.0) / dt;      /* (the velocity error is scaled by 1 / dt: a multiplication per joint and substep) */
  /* limb joints move synchronised: one common scale keeps every commanded
   * velocity within its limit, so the path is a straight line in joint space */
  real sync = R(1.0);
  for (int j = 0; j < RV_NLIMB; ++j) {
    if (!e->motor_on[j]) continue;
    real raw = rabs(e->motor_kp[j] * (e->motor_q[j] - e->q[j]) * inv_dt);
    if (raw > e->vmax_cmd[j]) sync = rmin(sync, e->vmax_cmd[j] / raw);
  }
  for (int j = 0; j < RV_NJ; ++j) {
    real vd = R(0.0);
    if (e->motor_on[j]) {
      vd = e->motor_kp[j] * (e->motor_q[j] - e->q[j]) * inv_dt;
      if (j < RV_NLIMB) vd = vd * sync;
      vd = rclamp(vd, -e->vmax_cmd[j], e->vmax_cmd[j]);
    }
    real dv = rclamp(vd - e->qd[j], -(real)a->a_max[j] * dt, (real)a->a_max[j] * dt);
    real qd = e->qd[j] + dv;
    real qn = e->q[j] + qd * dt;
    if (qn < (real)a->q_lo[j]) { qn = (real)a->q_lo[j]; qd = R(0.0); }
    if (qn > (real)a->q_hi[j]) { qn = (real)a->q_hi[j]; qd = R(0.0); }
    if (j >= RV_NLIMB) { e->fing_dv[j - RV_NLIMB] = dv; e->fing_vt[j - RV_NLIMB] = vd; e->fing_qd0[j - RV_NLIMB] = qd; }
    else { e->limb_dv[j] = dv; e->limb_vt[j] = vd; e->limb_qd0[j] = qd; }
    e->q[j] = qn; e->qd[j] = qd;
  }
}

/* --------------------------------------------------------- rigid bodies -- */
static void body_set_mass(const orc_world* w, orc_env* e, int b, real mass) {
  const rv_shape* s = &w->scene.shapes[e->bp[b].shape];
  orc_bparam* p = &e->bp[b];
  p->mass = mass; p->inv_mass = mass > R(0.0) ? R(1.0) / mass : R(0.0);      /* (mass 0: a static body) */
  real s2 = p->scale * p->scale;
  for (int k = 0; k < 3; ++k) p->inv_inertia[k] = mass > R(0.0) ? R(1.0) / (mass * s2 * (real)s->inertia_k[k]) : R(0.0);
  p->radius = (real)s->radius * p->scale + (real)w->cfg.margin;
}

static void body_prepare(const orc_world* w, orc_env* e, int b);
static void bodies_prepare(const orc_world* w, orc_env* e) {
  for (int b = 0; b < RV_MAXB; ++b) if (body_on(e, b)) body_prepare(w, e, b);
}
static void body_prepare(const orc_world* w, orc_env* e, int b) {
  {
    const rv_shape* s = &w->scene.shapes[e->bp[b].shape];
    real* m = e->rot[b];
    qmat(m, e->body[b].q);
    const real* ii = e->bp[b].inv_inertia;
    /* I^-1 world = R diag(ii) R^T */
    for (int r = 0; r < 3; ++r)
      for (int c = 0; c < 3; ++c)
        e->iinv[b][r * 3 + c] = m[r * 3 + 0] * ii[0] * m[c * 3 + 0] + m[r * 3 + 1] * ii[1] * m[c * 3 + 1] + m[r * 3 + 2] * ii[2] * m[c * 3 + 2];
    for (int h = 0; h < s->n_hulls; ++h)
      for (int i = 0; i < s->n_verts[h]; ++i) {
        real l[3] = {(real)s->verts[h][i][0] * e->bp[b].scale, (real)s->verts[h][i][1] * e->bp[b].scale, (real)s->verts[h][i][2] * e->bp[b].scale};
        real t[3]; m3mulv(t, m, l); v3add(e->wv[b][h][i], e->body[b].p, t);
      }
  }
}

static void table_prepare(const orc_world* w, orc_env* e) {
  const rv_config* c = &w->cfg;
  real mg = (real)c->margin;
  real hx = (real)c->table_half[0] - mg, hy = (real)c->table_half[1] - mg;
  real ztop = e->table_z - mg, zbot = e->table_z - (real)c->table_thickness + mg;
  for (int k = 0; k < 8; ++k) {
    e->tablev[k][0] = (real)c->table_center[0] + ((k & 1) ? hx : -hx);
    e->tablev[k][1] = (real)c->table_center[1] + ((k & 2) ? hy : -hy);
    e->tablev[k][2] = (k & 4) ? ztop : zbot;
    /* the ground: a 20 m x 20 m slab, 1 m thick, under the table */
    e->groundv[k][0] = (real)c->table_center[0] + ((k & 1) ? R(10.0) : R(-10.0));
    e->groundv[k][1] = (real)c->table_center[1] + ((k & 2) ? R(10.0) : R(-10.0));
    e->groundv[k][2] = (k & 4) ? (real)c->ground_z - mg : (real)c->ground_z - R(1.0) + mg;
  }
}

static void plane_space(const real* n, real* t1, real* t2) {
  if (rabs(n[2]) > R(0.7071067811865476)) {
    real a = n[1] * n[1] + n[2] * n[2];
    real k = R(1.0) / rsqrt_(a);
    t1[0] = R(0.0); t1[1] = -n[2] * k; t1[2] = n[1] * k;
    t2[0] = a * k; t2[1] = -n[0] * t1[2]; t2[2] = n[0] * t1[1];
  } else {
    real a = n[0] * n[0] + n[1] * n[1];
    real k = R(1.0) / rsqrt_(a);
    t1[0] = -n[1] * k; t1[1] = n[0] * k; t1[2] = R(0.0);
    t2[0] = -n[2] * t1[1]; t2[1] = n[2] * t1[0]; t2[2] = a * k;
  }
}

/* world -> local helpers for manifold points */
static void to_local_body(const orc_env* e, int b, const real* wp, real* lp) {
  real d[3]; v3sub(d, wp, e->body[b].p); m3tmulv(lp, e->rot[b], d);
}
static void to_world_body(const orc_env* e, int b, const real* lp, real* wp) {
  real t[3]; m3mulv(t, e->rot[b], lp); v3add(wp, e->body[b].p, t);
}
static void to_local_frame(const orc_env* e, int f, const real* wp, real* lp) {
  real d[3]; v3sub(d, wp, e->fpos[f]); m3tmulv(lp, e->frot[f], d);
}
static void to_world_frame(const orc_env* e, int f, const real* lp, real* wp) {
  real t[3]; m3mulv(t, e->frot[f], lp); v3add(wp, e->fpos[f], t);
}

/* kind: 0 = body-table, 1 = body-body, 2 = arm-body */
static void manifold_world_points(const orc_world* w, const orc_env* e, int kind, int a, int b, const orc_manifold* m, int i, real* wa, real* wb) {
  to_world_body(e, a, m->la[i], wa);
  if (kind == 0) v3cpy(wb, m->lb[i]);
  else if (kind == 1) to_world_body(e, b, m->lb[i], wb);
  else to_world_frame(e, w->scene.arm.col_frame[m->col[i]], m->lb[i], wb);
}

static int manifold_refresh(const orc_world* w, orc_env* e, int kind, int a, int b, orc_manifold* m) {
  int n0 = m->n;
  for (int i = m->n - 1; i >= 0; --i) {
    real wa[3], wb[3], d[3];
    const real brk = brk_of(w, e, kind, a, b, m->col[i]);
    manifold_world_points(w, e, kind, a, b, m, i, wa, wb);
    v3sub(d, wa, wb);
    real dist = v3dot(d, m->nrm[i]);
    m->dist[i] = dist;
    if (dist > brk) { orc_man_remove(m, i); continue; }
    real proj[3], dr[3];
    v3madd(proj, wa, m->nrm[i], -dist);
    v3sub(dr, wb, proj);
    if (v3dot(dr, dr) > brk * brk) orc_man_remove(m, i);
  }
  return n0 - m->n;
}

/* tangent-plane direction set used for one-shot manifold generation: the
 * frame (t1,t2) rotated by 0.37 rad so box edges rarely tie */
#define MAN_C R(0.932327)
#define MAN_S R(0.361615)
#define MAN_TAU R(0.1)
#define FEATURE_PERIOD 4

static void manifold_add_world(const orc_world* w, orc_env* e, int kind, int a, int b, int col, orc_manifold* m,
                               const real* wa, const real* wb, const real* n, real d, real brk) {
  real la[3], lb[3];
  to_local_body(e, a, wa, la);
  if (kind == 0) v3cpy(lb, wb);
  else if (kind == 1) to_local_body(e, b, wb, lb);
  else to_local_frame(e, w->scene.arm.col_frame[col], wb, lb);
  orc_man_add(m, la, lb, n, d, col, brk);
}

/* narrow phase of one convex pair: GJK/EPA witness point plus the feature
 * vertices of either hull that lie within the contact-breaking distance of
 * the other hull's support plane. */
static int collide_pair(const orc_world* w, orc_env* e, int kind, int a, int b, int col,
                        const real (*A)[3], int nA, const real (*B)[3], int nB,
                        const real* guess, orc_manifold* m, real* out_dist, const real brk, const int pair) {
  real mg = (real)w->cfg.margin;
  real n[3], dist, pa[3], pb[3];
  e->pairs_last++;
#ifdef ORC_DEBUG_PAIRS
  fprintf(stderr, "P %d %d %d %d %d\n", e->sim_steps, kind, a, b, col);
#endif
  if (!orc_gjk_epa_c(A, nA, B, nB, guess, brk + R(2.0) * mg, n, &dist, pa, pb,
                     m ? &m->gc : (orc_gjk_cache*)0, pair)) return 0;
  real d = dist - R(2.0) * mg;
  if (d > brk) return 0;
  if (!(v3dot(n, n) > R(0.5))) return 0;   /* safety net: never accept a non-unit normal */
  *out_dist = d;
  if (!m) return 1;
#ifdef ORC_DEBUG_GJK
  if (kind == 0 && n[2] < R(0.99) && !body_below_table(w, e, a))
    fprintf(stderr, "TILTED table normal: step %d body %d n (%.4f %.4f %.4f) d %.6g reason %d iters %d cache_n %d start_n %d end_n %d pa (%.5f %.5f %.5f) pb (%.5f %.5f %.5f)\n",
            e->sim_steps, a, (double)n[0], (double)n[1], (double)n[2], (double)d, orc_dbg_reason, orc_dbg_iters, orc_dbg_cache_n, orc_dbg_start_n, orc_dbg_end_n,
            (double)pa[0], (double)pa[1], (double)pa[2], (double)pb[0], (double)pb[1], (double)pb[2]);
#endif
#ifdef ORC_DEBUG_NP
  fprintf(stderr, "pair kind %d a %d: n (%.6f %.6f %.6f) dist %.6g pa (%.5f %.5f %.5f) pb (%.5f %.5f %.5f)\n", kind, a, (double)n[0], (double)n[1], (double)n[2], (double)d, (double)pa[0], (double)pa[1], (double)pa[2], (double)pb[0], (double)pb[1], (double)pb[2]);
#endif
  real wa[3], wb[3];
  v3madd(wa, pa, n, -mg); v3madd(wb, pb, n, mg);
  manifold_add_world(w, e, kind, a, b, col, m, wa, wb, n, d, brk);
  /* feature stage: only while the manifold is incomplete, or every
   * FEATURE_PERIOD-th full pass (cached points are refreshed every substep) */
  if (m->n >= 4 && m->age < FEATURE_PERIOD - 1) { m->age++; return 1; }
  m->age = 0;
  /* candidate vertices are picked along four tangent directions rotated off the plane axes (a box
   * aligned with them would offer a whole edge); a candidate is kept only inside the other hull's
   * extent along those four AND along the four plane axes themselves (an octagon: exact for a box
   * whose edges follow the axes, e.g. the table -- a body overhanging the table edge gets no
   * support beyond the edge) */
  real t1[3], t2[3], dir[8][3], extA[8], extB[8];
  plane_space(n, t1, t2);
  for (int k = 0; k < 3; ++k) {
    dir[0][k] = MAN_C * t1[k] + MAN_S * t2[k];
    dir[1][k] = MAN_C * t2[k] - MAN_S * t1[k];
    dir[2][k] = -dir[0][k];
    dir[3][k] = -dir[1][k];
    dir[4][k] = t1[k]; dir[5][k] = t2[k]; dir[6][k] = -t1[k]; dir[7][k] = -t2[k];
  }
  for (int j = 0; j < 8; ++j) {
    extA[j] = v3dot(A[orc_support(A, nA, dir[j])], dir[j]) + mg;
    extB[j] = v3dot(B[orc_support(B, nB, dir[j])], dir[j]) + mg;
  }
  for (int k = 0; k < 4; ++k) {
    real sd[3];
    /* vertex of A's contact feature, projected on B's support plane */
    v3madd(sd, dir[k], n, R(-1.0) / MAN_TAU);
    const real* va = A[orc_support(A, nA, sd)];
    real dv[3]; v3sub(dv, va, pb);
    real sep = v3dot(dv, n);
    real gap = sep - R(2.0) * mg;
#ifdef ORC_DEBUG_NP
    fprintf(stderr, "  A-cand k %d va (%.5f %.5f %.5f) gap %.6g\n", k, (double)va[0], (double)va[1], (double)va[2], (double)gap);
#endif
    if (gap <= brk) {
      real pt[3]; v3madd(pt, va, n, -sep);
      int ok = 1;
      for (int j = 0; j < 8; ++j) if (v3dot(pt, dir[j]) > extB[j]) ok = 0;
      if (ok) {
        v3madd(wa, va, n, -mg); v3madd(wb, pt, n, mg);
        manifold_add_world(w, e, kind, a, b, col, m, wa, wb, n, gap, brk);
      }
    }
    /* vertex of B's contact feature, projected on A's support plane */
    v3madd(sd, dir[k], n, R(1.0) / MAN_TAU);
    const real* vb = B[orc_support(B, nB, sd)];
    v3sub(dv, pa, vb);
    sep = v3dot(dv, n);
    gap = sep - R(2.0) * mg;
    if (gap <= brk) {
      real pt[3]; v3madd(pt, vb, n, sep);
      int ok = 1;
      for (int j = 0; j < 8; ++j) if (v3dot(pt, dir[j]) > extA[j]) ok = 0;
      if (ok) {
        v3madd(wa, pt, n, -mg); v3madd(wb, vb, n, mg);
        manifold_add_world(w, e, kind, a, b, col, m, wa, wb, n, gap, brk);
      }
    }
  }
  return 1;
}

static real sphere_aabb_dist2(const real* p, const real* lo, const real* hi) {
  real d2 = R(0.0);
  for (int k = 0; k < 3; ++k) {
    real d = R(0.0);
    if (p[k] < lo[k]) d = lo[k] - p[k];
    if (p[k] > hi[k]) d = p[k] - hi[k];
    d2 += d * d;
  }
  return d2;
}
/* squared distance between two axis-aligned boxes */
static real aabb_aabb_dist2(const real* lo_a, const real* hi_a, const real* lo_b, const real* hi_b) {
  real d2 = R(0.0);
  for (int k = 0; k < 3; ++k) {
    real d = R(0.0);
    if (hi_a[k] < lo_b[k]) d = lo_b[k] - hi_a[k];
    if (hi_b[k] < lo_a[k]) d = lo_a[k] - hi_b[k];
    d2 += d * d;
  }
  return d2;
}
static real sphere_box_dist2(const real* p, const real* c, const real* h) {
  real d2 = R(0.0);
  for (int k = 0; k < 3; ++k) {
    real d = rabs(p[k] - c[k]) - h[k];
    if (d > R(0.0)) d2 += d * d;
  }
  return d2;
}

static void collide_all(const orc_world* w, orc_env* e) {
  const rv_config* c = &w->cfg;
  real qd = (real)c->contact_query_dist;
  real tc[3] = {(real)c->table_center[0], (real)c->table_center[1], e->table_z - R(0.5) * (real)c->table_thickness};
  real th[3] = {(real)c->table_half[0], (real)c->table_half[1], R(0.5) * (real)c->table_thickness};
  int run[RV_NMAN];
  for (int i = 0; i < RV_NMAN; ++i) run[i] = 1;
  /* refresh */
  for (int b = 0; b < RV_MAXB; ++b) {
    if (!(e->bp[b].active && !e->bp[b].frozen)) { e->man[TIDX(b)].n = 0; e->man[AIDX(b)].n = 0; continue; }
    if (body_static(e, b)) { e->man[TIDX(b)].n = 0; e->man[AIDX(b)].n = 0; run[TIDX(b)] = 0; run[AIDX(b)] = 0; continue; }   /* a static body: pair manifolds only */
    if (e->bp[b].asleep) continue; /* manifolds of a sleeping body stay frozen */
    {
      orc_manifold* m = &e->man[TIDX(b)];
      int lost = manifold_refresh(w, e, 0, b, -1, m);
      m->acc += e->mot[b];
      /* (a body on fewer than three support points is rocking or tipping: its support is looked at every substep) */
      run[TIDX(b)] = (c->np_max_age <= 0) || m->n < 3 || lost > 0 || m->acc > (real)c->np_gate || (e->sim_steps % c->np_max_age) == 0;
    }
    if (e->arm_enabled) {
      /* arm - body pairs are gated the same way, on the body's plus the arm's travel */
      orc_manifold* m = &e->man[AIDX(b)];
      int lost = manifold_refresh(w, e, 2, b, -1, m);
      m->acc += e->mot[b] + e->arm_mot;
      run[AIDX(b)] = (c->np_max_age <= 0) || m->n == 0 || lost > 0 || m->acc > (real)c->np_gate || (e->sim_steps % c->np_max_age) == 0;
      if (run[AIDX(b)]) m->acc = R(0.0);
    } else e->man[AIDX(b)].n = 0;
  }
  for (int k = 0; k < RV_NBB; ++k) {
    int a = BB_A[k], b = BB_B[k];
    int on = e->bp[a].active && !e->bp[a].frozen && e->bp[b].active && !e->bp[b].frozen && !(body_static(e, a) && body_static(e, b));
    if (!on) { e->man[BBIDX(k)].n = 0; run[BBIDX(k)] = 0; continue; }
    if (e->bp[a].asleep || e->bp[b].asleep) continue;
    {
      orc_manifold* m = &e->man[BBIDX(k)];
      int lost = manifold_refresh(w, e, 1, a, b, m);
      m->acc += e->mot[a] + e->mot[b];
      run[BBIDX(k)] = (c->np_max_age <= 0) || m->n == 0 || lost > 0 || m->acc > (real)c->np_gate || (e->sim_steps % c->np_max_age) == 0;
    }
  }
  /* body - table */
  for (int b = 0; b < RV_MAXB; ++b) {
    if (!body_on(e, b) || !run[TIDX(b)]) continue;
    e->man[TIDX(b)].acc = R(0.0);
    real r = e->bp[b].radius + brk_body(e, c, b);
    const int below = body_below_table(w, e, b);
    real guess[3] = {R(0.0), R(0.0), R(1.0)};
    if (below) {
      if (e->body[b].p[2] - (real)c->ground_z >= r) continue;
    } else {
      if (sphere_box_dist2(e->body[b].p, tc, th) >= r * r) continue;
      guess[2] = e->body[b].p[2] - tc[2];
    }
    const rv_shape* s = &w->scene.shapes[e->bp[b].shape];
    for (int h = 0; h < s->n_hulls; ++h) {
      real d;
      collide_pair(w, e, 0, b, -1, -1, (const real(*)[3])e->wv[b][h], s->n_verts[h], (const real(*)[3])(below ? e->groundv : e->tablev), 8, guess, &e->man[TIDX(b)], &d, brk_body(e, c, b), h + (below ? 64 : 0));
    }
  }
  /* body - body */
  for (int k = 0; k < RV_NBB; ++k) {
    int a = BB_A[k], b = BB_B[k];
    if (!(body_on(e, a) && body_on(e, b)) || !run[BBIDX(k)]) continue;
    e->man[BBIDX(k)].acc = R(0.0);
    real d[3]; v3sub(d, e->body[a].p, e->body[b].p);
    real r = e->bp[a].radius + e->bp[b].radius + brk_bb(e, c, a, b);
    if (v3dot(d, d) >= r * r) continue;
    const rv_shape* sa = &w->scene.shapes[e->bp[a].shape];
    const rv_shape* sb = &w->scene.shapes[e->bp[b].shape];
    for (int ha = 0; ha < sa->n_hulls; ++ha)
      for (int hb = 0; hb < sb->n_hulls; ++hb) {
        real dd;
        collide_pair(w, e, 1, a, b, -1, (const real(*)[3])e->wv[a][ha], sa->n_verts[ha], (const real(*)[3])e->wv[b][hb], sb->n_verts[hb], d, &e->man[BBIDX(k)], &dd, brk_bb(e, c, a, b), ha * 8 + hb);
      }
  }
  /* arm */
  e->flag_arm_table = 0;
  for (int b = 0; b < RV_MAXB; ++b) e->flag_arm_body[b] = 0;
  if (e->arm_enabled) {
    for (int col = 0; col < RV_NCOL; ++col) {
      for (int b = 0; b < RV_MAXB; ++b) {
        if (!body_on(e, b) || !run[AIDX(b)]) continue;
        real d[3]; v3sub(d, e->body[b].p, e->colc[col]);
        real r = e->bp[b].radius + brk_ab(w, e, b, col);
#ifdef ORC_DEBUG_PAIRS
        if (e->sim_steps >= 9640 && e->sim_steps <= 9654 && b == 2 && col == 8) fprintf(stderr, "ORC step %d colmin %.6f %.6f %.6f colmax %.6f %.6f %.6f body %.6f %.6f %.6f r %.6g d2 %.6g run %d\n", e->sim_steps, (double)e->colmin[col][0], (double)e->colmin[col][1], (double)e->colmin[col][2], (double)e->colmax[col][0], (double)e->colmax[col][1], (double)e->colmax[col][2], (double)e->body[b].p[0], (double)e->body[b].p[1], (double)e->body[b].p[2], (double)r, (double)sphere_aabb_dist2(e->body[b].p, e->colmin[col], e->colmax[col]), run[AIDX(b)]);
#endif
        if (sphere_aabb_dist2(e->body[b].p, e->colmin[col], e->colmax[col]) >= r * r) continue;
        const rv_shape* s = &w->scene.shapes[e->bp[b].shape];
        for (int h = 0; h < s->n_hulls; ++h) {
          real dd;
          collide_pair(w, e, 2, b, -1, col, (const real(*)[3])e->wv[b][h], s->n_verts[h], (const real(*)[3])e->colv[col], 8, d, &e->man[AIDX(b)], &dd, brk_ab(w, e, b, col), col * 8 + h);
        }
      }
      /* arm - table: detection only (push_env.py:839-855) */
      real r = e->colr[col] + brk_col(w, col);
      const real minz = e->colmin[col][2];
      /* rejection: the flag needs dist < query_dist and dist >= minz - table_z - margin */
      if (minz - e->table_z - (real)c->margin >= qd) continue;
      if (sphere_box_dist2(e->colc[col], tc, th) < r * r) {
        real guess[3] = {R(0.0), R(0.0), R(1.0)}, dd;
        if (collide_pair(w, e, 0, 0, -1, col, (const real(*)[3])e->colv[col], 8, (const real(*)[3])e->tablev, 8, guess, NULL, &dd, brk_col(w, col), 0))
          if (dd < qd) e->flag_arm_table = 1;
      }
    }
    for (int b = 0; b < RV_MAXB; ++b) {
      const orc_manifold* m = &e->man[AIDX(b)];
      for (int i = 0; i < m->n; ++i) if (m->dist[i] < qd) e->flag_arm_body[b] = 1;
    }
  }
}

/* ---------------------------------------------------------------- PGS ---- */
typedef struct {
  real dir[3][3];     /* n, t1, t2 */
  real rxa[3][3], rxb[3][3];
  real aa[3][3], ab[3][3];
  real invk[3];
  real vbc[3];        /* dir . (kinematic surface velocity) */
  real target, mu;
  real jf[3]; int fidx;   /* dynamic finger (rv_config.finger_dynamics): Jacobian on finger joint 7 + fidx; -1: none */
  real cap;           /* largest normal impulse the row may carry (arm effort limit); 1e30: none */
} orc_row;

static void row_setup(const orc_world* w, orc_env* e, int kind, int a, int b, const orc_manifold* m, int i, orc_row* r) {
  const rv_config* c = &w->cfg;
  real dt = (real)c->dt;
  real wa[3], wb[3], ra[3], rb[3];
  manifold_world_points(w, e, kind, a, b, m, i, wa, wb);
  v3sub(ra, wa, e->body[a].p);
  v3cpy(r->dir[0], m->nrm[i]);
  plane_space(r->dir[0], r->dir[1], r->dir[2]);
  real vb_pt[3] = {R(0.0), R(0.0), R(0.0)};
  real imb = R(0.0);
  if (kind == 1) { v3sub(rb, wb, e->body[b].p); imb = e->bp[b].inv_mass; }
  else { rb[0] = rb[1] = rb[2] = R(0.0); }
  if (kind == 2) {
    int f = w->scene.arm.col_frame[m->col[i]];
    real d[3], cr[3];
    v3sub(d, wb, e->fpos[f]); v3cross(cr, e->fw[f], d); v3add(vb_pt, e->fv[f], cr);
  }
  /* a point on a finger pad of the force-limited gripper: the finger joint is a solver DOF */
  const int fing = c->finger_dynamics && kind == 2 && m->col[i] >= 8;
  const real fy[3] = {e->frot[7][1], e->frot[7][4], e->frot[7][7]};
  for (int k = 0; k < 3; ++k) {
    v3cross(r->rxa[k], ra, r->dir[k]);
    m3mulv(r->aa[k], e->iinv[a], r->rxa[k]);
    real kk = e->bp[a].inv_mass + v3dot(r->rxa[k], r->aa[k]);
    if (kind == 1) {
      v3cross(r->rxb[k], rb, r->dir[k]);
      m3mulv(r->ab[k], e->iinv[b], r->rxb[k]);
      kk += imb + v3dot(r->rxb[k], r->ab[k]);
    } else {
      r->rxb[k][0] = r->rxb[k][1] = r->rxb[k][2] = R(0.0);
      r->ab[k][0] = r->ab[k][1] = r->ab[k][2] = R(0.0);
    }
    real jf = R(0.0);
    if (fing) { jf = -v3dot(r->dir[k], fy); kk += jf * jf / (real)c->finger_mass; }
    r->invk[k] = R(1.0) / kk;
    r->vbc[k] = v3dot(r->dir[k], vb_pt);
    r->jf[k] = jf;
  }
  r->fidx = fing ? m->col[i] - 8 : -1;
  /* what the arm can push with along the normal: min over the joints upstream of the collider of
   * tau_j / |J_j . n| (J_j = axis_j x (p - p_j); a finger pad also slides along the hand's y) */
  r->cap = R(1e30);
  if (kind == 2 && c->arm_effort_limit) {
    const rv_arm* arm = &w->scene.arm;
    const int f = arm->col_frame[m->col[i]];
    const int fl = f < RV_NLIMB ? f : RV_NLIMB - 1;
    real worst = R(0.0);
    for (int j = 0; j < RV_NLIMB; ++j) {
      if (j > fl) continue;
      real d[3], lever[3];
      v3sub(d, wb, e->fpos[j]); v3cross(lever, e->axis[j], d);
      worst = rmax(worst, rabs(v3dot(lever, r->dir[0])) * (real)arm->inv_tau_max[j]);
    }
    if (f >= 8 && !c->finger_dynamics) worst = rmax(worst, rabs(v3dot(fy, r->dir[0])) * (real)arm->inv_tau_max[f - 1]);
    /* (the budget is shared equally by the points of the manifold) */
    if (worst > R(0.0)) r->cap = dt / (worst * (real)m->n);
  }
  real dist = m->dist[i];
  if (dist > R(0.0)) r->target = -dist / dt;
  else r->target = rmin((real)c->erp * rmax(-dist - (real)c->slop, R(0.0)) / dt, (real)c->max_pushout);
  real mub = (kind == 0) ? (body_below_table(w, e, a) ? (real)c->ground_friction : e->mu_table) : (kind == 1 ? e->bp[b].friction : (m->col[i] >= 8 ? e->mu_finger : (real)c->arm_friction));
  r->mu = e->bp[a].friction * mub;
}

static void row_apply(orc_env* e, int kind, int a, int b, const orc_row* r, int k, real dl) {
  v3madd(e->body[a].v, e->body[a].v, r->dir[k], dl * e->bp[a].inv_mass);
  v3madd(e->body[a].w, e->body[a].w, r->aa[k], dl);
  if (kind == 1) {
    v3madd(e->body[b].v, e->body[b].v, r->dir[k], -dl * e->bp[b].inv_mass);
    v3madd(e->body[b].w, e->body[b].w, r->ab[k], -dl);
  }
}
static real row_jv(const orc_env* e, int kind, int a, int b, const orc_row* r, int k) {
  real jv = v3dot(r->dir[k], e->body[a].v) + v3dot(r->rxa[k], e->body[a].w);
  if (kind == 1) jv -= v3dot(r->dir[k], e->body[b].v) + v3dot(r->rxb[k], e->body[b].w);
  else jv -= r->vbc[k];
  return jv;
}
static real point_solve(orc_env* e, int kind, int a, int b, orc_manifold* m, int i, const orc_row* r) {
  /* normal */
  real res;
  real jv = row_jv(e, kind, a, b, r, 0);
  real dl = (r->target - jv) * r->invk[0];
  real ln = rclamp(m->ln[i] + dl, R(0.0), r->cap);
  dl = ln - m->ln[i]; m->ln[i] = ln;
  res = rabs(dl);
  row_apply(e, kind, a, b, r, 0, dl);
  /* friction pyramid */
  real lim = r->mu * m->ln[i];
  jv = row_jv(e, kind, a, b, r, 1);
  dl = -jv * r->invk[1];
  real l1 = rclamp(m->lt1[i] + dl, -lim, lim);
  dl = l1 - m->lt1[i]; m->lt1[i] = l1;
  res = rmax(res, rabs(dl));
  row_apply(e, kind, a, b, r, 1, dl);
  jv = row_jv(e, kind, a, b, r, 2);
  dl = -jv * r->invk[2];
  real l2 = rclamp(m->lt2[i] + dl, -lim, lim);
  dl = l2 - m->lt2[i]; m->lt2[i] = l2;
  res = rmax(res, rabs(dl));
  row_apply(e, kind, a, b, r, 2, dl);
  return res;
}

/* ---- limb dynamics (rv_config.limb_dynamics) --------------------------------------
 * While the arm touches an awake body the seven limb joints are unknowns of the solver (PyBullet: the
 * arm is a btMultiBody whose POSITION_CONTROL motors are constraint rows of the same PGS,
 * controllable_body.py:458-466, bullet_physics.py:1061-1104).  The unknown is the DEVIATION dq of the
 * joint velocities from the ones the motor law of this substep commanded (limb_qd0): the link twists the
 * contact rows were set up with stay as they are, a contact row on a collider of frame f gets the
 * joint-space Jacobian  Ja_j = -dir . (axis_j x (p - p_j)), j <= min(f, 6), an impulse dl on it changes
 * dq by M^-1 Ja^T dl, and its effective mass gains Ja M^-1 Ja^T.  M(q) is the joint-space inertia of the
 * chain of eight masses (links 0..6 and the hand):
 *   M_jk = sum_{i >= max(j,k)} m_i (a_j x (c_i - p_j)) . (a_k x (c_i - p_k)) + (R_i^T a_j) . I_i (R_i^T a_k)
 * (the composite-rigid-body sum written out for revolute joints; velocity-product terms are neglected: the
 * limb moves at < 1 rad/s).  The solve starts from the joint velocities BEFORE the motor step of this
 * substep (dq = -limb_dv: the acceleration limits of the kinematic motor law are no statement about
 * torques).  Motor row j: J = e_j, target dq_j = commanded - current velocity, accumulated impulse within
 * +- tau_j dt minus the torque that holds the chain against gravity.  After the solve the joints move
 * with the solved velocity and the link frames are recomputed. */
typedef struct {
  real M[RV_NLIMB][RV_NLIMB], Mi[RV_NLIMB][RV_NLIMB];
  real lo[RV_NLIMB], hi[RV_NLIMB], tgt[RV_NLIMB];
  real Ja[RV_MAXB][4][3][RV_NLIMB], MiJ[RV_MAXB][4][3][RV_NLIMB], invk[RV_MAXB][4][3];
} orc_limb;
static void limb_prepare(const orc_world* w, orc_env* e, orc_row rows[][4], const int* use, orc_limb* L) {
  const rv_config* c = &w->cfg; const rv_arm* arm = &w->scene.arm;
  const real dt = (real)c->dt;
  real com[RV_NLIMB + 1][3];
  for (int i = 0; i <= RV_NLIMB; ++i) {
    real lc[3] = {(real)arm->link_com[i][0], (real)arm->link_com[i][1], (real)arm->link_com[i][2]}, t[3];
    m3mulv(t, e->frot[i], lc); v3add(com[i], e->fpos[i], t);
  }
  const real g[3] = {(real)c->gravity_xy[0], (real)c->gravity_xy[1], (real)c->gravity_z};
  real A[RV_NLIMB][2 * RV_NLIMB], Gq[RV_NLIMB];
  for (int j = 0; j < RV_NLIMB; ++j) {
    for (int k = 0; k < RV_NLIMB; ++k) {
      const int mx = j > k ? j : k;
      real acc = R(0.0);
      for (int i = mx; i <= RV_NLIMB; ++i) {
        real dj[3], dk[3], lj[3], lk[3], aj[3], ak[3];
        v3sub(dj, com[i], e->fpos[j]); v3cross(lj, e->axis[j], dj);
        v3sub(dk, com[i], e->fpos[k]); v3cross(lk, e->axis[k], dk);
        const real t = (real)arm->link_mass[i] * v3dot(lj, lk);
        m3tmulv(aj, e->frot[i], e->axis[j]); m3tmulv(ak, e->frot[i], e->axis[k]);
        const real r = (aj[0] * ak[0]) * (real)arm->link_inertia[i][0] + (aj[1] * ak[1]) * (real)arm->link_inertia[i][1]
                     + (aj[2] * ak[2]) * (real)arm->link_inertia[i][2];
        acc = acc + (t + r);
      }
      L->M[j][k] = acc; A[j][k] = acc; A[j][RV_NLIMB + k] = j == k ? R(1.0) : R(0.0);
    }
    real gq = R(0.0);
    for (int i = j; i <= RV_NLIMB; ++i) {
      real dj[3], lj[3];
      v3sub(dj, com[i], e->fpos[j]); v3cross(lj, e->axis[j], dj);
      gq = gq + (real)arm->link_mass[i] * v3dot(lj, g);
    }
    Gq[j] = gq;
  }
  /* M^-1 by Gauss-Jordan on [M | 1] without pivoting (M is symmetric positive definite); a column that has
   * been the pivot column is not touched again (the device runs one lane per column) */
  for (int p = 0; p < RV_NLIMB; ++p) {
    const real piv = A[p][p];
    for (int col = 0; col < 2 * RV_NLIMB; ++col) {
      if (col <= p) continue;
      const real ap = A[p][col] / piv;
      for (int r = 0; r < RV_NLIMB; ++r) if (r != p) A[r][col] = A[r][col] - A[r][p] * ap;
      A[p][col] = ap;
    }
  }
  for (int j = 0; j < RV_NLIMB; ++j) for (int k = 0; k < RV_NLIMB; ++k) L->Mi[j][k] = A[j][RV_NLIMB + k];
  for (int j = 0; j < RV_NLIMB; ++j) {
    const real hold = -Gq[j] * dt;
    const real tdt = arm->inv_tau_max[j] > 0.0f ? dt / (real)arm->inv_tau_max[j] : R(1e30);
    L->lo[j] = rmin(R(0.0), -tdt - hold); L->hi[j] = rmax(R(0.0), tdt - hold);
    L->tgt[j] = e->limb_vt[j] - e->limb_qd0[j];
  }
  for (int b = 0; b < RV_MAXB; ++b) {
    const orc_manifold* m = &e->man[AIDX(b)];
    if (!use[AIDX(b)]) continue;
    for (int i = 0; i < m->n; ++i) {
      real wa[3], wb[3];
      manifold_world_points(w, e, 2, b, -1, m, i, wa, wb);
      const int f = arm->col_frame[m->col[i]], fl = f < RV_NLIMB ? f : RV_NLIMB - 1;
      for (int k = 0; k < 3; ++k) {
        const orc_row* r = &rows[AIDX(b)][i];
        real* Ja = L->Ja[b][i][k]; real* MiJ = L->MiJ[b][i][k];
        for (int j = 0; j < RV_NLIMB; ++j) {
          real d[3], lever[3];
          v3sub(d, wb, e->fpos[j]); v3cross(lever, e->axis[j], d);
          Ja[j] = j <= fl ? -v3dot(r->dir[k], lever) : R(0.0);
        }
        real kk = R(1.0) / r->invk[k];
        for (int j = 0; j < RV_NLIMB; ++j) {
          real a_ = R(0.0);
          for (int x = 0; x < RV_NLIMB; ++x) a_ = a_ + L->Mi[j][x] * Ja[x];
          MiJ[j] = a_;
        }
        for (int j = 0; j < RV_NLIMB; ++j) kk = kk + Ja[j] * MiJ[j];
        L->invk[b][i][k] = R(1.0) / kk;
      }
    }
  }
}
/* diagnostics for the tests: joint-space inertia (49), its inverse (49), the motor-row bounds lo/hi (7 + 7)
 * of the CURRENT joint state of env i and the motor-row impulses of the last limb solve (7) */
void orc_limb_debug(orc_world* w, int i, double* M, double* Mi, double* lohi) {
  orc_env* e = &w->env[i];
  orc_limb L; orc_row rows[RV_NMAN][4]; int use[RV_NMAN];
  memset(use, 0, sizeof(use)); memset(rows, 0, sizeof(rows));
  arm_update_kinematics(w, e);
  limb_prepare(w, e, rows, use, &L);
  for (int j = 0; j < RV_NLIMB; ++j) {
    for (int k = 0; k < RV_NLIMB; ++k) { M[j * RV_NLIMB + k] = (double)L.M[j][k]; Mi[j * RV_NLIMB + k] = (double)L.Mi[j][k]; }
    lohi[j] = (double)L.lo[j]; lohi[RV_NLIMB + j] = (double)L.hi[j]; lohi[2 * RV_NLIMB + j] = (double)e->limb_lam[j];
  }
}
static real limb_jv(const real* Ja, const real* dq) {
  real a_ = R(0.0);
  for (int j = 0; j < RV_NLIMB; ++j) a_ = a_ + Ja[j] * dq[j];
  return a_;
}
static void limb_apply(const real* MiJ, real* dq, real dl) { for (int j = 0; j < RV_NLIMB; ++j) dq[j] = dq[j] + MiJ[j] * dl; }

/* PGS with the force-limited gripper (rv_config.finger_dynamics; Grasp4DofEnv).  The two finger
 * joints are dynamic 1-DoF bodies of mass finger_mass sliding along the hand's y axis: contact rows
 * on a finger pad (collider boxes 8 / 9) carry jf = -(dir . y) on the finger velocity, and each
 * finger has a POSITION_CONTROL motor row (bullet_physics.py:1061-1104) that pulls its velocity to
 * the commanded one with at most finger_max_force (the joint motors have already spent m * fing_dv
 * of that budget on the free motion).  Velocity-space Gauss-Seidel over all awake bodies and both
 * fingers as one system. */
/* lj / lm / lk: the limb Jacobians Ja[3][7], M^-1 Ja^T [3][7] and effective masses [3] of an arm row in
 * limb_dynamics mode (NULL otherwise), dq: the deviation of the joint velocities */
static real point_solve_g(orc_env* e, int a, orc_manifold* m, int i, const orc_row* r, real* qf, real imf,
                          const real (*lj)[RV_NLIMB], const real (*lm)[RV_NLIMB], const real* lk, real* dq) {
  const int fi = r->fidx;
  real jv = row_jv(e, 0, a, -1, r, 0);
  if (fi >= 0) jv += r->jf[0] * qf[fi];
  if (lj) jv += limb_jv(lj[0], dq);
  real dl = (r->target - jv) * (lj ? lk[0] : r->invk[0]);
  real ln = rclamp(m->ln[i] + dl, R(0.0), r->cap);
  dl = ln - m->ln[i]; m->ln[i] = ln;
  real res = rabs(dl);
  row_apply(e, 0, a, -1, r, 0, dl);
  if (fi >= 0) qf[fi] += r->jf[0] * dl * imf;
  if (lj) limb_apply(lm[0], dq, dl);
  real lim = r->mu * ln;
  jv = row_jv(e, 0, a, -1, r, 1);
  if (fi >= 0) jv += r->jf[1] * qf[fi];
  if (lj) jv += limb_jv(lj[1], dq);
  dl = -jv * (lj ? lk[1] : r->invk[1]);
  real l1 = rclamp(m->lt1[i] + dl, -lim, lim);
  dl = l1 - m->lt1[i]; m->lt1[i] = l1;
  res = rmax(res, rabs(dl));
  row_apply(e, 0, a, -1, r, 1, dl);
  if (fi >= 0) qf[fi] += r->jf[1] * dl * imf;
  if (lj) limb_apply(lm[1], dq, dl);
  jv = row_jv(e, 0, a, -1, r, 2);
  if (fi >= 0) jv += r->jf[2] * qf[fi];
  if (lj) jv += limb_jv(lj[2], dq);
  dl = -jv * (lj ? lk[2] : r->invk[2]);
  real l2 = rclamp(m->lt2[i] + dl, -lim, lim);
  dl = l2 - m->lt2[i]; m->lt2[i] = l2;
  res = rmax(res, rabs(dl));
  row_apply(e, 0, a, -1, r, 2, dl);
  if (fi >= 0) qf[fi] += r->jf[2] * dl * imf;
  if (lj) limb_apply(lm[2], dq, dl);
  return res;
}
/* The six rows of a user constraint on body b (a fixed joint to a frame of the world: the mocap-style
 * constraint ControllableConstraint servoes, controllable_constraint.py:21-170): three linear rows at
 * the pivot, three angular rows, Baumgarte-stabilised with rv_config.erp, accumulated impulse within
 * +- con_fmax dt per row (pybullet changeConstraint maxForce).  lam[6]: accumulated impulses. */
/* con_on encodes the joint: bits 0-3 the type (1 fixed: six rows, 2 point to point: the three linear rows), bits 4-7
 * the child body + 1 (0: the world).  With a child body the frame (con_tpos, con_tquat) is given in the CHILD's frame
 * and every row acts on both bodies; a child that is absent / frozen stands still like the world. */
#define CON_TYPE(x) ((x) & 15)
#define CON_CHILD(x) ((((x) >> 4) & 15) - 1)
static real constraint_solve(const orc_world* w, orc_env* e, int b, real* lam) {
  const rv_config* c = &w->cfg; const orc_bparam* P = &e->bp[b]; orc_body* B = &e->body[b];
  const real dt = (real)c->dt, lim = P->con_fmax * dt;
  const int ctype = CON_TYPE(P->con_on), cw = CON_CHILD(P->con_on);
  int cb = cw;
  /* child >= RV_MAXB: frame cw - RV_MAXB of the ARM (a link as the other party, bullet_physics.py:773-779): a kinematic frame
   * that moves with the link's twist and takes no impulse */
  const int lf = cw >= RV_MAXB ? cw - RV_MAXB : -1;
  if (lf >= 0) cb = -2;
  if (cb >= 0 && !body_on(e, cb)) cb = -2;
  real rot[9], r[3], wp[3], res = R(0.0);
  qmat(rot, B->q); m3mulv(r, rot, P->con_lpos); v3add(wp, B->p, r);
  real qw[4], qc[4], qe[4];
  qmul(qw, B->q, P->con_lquat);                                  /* the joint frame of the body, in the world */
  qc[0] = -qw[0]; qc[1] = -qw[1]; qc[2] = -qw[2]; qc[3] = qw[3];
  real tp[3] = {P->con_tpos[0], P->con_tpos[1], P->con_tpos[2]}, rc[3] = {R(0.0), R(0.0), R(0.0)};
  real tq[4] = {P->con_tquat[0], P->con_tquat[1], P->con_tquat[2], P->con_tquat[3]};
  real imc = R(0.0);
  real lv[3] = {R(0.0), R(0.0), R(0.0)}, lw[3] = {R(0.0), R(0.0), R(0.0)};      /* velocity of the link frame at the pivot, its spin */
  if (lf >= 0) {
    real rotc[9];
    qmat(rotc, e->fquat[lf]); m3mulv(rc, rotc, P->con_tpos); v3add(tp, e->fpos[lf], rc);
    qmul(tq, e->fquat[lf], P->con_tquat);
    if (e->arm_enabled) {
      real d[3], cr[3]; v3sub(d, wp, e->fpos[lf]); v3cross(cr, e->fw[lf], d); v3add(lv, e->fv[lf], cr); v3cpy(lw, e->fw[lf]);
    }
  } else if (cw >= 0) {
    real rotc[9];
    qmat(rotc, e->body[cw].q); m3mulv(rc, rotc, P->con_tpos); v3add(tp, e->body[cw].p, rc);
    qmul(tq, e->body[cw].q, P->con_tquat);
    if (cb >= 0) imc = e->bp[cb].inv_mass;
  }
  qmul(qe, tq, qc);                                              /* rotation that takes it to the target frame */
  const real sg = qe[3] < R(0.0) ? R(-2.0) : R(2.0);
  const real th[3] = {qe[0] * sg, qe[1] * sg, qe[2] * sg};
  /* prismatic (type 3; pybullet JOINT_PRISMATIC): the body slides along the x axis of the frame it is tied to -- two
   * linear rows along that frame's y and z axes, then the three angular rows */
  /* revolute (type 4; the reference's JOINT_TYPES_MAPPING 'revolute', bullet_physics.py:20-25): a hinge about the x axis of the
   * frame -- the three linear rows at the pivot and two angular rows along that frame's y and z axes */
  const int n_lin = ctype == 3 ? 2 : 3, n_ang = ctype == 4 ? 2 : 3, n_rows = ctype == 2 ? 3 : n_lin + n_ang;
  real rt[9], dtp[3];
  qmat(rt, tq); v3sub(dtp, tp, wp);
  for (int k = 0; k < n_rows; ++k) {
    real jl[3] = {R(0.0), R(0.0), R(0.0)}, ja[3] = {R(0.0), R(0.0), R(0.0)}, jc[3] = {R(0.0), R(0.0), R(0.0)}, ia[3], ic[3] = {R(0.0), R(0.0), R(0.0)}, bias;
    if (k < n_lin && ctype == 3) {
      real ek[3] = {R(0.0), R(0.0), R(0.0)}; ek[k + 1] = R(1.0);
      m3mulv(jl, rt, ek);                                        /* column k + 1 of the frame's rotation */
      /* ONE anchor for both parties: the parent's pivot wp (as Bullet's slider does).  Along the slide axis wp and the
       * child's frame origin are far apart; equal and opposite impulses at two different points would be a spurious
       * torque (round-4 advisor): the child's lever is wp - x_child, not its own frame origin */
      real rcw[3] = {R(0.0), R(0.0), R(0.0)};
      if (cw >= 0 && lf < 0) v3sub(rcw, wp, e->body[cw].p);
      v3cross(ja, r, jl); v3cross(jc, rcw, jl);
      bias = (real)c->erp * v3dot(jl, dtp) / dt;
    } else if (k < n_lin) {
      jl[k] = R(1.0);
      real ek[3] = {R(0.0), R(0.0), R(0.0)}; ek[k] = R(1.0);
      v3cross(ja, r, ek); v3cross(jc, rc, ek);
      bias = (real)c->erp * (tp[k] - wp[k]) / dt;
    } else if (ctype == 4) {
      real ek[3] = {R(0.0), R(0.0), R(0.0)}; ek[k - n_lin + 1] = R(1.0);
      m3mulv(ja, rt, ek);                                        /* column k - n_lin + 1 of the frame's rotation */
      v3cpy(jc, ja);
      bias = (real)c->erp * v3dot(ja, th) / dt;
    } else {
      ja[k - n_lin] = R(1.0); jc[k - n_lin] = R(1.0);
      bias = (real)c->erp * th[k - n_lin] / dt;
    }
    m3mulv(ia, e->iinv[b], ja);
    real kk = (k < n_lin ? P->inv_mass : R(0.0)) + v3dot(ja, ia);
    real jv = v3dot(jl, B->v) + v3dot(ja, B->w);
    if (lf >= 0) jv = jv - (k < n_lin ? v3dot(jl, lv) : v3dot(ja, lw));
    if (cb >= 0) {
      m3mulv(ic, e->iinv[cb], jc);
      kk = kk + ((k < n_lin ? imc : R(0.0)) + v3dot(jc, ic));
      jv = jv - (v3dot(jl, e->body[cb].v) + v3dot(jc, e->body[cb].w));
    }
    // a row no party of which can move (a static parent -- mass 0 -- constrained to the world, a link or another static
    // body): nothing to solve, as in Bullet, where a constraint on a fixed-base body is harmless (0 / 0 here otherwise)
    if (!(kk > R(0.0))) continue;
    real dl = (bias - jv) / kk;
    const real ln = rclamp(lam[k] + dl, -lim, lim);
    dl = ln - lam[k]; lam[k] = ln;
    res = rmax(res, rabs(dl));
    v3madd(B->v, B->v, jl, dl * P->inv_mass);
    v3madd(B->w, B->w, ia, dl);
    if (cb >= 0) {
      v3madd(e->body[cb].v, e->body[cb].v, jl, -(dl * imc));
      v3madd(e->body[cb].w, e->body[cb].w, ic, -dl);
    }
  }
  return res;
}
static int con_pair_member(const orc_env* e, int b) {
  int m = CON_TYPE(e->bp[b].con_on) != 0 && CON_CHILD(e->bp[b].con_on) >= 0;
  for (int x = 0; x < RV_MAXB; ++x) if (CON_TYPE(e->bp[x].con_on) != 0 && CON_CHILD(e->bp[x].con_on) == b) m = 1;
  return m;
}
static void solve_with_fingers(const orc_world* w, orc_env* e, orc_row rows[][4], const int* use, const orc_limb* L) {
  const rv_config* c = &w->cfg; const rv_arm* arm = &w->scene.arm;
  real dq[RV_NLIMB], lam_l[RV_NLIMB];
  for (int j = 0; j < RV_NLIMB; ++j) { dq[j] = L ? -e->limb_dv[j] : R(0.0); lam_l[j] = R(0.0); }   /* (the solve starts from the velocity before the motor step) */
  const int fd = c->finger_dynamics && e->arm_enabled;           /* the finger joints are DOFs of the system */
  const real mf = (real)c->finger_mass, imf = R(1.0) / (real)c->finger_mass, fdt = (real)c->finger_max_force * (real)c->dt;
  real qf[2] = {e->qd[RV_NLIMB], e->qd[RV_NLIMB + 1]}, lam_m[2] = {R(0.0), R(0.0)};
  real lam_c[RV_MAXB][6]; memset(lam_c, 0, sizeof(lam_c));
  real best = R(1e30); int since = 0;               /* rv_config.solver_stall */
  for (int it = -1; it < c->solver_iters; ++it) {
    real res = R(0.0);
    for (int b = 0; b < RV_MAXB; ++b) {
      if (!use[TIDX(b)]) continue;
      for (int kind = 0; kind <= 2; kind += 2) {
        const int mi = kind == 0 ? TIDX(b) : AIDX(b);
        orc_manifold* m = &e->man[mi];
        for (int i = 0; i < m->n; ++i) {
          const orc_row* r = &rows[mi][i];
          const int la = L && kind == 2;
          if (it < 0) {
            row_apply(e, 0, b, -1, r, 0, m->ln[i]); row_apply(e, 0, b, -1, r, 1, m->lt1[i]); row_apply(e, 0, b, -1, r, 2, m->lt2[i]);
            if (r->fidx >= 0) { qf[r->fidx] += r->jf[0] * m->ln[i] * imf; qf[r->fidx] += r->jf[1] * m->lt1[i] * imf; qf[r->fidx] += r->jf[2] * m->lt2[i] * imf; }
            if (la) { limb_apply(L->MiJ[b][i][0], dq, m->ln[i]); limb_apply(L->MiJ[b][i][1], dq, m->lt1[i]); limb_apply(L->MiJ[b][i][2], dq, m->lt2[i]); }
          } else res = rmax(res, point_solve_g(e, b, m, i, r, qf, imf, la ? L->Ja[b][i] : NULL, la ? L->MiJ[b][i] : NULL, la ? L->invk[b][i] : NULL, dq));
        }
      }
    }
    for (int rd = 0; rd < 3; ++rd)
      for (int x = 0; x < 2; ++x) {
        const int k = BB_ROUND[rd][x], mi = BBIDX(k);
        orc_manifold* m = &e->man[mi];
        if (!use[mi]) continue;
        for (int i = 0; i < m->n; ++i) {
          if (it < 0) {
            row_apply(e, 1, BB_A[k], BB_B[k], &rows[mi][i], 0, m->ln[i]); row_apply(e, 1, BB_A[k], BB_B[k], &rows[mi][i], 1, m->lt1[i]);
            row_apply(e, 1, BB_A[k], BB_B[k], &rows[mi][i], 2, m->lt2[i]);
          } else res = rmax(res, point_solve(e, 1, BB_A[k], BB_B[k], m, i, &rows[mi][i]));
        }
      }
    if (it < 0) continue;
    for (int b = 0; b < RV_MAXB; ++b) if (use[TIDX(b)] && e->bp[b].con_on) res = rmax(res, constraint_solve(w, e, b, lam_c[b]));
    for (int f = 0; fd && f < 2; ++f) {
      const real i0 = mf * e->fing_dv[f];
      real dl = (e->fing_vt[f] - qf[f]) * mf;
      const real ln = rclamp(lam_m[f] + dl, -fdt - i0, fdt - i0);
      dl = ln - lam_m[f]; lam_m[f] = ln;
      qf[f] += dl * imf;
      res = rmax(res, rabs(dl));
    }
    for (int j = 0; L && j < RV_NLIMB; ++j) {     /* limb motor rows */
      real dl = (L->tgt[j] - dq[j]) / L->Mi[j][j];
      const real ln = rclamp(lam_l[j] + dl, L->lo[j], L->hi[j]);
      dl = ln - lam_l[j]; lam_l[j] = ln;
      for (int k = 0; k < RV_NLIMB; ++k) dq[k] = dq[k] + L->Mi[k][j] * dl;
      res = rmax(res, rabs(dl));
    }
    if (res < (real)c->solver_tol) break;
    if (c->solver_stall > 0) { if (res < best) { best = res; since = 0; } else if (++since >= c->solver_stall) break; }
  }
  for (int j = 0; L && j < RV_NLIMB; ++j) {       /* the limb moves with the solved velocity */
    real qd = e->limb_qd0[j] + dq[j];
    real qn = e->q[j] + dq[j] * (real)c->dt;
    if (qn < (real)arm->q_lo[j]) { qn = (real)arm->q_lo[j]; qd = R(0.0); }
    if (qn > (real)arm->q_hi[j]) { qn = (real)arm->q_hi[j]; qd = R(0.0); }
    e->q[j] = qn; e->qd[j] = qd;
    e->limb_lam[j] = lam_l[j];
  }
  for (int f = 0; fd && f < 2; ++f) {
    const int j = RV_NLIMB + f;
    real qd = qf[f];
    real qn = e->q[j] + (qd - e->fing_qd0[f]) * (real)c->dt;
    if (qn < (real)arm->q_lo[j]) { qn = (real)arm->q_lo[j]; qd = R(0.0); }
    if (qn > (real)arm->q_hi[j]) { qn = (real)arm->q_hi[j]; qd = R(0.0); }
    e->q[j] = qn; e->qd[j] = qd;
  }
}

/* PGS in impulse space.  The rows of ALL islands of the env in the order the sequential
 * Gauss-Seidel visits them: body by body (ascending) the points of its table manifold, then of
 * its arm manifold; then the body-body manifolds in colour-round order; per point the normal
 * row, then the two friction rows.  Row r acts on body a_r with J = (+dir, +rxa) and, for a
 * body-body row, on b_r with (-dir, -rxb); an impulse dl on row s changes the velocity of a_s
 * by P0_s dl = (dir_s / m_a, aa_s) dl and of b_s by P1_s dl = -(dir_s / m_b, ab_s) dl.  Row
 * velocities g_r = J_r u - vbc_r evolve as g += A[:, s] dl, A_rs = J_r[a_s].P0_s + J_r[b_s].P1_s.
 * Same order, clamps and per-island residual test as the velocity-space solver below (which
 * remains the fallback for more than SOLVE_ROWS rows); velocities are rebuilt at the end. */
/* The residual an island's sweeps stop on.  An island all of whose bodies were below the sleep thresholds after the
 * last substep converges to rv_config.solver_tol_rest (< solver_tol): with the plain 1e-5 N s exit resting bodies
 * creep at ~4e-5 m/s -- a solver artefact that only deactivation hides (Bullet runs its 50 sweeps without an early
 * exit).  Islands that hold motor rows (fingers, limb) keep solver_tol. */
static real island_tol(const rv_config* c, const orc_env* e, const int* use, const int* label, int root) {
  if (!(c->solver_tol_rest > 0.0f && c->solver_tol_rest < c->solver_tol)) return (real)c->solver_tol;
  for (int b = 0; b < RV_MAXB; ++b) if (use[TIDX(b)] && label[b] == root && !(e->bp[b].sleep_count > 0)) return (real)c->solver_tol;
  return (real)c->solver_tol_rest;
}
#define SOLVE_ROWS 120
typedef struct { int mi, i, k, a, b, isl; } orc_rowid;
typedef struct { real l[3], a[3]; } orc_j6;
static real dotj(const orc_j6* j, const real* pl, const real* pa) { return v3dot(j->l, pl) + v3dot(j->a, pa); }
/* fing != 0 (rv_config.finger_dynamics, at most one awake body): the two finger joints are DOFs of the
 * system as well.  A contact row on a finger pad has the Jacobian entry jf on its finger's velocity (an
 * impulse dl changes that velocity by jf dl / finger_mass), and each finger has a POSITION_CONTROL motor
 * row (bullet_physics.py:1061-1104) after the contact rows: J = 1 on the finger, target = the commanded
 * velocity, impulse within +-finger_max_force dt minus what the joint motors of the light part already
 * spent on the free motion.  The Delassus matrix gets the finger terms, nothing else changes; the fingers
 * then move with the solved velocity. */
static void solve_rows(const orc_world* w, orc_env* e, orc_row rows[][4], const orc_rowid* id, int n_rows, const int* use, const int* big, const int* label, const int fing, const orc_limb* L, const int motor_isl) {
  const rv_config* c = &w->cfg;
  const rv_arm* arm = &w->scene.arm;
  static __thread real A[SOLVE_ROWS + 9][SOLVE_ROWS + 9];
  real g[SOLVE_ROWS + 9], lam[SOLVE_ROWS + 9], invk[SOLVE_ROWS + 9], bias[SOLVE_ROWS + 9], mu[SOLVE_ROWS + 9], cap[SOLVE_ROWS + 9];
  real jf[SOLVE_ROWS + 9], pf[SOLVE_ROWS + 9], mlo[2] = {R(0.0), R(0.0)}, mhi[2] = {R(0.0), R(0.0)}; int fi[SOLVE_ROWS + 9];
  const real mf = (real)c->finger_mass, imf = fing ? R(1.0) / (real)c->finger_mass : R(0.0), fdt = (real)c->finger_max_force * (real)c->dt;
  const real qf0[2] = {e->qd[RV_NLIMB], e->qd[RV_NLIMB + 1]};
  /* L != NULL (rv_config.limb_dynamics, at most one awake body): the seven limb joints are DOFs as well.  Rows
   * of the arm manifold and the seven limb motor rows (after the finger motor rows) are 'limb rows': row r has
   * the joint-space Jacobian ja[r] (a motor row: e_j) and the velocity change per unit impulse pj[r] = M^-1 ja^T
   * (a motor row: column j of M^-1); A_rs gains ja[r] . pj[s] */
  const int nfm = fing ? 2 : 0, nlm = L ? RV_NLIMB : 0;
  const int n_all = n_rows + nfm + nlm;
  int la[SOLVE_ROWS + 9]; real ja[SOLVE_ROWS + 9][RV_NLIMB], pj[SOLVE_ROWS + 9][RV_NLIMB], dq0[RV_NLIMB];
  for (int x = 0; x < RV_NLIMB; ++x) dq0[x] = L ? -e->limb_dv[x] : R(0.0);     /* the solve starts from the velocity before the motor step */
  int fisl = 0;      /* the island the motor rows belong to: the awake body's when there is at most one, else motor_isl */
  orc_j6 jx[SOLVE_ROWS][RV_MAXB];
  for (int r = 0; r < n_rows; ++r) {
    const orc_row* rw = &rows[id[r].mi][id[r].i]; const orc_manifold* mm = &e->man[id[r].mi];
    const int k = id[r].k, ra = id[r].a, rb = id[r].b, pi = id[r].i;
    memset(jx[r], 0, sizeof(jx[r]));
    invk[r] = rw->invk[k]; mu[r] = rw->mu; bias[r] = k == 0 ? rw->target : R(0.0); cap[r] = rw->cap;
    lam[r] = k == 0 ? mm->ln[pi] : (k == 1 ? mm->lt1[pi] : mm->lt2[pi]);
    real gg = v3dot(rw->dir[k], e->body[ra].v) + v3dot(rw->rxa[k], e->body[ra].w);
    v3cpy(jx[r][ra].l, rw->dir[k]); v3cpy(jx[r][ra].a, rw->rxa[k]);
    if (rb >= 0) {
      gg -= v3dot(rw->dir[k], e->body[rb].v) + v3dot(rw->rxb[k], e->body[rb].w);
      for (int x = 0; x < 3; ++x) { jx[r][rb].l[x] = -rw->dir[k][x]; jx[r][rb].a[x] = -rw->rxb[k][x]; }
    } else gg -= rw->vbc[k];
    jf[r] = R(0.0); pf[r] = R(0.0); fi[r] = -1;
    if (fing) {
      fi[r] = rw->fidx; fisl = id[r].isl;
      if (fi[r] >= 0) { jf[r] = rw->jf[k]; pf[r] = jf[r] * imf; gg += jf[r] * qf0[fi[r]]; }
    }
    la[r] = 0;
    if (L) {
      fisl = id[r].isl;
      if (id[r].mi >= AIDX(0)) {
        const int b_ = id[r].mi - AIDX(0);
        la[r] = 1; invk[r] = L->invk[b_][pi][k];
        real t = R(0.0);
        for (int x = 0; x < RV_NLIMB; ++x) { ja[r][x] = L->Ja[b_][pi][k][x]; pj[r][x] = L->MiJ[b_][pi][k][x]; t = t + ja[r][x] * dq0[x]; }
        gg += t;
      }
    }
    g[r] = gg;
  }
  if (motor_isl >= 0) fisl = motor_isl;
  for (int m = 0; fing && m < 2; ++m) {       /* motor rows */
    const int r = n_rows + m;
    const real i0 = mf * e->fing_dv[m];
    g[r] = qf0[m] - e->fing_vt[m]; lam[r] = R(0.0); invk[r] = mf; bias[r] = R(0.0); mu[r] = R(0.0); cap[r] = R(0.0);
    jf[r] = R(1.0); pf[r] = imf; fi[r] = m;
    mlo[m] = -fdt - i0; mhi[m] = fdt - i0;
    la[r] = 0;
  }
  for (int j = 0; j < nlm; ++j) {             /* limb motor rows */
    const int r = n_rows + nfm + j;
    g[r] = dq0[j] - L->tgt[j]; lam[r] = R(0.0); invk[r] = R(1.0) / L->Mi[j][j]; bias[r] = R(0.0); mu[r] = R(0.0); cap[r] = R(0.0);
    jf[r] = R(0.0); pf[r] = R(0.0); fi[r] = -1; la[r] = 1;
    for (int x = 0; x < RV_NLIMB; ++x) { ja[r][x] = x == j ? R(1.0) : R(0.0); pj[r][x] = L->Mi[x][j]; }
  }
  for (int r = 0; r < n_rows; ++r)
    for (int s2 = 0; s2 < n_rows; ++s2) {
      const orc_row* q = &rows[id[s2].mi][id[s2].i];
      const int ks = id[s2].k, as = id[s2].a, bs = id[s2].b;
      real pl[3] = {q->dir[ks][0] * e->bp[as].inv_mass, q->dir[ks][1] * e->bp[as].inv_mass, q->dir[ks][2] * e->bp[as].inv_mass};
      real a_ = dotj(&jx[r][as], pl, q->aa[ks]);
      if (bs >= 0) {
        real t[3] = {q->dir[ks][0] * e->bp[bs].inv_mass, q->dir[ks][1] * e->bp[bs].inv_mass, q->dir[ks][2] * e->bp[bs].inv_mass};
        real nl_[3] = {-t[0], -t[1], -t[2]}, na_[3] = {-q->ab[ks][0], -q->ab[ks][1], -q->ab[ks][2]};
        a_ = a_ + dotj(&jx[r][bs], nl_, na_);
      }
      if (fing && fi[r] >= 0 && fi[r] == fi[s2]) a_ = a_ + jf[r] * pf[s2];
      A[r][s2] = a_;
    }
  for (int m = 0; fing && m < 2; ++m) {
    const int q = n_rows + m;
    for (int r = 0; r < n_rows; ++r) { A[r][q] = fi[r] == m ? jf[r] * pf[q] : R(0.0); A[q][r] = fi[r] == m ? pf[r] : R(0.0); }
    for (int m2 = 0; m2 < 2; ++m2) A[q][n_rows + m2] = m == m2 ? pf[q] : R(0.0);
  }
  for (int j = 0; j < nlm; ++j) {
    const int q = n_rows + nfm + j;
    for (int r = 0; r < n_all; ++r) { A[r][q] = R(0.0); A[q][r] = R(0.0); }
  }
  for (int r = 0; L && r < n_all; ++r)
    for (int s2 = 0; s2 < n_all; ++s2) {
      if (!(la[r] && la[s2])) continue;
      real t = R(0.0);
      for (int x = 0; x < RV_NLIMB; ++x) t = t + ja[r][x] * pj[s2][x];
      A[r][s2] = A[r][s2] + t;
    }
  for (int s2 = 0; s2 < n_rows; ++s2) for (int r = 0; r < n_all; ++r) g[r] = g[r] + A[r][s2] * lam[s2];
  /* NORMALISED RESIDUAL FORM of the row step (round 5).  rr[r] = (bias[r] - g[r]) invk[r] is the change row r's impulse
   * would take if it were unbounded; a row step is  nl = clamp(lam[s] + rr[s]); d = nl - lam[s]; rr[r] += C[r][s] d  with
   * C[r][s] = -(A[r][s] invk[r]).  In exact arithmetic these are the iterates of  nl = clamp(lam + (bias - g) invk),
   * g += A d;  on the device a row step is v_add, v_med3, v_sub, the broadcast and ONE v_fma instead of thirteen
   * instructions with three selects (tools/ubench/sweep_step.hip: 92 -> 45 clocks per row step).  The update is a fused
   * multiply-add (fmaf / v_fma_f32: a single rounding, the same on host and device).  (Tracking T = lam + rr instead
   * would save the add, but T rounds at the scale of lam where rr rounds at the scale of the residual: run to
   * convergence in FP32 it left 5e-6 m/s where this form leaves 3e-8 -- tests/test_independent_pin.py.) */
  real rr[SOLVE_ROWS + 9];
  for (int r = 0; r < n_all; ++r) {
    rr[r] = (bias[r] - g[r]) * invk[r];
    for (int s2 = 0; s2 < n_all; ++s2) A[r][s2] = -(A[r][s2] * invk[r]);
  }
  int isl_rows = 0, done = 0;
  real best[RV_MAXB] = {R(1e30), R(1e30), R(1e30), R(1e30)}; int since[RV_MAXB] = {0, 0, 0, 0};
  for (int s2 = 0; s2 < n_rows; ++s2) isl_rows |= 1 << id[s2].isl;
  if (fing || L) isl_rows |= 1 << fisl;
  real tolx[RV_MAXB];
  for (int x = 0; x < RV_MAXB; ++x) tolx[x] = ((fing || L) && x == fisl) ? (real)c->solver_tol : island_tol(c, e, use, label, x);
  for (int it = 0; it < c->solver_iters; ++it) {
    real res[RV_MAXB] = {R(0.0), R(0.0), R(0.0), R(0.0)};
    for (int x = 0; x < RV_MAXB; ++x) if (((isl_rows >> x) & 1) && !((done >> x) & 1)) { e->cnt_sweeps++; if (it == 0) e->cnt_islands++; }
    real limtab[RV_NMAN][4];   /* friction bound of every point: mu x its normal impulse (the rows of a point need not be neighbours in the list) */
    for (int s2 = 0; s2 < n_rows; ++s2) {
      const int isl = id[s2].isl;
      if ((done >> isl) & 1) continue;
      real nl;
      const real lim = id[s2].k == 0 ? R(0.0) : limtab[id[s2].mi][id[s2].i];
      if (id[s2].k == 0) nl = rclamp(lam[s2] + rr[s2], R(0.0), cap[s2]);
      else nl = rclamp(lam[s2] + rr[s2], -lim, lim);
      const real d = nl - lam[s2];
      lam[s2] = nl; e->cnt_rowsteps++;
      if (id[s2].k == 0) limtab[id[s2].mi][id[s2].i] = mu[s2] * nl;
      res[isl] = rmax(res[isl], rabs(d));
      for (int r = 0; r < n_all; ++r) rr[r] = rfma(A[r][s2], d, rr[r]);
    }
    /* (the motor rows belong to island fisl and stop with it -- other islands may still be sweeping) */
    const int motors_on = !((done >> fisl) & 1);
    for (int m = 0; fing && motors_on && m < 2; ++m) {
      const int q = n_rows + m;
      const real nl = rclamp(lam[q] + rr[q], mlo[m], mhi[m]);
      const real d = nl - lam[q];
      lam[q] = nl;
      res[fisl] = rmax(res[fisl], rabs(d));
      for (int r = 0; r < n_all; ++r) rr[r] = rfma(A[r][q], d, rr[r]);
    }
    for (int j = 0; motors_on && j < nlm; ++j) {
      const int q = n_rows + nfm + j;
      const real nl = rclamp(lam[q] + rr[q], L->lo[j], L->hi[j]);
      const real d = nl - lam[q];
      lam[q] = nl;
      res[fisl] = rmax(res[fisl], rabs(d));
      for (int r = 0; r < n_all; ++r) rr[r] = rfma(A[r][q], d, rr[r]);
    }
    for (int x = 0; x < RV_MAXB; ++x) if (((isl_rows >> x) & 1) && res[x] < tolx[x]) done |= 1 << x;
    /* stalled islands (rv_config.solver_stall): no new smallest residual for that many sweeps */
    for (int x = 0; c->solver_stall > 0 && x < RV_MAXB; ++x) {
      if (!((isl_rows >> x) & 1) || ((done >> x) & 1)) continue;
      if (res[x] < best[x]) { best[x] = res[x]; since[x] = 0; }
      else if (++since[x] >= c->solver_stall) done |= 1 << x;
    }
#ifdef ORC_DEBUG_SOLVE
    fprintf(stderr, "it %d res %g %g %g %g | lam", it, (double)res[0], (double)res[1], (double)res[2], (double)res[3]);
    for (int s2 = 0; s2 < n_rows; ++s2) fprintf(stderr, " %.4g", (double)lam[s2]);
    fprintf(stderr, "\n");
#endif
    if ((done & isl_rows) == isl_rows) break;
  }
  for (int s2 = 0; s2 < n_rows; ++s2) {
    orc_manifold* mm = &e->man[id[s2].mi];
    if (id[s2].k == 0) mm->ln[id[s2].i] = lam[s2]; else if (id[s2].k == 1) mm->lt1[id[s2].i] = lam[s2]; else mm->lt2[id[s2].i] = lam[s2];
  }
  for (int X = 0; X < RV_MAXB; ++X) {
    if (!use[TIDX(X)] || big[label[X]]) continue;
    for (int cc = 0; cc < 6; ++cc) {
      real acc = cc < 3 ? e->body[X].v[cc] : e->body[X].w[cc - 3];
      for (int s2 = 0; s2 < n_rows; ++s2) {
        const int as = id[s2].a, bs = id[s2].b, ks = id[s2].k;
        if (as != X && bs != X) continue;
        const orc_row* q = &rows[id[s2].mi][id[s2].i];
        real coef;
        if (as == X) coef = cc < 3 ? q->dir[ks][cc] * e->bp[X].inv_mass : q->aa[ks][cc - 3];
        else coef = cc < 3 ? -(q->dir[ks][cc] * e->bp[X].inv_mass) : -q->ab[ks][cc - 3];
        acc = acc + coef * lam[s2];
      }
      if (cc < 3) e->body[X].v[cc] = acc; else e->body[X].w[cc - 3] = acc;
    }
  }
  for (int m = 0; fing && m < 2; ++m) {        /* the fingers move with the solved velocity */
    real qd = qf0[m];
    for (int s2 = 0; s2 < n_rows; ++s2) if (fi[s2] == m) qd = qd + pf[s2] * lam[s2];
    qd = qd + pf[n_rows + m] * lam[n_rows + m];
    const int j = RV_NLIMB + m;
    real qn = e->q[j] + (qd - e->fing_qd0[m]) * (real)c->dt;
    if (qn < (real)arm->q_lo[j]) { qn = (real)arm->q_lo[j]; qd = R(0.0); }
    if (qn > (real)arm->q_hi[j]) { qn = (real)arm->q_hi[j]; qd = R(0.0); }
    e->q[j] = qn; e->qd[j] = qd;
  }
  for (int x = 0; x < nlm; ++x) {              /* the limb moves with the solved velocity */
    real dq = dq0[x];
    for (int s2 = 0; s2 < n_all; ++s2) if (la[s2]) dq = dq + pj[s2][x] * lam[s2];
    real qd = e->limb_qd0[x] + dq;
    real qn = e->q[x] + dq * (real)c->dt;
    if (qn < (real)arm->q_lo[x]) { qn = (real)arm->q_lo[x]; qd = R(0.0); }
    if (qn > (real)arm->q_hi[x]) { qn = (real)arm->q_hi[x]; qd = R(0.0); }
    e->q[x] = qn; e->qd[x] = qd;
    e->limb_lam[x] = lam[n_rows + nfm + x];
  }
}

static void solve_contacts(const orc_world* w, orc_env* e) {
  const rv_config* c = &w->cfg;
  orc_row rows[RV_NMAN][4];
  /* manifolds that take part: both bodies awake (a sleeping body has none) */
  int use[RV_NMAN];
  for (int b = 0; b < RV_MAXB; ++b) { use[TIDX(b)] = body_on(e, b); use[AIDX(b)] = body_on(e, b); }
  for (int k = 0; k < RV_NBB; ++k) use[BBIDX(k)] = body_on(e, BB_A[k]) && body_on(e, BB_B[k]);
  /* islands: awake bodies coupled (transitively) by body-body manifolds that hold
   * points.  An island is an independent problem and stops on its own residual. */
  int label[RV_MAXB];
  for (int b = 0; b < RV_MAXB; ++b) label[b] = b;
  for (int pass = 0; pass < RV_MAXB; ++pass)
    for (int k = 0; k < RV_NBB; ++k) {
      if (!use[BBIDX(k)] || e->man[BBIDX(k)].n == 0) continue;
      int la = label[BB_A[k]], lb = label[BB_B[k]];
      int lo = la < lb ? la : lb;
      label[BB_A[k]] = lo; label[BB_B[k]] = lo;
    }
  /* islands of one or two bodies are solved in impulse space (solve_rows); bigger ones by the
   * velocity-space Gauss-Seidel below */
  int big[RV_MAXB];
  for (int b = 0; b < RV_MAXB; ++b) {
    int members = 0;
    for (int x = 0; x < RV_MAXB; ++x) members += (use[TIDX(x)] && label[x] == b);
    big[b] = use[TIDX(b)] && label[b] == b && members > 2;
  }
  /* row setup (+ warm-start scaling of the kept impulses) and the row list in visiting order */
  orc_rowid id[SOLVE_ROWS]; int n_rows = 0;
  for (int b = 0; b < RV_MAXB; ++b)
    for (int kind = 0; kind <= 2; kind += 2) {
      int mi = kind == 0 ? TIDX(b) : AIDX(b);
      orc_manifold* m = &e->man[mi];
      if (!use[mi]) continue;
      for (int i = 0; i < m->n; ++i) {
        row_setup(w, e, kind, b, -1, m, i, &rows[mi][i]);
        m->ln[i] *= (real)c->warmstart; m->lt1[i] *= (real)c->warmstart; m->lt2[i] *= (real)c->warmstart;
      }
    }
  /* visiting order of the rows of a body's own manifolds: bodies ascending, per body the table points
   * then the arm points, per point n, t1, t2 -- and the two members X < Y of a two-body island are
   * visited TOGETHER, slot by slot (X's row of slot t, then Y's row of slot t; slot = 3 * (4 * [arm] +
   * point) + row): their own rows do not couple with each other (only through the X-Y manifold, which
   * comes later), so the device solves the two blocks side by side */
  for (int b = 0; b < RV_MAXB; ++b) {
    if (!use[TIDX(b)] || big[label[b]]) continue;
    int partner = -1;
    for (int x = 0; x < RV_MAXB; ++x) if (x != b && use[TIDX(x)] && label[x] == label[b]) partner = x;
    if (partner >= 0 && partner < b) continue;       /* visited with the first member */
    for (int t = 0; t < 24; ++t)
      for (int side = 0; side < 2; ++side) {
        const int body = side == 0 ? b : partner;
        if (body < 0) continue;
        const int p = t / 3, k3 = t % 3, mi = p < 4 ? TIDX(body) : AIDX(body), i = p & 3;
        if (!use[mi] || i >= e->man[mi].n) continue;
        orc_rowid x = {mi, i, k3, body, -1, label[body]}; id[n_rows++] = x;
      }
  }
  for (int rd = 0; rd < 3; ++rd)
    for (int x = 0; x < 2; ++x) {
      int k = BB_ROUND[rd][x]; int mi = BBIDX(k);
      orc_manifold* m = &e->man[mi];
      if (!use[mi]) continue;
      for (int i = 0; i < m->n; ++i) {
        row_setup(w, e, 1, BB_A[k], BB_B[k], m, i, &rows[mi][i]);
        m->ln[i] *= (real)c->warmstart; m->lt1[i] *= (real)c->warmstart; m->lt2[i] *= (real)c->warmstart;
        if (!big[label[BB_A[k]]]) for (int k3 = 0; k3 < 3; ++k3) { orc_rowid y = {mi, i, k3, BB_A[k], BB_B[k], label[BB_A[k]]}; id[n_rows++] = y; }
      }
    }
  {
    /* an awake body with a user constraint: everything goes through the velocity-space system solver */
    int any_con = 0, limb = 0;
    for (int b = 0; b < RV_MAXB; ++b) any_con |= use[TIDX(b)] && e->bp[b].con_on;
    /* limb dynamics: an awake body touches the arm -> the joint velocities are unknowns too */
    if (c->limb_dynamics && e->arm_enabled) for (int b = 0; b < RV_MAXB; ++b) limb |= use[AIDX(b)] && e->man[AIDX(b)].n > 0;
    if (limb) {
      /* one awake body and no user constraint: impulse space with the limb (and finger) DOFs and motor rows;
       * else the velocity-space system solver */
      orc_limb L; limb_prepare(w, e, rows, use, &L);
      int n_on = 0;
      for (int b = 0; b < RV_MAXB; ++b) n_on += use[TIDX(b)];
      /* ... or several awake bodies of which ONE touches the arm and is an island by itself (the pushed body while
       * another one is still sliding on): the limb rows live in that island, the other islands of one or two bodies
       * are the usual independent problems of the same impulse-space solve */
      int n_arm = 0, lb = -1, any_big = 0, lone = 0;
      for (int b = 0; b < RV_MAXB; ++b) { if (use[AIDX(b)] && e->man[AIDX(b)].n > 0) { ++n_arm; lb = b; } any_big |= big[b]; }
      if (n_arm == 1) { int members = 0; for (int x = 0; x < RV_MAXB; ++x) members += (use[TIDX(x)] && label[x] == label[lb]); lone = members == 1; }
      const int fingers = c->finger_dynamics && e->arm_enabled;
      if (!any_con && !getenv("ORC_LIMB_SYS") && (n_on <= 1 || (!fingers && lone && !any_big)))
        solve_rows(w, e, rows, id, n_rows, use, big, label, fingers, &L, n_on <= 1 ? -1 : label[lb]);
      else solve_with_fingers(w, e, rows, use, &L);
      arm_update_kinematics(w, e);      /* the link frames follow the solved joint state */
      return;
    }
    if (any_con) { solve_with_fingers(w, e, rows, use, NULL); return; }
  }
  if (c->finger_dynamics && e->arm_enabled) {
    /* at most one awake body (a grasp scene): impulse space, fingers included; else the
     * velocity-space system solver */
    int n_on = 0;
    for (int b = 0; b < RV_MAXB; ++b) n_on += use[TIDX(b)];
    if (n_on <= 1) solve_rows(w, e, rows, id, n_rows, use, big, label, 1, NULL, -1);
    else solve_with_fingers(w, e, rows, use, NULL);
    return;
  }
  if (n_rows > 0) solve_rows(w, e, rows, id, n_rows, use, big, label, 0, NULL, -1);
  /* big islands: warm start first */
  for (int b = 0; b < RV_MAXB; ++b)
    for (int kind = 0; kind <= 2; kind += 2) {
      int mi = kind == 0 ? TIDX(b) : AIDX(b);
      orc_manifold* m = &e->man[mi];
      if (!use[mi] || !big[label[b]]) continue;
      for (int i = 0; i < m->n; ++i) {
        row_apply(e, kind, b, -1, &rows[mi][i], 0, m->ln[i]);
        row_apply(e, kind, b, -1, &rows[mi][i], 1, m->lt1[i]);
        row_apply(e, kind, b, -1, &rows[mi][i], 2, m->lt2[i]);
      }
    }
  for (int rd = 0; rd < 3; ++rd)
    for (int x = 0; x < 2; ++x) {
      int k = BB_ROUND[rd][x]; int mi = BBIDX(k);
      orc_manifold* m = &e->man[mi];
      if (!use[mi] || !big[label[BB_A[k]]]) continue;
      for (int i = 0; i < m->n; ++i) {
        row_apply(e, 1, BB_A[k], BB_B[k], &rows[mi][i], 0, m->ln[i]);
        row_apply(e, 1, BB_A[k], BB_B[k], &rows[mi][i], 1, m->lt1[i]);
        row_apply(e, 1, BB_A[k], BB_B[k], &rows[mi][i], 2, m->lt2[i]);
      }
    }
  for (int root = 0; root < RV_MAXB; ++root) {
    if (!big[root]) continue;
    real big_best = R(1e30); int big_since = 0;     /* rv_config.solver_stall */
    for (int it = 0; it < c->solver_iters; ++it) {
      real res = R(0.0);
      int rows_seen = 0;
      e->cnt_sweeps++; if (it == 0) e->cnt_islands++;
      for (int b = root; b < RV_MAXB; ++b) {
        if (!use[TIDX(b)] || label[b] != root) continue;
        orc_manifold* m = &e->man[TIDX(b)];
        for (int i = 0; i < m->n; ++i) { res = rmax(res, point_solve(e, 0, b, -1, m, i, &rows[TIDX(b)][i])); rows_seen++; }
        m = &e->man[AIDX(b)];
        for (int i = 0; i < m->n; ++i) { res = rmax(res, point_solve(e, 2, b, -1, m, i, &rows[AIDX(b)][i])); rows_seen++; }
      }
      for (int rd = 0; rd < 3; ++rd)
        for (int x = 0; x < 2; ++x) {
          int k = BB_ROUND[rd][x];
          if (!use[BBIDX(k)] || label[BB_A[k]] != root) continue;
          orc_manifold* m = &e->man[BBIDX(k)];
          for (int i = 0; i < m->n; ++i) { res = rmax(res, point_solve(e, 1, BB_A[k], BB_B[k], m, i, &rows[BBIDX(k)][i])); rows_seen++; }
        }
      if (rows_seen == 0 || res < island_tol(c, e, use, label, root)) break;   /* residual-based early exit */
      if (c->solver_stall > 0) { if (res < big_best) { big_best = res; big_since = 0; } else if (++big_since >= c->solver_stall) break; }
    }
  }
}

/* ------------------------------------------------- Simulator.step ------- */
/* Simulator.step (simulator.py:94-103): body.update() for the arm, then
 * physics.step() (bullet_physics.py:106-109), then num_steps += 1. */
static void sim_substep(const orc_world* w, orc_env* e) {
  const rv_config* c = &w->cfg;
  real dt = (real)c->dt;
  if (e->arm_enabled) {
    if (!w->ext_control) control_update(w, e);
    arm_motor_step(w, e);
    arm_update_kinematics(w, e);
  }
  /* wake sleeping bodies that an awake body or an arm collider comes near */
  {
    int wake[RV_MAXB];
    for (int b = 0; b < RV_MAXB; ++b) {
      wake[b] = 0;
      if (!(e->bp[b].active && !e->bp[b].frozen && e->bp[b].asleep)) continue;
      for (int a = 0; a < RV_MAXB; ++a) {
        /* only a MOVING neighbour wakes a sleeper (resting neighbours would ping-pong) */
        /* ... and moving means: left its 1 mm pose window within the last 50 substeps */
        if (a == b || !body_on(e, a) || e->bp[a].sleep_count > 0 || e->bp[a].still_count >= 50) continue;
        real d[3]; v3sub(d, e->body[a].p, e->body[b].p);
        real r = e->bp[a].radius + e->bp[b].radius + brk_bb(e, c, a, b);
        if (v3dot(d, d) < r * r) wake[b] = 1;
      }
      if (e->arm_enabled && e->arm_moving && !body_static(e, b)) {      /* (a static body has no manifold with the arm) */
        /* the arm wakes a sleeper when one of its boxes comes within the contact-
         * breaking distance of the body's hulls (= when a contact point would be
         * created); boxes whose AABB is farther than that from the body's are skipped */
        int nearf[RV_NCOL], near = 0;
        for (int col = 0; col < RV_NCOL; ++col) {
          real r = wake_range(w, e, b, col) + R(2.0) * (real)c->margin;
          nearf[col] = aabb_aabb_dist2(e->bp[b].aabb, e->bp[b].aabb + 3, e->colmin[col], e->colmax[col]) < r * r;
          near |= nearf[col];
        }
        if (near) {
          const rv_shape* s = &w->scene.shapes[e->bp[b].shape];
          real m[9]; qmat(m, e->body[b].q);
          real wv[RV_MAXH][RV_MAXV][3];
          for (int h = 0; h < s->n_hulls; ++h)
            for (int i = 0; i < s->n_verts[h]; ++i) {
              real l[3] = {(real)s->verts[h][i][0] * e->bp[b].scale, (real)s->verts[h][i][1] * e->bp[b].scale, (real)s->verts[h][i][2] * e->bp[b].scale};
              real t[3]; m3mulv(t, m, l); v3add(wv[h][i], e->body[b].p, t);
            }
          real mg = (real)c->margin;
          for (int col = 0; col < RV_NCOL && !wake[b]; ++col) {
            if (!nearf[col]) continue;
            const real brk = wake_range(w, e, b, col);
            real d[3]; v3sub(d, e->body[b].p, e->colc[col]);
            for (int h = 0; h < s->n_hulls && !wake[b]; ++h) {
              real n[3], dist, pa[3], pb[3];
              if (orc_gjk_epa((const real(*)[3])wv[h], s->n_verts[h], (const real(*)[3])e->colv[col], 8, d, brk + R(2.0) * mg, n, &dist, pa, pb))
                if (!(dist - R(2.0) * mg > brk)) wake[b] = 1;
            }
          }
        }
      }
    }
    /* a contact island is awake or asleep as a whole (Bullet deactivates islands, not bodies): a
     * sleeper that shares a manifold holding points with a body that is awake, or is being woken,
     * wakes as well -- otherwise that manifold would not be solved and the awake body would lose
     * its support */
    int any_wake = 0;      /* (islands sleep as a whole, so only a wake-up can leave a sleeper next to an awake body) */
    for (int b = 0; b < RV_MAXB; ++b) any_wake |= wake[b];
    if (any_wake) {
      int aw[RV_MAXB];
      for (int b = 0; b < RV_MAXB; ++b) aw[b] = body_on(e, b) || wake[b];
      for (int pass = 0; pass < 3; ++pass)
        for (int k = 0; k < RV_NBB; ++k) {
          const int a_ = BB_A[k], b_ = BB_B[k];
          if (!(e->bp[a_].active && !e->bp[a_].frozen && e->bp[b_].active && !e->bp[b_].frozen) || e->man[BBIDX(k)].n == 0) continue;
          if (aw[a_] && !aw[b_]) aw[b_] = 1;
          else if (aw[b_] && !aw[a_]) aw[a_] = 1;
        }
      for (int b = 0; b < RV_MAXB; ++b) if (aw[b] && e->bp[b].active && !e->bp[b].frozen && e->bp[b].asleep) wake[b] = 1;
    }
    for (int b = 0; b < RV_MAXB; ++b) if (wake[b]) {
      orc_bparam* P = &e->bp[b];
      P->asleep = 0; P->sleep_count = 0; P->deact_count = 0;
      /* open the pose window at the pose it was resting in */
      P->still_count = 1; P->undisturbed = 1;
      v3cpy(P->still_ref, e->body[b].p);
      for (int k = 0; k < 4; ++k) P->still_ref[3 + k] = e->body[b].q[k];
    }
  }
  for (int b = 0; b < RV_MAXB; ++b) {
    if (!body_on(e, b)) continue;
    if (body_static(e, b)) { e->mot[b] = R(0.0); continue; }
    orc_body* B = &e->body[b];
    B->v[0] += (real)c->gravity_xy[0] * dt; B->v[1] += (real)c->gravity_xy[1] * dt;
    B->v[2] += (real)c->gravity_z * dt;
    v3scale(B->v, B->v, (real)c->lin_damp);
    v3scale(B->w, B->w, (real)c->ang_damp);
    e->mot[b] = (v3len(B->v) + v3len(B->w) * e->bp[b].radius) * dt;
  }
  {
    int aw = 0, at = 0;
    for (int b = 0; b < RV_MAXB; ++b) aw |= body_on(e, b);
    e->awake_last += aw;
    if (e->arm_enabled) {
      /* the two rejection tests of the arm-table detector in collide_all() */
      real tc[3] = {(real)c->table_center[0], (real)c->table_center[1], e->table_z - R(0.5) * (real)c->table_thickness};
      real th[3] = {(real)c->table_half[0], (real)c->table_half[1], R(0.5) * (real)c->table_thickness};
      for (int col = 0; col < RV_NCOL; ++col) {
        real r = e->colr[col] + brk_col(w, col);
        if (!(e->colmin[col][2] - e->table_z - (real)c->margin >= (real)c->contact_query_dist) &&
            sphere_box_dist2(e->colc[col], tc, th) < r * r) at = 1;
      }
    }
    /* quiet substep: every body asleep (or absent) and no arm collider near the
     * table -> nothing to collide, solve or integrate */
    if (!aw && !at) {
      e->flag_arm_table = 0;
      for (int b = 0; b < RV_MAXB; ++b) e->flag_arm_body[b] = 0;
      e->sim_steps++;
      e->substeps_last++;
      return;
    }
  }
  bodies_prepare(w, e);
  collide_all(w, e);
  solve_contacts(w, e);
  int on_at_solve[RV_MAXB];      /* who is awake in this substep (the flags change in the loop below) */
  for (int b = 0; b < RV_MAXB; ++b) on_at_solve[b] = body_on(e, b);
  int ready[RV_MAXB] = {0, 0, 0, 0};   /* the body's own deactivation tests say it may sleep */
  for (int b = 0; b < RV_MAXB; ++b) {
    if (!body_on(e, b)) continue;
    orc_body* B = &e->body[b];
    /* rolling / spinning friction on the support: a resisting angular impulse of at most
     * rolling_friction x (normal impulse of the table manifold), never reversing the spin */
    if ((real)c->rolling_friction > R(0.0) && e->man[TIDX(b)].n > 0) {
      const orc_manifold* mt = &e->man[TIDX(b)];
      real nimp = R(0.0);
      for (int i = 0; i < mt->n; ++i) nimp += mt->ln[i];
      real wl = v3len(B->w);
      if (wl > R(0.0) && nimp > R(0.0)) {
        real dir[3], iw[3];
        v3scale(dir, B->w, R(1.0) / wl);
        m3mulv(iw, e->iinv[b], dir);
        real k = v3dot(dir, iw);
        real j = rmin((real)c->rolling_friction * nimp, wl / k);
        v3madd(B->w, B->w, iw, -j);
      }
    }
    real vv = v3dot(B->v, B->v), ww = v3dot(B->w, B->w);
    if (!body_static(e, b)) {      /* (a static body stays where it is: its velocities are zero, no impulse changes them) */
    v3madd(B->p, B->p, B->v, dt);
    real wq[4] = {B->w[0], B->w[1], B->w[2], R(0.0)}, dq[4];
    qmul(dq, wq, B->q);
    for (int k = 0; k < 4; ++k) B->q[k] += R(0.5) * dt * dq[k];
    qnormalize(B->q);
    if (B->p[2] < (real)c->ground_z - (real)c->fall_depth) {
      e->bp[b].frozen = 1;
      v3set(B->v, R(0.0), R(0.0), R(0.0)); v3set(B->w, R(0.0), R(0.0), R(0.0));
    }
    }
    /* substeps in a row below the sleep thresholds (the deactivation counter; also what makes a row a 'rest' row of the solver) */
    if (vv < (real)c->sleep_lin * (real)c->sleep_lin && ww < (real)c->sleep_ang * (real)c->sleep_ang) e->bp[b].sleep_count++;
    else e->bp[b].sleep_count = 0;
    if (c->sleep_steps > 0) {
      /* in-place oscillation: the pose has not left a small window around where it
       * was when the window opened */
      if ((real)c->sleep_pos_win > R(0.0)) {
        orc_bparam* P = &e->bp[b];
        int inside = 0;
        if (P->still_count > 0) {
          real dp[3]; v3sub(dp, B->p, P->still_ref);
          real dqm = R(0.0);
          for (int k = 0; k < 4; ++k) dqm = rmax(dqm, rabs(B->q[k] - P->still_ref[3 + k]));
          inside = v3dot(dp, dp) < (real)c->sleep_pos_win * (real)c->sleep_pos_win && dqm < (real)c->sleep_rot_win;
        }
        if (inside) P->still_count++;
        else {
          P->undisturbed = 0;
          P->still_count = 1;
          v3cpy(P->still_ref, B->p);
          for (int k = 0; k < 4; ++k) P->still_ref[3 + k] = B->q[k];
        }
      }
      /* a sleeper that was woken but never left the pose it was resting in goes back to
       * sleep after a quarter of the usual wait */
      int quick = e->bp[b].undisturbed && 4 * e->bp[b].still_count >= c->sleep_steps && 4 * e->bp[b].sleep_count >= c->sleep_steps;
      /* a body the force-limited gripper holds stays active (its island contains the moving fingers) */
      int held = (c->finger_dynamics && e->man[AIDX(b)].n > 0) || con_pair_member(e, b);
      /* Bullet's own rule (0.8 m/s, 1 rad/s, 2 s) for a body whose island is the body alone */
      int deact = 0;
      if (c->deact_steps > 0) {
        int free_ = e->man[AIDX(b)].n == 0;
        for (int k = 0; k < RV_NBB; ++k) {
          int a_ = BB_A[k], b_ = BB_B[k];
          if ((a_ == b || b_ == b) && e->man[BBIDX(k)].n != 0 && on_at_solve[a_ == b ? b_ : a_]) free_ = 0;
        }
        if (free_ && vv < (real)c->deact_lin * (real)c->deact_lin && ww < (real)c->deact_ang * (real)c->deact_ang) e->bp[b].deact_count++;
        else e->bp[b].deact_count = 0;
        deact = e->bp[b].deact_count >= c->deact_steps;
      }
      ready[b] = e->bp[b].frozen || (!held && (e->bp[b].sleep_count >= c->sleep_steps || e->bp[b].still_count >= c->sleep_steps || quick || deact));
    }
#ifdef ORC_TRACE_AWAKE
    /* diagnostic (tools): one line per awake body and substep of env ORC_TRACE_ENV */
    { static int tenv = -2; if (tenv == -2) { const char* t_ = getenv("ORC_TRACE_ENV"); tenv = t_ ? atoi(t_) : -1; }
      if (tenv == (int)(e - w->env)) fprintf(stderr, "T %d ph %d b %d v %.4f w %.3f sc %d st %d arm %d tab %d\n", e->sim_steps, e->phase, b,
          (double)rsqrt_(vv), (double)rsqrt_(ww), e->bp[b].sleep_count, e->bp[b].still_count, e->man[AIDX(b)].n, e->man[TIDX(b)].n); }
#endif
  }
  /* islands go to sleep as a whole: a body sleeps when every awake body it is coupled to
   * (transitively) by manifolds that hold points is ready as well */
  {
    int label[RV_MAXB];
    for (int b = 0; b < RV_MAXB; ++b) label[b] = b;
    for (int pass = 0; pass < RV_MAXB; ++pass)
      for (int k = 0; k < RV_NBB; ++k) {
        const int a_ = BB_A[k], b_ = BB_B[k];
        if (!(on_at_solve[a_] && on_at_solve[b_] && e->man[BBIDX(k)].n != 0)) continue;
        const int lo = label[a_] < label[b_] ? label[a_] : label[b_];
        label[a_] = lo; label[b_] = lo;
      }
    for (int b = 0; b < RV_MAXB; ++b) {
      if (!on_at_solve[b] || !ready[b] || e->bp[b].frozen) continue;
      int all = 1;
      for (int x = 0; x < RV_MAXB; ++x) if (on_at_solve[x] && label[x] == label[b] && !ready[x]) all = 0;
      if (!all) continue;
      orc_body* B = &e->body[b];
      {
        e->bp[b].asleep = 1;
        v3set(B->v, R(0.0), R(0.0), R(0.0)); v3set(B->w, R(0.0), R(0.0), R(0.0));
        /* world box of the resting hulls: what the arm has to come near to wake the body */
        const rv_shape* s = &w->scene.shapes[e->bp[b].shape];
        real m[9]; qmat(m, B->q);
        real* bx = e->bp[b].aabb;
        for (int k = 0; k < 3; ++k) { bx[k] = R(1e30); bx[3 + k] = R(-1e30); }
        for (int h = 0; h < s->n_hulls; ++h)
          for (int i = 0; i < s->n_verts[h]; ++i) {
            real l[3] = {(real)s->verts[h][i][0] * e->bp[b].scale, (real)s->verts[h][i][1] * e->bp[b].scale, (real)s->verts[h][i][2] * e->bp[b].scale};
            real t[3], p[3]; m3mulv(t, m, l); v3add(p, B->p, t);
            for (int k = 0; k < 3; ++k) { bx[k] = rmin(bx[k], p[k]); bx[3 + k] = rmax(bx[3 + k], p[k]); }
          }
        for (int k = 0; k < 3; ++k) { bx[k] -= (real)c->margin; bx[3 + k] += (real)c->margin; }
      }
    }
  }
  e->sim_steps++;
  e->substeps_last++;
}

/* Simulator.check_stable over a body mask (simulator.py:289-323) */
static int bodies_stable(const orc_env* e, unsigned mask, real lin_thr, real ang_thr) {
  for (int b = 0; b < RV_MAXB; ++b) {
    if (!((mask >> b) & 1u) || !e->bp[b].active) continue;
    real lv = v3len(e->body[b].v), av = v3len(e->body[b].w);
    if (lv >= lin_thr || av >= ang_thr) return 0;
  }
  return 1;
}
/* Simulator.wait_until_stable (simulator.py:325-376) */
static int wait_until_stable(const orc_world* w, orc_env* e, unsigned mask, real lin_thr, real ang_thr,
                             int check_after, int min_stable, int max_steps) {
  int num_steps = 0, num_stable = 0;
  for (;;) {
    sim_substep(w, e);
    num_steps++;
    if (num_steps < check_after) continue;
    if (bodies_stable(e, mask, lin_thr, ang_thr)) num_stable++;
    if (num_stable >= min_stable || num_steps >= max_steps) break;
  }
  return num_steps;
}

/* --------------------------------------------------- observation/reward -- */
static void compute_obs(orc_env* e) {
  memcpy(e->prev_obs_pos, e->obs_pos, sizeof(e->obs_pos));
  for (int b = 0; b < RV_MAXB; ++b) {
    if (body_movable(e, b)) v3cpy(e->obs_pos[b], e->body[b].p);      /* (self.movable_bodies: a static body is not one) */
    else v3set(e->obs_pos[b], R(0.0), R(0.0), R(0.0));
  }
}

/* check_on_tiles (push_reward.py:57-68): note the tile centres use `size` */
static int on_tiles(const real* xy, const float (*tiles)[2], int n, real size, const float* offset, real max_dist) {
  for (int i = 0; i < n; ++i) {
    real tx = (real)offset[0] + (real)tiles[i][0] * size;
    real ty = (real)offset[1] + (real)tiles[i][1] * size;
    if (rabs(xy[0] - tx) <= R(0.5) * max_dist && rabs(xy[1] - ty) <= R(0.5) * max_dist) return 1;
  }
  return 0;
}
/* get_tile_dists (push_reward.py:71-76) */
static real tile_dist(const real* xy, const float (*tiles)[2], int n, real size, const float* offset) {
  real best = R(1e30);
  for (int i = 0; i < n; ++i) {
    real dx = xy[0] - ((real)offset[0] + (real)tiles[i][0] * size);
    real dy = xy[1] - ((real)offset[1] + (real)tiles[i][1] * size);
    real d = rsqrt_(dx * dx + dy * dy);
    if (d < best) best = d;
  }
  return best;
}
static real task_score(const rv_config* c, real st[][3]) {
  real size = (real)c->tile_size;
  if (c->task == RV_TASK_CLEARING) {
    /* clearing_score (push_reward.py:101-107), including the overwritten min */
    real d1 = R(0.0), d3 = R(0.0);
    for (int b = 0; b < RV_MAXB; ++b) { d1 += rabs(st[b][0] - R(0.7)); d3 += rabs(st[b][1] + R(0.9)); }
    d1 /= (real)RV_MAXB; d3 /= (real)RV_MAXB;
    return -rmin(d1, d3);
  }
  return -tile_dist(st[0], c->goal, c->n_goal, size, c->tile_offset);
}
/* get_reward_fn(...).reward_fn with is_planning=False (push_reward.py:302-372) */
static void compute_reward(const orc_world* w, orc_env* e, real* reward, int* termination) {
  const rv_config* c = &w->cfg;
  if (c->task == RV_TASK_NONE) { *reward = R(1.0); *termination = 0; return; } /* dummy_reward_fn :34-47 */
  real size = (real)c->tile_size;
  int term = 0, goal = 0;
  if (c->task == RV_TASK_CROSSING)
    term = !on_tiles(e->obs_pos[0], c->region, c->n_region, size, c->tile_offset, size * R(1.5));
  if (c->task == RV_TASK_CLEARING) {
    goal = 1;
    for (int b = 0; b < RV_MAXB; ++b)
      if (on_tiles(e->obs_pos[b], c->region, c->n_region, size * R(1.25), c->tile_offset, size * R(1.25))) goal = 0;
  } else {
    goal = on_tiles(e->obs_pos[0], c->goal, c->n_goal, size, c->tile_offset, size);
  }
  goal = goal && !term;
  real r = R(0.0);
  r += R(100.0) * (real)goal;
  int penalty = term && !goal;
  r += R(-100.0) * (real)penalty;
  r += rabs(task_score(c, e->obs_pos) - task_score(c, e->prev_obs_pos));
  r += R(-1.0);
  *reward = r; *termination = term || goal;
}

/* --------------------------------------------------------- PushEnv ------- */
static void set_gripper_pose(real* pose, real x, real y, real z) {
  /* euler [pi, 0, 0] (push_env.py:771,782) */
  pose[0] = x; pose[1] = y; pose[2] = z;
  euler_to_quat(pose + 3, ORC_PI, R(0.0), R(0.0));
}
/* PushEnv._compute_waypoints (push_env.py:752-786) */
static void compute_waypoints(const rv_config* c, const real* action, real* start, real* end) {
  real lo0 = (real)c->cspace_low[0], hi0 = (real)c->cspace_high[0];
  real lo1 = (real)c->cspace_low[1], hi1 = (real)c->cspace_high[1];
  real lo2 = (real)c->cspace_low[2], hi2 = (real)c->cspace_high[2];
  real x = action[0] * (R(0.5) * (hi0 - lo0)) + R(0.5) * (hi0 + lo0);
  real y = action[1] * (R(0.5) * (hi1 - lo1)) + R(0.5) * (hi1 + lo1);
  real z = (real)c->finger_tip_offset + R(0.5) * (hi2 + lo2);
  set_gripper_pose(start, x, y, z);
  real ex = rclamp(x + action[2] * (real)c->translation_x, lo0, hi0);
  real ey = rclamp(y + action[3] * (real)c->translation_y, lo1, hi1);
  set_gripper_pose(end, ex, ey, z);
}
static int arm_touches_movables(const orc_env* e) {
  for (int b = 0; b < RV_MAXB; ++b) if (body_movable(e, b) && e->flag_arm_body[b]) return 1;
  return 0;
}
/* PushEnv._check_safety (push_env.py:857-898) */
static int check_safety(const orc_world* w, const orc_env* e, real start_z) {
  const rv_config* c = &w->cfg;
  if (e->phase == RV_PHASE_PRE) { if (arm_touches_movables(e)) return 0; }
  if (e->phase == RV_PHASE_START) {
    if (arm_touches_movables(e)) {
      real dist = e->fpos[7][2] - start_z;
      if (rabs(dist) <= R(0.01)) return 1;
      return 0;
    }
  }
  if (e->phase == RV_PHASE_DONE) {
    if (arm_touches_movables(e)) return 0;
    real lx = (real)c->table_center[0] - R(0.5) * (real)c->workspace_x_range, hx = (real)c->table_center[0] + R(0.5) * (real)c->workspace_x_range;
    real ly = (real)c->table_center[1] - R(0.5) * (real)c->workspace_y_range, hy = (real)c->table_center[1] + R(0.5) * (real)c->workspace_y_range;
    for (int b = 0; b < RV_MAXB; ++b) {
      if (!body_movable(e, b)) continue;
      const real* p = e->body[b].p;
      if (p[0] < lx || p[0] > hx || p[1] < ly || p[1] > hy) return 0;
    }
  }
  return 1;
}

/* PushEnv._execute_action (push_env.py:631-733) */
static void execute_action(const orc_world* w, orc_env* e) {
  const rv_config* c = &w->cfg;
  int G = c->num_goal_steps > 0 ? c->num_goal_steps : 1;
  real wp[RV_MAXG][2][7];
  for (int g = 0; g < G; ++g) compute_waypoints(c, e->action[g], wp[g][0], wp[g][1]);
  if (w->pose_f32)   /* Pose([[x, y, z], [pi, 0, 0]]): Euler angles are stored as float32 */
    for (int g = 0; g < G; ++g) for (int k = 0; k < 2; ++k)
      euler_to_quat(wp[g][k] + 3, (real)(float)ORC_PI, R(0.0), R(0.0));
  real start_z = (real)c->finger_tip_offset + R(0.5) * ((real)c->cspace_high[2] + (real)c->cspace_low[2]);
  e->is_safe = 1; e->is_effective = 1;
  e->phase = RV_PHASE_INITIAL;
  int num_waypoints = 0, interrupt = 0, has_budget = 0, max_phase_steps = 0;
  real start_pos[RV_MAXB][3], start_yaw[RV_MAXB];
  for (int b = 0; b < RV_MAXB; ++b) { v3cpy(start_pos[b], e->body[b].p); start_yaw[b] = quat_yaw(e->body[b].q); }
  while (e->phase != RV_PHASE_DONE) {
    sim_substep(w, e);
    if (e->sim_steps % c->steps_check != 0) continue;
    /* _is_phase_ready (push_env.py:812-837) */
    int ready = 0;
    if (interrupt) ready = 1;
    else if (arm_is_ready_limb(w, e) && robot_is_gripper_ready(w, e)) { arm_reset_targets(e); ready = 1; }
    else if (!has_budget) ready = 1;
    else if (e->sim_steps >= max_phase_steps) { arm_reset_targets(e); ready = 1; }
    if (ready) {
      /* _get_next_phase (push_env.py:788-810) */
      int next;
      if (interrupt && e->phase != RV_PHASE_POST && e->phase != RV_PHASE_OFFSTAGE) next = RV_PHASE_POST;
      else if (c->num_goal_steps > 0 && e->phase == RV_PHASE_POST && num_waypoints < c->num_goal_steps) next = RV_PHASE_PRE;
      else next = e->phase + 1;
#ifdef ORC_TRACE
      fprintf(stderr, "phase %d -> %d at sim_step %d (lt %d jt %d) ee=(%.3f %.3f %.3f) interrupt=%d\n", e->phase, next, e->sim_steps, e->lt.active, e->jt.active, (double)e->fpos[7][0], (double)e->fpos[7][1], (double)e->fpos[7][2], interrupt);
#endif
      e->phase = next;
      has_budget = 1;
      max_phase_steps = e->sim_steps;
      if (next == RV_PHASE_MOTION) max_phase_steps += c->max_motion_steps;
      else if (next == RV_PHASE_OFFSTAGE) max_phase_steps += c->max_offstage_steps;
      else max_phase_steps += c->max_phase_steps;
      if (next == RV_PHASE_PRE) {
        real pose[7]; memcpy(pose, wp[num_waypoints][0], sizeof(pose));
        pose[2] = (real)c->gripper_safe_height;
        robot_move_to_gripper_pose(w, e, pose);
      } else if (next == RV_PHASE_START) {
        robot_move_to_gripper_pose(w, e, wp[num_waypoints][0]);
      } else if (next == RV_PHASE_MOTION) {
        robot_move_to_gripper_pose(w, e, wp[num_waypoints][1]);
      } else if (next == RV_PHASE_POST) {
        num_waypoints++;
        real pose[7];
        v3cpy(pose, e->fpos[7]); memcpy(pose + 3, e->fquat[7], sizeof(real) * 4);
        if (w->pose_f32) for (int i = 3; i < 7; ++i) pose[i] = (real)(float)pose[i];   /* Pose([p, quaternion]) */
        pose[2] = (real)c->gripper_safe_height;
        robot_move_to_gripper_pose(w, e, pose);
      } else if (next == RV_PHASE_OFFSTAGE) {
        real off[RV_NLIMB];
        for (int j = 0; j < RV_NLIMB; ++j) off[j] = (real)c->offstage_positions[j];
        robot_move_to_joint_positions(w, e, off);
      }
    }
    interrupt = 0;
    /* _check_singularity (push_env.py:839-855) */
    if (e->phase == RV_PHASE_MOTION && e->flag_arm_table) interrupt = 1;
    if (!check_safety(w, e, start_z)) { interrupt = 1; e->is_safe = 0; }
    if (interrupt && e->phase == RV_PHASE_DONE) { e->done = 1; break; }
  }
  unsigned mask = 0;
  for (int b = 0; b < RV_MAXB; ++b) if (e->bp[b].active) mask |= 1u << b;
  wait_until_stable(w, e, mask, R(0.005), R(0.005), 100, 100, 2000);
  /* _check_effectiveness (push_env.py:900-923) */
  real dpos = R(0.0), dang = R(0.0);
  for (int b = 0; b < RV_MAXB; ++b) {
    if (!e->bp[b].active) continue;
    real d[3]; v3sub(d, e->body[b].p, start_pos[b]);
    dpos += v3len(d);
    real da = quat_yaw(e->body[b].q) - start_yaw[b];
    da = da + ORC_PI;
    da = da - R(2.0) * ORC_PI * R(floor)(da / (R(2.0) * ORC_PI));
    da = da - ORC_PI;
    dang += rabs(da);
  }
  if (dpos <= (real)c->min_delta_position && dang <= (real)c->min_delta_angle) e->is_effective = 0;
  e->num_total_steps++;
  e->num_unsafe += !e->is_safe;
  e->num_ineffective += !e->is_effective;
  e->num_useful += (e->is_safe && e->is_effective);
}

/* RobotEnv.step (robot_env.py:239-275) + PushEnv.step (push_env.py:599-629) */
static void env_step(const orc_world* w, orc_env* e) {
  const rv_config* c = &w->cfg;
  if (e->done) return;
  e->substeps_last = 0; e->awake_last = 0; e->pairs_last = 0;
  e->obs_num_steps = e->num_steps; e->obs_num_episodes = e->num_episodes;
  execute_action(w, e);
  e->l_unsafe += !e->is_safe; e->l_ineffective += !e->is_effective; e->l_useful += (e->is_safe && e->is_effective);
  e->num_steps++;
  compute_obs(e);
  real r; int term;
  compute_reward(w, e, &r, &term);
  e->last_reward = r;
  e->episode_reward += r;
  e->done = e->done || term;
  if (c->max_steps > 0 && e->num_steps >= c->max_steps) e->done = 1;
  if (e->done) {
    e->num_episodes++; e->l_episodes++;
    if (r >= (real)c->success_thresh) { e->num_successes++; e->l_successes++; }
  }
}

/* ------------------------------------------------------- Grasp4DofEnv ---- */
/* SawyerSim.move_along_gripper_path -> set_target_link_poses (sawyer_sim.py:310-360,
 * controllable_body.py:319-345): the poses are reached one after the other */
static void robot_move_along_gripper_path(const orc_world* w, orc_env* e, real poses[][7], int n) {
  robot_move_to_gripper_pose(w, e, poses[0]);
  e->lt.has_pose = 0; e->lt.nq = n;
  for (int q = 0; q < n; ++q) memcpy(e->lt.queue[q], poses[q], sizeof(real) * 7);
  lt_pop(&e->lt);
}
/* SawyerSim.move_to_gripper_pose(pose, straight_line=True) (sawyer_sim.py:259-276) */
static void robot_move_straight(const orc_world* w, orc_env* e, const real* pose) {
  const rv_config* c = &w->cfg;
  real d[3]; v3sub(d, pose, e->fpos[7]);
  int num = (int)(v3len(d) / (real)c->end_effector_step);
  if (num > RV_MAXQ - 1) num = RV_MAXQ - 1;
  real wps[RV_MAXQ][7];
  for (int i = 0; i < num; ++i) {
    real sc = (real)i / (real)num;
    v3madd(wps[i], e->fpos[7], d, sc);
    for (int k = 3; k < 7; ++k) wps[i][k] = pose[k];
  }
  memcpy(wps[num], pose, sizeof(real) * 7);
  robot_move_along_gripper_path(w, e, wps, num + 1);
}
/* Grasp4DofEnv._execute_action (grasp_4dof_env.py:213-345) */
static void execute_grasp(const orc_world* w, orc_env* e) {
  const rv_config* c = &w->cfg;
  real start[7];
  start[0] = e->action[0][0]; start[1] = e->action[0][1]; start[2] = e->action[0][2] + (real)c->finger_tip_offset;
  euler_to_quat(start + 3, R(0.0), w->pose_f32 ? (real)(float)ORC_PI : ORC_PI, w->pose_f32 ? (real)(float)e->action[0][3] : e->action[0][3]);
  e->is_safe = 1; e->is_effective = 1;
  e->phase = RV_GPHASE_INITIAL; e->num_action_steps = 0;
  while (e->phase != RV_GPHASE_DONE) {
    sim_substep(w, e);
    if (e->phase == RV_GPHASE_START) e->num_action_steps++;
    /* _is_phase_ready (:322-345) */
    int ready = 0;
    if (e->phase == RV_GPHASE_START && e->num_action_steps >= c->max_action_steps) ready = 1;
    else if ((e->phase == RV_GPHASE_START || e->phase == RV_GPHASE_END) && e->flag_arm_table) ready = 1;
    else if (arm_is_ready_limb(w, e) && robot_is_gripper_ready(w, e)) ready = 1;
    if (!ready) continue;
    e->phase = e->phase + 1;
    if (e->phase == RV_GPHASE_OVERHEAD) {
      real q[RV_NLIMB];
      for (int j = 0; j < RV_NLIMB; ++j) q[j] = (real)c->overhead_positions[j];
      robot_move_to_joint_positions(w, e, q);
    } else if (e->phase == RV_GPHASE_PRESTART) {
      real pose[7]; memcpy(pose, start, sizeof(pose));
      pose[2] = (real)c->gripper_safe_height;
      robot_move_to_gripper_pose(w, e, pose);
    } else if (e->phase == RV_GPHASE_START) {
      robot_move_straight(w, e, start);
      e->mu_finger = (real)c->grasp_mu_descend[0]; e->mu_table = (real)c->grasp_mu_descend[1];
    } else if (e->phase == RV_GPHASE_END) {
      robot_grip(w, e, R(1.0));
    } else if (e->phase == RV_GPHASE_POSTEND) {
      real pose[7];
      v3cpy(pose, e->fpos[7]); memcpy(pose + 3, e->fquat[7], sizeof(real) * 4);
      if (w->pose_f32) for (int i = 3; i < 7; ++i) pose[i] = (real)(float)pose[i];
      pose[2] = (real)c->gripper_safe_height;
      robot_move_straight(w, e, pose);
      e->mu_finger = (real)c->grasp_mu_lift[0]; e->mu_table = (real)c->grasp_mu_lift[1];
    }
  }
}
/* RobotEnv.step for Grasp4DofEnv + GraspReward.get_reward (grasp_reward.py:49-68) */
static void genv_step(const orc_world* w, orc_env* e) {
  if (e->done) return;
  e->substeps_last = 0; e->awake_last = 0; e->pairs_last = 0;
  e->obs_num_steps = e->num_steps; e->obs_num_episodes = e->num_episodes;
  execute_grasp(w, e);
  e->num_steps++;
  compute_obs(e);
  unsigned mask = 0;
  for (int b = 0; b < RV_MAXB; ++b) if (e->bp[b].active) mask |= 1u << b;
  wait_until_stable(w, e, mask, R(0.005), R(0.005), 100, 100, 2000);
  const int success = arm_touches_movables(e);
  const real r = success ? R(1.0) : R(0.0);
  e->is_effective = success;
  e->num_total_steps++;
  e->num_useful += success; e->l_useful += success;
  e->num_ineffective += !success; e->l_ineffective += !success;
  e->last_reward = r; e->episode_reward += r;
  e->done = 1;
  e->num_episodes++; e->l_episodes++;
  if (success) { e->num_successes++; e->l_successes++; }
}
static void step_env(const orc_world* w, orc_env* e) { if (w->cfg.env_type == RV_ENV_GRASP) genv_step(w, e); else env_step(w, e); }
/* RandomPolicy._action = action_space.sample() (see random_action in rv_dev_env.h) */
static void random_action(const rv_config* c, int gid, int macro_index, real* a) {
  int G = c->num_goal_steps > 0 ? c->num_goal_steps : 1;
  orc_rng g; rng_init(&g, c->seed_lo, c->seed_hi, (uint32_t)gid, STREAM_RANDOM, (uint32_t)macro_index);
  for (int x = 0; x < G * 4; ++x) a[x] = rng_uniform(&g, R(-1.0), R(1.0));
  if (c->env_type == RV_ENV_GRASP) {
    for (int k = 0; k < 3; ++k) a[k] = (real)c->grasp_cuboid_low[k] + ((real)c->grasp_cuboid_high[k] - (real)c->grasp_cuboid_low[k]) * (R(0.5) * (a[k] + R(1.0)));
    a[3] = ORC_PI * (a[3] + R(1.0));
  }
}

/* --------------------------------------------------------------- reset --- */
/* PushEnv._sample_body_poses / _sample_body_poses_on_tiles, intent version
 * (push_env.py:473-597; SURVEY.md Appendix B-7 documents why the reference's
 * RNG stream is not replicated). */
static void sample_poses(const orc_world* w, orc_env* e, orc_rng* g, int nb, real poses[][7]) {
  const rv_config* c = &w->cfg;
  for (;;) {
    int ok = 1;
    for (int i = 0; i < nb && ok; ++i) {
      int placed = 0;
      for (int att = 0; att <= 32 && !placed; ++att) {
        real x, y, z, roll, pitch, yaw;
        if (c->use_tiles) {
          const float (*tiles)[2] = (i == 0 && c->n_target > 0) ? c->target : c->obstacle;
          int nt = (i == 0 && c->n_target > 0) ? c->n_target : c->n_obstacle;
          int tid = rng_randint(g, nt);
          real cx = (real)c->tile_offset[0] + (real)tiles[tid][0] * (real)c->tile_size;
          real cy = (real)c->tile_offset[1] + (real)tiles[tid][1] * (real)c->tile_size;
          x = rng_uniform(g, cx - R(0.5) * (real)c->tile_size, cx + R(0.5) * (real)c->tile_size);
          y = rng_uniform(g, cy - R(0.5) * (real)c->tile_size, cy + R(0.5) * (real)c->tile_size);
          z = e->table_z + (real)c->safe_drop_height;
          roll = rng_uniform(g, -ORC_PI, ORC_PI);
          pitch = rng_uniform(g, R(-0.5) * ORC_PI, R(0.5) * ORC_PI);
          yaw = rng_uniform(g, -ORC_PI, ORC_PI);
        } else {
          x = rng_uniform(g, (real)c->pose_lo[0], (real)c->pose_hi[0]);
          y = rng_uniform(g, (real)c->pose_lo[1], (real)c->pose_hi[1]);
          z = e->table_z + rng_uniform(g, (real)c->pose_lo[2], (real)c->pose_hi[2]);
          roll = rng_uniform(g, (real)c->pose_lo[3], (real)c->pose_hi[3]);
          pitch = rng_uniform(g, (real)c->pose_lo[4], (real)c->pose_hi[4]);
          yaw = rng_uniform(g, (real)c->pose_lo[5], (real)c->pose_hi[5]);
        }
        int valid = 1;
        for (int j = 0; j < i; ++j) {
          real dx = x - poses[j][0], dy = y - poses[j][1];
          if (rsqrt_(dx * dx + dy * dy) < (real)c->margin_xy) { valid = 0; break; }
        }
        if (valid) {
          poses[i][0] = x; poses[i][1] = y; poses[i][2] = z;
          euler_to_quat(poses[i] + 3, roll, pitch, yaw);
          placed = 1;
        }
      }
      if (!placed) ok = 0;
    }
    if (ok) return;
  }
}

/* RobotEnv.reset (robot_env.py:204-237) for one env */
/* ArmEnv._reset_camera (arm_env.py:109-152; push_env.py:273-280): rv_config's calibration plus uniform noise, element by
 * element, from a Philox stream of its own (the scene of an episode does not depend on the camera noise) */
#define STREAM_CAMERA 4u
static void camera_reset(const rv_config* c, orc_env* e, int gid, int use_noise) {
  orc_rng g; rng_init(&g, c->seed_lo, c->seed_hi, (uint32_t)gid, STREAM_CAMERA, (uint32_t)e->reset_count);
  for (int k = 0; k < 17; ++k) {
    const real base = (real)(k < 5 ? c->cam_intrinsics[k] : (k < 14 ? c->cam_rotation[k - 5] : c->cam_translation[k - 14]));
    const real nz = use_noise ? rng_uniform(&g, -(real)c->cam_noise[k], (real)c->cam_noise[k]) : R(0.0);
    const real v = base + nz;
    if (k < 5) e->cam_intrinsics[k] = v; else if (k < 14) e->cam_rotation[k - 5] = v; else e->cam_translation[k - 14] = v;
  }
}
/* ArmEnv._reset_scene's wall (arm_env.py:94-99): simulator.add_body(SIM.WALL.PATH, SIM.WALL.POSE, is_static=True) -- a
 * static body (mass 0) in the last body slot */
static void wall_place(const orc_world* w, orc_env* e) {
  const rv_config* c = &w->cfg;
  if (!c->wall_use) return;
  const int b = RV_MAXB - 1;
  orc_bparam* p = &e->bp[b];
  p->active = 1; p->frozen = 0; p->asleep = 0; p->sleep_count = 0; p->deact_count = 0; p->still_count = 0; p->undisturbed = 0; p->con_on = 0;
  p->shape = c->wall_shape; p->scale = (real)c->wall_scale; p->friction = R(1.0);     /* (URDF template default, tools/templates/urdf_template.xml:11-22) */
  body_set_mass(w, e, b, R(0.0));
  for (int k = 0; k < 3; ++k) e->body[b].p[k] = (real)c->wall_pose[k];
  for (int k = 0; k < 4; ++k) e->body[b].q[k] = (real)c->wall_pose[3 + k];
  v3set(e->body[b].v, R(0.0), R(0.0), R(0.0)); v3set(e->body[b].w, R(0.0), R(0.0), R(0.0));
}
static void env_reset(const orc_world* w, orc_env* e, int gid) {
  const rv_config* c = &w->cfg;
  orc_rng g; rng_init(&g, c->seed_lo, c->seed_hi, (uint32_t)gid, STREAM_RESET, (uint32_t)e->reset_count);
  camera_reset(c, e, gid, 1);
  e->reset_count++;
  e->substeps_last = 0; e->awake_last = 0; e->pairs_last = 0;
  e->sim_steps = 0; e->num_steps = 0; e->episode_reward = R(0.0); e->last_reward = R(0.0);
  e->obs_num_steps = 0; e->obs_num_episodes = e->num_episodes;
  e->done = 0;
  e->phase = RV_PHASE_INITIAL; e->is_safe = 1; e->is_effective = 1;
  e->arm_enabled = 0;
  e->mu_finger = (real)c->arm_friction; e->mu_table = (real)c->table_friction; e->num_action_steps = 0;
  for (int i = 0; i < RV_NMAN; ++i) e->man[i].n = 0;
  e->flag_arm_table = 0;
  for (int b = 0; b < RV_MAXB; ++b) e->flag_arm_body[b] = 0;
  /* ArmEnv._reset_scene (arm_env.py:78-99) */
  e->table_z = (real)c->table_z + rng_uniform(&g, (real)c->table_height_range[0], (real)c->table_height_range[1]);
  table_prepare(w, e);
  /* PushEnv._reset_scene / _load_movable_bodies (push_env.py:331-471) */
  int nb = c->n_bodies_min + rng_randint(&g, c->n_bodies_max - c->n_bodies_min + 1);
  e->n_bodies = nb;
  if (c->env_type == RV_ENV_GRASP) {
    /* Grasp4DofEnv._reset_scene (grasp_4dof_env.py:166-198) */
    real poses[RV_MAXB][7];
    for (int b = 0; b < RV_MAXB; ++b) { e->bp[b].active = 0; e->bp[b].frozen = 0; e->bp[b].asleep = 0; e->bp[b].sleep_count = 0; e->bp[b].deact_count = 0; e->bp[b].still_count = 0; e->bp[b].undisturbed = 0; e->bp[b].con_on = 0; }
    e->n_bodies = 1;
    wall_place(w, e);
    sample_poses(w, e, &g, 1, poses);
    int shape = c->movable_shapes[rng_randint(&g, c->n_movable_shapes)];
    real scale = rng_uniform(&g, (real)c->scale_range[0], (real)c->scale_range[1]);
    real mass = rng_uniform(&g, (real)c->mass_range[0], (real)c->mass_range[1]);
    real fr = rng_uniform(&g, (real)c->friction_range[0], (real)c->friction_range[1]);
    orc_bparam* p = &e->bp[0];
    p->active = 1; p->shape = shape; p->scale = scale; p->friction = fr;
    body_set_mass(w, e, 0, mass);
    v3cpy(e->body[0].p, poses[0]); memcpy(e->body[0].q, poses[0] + 3, sizeof(real) * 4);
    v3set(e->body[0].v, R(0.0), R(0.0), R(0.0)); v3set(e->body[0].w, R(0.0), R(0.0), R(0.0));
  } else
  for (;;) {
    real poses[RV_MAXB][7];
    for (int b = 0; b < RV_MAXB; ++b) { e->bp[b].active = 0; e->bp[b].frozen = 0; e->bp[b].asleep = 0; e->bp[b].sleep_count = 0; e->bp[b].deact_count = 0; e->bp[b].still_count = 0; e->bp[b].undisturbed = 0; e->bp[b].con_on = 0; }
    for (int i = 0; i < RV_NMAN; ++i) e->man[i].n = 0;
    wall_place(w, e);
    sample_poses(w, e, &g, nb, poses);
    for (int i = 0; i < nb; ++i) {
      int use_target = (i == 0 && c->use_tiles && c->n_target > 0 && c->n_target_shapes > 0);
      int shape = use_target ? c->target_shapes[rng_randint(&g, c->n_target_shapes)]
                             : c->movable_shapes[rng_randint(&g, c->n_movable_shapes)];
      real scale = rng_uniform(&g, (real)c->scale_range[0], (real)c->scale_range[1]);
      orc_bparam* p = &e->bp[i];
      p->active = 1; p->frozen = 0; p->asleep = 0; p->sleep_count = 0; p->deact_count = 0; p->still_count = 0; p->undisturbed = 0; p->undisturbed = 0; p->shape = shape; p->scale = scale; p->friction = (real)c->drop_friction;
      body_set_mass(w, e, i, (real)c->drop_mass);
      v3cpy(e->body[i].p, poses[i]); memcpy(e->body[i].q, poses[i] + 3, sizeof(real) * 4);
      v3set(e->body[i].v, R(0.0), R(0.0), R(0.0)); v3set(e->body[i].w, R(0.0), R(0.0), R(0.0));
      wait_until_stable(w, e, 1u << i, R(0.1), R(0.1), 100, 100, 500);
      real mass = rng_uniform(&g, (real)c->mass_range[0], (real)c->mass_range[1]);
      real fr = rng_uniform(&g, (real)c->friction_range[0], (real)c->friction_range[1]);
      body_set_mass(w, e, i, mass); p->friction = fr;
    }
    int valid = 1;
    for (int i = 0; i < nb; ++i) if (e->body[i].p[2] < e->table_z) valid = 0;
    if (valid) break;
  }
  unsigned mask = 0;
  for (int b = 0; b < RV_MAXB; ++b) if (e->bp[b].active) mask |= 1u << b;
  wait_until_stable(w, e, mask, R(0.005), R(0.005), 100, 100, 2000);
  /* ArmEnv._reset_robot (arm_env.py:101-107) -> SawyerSim.reboot (sawyer_sim.py:86-171) */
  const rv_arm* a = &w->scene.arm;
  for (int j = 0; j < RV_NLIMB; ++j) { e->q[j] = (real)c->neutral_positions[j]; e->qd[j] = R(0.0); }
  e->q[7] = (real)a->q_hi[7]; e->q[8] = (real)a->q_lo[8]; e->qd[7] = e->qd[8] = R(0.0);
  for (int j = 0; j < RV_NJ; ++j) { e->motor_on[j] = 0; e->motor_q[j] = e->q[j]; e->motor_kp[j] = (real)c->kp; e->motor_kd[j] = (real)c->kd; e->vmax_cmd[j] = (real)a->v_max[j]; }
  arm_reset_targets(e);
  e->gripper_ready_time = R(0.0);
  e->arm_enabled = 1;
  arm_update_kinematics(w, e);
  if (c->open_gripper_when_reset) robot_grip(w, e, R(0.0));
  real off[RV_NLIMB];
  for (int j = 0; j < RV_NLIMB; ++j) off[j] = (real)c->offstage_positions[j];
  robot_move_to_joint_positions(w, e, off);
  if (c->env_type == RV_ENV_GRASP) {
    /* Grasp4DofEnv._reset_robot (grasp_4dof_env.py:206-211): robot.reset(OFFSTAGE) = move, then grip(0)
     * whose finger target replaces the limb target (one JointTarget per body) */
    robot_move_to_joint_positions(w, e, off);
    robot_grip(w, e, R(0.0));
  }
  e->has_prev = 0;
  compute_obs(e);
  memcpy(e->prev_obs_pos, e->obs_pos, sizeof(e->obs_pos));
}

/* ------------------------------------------------------------ C API ------ */
orc_world* orc_create(const rv_config* cfg, const rv_scene* scene) {
  orc_world* w = (orc_world*)calloc(1, sizeof(orc_world));
  w->cfg = *cfg; w->scene = *scene; w->n = cfg->n_envs;
  w->env = (orc_env*)calloc((size_t)w->n, sizeof(orc_env));
  for (int i = 0; i < w->n; ++i) {
    for (int b = 0; b < RV_MAXB; ++b) w->env[i].body[b].q[3] = R(1.0);
    for (int f = 0; f < RV_NFRAME; ++f) w->env[i].fquat[f][3] = R(1.0);
    w->env[i].done = 1; /* RobotEnv.__init__: self._done = True (robot_env.py:66) */
    w->env[i].mu_finger = (real)cfg->arm_friction; w->env[i].mu_table = (real)cfg->table_friction;
    camera_reset(cfg, &w->env[i], 0, 0);
  }
  return w;
}
void orc_destroy(orc_world* w) { if (w) { free(w->env); free(w); } }
int orc_is_double(void) { return (int)(sizeof(real) == 8); }

static void stats_begin(orc_world* w) {
  memset(&w->stats, 0, sizeof(w->stats));
  for (int i = 0; i < w->n; ++i) { orc_env* e = &w->env[i]; e->l_unsafe = e->l_ineffective = e->l_useful = e->l_episodes = e->l_successes = 0; }
}
static void stats_env(orc_world* w, const orc_env* e) {
  w->stats.unsafe += e->l_unsafe; w->stats.ineffective += e->l_ineffective; w->stats.useful += e->l_useful;
  w->stats.episodes_done += e->l_episodes; w->stats.successes += e->l_successes;
  w->stats.substeps += e->substeps_last;
  w->stats.awake_substeps += e->awake_last;
  if (e->substeps_last > w->stats.max_substeps) w->stats.max_substeps = e->substeps_last;
}

void orc_reset(orc_world* w, const uint8_t* mask) {
  stats_begin(w);
#pragma omp parallel for schedule(dynamic)
  for (int i = 0; i < w->n; ++i) {
    if (mask && !mask[i]) { w->env[i].substeps_last = 0; w->env[i].awake_last = 0; w->env[i].pairs_last = 0; continue; }
    env_reset(w, &w->env[i], w->cfg.env_id_offset + i);
  }
  for (int i = 0; i < w->n; ++i) stats_env(w, &w->env[i]);
}
void orc_set_actions(orc_world* w, const float* actions) {
  int G = w->cfg.num_goal_steps > 0 ? w->cfg.num_goal_steps : 1;
  for (int i = 0; i < w->n; ++i)
    for (int g = 0; g < G; ++g)
      for (int k = 0; k < 4; ++k) w->env[i].action[g][k] = (real)actions[(i * G + g) * 4 + k];
}
void orc_step_macro(orc_world* w) {
  stats_begin(w);
#pragma omp parallel for schedule(dynamic)
  for (int i = 0; i < w->n; ++i) {
    orc_env* e = &w->env[i];
    e->substeps_last = 0; e->awake_last = 0; e->pairs_last = 0;
    if (e->done) continue;
    step_env(w, e);
  }
  for (int i = 0; i < w->n; ++i) {
    const orc_env* e = &w->env[i];
    stats_env(w, e);
    if (e->substeps_last > 0) w->stats.env_steps++;
  }
}
/* generate_episode's inner loop with RandomPolicy (see rv_rollout in rovat.h) */
static void rollout_impl(orc_world* w, int n_steps, const int32_t* counts, int first_index, int auto_reset);
void orc_rollout(orc_world* w, int n_steps, int first_index, int auto_reset) { rollout_impl(w, n_steps, NULL, first_index, auto_reset); }
/* checker for rv_rollout_async: env i takes counts[i] steps (auto-reset) */
void orc_rollout_counts(orc_world* w, const int32_t* counts, int first_index) { rollout_impl(w, 0, counts, first_index, 1); }
static void rollout_impl(orc_world* w, int n_steps_all, const int32_t* counts, int first_index, int auto_reset) {
  stats_begin(w);
  int* stepped = (int*)calloc((size_t)w->n, sizeof(int));
  int* succ = (int*)calloc((size_t)w->n, sizeof(int));
#pragma omp parallel for schedule(dynamic)
  for (int i = 0; i < w->n; ++i) {
    orc_env* e = &w->env[i];
    int gid = w->cfg.env_id_offset + i;
    int sub = 0, aw = 0, pr = 0;
    e->substeps_last = 0; e->awake_last = 0; e->pairs_last = 0;
    const int n_steps = counts ? counts[i] : n_steps_all;
    for (int k = 0; k < n_steps; ++k) {
      if (e->done) {
        if (!auto_reset) break;
        env_reset(w, e, gid);
        sub += e->substeps_last; aw += e->awake_last; pr += e->pairs_last;
      }
      random_action(&w->cfg, gid, first_index + k, &e->action[0][0]);
      step_env(w, e);
      sub += e->substeps_last; aw += e->awake_last; pr += e->pairs_last;
      stepped[i]++;
      if (e->done && e->last_reward >= (real)w->cfg.success_thresh) succ[i]++;
    }
    e->substeps_last = sub; e->awake_last = aw; e->pairs_last = pr;
  }
  for (int i = 0; i < w->n; ++i) { stats_env(w, &w->env[i]); w->stats.env_steps += stepped[i]; }
  free(stepped); free(succ);
}
void orc_step_sub(orc_world* w, int n) {
  stats_begin(w);
#pragma omp parallel for schedule(dynamic)
  for (int i = 0; i < w->n; ++i) {
    w->env[i].substeps_last = 0; w->env[i].awake_last = 0; w->env[i].pairs_last = 0;
    for (int k = 0; k < n; ++k) sim_substep(w, &w->env[i]);
  }
  for (int i = 0; i < w->n; ++i) stats_env(w, &w->env[i]);
}
void orc_wait_until_stable(orc_world* w, float lin, float ang, int check_after, int min_stable, int max_steps) {
  stats_begin(w);
#pragma omp parallel for schedule(dynamic)
  for (int i = 0; i < w->n; ++i) {
    orc_env* e = &w->env[i];
    e->substeps_last = 0; e->awake_last = 0; e->pairs_last = 0;
    unsigned mask = 0;
    for (int b = 0; b < RV_MAXB; ++b) if (e->bp[b].active) mask |= 1u << b;
    wait_until_stable(w, e, mask, (real)lin, (real)ang, check_after, min_stable, max_steps);
  }
  for (int i = 0; i < w->n; ++i) stats_env(w, &w->env[i]);
}
void orc_policy_random(orc_world* w, int macro_index, float* actions) {
  int G = w->cfg.num_goal_steps > 0 ? w->cfg.num_goal_steps : 1;
  for (int i = 0; i < w->n; ++i) {
    real a[RV_MAXG * 4];
    random_action(&w->cfg, w->cfg.env_id_offset + i, macro_index, a);
    for (int k = 0; k < G * 4; ++k) actions[i * G * 4 + k] = (float)a[k];
  }
}

/* HeuristicPushSampler._sample (heuristic_push_sampler.py:66-123): candidates
 * are drawn from Philox keyed by (env, episode, step, attempt) so that the
 * lowest successful attempt index wins regardless of evaluation order. */
void orc_policy_heuristic(orc_world* w, int max_attempts, float* actions) {
  const rv_config* c = &w->cfg;
  int G = c->num_goal_steps > 0 ? c->num_goal_steps : 1;
  for (int i = 0; i < w->n; ++i) {
    const orc_env* e = &w->env[i];
    int nb = 0;
    for (int b = 0; b < RV_MAXB; ++b) nb += body_movable(e, b);
    if (nb == 0) nb = 1;
    /* counters come from the observation = the env.attributes snapshot (push_policy.py:46-49) */
    const int n_ep = e->obs_num_episodes, n_st = e->obs_num_steps;
    int body_id = n_ep % nb;
    real base = (real)n_ep * R(42.0);
    base = base - R(2.0) * ORC_PI * R(floor)(base / (R(2.0) * ORC_PI));
    real lo0 = (real)c->cspace_low[0], hi0 = (real)c->cspace_high[0], lo1 = (real)c->cspace_low[1], hi1 = (real)c->cspace_high[1];
    real start[2] = {0, 0}, motion[2] = {0, 0};
    for (int att = 0; att < max_attempts; ++att) {
      orc_rng g; rng_init(&g, c->seed_lo, c->seed_hi, (uint32_t)(c->env_id_offset + i), STREAM_HEUR,
                          (uint32_t)(n_ep * 64 + n_st));
      g.ctr[0] = (uint32_t)att * 2u;
      start[0] = rng_uniform(&g, R(-1.0), R(1.0)); start[1] = rng_uniform(&g, R(-1.0), R(1.0));
      real ang = base + rng_uniform(&g, R(-0.25) * ORC_PI, R(0.25) * ORC_PI);
      real s, co; rsincos(ang, &s, &co);
      motion[0] = rclamp(co + rng_uniform(&g, R(-0.3), R(0.3)), R(-1.0), R(1.0));
      motion[1] = rclamp(s + rng_uniform(&g, R(-0.3), R(0.3)), R(-1.0), R(1.0));
      real x = start[0] * (R(0.5) * (hi0 - lo0)) + R(0.5) * (hi0 + lo0);
      real y = start[1] * (R(0.5) * (hi1 - lo1)) + R(0.5) * (hi1 + lo1);
      real ex = rclamp(x + motion[0] * (real)c->translation_x, lo0, hi0);
      real ey = rclamp(y + motion[1] * (real)c->translation_y, lo1, hi1);
      int safe = 1;
      for (int b = 0; b < nb; ++b) {
        real dx = e->obs_pos[b][0] - x, dy = e->obs_pos[b][1] - y;
        if (!(rsqrt_(dx * dx + dy * dy) > R(0.05))) safe = 0;
      }
      if (!safe) continue;
      real dx1 = e->obs_pos[body_id][0] - x, dy1 = e->obs_pos[body_id][1] - y;
      real dx2 = e->obs_pos[body_id][0] - ex, dy2 = e->obs_pos[body_id][1] - ey;
      int clear = rsqrt_(dx1 * dx1 + dy1 * dy1) >= R(0.01) && rsqrt_(dx2 * dx2 + dy2 * dy2) >= R(0.01);
      if (!clear) break; /* effective and safe */
    }
    for (int g2 = 0; g2 < G; ++g2) {
      actions[(i * G + g2) * 4 + 0] = (float)start[0]; actions[(i * G + g2) * 4 + 1] = (float)start[1];
      actions[(i * G + g2) * 4 + 2] = (float)motion[0]; actions[(i * G + g2) * 4 + 3] = (float)motion[1];
    }
  }
}

void orc_get_body_state(orc_world* w, double* out) {
  for (int i = 0; i < w->n; ++i)
    for (int b = 0; b < RV_MAXB; ++b) {
      double* o = out + ((size_t)i * RV_MAXB + b) * 13;
      const orc_body* B = &w->env[i].body[b];
      for (int k = 0; k < 3; ++k) { o[k] = B->p[k]; o[7 + k] = B->v[k]; o[10 + k] = B->w[k]; }
      for (int k = 0; k < 4; ++k) o[3 + k] = B->q[k];
    }
}
void orc_set_body_state(orc_world* w, const double* in) {
  for (int i = 0; i < w->n; ++i)
    for (int b = 0; b < RV_MAXB; ++b) {
      const double* o = in + ((size_t)i * RV_MAXB + b) * 13;
      orc_body* B = &w->env[i].body[b];
      for (int k = 0; k < 3; ++k) { B->p[k] = (real)o[k]; B->v[k] = (real)o[7 + k]; B->w[k] = (real)o[10 + k]; }
      for (int k = 0; k < 4; ++k) B->q[k] = (real)o[3 + k];
      w->env[i].man[TIDX(b)].n = 0; w->env[i].man[AIDX(b)].n = 0;
      w->env[i].bp[b].asleep = 0; w->env[i].bp[b].sleep_count = 0; w->env[i].bp[b].deact_count = 0; w->env[i].bp[b].still_count = 0; w->env[i].bp[b].undisturbed = 0;
    }
  for (int i = 0; i < w->n; ++i) for (int k = 0; k < RV_NBB; ++k) w->env[i].man[BBIDX(k)].n = 0;
}
void orc_get_body_params(orc_world* w, double* out) {
  for (int i = 0; i < w->n; ++i)
    for (int b = 0; b < RV_MAXB; ++b) {
      double* o = out + ((size_t)i * RV_MAXB + b) * 8;
      const orc_bparam* p = &w->env[i].bp[b];
      o[0] = p->active; o[1] = p->shape; o[2] = p->scale; o[3] = p->mass; o[4] = p->friction; o[5] = p->frozen; o[6] = w->env[i].table_z; o[7] = p->asleep;
    }
}
void orc_set_body_params(orc_world* w, const double* in) {
  for (int i = 0; i < w->n; ++i) {
    orc_env* e = &w->env[i];
    for (int b = 0; b < RV_MAXB; ++b) {
      const double* o = in + ((size_t)i * RV_MAXB + b) * 8;
      orc_bparam* p = &e->bp[b];
      p->active = (int)o[0]; p->shape = (int)o[1]; p->scale = (real)o[2]; p->friction = (real)o[4]; p->frozen = (int)o[5]; p->asleep = 0; p->sleep_count = 0; p->deact_count = 0; p->still_count = 0; p->undisturbed = 0;
      if (b == 0) { e->table_z = (real)o[6]; table_prepare(w, e); }
      if (p->active) body_set_mass(w, e, b, (real)o[3]);
    }
    e->n_bodies = 0;
    for (int b = 0; b < RV_MAXB; ++b) e->n_bodies += e->bp[b].active;
  }
}
void orc_get_joint_state(orc_world* w, double* out) {
  for (int i = 0; i < w->n; ++i)
    for (int j = 0; j < RV_NJ; ++j) { out[((size_t)i * RV_NJ + j) * 2] = w->env[i].q[j]; out[((size_t)i * RV_NJ + j) * 2 + 1] = w->env[i].qd[j]; }
}
void orc_set_joint_state(orc_world* w, const double* in) {
  for (int i = 0; i < w->n; ++i) {
    orc_env* e = &w->env[i];
    for (int j = 0; j < RV_NJ; ++j) { e->q[j] = (real)in[((size_t)i * RV_NJ + j) * 2]; e->qd[j] = (real)in[((size_t)i * RV_NJ + j) * 2 + 1]; e->motor_q[j] = e->q[j]; }
    if (!e->arm_enabled) {
      const rv_arm* a = &w->scene.arm;
      for (int j = 0; j < RV_NJ; ++j) { e->motor_on[j] = 0; e->motor_kp[j] = (real)w->cfg.kp; e->motor_kd[j] = (real)w->cfg.kd; e->vmax_cmd[j] = (real)a->v_max[j]; }
      arm_reset_targets(e); e->gripper_ready_time = R(0.0); e->arm_enabled = 1;
    }
    arm_update_kinematics(w, e);
  }
}
void orc_get_link_poses(orc_world* w, double* out) {
  for (int i = 0; i < w->n; ++i)
    for (int f = 0; f < RV_NFRAME; ++f) {
      double* o = out + ((size_t)i * RV_NFRAME + f) * 7;
      for (int k = 0; k < 3; ++k) o[k] = w->env[i].fpos[f][k];
      for (int k = 0; k < 4; ++k) o[3 + k] = w->env[i].fquat[f][k];
    }
}
void orc_get_env_counters(orc_world* w, int32_t* out) {
  for (int i = 0; i < w->n; ++i) {
    const orc_env* e = &w->env[i]; int32_t* o = out + (size_t)i * RV_NCOUNTERS;
    o[8] = e->awake_last; o[9] = e->pairs_last;
    o[0] = e->sim_steps; o[1] = e->num_steps; o[2] = e->num_episodes; o[3] = e->phase; o[4] = e->done; o[5] = e->is_safe; o[6] = e->is_effective; o[7] = e->substeps_last;
  }
}
void orc_set_joint_targets(orc_world* w, const float* q) {
  for (int i = 0; i < w->n; ++i) {
    real pos[RV_NLIMB];
    for (int j = 0; j < RV_NLIMB; ++j) pos[j] = (real)q[i * RV_NLIMB + j];
    robot_move_to_joint_positions(w, &w->env[i], pos);
  }
}
void orc_set_link_target(orc_world* w, const float* pose) {
  for (int i = 0; i < w->n; ++i) {
    real p[7];
    for (int k = 0; k < 7; ++k) p[k] = (real)pose[i * 7 + k];
    robot_move_to_gripper_pose(w, &w->env[i], p);
  }
}
void orc_compute_ik(orc_world* w, const float* pose, double* q) {
  for (int i = 0; i < w->n; ++i) {
    real p[7], out[RV_NLIMB];
    for (int k = 0; k < 7; ++k) p[k] = (real)pose[i * 7 + k];
    arm_ik(w, w->env[i].q, p, out);
    for (int j = 0; j < RV_NLIMB; ++j) q[i * RV_NLIMB + j] = out[j];
  }
}
/* ---- hooks for tests/golden/gen_control_golden.py: the reference's unmodified
 * Simulator / SawyerSim / ControllableBody drive THIS arm model through a
 * Physics plugin, so that control_update() above can be pinned against them. */
void orc_set_external_control(orc_world* w, int on) { w->ext_control = on; }
void orc_set_pose_f32(orc_world* w, int on) { w->pose_f32 = on; }
/* position_control_array (bullet_physics.py:1061-1104) */
void orc_motor_targets(orc_world* w, int n, const int32_t* idx, const double* pos) {
  for (int i = 0; i < w->n; ++i) {
    orc_env* e = &w->env[i];
    for (int j = 0; j < RV_NLIMB; ++j) e->vmax_cmd[j] = (real)w->cfg.limb_max_velocity_ratio * (real)w->scene.arm.v_max[j];
    for (int k = 0; k < n; ++k) {
      int j = idx[k];
      e->motor_on[j] = 1; e->motor_q[j] = (real)pos[k];
      e->motor_kp[j] = (real)w->cfg.kp; e->motor_kd[j] = (real)w->cfg.kd;
    }
  }
}
/* rv_set_max_joint_velocities (controllable_body.py:357-372) */
void orc_set_max_joint_velocities(orc_world* w, const float* v) {
  for (int i = 0; i < w->n; ++i) for (int j = 0; j < RV_NLIMB; ++j) w->env[i].vmax_cmd[j] = (real)v[(size_t)i * RV_NLIMB + j];
}
void orc_compute_ik_seeded(orc_world* w, const double* seed, const double* pose, double* q) {
  real p[7], sd[RV_NLIMB], out[RV_NLIMB];
  for (int k = 0; k < 7; ++k) p[k] = (real)pose[k];
  for (int j = 0; j < RV_NLIMB; ++j) sd[j] = seed ? (real)seed[j] : w->env[0].q[j];
  arm_ik(w, sd, p, out);
  for (int j = 0; j < RV_NLIMB; ++j) q[j] = out[j];
}
void orc_set_link_path(orc_world* w, int n_poses, const float* poses) {
  for (int i = 0; i < w->n; ++i) {
    real wps[RV_MAXQ][7];
    for (int q = 0; q < n_poses; ++q) for (int k = 0; k < 7; ++k) wps[q][k] = (real)poses[q * 7 + k];
    robot_move_along_gripper_path(w, &w->env[i], wps, n_poses);
  }
}
/* Link.set_dynamics / Body.set_dynamics lateral friction of the finger tips and the table
 * (grasp_4dof_env.py:262-293); negative = leave unchanged */
/* per-env paths [N][n_poses][7] (rv_set_link_path) */
void orc_set_link_paths(orc_world* w, int n_poses, const float* poses) {
  for (int i = 0; i < w->n; ++i) {
    real wps[RV_MAXQ][7];
    for (int q = 0; q < n_poses; ++q) for (int k = 0; k < 7; ++k) wps[q][k] = (real)poses[((size_t)i * n_poses + q) * 7 + k];
    robot_move_along_gripper_path(w, &w->env[i], wps, n_poses);
  }
}
/* rv_get_robot_ready: [N][2] = (is_limb_ready, is_gripper_ready) */
void orc_get_camera(orc_world* w, double* out) {
  for (int i = 0; i < w->n; ++i) {
    const orc_env* e = &w->env[i]; double* o = out + (size_t)i * 17;
    for (int k = 0; k < 5; ++k) o[k] = (double)e->cam_intrinsics[k];
    for (int k = 0; k < 9; ++k) o[5 + k] = (double)e->cam_rotation[k];
    for (int k = 0; k < 3; ++k) o[14 + k] = (double)e->cam_translation[k];
  }
}
void orc_robot_ready(orc_world* w, uint8_t* out) {
  for (int i = 0; i < w->n; ++i) { out[2 * i] = (uint8_t)arm_is_ready_limb(w, &w->env[i]); out[2 * i + 1] = (uint8_t)robot_is_gripper_ready(w, &w->env[i]); }
}
void orc_set_friction(orc_world* w, double mu_finger, double mu_table) {
  for (int i = 0; i < w->n; ++i) {
    if (mu_finger >= 0.0) w->env[i].mu_finger = (real)mu_finger;
    if (mu_table >= 0.0) w->env[i].mu_table = (real)mu_table;
  }
}
/* move_to_gripper_pose(..., timeout=t): LinkTarget.set stop_time = start_time + timeout */
void orc_set_link_timeout(orc_world* w, double timeout) {
  for (int i = 0; i < w->n; ++i) w->env[i].lt.stop_t = w->env[i].lt.start_t + (real)timeout;
}
/* Simulator.add_constraint / Constraint.pose setter (bullet_physics.py:748-957) for body `body` of every env:
 * frame7 = joint frame in the body frame (NULL: identity), target7 = the world frame it is tied to,
 * max_force < 0: the constraint is removed */
void orc_set_constraint_ex(orc_world* w, int body, int child, int joint_type, const double* frame7, const double* target7, double max_force) {
  for (int i = 0; i < w->n; ++i) {
    orc_bparam* P = &w->env[i].bp[body];
    if (max_force < 0.0) P->con_on = 0;
    else {
      P->con_on = (joint_type >= 2 && joint_type <= 4 ? joint_type : 1) | ((child + 1) << 4); P->con_fmax = (real)max_force;
      for (int k = 0; k < 3; ++k) { P->con_lpos[k] = frame7 ? (real)frame7[k] : R(0.0); P->con_tpos[k] = (real)target7[k]; }
      for (int k = 0; k < 4; ++k) { P->con_lquat[k] = frame7 ? (real)frame7[3 + k] : (k == 3 ? R(1.0) : R(0.0)); P->con_tquat[k] = (real)target7[3 + k]; }
    }
    P->asleep = 0; P->sleep_count = 0; P->deact_count = 0; P->still_count = 0; P->undisturbed = 0;
    if (child >= 0 && child < RV_MAXB) { orc_bparam* C = &w->env[i].bp[child]; C->asleep = 0; C->sleep_count = 0; C->deact_count = 0; C->still_count = 0; C->undisturbed = 0; }
  }
}
void orc_set_constraint(orc_world* w, int body, const double* frame7, const double* target7, double max_force) {
  orc_set_constraint_ex(w, body, -1, 1, frame7, target7, max_force);
}
void orc_grip(orc_world* w, float value) { for (int i = 0; i < w->n; ++i) robot_grip(w, &w->env[i], (real)value); }
int orc_is_limb_ready(orc_world* w, int env) { return arm_is_ready_limb(w, &w->env[env]); }
int orc_is_gripper_ready(orc_world* w, int env) { return robot_is_gripper_ready(w, &w->env[env]); }
double orc_time(orc_world* w, int env) { return (double)sim_time(w, &w->env[env]); }

void orc_query_contacts(orc_world* w, uint8_t* out) {
  for (int i = 0; i < w->n; ++i) {
    const orc_env* e = &w->env[i]; uint8_t* o = out + (size_t)i * (2 + RV_MAXB);
    o[0] = (uint8_t)e->flag_arm_table; o[1] = (uint8_t)arm_touches_movables(e);
    for (int b = 0; b < RV_MAXB; ++b) o[2 + b] = (uint8_t)(e->bp[b].active && e->flag_arm_body[b]);
  }
}
/* diagnostic: the points of one manifold, out[4][13] = la, lb, nrm, dist, ln, lt1, lt2; returns n */
int orc_get_manifold(orc_world* w, int env, int mi, double* out) {
  const orc_manifold* m = &w->env[env].man[mi];
  for (int i = 0; i < 4; ++i) {
    double* o = out + 13 * i;
    for (int k = 0; k < 3; ++k) { o[k] = m->la[i][k]; o[3 + k] = m->lb[i][k]; o[6 + k] = m->nrm[i][k]; }
    o[9] = m->dist[i]; o[10] = m->ln[i]; o[11] = m->lt1[i]; o[12] = m->lt2[i];
  }
  return m->n;
}
void orc_get_manifold_counts(orc_world* w, int32_t* out) {
  for (int i = 0; i < w->n; ++i) for (int m = 0; m < RV_NMAN; ++m) out[(size_t)i * RV_NMAN + m] = w->env[i].man[m].n;
}
/* PoseObs (pose_obs.py:53-73, all four modalities) + attribute observations
 * (attribute_obs.py:16-115).  attrs: [N][5] num_episodes, num_steps, layout_id, is_safe, is_effective */
void orc_observe(orc_world* w, double* position, double* body_mask, int64_t* attrs,
                 double* pose, double* pose2d, double* yaw_cossin) {
  for (int i = 0; i < w->n; ++i) {
    const orc_env* e = &w->env[i];
    for (int b = 0; b < RV_MAXB; ++b) {
      const size_t ib = (size_t)i * RV_MAXB + b;
      for (int k = 0; k < 3; ++k) position[ib * 3 + k] = e->obs_pos[b][k];
      body_mask[ib] = body_movable(e, b);
      real eu[3] = {R(0.0), R(0.0), R(0.0)};
      if (body_movable(e, b)) quat_to_euler(e->body[b].q, eu);
      if (pose) for (int k = 0; k < 3; ++k) { pose[ib * 6 + k] = e->obs_pos[b][k]; pose[ib * 6 + 3 + k] = eu[k]; }
      if (pose2d) { pose2d[ib * 3] = e->obs_pos[b][0]; pose2d[ib * 3 + 1] = e->obs_pos[b][1]; pose2d[ib * 3 + 2] = eu[2]; }
      if (yaw_cossin) {
        real sn = R(0.0), cs = R(0.0);
        if (body_movable(e, b)) rsincos(eu[2], &sn, &cs);
        yaw_cossin[ib * 2] = cs; yaw_cossin[ib * 2 + 1] = sn;
      }
    }
    if (attrs) {
      attrs[i * 5 + 0] = e->obs_num_episodes; attrs[i * 5 + 1] = e->obs_num_steps; attrs[i * 5 + 2] = w->cfg.layout_id;
      attrs[i * 5 + 3] = e->is_safe; attrs[i * 5 + 4] = e->is_effective;
    }
  }
}
/* -------------------------------------------------- SegmentedPointCloudObs --
 * Restatement of the chain BulletCamera._frames (bullet_camera.py:188-235: depth +
 * segmentation render, depth linearised to eye z) -> Camera.deproject_depth_image
 * (camera.py:213-244) -> convert_segment_ids / group_by_labels
 * (point_cloud_utils.py:110-157; downsample :23-39: with replacement iff the body has
 * fewer than num_points pixels, zeros if it has none).  The render is a ray cast of the
 * convex hulls (face planes of rv_shape) and the table; the arm is off-stage when the
 * reference takes the observation and is not rendered.  Sampling: the num_points smallest
 * per-pixel Philox keys (a uniformly random subset) or num_points draws with replacement. */
#define STREAM_PC 7u
static void pc_pixel_dir_cam(const orc_env* c, real u, real v, real* d) {
  const real fx = (real)c->cam_intrinsics[0], fy = (real)c->cam_intrinsics[1], cx = (real)c->cam_intrinsics[2], cy = (real)c->cam_intrinsics[3], sk = (real)c->cam_intrinsics[4];
  real y = (v - cy) / fy;
  real x = (u - cx - sk * y) / fx;
  d[0] = x; d[1] = y; d[2] = R(1.0);
}
static void pc_cam_rot(const orc_env* c, real* Rm) { for (int i = 0; i < 9; ++i) Rm[i] = (real)c->cam_rotation[i]; }
static void pc_cam_position(const orc_env* c, real* o) {
  real Rm[9], t[3] = {(real)c->cam_translation[0], (real)c->cam_translation[1], (real)c->cam_translation[2]}, p[3];
  pc_cam_rot(c, Rm); m3tmulv(p, Rm, t);
  o[0] = -p[0]; o[1] = -p[1]; o[2] = -p[2];
}
static int pc_ray_hull(const float (*planes)[4], int n, real sc, real margin, const real* o, const real* d, real* t_hit, int* plane_hit) {
  real t0 = R(0.0), t1 = R(1e30); int ip = -1;
  for (int i = 0; i < n; ++i) {
    real nn[3] = {(real)planes[i][0], (real)planes[i][1], (real)planes[i][2]};
    real off = (real)planes[i][3] * sc + margin;
    real den = v3dot(nn, d);
    real num = off - v3dot(nn, o);
    if (den < R(0.0)) { real t = num / den; if (t > t0) { t0 = t; ip = i; } }
    else if (den > R(0.0)) { real t = num / den; if (t < t1) t1 = t; }
    else if (num < R(0.0)) return 0;
  }
  if (t0 > t1) return 0;
  *t_hit = t0;
  if (plane_hit) *plane_hit = ip;
  return 1;
}
static int pc_ray_table(const rv_config* c, real table_z, const real* o, const real* d, real* t_hit, int* axis_hit) {
  const real lo[3] = {(real)c->table_center[0] - (real)c->table_half[0], (real)c->table_center[1] - (real)c->table_half[1], table_z - (real)c->table_thickness};
  const real hi[3] = {(real)c->table_center[0] + (real)c->table_half[0], (real)c->table_center[1] + (real)c->table_half[1], table_z};
  real t0 = R(0.0), t1 = R(1e30); int ax = -1;
  for (int k = 0; k < 3; ++k) {
    if (d[k] != R(0.0)) {
      real a = (lo[k] - o[k]) / d[k], b = (hi[k] - o[k]) / d[k];
      real tn = a < b ? a : b, tf = a < b ? b : a;
      if (tn > t0) { t0 = tn; ax = a < b ? k : 3 + k; }
      if (tf < t1) t1 = tf;
    } else if (o[k] < lo[k] || o[k] > hi[k]) return 0;
  }
  if (t0 > t1) return 0;
  *t_hit = t0;
  if (axis_hit) *axis_hit = ax;
  return 1;
}
/* entry depth into the box |x_k| <= h_k (ray in the box frame) */
static int pc_ray_box(const real* h, const real* o, const real* d, real* t_hit, int* axis_hit) {
  real t0 = R(0.0), t1 = R(1e30); int ax = -1;
  for (int k = 0; k < 3; ++k) {
    if (d[k] != R(0.0)) {
      real a = (-h[k] - o[k]) / d[k], b = (h[k] - o[k]) / d[k];
      real tn = a < b ? a : b, tf = a < b ? b : a;
      if (tn > t0) { t0 = tn; ax = a < b ? k : 3 + k; }
      if (tf < t1) t1 = tf;
    } else if (o[k] < -h[k] || o[k] > h[k]) return 0;
  }
  if (t0 > t1) return 0;
  *t_hit = t0; *axis_hit = ax;
  return 1;
}
/* nearest hit of a pixel ray: body index, RV_MAXB = table, RV_MAXB + 1 = the arm (its link collider boxes), -1 = nothing */
static int pc_render_pixel_n(const orc_world* w, const orc_env* e, real rot[][9], const real* cam_o, const real* dw, real* depth, real* normal) {
  const rv_config* c = &w->cfg;
  real best = R(1e30); int who = -1;
  real nb[3] = {R(0.0), R(0.0), R(0.0)};
  for (int b = 0; b < RV_MAXB; ++b) {
    if (!e->bp[b].active) continue;
    const rv_shape* sh = &w->scene.shapes[e->bp[b].shape];
    real rel[3]; v3sub(rel, cam_o, e->body[b].p);
    real r = (real)sh->radius * e->bp[b].scale + (real)c->margin;
    real dd = v3dot(dw, dw), rd = v3dot(rel, dw);
    real perp2 = v3dot(rel, rel) - rd * rd / dd;
    if (perp2 > r * r) continue;
    real ol[3], dl[3];
    m3tmulv(ol, rot[b], rel); m3tmulv(dl, rot[b], dw);
    for (int h = 0; h < sh->n_hulls; ++h) {
      real t; int ip = -1;
      if (pc_ray_hull(sh->planes[h], sh->n_planes[h], e->bp[b].scale, (real)c->margin, ol, dl, &t, &ip) && t < best) {
        best = t; who = b;
        if (normal && ip >= 0) { real pn[3] = {(real)sh->planes[h][ip][0], (real)sh->planes[h][ip][1], (real)sh->planes[h][ip][2]}; m3mulv(nb, rot[b], pn); }
      }
    }
  }
  if (e->arm_enabled) {
    const rv_arm* arm = &w->scene.arm;
    for (int col = 0; col < RV_NCOL; ++col) {
      const int f = arm->col_frame[col];
      const real hh[3] = {(real)arm->col_half[col][0] + (real)c->margin, (real)arm->col_half[col][1] + (real)c->margin, (real)arm->col_half[col][2] + (real)c->margin};
      real Rm[9], cc[3] = {(real)arm->col_center[col][0], (real)arm->col_center[col][1], (real)arm->col_center[col][2]}, t3[3], cw[3], rel[3];
      qmat(Rm, e->fquat[f]);
      m3mulv(t3, Rm, cc); v3add(cw, e->fpos[f], t3);
      /* (the device keeps the box centre in a float snapshot) */
      cw[0] = (real)(float)cw[0]; cw[1] = (real)(float)cw[1]; cw[2] = (real)(float)cw[2];
      v3sub(rel, cam_o, cw);
      const real r2 = hh[0] * hh[0] + hh[1] * hh[1] + hh[2] * hh[2];
      const real dd = v3dot(dw, dw), rd = v3dot(rel, dw);
      const real perp2 = v3dot(rel, rel) - rd * rd / dd;
      if (perp2 > r2) continue;
      real ol[3], dl[3], t; int ia = -1;
      m3tmulv(ol, Rm, rel); m3tmulv(dl, Rm, dw);
      if (pc_ray_box(hh, ol, dl, &t, &ia) && t < best) {
        best = t; who = RV_MAXB + 1;
        if (normal && ia >= 0) { const real sg = ia < 3 ? R(-1.0) : R(1.0); const int k = ia % 3; real an[3] = {k == 0 ? sg : R(0.0), k == 1 ? sg : R(0.0), k == 2 ? sg : R(0.0)}; m3mulv(nb, Rm, an); }
      }
    }
  }
  real tt; int ax = -1;
  if (pc_ray_table(c, e->table_z, cam_o, dw, &tt, &ax) && tt < best) {
    best = tt; who = RV_MAXB;
    if (normal && ax >= 0) { const real sg = ax < 3 ? R(-1.0) : R(1.0); const int k = ax % 3; nb[0] = k == 0 ? sg : R(0.0); nb[1] = k == 1 ? sg : R(0.0); nb[2] = k == 2 ? sg : R(0.0); }
  }
  *depth = best;
  if (normal) { normal[0] = nb[0]; normal[1] = nb[1]; normal[2] = nb[2]; }
  return who;
}
static int pc_render_pixel(const orc_world* w, const orc_env* e, real rot[][9], const real* cam_o, const real* dw, real* depth) {
  return pc_render_pixel_n(w, e, rot, cam_o, dw, depth, (real*)0);
}
static void pc_deproject(const orc_env* c, const real* cam_o, real u, real v, real z, real* out) {
  real d[3], pc[3], Rm[9], pw[3];
  pc_pixel_dir_cam(c, u, v, d);
  pc[0] = d[0] * z; pc[1] = d[1] * z; pc[2] = d[2] * z;
  pc_cam_rot(c, Rm); m3tmulv(pw, Rm, pc);
  v3add(out, cam_o, pw);
}
static int pc_crop_ok(const rv_config* c, const real* p) {
  if (!c->use_crop) return 1;
  return p[0] >= (real)c->crop_min[0] && p[1] >= (real)c->crop_min[1] && p[2] >= (real)c->crop_min[2] &&
         p[0] <= (real)c->crop_max[0] && p[1] <= (real)c->crop_max[1] && p[2] <= (real)c->crop_max[2];
}
static uint32_t pc_hash(const rv_config* c, uint32_t gid, uint32_t rng_arg, uint32_t ctr) {
  uint32_t cc[4] = {ctr, rng_arg, gid, STREAM_PC}, kk[2] = {c->seed_lo, c->seed_hi}, out[4];
  philox4x32_10(cc, kk, out);
  return out[0];
}
static void pc_body_rots(const orc_env* e, real rot[][9]) { for (int b = 0; b < RV_MAXB; ++b) qmat(rot[b], e->body[b].q); }
/* the full depth / segmentation image of one env (tests: input of the reference-shaped
 * deproject -> convert_segment_ids -> group_by_labels pipeline).  segmask: body index,
 * RV_MAXB = table, RV_MAXB + 1 = arm, 255 = nothing; depth 0 where nothing is hit */
void orc_render(orc_world* w, int env, float* depth, uint8_t* segmask) {
  const rv_config* c = &w->cfg; const orc_env* e = &w->env[env];
  real rot[RV_MAXB][9], cam_o[3], Rm[9];
  pc_body_rots(e, rot); pc_cam_position(e, cam_o); pc_cam_rot(e, Rm);
  for (int v = 0; v < c->cam_height; ++v)
    for (int u = 0; u < c->cam_width; ++u) {
      real dc[3], dw[3], dep;
      pc_pixel_dir_cam(e, (real)u, (real)v, dc); m3tmulv(dw, Rm, dc);
      int who = pc_render_pixel(w, e, rot, cam_o, dw, &dep);
      if (who >= 0 && !(dep > (real)c->cam_near)) who = -1;
      depth[(size_t)v * c->cam_width + u] = who >= 0 ? (float)dep : 0.0f;
      segmask[(size_t)v * c->cam_width + u] = who >= 0 ? (uint8_t)who : (uint8_t)255;
    }
}
/* CameraObs 'rgb' (camera_obs.py:33-88): flat colours per body slot / table / background,
 * Lambert-shaded with the normal of the face hit, one fixed directional light; rgb: [H][W][3] */
void orc_render_rgb(orc_world* w, int env, uint8_t* rgb) {
  const rv_config* c = &w->cfg; const orc_env* e = &w->env[env];
  static const real base[RV_MAXB + 3][3] = {{230, 60, 60}, {60, 170, 230}, {250, 200, 40}, {90, 200, 110}, {150, 120, 90}, {185, 185, 195}, {30, 30, 30}};   /* bodies, table, arm, background */
  real rot[RV_MAXB][9], cam_o[3], Rm[9];
  pc_body_rots(e, rot); pc_cam_position(e, cam_o); pc_cam_rot(e, Rm);
  for (int v = 0; v < c->cam_height; ++v)
    for (int u = 0; u < c->cam_width; ++u) {
      real dc[3], dw[3], dep, n[3];
      pc_pixel_dir_cam(e, (real)u, (real)v, dc); m3tmulv(dw, Rm, dc);
      int who = pc_render_pixel_n(w, e, rot, cam_o, dw, &dep, n);
      if (who >= 0 && !(dep > (real)c->cam_near)) who = -1;
      const int idx = who < 0 ? RV_MAXB + 2 : who;
      real sh = R(1.0);
      if (who >= 0) {
        const real lam = n[0] * R(0.30151134) + n[1] * R(-0.30151134) + n[2] * R(0.90453403);
        sh = R(0.35) + R(0.65) * (lam > R(0.0) ? lam : R(0.0));
      }
      for (int k = 0; k < 3; ++k) rgb[((size_t)v * c->cam_width + u) * 3 + k] = (uint8_t)(int)(base[idx][k] * sh + R(0.5));
    }
}
static int cmp_u32(const void* a, const void* b) { uint32_t x = *(const uint32_t*)a, y = *(const uint32_t*)b; return x < y ? -1 : (x > y); }
/* out: [N][RV_MAXB][num_points][3] */
void orc_point_cloud(orc_world* w, float* out) {
  const rv_config* c = &w->cfg; const int P = c->num_points;
#pragma omp parallel for schedule(dynamic)
  for (int i = 0; i < w->n; ++i) {
    const orc_env* e = &w->env[i];
    const uint32_t gid = (uint32_t)(c->env_id_offset + i);
    const uint32_t rng_arg = (uint32_t)e->reset_count * 4096u + (uint32_t)e->num_steps;
    real rot[RV_MAXB][9], cam_o[3], Rm[9];
    pc_body_rots(e, rot); pc_cam_position(e, cam_o); pc_cam_rot(e, Rm);
    uint32_t* pix = (uint32_t*)malloc(sizeof(uint32_t) * RV_PC_MAXPIX * 3);
    uint32_t* key = pix + RV_PC_MAXPIX; uint32_t* sorted = key + RV_PC_MAXPIX;
    real* dep = (real*)malloc(sizeof(real) * RV_PC_MAXPIX);
    for (int b = 0; b < RV_MAXB; ++b) {
      float* o = out + ((size_t)i * RV_MAXB + b) * (size_t)P * 3;
      int n = 0;
      if (body_movable(e, b)) {      /* (the segmented cloud has the movable bodies; a static body is rendered -- it occludes -- but not sampled) */
        const rv_shape* sh = &w->scene.shapes[e->bp[b].shape];
        real mu = R(1e30), xu = R(-1e30), mv = R(1e30), xv = R(-1e30), mz = R(1e30);
        for (int h = 0; h < sh->n_hulls; ++h)
          for (int k = 0; k < sh->n_verts[h]; ++k) {
            const real sc = e->bp[b].scale;
            real l[3] = {(real)sh->verts[h][k][0] * sc, (real)sh->verts[h][k][1] * sc, (real)sh->verts[h][k][2] * sc}, rw[3], pw[3], pc[3];
            m3mulv(rw, rot[b], l); v3add(pw, e->body[b].p, rw);
            m3mulv(pc, Rm, pw); pc[0] += e->cam_translation[0]; pc[1] += e->cam_translation[1]; pc[2] += e->cam_translation[2];
            real z = pc[2];
            real u = (e->cam_intrinsics[0] * pc[0] + e->cam_intrinsics[4] * pc[1]) / pc[2] + e->cam_intrinsics[2];
            real v = e->cam_intrinsics[1] * pc[1] / pc[2] + e->cam_intrinsics[3];
            if (z < mz) mz = z;
            if (z > (real)c->cam_near) { if (u < mu) mu = u; if (u > xu) xu = u; if (v < mv) mv = v; if (v > xv) xv = v; }
          }
        if (mz > (real)c->cam_near) {
          const real W1 = (real)(c->cam_width - 1), H1 = (real)(c->cam_height - 1);
          const int u0 = (int)rclamp(R(floor)(mu) - R(1.0), R(0.0), W1), u1 = (int)rclamp(R(floor)(xu) + R(2.0), R(0.0), W1);
          const int v0 = (int)rclamp(R(floor)(mv) - R(1.0), R(0.0), H1), v1 = (int)rclamp(R(floor)(xv) + R(2.0), R(0.0), H1);
          const int ww = u1 - u0 + 1, hh = v1 - v0 + 1;
          const int total = (xu < R(0.0) || xv < R(0.0) || mu > W1 || mv > H1) ? 0 : ww * hh;
          /* a body with more than RV_PC_MAXPIX visible pixels: every stride-th one (scan order) is kept */
          int stride = 1;
          for (int pass = 0; pass < 2; ++pass) {
            n = 0;
            for (int idx = 0; idx < total; ++idx) {
              const int u = u0 + idx % ww, v = v0 + idx / ww;
              real dc[3], dw[3], d, pt[3];
              pc_pixel_dir_cam(e, (real)u, (real)v, dc); m3tmulv(dw, Rm, dc);
              int who = pc_render_pixel(w, e, rot, cam_o, dw, &d);
              if (!(who == b && d > (real)c->cam_near)) continue;
              pc_deproject(e, cam_o, (real)u, (real)v, d, pt);
              if (!pc_crop_ok(c, pt)) continue;
              if (n % stride == 0 && n / stride < RV_PC_MAXPIX) { pix[n / stride] = ((uint32_t)v << 16) | (uint32_t)u; dep[n / stride] = d; }
              n++;
            }
            if (n <= RV_PC_MAXPIX) break;
            if (pass == 0) stride = (n + RV_PC_MAXPIX - 1) / RV_PC_MAXPIX;
          }
          n = (n + stride - 1) / stride;
        }
      }
      if (n == 0) { for (int j = 0; j < P * 3; ++j) o[j] = 0.0f; continue; }
      if (n < P) {
        for (int j = 0; j < P; ++j) {
          uint32_t k = pc_hash(c, gid, rng_arg, ((uint32_t)b << 16) | (uint32_t)j) % (uint32_t)n;
          real pt[3]; pc_deproject(e, cam_o, (real)(pix[k] & 0xffffu), (real)(pix[k] >> 16), dep[k], pt);
          o[3 * j] = (float)pt[0]; o[3 * j + 1] = (float)pt[1]; o[3 * j + 2] = (float)pt[2];
        }
        continue;
      }
      for (int k = 0; k < n; ++k) { key[k] = pc_hash(c, gid, rng_arg, 0x80000000u | ((uint32_t)b << 16) | (uint32_t)k); sorted[k] = key[k]; }
      qsort(sorted, (size_t)n, sizeof(uint32_t), cmp_u32);
      const uint32_t T = sorted[P - 1];
      int n_lt = 0; for (int k = 0; k < n; ++k) n_lt += key[k] < T;
      int need = P - n_lt, outn = 0;
      for (int k = 0; k < n && outn < P; ++k) {
        int sel = key[k] < T;
        if (!sel && key[k] == T && need > 0) { sel = 1; need--; }
        if (!sel) continue;
        /* np.random.choice(replace=False) returns the subset in random order: the points go out in the order of
         * their keys (ties in scan order) */
        int rank = 0;
        for (int t = 0; t < n; ++t) rank += (key[t] < key[k] || (key[t] == key[k] && t < k)) ? 1 : 0;
        real pt[3]; pc_deproject(e, cam_o, (real)(pix[k] & 0xffffu), (real)(pix[k] >> 16), dep[k], pt);
        o[3 * rank] = (float)pt[0]; o[3 * rank + 1] = (float)pt[1]; o[3 * rank + 2] = (float)pt[2];
        outn++;
      }
    }
    free(pix); free(dep);
  }
}
void orc_reward(orc_world* w, double* reward, uint8_t* done) {
  /* an env that was not stepped by the last call (episode over) reports reward 0 */
  for (int i = 0; i < w->n; ++i) { reward[i] = w->env[i].substeps_last > 0 ? w->env[i].last_reward : R(0.0); done[i] = (uint8_t)w->env[i].done; }
}
void orc_get_episode_returns(orc_world* w, double* r) { for (int i = 0; i < w->n; ++i) r[i] = w->env[i].episode_reward; }
void orc_get_stats(orc_world* w, rv_macro_stats* s) { *s = w->stats; }

/* stand-alone reward evaluation for golden-vector tests (push_reward.py:302-372) */
void orc_eval_reward(const rv_config* cfg, const double* state, const double* next_state, double* reward, int32_t* term) {
  orc_world w; memset(&w, 0, sizeof(w)); w.cfg = *cfg;
  orc_env e; memset(&e, 0, sizeof(e));
  for (int b = 0; b < RV_MAXB; ++b) {
    e.prev_obs_pos[b][0] = (real)state[b * 2]; e.prev_obs_pos[b][1] = (real)state[b * 2 + 1];
    e.obs_pos[b][0] = (real)next_state[b * 2]; e.obs_pos[b][1] = (real)next_state[b * 2 + 1];
  }
  real r; int t;
  compute_reward(&w, &e, &r, &t);
  *reward = r; *term = t;
}
/* stand-alone waypoint evaluation (push_env.py:752-786) */
void orc_eval_waypoints(const rv_config* cfg, const float* action, double* start, double* end) {
  real a[4] = {(real)action[0], (real)action[1], (real)action[2], (real)action[3]}, s[7], e[7];
  compute_waypoints(cfg, a, s, e);
  for (int k = 0; k < 7; ++k) { start[k] = s[k]; end[k] = e[k]; }
}
/* Simulator.wait_until_stable counter logic (simulator.py:325-376) driven by a
 * scripted per-step stability flag instead of physics; stable[k-1] is the
 * check_stable() outcome after the k-th step. */
int orc_eval_wait_until_stable(const uint8_t* stable, int check_after, int min_stable, int max_steps) {
  int num_steps = 0, num_stable = 0;
  for (;;) {
    num_steps++;
    if (num_steps < check_after) continue;
    if (stable[num_steps - 1]) num_stable++;
    if (num_stable >= min_stable || num_steps >= max_steps) break;
  }
  return num_steps;
}
/* stand-alone GJK query for unit tests */
int orc_eval_gjk(const double* A, int nA, const double* B, int nB, double max_dist, double* out /* n3, dist, pa3, pb3 */) {
  real a[64][3], b[64][3];
  for (int i = 0; i < nA; ++i) for (int k = 0; k < 3; ++k) a[i][k] = (real)A[i * 3 + k];
  for (int i = 0; i < nB; ++i) for (int k = 0; k < 3; ++k) b[i][k] = (real)B[i * 3 + k];
  real ca[3] = {0, 0, 0}, cb[3] = {0, 0, 0}, g[3];
  for (int i = 0; i < nA; ++i) v3madd(ca, ca, a[i], R(1.0) / (real)nA);
  for (int i = 0; i < nB; ++i) v3madd(cb, cb, b[i], R(1.0) / (real)nB);
  v3sub(g, ca, cb);
  real n[3], dist, pa[3], pb[3];
  int hit = orc_gjk_epa((const real(*)[3])a, nA, (const real(*)[3])b, nB, g, (real)max_dist, n, &dist, pa, pb);
  if (hit) { for (int k = 0; k < 3; ++k) { out[k] = n[k]; out[4 + k] = pa[k]; out[7 + k] = pb[k]; } out[3] = dist; }
  return hit;
}

/* debugging aid: GJK calls / iterations since the library was loaded (-DORC_COUNT_GJK builds) */
/* diagnostics (bench.py, tools): island solves / sweeps / row steps of the impulse-space solver since the world was made;
 * OpenMP threads of the stepping entry points */
void orc_debug_solver_counts(orc_world* w, long* out) {
  out[0] = out[1] = out[2] = 0;
  for (int i = 0; i < w->n; ++i) { out[0] += w->env[i].cnt_islands; out[1] += w->env[i].cnt_sweeps; out[2] += w->env[i].cnt_rowsteps; }
}
void orc_set_num_threads(int n) {
#ifdef _OPENMP
  if (n > 0) omp_set_num_threads(n);
#else
  (void)n;
#endif
}
void orc_debug_gjk_counts(long* out) { out[0] = orc_gjk_calls; out[1] = orc_gjk_iters; }

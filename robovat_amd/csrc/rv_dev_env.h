// rv_dev_env.h — the per-env simulation program of the MI355X backend.
//
// Execution model: ONE wave64 owns ONE env for the whole
// launch.  The env's persistent state (`DevEnv`, ~5 KB) is copied from its
// contiguous HBM block into LDS once, every substep of the macro action runs
// out of LDS / registers, and the block is written back once at the end.  The
// program is written as a sequence of *lane phases*: inside a phase each of
// the 64 lanes works on its own item (a body, a manifold, a contact point, a
// hull vertex ...) and only reads data produced by earlier phases; phases are
// separated by a workgroup barrier (a workgroup is exactly one wave, so the
// barrier is an LDS wait).  With RV_ON_DEVICE == 0 (rv_dev_math.h: -DRV_EMULATE or a plain C++ compiler) the same source
// compiles as host C++ where a phase is a loop over 64 lanes (tests/emu, a debugging aid); the host bodies of the larger
// hooks are in tests/emu/rv_emu_hooks.h, included at the `#define RV_EMU_SECTION n` points and never seen by hipcc.
//
// Reference behaviour restated here (file:line in StanfordVL/robovat):
//   Simulator.step                      robovat/simulation/simulator.py:94-103
//   Simulator.wait_until_stable         robovat/simulation/simulator.py:325-376
//   ControllableBody.update & friends   robovat/simulation/controllable_body.py:387-595
//   SawyerSim.move_to_* / grip          robovat/robots/sawyer/sawyer_sim.py:186-408
//   PushEnv._execute_action & checks    robovat/envs/push/push_env.py:631-937
//   PushEnv._load_movable_bodies        robovat/envs/push/push_env.py:399-597
//   RobotEnv.reset / step               robovat/envs/robot_env.py:204-275
//   push_reward.get_reward_fn           robovat/reward_fns/push_reward.py:272-374
#pragma once
#include "../../include/rovat.h"
#include "rv_dev_collide.h"
#include "rv_dev_obs.h"

namespace rv {

#if RV_ON_DEVICE
#define RV_LANES_BEGIN { const int lane = (int)threadIdx.x;
#define RV_LANES_END } __syncthreads();
// a value every lane holds alike (read from the env's LDS block), moved to the scalar unit so that the control flow
// that hangs on it is compiled as uniform
#define RV_UNI(x) __builtin_amdgcn_readfirstlane((int)(x))
#else
#define RV_UNI(x) ((int)(x))
#define RV_LANES_BEGIN for (int lane = 0; lane < 64; ++lane) {
#define RV_LANES_END }
#endif

// profiling hook: leave the substep after phase group n, still advancing the step
// counter so that the control schedule (IK every 10 substeps) stays representative
#ifdef __HIP_DEVICE_COMPILE__
#define RV_STOP(n) if (K.stop_after == (n)) { if (threadIdx.x == 0) S.e.sim_steps++; __syncthreads(); return; }
#define RV_STOPL(n) if (K.stop_after == (n)) { if (threadIdx.x == 0) S.e.sim_steps++; __syncthreads(); return 0; }
#else
#define RV_STOP(n)
#define RV_STOPL(n)
#endif

// RV_PROFILE build only: lane 0 adds the shader-clock time since the last mark to slot i
#if defined(RV_PROFILE) && defined(__HIP_DEVICE_COMPILE__)
__device__ __forceinline__ void g_shared_prof(int i, unsigned long long t);
__device__ __forceinline__ void g_shared_profg(int g, int i, unsigned long long t);
#define RV_PROF(i) { if (threadIdx.x == 0) { unsigned long long t_ = __builtin_amdgcn_s_memtime(); g_shared_prof(i, t_); } }
// per-group marks inside the narrow phase: the first lane of 16-lane group g adds the time
// since ITS last mark to slot 12 + i (g = 0: table owners of bodies 0/2, g = 1: their arm owners)
#define RV_PROFG(i) { if ((int)threadIdx.x == 0) { unsigned long long t_ = __builtin_amdgcn_s_memtime(); g_shared_profg(0, (i), t_); } }
#else
#define RV_PROF(i)
#define RV_PROFG(i)
#endif
// RV_PROFILE build only: event counts in slots 32..39
#if defined(RV_PROFILE) && defined(__HIP_DEVICE_COMPILE__)
__device__ __forceinline__ void g_shared_cnt(int i, int n);
#define RV_PCNT(i, n) { if (threadIdx.x == 0) g_shared_cnt((i), (n)); }
#else
#define RV_PCNT(i, n)
#endif

#define RV_STREAM_RESET  1u
#define RV_STREAM_RANDOM 2u
#define RV_STREAM_HEUR   3u
#define RV_STREAM_CAMERA 4u
#define RV_STEPS_TO_CHECK_DONE 100   // controllable_body.py:21
#define RV_STEPS_TO_UPDATE_IK  10    // controllable_body.py:24

#define RV_TIDX(b) (b)
#define RV_BBIDX(k) (RV_MAXB + (k))
#define RV_AIDX(b) (RV_MAXB + RV_NBB + (b))

// JointTarget / LinkTarget (controllable_body.py:28-233)
struct JTarget {
  int active, n_idx, has_vel, has_stop;
  int from_ik;   // the target is the IK solution of the active link target: 1 = yes, 2 = yes and the
                 // solve ended on the residual test (a re-solve from it returns it unchanged)
  int idx[RV_NJ];
  float pos[RV_NJ];
  float start_t, stop_t, pos_thr, vel_thr;
};
struct LTarget {
  int active, has_pose, nq, has_stop;
  float pose[7];
  float queue[RV_MAXQ][7];
  float start_t, stop_t, pos_thr, vel_thr;
};

// Persistent per-env state: one contiguous block in HBM, staged in LDS.
struct DevEnv {
  float body[RV_MAXB][RV_BODY_STRIDE];
  int active[RV_MAXB], frozen[RV_MAXB], shape[RV_MAXB];
  int asleep[RV_MAXB], sleep_count[RV_MAXB], deact_count[RV_MAXB];
  int still_count[RV_MAXB]; float still_ref[RV_MAXB][7];   // pose window of the in-place oscillation test
  int undisturbed[RV_MAXB];   // woken, but has not left the pose window it was sleeping in
  float baabb[RV_MAXB][6];   // world box (lo, hi) of the hulls + margin, taken when the body fell asleep
  float scale[RV_MAXB], mass[RV_MAXB], inv_mass[RV_MAXB], inv_inertia[RV_MAXB][3], friction[RV_MAXB], radius[RV_MAXB];
  // user constraint (Simulator.add_constraint, simulator.py:166-224; bullet_physics.py:748-957): a fixed
  // joint between the frame (con_lpos, con_lquat) of the body and a frame of the world (con_tpos,
  // con_tquat), which can apply at most con_fmax N per row
  int con_on[RV_MAXB]; float con_lpos[RV_MAXB][3], con_lquat[RV_MAXB][4], con_tpos[RV_MAXB][3], con_tquat[RV_MAXB][4], con_fmax[RV_MAXB];
  float table_z;
  // the camera this env is observed with: rv_config's calibration + the noise drawn at its last reset (ArmEnv._reset_camera,
  // arm_env.py:109-152): fx, fy, cx, cy, skew; rotation (row-major); translation
  float cam_intrinsics[5], cam_rotation[9], cam_translation[3];
  int n_bodies;
  int arm_enabled;
  float q[RV_NJ], qd[RV_NJ];
  int motor_on[RV_NJ];
  float motor_q[RV_NJ], motor_kp[RV_NJ], motor_kd[RV_NJ], vmax_cmd[RV_NJ];
  JTarget jt;
  LTarget lt;
  float gripper_ready_time;
  float fpos[RV_NFRAME][3], fquat[RV_NFRAME][4];
  DevMan man[RV_NMAN];
  int flag_arm_table, flag_arm_body[RV_MAXB];
  int sim_steps, num_steps, num_episodes, done, phase, is_safe, is_effective, reset_count, substeps_last, awake_last, pairs_last;
  int stepped;      // this env executed an env.step() in the last macro launch
  // rv_step_begin / rv_step_poll (partial batches): an env.step() that is spread over several
  // launches.  in_step: 0 no step pending, 1 stepping, 2 a step was begun on a finished episode
  // (reported as finished with reward 0, like rv_step_macro skips it); step_stage: -1 not started,
  // 0 in the phase loop, 1 in the closing wait_until_stable; ms_*: the locals of _execute_action
  // that have to survive the launch boundary
  int in_step, step_stage;
  int ms_num_waypoints, ms_interrupt, ms_has_budget, ms_max_phase_steps, ms_wus_steps, ms_wus_stable;
  float ms_wp[RV_MAXG][2][7], ms_start_pos[RV_MAXB][3], ms_start_yaw[RV_MAXB];
  int reward_valid; // last_reward is the reward of the env.step() the last rv_step_macro / rollout gave this env
                    // (set by the step, cleared when a macro launch skips the env or the env is reset; other launches leave it)
  float episode_reward, last_reward;
  float action[RV_MAXG][4];
  float obs_pos[RV_MAXB][3], prev_obs_pos[RV_MAXB][3];
  float obs_quat[RV_MAXB][4];   // ... and the orientations at that moment (the observation of an env.step() is taken before get_reward, robot_env.py:246-248)
  int num_total_steps, num_unsafe, num_ineffective, num_useful, num_successes;
  // env.attributes (push_env.py:368-375, 637-644): the counters the attribute observations
  // and the heuristic policy see are captured at the START of _execute_action / _reset_scene
  int obs_num_steps, obs_num_episodes;
  // per-launch sums for rv_get_stats (a rollout launch takes several steps per env)
  int l_unsafe, l_ineffective, l_useful, l_episodes, l_successes;
  // lateral friction of the finger tips / the table (Link.set_dynamics, grasp_4dof_env.py:262-293);
  // PushEnv leaves them at PHYSICS.ARM_FRICTION / SIM.TABLE.FRICTION
  float mu_finger, mu_table;
  int num_action_steps;       // Grasp4DofEnv: substeps spent in the 'start' phase
  // rollouts through the task queue (rv_env_kernel.h): how many tasks of the running launch this block has been through, and
  // the sum of its other words when it was last handed on -- the next workgroup takes the block only when both are right
  int q_seq; unsigned q_sum;
#ifdef RV_PROFILE
  unsigned long long prof[48], prof_t, prof_t2[4];   // tools/prof_rollout.py: shader-clock time per substep part
#endif
};
static_assert(sizeof(DevEnv) % 4 == 0, "DevEnv is copied word-wise");

// one solver row set per contact point (normal + two friction directions)
struct Row {
  float dir[3][3], rxa[3][3], rxb[3][3], aa[3][3], ab[3][3];
  float invk[3], vbc[3];
  float target, mu;
  // dynamic finger (rv_config.finger_dynamics): the row also acts on finger joint 7 + fidx with
  // Jacobian jf = -(dir . slide axis); fidx = -1: no finger involved
  float jf[3]; int fidx;
  float cap;   // largest normal impulse the row may carry (arm effort limit); 1e30: none
};

struct Scratch {
  float frot[RV_NFRAME][9], fv[RV_NFRAME][3], fw[RV_NFRAME][3], axis[RV_NLIMB][3];
  float colv[RV_NCOL][8][3], colc[RV_NCOL][3], colr[RV_NCOL];
  float colmin[RV_NCOL][3], colmax[RV_NCOL][3];   // world AABB of each collider box
  int arm_moving;
  int colflag[RV_NCOL];
  float rot[RV_MAXB][9], iinv[RV_MAXB][9];
  float tablev[8][3], groundv[8][3];
  // solver rows + world hull vertices (rebuilt before every use: the solver borrows them as
  // scratch for the velocity update)
  struct {
    struct {
#if !RV_ON_DEVICE
      Row rows[RV_NMAN][4];      // host emulation only: on the device every solver sets its rows up in registers
#endif
      float wv[RV_MAXB][RV_MAXH][RV_MAXV][3];
    } r;
  } u;
  // macro-step locals that must survive across phases
  float wp[RV_MAXG][2][7];
  float gstart[7];                       // Grasp4DofEnv: the grasp pose of this step
  float start_pos[RV_MAXB][3], start_yaw[RV_MAXB];
  float poses[RV_MAXB][7];
  int num_waypoints, interrupt, has_budget, max_phase_steps;
  int loop_break, wus_steps, wus_stable, valid;
  int wake[RV_MAXB];
  // budget of the running sim_run_call (rv_step_poll): at most bud_sub substeps / bud_clk shader
  // clocks from bud_t0 in this launch (0 = no limit); suspended: the call returned on the budget
  int bud_sub, bud_sub0, suspended, wus_resume;
  unsigned long long bud_clk, bud_t0;
  int ready[RV_MAXB];                    // the body's own deactivation tests say it may sleep (islands sleep as a whole)
  float res[16];
  float mot[RV_MAXB];
  float sync;
  // ControllableBody._update_ik as a wave-wide solve: seed / solution, Jacobian, flags
  float ik_q[RV_NLIMB], ik_J[6][RV_NLIMB]; int ik_need, ik_conv;
  float lq[RV_NLIMB][4];
  float vdraw[RV_NJ], ratio[RV_NJ];      // motor phase: raw commanded velocity, limit factor
  float fing_dv[2], fing_vt[2], fing_qd0[2];   // finger motors this substep: velocity step taken, commanded velocity, velocity after it
  // limb dynamics (rv_config.limb_dynamics): the same of the limb joints; centres of mass, joint-space
  // inertia, the Gauss-Jordan workspace [M | 1] -> [. | M^-1], generalised gravity force, motor-row bounds
  // and targets; per arm contact row (12 x body + 3 x point + direction) the joint-space Jacobian,
  // M^-1 Ja^T and the effective mass
  float limb_dv[RV_NLIMB], limb_vt[RV_NLIMB], limb_qd0[RV_NLIMB];
  float lcom[RV_NLIMB + 1][3], lA[RV_NLIMB][2 * RV_NLIMB], lGq[RV_NLIMB];
  float llo[RV_NLIMB], lhi[RV_NLIMB], ltgt[RV_NLIMB];
  // (rows 0..11: the arm rows of the ONE awake body the lane-per-row solver handles; the last seven: the motor rows,
  // e_j and column j of M^-1.  The one-lane system solver, which takes the cases with several awake bodies, derives
  // the rows of a point when it visits it: limb_point_row)
  float lJa[12 + RV_NLIMB][RV_NLIMB], lMiJ[12 + RV_NLIMB][RV_NLIMB], linvk[12];
  int jmoving[RV_NJ], jchg[RV_NJ];
  float rvec[RV_NLIMB + 1][3];           // FK: link offsets rotated into the world
  int jt_applied;                        // the motor targets hold the current joint target (per launch)
  int atflag[RV_NCOL];                   // collider box may be within the contact-query distance of the table
  int kin_fresh;                         // FK / collider scratch matches the joint state (per launch)
  // "arm far" substeps (sim_substep_light): joint path lengths since the collider boxes were last computed, whether
  // those boxes exist at all in this launch, substeps taken far since then, and this substep's verdict
  float ftravel[RV_NJ]; int far_valid, far_n, far;
  int nearf[RV_MAXB][RV_NCOL], bnear[RV_MAXB], near_any;   // wake test stage 1 -> stage 2
  float sep[RV_MAXB][RV_NCOL], coltravel[RV_NCOL];          // distance-bound culling of the wake queries
  float cdelta[3][RV_NCOL];                                 // coasting: box travel bound per candidate length
  float clr_t[RV_NCOL], clr_b[RV_NCOL]; int clr_valid;      // coasting: clearances left (table, bodies)
  float jtravel[RV_NJ];                                     // coasting: joint path lengths of the chunk
  float ccoef[RV_NCOL][RV_NJ];                              // coasting: lever of joint j on collider box col (col_travelled's weights)
  float crun[RV_NCOL][RV_NJ];                               // ... as measured on the last fresh kinematics (coast_measure_clearances)
  int fused_n, fused_pending;
  int coast_unsafe[3];
  float jlen[RV_NLIMB + 1], colext[RV_NCOL];   // |jpos_i|; collider extent from its frame origin
  float fext[RV_NFRAME], fmot[RV_NFRAME];     // per frame: largest collider extent; vertex travel this substep
  int any_on;
  int pairs[4];
  // narrow-phase work list (per substep): refreshed distances / break flags per cached point,
  // (body, collider box) proximity flags, the owners that have convex queries to run
  float rf_dist[RV_NMAN * 4]; int rf_rm[RV_NMAN * 4];
  int cn[RV_MAXB][RV_NCOL];
  int ow_run[RV_NMAN + RV_NCOL], olist[RV_NMAN + RV_NCOL], n_olist;
  int wvneed[RV_MAXB];                   // body has hulls in a convex query of this substep
#if !RV_ON_DEVICE
  int rowmap[120], n_rows;   // host emulation of the impulse-space solver: rows in visiting order
#endif
  Rng rng;
};

struct Shared {
  DevEnv e;
  Scratch s;
  // launch constants staged in LDS: inside the out-of-line substep a pointer
  // argument is not provably wave-uniform, so reading rv_config / rv_arm
  // through it would be a vector global load (L2 latency) per field
  rv_config cfg;
  rv_arm arm;
  int n_hulls[RV_MAXB];
  int n_verts[RV_MAXB][RV_MAXH];
};

struct Consts {
  const rv_config* cfg;     // LDS copy (Shared::cfg) inside kernels
  const rv_arm* arm;        // LDS copy (Shared::arm) inside kernels
  const rv_scene* scene;    // global memory: hull vertices / inertia only
  int stop_after;
};

RV_DEV int bb_a(int k) { return k < 3 ? 0 : (k < 5 ? 1 : 2); }
RV_DEV int bb_b(int k) { return k == 0 ? 1 : (k == 1 ? 2 : (k == 2 ? 3 : (k == 3 ? 2 : 3))); }
// round-robin colouring: round r solves pairs {r, 5-r}, which touch disjoint bodies
RV_DEV int bb_round_pair(int r, int x) { return x == 0 ? r : 5 - r; }

// Contact-breaking threshold of a manifold = rv_config.breaking x the smaller "angular motion disc" of
// the two shapes (btCollisionShape::getContactBreakingThreshold, btPersistentManifold): the disc of a
// movable is its bounding radius about the body origin, that of a collider box its half diagonal;
// the table's and the ground's are larger than any of them.
RV_DEV float brk_body(const DevEnv& e, const rv_config* c, int b) { return c->breaking * (e.radius[b] - c->margin); }
RV_DEV float brk_col(const rv_arm* arm, const rv_config* c, int col) {
  const float* h = arm->col_half[col];
  return c->breaking * fsqrtr(h[0] * h[0] + h[1] * h[1] + h[2] * h[2]);
}
RV_DEV float brk_bb(const DevEnv& e, const rv_config* c, int a, int b) { return fminr(brk_body(e, c, a), brk_body(e, c, b)); }
RV_DEV float brk_ab(const DevEnv& e, const rv_arm* arm, const rv_config* c, int a, int col) { return fminr(brk_body(e, c, a), brk_col(arm, c, col)); }
// kind: 0 body-table, 1 body-body, 2 arm-body (col = the collider box)
RV_DEV float brk_of(const DevEnv& e, const rv_arm* arm, const rv_config* c, int kind, int a, int b, int col) {
  return kind == 0 ? brk_body(e, c, a) : (kind == 1 ? brk_bb(e, c, a, b) : brk_ab(e, arm, c, a, col));
}
// how close a moving collider box must come to the hulls of a sleeping body to wake it
RV_DEV float wake_range(const DevEnv& e, const rv_arm* arm, const rv_config* c, int b, int col) { return fminr(brk_ab(e, arm, c, b, col), c->wake_gap); }
RV_DEV float sim_time(const Shared& S, const Consts& K) { return K.cfg->dt * (float)S.e.sim_steps; }
RV_DEV int body_present(const DevEnv& e, int b) { return e.active[b] && !e.frozen[b]; }
RV_DEV int body_on(const DevEnv& e, int b) { return e.active[b] && !e.frozen[b] && !e.asleep[b]; }
// A STATIC body (Simulator.add_body(..., is_static=True), simulator.py:195-224; the wall of ArmEnv._reset_scene,
// arm_env.py:94-99): mass 0 as in Bullet -- inverse mass and inverse inertia 0.  It keeps its slot and its pair manifolds
// with the other bodies (their rows see a party that no impulse moves); it has no manifold with the table or the arm, takes
// no gravity and is not integrated; the arm does not wake it.  The env logic (observations, reward, safety, policies) looks
// at MOVABLE bodies only.
RV_DEV int body_static(const DevEnv& e, int b) { return e.inv_mass[b] == 0.0f; }
RV_DEV int body_movable(const DevEnv& e, int b) { return e.active[b] && !body_static(e, b); }
// start of a launch: nothing stepped yet (one lane)
RV_DEV void launch_counters_zero(DevEnv& e) {
  e.substeps_last = 0; e.awake_last = 0; e.pairs_last = 0; e.stepped = 0;
  e.l_unsafe = 0; e.l_ineffective = 0; e.l_useful = 0; e.l_episodes = 0; e.l_successes = 0;
}

// ------------------------------------------------------------------- arm --
// FK of the limb for joint vector q (registers / LDS), frames 0..7
struct LimbFK {
  v3 pos[RV_NLIMB + 1];
  q4 quat[RV_NLIMB + 1];
  v3 axis[RV_NLIMB];
};
// local joint rotation jquat_i o Rz(q_i)
RV_DEV q4 joint_local_quat(const rv_arm* a, int i, float qi) {
  float s, c; sincosr(qi * 0.5f, &s, &c);
  q4 qz; qz.x = 0.0f; qz.y = 0.0f; qz.z = s; qz.w = c;
  return qmul(ldq(a->jquat[i]), qz);
}
// chain composition given the local joint quaternions lq[0..6]
RV_DEV void fk_chain(const rv_arm* a, const q4* lq, LimbFK& F) {
  v3 pp = ld3(a->base_pos);
  q4 pq = ldq(a->base_quat);
#pragma unroll
  for (int i = 0; i < RV_NLIMB; ++i) {
    v3 po = add(pp, qrotv(pq, ld3(a->jpos[i])));
    q4 qf = qmul(pq, lq[i]);
    F.pos[i] = po; F.quat[i] = qf; F.axis[i] = qaxis_z(qf);
    pp = po; pq = qf;
  }
  F.pos[7] = add(pp, qrotv(pq, ld3(a->jpos[7])));
  F.quat[7] = qmul(pq, ldq(a->jquat[7]));
}
RV_DEV void fk_limb(const rv_arm* a, const float* q, LimbFK& F, float* frot_out /* [8][9] or null */) {
  q4 lq[RV_NLIMB];
#pragma unroll
  for (int i = 0; i < RV_NLIMB; ++i) lq[i] = joint_local_quat(a, i, q[i]);
  fk_chain(a, lq, F);
  if (frot_out) {
#pragma unroll
    for (int i = 0; i <= RV_NLIMB; ++i) stm(frot_out + 9 * i, qmat(F.quat[i]));
  }
}

// damped-least-squares IK (bullet_physics.py:1203-1262 call site), lane-serial
RV_DEV_NOINLINE int arm_ik(const Consts& K, const float* q0, const float* pose, float* out) {
  const rv_arm* a = K.arm;
  const rv_config* c = K.cfg;
  float q[RV_NLIMB];
  int conv = 0;
#pragma unroll
  for (int i = 0; i < RV_NLIMB; ++i) q[i] = q0[i];
  v3 tp = ld3(pose);
  q4 tq = ldq(pose + 3);
  RV_CNT(0, 1)
  for (int it = 0; it < c->ik_iters; ++it) {
    RV_CNT(1, 1)
    LimbFK F;
    fk_limb(a, q, F, nullptr);
    float err[6];
    v3 ep = sub(tp, F.pos[7]);
    err[0] = ep.x; err[1] = ep.y; err[2] = ep.z;
    q4 qc; qc.x = -F.quat[7].x; qc.y = -F.quat[7].y; qc.z = -F.quat[7].z; qc.w = F.quat[7].w;
    q4 qe = qmul(tq, qc);
    float sg = qe.w < 0.0f ? -2.0f : 2.0f;
    err[3] = qe.x * sg; err[4] = qe.y * sg; err[5] = qe.z * sg;
    float e2 = 0.0f;
#pragma unroll
    for (int k = 0; k < 6; ++k) e2 += err[k] * err[k];
    if (e2 < c->ik_residual * c->ik_residual) { conv = 1; break; }
    float J[6][RV_NLIMB];
#pragma unroll
    for (int j = 0; j < RV_NLIMB; ++j) {
      v3 cr = cross(F.axis[j], sub(F.pos[7], F.pos[j]));
      J[0][j] = cr.x; J[1][j] = cr.y; J[2][j] = cr.z;
      J[3][j] = F.axis[j].x; J[4][j] = F.axis[j].y; J[5][j] = F.axis[j].z;
    }
    float A[6][6];
    float lam2 = c->ik_damping * c->ik_damping;
#pragma unroll
    for (int r = 0; r < 6; ++r)
#pragma unroll
      for (int s = 0; s < 6; ++s) {
        float acc = 0.0f;
#pragma unroll
        for (int j = 0; j < RV_NLIMB; ++j) acc += J[r][j] * J[s][j];
        A[r][s] = acc + (r == s ? lam2 : 0.0f);
      }
    float L[6][6];
#pragma unroll
    for (int r = 0; r < 6; ++r)
#pragma unroll
      for (int s = 0; s <= r; ++s) {
        float acc = A[r][s];
#pragma unroll
        for (int k = 0; k < s; ++k) acc -= L[r][k] * L[s][k];
        if (r == s) L[r][r] = fsqrtr(fmaxr(acc, 1e-12f));
        else L[r][s] = acc / L[s][s];
      }
    float y[6];
#pragma unroll
    for (int r = 0; r < 6; ++r) {
      float acc = err[r];
#pragma unroll
      for (int k = 0; k < r; ++k) acc -= L[r][k] * y[k];
      y[r] = acc / L[r][r];
    }
#pragma unroll
    for (int r = 5; r >= 0; --r) {
      float acc = y[r];
#pragma unroll
      for (int k = r + 1; k < 6; ++k) acc -= L[k][r] * y[k];
      y[r] = acc / L[r][r];
    }
    float dq[RV_NLIMB], mx = 0.0f;
#pragma unroll
    for (int j = 0; j < RV_NLIMB; ++j) {
      float acc = 0.0f;
#pragma unroll
      for (int r = 0; r < 6; ++r) acc += J[r][j] * y[r];
      dq[j] = acc;
      mx = fmaxr(mx, fabsr(acc));
    }
    float sc = mx > c->ik_max_step ? c->ik_max_step / mx : 1.0f;
#pragma unroll
    for (int j = 0; j < RV_NLIMB; ++j) q[j] = fclampr(q[j] + dq[j] * sc, a->q_lo[j], a->q_hi[j]);
  }
#pragma unroll
  for (int i = 0; i < RV_NLIMB; ++i) out[i] = q[i];
  return conv;
}

// --------------------------------------------- ControllableBody restated --
RV_DEV void jt_reset(JTarget& t) { t.active = 0; t.n_idx = 0; t.has_stop = 0; t.from_ik = 0; }
RV_DEV void lt_reset(LTarget& t) { t.active = 0; t.has_pose = 0; t.nq = 0; t.has_stop = 0; }

RV_DEV int check_joints_reached(const DevEnv& e) {
  const JTarget& t = e.jt;
  if (!t.active) return 1;
  for (int i = 0; i < t.n_idx; ++i) {
    int j = t.idx[i];
    float dp = t.pos[i] - e.q[j];
    int pr = fabsr(dp) < t.pos_thr;
    int vr = 1;
    if (t.has_vel) { float dv = 0.0f - e.qd[j]; vr = fabsr(dv) < t.vel_thr; }
    if (!(pr && vr)) return 0;
  }
  return 1;
}
RV_DEV int check_link_target_done(const Shared& S, const Consts& K) {
  const LTarget& t = S.e.lt;
  if (!t.has_stop) return 1;
  if (sim_time(S, K) >= t.stop_t) return 1;
  if (!t.has_pose && t.nq == 0) return 1;
  return 0;
}
RV_DEV int check_joint_target_done(const Shared& S, const Consts& K) {
  const JTarget& t = S.e.jt;
  if (!t.has_stop) return 1;
  if (sim_time(S, K) >= t.stop_t) return 1;
  if (check_joints_reached(S.e)) return 1;
  return 0;
}
RV_DEV void lt_pop(LTarget& t) {
  if (t.nq == 0) { lt_reset(t); return; }
  for (int k = 0; k < 7; ++k) t.pose[k] = t.queue[0][k];
  for (int i = 1; i < t.nq; ++i) for (int k = 0; k < 7; ++k) t.queue[i - 1][k] = t.queue[i][k];
  t.nq--; t.has_pose = 1;
}
RV_DEV void arm_reset_targets(DevEnv& e) { lt_reset(e.lt); jt_reset(e.jt); }

// ControllableBody.update (controllable_body.py:387-413).  One lane runs it, in two parts around the
// IK solve, which is a job for the whole wave (arm_ik_wave): control_update_a() either finishes the
// update or leaves S.s.ik_need = 1 with the seed in S.s.ik_q; control_update_b() takes the solution.
RV_DEV void control_update_tail(Shared& S, const Consts& K, const int ik_updated) {
  DevEnv& e = S.e;
  if (ik_updated) {
    if (check_joints_reached(e)) {
      lt_pop(e.lt);                                  // next pose of the path: solve again
      if (e.jt.from_ik == 2) e.jt.from_ik = 1;
    }
  }
  if (e.jt.active) {
    if (e.sim_steps % RV_STEPS_TO_CHECK_DONE == 0 || ik_updated)
      if (check_joint_target_done(S, K)) jt_reset(e.jt);
  }
  if (e.jt.active) {
    // _update_position_control (controllable_body.py:458-466)
    for (int i = 0; i < e.jt.n_idx; ++i) {
      int j = e.jt.idx[i];
      e.motor_on[j] = 1; e.motor_q[j] = e.jt.pos[i];
      e.motor_kp[j] = K.cfg->kp; e.motor_kd[j] = K.cfg->kd;
    }
    S.s.jt_applied = 1;
  }
}
RV_DEV void control_update_a(Shared& S, const Consts& K) {
  DevEnv& e = S.e;
  S.s.ik_need = 0;
  {
    // most substeps nothing is due: no done-check (every 100 steps), no IK (every 10
    // steps), and the motor targets already hold the joint target (jt_applied).
    // One batch of LDS reads decides that.
    const int lt_on = e.lt.active, jt_on = e.jt.active, steps = e.sim_steps, applied = S.s.jt_applied;
    if (!lt_on && !jt_on) return;
    if (steps % RV_STEPS_TO_CHECK_DONE != 0 && jt_on && applied &&
        (!lt_on || steps % RV_STEPS_TO_UPDATE_IK != 0)) return;
  }
  int ik_updated = 0;
  if (e.lt.active) {
    if (e.sim_steps % RV_STEPS_TO_CHECK_DONE == 0)
      if (check_link_target_done(S, K)) lt_reset(e.lt);
  }
  if (e.lt.active) {
    if (e.sim_steps % RV_STEPS_TO_UPDATE_IK == 0 || !e.jt.active) {
      // _update_ik (controllable_body.py:468-499).  Skipped when the tracked target is
      // the converged solution of this very pose: solving again from it passes the
      // residual test at once and returns it unchanged.
      if (!(e.jt.active && e.jt.from_ik == 2)) {
        // seed: the previous IK solution while it is still being tracked, else the
        // current joint state
        const float* seed = (e.jt.active && e.jt.from_ik) ? e.jt.pos : e.q;
        for (int i = 0; i < RV_NLIMB; ++i) S.s.ik_q[i] = seed[i];
        S.s.ik_need = 1;
        return;
      }
      ik_updated = 1;
    }
  }
  control_update_tail(S, K, ik_updated);
}
RV_DEV void control_update_b(Shared& S, const Consts& K) {
  DevEnv& e = S.e;
  JTarget& t = e.jt;
  t.active = 1; t.n_idx = RV_NLIMB; t.has_vel = (e.lt.nq == 0); t.from_ik = S.s.ik_conv ? 2 : 1;
  for (int i = 0; i < RV_NLIMB; ++i) { t.idx[i] = i; t.pos[i] = S.s.ik_q[i]; }
  t.start_t = e.lt.start_t; t.stop_t = e.lt.stop_t; t.has_stop = 1;
  t.pos_thr = e.lt.pos_thr; t.vel_thr = e.lt.vel_thr;
  S.s.jt_applied = 0;
  control_update_tail(S, K, 1);
}
#if RV_ON_DEVICE
RV_DEV float ik_rdlane(float x, int l) { return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, x), l)); }
// arm_ik() by the whole wave, same arithmetic in the same order (bit-identical result):
// lane j < 7 owns joint j (its angle, its local quaternion, its Jacobian column); every lane runs
// the chain product and the 6x6 solve on broadcast values; lane 6 r + s sums A[r][s].
RV_DEV void arm_ik_wave(Shared& S, const Consts& K) {
  const rv_arm* a = K.arm; const rv_config* c = K.cfg;
  const int lane = (int)threadIdx.x;
  const int j = lane < RV_NLIMB ? lane : RV_NLIMB - 1;
  float q = S.s.ik_q[j];
  const float qlo = a->q_lo[j], qhi = a->q_hi[j];
  const v3 tp = ld3(S.e.lt.pose);
  const q4 tq = ldq(S.e.lt.pose + 3);
  const float lam2 = c->ik_damping * c->ik_damping;
  const float res2 = c->ik_residual * c->ik_residual;
  const int iters = __builtin_amdgcn_readfirstlane(c->ik_iters);
  const int ar = lane < 36 ? lane / 6 : 0, as = lane < 36 ? lane - 6 * (lane / 6) : 0;
  int conv = 0;
  for (int it = 0; it < iters; ++it) {
    // FK: the local quaternion of joint j on lane j, broadcast; the chain on every lane
    const q4 lqm = joint_local_quat(a, j, q);
    q4 lq[RV_NLIMB];
#pragma unroll
    for (int i = 0; i < RV_NLIMB; ++i) { lq[i].x = ik_rdlane(lqm.x, i); lq[i].y = ik_rdlane(lqm.y, i); lq[i].z = ik_rdlane(lqm.z, i); lq[i].w = ik_rdlane(lqm.w, i); }
    LimbFK F;
    fk_chain(a, lq, F);
    float err[6];
    v3 ep = sub(tp, F.pos[7]);
    err[0] = ep.x; err[1] = ep.y; err[2] = ep.z;
    q4 qc; qc.x = -F.quat[7].x; qc.y = -F.quat[7].y; qc.z = -F.quat[7].z; qc.w = F.quat[7].w;
    q4 qe = qmul(tq, qc);
    float sg = qe.w < 0.0f ? -2.0f : 2.0f;
    err[3] = qe.x * sg; err[4] = qe.y * sg; err[5] = qe.z * sg;
    float e2 = 0.0f;
#pragma unroll
    for (int k = 0; k < 6; ++k) e2 += err[k] * err[k];
    if (e2 < res2) { conv = 1; break; }
    // Jacobian column of joint j
    v3 axj = F.axis[0], pj = F.pos[0];
#pragma unroll
    for (int i = 1; i < RV_NLIMB; ++i) if (j == i) { axj = F.axis[i]; pj = F.pos[i]; }
    const v3 cr = cross(axj, sub(F.pos[7], pj));
    const float Jc[6] = {cr.x, cr.y, cr.z, axj.x, axj.y, axj.z};
    __syncthreads();                       // (the previous iteration's readers of ik_J are done)
    if (lane < RV_NLIMB) {
#pragma unroll
      for (int r = 0; r < 6; ++r) S.s.ik_J[r][lane] = Jc[r];
    }
    __syncthreads();
    float Ars;
    {
      float acc = 0.0f;
#pragma unroll
      for (int k = 0; k < RV_NLIMB; ++k) acc += S.s.ik_J[ar][k] * S.s.ik_J[as][k];
      Ars = acc + (ar == as ? lam2 : 0.0f);
    }
    // Cholesky and the two triangular solves on broadcast values
    float L[6][6];
#pragma unroll
    for (int r = 0; r < 6; ++r)
#pragma unroll
      for (int s2 = 0; s2 <= r; ++s2) {
        float acc = ik_rdlane(Ars, 6 * r + s2);
#pragma unroll
        for (int k = 0; k < s2; ++k) acc -= L[r][k] * L[s2][k];
        if (r == s2) L[r][r] = fsqrtr(fmaxr(acc, 1e-12f));
        else L[r][s2] = acc / L[s2][s2];
      }
    float y[6];
#pragma unroll
    for (int r = 0; r < 6; ++r) {
      float acc = err[r];
#pragma unroll
      for (int k = 0; k < r; ++k) acc -= L[r][k] * y[k];
      y[r] = acc / L[r][r];
    }
#pragma unroll
    for (int r = 5; r >= 0; --r) {
      float acc = y[r];
#pragma unroll
      for (int k = r + 1; k < 6; ++k) acc -= L[k][r] * y[k];
      y[r] = acc / L[r][r];
    }
    float dq = 0.0f;
#pragma unroll
    for (int r = 0; r < 6; ++r) dq += Jc[r] * y[r];
    float mx = 0.0f;
#pragma unroll
    for (int i = 0; i < RV_NLIMB; ++i) mx = fmaxr(mx, fabsr(ik_rdlane(dq, i)));
    const float sc = mx > c->ik_max_step ? c->ik_max_step / mx : 1.0f;
    q = fclampr(q + dq * sc, qlo, qhi);
  }
  __syncthreads();
  if (lane < RV_NLIMB) S.s.ik_q[lane] = q;
  if (lane == 0) S.s.ik_conv = conv;
  __syncthreads();
}
#endif
// ControllableBody.update for the env: every lane calls it
RV_DEV void control_update_phases(Shared& S, const Consts& K) {
  {
    // most substeps nothing is due (control_update_a's own first test, taken by every lane)
    const DevEnv& e = S.e;
    const int lt_on = e.lt.active, jt_on = e.jt.active, steps = e.sim_steps, applied = S.s.jt_applied;
    if (!lt_on && !jt_on) return;
    if (steps % RV_STEPS_TO_CHECK_DONE != 0 && jt_on && applied &&
        (!lt_on || steps % RV_STEPS_TO_UPDATE_IK != 0)) return;
  }
  RV_LANES_BEGIN
    if (lane == 0) control_update_a(S, K);
  RV_LANES_END
  if (S.s.ik_need) {
#if RV_ON_DEVICE
    arm_ik_wave(S, K);
#else
    {
      float qik[RV_NLIMB];
      S.s.ik_conv = arm_ik(K, S.s.ik_q, S.e.lt.pose, qik);
      for (int i = 0; i < RV_NLIMB; ++i) S.s.ik_q[i] = qik[i];
    }
#endif
    RV_LANES_BEGIN
      if (lane == 0) control_update_b(S, K);
    RV_LANES_END
  }
}
// ControllableBody.is_ready(limb joints) (controllable_body.py:565-595)
RV_DEV int arm_is_ready_limb(Shared& S, const Consts& K) {
  DevEnv& e = S.e;
  if (check_link_target_done(S, K)) lt_reset(e.lt);
  if (check_joint_target_done(S, K)) jt_reset(e.jt);
  if (e.lt.active) return 0;
  if (e.jt.active) {
    for (int i = 0; i < e.jt.n_idx; ++i) if (e.jt.idx[i] < RV_NLIMB) return 0;
  }
  return 1;
}
// SawyerSim.move_to_joint_positions (sawyer_sim.py:186-234)
RV_DEV void robot_move_to_joint_positions(Shared& S, const Consts& K, const float* pos) {
  DevEnv& e = S.e; const rv_config* c = K.cfg;
  arm_reset_targets(e);
  for (int j = 0; j < RV_NLIMB; ++j) e.vmax_cmd[j] = c->limb_max_velocity_ratio * K.arm->v_max[j];
  JTarget& t = e.jt;
  t.active = 1; t.n_idx = RV_NLIMB; t.has_vel = 1; t.from_ik = 0;
  S.s.jt_applied = 0;
  for (int i = 0; i < RV_NLIMB; ++i) { t.idx[i] = i; t.pos[i] = pos[i]; }
  t.start_t = sim_time(S, K); t.stop_t = t.start_t + c->limb_timeout; t.has_stop = 1;
  t.pos_thr = c->limb_position_threshold; t.vel_thr = c->velocity_threshold;
}
// SawyerSim.move_to_gripper_pose, straight_line=False (sawyer_sim.py:236-308)
RV_DEV void robot_move_to_gripper_pose(Shared& S, const Consts& K, const float* pose) {
  DevEnv& e = S.e; const rv_config* c = K.cfg;
  arm_reset_targets(e);
  for (int j = 0; j < RV_NLIMB; ++j) e.vmax_cmd[j] = c->limb_max_velocity_ratio * K.arm->v_max[j];
  LTarget& t = e.lt;
  t.active = 1; t.has_pose = 1; t.nq = 0;
  for (int k = 0; k < 7; ++k) t.pose[k] = pose[k];
  t.start_t = sim_time(S, K); t.stop_t = t.start_t + c->limb_timeout; t.has_stop = 1;
  t.pos_thr = c->limb_position_threshold; t.vel_thr = c->velocity_threshold;
}
// SawyerSim.grip (sawyer_sim.py:362-392)
RV_DEV void grip_env(DevEnv& e, const rv_arm* a, const rv_config* c, float value) {
  value = fclampr(value, 0.01f, 0.99f);
  float lpos = a->q_hi[7] - value * (a->q_hi[7] - a->q_lo[7]);
  float rpos = a->q_lo[8] + value * (a->q_hi[8] - a->q_lo[8]);
  const float now = c->dt * (float)e.sim_steps;
  JTarget& t = e.jt;
  t.active = 1; t.n_idx = 2; t.has_vel = 1; t.from_ik = 0;
  t.idx[0] = 7; t.idx[1] = 8; t.pos[0] = lpos; t.pos[1] = rpos;
  t.start_t = now; t.stop_t = t.start_t + 10000.0f; t.has_stop = 1;
  t.pos_thr = 0.008726640f; t.vel_thr = c->velocity_threshold;
  e.gripper_ready_time = now + 0.5f;
}
RV_DEV void robot_grip(Shared& S, const Consts& K, float value) {
  grip_env(S.e, K.arm, K.cfg, value);
  S.s.jt_applied = 0;
}

// ---------------------------------------------------------- rigid bodies --
RV_DEV void body_set_mass(DevEnv& e, const Consts& K, int b, float mass) {
  const rv_shape* s = &K.scene->shapes[e.shape[b]];
  e.mass[b] = mass; e.inv_mass[b] = mass > 0.0f ? 1.0f / mass : 0.0f;      // (mass 0: a static body)
  float s2 = e.scale[b] * e.scale[b];
  for (int k = 0; k < 3; ++k) e.inv_inertia[b][k] = mass > 0.0f ? 1.0f / (mass * s2 * s->inertia_k[k]) : 0.0f;
  e.radius[b] = s->radius * e.scale[b] + K.cfg->margin;
}
RV_DEV void cache_shape_meta(Shared& S, const Consts& K, int b) {
  const rv_shape* s = &K.scene->shapes[S.e.shape[b]];
  S.n_hulls[b] = s->n_hulls;
  for (int h = 0; h < RV_MAXH; ++h) S.n_verts[b][h] = s->n_verts[h];
}
RV_DEV void table_prepare(Shared& S, const Consts& K, int k) {
  const rv_config* c = K.cfg;
  float mg = c->margin;
  float hx = c->table_half[0] - mg, hy = c->table_half[1] - mg;
  float ztop = S.e.table_z - mg, zbot = S.e.table_z - c->table_thickness + mg;
  S.s.tablev[k][0] = c->table_center[0] + ((k & 1) ? hx : -hx);
  S.s.tablev[k][1] = c->table_center[1] + ((k & 2) ? hy : -hy);
  S.s.tablev[k][2] = (k & 4) ? ztop : zbot;
  // the ground: a 20 m x 20 m slab, 1 m thick, under the table
  S.s.groundv[k][0] = c->table_center[0] + ((k & 1) ? 10.0f : -10.0f);
  S.s.groundv[k][1] = c->table_center[1] + ((k & 2) ? 10.0f : -10.0f);
  S.s.groundv[k][2] = (k & 4) ? c->ground_z - mg : c->ground_z - 1.0f + mg;
}
// a body that is entirely below the table slab can only touch the ground: its "table"
// manifold then holds its contacts with the ground
RV_DEV int body_below_table(const DevEnv& e, const rv_config* c, int b) {
  return e.body[b][2] + e.radius[b] < e.table_z - c->table_thickness;
}

RV_DEV v3 to_local_body(const Shared& S, int b, v3 wp) { return tmulv(S.s.rot[b], sub(wp, ld3(S.e.body[b]))); }
RV_DEV v3 to_world_body(const Shared& S, int b, v3 lp) { return add(ld3(S.e.body[b]), mulv(S.s.rot[b], lp)); }
RV_DEV v3 to_local_frame(const Shared& S, int f, v3 wp) { return tmulv(S.s.frot[f], sub(wp, ld3(S.e.fpos[f]))); }
RV_DEV v3 to_world_frame(const Shared& S, int f, v3 lp) { return add(ld3(S.e.fpos[f]), mulv(S.s.frot[f], lp)); }

// kind: 0 body-table, 1 body-body, 2 arm-body
RV_DEV void point_world(const Shared& S, const Consts& K, int kind, int a, int b, const ManPoint& p, v3* wa, v3* wb) {
  *wa = to_world_body(S, a, p.la);
  if (kind == 0) *wb = p.lb;
  else if (kind == 1) *wb = to_world_body(S, b, p.lb);
  else *wb = to_world_frame(S, K.arm->col_frame[p.col], p.lb);
}
// refresh of ONE cached point: its current distance, and whether it broke (too far apart
// along the normal, or the two anchors drifted apart tangentially)
RV_DEV void refresh_point(const Shared& S, const Consts& K, int kind, int a, int b, const DevMan& m, int i, float* dist, int* rm) {
  ManPoint p;
  p.la = ld3(m.la[i]); p.lb = ld3(m.lb[i]); p.nrm = ld3(m.nrm[i]); p.col = m.col[i];
  const float brk = brk_of(S.e, K.arm, K.cfg, kind, a, b, p.col);
  v3 wa, wb;
  point_world(S, K, kind, a, b, p, &wa, &wb);
  float d = dot(sub(wa, wb), p.nrm);
  int r = 0;
  if (d > brk) r = 1;
  else {
    v3 proj = madd(wa, p.nrm, -d);
    v3 dr = sub(wb, proj);
    if (dot(dr, dr) > brk * brk) r = 1;
  }
  *dist = d; *rm = r;
}
// store the refreshed distances, drop the broken points (highest slot first); returns how many were lost
RV_DEV int refresh_apply(DevMan& m, const float* dist, const int* rm) {
  const int n0 = m.n;
#pragma unroll
  for (int i = 0; i < 4; ++i) if (i < n0) m.dist[i] = dist[i];
#pragma unroll
  for (int i = 3; i >= 0; --i) if (i < n0 && rm[i]) man_remove(m, i);
  return n0 - m.n;
}
RV_DEV void manifold_add_world(const Shared& S, const Consts& K, int kind, int a, int b, int col, DevMan& m, v3 wa, v3 wb, v3 n, float d, float brk) {
  v3 la = to_local_body(S, a, wa), lb;
  if (kind == 0) lb = wb;
  else if (kind == 1) lb = to_local_body(S, b, wb);
  else lb = to_local_frame(S, K.arm->col_frame[col], wb);
  man_add(m, la, lb, n, d, col, brk);
}

#define RV_MAN_C 0.932327f
#define RV_MAN_S 0.361615f
#define RV_MAN_TAU 0.1f
#define RV_FEATURE_PERIOD 4

// narrow phase of one convex pair; m == nullptr: distance only
RV_DEV int collide_pair(const Shared& S, const Consts& K, int kind, int a, int b, int col,
                        const float* A, int nA, const float* B, int nB, v3 guess, DevMan* m, float* out_dist, const float brk, const int pair) {
  float mg = K.cfg->margin;
  v3 n, pa, pb; float dist;
  RV_PROFG(0)
  const int hit_ = gjk_epa(A, nA, B, nB, guess, brk + 2.0f * mg, &n, &dist, &pa, &pb, nullptr, m ? &m->gc : nullptr, pair);
  RV_PROFG(1)
  if (!hit_) return 0;
  float d = dist - 2.0f * mg;
  if (d > brk) return 0;
  if (!(dot(n, n) > 0.5f)) return 0;   // safety net: never accept a non-unit normal
  *out_dist = d;
  if (!m) return 1;
  manifold_add_world(S, K, kind, a, b, col, *m, madd(pa, n, -mg), madd(pb, n, mg), n, d, brk);
  RV_PROFG(2)
  // feature stage: only while the manifold is incomplete, or every
  // RV_FEATURE_PERIOD-th full pass (cached points are refreshed every substep)
  {
    int age = m->age;
    if (m->n >= 4 && age < RV_FEATURE_PERIOD - 1) { m->age = age + 1; return 1; }
    m->age = 0;
  }
  // candidate vertices are picked along four tangent directions rotated off the plane axes (a box
  // aligned with them would offer a whole edge); a candidate is kept only inside the other hull's
  // extent along those four AND along the four plane axes themselves (an octagon: exact for a box
  // whose edges follow the axes, e.g. the table -- a body overhanging the table edge gets no
  // support beyond the edge)
  v3 t1, t2, dir[8];
  float extA[8], extB[8];
  plane_space(n, &t1, &t2);
  dir[0] = mk(RV_MAN_C * t1.x + RV_MAN_S * t2.x, RV_MAN_C * t1.y + RV_MAN_S * t2.y, RV_MAN_C * t1.z + RV_MAN_S * t2.z);
  dir[1] = mk(RV_MAN_C * t2.x - RV_MAN_S * t1.x, RV_MAN_C * t2.y - RV_MAN_S * t1.y, RV_MAN_C * t2.z - RV_MAN_S * t1.z);
  dir[2] = mk(-dir[0].x, -dir[0].y, -dir[0].z);
  dir[3] = mk(-dir[1].x, -dir[1].y, -dir[1].z);
  dir[4] = t1; dir[5] = t2; dir[6] = mk(-t1.x, -t1.y, -t1.z); dir[7] = mk(-t2.x, -t2.y, -t2.z);
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    extA[j] = support_proj(A, nA, dir[j]) + mg;
    extB[j] = support_proj(B, nB, dir[j]) + mg;
  }
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    float pj;
    v3 sd = madd(dir[k], n, -1.0f / RV_MAN_TAU);
    v3 va = support_v(A, nA, sd, &pj);
    float sep = dot(sub(va, pb), n);
    float gap = sep - 2.0f * mg;
    if (gap <= brk) {
      v3 pt = madd(va, n, -sep);
      int ok = 1;
#pragma unroll
      for (int j = 0; j < 8; ++j) if (dot(pt, dir[j]) > extB[j]) ok = 0;
      if (ok) manifold_add_world(S, K, kind, a, b, col, *m, madd(va, n, -mg), madd(pt, n, mg), n, gap, brk);
    }
    sd = madd(dir[k], n, 1.0f / RV_MAN_TAU);
    v3 vb = support_v(B, nB, sd, &pj);
    sep = dot(sub(pa, vb), n);
    gap = sep - 2.0f * mg;
    if (gap <= brk) {
      v3 pt = madd(vb, n, sep);
      int ok = 1;
#pragma unroll
      for (int j = 0; j < 8; ++j) if (dot(pt, dir[j]) > extA[j]) ok = 0;
      if (ok) manifold_add_world(S, K, kind, a, b, col, *m, madd(pt, n, -mg), madd(vb, n, mg), n, gap, brk);
    }
  }
  RV_PROFG(3)
  return 1;
}

RV_DEV float sphere_aabb_dist2(v3 p, const float* lo, const float* hi) {
  float d2 = 0.0f;
  float dx = 0.0f; if (p.x < lo[0]) dx = lo[0] - p.x; if (p.x > hi[0]) dx = p.x - hi[0]; d2 += dx * dx;
  float dy = 0.0f; if (p.y < lo[1]) dy = lo[1] - p.y; if (p.y > hi[1]) dy = p.y - hi[1]; d2 += dy * dy;
  float dz = 0.0f; if (p.z < lo[2]) dz = lo[2] - p.z; if (p.z > hi[2]) dz = p.z - hi[2]; d2 += dz * dz;
  return d2;
}
// squared distance between two axis-aligned boxes
RV_DEV float aabb_aabb_dist2(const float* lo_a, const float* hi_a, const float* lo_b, const float* hi_b) {
  float d2 = 0.0f;
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    float d = 0.0f;
    if (hi_a[k] < lo_b[k]) d = lo_b[k] - hi_a[k];
    if (hi_b[k] < lo_a[k]) d = lo_a[k] - hi_b[k];
    d2 += d * d;
  }
  return d2;
}
RV_DEV float sphere_box_dist2(v3 p, v3 c, v3 h) {
  float d2 = 0.0f;
  float dx = fabsr(p.x - c.x) - h.x; if (dx > 0.0f) d2 += dx * dx;
  float dy = fabsr(p.y - c.y) - h.y; if (dy > 0.0f) d2 += dy * dy;
  float dz = fabsr(p.z - c.z) - h.z; if (dz > 0.0f) d2 += dz * dz;
  return d2;
}

// ---- manifold owners of the narrow phase ----
// ids: T(b) = b (body - table), BB(k) = 4 + k (body - body), A(b) = 10 + b (arm - body),
// AT(col) = 14 + col (arm collider box - table, detection only: push_env.py:839-855)
struct OwnerInfo {
  int role;            // -1 nothing to query, 0 body-table, 1 body-body, 2 arm-body, 3 arm-table detector
  int kind;            // manifold kind (0 / 1 / 2) for point_world
  int a, b, mi;        // bodies (b = -1: none), manifold slot
  int clear, live;     // clear: drop the manifold; live: its cached points are refreshed
  int n_outer, n_inner;
  v3 guess0;
};
RV_DEV void owner_decode(const Shared& S, const Consts& K, int owner, int arm_on, OwnerInfo& o) {
  const rv_config* c = K.cfg; const DevEnv& e = S.e;
  v3 tc = mk(c->table_center[0], c->table_center[1], e.table_z - 0.5f * c->table_thickness);
  v3 th = mk(c->table_half[0], c->table_half[1], 0.5f * c->table_thickness);
  o.role = -1; o.kind = 0; o.a = 0; o.b = -1; o.mi = 0; o.clear = 0; o.live = 0; o.n_outer = 0; o.n_inner = 0;
  o.guess0 = mk(0.0f, 0.0f, 1.0f);
  if (owner < RV_MAXB) {
    const int a = owner; o.a = a; o.mi = RV_TIDX(a); o.kind = 0;
    if (!body_present(e, a) || body_static(e, a)) o.clear = 1;      // (a static body: pair manifolds only)
    else if (!e.asleep[a]) {
      o.live = 1;
      float r = e.radius[a] + brk_body(e, c, a);
      if (body_below_table(e, c, a)) {
        if (!(e.body[a][2] - c->ground_z >= r)) { o.role = 0; o.n_outer = 1; o.n_inner = S.n_hulls[a]; }
      } else if (!(sphere_box_dist2(ld3(e.body[a]), tc, th) >= r * r)) {
        o.role = 0; o.n_outer = 1; o.n_inner = S.n_hulls[a];
        o.guess0 = mk(0.0f, 0.0f, e.body[a][2] - tc.z);
      }
    }
  } else if (owner < RV_MAXB + RV_NBB) {
    const int k = owner - RV_MAXB, a = bb_a(k), b = bb_b(k);
    o.a = a; o.b = b; o.mi = RV_BBIDX(k); o.kind = 1;
    if (!(body_present(e, a) && body_present(e, b)) || (body_static(e, a) && body_static(e, b))) o.clear = 1;
    else if (!e.asleep[a] && !e.asleep[b]) {
      o.live = 1;
      v3 d = sub(ld3(e.body[a]), ld3(e.body[b]));
      float r = e.radius[a] + e.radius[b] + brk_bb(e, c, a, b);
      if (!(dot(d, d) >= r * r)) { o.role = 1; o.n_outer = S.n_hulls[a]; o.n_inner = S.n_hulls[b]; o.guess0 = d; }
    }
  } else if (owner < RV_NMAN) {
    const int a = owner - RV_MAXB - RV_NBB;
    o.a = a; o.mi = RV_AIDX(a); o.kind = 2;
    if (!body_present(e, a) || body_static(e, a)) o.clear = 1;
    else if (!e.asleep[a]) {
      if (!arm_on) o.clear = 1;
      else { o.live = 1; o.role = 2; o.n_outer = RV_NCOL; o.n_inner = S.n_hulls[a]; }
    }
  } else if (owner < RV_NMAN + RV_NCOL) {
    const int col = owner - RV_NMAN;
    o.a = col;
    if (arm_on) {
      float r = S.s.colr[col] + brk_col(K.arm, c, col);
      const float minz = S.s.colmin[col][2];
      // rejection: the flag needs dist < query_dist and dist >= minz - table_z - margin
      if (!(minz - e.table_z - c->margin >= c->contact_query_dist) &&
          sphere_box_dist2(ld3(S.s.colc[col]), tc, th) < r * r) { o.role = 3; o.n_outer = 1; o.n_inner = 1; }
    }
  }
}

// ------------------------------------------------------------------ PGS --
// ONE solver row of a contact point (k = 0 the normal row, 1 / 2 the friction rows): everything the solvers
// need of it.  row_setup() below is three calls of this; the lane-per-row solvers call it for their own row.
struct RowK { v3 dir, rxa, rxb, aa, ab; float invk, vbc, target, mu, jf, cap; int fidx; };
RV_DEV void row_setup_k(const Shared& S, const Consts& K, int kind, int a, int b, const ManPoint& p, const int k, const int n_pts, RowK& o) {
  const rv_config* c = K.cfg; const DevEnv& e = S.e;
  float dt = c->dt;
  v3 wa, wb;
  point_world(S, K, kind, a, b, p, &wa, &wb);
  v3 ra = sub(wa, ld3(e.body[a]));
  v3 d0 = p.nrm, d1, d2;
  plane_space(d0, &d1, &d2);
  v3 vb_pt = mk(0.0f, 0.0f, 0.0f);
  v3 rb = mk(0.0f, 0.0f, 0.0f);
  float imb = 0.0f;
  m3 iia = ldm(S.s.iinv[a]);
  m3 iib = iia;
  if (kind == 1) { rb = sub(wb, ld3(e.body[b])); imb = e.inv_mass[b]; iib = ldm(S.s.iinv[b]); }
  if (kind == 2) {
    int f = K.arm->col_frame[p.col];
    vb_pt = add(ld3(S.s.fv[f]), cross(ld3(S.s.fw[f]), sub(wb, ld3(e.fpos[f]))));
  }
  // a point on a finger pad of the force-limited gripper: the finger joint is a solver DOF
  const int fing = c->finger_dynamics && kind == 2 && p.col >= 8;
  const v3 fy = mk(S.s.frot[7][1], S.s.frot[7][4], S.s.frot[7][7]);   // slide axis = hand y
  float ima = e.inv_mass[a];
  {
    v3 dk = k == 0 ? d0 : (k == 1 ? d1 : d2);
    v3 rxa = cross(ra, dk);
    v3 aa = mulv(iia, rxa);
    float kk = ima + dot(rxa, aa);
    v3 rxb = mk(0.0f, 0.0f, 0.0f), ab = mk(0.0f, 0.0f, 0.0f);
    if (kind == 1) {
      rxb = cross(rb, dk);
      ab = mulv(iib, rxb);
      kk += imb + dot(rxb, ab);
    }
    float jf = 0.0f;
    if (fing) { jf = -dot(dk, fy); kk += jf * jf / c->finger_mass; }
    o.dir = dk; o.rxa = rxa; o.aa = aa; o.rxb = rxb; o.ab = ab;
    o.invk = 1.0f / kk;
    o.vbc = dot(dk, vb_pt);
    o.jf = jf;
  }
  o.fidx = fing ? p.col - 8 : -1;
  // what the arm can push with along the normal: min over the joints upstream of the collider of
  // tau_j / |J_j . n| (J_j = axis_j x (p - p_j); a finger pad also slides along the hand's y)
  o.cap = 1e30f;
  if (kind == 2 && c->arm_effort_limit) {
    const rv_arm* arm = K.arm;
    const int f = arm->col_frame[p.col];
    const int fl = f < RV_NLIMB ? f : RV_NLIMB - 1;
    float worst = 0.0f;
#pragma unroll
    for (int j = 0; j < RV_NLIMB; ++j) {
      if (j > fl) continue;
      const v3 lever = cross(ld3(S.s.axis[j]), sub(wb, ld3(e.fpos[j])));
      worst = fmaxr(worst, fabsr(dot(lever, d0)) * arm->inv_tau_max[j]);
    }
    if (f >= 8 && !c->finger_dynamics) worst = fmaxr(worst, fabsr(dot(fy, d0)) * arm->inv_tau_max[f - 1]);
    // (the budget is shared equally by the points of the manifold)
    if (worst > 0.0f) o.cap = dt / (worst * (float)n_pts);
  }
  float dist = p.dist;
  if (dist > 0.0f) o.target = -dist / dt;
  else o.target = fminr(c->erp * fmaxr(-dist - c->slop, 0.0f) / dt, c->max_pushout);
  float mub = (kind == 0) ? (body_below_table(e, c, a) ? c->ground_friction : e.mu_table) : (kind == 1 ? e.friction[b] : (p.col >= 8 ? e.mu_finger : c->arm_friction));
  o.mu = e.friction[a] * mub;
}
RV_DEV void row_setup(const Shared& S, const Consts& K, int kind, int a, int b, const ManPoint& p, Row& r, const int n_pts) {
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    RowK o;
    row_setup_k(S, K, kind, a, b, p, k, n_pts, o);
    st3(r.dir[k], o.dir); st3(r.rxa[k], o.rxa); st3(r.aa[k], o.aa); st3(r.rxb[k], o.rxb); st3(r.ab[k], o.ab);
    r.invk[k] = o.invk; r.vbc[k] = o.vbc; r.jf[k] = o.jf;
    r.fidx = o.fidx; r.cap = o.cap; r.target = o.target; r.mu = o.mu;
  }
}

// kind / bodies of manifold slot mi
RV_DEV void man_owner(int mi, int* kind, int* a, int* b) {
  if (mi < RV_MAXB) { *kind = 0; *a = mi; *b = -1; }
  else if (mi < RV_MAXB + RV_NBB) { *kind = 1; *a = bb_a(mi - RV_MAXB); *b = bb_b(mi - RV_MAXB); }
  else { *kind = 2; *a = mi - RV_MAXB - RV_NBB; *b = -1; }
}
// The residual an island's sweeps stop on: an island all of whose bodies were below the sleep thresholds after the last
// substep converges to rv_config.solver_tol_rest (< solver_tol) -- with the plain early exit resting bodies creep at
// ~4e-5 m/s, which only deactivation hides (Bullet: 50 sweeps, no exit).  Islands that hold finger / limb motor rows
// keep solver_tol.  rest_mask: bit b = body b is awake and NOT at rest (sleep_count == 0).
RV_DEV int unrest_mask(const DevEnv& e) {
  int m = 0;
#pragma unroll
  for (int b = 0; b < RV_MAXB; ++b) if (body_on(e, b) && !(e.sleep_count[b] > 0)) m |= 1 << b;
  return m;
}
RV_DEV float tol_of(const rv_config* c, const int unrest_members) {
  return (c->solver_tol_rest > 0.0f && c->solver_tol_rest < c->solver_tol && unrest_members == 0) ? c->solver_tol_rest : c->solver_tol;
}
// The Row of manifold point (mi, i) for the one-lane system solver of the host emulation: the record the
// row-setup phase left in LDS.  (Device: no Row records exist -- they were 13 KB of the env's LDS block; see
// SerialRows.)
#if !RV_ON_DEVICE
RV_DEV Row fetch_row(const Shared& S, const Consts& K, int mi, int i) { (void)K; return S.s.u.r.rows[mi][i]; }
#endif

// body velocity pair held in registers by the solving lane
struct BV { v3 v, w; };
RV_DEV BV ld_bv(const DevEnv& e, int b) { BV x; x.v = ld3(e.body[b] + 7); x.w = ld3(e.body[b] + 10); return x; }
RV_DEV void st_bv(DevEnv& e, int b, const BV& x) { st3(e.body[b] + 7, x.v); st3(e.body[b] + 10, x.w); }

RV_DEV void row_apply(BV& A, BV* B, float ima, float imb, const Row& r, int k, float dl) {
  A.v = madd(A.v, ld3(r.dir[k]), dl * ima);
  A.w = madd(A.w, ld3(r.aa[k]), dl);
  if (B) {
    B->v = madd(B->v, ld3(r.dir[k]), -dl * imb);
    B->w = madd(B->w, ld3(r.ab[k]), -dl);
  }
}
RV_DEV float row_jv(const BV& A, const BV* B, const Row& r, int k) {
  float jv = dot(ld3(r.dir[k]), A.v) + dot(ld3(r.rxa[k]), A.w);
  if (B) jv -= dot(ld3(r.dir[k]), B->v) + dot(ld3(r.rxb[k]), B->w);
  else jv -= r.vbc[k];
  return jv;
}
struct Lam { float n, t1, t2; };
RV_DEV float point_solve(BV& A, BV* B, float ima, float imb, Lam& l, const Row& r) {
  float jv = row_jv(A, B, r, 0);
  float dl = (r.target - jv) * r.invk[0];
  float ln = fclampr(l.n + dl, 0.0f, r.cap);
  dl = ln - l.n; l.n = ln;
  float res = fabsr(dl);
  row_apply(A, B, ima, imb, r, 0, dl);
  float lim = r.mu * ln;
  jv = row_jv(A, B, r, 1);
  dl = -jv * r.invk[1];
  float l1 = fclampr(l.t1 + dl, -lim, lim);
  dl = l1 - l.t1; l.t1 = l1;
  res = fmaxr(res, fabsr(dl));
  row_apply(A, B, ima, imb, r, 1, dl);
  jv = row_jv(A, B, r, 2);
  dl = -jv * r.invk[2];
  float l2 = fclampr(l.t2 + dl, -lim, lim);
  dl = l2 - l.t2; l.t2 = l2;
  res = fmaxr(res, fabsr(dl));
  row_apply(A, B, ima, imb, r, 2, dl);
  return res;
}
RV_DEV void warm_apply(BV& A, BV* B, float ima, float imb, const Lam& l, const Row& r) {
  row_apply(A, B, ima, imb, r, 0, l.n); row_apply(A, B, ima, imb, r, 1, l.t1); row_apply(A, B, ima, imb, r, 2, l.t2);
}

// ---- PGS with the force-limited gripper (rv_config.finger_dynamics; Grasp4DofEnv) ----------
// The two finger joints are dynamic 1-DoF bodies of mass finger_mass sliding along the hand's
// y axis: contact rows on a finger pad (collider boxes 8 / 9) carry the Jacobian entry
// jf = -(dir . y) on the finger velocity, and each finger has a POSITION_CONTROL motor row
// (bullet_physics.py:1061-1104) that pulls its velocity to the commanded one with at most
// finger_max_force: the joint motors of the light part have already spent m * fing_dv of that
// budget on the free motion, the row may add the rest.  Velocity-space Gauss-Seidel over ALL
// awake bodies and both fingers as one system, one lane (a grasp scene has one body).
// ---- limb dynamics (rv_config.limb_dynamics) ----------------------------------
// While the arm touches an awake body the seven limb joints are unknowns of the solver (PyBullet: the
// arm is a btMultiBody whose POSITION_CONTROL motors are constraint rows of the same PGS,
// controllable_body.py:458-466, bullet_physics.py:1061-1104).  The unknown is the DEVIATION dq of the
// joint velocities from the ones the motor law of this substep commanded (limb_qd0): the link twists
// the contact rows were set up with stay as they are, a contact row on a collider of frame f gets the
// joint-space Jacobian  Ja_j = -dir . (axis_j x (p - p_j)), j <= min(f, 6), an impulse dl on it changes
// dq by M^-1 Ja^T dl, and its effective mass gains Ja M^-1 Ja^T.  M(q) is the joint-space inertia of
// the chain of eight masses (links 0..6 and the hand):
//   M_jk = sum_{i >= max(j,k)} m_i (a_j x (c_i - p_j)) . (a_k x (c_i - p_k)) + (R_i^T a_j) . I_i (R_i^T a_k)
// (the composite-rigid-body sum written out for revolute joints; velocity-product terms are neglected:
// the limb moves at < 1 rad/s).  The solve starts from the joint velocities BEFORE the motor step of
// this substep (dq = -limb_dv: the acceleration limits of the kinematic motor law are no statement about
// torques).  Motor row j: J = e_j, target dq_j = commanded - current velocity, accumulated impulse within
// +- tau_j dt minus the torque that holds the chain against gravity.  After the solve the joints move
// with the solved velocity and the link frames are recomputed.
// limb_prepare: wave-wide -- one lane per entry of M, per column of the Gauss-Jordan elimination of
// [M | 1], per contact row; the Gauss-Seidel itself is the velocity-space system solver below.
// row k of arm point i of body b in limb_dynamics mode: joint-space Jacobian ja, M^-1 ja^T, 1 / effective mass
// (needs M^-1 in S.s.lA: limb_prepare)
RV_DEV void limb_point_row(const Shared& S, const Consts& K, int b, int i, int k, float* ja, float* mij, float* invk_out) {
  const DevEnv& e = S.e; const rv_arm* arm = K.arm;
  const DevMan& m = e.man[RV_AIDX(b)];
  const int f = arm->col_frame[m.col[i]], fl = f < RV_NLIMB ? f : RV_NLIMB - 1;
  const v3 wb = to_world_frame(S, f, ld3(m.lb[i]));
  ManPoint pt;
  pt.la = ld3(m.la[i]); pt.lb = ld3(m.lb[i]); pt.nrm = ld3(m.nrm[i]); pt.dist = m.dist[i]; pt.col = m.col[i];
  RowK r;
  row_setup_k(S, K, 2, b, -1, pt, k, m.n, r);       // (the row as the solver sets it up)
  const v3 dk = r.dir;
#pragma unroll
  for (int j = 0; j < RV_NLIMB; ++j) {
    const v3 lever = cross(ld3(S.s.axis[j]), sub(wb, ld3(e.fpos[j])));
    ja[j] = j <= fl ? -dot(dk, lever) : 0.0f;
  }
  float kk = 1.0f / r.invk;
#pragma unroll
  for (int j = 0; j < RV_NLIMB; ++j) {
    float a_ = 0.0f;
#pragma unroll
    for (int x = 0; x < RV_NLIMB; ++x) a_ = a_ + S.s.lA[j][RV_NLIMB + x] * ja[x];
    mij[j] = a_;
  }
#pragma unroll
  for (int j = 0; j < RV_NLIMB; ++j) kk = kk + ja[j] * mij[j];
  *invk_out = 1.0f / kk;
}
// lbody: the one awake body whose arm rows the lane-per-row solver will want (-1: none)
RV_DEV void limb_prepare(Shared& S, const Consts& K, const int lbody) {
  const rv_config* c = K.cfg; const rv_arm* arm = K.arm;
  RV_LANES_BEGIN
    if (lane <= RV_NLIMB) st3(S.s.lcom[lane], add(ld3(S.e.fpos[lane]), mulv(S.s.frot[lane], ld3(arm->link_com[lane]))));
  RV_LANES_END
  RV_LANES_BEGIN
    const DevEnv& e = S.e;
    if (lane < RV_NLIMB * RV_NLIMB) {
      const int j = lane / RV_NLIMB, k = lane - j * RV_NLIMB, mx = j > k ? j : k;
      const v3 aj = ld3(S.s.axis[j]), ak = ld3(S.s.axis[k]), pj = ld3(e.fpos[j]), pk = ld3(e.fpos[k]);
      float acc = 0.0f;
      for (int i = 0; i <= RV_NLIMB; ++i) {
        if (i < mx) continue;
        const v3 ci = ld3(S.s.lcom[i]);
        const v3 lj = cross(aj, sub(ci, pj)), lk = cross(ak, sub(ci, pk));
        const float t = arm->link_mass[i] * dot(lj, lk);
        const v3 bj = tmulv(S.s.frot[i], aj), bk = tmulv(S.s.frot[i], ak);
        const float r = (bj.x * bk.x) * arm->link_inertia[i][0] + (bj.y * bk.y) * arm->link_inertia[i][1]
                      + (bj.z * bk.z) * arm->link_inertia[i][2];
        acc = acc + (t + r);
      }
      S.s.lA[j][k] = acc; S.s.lA[j][RV_NLIMB + k] = j == k ? 1.0f : 0.0f;
    } else if (lane >= 56 && lane < 56 + RV_NLIMB) {
      const int j = lane - 56;
      const v3 aj = ld3(S.s.axis[j]), pj = ld3(e.fpos[j]), g = mk(c->gravity_xy[0], c->gravity_xy[1], c->gravity_z);
      float gq = 0.0f;
      for (int i = 0; i <= RV_NLIMB; ++i) {
        if (i < j) continue;
        const v3 lj = cross(aj, sub(ld3(S.s.lcom[i]), pj));
        gq = gq + arm->link_mass[i] * dot(lj, g);
      }
      S.s.lGq[j] = gq;
    }
  RV_LANES_END
  // M^-1 by Gauss-Jordan on [M | 1] without pivoting (M is symmetric positive definite), one lane per
  // column; a column that has been the pivot column is not touched again
  for (int p = 0; p < RV_NLIMB; ++p) {
    RV_LANES_BEGIN
      if (lane < 2 * RV_NLIMB && lane > p) {
        const int col = lane;
        const float ap = S.s.lA[p][col] / S.s.lA[p][p];
        for (int r = 0; r < RV_NLIMB; ++r) if (r != p) S.s.lA[r][col] = S.s.lA[r][col] - S.s.lA[r][p] * ap;
        S.s.lA[p][col] = ap;
      }
    RV_LANES_END
  }
  RV_LANES_BEGIN
    const DevEnv& e = S.e; const float dt = c->dt;
    if (lane < RV_NLIMB) {
      const int j = lane;
      const float hold = -S.s.lGq[j] * dt;
      const float tdt = arm->inv_tau_max[j] > 0.0f ? dt / arm->inv_tau_max[j] : 1e30f;
      S.s.llo[j] = fminr(0.0f, -tdt - hold); S.s.lhi[j] = fmaxr(0.0f, tdt - hold);
      S.s.ltgt[j] = S.s.limb_vt[j] - S.s.limb_qd0[j];
#pragma unroll
      for (int x = 0; x < RV_NLIMB; ++x) { S.s.lJa[12 + j][x] = x == j ? 1.0f : 0.0f; S.s.lMiJ[12 + j][x] = S.s.lA[x][RV_NLIMB + j]; }
    } else if (lbody >= 0 && lane >= 8 && lane < 8 + 12) {
      const int row = lane - 8, b = lbody, i = row / 3, k = row - i * 3;
      const DevMan& m = e.man[RV_AIDX(b)];
      if (body_on(e, b) && i < m.n) {
        float ja[RV_NLIMB], mij[RV_NLIMB], ik;
        limb_point_row(S, K, b, i, k, ja, mij, &ik);
#pragma unroll
        for (int j = 0; j < RV_NLIMB; ++j) { S.s.lJa[row][j] = ja[j]; S.s.lMiJ[row][j] = mij[j]; }
        S.s.linvk[row] = ik;
      }
    }
  RV_LANES_END
}
RV_DEV float limb_jv(const float* Ja, const float* dq) {
  float a_ = 0.0f;
#pragma unroll
  for (int j = 0; j < RV_NLIMB; ++j) a_ = a_ + Ja[j] * dq[j];
  return a_;
}
RV_DEV void limb_apply(const float* MiJ, float* dq, float dl) {
#pragma unroll
  for (int j = 0; j < RV_NLIMB; ++j) dq[j] = dq[j] + MiJ[j] * dl;
}
// lj / lm / lk: the limb Jacobians Ja[3][7], M^-1 Ja^T [3][7] and effective masses [3] of an arm row in
// limb_dynamics mode (null otherwise), dq: the deviation of the joint velocities
// (lon says whether the limb arrays are there: a test of the POINTER would not do -- on the device the caller's
// arrays live in private memory, where offset 0 is a valid address)
RV_DEV float point_solve_g(BV& A, float ima, Lam& l, const Row& r, float* qf, float imf, const int lon,
                           const float (*lj)[RV_NLIMB], const float (*lm)[RV_NLIMB], const float* lk, float* dq) {
  const int fi = r.fidx;
  float jv = row_jv(A, nullptr, r, 0);
  if (fi >= 0) jv += r.jf[0] * qf[fi];
  if (lon) jv += limb_jv(lj[0], dq);
  float dl = (r.target - jv) * (lon ? lk[0] : r.invk[0]);
  float ln = fclampr(l.n + dl, 0.0f, r.cap);
  dl = ln - l.n; l.n = ln;
  float res = fabsr(dl);
  row_apply(A, nullptr, ima, 0.0f, r, 0, dl);
  if (fi >= 0) qf[fi] += r.jf[0] * dl * imf;
  if (lon) limb_apply(lm[0], dq, dl);
  float lim = r.mu * ln;
  jv = row_jv(A, nullptr, r, 1);
  if (fi >= 0) jv += r.jf[1] * qf[fi];
  if (lon) jv += limb_jv(lj[1], dq);
  dl = -jv * (lon ? lk[1] : r.invk[1]);
  float l1 = fclampr(l.t1 + dl, -lim, lim);
  dl = l1 - l.t1; l.t1 = l1;
  res = fmaxr(res, fabsr(dl));
  row_apply(A, nullptr, ima, 0.0f, r, 1, dl);
  if (fi >= 0) qf[fi] += r.jf[1] * dl * imf;
  if (lon) limb_apply(lm[1], dq, dl);
  jv = row_jv(A, nullptr, r, 2);
  if (fi >= 0) jv += r.jf[2] * qf[fi];
  if (lon) jv += limb_jv(lj[2], dq);
  dl = -jv * (lon ? lk[2] : r.invk[2]);
  float l2 = fclampr(l.t2 + dl, -lim, lim);
  dl = l2 - l.t2; l.t2 = l2;
  res = fmaxr(res, fabsr(dl));
  row_apply(A, nullptr, ima, 0.0f, r, 2, dl);
  if (fi >= 0) qf[fi] += r.jf[2] * dl * imf;
  if (lon) limb_apply(lm[2], dq, dl);
  return res;
}
// The six rows of a user constraint on body b (a fixed joint to a frame of the world: the mocap-style
// constraint ControllableConstraint servoes, controllable_constraint.py:21-170): three linear rows at the
// pivot, three angular rows, Baumgarte-stabilised with rv_config.erp, accumulated impulse within
// +- con_fmax dt per row (pybullet changeConstraint maxForce).  lam[6]: accumulated impulses.
// con_on[b] encodes the joint: bits 0-3 the type (RV_CON_FIXED: six rows, RV_CON_P2P: point to point, the three linear
// rows only), bits 4-7 the child body + 1 (0: the world).  With a child body the frame (con_tpos, con_tquat) is given
// in the CHILD's frame and every row acts on both bodies (opposite signs); a child that is absent or frozen stands
// still like the world.
#define RV_CON_FIXED 1
#define RV_CON_P2P 2
#define RV_CON_PRISMATIC 3
#define RV_CON_REVOLUTE 4
#define RV_CON_TYPE(x) ((x) & 15)
#define RV_CON_CHILD(x) ((((x) >> 4) & 15) - 1)
RV_DEV float constraint_solve(Shared& S, const Consts& K, int b, float* lam) {
  DevEnv& e = S.e; const rv_config* c = K.cfg;
  const float dt = c->dt, lim = e.con_fmax[b] * dt, ima = e.inv_mass[b];
  const int ctype = RV_CON_TYPE(e.con_on[b]);
  int cb = RV_CON_CHILD(e.con_on[b]);
  const int cw = RV_CON_CHILD(e.con_on[b]);       // (-1: the frame is a world frame)
  // child >= RV_MAXB: frame cw - RV_MAXB of the ARM (a link as the other party, bullet_physics.py:773-779): a kinematic frame
  // that moves with the link's twist and takes no impulse
  const int lf = cw >= RV_MAXB ? cw - RV_MAXB : -1;
  if (lf >= 0) cb = -2;
  if (cb >= 0 && !body_on(e, cb)) cb = -2;        // (an absent / frozen / sleeping child: a fixed frame where it is)
  const m3 rot = qmat(ldq(e.body[b] + 3));
  const v3 r = mulv(rot, ld3(e.con_lpos[b])), wp = add(ld3(e.body[b]), r);
  const q4 qw = qmul(ldq(e.body[b] + 3), ldq(e.con_lquat[b]));
  q4 qc; qc.x = -qw.x; qc.y = -qw.y; qc.z = -qw.z; qc.w = qw.w;
  // the frame the joint is tied to, in the world
  v3 tpv = ld3(e.con_tpos[b]), rc = mk(0.0f, 0.0f, 0.0f);
  q4 tq = ldq(e.con_tquat[b]);
  float imc = 0.0f;
  v3 lv = mk(0.0f, 0.0f, 0.0f), lw = mk(0.0f, 0.0f, 0.0f);      // velocity of the link frame at the pivot, its spin
  if (lf >= 0) {
    const m3 rotc = qmat(ldq(e.fquat[lf]));
    rc = mulv(rotc, ld3(e.con_tpos[b]));
    tpv = add(ld3(e.fpos[lf]), rc);
    tq = qmul(ldq(e.fquat[lf]), ldq(e.con_tquat[b]));
    if (e.arm_enabled) { lv = add(ld3(S.s.fv[lf]), cross(ld3(S.s.fw[lf]), sub(wp, ld3(e.fpos[lf])))); lw = ld3(S.s.fw[lf]); }
  } else if (cw >= 0) {
    const m3 rotc = qmat(ldq(e.body[cw] + 3));
    rc = mulv(rotc, ld3(e.con_tpos[b]));
    tpv = add(ld3(e.body[cw]), rc);
    tq = qmul(ldq(e.body[cw] + 3), ldq(e.con_tquat[b]));
    if (cb >= 0) imc = e.inv_mass[cb];
  }
  const q4 qe = qmul(tq, qc);
  const float sg = qe.w < 0.0f ? -2.0f : 2.0f;
  const float th[3] = {qe.x * sg, qe.y * sg, qe.z * sg};
  const float tp[3] = {tpv.x, tpv.y, tpv.z}, wpa[3] = {wp.x, wp.y, wp.z};
  float res = 0.0f;
  // prismatic (RV_CON_PRISMATIC; pybullet JOINT_PRISMATIC): the body slides along the x axis of the frame it is tied
  // to -- two linear rows along that frame's y and z axes, then the three angular rows
  // revolute (RV_CON_REVOLUTE; 'revolute' of the reference's JOINT_TYPES_MAPPING, bullet_physics.py:20-25): a hinge about the x
  // axis of the frame -- the three linear rows at the pivot and two angular rows along that frame's y and z axes
  const int n_lin = ctype == RV_CON_PRISMATIC ? 2 : 3, n_ang = ctype == RV_CON_REVOLUTE ? 2 : 3, n_rows = ctype == RV_CON_P2P ? 3 : n_lin + n_ang;
  const m3 rt = qmat(tq);
  const v3 dtp = sub(tpv, wp);
  for (int k = 0; k < n_rows; ++k) {
    v3 jl = mk(0, 0, 0), ja = mk(0, 0, 0), jc = mk(0, 0, 0); float bias;
    if (k < n_lin && ctype == RV_CON_PRISMATIC) {
      jl = mulv(rt, mk(0.0f, k == 0 ? 1.0f : 0.0f, k == 1 ? 1.0f : 0.0f));      // column k + 1 of the frame's rotation
      // ONE anchor for both parties: the parent's pivot wp (as Bullet's slider does) -- along the slide axis wp and the
      // child's frame origin are far apart, and equal and opposite impulses at two points would be a spurious torque
      // (round-4 advisor): the child's lever is wp - x_child
      const v3 rcw = (cw >= 0 && lf < 0) ? sub(wp, ld3(e.body[cw])) : mk(0.0f, 0.0f, 0.0f);
      ja = cross(r, jl); jc = cross(rcw, jl);
      bias = c->erp * dot(jl, dtp) / dt;
    } else if (k < n_lin) {
      const v3 ek = mk(k == 0 ? 1.0f : 0.0f, k == 1 ? 1.0f : 0.0f, k == 2 ? 1.0f : 0.0f);
      jl = ek; ja = cross(r, ek); jc = cross(rc, ek);
      bias = c->erp * (tp[k] - wpa[k]) / dt;
    } else if (ctype == RV_CON_REVOLUTE) {
      const int a_ = k - n_lin;
      ja = mulv(rt, mk(0.0f, a_ == 0 ? 1.0f : 0.0f, a_ == 1 ? 1.0f : 0.0f)); jc = ja;      // column a_ + 1 of the frame's rotation
      bias = c->erp * dot(ja, mk(th[0], th[1], th[2])) / dt;
    } else {
      const int a_ = k - n_lin;
      ja = mk(a_ == 0 ? 1.0f : 0.0f, a_ == 1 ? 1.0f : 0.0f, a_ == 2 ? 1.0f : 0.0f); jc = ja;
      bias = c->erp * th[a_] / dt;
    }
    const v3 ia = mulv(ldm(S.s.iinv[b]), ja);
    float kk = (k < n_lin ? ima : 0.0f) + dot(ja, ia);
    float jv = dot(jl, ld3(e.body[b] + 7)) + dot(ja, ld3(e.body[b] + 10));
    if (lf >= 0) jv = jv - (k < n_lin ? dot(jl, lv) : dot(ja, lw));
    v3 ic = mk(0, 0, 0);
    if (cb >= 0) {
      ic = mulv(ldm(S.s.iinv[cb]), jc);
      kk = kk + ((k < n_lin ? imc : 0.0f) + dot(jc, ic));
      jv = jv - (dot(jl, ld3(e.body[cb] + 7)) + dot(jc, ld3(e.body[cb] + 10)));
    }
    // a row no party of which can move (a static parent -- mass 0 -- constrained to the world, a link or another static
    // body): nothing to solve, as in Bullet, where a constraint on a fixed-base body is harmless (0 / 0 here otherwise)
    if (!(kk > 0.0f)) continue;
    float dl = (bias - jv) / kk;
    const float ln = fclampr(lam[k] + dl, -lim, lim);
    dl = ln - lam[k]; lam[k] = ln;
    res = fmaxr(res, fabsr(dl));
    st3(e.body[b] + 7, madd(ld3(e.body[b] + 7), jl, dl * ima));
    st3(e.body[b] + 10, madd(ld3(e.body[b] + 10), ia, dl));
    if (cb >= 0) {
      st3(e.body[cb] + 7, madd(ld3(e.body[cb] + 7), jl, -(dl * imc)));
      st3(e.body[cb] + 10, madd(ld3(e.body[cb] + 10), ic, -dl));
    }
  }
  return res;
}
// is body b a party to a body - body constraint?  (the two stay awake together: they never deactivate)
RV_DEV int con_pair_member(const DevEnv& e, int b) {
  int m = RV_CON_TYPE(e.con_on[b]) != 0 && RV_CON_CHILD(e.con_on[b]) >= 0;
  for (int x = 0; x < RV_MAXB; ++x) if (RV_CON_TYPE(e.con_on[x]) != 0 && RV_CON_CHILD(e.con_on[x]) == b) m = 1;
  return m;
}
#if RV_ON_DEVICE
// Device: the one-lane system solver is run by EVERY lane of the wave (the same scalar program on the same LDS
// data: the lanes agree on every value they store), so that the row sets it visits can live in registers: lane
// 4 mi + i holds the Row of point i of manifold mi -- and, in limb mode, the limb rows of an arm point -- set up
// once per solve (serial_rows_setup), and a visit pulls them with v_readlane at a wave-uniform lane index.
// (Recomputing the rows at every visit, as the first version without LDS Row records did, made the rare envs
// that take this path the tail of the whole launch.)
struct SerialRows { Row my; float lja[3][RV_NLIMB], lmi[3][RV_NLIMB], llk[3]; };
RV_DEV float rdl_u(float x, int src) { return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, x), src)); }
RV_DEV Row serial_pull_row(const SerialRows& R, const int src) {
  Row r;
#pragma unroll
  for (int k = 0; k < 3; ++k) {
#pragma unroll
    for (int x = 0; x < 3; ++x) {
      r.dir[k][x] = rdl_u(R.my.dir[k][x], src); r.rxa[k][x] = rdl_u(R.my.rxa[k][x], src); r.rxb[k][x] = rdl_u(R.my.rxb[k][x], src);
      r.aa[k][x] = rdl_u(R.my.aa[k][x], src); r.ab[k][x] = rdl_u(R.my.ab[k][x], src);
    }
    r.invk[k] = rdl_u(R.my.invk[k], src); r.vbc[k] = rdl_u(R.my.vbc[k], src); r.jf[k] = rdl_u(R.my.jf[k], src);
  }
  r.target = rdl_u(R.my.target, src); r.mu = rdl_u(R.my.mu, src); r.cap = rdl_u(R.my.cap, src);
  r.fidx = __builtin_amdgcn_readlane(R.my.fidx, src);
  return r;
}
RV_DEV void serial_pull_limb(const SerialRows& R, const int src, float (*pja)[RV_NLIMB], float (*pmi)[RV_NLIMB], float* plk) {
#pragma unroll
  for (int k = 0; k < 3; ++k) {
#pragma unroll
    for (int x = 0; x < RV_NLIMB; ++x) { pja[k][x] = rdl_u(R.lja[k][x], src); pmi[k][x] = rdl_u(R.lmi[k][x], src); }
    plk[k] = rdl_u(R.llk[k], src);
  }
}
RV_DEV void serial_rows_setup(const Shared& S, const Consts& K, const int limb, SerialRows& R) {
  const DevEnv& e = S.e;
  const int lane = (int)threadIdx.x;
  const int L = lane < RV_NMAN * 4 ? lane : RV_NMAN * 4 - 1;
  const int mi = L >> 2, i = L & 3;
  int kind, a, b;
  man_owner(mi, &kind, &a, &b);
  const DevMan& m = e.man[mi];
  const bool use = lane < RV_NMAN * 4 && body_on(e, a) && (kind != 1 || body_on(e, b)) && i < m.n;
  ManPoint pt;
  pt.la = ld3(m.la[i]); pt.lb = ld3(m.lb[i]); pt.nrm = ld3(m.nrm[i]); pt.dist = m.dist[i]; pt.col = m.col[i];
  if (!use) pt.col = 0;                                  // (a stale collider index of an empty slot must not index anything)
  row_setup(S, K, kind, a, b, pt, R.my, m.n);            // (lanes without a point compute a row nobody asks for)
#pragma unroll
  for (int k = 0; k < 3; ++k) {
#pragma unroll
    for (int x = 0; x < RV_NLIMB; ++x) { R.lja[k][x] = 0.0f; R.lmi[k][x] = 0.0f; }
    R.llk[k] = 0.0f;
  }
  if (limb && kind == 2 && use) {
#pragma unroll
    for (int k = 0; k < 3; ++k) limb_point_row(S, K, a, i, k, R.lja[k], R.lmi[k], &R.llk[k]);
  }
}
#define RV_SERIAL_ROWS_ARG , const SerialRows& SR
#else
#define RV_SERIAL_ROWS_ARG
#endif
RV_DEV void solve_with_fingers(Shared& S, const Consts& K, const int limb RV_SERIAL_ROWS_ARG) {
  DevEnv& e = S.e; const rv_config* c = K.cfg; const rv_arm* arm = K.arm;
  float dq[RV_NLIMB], lam_l[RV_NLIMB];
#pragma unroll
  for (int j = 0; j < RV_NLIMB; ++j) { dq[j] = limb ? -S.s.limb_dv[j] : 0.0f; lam_l[j] = 0.0f; }   // (the solve starts from the velocity before the motor step)
  const int fd = c->finger_dynamics && e.arm_enabled;            // the finger joints are DOFs of the system
  const float mf = c->finger_mass, imf = 1.0f / c->finger_mass, fdt = c->finger_max_force * c->dt;
  float qf[2] = {e.qd[RV_NLIMB], e.qd[RV_NLIMB + 1]}, lam_m[2] = {0.0f, 0.0f};
  float lam_c[RV_MAXB][6];
  for (int b = 0; b < RV_MAXB; ++b) for (int k = 0; k < 6; ++k) lam_c[b][k] = 0.0f;
  float best = 1e30f; int since = 0;              // rv_config.solver_stall
  for (int it = -1; it < c->solver_iters; ++it) {   // it == -1: warm start
    float res = 0.0f;
    for (int b = 0; b < RV_MAXB; ++b) {
      if (!body_on(e, b)) continue;
      BV A = ld_bv(e, b); const float ima = e.inv_mass[b];
      for (int kind = 0; kind < 2; ++kind) {
        const int mi = kind == 0 ? RV_TIDX(b) : RV_AIDX(b);
        DevMan& m = e.man[mi];
        for (int i = 0; i < m.n; ++i) {
          const int la = limb && kind == 1;
          float pja[3][RV_NLIMB], pmi[3][RV_NLIMB], plk[3];
#if RV_ON_DEVICE
          const int src = __builtin_amdgcn_readfirstlane(mi * 4 + i);
          Row r = serial_pull_row(SR, src);
          if (la) serial_pull_limb(SR, src, pja, pmi, plk);
#else
          Row r = fetch_row(S, K, mi, i);
          if (la) { for (int kk = 0; kk < 3; ++kk) limb_point_row(S, K, b, i, kk, pja[kk], pmi[kk], &plk[kk]); }
#endif
          Lam l; l.n = m.ln[i]; l.t1 = m.lt1[i]; l.t2 = m.lt2[i];
          if (it < 0) {
            warm_apply(A, nullptr, ima, 0.0f, l, r);
            if (r.fidx >= 0) { qf[r.fidx] += r.jf[0] * l.n * imf; qf[r.fidx] += r.jf[1] * l.t1 * imf; qf[r.fidx] += r.jf[2] * l.t2 * imf; }
            if (la) { limb_apply(pmi[0], dq, l.n); limb_apply(pmi[1], dq, l.t1); limb_apply(pmi[2], dq, l.t2); }
          } else {
            res = fmaxr(res, point_solve_g(A, ima, l, r, qf, imf, la, pja, pmi, plk, dq));
            m.ln[i] = l.n; m.lt1[i] = l.t1; m.lt2[i] = l.t2;
          }
        }
      }
      st_bv(e, b, A);
    }
    for (int rd = 0; rd < 3; ++rd)
      for (int x = 0; x < 2; ++x) {
        const int k = bb_round_pair(rd, x);
        const int a_ = bb_a(k), b_ = bb_b(k);
        if (!(body_on(e, a_) && body_on(e, b_))) continue;
        DevMan& m = e.man[RV_BBIDX(k)];
        if (m.n == 0) continue;
        BV A = ld_bv(e, a_), B = ld_bv(e, b_);
        const float ima = e.inv_mass[a_], imb = e.inv_mass[b_];
        for (int i = 0; i < m.n; ++i) {
#if RV_ON_DEVICE
          Row r = serial_pull_row(SR, __builtin_amdgcn_readfirstlane(RV_BBIDX(k) * 4 + i));
#else
          Row r = fetch_row(S, K, RV_BBIDX(k), i);
#endif
          Lam l; l.n = m.ln[i]; l.t1 = m.lt1[i]; l.t2 = m.lt2[i];
          if (it < 0) warm_apply(A, &B, ima, imb, l, r);
          else { res = fmaxr(res, point_solve(A, &B, ima, imb, l, r)); m.ln[i] = l.n; m.lt1[i] = l.t1; m.lt2[i] = l.t2; }
        }
        st_bv(e, a_, A); st_bv(e, b_, B);
      }
    if (it < 0) continue;
    for (int b = 0; b < RV_MAXB; ++b) if (body_on(e, b) && e.con_on[b]) res = fmaxr(res, constraint_solve(S, K, b, lam_c[b]));
    for (int f = 0; fd && f < 2; ++f) {      // motor rows
      const float i0 = mf * S.s.fing_dv[f];
      float dl = (S.s.fing_vt[f] - qf[f]) * mf;
      const float ln = fclampr(lam_m[f] + dl, -fdt - i0, fdt - i0);
      dl = ln - lam_m[f]; lam_m[f] = ln;
      qf[f] += dl * imf;
      res = fmaxr(res, fabsr(dl));
    }
    for (int j = 0; limb && j < RV_NLIMB; ++j) {   // limb motor rows
      float dl = (S.s.ltgt[j] - dq[j]) / S.s.lA[j][RV_NLIMB + j];
      const float ln = fclampr(lam_l[j] + dl, S.s.llo[j], S.s.lhi[j]);
      dl = ln - lam_l[j]; lam_l[j] = ln;
#pragma unroll
      for (int k = 0; k < RV_NLIMB; ++k) dq[k] = dq[k] + S.s.lA[k][RV_NLIMB + j] * dl;
      res = fmaxr(res, fabsr(dl));
    }
    if (res < c->solver_tol) break;
    if (c->solver_stall > 0) { if (res < best) { best = res; since = 0; } else if (++since >= c->solver_stall) break; }
  }
  for (int j = 0; limb && j < RV_NLIMB; ++j) {   // the limb moves with the solved velocity
    float qd = S.s.limb_qd0[j] + dq[j];
    float qn = e.q[j] + dq[j] * c->dt;
    if (qn < arm->q_lo[j]) { qn = arm->q_lo[j]; qd = 0.0f; }
    if (qn > arm->q_hi[j]) { qn = arm->q_hi[j]; qd = 0.0f; }
    e.q[j] = qn; e.qd[j] = qd;
  }
  if (limb) S.s.kin_fresh = 0;                   // the frames no longer match the joints
  for (int f = 0; fd && f < 2; ++f) {        // the fingers move with the solved velocity
    const int j = RV_NLIMB + f;
    float qd = qf[f];
    float qn = e.q[j] + (qd - S.s.fing_qd0[f]) * c->dt;
    if (qn < arm->q_lo[j]) { qn = arm->q_lo[j]; qd = 0.0f; }
    if (qn > arm->q_hi[j]) { qn = arm->q_hi[j]; qd = 0.0f; }
    e.q[j] = qn; e.qd[j] = qd;
  }
}

// ---- PGS in impulse space, one lane per solver row -------------------------------------
// The rows of ALL islands of the env, in the order the sequential Gauss-Seidel visits them:
// body by body (ascending) the points of its table manifold, then of its arm manifold; then
// the body-body manifolds in colour-round order; per point the normal row, then the two
// friction rows.  Row r has J_r on body a_r (+dir, +rxa) and, for a body-body row, on b_r
// (-dir, -rxb); an impulse dl on row s changes the velocity of a_s by P0_s dl = (dir_s / m_a,
// aa_s) dl and of b_s by P1_s dl = -(dir_s / m_b, ab_s) dl.  So the row velocities
// g_r = J_r u - vbc_r evolve as g += A[:, s] dl with the Delassus matrix A_rs = J_r[a_s].P0_s +
// J_r[b_s].P1_s (zero when rows r and s share no body: islands stay independent by themselves).
// The iterates are those of the velocity-space solver (same order, clamps, per-island residual
// test); what changes is the cost of a row step: lane r keeps g_r, lambda_r and its row of A in
// registers, so a step is a handful of VALU instructions and one v_readlane broadcast instead
// of ~40 dependent instructions and a 56-word LDS row fetch.  The body velocities are rebuilt
// once at the end, u = u0 + sum_s P_s lambda_s.  More than RV_SOLVE_ROWS rows (three or four
// bodies in mutual contact): the env falls back to the velocity-space solver.
// Host emulation / oracle: the same arithmetic on arrays.
#define RV_SOLVE_ROWS 120
#define RV_ROW_PACK(mi, i, k, a, b, isl) ((mi) | ((i) << 8) | ((k) << 12) | ((a) << 16) | (((b) + 1) << 20) | ((isl) << 24))
#define RV_ROW_MI(x)  ((x) & 255)
#define RV_ROW_I(x)   (((x) >> 8) & 15)
#define RV_ROW_K(x)   (((x) >> 12) & 15)
#define RV_ROW_A(x)   (((x) >> 16) & 15)
#define RV_ROW_B(x)   ((((x) >> 20) & 15) - 1)
#define RV_ROW_ISL(x) (((x) >> 24) & 15)
#if !RV_ON_DEVICE
#define RV_EMU_SECTION 1
#include "../../tests/emu/rv_emu_hooks.h"      // host lane emulation (test scaffolding; not compiled into the product)
#undef RV_EMU_SECTION
#endif
struct J6 { v3 l, a; };
RV_DEV float dotj(const J6& j, v3 pl, v3 pa) { return dot(j.l, pl) + dot(j.a, pa); }
#if RV_ON_DEVICE
RV_DEV float rdlane(float x, int l) { return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, x), l)); }
RV_DEV v3 rdlane3(v3 x, int l) { return mk(rdlane(x.x, l), rdlane(x.y, l), rdlane(x.z, l)); }
// One island of TWO bodies X < Y, with a STATIC lane layout so that every index below is a compile-time
// constant (row data, the lane's row of the matrix and the sweep live in registers):
//   lanes  0..23  body X: table points 0..3, arm points 0..3, x 3 rows
//   lanes 24..47  body Y likewise
//   lanes 48..59  the X-Y manifold, points 0..3 x 3 rows
// which is the visiting order of the row list.  JX / JY: the lane's row acting on X / Y;
// PX / PY: what a unit impulse on the lane's row does to X / Y.  (Islands of ONE body are solve_singles'.)
// Row step in normalised residual form (see solve_singles): rr = (bias - g) invk, C_rs = -(A_rs invk_r);
// nl = med3(lam + rr, lo, hi); d = nl - lam; rr = fma(C[s], d_s, rr) with d_s broadcast by v_readlane.
RV_DEV int isl_row_on(int s, int ntx, int nax, int nty, int nay, int nxy) {
  const int blk = s < 24 ? 0 : (s < 48 ? 1 : 2);
  const int p = (s - 24 * blk) / 3;
  return blk == 0 ? (p < 4 ? p < ntx : p - 4 < nax) : (blk == 1 ? (p < 4 ? p < nty : p - 4 < nay) : p < nxy);
}
// lane S of the caller's own 16-lane group, to every lane of the group: the DPP control row_newbcast:S (gfx90a and
// later: "broadcast lane S of each row of 16 to the whole row") -- ONE VALU instruction; no trip through SGPRs, and not
// the ~100 clocks of the LDS crossbar that the ds_swizzle of the first version took on the critical path of every row step
// "is it this lane's turn" of the lane-per-row sweeps: (lane's row) == s for a compile-time s.  The compare is loop
// invariant, so the compiler computes the 12 (36) lane masks once, keeps them in SGPR pairs -- and, out of SGPRs, spills them
// into a VGPR: every row step then paid two v_readlane + a wait state to get its mask back.  An opaque copy of the lane's
// row index makes the compare part of the step: v_cmp_eq (inline constant) + v_cndmask through vcc, two instructions.
RV_DEV int opaque_i(int x) {
#if RV_ON_DEVICE
  asm volatile("" : "+v"(x));
#endif
  return x;
}
template <int S_> RV_DEV float grp_bcast(float x) {
  const int xi = __builtin_bit_cast(int, x);
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(xi, xi, 0x150 + S_, 0xf, 0xf, true));
}
template <int N_> RV_DEV int grp_ror_max(int x) {       // max with the lane N_ to the right in the 16-lane row (row_ror)
  const int y = __builtin_amdgcn_update_dpp(x, x, 0x120 + N_, 0xf, 0xf, true);
  return x > y ? x : y;
}
RV_DEV void solve_island2(Shared& S, const Consts& K, const int X, const int Y, const int kxy, const float tol) {
  DevEnv& e = S.e; const rv_config* c = K.cfg;
  const int lane = (int)threadIdx.x;
  const int ntx = __builtin_amdgcn_readfirstlane(e.man[RV_TIDX(X)].n), nax = __builtin_amdgcn_readfirstlane(e.man[RV_AIDX(X)].n);
  const int nty = __builtin_amdgcn_readfirstlane(e.man[RV_TIDX(Y)].n), nay = __builtin_amdgcn_readfirstlane(e.man[RV_AIDX(Y)].n);
  const int nxy = __builtin_amdgcn_readfirstlane(e.man[RV_BBIDX(kxy)].n);
  if (ntx + nax + nty + nay + nxy == 0) return;
  const int L = lane < 60 ? lane : 59;
  const int blk = L < 24 ? 0 : (L < 48 ? 1 : 2);
  const int p = (L - 24 * blk) / 3, k = (L - 24 * blk) - 3 * p, slot = p & 3;
  const int mi = blk == 0 ? (p < 4 ? RV_TIDX(X) : RV_AIDX(X)) : (blk == 1 ? (p < 4 ? RV_TIDX(Y) : RV_AIDX(Y)) : RV_BBIDX(kxy));
  const bool act = lane < 60 && isl_row_on(L, ntx, nax, nty, nay, nxy);
  DevMan& mm = e.man[mi];
  J6 JX, JY, PX, PY;
  JX.l = JX.a = JY.l = JY.a = PX.l = PX.a = PY.l = PY.a = mk(0, 0, 0);
  float invk = 0.0f, bias = 0.0f, mu = 0.0f, lam = 0.0f, g = 0.0f, cap = 0.0f;
  if (act) {
    // this lane's row, set up by the lane itself (row_setup_k: the arithmetic of the row-setup phase)
    const int kind = blk == 2 ? 1 : (p < 4 ? 0 : 2);
    const int ba = blk == 1 ? Y : X;          // body a of the row's manifold
    ManPoint pt;
    pt.la = ld3(mm.la[slot]); pt.lb = ld3(mm.lb[slot]); pt.nrm = ld3(mm.nrm[slot]); pt.dist = mm.dist[slot]; pt.col = mm.col[slot];
    RowK o;
    row_setup_k(S, K, kind, ba, blk == 2 ? Y : -1, pt, k, mm.n, o);
    const v3 dir = o.dir, rxa = o.rxa;
    invk = o.invk; mu = o.mu; bias = k == 0 ? o.target : 0.0f; cap = o.cap;
    // (the impulses kept from the last substep, scaled as the row-setup phase scales them)
    lam = (k == 0 ? mm.ln[slot] : (k == 1 ? mm.lt1[slot] : mm.lt2[slot])) * c->warmstart;
    g = dot(dir, ld3(e.body[ba] + 7)) + dot(rxa, ld3(e.body[ba] + 10));
    const v3 pl = scale(dir, e.inv_mass[ba]), pa = o.aa;
    if (blk == 0) { JX.l = dir; JX.a = rxa; PX.l = pl; PX.a = pa; g -= o.vbc; }
    else if (blk == 1) { JY.l = dir; JY.a = rxa; PY.l = pl; PY.a = pa; g -= o.vbc; }
    else {
      const v3 rxb = o.rxb;
      g -= dot(dir, ld3(e.body[Y] + 7)) + dot(rxb, ld3(e.body[Y] + 10));
      JX.l = dir; JX.a = rxa; PX.l = pl; PX.a = pa;
      JY.l = mk(-dir.x, -dir.y, -dir.z); JY.a = mk(-rxb.x, -rxb.y, -rxb.z);
      const v3 t = scale(dir, e.inv_mass[Y]), ab = o.ab;
      PY.l = mk(-t.x, -t.y, -t.z); PY.a = mk(-ab.x, -ab.y, -ab.z);
    }
  }
  RV_PROF(25)
  // this lane's row of the Delassus matrix.  The P vectors of all rows go through LDS (the hull-vertex
  // scratch is dead here): every lane reads the same address -- a broadcast -- instead of six v_readlane
  // trips through the scalar unit per column
  float* cb = &S.s.u.r.wv[0][0][0][0];
  if (lane < 60) {
    float* o = cb + 12 * lane;
    o[0] = PX.l.x; o[1] = PX.l.y; o[2] = PX.l.z; o[3] = PX.a.x; o[4] = PX.a.y; o[5] = PX.a.z;
    o[6] = PY.l.x; o[7] = PY.l.y; o[8] = PY.l.z; o[9] = PY.a.x; o[10] = PY.a.y; o[11] = PY.a.z;
  }
  __syncthreads();
  float A[60];
#pragma unroll
  for (int s = 0; s < 60; ++s) {
    float a_ = 0.0f;
    if (isl_row_on(s, ntx, nax, nty, nay, nxy)) {
      const float* q = cb + 12 * s;
      if (s < 24) a_ = dotj(JX, mk(q[0], q[1], q[2]), mk(q[3], q[4], q[5]));
      else if (s < 48) a_ = dotj(JY, mk(q[6], q[7], q[8]), mk(q[9], q[10], q[11]));
      else a_ = dotj(JX, mk(q[0], q[1], q[2]), mk(q[3], q[4], q[5])) + dotj(JY, mk(q[6], q[7], q[8]), mk(q[9], q[10], q[11]));
    }
    A[s] = a_;
  }
  __syncthreads();          // (cb is written again by the epilogue)
  // warm start: the impulses kept from the last substep act first, in visiting order (X and Y slot
  // by slot, then the pair manifold)
#pragma unroll
  for (int t = 0; t < 24; ++t) {
    if (isl_row_on(t, ntx, nax, nty, nay, nxy)) g = g + A[t] * rdlane(lam, t);
    if (isl_row_on(24 + t, ntx, nax, nty, nay, nxy)) g = g + A[24 + t] * rdlane(lam, 24 + t);
  }
#pragma unroll
  for (int s = 48; s < 60; ++s) if (isl_row_on(s, ntx, nax, nty, nay, nxy)) g = g + A[s] * rdlane(lam, s);
  // normalised residual form (rows that are off: lam = 0, rr = 0, bounds 0, a row of zeros)
  float T = 0.0f;          // (rr: the change the row's impulse would take if it were unbounded)
  if (act) T = (bias - g) * invk;
#pragma unroll
  for (int s = 0; s < 60; ++s) A[s] = act ? -(A[s] * invk) : 0.0f;
  const float hi = act ? cap : 0.0f;
  if (!act) mu = 0.0f;
  const int iters = c->solver_iters;
  RV_PROF(26)
  // (v_med3 gives what the ternaries of the host version give for every finite input; the residual of a sweep is the
  // largest |lam - lam at its start| over the rows -- a row changes once per sweep -- compared through its bit pattern,
  // whose integer order is the order of the magnitudes)
  const int toli = __builtin_bit_cast(int, tol);
  const int stall = c->solver_stall; int besti = 0x7f800000, since = 0;
  const bool in_y = lane >= 24 && lane < 48;
  const int slot_xy = lane < 24 ? lane : (lane < 48 ? lane - 24 : -1);      // (slot t of X is lane t, of Y lane 24 + t)
  for (int it = 0; it < iters; ++it) {
    const float lam0 = lam;
    const int sq = opaque_i(slot_xy), lq = opaque_i(lane);      // (opaque once per sweep: see opaque_i)
    // the own rows of X and of Y do not couple (A[x][y] = 0), so slot t of X (lane t) and slot t of Y (lane 24 + t) are
    // solved in the same step; every row adds X's change first, then Y's -- the visiting order of the row list
#pragma unroll
    for (int pp = 0; pp < 8; ++pp) {
      if (!(isl_row_on(3 * pp, ntx, nax, nty, nay, nxy) || isl_row_on(24 + 3 * pp, ntx, nax, nty, nay, nxy))) continue;
      float lim = 0.0f;
#pragma unroll
      for (int kk = 0; kk < 3; ++kk) {
        const int s = 3 * pp + kk;
        const float nl = kk == 0 ? __builtin_amdgcn_fmed3f(lam + T, 0.0f, hi) : __builtin_amdgcn_fmed3f(lam + T, -lim, lim);
        const float d = nl - lam;
        if (sq == s) lam = nl;
        const float sdx = rdlane(d, s), sdy = rdlane(d, s + 24);
        if (kk == 0) { const float ml = mu * nl; const float lx = rdlane(ml, s), ly = rdlane(ml, s + 24); lim = in_y ? ly : lx; }
        T = rv_fma(A[s], sdx, T);
        T = rv_fma(A[s + 24], sdy, T);
      }
    }
#pragma unroll
    for (int pp = 16; pp < 20; ++pp) {
      if (!isl_row_on(3 * pp, ntx, nax, nty, nay, nxy)) continue;
      float lim = 0.0f;
#pragma unroll
      for (int kk = 0; kk < 3; ++kk) {
        const int s = 3 * pp + kk;
        const float nl = kk == 0 ? __builtin_amdgcn_fmed3f(lam + T, 0.0f, hi) : __builtin_amdgcn_fmed3f(lam + T, -lim, lim);
        const float d = nl - lam;
        if (lq == s) lam = nl;
        const float sd = rdlane(d, s);
        if (kk == 0) lim = rdlane(mu * nl, s);   // friction bound of the point: mu x its normal impulse
        T = rv_fma(A[s], sd, T);
      }
    }
    int m = __builtin_bit_cast(int, lam - lam0) & 0x7fffffff;
    m = grp_ror_max<8>(m); m = grp_ror_max<4>(m); m = grp_ror_max<2>(m); m = grp_ror_max<1>(m);
    const int r0 = __builtin_amdgcn_readlane(m, 0), r1 = __builtin_amdgcn_readlane(m, 16), r2 = __builtin_amdgcn_readlane(m, 32), r3 = __builtin_amdgcn_readlane(m, 48);
    int resi = r0 > r1 ? r0 : r1; resi = resi > r2 ? resi : r2; resi = resi > r3 ? resi : r3;
    if (tol > 0.0f ? resi < toli : false) break;
    // stalled (rv_config.solver_stall): no new smallest residual for that many sweeps
    if (stall > 0) { if (resi < besti) { besti = resi; since = 0; } else if (++since >= stall) break; }
  }
  RV_PROF(27)
  // impulses back to the manifolds; what every row adds to X and Y goes through LDS (the hull-vertex
  // scratch is dead here), one lane per velocity component sums it in row order
  if (act) { if (k == 0) mm.ln[slot] = lam; else if (k == 1) mm.lt1[slot] = lam; else mm.lt2[slot] = lam; }
  if (lane < 60) {
    // (rows that are off hold lam = 0 and zero P: they write zeros, which the sums below may add)
    float* o = cb + 12 * lane;
    o[0] = PX.l.x * lam; o[1] = PX.l.y * lam; o[2] = PX.l.z * lam; o[3] = PX.a.x * lam; o[4] = PX.a.y * lam; o[5] = PX.a.z * lam;
    o[6] = PY.l.x * lam; o[7] = PY.l.y * lam; o[8] = PY.l.z * lam; o[9] = PY.a.x * lam; o[10] = PY.a.y * lam; o[11] = PY.a.z * lam;
  }
  __syncthreads();
  if (lane < 12) {
    const int isy = lane >= 6, cc = lane - 6 * isy, bd = isy ? Y : X;
    const float* src = cb + 6 * isy + cc;
    float acc = e.body[bd][7 + cc];
    // its own 24 rows, then the 12 pair rows: loads first, sums in row order
    float t[24];
    const int base = isy ? 24 : 0;
#pragma unroll
    for (int s = 0; s < 24; ++s) t[s] = src[12 * (base + s)];
#pragma unroll
    for (int s = 0; s < 24; ++s) acc = acc + t[s];
#pragma unroll
    for (int s = 0; s < 12; ++s) t[s] = src[12 * (48 + s)];
#pragma unroll
    for (int s = 0; s < 12; ++s) acc = acc + t[s];
    e.body[bd][7 + cc] = acc;
  }
  __syncthreads();
}
// ALL islands that are ONE body -- on the table (or the ground), touched by the arm or not; no point with another
// awake body, no finger / limb / constraint rows -- solved TOGETHER: 16 lanes per body, lane 16 b + 3 p + k holds row k of
// table point p of body b AND row k of its arm point p (two rows per lane).  The islands do not couple, so each keeps its
// own iterates, residual and exit; what is shared is the instruction stream: one Delassus build with 12 (+ 3 x 12 when
// some body has arm points) columns per lane, one sweep loop whose row step serves four bodies.  Without deactivation (the
// reference's most likely semantics) all four bodies of the scene are such islands in most substeps; with it, the pushed
// body is one.  The rows are set up by the solver lanes themselves (row_setup_k's arithmetic): no Row records go through LDS.
//
// Round 5: the row step in NORMALISED RESIDUAL FORM.  rr = (bias - g) invk is the change a row's impulse would take without
// its bounds; a row step is  nl = med3(lam + rr, lo, hi); d = nl - lam; lam = nl; rr_r = fma(C_rs, d, rr_r) for every row r
// of the island, with C_rs = -(A_rs invk_r).  Same iterates in exact arithmetic as nl = clamp(lam + (bias - g) invk),
// g += A d (oracle: solve_rows), but a row step is v_add, v_med3, v_sub, the row_newbcast broadcast and ONE v_fma per row
// the lane holds, instead of thirteen instructions with three selects (tools/ubench/sweep_step.hip, one wave per SIMD:
// 92 -> 45 clocks per row step); rows that are absent are rows of zeros (lam = rr = bounds = C = 0: their step is d = 0), an
// island that has stopped leaves through exec, the residual of a sweep is |lam - lam at its start| (a row changes once per
// sweep), and every lane keeps the exit state (best residual, sweeps since) of its own island in VGPRs: ONE ballot per
// sweep instead of four readlanes and a scalar ladder.  (Tracking T = lam + rr would save the add; it rounds at the scale
// of lam instead of the residual's and, run to convergence in FP32, leaves 5e-6 m/s where this form leaves 3e-8:
// tests/test_independent_pin.py.  The ubench measured no difference in time -- the chain is not what bounds a lone wave.)
template <bool ARMS>
RV_DEV void singles_sweeps(const int r, const int ntmax, const int namax, const int iters, const int toli, const int stall_n, bool alive,
                           float& lamT, float& TT, const float (&CTT)[12], const float (&CTA)[12], const float hiT, const float muT,
                           float& lamA, float& TA, const float (&CAT)[12], const float (&CAA)[12], const float hiA, const float muA) {
  int best = 0x7f800000, since = 0;
  for (int it = 0; it < iters; ++it) {
    if (alive) {
      const float lamT0 = lamT, lamA0 = lamA;
      const int rq = opaque_i(r);      // (the row index, opaque once per sweep: see opaque_i)
#define RV_SSTEP_T(pp_, kk_) { \
      constexpr int s_ = 3 * pp_ + kk_; \
      const float nl = kk_ == 0 ? __builtin_amdgcn_fmed3f(lamT + TT, 0.0f, hiT) : __builtin_amdgcn_fmed3f(lamT + TT, -lim, lim); \
      const float d = nl - lamT; \
      if (rq == s_) lamT = nl; \
      if (kk_ == 0) lim = grp_bcast<s_>(muT * nl); \
      const float bd = grp_bcast<s_>(d); \
      TT = rv_fma(CTT[s_], bd, TT); \
      if (ARMS) TA = rv_fma(CAT[s_], bd, TA); }
#define RV_SSTEP_A(pp_, kk_) { \
      constexpr int s_ = 3 * pp_ + kk_; \
      const float nl = kk_ == 0 ? __builtin_amdgcn_fmed3f(lamA + TA, 0.0f, hiA) : __builtin_amdgcn_fmed3f(lamA + TA, -lim, lim); \
      const float d = nl - lamA; \
      if (rq == s_) lamA = nl; \
      if (kk_ == 0) lim = grp_bcast<s_>(muA * nl); \
      const float bd = grp_bcast<s_>(d); \
      TT = rv_fma(CTA[s_], bd, TT); \
      TA = rv_fma(CAA[s_], bd, TA); }
#define RV_POINT_T(pp_) if (pp_ < ntmax) { float lim = 0.0f; RV_SSTEP_T(pp_, 0) RV_SSTEP_T(pp_, 1) RV_SSTEP_T(pp_, 2) }
#define RV_POINT_A(pp_) if (pp_ < namax) { float lim = 0.0f; RV_SSTEP_A(pp_, 0) RV_SSTEP_A(pp_, 1) RV_SSTEP_A(pp_, 2) }
      RV_POINT_T(0) RV_POINT_T(1) RV_POINT_T(2) RV_POINT_T(3)
      if (ARMS) { RV_POINT_A(0) RV_POINT_A(1) RV_POINT_A(2) RV_POINT_A(3) }
#undef RV_POINT_A
#undef RV_POINT_T
#undef RV_SSTEP_A
#undef RV_SSTEP_T
      // the island's residual of this sweep: the largest |change| of one of its rows (bit patterns: their integer order is
      // the order of the magnitudes); every island stops on its own tolerance / stall count
      int m = __builtin_bit_cast(int, lamT - lamT0) & 0x7fffffff;
      if (ARMS) { const int ma_ = __builtin_bit_cast(int, lamA - lamA0) & 0x7fffffff; m = m > ma_ ? m : ma_; }
      m = grp_ror_max<8>(m); m = grp_ror_max<4>(m); m = grp_ror_max<2>(m); m = grp_ror_max<1>(m);
      const bool better = m < best;
      best = better ? m : best;
      since = better ? 0 : since + 1;
      if (m < toli || since >= stall_n) alive = false;        // (a tolerance of 0: never; stall_n = INT_MAX: no stall exit)
    }
    if (__builtin_amdgcn_ballot_w64(alive) == 0) break;
  }
}
RV_DEV void solve_singles(Shared& S, const Consts& K, const int smask, const int unrest) {
  DevEnv& e = S.e; const rv_config* c = K.cfg;
  const int lane = (int)threadIdx.x;
  const int b = lane >> 4, r = lane & 15;
  const int rr = r < 12 ? r : 11;
  const int p = rr / 3, k = rr - 3 * p;
  DevMan& mm = e.man[RV_TIDX(b)];
  DevMan& ma = e.man[RV_AIDX(b)];
  const bool mine = ((smask >> b) & 1) != 0;
  const int nt = mine ? mm.n : 0, na = mine ? ma.n : 0;      // (uniform within the 16-lane group)
  const bool actT = r < 12 && p < nt, actA = r < 12 && p < na;
  const int n0 = __builtin_amdgcn_readlane(nt, 0), n1 = __builtin_amdgcn_readlane(nt, 16),
            n2 = __builtin_amdgcn_readlane(nt, 32), n3 = __builtin_amdgcn_readlane(nt, 48);
  int ntmax = n0 > n1 ? n0 : n1; ntmax = ntmax > n2 ? ntmax : n2; ntmax = ntmax > n3 ? ntmax : n3;
  const int a0 = __builtin_amdgcn_readlane(na, 0), a1 = __builtin_amdgcn_readlane(na, 16),
            a2 = __builtin_amdgcn_readlane(na, 32), a3 = __builtin_amdgcn_readlane(na, 48);
  int namax = a0 > a1 ? a0 : a1; namax = namax > a2 ? namax : a2; namax = namax > a3 ? namax : a3;
  const bool arms = namax > 0;                             // (wave-uniform: some island of this solve has arm points)
  // ---- this lane's table row: row_setup_k() for a body - table point, row k only
  J6 JT, PT, JA, PA;
  JT.l = JT.a = PT.l = PT.a = JA.l = JA.a = PA.l = PA.a = mk(0, 0, 0);
  float invkT = 0.0f, biasT = 0.0f, muT = 0.0f, lamT = 0.0f, gT = 0.0f;
  float invkA = 0.0f, biasA = 0.0f, muA = 0.0f, lamA = 0.0f, gA = 0.0f, capA = 0.0f;
  if (actT) {
    const float dt = c->dt;
    const v3 la = ld3(mm.la[p]), d0 = ld3(mm.nrm[p]);
    const v3 wa = to_world_body(S, b, la);
    const v3 ra = sub(wa, ld3(e.body[b]));
    v3 d1, d2;
    plane_space(d0, &d1, &d2);
    const v3 vb_pt = mk(0.0f, 0.0f, 0.0f);
    const m3 iia = ldm(S.s.iinv[b]);
    const float ima = e.inv_mass[b];
    const v3 dk = k == 0 ? d0 : (k == 1 ? d1 : d2);
    const v3 rxa = cross(ra, dk);
    const v3 aa = mulv(iia, rxa);
    const float kk = ima + dot(rxa, aa);
    invkT = 1.0f / kk;
    const float vbc = dot(dk, vb_pt);
    const float dist = mm.dist[p];
    float target;
    if (dist > 0.0f) target = -dist / dt;
    else target = fminr(c->erp * fmaxr(-dist - c->slop, 0.0f) / dt, c->max_pushout);
    const float mub = body_below_table(e, c, b) ? c->ground_friction : e.mu_table;
    muT = e.friction[b] * mub;
    biasT = k == 0 ? target : 0.0f;
    // (the impulses kept from the last substep, scaled as the row-setup phase scales them)
    lamT = (k == 0 ? mm.ln[p] : (k == 1 ? mm.lt1[p] : mm.lt2[p])) * c->warmstart;
    gT = dot(dk, ld3(e.body[b] + 7)) + dot(rxa, ld3(e.body[b] + 10));
    JT.l = dk; JT.a = rxa; PT.l = scale(dk, ima); PT.a = aa;
    gT -= vbc;
  }
  if (arms) {
    if (actA) {
      // ... and its arm row (the pushed body): the arithmetic of the row-setup phase for an arm - body point
      ManPoint pt;
      pt.la = ld3(ma.la[p]); pt.lb = ld3(ma.lb[p]); pt.nrm = ld3(ma.nrm[p]); pt.dist = ma.dist[p]; pt.col = ma.col[p];
      RowK o;
      row_setup_k(S, K, 2, b, -1, pt, k, ma.n, o);
      invkA = o.invk; muA = o.mu; biasA = k == 0 ? o.target : 0.0f; capA = o.cap;
      lamA = (k == 0 ? ma.ln[p] : (k == 1 ? ma.lt1[p] : ma.lt2[p])) * c->warmstart;
      gA = dot(o.dir, ld3(e.body[b] + 7)) + dot(o.rxa, ld3(e.body[b] + 10));
      JA.l = o.dir; JA.a = o.rxa; PA.l = scale(o.dir, e.inv_mass[b]); PA.a = o.aa;
      gA -= o.vbc;
    }
  }
  RV_PROF(25)
  // ---- Delassus rows: what a unit impulse on row s of the SAME body does to this lane's rows.  The P vectors go through
  // LDS (the hull-vertex scratch is dead here): a lane reads the 12 (24) of its group
  float* cb = &S.s.u.r.wv[0][0][0][0];
  {
    float* o = cb + 12 * lane;
    o[0] = PT.l.x; o[1] = PT.l.y; o[2] = PT.l.z; o[3] = PT.a.x; o[4] = PT.a.y; o[5] = PT.a.z;
    if (arms) { o[6] = PA.l.x; o[7] = PA.l.y; o[8] = PA.l.z; o[9] = PA.a.x; o[10] = PA.a.y; o[11] = PA.a.z; }
  }
  __syncthreads();
  // (column s of the matrix, as this lane's rows see it; then in target form: C = -(A invk), C_ss = 1 - A_ss invk_s)
  float CTT[12], CTA[12], CAT[12], CAA[12];
  float lsT[12], lsA[12];
#define RV_WS(s_) lsT[s_] = grp_bcast<s_>(lamT); if (arms) lsA[s_] = grp_bcast<s_>(lamA); else lsA[s_] = 0.0f;
  RV_WS(0) RV_WS(1) RV_WS(2) RV_WS(3) RV_WS(4) RV_WS(5) RV_WS(6) RV_WS(7) RV_WS(8) RV_WS(9) RV_WS(10) RV_WS(11)
#undef RV_WS
#pragma unroll
  for (int s = 0; s < 12; ++s) {
    const float* q = cb + 12 * (16 * b + s);
    CTT[s] = dotj(JT, mk(q[0], q[1], q[2]), mk(q[3], q[4], q[5]));
    CTA[s] = 0.0f; CAT[s] = 0.0f; CAA[s] = 0.0f;
  }
  if (arms) {
#pragma unroll
    for (int s = 0; s < 12; ++s) {
      const float* q = cb + 12 * (16 * b + s);
      CTA[s] = dotj(JT, mk(q[6], q[7], q[8]), mk(q[9], q[10], q[11]));
      CAT[s] = dotj(JA, mk(q[0], q[1], q[2]), mk(q[3], q[4], q[5]));
      CAA[s] = dotj(JA, mk(q[6], q[7], q[8]), mk(q[9], q[10], q[11]));
    }
  }
  __syncthreads();          // (cb is written again by the epilogue)
  // warm start, in visiting order: the table rows, then the arm rows
#pragma unroll
  for (int s = 0; s < 12; ++s) if (s / 3 < nt) { gT = gT + CTT[s] * lsT[s]; if (arms) gA = gA + CAT[s] * lsT[s]; }
  if (arms) {
#pragma unroll
    for (int s = 0; s < 12; ++s) if (s / 3 < na) { gT = gT + CTA[s] * lsA[s]; gA = gA + CAA[s] * lsA[s]; }
  }
  // normalised residual form (TT / TA hold rr).  Rows that are absent: lam = 0, rr = 0, bounds 0, a row of zeros
  float TT = 0.0f, TA = 0.0f;
  if (actT) TT = (biasT - gT) * invkT;
  if (actA) TA = (biasA - gA) * invkA;
#pragma unroll
  for (int s = 0; s < 12; ++s) CTT[s] = actT ? -(CTT[s] * invkT) : 0.0f;
  if (arms) {
#pragma unroll
    for (int s = 0; s < 12; ++s) {
      CTA[s] = actT ? -(CTA[s] * invkT) : 0.0f;
      CAT[s] = actA ? -(CAT[s] * invkA) : 0.0f;
      CAA[s] = actA ? -(CAA[s] * invkA) : 0.0f;
    }
  }
  const float hiT = actT ? 1e30f : 0.0f, hiA = actA ? capA : 0.0f;
  if (!actT) muT = 0.0f;
  if (!actA) muA = 0.0f;
  RV_PROF(26)
  // every island (= body) stops on its own tolerance: solver_tol_rest when the body is at rest (tol_of)
  const int toli = __builtin_bit_cast(int, tol_of(c, (unrest >> b) & 1));
  const int stall_n = c->solver_stall > 0 ? c->solver_stall : 0x7fffffff;
  const bool alive0 = mine && (nt + na) > 0;
  if (arms) singles_sweeps<true>(r, ntmax, namax, c->solver_iters, toli, stall_n, alive0, lamT, TT, CTT, CTA, hiT, muT, lamA, TA, CAT, CAA, hiA, muA);
  else singles_sweeps<false>(r, ntmax, namax, c->solver_iters, toli, stall_n, alive0, lamT, TT, CTT, CTA, hiT, muT, lamA, TA, CAT, CAA, hiA, muA);
  RV_PROF(27)
  // impulses back to the manifolds; the body velocities are rebuilt in row order through LDS
  if (actT) { if (k == 0) mm.ln[p] = lamT; else if (k == 1) mm.lt1[p] = lamT; else mm.lt2[p] = lamT; }
  if (actA) { if (k == 0) ma.ln[p] = lamA; else if (k == 1) ma.lt1[p] = lamA; else ma.lt2[p] = lamA; }
  {
    float* o = cb + 12 * lane;
    o[0] = PT.l.x * lamT; o[1] = PT.l.y * lamT; o[2] = PT.l.z * lamT; o[3] = PT.a.x * lamT; o[4] = PT.a.y * lamT; o[5] = PT.a.z * lamT;
    if (arms) { o[6] = PA.l.x * lamA; o[7] = PA.l.y * lamA; o[8] = PA.l.z * lamA; o[9] = PA.a.x * lamA; o[10] = PA.a.y * lamA; o[11] = PA.a.z * lamA; }
  }
  __syncthreads();
  if (r < 6 && (nt + na) > 0) {
    float acc = e.body[b][7 + r];
    float t[12];
#pragma unroll
    for (int s = 0; s < 12; ++s) t[s] = cb[12 * (16 * b + s) + r];
#pragma unroll
    for (int s = 0; s < 12; ++s) acc = acc + t[s];
    if (arms) {
#pragma unroll
      for (int s = 0; s < 12; ++s) t[s] = cb[12 * (16 * b + s) + 6 + r];
#pragma unroll
      for (int s = 0; s < 12; ++s) acc = acc + t[s];
    } else acc = acc + 0.0f;        // (the twelve -- empty -- arm rows of the row list: +0.0 each)
    e.body[b][7 + r] = acc;
  }
  __syncthreads();
}
// The force-limited gripper in impulse space (rv_config.finger_dynamics; at most one awake body X, or
// none: X < 0).  Layout: lanes 0..23 the rows of X (table points 0..3, arm points 0..3, x 3 rows), lanes
// 24 / 25 the POSITION_CONTROL motor rows of the two finger joints (bullet_physics.py:1061-1104).  The
// finger joints are DOFs of mass finger_mass: a contact row on a finger pad has the Jacobian entry jf on
// its finger's velocity, so A_rs gains jf_r jf_s / m for two rows on the same finger, a motor row couples
// with the rows of its finger through jf / m, and with itself through 1 / m.  Visiting order: the contact
// rows, then the two motor rows; the motor impulse stays within +-finger_max_force dt minus what the
// joint motors of the light part already spent on the free motion.  Same arithmetic as solve_rows(fing).
// LIMB (rv_config.limb_dynamics): lanes 26..32 are the motor rows of the seven limb joints; the rows of the
// arm manifold and these motor rows are 'limb rows' with a joint-space Jacobian ja (a motor row: e_j) and the
// velocity change per unit impulse pj = M^-1 ja^T (a motor row: column j of M^-1), both prepared in LDS by
// limb_prepare; the Delassus entry of two limb rows gains ja_r . pj_s.
template <bool LIMB>
RV_DEV void solve_island_fingers_t(Shared& S, const Consts& K, const int X, const int with_fingers) {
  DevEnv& e = S.e; const rv_config* c = K.cfg; const rv_arm* arm = K.arm;
  const int lane = (int)threadIdx.x;
  const int Xc = X >= 0 ? X : 0;
  const int ntx = X >= 0 ? __builtin_amdgcn_readfirstlane(e.man[RV_TIDX(Xc)].n) : 0;
  const int nax = X >= 0 ? __builtin_amdgcn_readfirstlane(e.man[RV_AIDX(Xc)].n) : 0;
  const float mf = c->finger_mass, imf = 1.0f / c->finger_mass, fdt = c->finger_max_force * c->dt;
  const int L = lane < 24 ? lane : 23;
  const int p = L / 3, k = L - 3 * p, slot = p & 3;
  const int mi = p < 4 ? RV_TIDX(Xc) : RV_AIDX(Xc);
  const bool act = lane < 24 && (p < 4 ? p < ntx : p - 4 < nax);
  const bool motor = with_fingers && (lane == 24 || lane == 25);
  const bool lmotor = LIMB && lane >= 26 && lane < 26 + RV_NLIMB;
  const int lj = lmotor ? lane - 26 : 0;
  constexpr int NA = LIMB ? 26 + RV_NLIMB : 26, CS = LIMB ? 16 : 8;
  const int mid = lane == 25 ? 1 : 0;
  DevMan& mm = e.man[mi];
  J6 JX, PX;
  JX.l = JX.a = PX.l = PX.a = mk(0, 0, 0);
  float invk = 0.0f, bias = 0.0f, mu = 0.0f, lam = 0.0f, g = 0.0f, cap = 1e30f, jf = 0.0f, pf = 0.0f, lo = 0.0f, hi = 0.0f;
  int fi = -1;
  const float qf0a = e.qd[RV_NLIMB], qf0b = e.qd[RV_NLIMB + 1];
  float row_invk = 0.0f;   // (1 / effective mass of the row before the limb terms: limb_prepare's input)
  v3 row_dir = mk(0, 0, 0);
  if (act) {
    // this lane's row, set up by the lane itself (row_setup_k: the arithmetic of the row-setup phase)
    ManPoint pt;
    pt.la = ld3(mm.la[slot]); pt.lb = ld3(mm.lb[slot]); pt.nrm = ld3(mm.nrm[slot]); pt.dist = mm.dist[slot]; pt.col = mm.col[slot];
    RowK o;
    row_setup_k(S, K, p < 4 ? 0 : 2, Xc, -1, pt, k, mm.n, o);
    const v3 dir = o.dir, rxa = o.rxa;
    invk = o.invk; mu = o.mu; bias = k == 0 ? o.target : 0.0f; cap = o.cap;
    row_invk = o.invk; row_dir = dir;
    lam = (k == 0 ? mm.ln[slot] : (k == 1 ? mm.lt1[slot] : mm.lt2[slot])) * c->warmstart;
    g = dot(dir, ld3(e.body[Xc] + 7)) + dot(rxa, ld3(e.body[Xc] + 10));
    JX.l = dir; JX.a = rxa; PX.l = scale(dir, e.inv_mass[Xc]); PX.a = o.aa;
    g -= o.vbc;
    fi = o.fidx;
    if (fi >= 0) { jf = o.jf; pf = jf * imf; g += jf * (fi == 0 ? qf0a : qf0b); }
  }
  (void)row_invk; (void)row_dir;
  if (motor) {
    const float i0 = mf * S.s.fing_dv[mid];
    g = (mid == 0 ? qf0a : qf0b) - S.s.fing_vt[mid]; invk = mf; jf = 1.0f; pf = imf; fi = mid;
    lo = -fdt - i0; hi = fdt - i0;
  }
  // limb rows
  const bool la = LIMB && ((act && p >= 4) || lmotor);
  const int lrow = lmotor ? 12 + lj : slot * 3 + k;
  float ja[RV_NLIMB], pj[RV_NLIMB];
#pragma unroll
  for (int x = 0; x < RV_NLIMB; ++x) { ja[x] = 0.0f; pj[x] = 0.0f; }
  if (LIMB) {
    if (la) {
#pragma unroll
      for (int x = 0; x < RV_NLIMB; ++x) { ja[x] = S.s.lJa[lrow][x]; pj[x] = S.s.lMiJ[lrow][x]; }
    }
    if (act && p >= 4) {
      invk = S.s.linvk[lrow];
      float t = 0.0f;
#pragma unroll
      for (int x = 0; x < RV_NLIMB; ++x) t = t + ja[x] * (-S.s.limb_dv[x]);     // (the solve starts from the velocity before the motor step)
      g += t;
    }
    if (lmotor) {
      g = (-S.s.limb_dv[lj]) - S.s.ltgt[lj]; invk = 1.0f / S.s.lA[lj][RV_NLIMB + lj];
      lo = S.s.llo[lj]; hi = S.s.lhi[lj];
    }
  }
  // this lane's row of the Delassus matrix: columns 0..23 the contact rows, 24 / 25 the motor rows
  float A[NA];
#pragma unroll
  for (int s = 0; s < 24; ++s) {
    float a_ = 0.0f;
    const int ps = s / 3;
    if (ps < 4 ? ps < ntx : ps - 4 < nax) {
      a_ = dotj(JX, rdlane3(PX.l, s), rdlane3(PX.a, s));
      const int fs = __builtin_amdgcn_readlane(fi, s);
      const float pfs = rdlane(pf, s);
      if (motor) a_ = fs == fi ? pfs : 0.0f;                    // a motor row sees the rows of its finger through jf_s / m
      else if (fi >= 0 && fs == fi) a_ = a_ + jf * pfs;
      if (LIMB) {
        if (lmotor) a_ = 0.0f;
        if (ps >= 4) {
          const float* pjs = S.s.lMiJ[s - 12];
          float t = 0.0f;
#pragma unroll
          for (int x = 0; x < RV_NLIMB; ++x) t = t + ja[x] * pjs[x];
          if (la) a_ = a_ + t;
        }
      }
    }
    A[s] = a_;
  }
#pragma unroll
  for (int m = 0; m < 2; ++m) A[24 + m] = (fi == m) ? jf * imf : 0.0f;   // (motor row m on itself: 1 x 1 / m)
  if (LIMB) {
#pragma unroll
    for (int j = 0; j < RV_NLIMB; ++j) {
      const float* pjs = S.s.lMiJ[12 + j];
      float t = 0.0f;
#pragma unroll
      for (int x = 0; x < RV_NLIMB; ++x) t = t + ja[x] * pjs[x];
      A[26 + j] = la ? 0.0f + t : 0.0f;
    }
  }
  // warm start: the contact impulses kept from the last substep (the motor rows start from zero)
#pragma unroll
  for (int s = 0; s < 24; ++s) { const int ps = s / 3; if (ps < 4 ? ps < ntx : ps - 4 < nax) g = g + A[s] * rdlane(lam, s); }
  // normalised residual form of the row step (see solve_singles): rr = (bias - g) invk (kept in T), C_rs = -(A_rs invk_r)
  const bool rowon = act || motor || lmotor;
  float T = 0.0f;
  if (rowon) T = (bias - g) * invk;
#pragma unroll
  for (int s = 0; s < NA; ++s) A[s] = rowon ? -(A[s] * invk) : 0.0f;
  const float capn = act ? cap : 0.0f;
  if (!act) mu = 0.0f;
  const int iters = c->solver_iters; const float tol = c->solver_tol;
  const int toli = __builtin_bit_cast(int, tol);
  const int stall = c->solver_stall; int besti = 0x7f800000, since = 0;
  for (int it = 0; it < iters; ++it) {
    const float lam0 = lam;
    const int lq = opaque_i(lane);      // (opaque once per sweep: see opaque_i)
#pragma unroll
    for (int pp = 0; pp < 8; ++pp) {
      if (!(pp < 4 ? pp < ntx : pp - 4 < nax)) continue;
      float lim = 0.0f;
#pragma unroll
      for (int kk = 0; kk < 3; ++kk) {
        const int s = 3 * pp + kk;
        const float nl = kk == 0 ? __builtin_amdgcn_fmed3f(lam + T, 0.0f, capn) : __builtin_amdgcn_fmed3f(lam + T, -lim, lim);
        const float d = nl - lam;
        if (lq == s) lam = nl;
        const float sd = rdlane(d, s);
        if (kk == 0) lim = rdlane(mu * nl, s);
        T = rv_fma(A[s], sd, T);
      }
    }
#pragma unroll
    for (int m = 0; m < 2; ++m) {
      const int s = 24 + m;
      const float nl = __builtin_amdgcn_fmed3f(lam + T, lo, hi);
      const float d = nl - lam;
      if (lq == s) lam = nl;
      T = rv_fma(A[s], rdlane(d, s), T);
    }
    if (LIMB) {
#pragma unroll
      for (int j = 0; j < RV_NLIMB; ++j) {
        const int s = 26 + j;
        const float nl = __builtin_amdgcn_fmed3f(lam + T, lo, hi);
        const float d = nl - lam;
        if (lq == s) lam = nl;
        T = rv_fma(A[s], rdlane(d, s), T);
      }
    }
    // the residual of the sweep: the largest |lam - lam at its start| over the rows (lanes 0 .. NA - 1: three 16-lane rows)
    int m_ = __builtin_bit_cast(int, lam - lam0) & 0x7fffffff;
    m_ = grp_ror_max<8>(m_); m_ = grp_ror_max<4>(m_); m_ = grp_ror_max<2>(m_); m_ = grp_ror_max<1>(m_);
    const int r0 = __builtin_amdgcn_readlane(m_, 0), r1 = __builtin_amdgcn_readlane(m_, 16), r2 = __builtin_amdgcn_readlane(m_, 32);
    int resi = r0 > r1 ? r0 : r1; resi = resi > r2 ? resi : r2;
    if (tol > 0.0f ? resi < toli : false) break;
    // stalled (rv_config.solver_stall): no new smallest residual for that many sweeps
    if (stall > 0) { if (resi < besti) { besti = resi; since = 0; } else if (++since >= stall) break; }
  }
  // impulses back to the manifolds; body, finger and limb velocities rebuilt in row order through LDS
  float* cb = &S.s.u.r.wv[0][0][0][0];
  if (act) { if (k == 0) mm.ln[slot] = lam; else if (k == 1) mm.lt1[slot] = lam; else mm.lt2[slot] = lam; }
  if (lane < NA) {
    float* o = cb + CS * lane;
    o[0] = PX.l.x * lam; o[1] = PX.l.y * lam; o[2] = PX.l.z * lam; o[3] = PX.a.x * lam; o[4] = PX.a.y * lam; o[5] = PX.a.z * lam;
    o[6] = pf * lam; o[7] = __builtin_bit_cast(float, fi);
    if (LIMB) {
#pragma unroll
      for (int x = 0; x < RV_NLIMB; ++x) o[8 + x] = pj[x] * lam;
    }
  }
  __syncthreads();
  if (lane < 6 && X >= 0) {
    float acc = e.body[Xc][7 + lane];
    float t[24];
#pragma unroll
    for (int s = 0; s < 24; ++s) t[s] = cb[CS * s + lane];
#pragma unroll
    for (int s = 0; s < 24; ++s) acc = acc + t[s];
    e.body[Xc][7 + lane] = acc;
  } else if (with_fingers && (lane == 8 || lane == 9)) {
    const int m = lane - 8;
    float qd = m == 0 ? qf0a : qf0b;
    for (int s = 0; s < 24; ++s) {
      const int ps = s / 3;
      if (!(ps < 4 ? ps < ntx : ps - 4 < nax)) continue;
      if (__builtin_bit_cast(int, cb[CS * s + 7]) == m) qd = qd + cb[CS * s + 6];
    }
    qd = qd + cb[CS * (24 + m) + 6];
    const int j = RV_NLIMB + m;
    float qn = e.q[j] + (qd - S.s.fing_qd0[m]) * c->dt;
    if (qn < arm->q_lo[j]) { qn = arm->q_lo[j]; qd = 0.0f; }
    if (qn > arm->q_hi[j]) { qn = arm->q_hi[j]; qd = 0.0f; }
    e.q[j] = qn; e.qd[j] = qd;
  } else if (LIMB && lane >= 16 && lane < 16 + RV_NLIMB) {
    // the limb moves with the solved velocity: dq = dq0 + sum over the limb rows, in row order
    const int x = lane - 16;
    float dq = -S.s.limb_dv[x];
    for (int s = 12; s < 24; ++s) {
      if (!((s / 3) - 4 < nax)) continue;
      dq = dq + cb[CS * s + 8 + x];
    }
    for (int j = 0; j < RV_NLIMB; ++j) dq = dq + cb[CS * (26 + j) + 8 + x];
    float qd = S.s.limb_qd0[x] + dq;
    float qn = e.q[x] + dq * c->dt;
    if (qn < arm->q_lo[x]) { qn = arm->q_lo[x]; qd = 0.0f; }
    if (qn > arm->q_hi[x]) { qn = arm->q_hi[x]; qd = 0.0f; }
    e.q[x] = qn; e.qd[x] = qd;
  }
  if (LIMB && lane == 63) S.s.kin_fresh = 0;
  __syncthreads();
}
RV_DEV void solve_island_fingers(Shared& S, const Consts& K, const int X, const int with_fingers, const int limb) {
  if (limb) solve_island_fingers_t<true>(S, K, X, with_fingers);
  else solve_island_fingers_t<false>(S, K, X, with_fingers);
}
#else
#define RV_EMU_SECTION 2
#include "../../tests/emu/rv_emu_hooks.h"      // host lane emulation (test scaffolding; not compiled into the product)
#undef RV_EMU_SECTION
#endif

// ---------------------------------------------------- Simulator.step -----
// One dt of Simulator.step (simulator.py:94-103): the arm's
// ControllableBody.update, then the physics step, then num_steps += 1.
// K.stop_after (profiling hook, 0 = off): return after a given phase group so that
// per-phase costs can be read off as differences (tools/prof_phases.sh)
// ---- arm phase groups (shared by the regular substep and the coasting path) ----
// ControllableBody.update + joint motors.  with_lq: also the local joint quaternions
// of the new positions and the per-joint "moving" flags (input of the FK phases).
// count_step: lane 0 advances the step counters (coasting substeps end here).
RV_DEV void arm_motor_phases(Shared& S, const Consts& K, const int with_lq, const int count_step) {
  const rv_config* c = K.cfg;
  const rv_arm* arm = K.arm;
  control_update_phases(S, K);
  // joint motors of the kinematic arm, (a) per joint: the raw
  // commanded velocity and the factor that would bring it within its limit
#if RV_ON_DEVICE
  {
    // (a) + (b) in one phase: the common scale is a 16-lane DPP min over the limb lanes (the
    // factors are positive floats, ordered like their bit patterns)
    const int lane = (int)threadIdx.x;
    const DevEnv& e0 = S.e; const int j0 = lane < RV_NJ ? lane : RV_NJ - 1;
    float vd0 = 0.0f, ratio0 = 1.0f;
    if (e0.motor_on[j0]) {
      vd0 = e0.motor_kp[j0] * (e0.motor_q[j0] - e0.q[j0]) * (1.0f / c->dt);
      float raw = fabsr(vd0);
      if (j0 < RV_NLIMB && raw > e0.vmax_cmd[j0]) ratio0 = e0.vmax_cmd[j0] / raw;
    }
    if (lane >= RV_NLIMB) ratio0 = 1.0f;
    int r = __builtin_bit_cast(int, ratio0);      // (<= 1: vmax / raw only where raw > vmax)
    r = min(r, __builtin_amdgcn_update_dpp(0, r, 0x128, 0xf, 0xf, false));
    r = min(r, __builtin_amdgcn_update_dpp(0, r, 0x124, 0xf, 0xf, false));
    r = min(r, __builtin_amdgcn_update_dpp(0, r, 0x122, 0xf, 0xf, false));
    r = min(r, __builtin_amdgcn_update_dpp(0, r, 0x121, 0xf, 0xf, false));
    if (lane < RV_NJ) { S.s.vdraw[lane] = vd0; S.s.ratio[lane] = __builtin_bit_cast(float, r); }
  }
#else
#define RV_EMU_SECTION 3
#include "../../tests/emu/rv_emu_hooks.h"      // host lane emulation (test scaffolding; not compiled into the product)
#undef RV_EMU_SECTION
#endif
  // (b) limb joints move synchronised: one common scale (the smallest factor)
  // keeps every commanded velocity within its limit, so the path is a straight
  // line in joint space.
  RV_LANES_BEGIN
    if (lane < RV_NJ) {
      DevEnv& e = S.e; int j = lane; float dt = c->dt;
      float sync = 1.0f;
#if RV_ON_DEVICE
      sync = S.s.ratio[j];      // (its own store: the common scale)
#else
#pragma unroll
      for (int k = 0; k < RV_NLIMB; ++k) sync = fminr(sync, S.s.ratio[k]);
#endif
      float vd = 0.0f;
      if (e.motor_on[j]) {
        vd = S.s.vdraw[j];
        if (j < RV_NLIMB) vd = vd * sync;
        vd = fclamp_pm(vd, e.vmax_cmd[j]);
      }
      float dv = fclamp_pm(vd - e.qd[j], arm->a_max[j] * dt);
      float qd = e.qd[j] + dv;
      float qn = e.q[j] + qd * dt;
      if (qn < arm->q_lo[j]) { qn = arm->q_lo[j]; qd = 0.0f; }
      if (qn > arm->q_hi[j]) { qn = arm->q_hi[j]; qd = 0.0f; }
      if (j >= RV_NLIMB) { S.s.fing_dv[j - RV_NLIMB] = dv; S.s.fing_vt[j - RV_NLIMB] = vd; S.s.fing_qd0[j - RV_NLIMB] = qd; }
      else if (c->limb_dynamics) { S.s.limb_dv[j] = dv; S.s.limb_vt[j] = vd; S.s.limb_qd0[j] = qd; }
      if (with_lq) S.s.jchg[j] = (qn != e.q[j]) || (qd != 0.0f);   // did the joint state change at all?
      e.q[j] = qn; e.qd[j] = qd;
      if (with_lq) {
        S.s.jmoving[j] = fabsr(qd) > 1e-3f;
        if (j < RV_NLIMB) stq(S.s.lq[j], joint_local_quat(arm, j, qn));
      }
      if (count_step == 1) {
        S.s.jtravel[j] += fabsr(qd) * dt;      // coasting: path length of the joint
        if (j == 0) { e.sim_steps++; e.substeps_last++; }
      }
      if (count_step == 2) S.s.ftravel[j] += fabsr(qd) * dt;   // an "arm far" substep: the boxes are not recomputed
    }
  RV_LANES_END
}
// local joint quaternions / moving flags from the stored joint state (kinematics refresh)
RV_DEV void arm_lq_phase(Shared& S, const Consts& K) {
  RV_LANES_BEGIN
    if (lane < RV_NJ) {
      const DevEnv& e = S.e; int j = lane;
      S.s.jmoving[j] = fabsr(e.qd[j]) > 1e-3f;
      if (j < RV_NLIMB) stq(S.s.lq[j], joint_local_quat(K.arm, j, e.q[j]));
    }
  RV_LANES_END
}
// forward kinematics from the local joint quaternions S.s.lq
RV_DEV void arm_fk_phases(Shared& S, const Consts& K) {
  const rv_arm* arm = K.arm;
  // (a) lane 0: the serial product of the joint quaternions
  RV_LANES_BEGIN
    if (lane == 0) {
      DevEnv& e = S.e;
      q4 pq = ldq(arm->base_quat);
#pragma unroll
      for (int i = 0; i < RV_NLIMB; ++i) { pq = qmul(pq, ldq(S.s.lq[i])); stq(e.fquat[i], pq); }
      stq(e.fquat[7], qmul(pq, ldq(arm->jquat[7])));
    } else if (lane == 1) {
      int mv = 0;
#pragma unroll
      for (int j = 0; j < RV_NJ; ++j) mv |= S.s.jmoving[j];
      S.s.arm_moving = mv;
    }
  RV_LANES_END
  // (b) lanes 0-7: each link's offset rotated into the world, joint axis, rotation matrix
  RV_LANES_BEGIN
    DevEnv& e = S.e;
    if (lane <= RV_NLIMB) {
      int i = lane;
      q4 par = ldq(i == 0 ? arm->base_quat : e.fquat[i == 0 ? 0 : i - 1]);
      st3(S.s.rvec[i], qrotv(par, ld3(arm->jpos[i])));
      q4 qi = ldq(e.fquat[i]);
      if (i < RV_NLIMB) st3(S.s.axis[i], qaxis_z(qi));
      stm(S.s.frot[i], qmat(qi));
    } else if (lane == 8 || lane == 9) {
      q4 q7 = ldq(e.fquat[7]);
      stm(S.s.frot[lane], qmat(q7));
      stq(e.fquat[lane], q7);
    }
  RV_LANES_END
  // (c) lanes 0-9: frame origins = the running sum of the offsets, in chain order
  // (bit-identical to the serial chain); the two finger frames slide along hand y
  RV_LANES_BEGIN
    if (lane < RV_NFRAME) {
      DevEnv& e = S.e;
      const int n = lane < 8 ? lane : 7;
      v3 p = ld3(arm->base_pos);
#pragma unroll
      for (int k = 0; k < 8; ++k) if (k <= n) p = add(p, ld3(S.s.rvec[k]));
      if (lane >= 8) {
        int k = lane - 8;
        v3 yax = mk(S.s.frot[7][1], S.s.frot[7][4], S.s.frot[7][7]);
        float off = arm->finger_y0[k] + e.q[7 + k];
        p = madd(p, yax, off);
      }
      st3(e.fpos[lane], p);
    }
  RV_LANES_END
}
// collider boxes in the world: centres, AABBs (centre +- |R| half: no vertices needed) and the arm-table gate.
// The eight vertices of a box are only the input of a convex query: arm_box_vertices() computes them for the
// boxes that are about to take part in one (a wake query, an arm - body or an arm - table query).
RV_DEV void arm_collider_phases(Shared& S, const Consts& K, const int arm_on) {
  const rv_config* c = K.cfg;
  const rv_arm* arm = K.arm;
  RV_LANES_BEGIN
    if (lane >= 16 && lane < 16 + RV_MAXB) { S.s.wake[lane - 16] = 0; S.s.bnear[lane - 16] = 0; }
    if (lane == 24) S.s.near_any = 0;
    // (no arm in this substep -- the bodies settling after a reset: the boxes do not travel.  The wake test reads the travel
    // whatever arm_on is and keeps its distance bounds; in a launch that BEGINS with a reset nothing had written it yet, and a
    // negative left-over in LDS turned the bounds into "far away for ever" -- found by the poisoned-LDS build, round 5)
    if (!arm_on && lane < RV_NCOL) S.s.coltravel[lane] = 0.0f;
    if (arm_on && lane < RV_NCOL) {
      const int col = lane; const int f = arm->col_frame[col];
      const v3 cc = ld3(arm->col_center[col]), hh = ld3(arm->col_half[col]);
      const float* R = S.s.frot[f];
      const v3 cw = add(ld3(S.e.fpos[f]), mulv(R, cc));
      st3(S.s.colc[col], cw);
      const float colr = fsqrtr(hh.x * hh.x + hh.y * hh.y + hh.z * hh.z) + c->margin;
      S.s.colr[col] = colr;
      S.s.colflag[col] = 0;
      const float cwa[3] = {cw.x, cw.y, cw.z};
      float lo_z = 0.0f;
#pragma unroll
      for (int x = 0; x < 3; ++x) {
        const float ext = fabsr(R[3 * x]) * hh.x + fabsr(R[3 * x + 1]) * hh.y + fabsr(R[3 * x + 2]) * hh.z;
        const float lo = cwa[x] - ext;
        S.s.colmin[col][x] = lo; S.s.colmax[col][x] = cwa[x] + ext;
        if (x == 2) lo_z = lo;
      }
      // can the box be within the contact-query distance of the table?  (the two rejection tests of the
      // arm-table detector)
      v3 tc = mk(c->table_center[0], c->table_center[1], S.e.table_z - 0.5f * c->table_thickness);
      v3 th = mk(c->table_half[0], c->table_half[1], 0.5f * c->table_thickness);
      float r = colr + brk_col(arm, c, col);
      S.s.atflag[col] = (!(lo_z - S.e.table_z - c->margin >= c->contact_query_dist) &&
                         sphere_box_dist2(cw, tc, th) < r * r) ? 1 : 0;
      // how far a vertex of this box can have moved in this substep (joint travel x reach)
      {
        const DevEnv& e = S.e; const int fl = f < 7 ? f : 7;
        float tr = 0.0f, reach = S.s.colext[col];
        for (int j = fl; j >= 0; --j) {
          if (j < RV_NLIMB) tr += reach * (fabsr(e.qd[j]) * c->dt);
          reach += S.s.jlen[j];
        }
        if (f >= 8) tr += fabsr(e.qd[f - 1]) * c->dt;
        S.s.coltravel[col] = tr * 1.02f + 1e-7f;
      }
    }
    if (lane == 63) { S.s.kin_fresh = arm_on; S.s.clr_valid = 0; S.s.far_valid = arm_on; }   // left-over clearances are for coasting chains only
    if (lane >= 32 && lane < 32 + RV_NJ) S.s.ftravel[lane - 32] = 0.0f;
  RV_LANES_END
}
// the eight world vertices of collider box col (lane k < 8 of the caller's choice)
RV_DEV void arm_box_vertex(Shared& S, const Consts& K, const int col, const int k) {
  const rv_arm* arm = K.arm;
  const int f = arm->col_frame[col];
  const v3 cc = ld3(arm->col_center[col]), hh = ld3(arm->col_half[col]);
  const v3 l = mk(cc.x + ((k & 1) ? hh.x : -hh.x), cc.y + ((k & 2) ? hh.y : -hh.y), cc.z + ((k & 4) ? hh.z : -hh.z));
  st3(S.s.colv[col][k], add(ld3(S.e.fpos[f]), mulv(S.s.frot[f], l)));
}

#ifdef RV_EMU_COUNT
static long rv_emu_coasted = 0;
#endif
// ---- coasting: substeps that provably cannot touch anything -------------------
// While every body sleeps and every collider box is farther from the table and
// from the bodies than the arm can travel in m substeps, those m substeps reduce
// to control + joint motors; FK / colliders are refreshed once at the end.  The
// bound is rigorous (joint travel from |qd|, v_max and a_max; vertex travel <=
// sum of joint travel x reach), so the result is bit-identical to stepping.
// Returns m (0: step normally).  Needs fresh kinematics (S.s.kin_fresh).
// clearances of collider box col on fresh kinematics: how far the box is from the contact range
// of the table and of the nearest body
RV_DEV void coast_measure_clearances(Shared& S, const Consts& K, const int col) {
  const rv_config* c = K.cfg; const DevEnv& e = S.e;
  v3 tc = mk(c->table_center[0], c->table_center[1], e.table_z - 0.5f * c->table_thickness);
  v3 th = mk(c->table_half[0], c->table_half[1], 0.5f * c->table_thickness);
  const float zc = S.s.colmin[col][2] - e.table_z - c->margin - c->contact_query_dist;
  const float sc = fsqrtr(sphere_box_dist2(ld3(S.s.colc[col]), tc, th)) - (S.s.colr[col] + brk_col(K.arm, c, col));
  const float tclear = zc > sc ? zc : sc;                 // either test rejecting is enough
  float bclear = 1e30f;
  for (int b = 0; b < RV_MAXB; ++b) {
    if (!body_present(e, b) || body_static(e, b)) continue;      // (the arm does not wake a static body)
    float d = fsqrtr(aabb_aabb_dist2(e.baabb[b], e.baabb[b] + 3, S.s.colmin[col], S.s.colmax[col])) - (wake_range(e, K.arm, c, b, col) + 2.0f * c->margin);
    d = fmaxr(d, S.s.sep[b][col]);   // the distance bound left by the last wake query, if better
    bclear = fminr(bclear, d);
  }
  S.s.clr_t[col] = tclear; S.s.clr_b[col] = bclear;
  // Levers for the travel bound of the fused loop.  ccoef (the length of the chain from the joint
  // to the box, valid in any configuration) is what a folded arm never reaches: the distance R_j
  // from joint j's origin to the farthest point of the box, measured now, changes only through the
  // joints between j and the box, i.e. by no more than the box itself travels -- and the loop never
  // lets that exceed the clearance measured here.  So R_j + clearance bounds the lever for as long
  // as these clearances are in use.
  {
    const rv_arm* arm = K.arm;
    const int f = arm->col_frame[col]; const int fl = f < RV_NLIMB ? f : RV_NLIMB - 1;
    const float slack = fmaxr(fminr(tclear, 0.5f * bclear), 0.0f);
    const v3 cc = ld3(S.s.colc[col]);
    for (int j = 0; j < RV_NJ; ++j) {
      float L = S.s.ccoef[col][j];
      if (j <= fl && j < RV_NLIMB) L = fminr(L, (len(sub(cc, ld3(e.fpos[j]))) + S.s.colr[col] + slack) * 1.001f);
      S.s.crun[col][j] = L;
    }
  }
}
RV_DEV int coast_budget(Shared& S, const Consts& K, const int remaining, int* kidx) {
  const rv_config* c = K.cfg;
  const rv_arm* arm = K.arm;
  if (K.stop_after != 0 || !S.e.arm_enabled || remaining < 2) return 0;
  const int fresh = S.s.kin_fresh;
  if (!fresh && !S.s.clr_valid) return 0;
  {
    int any_on = 0;
#pragma unroll
    for (int b = 0; b < RV_MAXB; ++b) any_on |= body_on(S.e, b);
    if (any_on) return 0;
  }
  const int m0 = remaining < 16 ? remaining : 16;
  RV_LANES_BEGIN
    if (lane < 3) S.s.coast_unsafe[lane] = 0;
  RV_LANES_END
  RV_LANES_BEGIN
    if (lane < RV_NCOL) {
      const DevEnv& e = S.e;
      const int col = lane, f = arm->col_frame[col];
      const int fl = f < 7 ? f : 7;                 // last limb frame the box depends on
      const float dt = c->dt;
      // clearances: measured on fresh kinematics, else what the previous coasting
      // chunks left of them (each chunk subtracts its travel bound)
      if (fresh) coast_measure_clearances(S, K, col);
      const float tclear = S.s.clr_t[col], bclear = S.s.clr_b[col];
      if (lane == 0) S.s.clr_valid = 1;
      for (int k = 0; k < 3; ++k) {
        const int m = m0 >> k;
        if (m < 2) { S.s.coast_unsafe[k] = 1; continue; }
        const float mf = (float)m;
        // vertex travel bound over m substeps
        float delta = 0.0f, reach = S.s.colext[col];
        for (int j = fl; j >= 0; --j) {
          if (j < RV_NLIMB) {
            float q0 = fabsr(e.qd[j]);
            float vcap = fmaxr(q0, e.vmax_cmd[j]);
            float vacc = q0 + mf * arm->a_max[j] * dt;
            delta += reach * (mf * dt * fminr(vcap, vacc));
          }
          reach += S.s.jlen[j];
        }
        if (f >= 8) {
          int j = f - 1;                            // finger joint 7 / 8 slides the box
          float q0 = fabsr(e.qd[j]);
          delta += mf * dt * fminr(fmaxr(q0, e.vmax_cmd[j]), q0 + mf * arm->a_max[j] * dt);
        }
        delta = delta * 1.02f + 1e-4f;
        S.s.cdelta[k][col] = delta;
        if (!(tclear > delta) || !(bclear > 2.0f * delta)) S.s.coast_unsafe[k] = 1;
      }
    }
  RV_LANES_END
  for (int k = 0; k < 3; ++k) if (!S.s.coast_unsafe[k]) { *kidx = k; return m0 >> k; }
  return fresh ? 0 : -1;      // -1: the left-over clearances do not suffice; measure again and retry
}
// r substeps of the joint motors alone (arm_motor_phases (a)+(b) with a no-op control).
// Device: lane j < 9 keeps joint j in registers; the common scale of the limb joints
// is a 16-lane DPP min all-reduce (min is exact, so the order does not matter).
// Host emulation: the same arithmetic, joint by joint.
#if RV_ON_DEVICE
template <int N> RV_DEV float row_ror_min(float x) {
  float o = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x120 + N, 0xf, 0xf, false));
  return o < x ? o : x;
}
RV_DEV void motors_only_substeps(Shared& S, const Consts& K, const int r) {
  const rv_config* c = K.cfg; const rv_arm* arm = K.arm;
  DevEnv& e = S.e;
  const int lane = (int)threadIdx.x;
  const int j = lane < RV_NJ ? lane : RV_NJ - 1;
  const bool mine = lane < RV_NJ;
  const float dt = c->dt;
  float q = e.q[j], qd = e.qd[j];
  const int on = e.motor_on[j];
  const float kp = e.motor_kp[j], mq = e.motor_q[j], vmax = e.vmax_cmd[j];
  const float amax_dt = arm->a_max[j] * dt, lo = arm->q_lo[j], hi = arm->q_hi[j];
  float trav = 0.0f;
  for (int i = 0; i < r; ++i) {
    float vd = 0.0f, ratio = 1.0f;
    if (on) {
      vd = kp * (mq - q) * (1.0f / dt);
      float raw = fabsr(vd);
      if (j < RV_NLIMB && raw > vmax) ratio = vmax / raw;
    }
    if (!mine) ratio = 1.0f;
    float sync = ratio;      // (<= 1: vmax / raw only where raw > vmax)
    sync = row_ror_min<8>(sync); sync = row_ror_min<4>(sync); sync = row_ror_min<2>(sync); sync = row_ror_min<1>(sync);
    float vdd = 0.0f;
    if (on) {
      vdd = vd;
      if (j < RV_NLIMB) vdd = vdd * sync;
      vdd = fclamp_pm(vdd, vmax);
    }
    float dv = fclamp_pm(vdd - qd, amax_dt);
    float qdn = qd + dv;
    float qn = q + qdn * dt;
    if (qn < lo) { qn = lo; qdn = 0.0f; }
    if (qn > hi) { qn = hi; qdn = 0.0f; }
    q = qn; qd = qdn;
    trav += fabsr(qdn) * dt;
  }
  if (mine) { e.q[j] = q; e.qd[j] = qd; S.s.jtravel[j] += trav; }
  if (lane == 0) { e.sim_steps += r; e.substeps_last += r; }
  __syncthreads();
}
#else
#define RV_EMU_SECTION 4
#include "../../tests/emu/rv_emu_hooks.h"      // host lane emulation (test scaffolding; not compiled into the product)
#undef RV_EMU_SECTION
#endif

// m coasting substeps, then the kinematics of the final joint state
// upper bound of the distance a vertex of collider box col travelled, from the joint
// path lengths accumulated while coasting
RV_DEV float col_travelled(const Shared& S, const Consts& K, const int col) {
  const int f = K.arm->col_frame[col]; const int fl = f < 7 ? f : 7;
  float tr = 0.0f, reach = S.s.colext[col];
  for (int j = fl; j >= 0; --j) {
    if (j < RV_NLIMB) tr += reach * S.s.jtravel[j];
    reach += S.s.jlen[j];
  }
  if (f >= 8) tr += S.s.jtravel[f - 1];
  return tr * 1.02f + 1e-6f;
}
// after coasted substeps: FK / colliders are NOT refreshed: consecutive chunks run on the clearances left
// over (arm_refresh_kinematics is called by whoever needs frames or fresh clearances)
RV_DEV void coast_finish(Shared& S, const Consts& K) {
  RV_LANES_BEGIN
    DevEnv& e = S.e;
    if (lane == 0) e.flag_arm_table = 0;
    if (lane >= 1 && lane <= RV_MAXB) e.flag_arm_body[lane - 1] = 0;
    // what the boxes really travelled (joint path lengths x reach; never more than the
    // bound the chunk was admitted with) comes off the distance bounds and clearances
    if (lane >= 8 && lane < 8 + RV_MAXB * RV_NCOL) {
      int t = lane - 8, b = t / RV_NCOL, col = t - b * RV_NCOL;
      S.s.sep[b][col] = fmaxr(S.s.sep[b][col] - col_travelled(S, K, col), 0.0f);
    }
    if (lane >= 48 && lane < 48 + RV_NCOL) {
      int col = lane - 48; float d = col_travelled(S, K, col);
      S.s.clr_t[col] = S.s.clr_t[col] - d; S.s.clr_b[col] = S.s.clr_b[col] - 2.0f * d;
    }
    if (lane == 63) S.s.kin_fresh = 0;
  RV_LANES_END
}
RV_DEV void coast_substeps(Shared& S, const Consts& K, const int m, const int kidx) {
#ifdef RV_EMU_COUNT
  rv_emu_coasted += m;
#endif
  RV_CNT(8, m)
  RV_LANES_BEGIN
    if (lane < RV_NJ) S.s.jtravel[lane] = 0.0f;
  RV_LANES_END
  for (int i = 0; i < m;) {
    // substeps in which ControllableBody.update has nothing to do (no done-check, no
    // IK, motor targets already applied) leave only the joint motors: run those r
    // substeps in registers, without LDS round trips
    int r = 0;
    {
      const int lt_on = S.e.lt.active, jt_on = S.e.jt.active, steps = S.e.sim_steps, applied = S.s.jt_applied;
      while (i + r < m) {
        const int st = steps + r;
        const int noop = (!lt_on && !jt_on) ||
                         (st % RV_STEPS_TO_CHECK_DONE != 0 && jt_on && applied && (!lt_on || st % RV_STEPS_TO_UPDATE_IK != 0));
        if (!noop) break;
        ++r;
      }
    }
    if (r >= 2) { motors_only_substeps(S, K, r); i += r; }
    else { arm_motor_phases(S, K, 0, 1); i += 1; }
  }
  coast_finish(S, K);
}

// ---- fused coasting of the phase loop ---------------------------------------
// While the arm moves through free space the phase loop of PushEnv._execute_action is, substep
// after substep: ControllableBody.update finds nothing to change (the IK solution it tracks has
// converged, nothing is reached, nothing timed out), the joint motors advance, and every
// STEPS_CHECK substeps the phase machine finds the robot "not ready".  coast_fused() runs exactly
// those substeps with the joint state in registers (lane j = joint j), for as long as
//   * ControllableBody.update(st) provably has no effect            (ctl_update_noop)
//   * the tick after a substep provably has no effect                (ctl_tick_noop)
//   * no collider box can have come within contact range of anything: the joint path lengths,
//     weighted with the largest lever of each joint, stay below the smallest clearance left.
// It stops BEFORE the first substep / AT the first tick that needs the real code, so the state it
// leaves is the state plain stepping would have at that point (bit for bit).
struct CoastCtl {
  int lt_on, jt_on, applied, from_ik, lt_has_stop, lt_more, jt_has_stop, jt_limb;
  float lt_stop, jt_stop, dt;
  int interrupt, has_budget, max_phase_steps;
  // Grasp4DofEnv (its phase machine ticks after every substep): phase, substeps spent in 'start'
  int grasp, g_phase, g_action_steps, g_max_action;
  float g_ready_time;
};
RV_DEV int ctl_link_done(const CoastCtl& C, int st) {        // check_link_target_done at step st
  if (!C.lt_has_stop) return 1;
  if (C.dt * (float)st >= C.lt_stop) return 1;
  if (!C.lt_more) return 1;
  return 0;
}
RV_DEV int ctl_joint_done(const CoastCtl& C, int st, int reached) {   // check_joint_target_done
  if (!C.jt_has_stop) return 1;
  if (C.dt * (float)st >= C.jt_stop) return 1;
  if (reached) return 1;
  return 0;
}
// does control_update() at step st have anything to do at all?
RV_DEV int ctl_update_due(const CoastCtl& C, int st) {
  if (!C.lt_on && !C.jt_on) return 0;
  if (st % RV_STEPS_TO_CHECK_DONE != 0 && C.jt_on && C.applied && (!C.lt_on || st % RV_STEPS_TO_UPDATE_IK != 0)) return 0;
  return 1;
}
// ... and if it has: does it leave everything as it is?  reached = check_joints_reached now
RV_DEV int ctl_update_noop(const CoastCtl& C, int st, int reached) {
  if (!(C.jt_on && C.applied)) return 0;
  int ik_updated = 0;
  if (C.lt_on) {
    if (st % RV_STEPS_TO_CHECK_DONE == 0 && ctl_link_done(C, st)) return 0;
    if (st % RV_STEPS_TO_UPDATE_IK == 0) {
      if (C.from_ik != 2) return 0;          // the IK would be solved again
      ik_updated = 1;
      if (reached) return 0;                 // next pose of the path
    }
  }
  if (st % RV_STEPS_TO_CHECK_DONE == 0 || ik_updated) if (ctl_joint_done(C, st, reached)) return 0;
  return 1;                                  // (the motor targets are written again with the values they hold)
}
// phase_tick() at step st: not ready, nothing reset, no interrupt (the contact flags are clear
// while coasting, so check_safety passes in the phases the loop visits)
RV_DEV int ctl_tick_noop(const CoastCtl& C, int st, int reached) {
  if (C.interrupt || !C.has_budget || st >= C.max_phase_steps) return 0;
  if (C.lt_on && ctl_link_done(C, st)) return 0;
  if (C.jt_on && ctl_joint_done(C, st, reached)) return 0;
  return C.lt_on || (C.jt_on && C.jt_limb);
}
RV_DEV CoastCtl coast_ctl_load(const Shared& S, const Consts& K) {
  const DevEnv& e = S.e;
  CoastCtl C;
  C.lt_on = e.lt.active; C.jt_on = e.jt.active; C.applied = S.s.jt_applied; C.from_ik = e.jt.from_ik;
  C.lt_has_stop = e.lt.has_stop; C.lt_more = e.lt.has_pose || e.lt.nq != 0; C.jt_has_stop = e.jt.has_stop;
  C.jt_limb = 0;
  for (int i = 0; i < e.jt.n_idx; ++i) if (e.jt.idx[i] < RV_NLIMB) C.jt_limb = 1;
  C.lt_stop = e.lt.stop_t; C.jt_stop = e.jt.stop_t; C.dt = K.cfg->dt;
  C.interrupt = S.s.interrupt; C.has_budget = S.s.has_budget; C.max_phase_steps = S.s.max_phase_steps;
  C.grasp = K.cfg->env_type == RV_ENV_GRASP; C.g_phase = e.phase; C.g_action_steps = e.num_action_steps;
  C.g_max_action = K.cfg->max_action_steps; C.g_ready_time = e.gripper_ready_time;
  return C;
}
// gphase_tick() at step st (Grasp4DofEnv, after EVERY substep): nothing but the count of 'start'
// substeps changes.  start_ticks: ticks since C was loaded, this one included.  (The arm - table
// flag is clear while coasting.)
RV_DEV int ctl_gtick_noop(const CoastCtl& C, int st, int reached, int start_ticks) {
  if (C.g_phase == RV_GPHASE_START && C.g_action_steps + start_ticks >= C.g_max_action) return 0;   // the grasping motion is stuck
  if (C.lt_on && ctl_link_done(C, st)) return 0;
  if (C.jt_on && ctl_joint_done(C, st, reached)) return 0;
  if (C.lt_on || (C.jt_on && C.jt_limb)) return 1;                // limb not ready
  return C.dt * (float)st < C.g_ready_time;                       // limb ready: the gripper is not
}
#if RV_ON_DEVICE
template <int N> RV_DEV float row_ror_add(float x) {
  float o = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x120 + N, 0xf, 0xf, false));
  return o + x;
}
RV_DEV int uni(int x) { return __builtin_amdgcn_readfirstlane(x); }
RV_DEV float unif(float x) { return __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, x))); }
#endif
// returns the number of substeps taken; *tick_pending (why it stopped): 1 the clearance is used up,
// 2 at a tick the phase machine must see, 3 max_n substeps taken.  steps_check == 0: no phase
// machine (plain stepping, wait_until_stable with every body asleep)
RV_DEV void arm_refresh_kinematics(Shared& S, const Consts& K);
RV_DEV int coast_fused(Shared& S, const Consts& K, const int steps_check, const int max_n, int* tick_pending) {
  const rv_config* c = K.cfg; const rv_arm* arm = K.arm;
  *tick_pending = 0;
  if (K.stop_after != 0 || !S.e.arm_enabled) return 0;
  const int fresh = S.s.kin_fresh;
  if (!fresh && !S.s.clr_valid) return 0;
  {
    int any_on = 0;
#pragma unroll
    for (int b = 0; b < RV_MAXB; ++b) any_on |= body_on(S.e, b);
    if (any_on) return 0;
  }
  if (!fresh) {       // clearances left over by the last chunk: anything left at all?
    float cmin = 1e30f;
    for (int col = 0; col < RV_NCOL; ++col) cmin = fminr(cmin, fminr(S.s.clr_t[col], 0.5f * S.s.clr_b[col]));
    if (!(cmin > 2e-4f)) { RV_CNT(11, 1) *tick_pending = 1; return 0; }
  }
  RV_LANES_BEGIN
    if (fresh && lane < RV_NCOL) coast_measure_clearances(S, K, lane);
    if (lane == 63) { S.s.clr_valid = 1; S.s.fused_n = 0; }
    if (lane >= 32 && lane < 32 + RV_NJ) S.s.jtravel[lane - 32] = 0.0f;
  RV_LANES_END
  RV_PROF(21)
  // segments: between two of them ControllableBody.update does real work (a new IK solution,
  // the next pose of a path, ...) on lane 0; the clearances and path lengths carry over
  int skip_st = -1;            // the update of this step has been executed already
  for (;;) {
  const int st0 = S.e.sim_steps;
  int pending = 0;             // 0 update due, 1 clearance used up, 2 tick
#if RV_ON_DEVICE
  CoastCtl C = coast_ctl_load(S, K);
  C.lt_on = uni(C.lt_on); C.jt_on = uni(C.jt_on); C.applied = uni(C.applied); C.from_ik = uni(C.from_ik);
  C.lt_has_stop = uni(C.lt_has_stop); C.lt_more = uni(C.lt_more); C.jt_has_stop = uni(C.jt_has_stop); C.jt_limb = uni(C.jt_limb);
  C.lt_stop = unif(C.lt_stop); C.jt_stop = unif(C.jt_stop); C.dt = unif(C.dt);
  C.interrupt = uni(C.interrupt); C.has_budget = uni(C.has_budget); C.max_phase_steps = uni(C.max_phase_steps);
  C.grasp = uni(C.grasp); C.g_phase = uni(C.g_phase); C.g_action_steps = uni(C.g_action_steps); C.g_max_action = uni(C.g_max_action);
  C.g_ready_time = unif(C.g_ready_time);
  DevEnv& e = S.e;
  const int lane = (int)threadIdx.x;
  const int j = lane < RV_NJ ? lane : RV_NJ - 1;
  const bool mine = lane < RV_NJ;
  const bool limb = j < RV_NLIMB;
  const float dt = C.dt;
  float q = e.q[j], qd = e.qd[j];
  const bool on = e.motor_on[j] != 0;
  const float kp = e.motor_kp[j], mq = e.motor_q[j], vmax = e.vmax_cmd[j];
  const float amax_dt = arm->a_max[j] * dt, lo = arm->q_lo[j], hi = arm->q_hi[j];
  // lanes 16 .. 16 + RV_NCOL - 1 watch one collider box each
  const bool iscol = lane >= 16 && lane < 16 + RV_NCOL;
  const int col = iscol ? lane - 16 : 0;
  float cf[RV_NJ];
#pragma unroll
  for (int k = 0; k < RV_NJ; ++k) cf[k] = S.s.crun[col][k];
  const float clt = S.s.clr_t[col], clb = S.s.clr_b[col];
  // this lane's joint in the joint target
  int tgt = 0; float tpos = 0.0f;
  {
    const int n_idx = uni(e.jt.n_idx);
    for (int i = 0; i < n_idx; ++i) if (e.jt.idx[i] == lane) { tgt = 1; tpos = e.jt.pos[i]; }
  }
  const float pos_thr = unif(e.jt.pos_thr), vel_thr = unif(e.jt.vel_thr);
  const int has_vel = uni(e.jt.has_vel);
  float trav = S.s.jtravel[j];
  int st = uni(st0);
  // the control schedule as counters (st % 10, st % 100, st % steps_check)
  // (function arguments are not provably wave-uniform: without the readfirstlane the loop control
  // below is compiled as divergent code, exec-mask updates and all)
  const int sc = uni(steps_check);
  int k10 = st % RV_STEPS_TO_UPDATE_IK, k100 = st % RV_STEPS_TO_CHECK_DONE, kchk = sc > 0 ? st % sc : 0;
  int left = uni(max_n) - uni(S.s.fused_n);
  const bool any_tgt = C.lt_on || C.jt_on, quiet_ok = C.jt_on && C.applied;
  const int skip = uni(skip_st);
  // The out-of-reach test need not run every substep: within a segment |qd_j| never exceeds
  // vb_j = max(|qd_j| now, commanded limit) (the motor law moves qd towards a target inside the
  // limit), so a box travels at most Bc = sum_j lever_j vb_j dt per substep.  After a test that
  // passed with travel bound T, the next n substeps pass it as well when
  // 1.05 (T + (n + 1) Bc) + 2e-4 <= slack (3 % and 1e-4 more than the test itself asks for: far
  // above the rounding of the sums) -- they are taken unchecked, then the test runs again on the
  // exact path lengths.  Same substeps taken, same state: only the number of tests changes.
  // (a second bound of the same kind: a joint gains at most a_max dt of speed per substep, so the
  // next n substeps move a box by at most n B1 + n (n + 1) / 2 B2 with B1 = sum_j lever_j |qd_j| dt
  // now and B2 = sum_j lever_j a_j dt^2 -- much less than n Bc while the joints are slow; the
  // larger of the two counts is used)
  float Bc = 0.0f, B2 = 0.0f;
  {
    const float vb = on ? fmaxr(fabsr(qd), vmax) : fabsr(qd);
#pragma unroll
    for (int k = 0; k < RV_NJ; ++k) { Bc = __builtin_fmaf(cf[k], rdlane(vb, k), Bc); B2 = __builtin_fmaf(cf[k], rdlane(amax_dt, k), B2); }
    Bc = Bc * dt; B2 = B2 * dt;
  }
  const float slack = fminr(clt, 0.5f * clb);
  RV_PCNT(35, 1)
  int free_left = 0;
  for (;;) {
    {
      // Substeps at which nothing has to look at anything: no ControllableBody.update due at their
      // top (the counters say when the next one is), no out-of-reach test (free_left), not the last
      // one allowed, no tick of the phase machine after them.  Those m substeps are the joint
      // motors alone -- a loop without a single scalar decision in it.
      int m = free_left < left - 1 ? free_left : left - 1;
      if (sc > 0) m = m < sc - 1 - kchk ? m : sc - 1 - kchk;
      if (any_tgt) {
        if (!quiet_ok) m = 0;
        else {
          const int u100 = k100 == 0 ? 0 : RV_STEPS_TO_CHECK_DONE - k100;
          m = m < u100 ? m : u100;
          if (C.lt_on) { const int u10 = k10 == 0 ? 0 : RV_STEPS_TO_UPDATE_IK - k10; m = m < u10 ? m : u10; }
        }
      }
      if (m > 0) {
        for (int i = 0; i < m; ++i) {
          float vd = 0.0f;
          if (on) vd = kp * (mq - q) * (1.0f / dt);
          const float raw = fabsr(vd);
          const bool sat = on && limb && mine && raw > vmax;
          float sync = 1.0f;
          if (__builtin_amdgcn_ballot_w64(sat) != 0) {
            float ratio = 1.0f;
            if (sat) ratio = vmax / raw;
            int r = __builtin_bit_cast(int, ratio);      // (<= 1: vmax / raw only where raw > vmax)
            r = min(r, __builtin_amdgcn_update_dpp(0, r, 0x128, 0xf, 0xf, false));
            r = min(r, __builtin_amdgcn_update_dpp(0, r, 0x124, 0xf, 0xf, false));
            r = min(r, __builtin_amdgcn_update_dpp(0, r, 0x122, 0xf, 0xf, false));
            r = min(r, __builtin_amdgcn_update_dpp(0, r, 0x121, 0xf, 0xf, false));
            sync = __builtin_bit_cast(float, r);
          }
          float vdd = 0.0f;
          if (on) {
            vdd = vd;
            if (limb) vdd = vdd * sync;
            vdd = fclamp_pm(vdd, vmax);
          }
          const float dv = fclamp_pm(vdd - qd, amax_dt);
          float qdn = qd + dv;
          float qn = q + qdn * dt;
          if (qn < lo) { qn = lo; qdn = 0.0f; }
          if (qn > hi) { qn = hi; qdn = 0.0f; }
          trav = trav + fabsr(qdn) * dt;
          q = qn; qd = qdn;
        }
        RV_PCNT(32, m)
        st += m; left -= m; free_left -= m; kchk += m; k100 += m;
        k10 += m; if (k10 >= RV_STEPS_TO_UPDATE_IK) k10 -= RV_STEPS_TO_UPDATE_IK * (k10 / RV_STEPS_TO_UPDATE_IK);
      }
    }
    RV_PROF(44)
    if (any_tgt && (!quiet_ok || k100 == 0 || (C.lt_on && k10 == 0)) && st != skip) {
      int reached = 1;
      if (C.jt_on) {
        const bool ok = fabsr(tpos - q) < pos_thr && (!has_vel || fabsr(0.0f - qd) < vel_thr);
        reached = __builtin_amdgcn_ballot_w64(tgt && !ok) == 0;
      }
      if (!ctl_update_noop(C, st, reached)) {
        // the update runs between two segments -- provided this substep can be taken whatever the
        // update commands (|velocity step| <= a_max dt), so that it is not run twice
        const float travub = trav + (fabsr(qd) + amax_dt) * dt;
        float T = 0.0f;
#pragma unroll
        for (int k = 0; k < RV_NJ; ++k) T = __builtin_fmaf(cf[k], rdlane(travub, k), T);
        const float D = T * 1.02f + 1e-4f;
        if (__builtin_amdgcn_ballot_w64(iscol && !(clt > D && clb > 2.0f * D)) != 0) pending = 1;
        break;
      }
    }
    RV_PROF(45)
    float vd = 0.0f;
    if (on) vd = kp * (mq - q) * (1.0f / dt);
    const float raw = fabsr(vd);
    const bool sat = on && limb && mine && raw > vmax;
    float sync = 1.0f;
    if (__builtin_amdgcn_ballot_w64(sat) != 0) {       // some limb joint is over its speed limit: common scale
      float ratio = 1.0f;
      if (sat) ratio = vmax / raw;
      // ratios are positive floats: their order is the order of their bit patterns
      int r = __builtin_bit_cast(int, ratio);      // (<= 1)
      r = min(r, __builtin_amdgcn_update_dpp(0, r, 0x128, 0xf, 0xf, false));
      r = min(r, __builtin_amdgcn_update_dpp(0, r, 0x124, 0xf, 0xf, false));
      r = min(r, __builtin_amdgcn_update_dpp(0, r, 0x122, 0xf, 0xf, false));
      r = min(r, __builtin_amdgcn_update_dpp(0, r, 0x121, 0xf, 0xf, false));
      sync = __builtin_bit_cast(float, r);
    }
    float vdd = 0.0f;
    if (on) {
      vdd = vd;
      if (limb) vdd = vdd * sync;
      vdd = fclamp_pm(vdd, vmax);
    }
    float dv = fclamp_pm(vdd - qd, amax_dt);
    float qdn = qd + dv;
    float qn = q + qdn * dt;
    if (qn < lo) { qn = lo; qdn = 0.0f; }
    if (qn > hi) { qn = hi; qdn = 0.0f; }
    const float travn = trav + fabsr(qdn) * dt;
    // (|speed| of the joint after this substep, for the look-ahead below: computed HERE, by every lane -- the box lanes
    // read the joint lanes' values with v_readlane inside a branch only the box lanes take)
    const float sp = fabsr(qdn);
    // the boxes after this substep: still out of reach of everything?
    RV_PCNT(33, 1)
    if (free_left > 0) --free_left;
    else {
      RV_PCNT(34, 1)
      float T = 0.0f;
#pragma unroll
      for (int k = 0; k < RV_NJ; ++k) T = __builtin_fmaf(cf[k], rdlane(travn, k), T);
      const float D = T * 1.02f + 1e-4f;
      if (__builtin_amdgcn_ballot_w64(iscol && !(clt > D && clb > 2.0f * D)) != 0) { pending = 1; break; }   // no: this substep is not taken
      // how many of the next substeps need no test (smallest count over the box lanes 16 .. 25)
      float nf = 1e6f;
      // (the cross-lane reads of the joint lanes' speeds happen HERE, where every lane is active: inside the branch below only
      // the box lanes are, and a v_readlane of a lane that is inactive where it is executed is undefined to the compiler --
      // it may sink the producer of `sp` into the branch, where the joint lanes never execute it.  Round 4 found this as a
      // parity break of an `exact on paper' rewrite; round 5 again, with |x| as a source modifier: now it is out of the branch)
      float B1 = 0.0f;
#pragma unroll
      for (int k = 0; k < RV_NJ; ++k) B1 = __builtin_fmaf(cf[k], rdlane(sp, k), B1);
      if (iscol && Bc > 0.0f) {
        const float room = (slack - 2e-4f) * (1.0f / 1.05f) - T;
        nf = room / Bc - 1.0f;
        if (B2 > 0.0f && room > 0.0f) {
          const float hb = B1 * dt + 0.5f * B2;          // n hb + n^2 B2 / 2 <= room
          const float nq = 0.98f * (fsqrtr(hb * hb + 2.0f * B2 * room) - hb) / B2 - 1.0f;
          nf = fmaxr(nf, nq);
        }
        nf = fminr(nf, 1e6f);
      }
      nf = row_ror_min<8>(nf); nf = row_ror_min<4>(nf); nf = row_ror_min<2>(nf); nf = row_ror_min<1>(nf);
      const float nmin = rdlane(nf, 16);
      free_left = nmin >= 1.0f ? (int)nmin : 0;
    }
    RV_PROF(46)
    q = qn; qd = qdn; trav = travn; ++st;
    if (++k10 == RV_STEPS_TO_UPDATE_IK) k10 = 0;
    if (++k100 == RV_STEPS_TO_CHECK_DONE) k100 = 0;
    --left;
    if (++kchk == sc) {
      kchk = 0;
      int reached = 1;
      if (C.jt_on) {
        const bool ok = fabsr(tpos - q) < pos_thr && (!has_vel || fabsr(0.0f - qd) < vel_thr);
        reached = __builtin_amdgcn_ballot_w64(tgt && !ok) == 0;
      }
      if (!(C.grasp ? ctl_gtick_noop(C, st, reached, st - uni(st0)) : ctl_tick_noop(C, st, reached))) { pending = 2; break; }
    }
    if (left == 0) { pending = 3; break; }      // (after the tick test: a tick that is due is never skipped)
    RV_PROF(47)
  }
  const int n = st - uni(st0);
  if (n > 0) {
    if (mine) { e.q[j] = q; e.qd[j] = qd; S.s.jtravel[j] = trav; }
    if (lane == 0) {
      e.sim_steps += n; e.substeps_last += n; S.s.fused_n += n;
      // Grasp4DofEnv: the ticks this segment passed over counted their 'start' substeps (the tick
      // that ended it, if any, is executed -- and counted -- by the caller)
      if (C.grasp && C.g_phase == RV_GPHASE_START) e.num_action_steps += n - (pending == 2 ? 1 : 0);
    }
  }
  if (lane == 0) S.s.fused_pending = pending;
  __syncthreads();
#else
#define RV_EMU_SECTION 5
#include "../../tests/emu/rv_emu_hooks.h"      // host lane emulation (test scaffolding; not compiled into the product)
#undef RV_EMU_SECTION
#endif
  if (S.s.fused_pending != 0) break;
  // ControllableBody.update of this step for real, then on with its motors
  skip_st = S.e.sim_steps;
  control_update_phases(S, K);
  }
  RV_PROF(22)
  const int n_taken = S.s.fused_n;
  *tick_pending = S.s.fused_pending;
  if (n_taken == 0) return 0;
#ifdef RV_EMU_COUNT
  rv_emu_coasted += n_taken;
#endif
  coast_finish(S, K);
  RV_PROF(19)
  return n_taken;
}
// frames, colliders and AABBs of the current joint state (after coasting)
RV_DEV void arm_refresh_kinematics(Shared& S, const Consts& K) {
  arm_lq_phase(S, K);
  arm_fk_phases(S, K);
  arm_collider_phases(S, K, 1);
}
// up to `want` coasted substeps with the fused loop; the kinematics are measured again whenever the
// clearance is used up (fresh clearances that buy nothing: close to something -> the caller steps)
RV_DEV int coast_run_body(Shared& S, const Consts& K, const int steps_check, const int want, int* why_out) {
  int why = 0, n = 0;
  for (;;) {
    n += coast_fused(S, K, steps_check, want - n, &why);
    RV_PROF(11)
    if (why != 1 || S.s.kin_fresh) break;
#if RV_ON_DEVICE
    if (S.s.bud_clk != 0 && __builtin_amdgcn_s_memtime() - S.s.bud_t0 > S.s.bud_clk) break;   // rv_step_poll: out of time
#endif
    arm_refresh_kinematics(S, K);
    RV_PROF(10)
    RV_CNT(12, 1)
  }
  *why_out = why;
  return n;
}
#if defined(RV_COAST_NOINLINE) && RV_ON_DEVICE
// A build variant of the 256-register kernel (RV_OCC2_COAST_OUT_OF_LINE, off): the coasting run as a function of its own.
// Inlined into the substep loop its hottest loops -- one iteration per coasted substep, 97 % of all substeps -- carry reloads of
// values the surrounding code pushed out to scratch (-Rpass-missed=regalloc: 22 + 4 + 1 reloads in the three nested loops of
// coast_fused, 49 in this one); as a callee it is allocated by itself and those loops are clean -- but a run of coasted
// substeps is short (tens of substeps) and the save / restore of the caller's live registers per call costs more than the
// reloads did: measured - 9 % on config 5, - 11 % on config 4 (profiles/r06_k_occ2_coast_out_of_line.txt).  (n < 2^28; why in the top bits)
RV_DEV_NOINLINE int coast_run_fn(const rv_scene* scene, int stop_after, int steps_check, int want);      // (defined below g_shared)
RV_DEV int coast_run(Shared& S, const Consts& K, const int steps_check, const int want, int* why_out) {
  (void)S;
  const int r = coast_run_fn(K.scene, K.stop_after, steps_check, want);
  *why_out = (int)((unsigned)r >> 28);
  return r & 0x0fffffff;
}
#else
RV_DEV int coast_run(Shared& S, const Consts& K, const int steps_check, const int want, int* why_out) { return coast_run_body(S, K, steps_check, want, why_out); }
#endif

// link twist of frame f (used by the arm-body contact rows; base is static): w_f = sum_k axis_k qd_k,
// v_f = sum_k (axis_k qd_k) x (p_f - p_k) over the joints upstream of f; the fingers add their slide along
// the hand's y axis.  fmot: how far a collider vertex riding on the frame can travel in this substep.
// Runs on idle lanes of the body-velocity phase of the light part (one phase less per heavy substep).
RV_DEV void arm_twist_lane(Shared& S, const Consts& K, const int f) {
  const rv_config* c = K.cfg;
  const DevEnv& e = S.e;
  const int kmax = f < RV_NLIMB ? f : RV_NLIMB - 1;
  v3 pf = ld3(e.fpos[f]);
  v3 fw = mk(0, 0, 0), fv = mk(0, 0, 0);
#pragma unroll
  for (int k = 0; k < RV_NLIMB; ++k) if (k <= kmax) {
    v3 u = scale(ld3(S.s.axis[k]), e.qd[k]);
    fw = add(fw, u);
    fv = add(fv, cross(u, sub(pf, ld3(e.fpos[k]))));
  }
  // the slide of a finger along the hand's y axis (a solver DOF of its own in finger_dynamics mode)
  if (f >= 8 && !c->finger_dynamics) fv = madd(fv, mk(S.s.frot[7][1], S.s.frot[7][4], S.s.frot[7][7]), e.qd[f - 1]);
  st3(S.s.fv[f], fv); st3(S.s.fw[f], fw);
  S.s.fmot[f] = (len(fv) + len(fw) * S.s.fext[f]) * c->dt;
  if (f >= 8 && c->finger_dynamics) S.s.fmot[f] += fabsr(e.qd[f - 1]) * c->dt;
}
RV_DEV int sim_substep_light(Shared& S, const Consts& K) {
  const rv_config* c = K.cfg;
  const int arm_on = S.e.arm_enabled;

  // An arm that did not move (every joint at rest: the settle after a push, the wait for the
  // bodies to come to rest) has the frames, collider boxes and table flags of the last substep:
  // only the per-substep flags are reset.  Exact: the skipped phases would recompute the same values.
  int arm_static = 0;
  // "Arm far": while a body is awake but no collider box can be within contact / wake range of any body or of the
  // table -- judged on the boxes as last computed, inflated by what the joints have travelled since plus what they
  // can travel in this substep -- the arm's frames are not needed: no arm - body or arm - table query would run,
  // no sleeper would be woken by it.  Such a substep moves the joints only (control update + motors); forward
  // kinematics, collider boxes, arm wake tests and link twists wait until a box may be near something or somebody
  // needs the frames (kin_fresh = 0).  Exact: every skipped test would have said "no".  (Without deactivation --
  // the reference's most likely semantics -- the arm is far in ~80 % of the substeps.)
  int far = 0;
#if !RV_ON_DEVICE
  static const int rv_no_far = getenv("RV_NO_FAR") != nullptr;     // (host emulation: debugging aid)
#else
  const int rv_no_far = 0;
#endif
  if (arm_on && K.stop_after == 0 && S.s.far_valid && !rv_no_far && !c->finger_dynamics && !c->limb_dynamics) {
    int ok = 0;
#pragma unroll
    for (int b = 0; b < RV_MAXB; ++b) ok |= body_on(S.e, b);
#pragma unroll
    for (int b = 0; b < RV_MAXB; ++b) if (S.e.man[RV_AIDX(b)].n != 0) ok = 0;
#pragma unroll
    for (int b = 0; b < RV_MAXB; ++b) if (RV_CON_TYPE(S.e.con_on[b]) != 0 && RV_CON_CHILD(S.e.con_on[b]) >= RV_MAXB) ok = 0;   // (a body tied to a LINK: the frames are needed)
    if (ok) {
      int near_any = 0;
      RV_LANES_BEGIN
        const DevEnv& e = S.e; const rv_arm* arm = K.arm;
        int near = 0;
        if (lane < RV_MAXB * RV_NCOL + RV_NCOL) {
          const int isb = lane < RV_MAXB * RV_NCOL;
          const int b = isb ? lane / RV_NCOL : 0, col = isb ? lane - b * RV_NCOL : lane - RV_MAXB * RV_NCOL;
          // travel bound of a vertex of the box: chain lever x joint path (so far, plus this substep's worst case)
          float T = 0.0f;
#pragma unroll
          for (int j = 0; j < RV_NJ; ++j) T = T + S.s.ccoef[col][j] * (S.s.ftravel[j] + (fabsr(e.qd[j]) + arm->a_max[j] * c->dt) * c->dt);
          T = T * 1.02f + 1e-4f;
          float lo[3], hi[3];
#pragma unroll
          for (int x = 0; x < 3; ++x) { lo[x] = S.s.colmin[col][x] - T; hi[x] = S.s.colmax[col][x] + T; }
          if (isb) {
            if (body_present(e, b) && !body_static(e, b)) {      // (a static body has no business with the arm)
              if (e.asleep[b]) {
                const float r = wake_range(e, arm, c, b, col) + 2.0f * c->margin;
                near = aabb_aabb_dist2(e.baabb[b], e.baabb[b] + 3, lo, hi) < r * r;
              }
              // (the body itself moves before the heavy part looks: its travel of this substep is bounded by its speed
              // after gravity; 2 x that and 1 mm for the velocity the solver may add.  A SLEEPER takes this test too:
              // another body may wake it in this very substep, and the heavy part then treats it like any awake body --
              // round 5, found by a pair count that was one short of the oracle's: the skipped query could not have
              // found a contact, but "every skipped test would have said no" has to hold for the cull tests as well)
              const float vb = (len(ld3(e.body[b] + 7)) + len(ld3(e.body[b] + 10)) * e.radius[b] + fabsr(c->gravity_z) * c->dt) * c->dt;
              const float r = e.radius[b] + brk_ab(e, arm, c, b, col) + 2.0f * vb + 1e-3f;
              near |= !(sphere_aabb_dist2(ld3(e.body[b]), lo, hi) >= r * r);
            }
          } else {
            // the arm - table gate: the box stays above the contact-query distance of the table top
            near = !(lo[2] - e.table_z - c->margin >= c->contact_query_dist);
          }
        }
#if RV_ON_DEVICE
        near_any = __builtin_amdgcn_ballot_w64(near != 0) != 0;
#else
        near_any |= near;
#endif
      RV_LANES_END
      far = !near_any;
    }
  }
  if (far) {
    arm_motor_phases(S, K, 0, 2);
    RV_PROF(40)
    RV_LANES_BEGIN
      if (lane < RV_MAXB) { S.s.wake[lane] = 0; S.s.bnear[lane] = 0; }
      if (lane == 8) S.s.near_any = 0;
      if (lane == 9) S.s.arm_moving = 0;
      if (lane >= 16 && lane < 16 + RV_NCOL) { S.s.colflag[lane - 16] = 0; S.s.atflag[lane - 16] = 0; }
      if (lane == 63) { S.s.kin_fresh = 0; S.s.clr_valid = 0; S.s.far = 1; S.s.far_n = S.s.far_n + 1; }
    RV_LANES_END
  } else
  if (arm_on) {
    arm_motor_phases(S, K, 1, 0);
    RV_PROF(40)
    RV_STOPL(12)
    if (S.s.kin_fresh && K.stop_after == 0) {
      arm_static = 1;
#pragma unroll
      for (int j = 0; j < RV_NJ; ++j) if (S.s.jchg[j]) arm_static = 0;
    }
    if (!arm_static) arm_fk_phases(S, K);
  }
  RV_PROF(41)
  RV_STOPL(1)
  if (arm_static) {
    RV_LANES_BEGIN
      if (lane < RV_MAXB) { S.s.wake[lane] = 0; S.s.bnear[lane] = 0; }
      if (lane == 8) S.s.near_any = 0;
      if (lane == 9) S.s.arm_moving = 0;
      if (lane >= 16 && lane < 16 + RV_NCOL) { S.s.colflag[lane - 16] = 0; S.s.coltravel[lane - 16] = 0.0f * 1.02f + 1e-7f; }
      if (lane == 63) S.s.clr_valid = 0;
    RV_LANES_END
  } else if (!far) {
    arm_collider_phases(S, K, arm_on);
  }
  RV_PROF(42)
  RV_STOPL(16)

  // wake test.  A sleeping body is woken by a MOVING awake body nearby, or by the
  // moving arm when one of its boxes comes within the contact-breaking distance of
  // the body's hulls (= when a contact point would be created).  Stage 1, one lane
  // per (body, box) and per ordered (body, neighbour) pair: boxes whose AABB is
  // within that distance of the sleeper's cached AABB are flagged for stage 2.
  RV_LANES_BEGIN
    const DevEnv& e = S.e;
    if (lane < RV_MAXB * RV_NCOL) {
      int b = lane / RV_NCOL, col = lane - b * RV_NCOL;
      int nr = 0;
      // sep: a lower bound of (hull distance - contact range) left over from the last
      // distance query, minus the box's travel since (a sleeper does not move); while it
      // is positive the query cannot hit and is skipped.  Exact: only the work changes.
      float sep = S.s.sep[b][col] - S.s.coltravel[col];
      if (S.s.far_n != 0) sep = 0.0f;        // ("arm far" substeps did not keep the bound up to date: it starts again)
      if (arm_on && S.s.arm_moving && body_present(e, b) && e.asleep[b] && !body_static(e, b)) {
        float r = wake_range(e, K.arm, c, b, col) + 2.0f * c->margin;
        nr = aabb_aabb_dist2(e.baabb[b], e.baabb[b] + 3, S.s.colmin[col], S.s.colmax[col]) < r * r;
        if (nr && sep > 0.0f) nr = 0;
      }
      S.s.sep[b][col] = sep > 0.0f ? sep : 0.0f;
      S.s.nearf[b][col] = nr;
      if (nr) { S.s.bnear[b] = 1; S.s.near_any = 1; }
    } else if (lane < RV_MAXB * RV_NCOL + RV_MAXB * (RV_MAXB - 1)) {
      int t = lane - RV_MAXB * RV_NCOL;
      int b = t / (RV_MAXB - 1), a = t - b * (RV_MAXB - 1);
      if (a >= b) a++;
      // only a MOVING neighbour wakes a sleeper (resting neighbours would ping-pong)
      // ... and moving means: left its 1 mm pose window within the last 50 substeps
      if (body_present(e, b) && e.asleep[b] && body_on(e, a) && !(e.sleep_count[a] > 0) && !(e.still_count[a] >= 50)) {
        v3 d = sub(ld3(e.body[a]), ld3(e.body[b]));
        float r = e.radius[a] + e.radius[b] + brk_bb(e, c, a, b);
        if (dot(d, d) < r * r) S.s.wake[b] = 1;
      }
    }
  RV_LANES_END
  const int near_any = S.s.near_any;
  if (near_any) {
    // stage 2a: world hull vertices of the flagged sleepers, and the vertices of the boxes near them
    RV_LANES_BEGIN
      for (int item = lane; item < RV_NCOL * 8; item += 64) {
        const int col = item >> 3;
        int nd = 0;
#pragma unroll
        for (int b = 0; b < RV_MAXB; ++b) nd |= S.s.nearf[b][col];
        if (nd) arm_box_vertex(S, K, col, item & 7);
      }
      for (int item = lane; item < RV_MAXB * RV_MAXH * RV_MAXV; item += 64) {
        int b = item / (RV_MAXH * RV_MAXV), h = (item / RV_MAXV) % RV_MAXH, i = item % RV_MAXV;
        if (!S.s.bnear[b]) continue;
        if (h >= S.n_hulls[b] || i >= S.n_verts[b][h]) continue;
        const rv_shape* s = &K.scene->shapes[S.e.shape[b]];
        float sc = S.e.scale[b];
        v3 l = mk(s->verts[h][i][0] * sc, s->verts[h][i][1] * sc, s->verts[h][i][2] * sc);
        st3(S.s.u.r.wv[b][h][i], add(ld3(S.e.body[b]), mulv(qmat(ldq(S.e.body[b] + 3)), l)));
      }
    RV_LANES_END
    // stage 2b: one 16-lane group per body runs the distance queries, box by box
    RV_LANES_BEGIN
      const DevEnv& e = S.e;
      const int b = lane >> 4;
#if !RV_ON_DEVICE
      if ((lane & 15) != 0) continue;   // host emulation: one lane per group does the work
#endif
      if (S.s.bnear[b] && !S.s.wake[b]) {
        const float mg = c->margin;
        int hit = 0;
        for (int col = 0; col < RV_NCOL && !hit; ++col) {
          if (!S.s.nearf[b][col]) continue;
          const float brk = wake_range(e, K.arm, c, b, col);
          v3 d = sub(ld3(e.body[b]), ld3(S.s.colc[col]));
          float lbmin = 1e30f;
          for (int h = 0; h < S.n_hulls[b] && !hit; ++h) {
            v3 n, pa, pb; float dist, lb = 0.0f;
            if (gjk_epa(&S.s.u.r.wv[b][h][0][0], S.n_verts[b][h], &S.s.colv[col][0][0], 8, d, brk + 2.0f * mg, &n, &dist, &pa, &pb, &lb))
              if (!(dist - 2.0f * mg > brk)) hit = 1;
            lbmin = fminr(lbmin, lb);
          }
          if (!hit && (lane & 15) == 0) S.s.sep[b][col] = fmaxr(lbmin - 2.0f * mg - brk - 1e-5f, 0.0f);
        }
        if (hit) S.s.wake[b] = 1;
      }
    RV_LANES_END
  }
  RV_PROF(43)
  RV_STOPL(17)

  // body velocity update + rotations
  // A contact island is awake or asleep as a whole (Bullet deactivates islands, not bodies): a
  // sleeper that shares a manifold holding points with a body that is awake, or is being woken,
  // wakes as well -- otherwise that manifold would not be solved and the awake body would lose its
  // support.  Every body lane works the closure out for itself.
  int wake_me = 0;
  RV_LANES_BEGIN
    if (lane < RV_MAXB) {
      const DevEnv& e = S.e;
      // (islands sleep as a whole, so only a wake-up can leave a sleeper next to an awake body)
      int any_wake = 0;
#pragma unroll
      for (int x = 0; x < RV_MAXB; ++x) any_wake |= S.s.wake[x];
      int me = 0;
      if (any_wake) {
        int aw[RV_MAXB], pr[RV_MAXB], np[RV_NBB];
#pragma unroll
        for (int x = 0; x < RV_MAXB; ++x) { pr[x] = body_present(e, x); aw[x] = (pr[x] && !e.asleep[x]) || S.s.wake[x]; }
#pragma unroll
        for (int k = 0; k < RV_NBB; ++k) np[k] = e.man[RV_BBIDX(k)].n;
#pragma unroll
        for (int pass = 0; pass < 3; ++pass)
#pragma unroll
          for (int k = 0; k < RV_NBB; ++k) {
            const int a_ = bb_a(k), b_ = bb_b(k);
            if (!(pr[a_] && pr[b_]) || np[k] == 0) continue;
            if (aw[a_] && !aw[b_]) aw[b_] = 1;
            else if (aw[b_] && !aw[a_]) aw[a_] = 1;
          }
#pragma unroll
        for (int x = 0; x < RV_MAXB; ++x) if (x == lane) me = aw[x] && pr[x] && e.asleep[x];
      }
#if RV_ON_DEVICE
      wake_me = me;
#else
      S.s.ready[lane] = me;      // (host emulation: lane-local values do not survive the phase)
#endif
    }
#if !RV_ON_DEVICE
  RV_LANES_END
  RV_LANES_BEGIN
#endif
    if (arm_on && !far && lane >= 16 && lane < 16 + RV_NFRAME) arm_twist_lane(S, K, lane - 16);
    if (lane == 6) { S.s.far = far; if (!far) S.s.far_n = 0; }
    if (lane < RV_MAXB) {
      int b = lane; DevEnv& e = S.e;
#if !RV_ON_DEVICE
      wake_me = S.s.ready[lane];
#endif
      if (S.s.wake[b] || wake_me) {
        e.asleep[b] = 0; e.sleep_count[b] = 0; e.deact_count[b] = 0;
        // open the pose window at the pose it was resting in
        e.still_count[b] = 1; e.undisturbed[b] = 1;
        st3(e.still_ref[b], ld3(e.body[b])); stq(e.still_ref[b] + 3, ldq(e.body[b] + 3));
      }
      if (body_on(e, b)) {
        float dt = c->dt;
        if (body_static(e, b)) S.s.mot[b] = 0.0f;      // (no gravity on a static body; its velocities stay zero)
        else {
        e.body[b][7] += c->gravity_xy[0] * dt; e.body[b][8] += c->gravity_xy[1] * dt;
        e.body[b][9] += c->gravity_z * dt;
        v3 v = scale(ld3(e.body[b] + 7), c->lin_damp);
        v3 w = scale(ld3(e.body[b] + 10), c->ang_damp);
        st3(e.body[b] + 7, v); st3(e.body[b] + 10, w);
        S.s.mot[b] = (len(v) + len(w) * e.radius[b]) * dt;
        }
        m3 m = qmat(ldq(e.body[b] + 3));
        stm(S.s.rot[b], m);
        const float* ii = e.inv_inertia[b];
        for (int r = 0; r < 3; ++r)
          for (int cc2 = 0; cc2 < 3; ++cc2)
            S.s.iinv[b][r * 3 + cc2] = m.m[r * 3 + 0] * ii[0] * m.m[cc2 * 3 + 0] + m.m[r * 3 + 1] * ii[1] * m.m[cc2 * 3 + 1] + m.m[r * 3 + 2] * ii[2] * m.m[cc2 * 3 + 2];
      }
    }
  RV_LANES_END
  RV_STOPL(18)

  // uniform: is any body awake?  can any arm collider be within the contact-query
  // distance of the table top?
  int any_on = 0, at_possible = 0;
#pragma unroll
  for (int b = 0; b < RV_MAXB; ++b) any_on |= body_on(S.e, b);
  if (arm_on) {
#pragma unroll
    for (int col = 0; col < RV_NCOL; ++col) at_possible |= S.s.atflag[col];
  }
  // quiet substep: every body asleep (or absent) and no arm collider near the
  // table -> nothing to collide, solve or integrate
  if (!any_on && !at_possible) {
    RV_LANES_BEGIN
      DevEnv& e = S.e;
      if (lane == 0) { e.flag_arm_table = 0; e.sim_steps++; e.substeps_last++; }
      if (lane >= 1 && lane <= RV_MAXB) e.flag_arm_body[lane - 1] = 0;
    RV_LANES_END
    return 0;
  }
  RV_LANES_BEGIN
    if (lane == 0) S.s.any_on = any_on;
  RV_LANES_END
  return 1;
}

// The rest of the step: contacts, solver, integration.
RV_DEV void sim_substep_heavy(Shared& S, const Consts& K) {
  const rv_config* c = K.cfg;
  const rv_arm* arm = K.arm;
  const int arm_on = S.e.arm_enabled && !S.s.far;     // ("arm far" substep: no box can be near anything, the frames are stale)
  // (the link twists are computed by idle lanes of the last light phase, arm_twist_lane)
  RV_STOP(2)
  RV_PROF(2)
  // manifold refresh + narrow phase.  24 manifold owners (4 body-table, 6 body-body, 4
  // arm-body, 10 arm-table detectors) write disjoint manifolds, so the schedule is free:
  //  (0) one lane per cached point: refreshed distance / break test;
  //      one lane per (body, collider box): bounding-sphere / box proximity;
  //  (1) one lane per owner: drop broken points, motion gate -> does the owner run convex
  //      queries in this substep?
  //  (2) lane 0: the compact list of owners that do;
  //  (3) the wave splits into four 16-lane groups that take owners off the list.  All lanes
  //      of a group execute the same scalar program redundantly -- except inside support_v(),
  //      where each lane holds one hull vertex and a DPP all-reduce picks the extreme one.
  //      Every convex pair of every owner goes through ONE collide_pair call site.
  RV_LANES_BEGIN
    const DevEnv& e = S.e;
    if (lane < RV_NMAN * 4) {
      const int mi = lane >> 2, i = lane & 3;
      // whose manifold, and is it refreshed?  (the same conditions as owner_decode's `live`)
      int kind, a, b = -1, live;
      if (mi < RV_MAXB) { kind = 0; a = mi; live = body_present(e, a) && !e.asleep[a] && !body_static(e, a); }
      else if (mi < RV_MAXB + RV_NBB) {
        kind = 1; a = bb_a(mi - RV_MAXB); b = bb_b(mi - RV_MAXB);
        live = body_present(e, a) && body_present(e, b) && !e.asleep[a] && !e.asleep[b] && !(body_static(e, a) && body_static(e, b));
      } else { kind = 2; a = mi - RV_MAXB - RV_NBB; live = body_present(e, a) && !e.asleep[a] && arm_on && !body_static(e, a); }
      float d = 0.0f; int rm = 0;
      if (live && i < e.man[mi].n) refresh_point(S, K, kind, a, b, e.man[mi], i, &d, &rm);
      S.s.rf_dist[lane] = d; S.s.rf_rm[lane] = rm;
    }
    if (lane >= 60) S.s.wvneed[lane - 60] = 0;
#ifdef RV_DIAG_AWAKE_BODIES   // diagnostic build (tools/diag_lockstep.py): count awake BODIES per substep
    if (lane == 59) { int nb_ = 0; for (int b = 0; b < RV_MAXB; ++b) nb_ += body_on(S.e, b); S.e.awake_last += nb_ * 65536 + S.s.any_on; }
#else
    if (lane == 59) S.e.awake_last += S.s.any_on;
#endif
  RV_LANES_END
  RV_LANES_BEGIN
    DevEnv& e = S.e;
    int runs = 0;
    if (lane < RV_NMAN + RV_NCOL) {
      const int owner = lane;
      OwnerInfo o;
      owner_decode(S, K, owner, arm_on, o);
      if (o.clear) e.man[o.mi].n = 0;
      int n_pairs = o.n_outer * o.n_inner;
      if (o.live) {
        DevMan& m = e.man[o.mi];
        const int lost = refresh_apply(m, &S.s.rf_dist[o.mi * 4], &S.s.rf_rm[o.mi * 4]);
        // narrow-phase gating on the travel of the two shapes since the last full pass
        float mo = S.s.mot[o.a];
        if (owner >= RV_MAXB + RV_NBB) {
          float am = 0.0f;
#pragma unroll
          for (int f = 0; f < RV_NFRAME; ++f) am = fmaxr(am, S.s.fmot[f]);
          mo = mo + am;
        }
        if (o.b >= 0) mo = mo + S.s.mot[o.b];
        float acc = m.acc + mo;
        // (a body on fewer than three support points is rocking or tipping: its support is looked at every substep)
        const int run = (c->np_max_age <= 0) || m.n == 0 || (owner < RV_MAXB && m.n < 3) || lost > 0 || acc > c->np_gate || (e.sim_steps % c->np_max_age) == 0;
        if (run) acc = 0.0f; else n_pairs = 0;
        m.acc = acc;
      }
      S.s.ow_run[owner] = n_pairs > 0;
      runs = n_pairs > 0;
      if (n_pairs > 0 && o.role >= 0 && o.role <= 2) { S.s.wvneed[o.a] = 1; if (o.b >= 0) S.s.wvneed[o.b] = 1; }
    }
    if (lane >= RV_NMAN + RV_NCOL && lane < RV_NMAN + RV_NCOL + RV_MAXB * RV_NCOL / 2) {
      // (body, box) proximity for the arm-body owners, two pairs per lane
      for (int t = (lane - RV_NMAN - RV_NCOL) * 2; t < (lane - RV_NMAN - RV_NCOL) * 2 + 2; ++t) {
        const int b = t / RV_NCOL, col = t - b * RV_NCOL;
        int near = 0;
        if (arm_on && body_on(e, b) && !body_static(e, b)) {
          const float r = e.radius[b] + brk_ab(e, arm, c, b, col);
          near = !(sphere_aabb_dist2(ld3(e.body[b]), S.s.colmin[col], S.s.colmax[col]) >= r * r);
        }
        S.s.cn[b][col] = near;
      }
    }
#if RV_ON_DEVICE
    {
      // the compact list of owners that run, in owner order
      const unsigned long long mk = __builtin_amdgcn_ballot_w64(runs != 0);
      if (runs) S.s.olist[__builtin_popcountll(mk & ((1ull << lane) - 1ull))] = lane;
      if (lane == 0) S.s.n_olist = __builtin_popcountll(mk);
    }
#else
    (void)runs;
#endif
  RV_LANES_END
#if !RV_ON_DEVICE
  RV_LANES_BEGIN
    if (lane == 0) {
      int n = 0;
      for (int o = 0; o < RV_NMAN + RV_NCOL; ++o) if (S.s.ow_run[o]) S.s.olist[n++] = o;
      S.s.n_olist = n;
    }
  RV_LANES_END
#endif
  // world hull vertices of the bodies, and vertices of the collider boxes, that take part in a convex query
  {
    int any = 0;
#pragma unroll
    for (int b = 0; b < RV_MAXB; ++b) any |= S.s.wvneed[b];
    if (arm_on) {
#pragma unroll
      for (int col = 0; col < RV_NCOL; ++col) any |= S.s.atflag[col];
    }
    if (any) {
      RV_LANES_BEGIN
        if (arm_on) {
          for (int item = lane; item < RV_NCOL * 8; item += 64) {
            const int col = item >> 3;
            int nd = S.s.atflag[col];
#pragma unroll
            for (int b = 0; b < RV_MAXB; ++b) nd |= S.s.cn[b][col];
            if (nd) arm_box_vertex(S, K, col, item & 7);
          }
        }
        for (int b = 0; b < RV_MAXB; ++b) {
          if (!S.s.wvneed[b]) continue;
          const rv_shape* s = &K.scene->shapes[S.e.shape[b]];
          const float sc = S.e.scale[b];
          const int n_items = S.n_hulls[b] * RV_MAXV;
          for (int item = lane; item < n_items; item += 64) {
            const int h = item / RV_MAXV, i = item % RV_MAXV;
            if (i >= S.n_verts[b][h]) continue;
            v3 l = mk(s->verts[h][i][0] * sc, s->verts[h][i][1] * sc, s->verts[h][i][2] * sc);
            st3(S.s.u.r.wv[b][h][i], add(ld3(S.e.body[b]), mulv(S.s.rot[b], l)));
          }
        }
      RV_LANES_END
    }
  }
  RV_PROF(18)
  RV_LANES_BEGIN
    DevEnv& e = S.e;
    const int slot = lane >> 4;
#if !RV_ON_DEVICE
    if ((lane & 15) != 0) continue;   // host emulation: one lane per group does the work
#endif
    RV_PROFG(5)   // (profiling build) time outside the narrow phase goes to a dump slot
    int my_pairs = 0;
    const int n_list = S.s.n_olist;
    for (int idx = slot; idx < n_list; idx += 4) {
      const int owner = S.s.olist[idx];
      OwnerInfo o;
      owner_decode(S, K, owner, arm_on, o);
      const int role = o.role, a = o.a, b = o.b, mi = o.mi, n_inner = o.n_inner;
      const int n_pairs = o.n_outer * o.n_inner;
      for (int t = 0; t < n_pairs; ++t) {
        int io = t / n_inner, ii = t - io * n_inner;
        const float* A; const float* B; int nA, nB, ckind, col = -1;
        v3 guess = o.guess0;
        if (role == 0) { A = &S.s.u.r.wv[a][ii][0][0]; nA = S.n_verts[a][ii]; B = body_below_table(e, c, a) ? &S.s.groundv[0][0] : &S.s.tablev[0][0]; nB = 8; ckind = 0; }
        else if (role == 1) { A = &S.s.u.r.wv[a][io][0][0]; nA = S.n_verts[a][io]; B = &S.s.u.r.wv[b][ii][0][0]; nB = S.n_verts[b][ii]; ckind = 1; }
        else if (role == 2) {
          col = io;
          if (!S.s.cn[a][col]) continue;
          A = &S.s.u.r.wv[a][ii][0][0]; nA = S.n_verts[a][ii]; B = &S.s.colv[col][0][0]; nB = 8; ckind = 2;
          guess = sub(ld3(e.body[a]), ld3(S.s.colc[col]));
        } else { col = a; A = &S.s.colv[col][0][0]; nA = 8; B = &S.s.tablev[0][0]; nB = 8; ckind = 0; }
        float dd;
        my_pairs++;
#ifdef RV_DEBUG_PAIRS
        fprintf(stderr, "P %d %d %d %d %d\n", e.sim_steps, ckind, role == 3 ? 0 : a, b, col);
#endif
        const float brk = role == 3 ? brk_col(arm, c, col) : brk_of(e, arm, c, ckind, a, b, col);
        // which pair of hulls of the manifold (the key of its simplex cache)
        const int pair = role == 0 ? ii + (body_below_table(e, c, a) ? 64 : 0) : (role == 1 ? io * 8 + ii : (role == 2 ? col * 8 + ii : 0));
        int hit = collide_pair(S, K, ckind, role == 3 ? 0 : a, b, col, A, nA, B, nB, guess, role == 3 ? nullptr : &e.man[mi], &dd, brk, pair);
        if (role == 3 && hit && dd < c->contact_query_dist) S.s.colflag[col] = 1;
#ifdef RV_EMU_COUNT
        // queries by role; of them: no contact found (separation beyond the breaking distance)
        rv_emu_dbg2[40 + role] += 1; if (!hit) rv_emu_dbg2[44 + role] += 1;
#endif
      }
    }
    if ((lane & 15) == 0) S.s.pairs[slot] = my_pairs;
  RV_LANES_END

  RV_STOP(3)
  RV_PROF(3)
  // PGS over islands: awake bodies coupled (transitively) by body-body manifolds
  // that hold points.  Every island stops on its own residual.  All rows of the env are
  // solved by the whole wave in impulse space, one lane per row (solve_rows above).
  int label[RV_MAXB], on_[RV_MAXB], act_[RV_NBB];
#pragma unroll
  for (int b = 0; b < RV_MAXB; ++b) { label[b] = b; on_[b] = body_on(S.e, b); }
  int on_mask = 0;
#pragma unroll
  for (int b = 0; b < RV_MAXB; ++b) on_mask |= on_[b] ? (1 << b) : 0;
#pragma unroll
  for (int k = 0; k < RV_NBB; ++k) act_[k] = on_[bb_a(k)] && on_[bb_b(k)] && S.e.man[RV_BBIDX(k)].n != 0;
#pragma unroll
  for (int pass = 0; pass < RV_MAXB; ++pass)
#pragma unroll
    for (int k = 0; k < RV_NBB; ++k) {
      const int a_ = bb_a(k), b_ = bb_b(k);
      if (!act_[k]) continue;
      const int lo = label[a_] < label[b_] ? label[a_] : label[b_];
      label[a_] = lo; label[b_] = lo;
    }
  // members of every island; islands of one or two bodies go to the impulse-space solver
  int mem_[RV_MAXB], big_[RV_MAXB];
#pragma unroll
  for (int b = 0; b < RV_MAXB; ++b) {
    int m_ = 0;
#pragma unroll
    for (int x = 0; x < RV_MAXB; ++x) m_ += (on_[x] && label[x] == b);
    mem_[b] = (on_[b] && label[b] == b) ? m_ : 0;
    big_[b] = mem_[b] > 2;
  }
#ifdef RV_EMU_COUNT
  { int n1 = 0, n2 = 0, n3 = 0; for (int b = 0; b < RV_MAXB; ++b) { n1 += mem_[b] == 1; n2 += mem_[b] == 2; n3 += mem_[b] > 2; }
    rv_emu_cnt[13] += n1; rv_emu_cnt[16] += n2; rv_emu_cnt[17] += n3;
    rv_emu_cnt[14] += (n3 > 0); rv_emu_cnt[15] += (n1 + n2 >= 2); }
#endif
#ifdef RV_EMU_COUNT
  { int arm_pts = 0, tab_pts = 0, slow = 1, near = 0;
    for (int b = 0; b < RV_MAXB; ++b) if (on_[b]) {
      arm_pts += S.e.man[RV_AIDX(b)].n; tab_pts += S.e.man[RV_TIDX(b)].n;
      if (len(ld3(S.e.body[b] + 7)) > 0.02f || len(ld3(S.e.body[b] + 10)) > 0.5f) slow = 0;
      for (int col = 0; col < RV_NCOL; ++col) near |= S.s.cn[b][col];
    }
    rv_emu_cnt[24] += arm_pts > 0; rv_emu_cnt[25] += (arm_pts == 0 && near); rv_emu_cnt[26] += (arm_pts == 0 && !near && S.s.arm_moving);
    rv_emu_cnt[27] += (arm_pts == 0 && !near && !S.s.arm_moving); rv_emu_cnt[28] += slow; rv_emu_cnt[29] += (slow && arm_pts == 0); }
#endif
  const int with_fingers = c->finger_dynamics && arm_on;
  // force-limited gripper: at most one awake body (a grasp scene) -> impulse space with the two
  // finger DOFs and their motor rows, one lane per row; else the velocity-space system solver
  int n_on = 0, the_body = -1;
#pragma unroll
  for (int b = RV_MAXB - 1; b >= 0; --b) if (on_[b]) { ++n_on; the_body = b; }
  // an awake body with a user constraint: everything goes through the velocity-space system solver
  int any_con = 0;
#pragma unroll
  for (int b = 0; b < RV_MAXB; ++b) any_con |= on_[b] && S.e.con_on[b];
  // limb dynamics: an awake body touches the arm -> the joint velocities are unknowns too (system solver)
  int limb = 0;
  if (c->limb_dynamics && arm_on) {
#pragma unroll
    for (int b = 0; b < RV_MAXB; ++b) limb |= on_[b] && S.e.man[RV_AIDX(b)].n > 0;
  }
  // one awake body at most and no user constraint: impulse space, one lane per row, with the finger / limb
  // DOFs and their motor rows; else the velocity-space system solver
  // ... or several awake bodies of which ONE touches the arm and is an island by itself (the pushed body while
  // another one is still sliding on): the limb rows live in that island, the other islands of one or two bodies
  // are the usual independent problems
  int lone = 0;
  if (limb) {
    int n_arm = 0, lb = -1, any_big = 0;
#pragma unroll
    for (int b = RV_MAXB - 1; b >= 0; --b) { if (on_[b] && S.e.man[RV_AIDX(b)].n > 0) { ++n_arm; lb = b; } any_big |= big_[b]; }
    int alone = 0;
#pragma unroll
    for (int b = 0; b < RV_MAXB; ++b) if (b == lb) alone = mem_[b] == 1;
    lone = n_arm == 1 && alone && !any_big && !with_fingers && n_on > 1;
    if (lone) the_body = lb;
  }
  const int fing_fast = (with_fingers || limb) && !any_con && (n_on <= 1 || lone);
  // the other islands keep their lane-per-row solvers unless the one-lane system solver takes everything
  const int others_ok = !with_fingers && !any_con && (!limb || lone);
  // islands that are one body on the table and nothing else are set up and solved together by
  // solve_singles(): their rows never go through the Row records
  int smask = 0;
  const int rows_all = ((with_fingers || limb) && !fing_fast) || (any_con && !limb);
  (void)rows_all;
#if RV_ON_DEVICE
  if (others_ok) {
#pragma unroll
    for (int b = 0; b < RV_MAXB; ++b) if (on_[b] && mem_[b] == 1 && !(lone && b == the_body)) smask |= 1 << b;     // (arm points or not)
    smask = __builtin_amdgcn_readfirstlane(smask);
  }
#endif
  // solver row setup (one lane per contact point) + contact flags
  RV_LANES_BEGIN
    DevEnv& e = S.e;
    if (lane < RV_NMAN * 4) {
      int mi = lane >> 2, i = lane & 3;
      DevMan& m = e.man[mi];
      int kind, a, b = -1;
      if (mi < RV_MAXB) { kind = 0; a = mi; }
      else if (mi < RV_MAXB + RV_NBB) { kind = 1; a = bb_a(mi - RV_MAXB); b = bb_b(mi - RV_MAXB); }
      else { kind = 2; a = mi - RV_MAXB - RV_NBB; }
      int use = body_on(e, a) && (kind != 1 || body_on(e, b));
#if RV_ON_DEVICE
      // the lane-per-row solvers (solve_singles, solve_island2) set their rows up themselves; Row records are
      // for the velocity-space paths only: an island of three or four bodies, the one-lane system solver
      // (and, for now, the finger / limb solver)
      {
        int big_a = 0;
#pragma unroll
        for (int x = 0; x < RV_MAXB; ++x) if (x == a) big_a = big_[label[x]];
        (void)big_a;
        if (!rows_all) use = 0;
      }
#endif
      if (use && i < m.n) {
        ManPoint p;
        p.la = ld3(m.la[i]); p.lb = ld3(m.lb[i]); p.nrm = ld3(m.nrm[i]); p.dist = m.dist[i]; p.col = m.col[i];
        p.ln = m.ln[i]; p.lt1 = m.lt1[i]; p.lt2 = m.lt2[i];
#if !RV_ON_DEVICE
        Row r;
        row_setup(S, K, kind, a, b, p, r, m.n);
        S.s.u.r.rows[mi][i] = r;
#endif
        m.ln[i] = p.ln * c->warmstart; m.lt1[i] = p.lt1 * c->warmstart; m.lt2[i] = p.lt2 * c->warmstart;
      }
    } else if (lane == 61) {
      e.pairs_last += S.s.pairs[0] + S.s.pairs[1] + S.s.pairs[2] + S.s.pairs[3];
    } else if (lane == 56) {
      int f = 0;
      if (arm_on) for (int col = 0; col < RV_NCOL; ++col) f |= S.s.colflag[col];
      e.flag_arm_table = f;
    } else if (lane >= 57 && lane < 57 + RV_MAXB) {
      int b = lane - 57; int f = 0;
      if (arm_on) {
        const DevMan& m = e.man[RV_AIDX(b)];
        for (int i = 0; i < m.n; ++i) if (m.dist[i] < c->contact_query_dist) f = 1;
      }
      e.flag_arm_body[b] = f;
    }
  RV_LANES_END

  RV_STOP(4)
  RV_PROF(4)
  any_con |= limb;
  if (limb) limb_prepare(S, K, fing_fast ? the_body : -1);
  if (((with_fingers || limb) && !fing_fast) || (any_con && !limb)) {
#if RV_ON_DEVICE
    {
      SerialRows SR;
      serial_rows_setup(S, K, limb, SR);
      __syncthreads();
      solve_with_fingers(S, K, limb, SR);      // (every lane: see SerialRows)
      __syncthreads();
    }
#else
    RV_LANES_BEGIN
      if (lane == 0) solve_with_fingers(S, K, limb);
    RV_LANES_END
#endif
  }
#if RV_ON_DEVICE
  if (fing_fast) solve_island_fingers(S, K, __builtin_amdgcn_readfirstlane(the_body), with_fingers, limb);
#endif
#if RV_ON_DEVICE
  {
    // the partner / pair of every two-body island, then ONE instance of the island solver in
    // the instruction stream, entered once per island of one or two bodies
    int isl_y[RV_MAXB], isl_k[RV_MAXB];
#pragma unroll
    for (int b = 0; b < RV_MAXB; ++b) {
      int y_ = -1;
#pragma unroll
      for (int x = RV_MAXB - 1; x > 0; --x) if (x > b && on_[x] && label[x] == b) y_ = x;
      int kxy = 0;
#pragma unroll
      for (int kk = 0; kk < RV_NBB; ++kk) if (bb_a(kk) == b && bb_b(kk) == y_) kxy = kk;
      isl_y[b] = mem_[b] == 2 ? y_ : -1; isl_k[b] = mem_[b] == 2 ? kxy : 0;
    }
    const int unrest = __builtin_amdgcn_readfirstlane(unrest_mask(S.e));
    if (smask) solve_singles(S, K, smask, unrest);
#pragma nounroll
    for (int b = 0; b < RV_MAXB; ++b) {
      if ((smask >> b) & 1) continue;
      int m_ = 0, y_ = -1, kxy = 0;
#pragma unroll
      for (int x = 0; x < RV_MAXB; ++x) if (x == b) { m_ = mem_[x]; y_ = isl_y[x]; kxy = isl_k[x]; }
      m_ = (!others_ok || (lone && b == the_body)) ? 0 : __builtin_amdgcn_readfirstlane(m_);
      if (m_ == 2) {       // (an island of one body is solve_singles')
        const int y1 = __builtin_amdgcn_readfirstlane(y_);
        solve_island2(S, K, b, y1, __builtin_amdgcn_readfirstlane(kxy), tol_of(c, unrest & ((1 << b) | (1 << y1))));
      }
    }
  }
#else
#define RV_EMU_SECTION 6
#include "../../tests/emu/rv_emu_hooks.h"      // host lane emulation (test scaffolding; not compiled into the product)
#undef RV_EMU_SECTION
#endif
  if (limb) { arm_lq_phase(S, K); arm_fk_phases(S, K); }   // the link frames follow the solved joint state
  // an island of three or four bodies (there can be only one): velocity-space Gauss-Seidel in the
  // order  bodies (their table and arm points), then the three rounds of body pairs.  Bodies do
  // not share anything in the first stage and the two pairs of a round touch disjoint bodies, so
  // one lane per body / per pair of the round runs them side by side: same arithmetic, same order.
  int big_root = -1;
#pragma unroll
  for (int b = RV_MAXB - 1; b >= 0; --b) if (big_[b]) big_root = b;
  RV_PROF(24)
#if RV_ON_DEVICE
  if (!with_fingers && !any_con && __builtin_amdgcn_readfirstlane(big_root) >= 0) {
    // Device: no Row records in LDS.  Lane 4 mi + i sets up the row set of point i of manifold mi in its
    // REGISTERS (row_setup(): what the row-setup phase computes) and scales the impulses kept from the last
    // substep; the body / pair lanes of the sweep below pull the row sets they visit out of those lanes with
    // ds_bpermute (the LDS crossbar: one instruction per word, any lane to any lane).
    const int root = __builtin_amdgcn_readfirstlane(big_root);
    const int lane = (int)threadIdx.x;
    DevEnv& e = S.e;
    Row my;
    {
      const int L = lane < RV_NMAN * 4 ? lane : RV_NMAN * 4 - 1;
      const int mi = L >> 2, i = L & 3;
      int kind, a, b;
      man_owner(mi, &kind, &a, &b);
      DevMan& m = e.man[mi];
      int in_big = 0;
#pragma unroll
      for (int x = 0; x < RV_MAXB; ++x) if (x == a) in_big = on_[x] && label[x] == root;
      const int use = lane < RV_NMAN * 4 && in_big && (kind != 1 || body_on(e, b)) && i < m.n;
      ManPoint pt;
      pt.la = ld3(m.la[i]); pt.lb = ld3(m.lb[i]); pt.nrm = ld3(m.nrm[i]); pt.dist = m.dist[i]; pt.col = m.col[i];
      row_setup(S, K, kind, a, b, pt, my, m.n);          // (lanes without a point compute a row nobody asks for)
      if (use) { m.ln[i] = m.ln[i] * c->warmstart; m.lt1[i] = m.lt1[i] * c->warmstart; m.lt2[i] = m.lt2[i] * c->warmstart; }
    }
    __syncthreads();
    auto pull = [&](float x, int src) { return __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(src << 2, __builtin_bit_cast(int, x))); };
    auto pull_row = [&](int src, bool pair) {
      Row r;
#pragma unroll
      for (int k = 0; k < 3; ++k) {
#pragma unroll
        for (int x = 0; x < 3; ++x) {
          r.dir[k][x] = pull(my.dir[k][x], src); r.rxa[k][x] = pull(my.rxa[k][x], src); r.aa[k][x] = pull(my.aa[k][x], src);
          if (pair) { r.rxb[k][x] = pull(my.rxb[k][x], src); r.ab[k][x] = pull(my.ab[k][x], src); } else { r.rxb[k][x] = 0.0f; r.ab[k][x] = 0.0f; }
        }
        r.invk[k] = pull(my.invk[k], src); r.vbc[k] = pull(my.vbc[k], src); r.jf[k] = 0.0f;
      }
      r.target = pull(my.target, src); r.mu = pull(my.mu, src); r.cap = pull(my.cap, src); r.fidx = -1;
      return r;
    };
    float big_best = 1e30f; int big_since = 0;        // rv_config.solver_stall
    float big_tol;                                     // the island's own tolerance (tol_of)
    {
      const int unrest = __builtin_amdgcn_readfirstlane(unrest_mask(S.e));
      int u_ = 0;
#pragma unroll
      for (int x = 0; x < RV_MAXB; ++x) if (on_[x] && label[x] == root) u_ |= (unrest >> x) & 1;
      big_tol = tol_of(c, __builtin_amdgcn_readfirstlane(u_));
    }
    const int iters = c->solver_iters;
    for (int it = -1; it < iters; ++it) {             // it == -1: warm start
      {
        const int b = lane < RV_MAXB ? lane : 0;
        int mine = 0;
#pragma unroll
        for (int x = 0; x < RV_MAXB; ++x) if (x == b) mine = on_[x] && label[x] == root;
        const bool active = lane < RV_MAXB && mine;
        float res = 0.0f;
        BV A = ld_bv(e, b); const float ima = e.inv_mass[b];
#pragma unroll
        for (int kind = 0; kind < 2; ++kind) {
          const int mi = kind == 0 ? RV_TIDX(b) : RV_AIDX(b);
          DevMan& m = e.man[mi];
          const int mn = m.n;
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const bool need = active && i < mn;
            if (__builtin_amdgcn_ballot_w64(need) == 0) continue;
            const Row r = pull_row(need ? mi * 4 + i : lane, false);
            if (need) {
              Lam l; l.n = m.ln[i]; l.t1 = m.lt1[i]; l.t2 = m.lt2[i];
              if (it < 0) warm_apply(A, nullptr, ima, 0.0f, l, r);
              else { res = fmaxr(res, point_solve(A, nullptr, ima, 0.0f, l, r)); m.ln[i] = l.n; m.lt1[i] = l.t1; m.lt2[i] = l.t2; }
            }
          }
        }
        if (active) st_bv(e, b, A);
        if (lane < RV_MAXB) S.s.res[lane] = res;
      }
      __syncthreads();
      for (int rd = 0; rd < 3; ++rd) {
        const int x = lane < 2 ? lane : 0;
        const int k = bb_round_pair(rd, x);
        const int a_ = bb_a(k), b_ = bb_b(k);
        int la_ = 0;
#pragma unroll
        for (int y = 0; y < RV_MAXB; ++y) if (y == a_) la_ = label[y];
        DevMan& m = e.man[RV_BBIDX(k)];
        const int mn = m.n;
        const bool active = lane < 2 && body_on(e, a_) && body_on(e, b_) && la_ == root && mn != 0;
        float res = 0.0f;
        BV A = ld_bv(e, a_), B = ld_bv(e, b_);
        const float ima = e.inv_mass[a_], imb = e.inv_mass[b_];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const bool need = active && i < mn;
          if (__builtin_amdgcn_ballot_w64(need) == 0) continue;
          const Row r = pull_row(need ? RV_BBIDX(k) * 4 + i : lane, true);
          if (need) {
            Lam l; l.n = m.ln[i]; l.t1 = m.lt1[i]; l.t2 = m.lt2[i];
            if (it < 0) warm_apply(A, &B, ima, imb, l, r);
            else { res = fmaxr(res, point_solve(A, &B, ima, imb, l, r)); m.ln[i] = l.n; m.lt1[i] = l.t1; m.lt2[i] = l.t2; }
          }
        }
        if (active) { st_bv(e, a_, A); st_bv(e, b_, B); }
        if (lane < 2) S.s.res[4 + 2 * rd + lane] = res;
        __syncthreads();
      }
      float res = 0.0f;
#pragma unroll
      for (int t = 0; t < 10; ++t) res = fmaxr(res, S.s.res[t]);
      res = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, res)));
      if (it >= 0 && res < big_tol) break;
      if (it >= 0 && c->solver_stall > 0) { if (res < big_best) { big_best = res; big_since = 0; } else if (++big_since >= c->solver_stall) break; }
      __syncthreads();          // (S.s.res is written again by the next sweep)
    }
  }
#else
#define RV_EMU_SECTION 7
#include "../../tests/emu/rv_emu_hooks.h"      // host lane emulation (test scaffolding; not compiled into the product)
#undef RV_EMU_SECTION
#endif
  RV_STOP(5)
  RV_PROF(5)
  // integrate positions, freeze fallen bodies, counters
  RV_LANES_BEGIN
    DevEnv& e = S.e;
    if (lane < RV_MAXB) {
      int b = lane;
      if (body_on(e, b)) {
        float dt = c->dt;
        // rolling / spinning friction on the support: a resisting angular impulse of at most
        // rolling_friction x (normal impulse of the table manifold), never reversing the spin
        if (c->rolling_friction > 0.0f && e.man[RV_TIDX(b)].n > 0) {
          const DevMan& mt = e.man[RV_TIDX(b)];
          float nimp = 0.0f;
          for (int i = 0; i < mt.n; ++i) nimp += mt.ln[i];
          v3 w0 = ld3(e.body[b] + 10);
          float wl = len(w0);
          if (wl > 0.0f && nimp > 0.0f) {
            v3 dir = scale(w0, 1.0f / wl);
            v3 iw = mulv(S.s.iinv[b], dir);
            float k = dot(dir, iw);
            float j = fminr(c->rolling_friction * nimp, wl / k);
            st3(e.body[b] + 10, madd(w0, iw, -j));
          }
        }
        v3 v = ld3(e.body[b] + 7), w = ld3(e.body[b] + 10);
        v3 p = ld3(e.body[b]);
        q4 q = ldq(e.body[b] + 3);
        if (!body_static(e, b)) {      // (a static body stays where it is: its velocities are zero, no impulse changes them)
        p = madd(p, v, dt);
        st3(e.body[b], p);
        q4 wq; wq.x = w.x; wq.y = w.y; wq.z = w.z; wq.w = 0.0f;
        q4 dq = qmul(wq, q);
        q.x += 0.5f * dt * dq.x; q.y += 0.5f * dt * dq.y; q.z += 0.5f * dt * dq.z; q.w += 0.5f * dt * dq.w;
        q = qnormalize(q);
        stq(e.body[b] + 3, q);
        if (p.z < c->ground_z - c->fall_depth) {
          e.frozen[b] = 1;
          st3(e.body[b] + 7, mk(0, 0, 0)); st3(e.body[b] + 10, mk(0, 0, 0));
        }
        }
        // substeps in a row below the sleep thresholds (the deactivation counter; also what makes a row a 'rest' row of the solver)
        if (dot(v, v) < c->sleep_lin * c->sleep_lin && dot(w, w) < c->sleep_ang * c->sleep_ang) e.sleep_count[b]++;
        else { e.sleep_count[b] = 0; }
        if (c->sleep_steps > 0) {
          // in-place oscillation: the pose has not left a small window around where
          // it was when the window opened
          if (c->sleep_pos_win > 0.0f) {
            int inside = 0;
            if (e.still_count[b] > 0) {
              v3 dp = sub(p, ld3(e.still_ref[b]));
              float dqm = fmaxr(fmaxr(fmaxr(fmaxr(0.0f, fabsr(q.x - e.still_ref[b][3])), fabsr(q.y - e.still_ref[b][4])),
                                      fabsr(q.z - e.still_ref[b][5])), fabsr(q.w - e.still_ref[b][6]));
              inside = dot(dp, dp) < c->sleep_pos_win * c->sleep_pos_win && dqm < c->sleep_rot_win;
            }
            if (inside) e.still_count[b]++;
            else { e.undisturbed[b] = 0; e.still_count[b] = 1; st3(e.still_ref[b], p); stq(e.still_ref[b] + 3, q); }
          }
          // a sleeper that was woken but never left the pose it was resting in goes back
          // to sleep after a quarter of the usual wait
          const int quick = e.undisturbed[b] && 4 * e.still_count[b] >= c->sleep_steps && 4 * e.sleep_count[b] >= c->sleep_steps;
          // a body the force-limited gripper holds stays active (its island contains the moving fingers)
          const int held = (c->finger_dynamics && e.man[RV_AIDX(b)].n > 0) || con_pair_member(e, b);
          // Bullet's own rule (0.8 m/s, 1 rad/s, 2 s) for a body whose island is the body alone:
          // no arm contact points, no contact points with another awake body
          int deact = 0;
          if (c->deact_steps > 0) {
            int free_ = e.man[RV_AIDX(b)].n == 0;
#pragma unroll
            for (int k = 0; k < RV_NBB; ++k) {
              const int a_ = bb_a(k), b_ = bb_b(k);
              if ((a_ == b || b_ == b) && e.man[RV_BBIDX(k)].n != 0 && ((on_mask >> (a_ == b ? b_ : a_)) & 1)) free_ = 0;
            }
            if (free_ && dot(v, v) < c->deact_lin * c->deact_lin && dot(w, w) < c->deact_ang * c->deact_ang) e.deact_count[b]++;
            else e.deact_count[b] = 0;
            deact = e.deact_count[b] >= c->deact_steps;
          }
          S.s.ready[b] = e.frozen[b] || (!held && (e.sleep_count[b] >= c->sleep_steps || e.still_count[b] >= c->sleep_steps || quick || deact));
        }
      }
    }
    if (lane == 32) { e.sim_steps++; e.substeps_last++; }
#if RV_ON_DEVICE
  }
  // islands go to sleep as a whole: a body sleeps when every awake body it is coupled to
  // (transitively) by manifolds that hold points is ready as well (the ready flags of the four
  // body lanes travel in a ballot: no LDS round trip)
  if (c->sleep_steps > 0) {
    const int lane = (int)threadIdx.x;
    DevEnv& e = S.e;
    const int rb = lane < RV_MAXB ? lane : 0;
    const int my_ready = lane < RV_MAXB && S.s.ready[rb];   // (its own store; lanes of bodies that were not awake hold stale flags nobody looks at)
    const unsigned rmask = (unsigned)__builtin_amdgcn_ballot_w64(my_ready != 0);
    if (lane < RV_MAXB) {
      int b = lane;
      int mine = 0, all = 1;
#pragma unroll
      for (int x = 0; x < RV_MAXB; ++x) if (x == b) mine = on_[x] && label[x] >= 0;
      int lb = 0;
#pragma unroll
      for (int x = 0; x < RV_MAXB; ++x) if (x == b) lb = label[x];
#pragma unroll
      for (int x = 0; x < RV_MAXB; ++x) if (on_[x] && label[x] == lb && !((rmask >> x) & 1u)) all = 0;
      if (mine && ((rmask >> b) & 1u) && !e.frozen[b] && all) {
#else
#define RV_EMU_SECTION 8
#include "../../tests/emu/rv_emu_hooks.h"      // host lane emulation (test scaffolding; not compiled into the product)
#undef RV_EMU_SECTION
#endif
        {
          {
            const v3 p = ld3(e.body[b]); const q4 q = ldq(e.body[b] + 3);
            e.asleep[b] = 1;
            st3(e.body[b] + 7, mk(0, 0, 0)); st3(e.body[b] + 10, mk(0, 0, 0));
            // world box of the resting hulls: what the arm has to come near to wake the body
            const rv_shape* sh = &K.scene->shapes[e.shape[b]];
            const m3 m = qmat(q);
            const float sc = e.scale[b];
            float lo[3] = {1e30f, 1e30f, 1e30f}, hi[3] = {-1e30f, -1e30f, -1e30f};
            for (int h = 0; h < S.n_hulls[b]; ++h)
              for (int i = 0; i < S.n_verts[b][h]; ++i) {
                v3 l = mk(sh->verts[h][i][0] * sc, sh->verts[h][i][1] * sc, sh->verts[h][i][2] * sc);
                v3 pw = add(p, mulv(m, l));
                lo[0] = fminr(lo[0], pw.x); lo[1] = fminr(lo[1], pw.y); lo[2] = fminr(lo[2], pw.z);
                hi[0] = fmaxr(hi[0], pw.x); hi[1] = fmaxr(hi[1], pw.y); hi[2] = fmaxr(hi[2], pw.z);
              }
            for (int k = 0; k < 3; ++k) { e.baabb[b][k] = lo[k] - c->margin; e.baabb[b][3 + k] = hi[k] + c->margin; }
            for (int col = 0; col < RV_NCOL; ++col) S.s.sep[b][col] = 0.0f;   // new resting pose
          }
        }
      }
    }
#if RV_ON_DEVICE
  }
  __syncthreads();
#else
  RV_LANES_END
  }
#endif
}

// The env block lives in ONE statically addressed LDS object so that the
// (large) substep body can be a real function with a single copy in the
// instruction stream instead of being inlined at every call site.
#if RV_ON_DEVICE
__shared__ Shared g_shared;
#else
static thread_local Shared g_shared;
#endif
#if defined(RV_PROFILE) && defined(__HIP_DEVICE_COMPILE__)
__device__ __forceinline__ void g_shared_prof(int i, unsigned long long t) {
  g_shared.e.prof[i] += t - g_shared.e.prof_t; g_shared.e.prof_t = t;
}
__device__ __forceinline__ void g_shared_cnt(int i, int n) { g_shared.e.prof[i] += (unsigned long long)n; }
__device__ __forceinline__ void g_shared_profg(int g, int i, unsigned long long t) {
  g_shared.e.prof[12 + g * 6 + i] += t - g_shared.e.prof_t2[g]; g_shared.e.prof_t2[g] = t;
}
#endif
// Consts whose cfg / arm point at the LDS copies (statically known address)
RV_DEV Consts lds_consts(const rv_scene* scene, int stop_after) {
  Consts K; K.cfg = &g_shared.cfg; K.arm = &g_shared.arm; K.scene = scene; K.stop_after = stop_after;
  return K;
}
#if defined(RV_COAST_NOINLINE) && RV_ON_DEVICE
RV_DEV_NOINLINE int coast_run_fn(const rv_scene* scene, int stop_after, int steps_check, int want) {
  const Consts K = lds_consts(scene, stop_after);
  int why = 0;
  const int n = coast_run_body(g_shared, K, steps_check, want, &why);
  return n | (why << 28);
}
#endif
// Simulator.check_stable over a body mask (simulator.py:289-323)
RV_DEV int bodies_stable(const DevEnv& e, unsigned mask, float lin_thr, float ang_thr) {
  for (int b = 0; b < RV_MAXB; ++b) {
    if (!((mask >> b) & 1u) || !e.active[b]) continue;
    float lv = len(ld3(e.body[b] + 7)), av = len(ld3(e.body[b] + 10));
    if (lv >= lin_thr || av >= ang_thr) return 0;
  }
  return 1;
}
RV_DEV unsigned active_mask(const DevEnv& e) {
  unsigned m = 0;
  for (int b = 0; b < RV_MAXB; ++b) if (e.active[b]) m |= 1u << b;
  return m;
}
RV_DEV void phase_tick(Shared& S, const Consts& K);
RV_DEV void gphase_tick(Shared& S, const Consts& K);
// Runs substeps inside ONE out-of-line function, so that the call overhead
// (callee-saved registers: ~140 VGPRs saved and restored per call) is paid per call
// and not per substep, and the light part exists once in the instruction stream.
//   n_fixed > 0 : exactly n_fixed times Simulator.step
//   n_fixed == 0: Simulator.wait_until_stable (simulator.py:325-376); mask == 0 => all active
//   n_fixed < 0 : the phase loop of PushEnv._execute_action (push_env.py:648-719: step
//                 until the next multiple of STEPS_CHECK, phase_tick, until 'done'),
//                 followed by its closing wait_until_stable -- one call per env.step()
#ifdef RV_HEAVY_NOINLINE      // (an option of the two-waves-per-SIMD build: the heavy part as a function of its own)
RV_DEV_NOINLINE void sim_substep_heavy_fn(const rv_scene* scene, int stop_after) { const Consts K = lds_consts(scene, stop_after); sim_substep_heavy(g_shared, K); }
RV_DEV void sim_substep_heavy_call(Shared& S, const Consts& K) { (void)S; sim_substep_heavy_fn(K.scene, K.stop_after); }
#endif
// What one run of the substep loop is asked to do (the arguments of the former sim_run_call)
struct RunReq { int n_arg; unsigned mask; float lin_thr, ang_thr; int check_after, min_stable, max_steps; };
RV_DEV RunReq run_req(int n_arg, unsigned mask = 0u, float lin = 0.0f, float ang = 0.0f, int ca = 0, int ms = 0, int mx = 0) {
  RunReq r; r.n_arg = n_arg; r.mask = mask; r.lin_thr = lin; r.ang_thr = ang; r.check_after = ca; r.min_stable = ms; r.max_steps = mx;
  return r;
}
// The substep loop.  It exists ONCE in every env kernel: env_program() -- the reset / env.step() / rollout logic
// written as a resumable program -- hands it one request after the other from a single call site, so nothing of it
// is a function call (until round 4 it was an out-of-line function entered per env.step(), and its heavy part a
// second one entered per awake substep: each call saved and restored ~200 registers through scratch memory).
RV_DEV void sim_run_body(Shared& S, const Consts& K, const RunReq& rq) {
  const rv_scene* scene = K.scene; const int stop_after = K.stop_after; (void)scene; (void)stop_after;
  const int n_arg = rq.n_arg; const unsigned mask = rq.mask; const float lin_thr = rq.lin_thr, ang_thr = rq.ang_thr;
  const int check_after = rq.check_after, min_stable = rq.min_stable, max_steps = rq.max_steps;
  int phase_mode = n_arg < 0;
  const int grasp_mode = n_arg == -2;     // Grasp4DofEnv: its phase machine looks at the world after EVERY substep
  // rv_step_poll: a budget of substeps and / or shader clocks for this launch; the call may return
  // between any two substeps (S.s.suspended) and is entered again by the next launch -- the
  // substep sequence, and so every result, is the one of an uninterrupted call
#if RV_ON_DEVICE
  const int bud_sub = __builtin_amdgcn_readfirstlane(S.s.bud_sub);
  const unsigned long long bud_clk = S.s.bud_clk;
#else
  const int bud_sub = S.s.bud_sub; const unsigned long long bud_clk = 0;
#endif
  const int bud_any = bud_sub > 0 || bud_clk != 0;
  int suspended = 0;
  for (;;) {                  // one pass, except in phase mode
  int n_fixed = n_arg;
  if (phase_mode) {
    if (S.e.phase == RV_PHASE_DONE) {
      if (grasp_mode) break;              // (the reward's wait_until_stable comes after the observation)
      phase_mode = 0; n_fixed = 0;        // -> closing wait_until_stable
    } else n_fixed = grasp_mode ? 1 : K.cfg->steps_check - (S.e.sim_steps % K.cfg->steps_check);
  }
  if (n_fixed == 0) {
    RV_LANES_BEGIN
      if (lane == 0) {
        if (S.s.wus_resume) { S.s.wus_steps = S.e.ms_wus_steps; S.s.wus_stable = S.e.ms_wus_stable; S.s.wus_resume = 0; }
        else { S.s.wus_steps = 0; S.s.wus_stable = 0; }
        S.s.loop_break = 0;
        if (n_arg == -1) S.e.step_stage = 1;      // the closing wait_until_stable of env.step()
      }
    RV_LANES_END
  }
  int taken = 0;
  int coast_wait = 0;     // regular substeps to take before coasting is considered again
  int coast_fail = 0;     // attempts in a row that bought nothing: the wait doubles (1, 2, 4, 8)
  for (;;) {
    RV_PROF(7)
    int bud_left = 1 << 30;
    if (bud_any) {
      int stop = 0;
      if (bud_sub > 0) { bud_left = bud_sub - (S.e.substeps_last - S.s.bud_sub0); if (bud_left <= 0) stop = 1; }
#if RV_ON_DEVICE
      if (bud_clk != 0) { if (__builtin_amdgcn_s_memtime() - S.s.bud_t0 > bud_clk) stop = 1; if (bud_left > 128) bud_left = 128; }
#endif
      if (stop) { suspended = 1; break; }
    }
    if (phase_mode && coast_wait == 0) {
      // free-space motion: as many substeps (and ticks that change nothing) as provably possible
      // (Grasp4DofEnv: a tick after every substep)
      int why = 0;
      const int n = coast_run(S, K, grasp_mode ? 1 : K.cfg->steps_check, bud_left, &why);
      if (n > 0) {
        coast_fail = 0;
        if (why == 2) break;               // the phase machine has something to do at this tick
        n_fixed = grasp_mode ? 1 : K.cfg->steps_check - (S.e.sim_steps % K.cfg->steps_check); taken = 0;
      }
      // the substep the loop stopped at is a regular one; after attempts that bought nothing
      // (awake bodies / close to something) step normally for a while
      if (n == 0) { coast_wait = 1 << (coast_fail < 3 ? coast_fail : 3); ++coast_fail; }
    } else if (!phase_mode && n_fixed > 0 && coast_wait == 0) {
      int why = 0;
      const int n = coast_run(S, K, 0, n_fixed - taken, &why);
      taken += n;
      if (taken >= n_fixed) break;
      if (n > 0) coast_fail = 0; else { coast_wait = 1 << (coast_fail < 3 ? coast_fail : 3); ++coast_fail; }
    } else if (n_fixed > 0 && coast_wait == 0) {
      // (Grasp4DofEnv ticks its phase machine every substep: chunks of one do not coast)
      int kidx = 0;
      int m = coast_budget(S, K, n_fixed - taken, &kidx);
      if (m < 0) { arm_refresh_kinematics(S, K); m = coast_budget(S, K, n_fixed - taken, &kidx); }
      RV_PROF(10)
      if (m >= 2) {
        coast_substeps(S, K, m, kidx);
        RV_PROF(11)
        taken += m;
        if (taken >= n_fixed) break;
        continue;
      }
      coast_wait = 8;
    }
    if (n_fixed == 0 && coast_wait == 0) {
      // wait_until_stable with every body asleep: the stability verdict cannot change
      // while coasting, so the loop's counters can be advanced m substeps at once
      const unsigned mk_ = mask ? mask : active_mask(S.e);
      const int st_ok = bodies_stable(S.e, mk_, lin_thr, ang_thr);
      int T = 0;
      {
        int st = S.s.wus_steps, sb = S.s.wus_stable, fin = 0;
        const int tmax = bud_left < 256 ? bud_left : 256;
        while (T < tmax && !fin) {
          ++T; ++st;
          if (st >= check_after) { if (st_ok) ++sb; if (sb >= min_stable || st >= max_steps) fin = 1; }
        }
      }
      int m = 0;
      {
        int any_on = 0;
#pragma unroll
        for (int b = 0; b < RV_MAXB; ++b) any_on |= body_on(S.e, b);
        if (!S.e.arm_enabled && !any_on && K.stop_after == 0) {
          // no arm, nothing awake: these substeps only count
          m = T;
          RV_LANES_BEGIN
            DevEnv& e = S.e;
            if (lane == 0) { e.flag_arm_table = 0; e.sim_steps += m; e.substeps_last += m; }
            if (lane >= 1 && lane <= RV_MAXB) e.flag_arm_body[lane - 1] = 0;
          RV_LANES_END
        } else {
          int why = 0;
          m = coast_run(S, K, 0, T, &why);
        }
      }
      if (m == 0) {
        const int T16 = T < 16 ? T : 16;
        int kidx = 0;
        m = coast_budget(S, K, T16, &kidx);
        if (m < 0) { arm_refresh_kinematics(S, K); m = coast_budget(S, K, T16, &kidx); }
        if (m >= 2) coast_substeps(S, K, m, kidx); else m = 0;
      }
      if (m > 0) {
        RV_LANES_BEGIN
          if (lane == 0) {
            int ws = S.s.wus_steps, wb = S.s.wus_stable, lb = 0;
            for (int i = 0; i < m; ++i) {
              ws++;
              if (ws >= check_after) {
                if (st_ok) wb++;
                if (wb >= min_stable || ws >= max_steps) lb = 1;
              }
            }
            S.s.wus_steps = ws; S.s.wus_stable = wb; if (lb) S.s.loop_break = 1;
          }
        RV_LANES_END
        RV_PROF(0)
        coast_fail = 0;
        if (S.s.loop_break) break;
        continue;
      }
      coast_wait = 1 << (coast_fail < 3 ? coast_fail : 3); ++coast_fail;
    }
    if (coast_wait > 0) --coast_wait;
    RV_CNT(9, 1)
    if (sim_substep_light(S, K)) {
      RV_CNT(10, 1)
      RV_PROF(1)
#ifdef RV_HEAVY_NOINLINE
      sim_substep_heavy_call(S, K);
#else
      sim_substep_heavy(S, K);
#endif
      RV_PROF(6)
    } else {
      RV_PROF(0)
    }
    if (n_fixed > 0) {
      if (++taken >= n_fixed) break;
      continue;
    }
    RV_LANES_BEGIN
      if (lane == 0) {
        S.s.wus_steps++;
        if (S.s.wus_steps >= check_after) {
          unsigned mk_ = mask ? mask : active_mask(S.e);
          if (bodies_stable(S.e, mk_, lin_thr, ang_thr)) S.s.wus_stable++;
          if (S.s.wus_stable >= min_stable || S.s.wus_steps >= max_steps) S.s.loop_break = 1;
        }
      }
    RV_LANES_END
    if (S.s.loop_break) break;
  }
  if (suspended) break;
  if (!phase_mode) break;
  // the phase machine looks at the world every STEPS_CHECK substeps (push_env.py:652-661);
  // it reads the end-effector frame only in these situations
  if (!S.s.kin_fresh && S.e.arm_enabled &&
      (grasp_mode || S.e.phase == RV_PHASE_START || S.e.phase == RV_PHASE_MOTION || S.s.interrupt)) arm_refresh_kinematics(S, K);
  RV_PROF(8)
  RV_LANES_BEGIN
    if (lane == 0) { if (grasp_mode) gphase_tick(S, K); else phase_tick(S, K); }
  RV_LANES_END
  RV_PROF(9)
  }
  if (!S.s.kin_fresh && S.e.arm_enabled) arm_refresh_kinematics(S, K);   // leave with frames that match the joints
  if (bud_any) {
    RV_LANES_BEGIN
      if (lane == 0) S.s.suspended = suspended;
    RV_LANES_END
  }
}
#ifdef RV_SIM_RUN_NOINLINE
// The two-waves-per-SIMD build keeps the loop as ONE function (a boundary for the register allocator under its
// 256-register cap).  It addresses the env through the file-scope LDS object and rebuilds Consts from scalars: through
// pointer arguments the LDS accesses would be compiled as flat loads.
RV_DEV_NOINLINE void sim_run_fn(const rv_scene* scene, int stop_after, int n_arg, unsigned mask, float lin_thr, float ang_thr,
                                int check_after, int min_stable, int max_steps) {
  const Consts K = lds_consts(scene, stop_after);
  sim_run_body(g_shared, K, run_req(n_arg, mask, lin_thr, ang_thr, check_after, min_stable, max_steps));
}
RV_DEV void sim_run(Shared& S, const Consts& K, const RunReq& rq) {
  (void)S;
  sim_run_fn(K.scene, K.stop_after, rq.n_arg, rq.mask, rq.lin_thr, rq.ang_thr, rq.check_after, rq.min_stable, rq.max_steps);
}
#else
RV_DEV void sim_run(Shared& S, const Consts& K, const RunReq& rq) { sim_run_body(S, K, rq); }
#endif

// ------------------------------------------------- observation / reward --
RV_DEV void compute_obs(DevEnv& e) {
  for (int b = 0; b < RV_MAXB; ++b)
    for (int k = 0; k < 3; ++k) {
      e.prev_obs_pos[b][k] = e.obs_pos[b][k];
      e.obs_pos[b][k] = body_movable(e, b) ? e.body[b][k] : 0.0f;      // (self.movable_bodies: a static body is not one)
    }
  for (int b = 0; b < RV_MAXB; ++b)
    for (int k = 0; k < 4; ++k) e.obs_quat[b][k] = e.body[b][3 + k];
}
// pose snapshot for the point-cloud render (rv_dev_obs.h)
RV_DEV void obs_snap_arm(const DevEnv& e, const rv_arm* arm, ObsSnap& s);
// at_obs: the body poses as they were when the env took its observation (RobotEnv.step observes BEFORE get_reward,
// robot_env.py:246-248 -- GraspReward then waits until the object is stable) instead of the current ones
RV_DEV void obs_snap_fill(const DevEnv& e, const rv_arm* arm, ObsSnap& s, const int at_obs = 0) {
  for (int b = 0; b < RV_MAXB; ++b) {
    for (int k = 0; k < 7; ++k) s.pose[b][k] = !at_obs ? e.body[b][k] : (k < 3 ? e.obs_pos[b][k] : e.obs_quat[b][k - 3]);
    s.scale[b] = e.scale[b];
    s.shape[b] = e.active[b] ? e.shape[b] : -1;
    s.is_static[b] = body_static(e, b);
  }
  s.table_z = e.table_z;
  for (int k = 0; k < 5; ++k) s.cam_intrinsics[k] = e.cam_intrinsics[k];
  for (int k = 0; k < 9; ++k) s.cam_rotation[k] = e.cam_rotation[k];
  for (int k = 0; k < 3; ++k) s.cam_translation[k] = e.cam_translation[k];
  s.rng_arg = (uint32_t)e.reset_count * 4096u + (uint32_t)e.num_steps;
  s.arm_on = e.arm_enabled;
  if (s.arm_on) obs_snap_arm(e, arm, s);
}
// ... and the arm's link boxes (world centre, frame quaternion) from the link frames
RV_DEV void obs_snap_arm(const DevEnv& e, const rv_arm* arm, ObsSnap& s) {
  for (int col = 0; col < RV_NCOL; ++col) {
    const int f = arm->col_frame[col];
    const q4 q = ldq(e.fquat[f]);
    st3(s.arm_c[col], add(ld3(e.fpos[f]), mulv(qmat(q), ld3(arm->col_center[col]))));
    stq(s.arm_q[col], q);
  }
}
// one observation row: PoseObs in its four modalities (pose_obs.py:53-73), the attribute
// observations (attribute_obs.py:16-115; env.attributes snapshot), body mask.  NULL members
// are skipped.  e == nullptr writes the zero row of a step that was not taken.
RV_DEV void obs_write_row(const DevEnv* e, const rv_obs_buffers& o, size_t row, const rv_config* cfg) {
  for (int b = 0; b < RV_MAXB; ++b) {
    const size_t ib = row * RV_MAXB + b;
    const int on = e ? body_movable(*e, b) : 0;
    float pos[3] = {0.0f, 0.0f, 0.0f}, eu[3] = {0.0f, 0.0f, 0.0f};
    if (e) for (int k = 0; k < 3; ++k) pos[k] = e->obs_pos[b][k];
    if (o.d_position) for (int k = 0; k < 3; ++k) o.d_position[ib * 3 + k] = pos[k];
    if (o.d_body_mask) o.d_body_mask[ib] = (float)on;
    if (o.d_pose || o.d_pose2d || o.d_yaw_cossin) {
      if (on) quat_to_euler(ldq(e->body[b] + 3), eu);
      if (o.d_pose) for (int k = 0; k < 3; ++k) { o.d_pose[ib * 6 + k] = pos[k]; o.d_pose[ib * 6 + 3 + k] = eu[k]; }
      if (o.d_pose2d) { o.d_pose2d[ib * 3] = pos[0]; o.d_pose2d[ib * 3 + 1] = pos[1]; o.d_pose2d[ib * 3 + 2] = eu[2]; }
      if (o.d_yaw_cossin) {
        float sn = 0.0f, cs = 0.0f;
        if (on) sincosr(eu[2], &sn, &cs);
        o.d_yaw_cossin[ib * 2] = cs; o.d_yaw_cossin[ib * 2 + 1] = sn;
      }
    }
  }
  if (o.d_num_episodes) o.d_num_episodes[row] = e ? e->obs_num_episodes : 0;
  if (o.d_num_steps) o.d_num_steps[row] = e ? e->obs_num_steps : 0;
  if (o.d_layout_id) o.d_layout_id[row] = e ? cfg->layout_id : 0;
  if (o.d_is_safe) o.d_is_safe[row] = e ? e->is_safe : 0;
  if (o.d_is_effective) o.d_is_effective[row] = e ? e->is_effective : 0;
}
// what a rollout records per env.step(): reward, done, observation row, pose snapshot
struct RolloutRec {
  float* rewards; uint8_t* dones;
  rv_obs_buffers obs; int has_obs;
  ObsSnap* snaps;
  // rv_rollout_record_full: the action of every step, the observation after every auto-reset
  float* actions; uint8_t* resets;
  rv_obs_buffers robs; int has_robs;
  ObsSnap* rsnaps;
};
RV_DEV void rollout_record(const RolloutRec& r, const DevEnv* e, size_t row, const rv_config* cfg, const rv_arm* arm) {
  if (r.rewards) r.rewards[row] = e ? e->last_reward : 0.0f;
  if (r.dones) r.dones[row] = (uint8_t)(e ? e->done : 1);
  if (r.has_obs) obs_write_row(e, r.obs, row, cfg);
  if (r.snaps) {
    if (e) obs_snap_fill(*e, arm, r.snaps[row], 1);
    else { for (int b = 0; b < RV_MAXB; ++b) r.snaps[row].shape[b] = -1; r.snaps[row].arm_on = 0; }
  }
}

RV_DEV int on_tiles(const float* xy, const float (*tiles)[2], int n, float size, const float* offset, float max_dist) {
  for (int i = 0; i < n; ++i) {
    float tx = offset[0] + tiles[i][0] * size;
    float ty = offset[1] + tiles[i][1] * size;
    if (fabsr(xy[0] - tx) <= 0.5f * max_dist && fabsr(xy[1] - ty) <= 0.5f * max_dist) return 1;
  }
  return 0;
}
RV_DEV float tile_dist(const float* xy, const float (*tiles)[2], int n, float size, const float* offset) {
  float best = 1e30f;
  for (int i = 0; i < n; ++i) {
    float dx = xy[0] - (offset[0] + tiles[i][0] * size);
    float dy = xy[1] - (offset[1] + tiles[i][1] * size);
    float d = fsqrtr(dx * dx + dy * dy);
    if (d < best) best = d;
  }
  return best;
}
RV_DEV float task_score(const rv_config* c, const float (*st)[3]) {
  float size = c->tile_size;
  if (c->task == RV_TASK_CLEARING) {
    float d1 = 0.0f, d3 = 0.0f;
    for (int b = 0; b < RV_MAXB; ++b) { d1 += fabsr(st[b][0] - 0.7f); d3 += fabsr(st[b][1] + 0.9f); }
    d1 /= (float)RV_MAXB; d3 /= (float)RV_MAXB;
    return -fminr(d1, d3);
  }
  return -tile_dist(st[0], c->goal, c->n_goal, size, c->tile_offset);
}
// get_reward_fn(...).reward_fn, is_planning=False (push_reward.py:302-372)
RV_DEV void compute_reward(const DevEnv& e, const rv_config* c, float* reward, int* termination) {
  if (c->task == RV_TASK_NONE) { *reward = 1.0f; *termination = 0; return; }
  float size = c->tile_size;
  int term = 0, goal = 0;
  if (c->task == RV_TASK_CROSSING)
    term = !on_tiles(e.obs_pos[0], c->region, c->n_region, size, c->tile_offset, size * 1.5f);
  if (c->task == RV_TASK_CLEARING) {
    goal = 1;
    for (int b = 0; b < RV_MAXB; ++b)
      if (on_tiles(e.obs_pos[b], c->region, c->n_region, size * 1.25f, c->tile_offset, size * 1.25f)) goal = 0;
  } else {
    goal = on_tiles(e.obs_pos[0], c->goal, c->n_goal, size, c->tile_offset, size);
  }
  goal = goal && !term;
  float r = 0.0f;
  r += 100.0f * (float)goal;
  int penalty = term && !goal;
  r += -100.0f * (float)penalty;
  r += fabsr(task_score(c, e.obs_pos) - task_score(c, e.prev_obs_pos));
  r += -1.0f;
  *reward = r; *termination = term || goal;
}

// ---------------------------------------------------------------- PushEnv --
RV_DEV void set_gripper_pose(float* pose, float x, float y, float z) {
  pose[0] = x; pose[1] = y; pose[2] = z;
  stq(pose + 3, euler_to_quat(RV_PI, 0.0f, 0.0f));
}
// PushEnv._compute_waypoints (push_env.py:752-786)
RV_DEV void compute_waypoints(const rv_config* c, const float* action, float* start, float* end) {
  float lo0 = c->cspace_low[0], hi0 = c->cspace_high[0];
  float lo1 = c->cspace_low[1], hi1 = c->cspace_high[1];
  float lo2 = c->cspace_low[2], hi2 = c->cspace_high[2];
  float x = action[0] * (0.5f * (hi0 - lo0)) + 0.5f * (hi0 + lo0);
  float y = action[1] * (0.5f * (hi1 - lo1)) + 0.5f * (hi1 + lo1);
  float z = c->finger_tip_offset + 0.5f * (hi2 + lo2);
  set_gripper_pose(start, x, y, z);
  float ex = fclampr(x + action[2] * c->translation_x, lo0, hi0);
  float ey = fclampr(y + action[3] * c->translation_y, lo1, hi1);
  set_gripper_pose(end, ex, ey, z);
}
RV_DEV int arm_touches_movables(const DevEnv& e) {
  for (int b = 0; b < RV_MAXB; ++b) if (body_movable(e, b) && e.flag_arm_body[b]) return 1;
  return 0;
}
// PushEnv._check_safety (push_env.py:857-898)
RV_DEV int check_safety(const DevEnv& e, const rv_config* c, float start_z) {
  if (e.phase == RV_PHASE_PRE) { if (arm_touches_movables(e)) return 0; }
  if (e.phase == RV_PHASE_START) {
    if (arm_touches_movables(e)) {
      float dist = e.fpos[7][2] - start_z;
      if (fabsr(dist) <= 0.01f) return 1;
      return 0;
    }
  }
  if (e.phase == RV_PHASE_DONE) {
    if (arm_touches_movables(e)) return 0;
    float lx = c->table_center[0] - 0.5f * c->workspace_x_range, hx = c->table_center[0] + 0.5f * c->workspace_x_range;
    float ly = c->table_center[1] - 0.5f * c->workspace_y_range, hy = c->table_center[1] + 0.5f * c->workspace_y_range;
    for (int b = 0; b < RV_MAXB; ++b) {
      if (!body_movable(e, b)) continue;
      const float* p = e.body[b];
      if (p[0] < lx || p[0] > hx || p[1] < ly || p[1] > hy) return 0;
    }
  }
  return 1;
}

// one phase-machine tick of PushEnv._execute_action (push_env.py:662-720), lane 0
RV_DEV void phase_tick(Shared& S, const Consts& K) {
  DevEnv& e = S.e; Scratch& s = S.s; const rv_config* c = K.cfg;
  float start_z = c->finger_tip_offset + 0.5f * (c->cspace_high[2] + c->cspace_low[2]);
  int ready = 0;
  if (s.interrupt) ready = 1;
  else if (arm_is_ready_limb(S, K) && (sim_time(S, K) >= e.gripper_ready_time)) { arm_reset_targets(e); ready = 1; }
  else if (!s.has_budget) ready = 1;
  else if (e.sim_steps >= s.max_phase_steps) { arm_reset_targets(e); ready = 1; }
  if (ready) {
    int next;
    if (s.interrupt && e.phase != RV_PHASE_POST && e.phase != RV_PHASE_OFFSTAGE) next = RV_PHASE_POST;
    else if (c->num_goal_steps > 0 && e.phase == RV_PHASE_POST && s.num_waypoints < c->num_goal_steps) next = RV_PHASE_PRE;
    else next = e.phase + 1;
    e.phase = next;
    s.has_budget = 1;
    s.max_phase_steps = e.sim_steps;
    if (next == RV_PHASE_MOTION) s.max_phase_steps += c->max_motion_steps;
    else if (next == RV_PHASE_OFFSTAGE) s.max_phase_steps += c->max_offstage_steps;
    else s.max_phase_steps += c->max_phase_steps;
    if (next == RV_PHASE_PRE) {
      float pose[7];
      for (int k = 0; k < 7; ++k) pose[k] = s.wp[s.num_waypoints][0][k];
      pose[2] = c->gripper_safe_height;
      robot_move_to_gripper_pose(S, K, pose);
    } else if (next == RV_PHASE_START) {
      robot_move_to_gripper_pose(S, K, s.wp[s.num_waypoints][0]);
    } else if (next == RV_PHASE_MOTION) {
      robot_move_to_gripper_pose(S, K, s.wp[s.num_waypoints][1]);
    } else if (next == RV_PHASE_POST) {
      s.num_waypoints++;
      float pose[7];
      for (int k = 0; k < 3; ++k) pose[k] = e.fpos[7][k];
      for (int k = 0; k < 4; ++k) pose[3 + k] = e.fquat[7][k];
      pose[2] = c->gripper_safe_height;
      robot_move_to_gripper_pose(S, K, pose);
    } else if (next == RV_PHASE_OFFSTAGE) {
      float off[RV_NLIMB];
      for (int j = 0; j < RV_NLIMB; ++j) off[j] = c->offstage_positions[j];
      robot_move_to_joint_positions(S, K, off);
    }
  }
  s.interrupt = 0;
  if (e.phase == RV_PHASE_MOTION && e.flag_arm_table) s.interrupt = 1;
  if (!check_safety(e, c, start_z)) { s.interrupt = 1; e.is_safe = 0; }
  if (s.interrupt && e.phase == RV_PHASE_DONE) e.done = 1;
}

// RobotEnv.step (robot_env.py:239-275) + PushEnv._execute_action
// (push_env.py:631-733) + PushEnv.step bookkeeping (push_env.py:599-629)
RV_DEV void env_step_prologue(Shared& S, const Consts& K, int zero_counters, int count_step) {
  const rv_config* c = K.cfg;
  RV_LANES_BEGIN
    if (lane == 0) {
      DevEnv& e = S.e; Scratch& s = S.s;
      if (zero_counters) { e.substeps_last = 0; e.awake_last = 0; e.pairs_last = 0; }
      if (count_step) { e.stepped += 1; e.reward_valid = 1; }
      e.step_stage = 0;
      e.obs_num_steps = e.num_steps; e.obs_num_episodes = e.num_episodes;
      int G = c->num_goal_steps > 0 ? c->num_goal_steps : 1;
      for (int g = 0; g < G; ++g) compute_waypoints(c, e.action[g], s.wp[g][0], s.wp[g][1]);
      e.is_safe = 1; e.is_effective = 1;
      e.phase = RV_PHASE_INITIAL;
      s.num_waypoints = 0; s.interrupt = 0; s.has_budget = 0; s.max_phase_steps = 0;
      for (int b = 0; b < RV_MAXB; ++b) {
        for (int k = 0; k < 3; ++k) s.start_pos[b][k] = e.body[b][k];
        s.start_yaw[b] = quat_yaw(ldq(e.body[b] + 3));
      }
    }
  RV_LANES_END
  RV_PROF(29)
}
RV_DEV void env_step_epilogue(Shared& S, const Consts& K) {
  const rv_config* c = K.cfg;
  RV_LANES_BEGIN
    if (lane == 0) {
      DevEnv& e = S.e; Scratch& s = S.s;
      e.step_stage = -1;
      // _check_effectiveness (push_env.py:900-923)
      float dpos = 0.0f, dang = 0.0f;
      for (int b = 0; b < RV_MAXB; ++b) {
        if (!e.active[b]) continue;
        dpos += len(sub(ld3(e.body[b]), ld3(s.start_pos[b])));
        float da = quat_yaw(ldq(e.body[b] + 3)) - s.start_yaw[b];
        da = da + RV_PI;
        da = da - 2.0f * RV_PI * ffloorr(da / (2.0f * RV_PI));
        da = da - RV_PI;
        dang += fabsr(da);
      }
      if (dpos <= c->min_delta_position && dang <= c->min_delta_angle) e.is_effective = 0;
      e.num_total_steps++;
      e.num_unsafe += !e.is_safe; e.l_unsafe += !e.is_safe;
      e.num_ineffective += !e.is_effective; e.l_ineffective += !e.is_effective;
      e.num_useful += (e.is_safe && e.is_effective); e.l_useful += (e.is_safe && e.is_effective);
      e.num_steps++;
      compute_obs(e);
      float r; int term;
      compute_reward(e, c, &r, &term);
      e.last_reward = r;
      e.episode_reward += r;
      e.done = e.done || term;
      if (c->max_steps > 0 && e.num_steps >= c->max_steps) e.done = 1;
      if (e.done) {
        e.num_episodes++; e.l_episodes++;
        if (r >= c->success_thresh) { e.num_successes++; e.l_successes++; }
      }
    }
  RV_LANES_END
  RV_PROF(31)
}
// (env.step() = env_step_prologue, one run of the substep loop with run_req(-1, ...), env_step_epilogue: env_program)
// rv_step_poll: the env.step() this env is in the middle of (S.e.in_step == 1), continued within the budget set
// in S.s.bud_*: what comes before the run of the substep loop ...
RV_DEV void env_pstep_begin(Shared& S, const Consts& K) {
  if (S.e.step_stage < 0) env_step_prologue(S, K, 0, 0);
  else {
    RV_LANES_BEGIN
      if (lane == 0) {
        DevEnv& e = S.e; Scratch& s = S.s;
        for (int g = 0; g < RV_MAXG; ++g) for (int x = 0; x < 2; ++x) for (int k = 0; k < 7; ++k) s.wp[g][x][k] = e.ms_wp[g][x][k];
        for (int b = 0; b < RV_MAXB; ++b) { for (int k = 0; k < 3; ++k) s.start_pos[b][k] = e.ms_start_pos[b][k]; s.start_yaw[b] = e.ms_start_yaw[b]; }
        s.num_waypoints = e.ms_num_waypoints; s.interrupt = e.ms_interrupt; s.has_budget = e.ms_has_budget; s.max_phase_steps = e.ms_max_phase_steps;
        s.wus_resume = e.step_stage == 1;
      }
    RV_LANES_END
  }
}
// ... and after it; returns 1 when the step completed in this launch
RV_DEV int env_pstep_end(Shared& S, const Consts& K) {
  if (S.s.suspended) {
    RV_LANES_BEGIN
      if (lane == 0) {
        DevEnv& e = S.e; const Scratch& s = S.s;
        for (int g = 0; g < RV_MAXG; ++g) for (int x = 0; x < 2; ++x) for (int k = 0; k < 7; ++k) e.ms_wp[g][x][k] = s.wp[g][x][k];
        for (int b = 0; b < RV_MAXB; ++b) { for (int k = 0; k < 3; ++k) e.ms_start_pos[b][k] = s.start_pos[b][k]; e.ms_start_yaw[b] = s.start_yaw[b]; }
        e.ms_num_waypoints = s.num_waypoints; e.ms_interrupt = s.interrupt; e.ms_has_budget = s.has_budget; e.ms_max_phase_steps = s.max_phase_steps;
        e.ms_wus_steps = s.wus_steps; e.ms_wus_stable = s.wus_stable;
      }
    RV_LANES_END
    return 0;
  }
  env_step_epilogue(S, K);
  RV_LANES_BEGIN
    if (lane == 0) { S.e.in_step = 0; S.e.stepped += 1; S.e.reward_valid = 1; }
  RV_LANES_END
  return 1;
}

// ------------------------------------------------------------- Grasp4DofEnv --
// SawyerSim.move_along_gripper_path -> set_target_link_poses (sawyer_sim.py:310-360,
// controllable_body.py:319-345): the poses are reached one after the other
RV_DEV void robot_move_along_gripper_path(Shared& S, const Consts& K, const float (*poses)[7], int n) {
  DevEnv& e = S.e; const rv_config* c = K.cfg;
  arm_reset_targets(e);
  for (int j = 0; j < RV_NLIMB; ++j) e.vmax_cmd[j] = c->limb_max_velocity_ratio * K.arm->v_max[j];
  LTarget& t = e.lt;
  t.active = 1; t.has_pose = 0; t.nq = n;
  for (int q = 0; q < n; ++q) for (int k = 0; k < 7; ++k) t.queue[q][k] = poses[q][k];
  t.start_t = sim_time(S, K); t.stop_t = t.start_t + c->limb_timeout; t.has_stop = 1;
  t.pos_thr = c->limb_position_threshold; t.vel_thr = c->velocity_threshold;
  lt_pop(t);
}
// SawyerSim.move_to_gripper_pose(pose, straight_line=True) (sawyer_sim.py:259-276): way points
// every END_EFFECTOR_STEP along the segment from the current end-effector position, all with
// the orientation of the target
RV_DEV void robot_move_straight(Shared& S, const Consts& K, const float* pose) {
  DevEnv& e = S.e; const rv_config* c = K.cfg;
  v3 p0 = ld3(e.fpos[7]), d = sub(ld3(pose), p0);
  int num = (int)(len(d) / c->end_effector_step);
  if (num > RV_MAXQ - 1) num = RV_MAXQ - 1;
  float wps[RV_MAXQ][7];
  for (int i = 0; i < num; ++i) {
    float sc = (float)i / (float)num;
    st3(wps[i], madd(p0, d, sc));
    for (int k = 3; k < 7; ++k) wps[i][k] = pose[k];
  }
  for (int k = 0; k < 7; ++k) wps[num][k] = pose[k];
  robot_move_along_gripper_path(S, K, wps, num + 1);
}
// one pass of the loop of Grasp4DofEnv._execute_action (grasp_4dof_env.py:234-293) after
// simulator.step(): _is_phase_ready (:322-345), _get_next_phase (:295-320) and the commands
// of the phase that starts; lane 0
RV_DEV void gphase_tick(Shared& S, const Consts& K) {
  DevEnv& e = S.e; Scratch& s = S.s; const rv_config* c = K.cfg;
  if (e.phase == RV_GPHASE_START) e.num_action_steps++;
  int ready = 0;
  if (e.phase == RV_GPHASE_START && e.num_action_steps >= c->max_action_steps) ready = 1;     // the grasping motion is stuck
  else if ((e.phase == RV_GPHASE_START || e.phase == RV_GPHASE_END) && e.flag_arm_table) ready = 1;   // the gripper contacts the table
  else if (arm_is_ready_limb(S, K) && sim_time(S, K) >= e.gripper_ready_time) ready = 1;
  if (!ready) return;
  e.phase = e.phase + 1;
  if (e.phase == RV_GPHASE_OVERHEAD) {
    float q[RV_NLIMB];
    for (int j = 0; j < RV_NLIMB; ++j) q[j] = c->overhead_positions[j];
    robot_move_to_joint_positions(S, K, q);
  } else if (e.phase == RV_GPHASE_PRESTART) {
    float pose[7];
    for (int k = 0; k < 7; ++k) pose[k] = s.gstart[k];
    pose[2] = c->gripper_safe_height;
    robot_move_to_gripper_pose(S, K, pose);
  } else if (e.phase == RV_GPHASE_START) {
    robot_move_straight(S, K, s.gstart);
    e.mu_finger = c->grasp_mu_descend[0]; e.mu_table = c->grasp_mu_descend[1];   // "prevent problems caused by unrealistic frictions"
  } else if (e.phase == RV_GPHASE_END) {
    robot_grip(S, K, 1.0f);
  } else if (e.phase == RV_GPHASE_POSTEND) {
    float pose[7];
    for (int k = 0; k < 3; ++k) pose[k] = e.fpos[7][k];
    for (int k = 0; k < 4; ++k) pose[3 + k] = e.fquat[7][k];
    pose[2] = c->gripper_safe_height;
    robot_move_straight(S, K, pose);
    e.mu_finger = c->grasp_mu_lift[0]; e.mu_table = c->grasp_mu_lift[1];
  }
}
// RobotEnv.step (robot_env.py:239-275) for Grasp4DofEnv: _execute_action (grasp_4dof_env.py:213-293),
// observation, GraspReward.get_reward (grasp_reward.py:49-68: wait until the object is stable, success =
// the arm still touches it; the episode ends after one grasp)
// (three segments around two runs of the substep loop -- run_req(-2): the phase loop; then the reward's
// wait_until_stable -- see env_program)
RV_DEV void genv_step_begin(Shared& S, const Consts& K, int zero_counters, int count_step = 1) {
  const rv_config* c = K.cfg;
  RV_LANES_BEGIN
    if (lane == 0) {
      DevEnv& e = S.e; Scratch& s = S.s;
      if (zero_counters) { e.substeps_last = 0; e.awake_last = 0; e.pairs_last = 0; }
      if (count_step) { e.stepped += 1; e.reward_valid = 1; }
      e.step_stage = 0;
      e.obs_num_steps = e.num_steps; e.obs_num_episodes = e.num_episodes;
      // start = Pose([[x, y, z + FINGER_TIP_OFFSET], [0, pi, angle]])
      s.gstart[0] = e.action[0][0]; s.gstart[1] = e.action[0][1]; s.gstart[2] = e.action[0][2] + c->finger_tip_offset;
      stq(s.gstart + 3, euler_to_quat(0.0f, RV_PI, e.action[0][3]));
      e.is_safe = 1; e.is_effective = 1;
      e.phase = RV_GPHASE_INITIAL; e.num_action_steps = 0;
    }
  RV_LANES_END
}
RV_DEV void genv_step_observe(Shared& S) {
  RV_LANES_BEGIN
    if (lane == 0) { S.e.num_steps++; compute_obs(S.e); }
  RV_LANES_END
}
RV_DEV void genv_step_end(Shared& S) {
  RV_LANES_BEGIN
    if (lane == 0) {
      DevEnv& e = S.e;
      e.step_stage = -1;
      const int success = arm_touches_movables(e);
      const float r = success ? 1.0f : 0.0f;
      e.is_effective = success;
      e.num_total_steps++;
      e.num_useful += success; e.l_useful += success;
      e.num_ineffective += !success; e.l_ineffective += !success;
      e.last_reward = r; e.episode_reward += r;
      e.done = 1;                         // terminate_after_grasp
      e.num_episodes++; e.l_episodes++;
      if (success) { e.num_successes++; e.l_successes++; }
    }
  RV_LANES_END
}

// rv_step_poll on a Grasp4DofEnv (grasp_4dof_env.py:213-293 driven in partial batches): the env.step() this env is in the
// middle of, continued within the launch's budget.  step_stage: -1 not begun, 0 in the phase loop, 1 in the reward's
// wait_until_stable.  What the phase machine keeps outside the env block is the start pose -- a function of the action --
// and the two counters of wait_until_stable (ms_wus_*), so a suspended step resumes exactly where it stopped.
// Returns which run of the substep loop comes next: 0 the phase loop, 1 the closing wait
RV_DEV int genv_pstep_begin(Shared& S, const Consts& K) {
  const int stage = RV_UNI(S.e.step_stage);
  if (stage < 0) { genv_step_begin(S, K, 0, 0); return 0; }
  RV_LANES_BEGIN
    if (lane == 0) {
      DevEnv& e = S.e; Scratch& s = S.s; const rv_config* c = K.cfg;
      s.gstart[0] = e.action[0][0]; s.gstart[1] = e.action[0][1]; s.gstart[2] = e.action[0][2] + c->finger_tip_offset;
      stq(s.gstart + 3, euler_to_quat(0.0f, RV_PI, e.action[0][3]));
      s.wus_resume = stage == 1;
    }
  RV_LANES_END
  return stage == 1;
}
// after the phase loop: 0 = suspended in it (the launch ends), 1 = it is over: observation taken, the closing wait comes next
RV_DEV int genv_pstep_observe(Shared& S) {
  if (S.s.suspended) return 0;
  genv_step_observe(S);
  RV_LANES_BEGIN
    if (lane == 0) S.e.step_stage = 1;
  RV_LANES_END
  return 1;
}
// after the closing wait: 1 when the step completed in this launch
RV_DEV int genv_pstep_end(Shared& S) {
  if (S.s.suspended) {
    RV_LANES_BEGIN
      if (lane == 0) { S.e.ms_wus_steps = S.s.wus_steps; S.e.ms_wus_stable = S.s.wus_stable; }
    RV_LANES_END
    return 0;
  }
  genv_step_end(S);
  RV_LANES_BEGIN
    if (lane == 0) { S.e.in_step = 0; S.e.stepped += 1; S.e.reward_valid = 1; }
  RV_LANES_END
  return 1;
}

// --------------------------------------------------------------- reset ---
// rejection sampling of one layout (push_env.py:473-597, intent version), lane 0
RV_DEV void sample_poses(Shared& S, const Consts& K, int nb) {
  const rv_config* c = K.cfg; DevEnv& e = S.e; Rng& g = S.s.rng;
  for (;;) {
    int ok = 1;
    for (int i = 0; i < nb && ok; ++i) {
      int placed = 0;
      for (int att = 0; att <= 32 && !placed; ++att) {
        float x, y, z, roll, pitch, yaw;
        if (c->use_tiles) {
          int use_t = (i == 0 && c->n_target > 0);
          const float (*tiles)[2] = use_t ? c->target : c->obstacle;
          int nt = use_t ? c->n_target : c->n_obstacle;
          int tid = rng_randint(g, nt);
          float cx = c->tile_offset[0] + tiles[tid][0] * c->tile_size;
          float cy = c->tile_offset[1] + tiles[tid][1] * c->tile_size;
          x = rng_uniform(g, cx - 0.5f * c->tile_size, cx + 0.5f * c->tile_size);
          y = rng_uniform(g, cy - 0.5f * c->tile_size, cy + 0.5f * c->tile_size);
          z = e.table_z + c->safe_drop_height;
          roll = rng_uniform(g, -RV_PI, RV_PI);
          pitch = rng_uniform(g, -0.5f * RV_PI, 0.5f * RV_PI);
          yaw = rng_uniform(g, -RV_PI, RV_PI);
        } else {
          x = rng_uniform(g, c->pose_lo[0], c->pose_hi[0]);
          y = rng_uniform(g, c->pose_lo[1], c->pose_hi[1]);
          z = e.table_z + rng_uniform(g, c->pose_lo[2], c->pose_hi[2]);
          roll = rng_uniform(g, c->pose_lo[3], c->pose_hi[3]);
          pitch = rng_uniform(g, c->pose_lo[4], c->pose_hi[4]);
          yaw = rng_uniform(g, c->pose_lo[5], c->pose_hi[5]);
        }
        int valid = 1;
        for (int j = 0; j < i; ++j) {
          float dx = x - S.s.poses[j][0], dy = y - S.s.poses[j][1];
          if (fsqrtr(dx * dx + dy * dy) < c->margin_xy) { valid = 0; break; }
        }
        if (valid) {
          S.s.poses[i][0] = x; S.s.poses[i][1] = y; S.s.poses[i][2] = z;
          stq(S.s.poses[i] + 3, euler_to_quat(roll, pitch, yaw));
          placed = 1;
        }
      }
      if (!placed) ok = 0;
    }
    if (ok) return;
  }
}

// RobotEnv.reset for one env (robot_env.py:204-237), in the segments env_program runs between the settle waits
// (each wait is one request to the substep loop).  (1) counters, table height, body count; Grasp4DofEnv: its object
// ArmEnv._reset_camera (arm_env.py:109-152): the calibration of rv_config plus uniform noise, element by element.  A
// stream of its own: the scene of an episode does not depend on whether the camera is perturbed (noise 0: x + 0 = x).
RV_DEV void camera_reset(DevEnv& e, const rv_config* c, int gid, int use_noise) {
  Rng g = rng_init(c->seed_lo, c->seed_hi, (uint32_t)gid, RV_STREAM_CAMERA, (uint32_t)e.reset_count);
  for (int k = 0; k < 17; ++k) {
    const float base = k < 5 ? c->cam_intrinsics[k] : (k < 14 ? c->cam_rotation[k - 5] : c->cam_translation[k - 14]);
    const float nz = use_noise ? rng_uniform(g, -c->cam_noise[k], c->cam_noise[k]) : 0.0f;
    const float v = base + nz;
    if (k < 5) e.cam_intrinsics[k] = v; else if (k < 14) e.cam_rotation[k - 5] = v; else e.cam_translation[k - 14] = v;
  }
}
// ArmEnv._reset_scene's wall (arm_env.py:94-99): simulator.add_body(SIM.WALL.PATH, SIM.WALL.POSE, is_static=True) -- a
// static body (mass 0) in the last body slot (lane 0)
RV_DEV void wall_place(Shared& S, const Consts& K) {
  const rv_config* c = K.cfg; DevEnv& e = S.e;
  if (!c->wall_use) return;
  const int b = RV_MAXB - 1;
  e.active[b] = 1; e.frozen[b] = 0; e.asleep[b] = 0; e.sleep_count[b] = 0; e.deact_count[b] = 0; e.still_count[b] = 0; e.undisturbed[b] = 0; e.con_on[b] = 0;
  e.shape[b] = c->wall_shape; e.scale[b] = c->wall_scale; e.friction[b] = 1.0f;     // (URDF template default, tools/templates/urdf_template.xml:11-22)
  body_set_mass(e, K, b, 0.0f);
  cache_shape_meta(S, K, b);
  for (int k = 0; k < 7; ++k) e.body[b][k] = c->wall_pose[k];
  for (int k = 7; k < 13; ++k) e.body[b][k] = 0.0f;
}
RV_DEV void env_reset_begin(Shared& S, const Consts& K, int gid, int zero_counters) {
  const rv_config* c = K.cfg;
  RV_LANES_BEGIN
    DevEnv& e = S.e;
    if (lane == 0) {
      // per-launch scratch state that a reset kernel does not get from env_enter
      S.s.jt_applied = 0; S.s.kin_fresh = 0; S.s.clr_valid = 0; S.s.bud_sub = 0; S.s.bud_clk = 0; S.s.suspended = 0; S.s.wus_resume = 0; S.s.far_valid = 0; S.s.far_n = 0; S.s.far = 0;
      e.in_step = 0; e.step_stage = -1;
      S.s.rng = rng_init(c->seed_lo, c->seed_hi, (uint32_t)gid, RV_STREAM_RESET, (uint32_t)e.reset_count);
      camera_reset(e, c, gid, 1);
      e.reset_count++;
      if (zero_counters) launch_counters_zero(e);
      e.reward_valid = 0;
      e.sim_steps = 0; e.num_steps = 0; e.obs_num_steps = 0; e.obs_num_episodes = e.num_episodes; e.episode_reward = 0.0f; e.last_reward = 0.0f;
      e.done = 0; e.phase = RV_PHASE_INITIAL; e.is_safe = 1; e.is_effective = 1;
      e.arm_enabled = 0;
      e.mu_finger = c->arm_friction; e.mu_table = c->table_friction; e.num_action_steps = 0;
      e.flag_arm_table = 0;
      for (int b = 0; b < RV_MAXB; ++b) e.flag_arm_body[b] = 0;
      // ArmEnv._reset_scene (arm_env.py:78-99)
      e.table_z = c->table_z + rng_uniform(S.s.rng, c->table_height_range[0], c->table_height_range[1]);
      e.n_bodies = c->n_bodies_min + rng_randint(S.s.rng, c->n_bodies_max - c->n_bodies_min + 1);
      S.s.valid = 0;
    }
    if (lane < RV_NMAN) e.man[lane].n = 0;
  RV_LANES_END
  RV_LANES_BEGIN
    if (lane < 8) table_prepare(S, K, lane);
  RV_LANES_END
  if (c->env_type == RV_ENV_GRASP) {
    // Grasp4DofEnv._reset_scene (grasp_4dof_env.py:166-198): one graspable object at
    // Pose.uniform(SIM.GRASPABLE.POSE) with a uniform scale, then wait_until_stable(graspable)
    RV_LANES_BEGIN
      DevEnv& e = S.e;
      if (lane == 0) {
        Rng& g = S.s.rng;
        for (int b = 0; b < RV_MAXB; ++b) { e.active[b] = 0; e.frozen[b] = 0; e.asleep[b] = 0; e.sleep_count[b] = 0; e.deact_count[b] = 0; e.still_count[b] = 0; e.undisturbed[b] = 0; e.con_on[b] = 0; }
        e.n_bodies = 1;
        wall_place(S, K);
        sample_poses(S, K, 1);
        int shape = c->movable_shapes[rng_randint(g, c->n_movable_shapes)];
        float sc = rng_uniform(g, c->scale_range[0], c->scale_range[1]);
        float mass = rng_uniform(g, c->mass_range[0], c->mass_range[1]);
        float fr = rng_uniform(g, c->friction_range[0], c->friction_range[1]);
        e.active[0] = 1; e.shape[0] = shape; e.scale[0] = sc; e.friction[0] = fr;
        body_set_mass(e, K, 0, mass);
        cache_shape_meta(S, K, 0);
        for (int k = 0; k < 7; ++k) e.body[0][k] = S.s.poses[0][k];
        for (int k = 7; k < 13; ++k) e.body[0][k] = 0.0f;
        S.s.valid = 1;
      }
    RV_LANES_END
  }
}
// (2) PushEnv._load_movable_bodies (push_env.py:399-471): a new layout (while the last one was not valid)
RV_DEV void env_reset_layout(Shared& S, const Consts& K) {
  RV_LANES_BEGIN
    DevEnv& e = S.e;
    if (lane == 0) {
      for (int b = 0; b < RV_MAXB; ++b) { e.active[b] = 0; e.frozen[b] = 0; e.asleep[b] = 0; e.sleep_count[b] = 0; e.deact_count[b] = 0; e.still_count[b] = 0; e.undisturbed[b] = 0; e.con_on[b] = 0; }
      wall_place(S, K);
      sample_poses(S, K, e.n_bodies);
    }
    if (lane >= 1 && lane <= RV_NMAN) e.man[lane - 1].n = 0;
  RV_LANES_END
}
// (3) body i is dropped; then wait_until_stable(body i, 0.1, 0.1, 100, 100, 500)
RV_DEV void env_reset_place(Shared& S, const Consts& K, const int i) {
  const rv_config* c = K.cfg;
  RV_LANES_BEGIN
    if (lane == 0) {
      DevEnv& e = S.e; Rng& g = S.s.rng;
      int use_target = (i == 0 && c->use_tiles && c->n_target > 0 && c->n_target_shapes > 0);
      int shape = use_target ? c->target_shapes[rng_randint(g, c->n_target_shapes)]
                             : c->movable_shapes[rng_randint(g, c->n_movable_shapes)];
      float sc = rng_uniform(g, c->scale_range[0], c->scale_range[1]);
      e.active[i] = 1; e.frozen[i] = 0; e.asleep[i] = 0; e.sleep_count[i] = 0; e.deact_count[i] = 0; e.still_count[i] = 0; e.undisturbed[i] = 0; e.shape[i] = shape; e.scale[i] = sc; e.friction[i] = c->drop_friction;
      body_set_mass(e, K, i, c->drop_mass);
      cache_shape_meta(S, K, i);
      for (int k = 0; k < 3; ++k) e.body[i][k] = S.s.poses[i][k];
      for (int k = 0; k < 4; ++k) e.body[i][3 + k] = S.s.poses[i][3 + k];
      for (int k = 7; k < 13; ++k) e.body[i][k] = 0.0f;
    }
  RV_LANES_END
}
// (4) ... and gets its mass and friction
RV_DEV void env_reset_after_body(Shared& S, const Consts& K, const int i) {
  const rv_config* c = K.cfg;
  RV_LANES_BEGIN
    if (lane == 0) {
      DevEnv& e = S.e; Rng& g = S.s.rng;
      float mass = rng_uniform(g, c->mass_range[0], c->mass_range[1]);
      float fr = rng_uniform(g, c->friction_range[0], c->friction_range[1]);
      body_set_mass(e, K, i, mass); e.friction[i] = fr;
    }
  RV_LANES_END
}
// (5) a body that ended below the table top invalidates the layout
RV_DEV void env_reset_validate(Shared& S, const int nb) {
  RV_LANES_BEGIN
    if (lane == 0) {
      int valid = 1;
      for (int i = 0; i < nb; ++i) if (S.e.body[i][2] < S.e.table_z) valid = 0;
      S.s.valid = valid;
    }
  RV_LANES_END
}
// (6) after the closing wait_until_stable(all, 0.005, 0.005, 100, 100, 2000): the robot
RV_DEV void env_reset_robot(Shared& S, const Consts& K) {
  const rv_config* c = K.cfg;
  // ArmEnv._reset_robot (arm_env.py:101-107) -> SawyerSim.reboot (sawyer_sim.py:86-171)
  RV_LANES_BEGIN
    if (lane == 0) {
      DevEnv& e = S.e; const rv_arm* a = K.arm;
      for (int j = 0; j < RV_NLIMB; ++j) { e.q[j] = c->neutral_positions[j]; e.qd[j] = 0.0f; }
      e.q[7] = a->q_hi[7]; e.q[8] = a->q_lo[8]; e.qd[7] = 0.0f; e.qd[8] = 0.0f;
      for (int j = 0; j < RV_NJ; ++j) { e.motor_on[j] = 0; e.motor_q[j] = e.q[j]; e.motor_kp[j] = c->kp; e.motor_kd[j] = c->kd; e.vmax_cmd[j] = a->v_max[j]; }
      S.s.jt_applied = 0;
      arm_reset_targets(e);
      e.gripper_ready_time = 0.0f;
      e.arm_enabled = 1;
      if (c->open_gripper_when_reset) robot_grip(S, K, 0.0f);
      float off[RV_NLIMB];
      for (int j = 0; j < RV_NLIMB; ++j) off[j] = c->offstage_positions[j];
      robot_move_to_joint_positions(S, K, off);
      if (c->env_type == RV_ENV_GRASP) {
        // Grasp4DofEnv._reset_robot (grasp_4dof_env.py:206-211): robot.reset(OFFSTAGE_POSITIONS) =
        // move_to_joint_positions, then grip(0) -- whose finger target REPLACES the limb target
        // (one JointTarget per body, controllable_body.py:263-300)
        robot_move_to_joint_positions(S, K, off);
        robot_grip(S, K, 0.0f);
      }
      compute_obs(e);
      for (int b = 0; b < RV_MAXB; ++b) for (int k = 0; k < 3; ++k) e.prev_obs_pos[b][k] = e.obs_pos[b][k];
      // link frames of the rebooted arm, for getters before the first substep
      LimbFK F;
      fk_limb(a, e.q, F, &S.s.frot[0][0]);
      for (int i = 0; i <= RV_NLIMB; ++i) { st3(e.fpos[i], F.pos[i]); stq(e.fquat[i], F.quat[i]); }
      v3 yax = mk(S.s.frot[7][1], S.s.frot[7][4], S.s.frot[7][7]);
      for (int k = 0; k < 2; ++k) {
        st3(e.fpos[8 + k], madd(F.pos[7], yax, a->finger_y0[k] + e.q[7 + k]));
        stq(e.fquat[8 + k], F.quat[7]);
      }
    }
  RV_LANES_END
}

// RandomPolicy._action (random_policy.py:14-23) = action_space.sample(): PushEnv U(-1,1)^(G*4)
// (push_env.py:253-267); Grasp4DofEnv uniform in ACTION.CUBOID x [0, 2 pi] (grasp_4dof_env.py:144-150).
// Philox keyed by (seed, global env id, macro index).
RV_DEV void random_action(const rv_config* c, int gid, int macro_index, float* a /* [G][4] */) {
  int G = c->num_goal_steps > 0 ? c->num_goal_steps : 1;
  Rng g = rng_init(c->seed_lo, c->seed_hi, (uint32_t)gid, RV_STREAM_RANDOM, (uint32_t)macro_index);
  for (int x = 0; x < G * 4; ++x) a[x] = rng_uniform(g, -1.0f, 1.0f);
  if (c->env_type == RV_ENV_GRASP) {
    for (int k = 0; k < 3; ++k) a[k] = c->grasp_cuboid_low[k] + (c->grasp_cuboid_high[k] - c->grasp_cuboid_low[k]) * (0.5f * (a[k] + 1.0f));
    a[3] = RV_PI * (a[3] + 1.0f);
  }
}
// ---- the env program ------------------------------------------------------------------------------
// Everything an env kernel does -- RobotEnv.reset, env.step() for PushEnv and Grasp4DofEnv, the rollout loop
// of generate_episode with the on-device RandomPolicy (episode_generation.py:44-46, random_policy.py:14-23),
// n plain substeps, wait_until_stable, the partial step of rv_step_poll -- written as ONE resumable program:
// straight-line segments (the functions above) between REQUESTS to the substep loop.  The driver below runs a
// segment, and when the segment ends on a request, the loop -- from a single call site, so sim_run() is inlined
// once per kernel and no function call (with its register save / restore through scratch) is left on the path.
// Rollout: n_steps x { action = policy(obs); env.step(action) } for this env; budget != nullptr: asynchronous
// rollout -- the envs of the launch share a pool of env.step() calls; each env takes its next step while the pool
// lasts, so fast envs take more steps than slow ones and no SIMD idles (the reference's worker processes are just
// as independent, tools/parallel_run.py:54-90).  The k-th step an env takes uses macro index first_index + k.
enum { RV_PROG_RESET = 0, RV_PROG_MACRO = 1, RV_PROG_SUB = 2, RV_PROG_WAIT = 3, RV_PROG_ROLLOUT = 4, RV_PROG_PARTIAL = 5 };
struct ProgArgs {
  int gid;                                   // global env id (reset, rollout)
  int n_steps;                               // SUB: substeps; ROLLOUT: env.step() calls
  float lin_thr, ang_thr; int check_after, min_stable, max_steps;   // WAIT
  int first_index, auto_reset; RolloutRec rec; int env, n_envs; int* budget;   // ROLLOUT
  // ROLLOUT as one task of a queue (rv_env_kernel.h): this call takes the steps k0 .. k_stop - 1 of the env's n_steps
  // (k_stop = 0: all of them, from k0 = 0 -- a plain launch); the per-launch counters are zeroed by the task that starts at 0
  int k0, k_stop;
};
// returns (RV_PROG_PARTIAL) 1 when the env.step() completed in this launch
// The segments of the env program as functions of their own (RV_SEGMENTS_NOINLINE: the two-waves-per-SIMD build).  Round 4
// kept the substep LOOP out of line in that build and paid for it with the callee-saved registers of that one big function
// (~100 KB of scratch per call: 49 - 130 GB of L2 write-outs per launch, profiles/r04_final_c{3,4,5}_pmc.txt).  Inverted in
// round 5: the loop is inlined into the kernel (a kernel saves nothing) and what is called are the short straight-line
// segments between two runs of it -- a callee only saves the callee-saved registers it uses itself.  Like sim_run_fn did,
// a segment addresses the env through the file-scope LDS object and rebuilds Consts from scalars (through pointer
// arguments the LDS accesses would be compiled as flat loads).
#ifdef RV_SEGMENTS_NOINLINE
#define RV_SEG_K const Consts K = lds_consts(scene, 0); Shared& S = g_shared;
RV_DEV_NOINLINE void seg_step_prologue(const rv_scene* scene, int zero_counters, int count_step) { RV_SEG_K env_step_prologue(S, K, zero_counters, count_step); }
RV_DEV_NOINLINE void seg_step_epilogue(const rv_scene* scene) { RV_SEG_K env_step_epilogue(S, K); }
RV_DEV_NOINLINE void seg_pstep_begin(const rv_scene* scene) { RV_SEG_K env_pstep_begin(S, K); }
RV_DEV_NOINLINE int seg_pstep_end(const rv_scene* scene) { RV_SEG_K return env_pstep_end(S, K); }
RV_DEV_NOINLINE void seg_gstep_begin(const rv_scene* scene, int zero_counters) { RV_SEG_K genv_step_begin(S, K, zero_counters); }
RV_DEV_NOINLINE void seg_gstep_observe(const rv_scene* scene) { (void)scene; genv_step_observe(g_shared); }
RV_DEV_NOINLINE void seg_gstep_end(const rv_scene* scene) { (void)scene; genv_step_end(g_shared); }
RV_DEV_NOINLINE int seg_gpstep_begin(const rv_scene* scene) { RV_SEG_K return genv_pstep_begin(S, K); }
RV_DEV_NOINLINE int seg_gpstep_observe(const rv_scene* scene) { (void)scene; return genv_pstep_observe(g_shared); }
RV_DEV_NOINLINE int seg_gpstep_end(const rv_scene* scene) { (void)scene; return genv_pstep_end(g_shared); }
RV_DEV_NOINLINE void seg_reset_begin(const rv_scene* scene, int gid, int zero_counters) { RV_SEG_K env_reset_begin(S, K, gid, zero_counters); }
RV_DEV_NOINLINE void seg_reset_layout(const rv_scene* scene) { RV_SEG_K env_reset_layout(S, K); }
RV_DEV_NOINLINE void seg_reset_place(const rv_scene* scene, int i) { RV_SEG_K env_reset_place(S, K, i); }
RV_DEV_NOINLINE void seg_reset_after_body(const rv_scene* scene, int i) { RV_SEG_K env_reset_after_body(S, K, i); }
RV_DEV_NOINLINE void seg_reset_validate(const rv_scene* scene, int nb) { (void)scene; env_reset_validate(g_shared, nb); }
RV_DEV_NOINLINE void seg_reset_robot(const rv_scene* scene) { RV_SEG_K env_reset_robot(S, K); }
#undef RV_SEG_K
#define RV_SEG(inl_, out_) out_
#else
#define RV_SEG(inl_, out_) inl_
#endif
RV_DEV int env_program(Shared& S, const Consts& K, const int prog, const ProgArgs& A) {
  const rv_config* c = K.cfg;
  enum { PC_DONE = 0, PC_RESET_BEGIN, PC_RESET_LAYOUT, PC_RESET_BODY, PC_RESET_BODY_AFTER, PC_RESET_FINAL, PC_RESET_ROBOT,
         PC_STEP_BEGIN, PC_STEP_END, PC_GSTEP_OBS, PC_GSTEP_END, PC_PSTEP_END, PC_GPSTEP_OBS, PC_GPSTEP_END,
         PC_ROLL_TOP, PC_ROLL_ACTION, PC_ROLL_AFTER_STEP, PC_ROLL_TAIL };
  int pc = PC_DONE, ret_reset = PC_DONE, ret_step = PC_DONE;
  int reset_zero = 1, step_zero = 1;           // zero the per-launch counters (not inside a rollout)
  int ri = 0, rnb = 0;                         // reset: body index, body count
  int k = A.k0, k_end = A.k0, was_reset = 0;   // rollout
  const int k_hi = A.k_stop > 0 ? A.k_stop : A.n_steps;
  int fin = 0;
  const int grasp = RV_UNI(c->env_type == RV_ENV_GRASP);
  RunReq rq = run_req(0);
  int run = 0;
  if (prog == RV_PROG_RESET) pc = PC_RESET_BEGIN;
  else if (prog == RV_PROG_MACRO) pc = PC_STEP_BEGIN;
  else if (prog == RV_PROG_SUB) { if (A.n_steps > 0) { rq = run_req(A.n_steps); run = 1; } }
  else if (prog == RV_PROG_WAIT) { rq = run_req(0, 0u, A.lin_thr, A.ang_thr, A.check_after, A.min_stable, A.max_steps); run = 1; }
  else if (prog == RV_PROG_PARTIAL && grasp) {
    const int in_wait = RV_UNI(RV_SEG(genv_pstep_begin(S, K), seg_gpstep_begin(K.scene)));
    if (in_wait) { rq = run_req(0, 0u, 0.005f, 0.005f, 100, 100, 2000); run = 1; pc = PC_GPSTEP_END; }
    else { rq = run_req(-2); run = 1; pc = PC_GPSTEP_OBS; }
  }
  else if (prog == RV_PROG_PARTIAL) { RV_SEG(env_pstep_begin(S, K), seg_pstep_begin(K.scene)); rq = run_req(-1, 0u, 0.005f, 0.005f, 100, 100, 2000); run = 1; pc = PC_PSTEP_END; }
  else {
    if (A.k0 == 0) {
      RV_LANES_BEGIN
        if (lane == 0) launch_counters_zero(S.e);
      RV_LANES_END
    }
    reset_zero = 0; step_zero = 0;
    pc = PC_ROLL_TOP;
  }
  for (;;) {
    if (run) { sim_run(S, K, rq); run = 0; }        // THE call site of the substep loop
    if (pc == PC_DONE) break;
    switch (pc) {
      // ---- RobotEnv.reset
      case PC_RESET_BEGIN:
        RV_SEG(env_reset_begin(S, K, A.gid, reset_zero), seg_reset_begin(K.scene, A.gid, reset_zero));
        pc = PC_RESET_LAYOUT;
        break;
      case PC_RESET_LAYOUT:
        if (RV_UNI(S.s.valid)) { pc = PC_RESET_FINAL; break; }
        RV_SEG(env_reset_layout(S, K), seg_reset_layout(K.scene));
        rnb = RV_UNI(S.e.n_bodies); ri = 0;
        pc = PC_RESET_BODY;
        break;
      case PC_RESET_BODY:
        if (ri < rnb) {
          RV_SEG(env_reset_place(S, K, ri), seg_reset_place(K.scene, ri));
          RV_PROF(28)
          rq = run_req(0, 1u << ri, 0.1f, 0.1f, 100, 100, 500); run = 1;      // wait_until_stable(body)
          pc = PC_RESET_BODY_AFTER;
        } else {
          RV_SEG(env_reset_validate(S, rnb), seg_reset_validate(K.scene, rnb));
          pc = PC_RESET_LAYOUT;
        }
        break;
      case PC_RESET_BODY_AFTER:
        RV_SEG(env_reset_after_body(S, K, ri), seg_reset_after_body(K.scene, ri));
        ++ri;
        pc = PC_RESET_BODY;
        break;
      case PC_RESET_FINAL:
        RV_PROF(28)
        rq = run_req(0, 0u, 0.005f, 0.005f, 100, 100, 2000); run = 1;          // wait_until_stable(all)
        pc = PC_RESET_ROBOT;
        break;
      case PC_RESET_ROBOT:
        RV_SEG(env_reset_robot(S, K), seg_reset_robot(K.scene));
        pc = ret_reset;
        break;
      // ---- RobotEnv.step (robot_env.py:239-275)
      case PC_STEP_BEGIN:
        if (grasp) {
          RV_SEG(genv_step_begin(S, K, step_zero), seg_gstep_begin(K.scene, step_zero));
          rq = run_req(-2); run = 1;                                           // the phase loop of Grasp4DofEnv
          pc = PC_GSTEP_OBS;
        } else {
          RV_SEG(env_step_prologue(S, K, step_zero, 1), seg_step_prologue(K.scene, step_zero, 1));
          rq = run_req(-1, 0u, 0.005f, 0.005f, 100, 100, 2000); run = 1;       // phase loop + closing wait_until_stable
          pc = PC_STEP_END;
        }
        break;
      case PC_STEP_END:
        RV_PROF(30)
        RV_SEG(env_step_epilogue(S, K), seg_step_epilogue(K.scene));
        pc = ret_step;
        break;
      case PC_GSTEP_OBS:
        RV_SEG(genv_step_observe(S), seg_gstep_observe(K.scene));
        rq = run_req(0, 0u, 0.005f, 0.005f, 100, 100, 2000); run = 1;          // GraspReward: wait until the object is stable
        pc = PC_GSTEP_END;
        break;
      case PC_GSTEP_END:
        RV_SEG(genv_step_end(S), seg_gstep_end(K.scene));
        pc = ret_step;
        break;
      case PC_PSTEP_END:
        fin = RV_UNI(RV_SEG(env_pstep_end(S, K), seg_pstep_end(K.scene)));
        pc = PC_DONE;
        break;
      case PC_GPSTEP_OBS:
        if (RV_UNI(RV_SEG(genv_pstep_observe(S), seg_gpstep_observe(K.scene)))) {
          rq = run_req(0, 0u, 0.005f, 0.005f, 100, 100, 2000); run = 1;        // GraspReward: wait until the object is stable
          pc = PC_GPSTEP_END;
        } else pc = PC_DONE;
        break;
      case PC_GPSTEP_END:
        fin = RV_UNI(RV_SEG(genv_pstep_end(S), seg_gpstep_end(K.scene)));
        pc = PC_DONE;
        break;
      // ---- the rollout loop
      case PC_ROLL_TOP: {
        if (!(A.budget != nullptr || k < k_hi)) { pc = PC_ROLL_TAIL; break; }
        if (A.budget != nullptr) {
          RV_LANES_BEGIN
            if (lane == 0) {
#if defined(__HIP_DEVICE_COMPILE__)
              int left = atomicSub(A.budget, 1);
#else
              int left = (*A.budget)--;
#endif
              S.s.loop_break = !(left > 0);
            }
          RV_LANES_END
          if (RV_UNI(S.s.loop_break)) { pc = PC_ROLL_TAIL; break; }
        }
        was_reset = 0;
        if (RV_UNI(S.e.done)) {
          if (!A.auto_reset) { pc = PC_ROLL_TAIL; break; }
          was_reset = 1;
          ret_reset = PC_ROLL_ACTION;
          pc = PC_RESET_BEGIN;
          break;
        }
        pc = PC_ROLL_ACTION;
        break;
      }
      case PC_ROLL_ACTION:
        if (was_reset) { RV_PROF(28) }
        RV_LANES_BEGIN
          if (lane == 0) {
            random_action(c, A.gid, A.first_index + k, &S.e.action[0][0]);
            if (A.budget == nullptr) {
              const size_t row = (size_t)k * A.n_envs + A.env;
              if (A.rec.actions) {
                const int G = c->num_goal_steps > 0 ? c->num_goal_steps : 1;
                for (int x = 0; x < G * 4; ++x) A.rec.actions[row * (size_t)(G * 4) + x] = (&S.e.action[0][0])[x];
              }
              if (A.rec.resets) A.rec.resets[row] = (uint8_t)was_reset;
              if (was_reset) {      // what env.reset() returned (robot_env.py:204-237)
                if (A.rec.has_robs) obs_write_row(&S.e, A.rec.robs, row, c);
                if (A.rec.rsnaps) obs_snap_fill(S.e, K.arm, A.rec.rsnaps[row], 1);
              }
            }
          }
        RV_LANES_END
        RV_PROF(20)
        ret_step = PC_ROLL_AFTER_STEP;
        pc = PC_STEP_BEGIN;
        break;
      case PC_ROLL_AFTER_STEP:
        RV_LANES_BEGIN
          if (lane == 0 && A.budget == nullptr) rollout_record(A.rec, &S.e, (size_t)k * A.n_envs + A.env, c, K.arm);
        RV_LANES_END
        RV_PROF(23)
        k_end = k + 1;
        ++k;
        pc = PC_ROLL_TOP;
        break;
      case PC_ROLL_TAIL:
        // steps not taken (episode over, no auto-reset): reward 0, done
        if (A.budget == nullptr) {
          RV_LANES_BEGIN
            for (int kk = k_end + lane; kk < k_hi; kk += 64) rollout_record(A.rec, nullptr, (size_t)kk * A.n_envs + A.env, c, K.arm);
          RV_LANES_END
        }
        pc = PC_DONE;
        break;
      default:
        pc = PC_DONE;
        break;
    }
  }
  return fin;
}

// rebuild the per-launch caches that are not part of the persistent block
RV_DEV void env_enter(Shared& S, const Consts& K) {
  RV_LANES_BEGIN
    if (lane == 63) { S.s.jt_applied = 0; S.s.kin_fresh = 0; S.s.clr_valid = 0; S.s.bud_sub = 0; S.s.bud_clk = 0; S.s.suspended = 0; S.s.wus_resume = 0; S.s.far_valid = 0; S.s.far_n = 0; S.s.far = 0; }
    if (lane < RV_MAXB * RV_NCOL) S.s.sep[lane / RV_NCOL][lane % RV_NCOL] = 0.0f;
    if (lane < 8) table_prepare(S, K, lane);
    if (lane >= 48 && lane < 48 + RV_NLIMB + 1) { int i = lane - 48; S.s.jlen[i] = len(ld3(K.arm->jpos[i])); }
    if (lane >= 20 && lane < 20 + RV_NCOL) {
      const rv_arm* a = K.arm; int col = lane - 20; int f = a->col_frame[col];
      float ext = len(ld3(a->col_center[col])) + len(ld3(a->col_half[col]));
      if (f >= 8) ext += fabsr(a->finger_y0[f - 8]) + fmaxr(fabsr(a->q_lo[f - 1]), fabsr(a->q_hi[f - 1]));
      S.s.colext[col] = ext;
    }
    if (lane >= 8 && lane < 8 + RV_NFRAME) {
      int f = lane - 8;
      stm(S.s.frot[f], qmat(ldq(S.e.fquat[f])));
      float ext = 0.0f;
      for (int col = 0; col < RV_NCOL; ++col)
        if (K.arm->col_frame[col] == f) ext = fmaxr(ext, len(ld3(K.arm->col_center[col])) + len(ld3(K.arm->col_half[col])));
      S.s.fext[f] = ext;
    }
    if (lane >= 32 && lane < 32 + RV_MAXB) {
      int b = lane - 32;
      stm(S.s.rot[b], qmat(ldq(S.e.body[b] + 3)));
      if (S.e.active[b]) cache_shape_meta(S, K, b); else S.n_hulls[b] = 0;
    }
  RV_LANES_END
  RV_LANES_BEGIN
    if (lane < RV_NCOL) {
      // col_travelled() as weights: travel of box col = sum_j ccoef[col][j] * (path length of joint j)
      const rv_arm* a = K.arm; const int col = lane;
      const int f = a->col_frame[col]; const int fl = f < 7 ? f : 7;
      for (int j = 0; j < RV_NJ; ++j) S.s.ccoef[col][j] = 0.0f;
      float reach = S.s.colext[col];
      for (int j = fl; j >= 0; --j) {
        if (j < RV_NLIMB) S.s.ccoef[col][j] = reach;
        reach += S.s.jlen[j];
      }
      if (f >= 8) S.s.ccoef[col][f - 1] = 1.0f;
    }
  RV_LANES_END
}

}  // namespace rv

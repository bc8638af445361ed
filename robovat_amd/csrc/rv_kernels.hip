// rv_kernels.hip — __global__ kernels and the C ABI of librovat_hip.so
// (include/rovat.h).  gfx950 only; built with hipcc --offload-arch=gfx950.
//
// Kernel inventory:
//   k_env<MODE>     one wave64 per env, the whole reset / macro step / n
//                   substeps / settle loop out of LDS (rv_dev_env.h)
//   k_*             small one-thread-per-env accessors behind the getters,
//                   setters, observation, reward and policy entry points
#include <hip/hip_runtime.h>
#include <cstdio>
#include <stdint.h>
#include <string.h>
#include <stdlib.h>
#include <stddef.h>
#include <string>
#include <vector>
#include <mutex>
#include <new>

#include "../../include/rovat.h"
#include "rv_dev_env.h"
#include "rv_dev_obs.h"

using namespace rv;

#include "rv_env_kernel.h"

#define ENV_THREAD() const int i = (int)(blockIdx.x * blockDim.x + threadIdx.x); if (i >= n) return; DevEnv& e = envs[i];

__global__ void k_init(DevEnv* envs, int n, float mu_finger, float mu_table, const rv_config* cfg) {
  ENV_THREAD();
  uint32_t* p = reinterpret_cast<uint32_t*>(&e);
  for (int k = 0; k < (int)(sizeof(DevEnv) / 4); ++k) p[k] = 0u;
  for (int b = 0; b < RV_MAXB; ++b) e.body[b][6] = 1.0f;
  for (int f = 0; f < RV_NFRAME; ++f) e.fquat[f][3] = 1.0f;
  e.done = 1;  // RobotEnv.__init__: self._done = True (robot_env.py:66)
  e.mu_finger = mu_finger; e.mu_table = mu_table;
  camera_reset(e, cfg, 0, 0);      // (the unperturbed calibration until the first env.reset())
}
__global__ void k_get_body_state(const DevEnv* envs, int n, float* out) {
  const int i = (int)(blockIdx.x * blockDim.x + threadIdx.x); if (i >= n) return;
  for (int b = 0; b < RV_MAXB; ++b) for (int k = 0; k < 13; ++k) out[((size_t)i * RV_MAXB + b) * 13 + k] = envs[i].body[b][k];
}
__global__ void k_set_body_state(DevEnv* envs, int n, const float* in) {
  ENV_THREAD();
  for (int b = 0; b < RV_MAXB; ++b) for (int k = 0; k < 13; ++k) e.body[b][k] = in[((size_t)i * RV_MAXB + b) * 13 + k];
  for (int m = 0; m < RV_NMAN; ++m) e.man[m].n = 0;
  for (int b = 0; b < RV_MAXB; ++b) { e.asleep[b] = 0; e.sleep_count[b] = 0; e.deact_count[b] = 0; e.still_count[b] = 0; e.undisturbed[b] = 0; }
}
__global__ void k_get_body_params(const DevEnv* envs, int n, float* out) {
  const int i = (int)(blockIdx.x * blockDim.x + threadIdx.x); if (i >= n) return;
  const DevEnv& e = envs[i];
  for (int b = 0; b < RV_MAXB; ++b) {
    float* o = out + ((size_t)i * RV_MAXB + b) * 8;
    o[0] = (float)e.active[b]; o[1] = (float)e.shape[b]; o[2] = e.scale[b]; o[3] = e.mass[b]; o[4] = e.friction[b];
    o[5] = (float)e.frozen[b]; o[6] = e.table_z; o[7] = (float)e.asleep[b];
  }
}
__global__ void k_set_body_params(DevEnv* envs, int n, const float* in, const rv_config* cfg, const rv_scene* scene) {
  ENV_THREAD();
  Consts K; K.cfg = cfg; K.arm = &scene->arm; K.scene = scene; K.stop_after = 0;
  int nb = 0;
  for (int b = 0; b < RV_MAXB; ++b) {
    const float* o = in + ((size_t)i * RV_MAXB + b) * 8;
    e.active[b] = (int)o[0]; e.shape[b] = (int)o[1]; e.scale[b] = o[2]; e.friction[b] = o[4]; e.frozen[b] = (int)o[5]; e.asleep[b] = 0; e.sleep_count[b] = 0; e.deact_count[b] = 0; e.still_count[b] = 0; e.undisturbed[b] = 0;
    if (b == 0) e.table_z = o[6];
    if (e.active[b]) { body_set_mass(e, K, b, o[3]); nb++; }
  }
  e.n_bodies = nb;
}
__global__ void k_get_joint_state(const DevEnv* envs, int n, float* out) {
  const int i = (int)(blockIdx.x * blockDim.x + threadIdx.x); if (i >= n) return;
  for (int j = 0; j < RV_NJ; ++j) { out[((size_t)i * RV_NJ + j) * 2] = envs[i].q[j]; out[((size_t)i * RV_NJ + j) * 2 + 1] = envs[i].qd[j]; }
}
__device__ void update_link_frames(DevEnv& e, const rv_arm* a) {
  LimbFK F;
  fk_limb(a, e.q, F, nullptr);
  for (int f = 0; f <= RV_NLIMB; ++f) { st3(e.fpos[f], F.pos[f]); stq(e.fquat[f], F.quat[f]); }
  m3 r7 = qmat(F.quat[7]);
  v3 yax = mk(r7.m[1], r7.m[4], r7.m[7]);
  for (int k = 0; k < 2; ++k) {
    st3(e.fpos[8 + k], madd(F.pos[7], yax, a->finger_y0[k] + e.q[7 + k]));
    stq(e.fquat[8 + k], F.quat[7]);
  }
}
__global__ void k_set_joint_state(DevEnv* envs, int n, const float* in, const rv_config* cfg, const rv_scene* scene) {
  ENV_THREAD();
  const rv_arm* a = &scene->arm;
  for (int j = 0; j < RV_NJ; ++j) { e.q[j] = in[((size_t)i * RV_NJ + j) * 2]; e.qd[j] = in[((size_t)i * RV_NJ + j) * 2 + 1]; e.motor_q[j] = e.q[j]; }
  if (!e.arm_enabled) {
    for (int j = 0; j < RV_NJ; ++j) { e.motor_on[j] = 0; e.motor_kp[j] = cfg->kp; e.motor_kd[j] = cfg->kd; e.vmax_cmd[j] = a->v_max[j]; }
    arm_reset_targets(e); e.gripper_ready_time = 0.0f; e.arm_enabled = 1;
  }
  update_link_frames(e, a);
}
__global__ void k_get_link_poses(const DevEnv* envs, int n, float* out) {
  const int i = (int)(blockIdx.x * blockDim.x + threadIdx.x); if (i >= n) return;
  for (int f = 0; f < RV_NFRAME; ++f) {
    float* o = out + ((size_t)i * RV_NFRAME + f) * 7;
    for (int k = 0; k < 3; ++k) o[k] = envs[i].fpos[f][k];
    for (int k = 0; k < 4; ++k) o[3 + k] = envs[i].fquat[f][k];
  }
}
#ifdef RV_PROFILE
__global__ void k_debug_profile(const DevEnv* envs, int n, unsigned long long* out) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) for (int k = 0; k < 48; ++k) out[(size_t)i * 48 + k] = envs[i].prof[k];
}
#endif
__global__ void k_get_env_counters(const DevEnv* envs, int n, int32_t* out) {
  const int i = (int)(blockIdx.x * blockDim.x + threadIdx.x); if (i >= n) return;
  const DevEnv& e = envs[i]; int32_t* o = out + (size_t)i * RV_NCOUNTERS;
  o[8] = e.awake_last; o[9] = e.pairs_last;
  o[0] = e.sim_steps; o[1] = e.num_steps; o[2] = e.num_episodes; o[3] = e.phase; o[4] = e.done; o[5] = e.is_safe; o[6] = e.is_effective; o[7] = e.substeps_last;
}
__global__ void k_set_actions(DevEnv* envs, int n, const float* a, int G) {
  ENV_THREAD();
  for (int g = 0; g < G; ++g) for (int k = 0; k < 4; ++k) e.action[g][k] = a[((size_t)i * G + g) * 4 + k];
}
// rv_step_begin: the next action of the flagged envs; they are stepping from now on
__global__ void k_step_begin(DevEnv* envs, int n, const float* actions, const uint8_t* mask, int G) {
  ENV_THREAD();
  if (mask && !mask[i]) return;
  if (e.in_step == 1) return;                     // (still in the middle of its step: the action is ignored)
  for (int g2 = 0; g2 < G; ++g2) for (int k = 0; k < 4; ++k) e.action[g2][k] = actions[((size_t)i * G + g2) * 4 + k];
  e.in_step = e.done ? 2 : 1;
  e.step_stage = -1;
}
__global__ void k_reset_targets(DevEnv* envs, int n) {
  ENV_THREAD();
  arm_reset_targets(e);   // ControllableBody.reset_targets (controllable_body.py:347-350)
}
__global__ void k_set_int(int* p, int v) { *p = v; }
// BulletPhysics.position_control_array (bullet_physics.py:1061-1104): POSITION_CONTROL
// motor targets for the joints whose mask byte is set
__global__ void k_set_motor_targets(DevEnv* envs, int n, const float* q, const uint8_t* mask, const rv_config* cfg) {
  ENV_THREAD();
  for (int j = 0; j < RV_NJ; ++j) {
    if (mask && !mask[(size_t)i * RV_NJ + j]) continue;
    e.motor_on[j] = 1; e.motor_q[j] = q[(size_t)i * RV_NJ + j]; e.motor_kp[j] = cfg->kp; e.motor_kd[j] = cfg->kd;
  }
}
struct ConArgs { int body, child, type; float lp[3], lq[4], tp[3], tq[4], fmax; };
__global__ void k_set_constraint(DevEnv* envs, int n, ConArgs a) {
  ENV_THREAD();
  const int b = a.body;
  // (a body that went to sleep hanging from its constraint must notice that it changed or is gone)
  if (a.fmax < 0.0f) e.con_on[b] = 0;
  else {
    e.con_on[b] = a.type | ((a.child + 1) << 4); e.con_fmax[b] = a.fmax;      // (RV_CON_TYPE / RV_CON_CHILD, rv_dev_env.h)
    for (int k = 0; k < 3; ++k) { e.con_lpos[b][k] = a.lp[k]; e.con_tpos[b][k] = a.tp[k]; }
    for (int k = 0; k < 4; ++k) { e.con_lquat[b][k] = a.lq[k]; e.con_tquat[b][k] = a.tq[k]; }
  }
  e.asleep[b] = 0; e.sleep_count[b] = 0; e.deact_count[b] = 0; e.still_count[b] = 0; e.undisturbed[b] = 0;
  if (a.child >= 0 && a.child < RV_MAXB) { const int cb = a.child; e.asleep[cb] = 0; e.sleep_count[cb] = 0; e.deact_count[cb] = 0; e.still_count[cb] = 0; e.undisturbed[cb] = 0; }
}
__global__ void k_set_friction(DevEnv* envs, int n, float mu_finger, float mu_table) {
  ENV_THREAD();
  if (mu_finger >= 0.0f) e.mu_finger = mu_finger;
  if (mu_table >= 0.0f) e.mu_table = mu_table;
}
__global__ void k_grip(DevEnv* envs, int n, float value, const rv_config* cfg, const rv_scene* scene) {
  ENV_THREAD();
  grip_env(e, &scene->arm, cfg, value);
}
__global__ void k_set_joint_targets(DevEnv* envs, int n, const float* q, const rv_config* cfg, const rv_scene* scene, float timeout, float threshold) {
  ENV_THREAD();
  // SawyerSim.move_to_joint_positions (sawyer_sim.py:186-234)
  arm_reset_targets(e);
  for (int j = 0; j < RV_NLIMB; ++j) e.vmax_cmd[j] = cfg->limb_max_velocity_ratio * scene->arm.v_max[j];
  JTarget& t = e.jt;
  t.active = 1; t.n_idx = RV_NLIMB; t.has_vel = 1; t.from_ik = 0;
  for (int j = 0; j < RV_NLIMB; ++j) { t.idx[j] = j; t.pos[j] = q[(size_t)i * RV_NLIMB + j]; }
  t.start_t = cfg->dt * (float)e.sim_steps; t.stop_t = t.start_t + (timeout > 0.0f ? timeout : cfg->limb_timeout); t.has_stop = 1;
  t.pos_thr = threshold > 0.0f ? threshold : cfg->limb_position_threshold; t.vel_thr = cfg->velocity_threshold;
}
__global__ void k_set_link_target(DevEnv* envs, int n, const float* pose, const rv_config* cfg, const rv_scene* scene, float timeout, float threshold) {
  ENV_THREAD();
  // SawyerSim.move_to_gripper_pose (sawyer_sim.py:236-308)
  arm_reset_targets(e);
  for (int j = 0; j < RV_NLIMB; ++j) e.vmax_cmd[j] = cfg->limb_max_velocity_ratio * scene->arm.v_max[j];
  LTarget& t = e.lt;
  t.active = 1; t.has_pose = 1; t.nq = 0;
  for (int k = 0; k < 7; ++k) t.pose[k] = pose[(size_t)i * 7 + k];
  t.start_t = cfg->dt * (float)e.sim_steps; t.stop_t = t.start_t + (timeout > 0.0f ? timeout : cfg->limb_timeout); t.has_stop = 1;
  t.pos_thr = threshold > 0.0f ? threshold : cfg->limb_position_threshold; t.vel_thr = cfg->velocity_threshold;
}
__global__ void k_set_link_path(DevEnv* envs, int n, const float* poses, int n_poses, const rv_config* cfg, const rv_scene* scene, float timeout, float threshold) {
  ENV_THREAD();
  // SawyerSim.move_along_gripper_path (sawyer_sim.py:310-360) -> ControllableBody.set_target_link_poses
  // (controllable_body.py:322-345): the poses are queued, the first one becomes the link target
  arm_reset_targets(e);
  for (int j = 0; j < RV_NLIMB; ++j) e.vmax_cmd[j] = cfg->limb_max_velocity_ratio * scene->arm.v_max[j];
  LTarget& t = e.lt;
  t.active = 1; t.has_pose = 0; t.nq = n_poses;
  for (int q = 0; q < n_poses; ++q) for (int k = 0; k < 7; ++k) t.queue[q][k] = poses[((size_t)i * n_poses + q) * 7 + k];
  t.start_t = cfg->dt * (float)e.sim_steps; t.stop_t = t.start_t + (timeout > 0.0f ? timeout : cfg->limb_timeout); t.has_stop = 1;
  t.pos_thr = threshold > 0.0f ? threshold : cfg->limb_position_threshold; t.vel_thr = cfg->velocity_threshold;
  lt_pop(t);
}
__global__ void k_set_max_joint_velocities(DevEnv* envs, int n, const float* v) {
  ENV_THREAD();
  // ControllableBody.set_max_joint_velocities (controllable_body.py:357-372): the speed limits of the limb joints that the
  // running (and the next) targets are followed with -- SawyerSim.move_to_*(speed=...) sends them with every motion command
  // (sawyer_sim.py:212-220, 285-293, 336-344); a target setter puts LIMB_MAX_VELOCITY_RATIO x the URDF limits back
  for (int j = 0; j < RV_NLIMB; ++j) e.vmax_cmd[j] = v[(size_t)i * RV_NLIMB + j];
}
__global__ void k_robot_ready(DevEnv* envs, int n, uint8_t* out, const rv_config* cfg) {
  ENV_THREAD();
  // SawyerSim.is_limb_ready -> ControllableBody.is_ready(limb joints) (sawyer_sim.py:394-400, controllable_body.py:565-595):
  // like the reference's, the query retires targets that are done (reached, timed out, path exhausted)
  const float now = cfg->dt * (float)e.sim_steps;
  {
    const LTarget& t = e.lt;
    if (!t.has_stop || now >= t.stop_t || (!t.has_pose && t.nq == 0)) lt_reset(e.lt);
  }
  {
    const JTarget& t = e.jt;
    if (!t.has_stop || now >= t.stop_t || check_joints_reached(e)) jt_reset(e.jt);
  }
  int ready = 1;
  if (e.lt.active) ready = 0;                 // (every limb joint index < end-effector link index)
  if (e.jt.active) for (int k = 0; k < e.jt.n_idx; ++k) if (e.jt.idx[k] < RV_NLIMB) ready = 0;
  out[(size_t)i * 2] = (uint8_t)ready;
  // SawyerSim.is_gripper_ready (sawyer_sim.py:402-408): 0.5 s of simulated time after the last grip command
  out[(size_t)i * 2 + 1] = (uint8_t)(now >= e.gripper_ready_time);
}
__global__ void k_compute_ik(const DevEnv* envs, int n, const float* pose, float* q, const rv_config* cfg, const rv_scene* scene) {
  const int i = (int)(blockIdx.x * blockDim.x + threadIdx.x); if (i >= n) return;
  Consts K; K.cfg = cfg; K.arm = &scene->arm; K.scene = scene; K.stop_after = 0;
  float p[7], q0[RV_NLIMB], out[RV_NLIMB];
  for (int k = 0; k < 7; ++k) p[k] = pose[(size_t)i * 7 + k];
  for (int j = 0; j < RV_NLIMB; ++j) q0[j] = envs[i].q[j];
  arm_ik(K, q0, p, out);
  for (int j = 0; j < RV_NLIMB; ++j) q[(size_t)i * RV_NLIMB + j] = out[j];
}
__global__ void k_query_contacts(const DevEnv* envs, int n, uint8_t* out) {
  const int i = (int)(blockIdx.x * blockDim.x + threadIdx.x); if (i >= n) return;
  const DevEnv& e = envs[i]; uint8_t* o = out + (size_t)i * (2 + RV_MAXB);
  o[0] = (uint8_t)e.flag_arm_table; o[1] = (uint8_t)arm_touches_movables(e);
  for (int b = 0; b < RV_MAXB; ++b) o[2 + b] = (uint8_t)(e.active[b] && e.flag_arm_body[b]);      // (a static body: never)
}
__global__ void k_get_camera(const DevEnv* envs, int n, float* out) {
  const int i = (int)(blockIdx.x * blockDim.x + threadIdx.x); if (i >= n) return;
  const DevEnv& e = envs[i]; float* o = out + (size_t)i * 17;
  for (int k = 0; k < 5; ++k) o[k] = e.cam_intrinsics[k];
  for (int k = 0; k < 9; ++k) o[5 + k] = e.cam_rotation[k];
  for (int k = 0; k < 3; ++k) o[14 + k] = e.cam_translation[k];
}
__global__ void k_manifold_counts(const DevEnv* envs, int n, int32_t* out) {
  const int i = (int)(blockIdx.x * blockDim.x + threadIdx.x); if (i >= n) return;
  for (int m = 0; m < RV_NMAN; ++m) out[(size_t)i * RV_NMAN + m] = envs[i].man[m].n;
}
__global__ void k_observe(const DevEnv* envs, int n, rv_obs_buffers o, const rv_config* cfg) {
  const int i = (int)(blockIdx.x * blockDim.x + threadIdx.x); if (i >= n) return;
  obs_write_row(&envs[i], o, (size_t)i, cfg);
}
__global__ void k_obs_snap(const DevEnv* envs, int n, ObsSnap* snaps, const rv_scene* scene) {
  const int i = (int)(blockIdx.x * blockDim.x + threadIdx.x); if (i >= n) return;
  obs_snap_fill(envs[i], &scene->arm, snaps[i]);
}
RV_DEV float wave_min(float x) { for (int o = 32; o > 0; o >>= 1) { float y = __shfl_xor(x, o); x = y < x ? y : x; } return x; }
RV_DEV float wave_max(float x) { for (int o = 32; o > 0; o >>= 1) { float y = __shfl_xor(x, o); x = y > x ? y : x; } return x; }
// SegmentedPointCloudObs (camera_obs.py:182-238): one wave per (observation, body); see rv_dev_obs.h
__global__ __launch_bounds__(64) void k_point_cloud(const ObsSnap* snaps, int n_snaps, int n_envs, float* out,
                                                    const rv_config* c, const rv_scene* scene) {
  __shared__ uint32_t s_pix[RV_PC_MAXPIX];
  __shared__ float s_dep[RV_PC_MAXPIX];
  __shared__ uint32_t s_key[RV_PC_MAXPIX];
  __shared__ float s_rot[RV_MAXB][9];
  const int lane = (int)threadIdx.x;
  const int si = (int)blockIdx.x / RV_MAXB, b = (int)blockIdx.x % RV_MAXB;
  if (si >= n_snaps) return;
  const ObsSnap& s = snaps[si];
  const int P = c->num_points;
  float* o = out + ((size_t)si * RV_MAXB + b) * (size_t)P * 3;
  const uint32_t gid = (uint32_t)(c->env_id_offset + si % n_envs);
  if (lane < RV_MAXB) { m3 m = qmat(ldq(s.pose[lane] + 3)); stm(s_rot[lane], m); }
  __syncthreads();
  int n = 0;
  const v3 cam_o = cam_position(&s);
  if (s.shape[b] >= 0 && !s.is_static[b]) {
    // screen rectangle of the body
    const rv_shape* sh = &scene->shapes[s.shape[b]];
    float mu = 1e30f, xu = -1e30f, mv = 1e30f, xv = -1e30f, mz = 1e30f;
    {
      const int h = lane >> 4, i = lane & 15;
      if (h < sh->n_hulls && i < sh->n_verts[h]) {
        const float sc = s.scale[b];
        v3 pw = add(ld3(s.pose[b]), mulv(s_rot[b], mk(sh->verts[h][i][0] * sc, sh->verts[h][i][1] * sc, sh->verts[h][i][2] * sc)));
        float u, v, z;
        project_vertex(&s, pw, &u, &v, &z);
        mz = z;
        if (z > c->cam_near) { mu = xu = u; mv = xv = v; }
      }
    }
    mu = wave_min(mu); xu = wave_max(xu); mv = wave_min(mv); xv = wave_max(xv); mz = wave_min(mz);
    if (mz > c->cam_near) {
      const float W1 = (float)(c->cam_width - 1), H1 = (float)(c->cam_height - 1);
      const int u0 = (int)fclampr(ffloorr(mu) - 1.0f, 0.0f, W1), u1 = (int)fclampr(ffloorr(xu) + 2.0f, 0.0f, W1);
      const int v0 = (int)fclampr(ffloorr(mv) - 1.0f, 0.0f, H1), v1 = (int)fclampr(ffloorr(xv) + 2.0f, 0.0f, H1);
      const int w = u1 - u0 + 1, hh = v1 - v0 + 1;
      const int total = (xu < 0.0f || xv < 0.0f || mu > W1 || mv > H1) ? 0 : w * hh;
      // which link boxes of the arm can cover a pixel of this rectangle at all?  (bounding sphere of the box
      // projected as a disc, generously; the arm is off-stage when the envs observe, so the answer is
      // almost always "none" and the pixels pay nothing for it)
      unsigned arm_mask = 0u;
      if (s.arm_on == 1) {
        bool may = false;
        if (lane < RV_NCOL) {
          const rv_arm* arm = &scene->arm;
          const float hx = arm->col_half[lane][0] + c->margin, hy = arm->col_half[lane][1] + c->margin, hz = arm->col_half[lane][2] + c->margin;
          const float r = fsqrtr(hx * hx + hy * hy + hz * hz) * 1.01f + 1e-4f;
          float uc, vc, zc;
          project_vertex(&s, ld3(s.arm_c[lane]), &uc, &vc, &zc);
          if (zc - r <= c->cam_near) may = zc + r > 0.0f;          // too close to bound its disc: keep it
          else {
            const float f = fmaxr(fabsr(s.cam_intrinsics[0]), fabsr(s.cam_intrinsics[1])) + fabsr(s.cam_intrinsics[4]);     // (this env's calibration: camera noise)
            const float rp = r * f / (zc - r) * 1.05f + 2.0f;
            may = !(uc + rp < (float)u0 || uc - rp > (float)u1 || vc + rp < (float)v0 || vc - rp > (float)v1);
          }
        }
        arm_mask = (unsigned)(__ballot(may) & 0x3ffull);
      }
      // pass 0 keeps every visible pixel; a body with more than RV_PC_MAXPIX of them is cast again and every
      // stride-th visible pixel (scan order) is kept, so that the sample covers the whole body
      int stride = 1;
      for (int pass = 0; pass < 2; ++pass) {
        n = 0;
        for (int base = 0; base < total; base += 64) {
          const int idx = base + lane;
          bool vis = false; float dep = 0.0f; int u = 0, v = 0;
          if (idx < total) {
            u = u0 + idx % w; v = v0 + idx / w;
            v3 dw = cam_to_world_dir(&s, pixel_dir_cam(&s, (float)u, (float)v));
            int who = render_pixel(c, scene, s, s_rot, cam_o, dw, &dep, nullptr, arm_mask);
            vis = (who == b) && dep > c->cam_near && crop_ok(c, deproject(&s, cam_o, (float)u, (float)v, dep));
          }
          const unsigned long long bal = __ballot(vis);
          const int rank = n + (int)__popcll(bal & ((1ull << lane) - 1ull));
          const int pos = rank / stride;
          if (vis && rank % stride == 0 && pos < RV_PC_MAXPIX) { s_pix[pos] = ((uint32_t)v << 16) | (uint32_t)u; s_dep[pos] = dep; }
          n += (int)__popcll(bal);
        }
        if (n <= RV_PC_MAXPIX) break;
        if (pass == 0) stride = (n + RV_PC_MAXPIX - 1) / RV_PC_MAXPIX;
      }
      n = (n + stride - 1) / stride;
    }
  }
  __syncthreads();
  if (n == 0) {                                  // group_by_labels: zeros when the body has no pixel
    for (int j = lane; j < P * 3; j += 64) o[j] = 0.0f;
    return;
  }
  if (n < P) {                                   // downsample: with replacement iff fewer than num_points
    for (int j = lane; j < P; j += 64) {
      const uint32_t i = pc_hash(c, gid, s.rng_arg, RV_PC_DRAW_CTR(b, j)) % (uint32_t)n;
      const uint32_t px = s_pix[i];
      st3(o + 3 * j, deproject(&s, cam_o, (float)(px & 0xffffu), (float)(px >> 16), s_dep[i]));
    }
    return;
  }
  // uniformly random P-subset: the P smallest keys.  Radix select of the P-th smallest key T
  for (int i = lane; i < n; i += 64) s_key[i] = pc_hash(c, gid, s.rng_arg, RV_PC_KEY_CTR(b, i));
  __syncthreads();
  uint32_t prefix = 0u, mask = 0u; int k = P;
  for (int bit = 31; bit >= 0; --bit) {
    const uint32_t m2 = mask | (1u << bit);
    int cnt = 0;
    for (int base = 0; base < n; base += 64) {
      const int i = base + lane;
      const bool z = i < n && (s_key[i] & m2) == prefix;     // high bits match and this bit is 0
      cnt += (int)__popcll(__ballot(z));
    }
    if (k > cnt) { k -= cnt; prefix |= (1u << bit); }
    mask = m2;
  }
  const uint32_t T = prefix;       // k = how many keys equal to T are still needed (scan order)
  // the selected pixels are compacted IN PLACE to the front of the three arrays (a lane writes at or below its own
  // index, after every lane of the chunk has read its element) ...
  int outn = 0, ties = 0;
  for (int base = 0; base < n; base += 64) {
    const int i = base + lane;
    const uint32_t key = i < n ? s_key[i] : 0xffffffffu;
    const uint32_t px = i < n ? s_pix[i] : 0u;
    const float dp = i < n ? s_dep[i] : 0.0f;
    const bool lt = i < n && key < T;
    const bool eq = i < n && key == T;
    const unsigned long long beq = __ballot(eq);
    const int tie_rank = ties + (int)__popcll(beq & ((1ull << lane) - 1ull));
    const bool sel = lt || (eq && tie_rank < k);
    const unsigned long long bs = __ballot(sel);
    const int pos = outn + (int)__popcll(bs & ((1ull << lane) - 1ull));
    if (sel && pos < P) { s_key[pos] = key; s_pix[pos] = px; s_dep[pos] = dp; }
    outn += (int)__popcll(bs); ties += (int)__popcll(beq);
  }
  __syncthreads();
  // ... and go out in the order of their keys (ties in scan order): np.random.choice(replace=False) returns the
  // subset in random order (point_cloud_utils.py:23-39).  Rank of an element = how many selected ones precede it;
  // a lane ranks up to four elements per pass over the keys
  for (int j0 = 0; j0 < P; j0 += 256) {
    uint32_t kq[4]; int rq[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) { const int jj = j0 + lane + 64 * q; kq[q] = jj < P ? s_key[jj] : 0u; rq[q] = 0; }
    for (int t = 0; t < P; ++t) {
      const uint32_t kt = s_key[t];
#pragma unroll
      for (int q = 0; q < 4; ++q) rq[q] += (kt < kq[q] || (kt == kq[q] && t < j0 + lane + 64 * q)) ? 1 : 0;
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int jj = j0 + lane + 64 * q;
      if (jj < P) {
        const uint32_t px = s_pix[jj];
        st3(o + 3 * rq[q], deproject(&s, cam_o, (float)(px & 0xffffu), (float)(px >> 16), s_dep[jj]));
      }
    }
  }
}
// CameraObs 'depth' / 'segmask' (camera_obs.py:33-88 over BulletCamera._frames,
// bullet_camera.py:188-235): the same ray cast as the point cloud, one thread per pixel.
// depth: eye z, 0 where nothing is hit; segmask: body index, RV_MAXB = table, RV_MAXB + 1 = arm, 255 = nothing
__global__ void k_render(const ObsSnap* snaps, int n, float* depth, uint8_t* seg, const rv_config* c, const rv_scene* scene) {
  const int H = c->cam_height, W = c->cam_width;
  const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= (size_t)n * H * W) return;
  const int i = (int)(t / ((size_t)H * W)); const int px = (int)(t % ((size_t)H * W));
  const int v = px / W, u = px - v * W;
  const ObsSnap& s = snaps[i];
  float rot[RV_MAXB][9];
  for (int b = 0; b < RV_MAXB; ++b) { m3 m = qmat(ldq(s.pose[b] + 3)); stm(rot[b], m); }
  const v3 cam_o = cam_position(&s);
  float d;
  int who = render_pixel(c, scene, s, rot, cam_o, cam_to_world_dir(&s, pixel_dir_cam(&s, (float)u, (float)v)), &d);
  if (who >= 0 && !(d > c->cam_near)) who = -1;
  if (depth) depth[t] = who >= 0 ? d : 0.0f;
  if (seg) seg[t] = who >= 0 ? (uint8_t)who : (uint8_t)255;
}
__global__ void k_reward(const DevEnv* envs, int n, float* reward, uint8_t* done) {
  const int i = (int)(blockIdx.x * blockDim.x + threadIdx.x); if (i >= n) return;
  // an env whose episode is over is not stepped by rv_step_macro (the reference raises
  // "Forget to reset?", robot_env.py:244-245): it reports reward 0, done
  if (reward) reward[i] = envs[i].reward_valid ? envs[i].last_reward : 0.0f;
  if (done) done[i] = (uint8_t)envs[i].done;
}
__global__ void k_returns(const DevEnv* envs, int n, float* r) {
  const int i = (int)(blockIdx.x * blockDim.x + threadIdx.x); if (i >= n) return;
  r[i] = envs[i].episode_reward;
}
__global__ void k_policy_random(int n, const rv_config* cfg, int macro_index, float* actions) {
  const int i = (int)(blockIdx.x * blockDim.x + threadIdx.x); if (i >= n) return;
  int G = cfg->num_goal_steps > 0 ? cfg->num_goal_steps : 1;
  random_action(cfg, cfg->env_id_offset + i, macro_index, actions + (size_t)i * G * 4);
}
// HeuristicPushSampler._sample (heuristic_push_sampler.py:66-123): one wave
// per env, 64 candidate pushes per round; the lowest successful attempt wins.
__global__ __launch_bounds__(64) void k_policy_heuristic(const DevEnv* envs, int n, const rv_config* c, int max_attempts, float* actions) {
  const int i = (int)blockIdx.x; if (i >= n) return;
  const int lane = (int)threadIdx.x;
  const DevEnv& e = envs[i];
  int G = c->num_goal_steps > 0 ? c->num_goal_steps : 1;
  int nb = 0;
  for (int b = 0; b < RV_MAXB; ++b) nb += body_movable(e, b);
  if (nb == 0) nb = 1;
  // the policy reads the counters from the observation (push_policy.py:46-49), i.e. the
  // env.attributes snapshot; like the reference it assumes the first nb slots are the bodies
  // (heuristic_push_sampler.py:70: position[:num_bodies])
  const int n_ep = e.obs_num_episodes, n_st = e.obs_num_steps;
  int body_id = n_ep % nb;
  float base = (float)n_ep * 42.0f;
  base = base - 2.0f * RV_PI * ffloorr(base / (2.0f * RV_PI));
  float lo0 = c->cspace_low[0], hi0 = c->cspace_high[0], lo1 = c->cspace_low[1], hi1 = c->cspace_high[1];
  float s0 = 0.0f, s1 = 0.0f, m0 = 0.0f, m1 = 0.0f;
  int found_att = -1;
  for (int base_att = 0; base_att < max_attempts; base_att += 64) {
    int att = base_att + lane;
    bool good = false;
    if (att < max_attempts) {
      Rng g = rng_init(c->seed_lo, c->seed_hi, (uint32_t)(c->env_id_offset + i), RV_STREAM_HEUR, (uint32_t)(n_ep * 64 + n_st));
      g.c0 = (uint32_t)att * 2u;
      s0 = rng_uniform(g, -1.0f, 1.0f); s1 = rng_uniform(g, -1.0f, 1.0f);
      float ang = base + rng_uniform(g, -0.25f * RV_PI, 0.25f * RV_PI);
      float sn, co; sincosr(ang, &sn, &co);
      m0 = fclampr(co + rng_uniform(g, -0.3f, 0.3f), -1.0f, 1.0f);
      m1 = fclampr(sn + rng_uniform(g, -0.3f, 0.3f), -1.0f, 1.0f);
      float x = s0 * (0.5f * (hi0 - lo0)) + 0.5f * (hi0 + lo0);
      float y = s1 * (0.5f * (hi1 - lo1)) + 0.5f * (hi1 + lo1);
      float ex = fclampr(x + m0 * c->translation_x, lo0, hi0);
      float ey = fclampr(y + m1 * c->translation_y, lo1, hi1);
      int safe = 1;
      for (int b = 0; b < nb; ++b) {
        float dx = e.obs_pos[b][0] - x, dy = e.obs_pos[b][1] - y;
        if (!(fsqrtr(dx * dx + dy * dy) > 0.05f)) safe = 0;
      }
      float dx1 = e.obs_pos[body_id][0] - x, dy1 = e.obs_pos[body_id][1] - y;
      float dx2 = e.obs_pos[body_id][0] - ex, dy2 = e.obs_pos[body_id][1] - ey;
      int clear = fsqrtr(dx1 * dx1 + dy1 * dy1) >= 0.01f && fsqrtr(dx2 * dx2 + dy2 * dy2) >= 0.01f;
      good = safe && !clear;
    }
    unsigned long long ballot = __ballot(good);
    if (ballot) { found_att = base_att + (int)__ffsll((long long)ballot) - 1; break; }
  }
  // winner (or, like the reference, the last attempt drawn) writes the action
  int writer = found_att >= 0 ? (found_att & 63) : ((max_attempts - 1) & 63);
  if (lane == writer) {
    for (int g2 = 0; g2 < G; ++g2) {
      float* a = actions + ((size_t)i * G + g2) * 4;
      a[0] = s0; a[1] = s1; a[2] = m0; a[3] = m1;
    }
  }
}
__global__ void k_stats(const DevEnv* envs, int n, rv_macro_stats* st, float success_thresh) {
  const int i = (int)(blockIdx.x * blockDim.x + threadIdx.x); if (i >= n) return;
  const DevEnv& e = envs[i];
  typedef unsigned long long u64;
  if (e.substeps_last > 0) {
    atomicAdd((u64*)&st->substeps, (u64)e.substeps_last);
    atomicMax((u64*)&st->max_substeps, (u64)e.substeps_last);
    atomicAdd((u64*)&st->awake_substeps, (u64)e.awake_last);
  }
  if (e.stepped) {
    atomicAdd((u64*)&st->env_steps, (u64)e.stepped);
    // per-launch sums kept by env_step (a rollout launch takes several steps per env)
    if (e.l_unsafe) atomicAdd((u64*)&st->unsafe, (u64)e.l_unsafe);
    if (e.l_ineffective) atomicAdd((u64*)&st->ineffective, (u64)e.l_ineffective);
    if (e.l_useful) atomicAdd((u64*)&st->useful, (u64)e.l_useful);
    if (e.l_episodes) atomicAdd((u64*)&st->episodes_done, (u64)e.l_episodes);
    if (e.l_successes) atomicAdd((u64*)&st->successes, (u64)e.l_successes);
  }
}

__global__ void k_render_rgb(const ObsSnap* snaps, int n, uint8_t* rgb, const rv_config* c, const rv_scene* scene) {
  const int H = c->cam_height, W = c->cam_width;
  const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= (size_t)n * H * W) return;
  const int i = (int)(t / ((size_t)H * W)); const int px = (int)(t % ((size_t)H * W));
  const int v = px / W, u = px - v * W;
  const ObsSnap& s = snaps[i];
  float rot[RV_MAXB][9];
  for (int b = 0; b < RV_MAXB; ++b) { m3 m = qmat(ldq(s.pose[b] + 3)); stm(rot[b], m); }
  const v3 cam_o = cam_position(&s);
  float d; v3 nrm;
  int who = render_pixel(c, scene, s, rot, cam_o, cam_to_world_dir(&s, pixel_dir_cam(&s, (float)u, (float)v)), &d, &nrm);
  if (who >= 0 && !(d > c->cam_near)) who = -1;
  shade_rgb(who, nrm, rgb + t * 3);
}

// ------------------------------------------------------------------ host ABI
struct rv_world {
  rv_config cfg;
  int device;
  int n;
  hipStream_t stream;
  rv_config* d_cfg;
  rv_scene* d_scene;
  DevEnv* d_envs;
  rv_macro_stats* d_stats;
  int* d_budget;
  ObsSnap* d_snaps; size_t n_snaps_cap;   // pose snapshots for the point-cloud render
  hipEvent_t ev0, ev1;
  bool timed;
  int auto_reset;                         // rv_set_auto_reset
  int occ2;                               // more envs than SIMDs: launch k_env_occ2 (rv_env_kernel.h)
  // rollouts of a world with more envs than the GPU has wave slots go through task queues, one per XCD (rv_env_kernel.h):
  // q_grid workgroups (what is resident at a time), d_q = [RV_Q_CTL_WORDS control words][RV_Q_NQ rings of q_ring slots];
  // q_launch numbers the queued launches, q_used: a queued launch ran since rv_get_stats last looked at its error word
  int q_grid; int* d_q; size_t q_cap; int q_launch; bool q_used;
};

static thread_local std::string g_err;
static int fail(int code, const std::string& msg) { g_err = msg; return code; }
#define HIPCHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) return fail(RV_ERR_HIP, std::string(#x) + ": " + hipGetErrorString(e_)); } while (0)
#define WCHK(w) do { if (!(w)) return fail(RV_ERR_VALUE, "null world"); HIPCHK(hipSetDevice((w)->device)); } while (0)

static inline dim3 grid1(int n) { return dim3((unsigned)((n + 127) / 128)); }
#define TPB 128

// the register-rich kernel while every env has a SIMD of its own, the two-waves-per-SIMD build beyond that
static void launch_k_env(rv_world* w, int mode, const EnvKernelArgs& a) {
  const int n_grid = a.q_slots ? w->q_grid : w->n;
  if (w->occ2) rv_launch_k_env_occ2(mode, a, n_grid, w->stream);
  else rv_launch_k_env_here(mode, a, n_grid, w->stream);
}
#define RV_QUEUE_MIN_STEPS 10
__global__ void k_queue_init(int* q, long long words) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < words) q[i] = i < RV_Q_CTL_WORDS ? 0 : -1;      // counters 0 (the error word too), every slot "not yet published"
}
// MODE_ROLLOUT of n_steps through the task queues?  Worlds with more envs than resident workgroups (RV_QUEUE=0: never)
// pool > 0: the work-conserving rollout (rv_rollout_async) -- `pool` tasks in all, an env goes back to the tail after every
// step, so the envs take turns and one in a slow state simply gets fewer of them
static int queue_setup(rv_world* w, int n_steps, EnvKernelArgs& a, long long pool = 0) {
  a.q_slots = nullptr; a.q_ctl = nullptr; a.q_cap = 0; a.q_total = 0; a.q_pool = 0; a.q_launch = 0; a.q_wt = 0; a.q_sticky = 0; a.q_debug = 0; a.q_global = 0; a.q_steal = 0;
  const char* q = getenv("RV_QUEUE");      // (read per launch: the tests compare the two schedules in one process)
  // (measured with the per-XCD queues and the keep rule of k_env, profiles/r06_queue_variants.txt -- 8192 envs: 20 steps + 14 %, 10 steps
  // + 13 ... 16 %, 5 steps + 16 ... 19 %, 2 steps - 5 %; 4096 concave envs x 10 steps, bound by their slowest env: + - 0;
  // 8192 envs without deactivation, 8 steps, bound by envs that get slower step after step: - 6 %.  Rollouts of 10 steps and
  // more go through the queues.  RV_QUEUE=1 forces them, 0 forbids them, RV_QUEUE_MIN_STEPS moves the threshold)
  const char* qm = getenv("RV_QUEUE_MIN_STEPS");
  const int min_steps = qm ? atoi(qm) : RV_QUEUE_MIN_STEPS;
  if ((q && atoi(q) == 0) || w->q_grid <= 0 || w->n <= w->q_grid || (pool == 0 && n_steps < min_steps && !(q && atoi(q) == 1))) return RV_OK;
  const size_t total = pool > 0 ? (size_t)pool : (size_t)w->n * (size_t)n_steps;      // tasks that are begun
  if (total > ((size_t)1 << 24)) return RV_OK;      // (8 rings of `total` ints: 512 MB at most)
  // a ring holds what ONE XCD may be handed in the worst case: every task of the launch (an env is published once per
  // step it has left; in a pool once per task that was begun)
  const size_t ring = total;
  const size_t need = (size_t)RV_Q_CTL_WORDS + (size_t)RV_Q_NQ * ring;
  if (w->q_cap < need) {
    if (w->d_q) { HIPCHK(hipStreamSynchronize(w->stream)); HIPCHK(hipFree(w->d_q)); w->d_q = nullptr; w->q_cap = 0; }
    HIPCHK(hipMalloc(&w->d_q, need * sizeof(int)));
    w->q_cap = need;
  }
  if (w->q_used) {      // (the error word of the last queued launch, before it is zeroed)
    int err = 0;
    HIPCHK(hipMemcpyAsync(&err, w->d_q + RV_Q_ERR, sizeof(int), hipMemcpyDeviceToHost, w->stream));
    HIPCHK(hipStreamSynchronize(w->stream));
    w->q_used = false;
    if (err) return fail(RV_ERR_STATE, "task queue: a block arrived with the wrong step / launch number, or a ring overflowed (code " + std::to_string(err) + ") in the previous queued launch");
  }
  a.q_ctl = w->d_q; a.q_slots = w->d_q + RV_Q_CTL_WORDS; a.q_cap = (int)ring; a.q_total = (int)total; a.q_pool = pool > 0 ? 1 : 0;
  a.q_launch = ++w->q_launch;
  a.q_debug = getenv("RV_QUEUE_DEBUG") != nullptr;
  a.q_global = getenv("RV_QUEUE_GLOBAL") != nullptr;
  { const char* st = getenv("RV_QUEUE_STICKY"); a.q_sticky = st ? atoi(st) : 1; }
  { const char* wt = getenv("RV_QUEUE_WT"); a.q_wt = wt ? atoi(wt) : 1; }      // (measurement aid: 0 = plain stores / loads of the block)
  { const char* sl = getenv("RV_QUEUE_STEAL"); a.q_steal = (sl ? atoi(sl) : 0) && a.q_wt; }      // (off: measured, no gain -- rv_env_kernel.h)
  hipLaunchKernelGGL(k_queue_init, dim3((unsigned)((need + 255) / 256)), dim3(256), 0, w->stream, w->d_q, (long long)need);
  HIPCHK(hipGetLastError());
  w->q_used = true;
  return RV_OK;
}
// RV_POISON_LDS builds (tools/build_poison.py): which words of the scratch block start as garbage; everything by default
static void poison_range(EnvKernelArgs& a) {
  const char* lo = getenv("RV_POISON_LO"); const char* hi = getenv("RV_POISON_HI");
  a.poison_lo = lo ? atoi(lo) : 0; a.poison_hi = hi ? atoi(hi) : 0x7fffffff;
}
template <int MODE>
static int launch_env(rv_world* w, const uint8_t* mask, int n_sub, float lin, float ang, int ca, int ms, int mx,
                      int first_index = 0, int auto_reset = 0, const RolloutRec* rec = nullptr,
                      int* budget = nullptr, int32_t* steps_taken = nullptr, long long pool_tasks = 0) {
  EnvKernelArgs a;
  a.budget = budget; a.steps_taken = steps_taken; a.budget_clk = 0; a.finished = nullptr;
  a.first_index = first_index; a.auto_reset = auto_reset;
  memset(&a.rec, 0, sizeof(a.rec));
  if (rec) a.rec = *rec;
  a.cfg = w->d_cfg; a.scene = w->d_scene; a.envs = w->d_envs; a.mask = mask; a.n_envs = w->n;
  { const char* ds = getenv("RV_DEBUG_STOP"); a.stop_after = ds ? atoi(ds) : 0; }
  a.n_substeps = n_sub; a.lin_thr = lin; a.ang_thr = ang; a.check_after = ca; a.min_stable = ms; a.max_steps = mx;
  a.q_slots = nullptr; a.q_ctl = nullptr; a.q_cap = 0; a.q_total = 0; a.q_pool = 0; a.q_launch = 0; a.q_wt = 0; a.q_sticky = 0; a.q_debug = 0; a.q_global = 0; a.q_steal = 0;
  poison_range(a);
  if (MODE == MODE_ROLLOUT && budget == nullptr) { int rc = queue_setup(w, n_sub, a); if (rc != RV_OK) return rc; }
  if (MODE == MODE_ROLLOUT && budget != nullptr && pool_tasks > 0) {
    // the work-conserving rollout of a world with more envs than resident workgroups: through the queue, or the
    // workgroups that are resident first would use the pool up before the others have started
    int rc = queue_setup(w, 0, a, pool_tasks); if (rc != RV_OK) return rc;
    if (a.q_slots) { a.budget = nullptr; a.n_substeps = 0x7fffffff; }      // (an env always has "steps left": the pool ends the launch)
  }
  HIPCHK(hipMemsetAsync(w->d_stats, 0, sizeof(rv_macro_stats), w->stream));
  HIPCHK(hipEventRecord(w->ev0, w->stream));
  launch_k_env(w, MODE, a);
  HIPCHK(hipGetLastError());
  HIPCHK(hipEventRecord(w->ev1, w->stream));
  w->timed = true;
  hipLaunchKernelGGL(k_stats, grid1(w->n), dim3(TPB), 0, w->stream, w->d_envs, w->n, w->d_stats, w->cfg.success_thresh);
  HIPCHK(hipGetLastError());
  return RV_OK;
}

static int ensure_snaps(rv_world* w, size_t n) {
  if (w->n_snaps_cap >= n) return RV_OK;
  if (w->d_snaps) { HIPCHK(hipStreamSynchronize(w->stream)); HIPCHK(hipFree(w->d_snaps)); w->d_snaps = nullptr; w->n_snaps_cap = 0; }
  HIPCHK(hipMalloc(&w->d_snaps, sizeof(ObsSnap) * n));
  w->n_snaps_cap = n;
  return RV_OK;
}
static int launch_point_cloud(rv_world* w, size_t n_snaps, float* d_out) {
  hipLaunchKernelGGL(k_point_cloud, dim3((unsigned)(n_snaps * RV_MAXB)), dim3(64), 0, w->stream,
                     w->d_snaps, (int)n_snaps, w->n, d_out, w->d_cfg, w->d_scene);
  HIPCHK(hipGetLastError());
  return RV_OK;
}

extern "C" {

const char* rv_last_error(void) { return g_err.c_str(); }

int rv_create(const rv_config* cfg, const rv_scene* scene, int device, rv_world** out) {
  if (!cfg || !scene || !out) return fail(RV_ERR_VALUE, "rv_create: null argument");
  if (cfg->n_envs <= 0) return fail(RV_ERR_VALUE, "rv_create: n_envs must be positive");
  if (cfg->n_bodies_max > RV_MAXB || cfg->n_bodies_min < 1 || cfg->n_bodies_min > cfg->n_bodies_max)
    return fail(RV_ERR_VALUE, "rv_create: body count outside [1, RV_MAXB]");
  if (scene->n_shapes <= 0 || scene->n_shapes > RV_MAX_SHAPES) return fail(RV_ERR_VALUE, "rv_create: bad n_shapes");
  if (cfg->wall_use && (cfg->n_bodies_max > RV_MAXB - 1 || cfg->wall_shape < 0 || cfg->wall_shape >= scene->n_shapes || !(cfg->wall_scale > 0.0f)))
    return fail(RV_ERR_VALUE, "rv_create: wall_use needs n_bodies_max <= RV_MAXB - 1, a shape template of the scene and a positive scale");
  int ndev = 0;
  hipError_t e0 = hipGetDeviceCount(&ndev);
  if (e0 != hipSuccess || ndev == 0)
    return fail(RV_ERR_HIP, "rv_create: no HIP device available (librovat_hip has no CPU fallback)");
  if (device < 0 || device >= ndev) return fail(RV_ERR_VALUE, "rv_create: bad device index");
  HIPCHK(hipSetDevice(device));
  rv_world* w = new (std::nothrow) rv_world();
  if (!w) return fail(RV_ERR_STATE, "rv_create: out of host memory");
  w->cfg = *cfg; w->device = device; w->n = cfg->n_envs; w->stream = nullptr; w->timed = false;
  w->d_snaps = nullptr; w->n_snaps_cap = 0;
  {
    // one wave per env: with more envs than SIMDs the two-waves-per-SIMD build of the env kernel pays
    // (RV_ENV_OCC=1 / 2 in the environment forces a build: measurements)
    int cus = 0;
    HIPCHK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device));
    w->occ2 = w->n > 4 * cus;
    w->q_grid = 0; w->d_q = nullptr; w->q_cap = 0; w->q_launch = 0; w->q_used = false;
    const char* f = getenv("RV_ENV_OCC");
    if (f && (f[0] == '1' || f[0] == '2')) w->occ2 = f[0] == '2';
    // the task queue's grid: every workgroup the GPU keeps resident of the kernel this world launches
    // (the occupancy query does not see the waves-per-SIMD cap of the build: one wave per SIMD at 512 registers, two at
    // 256 -- and the 20 KB LDS block allows eight workgroups per CU at most)
    int per_cu = w->occ2 ? rv_k_env_occ2_blocks_per_cu() : rv_k_env_blocks_per_cu_here();
    const int cap = w->occ2 ? 8 : 4;
    if (per_cu > cap) per_cu = cap;
    w->q_grid = per_cu > 0 ? per_cu * cus : 0;
  }
  HIPCHK(hipMalloc(&w->d_cfg, sizeof(rv_config)));
  HIPCHK(hipMalloc(&w->d_scene, sizeof(rv_scene)));
  HIPCHK(hipMalloc(&w->d_envs, sizeof(DevEnv) * (size_t)w->n));
  HIPCHK(hipMalloc(&w->d_stats, sizeof(rv_macro_stats)));
  HIPCHK(hipMalloc(&w->d_budget, sizeof(int)));
  HIPCHK(hipMemcpy(w->d_cfg, cfg, sizeof(rv_config), hipMemcpyHostToDevice));
  HIPCHK(hipMemcpy(w->d_scene, scene, sizeof(rv_scene), hipMemcpyHostToDevice));
  HIPCHK(hipMemset(w->d_stats, 0, sizeof(rv_macro_stats)));
  HIPCHK(hipEventCreate(&w->ev0));
  HIPCHK(hipEventCreate(&w->ev1));
  hipLaunchKernelGGL(k_init, grid1(w->n), dim3(TPB), 0, w->stream, w->d_envs, w->n, cfg->arm_friction, cfg->table_friction, w->d_cfg);
  HIPCHK(hipGetLastError());
  HIPCHK(hipStreamSynchronize(w->stream));
  *out = w;
  return RV_OK;
}

int rv_destroy(rv_world* w) {
  if (!w) return RV_OK;
  (void)hipSetDevice(w->device);
  (void)hipStreamSynchronize(w->stream);
  (void)hipFree(w->d_cfg); (void)hipFree(w->d_scene);
  (void)hipFree(w->d_envs); (void)hipFree(w->d_stats); (void)hipFree(w->d_budget); if (w->d_q) (void)hipFree(w->d_q);
  if (w->d_snaps) (void)hipFree(w->d_snaps);
  (void)hipEventDestroy(w->ev0); (void)hipEventDestroy(w->ev1);
  delete w;
  return RV_OK;
}

int rv_set_stream(rv_world* w, void* s) { WCHK(w); w->stream = (hipStream_t)s; return RV_OK; }
int rv_synchronize(rv_world* w) { WCHK(w); HIPCHK(hipStreamSynchronize(w->stream)); return RV_OK; }
int rv_num_envs(const rv_world* w) { return w ? w->n : 0; }

int rv_reset(rv_world* w, const uint8_t* d_env_mask) { WCHK(w); return launch_env<MODE_RESET>(w, d_env_mask, 0, 0, 0, 0, 0, 0); }
int rv_step_macro(rv_world* w) { WCHK(w); return launch_env<MODE_MACRO>(w, nullptr, 0, 0, 0, 0, 0, 0); }
int rv_rollout(rv_world* w, int32_t n_steps, int32_t first_macro_index, int32_t auto_reset, float* d_rewards, uint8_t* d_dones) {
  WCHK(w);
  if (n_steps <= 0) return fail(RV_ERR_VALUE, "rv_rollout: n_steps must be positive");
  RolloutRec rec; memset(&rec, 0, sizeof(rec));
  rec.rewards = d_rewards; rec.dones = d_dones;
  return launch_env<MODE_ROLLOUT>(w, nullptr, n_steps, 0, 0, 0, 0, 0, first_macro_index, auto_reset, &rec);
}
int rv_rollout_record(rv_world* w, int32_t n_steps, int32_t first_macro_index, int32_t auto_reset,
                      float* d_rewards, uint8_t* d_dones, const rv_obs_buffers* step_obs) {
  WCHK(w);
  if (n_steps <= 0) return fail(RV_ERR_VALUE, "rv_rollout_record: n_steps must be positive");
  RolloutRec rec; memset(&rec, 0, sizeof(rec));
  rec.rewards = d_rewards; rec.dones = d_dones;
  float* d_pc = nullptr;
  if (step_obs) {
    rec.obs = *step_obs; rec.has_obs = 1; d_pc = step_obs->d_point_cloud;
    rec.obs.d_point_cloud = nullptr;
  }
  const size_t rows = (size_t)n_steps * (size_t)w->n;
  if (d_pc) {
    if (w->cfg.num_points <= 0 || w->cfg.num_points > RV_PC_MAXPIX) return fail(RV_ERR_VALUE, "rv_rollout_record: num_points outside [1, RV_PC_MAXPIX]");
    int rc = ensure_snaps(w, rows); if (rc != RV_OK) return rc;
    rec.snaps = w->d_snaps;
  }
  int rc = launch_env<MODE_ROLLOUT>(w, nullptr, n_steps, 0, 0, 0, 0, 0, first_macro_index, auto_reset, &rec);
  if (rc != RV_OK) return rc;
  // the segmented point clouds of all n_steps x N observations, rendered together
  if (d_pc) return launch_point_cloud(w, rows, d_pc);
  return RV_OK;
}
int rv_rollout_record_full(rv_world* w, int32_t n_steps, int32_t first_macro_index, int32_t auto_reset,
                           float* d_rewards, uint8_t* d_dones, const rv_obs_buffers* step_obs, const rv_rollout_extra* extra) {
  WCHK(w);
  if (n_steps <= 0) return fail(RV_ERR_VALUE, "rv_rollout_record_full: n_steps must be positive");
  RolloutRec rec; memset(&rec, 0, sizeof(rec));
  rec.rewards = d_rewards; rec.dones = d_dones;
  float *d_pc = nullptr, *d_rpc = nullptr;
  if (step_obs) { rec.obs = *step_obs; rec.has_obs = 1; d_pc = step_obs->d_point_cloud; rec.obs.d_point_cloud = nullptr; }
  if (extra) {
    rec.actions = extra->d_actions; rec.resets = extra->d_reset;
    rec.robs = extra->reset_obs; rec.has_robs = 1; d_rpc = extra->reset_obs.d_point_cloud; rec.robs.d_point_cloud = nullptr;
  }
  const size_t rows = (size_t)n_steps * (size_t)w->n;
  if (d_pc || d_rpc) {
    if (w->cfg.num_points <= 0 || w->cfg.num_points > RV_PC_MAXPIX) return fail(RV_ERR_VALUE, "rv_rollout_record_full: num_points outside [1, RV_PC_MAXPIX]");
    int rc = ensure_snaps(w, 2 * rows); if (rc != RV_OK) return rc;
    if (d_pc) rec.snaps = w->d_snaps;
    if (d_rpc) {
      rec.rsnaps = w->d_snaps + rows;
      // rows without a reset keep shape = -1 for every body (all bits set): empty clouds
      HIPCHK(hipMemsetAsync(rec.rsnaps, 0xFF, sizeof(ObsSnap) * rows, w->stream));
    }
  }
  if (extra && extra->d_reset) HIPCHK(hipMemsetAsync(extra->d_reset, 0, rows, w->stream));
  int rc = launch_env<MODE_ROLLOUT>(w, nullptr, n_steps, 0, 0, 0, 0, 0, first_macro_index, auto_reset, &rec);
  if (rc != RV_OK) return rc;
  if (d_pc) { rc = launch_point_cloud(w, rows, d_pc); if (rc != RV_OK) return rc; }
  if (d_rpc) {
    hipLaunchKernelGGL(k_point_cloud, dim3((unsigned)(rows * RV_MAXB)), dim3(64), 0, w->stream,
                       w->d_snaps + rows, (int)rows, w->n, d_rpc, w->d_cfg, w->d_scene);
    HIPCHK(hipGetLastError());
  }
  return RV_OK;
}
int rv_rollout_async(rv_world* w, int32_t total_env_steps, int32_t first_macro_index, int32_t* d_steps_taken) {
  WCHK(w);
  if (total_env_steps <= 0) return fail(RV_ERR_VALUE, "rv_rollout_async: total_env_steps must be positive");
  hipLaunchKernelGGL(k_set_int, dim3(1), dim3(1), 0, w->stream, w->d_budget, (int)total_env_steps);
  HIPCHK(hipGetLastError());
  return launch_env<MODE_ROLLOUT>(w, nullptr, 0, 0, 0, 0, 0, 0, first_macro_index, 1, nullptr, w->d_budget, d_steps_taken, (long long)total_env_steps);
}
int rv_step_begin(rv_world* w, const float* d_actions, const uint8_t* d_mask) {
  WCHK(w);
  if (!d_actions) return fail(RV_ERR_VALUE, "rv_step_begin: null buffer");
  const int G = w->cfg.num_goal_steps > 0 ? w->cfg.num_goal_steps : 1;
  hipLaunchKernelGGL(k_step_begin, grid1(w->n), dim3(TPB), 0, w->stream, w->d_envs, w->n, d_actions, d_mask, G);
  HIPCHK(hipGetLastError());
  return RV_OK;
}
int rv_set_auto_reset(rv_world* w, int32_t on) { WCHK(w); w->auto_reset = on != 0; return RV_OK; }
int rv_step_poll(rv_world* w, int32_t max_substeps, int32_t max_usec, uint8_t* d_finished,
                 const rv_obs_buffers* obs, float* d_reward, uint8_t* d_done) {
  WCHK(w);
  if (!d_finished) return fail(RV_ERR_VALUE, "rv_step_poll: null buffer");
  if (max_substeps < 0 || max_usec < 0) return fail(RV_ERR_VALUE, "rv_step_poll: negative budget");
  EnvKernelArgs a;
  memset(&a, 0, sizeof(a));
  a.cfg = w->d_cfg; a.scene = w->d_scene; a.envs = w->d_envs; a.n_envs = w->n;
  a.n_substeps = max_substeps; a.finished = d_finished; a.auto_reset = w->auto_reset;
  poison_range(a);
  a.rec.rewards = d_reward; a.rec.dones = d_done;
  float* d_pc = nullptr;
  if (obs) {
    a.rec.obs = *obs; a.rec.has_obs = 1; d_pc = obs->d_point_cloud; a.rec.obs.d_point_cloud = nullptr;
    if (d_pc) {
      if (w->cfg.num_points <= 0 || w->cfg.num_points > RV_PC_MAXPIX) return fail(RV_ERR_VALUE, "rv_step_poll: num_points outside [1, RV_PC_MAXPIX]");
      int rc = ensure_snaps(w, (size_t)w->n); if (rc != RV_OK) return rc;
      a.rec.snaps = w->d_snaps;
      HIPCHK(hipMemsetAsync(w->d_snaps, 0xFF, sizeof(ObsSnap) * (size_t)w->n, w->stream));   // shape = -1: no cloud for envs that do not finish
    }
  }
  if (max_usec > 0) {
    int khz = 0;
    HIPCHK(hipDeviceGetAttribute(&khz, hipDeviceAttributeClockRate, w->device));   // s_memtime ticks at the shader clock on gfx950
    a.budget_clk = (unsigned long long)max_usec * (unsigned long long)(khz > 0 ? khz : 2400000) / 1000ull;
  }
  HIPCHK(hipMemsetAsync(w->d_stats, 0, sizeof(rv_macro_stats), w->stream));
  HIPCHK(hipEventRecord(w->ev0, w->stream));
  launch_k_env(w, MODE_PARTIAL, a);
  HIPCHK(hipGetLastError());
  HIPCHK(hipEventRecord(w->ev1, w->stream));
  w->timed = true;
  hipLaunchKernelGGL(k_stats, grid1(w->n), dim3(TPB), 0, w->stream, w->d_envs, w->n, w->d_stats, w->cfg.success_thresh);
  HIPCHK(hipGetLastError());
  if (d_pc) return launch_point_cloud(w, (size_t)w->n, d_pc);
  return RV_OK;
}
int rv_step_sub(rv_world* w, int32_t n) {
  WCHK(w);
  if (n < 0) return fail(RV_ERR_VALUE, "rv_step_sub: negative substep count");
  return launch_env<MODE_SUB>(w, nullptr, n, 0, 0, 0, 0, 0);
}
int rv_wait_until_stable(rv_world* w, float lin, float ang, int32_t ca, int32_t ms, int32_t mx) {
  WCHK(w);
  if (mx <= 0) return fail(RV_ERR_VALUE, "rv_wait_until_stable: max_steps must be positive");
  return launch_env<MODE_WAIT>(w, nullptr, 0, lin, ang, ca, ms, mx);
}

#define SIMPLE_LAUNCH(kern, ...) do { hipLaunchKernelGGL(kern, grid1(w->n), dim3(TPB), 0, w->stream, __VA_ARGS__); HIPCHK(hipGetLastError()); } while (0)
#define NEED(p, name) do { if (!(p)) return fail(RV_ERR_VALUE, name ": null buffer"); } while (0)

int rv_set_actions(rv_world* w, const float* d) {
  WCHK(w); NEED(d, "rv_set_actions");
  int G = w->cfg.num_goal_steps > 0 ? w->cfg.num_goal_steps : 1;
  SIMPLE_LAUNCH(k_set_actions, w->d_envs, w->n, d, G); return RV_OK;
}
int rv_policy_random(rv_world* w, int32_t macro_index, float* d) {
  WCHK(w); NEED(d, "rv_policy_random");
  SIMPLE_LAUNCH(k_policy_random, w->n, w->d_cfg, macro_index, d); return RV_OK;
}
int rv_policy_heuristic(rv_world* w, int32_t max_attempts, float* d) {
  WCHK(w); NEED(d, "rv_policy_heuristic");
  if (max_attempts <= 0) return fail(RV_ERR_VALUE, "rv_policy_heuristic: max_attempts must be positive");
  hipLaunchKernelGGL(k_policy_heuristic, dim3((unsigned)w->n), dim3(64), 0, w->stream, w->d_envs, w->n, w->d_cfg, max_attempts, d);
  HIPCHK(hipGetLastError());
  return RV_OK;
}
int rv_get_body_state(rv_world* w, float* d) { WCHK(w); NEED(d, "rv_get_body_state"); SIMPLE_LAUNCH(k_get_body_state, w->d_envs, w->n, d); return RV_OK; }
int rv_set_body_state(rv_world* w, const float* d) { WCHK(w); NEED(d, "rv_set_body_state"); SIMPLE_LAUNCH(k_set_body_state, w->d_envs, w->n, d); return RV_OK; }
int rv_get_body_params(rv_world* w, float* d) { WCHK(w); NEED(d, "rv_get_body_params"); SIMPLE_LAUNCH(k_get_body_params, w->d_envs, w->n, d); return RV_OK; }
int rv_set_body_params(rv_world* w, const float* d) { WCHK(w); NEED(d, "rv_set_body_params"); SIMPLE_LAUNCH(k_set_body_params, w->d_envs, w->n, d, w->d_cfg, w->d_scene); return RV_OK; }
int rv_get_joint_state(rv_world* w, float* d) { WCHK(w); NEED(d, "rv_get_joint_state"); SIMPLE_LAUNCH(k_get_joint_state, w->d_envs, w->n, d); return RV_OK; }
int rv_set_joint_state(rv_world* w, const float* d) { WCHK(w); NEED(d, "rv_set_joint_state"); SIMPLE_LAUNCH(k_set_joint_state, w->d_envs, w->n, d, w->d_cfg, w->d_scene); return RV_OK; }
int rv_get_link_poses(rv_world* w, float* d) { WCHK(w); NEED(d, "rv_get_link_poses"); SIMPLE_LAUNCH(k_get_link_poses, w->d_envs, w->n, d); return RV_OK; }
#ifdef RV_PROFILE
// profiling build only (tools/prof_rollout.py): per-env shader-clock time per substep part
int rv_debug_profile(rv_world* w, unsigned long long* d) { WCHK(w); NEED(d, "rv_debug_profile"); SIMPLE_LAUNCH(k_debug_profile, w->d_envs, w->n, d); return RV_OK; }
#endif
int rv_get_env_counters(rv_world* w, int32_t* d) { WCHK(w); NEED(d, "rv_get_env_counters"); SIMPLE_LAUNCH(k_get_env_counters, w->d_envs, w->n, d); return RV_OK; }
int rv_set_joint_targets(rv_world* w, const float* d, float timeout, float threshold) { WCHK(w); NEED(d, "rv_set_joint_targets"); SIMPLE_LAUNCH(k_set_joint_targets, w->d_envs, w->n, d, w->d_cfg, w->d_scene, timeout, threshold); return RV_OK; }
int rv_set_link_target(rv_world* w, const float* d, float timeout, float threshold) { WCHK(w); NEED(d, "rv_set_link_target"); SIMPLE_LAUNCH(k_set_link_target, w->d_envs, w->n, d, w->d_cfg, w->d_scene, timeout, threshold); return RV_OK; }
int rv_set_link_path(rv_world* w, const float* d, int32_t n_poses, float timeout, float threshold) {
  WCHK(w); NEED(d, "rv_set_link_path");
  if (n_poses < 1 || n_poses > RV_MAXQ) return fail(RV_ERR_VALUE, "rv_set_link_path: 1 <= n_poses <= RV_MAXQ");
  SIMPLE_LAUNCH(k_set_link_path, w->d_envs, w->n, d, n_poses, w->d_cfg, w->d_scene, timeout, threshold); return RV_OK;
}
int rv_set_max_joint_velocities(rv_world* w, const float* d) {
  WCHK(w); NEED(d, "rv_set_max_joint_velocities");
  SIMPLE_LAUNCH(k_set_max_joint_velocities, w->d_envs, w->n, d); return RV_OK;
}
int rv_get_camera(rv_world* w, float* d) { WCHK(w); NEED(d, "rv_get_camera"); SIMPLE_LAUNCH(k_get_camera, w->d_envs, w->n, d); return RV_OK; }
int rv_get_robot_ready(rv_world* w, uint8_t* d) { WCHK(w); NEED(d, "rv_get_robot_ready"); SIMPLE_LAUNCH(k_robot_ready, w->d_envs, w->n, d, w->d_cfg); return RV_OK; }
int rv_set_motor_targets(rv_world* w, const float* d_q, const uint8_t* d_mask) {
  WCHK(w); NEED(d_q, "rv_set_motor_targets");
  SIMPLE_LAUNCH(k_set_motor_targets, w->d_envs, w->n, d_q, d_mask, w->d_cfg); return RV_OK;
}
int rv_set_gravity(rv_world* w, const float* g) {
  WCHK(w); NEED(g, "rv_set_gravity");
  w->cfg.gravity_xy[0] = g[0]; w->cfg.gravity_xy[1] = g[1]; w->cfg.gravity_z = g[2];
  HIPCHK(hipMemcpyAsync(w->d_cfg, &w->cfg, sizeof(rv_config), hipMemcpyHostToDevice, w->stream));
  HIPCHK(hipStreamSynchronize(w->stream));
  return RV_OK;
}
int rv_set_friction(rv_world* w, float mu_finger, float mu_table) {
  WCHK(w);
  SIMPLE_LAUNCH(k_set_friction, w->d_envs, w->n, mu_finger, mu_table);
  return RV_OK;
}
int rv_set_constraint_ex(rv_world* w, int32_t body, int32_t child, int32_t joint_type, const float* frame7, const float* child_frame7, float max_force) {
  WCHK(w);
  if (body < 0 || body >= RV_MAXB) return fail(RV_ERR_VALUE, "rv_set_constraint: not a movable body slot");
  if (child < -1 || child >= RV_MAXB + RV_NFRAME || child == body) return fail(RV_ERR_VALUE, "rv_set_constraint: the child is the world (-1), another movable body slot or RV_CHILD_LINK(frame)");
  if (child >= RV_MAXB && joint_type != RV_JOINT_FIXED && joint_type != RV_JOINT_POINT2POINT) return fail(RV_ERR_NOTIMPL, "rv_set_constraint: a link of the arm as the other party: fixed and point2point joints");
  if (joint_type != RV_JOINT_FIXED && joint_type != RV_JOINT_POINT2POINT && joint_type != RV_JOINT_PRISMATIC && joint_type != RV_JOINT_REVOLUTE) return fail(RV_ERR_NOTIMPL, "rv_set_constraint: joint types built: fixed, point2point, prismatic, revolute");
  if (max_force >= 0.0f && !child_frame7) return fail(RV_ERR_VALUE, "rv_set_constraint: null target");
  ConArgs a; memset(&a, 0, sizeof(a));
  a.body = body; a.child = child; a.type = joint_type == RV_JOINT_POINT2POINT ? 2 : (joint_type == RV_JOINT_PRISMATIC ? 3 : (joint_type == RV_JOINT_REVOLUTE ? 4 : 1)); a.fmax = max_force; a.lq[3] = 1.0f; a.tq[3] = 1.0f;
  if (max_force >= 0.0f) {
    if (frame7) { for (int k = 0; k < 3; ++k) a.lp[k] = frame7[k]; for (int k = 0; k < 4; ++k) a.lq[k] = frame7[3 + k]; }
    for (int k = 0; k < 3; ++k) a.tp[k] = child_frame7[k];
    for (int k = 0; k < 4; ++k) a.tq[k] = child_frame7[3 + k];
  }
  SIMPLE_LAUNCH(k_set_constraint, w->d_envs, w->n, a);
  return RV_OK;
}
int rv_set_constraint(rv_world* w, int32_t body, const float* frame7, const float* target7, float max_force) {
  return rv_set_constraint_ex(w, body, -1, RV_JOINT_FIXED, frame7, target7, max_force);
}
int rv_grip(rv_world* w, float value) { WCHK(w); SIMPLE_LAUNCH(k_grip, w->d_envs, w->n, value, w->d_cfg, w->d_scene); return RV_OK; }
int rv_reset_targets(rv_world* w) { WCHK(w); SIMPLE_LAUNCH(k_reset_targets, w->d_envs, w->n); return RV_OK; }
int rv_get_state_ptrs(rv_world* w, rv_state_view* v) {
  WCHK(w); NEED(v, "rv_get_state_ptrs");
  v->d_envs = w->d_envs; v->env_stride_bytes = (int64_t)sizeof(DevEnv);
  v->off_body = (int64_t)offsetof(DevEnv, body); v->off_active = (int64_t)offsetof(DevEnv, active);
  v->off_joint_q = (int64_t)offsetof(DevEnv, q); v->off_joint_qd = (int64_t)offsetof(DevEnv, qd);
  v->off_link_pos = (int64_t)offsetof(DevEnv, fpos); v->off_link_quat = (int64_t)offsetof(DevEnv, fquat);
  v->off_obs_pos = (int64_t)offsetof(DevEnv, obs_pos); v->off_table_z = (int64_t)offsetof(DevEnv, table_z);
  return RV_OK;
}
#ifndef RV_SOURCE_HASH
#define RV_SOURCE_HASH "unknown"
#endif
const char* rv_source_hash(void) { return RV_SOURCE_HASH; }
int rv_compute_ik(rv_world* w, const float* d_pose, float* d_q) {
  WCHK(w); NEED(d_pose, "rv_compute_ik"); NEED(d_q, "rv_compute_ik");
  SIMPLE_LAUNCH(k_compute_ik, w->d_envs, w->n, d_pose, d_q, w->d_cfg, w->d_scene); return RV_OK;
}
int rv_query_contacts(rv_world* w, uint8_t* d) { WCHK(w); NEED(d, "rv_query_contacts"); SIMPLE_LAUNCH(k_query_contacts, w->d_envs, w->n, d); return RV_OK; }
int rv_get_manifold_counts(rv_world* w, int32_t* d) { WCHK(w); NEED(d, "rv_get_manifold_counts"); SIMPLE_LAUNCH(k_manifold_counts, w->d_envs, w->n, d); return RV_OK; }
int rv_observe(rv_world* w, const rv_obs_buffers* obs) {
  WCHK(w); NEED(obs, "rv_observe");
  SIMPLE_LAUNCH(k_observe, w->d_envs, w->n, *obs, w->d_cfg);
  if (obs->d_point_cloud) {
    if (w->cfg.num_points <= 0 || w->cfg.num_points > RV_PC_MAXPIX) return fail(RV_ERR_VALUE, "rv_observe: num_points outside [1, RV_PC_MAXPIX]");
    int rc = ensure_snaps(w, (size_t)w->n); if (rc != RV_OK) return rc;
    SIMPLE_LAUNCH(k_obs_snap, w->d_envs, w->n, w->d_snaps, w->d_scene);
    rc = launch_point_cloud(w, w->n, obs->d_point_cloud); if (rc != RV_OK) return rc;
  }
  return RV_OK;
}
int rv_render_rgb(rv_world* w, uint8_t* d_rgb) {
  WCHK(w); NEED(d_rgb, "rv_render_rgb");
  int rc = ensure_snaps(w, (size_t)w->n); if (rc != RV_OK) return rc;
  SIMPLE_LAUNCH(k_obs_snap, w->d_envs, w->n, w->d_snaps, w->d_scene);
  const size_t total = (size_t)w->n * (size_t)w->cfg.cam_height * (size_t)w->cfg.cam_width;
  hipLaunchKernelGGL(k_render_rgb, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, w->stream, w->d_snaps, w->n, d_rgb, w->d_cfg, w->d_scene);
  HIPCHK(hipGetLastError());
  return RV_OK;
}
int rv_render(rv_world* w, float* d_depth, uint8_t* d_segmask) {
  WCHK(w);
  if (!d_depth && !d_segmask) return fail(RV_ERR_VALUE, "rv_render: no output buffer");
  int rc = ensure_snaps(w, (size_t)w->n); if (rc != RV_OK) return rc;
  SIMPLE_LAUNCH(k_obs_snap, w->d_envs, w->n, w->d_snaps, w->d_scene);
  const size_t total = (size_t)w->n * (size_t)w->cfg.cam_height * (size_t)w->cfg.cam_width;
  hipLaunchKernelGGL(k_render, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, w->stream, w->d_snaps, w->n, d_depth, d_segmask, w->d_cfg, w->d_scene);
  HIPCHK(hipGetLastError());
  return RV_OK;
}
int rv_reward(rv_world* w, float* d_reward, uint8_t* d_done) { WCHK(w); SIMPLE_LAUNCH(k_reward, w->d_envs, w->n, d_reward, d_done); return RV_OK; }
int rv_get_episode_returns(rv_world* w, float* d) { WCHK(w); NEED(d, "rv_get_episode_returns"); SIMPLE_LAUNCH(k_returns, w->d_envs, w->n, d); return RV_OK; }
int rv_get_stats(rv_world* w, rv_macro_stats* h) {
  WCHK(w); NEED(h, "rv_get_stats");
  HIPCHK(hipMemcpyAsync(h, w->d_stats, sizeof(rv_macro_stats), hipMemcpyDeviceToHost, w->stream));
  int q_err = 0;
  if (w->q_used) HIPCHK(hipMemcpyAsync(&q_err, w->d_q + RV_Q_ERR, sizeof(int), hipMemcpyDeviceToHost, w->stream));
  HIPCHK(hipStreamSynchronize(w->stream));
  if (w->q_used && getenv("RV_QUEUE_DEBUG")) {      // (measurement aid: what every XCD's queue did in the last queued launch)
    int dbg[8 * RV_Q_NQ];
    HIPCHK(hipMemcpy(dbg, w->d_q + RV_Q_DBG(0), sizeof(dbg), hipMemcpyDeviceToHost));
    int t_max = 0;
    for (int x = 0; x < RV_Q_NQ; ++x) if (dbg[8 * x] > 0 && dbg[8 * x + 2] > t_max) t_max = dbg[8 * x + 2];
    for (int x = 0; x < RV_Q_NQ; ++x)      // (s_memrealtime: 100 MHz, one counter for the device; >> 4: units of 0.16 us)
      fprintf(stderr, "queue of XCD %d: %d workgroups, %d tasks, %d envs bound to it, %d times an env was kept, %d tasks taken from another XCD's queue, its last workgroup left %.2f ms before the last of all\n",
              x, dbg[8 * x + 4], dbg[8 * x], dbg[8 * x + 3], dbg[8 * x + 1], dbg[8 * x + 5], dbg[8 * x] > 0 ? (t_max - dbg[8 * x + 2]) * 16.0 / 100e6 * 1e3 : 0.0);
  }
  if (w->q_used) {
    w->q_used = false;
    // (rv_env_kernel.h: the hand-over of an env block between two tasks is asserted, never repaired)
    if (q_err) return fail(RV_ERR_STATE, "task queue: a block arrived with the wrong step / launch number, or a ring overflowed (code " + std::to_string(q_err) + ")");
  }
  return RV_OK;
}
int rv_last_kernel_ms(rv_world* w, float* h_ms) {
  WCHK(w); NEED(h_ms, "rv_last_kernel_ms");
  if (!w->timed) return fail(RV_ERR_STATE, "rv_last_kernel_ms: no env kernel launched yet");
  HIPCHK(hipEventSynchronize(w->ev1));
  HIPCHK(hipEventElapsedTime(h_ms, w->ev0, w->ev1));
  return RV_OK;
}

}  // extern "C"

// rv_dev_obs.h — SegmentedPointCloudObs on the device.
//
// Reference chain (StanfordVL/robovat):
//   BulletCamera._frames            robovat/simulation/camera/bullet_camera.py:188-235
//     render depth + segmask of the 424x512 simulated Kinect2, depth linearised to eye z
//   Camera.deproject_depth_image    robovat/perception/camera/camera.py:213-244
//     point = cam_position + R^T (depth * K^-1 [u, v, 1])
//   convert_segment_ids / group_by_labels   robovat/perception/point_cloud_utils.py:110-157
//     per body: num_points pixels drawn from the body's visible pixels, WITH replacement
//     iff it has fewer than num_points, zeros when it has none (downsample, :23-39)
//
// Here: one wave64 per (observation, body).  The wave ray-casts the pixels of the body's
// screen rectangle against the convex hulls of every body (face planes of rv_shape) and
// the table, keeps the pixels on which this body is the nearest hit (segmentation),
// compacts them into LDS in scan order, and samples: a uniformly random num_points-subset
// (the num_points smallest of per-pixel Philox keys, found by a 32-step radix select, emitted in
// key order; a body with more than RV_PC_MAXPIX visible pixels keeps every stride-th one) or
// num_points draws with replacement.  The observation is a pure function of a small pose
// snapshot (ObsSnap), so a rollout records one snapshot per env.step() and all clouds of
// the launch are rendered together afterwards.
#pragma once
#include "../../include/rovat.h"
#include "rv_dev_math.h"

namespace rv {

#define RV_STREAM_PC 7u

// what the camera needs to know about one env at observation time
struct ObsSnap {
  float pose[RV_MAXB][7];   // pos3, quat4 (xyzw)
  float scale[RV_MAXB];
  int shape[RV_MAXB];       // -1: body absent
  int is_static[RV_MAXB];   // a static body (the wall): rendered -- it occludes -- but not one of the segmented clouds
  float table_z;
  float cam_intrinsics[5], cam_rotation[9], cam_translation[3];   // the env's camera (DevEnv: rv_config's calibration + reset noise)
  uint32_t rng_arg;         // reset_count * 4096 + num_steps: one sampling stream per observation
  // the arm as the camera sees it: the collider boxes of the links (world centre, frame quaternion); arm_on = 0: no arm
  int arm_on;
  float arm_c[RV_NCOL][3], arm_q[RV_NCOL][4];
};

struct CamRay { v3 o, d; };   // world ray; the parameter along d is the eye-space depth

// ray of pixel (u, v): K^-1 [u, v, 1] in the camera frame, rotated into the world
RV_DEV v3 pixel_dir_cam(const ObsSnap* c, float u, float v) {
  const float fx = c->cam_intrinsics[0], fy = c->cam_intrinsics[1], cx = c->cam_intrinsics[2], cy = c->cam_intrinsics[3], sk = c->cam_intrinsics[4];
  float y = (v - cy) / fy;
  float x = (u - cx - sk * y) / fx;
  return mk(x, y, 1.0f);
}
RV_DEV v3 cam_to_world_dir(const ObsSnap* c, v3 d) { return tmulv(c->cam_rotation, d); }
RV_DEV v3 cam_position(const ObsSnap* c) {
  v3 t = ld3(c->cam_translation);
  v3 p = tmulv(c->cam_rotation, t);
  return mk(-p.x, -p.y, -p.z);
}

// entry depth of the ray into one convex hull given by planes n.x <= d*scale + margin, in the
// body frame; returns 0 on a miss
RV_DEV int ray_hull(const float (*planes)[4], int n, float sc, float margin, v3 o, v3 d, float* t_hit, int* plane_hit = nullptr) {
  float t0 = 0.0f, t1 = 1e30f; int ip = -1;
  for (int i = 0; i < n; ++i) {
    v3 nn = mk(planes[i][0], planes[i][1], planes[i][2]);
    float off = planes[i][3] * sc + margin;
    float den = dot(nn, d);
    float num = off - dot(nn, o);
    if (den < 0.0f) { float t = num / den; if (t > t0) { t0 = t; ip = i; } }
    else if (den > 0.0f) { float t = num / den; if (t < t1) t1 = t; }
    else if (num < 0.0f) return 0;
  }
  if (t0 > t1) return 0;
  *t_hit = t0;
  if (plane_hit) *plane_hit = ip;
  return 1;
}
// entry depth into the axis-aligned table slab (axis_hit: 0..2 = the -x/-y/-z face, 3..5 = +x/+y/+z)
RV_DEV int ray_table(const rv_config* c, float table_z, v3 o, v3 d, float* t_hit, int* axis_hit = nullptr) {
  const float lo[3] = {c->table_center[0] - c->table_half[0], c->table_center[1] - c->table_half[1], table_z - c->table_thickness};
  const float hi[3] = {c->table_center[0] + c->table_half[0], c->table_center[1] + c->table_half[1], table_z};
  const float oo[3] = {o.x, o.y, o.z}, dd[3] = {d.x, d.y, d.z};
  float t0 = 0.0f, t1 = 1e30f; int ax = -1;
  for (int k = 0; k < 3; ++k) {
    if (dd[k] != 0.0f) {
      float a = (lo[k] - oo[k]) / dd[k], b = (hi[k] - oo[k]) / dd[k];
      float tn = a < b ? a : b, tf = a < b ? b : a;
      if (tn > t0) { t0 = tn; ax = a < b ? k : 3 + k; }
      if (tf < t1) t1 = tf;
    } else if (oo[k] < lo[k] || oo[k] > hi[k]) return 0;
  }
  if (t0 > t1) return 0;
  *t_hit = t0;
  if (axis_hit) *axis_hit = ax;
  return 1;
}

// entry depth into the box |x_k| <= h_k (ray in the box frame; axis_hit as in ray_table)
RV_DEV int ray_box(const float* h, v3 o, v3 d, float* t_hit, int* axis_hit) {
  const float oo[3] = {o.x, o.y, o.z}, dd[3] = {d.x, d.y, d.z};
  float t0 = 0.0f, t1 = 1e30f; int ax = -1;
  for (int k = 0; k < 3; ++k) {
    if (dd[k] != 0.0f) {
      float a = (-h[k] - oo[k]) / dd[k], b = (h[k] - oo[k]) / dd[k];
      float tn = a < b ? a : b, tf = a < b ? b : a;
      if (tn > t0) { t0 = tn; ax = a < b ? k : 3 + k; }
      if (tf < t1) t1 = tf;
    } else if (oo[k] < -h[k] || oo[k] > h[k]) return 0;
  }
  if (t0 > t1) return 0;
  *t_hit = t0; *axis_hit = ax;
  return 1;
}

// nearest hit of the pixel ray: body index, RV_MAXB for the table, RV_MAXB + 1 for the arm (its link collider
// boxes; arm_mask: the boxes that can be hit at all, a conservative pre-selection of the caller), -1 for nothing
// normal (optional): outward world normal of the surface that is hit
RV_DEV int render_pixel(const rv_config* c, const rv_scene* scene, const ObsSnap& s, const float (*rot)[9],
                        v3 cam_o, v3 dw, float* depth, v3* normal = nullptr, unsigned arm_mask = 0xffffffffu) {
  float best = 1e30f; int who = -1;
  v3 nb = mk(0.0f, 0.0f, 0.0f);
  for (int b = 0; b < RV_MAXB; ++b) {
    if (s.shape[b] < 0) continue;
    const rv_shape* sh = &scene->shapes[s.shape[b]];
    v3 rel = sub(cam_o, ld3(s.pose[b]));
    // bounding sphere: skip bodies the ray passes clear of
    float r = sh->radius * s.scale[b] + c->margin;
    float dd = dot(dw, dw), rd = dot(rel, dw);
    float perp2 = dot(rel, rel) - rd * rd / dd;
    if (perp2 > r * r) continue;
    v3 ol = tmulv(rot[b], rel), dl = tmulv(rot[b], dw);
    for (int h = 0; h < sh->n_hulls; ++h) {
      float t; int ip = -1;
      if (ray_hull(sh->planes[h], sh->n_planes[h], s.scale[b], c->margin, ol, dl, &t, &ip) && t < best) {
        best = t; who = b;
        if (normal && ip >= 0) nb = mulv(rot[b], mk(sh->planes[h][ip][0], sh->planes[h][ip][1], sh->planes[h][ip][2]));
      }
    }
  }
  if (s.arm_on == 1 && arm_mask) {
    const rv_arm* arm = &scene->arm;
    for (int col = 0; col < RV_NCOL; ++col) {
      if (!((arm_mask >> col) & 1u)) continue;
      const float hx = arm->col_half[col][0] + c->margin, hy = arm->col_half[col][1] + c->margin, hz = arm->col_half[col][2] + c->margin;
      v3 rel = sub(cam_o, ld3(s.arm_c[col]));
      float r2 = hx * hx + hy * hy + hz * hz;
      float dd = dot(dw, dw), rd = dot(rel, dw);
      float perp2 = dot(rel, rel) - rd * rd / dd;
      if (perp2 > r2) continue;
      const m3 R = qmat(ldq(s.arm_q[col]));
      const float hh[3] = {hx, hy, hz};
      float t; int ia = -1;
      if (ray_box(hh, tmulv(R, rel), tmulv(R, dw), &t, &ia) && t < best) {
        best = t; who = RV_MAXB + 1;
        if (normal && ia >= 0) { const float sg = ia < 3 ? -1.0f : 1.0f; const int k = ia % 3; nb = mulv(R, mk(k == 0 ? sg : 0.0f, k == 1 ? sg : 0.0f, k == 2 ? sg : 0.0f)); }
      }
    }
  }
  float tt; int ax = -1;
  if (ray_table(c, s.table_z, cam_o, dw, &tt, &ax) && tt < best) {
    best = tt; who = RV_MAXB;
    if (normal && ax >= 0) { const float sg = ax < 3 ? -1.0f : 1.0f; const int k = ax % 3; nb = mk(k == 0 ? sg : 0.0f, k == 1 ? sg : 0.0f, k == 2 ? sg : 0.0f); }
  }
  *depth = best;
  if (normal) *normal = nb;
  return who;
}

// CameraObs 'rgb' (camera_obs.py:33-88; bullet_camera.py:188-235 renders with pybullet's default
// light): flat colours per body slot, table and background, Lambert-shaded with the normal of the
// face that is hit under one fixed directional light
RV_DEV void shade_rgb(int who, v3 n, uint8_t* out) {
  const float base[RV_MAXB + 3][3] = {{230.0f, 60.0f, 60.0f}, {60.0f, 170.0f, 230.0f}, {250.0f, 200.0f, 40.0f}, {90.0f, 200.0f, 110.0f},
                                      {150.0f, 120.0f, 90.0f}, {185.0f, 185.0f, 195.0f}, {30.0f, 30.0f, 30.0f}};   // bodies, table, arm, background
  const int idx = who < 0 ? RV_MAXB + 2 : who;
  float sh = 1.0f;
  if (who >= 0) {
    const float lam = n.x * 0.30151134f + n.y * -0.30151134f + n.z * 0.90453403f;
    sh = 0.35f + 0.65f * (lam > 0.0f ? lam : 0.0f);
  }
  for (int k = 0; k < 3; ++k) out[k] = (uint8_t)(int)(base[idx][k] * sh + 0.5f);
}
// point of pixel (u, v) at eye depth z (Camera.deproject_pixel, camera.py:195-211)
RV_DEV v3 deproject(const ObsSnap* c, v3 cam_o, float u, float v, float z) {
  v3 pc = scale(pixel_dir_cam(c, u, v), z);
  return add(cam_o, tmulv(c->cam_rotation, pc));
}
RV_DEV int crop_ok(const rv_config* c, v3 p) {
  if (!c->use_crop) return 1;
  return p.x >= c->crop_min[0] && p.y >= c->crop_min[1] && p.z >= c->crop_min[2] &&
         p.x <= c->crop_max[0] && p.y <= c->crop_max[1] && p.z <= c->crop_max[2];
}
// screen rectangle (inclusive, clamped) of one body: its hull vertices projected, one pixel
// of slack for the collision margin.  Returns 0 when the body is not in front of the camera.
RV_DEV void project_vertex(const ObsSnap* c, v3 pw, float* u, float* v, float* z) {
  v3 pc = add(mulv(c->cam_rotation, pw), ld3(c->cam_translation));
  const float fx = c->cam_intrinsics[0], fy = c->cam_intrinsics[1], cx = c->cam_intrinsics[2], cy = c->cam_intrinsics[3], sk = c->cam_intrinsics[4];
  *z = pc.z;
  *u = (fx * pc.x + sk * pc.y) / pc.z + cx;
  *v = fy * pc.y / pc.z + cy;
}
// sampling keys / draws: one Philox block per item, first word
RV_DEV uint32_t pc_hash(const rv_config* c, uint32_t gid, uint32_t rng_arg, uint32_t ctr) {
  uint32_t o0, o1, o2, o3;
  philox(ctr, rng_arg, gid, RV_STREAM_PC, c->seed_lo, c->seed_hi, &o0, &o1, &o2, &o3);
  return o0;
}
#define RV_PC_KEY_CTR(b, i)  (0x80000000u | ((uint32_t)(b) << 16) | (uint32_t)(i))
#define RV_PC_DRAW_CTR(b, j) (((uint32_t)(b) << 16) | (uint32_t)(j))

}  // namespace rv

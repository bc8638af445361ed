/*
 * orc_math.h — scalar math for the CPU oracle (TEST INFRASTRUCTURE ONLY).
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * use anything under oracle/.  The product (robovat_amd/csrc) never includes
 * this file.
 *
 * Conventions restated from the reference:
 *   - quaternions are xyzw         (third_party/transformations.py:1142-1180)
 *   - Euler angles are static xyz  (third_party/transformations.py:1034-1081,
 *                                   robovat/math/orientation.py:71-104)
 * The float build avoids libm transcendentals (own sincos/atan2 below) so
 * that the HIP kernels can reproduce it operation for operation.
 */
#ifndef ORC_MATH_H_
#define ORC_MATH_H_

#include <math.h>
#include <stdint.h>
#include <string.h>

#ifdef ORC_DOUBLE
typedef double real;
#define R(x) x
#define rsqrt_(x) sqrt(x)
#define rabs(x) fabs(x)
#define rrint(x) rint(x)
#define rfma(a, b, c) __builtin_fma(a, b, c)     /* ONE rounding (the impulse-space solver's row update) */
#else
typedef float real;
#define R(x) x##f
#define rsqrt_(x) sqrtf(x)
#define rabs(x) fabsf(x)
#define rrint(x) rintf(x)
#define rfma(a, b, c) __builtin_fmaf(a, b, c)    /* = v_fma_f32 on the device */
#endif

#define ORC_PI R(3.14159265358979323846)

static inline real rmin(real a, real b) { return a < b ? a : b; }
static inline real rmax(real a, real b) { return a > b ? a : b; }
static inline real rclamp(real x, real lo, real hi) { return x < lo ? lo : (x > hi ? hi : x); }

static inline void v3set(real* o, real x, real y, real z) { o[0] = x; o[1] = y; o[2] = z; }
static inline void v3cpy(real* o, const real* a) { o[0] = a[0]; o[1] = a[1]; o[2] = a[2]; }
static inline void v3add(real* o, const real* a, const real* b) { o[0] = a[0] + b[0]; o[1] = a[1] + b[1]; o[2] = a[2] + b[2]; }
static inline void v3sub(real* o, const real* a, const real* b) { o[0] = a[0] - b[0]; o[1] = a[1] - b[1]; o[2] = a[2] - b[2]; }
static inline void v3scale(real* o, const real* a, real s) { o[0] = a[0] * s; o[1] = a[1] * s; o[2] = a[2] * s; }
/* o = a + b*s */
static inline void v3madd(real* o, const real* a, const real* b, real s) { o[0] = a[0] + b[0] * s; o[1] = a[1] + b[1] * s; o[2] = a[2] + b[2] * s; }
static inline real v3dot(const real* a, const real* b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }
static inline void v3cross(real* o, const real* a, const real* b) {
  real x = a[1] * b[2] - a[2] * b[1];
  real y = a[2] * b[0] - a[0] * b[2];
  real z = a[0] * b[1] - a[1] * b[0];
  o[0] = x; o[1] = y; o[2] = z;
}
static inline real v3len(const real* a) { return rsqrt_(v3dot(a, a)); }

/* own sincos / atan2 (cephes-style polynomials) for the float build */
static inline void rsincos(real x, real* s, real* c) {
#ifdef ORC_DOUBLE
  *s = sin(x); *c = cos(x);
#else
  float k = rintf(x * 0.636619772367581343f);
  float r = ((x - k * 1.5703125f) - k * 4.837512969970703125e-4f) - k * 7.54978995489188216e-8f;
  float z = r * r;
  float sp = r + r * z * (-1.6666654611e-1f + z * (8.3321608736e-3f + z * -1.9515295891e-4f));
  float cp = 1.0f - 0.5f * z + z * z * (4.166664568298827e-2f + z * (-1.388731625493765e-3f + z * 2.443315711809948e-5f));
  int q = ((int)k) & 3;
  float ss = (q & 1) ? cp : sp;
  float cc = (q & 1) ? sp : cp;
  if (q == 1 || q == 2) cc = -cc;
  if (q >= 2) ss = -ss;
  *s = ss; *c = cc;
#endif
}

static inline real ratan_pos(real x) { /* x >= 0 */
#ifdef ORC_DOUBLE
  return atan(x);
#else
  float y0 = 0.0f;
  if (x > 2.414213562373095f) { y0 = 1.5707963267948966f; x = -1.0f / x; }
  else if (x > 0.4142135623730950f) { y0 = 0.7853981633974483f; x = (x - 1.0f) / (x + 1.0f); }
  float z = x * x;
  float y = (((8.05374449538e-2f * z - 1.38776856032e-1f) * z + 1.99777106478e-1f) * z - 3.33329491539e-1f) * z * x + x;
  return y0 + y;
#endif
}
static inline real ratan2(real y, real x) {
#ifdef ORC_DOUBLE
  return atan2(y, x);
#else
  if (x == 0.0f) {
    if (y > 0.0f) return 1.5707963267948966f;
    if (y < 0.0f) return -1.5707963267948966f;
    return 0.0f;
  }
  float a = ratan_pos(fabsf(y / x));
  if (x < 0.0f) a = 3.14159265358979323846f - a;
  return y < 0.0f ? -a : a;
#endif
}

/* ---- quaternions, xyzw ---- */
static inline void qmul(real* o, const real* a, const real* b) {
  real x = a[3] * b[0] + a[0] * b[3] + a[1] * b[2] - a[2] * b[1];
  real y = a[3] * b[1] - a[0] * b[2] + a[1] * b[3] + a[2] * b[0];
  real z = a[3] * b[2] + a[0] * b[1] - a[1] * b[0] + a[2] * b[3];
  real w = a[3] * b[3] - a[0] * b[0] - a[1] * b[1] - a[2] * b[2];
  o[0] = x; o[1] = y; o[2] = z; o[3] = w;
}
static inline void qnormalize(real* q) {
  real n = rsqrt_(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
  real inv = R(1.0) / n;
  q[0] *= inv; q[1] *= inv; q[2] *= inv; q[3] *= inv;
}
/* rotate v by unit quaternion q: v + w t + u x t with t = 2 (u x v) */
static inline void qrotv(real* o, const real* q, const real* v) {
  real t[3], c[3];
  v3cross(t, q, v); v3scale(t, t, R(2.0));
  v3cross(c, q, t);
  o[0] = v[0] + q[3] * t[0] + c[0]; o[1] = v[1] + q[3] * t[1] + c[1]; o[2] = v[2] + q[3] * t[2] + c[2];
}
/* rotation matrix (row major) of a unit quaternion */
static inline void qmat(real* m, const real* q) {
  real x = q[0], y = q[1], z = q[2], w = q[3];
  real xx = x * x, yy = y * y, zz = z * z;
  real xy = x * y, xz = x * z, yz = y * z, wx = w * x, wy = w * y, wz = w * z;
  m[0] = R(1.0) - R(2.0) * (yy + zz); m[1] = R(2.0) * (xy - wz);          m[2] = R(2.0) * (xz + wy);
  m[3] = R(2.0) * (xy + wz);          m[4] = R(1.0) - R(2.0) * (xx + zz); m[5] = R(2.0) * (yz - wx);
  m[6] = R(2.0) * (xz - wy);          m[7] = R(2.0) * (yz + wx);          m[8] = R(1.0) - R(2.0) * (xx + yy);
}
static inline void m3mulv(real* o, const real* m, const real* v) {
  real x = m[0] * v[0] + m[1] * v[1] + m[2] * v[2];
  real y = m[3] * v[0] + m[4] * v[1] + m[5] * v[2];
  real z = m[6] * v[0] + m[7] * v[1] + m[8] * v[2];
  o[0] = x; o[1] = y; o[2] = z;
}
static inline void m3tmulv(real* o, const real* m, const real* v) {
  real x = m[0] * v[0] + m[3] * v[1] + m[6] * v[2];
  real y = m[1] * v[0] + m[4] * v[1] + m[7] * v[2];
  real z = m[2] * v[0] + m[5] * v[1] + m[8] * v[2];
  o[0] = x; o[1] = y; o[2] = z;
}
/* euler (static xyz: roll, pitch, yaw) -> quaternion xyzw
 * (transformations.py quaternion_from_euler, axes='sxyz') */
static inline void euler_to_quat(real* q, real roll, real pitch, real yaw) {
  real si, ci, sj, cj, sk, ck;
  rsincos(roll * R(0.5), &si, &ci);
  rsincos(pitch * R(0.5), &sj, &cj);
  rsincos(yaw * R(0.5), &sk, &ck);
  q[0] = si * cj * ck - ci * sj * sk;
  q[1] = ci * sj * ck + si * cj * sk;
  q[2] = ci * cj * sk - si * sj * ck;
  q[3] = ci * cj * ck + si * sj * sk;
}
/* yaw (euler[2], static xyz) of a unit quaternion */
static inline real quat_yaw(const real* q) {
  real x = q[0], y = q[1], z = q[2], w = q[3];
  return ratan2(R(2.0) * (w * z + x * y), R(1.0) - R(2.0) * (y * y + z * z));
}

/* static-xyz Euler angles of a unit quaternion (transformations.py:1034-1085
 * euler_from_matrix, axes 'sxyz', on the matrix of q): roll, pitch, yaw */
static inline void quat_to_euler(const real* q, real* e) {
  real m[9]; qmat(m, q);
  real cy = rsqrt_(m[0] * m[0] + m[3] * m[3]);
  if (cy > R(1e-6)) {
    e[0] = ratan2(m[7], m[8]);
    e[1] = ratan2(-m[6], cy);
    e[2] = ratan2(m[3], m[0]);
  } else {
    e[0] = ratan2(-m[5], m[4]);
    e[1] = ratan2(-m[6], cy);
    e[2] = R(0.0);
  }
}

/* ---- Philox4x32-10 counter-based RNG (results invariant to GPU count) ---- */
typedef struct { uint32_t key[2]; uint32_t ctr[4]; uint32_t buf[4]; int idx; } orc_rng;

static inline void philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]) {
  uint32_t c0 = ctr[0], c1 = ctr[1], c2 = ctr[2], c3 = ctr[3];
  uint32_t k0 = key[0], k1 = key[1];
  for (int r = 0; r < 10; ++r) {
    uint64_t p0 = (uint64_t)0xD2511F53u * c0;
    uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
    uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
    uint32_t n1 = (uint32_t)p1;
    uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
    uint32_t n3 = (uint32_t)p0;
    c0 = n0; c1 = n1; c2 = n2; c3 = n3;
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
  out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}
static inline void rng_init(orc_rng* g, uint32_t seed_lo, uint32_t seed_hi, uint32_t gid, uint32_t stream, uint32_t arg) {
  g->key[0] = seed_lo; g->key[1] = seed_hi;
  g->ctr[0] = 0; g->ctr[1] = arg; g->ctr[2] = gid; g->ctr[3] = stream;
  g->idx = 4;
}
static inline uint32_t rng_u32(orc_rng* g) {
  if (g->idx == 4) { philox4x32_10(g->ctr, g->key, g->buf); g->ctr[0] += 1; g->idx = 0; }
  return g->buf[g->idx++];
}
static inline real rng_uniform01(orc_rng* g) { return (real)(rng_u32(g) >> 8) * R(5.9604644775390625e-8); }
static inline real rng_uniform(orc_rng* g, real lo, real hi) { return lo + (hi - lo) * rng_uniform01(g); }
static inline int rng_randint(orc_rng* g, int n) { return (int)(rng_u32(g) % (uint32_t)n); }

#endif /* ORC_MATH_H_ */

// rv_dev_math.h — scalar float math used inside the CDNA4 kernels.
//
// Everything here is a plain inline function so that the same source compiles
// as HIP device code (the product) and, with -DRV_EMULATE, as host C++ for the
// lane-emulation harness under tests/emu (a debugging aid for the kernel
// logic, never a product path).  No libm transcendentals: sincos / atan2 are
// short polynomials so that results do not depend on the math library.
#pragma once
#include <stdint.h>

// RV_ON_DEVICE -- defined HERE and nowhere else: 1 when the source is compiled by hipcc for the product, 0 when it is
// compiled for the host by the lane emulator of tests/emu (-DRV_EMULATE, or any plain C++ compiler).  The kernel headers
// test only this macro; the host bodies of the larger emulation hooks live in tests/emu/rv_emu_hooks.h.
#if defined(__HIPCC__) && !defined(RV_EMULATE)
#define RV_ON_DEVICE 1
#else
#define RV_ON_DEVICE 0
#endif
#if RV_ON_DEVICE
#include <hip/hip_runtime.h>
#define RV_DEV __device__ __forceinline__
#define RV_DEV_NOINLINE __device__ __noinline__
#else
#include <math.h>
#define RV_DEV static inline
#define RV_DEV_NOINLINE static
#endif

namespace rv {

#ifdef RV_EMU_COUNT
static long rv_emu_cnt[48];      // ad-hoc event counters of the host emulation (tools only)
static long rv_emu_dbg[16], rv_emu_dbg2[48];
#define RV_CNT(i, n) rv_emu_cnt[i] += (n);
#else
#define RV_CNT(i, n)
#endif

#define RV_PI 3.14159265358979323846f

RV_DEV float fminr(float a, float b) { return a < b ? a : b; }
RV_DEV float fmaxr(float a, float b) { return a > b ? a : b; }
RV_DEV float fclampr(float x, float lo, float hi) { return x < lo ? lo : (x > hi ? hi : x); }
// |x|: the sign bit cleared (fabsf of the oracle) -- a source modifier on the device, not an instruction
RV_DEV float fabsr(float x) { return __builtin_fabsf(x); }
// clamp to [-b, b] for a bound b > 0: v_med3_f32 on the device, one instruction instead of two compare - select pairs;
// equal to fclampr(x, -b, b) for every non-NaN x (a zero of either sign included: -b < x < b returns x itself)
RV_DEV float fclamp_pm(float x, float b) {
#if RV_ON_DEVICE
  return __builtin_amdgcn_fmed3f(x, -b, b);
#else
  return fclampr(x, -b, b);
#endif
}
// the ONE explicit fused multiply-add of the build (the row update of the impulse-space solvers): a single rounding,
// v_fma_f32 on the device, fmaf on the host emulator and in the oracle (rfma); everything else is -ffp-contract=off
RV_DEV float rv_fma(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
// sqrtf() lowers to the correctly rounded v_sqrt_f32 + fma fix-up sequence on
// gfx950 (ROCm 7.2); __fsqrt_rn() lowers to the bare ~1 ulp v_sqrt_f32 and
// breaks bit-parity with the CPU oracle.
RV_DEV float fsqrtr(float x) { return sqrtf(x); }
RV_DEV float frintr(float x) { return rintf(x); }
RV_DEV float ffloorr(float x) { return floorf(x); }

struct v3 { float x, y, z; };
RV_DEV v3 mk(float x, float y, float z) { v3 r; r.x = x; r.y = y; r.z = z; return r; }
RV_DEV v3 ld3(const float* p) { return mk(p[0], p[1], p[2]); }
RV_DEV void st3(float* p, v3 a) { p[0] = a.x; p[1] = a.y; p[2] = a.z; }
RV_DEV v3 add(v3 a, v3 b) { return mk(a.x + b.x, a.y + b.y, a.z + b.z); }
RV_DEV v3 sub(v3 a, v3 b) { return mk(a.x - b.x, a.y - b.y, a.z - b.z); }
RV_DEV v3 scale(v3 a, float s) { return mk(a.x * s, a.y * s, a.z * s); }
RV_DEV v3 madd(v3 a, v3 b, float s) { return mk(a.x + b.x * s, a.y + b.y * s, a.z + b.z * s); }
RV_DEV float dot(v3 a, v3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
RV_DEV v3 cross(v3 a, v3 b) {
  return mk(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
RV_DEV float len(v3 a) { return fsqrtr(dot(a, a)); }

RV_DEV void sincosr(float x, float* s, float* c) {
  float k = frintr(x * 0.636619772367581343f);
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
}
RV_DEV float atan_pos(float x) {
  float y0 = 0.0f;
  if (x > 2.414213562373095f) { y0 = 1.5707963267948966f; x = -1.0f / x; }
  else if (x > 0.4142135623730950f) { y0 = 0.7853981633974483f; x = (x - 1.0f) / (x + 1.0f); }
  float z = x * x;
  float y = (((8.05374449538e-2f * z - 1.38776856032e-1f) * z + 1.99777106478e-1f) * z - 3.33329491539e-1f) * z * x + x;
  return y0 + y;
}
RV_DEV float atan2r(float y, float x) {
  if (x == 0.0f) {
    if (y > 0.0f) return 1.5707963267948966f;
    if (y < 0.0f) return -1.5707963267948966f;
    return 0.0f;
  }
  float a = atan_pos(fabsr(y / x));
  if (x < 0.0f) a = RV_PI - a;
  return y < 0.0f ? -a : a;
}

// ---- quaternions (xyzw) and 3x3 matrices (row major) ----
struct q4 { float x, y, z, w; };
RV_DEV q4 ldq(const float* p) { q4 q; q.x = p[0]; q.y = p[1]; q.z = p[2]; q.w = p[3]; return q; }
RV_DEV void stq(float* p, q4 q) { p[0] = q.x; p[1] = q.y; p[2] = q.z; p[3] = q.w; }
RV_DEV q4 qmul(q4 a, q4 b) {
  q4 o;
  o.x = a.w * b.x + a.x * b.w + a.y * b.z - a.z * b.y;
  o.y = a.w * b.y - a.x * b.z + a.y * b.w + a.z * b.x;
  o.z = a.w * b.z + a.x * b.y - a.y * b.x + a.z * b.w;
  o.w = a.w * b.w - a.x * b.x - a.y * b.y - a.z * b.z;
  return o;
}
RV_DEV q4 qnormalize(q4 q) {
  float n = fsqrtr(q.x * q.x + q.y * q.y + q.z * q.z + q.w * q.w);
  float inv = 1.0f / n;
  q.x *= inv; q.y *= inv; q.z *= inv; q.w *= inv;
  return q;
}
// rotate v by unit quaternion q: v + w t + u x t with t = 2 (u x v)
RV_DEV v3 qrotv(q4 q, v3 v) {
  v3 u = mk(q.x, q.y, q.z);
  v3 t = scale(cross(u, v), 2.0f);
  v3 c = cross(u, t);
  return mk(v.x + q.w * t.x + c.x, v.y + q.w * t.y + c.y, v.z + q.w * t.z + c.z);
}
// third column of qmat(q): the joint axis z in the world frame (same expressions as qmat)
RV_DEV v3 qaxis_z(q4 q) {
  float x = q.x, y = q.y, z = q.z, w = q.w;
  float xx = x * x, yy = y * y;
  float xz = x * z, yz = y * z, wx = w * x, wy = w * y;
  return mk(2.0f * (xz + wy), 2.0f * (yz - wx), 1.0f - 2.0f * (xx + yy));
}
struct m3 { float m[9]; };
RV_DEV m3 qmat(q4 q) {
  float x = q.x, y = q.y, z = q.z, w = q.w;
  float xx = x * x, yy = y * y, zz = z * z;
  float xy = x * y, xz = x * z, yz = y * z, wx = w * x, wy = w * y, wz = w * z;
  m3 r;
  r.m[0] = 1.0f - 2.0f * (yy + zz); r.m[1] = 2.0f * (xy - wz);          r.m[2] = 2.0f * (xz + wy);
  r.m[3] = 2.0f * (xy + wz);          r.m[4] = 1.0f - 2.0f * (xx + zz); r.m[5] = 2.0f * (yz - wx);
  r.m[6] = 2.0f * (xz - wy);          r.m[7] = 2.0f * (yz + wx);          r.m[8] = 1.0f - 2.0f * (xx + yy);
  return r;
}
RV_DEV m3 ldm(const float* p) { m3 r; for (int i = 0; i < 9; ++i) r.m[i] = p[i]; return r; }
RV_DEV void stm(float* p, const m3& a) { for (int i = 0; i < 9; ++i) p[i] = a.m[i]; }
RV_DEV v3 mulv(const m3& a, v3 v) {
  return mk(a.m[0] * v.x + a.m[1] * v.y + a.m[2] * v.z,
            a.m[3] * v.x + a.m[4] * v.y + a.m[5] * v.z,
            a.m[6] * v.x + a.m[7] * v.y + a.m[8] * v.z);
}
RV_DEV v3 tmulv(const m3& a, v3 v) {
  return mk(a.m[0] * v.x + a.m[3] * v.y + a.m[6] * v.z,
            a.m[1] * v.x + a.m[4] * v.y + a.m[7] * v.z,
            a.m[2] * v.x + a.m[5] * v.y + a.m[8] * v.z);
}
// mulv / tmulv straight from memory (LDS or global)
RV_DEV v3 mulv(const float* m, v3 v) {
  return mk(m[0] * v.x + m[1] * v.y + m[2] * v.z,
            m[3] * v.x + m[4] * v.y + m[5] * v.z,
            m[6] * v.x + m[7] * v.y + m[8] * v.z);
}
RV_DEV v3 tmulv(const float* m, v3 v) {
  return mk(m[0] * v.x + m[3] * v.y + m[6] * v.z,
            m[1] * v.x + m[4] * v.y + m[7] * v.z,
            m[2] * v.x + m[5] * v.y + m[8] * v.z);
}
RV_DEV q4 euler_to_quat(float roll, float pitch, float yaw) {
  float si, ci, sj, cj, sk, ck;
  sincosr(roll * 0.5f, &si, &ci);
  sincosr(pitch * 0.5f, &sj, &cj);
  sincosr(yaw * 0.5f, &sk, &ck);
  q4 q;
  q.x = si * cj * ck - ci * sj * sk;
  q.y = ci * sj * ck + si * cj * sk;
  q.z = ci * cj * sk - si * sj * ck;
  q.w = ci * cj * ck + si * sj * sk;
  return q;
}
RV_DEV float quat_yaw(q4 q) {
  return atan2r(2.0f * (q.w * q.z + q.x * q.y), 1.0f - 2.0f * (q.y * q.y + q.z * q.z));
}

// static-xyz Euler angles of a unit quaternion (transformations.py euler_from_matrix,
// axes 'sxyz', on the matrix of q): roll, pitch, yaw
RV_DEV void quat_to_euler(q4 q, float* e) {
  m3 m = qmat(q);
  float cy = fsqrtr(m.m[0] * m.m[0] + m.m[3] * m.m[3]);
  if (cy > 1e-6f) {
    e[0] = atan2r(m.m[7], m.m[8]);
    e[1] = atan2r(-m.m[6], cy);
    e[2] = atan2r(m.m[3], m.m[0]);
  } else {
    e[0] = atan2r(-m.m[5], m.m[4]);
    e[1] = atan2r(-m.m[6], cy);
    e[2] = 0.0f;
  }
}

// ---- Philox4x32-10 ----
struct Rng { uint32_t key0, key1; uint32_t c0, c1, c2, c3; uint32_t b0, b1, b2, b3; int idx; };
RV_DEV void philox(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1,
                   uint32_t* o0, uint32_t* o1, uint32_t* o2, uint32_t* o3) {
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
  *o0 = c0; *o1 = c1; *o2 = c2; *o3 = c3;
}
RV_DEV Rng rng_init(uint32_t seed_lo, uint32_t seed_hi, uint32_t gid, uint32_t stream, uint32_t arg) {
  Rng g; g.key0 = seed_lo; g.key1 = seed_hi; g.c0 = 0; g.c1 = arg; g.c2 = gid; g.c3 = stream;
  g.b0 = g.b1 = g.b2 = g.b3 = 0; g.idx = 4;
  return g;
}
RV_DEV uint32_t rng_u32(Rng& g) {
  if (g.idx == 4) { philox(g.c0, g.c1, g.c2, g.c3, g.key0, g.key1, &g.b0, &g.b1, &g.b2, &g.b3); g.c0 += 1; g.idx = 0; }
  uint32_t r = g.idx == 0 ? g.b0 : (g.idx == 1 ? g.b1 : (g.idx == 2 ? g.b2 : g.b3));
  g.idx++;
  return r;
}
RV_DEV float rng_uniform01(Rng& g) { return (float)(rng_u32(g) >> 8) * 5.9604644775390625e-8f; }
RV_DEV float rng_uniform(Rng& g, float lo, float hi) { return lo + (hi - lo) * rng_uniform01(g); }
RV_DEV int rng_randint(Rng& g, int n) { return (int)(rng_u32(g) % (uint32_t)n); }

}  // namespace rv

// rv_dev_collide.h — GJK / EPA narrow phase and persistent 4-point contact
// manifolds for the CDNA4 env kernel.
//
// A 16-lane group owns one manifold slot at a time and runs the convex queries
// of its pair(s): one scalar program executed redundantly by the group, except
// in support_v() where every lane holds one hull vertex and DPP all-reduces pick
// the extreme one.  The simplex lives in registers (all array indices are
// compile-time after unrolling), hull vertices are read from the env's LDS
// block, the rarely needed EPA polytope lives in per-lane private memory.
#pragma once
#include "rv_dev_math.h"

namespace rv {

#define RV_GJK_MAX_ITERS 32
#define RV_GJK_REL_TOL 1e-4f
#define RV_GJK_PROGRESS_TOL 1e-6f
#define RV_EPA_MAX_VERTS 24
#define RV_EPA_MAX_FACES 48
#define RV_EPA_MAX_EDGES 32
#define RV_EPA_MAX_ITERS 32
#define RV_EPA_TOL 1e-6f
#define RV_MAXV_REG 16

struct Simplex {
  v3 w[4], a[4], b[4];
  float lam[4];
  int n;
  int ia[4], ib[4];   // vertex indices of a[], b[] (for the simplex cache)
};
// the closest feature found by the last convex query of a manifold: the next query of the SAME pair of
// hulls starts from it (temporal coherence; Bullet keeps a cached separating axis per pair)
struct GjkCache { int n, pair; int ia[3], ib[3]; };

// EPA polytope workspace (per-lane private memory)
struct EpaWork {
  float W[RV_EPA_MAX_VERTS][3], VA[RV_EPA_MAX_VERTS][3], VB[RV_EPA_MAX_VERTS][3];
  int fi[RV_EPA_MAX_FACES][3];
  float fn[RV_EPA_MAX_FACES][3];
  float fd[RV_EPA_MAX_FACES];
  int alive[RV_EPA_MAX_FACES];
  int ea[RV_EPA_MAX_EDGES], eb[RV_EPA_MAX_EDGES];
};

// persistent manifold (one per pair slot), 63 words
struct DevMan {
  int n;
  int col[4];
  float la[4][3], lb[4][3], nrm[4][3];
  float dist[4], ln[4], lt1[4], lt2[4];
  float acc;   // relative motion since the last full narrow phase
  int age;     // full passes since the last feature stage
  GjkCache gc;
};

// Support vertex (by value) of a hull stored as n <= 16 packed xyz triples.
//
// Device: the 16 lanes of a group each hold one vertex (index clamped to n-1),
// take the dot product and run two 4-step DPP row-rotate all-reduces: the largest
// projection, then the lowest lane index attaining it -- exactly what the serial
// first-maximum loop returns (a clamped duplicate can never beat its original).
// Every lane of the group ends up with the same result.
// Host emulation: the serial loop.
#if RV_ON_DEVICE
template <int N> RV_DEV float row_ror_f(float x) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x120 + N, 0xf, 0xf, false));
}
template <int N> RV_DEV int row_ror_i(int x) { return __builtin_amdgcn_update_dpp(0, x, 0x120 + N, 0xf, 0xf, false); }
// two-stage reduction: the largest projection (4 DPP max steps), then the lowest lane
// index that attains it (4 DPP min steps) -- exactly the serial first-maximum -- and one
// broadcast LDS read of the winning vertex
// (v_max_f32 on the rotated value: equal to the ternary for every finite input up to the sign of a zero,
// which neither the index search `val == m` nor the uses of the projection can see)
// (ONE instruction, v_max_f32_dpp, behind the two wait states a DPP read of a just-written VGPR needs.  Through
// __builtin_fmaxf(update_dpp(0, x), x) a step was five issue slots: the move of the `old' operand, the wait state, the DPP
// move, a v_max x, x that quiets a possible signalling NaN of the moved value, and the maximum itself)
template <int N> RV_DEV float row_ror_fmax(float x) {
  float r;
  if (N == 8) asm("s_nop 1\n\tv_max_f32_dpp %0, %1, %1 row_ror:8 row_mask:0xf bank_mask:0xf" : "=v"(r) : "v"(x));
  else if (N == 4) asm("s_nop 1\n\tv_max_f32_dpp %0, %1, %1 row_ror:4 row_mask:0xf bank_mask:0xf" : "=v"(r) : "v"(x));
  else if (N == 2) asm("s_nop 1\n\tv_max_f32_dpp %0, %1, %1 row_ror:2 row_mask:0xf bank_mask:0xf" : "=v"(r) : "v"(x));
  else asm("s_nop 1\n\tv_max_f32_dpp %0, %1, %1 row_ror:1 row_mask:0xf bank_mask:0xf" : "=v"(r) : "v"(x));
  return r;
}
template <int N> RV_DEV int row_ror_imin(int x) {
  int o = row_ror_i<N>(x);
  return o < x ? o : x;
}
RV_DEV v3 support_v(const float* verts, int n, v3 d, float* proj, int* idx_out = nullptr) {
  const int sl = (int)threadIdx.x & 15;
  const int j = sl < n ? sl : n - 1;
  const float val = dot(mk(verts[3 * j], verts[3 * j + 1], verts[3 * j + 2]), d);
  float m = val;
  m = row_ror_fmax<8>(m); m = row_ror_fmax<4>(m); m = row_ror_fmax<2>(m); m = row_ror_fmax<1>(m);
  int idx = (val == m) ? j : 16;
  idx = row_ror_imin<8>(idx); idx = row_ror_imin<4>(idx); idx = row_ror_imin<2>(idx); idx = row_ror_imin<1>(idx);
  *proj = m;
  if (idx_out) *idx_out = idx;
  return mk(verts[3 * idx], verts[3 * idx + 1], verts[3 * idx + 2]);
}
// largest projection only (no witness vertex)
RV_DEV float support_proj(const float* verts, int n, v3 d) {
  const int sl = (int)threadIdx.x & 15;
  const int j = sl < n ? sl : n - 1;
  float m = dot(mk(verts[3 * j], verts[3 * j + 1], verts[3 * j + 2]), d);
  m = row_ror_fmax<8>(m); m = row_ror_fmax<4>(m); m = row_ror_fmax<2>(m); m = row_ror_fmax<1>(m);
  return m;
}
#else
RV_DEV float support_proj(const float* verts, int n, v3 d) {
  float bd = dot(ld3(verts), d);
  for (int i = 1; i < n; ++i) {
    float x = dot(ld3(verts + 3 * i), d);
    if (x > bd) bd = x;
  }
  return bd;
}
RV_DEV v3 support_v(const float* verts, int n, v3 d, float* proj, int* idx_out = nullptr) {
  v3 best = ld3(verts);
  float bd = dot(best, d);
  int bi = 0;
  for (int i = 1; i < n; ++i) {
    v3 p = ld3(verts + 3 * i);
    float x = dot(p, d);
    if (x > bd) { bd = x; best = p; bi = i; }
  }
  *proj = bd;
  if (idx_out) *idx_out = bi;
  return best;
}
#endif

RV_DEV int support(const float* verts, int n, v3 d) {
  int best = 0;
  float bd = dot(ld3(verts), d);
  for (int i = 1; i < n; ++i) {
    float x = dot(ld3(verts + 3 * i), d);
    if (x > bd) { bd = x; best = i; }
  }
  return best;
}

RV_DEV void closest_segment(v3 p0, v3 p1, float* l0, float* l1) {
  v3 d = sub(p1, p0);
  float dd = dot(d, d);
  if (!(dd > 0.0f)) { *l0 = 0.0f; *l1 = 1.0f; return; }
  float t = -dot(p0, d) / dd;
  if (t <= 0.0f) { *l0 = 1.0f; *l1 = 0.0f; }
  else if (t >= 1.0f) { *l0 = 0.0f; *l1 = 1.0f; }
  else { *l0 = 1.0f - t; *l1 = t; }
}

RV_DEV void closest_triangle(v3 a, v3 b, v3 c, float* l0, float* l1, float* l2) {
  v3 ab = sub(b, a), ac = sub(c, a);
  float d1 = -dot(ab, a), d2 = -dot(ac, a);
  if (d1 <= 0.0f && d2 <= 0.0f) { *l0 = 1.0f; *l1 = 0.0f; *l2 = 0.0f; return; }
  float d3 = -dot(ab, b), d4 = -dot(ac, b);
  if (d3 >= 0.0f && d4 <= d3) { *l0 = 0.0f; *l1 = 1.0f; *l2 = 0.0f; return; }
  float vc = d1 * d4 - d3 * d2;
  if (vc <= 0.0f && d1 >= 0.0f && d3 <= 0.0f) {
    float v = d1 / (d1 - d3);
    *l0 = 1.0f - v; *l1 = v; *l2 = 0.0f; return;
  }
  float d5 = -dot(ab, c), d6 = -dot(ac, c);
  if (d6 >= 0.0f && d5 <= d6) { *l0 = 0.0f; *l1 = 0.0f; *l2 = 1.0f; return; }
  float vb = d5 * d2 - d1 * d6;
  if (vb <= 0.0f && d2 >= 0.0f && d6 <= 0.0f) {
    float w = d2 / (d2 - d6);
    *l0 = 1.0f - w; *l1 = 0.0f; *l2 = w; return;
  }
  float va = d3 * d6 - d5 * d4;
  if (va <= 0.0f && (d4 - d3) >= 0.0f && (d5 - d6) >= 0.0f) {
    float w = (d4 - d3) / ((d4 - d3) + (d5 - d6));
    *l0 = 0.0f; *l1 = 1.0f - w; *l2 = w; return;
  }
  float s = va + vb + vc;
  if (!(s > 0.0f)) {
    // degenerate (collinear) triangle: best of the three edges
    float best = -1.0f;
    float r0 = 0.0f, r1 = 0.0f, r2 = 1.0f;
    {
      float e0, e1; closest_segment(a, b, &e0, &e1);
      v3 p = mk(a.x * e0 + b.x * e1, a.y * e0 + b.y * e1, a.z * e0 + b.z * e1);
      float dd = dot(p, p);
      if (best < 0.0f || dd < best) { best = dd; r0 = e0; r1 = e1; r2 = 0.0f; }
    }
    {
      float e0, e1; closest_segment(b, c, &e0, &e1);
      v3 p = mk(b.x * e0 + c.x * e1, b.y * e0 + c.y * e1, b.z * e0 + c.z * e1);
      float dd = dot(p, p);
      if (best < 0.0f || dd < best) { best = dd; r0 = 0.0f; r1 = e0; r2 = e1; }
    }
    {
      float e0, e1; closest_segment(c, a, &e0, &e1);
      v3 p = mk(c.x * e0 + a.x * e1, c.y * e0 + a.y * e1, c.z * e0 + a.z * e1);
      float dd = dot(p, p);
      if (best < 0.0f || dd < best) { best = dd; r0 = e1; r1 = 0.0f; r2 = e0; }
    }
    *l0 = r0; *l1 = r1; *l2 = r2;
    return;
  }
  float denom = 1.0f / s;
  float v = vb * denom, w = vc * denom;
  *l0 = 1.0f - v - w; *l1 = v; *l2 = w;
}

// barycentric weights of the point of the simplex closest to the origin (zero for the vertices
// that do not support it); returns 1 when a tetrahedron encloses the origin.  s is not modified.
RV_DEV int simplex_weights(const Simplex& s, float* l) {
  l[0] = l[1] = l[2] = l[3] = 0.0f;
  if (s.n == 1) {
    l[0] = 1.0f;
  } else if (s.n == 2) {
    closest_segment(s.w[0], s.w[1], &l[0], &l[1]);
  } else if (s.n == 3) {
    closest_triangle(s.w[0], s.w[1], s.w[2], &l[0], &l[1], &l[2]);
  } else {
    const int F[4][4] = {{0, 1, 2, 3}, {0, 1, 3, 2}, {0, 2, 3, 1}, {1, 2, 3, 0}};
    int any_outside = 0;
    float best = -1.0f;
#pragma unroll
    for (int f = 0; f < 4; ++f) {
      v3 p = s.w[F[f][0]], q = s.w[F[f][1]], r = s.w[F[f][2]], o = s.w[F[f][3]];
      v3 nrm = cross(sub(q, p), sub(r, p));
      float sp = dot(nrm, sub(o, p));
      float so = -dot(nrm, p);
      if (!(sp * so > 0.0f)) {
        any_outside = 1;
        float f0, f1, f2;
        closest_triangle(p, q, r, &f0, &f1, &f2);
        v3 c = mk(p.x * f0 + q.x * f1 + r.x * f2, p.y * f0 + q.y * f1 + r.y * f2, p.z * f0 + q.z * f1 + r.z * f2);
        float dd = dot(c, c);
        if (best < 0.0f || dd < best) {
          best = dd;
          l[0] = l[1] = l[2] = l[3] = 0.0f;
          l[F[f][0]] = f0; l[F[f][1]] = f1; l[F[f][2]] = f2;
        }
      }
    }
    if (!any_outside) return 1;
  }
  return 0;
}
// the closest point those weights give: sum of l_i w_i over the supporting vertices, in order
RV_DEV v3 simplex_point(const Simplex& s, const float* l) {
  v3 v = mk(0.0f, 0.0f, 0.0f);
#pragma unroll
  for (int i = 0; i < 4; ++i) if (i < s.n && l[i] > 0.0f) v = madd(v, s.w[i], l[i]);
  return v;
}
// reduce the simplex to the vertices with positive weight (order preserved), store the weights
RV_DEV void simplex_commit(Simplex& s, const float* l) {
  int m = 0;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    if (i < s.n && l[i] > 0.0f) {
#pragma unroll
      for (int j = 0; j <= i; ++j) {
        if (j == m) { s.w[j] = s.w[i]; s.a[j] = s.a[i]; s.b[j] = s.b[i]; s.ia[j] = s.ia[i]; s.ib[j] = s.ib[i]; s.lam[j] = l[i]; }
      }
      ++m;
    }
  }
  s.n = m;
}
// returns 1 when a tetrahedron encloses the origin
RV_DEV int simplex_solve(Simplex& s, v3* vout) {
  float l[4];
  if (simplex_weights(s, l)) return 1;
  *vout = simplex_point(s, l);
  simplex_commit(s, l);
  return 0;
}

// EPA on an origin-enclosing tetrahedron; polytope in the private workspace E.
// (the four seed vertices come in a struct of their own: passing the GJK simplex by reference to
// this out-of-line function would pin it in private memory for the whole query)
struct EpaSeed { v3 w[4], a[4], b[4]; };
RV_DEV_NOINLINE void epa(const float* A, int nA, const float* B, int nB, const EpaSeed& s,
                         v3* out_nf, float* out_depth, v3* pa, v3* pb) {
  EpaWork E;  // private (scratch) memory: deep penetration is rare, LDS is not spent on it
  int nv = 4, nf = 0;
#pragma unroll
  for (int i = 0; i < 4; ++i) { st3(E.W[i], s.w[i]); st3(E.VA[i], s.a[i]); st3(E.VB[i], s.b[i]); }
  const int F[4][4] = {{0, 1, 2, 3}, {0, 1, 3, 2}, {0, 2, 3, 1}, {1, 2, 3, 0}};
  for (int f = 0; f < 4; ++f) {
    int i = F[f][0], j = F[f][1], k = F[f][2], o = F[f][3];
    v3 wi = ld3(E.W[i]);
    v3 n = cross(sub(ld3(E.W[j]), wi), sub(ld3(E.W[k]), wi));
    if (dot(n, sub(ld3(E.W[o]), wi)) > 0.0f) { int t = j; j = k; k = t; n = scale(n, -1.0f); }
    float ln = len(n);
    int degen = !(ln > 1e-12f);
    if (degen) ln = 1.0f;
    n = scale(n, 1.0f / ln);
    // a zero-area face has no normal: give it infinite distance so it is never selected
    E.fi[nf][0] = i; E.fi[nf][1] = j; E.fi[nf][2] = k; st3(E.fn[nf], n); E.fd[nf] = degen ? 1e30f : dot(n, wi); E.alive[nf] = 1; ++nf;
  }
  int bestf = 0;
  for (int it = 0; it < RV_EPA_MAX_ITERS; ++it) {
    bestf = -1;
    for (int f = 0; f < nf; ++f) if (E.alive[f] && (bestf < 0 || E.fd[f] < E.fd[bestf])) bestf = f;
    v3 fnb = ld3(E.fn[bestf]);
    int ia = support(A, nA, fnb);
    int ib = support(B, nB, scale(fnb, -1.0f));
    v3 va = ld3(A + 3 * ia), vb = ld3(B + 3 * ib);
    v3 w = sub(va, vb);
    if (dot(fnb, w) - E.fd[bestf] < RV_EPA_TOL || nv >= RV_EPA_MAX_VERTS) break;
    int en = 0;
    for (int f = 0; f < nf; ++f) {
      if (!E.alive[f]) continue;
      v3 d = sub(w, ld3(E.W[E.fi[f][0]]));
      if (dot(ld3(E.fn[f]), d) > 0.0f) {
        E.alive[f] = 0;
        for (int e = 0; e < 3; ++e) {
          int p = E.fi[f][e], q = E.fi[f][(e + 1) % 3];
          int found = -1;
          for (int x = 0; x < en; ++x) if (E.ea[x] == q && E.eb[x] == p) { found = x; break; }
          if (found >= 0) { E.ea[found] = E.ea[en - 1]; E.eb[found] = E.eb[en - 1]; --en; }
          else if (en < RV_EPA_MAX_EDGES) { E.ea[en] = p; E.eb[en] = q; ++en; }
        }
      }
    }
    if (en == 0) break;
    st3(E.W[nv], w); st3(E.VA[nv], va); st3(E.VB[nv], vb);
    int overflow = 0;
    for (int x = 0; x < en; ++x) {
      int slot = -1;
      for (int f = 0; f < nf; ++f) if (!E.alive[f]) { slot = f; break; }
      if (slot < 0) { if (nf < RV_EPA_MAX_FACES) slot = nf++; else { overflow = 1; break; } }
      int i = E.ea[x], j = E.eb[x], k = nv;
      v3 wi = ld3(E.W[i]);
      v3 n = cross(sub(ld3(E.W[j]), wi), sub(ld3(E.W[k]), wi));
      float ln = len(n);
      int degen = !(ln > 1e-12f);
      if (degen) ln = 1.0f;
      n = scale(n, 1.0f / ln);
      float d = dot(n, wi);
      if (d < 0.0f) { int t = i; i = j; j = t; n = scale(n, -1.0f); d = -d; }
      if (degen) d = 1e30f;
      E.fi[slot][0] = i; E.fi[slot][1] = j; E.fi[slot][2] = k; st3(E.fn[slot], n); E.fd[slot] = d; E.alive[slot] = 1;
    }
    ++nv;
    if (overflow) break;
  }
  bestf = -1;
  for (int f = 0; f < nf; ++f) if (E.alive[f] && (bestf < 0 || E.fd[f] < E.fd[bestf])) bestf = f;
  v3 fnb = ld3(E.fn[bestf]);
  v3 c = scale(fnb, E.fd[bestf]);
  int i0 = E.fi[bestf][0], i1 = E.fi[bestf][1], i2 = E.fi[bestf][2];
  float l0, l1, l2;
  closest_triangle(sub(ld3(E.W[i0]), c), sub(ld3(E.W[i1]), c), sub(ld3(E.W[i2]), c), &l0, &l1, &l2);
  *pa = mk(E.VA[i0][0] * l0 + E.VA[i1][0] * l1 + E.VA[i2][0] * l2,
           E.VA[i0][1] * l0 + E.VA[i1][1] * l1 + E.VA[i2][1] * l2,
           E.VA[i0][2] * l0 + E.VA[i1][2] * l1 + E.VA[i2][2] * l2);
  *pb = mk(E.VB[i0][0] * l0 + E.VB[i1][0] * l1 + E.VB[i2][0] * l2,
           E.VB[i0][1] * l0 + E.VB[i1][1] * l1 + E.VB[i2][1] * l2,
           E.VB[i0][2] * l0 + E.VB[i1][2] * l1 + E.VB[i2][2] * l2);
  // c lies on the facet of the difference body that CONTAINS the best triangle, not necessarily inside the
  // triangle: the clamped barycentric point then gives witnesses with pa - pb != c (anchors centimetres apart
  // tangentially, dropped as a broken point by the next refresh).  The a_i all lie on A's support face for
  // this normal, the b_i on B's: a feature that is a single vertex is that body's witness and the other is
  // its projection; otherwise the mismatch is split (oracle: orc_epa, same arithmetic)
  {
    const bool vtx_a = E.VA[i0][0] == E.VA[i1][0] && E.VA[i0][1] == E.VA[i1][1] && E.VA[i0][2] == E.VA[i1][2] &&
                       E.VA[i0][0] == E.VA[i2][0] && E.VA[i0][1] == E.VA[i2][1] && E.VA[i0][2] == E.VA[i2][2];
    const bool vtx_b = E.VB[i0][0] == E.VB[i1][0] && E.VB[i0][1] == E.VB[i1][1] && E.VB[i0][2] == E.VB[i1][2] &&
                       E.VB[i0][0] == E.VB[i2][0] && E.VB[i0][1] == E.VB[i2][1] && E.VB[i0][2] == E.VB[i2][2];
    v3 qa = *pa, qb = *pb;
    if (vtx_a) qb = sub(qa, c);
    else if (vtx_b) qa = add(qb, c);
    else {
      const v3 dl = sub(sub(qa, qb), c);
      qa = mk(qa.x - 0.5f * dl.x, qa.y - 0.5f * dl.y, qa.z - 0.5f * dl.z);
      qb = mk(qb.x + 0.5f * dl.x, qb.y + 0.5f * dl.y, qb.z + 0.5f * dl.z);
    }
    *pa = qa; *pb = qb;
  }
  *out_nf = fnb;
  *out_depth = E.fd[bestf];
}

// The closest point of the GJK simplex IS the origin but the simplex is a vertex, a segment or a triangle (two cores that
// overlap mirror-symmetrically about the origin of the difference body: crossed edges exactly centred, a face centred on a
// face): grow it into a tetrahedron of non-zero volume that has the origin inside or on its boundary, so that EPA can
// measure the overlap (orc_simplex_expand).  Returns 0 when the difference body is flat in every direction tried.
RV_DEV void diff_support(const float* A, int nA, const float* B, int nB, v3 d, v3* w, v3* a, v3* b) {
  float pj;
  *a = support_v(A, nA, d, &pj); *b = support_v(B, nB, scale(d, -1.0f), &pj); *w = sub(*a, *b);
}
RV_DEV int simplex_expand(const float* A, int nA, const float* B, int nB, Simplex& s) {
  if (s.n == 1) {
    int ok = 0;
    for (int k = 0; k < 6 && !ok; ++k) {
      const float sg = (k & 1) ? -1.0f : 1.0f;
      const v3 d = (k >> 1) == 0 ? mk(sg, 0.0f, 0.0f) : ((k >> 1) == 1 ? mk(0.0f, sg, 0.0f) : mk(0.0f, 0.0f, sg));
      v3 w, a, b;
      diff_support(A, nA, B, nB, d, &w, &a, &b);
      const v3 e = sub(w, s.w[0]);
      if (dot(e, e) > 1e-12f) { s.w[1] = w; s.a[1] = a; s.b[1] = b; s.n = 2; ok = 1; }
    }
    if (!ok) return 0;
  }
  if (s.n == 2) {
    const v3 d = sub(s.w[1], s.w[0]);
    const float dd = dot(d, d);
    int ax = 0;
    if (fabsr(d.y) < fabsr(d.x)) ax = 1;
    if (fabsr(d.z) < fabsr(ax == 0 ? d.x : d.y)) ax = 2;
    const v3 e0 = ax == 0 ? mk(1.0f, 0.0f, 0.0f) : (ax == 1 ? mk(0.0f, 1.0f, 0.0f) : mk(0.0f, 0.0f, 1.0f));
    const v3 u = cross(d, e0), v = cross(d, u);
    int ok = 0;
    for (int k = 0; k < 4 && !ok; ++k) {
      const v3 dir = scale(k < 2 ? u : v, (k & 1) ? -1.0f : 1.0f);
      v3 w, a, b;
      diff_support(A, nA, B, nB, dir, &w, &a, &b);
      const v3 e = sub(w, s.w[0]), cr = cross(e, d);
      if (dot(e, e) > 1e-12f && dot(cr, cr) > 1e-6f * dd * dot(e, e)) { s.w[2] = w; s.a[2] = a; s.b[2] = b; s.n = 3; ok = 1; }
    }
    if (!ok) return 0;
  }
  if (s.n == 3) {
    const v3 nrm = cross(sub(s.w[1], s.w[0]), sub(s.w[2], s.w[0]));
    const float nl = fsqrtr(dot(nrm, nrm));
    if (!(nl > 0.0f)) return 0;
    int ok = 0;
    for (int k = 0; k < 2 && !ok; ++k) {
      const v3 dir = scale(nrm, k ? -1.0f : 1.0f);
      v3 w, a, b;
      diff_support(A, nA, B, nB, dir, &w, &a, &b);
      const v3 e = sub(w, s.w[0]);
      if (dot(e, dir) > 1e-6f * nl) { s.w[3] = w; s.a[3] = a; s.b[3] = b; s.n = 4; ok = 1; }
    }
    if (!ok) return 0;
  }
  return s.n == 4;
}

// GJK distance with EPA fallback.  Returns 0 when farther apart than max_dist.
// lb_out (optional): a rigorous lower bound of the distance between the two hulls
// (the largest separating-plane bound seen), valid also when the query misses.
RV_DEV int gjk_epa(const float* A, int nA, const float* B, int nB, v3 guess, float max_dist,
                   v3* n, float* dist, v3* pa, v3* pb, float* lb_out = nullptr, GjkCache* gc = nullptr, int pair = 0) {
  float lb = 0.0f;
  Simplex s; s.n = 0;
#pragma unroll
  for (int i = 0; i < 4; ++i) { s.w[i] = mk(0, 0, 0); s.a[i] = mk(0, 0, 0); s.b[i] = mk(0, 0, 0); s.lam[i] = 0.0f; s.ia[i] = 0; s.ib[i] = 0; }
  v3 v = guess;
  if (!(dot(v, v) > 1e-12f)) v = mk(1.0f, 0.0f, 0.0f);
  int have_v = 0, penetrating = 0;
  if (gc) {
    // start from the closest feature of the last query of this pair, if there is one
    const int cn = gc->n, cp = gc->pair;
    int cia[3], cib[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) { cia[i] = gc->ia[i]; cib[i] = gc->ib[i]; }
    if (cn > 0 && cp == pair) {
      int ok = 1;
#pragma unroll
      for (int i = 0; i < 3; ++i) if (i < cn && (cia[i] >= nA || cib[i] >= nB)) ok = 0;
      if (ok) {
#pragma unroll
        for (int i = 0; i < 3; ++i) if (i < cn) {
          s.a[i] = mk(A[3 * cia[i]], A[3 * cia[i] + 1], A[3 * cia[i] + 2]);
          s.b[i] = mk(B[3 * cib[i]], B[3 * cib[i] + 1], B[3 * cib[i] + 2]);
          s.w[i] = sub(s.a[i], s.b[i]); s.ia[i] = cia[i]; s.ib[i] = cib[i];
        }
        s.n = cn;
        v3 v0;
        if (!simplex_solve(s, &v0) && dot(v0, v0) > 1e-14f) { v = v0; have_v = 1; }
        else s.n = 0;
      }
    }
    gc->n = 0;
  }
  RV_CNT(18, 1)
  for (int it = 0; it < RV_GJK_MAX_ITERS; ++it) {
    RV_CNT(19, 1)
    float pja, pjb;
    int ia_ = 0, ib_ = 0;
    v3 va = support_v(A, nA, scale(v, -1.0f), &pja, &ia_), vb = support_v(B, nB, v, &pjb, &ib_);
    v3 w = sub(va, vb);
    float vv = dot(v, v), vw = dot(v, w);
    if (lb_out && vw > 0.0f) { float l = vw / fsqrtr(vv); if (l > lb) lb = l; *lb_out = lb; }
    if (vw > 0.0f && vw * vw > max_dist * max_dist * vv) return 0;
    int dup = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k)
      if (k < s.n && s.w[k].x == w.x && s.w[k].y == w.y && s.w[k].z == w.z) dup = 1;
    if (dup) break;
    if (have_v && vv - vw <= RV_GJK_REL_TOL * vv) break;
#pragma unroll
    for (int k = 0; k < 4; ++k) if (k == s.n) { s.w[k] = w; s.a[k] = va; s.b[k] = vb; s.ia[k] = ia_; s.ib[k] = ib_; }
    s.n++;
    float l[4];
    if (simplex_weights(s, l)) { penetrating = 1; break; }
    const v3 vc = simplex_point(s, l);
    float vn = dot(vc, vc);
    if (!(vn > 1e-14f)) { simplex_commit(s, l); v = vc; penetrating = 2; break; }
    // no progress: the new support point does not bring the closest point closer (face-face
    // contacts would otherwise cycle through the vertices of the touching faces).  The new point
    // is DROPPED and the query ends on the simplex it had: with four nearly coplanar points (a
    // cached face feature plus one more vertex of the same face) the sub-simplex chosen in FP32
    // can be farther from the origin than the one before, with a normal tilted by 20 degrees
    if (have_v && vv - vn <= RV_GJK_PROGRESS_TOL * vv) { s.n--; break; }
    simplex_commit(s, l); v = vc;
    have_v = 1;
  }
  v3 ta = mk(0, 0, 0), tb = mk(0, 0, 0);
  if (penetrating == 2) {
    // the origin lies ON a vertex / segment / triangle of the simplex: the witnesses of "touching" first, then (round 5)
    // the simplex is grown into a tetrahedron and EPA decides whether the cores merely touch or overlap
#pragma unroll
    for (int i = 0; i < 4; ++i) if (i < s.n) { ta = madd(ta, s.a[i], s.lam[i]); tb = madd(tb, s.b[i], s.lam[i]); }
    if (simplex_expand(A, nA, B, nB, s)) penetrating = 3;
  }
  if (penetrating == 1 || penetrating == 3) {
    v3 nf; float depth;
    RV_CNT(20, 1)
    EpaSeed seed;
#pragma unroll
    for (int i = 0; i < 4; ++i) { seed.w[i] = s.w[i]; seed.a[i] = s.a[i]; seed.b[i] = s.b[i]; }
    epa(A, nA, B, nB, seed, &nf, &depth, pa, pb);
    if (depth < 1e29f) {
      *n = scale(nf, -1.0f);
      *dist = -depth;
      return 1;
    }
    // fully degenerate polytope: treat as touching along the guess (after a failed expansion: with the witnesses of the
    // simplex GJK ended on; after GJK's own tetrahedron: with whatever EPA left, as the oracle does)
    if (penetrating == 3) { *pa = ta; *pb = tb; }
    penetrating = 4;
  }
  if (penetrating == 2) { *pa = ta; *pb = tb; }
  else if (penetrating == 0) {
    v3 qa = mk(0, 0, 0), qb = mk(0, 0, 0);
#pragma unroll
    for (int i = 0; i < 4; ++i) if (i < s.n) { qa = madd(qa, s.a[i], s.lam[i]); qb = madd(qb, s.b[i], s.lam[i]); }
    *pa = qa; *pb = qb;
  }
  if (penetrating == 2 || penetrating == 4) {
    v3 g = guess;
    float gl = len(g);
    if (!(gl > 1e-6f)) { g = mk(0.0f, 0.0f, 1.0f); gl = 1.0f; }
    *n = scale(g, 1.0f / gl);
    *dist = 0.0f;
    return 1;
  }
  float d = len(v);
  if (d > max_dist) return 0;
  v3 nn = scale(v, 1.0f / d);
  // The direction of v = sum lam_i w_i carries the rounding of the barycentric weights: for two
  // parallel faces a few mm apart it is off by degrees in FP32 (the weights of a cm-sized
  // triangle resolve the closest point to ~0.1 mm only).  The separating direction is a property
  // of the closest FEATURE: the plane normal of a triangle, the perpendicular from the origin to
  // the line of a segment -- neither needs the weights.
  if (s.n == 3) {
    v3 e1 = sub(s.w[1], s.w[0]), e2 = sub(s.w[2], s.w[0]);
    v3 nf = cross(e1, e2);
    float l2 = dot(nf, nf);
    if (l2 > 1e-6f * dot(e1, e1) * dot(e2, e2)) {
      nf = scale(nf, 1.0f / fsqrtr(l2));
      if (dot(nf, v) < 0.0f) nf = scale(nf, -1.0f);
      float dd = dot(nf, s.w[0]);
      if (dd > 0.0f) { nn = nf; d = dd; }
    }
  } else if (s.n == 2) {
    v3 e1 = sub(s.w[1], s.w[0]);
    float ee = dot(e1, e1);
    if (ee > 0.0f) {
      v3 vp = madd(s.w[0], e1, -(dot(s.w[0], e1) / ee));
      float l = len(vp);
      if (l > 0.0f) { nn = scale(vp, 1.0f / l); d = l; }
    }
  }
  *n = nn;
  *dist = d;
  if (gc && s.n <= 3) {
    gc->n = s.n; gc->pair = pair;
#pragma unroll
    for (int i = 0; i < 3; ++i) if (i < s.n) { gc->ia[i] = s.ia[i]; gc->ib[i] = s.ib[i]; }
  }
  return 1;
}

// ---------------------------------------------------------------- manifold
// The manifold stays in LDS; each operation first stages what it needs in
// registers (independent loads -> one batch), decides, then writes back only
// the slot it changes.
struct ManPoint { v3 la, lb, nrm; float dist, ln, lt1, lt2; int col; };
RV_DEV ManPoint man_get(const DevMan& m, int i) {
  ManPoint p;
  p.la = ld3(m.la[i]); p.lb = ld3(m.lb[i]); p.nrm = ld3(m.nrm[i]);
  p.dist = m.dist[i]; p.ln = m.ln[i]; p.lt1 = m.lt1[i]; p.lt2 = m.lt2[i]; p.col = m.col[i];
  return p;
}
RV_DEV void man_put(DevMan& m, int i, const ManPoint& p) {
  st3(m.la[i], p.la); st3(m.lb[i], p.lb); st3(m.nrm[i], p.nrm);
  m.dist[i] = p.dist; m.ln[i] = p.ln; m.lt1[i] = p.lt1; m.lt2[i] = p.lt2; m.col[i] = p.col;
}
RV_DEV void man_remove(DevMan& m, int i) {
  int last = m.n - 1;
  if (i != last) { ManPoint p = man_get(m, last); man_put(m, i, p); }
  m.n = last;
}

RV_DEV float area4(v3 p0, v3 p1, v3 p2, v3 p3) {
  v3 c0 = cross(sub(p0, p1), sub(p2, p3));
  v3 c1 = cross(sub(p0, p2), sub(p1, p3));
  v3 c2 = cross(sub(p0, p3), sub(p1, p2));
  return fmaxr(fmaxr(dot(c0, c0), dot(c1, c1)), dot(c2, c2));
}

RV_DEV void man_add(DevMan& m, v3 la, v3 lb, v3 nrm, float dist, int col, float breaking) {
  const int n = m.n;
  v3 q[4]; float qd[4]; int qc[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) { q[i] = ld3(m.la[i]); qd[i] = m.dist[i]; qc[i] = m.col[i]; }
  int slot = -1;
  float best = breaking * breaking;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    if (i < n) {
      v3 d = sub(q[i], la);
      float dd = dot(d, d);
      if (dd < best && qc[i] == col) { best = dd; slot = i; }
    }
  }
  int keep_impulse = 0;
  if (slot >= 0) {
    keep_impulse = 1;
  } else if (n < 4) {
    slot = n; m.n = n + 1;
  } else {
    int deepest = -1; float dmin = dist;
#pragma unroll
    for (int i = 0; i < 4; ++i) if (qd[i] < dmin) { dmin = qd[i]; deepest = i; }
    float r0 = (deepest == 0) ? -1.0f : area4(la, q[1], q[2], q[3]);
    float r1 = (deepest == 1) ? -1.0f : area4(la, q[0], q[2], q[3]);
    float r2 = (deepest == 2) ? -1.0f : area4(la, q[0], q[1], q[3]);
    float r3 = (deepest == 3) ? -1.0f : area4(la, q[0], q[1], q[2]);
    slot = 0; float rb = r0;
    if (r1 > rb) { rb = r1; slot = 1; }
    if (r2 > rb) { rb = r2; slot = 2; }
    if (r3 > rb) { rb = r3; slot = 3; }
  }
  st3(m.la[slot], la); st3(m.lb[slot], lb); st3(m.nrm[slot], nrm);
  m.dist[slot] = dist; m.col[slot] = col;
  if (!keep_impulse) { m.ln[slot] = 0.0f; m.lt1[slot] = 0.0f; m.lt2[slot] = 0.0f; }
}

RV_DEV void plane_space(v3 n, v3* t1, v3* t2) {
  if (fabsr(n.z) > 0.7071067811865476f) {
    float a = n.y * n.y + n.z * n.z;
    float k = 1.0f / fsqrtr(a);
    *t1 = mk(0.0f, -n.z * k, n.y * k);
    *t2 = mk(a * k, -n.x * t1->z, n.x * t1->y);
  } else {
    float a = n.x * n.x + n.y * n.y;
    float k = 1.0f / fsqrtr(a);
    *t1 = mk(-n.y * k, n.x * k, 0.0f);
    *t2 = mk(-n.z * t1->y, n.z * t1->x, a * k);
  }
}

}  // namespace rv

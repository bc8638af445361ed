/*
 * orc_collide.h — GJK / EPA narrow phase and persistent 4-point manifolds of
 * the CPU oracle (TEST INFRASTRUCTURE ONLY; see orc_math.h).
 *
 * What it restates: the arithmetic behind `pybullet.stepSimulation`
 * (robovat/simulation/physics/bullet_physics.py:106-109) for convex-hull
 * pairs.  PyBullet 2.6.5 itself is a third-party wheel that is not vendored
 * in the reference tree (requirements.txt:9) — PARITY UNPINNED for this part:
 * the algorithm below is the published GJK (Gilbert-Johnson-Keerthi 1988,
 * Ericson "Real-Time Collision Detection" §5.1/§9.5 closest-point
 * sub-algorithms), EPA (van den Bergen 2001) and Bullet-style persistent
 * contact manifolds (contact cache keyed on local points, area-maximising
 * 4-point reduction), as DESIGN.md §3 specifies them.
 */
#ifndef ORC_COLLIDE_H_
#define ORC_COLLIDE_H_

#include "orc_math.h"

#define GJK_MAX_ITERS 32
#define GJK_REL_TOL R(1e-4)
#define GJK_PROGRESS_TOL R(1e-6)
#define EPA_MAX_VERTS 24
#define EPA_MAX_FACES 48
#define EPA_MAX_EDGES 32
#define EPA_MAX_ITERS 32
#define EPA_TOL R(1e-6)

typedef struct {
  real w[4][3], a[4][3], b[4][3];
  real lam[4];
  int n;
  int ia[4], ib[4];   /* vertex indices of a[], b[] (for the simplex cache) */
} orc_simplex;

/* the closest feature found by the last query of a pair: the next query of the SAME pair starts from it
 * (Bullet keeps a cached separating axis per pair for the same reason: temporal coherence) */
typedef struct { int n, pair; int ia[3], ib[3]; } orc_gjk_cache;

static inline int orc_support(const real (*verts)[3], int n, const real* d) {
  int best = 0;
  real bd = v3dot(verts[0], d);
  for (int i = 1; i < n; ++i) {
    real x = v3dot(verts[i], d);
    if (x > bd) { bd = x; best = i; }
  }
  return best;
}

/* closest point to the origin on segment (p0,p1): barycentrics l0,l1 */
static inline void orc_closest_segment(const real* p0, const real* p1, real* l) {
  real d[3]; v3sub(d, p1, p0);
  real dd = v3dot(d, d);
  if (!(dd > R(0.0))) { l[0] = R(0.0); l[1] = R(1.0); return; }
  real t = -v3dot(p0, d) / dd;
  if (t <= R(0.0)) { l[0] = R(1.0); l[1] = R(0.0); }
  else if (t >= R(1.0)) { l[0] = R(0.0); l[1] = R(1.0); }
  else { l[0] = R(1.0) - t; l[1] = t; }
}

/* closest point to the origin on triangle (a,b,c) — Ericson §5.1.5 with p = 0 */
static inline void orc_closest_triangle(const real* a, const real* b, const real* c, real* l) {
  real ab[3], ac[3];
  v3sub(ab, b, a); v3sub(ac, c, a);
  real d1 = -v3dot(ab, a), d2 = -v3dot(ac, a);
  if (d1 <= R(0.0) && d2 <= R(0.0)) { l[0] = R(1.0); l[1] = R(0.0); l[2] = R(0.0); return; }
  real d3 = -v3dot(ab, b), d4 = -v3dot(ac, b);
  if (d3 >= R(0.0) && d4 <= d3) { l[0] = R(0.0); l[1] = R(1.0); l[2] = R(0.0); return; }
  real vc = d1 * d4 - d3 * d2;
  if (vc <= R(0.0) && d1 >= R(0.0) && d3 <= R(0.0)) {
    real v = d1 / (d1 - d3);
    l[0] = R(1.0) - v; l[1] = v; l[2] = R(0.0); return;
  }
  real d5 = -v3dot(ab, c), d6 = -v3dot(ac, c);
  if (d6 >= R(0.0) && d5 <= d6) { l[0] = R(0.0); l[1] = R(0.0); l[2] = R(1.0); return; }
  real vb = d5 * d2 - d1 * d6;
  if (vb <= R(0.0) && d2 >= R(0.0) && d6 <= R(0.0)) {
    real w = d2 / (d2 - d6);
    l[0] = R(1.0) - w; l[1] = R(0.0); l[2] = w; return;
  }
  real va = d3 * d6 - d5 * d4;
  if (va <= R(0.0) && (d4 - d3) >= R(0.0) && (d5 - d6) >= R(0.0)) {
    real w = (d4 - d3) / ((d4 - d3) + (d5 - d6));
    l[0] = R(0.0); l[1] = R(1.0) - w; l[2] = w; return;
  }
  real s = va + vb + vc;
  if (!(s > R(0.0))) {
    /* degenerate (collinear) triangle: best of the three edges */
    real le[2], p[3], best = R(-1.0);
    const real* vs[3] = {a, b, c};
    l[0] = R(0.0); l[1] = R(0.0); l[2] = R(1.0);
    for (int e = 0; e < 3; ++e) {
      int i = e, j = (e + 1) % 3;
      orc_closest_segment(vs[i], vs[j], le);
      p[0] = vs[i][0] * le[0] + vs[j][0] * le[1];
      p[1] = vs[i][1] * le[0] + vs[j][1] * le[1];
      p[2] = vs[i][2] * le[0] + vs[j][2] * le[1];
      real dd = v3dot(p, p);
      if (best < R(0.0) || dd < best) {
        best = dd; l[0] = l[1] = l[2] = R(0.0); l[i] = le[0]; l[j] = le[1];
      }
    }
    return;
  }
  real denom = R(1.0) / s;
  real v = vb * denom, w = vc * denom;
  l[0] = R(1.0) - v - w; l[1] = v; l[2] = w;
}

/* Barycentric weights l[4] of the point of the simplex closest to the origin (zero for the
 * vertices that do not support it).  Returns 1 when the origin is enclosed by a tetrahedron.
 * The simplex is not modified. */
static inline int orc_simplex_weights(const orc_simplex* s, real* l) {
  l[0] = l[1] = l[2] = l[3] = R(0.0);
  if (s->n == 1) {
    l[0] = R(1.0);
  } else if (s->n == 2) {
    orc_closest_segment(s->w[0], s->w[1], l);
  } else if (s->n == 3) {
    orc_closest_triangle(s->w[0], s->w[1], s->w[2], l);
  } else {
    static const int F[4][4] = {{0, 1, 2, 3}, {0, 1, 3, 2}, {0, 2, 3, 1}, {1, 2, 3, 0}};
    int any_outside = 0;
    real best = R(-1.0);
    for (int f = 0; f < 4; ++f) {
      const real* p = s->w[F[f][0]]; const real* q = s->w[F[f][1]];
      const real* r = s->w[F[f][2]]; const real* o = s->w[F[f][3]];
      real pq[3], pr[3], nrm[3], po[3];
      v3sub(pq, q, p); v3sub(pr, r, p); v3cross(nrm, pq, pr); v3sub(po, o, p);
      real sp = v3dot(nrm, po);
      real so = -v3dot(nrm, p);
      if (sp * so > R(0.0)) continue; /* origin strictly on the inner side */
      any_outside = 1;
      real lf[3], c[3];
      orc_closest_triangle(p, q, r, lf);
      for (int k = 0; k < 3; ++k) c[k] = p[k] * lf[0] + q[k] * lf[1] + r[k] * lf[2];
      real dd = v3dot(c, c);
      if (best < R(0.0) || dd < best) {
        best = dd;
        l[0] = l[1] = l[2] = l[3] = R(0.0);
        l[F[f][0]] = lf[0]; l[F[f][1]] = lf[1]; l[F[f][2]] = lf[2];
      }
    }
    if (!any_outside) return 1;
  }
  return 0;
}
/* the closest point those weights give: sum of l_i w_i over the supporting vertices, in order */
static inline void orc_simplex_point(const orc_simplex* s, const real* l, real* v) {
  v[0] = v[1] = v[2] = R(0.0);
  for (int i = 0; i < s->n; ++i) if (l[i] > R(0.0)) v3madd(v, v, s->w[i], l[i]);
}
/* reduce the simplex to the vertices with positive weight (order preserved), store the weights */
static inline void orc_simplex_commit(orc_simplex* s, const real* l) {
  int m = 0;
  for (int i = 0; i < s->n; ++i) {
    if (l[i] > R(0.0)) {
      if (m != i) { v3cpy(s->w[m], s->w[i]); v3cpy(s->a[m], s->a[i]); v3cpy(s->b[m], s->b[i]); s->ia[m] = s->ia[i]; s->ib[m] = s->ib[i]; }
      s->lam[m] = l[i];
      ++m;
    }
  }
  s->n = m;
}
/* Reduce the simplex to the feature closest to the origin; writes v (closest
 * point) and lam.  Returns 1 when the origin is enclosed by a tetrahedron. */
static inline int orc_simplex_solve(orc_simplex* s, real* v) {
  real l[4];
  if (orc_simplex_weights(s, l)) return 1;
  orc_simplex_point(s, l, v);
  orc_simplex_commit(s, l);
  return 0;
}

/* EPA on a tetrahedron that encloses the origin.  Outputs the penetration
 * depth (>= 0), the face normal nf (pointing away from the origin) and the
 * witness points on A and B. */
static inline void orc_epa(const real (*A)[3], int nA, const real (*B)[3], int nB,
                           const orc_simplex* s, real* out_nf, real* out_depth, real* pa, real* pb) {
  real W[EPA_MAX_VERTS][3], VA[EPA_MAX_VERTS][3], VB[EPA_MAX_VERTS][3];
  int fi[EPA_MAX_FACES][3]; real fn[EPA_MAX_FACES][3]; real fd[EPA_MAX_FACES]; int alive[EPA_MAX_FACES];
  int nv = 4, nf = 0;
  for (int i = 0; i < 4; ++i) { v3cpy(W[i], s->w[i]); v3cpy(VA[i], s->a[i]); v3cpy(VB[i], s->b[i]); }
  static const int F[4][4] = {{0, 1, 2, 3}, {0, 1, 3, 2}, {0, 2, 3, 1}, {1, 2, 3, 0}};
  for (int f = 0; f < 4; ++f) {
    int i = F[f][0], j = F[f][1], k = F[f][2], o = F[f][3];
    real e1[3], e2[3], n[3], eo[3];
    v3sub(e1, W[j], W[i]); v3sub(e2, W[k], W[i]); v3cross(n, e1, e2); v3sub(eo, W[o], W[i]);
    if (v3dot(n, eo) > R(0.0)) { int t = j; j = k; k = t; v3scale(n, n, R(-1.0)); }
    real len = v3len(n);
    int degen = !(len > R(1e-12));
    if (degen) { len = R(1.0); }
    v3scale(n, n, R(1.0) / len);
    /* a zero-area face has no normal: give it infinite distance so it is never selected */
    fi[nf][0] = i; fi[nf][1] = j; fi[nf][2] = k; v3cpy(fn[nf], n); fd[nf] = degen ? R(1e30) : v3dot(n, W[i]); alive[nf] = 1; ++nf;
  }
  int bestf = 0;
  for (int it = 0; it < EPA_MAX_ITERS; ++it) {
    bestf = -1;
    for (int f = 0; f < nf; ++f) if (alive[f] && (bestf < 0 || fd[f] < fd[bestf])) bestf = f;
    real nd[3]; v3scale(nd, fn[bestf], R(-1.0));
    int ia = orc_support(A, nA, fn[bestf]);
    int ib = orc_support(B, nB, nd);
    real w[3]; v3sub(w, A[ia], B[ib]);
    if (v3dot(fn[bestf], w) - fd[bestf] < EPA_TOL || nv >= EPA_MAX_VERTS) break;
    /* expand */
    int en = 0; int ea[EPA_MAX_EDGES], eb[EPA_MAX_EDGES];
    for (int f = 0; f < nf; ++f) {
      if (!alive[f]) continue;
      real d[3]; v3sub(d, w, W[fi[f][0]]);
      if (v3dot(fn[f], d) > R(0.0)) {
        alive[f] = 0;
        for (int e = 0; e < 3; ++e) {
          int p = fi[f][e], q = fi[f][(e + 1) % 3];
          int found = -1;
          for (int x = 0; x < en; ++x) if (ea[x] == q && eb[x] == p) { found = x; break; }
          if (found >= 0) { ea[found] = ea[en - 1]; eb[found] = eb[en - 1]; --en; }
          else if (en < EPA_MAX_EDGES) { ea[en] = p; eb[en] = q; ++en; }
        }
      }
    }
    if (en == 0) break;
    v3cpy(W[nv], w); v3cpy(VA[nv], A[ia]); v3cpy(VB[nv], B[ib]);
    int overflow = 0;
    for (int x = 0; x < en; ++x) {
      int slot = -1;
      for (int f = 0; f < nf; ++f) if (!alive[f]) { slot = f; break; }
      if (slot < 0) { if (nf < EPA_MAX_FACES) slot = nf++; else { overflow = 1; break; } }
      int i = ea[x], j = eb[x], k = nv;
      real e1[3], e2[3], n[3];
      v3sub(e1, W[j], W[i]); v3sub(e2, W[k], W[i]); v3cross(n, e1, e2);
      real len = v3len(n);
      int degen = !(len > R(1e-12));
      if (degen) { len = R(1.0); }
      v3scale(n, n, R(1.0) / len);
      real d = v3dot(n, W[i]);
      if (d < R(0.0)) { int t = i; i = j; j = t; v3scale(n, n, R(-1.0)); d = -d; }
      if (degen) d = R(1e30);
      fi[slot][0] = i; fi[slot][1] = j; fi[slot][2] = k; v3cpy(fn[slot], n); fd[slot] = d; alive[slot] = 1;
    }
    ++nv;
    if (overflow) break;
  }
  bestf = -1;
  for (int f = 0; f < nf; ++f) if (alive[f] && (bestf < 0 || fd[f] < fd[bestf])) bestf = f;
  /* project the origin on the best face */
  real c[3], p0[3], p1[3], p2[3], l[3];
  v3scale(c, fn[bestf], fd[bestf]);
  v3sub(p0, W[fi[bestf][0]], c); v3sub(p1, W[fi[bestf][1]], c); v3sub(p2, W[fi[bestf][2]], c);
  orc_closest_triangle(p0, p1, p2, l);
  for (int k = 0; k < 3; ++k) {
    pa[k] = VA[fi[bestf][0]][k] * l[0] + VA[fi[bestf][1]][k] * l[1] + VA[fi[bestf][2]][k] * l[2];
    pb[k] = VB[fi[bestf][0]][k] * l[0] + VB[fi[bestf][1]][k] * l[1] + VB[fi[bestf][2]][k] * l[2];
  }
  /* c lies on the facet of the difference body that CONTAINS the best triangle, not necessarily inside the
   * triangle itself (a small box 2 mm deep in the table slab: the facet is the whole table top, the polytope's
   * triangle a sliver of it): the clamped barycentric point then gives witnesses with pa - pb != c, i.e. two
   * anchors centimetres apart tangentially that the next refresh drops as a broken point.  The three a_i all
   * lie on A's support face for this normal, the three b_i on B's: when one of the two features is a single
   * vertex it is that body's witness and the other is its projection; otherwise the mismatch is split.
   * (found by the closed-form pin tests/test_independent_pin.py) */
  {
    const int i0 = fi[bestf][0], i1 = fi[bestf][1], i2 = fi[bestf][2];
    const int vtx_a = VA[i0][0] == VA[i1][0] && VA[i0][1] == VA[i1][1] && VA[i0][2] == VA[i1][2] &&
                      VA[i0][0] == VA[i2][0] && VA[i0][1] == VA[i2][1] && VA[i0][2] == VA[i2][2];
    const int vtx_b = VB[i0][0] == VB[i1][0] && VB[i0][1] == VB[i1][1] && VB[i0][2] == VB[i1][2] &&
                      VB[i0][0] == VB[i2][0] && VB[i0][1] == VB[i2][1] && VB[i0][2] == VB[i2][2];
    if (vtx_a) { for (int k = 0; k < 3; ++k) pb[k] = pa[k] - c[k]; }
    else if (vtx_b) { for (int k = 0; k < 3; ++k) pa[k] = pb[k] + c[k]; }
    else {
      for (int k = 0; k < 3; ++k) {
        const real dl = (pa[k] - pb[k]) - c[k];
        pa[k] = pa[k] - R(0.5) * dl; pb[k] = pb[k] + R(0.5) * dl;
      }
    }
  }
  v3cpy(out_nf, fn[bestf]);
  *out_depth = fd[bestf];
}

/* The closest point of the GJK simplex IS the origin but the simplex is a vertex, a segment or a triangle (two cores that
 * overlap mirror-symmetrically about the origin of the difference body: crossed edges exactly centred, a face centred on
 * a face): grow it into a tetrahedron of non-zero volume that has the origin inside or on its boundary, so that EPA can
 * measure the overlap.  Returns 0 when the difference body is flat in every direction tried (the cores really only touch).
 * Support points of the difference body A - B in direction d: support(A, d) - support(B, -d). */
static inline void orc_diff_support(const real (*A)[3], int nA, const real (*B)[3], int nB, const real* d, real* w, real* a, real* b) {
  real nd[3]; v3scale(nd, d, R(-1.0));
  v3cpy(a, A[orc_support(A, nA, d)]); v3cpy(b, B[orc_support(B, nB, nd)]); v3sub(w, a, b);
}
static inline int orc_simplex_expand(const real (*A)[3], int nA, const real (*B)[3], int nB, orc_simplex* s) {
  if (s->n == 1) {
    int ok = 0;
    for (int k = 0; k < 6 && !ok; ++k) {
      real d[3] = {R(0.0), R(0.0), R(0.0)}; d[k >> 1] = (k & 1) ? R(-1.0) : R(1.0);
      real w[3], a[3], b[3], e[3];
      orc_diff_support(A, nA, B, nB, d, w, a, b);
      v3sub(e, w, s->w[0]);
      if (v3dot(e, e) > R(1e-12)) { v3cpy(s->w[1], w); v3cpy(s->a[1], a); v3cpy(s->b[1], b); s->n = 2; ok = 1; }
    }
    if (!ok) return 0;
  }
  if (s->n == 2) {
    real d[3]; v3sub(d, s->w[1], s->w[0]);
    const real dd = v3dot(d, d);
    int ax = 0;
    if (rabs(d[1]) < rabs(d[ax])) ax = 1;
    if (rabs(d[2]) < rabs(d[ax])) ax = 2;
    real e0[3] = {R(0.0), R(0.0), R(0.0)}; e0[ax] = R(1.0);
    real u[3], v[3]; v3cross(u, d, e0); v3cross(v, d, u);
    int ok = 0;
    for (int k = 0; k < 4 && !ok; ++k) {
      real dir[3];
      v3scale(dir, k < 2 ? u : v, (k & 1) ? R(-1.0) : R(1.0));
      real w[3], a[3], b[3], e[3], cr[3];
      orc_diff_support(A, nA, B, nB, dir, w, a, b);
      v3sub(e, w, s->w[0]); v3cross(cr, e, d);
      if (v3dot(e, e) > R(1e-12) && v3dot(cr, cr) > R(1e-6) * dd * v3dot(e, e)) { v3cpy(s->w[2], w); v3cpy(s->a[2], a); v3cpy(s->b[2], b); s->n = 3; ok = 1; }
    }
    if (!ok) return 0;
  }
  if (s->n == 3) {
    real e1[3], e2[3], nrm[3];
    v3sub(e1, s->w[1], s->w[0]); v3sub(e2, s->w[2], s->w[0]); v3cross(nrm, e1, e2);
    const real nl = rsqrt_(v3dot(nrm, nrm));
    if (!(nl > R(0.0))) return 0;
    int ok = 0;
    for (int k = 0; k < 2 && !ok; ++k) {
      real dir[3]; v3scale(dir, nrm, k ? R(-1.0) : R(1.0));
      real w[3], a[3], b[3], e[3];
      orc_diff_support(A, nA, B, nB, dir, w, a, b);
      v3sub(e, w, s->w[0]);
      if (v3dot(e, dir) > R(1e-6) * nl) { v3cpy(s->w[3], w); v3cpy(s->a[3], a); v3cpy(s->b[3], b); s->n = 4; ok = 1; }
    }
    if (!ok) return 0;
  }
  return s->n == 4;
}

/* GJK distance between convex vertex sets A and B (world frame), with EPA on
 * overlap.  Returns 0 if the sets are farther apart than max_dist, else 1 and
 * n (unit, from B towards A), signed core distance (negative = overlap) and
 * witness points on the two cores.  guess = initial search direction. */
static long orc_gjk_calls = 0, orc_gjk_iters = 0;
#ifdef ORC_DEBUG_GJK
static __thread int orc_dbg_reason, orc_dbg_iters, orc_dbg_cache_n, orc_dbg_start_n, orc_dbg_end_n;
#define ORC_DBG(x) x
#else
#define ORC_DBG(x)
#endif
static inline int orc_gjk_epa_c(const real (*A)[3], int nA, const real (*B)[3], int nB,
                                const real* guess, real max_dist,
                                real* n, real* dist, real* pa, real* pb, orc_gjk_cache* gc, int pair) {
  orc_simplex s; s.n = 0;
  real v[3]; v3cpy(v, guess);
  if (!(v3dot(v, v) > R(1e-12))) v3set(v, R(1.0), R(0.0), R(0.0));
  int have_v = 0, penetrating = 0;
  ORC_DBG(orc_dbg_reason = 0; orc_dbg_iters = 0; orc_dbg_cache_n = (gc && gc->pair == pair) ? gc->n : 0; orc_dbg_start_n = 0;)
  if (gc && gc->n > 0 && gc->pair == pair) {
    int ok = 1;
    for (int i = 0; i < gc->n; ++i) if (gc->ia[i] >= nA || gc->ib[i] >= nB) ok = 0;
    if (ok) {
      real v0[3];
      for (int i = 0; i < gc->n; ++i) {
        v3cpy(s.a[i], A[gc->ia[i]]); v3cpy(s.b[i], B[gc->ib[i]]); v3sub(s.w[i], s.a[i], s.b[i]);
        s.ia[i] = gc->ia[i]; s.ib[i] = gc->ib[i];
      }
      s.n = gc->n;
      if (!orc_simplex_solve(&s, v0) && v3dot(v0, v0) > R(1e-14)) { v3cpy(v, v0); have_v = 1; }
      else s.n = 0;
      ORC_DBG(orc_dbg_start_n = s.n;)
    }
  }
  if (gc) gc->n = 0;
#ifdef ORC_COUNT_GJK
  orc_gjk_calls++;
#endif
  for (int it = 0; it < GJK_MAX_ITERS; ++it) {
#ifdef ORC_COUNT_GJK
    orc_gjk_iters++;
#endif
    real nv[3]; v3scale(nv, v, R(-1.0));
    int ia = orc_support(A, nA, nv);
    int ib = orc_support(B, nB, v);
    real w[3]; v3sub(w, A[ia], B[ib]);
    real vv = v3dot(v, v), vw = v3dot(v, w);
    if (vw > R(0.0) && vw * vw > max_dist * max_dist * vv) return 0;
    int dup = 0;
    for (int k = 0; k < s.n; ++k)
      if (s.w[k][0] == w[0] && s.w[k][1] == w[1] && s.w[k][2] == w[2]) dup = 1;
    ORC_DBG(orc_dbg_iters = it + 1;)
    if (dup) { ORC_DBG(orc_dbg_reason = 1;) break; }
    if (have_v && vv - vw <= GJK_REL_TOL * vv) { ORC_DBG(orc_dbg_reason = 2;) break; }
    v3cpy(s.w[s.n], w); v3cpy(s.a[s.n], A[ia]); v3cpy(s.b[s.n], B[ib]); s.ia[s.n] = ia; s.ib[s.n] = ib; s.n++;
    real l[4], vc[3];
    if (orc_simplex_weights(&s, l)) { penetrating = 1; break; }
    orc_simplex_point(&s, l, vc);
    real vn = v3dot(vc, vc);
    if (!(vn > R(1e-14))) { orc_simplex_commit(&s, l); v3cpy(v, vc); penetrating = 2; break; }
    /* no progress: the new support point does not bring the closest point closer (face-face
     * contacts would otherwise cycle through the vertices of the touching faces).  The new point
     * is DROPPED and the query ends on the simplex it had: with four nearly coplanar points (a
     * cached face feature plus one more vertex of the same face) the sub-simplex chosen in FP32
     * can be farther from the origin than the one before, with a normal tilted by 20 degrees */
    if (have_v && vv - vn <= GJK_PROGRESS_TOL * vv) { s.n--; ORC_DBG(orc_dbg_reason = 3;) break; }
    orc_simplex_commit(&s, l); v3cpy(v, vc);
    have_v = 1;
  }
  ORC_DBG(orc_dbg_end_n = s.n; if (penetrating) orc_dbg_reason = 10 + penetrating;)
  real ta[3] = {R(0.0), R(0.0), R(0.0)}, tb[3] = {R(0.0), R(0.0), R(0.0)};
  if (penetrating == 2) {
    /* the origin lies ON a vertex / segment / triangle of the simplex: the witnesses of "touching" first, then (round 5)
     * the simplex is grown into a tetrahedron and EPA decides whether the cores merely touch or overlap */
    for (int i = 0; i < s.n; ++i) { v3madd(ta, ta, s.a[i], s.lam[i]); v3madd(tb, tb, s.b[i], s.lam[i]); }
    if (orc_simplex_expand(A, nA, B, nB, &s)) penetrating = 3;
  }
  if (penetrating == 1 || penetrating == 3) {
    real nf[3], depth;
    orc_epa(A, nA, B, nB, &s, nf, &depth, pa, pb);
    if (depth < R(1e29)) {
      v3scale(n, nf, R(-1.0));
      *dist = -depth;
      return 1;
    }
    /* fully degenerate polytope: treat as touching along the guess */
    if (penetrating == 3) { v3cpy(pa, ta); v3cpy(pb, tb); }
    penetrating = 2;
    goto touching;
  }
  if (penetrating == 2) { v3cpy(pa, ta); v3cpy(pb, tb); }
  else {
    pa[0] = pa[1] = pa[2] = R(0.0); pb[0] = pb[1] = pb[2] = R(0.0);
    for (int i = 0; i < s.n; ++i) { v3madd(pa, pa, s.a[i], s.lam[i]); v3madd(pb, pb, s.b[i], s.lam[i]); }
  }
touching:
  if (penetrating == 2) {
    /* cores touch on a lower-dimensional simplex: zero depth along the guess */
    real g[3]; v3cpy(g, guess);
    real gl = v3len(g);
    if (!(gl > R(1e-6))) { v3set(g, R(0.0), R(0.0), R(1.0)); gl = R(1.0); }
    v3scale(n, g, R(1.0) / gl);
    *dist = R(0.0);
    return 1;
  }
  real d = v3len(v);
  if (d > max_dist) return 0;
  v3scale(n, v, R(1.0) / d);
  /* The direction of v = sum lam_i w_i carries the rounding of the barycentric weights: for two
   * parallel faces a few mm apart it is off by degrees in FP32 (the weights of a cm-sized
   * triangle resolve the closest point to ~0.1 mm only).  The separating direction is a property
   * of the closest FEATURE: the plane normal of a triangle, the perpendicular from the origin to
   * the line of a segment -- neither needs the weights. */
  if (s.n == 3) {
    real e1[3], e2[3], nf[3];
    v3sub(e1, s.w[1], s.w[0]); v3sub(e2, s.w[2], s.w[0]);
    v3cross(nf, e1, e2);
    real l2 = v3dot(nf, nf);
    if (l2 > R(1e-6) * v3dot(e1, e1) * v3dot(e2, e2)) {
      v3scale(nf, nf, R(1.0) / rsqrt_(l2));
      if (v3dot(nf, v) < R(0.0)) v3scale(nf, nf, R(-1.0));
      real dd = v3dot(nf, s.w[0]);
      if (dd > R(0.0)) { v3cpy(n, nf); d = dd; }
    }
  } else if (s.n == 2) {
    real e1[3], vp[3];
    v3sub(e1, s.w[1], s.w[0]);
    real ee = v3dot(e1, e1);
    if (ee > R(0.0)) {
      v3madd(vp, s.w[0], e1, -(v3dot(s.w[0], e1) / ee));
      real l = v3len(vp);
      if (l > R(0.0)) { v3scale(n, vp, R(1.0) / l); d = l; }
    }
  }
  *dist = d;
  if (gc && s.n <= 3) {
    gc->n = s.n; gc->pair = pair;
    for (int i = 0; i < s.n; ++i) { gc->ia[i] = s.ia[i]; gc->ib[i] = s.ib[i]; }
  }
  return 1;
}
static inline int orc_gjk_epa(const real (*A)[3], int nA, const real (*B)[3], int nB,
                              const real* guess, real max_dist,
                              real* n, real* dist, real* pa, real* pb) {
  return orc_gjk_epa_c(A, nA, B, nB, guess, max_dist, n, dist, pa, pb, (orc_gjk_cache*)0, 0);
}

/* ---------------- persistent manifold ---------------- */
typedef struct {
  int n;
  real la[4][3];   /* contact point on A, in A's body frame            */
  real lb[4][3];   /* contact point on B, in B's frame (world if static) */
  real nrm[4][3];  /* world normal, from B towards A                    */
  real dist[4];
  real ln[4], lt1[4], lt2[4]; /* accumulated impulses (warm start)      */
  int  col[4];     /* arm collider id for arm-body manifolds, else -1   */
  real acc;        /* relative motion since the last full narrow phase  */
  int  age;        /* full passes since the last feature stage          */
  orc_gjk_cache gc; /* closest feature of the last convex query of this manifold */
} orc_manifold;

static inline void orc_man_remove(orc_manifold* m, int i) {
  int last = m->n - 1;
  if (i != last) {
    v3cpy(m->la[i], m->la[last]); v3cpy(m->lb[i], m->lb[last]); v3cpy(m->nrm[i], m->nrm[last]);
    m->dist[i] = m->dist[last]; m->ln[i] = m->ln[last]; m->lt1[i] = m->lt1[last]; m->lt2[i] = m->lt2[last];
    m->col[i] = m->col[last];
  }
  m->n = last;
}

static inline real orc_area4(const real* p0, const real* p1, const real* p2, const real* p3) {
  real a0[3], b0[3], c0[3], c1[3], c2[3];
  v3sub(a0, p0, p1); v3sub(b0, p2, p3); v3cross(c0, a0, b0);
  v3sub(a0, p0, p2); v3sub(b0, p1, p3); v3cross(c1, a0, b0);
  v3sub(a0, p0, p3); v3sub(b0, p1, p2); v3cross(c2, a0, b0);
  return rmax(rmax(v3dot(c0, c0), v3dot(c1, c1)), v3dot(c2, c2));
}

/* add (or merge) a contact; la/lb local points, nrm world normal */
static inline void orc_man_add(orc_manifold* m, const real* la, const real* lb, const real* nrm,
                               real dist, int col, real breaking) {
  int slot = -1;
  real best = breaking * breaking;
  for (int i = 0; i < m->n; ++i) {
    real d[3]; v3sub(d, m->la[i], la);
    real dd = v3dot(d, d);
    if (dd < best && m->col[i] == col) { best = dd; slot = i; }
  }
  int keep_impulse = 0;
  if (slot >= 0) {
    keep_impulse = 1;
  } else if (m->n < 4) {
    slot = m->n++;
  } else {
    /* keep the deepest point, maximise the contact area (Bullet-style) */
    int deepest = -1; real dmin = dist;
    for (int i = 0; i < 4; ++i) if (m->dist[i] < dmin) { dmin = m->dist[i]; deepest = i; }
    real r[4];
    r[0] = (deepest == 0) ? R(-1.0) : orc_area4(la, m->la[1], m->la[2], m->la[3]);
    r[1] = (deepest == 1) ? R(-1.0) : orc_area4(la, m->la[0], m->la[2], m->la[3]);
    r[2] = (deepest == 2) ? R(-1.0) : orc_area4(la, m->la[0], m->la[1], m->la[3]);
    r[3] = (deepest == 3) ? R(-1.0) : orc_area4(la, m->la[0], m->la[1], m->la[2]);
    slot = 0;
    for (int i = 1; i < 4; ++i) if (r[i] > r[slot]) slot = i;
  }
  v3cpy(m->la[slot], la); v3cpy(m->lb[slot], lb); v3cpy(m->nrm[slot], nrm);
  m->dist[slot] = dist; m->col[slot] = col;
  if (!keep_impulse) { m->ln[slot] = R(0.0); m->lt1[slot] = R(0.0); m->lt2[slot] = R(0.0); }
}

#endif /* ORC_COLLIDE_H_ */

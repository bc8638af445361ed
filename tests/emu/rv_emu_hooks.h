// rv_emu_hooks.h -- the HOST bodies of the lane emulation (tests/emu/rv_emu.cpp), moved out of the product header
// robovat_amd/csrc/rv_dev_env.h (round 5).  rv_dev_env.h includes this file at the hook points below, ONLY when it is compiled
// for the host (RV_ON_DEVICE == 0: -DRV_EMULATE or a plain C++ compiler); hipcc never sees it.  A section is the host-side
// counterpart of the device code at the same place: a lane phase is a loop over 64 lanes, a wave builtin its scalar equivalent,
// the lane-per-row solvers are replaced by the row-list solver the oracle runs.  TEST SCAFFOLDING -- not a product path.
// (No include guard: every inclusion selects one section through RV_EMU_SECTION.)

#if RV_EMU_SECTION == 1      // rv_dev_env.h: // the row list of the islands of one or two bodies (host emulation)
// the row list of the islands of one or two bodies (host emulation)
RV_DEV int solver_row_list(Shared& S, const int* label, const int* on_, const int* act_, const int* big_) {
  DevEnv& e = S.e;
  int n = 0;
  // a body's own rows: bodies ascending; the two members X < Y of a two-body island are visited
  // together, slot by slot (X's row of slot t, then Y's; slot = 3 * (4 * [arm] + point) + row)
  for (int b = 0; b < RV_MAXB; ++b) {
    if (!on_[b] || big_[label[b]]) continue;
    int partner = -1;
    for (int x = 0; x < RV_MAXB; ++x) if (x != b && on_[x] && label[x] == label[b]) partner = x;
    if (partner >= 0 && partner < b) continue;
    for (int t = 0; t < 24; ++t)
      for (int side = 0; side < 2; ++side) {
        const int body = side == 0 ? b : partner;
        if (body < 0) continue;
        const int p = t / 3, k = t % 3, mi = p < 4 ? RV_TIDX(body) : RV_AIDX(body), i = p & 3;
        if (i >= e.man[mi].n) continue;
        if (n + 1 > RV_SOLVE_ROWS) return -1;
        S.s.rowmap[n++] = RV_ROW_PACK(mi, i, k, body, -1, label[body]);
      }
  }
  for (int rd = 0; rd < 3; ++rd)
    for (int x = 0; x < 2; ++x) {
      const int kp = bb_round_pair(rd, x);
      if (!act_[kp] || big_[label[bb_a(kp)]]) continue;
      const int np_ = e.man[RV_BBIDX(kp)].n;
      if (n + 3 * np_ > RV_SOLVE_ROWS) return -1;
      for (int i = 0; i < np_; ++i) for (int k = 0; k < 3; ++k) S.s.rowmap[n++] = RV_ROW_PACK(RV_BBIDX(kp), i, k, bb_a(kp), bb_b(kp), label[bb_a(kp)]);
    }
  return n;
}
#endif

#if RV_EMU_SECTION == 2      // rv_dev_env.h: // fing != 0 (rv_config.finger_dynamics, at most one awake body): the two finger joints ar
// fing != 0 (rv_config.finger_dynamics, at most one awake body): the two finger joints are DOFs of
// the system as well -- contact rows on a finger pad carry jf on their finger's velocity, each finger
// has a motor row after the contact rows (see solve_island_fingers, the device version)
RV_DEV void solve_rows(Shared& S, const Consts& K, const int n_rows, const int fing, const int limb, const int motor_isl, const float* isl_tol) {
  DevEnv& e = S.e; const rv_config* c = K.cfg; const rv_arm* arm = K.arm;
  static thread_local float A[RV_SOLVE_ROWS + 9][RV_SOLVE_ROWS + 9];
  float g[RV_SOLVE_ROWS + 9], lam[RV_SOLVE_ROWS + 9], invk[RV_SOLVE_ROWS + 9], bias[RV_SOLVE_ROWS + 9], mu[RV_SOLVE_ROWS + 9], cap[RV_SOLVE_ROWS + 9];
  float jf[RV_SOLVE_ROWS + 9], pf[RV_SOLVE_ROWS + 9], mlo[2] = {0.0f, 0.0f}, mhi[2] = {0.0f, 0.0f}; int fi[RV_SOLVE_ROWS + 9];
  const float mf = c->finger_mass, imf = fing ? 1.0f / c->finger_mass : 0.0f, fdt = c->finger_max_force * c->dt;
  const float qf0[2] = {e.qd[RV_NLIMB], e.qd[RV_NLIMB + 1]};
  // limb != 0 (rv_config.limb_dynamics, at most one awake body): the seven limb joints are DOFs as well.
  // Rows of the arm manifold and the seven limb motor rows (after the finger motor rows) are 'limb rows':
  // row r has the joint-space Jacobian ja[r] (a motor row: e_j) and the velocity change per unit impulse
  // pj[r] = M^-1 ja^T (a motor row: column j of M^-1); A_rs gains ja[r] . pj[s]
  const int nfm = fing ? 2 : 0, nlm = limb ? RV_NLIMB : 0;
  const int n_all = n_rows + nfm + nlm;
  int la[RV_SOLVE_ROWS + 9]; float ja[RV_SOLVE_ROWS + 9][RV_NLIMB], pj[RV_SOLVE_ROWS + 9][RV_NLIMB], dq0[RV_NLIMB];
  for (int x = 0; x < RV_NLIMB; ++x) dq0[x] = limb ? -S.s.limb_dv[x] : 0.0f;     // the solve starts from the velocity before the motor step
  int fisl = 0;
  J6 jx[RV_SOLVE_ROWS][RV_MAXB];
  for (int r = 0; r < n_rows; ++r) {
    const int rm = S.s.rowmap[r];
    const int mi = RV_ROW_MI(rm), pi = RV_ROW_I(rm), k = RV_ROW_K(rm), ra = RV_ROW_A(rm), rb = RV_ROW_B(rm);
    const Row& R = S.s.u.r.rows[mi][pi];
    const DevMan& mm = e.man[mi];
    for (int x = 0; x < RV_MAXB; ++x) { jx[r][x].l = mk(0, 0, 0); jx[r][x].a = mk(0, 0, 0); }
    const v3 dir = ld3(R.dir[k]), rxa = ld3(R.rxa[k]);
    invk[r] = R.invk[k]; mu[r] = R.mu; bias[r] = k == 0 ? R.target : 0.0f; cap[r] = R.cap;
    lam[r] = k == 0 ? mm.ln[pi] : (k == 1 ? mm.lt1[pi] : mm.lt2[pi]);
    float gg = dot(dir, ld3(e.body[ra] + 7)) + dot(rxa, ld3(e.body[ra] + 10));
    v3 nd = mk(0, 0, 0), nrxb = mk(0, 0, 0);
    if (rb >= 0) {
      const v3 rxb = ld3(R.rxb[k]);
      gg -= dot(dir, ld3(e.body[rb] + 7)) + dot(rxb, ld3(e.body[rb] + 10));
      nd = mk(-dir.x, -dir.y, -dir.z); nrxb = mk(-rxb.x, -rxb.y, -rxb.z);
    } else gg -= R.vbc[k];
    jf[r] = 0.0f; pf[r] = 0.0f; fi[r] = -1;
    if (fing) {
      fi[r] = R.fidx; fisl = RV_ROW_ISL(rm);
      if (fi[r] >= 0) { jf[r] = R.jf[k]; pf[r] = jf[r] * imf; gg += jf[r] * qf0[fi[r]]; }
    }
    la[r] = 0;
    if (limb) {
      fisl = RV_ROW_ISL(rm);
      if (mi >= RV_AIDX(0)) {
        const int lrow = pi * 3 + k;
        la[r] = 1; invk[r] = S.s.linvk[lrow];
        float t = 0.0f;
        for (int x = 0; x < RV_NLIMB; ++x) { ja[r][x] = S.s.lJa[lrow][x]; pj[r][x] = S.s.lMiJ[lrow][x]; t = t + ja[r][x] * dq0[x]; }
        gg += t;
      }
    }
    g[r] = gg;
    jx[r][ra].l = dir; jx[r][ra].a = rxa;
    if (rb >= 0) { jx[r][rb].l = nd; jx[r][rb].a = nrxb; }
  }
  if (motor_isl >= 0) fisl = motor_isl;
  for (int m = 0; fing && m < 2; ++m) {       // motor rows
    const int r = n_rows + m;
    const float i0 = mf * S.s.fing_dv[m];
    g[r] = qf0[m] - S.s.fing_vt[m]; lam[r] = 0.0f; invk[r] = mf; bias[r] = 0.0f; mu[r] = 0.0f; cap[r] = 0.0f;
    jf[r] = 1.0f; pf[r] = imf; fi[r] = m;
    mlo[m] = -fdt - i0; mhi[m] = fdt - i0;
    la[r] = 0;
  }
  for (int j = 0; j < nlm; ++j) {             // limb motor rows
    const int r = n_rows + nfm + j, lrow = 12 + j;
    g[r] = dq0[j] - S.s.ltgt[j]; lam[r] = 0.0f; invk[r] = 1.0f / S.s.lA[j][RV_NLIMB + j]; bias[r] = 0.0f; mu[r] = 0.0f; cap[r] = 0.0f;
    jf[r] = 0.0f; pf[r] = 0.0f; fi[r] = -1; la[r] = 1;
    for (int x = 0; x < RV_NLIMB; ++x) { ja[r][x] = S.s.lJa[lrow][x]; pj[r][x] = S.s.lMiJ[lrow][x]; }
  }
  for (int r = 0; r < n_rows; ++r)
    for (int s = 0; s < n_rows; ++s) {
      const int q = S.s.rowmap[s];
      const Row& Q = S.s.u.r.rows[RV_ROW_MI(q)][RV_ROW_I(q)];
      const int ks = RV_ROW_K(q), as = RV_ROW_A(q), bs = RV_ROW_B(q);
      const v3 ds = ld3(Q.dir[ks]);
      float a_ = dotj(jx[r][as], scale(ds, e.inv_mass[as]), ld3(Q.aa[ks]));
      if (bs >= 0) {
        const v3 t = scale(ds, e.inv_mass[bs]), ab = ld3(Q.ab[ks]);
        a_ = a_ + dotj(jx[r][bs], mk(-t.x, -t.y, -t.z), mk(-ab.x, -ab.y, -ab.z));
      }
      if (fing && fi[r] >= 0 && fi[r] == fi[s]) a_ = a_ + jf[r] * pf[s];
      A[r][s] = a_;
    }
  for (int m = 0; fing && m < 2; ++m) {
    const int q = n_rows + m;
    for (int r = 0; r < n_rows; ++r) { A[r][q] = fi[r] == m ? jf[r] * pf[q] : 0.0f; A[q][r] = fi[r] == m ? pf[r] : 0.0f; }
    for (int m2 = 0; m2 < 2; ++m2) A[q][n_rows + m2] = m == m2 ? pf[q] : 0.0f;
  }
  for (int j = 0; j < nlm; ++j) {
    const int q = n_rows + nfm + j;
    for (int r = 0; r < n_all; ++r) { A[r][q] = 0.0f; A[q][r] = 0.0f; }
  }
  for (int r = 0; limb && r < n_all; ++r)
    for (int s = 0; s < n_all; ++s) {
      if (!(la[r] && la[s])) continue;
      float t = 0.0f;
      for (int x = 0; x < RV_NLIMB; ++x) t = t + ja[r][x] * pj[s][x];
      A[r][s] = A[r][s] + t;
    }
  for (int s = 0; s < n_rows; ++s) for (int r = 0; r < n_all; ++r) g[r] = g[r] + A[r][s] * lam[s];
  // normalised residual form of the row step (see solve_singles / oracle solve_rows): rr = (bias - g) invk, C = -(A invk)
  float rr[RV_SOLVE_ROWS + 9];
  for (int r = 0; r < n_all; ++r) {
    rr[r] = (bias[r] - g[r]) * invk[r];
    for (int s = 0; s < n_all; ++s) A[r][s] = -(A[r][s] * invk[r]);
  }
  int isl_rows = 0, done = 0;
  float best[RV_MAXB] = {1e30f, 1e30f, 1e30f, 1e30f}; int since[RV_MAXB] = {0, 0, 0, 0};
  for (int s = 0; s < n_rows; ++s) isl_rows |= 1 << RV_ROW_ISL(S.s.rowmap[s]);
  if (fing || limb) isl_rows |= 1 << fisl;
  RV_CNT(21, 1) RV_CNT(23, n_rows)
  for (int it = 0; it < c->solver_iters; ++it) {
    RV_CNT(22, 1)
    float res[RV_MAXB] = {0.0f, 0.0f, 0.0f, 0.0f};
    float limtab[RV_NMAN][4];    // friction bound of every point: mu x its normal impulse (the rows of a point need not be neighbours in the list)
    for (int s = 0; s < n_rows; ++s) {
      const int q = S.s.rowmap[s];
      const int isl = RV_ROW_ISL(q);
      if ((done >> isl) & 1) continue;
      float nl;
      const float lim = RV_ROW_K(q) == 0 ? 0.0f : limtab[RV_ROW_MI(q)][RV_ROW_I(q)];
      if (RV_ROW_K(q) == 0) nl = fclampr(lam[s] + rr[s], 0.0f, cap[s]);
      else nl = fclampr(lam[s] + rr[s], -lim, lim);
      const float d = nl - lam[s];
      lam[s] = nl;
      if (RV_ROW_K(q) == 0) limtab[RV_ROW_MI(q)][RV_ROW_I(q)] = mu[s] * nl;
      res[isl] = fmaxr(res[isl], fabsr(d));
#ifdef RV_EMU_COUNT
      if (it == c->solver_iters - 1 && fabsr(d) >= c->solver_tol) {
        const int mi_ = RV_ROW_MI(q), cls = (mi_ < RV_MAXB ? 0 : (mi_ < RV_MAXB + RV_NBB ? 1 : 2)) * 2 + (RV_ROW_K(q) != 0);
        rv_emu_dbg[cls] += 1; if (RV_ROW_K(q) == 0 && nl >= cap[s]) rv_emu_dbg[6] += 1; if (RV_ROW_K(q) != 0 && (nl >= lim || nl <= -lim)) rv_emu_dbg[7] += 1;
      }
#endif
      for (int r = 0; r < n_all; ++r) rr[r] = rv_fma(A[r][s], d, rr[r]);
    }
    // (the motor rows belong to island fisl and stop with it -- other islands may still be sweeping)
    const int motors_on = !((done >> fisl) & 1);
    for (int m = 0; fing && motors_on && m < 2; ++m) {
      const int q = n_rows + m;
      const float nl = fclampr(lam[q] + rr[q], mlo[m], mhi[m]);
      const float d = nl - lam[q];
      lam[q] = nl;
      res[fisl] = fmaxr(res[fisl], fabsr(d));
      for (int r = 0; r < n_all; ++r) rr[r] = rv_fma(A[r][q], d, rr[r]);
    }
    for (int j = 0; motors_on && j < nlm; ++j) {
      const int q = n_rows + nfm + j;
      const float nl = fclampr(lam[q] + rr[q], S.s.llo[j], S.s.lhi[j]);
      const float d = nl - lam[q];
      lam[q] = nl;
      res[fisl] = fmaxr(res[fisl], fabsr(d));
      for (int r = 0; r < n_all; ++r) rr[r] = rv_fma(A[r][q], d, rr[r]);
    }
#ifdef RV_EMU_COUNT
    for (int x = 0; x < RV_MAXB; ++x) if (((isl_rows >> x) & 1) && !((done >> x) & 1) && (res[x] < c->solver_tol || it == c->solver_iters - 1)) {
      int nr = 0, arm_rows = 0;
      for (int s = 0; s < n_rows; ++s) if (RV_ROW_ISL(S.s.rowmap[s]) == x) { ++nr; arm_rows += RV_ROW_MI(S.s.rowmap[s]) >= RV_MAXB + RV_NBB; }
      const int cap = !(res[x] < c->solver_tol);
      rv_emu_cnt[35] += 1; rv_emu_cnt[33] += (long)nr * (it + 1);
      if (cap && arm_rows && e.phase >= 0 && e.phase < 8) rv_emu_cnt[38 + e.phase] += 1;
      if (cap && arm_rows) { float nn = 0.0f; for (int s = 0; s < n_rows; ++s) if (RV_ROW_ISL(S.s.rowmap[s]) == x && RV_ROW_K(S.s.rowmap[s]) == 0 && RV_ROW_MI(S.s.rowmap[s]) >= RV_MAXB + RV_NBB) nn += lam[s];
        if (nn / c->dt > 100.0f) rv_emu_cnt[46] += 1; if (nn / c->dt > 1000.0f) rv_emu_cnt[47] += 1; }
      if (cap) { rv_emu_cnt[arm_rows ? 31 : 32] += 1; rv_emu_cnt[34] += (long)nr * (it + 1); if (res[x] > 10.0f * c->solver_tol) rv_emu_cnt[37] += 1; }
      else rv_emu_cnt[36] += it + 1;
    }
#endif
    for (int x = 0; x < RV_MAXB; ++x) if (((isl_rows >> x) & 1) && res[x] < (((fing || limb) && x == fisl) ? c->solver_tol : isl_tol[x])) done |= 1 << x;
    // stalled islands (rv_config.solver_stall): no new smallest residual for that many sweeps
    for (int x = 0; c->solver_stall > 0 && x < RV_MAXB; ++x) {
      if (!((isl_rows >> x) & 1) || ((done >> x) & 1)) continue;
      if (res[x] < best[x]) { best[x] = res[x]; since[x] = 0; }
      else if (++since[x] >= c->solver_stall) { done |= 1 << x; RV_CNT(7, 1) }
    }
    if ((done & isl_rows) == isl_rows) break;
    if (it == c->solver_iters - 1) { RV_CNT(30, 1) }
  }
  for (int s = 0; s < n_rows; ++s) {
    const int q = S.s.rowmap[s];
    DevMan& mm = e.man[RV_ROW_MI(q)];
    const int pi = RV_ROW_I(q), ks = RV_ROW_K(q);
    if (ks == 0) mm.ln[pi] = lam[s]; else if (ks == 1) mm.lt1[pi] = lam[s]; else mm.lt2[pi] = lam[s];
  }
  for (int X = 0; X < RV_MAXB; ++X) {
    if (!body_on(e, X)) continue;
    for (int cc = 0; cc < 6; ++cc) {
      float acc = e.body[X][7 + cc];
      for (int s = 0; s < n_rows; ++s) {
        const int q = S.s.rowmap[s];
        const int as = RV_ROW_A(q), bs = RV_ROW_B(q), ks = RV_ROW_K(q);
        if (as != X && bs != X) continue;
        const Row& Q = S.s.u.r.rows[RV_ROW_MI(q)][RV_ROW_I(q)];
        float coef;
        if (as == X) coef = cc < 3 ? Q.dir[ks][cc] * e.inv_mass[X] : Q.aa[ks][cc - 3];
        else coef = cc < 3 ? -(Q.dir[ks][cc] * e.inv_mass[X]) : -Q.ab[ks][cc - 3];
        acc = acc + coef * lam[s];
      }
      e.body[X][7 + cc] = acc;
    }
  }
  for (int m = 0; fing && m < 2; ++m) {        // the fingers move with the solved velocity
    float qd = qf0[m];
    for (int s = 0; s < n_rows; ++s) if (fi[s] == m) qd = qd + pf[s] * lam[s];
    qd = qd + pf[n_rows + m] * lam[n_rows + m];
    const int j = RV_NLIMB + m;
    float qn = e.q[j] + (qd - S.s.fing_qd0[m]) * c->dt;
    if (qn < arm->q_lo[j]) { qn = arm->q_lo[j]; qd = 0.0f; }
    if (qn > arm->q_hi[j]) { qn = arm->q_hi[j]; qd = 0.0f; }
    e.q[j] = qn; e.qd[j] = qd;
  }
  for (int x = 0; x < nlm; ++x) {              // the limb moves with the solved velocity
    float dq = dq0[x];
    for (int s = 0; s < n_all; ++s) if (la[s]) dq = dq + pj[s][x] * lam[s];
    float qd = S.s.limb_qd0[x] + dq;
    float qn = e.q[x] + dq * c->dt;
    if (qn < arm->q_lo[x]) { qn = arm->q_lo[x]; qd = 0.0f; }
    if (qn > arm->q_hi[x]) { qn = arm->q_hi[x]; qd = 0.0f; }
    e.q[x] = qn; e.qd[x] = qd;
  }
  if (limb) S.s.kin_fresh = 0;
}
#endif

#if RV_EMU_SECTION == 3      // rv_dev_env.h: RV_LANES_BEGIN
  RV_LANES_BEGIN
    if (lane < RV_NJ) {
      const DevEnv& e = S.e; int j = lane;
      float vd = 0.0f, ratio = 1.0f;
      if (e.motor_on[j]) {
        vd = e.motor_kp[j] * (e.motor_q[j] - e.q[j]) * (1.0f / c->dt);
        float raw = fabsr(vd);
        if (j < RV_NLIMB && raw > e.vmax_cmd[j]) ratio = e.vmax_cmd[j] / raw;
      }
      S.s.vdraw[j] = vd; S.s.ratio[j] = ratio;
    }
  RV_LANES_END
#endif

#if RV_EMU_SECTION == 4      // rv_dev_env.h: RV_DEV void motors_only_substeps(Shared& S, const Consts& K, const int r) {
RV_DEV void motors_only_substeps(Shared& S, const Consts& K, const int r) {
  const rv_config* c = K.cfg; const rv_arm* arm = K.arm;
  DevEnv& e = S.e;
  const float dt = c->dt;
  for (int i = 0; i < r; ++i) {
    float vdr[RV_NJ], sync = 1.0f;
    for (int j = 0; j < RV_NJ; ++j) {
      float vd = 0.0f, ratio = 1.0f;
      if (e.motor_on[j]) {
        vd = e.motor_kp[j] * (e.motor_q[j] - e.q[j]) * (1.0f / dt);
        float raw = fabsr(vd);
        if (j < RV_NLIMB && raw > e.vmax_cmd[j]) ratio = e.vmax_cmd[j] / raw;
      }
      vdr[j] = vd;
      sync = fminr(sync, ratio);
    }
    for (int j = 0; j < RV_NJ; ++j) {
      float vdd = 0.0f;
      if (e.motor_on[j]) {
        vdd = vdr[j];
        if (j < RV_NLIMB) vdd = vdd * sync;
        vdd = fclampr(vdd, -e.vmax_cmd[j], e.vmax_cmd[j]);
      }
      float dv = fclampr(vdd - e.qd[j], -arm->a_max[j] * dt, arm->a_max[j] * dt);
      float qd = e.qd[j] + dv;
      float qn = e.q[j] + qd * dt;
      if (qn < arm->q_lo[j]) { qn = arm->q_lo[j]; qd = 0.0f; }
      if (qn > arm->q_hi[j]) { qn = arm->q_hi[j]; qd = 0.0f; }
      e.q[j] = qn; e.qd[j] = qd;
      S.s.jtravel[j] += fabsr(qd) * dt;
    }
    e.sim_steps++; e.substeps_last++;
  }
}
#endif

#if RV_EMU_SECTION == 5      // rv_dev_env.h: {
  {
    DevEnv& e = S.e;
    CoastCtl C = coast_ctl_load(S, K);
    const float dt = c->dt;
    float trav[RV_NJ];
    for (int j = 0; j < RV_NJ; ++j) trav[j] = S.s.jtravel[j];
    int st = st0;
    for (;;) {
      if (st != skip_st && ctl_update_due(C, st) && !ctl_update_noop(C, st, check_joints_reached(e))) {
        RV_CNT(4, 1)
        for (int col = 0; col < RV_NCOL; ++col) {
          float Tc = 0.0f;
          for (int j = 0; j < RV_NJ; ++j) Tc = Tc + S.s.crun[col][j] * (trav[j] + (fabsr(e.qd[j]) + arm->a_max[j] * dt) * dt);
          const float D = Tc * 1.02f + 1e-4f;
          if (!(S.s.clr_t[col] > D && S.s.clr_b[col] > 2.0f * D)) pending = 1;
        }
        break;
      }
      float vdr[RV_NJ], sync = 1.0f, qn_[RV_NJ], qdn_[RV_NJ], tn_[RV_NJ];
      for (int j = 0; j < RV_NJ; ++j) {
        float vd = 0.0f, ratio = 1.0f;
        if (e.motor_on[j]) {
          vd = e.motor_kp[j] * (e.motor_q[j] - e.q[j]) * (1.0f / dt);
          float raw = fabsr(vd);
          if (j < RV_NLIMB && raw > e.vmax_cmd[j]) ratio = e.vmax_cmd[j] / raw;
        }
        vdr[j] = vd;
        sync = fminr(sync, ratio);
      }
      for (int j = 0; j < RV_NJ; ++j) {
        float vdd = 0.0f;
        if (e.motor_on[j]) {
          vdd = vdr[j];
          if (j < RV_NLIMB) vdd = vdd * sync;
          vdd = fclampr(vdd, -e.vmax_cmd[j], e.vmax_cmd[j]);
        }
        float dv = fclampr(vdd - e.qd[j], -arm->a_max[j] * dt, arm->a_max[j] * dt);
        float qd = e.qd[j] + dv;
        float qn = e.q[j] + qd * dt;
        if (qn < arm->q_lo[j]) { qn = arm->q_lo[j]; qd = 0.0f; }
        if (qn > arm->q_hi[j]) { qn = arm->q_hi[j]; qd = 0.0f; }
        qn_[j] = qn; qdn_[j] = qd; tn_[j] = trav[j] + fabsr(qd) * dt;
      }
      int out_of_reach = 1;
      for (int col = 0; col < RV_NCOL; ++col) {
        float Tc = 0.0f;
        for (int j = 0; j < RV_NJ; ++j) Tc = Tc + S.s.crun[col][j] * tn_[j];
        const float D = Tc * 1.02f + 1e-4f;
        if (!(S.s.clr_t[col] > D && S.s.clr_b[col] > 2.0f * D)) {
          out_of_reach = 0;
#ifdef RV_EMU_COUNT
          rv_emu_dbg2[col * 2 + !(S.s.clr_t[col] > D)] += 1;
          rv_emu_dbg2[20 + col] += (long)(1e6f * (!(S.s.clr_t[col] > D) ? S.s.clr_t[col] : 0.5f * S.s.clr_b[col]));
          rv_emu_dbg2[30 + col] += st - st0;
#endif
        }
      }
      if (!out_of_reach) { RV_CNT(5, 1) pending = 1; break; }
      for (int j = 0; j < RV_NJ; ++j) { e.q[j] = qn_[j]; e.qd[j] = qdn_[j]; trav[j] = tn_[j]; }
      ++st;
      if (steps_check > 0 && st % steps_check == 0 &&
          !(C.grasp ? ctl_gtick_noop(C, st, check_joints_reached(e), st - st0) : ctl_tick_noop(C, st, check_joints_reached(e)))) { pending = 2; break; }
      if (S.s.fused_n + (st - st0) >= max_n) { pending = 3; break; }
    }
    const int n = st - st0;
    for (int j = 0; j < RV_NJ; ++j) S.s.jtravel[j] = trav[j];
    if (C.grasp && C.g_phase == RV_GPHASE_START) e.num_action_steps += n - (pending == 2 ? 1 : 0);
    e.sim_steps += n; e.substeps_last += n;
    S.s.fused_n += n; S.s.fused_pending = pending;
    RV_CNT(2, 1) RV_CNT(3, n) RV_CNT(6, pending == 2)
  }
#endif

#if RV_EMU_SECTION == 6      // rv_dev_env.h: RV_LANES_BEGIN
  RV_LANES_BEGIN
    if (lane == 0) S.s.n_rows = (!fing_fast && (with_fingers || any_con)) ? 0 : solver_row_list(S, label, on_, act_, big_);
  RV_LANES_END
  float isl_tol[RV_MAXB];
  {
    const int unrest = unrest_mask(S.e);
    for (int x = 0; x < RV_MAXB; ++x) {
      int u_ = 0;
      for (int b = 0; b < RV_MAXB; ++b) if (on_[b] && label[b] == x) u_ |= (unrest >> b) & 1;
      isl_tol[x] = tol_of(c, u_);
    }
  }
  if (fing_fast) solve_rows(S, K, S.s.n_rows < 0 ? 0 : S.s.n_rows, with_fingers, limb, lone ? label[the_body] : -1, isl_tol);
  else if (S.s.n_rows > 0) solve_rows(S, K, S.s.n_rows, 0, 0, -1, isl_tol);
#endif

#if RV_EMU_SECTION == 7      // rv_dev_env.h: if (!with_fingers && !any_con && big_root >= 0) {
  if (!with_fingers && !any_con && big_root >= 0) {
    const int root = big_root;
    float big_best = 1e30f; int big_since = 0;        // rv_config.solver_stall
    for (int it = -1; it < c->solver_iters; ++it) {   // it == -1: warm start
      RV_LANES_BEGIN
        DevEnv& e = S.e;
        if (lane < RV_MAXB) {
          const int b = lane;
          float res = 0.0f;
          if (body_on(e, b) && label[b] == root) {
            BV A = ld_bv(e, b); const float ima = e.inv_mass[b];
            for (int kind = 0; kind < 2; ++kind) {
              DevMan& m = e.man[kind == 0 ? RV_TIDX(b) : RV_AIDX(b)];
              const int mi = kind == 0 ? RV_TIDX(b) : RV_AIDX(b);
              for (int i = 0; i < m.n; ++i) {
                Row r = S.s.u.r.rows[mi][i];
                Lam l; l.n = m.ln[i]; l.t1 = m.lt1[i]; l.t2 = m.lt2[i];
                if (it < 0) warm_apply(A, nullptr, ima, 0.0f, l, r);
                else { res = fmaxr(res, point_solve(A, nullptr, ima, 0.0f, l, r)); m.ln[i] = l.n; m.lt1[i] = l.t1; m.lt2[i] = l.t2; }
              }
            }
            st_bv(e, b, A);
          }
          S.s.res[b] = res;
        }
      RV_LANES_END
      for (int rd = 0; rd < 3; ++rd) {
        RV_LANES_BEGIN
          DevEnv& e = S.e;
          if (lane < 2) {
            const int x = lane;
            float res = 0.0f;
            const int k = bb_round_pair(rd, x);
            const int a_ = bb_a(k), b_ = bb_b(k);
            if (body_on(e, a_) && body_on(e, b_) && label[a_] == root && e.man[RV_BBIDX(k)].n != 0) {
              DevMan& m = e.man[RV_BBIDX(k)];
              BV A = ld_bv(e, a_), B = ld_bv(e, b_);
              const float ima = e.inv_mass[a_], imb = e.inv_mass[b_];
              for (int i = 0; i < m.n; ++i) {
                Row r = S.s.u.r.rows[RV_BBIDX(k)][i];
                Lam l; l.n = m.ln[i]; l.t1 = m.lt1[i]; l.t2 = m.lt2[i];
                if (it < 0) warm_apply(A, &B, ima, imb, l, r);
                else { res = fmaxr(res, point_solve(A, &B, ima, imb, l, r)); m.ln[i] = l.n; m.lt1[i] = l.t1; m.lt2[i] = l.t2; }
              }
              st_bv(e, a_, A); st_bv(e, b_, B);
            }
            S.s.res[4 + 2 * rd + x] = res;
          }
        RV_LANES_END
      }
      float res = 0.0f;
#pragma unroll
      for (int t = 0; t < 10; ++t) res = fmaxr(res, S.s.res[t]);
      if (it >= 0 && res < isl_tol[root]) break;
      if (it >= 0 && c->solver_stall > 0) { if (res < big_best) { big_best = res; big_since = 0; } else if (++big_since >= c->solver_stall) break; }
    }
  }

#endif

#if RV_EMU_SECTION == 8      // rv_dev_env.h: RV_LANES_END
  RV_LANES_END
  // islands go to sleep as a whole: a body sleeps when every awake body it is coupled to
  // (transitively) by manifolds that hold points is ready as well
  if (c->sleep_steps > 0) {
  RV_LANES_BEGIN
    DevEnv& e = S.e;
    if (lane < RV_MAXB) {
      int b = lane;
      int mine = 0, all = 1;
#pragma unroll
      for (int x = 0; x < RV_MAXB; ++x) if (x == b) mine = on_[x] && label[x] >= 0;
      int lb = 0;
#pragma unroll
      for (int x = 0; x < RV_MAXB; ++x) if (x == b) lb = label[x];
#pragma unroll
      for (int x = 0; x < RV_MAXB; ++x) if (on_[x] && label[x] == lb && !S.s.ready[x]) all = 0;
      if (mine && S.s.ready[b] && !e.frozen[b] && all) {
#endif


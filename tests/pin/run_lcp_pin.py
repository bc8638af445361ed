"""Runs the independent contact-solve pin (tests/pin/lcp_pin.py) over >= 200 random contact situations
and prints the table that tests/test_independent_pin.py asserts on (also: profiles/r04_lcp_pin.txt).

    python tests/pin/run_lcp_pin.py [--envs 28] [--seed 0] [--hip]

--hip (GPU box): the substep is ALSO run by librovat_hip.so (float) from the same states; its body
velocities must equal the float oracle's bit for bit and are checked against the same certificate.
"""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from robovat_amd import configs, scenes, abi   # noqa: E402
from oracle import orc                          # noqa: E402
from tests.pin import lcp_pin as L              # noqa: E402


def collect(n_envs=28, seed=0, double=True, hip=False):
    scene, names = scenes.make_scene()

    def cfg_of(**over):
        return configs.make_rv_config(env_cfg=configs.push_env_config(**over), n_envs=n_envs, seed=5 + seed, shape_names=names)
    gen = orc.OracleWorld(cfg_of(), scene, double=True)
    states, P = L.make_states(gen, n_envs, seed)
    cfg = cfg_of(**L.CONVERGED)
    W = orc.OracleWorld(cfg, scene, double=double)
    Wsh = orc.OracleWorld(cfg_of(**L.SHIPPED), scene, double=double)
    H = None
    if hip:
        import torch
        from robovat_amd import lib
        H = lib.World(cfg, scene, device=0)
    bt = 1e-12 if double else 3e-6
    recs, hip_diff = [], 0.0
    for si, St in enumerate(states):
        W.set_body_params(P); W.set_body_state(St); W.step_sub(1)
        after, mc = W.body_state(), W.manifold_counts()
        if H is not None:
            H.set_body_params(P.astype(np.float32)); H.set_body_state(St.astype(np.float32)); H.step_sub(1)
            hv = H.body_state().cpu().numpy().astype(np.float64)
            hip_diff = max(hip_diff, float(np.abs(hv[..., 7:13] - after[..., 7:13]).max()))
        # the shipped solver (50 sweeps, early exit, stall exit) on its own third warm-started substep
        Wsh.set_body_params(P); Wsh.set_body_state(St); Wsh.step_sub(2)
        St_sh = Wsh.body_state(); Wsh.step_sub(1); mc_sh = Wsh.manifold_counts()
        for e in range(n_envs):
            man = {mi: W.manifold(e, mi) for mi in range(abi.RV_NMAN) if mc[e, mi] > 0}
            if not man:
                continue
            J, Minv, us, c, mu, rows = L.build_problem(cfg, scene, St[e], P[e], man, P[e, 0, 6])
            lam_pgs = np.array([man[mi][1][i, 10:13] for mi, i in rows]).reshape(-1)
            act = np.repeat(P[e, :, 0] > 0, 6)
            got = after[e, :, 7:13].reshape(-1)
            # (1) the velocities the build reports are the ones its impulses imply
            readback = np.abs(us + Minv @ J.T @ lam_pgs - got)[act].max()
            # (2) certificate: the converged impulses solve the independently built problem
            vel, bnd = L.mcp_violation(J, Minv, us, c, mu, lam_pgs, btol=bt)
            # (3) direct linear solve on the same active set; is ITS result a solution, and the same motion?
            lam_d, u_d, cond = L.solve_active_set(J, Minv, us, c, mu, lam_pgs, btol=bt)
            vel_d, bnd_d = L.mcp_violation(J, Minv, us, c, mu, lam_d, btol=1e-9)
            direct_ok = vel_d < 1e-9 and bnd_d < 1e-12 and cond < 1e7
            diff_d = np.abs(u_d - got)[act].max()
            # (4) what the shipped early exits leave (velocity violation of its own warm-started substep)
            man_s = {mi: Wsh.manifold(e, mi) for mi in range(abi.RV_NMAN) if mc_sh[e, mi] > 0}
            vel_s = 0.0
            if man_s:
                Js, Ms, uss, cs, mus, rows_s = L.build_problem(cfg, scene, St_sh[e], P[e], man_s, P[e, 0, 6])
                lam_s = np.array([man_s[mi][1][i, 10:13] for mi, i in rows_s]).reshape(-1)
                vel_s, _ = L.mcp_violation(Js, Ms, uss, cs, mus, lam_s, btol=bt)
            nbb = sum(1 for r in rows if 4 <= r[0] < 10)
            nbody = int(sum(1 for b in range(abi.RV_MAXB) if P[e, b, 0] > 0 and mc[e, b] > 0))
            recs.append(dict(set=si, env=e, points=len(rows), bb_points=nbb, bodies=nbody, readback=readback, vel=vel, bnd=bnd,
                             direct_ok=direct_ok, diff_direct=diff_d, shipped_vel=vel_s))
    if H is not None:
        H.close()
    return recs, hip_diff


def table(recs, title):
    A = lambda k: np.array([r[k] for r in recs], dtype=np.float64)
    lines = ['# ' + title,
             'cases %d (bodies in contact per case: 1..%d; %d cases hold body-body points; %d..%d contact points per case)'
             % (len(recs), int(A('bodies').max()), int((A('bb_points') > 0).sum()), int(A('points').min()), int(A('points').max()))]

    def row(name, v, unit):
        lines.append('%-66s median %.2e  p90 %.2e  p99 %.2e  max %.2e %s' % (name, np.median(v), np.percentile(v, 90), np.percentile(v, 99), v.max(), unit))
    row('velocities reported vs implied by the impulses read back', A('readback'), 'm/s')
    row('certificate: MCP violation of the converged PGS impulses', A('vel'), 'm/s')
    row('certificate: friction-bound violation', np.maximum(A('bnd'), 0.0), 'N s')
    ok = A('direct_ok') > 0
    lines.append('direct linear solve on the same active set: well-conditioned (cond < 1e7) and itself an exact solution in %d of %d '
                 'cases (the others: redundant points -- the sliding-friction problem then has a continuum of solutions)' % (ok.sum(), len(recs)))
    if ok.any():
        row('  ... body velocities, direct solve vs PGS (those cases)', A('diff_direct')[ok], 'm/s')
    row('shipped solver (50 sweeps, 1e-5 N s exit, stall exit), 3rd warm substep', A('shipped_vel'), 'm/s')
    return lines


if __name__ == '__main__':
    ap = argparse.ArgumentParser()
    ap.add_argument('--envs', type=int, default=28)
    ap.add_argument('--seed', type=int, default=0)
    ap.add_argument('--hip', action='store_true')
    args = ap.parse_args()
    for double in (True, False):
        recs, hd = collect(args.envs, args.seed, double, hip=args.hip and not double)
        print('\n'.join(table(recs, 'oracle %s, PGS run to convergence (6000 sweeps, no early exit)' % ('float64' if double else 'float32'))))
        if args.hip and not double:
            print('librovat_hip.so, same states, same substep: max |v_hip - v_oracle_f32| = %.3e' % hd)

"""Independent numerical pin of the contact solve (the physics arithmetic of
pybullet.stepSimulation, bullet_physics.py:106-109, cannot be run here -- this is the check that
does not involve this repo's solver).

For one substep the velocity-level contact problem is a mixed complementarity problem in the
impulses lambda of the contact rows:

    normal row   0 <= lambda_n  _|_  (J u - c)_n >= 0
    friction     lambda_t = clamp to [-mu lambda_n, mu lambda_n] of the row's unconstrained solution
                 (Bullet's friction PYRAMID: two tangent rows per point, each a box row)
    u = u* + M^-1 J^T lambda

Everything on the right is built HERE in float64 numpy from the manifold points the solver was
given (anchors, normals, distances), the body poses, masses, inertias and friction coefficients:
nothing of the solver's row data is read.  The problem is solved with scipy (bounded quasi-Newton
on the box QP, outer fixed point on the friction bounds) and the body velocities it implies are
compared with the ones the build's PGS produced.  With redundant contact points the impulses are
not unique; J^T lambda -- the body velocities -- is, so that is what is compared.
"""
import numpy as np
from scipy import optimize

from robovat_amd import abi

BB = [(0, 1), (0, 2), (0, 3), (1, 2), (1, 3), (2, 3)]


def quat_mat(q):
    x, y, z, w = q
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                     [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                     [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])


def plane_space(n):
    """btPlaneSpace1 (Bullet's tangent basis for a contact normal)."""
    if abs(n[2]) > 0.7071067811865476:
        a = n[1] * n[1] + n[2] * n[2]
        k = 1.0 / np.sqrt(a)
        t1 = np.array([0.0, -n[2] * k, n[1] * k])
        t2 = np.array([a * k, -n[0] * t1[2], n[0] * t1[1]])
    else:
        a = n[0] * n[0] + n[1] * n[1]
        k = 1.0 / np.sqrt(a)
        t1 = np.array([-n[1] * k, n[0] * k, 0.0])
        t2 = np.array([-n[2] * t1[1], n[2] * t1[0], a * k])
    return t1, t2


def build_problem(cfg, scene, state0, params, manifolds, table_z, mu_table=None):
    """state0 [B,13]: poses and velocities at the START of the substep; params [B,8]; manifolds:
    {mi: (n, pts[4,13])} AFTER it (anchors are body-local: they are what the solve saw).
    Returns (J [R,6B], Minv [6B], ustar [6B], c [R], mu [R/3], rows) for the awake, active bodies."""
    B = abi.RV_MAXB
    dt = cfg.dt
    g = np.array([cfg.gravity_xy[0], cfg.gravity_xy[1], cfg.gravity_z])
    Minv = np.zeros((6 * B, 6 * B))
    ustar = np.zeros(6 * B)
    R, pos = [], []
    for b in range(B):
        if not params[b, 0]:
            R.append(np.eye(3)); pos.append(np.zeros(3)); continue
        rot = quat_mat(state0[b, 3:7])
        R.append(rot); pos.append(state0[b, :3])
        mass, sc = params[b, 3], params[b, 2]
        ik = np.array(scene.shapes[int(params[b, 1])].inertia_k[:3])
        iinv_local = 1.0 / (mass * sc * sc * ik)
        Minv[6 * b:6 * b + 3, 6 * b:6 * b + 3] = np.eye(3) / mass
        Minv[6 * b + 3:6 * b + 6, 6 * b + 3:6 * b + 6] = rot @ np.diag(iinv_local) @ rot.T
        # semi-implicit Euler: gravity, then Bullet's damping factor
        ustar[6 * b:6 * b + 3] = (state0[b, 7:10] + g * dt) * cfg.lin_damp
        ustar[6 * b + 3:6 * b + 6] = state0[b, 10:13] * cfg.ang_damp
    Jr, cr, mur, rows = [], [], [], []
    mu_t = cfg.table_friction if mu_table is None else mu_table
    for mi, (n, pts) in sorted(manifolds.items()):
        if mi < B:
            kind, a, b = 0, mi, -1
        elif mi < B + len(BB):
            kind = 1; a, b = BB[mi - B]
        else:
            kind, a, b = 2, mi - B - len(BB), -1
        for i in range(n):
            la, lb, nrm, dist = pts[i, 0:3], pts[i, 3:6], pts[i, 6:9], pts[i, 9]
            wa = pos[a] + R[a] @ la
            ra = wa - pos[a]
            t1, t2 = plane_space(nrm)
            if kind == 1:
                wb = pos[b] + R[b] @ lb
                rb = wb - pos[b]
                mub = params[b, 4]
            elif kind == 0:
                mub = mu_t                 # (the scenes of this test stay on the table: no ground contacts)
            else:
                mub = cfg.arm_friction
            for k, d in enumerate((nrm, t1, t2)):
                row = np.zeros(6 * B)
                row[6 * a:6 * a + 3] = d
                row[6 * a + 3:6 * a + 6] = np.cross(ra, d)
                if kind == 1:
                    row[6 * b:6 * b + 3] = -d
                    row[6 * b + 3:6 * b + 6] = -np.cross(rb, d)
                Jr.append(row)
                if k == 0:
                    # speculative contact for a positive distance, Baumgarte push-out otherwise
                    cr.append(-dist / dt if dist > 0 else min(cfg.erp * max(-dist - cfg.slop, 0.0) / dt, cfg.max_pushout))
                else:
                    cr.append(0.0)
            mur.append(params[a, 4] * mub)
            rows.append((mi, i))
    return np.array(Jr).reshape(-1, 6 * B), Minv, ustar, np.array(cr), np.array(mur), rows


def solve_active_set(J, Minv, ustar, c, mu, lam_ref, btol=1e-12):
    """The contact problem solved DIRECTLY, by one linear solve (scipy.linalg.lstsq, float64), on the active
    set read off a reference solution: which points carry a normal impulse, and which friction rows sit at
    +mu lambda_n / -mu lambda_n (sliding) or strictly inside (sticking).  Unknowns x: the free normal and
    the sticking friction impulses; a sliding row is +-mu times its point's normal unknown, an open point
    is zero: lambda = T x.  Equations: w = J u - c = 0 on the free normal and the sticking rows.  Nothing
    of lam_ref but the pattern is used; whether the result solves the FULL problem (signs, bounds, open rows)
    is then a question for mcp_violation().  Returns (lambda, u, cond): cond = condition number of the
    system (a rank-deficient system = redundant points: the sliding-friction problem then has a continuum of
    exact solutions with DIFFERENT body velocities, and agreement with any particular one means nothing)."""
    from scipy import linalg
    R = J.shape[0]
    if R == 0:
        return np.zeros(0), ustar.copy(), 1.0
    A = J @ Minv @ J.T
    bvec = J @ ustar - c
    cols, eqs = [], []          # unknown -> column of T ; equation rows
    T = np.zeros((R, 0))
    for p in range(len(mu)):
        ln = lam_ref[3 * p]
        if not ln > 0.0:
            continue
        col = np.zeros(R); col[3 * p] = 1.0
        lim = mu[p] * ln
        for k in (1, 2):
            lt = lam_ref[3 * p + k]
            if lt >= lim * (1 - btol):
                col[3 * p + k] = mu[p]
            elif lt <= -lim * (1 - btol):
                col[3 * p + k] = -mu[p]
        cols.append(col); eqs.append(3 * p)
        for k in (1, 2):
            lt = lam_ref[3 * p + k]
            if -lim * (1 - btol) < lt < lim * (1 - btol):
                col = np.zeros(R); col[3 * p + k] = 1.0
                cols.append(col); eqs.append(3 * p + k)
    if not cols:
        return np.zeros(R), ustar.copy(), 1.0
    T = np.array(cols).T
    K = A[eqs] @ T
    x = linalg.lstsq(K, -bvec[eqs], cond=1e-13)[0]
    lam = T @ x
    return lam, ustar + Minv @ J.T @ lam, float(np.linalg.cond(K))


def friction_box(mu, lam_n):
    """Bounds of the box problem once the normal impulses the friction pyramid hangs on are given."""
    R = 3 * len(mu)
    lo, hi = np.zeros(R), np.full(R, np.inf)
    lim = mu * lam_n
    lo[1::3] = -lim; hi[1::3] = lim; lo[2::3] = -lim; hi[2::3] = lim
    return lo, hi


def mcp_violation(J, Minv, ustar, c, mu, lam, btol=1e-12):
    """Largest violation (m/s, resp. N s for a sign) of the conditions an exact solution of the contact
    problem satisfies, for ANY candidate lambda -- a certificate that does not depend on how lambda was found:
      normal:   lambda_n >= 0, w_n >= 0, and w_n = 0 where lambda_n > 0
      friction: |lambda_t| <= mu lambda_n; w_t = 0 inside the bounds, w_t <= 0 at the upper, >= 0 at the lower bound
    with w = J u - c the row velocities after the impulses (btol: relative slack in 'at the bound', for impulses
    that were clamped in float32).  Returns (velocity violation, bound violation)."""
    u = ustar + Minv @ J.T @ lam
    w = J @ u - c
    vel = bnd = 0.0
    for p in range(len(mu)):
        ln, wn = lam[3 * p], w[3 * p]
        bnd = max(bnd, -min(ln, 0.0))
        vel = max(vel, -min(wn, 0.0), abs(wn) if ln > 0 else 0.0)
        lim = mu[p] * ln
        for k in (1, 2):
            lt, wt = lam[3 * p + k], w[3 * p + k]
            bnd = max(bnd, abs(lt) - lim)
            if lim <= 0.0:
                continue                      # no normal force, no friction: the row may slide freely
            if lt >= lim * (1 - btol):
                vel = max(vel, max(wt, 0.0))
            elif lt <= -lim * (1 - btol):
                vel = max(vel, max(-wt, 0.0))
            else:
                vel = max(vel, abs(wt))
    return vel, bnd


# ---------------------------------------------------------------- scenes --
CONVERGED = {'PHYSICS.SOLVER_ITERS': 6000, 'PHYSICS.SOLVER_TOL': 0.0, 'PHYSICS.SOLVER_STALL': 0,
             'PHYSICS.ROLLING_FRICTION': 0.0, 'PHYSICS.SLEEP_STEPS': 0}
SHIPPED = {'PHYSICS.ROLLING_FRICTION': 0.0, 'PHYSICS.SLEEP_STEPS': 0}      # 50 sweeps, 1e-5 N s early exit, stall exit


def make_states(gen, n_envs, seed):
    """Random contact situations, as [T][N][B][13] body states + body params, from a settled reset of the
    oracle world `gen` (which is only a scene generator here): bodies at rest, sliding and spinning on the
    table (1-4 of them per env), and pairs teleported into slight overlap so that body-body manifolds hold
    points (with the table points of both: coupled islands)."""
    rng = np.random.RandomState(seed)
    gen.reset()
    S0, P = gen.body_state(), gen.body_params()
    S0[..., 7:] = 0.0
    out = []
    # (a) sliding / spinning singles
    for amp in (0.0, 0.05, 0.3, 1.0):
        S = S0.copy()
        ang = rng.uniform(-np.pi, np.pi, (n_envs, abi.RV_MAXB))
        sp = amp * rng.uniform(0.2, 1.0, (n_envs, abi.RV_MAXB))
        S[..., 7] = sp * np.cos(ang); S[..., 8] = sp * np.sin(ang)
        S[..., 12] = 6.0 * amp * rng.uniform(-1, 1, (n_envs, abi.RV_MAXB))
        S[..., 10:12] = 2.0 * amp * rng.uniform(-1, 1, (n_envs, abi.RV_MAXB, 2))      # tipping
        out.append(S)
    # (b) pairs in contact: body j next to body i, closing speed up to 0.3 m/s; (c) a chain of three
    for chain in (2, 3):
        for rep in range(2):
            S = S0.copy()
            for e in range(n_envs):
                act = [b for b in range(abi.RV_MAXB) if P[e, b, 0] > 0]
                if len(act) < chain:
                    continue
                order = list(rng.permutation(act)[:chain])
                for a, b in zip(order[:-1], order[1:]):
                    th = rng.uniform(-np.pi, np.pi)
                    ra = scene_radius(gen, P[e, a]); rb = scene_radius(gen, P[e, b])
                    d = (ra + rb) * rng.uniform(0.55, 0.8)
                    S[e, b, 0] = S[e, a, 0] + d * np.cos(th); S[e, b, 1] = S[e, a, 1] + d * np.sin(th)
                    v = rng.uniform(0.0, 0.3)
                    S[e, b, 7] = -v * np.cos(th); S[e, b, 8] = -v * np.sin(th)
                    S[e, b, 12] = rng.uniform(-2, 2)
            out.append(S)
    return out, P


def scene_radius(world, prm):
    return world.scene.shapes[int(prm[1])].radius * prm[2]

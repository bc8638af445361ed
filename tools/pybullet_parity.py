"""Run-time PyBullet probe and rigid-body pose-parity harness (SURVEY.md 8c last row).

The reference's physics is the third-party wheel `pybullet==2.6.5` (requirements.txt:9); its call sites on
the path are bullet_physics.py:106-109 (stepSimulation), :143-186 (bodies), :197-249 (pose / velocity
getters).  The wheel is not part of /root/reference and cannot be installed here, so whether it exists is
asked of the machine that runs this file, every time:

    probe()            -> (module or None, status string made from what `import pybullet` actually did)
    pose_parity(...)   -> identical scenes built through createCollisionShape / createMultiBody, stepped
                          next to the FP64 oracle and the HIP path; max / p99 position and angle error
                          after 1 / 10 / 100 substeps
    time_step_simulation(...) -> baseline B1: stepSimulation on 1 and `nproc` processes

This is harness code authored here; it drives a third-party module and contains none of the reference's
Python.  bench.py, tests/test_pybullet_probe.py and tests/test_gpu_pybullet_parity.py are its callers.
"""
import importlib
import multiprocessing
import os
import time

import numpy as np


def probe():
    """Import pybullet NOW.  Returns (module or None, status).  The status string is built from the result of the import
    -- the module's version when it is there, the exception it raised when it is not -- never a constant."""
    try:
        pb = importlib.import_module('pybullet')
    except BaseException as ex:          # noqa: BLE001 -- a broken wheel may raise anything (ImportError, OSError from dlopen ...)
        return None, 'unmeasured: `import pybullet` raised %s: %s' % (type(ex).__name__, ex)
    ver = None
    try:
        ver = pb.getAPIVersion()
    except Exception as ex:              # noqa: BLE001
        ver = 'getAPIVersion raised %r' % (ex,)
    return pb, 'importable: pybullet API version %s from %s' % (ver, getattr(pb, '__file__', '?'))


# ------------------------------------------------------------------------------------------------ scene building
def _shape_hulls(scene, shape_id, scale):
    """Hull vertex arrays [n, 3] (float64, scaled) of shape template `shape_id` of an rv_scene."""
    sh = scene.shapes[int(shape_id)]
    out = []
    for h in range(sh.n_hulls):
        n = sh.n_verts[h]
        out.append(np.array([[sh.verts[h][i][k] for k in range(3)] for i in range(n)], dtype=np.float64) * float(scale))
    return out, [float(sh.inertia_k[k]) for k in range(3)]


class BulletEnv(object):
    """One env of the build's scene in one PyBullet DIRECT client: the table top as a static GEOM_BOX, every
    active movable as GEOM_MESH convex hull(s) in its centre-of-mass / principal-axes frame (which is how
    the build's shape templates are stored, robovat_amd/scenes.py) with the build's mass, principal inertia,
    lateral / rolling / spinning friction and damping, nothing put to sleep (the reference passes no
    URDF_ENABLE_SLEEPING, bullet_physics.py:173-181)."""

    def __init__(self, pb, cfg, scene, params, env_cfg=None):
        self.pb = pb
        self.cid = pb.connect(pb.DIRECT)
        c = self.cid
        pb.resetSimulation(physicsClientId=c)
        pb.setTimeStep(float(cfg.dt), physicsClientId=c)                                     # bullet_physics.py:101
        pb.setGravity(float(cfg.gravity_xy[0]), float(cfg.gravity_xy[1]), float(cfg.gravity_z), physicsClientId=c)   # simulator.py:27
        lin_damping = ang_damping = 0.04                                                      # Bullet's default
        if env_cfg is not None:
            lin_damping = float(env_cfg.PHYSICS.LINEAR_DAMPING); ang_damping = float(env_cfg.PHYSICS.ANGULAR_DAMPING)
        table_z = float(params[0][6]) if float(params[0][6]) != 0.0 else float(cfg.table_z)
        half = [float(cfg.table_half[0]), float(cfg.table_half[1]), 0.5 * float(cfg.table_thickness)]
        col = pb.createCollisionShape(pb.GEOM_BOX, halfExtents=half, physicsClientId=c)
        self.table = pb.createMultiBody(baseMass=0.0, baseCollisionShapeIndex=col,
                                        basePosition=[float(cfg.table_center[0]), float(cfg.table_center[1]), table_z - half[2]],
                                        physicsClientId=c)
        pb.changeDynamics(self.table, -1, lateralFriction=float(cfg.table_friction), rollingFriction=0.0, spinningFriction=0.0,
                          restitution=0.0, physicsClientId=c)
        self.bodies = []
        for b, p in enumerate(params):
            if float(p[0]) <= 0:
                self.bodies.append(None)
                continue
            hulls, ik = _shape_hulls(scene, p[1], p[2])
            mass, scale = float(p[3]), float(p[2])
            if len(hulls) == 1:
                col = pb.createCollisionShape(pb.GEOM_MESH, vertices=hulls[0].tolist(), physicsClientId=c)
            else:
                col = pb.createCollisionShapeArray([pb.GEOM_MESH] * len(hulls), vertices=[h.tolist() for h in hulls], physicsClientId=c)
            uid = pb.createMultiBody(baseMass=mass, baseCollisionShapeIndex=col, basePosition=[0, 0, 10.0 + b], physicsClientId=c)
            pb.changeDynamics(uid, -1, lateralFriction=float(p[4]), rollingFriction=float(cfg.rolling_friction),
                              spinningFriction=float(cfg.rolling_friction),            # body.py:229 passes spinning = rolling
                              restitution=0.0, linearDamping=lin_damping, angularDamping=ang_damping,
                              localInertiaDiagonal=[mass * k * scale * scale for k in ik],
                              activationState=pb.ACTIVATION_STATE_DISABLE_SLEEPING, physicsClientId=c)
            self.bodies.append(uid)

    def set_state(self, state):
        pb, c = self.pb, self.cid
        for uid, s in zip(self.bodies, state):
            if uid is None:
                continue
            pb.resetBasePositionAndOrientation(uid, [float(x) for x in s[0:3]], [float(x) for x in s[3:7]], physicsClientId=c)
            pb.resetBaseVelocity(uid, [float(x) for x in s[7:10]], [float(x) for x in s[10:13]], physicsClientId=c)

    def step(self, n):
        for _ in range(int(n)):
            self.pb.stepSimulation(physicsClientId=self.cid)                                # bullet_physics.py:106-109

    def get_state(self, n_slots):
        pb, c = self.pb, self.cid
        out = np.zeros((n_slots, 13))
        for b, uid in enumerate(self.bodies):
            if uid is None:
                continue
            pos, quat = pb.getBasePositionAndOrientation(uid, physicsClientId=c)           # bullet_physics.py:197-211
            lin, ang = pb.getBaseVelocity(uid, physicsClientId=c)                           # bullet_physics.py:226-249
            out[b] = list(pos) + list(quat) + list(lin) + list(ang)
        return out

    def close(self):
        try:
            self.pb.disconnect(physicsClientId=self.cid)
        except Exception:                # noqa: BLE001
            pass


def _errors(got, want, on):
    from robovat_amd.math import rotations
    perr = np.linalg.norm(got[..., :3] - want[..., :3], axis=-1)[on]
    ang = rotations.quaternion_angle(got[..., 3:7], want[..., 3:7])[on]
    return {'max_pos_m': float(perr.max()), 'p99_pos_m': float(np.percentile(perr, 99)), 'median_pos_m': float(np.median(perr)),
            'max_angle_rad': float(ang.max()), 'p99_angle_rad': float(np.percentile(ang, 99)), 'median_angle_rad': float(np.median(ang))}


def pose_parity(pb, cfg, scene, state, params, runners, horizons=(1, 10, 100), env_cfg=None):
    """Step PyBullet and every runner in `runners` (name -> object with set_body_state / step_sub / body_state, already
    holding `params`) from the identical `state` [N][B][13]; report each runner's pose error against PyBullet at
    `horizons` substeps.  No arm in the PyBullet scene: use states in which the arm is away from the bodies (the
    pose_err scene of bench.py: bodies settled on the table, shoved at 0.2 m/s, arm at its reset pose above)."""
    n, nb = state.shape[0], state.shape[1]
    on = params[:, :, 0] > 0
    envs = [BulletEnv(pb, cfg, scene, params[i], env_cfg=env_cfg) for i in range(n)]
    for i, e in enumerate(envs):
        e.set_state(state[i])
    for r in runners.values():
        r.set_body_state(state)
    out, done = {}, 0
    for h in horizons:
        for e in envs:
            e.step(h - done)
        ref = np.stack([e.get_state(nb) for e in envs])
        for name, r in runners.items():
            r.step_sub(h - done)
            got = r.body_state()
            got = got.cpu().numpy() if hasattr(got, 'cpu') else np.asarray(got)
            out.setdefault(name, {})['substeps_%d' % h] = _errors(got.astype(np.float64), ref, on)
        done = h
    for e in envs:
        e.close()
    return out


# ------------------------------------------------------------------------------------------------ baseline B1
def _time_worker(args):
    cfg_bytes, n_substeps, seed = args
    pb, status = probe()
    if pb is None:
        return None
    from robovat_amd import configs, scenes
    from oracle import orc
    scene, names = scenes.make_scene()
    cfg = configs.make_rv_config(n_envs=1, shape_names=names, seed=seed)
    w = orc.OracleWorld(cfg, scene, double=False)
    w.reset()
    env = BulletEnv(pb, cfg, scene, w.body_params()[0])
    st = w.body_state()[0]
    st[:, 7] += 0.2
    env.set_state(st)
    t0 = time.perf_counter()
    env.step(n_substeps)
    el = time.perf_counter() - t0
    env.close()
    return el


def time_step_simulation(n_substeps=5000, procs=None):
    """stepSimulation of one PushEnv scene (table + 4 shoved bodies, no arm) on 1 and `procs` processes
    (one env per process, as tools/parallel_run.py:62-78 starts them).  sim_steps/s of each."""
    pb, status = probe()
    if pb is None:
        return {'status': status}
    from oracle import orc
    procs = procs or orc.effective_cpus()
    one = _time_worker((None, n_substeps, 0))
    out = {'status': status, 'substeps': n_substeps, 'one_process_sim_steps_per_s': n_substeps / one}
    ctx = multiprocessing.get_context('spawn')
    with ctx.Pool(procs) as pool:
        t0 = time.perf_counter()
        els = pool.map(_time_worker, [(None, n_substeps, s) for s in range(procs)])
        wall = time.perf_counter() - t0
    out.update({'processes': procs, 'all_processes_sim_steps_per_s': procs * n_substeps / max(max(els), 1e-9),
                'wall_incl_startup_s': wall,
                'note': 'table + 4 convex movables shoved at 0.2 m/s, no arm (the arm is the URDF the reference does not ship)'})
    return out


if __name__ == '__main__':
    import json
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    pb_, status_ = probe()
    print(json.dumps({'pybullet': status_}))
    if pb_ is not None:
        print(json.dumps(time_step_simulation()))

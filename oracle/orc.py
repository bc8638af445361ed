"""ctypes front-end of the CPU oracle.  TEST INFRASTRUCTURE ONLY.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s
``cpu_baseline`` leg may import this module; the product package
``robovat_amd`` never does.
"""
import ctypes as C
import os
import subprocess

import numpy as np

from robovat_amd import abi

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIBS = {}


def effective_cpus():
    """Host threads this process may actually keep busy: the smallest of the CPU count, the affinity mask and the
    cgroup CPU quota (a GPU box handed out as a slice of a node reports all 256 hardware threads in os.cpu_count() but
    throttles the container to its quota -- 16 CPUs on this pool: 256 busy OpenMP threads then run 40 x slower per thread
    than 16 do)."""
    n = os.cpu_count() or 1
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except (AttributeError, OSError):
        pass
    try:
        with open('/sys/fs/cgroup/cpu.max') as f:
            quota, period = f.read().split()[:2]
        if quota != 'max':
            n = min(n, max(1, int(float(quota) / float(period) + 0.5)))
    except (OSError, ValueError):
        try:
            with open('/sys/fs/cgroup/cpu/cpu.cfs_quota_us') as f:
                q = int(f.read())
            with open('/sys/fs/cgroup/cpu/cpu.cfs_period_us') as f:
                per = int(f.read())
            if q > 0:
                n = min(n, max(1, int(q / per + 0.5)))
        except (OSError, ValueError):
            pass
    return n


def build(force=False):
    """Compile liborc_f32.so / liborc_f64.so with the Makefile next to this file."""
    targets = [os.path.join(_HERE, n) for n in ('liborc_f32.so', 'liborc_f64.so')]
    srcs = [os.path.join(_HERE, n) for n in ('rv_oracle.c', 'orc_math.h', 'orc_collide.h')]
    srcs.append(os.path.join(_HERE, '..', 'include', 'rovat.h'))
    stale = force or any(
        (not os.path.exists(t)) or any(os.path.getmtime(s) > os.path.getmtime(t) for s in srcs)
        for t in targets)
    if stale:
        subprocess.run(['make', '-C', _HERE, '-s', '-B'], check=True)
    return targets


NATIVE_FLAGS = ['-O3', '-march=native', '-std=gnu11', '-fPIC', '-ffp-contract=off', '-fno-fast-math', '-fopenmp', '-Wno-unused-function']


def _cpu_model():
    try:
        with open('/proc/cpuinfo') as f:
            for ln in f:
                if ln.startswith('model name'):
                    return ln.split(':', 1)[1].strip()
    except OSError:
        pass
    return 'unknown'


def build_native():
    """the float oracle compiled `-O3 -march=native` for the TIMED cpu legs of bench.py -- on the machine
    that runs them (a -march=native binary built in the build container need not run on the GPU box's host CPU), so it is
    compiled at first use and rebuilt when the CPU model or the sources changed.  Same sources, -ffp-contract=off and no
    fast-math as the bit-exact target: the arithmetic is that of liborc_f32.so (tests/test_oracle_native.py)."""
    out = os.path.join(_HERE, 'liborc_f32_native.so')
    stamp = out + '.host'
    srcs = [os.path.join(_HERE, n) for n in ('rv_oracle.c', 'orc_math.h', 'orc_collide.h')] + [os.path.join(_HERE, '..', 'include', 'rovat.h')]
    model = _cpu_model()
    fresh = os.path.exists(out) and os.path.exists(stamp) and open(stamp).read() == model and \
        all(os.path.getmtime(s) <= os.path.getmtime(out) for s in srcs)
    if not fresh:
        subprocess.run(['gcc'] + NATIVE_FLAGS + [os.path.join(_HERE, 'rv_oracle.c'), '-o', out, '-shared', '-lm', '-fopenmp'], check=True)
        with open(stamp, 'w') as f:
            f.write(model)
    return out


def _lib(double, native=False):
    key = 'native' if native else bool(double)
    if key not in _LIBS:
        build()
        lib = C.CDLL(build_native() if native else os.path.join(_HERE, 'liborc_f64.so' if double else 'liborc_f32.so'))
        lib.orc_create.restype = C.c_void_p
        lib.orc_create.argtypes = [C.POINTER(abi.rv_config), C.POINTER(abi.rv_scene)]
        for name in ('orc_destroy', 'orc_reset', 'orc_set_actions', 'orc_step_macro', 'orc_step_sub',
                     'orc_wait_until_stable', 'orc_rollout', 'orc_policy_random', 'orc_policy_heuristic',
                     'orc_get_body_state', 'orc_set_body_state', 'orc_get_body_params',
                     'orc_set_body_params', 'orc_get_joint_state', 'orc_set_joint_state',
                     'orc_get_link_poses', 'orc_get_env_counters', 'orc_set_joint_targets',
                     'orc_set_link_target', 'orc_compute_ik', 'orc_query_contacts',
                     'orc_get_manifold_counts', 'orc_observe', 'orc_reward',
                     'orc_get_episode_returns', 'orc_get_stats', 'orc_eval_reward',
                     'orc_eval_waypoints', 'orc_set_external_control', 'orc_motor_targets',
                     'orc_compute_ik_seeded', 'orc_set_link_path', 'orc_grip', 'orc_set_link_timeout', 'orc_set_pose_f32',
                     'orc_debug_solver_counts', 'orc_set_num_threads', 'orc_set_link_paths', 'orc_robot_ready', 'orc_get_camera', 'orc_rollout_counts', 'orc_render', 'orc_point_cloud', 'orc_set_friction', 'orc_set_constraint', 'orc_render_rgb', 'orc_set_constraint_ex', 'orc_set_max_joint_velocities'):
            getattr(lib, name).restype = None
        lib.orc_is_limb_ready.restype = C.c_int
        lib.orc_is_gripper_ready.restype = C.c_int
        lib.orc_time.restype = C.c_double
        lib.orc_eval_gjk.restype = C.c_int
        lib.orc_get_manifold.restype = C.c_int
        lib.orc_eval_wait_until_stable.restype = C.c_int
        # OpenMP would start one thread per hardware thread of the NODE; a container that is a slice of it (cgroup quota)
        # is throttled into the ground by that: as many threads as the process may keep busy
        lib.orc_set_num_threads(C.c_int(effective_cpus()))
        _LIBS[key] = lib
    return _LIBS[key]


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


class OracleWorld(object):
    """N independent envs stepped on the CPU (OpenMP over envs)."""

    def __init__(self, cfg, scene, double=False, native=False):
        assert not (double and native)
        self.lib = _lib(double, native)
        self.double = double
        self.cfg = cfg
        self.scene = scene
        self.n = cfg.n_envs
        self.G = cfg.num_goal_steps if cfg.num_goal_steps > 0 else 1
        self.h = C.c_void_p(self.lib.orc_create(C.byref(cfg), C.byref(scene)))

    def __del__(self):
        if getattr(self, 'h', None):
            self.lib.orc_destroy(self.h)
            self.h = None

    def reset(self, mask=None):
        if mask is None:
            self.lib.orc_reset(self.h, None)
        else:
            m = np.ascontiguousarray(mask, dtype=np.uint8)
            self.lib.orc_reset(self.h, _p(m))

    def set_actions(self, actions):
        a = np.ascontiguousarray(actions, dtype=np.float32).reshape(self.n, self.G, 4)
        self.lib.orc_set_actions(self.h, _p(a))

    def step_macro(self):
        self.lib.orc_step_macro(self.h)

    def rollout(self, n_steps, first_macro_index=0, auto_reset=True):
        self.lib.orc_rollout(self.h, C.c_int(n_steps), C.c_int(first_macro_index), C.c_int(int(bool(auto_reset))))

    def rollout_counts(self, counts, first_macro_index=0):
        c = np.ascontiguousarray(counts, dtype=np.int32)
        assert c.shape == (self.n,)
        self.lib.orc_rollout_counts(self.h, _p(c), C.c_int(first_macro_index))

    def step_sub(self, n):
        self.lib.orc_step_sub(self.h, C.c_int(n))

    def wait_until_stable(self, lin=0.005, ang=0.005, check_after=100, min_stable=100, max_steps=2000):
        self.lib.orc_wait_until_stable(self.h, C.c_float(lin), C.c_float(ang), C.c_int(check_after),
                                       C.c_int(min_stable), C.c_int(max_steps))

    def policy_random(self, macro_index):
        a = np.zeros((self.n, self.G, 4), dtype=np.float32)
        self.lib.orc_policy_random(self.h, C.c_int(macro_index), _p(a))
        return a

    def policy_heuristic(self, max_attempts=20000):
        a = np.zeros((self.n, self.G, 4), dtype=np.float32)
        self.lib.orc_policy_heuristic(self.h, C.c_int(max_attempts), _p(a))
        return a

    def _get(self, fn, shape, dtype=np.float64):
        a = np.zeros(shape, dtype=dtype)
        getattr(self.lib, fn)(self.h, _p(a))
        return a

    def body_state(self):
        return self._get('orc_get_body_state', (self.n, abi.RV_MAXB, 13))

    def set_body_state(self, s):
        a = np.ascontiguousarray(s, dtype=np.float64).reshape(self.n, abi.RV_MAXB, 13)
        self.lib.orc_set_body_state(self.h, _p(a))

    def body_params(self):
        return self._get('orc_get_body_params', (self.n, abi.RV_MAXB, 8))

    def set_body_params(self, p):
        a = np.ascontiguousarray(p, dtype=np.float64).reshape(self.n, abi.RV_MAXB, 8)
        self.lib.orc_set_body_params(self.h, _p(a))

    def joint_state(self):
        return self._get('orc_get_joint_state', (self.n, abi.RV_NJ, 2))

    def set_joint_state(self, s):
        a = np.ascontiguousarray(s, dtype=np.float64).reshape(self.n, abi.RV_NJ, 2)
        self.lib.orc_set_joint_state(self.h, _p(a))

    def link_poses(self):
        return self._get('orc_get_link_poses', (self.n, abi.RV_NFRAME, 7))

    def env_counters(self):
        return self._get('orc_get_env_counters', (self.n, abi.RV_NCOUNTERS), np.int32)

    def set_joint_targets(self, q):
        a = np.ascontiguousarray(q, dtype=np.float32).reshape(self.n, abi.RV_NLIMB)
        self.lib.orc_set_joint_targets(self.h, _p(a))

    def set_link_target(self, pose):
        a = np.ascontiguousarray(pose, dtype=np.float32).reshape(self.n, 7)
        self.lib.orc_set_link_target(self.h, _p(a))

    def compute_ik(self, pose):
        a = np.ascontiguousarray(pose, dtype=np.float32).reshape(self.n, 7)
        q = np.zeros((self.n, abi.RV_NLIMB), dtype=np.float64)
        self.lib.orc_compute_ik(self.h, _p(a), _p(q))
        return q

    # ---- control-logic hooks (tests/golden/gen_control_golden.py, tests/test_control_golden.py)
    def set_external_control(self, on):
        self.lib.orc_set_external_control(self.h, C.c_int(int(bool(on))))

    def set_pose_f32(self, on):
        self.lib.orc_set_pose_f32(self.h, C.c_int(int(bool(on))))

    def motor_targets(self, idx, pos):
        i = np.ascontiguousarray(idx, dtype=np.int32); q = np.ascontiguousarray(pos, dtype=np.float64)
        self.lib.orc_motor_targets(self.h, C.c_int(len(i)), _p(i), _p(q))

    def compute_ik_seeded(self, seed, pose):
        a = np.ascontiguousarray(pose, dtype=np.float64).reshape(7)
        q = np.zeros(abi.RV_NLIMB, dtype=np.float64)
        sd = None if seed is None else np.ascontiguousarray(seed, dtype=np.float64)
        self.lib.orc_compute_ik_seeded(self.h, None if sd is None else _p(sd), _p(a), _p(q))
        return q

    def set_link_path(self, poses):
        a = np.ascontiguousarray(poses, dtype=np.float32).reshape(-1, 7)
        self.lib.orc_set_link_path(self.h, C.c_int(a.shape[0]), _p(a))

    def set_link_paths(self, poses):
        a = np.ascontiguousarray(poses, dtype=np.float32)
        if a.ndim == 2:
            a = np.ascontiguousarray(np.broadcast_to(a[None], (self.n,) + a.shape))
        self.lib.orc_set_link_paths(self.h, C.c_int(a.shape[1]), _p(a))

    def set_max_joint_velocities(self, v):
        v = np.ascontiguousarray(np.broadcast_to(np.asarray(v, np.float32), (self.n, abi.RV_NLIMB)))
        self.lib.orc_set_max_joint_velocities(self.h, _p(v))

    def robot_ready(self):
        return self._get('orc_robot_ready', (self.n, 2), np.uint8)

    def camera(self):
        return self._get('orc_get_camera', (self.n, 17))

    def set_link_timeout(self, timeout):
        self.lib.orc_set_link_timeout(self.h, C.c_double(timeout))

    def grip(self, value):
        self.lib.orc_grip(self.h, C.c_float(value))

    def set_constraint(self, body, target7, frame7=None, max_force=500.0, child=-1, joint_type='fixed'):
        t = np.ascontiguousarray(target7 if target7 is not None else [0, 0, 0, 0, 0, 0, 1], dtype=np.float64)
        f = None if frame7 is None else np.ascontiguousarray(frame7, dtype=np.float64)
        self.lib.orc_set_constraint_ex(self.h, C.c_int(int(body)), C.c_int(int(child)), C.c_int({'fixed': 1, 'point2point': 2, 'prismatic': 3, 'revolute': 4}[joint_type]),
                                       None if f is None else _p(f), _p(t), C.c_double(max_force))

    def remove_constraint(self, body):
        self.lib.orc_set_constraint(self.h, C.c_int(int(body)), None, None, C.c_double(-1.0))

    def set_friction(self, mu_finger=-1.0, mu_table=-1.0):
        self.lib.orc_set_friction(self.h, C.c_double(mu_finger), C.c_double(mu_table))

    def is_limb_ready(self, env=0):
        return bool(self.lib.orc_is_limb_ready(self.h, C.c_int(env)))

    def is_gripper_ready(self, env=0):
        return bool(self.lib.orc_is_gripper_ready(self.h, C.c_int(env)))

    def time(self, env=0):
        return float(self.lib.orc_time(self.h, C.c_int(env)))

    def query_contacts(self):
        return self._get('orc_query_contacts', (self.n, 2 + abi.RV_MAXB), np.uint8)

    def manifold(self, env, mi):
        """(n, [4, 13]) points of one manifold: la, lb, nrm, dist, ln, lt1, lt2 (diagnostics)."""
        out = np.zeros((4, 13))
        n = self.lib.orc_get_manifold(self.h, C.c_int(env), C.c_int(mi), _p(out))
        return n, out

    def limb_debug(self, env=0):
        """(M [7,7], M^-1 [7,7], lo [7], hi [7]) of rv_config.limb_dynamics at the current joint state;
        the motor-row impulses of the last limb solve are left in ``last_motor_impulse``."""
        M = np.zeros((7, 7)); Mi = np.zeros((7, 7)); lohi = np.zeros(21)
        self.lib.orc_limb_debug(self.h, C.c_int(env), _p(M), _p(Mi), _p(lohi))
        self.last_motor_impulse = lohi[14:]
        return M, Mi, lohi[:7], lohi[7:14]

    def manifold_counts(self):
        return self._get('orc_get_manifold_counts', (self.n, abi.RV_NMAN), np.int32)

    def observe(self, full=False):
        pos = np.zeros((self.n, abi.RV_MAXB, 3)); mask = np.zeros((self.n, abi.RV_MAXB))
        if not full:
            self.lib.orc_observe(self.h, _p(pos), _p(mask), None, None, None, None)
            return pos, mask
        attrs = np.zeros((self.n, 5), dtype=np.int64)
        pose = np.zeros((self.n, abi.RV_MAXB, 6)); pose2d = np.zeros((self.n, abi.RV_MAXB, 3))
        ycs = np.zeros((self.n, abi.RV_MAXB, 2))
        self.lib.orc_observe(self.h, _p(pos), _p(mask), _p(attrs), _p(pose), _p(pose2d), _p(ycs))
        return {'position': pos, 'body_mask': mask, 'num_episodes': attrs[:, 0], 'num_steps': attrs[:, 1],
                'layout_id': attrs[:, 2], 'is_safe': attrs[:, 3], 'is_effective': attrs[:, 4],
                'pose': pose, 'pose2d': pose2d, 'yaw_cossin': ycs}

    def render(self, env=0):
        """Depth (eye z, 0 = nothing) and segmentation image (body index, RV_MAXB = table,
        255 = nothing) of one env, as the point-cloud oracle sees it."""
        h, w = int(self.cfg.cam_height), int(self.cfg.cam_width)
        depth = np.zeros((h, w), dtype=np.float32); seg = np.zeros((h, w), dtype=np.uint8)
        self.lib.orc_render(self.h, C.c_int(env), _p(depth), _p(seg))
        return depth, seg

    def render_rgb(self, env=0):
        h, w = int(self.cfg.cam_height), int(self.cfg.cam_width)
        rgb = np.zeros((h, w, 3), dtype=np.uint8)
        self.lib.orc_render_rgb(self.h, C.c_int(env), _p(rgb))
        return rgb

    def point_cloud(self):
        out = np.zeros((self.n, abi.RV_MAXB, int(self.cfg.num_points), 3), dtype=np.float32)
        self.lib.orc_point_cloud(self.h, _p(out))
        return out

    def reward(self):
        r = np.zeros(self.n); d = np.zeros(self.n, dtype=np.uint8)
        self.lib.orc_reward(self.h, _p(r), _p(d))
        return r, d

    def episode_returns(self):
        return self._get('orc_get_episode_returns', (self.n,))

    def solver_counts(self):
        """(island solves, sweeps, row steps) of the impulse-space / big-island solvers since the world was created."""
        out = (C.c_long * 3)()
        self.lib.orc_debug_solver_counts(self.h, out)
        return {'island_solves': int(out[0]), 'sweeps': int(out[1]), 'row_steps': int(out[2])}

    def set_num_threads(self, n):
        self.lib.orc_set_num_threads(C.c_int(int(n)))

    def stats(self):
        s = abi.rv_macro_stats()
        self.lib.orc_get_stats(self.h, C.byref(s))
        return {k: getattr(s, k) for k, _ in s._fields_}


def set_num_threads(n, double=False):
    """OpenMP threads of the stepping entry points of both builds (process-wide)."""
    for d in (False, True):
        _lib(d).orc_set_num_threads(C.c_int(int(n)))


def eval_reward(cfg, state, next_state, double=False):
    lib = _lib(double)
    s = np.ascontiguousarray(state, dtype=np.float64).reshape(abi.RV_MAXB, 2)
    n = np.ascontiguousarray(next_state, dtype=np.float64).reshape(abi.RV_MAXB, 2)
    r = C.c_double(); t = C.c_int32()
    lib.orc_eval_reward(C.byref(cfg), _p(s), _p(n), C.byref(r), C.byref(t))
    return r.value, bool(t.value)


def eval_waypoints(cfg, action, double=False):
    lib = _lib(double)
    a = np.ascontiguousarray(action, dtype=np.float32).reshape(4)
    s = np.zeros(7); e = np.zeros(7)
    lib.orc_eval_waypoints(C.byref(cfg), _p(a), _p(s), _p(e))
    return s, e


def eval_gjk(A, B, max_dist=1e9, double=False):
    lib = _lib(double)
    A = np.ascontiguousarray(A, dtype=np.float64); B = np.ascontiguousarray(B, dtype=np.float64)
    out = np.zeros(10)
    hit = lib.orc_eval_gjk(_p(A), C.c_int(len(A)), _p(B), C.c_int(len(B)), C.c_double(max_dist), _p(out))
    if not hit:
        return None
    return dict(n=out[0:3].copy(), dist=out[3], pa=out[4:7].copy(), pb=out[7:10].copy())


def eval_wait_until_stable(stable, check_after=100, min_stable=100, max_steps=2000):
    lib = _lib(False)
    a = np.ascontiguousarray(stable, dtype=np.uint8)
    assert len(a) >= max_steps
    return lib.orc_eval_wait_until_stable(_p(a), C.c_int(check_after), C.c_int(min_stable), C.c_int(max_steps))

"""Loader and thin Python binding of ``librovat_hip.so`` (``include/rovat.h``).

The library is the product: there is NO CPU fallback.  Importing this module
is cheap; ``load()`` raises ``RuntimeError`` if the shared object has not been
built (``python -c "import __graft_entry__ as g; g.build()"``) and
``World(...)`` raises if no HIP device is present.
"""
import ctypes as C
import os
import subprocess

import numpy as np

from robovat_amd import abi

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get('RV_LIB', os.path.join(_HERE, 'librovat_hip.so'))
CSRC = os.path.join(_HERE, 'csrc')
HIPCC_FLAGS = ['-O3', '-std=c++17', '--offload-arch=gfx950', '-ffp-contract=off',
               '-fPIC', '-shared']

# every symbol include/rovat.h declares (checked by tests/test_abi.py)
SYMBOLS = [
    'rv_create', 'rv_destroy', 'rv_last_error', 'rv_set_stream', 'rv_synchronize',
    'rv_num_envs', 'rv_reset', 'rv_set_actions', 'rv_step_macro', 'rv_policy_random',
    'rv_policy_heuristic', 'rv_rollout', 'rv_rollout_async', 'rv_step_sub', 'rv_wait_until_stable', 'rv_get_body_state',
    'rv_set_body_state', 'rv_get_body_params', 'rv_set_body_params',
    'rv_get_joint_state', 'rv_set_joint_state', 'rv_get_link_poses',
    'rv_get_env_counters', 'rv_set_joint_targets', 'rv_set_link_target',
    'rv_compute_ik', 'rv_query_contacts', 'rv_get_manifold_counts', 'rv_observe',
    'rv_reward', 'rv_get_episode_returns', 'rv_get_stats', 'rv_last_kernel_ms',
    'rv_reset_targets', 'rv_get_state_ptrs', 'rv_source_hash', 'rv_set_motor_targets', 'rv_grip',
    'rv_rollout_record', 'rv_render', 'rv_set_gravity', 'rv_rollout_record_full', 'rv_step_begin', 'rv_step_poll', 'rv_set_constraint', 'rv_render_rgb', 'rv_set_friction', 'rv_set_auto_reset',
    'rv_set_constraint_ex', 'rv_set_link_path', 'rv_get_robot_ready', 'rv_get_camera', 'rv_set_max_joint_velocities',
]

_EXC = {abi.RV_ERR_VALUE: ValueError, abi.RV_ERR_STATE: RuntimeError,
        abi.RV_ERR_HIP: RuntimeError, abi.RV_ERR_NOTIMPL: NotImplementedError}
_lib = None


SOURCES = [os.path.join(CSRC, n) for n in sorted(os.listdir(CSRC)) if n.endswith(('.hip', '.h'))]
SOURCES.append(os.path.join(_HERE, '..', 'include', 'rovat.h'))


def source_hash():
    """sha256 over the sources librovat_hip.so is compiled from (file names and bytes)."""
    import hashlib
    h = hashlib.sha256()
    for path in SOURCES:
        h.update(os.path.basename(path).encode())
        with open(path, 'rb') as f:
            h.update(f.read())
    return h.hexdigest()


def built_source_hash():
    """The hash baked into the loaded binary (rv_source_hash)."""
    lib = load()
    return lib.rv_source_hash().decode()


def build(force=False, verbose=False):
    """Compile csrc/rv_kernels.hip for gfx950 into librovat_hip.so (in-tree).
    The sha256 of the sources is baked into the binary (rv_source_hash)."""
    if (not force and os.path.exists(LIB_PATH) and
            all(os.path.getmtime(d) <= os.path.getmtime(LIB_PATH) for d in SOURCES)):
        return LIB_PATH
    hipcc = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
    compile_lib(LIB_PATH, hipcc=hipcc, verbose=verbose)
    return LIB_PATH


def compile_lib(out, extra=(), hipcc='/opt/rocm/bin/hipcc', verbose=False):
    """The two translation units (rv_kernels.hip: C ABI + the register-rich env kernel; rv_kernels_occ2.hip: the
    env kernel for two waves per SIMD) compiled side by side, then linked into one shared object.  Objects and the
    linked library are made in a private temporary directory next to `out` and the library is moved into place with
    one rename: concurrent builds (the multi-rank tests) never see each other's half-written files, and a failed
    translation unit leaves neither a running compiler nor objects behind."""
    import shutil
    import tempfile
    srcs = [os.path.join(CSRC, 'rv_kernels.hip'), os.path.join(CSRC, 'rv_kernels_occ2.hip')]
    flags = [f for f in HIPCC_FLAGS if f != '-shared'] + ['-DRV_SOURCE_HASH="%s"' % source_hash()] + list(extra)
    tmp = tempfile.mkdtemp(prefix='.build_', dir=os.path.dirname(os.path.abspath(out)))
    procs = []
    try:
        objs = []
        for src in srcs:
            obj = os.path.join(tmp, os.path.basename(src) + '.o')
            cmd = [hipcc] + flags + ['-c', src, '-o', obj]
            if verbose:
                print(' '.join(cmd))
            procs.append(subprocess.Popen(cmd)); objs.append(obj)
        codes = [p.wait() for p in procs]          # (every compiler is waited for before an error is raised)
        if any(codes):
            raise subprocess.CalledProcessError(next(c for c in codes if c), 'hipcc')
        linked = os.path.join(tmp, os.path.basename(out))
        cmd = [hipcc, '--offload-arch=gfx950', '-fPIC', '-shared'] + objs + ['-o', linked]
        if verbose:
            print(' '.join(cmd))
        subprocess.run(cmd, check=True)
        os.replace(linked, out)
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill(); p.wait()
        shutil.rmtree(tmp, ignore_errors=True)


def load():
    """dlopen librovat_hip.so; fails loudly when it is missing."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            'librovat_hip.so is not built (%s). Run __graft_entry__.build(); '
            'there is no CPU fallback.' % LIB_PATH)
    lib = C.CDLL(LIB_PATH)
    lib.rv_last_error.restype = C.c_char_p
    lib.rv_source_hash.restype = C.c_char_p
    lib.rv_create.argtypes = [C.POINTER(abi.rv_config), C.POINTER(abi.rv_scene),
                              C.c_int, C.POINTER(C.c_void_p)]
    for name in SYMBOLS:
        fn = getattr(lib, name)
        if name not in ('rv_last_error', 'rv_source_hash'):
            fn.restype = C.c_int
    vp, i32, f32 = C.c_void_p, C.c_int32, C.c_float
    lib.rv_destroy.argtypes = [vp]
    lib.rv_set_stream.argtypes = [vp, vp]
    lib.rv_synchronize.argtypes = [vp]
    lib.rv_num_envs.argtypes = [vp]
    lib.rv_reset.argtypes = [vp, vp]
    lib.rv_step_macro.argtypes = [vp]
    lib.rv_step_sub.argtypes = [vp, i32]
    lib.rv_step_begin.argtypes = [vp, vp, vp]
    lib.rv_step_poll.argtypes = [vp, i32, i32, vp, C.POINTER(abi.rv_obs_buffers), vp, vp]
    lib.rv_rollout.argtypes = [vp, i32, i32, i32, vp, vp]
    lib.rv_rollout_async.argtypes = [vp, i32, i32, vp]
    lib.rv_rollout_record.argtypes = [vp, i32, i32, i32, vp, vp, C.POINTER(abi.rv_obs_buffers)]
    lib.rv_rollout_record_full.argtypes = [vp, i32, i32, i32, vp, vp, C.POINTER(abi.rv_obs_buffers), C.POINTER(abi.rv_rollout_extra)]
    lib.rv_wait_until_stable.argtypes = [vp, f32, f32, i32, i32, i32]
    lib.rv_policy_random.argtypes = [vp, i32, vp]
    lib.rv_policy_heuristic.argtypes = [vp, i32, vp]
    lib.rv_observe.argtypes = [vp, C.POINTER(abi.rv_obs_buffers)]
    lib.rv_reward.argtypes = [vp, vp, vp]
    lib.rv_compute_ik.argtypes = [vp, vp, vp]
    lib.rv_get_stats.argtypes = [vp, C.POINTER(abi.rv_macro_stats)]
    lib.rv_last_kernel_ms.argtypes = [vp, C.POINTER(C.c_float)]
    lib.rv_reset_targets.argtypes = [vp]
    lib.rv_render.argtypes = [vp, vp, vp]
    lib.rv_render_rgb.argtypes = [vp, vp]
    lib.rv_set_motor_targets.argtypes = [vp, vp, vp]
    lib.rv_grip.argtypes = [vp, f32]
    lib.rv_set_gravity.argtypes = [vp, C.POINTER(C.c_float)]
    lib.rv_set_friction.argtypes = [vp, f32, f32]
    lib.rv_set_auto_reset.argtypes = [vp, i32]
    lib.rv_set_constraint.argtypes = [vp, i32, C.POINTER(C.c_float), C.POINTER(C.c_float), f32]
    lib.rv_set_constraint_ex.argtypes = [vp, i32, i32, i32, C.POINTER(C.c_float), C.POINTER(C.c_float), f32]
    lib.rv_get_state_ptrs.argtypes = [vp, C.POINTER(abi.rv_state_view)]
    lib.rv_set_joint_targets.argtypes = [vp, vp, f32, f32]
    lib.rv_set_link_target.argtypes = [vp, vp, f32, f32]
    lib.rv_set_link_path.argtypes = [vp, vp, i32, f32, f32]
    lib.rv_get_robot_ready.argtypes = [vp, vp]
    lib.rv_set_max_joint_velocities.argtypes = [vp, vp]
    lib.rv_get_camera.argtypes = [vp, vp]
    for name in ('rv_set_actions', 'rv_get_body_state', 'rv_set_body_state',
                 'rv_get_body_params', 'rv_set_body_params', 'rv_get_joint_state',
                 'rv_set_joint_state', 'rv_get_link_poses', 'rv_get_env_counters',
                 'rv_query_contacts', 'rv_get_manifold_counts', 'rv_get_episode_returns'):
        getattr(lib, name).argtypes = [vp, vp]
    _lib = lib
    return lib


def check(status):
    if status != abi.RV_OK:
        msg = load().rv_last_error().decode('utf-8', 'replace')
        raise _EXC.get(status, RuntimeError)(msg)


class World(object):
    """N batched envs on one GPU.  Buffers are torch tensors on that device;
    the C ABI only ever sees their raw ``data_ptr()``."""

    def __init__(self, cfg, scene, device=0):
        import torch
        self.torch = torch
        self.lib = load()
        self.cfg = cfg
        self.scene = scene
        self.n = int(cfg.n_envs)
        self.G = cfg.num_goal_steps if cfg.num_goal_steps > 0 else 1
        self.device_index = int(device)
        self.device = torch.device('cuda', self.device_index)
        self.h = C.c_void_p()
        check(self.lib.rv_create(C.byref(cfg), C.byref(scene), self.device_index, C.byref(self.h)))
        # run on torch's current stream so torch ops and kernels are ordered
        self.use_current_stream()

    def use_current_stream(self):
        stream = self.torch.cuda.current_stream(self.device)
        check(self.lib.rv_set_stream(self.h, C.c_void_p(stream.cuda_stream)))

    def close(self):
        if getattr(self, 'h', None) and self.h:
            self.lib.rv_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- helpers
    def _new(self, shape, dtype):
        return self.torch.empty(shape, dtype=dtype, device=self.device)

    def _in(self, x, shape, dtype):
        t = self.torch.as_tensor(x, dtype=dtype, device=self.device).reshape(shape).contiguous()
        return t

    @staticmethod
    def _ptr(t):
        return C.c_void_p(t.data_ptr())

    def synchronize(self):
        check(self.lib.rv_synchronize(self.h))

    # -- env stepping
    def reset(self, mask=None):
        if mask is None:
            check(self.lib.rv_reset(self.h, None))
        else:
            m = self._in(mask, (self.n,), self.torch.uint8)
            check(self.lib.rv_reset(self.h, self._ptr(m)))

    def set_actions(self, actions):
        a = self._in(actions, (self.n, self.G, 4), self.torch.float32)
        check(self.lib.rv_set_actions(self.h, self._ptr(a)))

    def step_macro(self):
        check(self.lib.rv_step_macro(self.h))

    def step_begin(self, actions, mask=None):
        """rv_step_begin: the next action of the envs flagged in ``mask`` (uint8 [N], None = all)."""
        a = self._in(actions, (self.n, self.G, 4), self.torch.float32)
        m = None if mask is None else self._in(mask, (self.n,), self.torch.uint8)
        check(self.lib.rv_step_begin(self.h, self._ptr(a), None if m is None else self._ptr(m)))

    def set_auto_reset(self, on=True):
        """rv_set_auto_reset: step_begin on a finished episode resets the env; the next poll returns the reset observation."""
        check(self.lib.rv_set_auto_reset(self.h, int(bool(on))))

    def step_poll(self, max_substeps=0, max_usec=0, out=None):
        """rv_step_poll: advance the stepping envs within the budget; returns uint8 [N], 1 = this
        env's env.step() completed in this launch.  ``out`` (from ``poll_buffers``): the observation,
        reward and done of the envs that finished are written to their rows."""
        f = self._new((self.n,), self.torch.uint8)
        if out is None:
            check(self.lib.rv_step_poll(self.h, int(max_substeps), int(max_usec), self._ptr(f), None, None, None))
        else:
            check(self.lib.rv_step_poll(self.h, int(max_substeps), int(max_usec), self._ptr(f), C.byref(out['_buffers']),
                                        self._ptr(out['reward']), self._ptr(out['done'])))
        return f

    def poll_buffers(self, point_cloud=True, pose_modes=False):
        """Persistent [N]-row buffers for ``step_poll(out=...)``: {'obs': dict, 'reward', 'done'}."""
        obs, b = self._obs_buffers((self.n,), point_cloud, pose_modes)
        return {'obs': obs, '_buffers': b, 'reward': self.torch.zeros((self.n,), dtype=self.torch.float32, device=self.device),
                'done': self.torch.zeros((self.n,), dtype=self.torch.uint8, device=self.device)}

    def rollout(self, n_steps, first_macro_index=0, auto_reset=True, record=False):
        """n_steps x (RandomPolicy action -> env.step) per env in one launch."""
        r = d = None
        if record:
            r = self.torch.zeros((int(n_steps), self.n), dtype=self.torch.float32, device=self.device)
            d = self.torch.ones((int(n_steps), self.n), dtype=self.torch.uint8, device=self.device)
        check(self.lib.rv_rollout(self.h, int(n_steps), int(first_macro_index), int(bool(auto_reset)),
                                  self._ptr(r) if record else None, self._ptr(d) if record else None))
        return r, d

    def _obs_buffers(self, lead, point_cloud, pose_modes):
        """Zeroed observation tensors with leading shape ``lead`` + the rv_obs_buffers of their pointers."""
        t = self.torch

        def z(shape, dtype):
            return t.zeros(tuple(lead) + shape, dtype=dtype, device=self.device)
        out = {
            'position': z((abi.RV_MAXB, 3), t.float32), 'body_mask': z((abi.RV_MAXB,), t.float32),
            'num_episodes': z((), t.int64), 'num_steps': z((), t.int64), 'layout_id': z((), t.int64),
            'is_safe': z((), t.int64), 'is_effective': z((), t.int64),
        }
        b = abi.rv_obs_buffers()
        b.d_position = out['position'].data_ptr(); b.d_body_mask = out['body_mask'].data_ptr()
        b.d_num_episodes = out['num_episodes'].data_ptr(); b.d_num_steps = out['num_steps'].data_ptr()
        b.d_layout_id = out['layout_id'].data_ptr(); b.d_is_safe = out['is_safe'].data_ptr()
        b.d_is_effective = out['is_effective'].data_ptr()
        if point_cloud:
            out['point_cloud'] = z((abi.RV_MAXB, int(self.cfg.num_points), 3), t.float32)
            b.d_point_cloud = out['point_cloud'].data_ptr()
        if pose_modes:      # the other PoseObs modalities (pose_obs.py:53-73)
            out['pose'] = z((abi.RV_MAXB, 6), t.float32)
            out['pose2d'] = z((abi.RV_MAXB, 3), t.float32)
            out['yaw_cossin'] = z((abi.RV_MAXB, 2), t.float32)
            b.d_pose = out['pose'].data_ptr(); b.d_pose2d = out['pose2d'].data_ptr()
            b.d_yaw_cossin = out['yaw_cossin'].data_ptr()
        return out, b

    def rollout_record(self, n_steps, first_macro_index=0, auto_reset=True, point_cloud=True, pose_modes=False):
        """The rollout returning what every env.step() returns: per-step observation rows
        [n_steps, N, ...], rewards and dones (rv_rollout_record)."""
        k = int(n_steps)
        obs, b = self._obs_buffers((k, self.n), point_cloud, pose_modes)
        r = self.torch.zeros((k, self.n), dtype=self.torch.float32, device=self.device)
        d = self.torch.ones((k, self.n), dtype=self.torch.uint8, device=self.device)
        check(self.lib.rv_rollout_record(self.h, k, int(first_macro_index), int(bool(auto_reset)),
                                         self._ptr(r), self._ptr(d), C.byref(b)))
        return obs, r, d

    def rollout_record_full(self, n_steps, first_macro_index=0, auto_reset=True, point_cloud=True, pose_modes=False):
        """rollout_record plus what a complete episode record needs (rv_rollout_record_full): returns
        (obs, rewards, dones, extra) with extra = {'actions' [K, N, G, 4], 'reset' [K, N] uint8,
        'reset_obs' dict of [K, N, ...] observations env.reset() returned before the steps that
        an auto-reset preceded (zero rows elsewhere)}; feed io.hdf5_utils.episodes_from_rollout."""
        k = int(n_steps)
        obs, b = self._obs_buffers((k, self.n), point_cloud, pose_modes)
        robs, rb = self._obs_buffers((k, self.n), point_cloud, pose_modes)
        r = self.torch.zeros((k, self.n), dtype=self.torch.float32, device=self.device)
        d = self.torch.ones((k, self.n), dtype=self.torch.uint8, device=self.device)
        acts = self.torch.zeros((k, self.n, self.G, 4), dtype=self.torch.float32, device=self.device)
        rst = self.torch.zeros((k, self.n), dtype=self.torch.uint8, device=self.device)
        ex = abi.rv_rollout_extra()
        ex.d_actions = acts.data_ptr(); ex.d_reset = rst.data_ptr(); ex.reset_obs = rb
        check(self.lib.rv_rollout_record_full(self.h, k, int(first_macro_index), int(bool(auto_reset)),
                                              self._ptr(r), self._ptr(d), C.byref(b), C.byref(ex)))
        return obs, r, d, {'actions': acts, 'reset': rst, 'reset_obs': robs}

    def rollout_async(self, total_env_steps, first_macro_index=0):
        """total_env_steps x (RandomPolicy action -> env.step) shared by all envs: every env
        keeps stepping (auto-reset) while the pool lasts.  Returns steps taken per env."""
        taken = self._new((self.n,), self.torch.int32)
        check(self.lib.rv_rollout_async(self.h, int(total_env_steps), int(first_macro_index), self._ptr(taken)))
        return taken

    def step_sub(self, n):
        check(self.lib.rv_step_sub(self.h, int(n)))

    def wait_until_stable(self, lin=0.005, ang=0.005, check_after=100, min_stable=100, max_steps=2000):
        check(self.lib.rv_wait_until_stable(self.h, lin, ang, check_after, min_stable, max_steps))

    def policy_random(self, macro_index):
        a = self._new((self.n, self.G, 4), self.torch.float32)
        check(self.lib.rv_policy_random(self.h, int(macro_index), self._ptr(a)))
        return a

    def policy_heuristic(self, max_attempts=20000):
        a = self._new((self.n, self.G, 4), self.torch.float32)
        check(self.lib.rv_policy_heuristic(self.h, int(max_attempts), self._ptr(a)))
        return a

    # -- state
    def _get(self, fn, shape, dtype):
        t = self._new(shape, dtype)
        check(getattr(self.lib, fn)(self.h, self._ptr(t)))
        return t

    def body_state(self):
        return self._get('rv_get_body_state', (self.n, abi.RV_MAXB, 13), self.torch.float32)

    def set_body_state(self, s):
        check(self.lib.rv_set_body_state(self.h, self._ptr(self._in(s, (self.n, abi.RV_MAXB, 13), self.torch.float32))))

    def body_params(self):
        return self._get('rv_get_body_params', (self.n, abi.RV_MAXB, 8), self.torch.float32)

    def set_body_params(self, p):
        check(self.lib.rv_set_body_params(self.h, self._ptr(self._in(p, (self.n, abi.RV_MAXB, 8), self.torch.float32))))

    def joint_state(self):
        return self._get('rv_get_joint_state', (self.n, abi.RV_NJ, 2), self.torch.float32)

    def set_joint_state(self, s):
        check(self.lib.rv_set_joint_state(self.h, self._ptr(self._in(s, (self.n, abi.RV_NJ, 2), self.torch.float32))))

    def link_poses(self):
        return self._get('rv_get_link_poses', (self.n, abi.RV_NFRAME, 7), self.torch.float32)

    def env_counters(self):
        return self._get('rv_get_env_counters', (self.n, abi.RV_NCOUNTERS), self.torch.int32)

    def set_joint_targets(self, q, timeout=0.0, threshold=0.0):
        """timeout / threshold <= 0: the robot config's LIMB_TIMEOUT / LIMB_POSITION_THRESHOLD."""
        check(self.lib.rv_set_joint_targets(self.h, self._ptr(self._in(q, (self.n, abi.RV_NLIMB), self.torch.float32)),
                                            float(timeout), float(threshold)))

    def set_link_target(self, pose, timeout=0.0, threshold=0.0):
        check(self.lib.rv_set_link_target(self.h, self._ptr(self._in(pose, (self.n, 7), self.torch.float32)),
                                          float(timeout), float(threshold)))

    def set_link_path(self, poses, timeout=0.0, threshold=0.0):
        """poses [N, n_poses, 7] (or [n_poses, 7]: the same path for every env): SawyerSim.move_along_gripper_path
        -> ControllableBody.set_target_link_poses."""
        p = self.torch.as_tensor(poses, dtype=self.torch.float32, device=self.device)
        if p.dim() == 2:
            p = p[None].expand(self.n, -1, -1)
        n_poses = int(p.shape[1])
        p = p.reshape(self.n, n_poses, 7).contiguous()
        check(self.lib.rv_set_link_path(self.h, self._ptr(p), n_poses, float(timeout), float(threshold)))

    def set_max_joint_velocities(self, vmax):
        """rv_set_max_joint_velocities: speed limits [N, 7] (or [7]) of the limb joints for the targets being followed."""
        v = self._in(np.broadcast_to(np.asarray(vmax, np.float32), (self.n, abi.RV_NLIMB)).copy(), (self.n, abi.RV_NLIMB), self.torch.float32)
        check(self.lib.rv_set_max_joint_velocities(self.h, self._ptr(v)))

    def robot_ready(self):
        """[N, 2] uint8: (is_limb_ready, is_gripper_ready) -- retires finished targets like the reference's query."""
        return self._get('rv_get_robot_ready', (self.n, 2), self.torch.uint8)

    def reset_targets(self):
        check(self.lib.rv_reset_targets(self.h))

    def set_motor_targets(self, q, mask=None):
        """position_control_array: POSITION_CONTROL targets for the masked joints."""
        qt = self._in(q, (self.n, abi.RV_NJ), self.torch.float32)
        mt = None if mask is None else self._in(mask, (self.n, abi.RV_NJ), self.torch.uint8)
        check(self.lib.rv_set_motor_targets(self.h, self._ptr(qt), None if mt is None else self._ptr(mt)))

    def grip(self, value):
        check(self.lib.rv_grip(self.h, float(value)))

    JOINT_TYPES = {'revolute': 0, 'prismatic': 1, 'fixed': 4, 'point2point': 5}      # pybullet.JOINT_REVOLUTE / _PRISMATIC / _FIXED / _POINT2POINT: the reference's JOINT_TYPES_MAPPING

    def set_constraint(self, body, target7, frame7=None, max_force=500.0, child=-1, joint_type='fixed'):
        """rv_set_constraint_ex: tie frame7 (in the body frame; None = the body frame) of movable body ``body`` to
        the frame target7 -- of the world (child = -1) or, given in its frame, of movable body ``child`` or of frame f of the arm
        (child = abi.RV_CHILD_LINK(f): fixed / point2point only) -- by a
        'fixed', a 'point2point' or a 'prismatic' joint (sliding along the x axis of the target frame) with at most
        max_force N per row; max_force < 0 removes it."""
        if joint_type not in self.JOINT_TYPES:
            raise NotImplementedError("joint types built: 'fixed', 'point2point', 'prismatic', 'revolute' (not %r)" % (joint_type,))
        t = (C.c_float * 7)(*[float(x) for x in (target7 if target7 is not None else [0, 0, 0, 0, 0, 0, 1])])
        f = None if frame7 is None else (C.c_float * 7)(*[float(x) for x in frame7])
        check(self.lib.rv_set_constraint_ex(self.h, int(body), int(child), self.JOINT_TYPES[joint_type], f, t, float(max_force)))

    def remove_constraint(self, body):
        check(self.lib.rv_set_constraint(self.h, int(body), None, None, -1.0))

    def set_friction(self, mu_finger=-1.0, mu_table=-1.0):
        """rv_set_friction: lateral friction of the finger-tip pads / the table top (negative: unchanged)."""
        check(self.lib.rv_set_friction(self.h, float(mu_finger), float(mu_table)))

    def set_gravity(self, gravity):
        g = (C.c_float * 3)(float(gravity[0]), float(gravity[1]), float(gravity[2]))
        check(self.lib.rv_set_gravity(self.h, g))

    def state_view(self):
        """Zero-copy READ-ONLY torch views of the resident env blocks (rv_get_state_ptrs)."""
        v = abi.rv_state_view()
        check(self.lib.rv_get_state_ptrs(self.h, C.byref(v)))
        words = int(v.env_stride_bytes) // 4

        class _Raw(object):
            pass
        raw = _Raw()
        raw.__cuda_array_interface__ = {'shape': (self.n, words), 'typestr': '<f4', 'data': (int(v.d_envs), False),
                                        'version': 2, 'strides': None}
        t = self.torch
        blocks = t.as_tensor(raw, device=self.device)

        def f(off, count, shape):
            return blocks[:, int(off) // 4: int(off) // 4 + count].reshape((self.n,) + shape)
        return {
            'body': f(v.off_body, abi.RV_MAXB * 13, (abi.RV_MAXB, 13)),
            'active': f(v.off_active, abi.RV_MAXB, (abi.RV_MAXB,)).view(t.int32),
            'joint_q': f(v.off_joint_q, abi.RV_NJ, (abi.RV_NJ,)),
            'joint_qd': f(v.off_joint_qd, abi.RV_NJ, (abi.RV_NJ,)),
            'link_pos': f(v.off_link_pos, abi.RV_NFRAME * 3, (abi.RV_NFRAME, 3)),
            'link_quat': f(v.off_link_quat, abi.RV_NFRAME * 4, (abi.RV_NFRAME, 4)),
            'obs_pos': f(v.off_obs_pos, abi.RV_MAXB * 3, (abi.RV_MAXB, 3)),
            'table_z': f(v.off_table_z, 1, ()),
        }

    def compute_ik(self, pose):
        p = self._in(pose, (self.n, 7), self.torch.float32)
        q = self._new((self.n, abi.RV_NLIMB), self.torch.float32)
        check(self.lib.rv_compute_ik(self.h, self._ptr(p), self._ptr(q)))
        return q

    def camera(self):
        """[N, 17] float32: fx, fy, cx, cy, skew, rotation (row-major), translation -- the calibration each env is observed
        with (rv_config's plus the noise of its last reset, KINECT2.DEPTH.*_NOISE)."""
        return self._get('rv_get_camera', (self.n, 17), self.torch.float32)

    def query_contacts(self):
        return self._get('rv_query_contacts', (self.n, 2 + abi.RV_MAXB), self.torch.uint8)

    def manifold_counts(self):
        return self._get('rv_get_manifold_counts', (self.n, abi.RV_NMAN), self.torch.int32)

    def observe(self, point_cloud=False, pose_modes=False):
        out, b = self._obs_buffers((self.n,), point_cloud, pose_modes)
        check(self.lib.rv_observe(self.h, C.byref(b)))
        return out

    def render(self, segmask=True):
        """Depth image [N, H, W] (eye z, 0 = nothing hit) and segmentation mask of the simulated camera."""
        h, w = int(self.cfg.cam_height), int(self.cfg.cam_width)
        depth = self._new((self.n, h, w), self.torch.float32)
        seg = self._new((self.n, h, w), self.torch.uint8) if segmask else None
        check(self.lib.rv_render(self.h, self._ptr(depth), self._ptr(seg) if segmask else None))
        return depth, seg

    def render_rgb(self):
        """rv_render_rgb: uint8 [N, H, W, 3]."""
        rgb = self._new((self.n, int(self.cfg.cam_height), int(self.cfg.cam_width), 3), self.torch.uint8)
        check(self.lib.rv_render_rgb(self.h, self._ptr(rgb)))
        return rgb

    def reward(self):
        r = self._new((self.n,), self.torch.float32)
        d = self._new((self.n,), self.torch.uint8)
        check(self.lib.rv_reward(self.h, self._ptr(r), self._ptr(d)))
        return r, d

    def episode_returns(self):
        return self._get('rv_get_episode_returns', (self.n,), self.torch.float32)

    def stats(self):
        s = abi.rv_macro_stats()
        check(self.lib.rv_get_stats(self.h, C.byref(s)))
        return {k: getattr(s, k) for k, _ in s._fields_}

    def last_kernel_ms(self):
        ms = C.c_float()
        check(self.lib.rv_last_kernel_ms(self.h, C.byref(ms)))
        return float(ms.value)

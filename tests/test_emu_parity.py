"""The kernel program (robovat_amd/csrc/rv_dev_env.h) compiled for the host by
the lane emulator (tests/emu) must match the float oracle bit for bit.  This
is the CPU-side check of the kernel's lane/phase decomposition; the real HIP
parity tests are tests/test_gpu_parity.py (-m gpu)."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

from robovat_amd import abi, configs, scenes

EMU_DIR = os.path.join(os.path.dirname(__file__), 'emu')


@pytest.fixture(scope='module')
def emu():
    so = os.path.join(EMU_DIR, 'librv_emu.so')
    src = os.path.join(EMU_DIR, 'rv_emu.cpp')
    csrc = os.path.join(EMU_DIR, '..', '..', 'robovat_amd', 'csrc')
    deps = [src, os.path.join(EMU_DIR, 'rv_emu_hooks.h')] + [os.path.join(csrc, n) for n in ('rv_dev_env.h', 'rv_dev_collide.h', 'rv_dev_math.h')]
    if not os.path.exists(so) or any(os.path.getmtime(d) > os.path.getmtime(so) for d in deps):
        subprocess.run(['g++', '-O2', '-std=c++17', '-fPIC', '-ffp-contract=off', '-mfma', '-fopenmp', '-shared', src, '-o', so], check=True)
    lib = C.CDLL(so)
    lib.emu_create.restype = C.c_void_p
    lib.emu_create.argtypes = [C.POINTER(abi.rv_config), C.POINTER(abi.rv_scene)]
    return lib


class Emu(object):
    def __init__(self, lib, cfg, scene):
        self.lib, self.n = lib, cfg.n_envs
        self.h = C.c_void_p(lib.emu_create(C.byref(cfg), C.byref(scene)))

    def _get(self, fn, shape, dt):
        a = np.zeros(shape, dt); getattr(self.lib, fn)(self.h, a.ctypes.data_as(C.c_void_p)); return a

    def body_state(self): return self._get('emu_get_body_state', (self.n, abi.RV_MAXB, 13), np.float32)
    def joint_state(self): return self._get('emu_get_joint_state', (self.n, abi.RV_NJ, 2), np.float32)
    def counters(self): return self._get('emu_get_env_counters', (self.n, abi.RV_NCOUNTERS), np.int32)
    def manifolds(self): return self._get('emu_get_manifold_counts', (self.n, abi.RV_NMAN), np.int32)
    def link_poses(self): return self._get('emu_get_link_poses', (self.n, abi.RV_NFRAME, 7), np.float32)


def _check(e, ref):
    assert np.array_equal(e.body_state(), ref.body_state().astype(np.float32))
    assert np.array_equal(e.joint_state(), ref.joint_state().astype(np.float32))
    assert np.array_equal(e.counters(), ref.env_counters())
    assert np.array_equal(e.manifolds(), ref.manifold_counts())
    assert np.array_equal(e.link_poses(), ref.link_poses().astype(np.float32))


@pytest.mark.parametrize('over', [{}, dict(TASK_NAME='crossing', LAYOUT_ID=0, MOVABLE_NAME='CONCAVE', MAX_STEPS=3),
                                  dict(MIN_MOVABLE_BODIES=1, MAX_MOVABLE_BODIES=4, NUM_GOAL_STEPS=2),
                                  {'PHYSICS.ARM_EFFORT_LIMIT': 1}, {'PHYSICS.GRAVITY_XY': (0.3, -0.2)},
                                  {'PHYSICS.LIMB_DYNAMICS': 1}, {'PHYSICS.SOLVER_STALL': 0}, {'PHYSICS.SOLVER_STALL': 3},
                                  {'PHYSICS.SLEEP_STEPS': 0, 'MAX_STEPS': 2}, {'PHYSICS.SOLVER_TOL_REST': 1e-7}, {'PHYSICS.SLEEP_STEPS': 0, 'PHYSICS.SOLVER_TOL_REST': 0.0, 'MAX_STEPS': 2}])
def test_emulated_kernel_is_bit_exact_vs_float_oracle(emu, over):
    from oracle import orc
    scene, names = scenes.make_scene()
    cfg = configs.make_rv_config(env_cfg=configs.push_env_config(**over), n_envs=6, seed=17, shape_names=names)
    ref = orc.OracleWorld(cfg, scene, double=False)
    e = Emu(emu, cfg, scene)
    ref.reset(); emu.emu_reset(e.h, None)
    _check(e, ref)
    for k in range(2):
        a = ref.policy_random(k)
        ref.set_actions(a); emu.emu_set_actions(e.h, a.ctypes.data_as(C.c_void_p))
        ref.step_macro(); emu.emu_step_macro(e.h)
        _check(e, ref)
        r = np.zeros(6, np.float32); d = np.zeros(6, np.uint8)
        emu.emu_reward(e.h, r.ctypes.data_as(C.c_void_p), d.ctypes.data_as(C.c_void_p))
        rr, rd = ref.reward()
        assert np.array_equal(r, rr.astype(np.float32)) and np.array_equal(d, rd)


def test_masked_reset_only_touches_masked_envs(emu):
    from oracle import orc
    scene, names = scenes.make_scene()
    cfg = configs.make_rv_config(n_envs=4, seed=2, shape_names=names)
    ref = orc.OracleWorld(cfg, scene); e = Emu(emu, cfg, scene)
    ref.reset(); emu.emu_reset(e.h, None)
    before = e.body_state().copy()
    mask = np.array([0, 1, 0, 1], np.uint8)
    ref.reset(mask); emu.emu_reset(e.h, mask.ctypes.data_as(C.c_void_p))
    after = e.body_state()
    assert np.array_equal(after[[0, 2]], before[[0, 2]]) and not np.array_equal(after[[1, 3]], before[[1, 3]])
    _check(e, ref)


def test_rollout_equals_lockstep_and_oracle(emu):
    """rv_rollout semantics: K x (RandomPolicy action -> env.step) in one launch
    gives exactly the lock-step sequence (policy_random -> set_actions ->
    step_macro, reset of finished envs in between)."""
    from oracle import orc
    scene, names = scenes.make_scene()
    cfg = configs.make_rv_config(env_cfg=configs.push_env_config(MAX_STEPS=2), n_envs=5, seed=23, shape_names=names)
    ref_lock = orc.OracleWorld(cfg, scene); ref_roll = orc.OracleWorld(cfg, scene); e = Emu(emu, cfg, scene)
    for w in (ref_lock, ref_roll):
        w.reset()
    emu.emu_reset(e.h, None)
    K = 3
    for k in range(K):
        done = ref_lock.env_counters()[:, 4].astype(np.uint8)
        if done.any():
            ref_lock.reset(done)
        ref_lock.set_actions(ref_lock.policy_random(10 + k)); ref_lock.step_macro()
    ref_roll.rollout(K, 10, True)
    emu.emu_rollout(e.h, K, 10, 1)
    assert np.array_equal(ref_roll.body_state(), ref_lock.body_state())
    assert np.array_equal(ref_roll.env_counters()[:, :7], ref_lock.env_counters()[:, :7])
    _check(e, ref_roll)
    assert ref_roll.stats()['env_steps'] == K * 5


def test_long_rollout_with_resets_is_bit_exact(emu):
    """Several episodes per env in one launch (auto-reset), enough envs and steps to
    exercise coasting chains, lazy kinematics, the wake queries and their distance
    bounds across resets."""
    from oracle import orc
    scene, names = scenes.make_scene()
    cfg = configs.make_rv_config(env_cfg=configs.push_env_config(MAX_STEPS=3), n_envs=24, seed=77, shape_names=names)
    ref = orc.OracleWorld(cfg, scene); e = Emu(emu, cfg, scene)
    ref.reset(); emu.emu_reset(e.h, None)
    ref.rollout(8, 2, True)
    emu.emu_rollout(e.h, 8, 2, 1)
    _check(e, ref)


def test_grasp_env_is_bit_exact_vs_float_oracle(emu):
    """Grasp4DofEnv (BASELINE config 4) through the emulated kernel: reset, aimed and random
    grasps (force-limited gripper, per-substep phase machine, friction switches, GraspReward),
    then a rollout with auto-reset -- equal to the float oracle bit for bit."""
    from oracle import orc
    from robovat_amd.math import rotations
    env_cfg = configs.grasp_env_config()
    scene, names = scenes.make_scene(env_cfg=env_cfg)
    cfg = configs.make_rv_config(env_cfg=env_cfg, n_envs=8, seed=3, shape_names=names)
    ref = orc.OracleWorld(cfg, scene, double=False)
    e = Emu(emu, cfg, scene)
    ref.reset(); emu.emu_reset(e.h, None)
    _check(e, ref)
    a = ref.policy_random(0)
    st = ref.body_state()
    for i in range(0, 8, 2):         # every other env: a grasp aimed at the object
        a[i, 0, :2] = st[i, 0, :2]
        a[i, 0, 3] = rotations.euler_from_quaternion(st[i, 0, 3:7])[2]
    ref.set_actions(a); emu.emu_set_actions(e.h, a.ctypes.data_as(C.c_void_p))
    ref.step_macro(); emu.emu_step_macro(e.h)
    _check(e, ref)
    r = np.zeros(8, np.float32); d = np.zeros(8, np.uint8)
    emu.emu_reward(e.h, r.ctypes.data_as(C.c_void_p), d.ctypes.data_as(C.c_void_p))
    rr, rd = ref.reward()
    assert np.array_equal(r, rr.astype(np.float32)) and np.array_equal(d, rd) and d.all()
    assert 0 < r.sum() < 8                       # some grasps hold, some miss
    ref.rollout(2, 1, True); emu.emu_rollout(e.h, 2, 1, 1)
    _check(e, ref)


@pytest.mark.parametrize('budget', [37, 500, 2500])
def test_partial_steps_equal_whole_steps(emu, budget):
    """rv_step_begin / rv_step_poll (lane emulator, substep budget): an env.step() cut into launches
    of at most `budget` substeps gives the states, counters and rewards of rv_step_macro, bit for
    bit, whatever the budget (37 cuts inside phase segments, coasting runs and the closing settle)."""
    from oracle import orc
    scene, names = scenes.make_scene()
    cfg = configs.make_rv_config(env_cfg=configs.push_env_config(MAX_STEPS=3), n_envs=5, seed=23, shape_names=names)
    ref = orc.OracleWorld(cfg, scene, double=False)
    e = Emu(emu, cfg, scene)
    ref.reset(); emu.emu_reset(e.h, None)
    for k in range(2):
        a = ref.policy_random(k)
        ref.set_actions(a); ref.step_macro()
        emu.emu_step_begin(e.h, a.ctypes.data_as(C.c_void_p), None)
        done_mask = np.zeros(5, np.uint8); polls = 0
        while not done_mask.all():
            fin = np.zeros(5, np.uint8)
            emu.emu_step_poll(e.h, budget, fin.ctypes.data_as(C.c_void_p))
            assert not (fin & done_mask).any()                  # an env finishes once
            done_mask |= fin; polls += 1
            assert polls < 2000
        assert polls > 1 or budget > 2000
        assert np.array_equal(e.body_state(), ref.body_state().astype(np.float32))
        assert np.array_equal(e.joint_state(), ref.joint_state().astype(np.float32))
        cr, ce = ref.env_counters(), e.counters()
        assert np.array_equal(ce[:, :7], cr[:, :7])              # sim_steps, num_steps, episodes, phase, done, safe, effective
        assert np.array_equal(e.manifolds(), ref.manifold_counts())
        r = np.zeros(5, np.float32); d = np.zeros(5, np.uint8)
        emu.emu_reward(e.h, r.ctypes.data_as(C.c_void_p), d.ctypes.data_as(C.c_void_p))
        rr, rd = ref.reward()
        assert np.array_equal(d, rd)


def test_emulated_grasp_env_is_bit_exact_vs_float_oracle(emu):
    """Grasp4DofEnv (force-limited gripper: impulse-space solver with the two finger DOFs and their motor
    rows; phase machine ticking after every substep, fused coasting included) on the lane emulator."""
    from oracle import orc
    genv = configs.grasp_env_config()
    scene, names = scenes.make_scene(env_cfg=genv)
    cfg = configs.make_rv_config(env_cfg=genv, n_envs=6, seed=5, shape_names=names)
    ref = orc.OracleWorld(cfg, scene, double=False)
    e = Emu(emu, cfg, scene)
    ref.reset(); emu.emu_reset(e.h, None)
    _check(e, ref)
    for k in range(2):
        a = ref.policy_random(k)
        ref.set_actions(a); emu.emu_set_actions(e.h, a.ctypes.data_as(C.c_void_p))
        ref.step_macro(); emu.emu_step_macro(e.h)
        _check(e, ref)
        ref.reset(); emu.emu_reset(e.h, None)          # every grasp is a whole episode
        _check(e, ref)


@pytest.mark.parametrize('budget', [53, 700, 6000])
def test_partial_grasp_steps_equal_whole_steps(emu, budget):
    """rv_step_begin / rv_step_poll on a Grasp4DofEnv (lane emulator, substep budget): the step cut inside the phase loop
    (which ticks after every substep), inside the reward's wait_until_stable, or not at all ends where rv_step_macro ends."""
    from oracle import orc
    genv = configs.grasp_env_config()
    scene, names = scenes.make_scene(env_cfg=genv)
    cfg = configs.make_rv_config(env_cfg=genv, n_envs=6, seed=5, shape_names=names)
    ref = orc.OracleWorld(cfg, scene, double=False)
    e = Emu(emu, cfg, scene)
    ref.reset(); emu.emu_reset(e.h, None)
    a = ref.policy_random(0)
    st = ref.body_state()
    for i in range(0, 6, 2):
        a[i, 0, :2] = st[i, 0, :2]
    ref.set_actions(a); ref.step_macro()
    emu.emu_step_begin(e.h, a.ctypes.data_as(C.c_void_p), None)
    done_mask = np.zeros(6, np.uint8); polls = 0
    while not done_mask.all():
        fin = np.zeros(6, np.uint8)
        emu.emu_step_poll(e.h, budget, fin.ctypes.data_as(C.c_void_p))
        assert not (fin & done_mask).any()
        done_mask |= fin; polls += 1
        assert polls < 4000
    assert polls > 1 or budget > 5000
    assert np.array_equal(e.body_state(), ref.body_state().astype(np.float32))
    assert np.array_equal(e.joint_state(), ref.joint_state().astype(np.float32))
    assert np.array_equal(e.counters()[:, :7], ref.env_counters()[:, :7])
    assert np.array_equal(e.manifolds(), ref.manifold_counts())
    r = np.zeros(6, np.float32); d = np.zeros(6, np.uint8)
    emu.emu_reward(e.h, r.ctypes.data_as(C.c_void_p), d.ctypes.data_as(C.c_void_p))
    rr, rd = ref.reward()
    assert np.array_equal(r, rr.astype(np.float32)) and d.all()


def test_concentric_overlaps_run_epa_from_a_grown_simplex(emu):
    """Bodies teleported INTO each other (same centre, same orientation: identical shapes give a mirror-symmetric
    difference body, GJK's closest point is the origin on a segment / triangle): the grown-simplex + EPA path of
    rv_dev_collide.h equals the oracle's, and the pair manifolds hold deep points (until round 5: depth 0)."""
    from oracle import orc
    scene, names = scenes.make_scene()
    n = 32
    cfg = configs.make_rv_config(env_cfg=configs.push_env_config(), n_envs=n, seed=5, shape_names=names)
    ref = orc.OracleWorld(cfg, scene, double=False)
    e = Emu(emu, cfg, scene)
    ref.reset(); emu.emu_reset(e.h, None)
    st = ref.body_state().copy()
    for a, b in ((0, 1), (2, 3)):
        st[:, b, :7] = st[:, a, :7]
    st[:, :, 2] += 0.03; st[:, :, 7:] = 0
    ref.set_body_state(st); emu.emu_set_body_state(e.h, st.astype(np.float32).ctypes.data_as(C.c_void_p))
    ref.step_sub(1); emu.emu_step_sub(e.h, 1)
    assert np.array_equal(e.body_state(), ref.body_state().astype(np.float32)) and np.array_equal(e.manifolds(), ref.manifold_counts())
    same_shape = ref.body_params()[:, 0, 1] == ref.body_params()[:, 1, 1]
    deep = sum(1 for i in range(n) if same_shape[i] and ref.manifold(i, abi.RV_MAXB)[0] > 0 and ref.manifold(i, abi.RV_MAXB)[1][:, 9].min() < -0.005)
    assert same_shape.sum() >= 4 and deep == same_shape.sum(), (deep, same_shape.sum())
    ref.step_sub(30); emu.emu_step_sub(e.h, 30)
    assert np.array_equal(e.body_state(), ref.body_state().astype(np.float32))


@pytest.mark.parametrize('garbage', ['0xe846f640', '0x7f7ffffe', '0x00000000', '0x3f9d7a31'])
def test_a_launch_that_begins_with_a_reset_reads_nothing_left_over_in_the_scratch_block(emu, garbage, monkeypatch):
    """One-step rollouts with auto_reset over episodes of two steps: every other launch begins with reset + settle (no arm
    in those substeps) and goes straight on to a step.  The scratch block starts as `garbage` (tests/emu: RV_EMU_POISON; NaNs
    otherwise; an odd word is hashed per position).  With a huge negative float the wake test's box-travel scratch, which only substeps WITH the arm used to
    write, turned the distance bounds of the sleepers into "far for ever" (round 5: 11 of 64 envs differed on the GPU's
    poisoned-LDS build, tools/diag_poison_bisect.py named the word)."""
    from oracle import orc
    monkeypatch.setenv('RV_EMU_POISON', garbage)
    scene, names = scenes.make_scene()
    cfg = configs.make_rv_config(env_cfg=configs.push_env_config(MAX_STEPS=2), n_envs=24, seed=78, shape_names=names)
    ref = orc.OracleWorld(cfg, scene, double=False)
    e = Emu(emu, cfg, scene)
    ref.reset(); emu.emu_reset(e.h, None)
    for k in range(4):
        ref.rollout(1, k, True); emu.emu_rollout(e.h, 1, k, 1)
        _check(e, ref)

"""Static bodies (Simulator.add_body(..., is_static=True), simulator.py:195-224 -> bullet_physics.py:143-181) and the wall
of ArmEnv._reset_scene (arm_env.py:94-99: `if SIM.WALL.USE: simulator.add_body(WALL.PATH, WALL.POSE, is_static=True)`).

A static body is a body of mass 0 (Bullet's convention): it keeps its slot and its pair manifolds with the movable bodies,
nothing moves it, the env logic (observations, reward, safety, policies) does not count it among the movables.

  * closed form: a box sliding towards a wall stops AT the wall (it would have slid on without it); the wall's pose does
    not change by a bit; a body resting against the wall stays there
  * the kernel program (host lane emulation) == float oracle with SIM.WALL.USE through reset + macro steps, with and
    without deactivation
  * -m gpu: HIP == float oracle bit for bit with the wall across the workspace (rollouts with resets, bodies thrown at the
    wall), both builds of the env kernel; the wall is rendered (it has a segment id) but is no row of the pose observation
"""
import ctypes as C

import numpy as np
import pytest

from robovat_amd import abi, configs, scenes

from test_kat_contact import BACKENDS, Q0, _Np, _bodies     # noqa: F401

WALL_HALF = (0.02, 0.65, 0.4)       # scenes.default_shape_hulls: 'wall'
BOX_HALF_X = 0.035


def _world(backend, n=1, **over):
    scene, names = scenes.make_scene()
    cfg = configs.make_rv_config(env_cfg=configs.push_env_config(**over), n_envs=n, seed=1, shape_names=names)
    if backend == 'hip':
        from robovat_amd import lib
        return _Np(lib.World(cfg, scene, device=0)), cfg, names
    from oracle import orc
    return orc.OracleWorld(cfg, scene, double=(backend == 'oracle64')), cfg, names


@pytest.mark.parametrize('backend', BACKENDS)
@pytest.mark.parametrize('sleep', [None, 0])
def test_a_sliding_box_stops_at_a_static_wall(backend, sleep):
    over = {} if sleep is None else {'PHYSICS.SLEEP_STEPS': 0}
    face = 0.72 - WALL_HALF[0]                       # the wall's near face
    finals = {}
    for with_wall in (False, True):
        w, cfg, names = _world(backend, **over)
        rows = [(0, 0.2, 0.2, (0.55, 0.0, 0.031), Q0, (1.0, 0, 0))]
        if with_wall:
            rows.append((names.index('wall'), 0.0, 1.0, (0.72, 0.0, 0.4), Q0, (0, 0, 0)))
        _bodies(w, rows)
        wall0 = np.asarray(w.body_state())[0, 1].copy()
        w.step_sub(800)
        st = np.asarray(w.body_state())[0]
        finals[with_wall] = st[0, 0]
        assert np.abs(st[0, 7:13]).max() < 1e-3, st[0, 7:13]          # at rest
        if with_wall:
            # nothing moved the wall: not by a bit
            assert np.array_equal(np.asarray(w.body_state())[0, 1], wall0)
            # the box's front face is at the wall's, within the collision margins and the allowed penetration
            gap = face - (st[0, 0] + BOX_HALF_X)
            assert -2e-3 < gap < 4e-3, gap
            m = np.asarray(w.manifold_counts())[0]
            assert m[abi.RV_MAXB + 0] > 0                             # the (0, 1) pair manifold holds the contact
            assert m[1] == 0 and m[abi.RV_MAXB + abi.RV_NBB + 1] == 0   # a static body has no table / arm manifold
    # without the wall the box would have slid through where the wall stands
    assert finals[False] + BOX_HALF_X > face + 0.02, finals
    assert finals[True] < finals[False] - 0.02, finals


@pytest.mark.parametrize('backend', BACKENDS)
def test_a_box_leaning_on_a_static_body_stays(backend):
    """Tilted gravity presses a box against the wall: it neither sinks into it nor creeps along it, the wall stays."""
    w, cfg, names = _world(backend, **{'PHYSICS.GRAVITY_XY': (2.0, 0.0)})
    face = 0.72 - WALL_HALF[0]
    _bodies(w, [(0, 0.2, 0.1, (face - BOX_HALF_X - 0.004, 0.0, 0.031), Q0, (0, 0, 0)),
                (names.index('wall'), 0.0, 1.0, (0.72, 0.0, 0.4), Q0, (0, 0, 0))])
    wall0 = np.asarray(w.body_state())[0, 1].copy()
    w.step_sub(600)
    a = np.asarray(w.body_state())[0].copy()
    w.step_sub(600)
    b = np.asarray(w.body_state())[0]
    assert np.array_equal(b[1], wall0)
    assert np.abs(b[0, :3] - a[0, :3]).max() < 2e-4, b[0, :3] - a[0, :3]
    gap = face - (b[0, 0] + BOX_HALF_X)
    assert -2e-3 < gap < 4e-3, gap


def test_wall_config_is_validated():
    scene, names = scenes.make_scene()
    with pytest.raises(ValueError):
        configs.make_rv_config(env_cfg=configs.push_env_config(**{'SIM.WALL.USE': True}), n_envs=1, shape_names=names)    # 4 movables + wall
    with pytest.raises(ValueError):
        configs.make_rv_config(env_cfg=configs.push_env_config(**{'SIM.WALL.USE': True, 'SIM.WALL.SHAPE': 'nope', 'MAX_MOVABLE_BODIES': 3, 'MIN_MOVABLE_BODIES': 3}), n_envs=1, shape_names=names)
    cfg = configs.make_rv_config(env_cfg=configs.push_env_config(**{'SIM.WALL.USE': True, 'MAX_MOVABLE_BODIES': 3, 'MIN_MOVABLE_BODIES': 2}), n_envs=1, shape_names=names)
    assert cfg.wall_use == 1 and cfg.wall_shape == names.index('wall') and list(cfg.wall_pose)[:3] == pytest.approx([1.0, 0.0, 0.4])


WALLS = [{'SIM.WALL.USE': True, 'SIM.WALL.POSE': [[0.66, 0.0, 0.4], [0, 0, 0]], 'MAX_MOVABLE_BODIES': 3, 'MIN_MOVABLE_BODIES': 3},
         {'SIM.WALL.USE': True, 'SIM.WALL.POSE': [[0.6, 0.1, 0.4], [0, 0, 1.5708]], 'MAX_MOVABLE_BODIES': 3, 'MIN_MOVABLE_BODIES': 2,
          'PHYSICS.SLEEP_STEPS': 0, 'MAX_STEPS': 2}]


def _throw_at_the_wall(cfg, st, par):
    """Body states with every movable moving towards the wall's plane at 0.8 m/s (so that the pair manifolds with the wall
    fill up in every env), the wall untouched."""
    st = st.copy()
    wp = np.asarray(list(cfg.wall_pose)[:3])
    for i in range(st.shape[0]):
        for b in range(abi.RV_MAXB - 1):
            if par[i, b, 0] == 0:
                continue
            d = wp[:2] - st[i, b, :2]
            d = d / max(np.linalg.norm(d), 1e-6)
            st[i, b, 7:9] = 0.8 * d
    return st


@pytest.mark.parametrize('over', WALLS)
def test_emulated_kernel_with_a_wall_is_bit_exact_vs_float_oracle(over):
    import test_emu_parity as T
    from oracle import orc
    import os, subprocess
    so = os.path.join(T.EMU_DIR, 'librv_emu.so')
    if not os.path.exists(so):
        subprocess.run(['g++', '-O2', '-std=c++17', '-fPIC', '-ffp-contract=off', '-mfma', '-fopenmp', '-shared', os.path.join(T.EMU_DIR, 'rv_emu.cpp'), '-o', so], check=True)
    lib = C.CDLL(so)
    lib.emu_create.restype = C.c_void_p
    lib.emu_create.argtypes = [C.POINTER(abi.rv_config), C.POINTER(abi.rv_scene)]
    scene, names = scenes.make_scene()
    cfg = configs.make_rv_config(env_cfg=configs.push_env_config(**over), n_envs=6, seed=17, shape_names=names)
    ref = orc.OracleWorld(cfg, scene, double=False)
    e = T.Emu(lib, cfg, scene)
    ref.reset(); lib.emu_reset(e.h, None)
    T._check(e, ref)
    wall = ref.body_state()[:, abi.RV_MAXB - 1].copy()
    assert np.allclose(wall[:, :3], list(cfg.wall_pose)[:3]) and np.all(ref.body_params()[:, abi.RV_MAXB - 1, 3] == 0)     # mass 0
    # bodies thrown at the wall, then two env.step()s
    st = _throw_at_the_wall(cfg, ref.body_state(), ref.body_params())
    ref.set_body_state(st)
    s32 = np.ascontiguousarray(st, np.float32); lib.emu_set_body_state(e.h, s32.ctypes.data_as(C.c_void_p))
    ref.step_sub(300); lib.emu_step_sub(e.h, 300)
    T._check(e, ref)
    touched = ref.manifold_counts()[:, [abi.RV_MAXB + 2, abi.RV_MAXB + 4, abi.RV_MAXB + 5]].sum()
    assert touched > 0          # pair manifolds (0, 3), (1, 3), (2, 3): somebody reached the wall
    for k in range(2):
        a = ref.policy_random(k)
        ref.set_actions(a); lib.emu_set_actions(e.h, a.ctypes.data_as(C.c_void_p))
        ref.step_macro(); lib.emu_step_macro(e.h)
        T._check(e, ref)
    assert np.array_equal(ref.body_state()[:, abi.RV_MAXB - 1], wall)


def test_the_wall_is_not_a_movable_body_of_the_env():
    """Observation rows, body mask and the heuristic policy's body count see the movables only (self.movable_bodies)."""
    from oracle import orc
    scene, names = scenes.make_scene()
    cfg = configs.make_rv_config(env_cfg=configs.push_env_config(**WALLS[0]), n_envs=4, seed=3, shape_names=names)
    ref = orc.OracleWorld(cfg, scene, double=False)
    ref.reset()
    obs = ref.observe(full=True)
    mask = np.asarray(obs['body_mask']).reshape(4, abi.RV_MAXB)
    assert np.all(mask[:, :3] == 1) and np.all(mask[:, 3] == 0)
    pos = np.asarray(obs['position']).reshape(4, abi.RV_MAXB, 3)
    assert np.all(pos[:, 3] == 0)


@pytest.mark.gpu
@pytest.mark.parametrize('over', WALLS)
@pytest.mark.parametrize('occ', ['1', '2'])
def test_hip_with_a_wall_matches_float_oracle_bit_for_bit(over, occ, monkeypatch):
    monkeypatch.setenv('RV_ENV_OCC', occ)
    from test_gpu_parity import _worlds, _cmp
    world, ref, cfg = _worlds(32, seed=9, **over)
    world.reset(); ref.reset()
    assert _cmp(world, ref, 0.0) == 0.0
    wall = ref.body_state()[:, abi.RV_MAXB - 1].astype(np.float32).copy()
    st = _throw_at_the_wall(cfg, ref.body_state(), ref.body_params())
    ref.set_body_state(st); world.set_body_state(st)
    world.step_sub(300); ref.step_sub(300)
    assert _cmp(world, ref, 0.0) == 0.0
    assert ref.manifold_counts()[:, [abi.RV_MAXB + 2, abi.RV_MAXB + 4, abi.RV_MAXB + 5]].sum() > 0
    world.rollout(4, first_macro_index=0, auto_reset=True, record=False)
    ref.rollout(4, 0, True)
    assert _cmp(world, ref, 0.0) == 0.0
    ws, rs = world.stats(), ref.stats()
    for k in ('env_steps', 'substeps', 'awake_substeps'):
        assert ws[k] == rs[k], k
    assert np.array_equal(world.body_state().cpu().numpy()[:, abi.RV_MAXB - 1], wall)
    # the observation: the wall is no row of it; the camera sees it
    got = world.observe(); want = ref.observe(full=True)
    assert np.array_equal(got['body_mask'].cpu().numpy().reshape(-1), np.asarray(want['body_mask'], np.float32).reshape(-1))
    assert np.array_equal(got['position'].cpu().numpy().reshape(-1), np.asarray(want['position'], np.float32).reshape(-1))
    assert np.all(got['body_mask'].cpu().numpy().reshape(-1, abi.RV_MAXB)[:, abi.RV_MAXB - 1] == 0)
    depth, seg = world.render()
    seen = 0
    for i in (0, 7):
        rd, rs_ = ref.render(i)
        assert np.array_equal(seg[i].cpu().numpy(), rs_) and np.array_equal(depth[i].cpu().numpy(), rd)
        seen += int((rs_ == abi.RV_MAXB - 1).sum())
    assert seen > 0        # the wall has pixels of its own


@pytest.mark.parametrize('backend', BACKENDS)
def test_a_constraint_on_a_static_body_is_harmless(backend):
    """A fixed / point-to-point constraint whose parent is a static body (mass 0) tied to the world: every row has zero
    effective mass.  In Bullet a constraint on a fixed-base body does nothing; here the row used to divide 0 by 0 and the
    NaN spread through the pair rows to every body touching the wall (advisor, round 5).  The wall stays where it is bit for
    bit, the box that slides into it stops at it, nothing is NaN."""
    w, cfg, names = _world(backend)
    face = 0.72 - WALL_HALF[0]
    _bodies(w, [(0, 0.2, 0.2, (0.55, 0.0, 0.031), Q0, (1.0, 0, 0)),
                (names.index('wall'), 0.0, 1.0, (0.72, 0.0, 0.4), Q0, (0, 0, 0))])
    w.set_constraint(1, [0.72, 0.0, 0.45, 0, 0, 0, 1], max_force=50.0)                      # world frame 5 cm above: a bias
    w.set_constraint(0, [0.0, 0.0, 0.0, 0, 0, 0, 1], max_force=0.0, child=1, joint_type='point2point')   # a powerless joint to the static body
    wall0 = np.asarray(w.body_state())[0, 1].copy()
    w.step_sub(800)
    st = np.asarray(w.body_state())[0]
    assert np.isfinite(st[:2]).all(), st[:2]
    assert np.array_equal(st[1], wall0)
    gap = face - (st[0, 0] + BOX_HALF_X)
    assert -2e-3 < gap < 4e-3, gap

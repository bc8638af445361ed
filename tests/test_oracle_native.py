"""The `-O3 -march=native` build of the float oracle that bench.py times computes what the parity target
(liborc_f32.so, -O2) computes, bit for bit: same sources, -ffp-contract=off, no fast-math."""
import numpy as np


def test_native_build_equals_the_parity_target(scene_and_names):
    from robovat_amd import configs
    from oracle import orc
    scene, names = scene_and_names
    cfg = configs.make_rv_config(n_envs=6, shape_names=names, seed=5)
    a, b = orc.OracleWorld(cfg, scene), orc.OracleWorld(cfg, scene, native=True)
    for w in (a, b):
        w.reset()
        w.set_actions(w.policy_random(0)); w.step_macro()
        w.set_actions(w.policy_random(1)); w.step_macro()
    assert np.array_equal(a.body_state(), b.body_state())
    assert np.array_equal(a.joint_state(), b.joint_state())
    assert a.stats()['substeps'] == b.stats()['substeps']

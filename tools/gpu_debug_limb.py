import sys; sys.path.insert(0, '.')
import numpy as np
from robovat_amd import abi, configs, scenes, lib
from oracle import orc
scene, names = scenes.make_scene()
for name, ecfg in (('push1', configs.push_env_config(**{'PHYSICS.LIMB_DYNAMICS': 1, 'MIN_MOVABLE_BODIES': 1, 'MAX_MOVABLE_BODIES': 1})),
                   ('push14', configs.push_env_config(**{'PHYSICS.LIMB_DYNAMICS': 1, 'MIN_MOVABLE_BODIES': 1, 'MAX_MOVABLE_BODIES': 4})),
                   ('grasp', configs.grasp_env_config(**{'PHYSICS.LIMB_DYNAMICS': 1}))):
    cfg = configs.make_rv_config(env_cfg=ecfg, n_envs=48, seed=5, shape_names=names)
    w, ref = lib.World(cfg, scene, device=0), orc.OracleWorld(cfg, scene, double=False)
    w.reset(); ref.reset()
    for k in range(4):
        a = ref.policy_random(k)
        import torch
        w.set_actions(torch.from_numpy(a).cuda()); ref.set_actions(a)
        w.step_macro(); ref.step_macro()
        b, rb = w.body_state().cpu().numpy(), ref.body_state().astype(np.float32)
        j, rj = w.joint_state().cpu().numpy(), ref.joint_state().astype(np.float32)
        bad = np.nonzero((b != rb).any(axis=(1, 2)) | (j != rj).any(axis=(1, 2)))[0]
        print(name, 'step', k, 'bad envs', bad[:10], 'max body diff %.3g joint diff %.3g' % (np.abs(b - rb).max(), np.abs(j - rj).max()))
        if len(bad): break
    w.close()

"""GPU-box debugging aid: first substep window in which the HIP path leaves the host lane emulator (which equals the
float oracle) in limb-dynamics mode.  Both step env `ENV` with step_begin / step_poll(max_substeps=W)."""
import sys, os, subprocess, ctypes as C
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from robovat_amd import abi, configs, scenes, lib
ENV, W = int(sys.argv[1]) if len(sys.argv) > 1 else 40, int(sys.argv[2]) if len(sys.argv) > 2 else 25
KSTEP, COARSE = (int(sys.argv[3]), int(sys.argv[4])) if len(sys.argv) > 4 else (-1, 0)   # fine polls only in step KSTEP after COARSE substeps
emu_so = os.path.join(ROOT, 'tests', 'emu', 'librv_emu.so')
subprocess.run(['g++', '-O2', '-std=c++17', '-fPIC', '-ffp-contract=off', '-fopenmp', '-shared', os.path.join(ROOT, 'tests', 'emu', 'rv_emu.cpp'), '-o', emu_so], check=True)
emu = C.CDLL(emu_so)
emu.emu_create.restype = C.c_void_p
emu.emu_create.argtypes = [C.POINTER(abi.rv_config), C.POINTER(abi.rv_scene)]
scene, names = scenes.make_scene()
ecfg = configs.push_env_config(**{'PHYSICS.LIMB_DYNAMICS': 1, 'MIN_MOVABLE_BODIES': 1, 'MAX_MOVABLE_BODIES': 4})
cfg = configs.make_rv_config(env_cfg=ecfg, n_envs=1, env_id_offset=ENV, seed=5, shape_names=names)
w = lib.World(cfg, scene, device=0)
h = C.c_void_p(emu.emu_create(C.byref(cfg), C.byref(scene)))
w.reset(); emu.emu_reset(h, None)
p = lambda a: a.ctypes.data_as(C.c_void_p)
def estate():
    a = np.zeros((1, 4, 13), np.float32); emu.emu_get_body_state(h, p(a)); return a
def ecount():
    a = np.zeros((1, abi.RV_NCOUNTERS), np.int32); emu.emu_get_env_counters(h, p(a)); return a
def eman():
    a = np.zeros((1, abi.RV_NMAN), np.int32); emu.emu_get_manifold_counts(h, p(a)); return a
for k in range(4):
    act = w.policy_random(k)
    a_np = act.cpu().numpy().astype(np.float32)
    w.step_begin(act); emu.emu_step_begin(h, p(a_np), None)
    fin = np.zeros(1, np.uint8); polls = 0; hist = []
    while True:
        ww = W if (KSTEP < 0 or (k == KSTEP and polls >= 1)) else (COARSE if k == KSTEP else 100000)
        f = w.step_poll(max_substeps=ww); emu.emu_step_poll(h, ww, p(fin)); polls += 1
        if k == KSTEP and polls > 1500: print("no divergence in 1500 fine polls"); sys.exit(0)
        hist.append((polls + 1, w.manifold_counts().cpu().numpy()[0].tolist(), w.body_params().cpu().numpy()[0, :, 7].tolist()))
        hs, es = w.body_state().cpu().numpy(), estate()
        hj = w.joint_state().cpu().numpy(); ej = np.zeros((1, abi.RV_NJ, 2), np.float32); emu.emu_get_joint_state(h, p(ej))
        if not np.array_equal(hj, ej):
            print('step', k, 'poll', polls, 'JOINT DIFF max %.3e' % np.abs(hj - ej).max(), 'per joint', np.abs(hj - ej)[0].max(axis=1))
            print(' manifolds hip', w.manifold_counts().cpu().numpy()[0], 'asleep', w.body_params().cpu().numpy()[0, :, 7], 'body diff', np.abs(hs - es).max())
            for hh in hist[-6:]: print('  history (after poll, manifolds, asleep):', hh)
            sys.exit(0)
        if not np.array_equal(hs, es):
            print('step', k, 'poll', polls, 'substeps ~', polls * W, 'DIFF max %.3e' % np.abs(hs - es).max())
            print(' counters hip', w.env_counters().cpu().numpy()[0], 'emu', ecount()[0])
            print(' manifolds hip', w.manifold_counts().cpu().numpy()[0], 'emu', eman()[0])
            print(' asleep hip', w.body_params().cpu().numpy()[0, :, 7], ' vel diff per body', np.abs(hs - es)[0].max(axis=1))
            sys.exit(0)
        if int(f[0]) or fin[0]:
            break
    print('step', k, 'equal after', polls, 'polls')

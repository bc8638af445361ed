"""GPU-box debugging aid: which env / env.step() of a parity-sweep case leaves the float oracle first, and (via the host
lane emulator, which equals the oracle) in which substep window.    python tools/gpu_debug_sweep.py SEED N STEPS [W]"""
import sys, os, subprocess, ctypes as C
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from robovat_amd import abi, configs, scenes, lib
from oracle import orc
SEED, N, STEPS = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
W = int(sys.argv[4]) if len(sys.argv) > 4 else 20
FINE_AFTER = int(os.environ.get('FINE_AFTER', '-1'))      # from that poll on: one substep per poll
OVER = {'PHYSICS.LIMB_DYNAMICS': 1, 'MIN_MOVABLE_BODIES': 1, 'MAX_MOVABLE_BODIES': 4}
for kv in sys.argv[5:]:
    k_, v_ = kv.split('=', 1); OVER[k_] = eval(v_)
scene, names = scenes.make_scene()
ecfg = configs.push_env_config(**OVER)
cfg = configs.make_rv_config(env_cfg=ecfg, n_envs=N, seed=SEED, shape_names=names)
w = lib.World(cfg, scene, device=0); o = orc.OracleWorld(cfg, scene, double=False)
w.reset(); o.reset()
first = None
for k in range(STEPS):
    a = o.policy_random(k)
    w.set_actions(a); o.set_actions(a); w.step_macro(); o.step_macro()
    hs, os_ = w.body_state().cpu().numpy(), o.body_state().astype(np.float32)
    hj, oj = w.joint_state().cpu().numpy(), o.joint_state().astype(np.float32)
    bad = np.nonzero((np.abs(hs - os_).reshape(N, -1).max(1) > 0) | (np.abs(hj - oj).reshape(N, -1).max(1) > 0))[0]
    print('step', k, 'mismatching envs', bad[:12], flush=True)
    if len(bad):
        first = (k, int(bad[0])); break
    r, d = o.reward()
    m = (d != 0).astype(np.uint8)
    if m.any():
        w.reset(mask=m); o.reset(mask=m)
if first is None:
    print('no mismatch'); sys.exit(0)
K, ENV = first
print('first mismatch: step', K, 'env', ENV, ' active bodies', o.body_params()[ENV, :, 0], flush=True)
# the same env alone, HIP against the emulator, in polls of W substeps
emu_so = os.path.join(ROOT, 'tests', 'emu', 'librv_emu.so')
subprocess.run(['g++', '-O2', '-std=c++17', '-fPIC', '-ffp-contract=off', '-fopenmp', '-shared', os.path.join(ROOT, 'tests', 'emu', 'rv_emu.cpp'), '-o', emu_so], check=True)
emu = C.CDLL(emu_so)
emu.emu_create.restype = C.c_void_p
emu.emu_create.argtypes = [C.POINTER(abi.rv_config), C.POINTER(abi.rv_scene)]
cfg1 = configs.make_rv_config(env_cfg=ecfg, n_envs=1, env_id_offset=ENV, seed=SEED, shape_names=names)
w1 = lib.World(cfg1, scene, device=0)
h = C.c_void_p(emu.emu_create(C.byref(cfg1), C.byref(scene)))
w1.reset(); emu.emu_reset(h, None)
p = lambda a: a.ctypes.data_as(C.c_void_p)
def estate():
    a = np.zeros((1, 4, 13), np.float32); emu.emu_get_body_state(h, p(a)); return a
def ejoint():
    a = np.zeros((1, abi.RV_NJ, 2), np.float32); emu.emu_get_joint_state(h, p(a)); return a
def eman():
    a = np.zeros((1, abi.RV_NMAN), np.int32); emu.emu_get_manifold_counts(h, p(a)); return a
for k in range(K + 1):
    act = w1.policy_random(k)
    a_np = act.cpu().numpy().astype(np.float32)
    w1.step_begin(act); emu.emu_step_begin(h, p(a_np), None)
    fin = np.zeros(1, np.uint8); polls = 0; hist = []
    while True:
        ww = (1 if (FINE_AFTER >= 0 and polls >= FINE_AFTER) else W) if k == K else 100000
        f = w1.step_poll(max_substeps=ww); emu.emu_step_poll(h, ww, p(fin)); polls += 1
        hs, es = w1.body_state().cpu().numpy(), estate()
        hj, ej = w1.joint_state().cpu().numpy(), ejoint()
        hist.append((polls, w1.manifold_counts().cpu().numpy()[0].tolist(), w1.body_params().cpu().numpy()[0, :, 7].tolist()))
        if not (np.array_equal(hs, es) and np.array_equal(hj, ej)):
            print('step', k, 'poll', polls, '(substeps ~%d)' % (polls * ww), 'body diff %.3e joint diff %.3e' % (np.abs(hs - es).max(), np.abs(hj - ej).max()))
            print(' per body', np.abs(hs - es)[0].max(axis=1), 'per joint', np.abs(hj - ej)[0].max(axis=1))
            print(' manifolds hip', w1.manifold_counts().cpu().numpy()[0], 'emu', eman()[0])
            print(' hip body0', hs[0, 0], '\n emu body0', es[0, 0]); print(' hip joints', hj[0, :, 1], '\n emu joints', ej[0, :, 1])
            for hh in hist[-5:]: print('  history (poll, manifolds, asleep):', hh)
            sys.exit(0)
        if int(f[0]) or fin[0]:
            break
        if polls > 4000: print('no divergence in 4000 polls'); sys.exit(0)
    print('step', k, 'equal after', polls, 'polls', flush=True)
    r, d = w1.reward()
    if int(d[0]):
        w1.reset(); emu.emu_reset(h, None)
print('single-env run did not reproduce it')

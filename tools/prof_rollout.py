"""Where does the time of the bench workload go?  (profiling build, -DRV_PROFILE)

    RV_LIB=robovat_amd/librovat_hip_prof.so python tools/prof_rollout.py [n_envs] [steps]

Lane 0 of every env accumulates shader-clock time per substep part; this prints
the split over all envs and for the slowest env (which sets the launch time).
Build the profiling library first (here, without a GPU):
    python tools/prof_rollout.py --build
"""
import ctypes as C
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
PROF_LIB = os.path.join(ROOT, 'robovat_amd', 'librovat_hip_prof.so')

if '--build' in sys.argv:
    from robovat_amd import lib
    cmd = ['/opt/rocm/bin/hipcc'] + lib.HIPCC_FLAGS + ['-DRV_PROFILE', os.path.join(lib.CSRC, 'rv_kernels.hip'), '-o', PROF_LIB]
    subprocess.run(cmd, check=True)
    print('built', PROF_LIB)
    sys.exit(0)

os.environ['RV_LIB'] = PROF_LIB
import numpy as np
import torch
from robovat_amd import configs, scenes, lib

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
scene, names = scenes.make_scene()
seed = int(sys.argv[4]) if len(sys.argv) > 4 else 1234
cfg = configs.make_rv_config(n_envs=n, seed=seed, shape_names=names)
w = lib.World(cfg, scene, 0)
L = lib.load()


def prof():
    out = torch.zeros((n, 24), dtype=torch.int64, device=w.device)
    rc = L.rv_debug_profile(w.h, C.c_void_p(out.data_ptr()))
    assert rc == 0
    w.synchronize()
    return out.cpu().numpy().astype(np.float64)


warm = int(sys.argv[3]) if len(sys.argv) > 3 else 0      # launches of `steps` steps before the measured one (bench.py: 5)
w.reset(); w.synchronize()
first = 0
if warm == 0:
    w.rollout(1, first_macro_index=0, auto_reset=True, record=False); w.synchronize(); first = 1
for _ in range(warm):
    w.rollout(steps, first_macro_index=first, auto_reset=True, record=False); w.synchronize(); first += steps
p0 = prof()
w.rollout(steps, first_macro_index=first, auto_reset=True, record=False); w.synchronize()
ms = w.last_kernel_ms()
p = prof() - p0
st = w.stats()
names_ = ['quiet substeps (light part only)', 'light part of non-quiet substeps', 'heavy: twists + hull vertices',
          'heavy: narrow phase', 'heavy: row setup', 'heavy: solver (epilogue + fallback)', 'heavy: integrate + return', 'solver: islands + row loads (+ between substeps)',
          'solver: Delassus rows + warm start', 'solver: sweeps', 'coast: budget (+refresh)', 'coast: control + motors']
names_.append('heavy: narrow phase prep (refresh, gate, list)')
tot = p[:, :7].sum(axis=1) + p[:, 18]
slow = int(np.argmax(tot))
clk = tot.max() / (ms * 1e-3)
print('rollout of %d steps x %d envs: %.1f ms; slowest env = %.3g clocks => counter at %.1f MHz' % (steps, n, ms, tot.max(), clk / 1e6))
print('substeps %d, awake fraction %.3f' % (st['substeps'], st['awake_substeps'] / max(st['substeps'], 1)))
print('%-36s %10s %10s' % ('part', 'mean env', 'slowest'))
for k in list(range(12)) + [18]:
    print('%-46s %9.1f%% %9.1f%%' % (names_[k if k < 12 else 12], 100 * p[:, k].mean() / tot.mean(), 100 * p[slow, k] / tot[slow]))
xn = {19: 'fused: finish', 20: 'random_action', 21: 'fused: entry (clearances, loads)', 22: 'fused: substep loop', 23: 'rollout_record'}
for k in sorted(xn):
    print('%-46s %9.1f%% %9.1f%%' % (xn[k], 100 * p[:, k].mean() / tot.mean(), 100 * p[slow, k] / tot[slow]))
allt = p[:, :12].sum(axis=1) + p[:, 18:24].sum(axis=1)
print('all slots / (slots 0-6,18): mean %.3f slowest %.3f' % (allt.mean() / tot.mean(), allt[slow] / tot[slow]))
print('mean env busy time / slowest env = %.3f' % (tot.mean() / tot.max()))
gn = ['other (culling, loop, idle rounds)', 'GJK/EPA', 'first manifold point', 'feature stage', 'manifold refresh', '-']
for g, gname in ((0, 'group 0 of the query stage'),):
    gt = p[:, 12 + 6 * g: 18 + 6 * g]
    print(gname + ': share of the narrow-phase slot, mean env / slowest env')
    for k in range(5):
        print('   %-36s %6.1f%% %6.1f%%' % (gn[k], 100 * gt[:, k].mean() / p[:, 3].mean(), 100 * gt[slow, k] / p[slow, 3]))
cnt = w.env_counters().cpu().numpy()
order = np.argsort(-tot)[:6]
print('slowest envs: id, ms, substeps, awake, pairs')
for i in order:
    print('  %4d %7.1f %7d %6d %6d   parts%% %s' % (i, tot[i] / clk * 1e3, cnt[i, 7], cnt[i, 8], cnt[i, 9],
          np.round(100 * np.append(p[i, :7], p[i, 18]) / tot[i], 1)))

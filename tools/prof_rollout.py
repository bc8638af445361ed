"""Where does the time of the bench workload go?  (profiling build, -DRV_PROFILE)

    python tools/prof_rollout.py --build            # here, without a GPU
    python tools/prof_rollout.py [--envs 1024] [--steps 20] [--warm 1] [--seed 1234] [--over KEY=VALUE ...]

Lane 0 of every env adds the shader-clock time since its last mark to one of 32 slots
(RV_PROF(i) in rv_dev_env.h; every slot has ONE meaning).  The table gives each slot as
a share of the env's TOTAL marked time (the columns sum to 100 %), for the mean env and
for the slowest env (which sets the launch time), and the busy ratio mean / slowest.
The profiling library is a separate binary (librovat_hip_prof.so); the product library
has no marks.
"""
import argparse
import ctypes as C
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
PROF_LIB = os.path.join(ROOT, 'robovat_amd', 'librovat_hip_prof.so')

if '--build' in sys.argv:
    from robovat_amd import lib
    lib.compile_lib(PROF_LIB, extra=['-DRV_PROFILE'])
    print('built', PROF_LIB)
    sys.exit(0)

ap = argparse.ArgumentParser()
ap.add_argument('--envs', type=int, default=1024)
ap.add_argument('--steps', type=int, default=20)
ap.add_argument('--warm', type=int, default=1, help='launches of `steps` steps before the measured one')
ap.add_argument('--warm-steps', type=int, default=None, help='steps per warm launch (default: --steps); bench.py = --warm 1 --warm-steps 5')
ap.add_argument('--seed', type=int, default=1234)
ap.add_argument('--over', nargs='*', default=[], help='config overrides, e.g. PHYSICS.SLEEP_STEPS=0')
ap.add_argument('--top', type=int, default=6)
ap.add_argument('--grasp', action='store_true', help='BASELINE config 4: Grasp4DofEnv (random CUBOID grasps)')
args = ap.parse_args()

os.environ['RV_LIB'] = PROF_LIB
os.environ['RV_QUEUE'] = '0'      # (the per-env clock marks need an env to stay with one workgroup: the plain launch)
import numpy as np   # noqa: E402
import torch         # noqa: E402
from robovat_amd import configs, scenes, lib   # noqa: E402

NS = 48
SLOTS = {
    0: 'quiet substep: rest of the light part', 1: 'non-quiet substep: rest of the light part (body velocities, quiet test)',
    40: 'light: ControllableBody.update + joint motors', 41: 'light: forward kinematics', 42: 'light: collider boxes, AABBs, gates',
    43: 'light: wake tests', 2: 'heavy: link twists',
    18: 'heavy: narrow-phase prep (refresh, gate, work list, hull vertices)', 3: 'heavy: narrow-phase queries (GJK/EPA, features)',
    4: 'heavy: solver row setup + flags', 25: 'solver: island entry + row loads', 26: 'solver: Delassus rows + warm start',
    27: 'solver: sweeps', 24: 'solver: island glue + epilogues', 5: 'solver: island of 3-4 bodies (velocity space)',
    6: 'heavy: integrate, sleep tests, return', 7: 'substep loop top', 21: 'coast: entry (clearances, loads)',
    22: 'coast: fused substep loop (entry / exit of the loop)', 44: 'coast loop: check-free substeps', 45: 'coast loop: is a controller update due / a no-op', 46: 'coast loop: motor step + out-of-reach test + look-ahead', 47: 'coast loop: counters + tick test', 19: 'coast: finish', 11: 'coast: after a fused run', 10: 'coast: kinematics re-measured',
    8: 'tick: kinematics refresh', 9: 'tick: phase machine', 29: 'env.step prologue', 30: 'env.step: after the run call',
    31: 'env.step epilogue (effectiveness, obs, reward)', 28: 'reset (drop and settle)', 20: 'random_action', 23: 'rollout_record',
}
GROUP = ['other (culling, loop, idle rounds)', 'GJK/EPA', 'first manifold point', 'feature stage', 'manifold refresh', '-']

over = {}
for kv in args.over:
    k, v = kv.split('=')
    try:
        over[k] = int(v)
    except ValueError:
        try:
            over[k] = float(v)
        except ValueError:
            over[k] = v
if args.grasp:
    env_cfg = configs.grasp_env_config(**over)
    scene, names = scenes.make_scene(env_cfg=env_cfg)
else:
    scene, names = scenes.make_scene()
    env_cfg = configs.push_env_config(**over)
cfg = configs.make_rv_config(env_cfg=env_cfg, n_envs=args.envs, seed=args.seed, shape_names=names)
w = lib.World(cfg, scene, 0)
L = lib.load()
if lib.built_source_hash() != lib.source_hash():
    sys.exit('librovat_hip_prof.so was built from other sources: python tools/prof_rollout.py --build')
n = args.envs


def prof():
    out = torch.zeros((n, NS), dtype=torch.int64, device=w.device)
    rc = L.rv_debug_profile(w.h, C.c_void_p(out.data_ptr()))
    assert rc == 0
    w.synchronize()
    return out.cpu().numpy().astype(np.float64)


w.reset(); w.synchronize()
first = 0
ws = args.steps if args.warm_steps is None else args.warm_steps
for _ in range(args.warm):
    w.rollout(ws, first_macro_index=first, auto_reset=True, record=False); w.synchronize(); first += ws
p0 = prof()
w.rollout(args.steps, first_macro_index=first, auto_reset=True, record=False); w.synchronize()
ms = w.last_kernel_ms()
p = prof() - p0
st = w.stats()
main = [k for k in range(32) if not (12 <= k < 18)] + [40, 41, 42, 43, 44, 45, 46, 47]
tot = p[:, main].sum(axis=1)
slow = int(np.argmax(tot))
clk = tot.max() / (ms * 1e-3)
print('# tools/prof_rollout.py --envs %d --steps %d --warm %d --seed %d %s' % (n, args.steps, args.warm, args.seed, ' '.join(args.over)))
print('rollout of %d steps x %d envs: %.1f ms kernel; slowest env marks %.3g clocks => counter at %.1f MHz' % (args.steps, n, ms, tot.max(), clk / 1e6))
print('substeps %d, awake fraction %.4f, env-steps/s (kernel only) %.0f' % (st['substeps'], st['awake_substeps'] / max(st['substeps'], 1), st['env_steps'] / (ms * 1e-3)))
print('mean env busy time / slowest env = %.3f' % (tot.mean() / tot.max()))
print('%-72s %9s %9s %12s' % ('part (share of the env\'s own total)', 'mean env', 'slowest', 'slowest ms'))
s_mean = s_slow = 0.0
for k in SLOTS:
    a, b = 100 * p[:, k].mean() / tot.mean(), 100 * p[slow, k] / tot[slow]
    s_mean += a; s_slow += b
    print('%-72s %8.1f%% %8.1f%% %12.2f' % (SLOTS[k], a, b, p[slow, k] / clk * 1e3))
rest = [k for k in main if k not in SLOTS]
a, b = 100 * p[:, rest].sum(1).mean() / tot.mean(), 100 * p[slow, rest].sum() / tot[slow]
print('%-72s %8.1f%% %8.1f%%' % ('(unnamed slots)', a, b))
print('%-72s %8.1f%% %8.1f%%' % ('sum', s_mean + a, s_slow + b))
gt = p[:, 12:18]
print('group 0 of the query stage: share of the narrow-phase query slot, mean env / slowest env')
for k in range(5):
    print('   %-40s %6.1f%% %6.1f%%' % (GROUP[k], 100 * gt[:, k].mean() / max(p[:, 3].mean(), 1), 100 * gt[slow, k] / max(p[slow, 3], 1)))
cnt = w.env_counters().cpu().numpy()
order = np.argsort(-tot)[:args.top]
heavy = [1, 2, 18, 3, 4, 25, 26, 27, 24, 5, 6]
light4 = [40, 41, 42, 43]
print('slowest envs: id, ms, substeps, awake substeps, convex pairs | % of own time: light, twists, np prep, np queries, rows, isl entry, Delassus, sweeps, glue, big island, integrate | coast+ticks')
for i in order:
    sh = 100 * p[i, heavy] / tot[i]
    print('  %4d %7.1f %7d %6d %6d | %s | %.1f' % (i, tot[i] / clk * 1e3, cnt[i, 7], cnt[i, 8], cnt[i, 9], ' '.join('%4.1f' % x for x in sh),
                                                   100 - sh.sum()))
print('coast loop counts, mean env / slowest env: substeps in the check-free path %.0f / %.0f, full iterations %.0f / %.0f, out-of-reach tests %.0f / %.0f, segments %.0f / %.0f'
      % (p[:, 32].mean(), p[slow, 32], p[:, 33].mean(), p[slow, 33], p[:, 34].mean(), p[slow, 34], p[:, 35].mean(), p[slow, 35]))
q = np.percentile(tot / clk * 1e3, [50, 90, 99, 100])
print('env busy time ms: p50 %.1f  p90 %.1f  p99 %.1f  max %.1f' % tuple(q))
w.close()

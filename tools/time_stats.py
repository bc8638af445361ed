import sys, time
sys.path.insert(0, '.')
import torch
from robovat_amd import configs, scenes, lib
scene, names = scenes.make_scene()
cfg = configs.make_rv_config(n_envs=1024, seed=1234, shape_names=names)
w = lib.World(cfg, scene, 0); w.reset()
w.set_actions(w.policy_random(0)); w.step_macro()
torch.cuda.synchronize()
cnt = torch.zeros(1024, dtype=torch.long, device='cuda'); fin = torch.rand(1024, device='cuda') > 0.9; ar = torch.arange(1024, device='cuda')
A = torch.stack([w.policy_random(k) for k in range(80)])
def t(f, n=50):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize(); return 1e3 * (time.perf_counter() - t0) / n
print('stats() %.3f ms' % t(lambda: w.stats()))
print('int(fin.sum()) %.3f ms' % t(lambda: int(fin.sum())))
def f1(): cnt[fin] += 1
print('cnt[fin] += 1 %.3f ms' % t(f1))
print('bool(live.any()) %.3f ms' % t(lambda: bool(fin.any())))
print('bool((cnt == 0).any()) %.3f ms' % t(lambda: bool((cnt == 0).any())))
print('A[cnt.clamp, ar] %.3f ms' % t(lambda: A[cnt.clamp(max=79), ar]))
print('step_begin %.3f ms' % t(lambda: w.step_begin(A[0], mask=fin.to(torch.uint8))))
print('last_kernel_ms %.3f ms' % t(lambda: w.last_kernel_ms()))

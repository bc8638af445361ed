import sys, time
sys.path.insert(0, '.')
import torch
from robovat_amd import configs, scenes, lib
scene, names = scenes.make_scene()
n = 1024
cfg = configs.make_rv_config(n_envs=n, seed=1234, shape_names=names)
w = lib.World(cfg, scene, 0)
w.reset()
for k in range(5):
    w.set_actions(w.policy_random(k)); w.step_macro()
A = torch.stack([w.policy_random(5 + k) for k in range(80)])
out = w.poll_buffers(point_cloud=True)
cnt = torch.zeros(n, dtype=torch.long, device='cuda'); ar = torch.arange(n, device='cuda')
for usec in (100, 500, 1500, 3000):
    w.step_poll()  # drain
    cnt.zero_()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    w.step_begin(A[0])
    done, polls, t_poll, t_host, kms = 0, 0, 0.0, 0.0, 0.0
    while done < 20 * n:
        ta = time.perf_counter()
        fin = w.step_poll(max_usec=usec, out=out).bool()
        nf = int(fin.sum())
        tb = time.perf_counter(); t_poll += tb - ta; kms += w.last_kernel_ms()
        polls += 1
        if nf:
            live = fin & ~out['done'].bool()          # (an env whose episode ended stops, as in the lock-step leg)
            done += int(live.sum()); cnt[live] += 1
            w.step_begin(A[cnt.clamp(max=79), ar], mask=live.to(torch.uint8))
        t_host += time.perf_counter() - tb
    torch.cuda.synchronize(); el = time.perf_counter() - t0
    print('poll_usec %d: %.0f env-steps/s, %d polls, %.2f ms per poll call (kernel %.2f ms), %.2f ms host per poll, steps/env %d..%d' % (usec, done / el, polls, 1e3 * t_poll / polls, kms / polls, 1e3 * t_host / polls, int(cnt.min()), int(cnt.max())))

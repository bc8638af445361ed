import os, sys
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
import numpy as np
os.environ['RV_QUEUE']='1'
import test_gpu_scale as T
n, per_env = 4096 + 512, 3
world = T._world(n, seed=78, MAX_STEPS=2); world.reset()
taken = world.rollout_async(per_env * n, first_macro_index=0).cpu().numpy()
got = world.body_state().cpu().numpy(); world.close()
ref = T._world(n, seed=78, MAX_STEPS=2); ref.reset()
orc = T._oracle(64, seed=78, MAX_STEPS=2); orc.reset()
want = ref.body_state().cpu().numpy().copy(); wo = orc.body_state().astype(np.float32).copy()
for k in range(int(taken.max())):
    ref.rollout(1, first_macro_index=k, auto_reset=True); orc.rollout(1, k, True)
    cur = ref.body_state().cpu().numpy(); co = orc.body_state().astype(np.float32)
    sel = taken == k + 1
    want[sel] = cur[sel]; wo[sel[:64]] = co[sel[:64]]
bad_q = np.where((got != want).any((1,2)))[0]
print('queue-async vs plain lockstep: differing envs', len(bad_q), bad_q[:10], 'taken there', taken[bad_q[:10]])
print('plain lockstep vs oracle (first 64):', int((want[:64] != wo).any((1,2)).sum()), ' queue-async vs oracle (first 64):', int((got[:64] != wo).any((1,2)).sum()))

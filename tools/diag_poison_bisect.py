"""Which words of the LDS scratch block does a launch read before it writes them?  Run on the GPU box with a poisoned build:
    RV_LIB=build/librovat_poison_big.so python tools/diag_poison_bisect.py
The case: 1-step rollouts with auto_reset of 64 envs whose episodes are 2 steps long, against the oracle.  The poisoned range
[RV_POISON_LO, RV_POISON_HI) is narrowed by bisection; tools/scratch_offsets.cpp names the field at a word offset."""
import os, sys
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import numpy as np
os.environ['RV_QUEUE'] = '0'
import test_gpu_scale as T
N, STEPS = 64, 4
orc = T._oracle(N, seed=78, MAX_STEPS=2); orc.reset()
want = []
for k in range(STEPS):
    orc.rollout(1, k, True); want.append(orc.body_state().astype(np.float32).copy())
def bad(lo, hi):
    os.environ['RV_POISON_LO'] = str(lo); os.environ['RV_POISON_HI'] = str(hi)
    w = T._world(N, seed=78, MAX_STEPS=2); w.reset(); n = 0
    for k in range(STEPS):
        w.rollout(1, first_macro_index=k, auto_reset=True)
        n = max(n, int((w.body_state().cpu().numpy() != want[k]).any((1, 2)).sum()))
    w.close(); return n
TOP = 5100
print('all poisoned:', bad(0, TOP), ' none:', bad(0, 0), flush=True)
if bad(0, TOP):
    lo, hi = 0, TOP
    a, b = 0, TOP                      # smallest hi with bad(0, hi) > 0
    while b - a > 1:
        m = (a + b) // 2
        if bad(0, m): b = m
        else: a = m
    hi = b
    a, b = 0, hi                       # largest lo with bad(lo, hi) > 0
    while b - a > 1:
        m = (a + b) // 2
        if bad(m, hi): a = m
        else: b = m
    lo = a
    print('minimal range: words [%d, %d)  differing envs %d' % (lo, hi, bad(lo, hi)), flush=True)
    print('without that range:', bad(0, lo), '+', bad(hi, TOP), flush=True)

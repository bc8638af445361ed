#!/usr/bin/env python
"""What is the env with the most awake substeps of a lock-step env.step() doing?  (CPU, float oracle)

    python tools/trace_awake.py [step=6] [n_envs=256]

Pass 1 finds the env with the most awake substeps in that step; pass 2 re-runs with a -DORC_TRACE_AWAKE build of
the oracle that prints one line per awake body and substep of that env, and summarises them as segments of
(phase, arm contact, below the sleep speeds, table points).  Typical answer: one long push -- ~1000 substeps of
arm - body contact with the body sliding on four table points -- plus the 200-substep waits before it sleeps."""
import ctypes as C
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
K = int(sys.argv[1]) if len(sys.argv) > 1 else 6
N = int(sys.argv[2]) if len(sys.argv) > 2 else 256

if len(sys.argv) > 3:                        # pass 2 (re-entered below): trace env sys.argv[3]
    so = '/tmp/liborc_trace.so'
    subprocess.run(['gcc', '-O2', '-std=gnu11', '-fPIC', '-ffp-contract=off', '-fno-fast-math', '-fopenmp', '-Wno-unused-function', '-DORC_TRACE_AWAKE',
                    os.path.join(ROOT, 'oracle', 'rv_oracle.c'), '-o', so, '-shared', '-lm', '-fopenmp'], check=True)
    os.environ['ORC_TRACE_ENV'] = sys.argv[3]
    real = C.CDLL
    C.CDLL = lambda path, *a, **k: real(so if str(path).endswith('liborc_f32.so') else path, *a, **k)

import numpy as np  # noqa: E402
from robovat_amd import configs, scenes  # noqa: E402
from oracle import orc  # noqa: E402

scene, names = scenes.make_scene()
w = orc.OracleWorld(configs.make_rv_config(n_envs=N, seed=1234, shape_names=names), scene, double=False)
w.reset()
for k in range(K + 1):
    w.set_actions(w.policy_random(k))
    if len(sys.argv) > 3 and k == K:
        sys.stderr.write('BEGIN\n'); sys.stderr.flush()
    w.step_macro()
    if len(sys.argv) > 3 and k == K:
        sys.stderr.write('END\n'); sys.stderr.flush()
if len(sys.argv) <= 3:
    c = w.env_counters()
    i = int(np.argmax(c[:, 8]))
    print('step %d: env %d has the most awake substeps (%d of %d)' % (K, i, c[i, 8], c[i, 7]))
    out = subprocess.run([sys.executable, os.path.abspath(__file__), str(K), str(N), str(i)], capture_output=True, text=True).stderr
    rows = [l.split() for l in out.split('BEGIN\n')[1].split('END')[0].split('\n') if l.startswith('T ')]
    seg = []
    for r in rows:
        t, ph, v, wv, sc, arm, tab = int(r[1]), int(r[3]), float(r[7]), float(r[9]), int(r[11]), int(r[15]), int(r[17])
        key = (ph, arm > 0, v < 0.02 and wv < 0.5, tab)
        if seg and seg[-1][0] == key and t == seg[-1][2] + 1:
            seg[-1][2] = t; seg[-1][3] = max(seg[-1][3], sc); seg[-1][4] = max(seg[-1][4], v)
        else:
            seg.append([key, t, t, sc, v])
    for key, t0, t1, sc, v in seg:
        if t1 - t0 >= 15:
            print('  phase %d, arm contact %d, below the sleep speeds %d, %d table points: substeps %d..%d (%d), sleep counter up to %d, speed up to %.3f m/s'
                  % (key[0], key[1], key[2], key[3], t0, t1, t1 - t0 + 1, sc, v))

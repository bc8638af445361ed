#!/usr/bin/env python
"""Event counts of the env program, from the host lane-emulator built with -DRV_EMU_COUNT
(CPU only; a tool, never a product path): IK calls, fused-coasting runs and why they end,
substeps by kind, island sizes, GJK iterations, solver iterations.

    python tools/emu_counts.py [n_envs=16] [steps=4]
"""
import ctypes as C
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from robovat_amd import abi, configs, scenes  # noqa: E402

so = '/tmp/librv_emu_cnt.so'
subprocess.run(['g++', '-O2', '-std=c++17', '-fPIC', '-ffp-contract=off', '-mfma', '-shared', '-DRV_EMU_COUNT', '-I' + os.path.join(ROOT, 'include'),
                os.path.join(ROOT, 'tests', 'emu', 'rv_emu.cpp'), '-o', so], check=True)
lib = C.CDLL(so)
lib.emu_create.restype = C.c_void_p
lib.emu_create.argtypes = [C.POINTER(abi.rv_config), C.POINTER(abi.rv_scene)]
n = int(sys.argv[1]) if len(sys.argv) > 1 else 16
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 4
scene, names = scenes.make_scene()
over = {}
for kv in sys.argv[3:]:                      # config overrides, e.g. PHYSICS.SOLVER_STALL=8
    k_, v_ = kv.split('=', 1)
    over[k_] = eval(v_)
cfg = configs.make_rv_config(env_cfg=configs.push_env_config(**over), n_envs=n, seed=1234, shape_names=names)
h = C.c_void_p(lib.emu_create(C.byref(cfg), C.byref(scene)))
lib.emu_reset(h, None)
out = (C.c_long * 48)()
lib.emu_get_counts(out)
base = list(out)
lib.emu_rollout(h, steps, 0, 1)
lib.emu_get_counts(out)
c = [a - b for a, b in zip(out, base)]
per = float(steps * n)
rows = [('IK calls', 0), ('IK iterations', 1), ('fused coasting runs', 2), ('fused substeps', 3), ('  ended: controller has work', 4),
        ('  ended: clearance used up', 5), ('  ended: tick of the phase machine', 6), ('kinematics re-measured in between', 12),
        ('chunked coasting substeps (old path)', 8), ('regular substeps', 9), ('  of them heavy', 10)]
for name, k in rows:
    print('%-40s %10d   %8.1f per env.step()' % (name, c[k], c[k] / per))
print('islands: 1 body %d, 2 bodies %d, 3+ %d' % (c[13], c[16], c[17]))
print('GJK: %d calls, %.2f iterations per call, EPA %d' % (c[18], c[19] / max(c[18], 1), c[20]))
print('solver (host row list): %d solves, %.2f iterations, %.1f rows per solve' % (c[21], c[22] / max(c[21], 1), c[23] / max(c[21], 1)))
print('solves that ran into the iteration cap: %d' % c[30])
print('heavy substeps: arm contact points %d | arm near %d | arm far, moving %d | arm far, static %d ; bodies below the sleep speeds %d'
      % (c[24], c[25], c[26], c[27], c[28]))
print('islands solved (1-2 bodies): %d; ran into the cap: %d with arm rows, %d without (%d of them with a residual > 10 x tol); mean sweeps of the others %.2f'
      % (c[35], c[31], c[32], c[37], c[36] / max(c[35] - c[31] - c[32], 1)))
print('solver work (rows x sweeps): %d, of which in islands that ran into the cap: %d (%.1f %%)' % (c[33], c[34], 100.0 * c[34] / max(c[33], 1)))
print('cap hits with arm rows by phase (initial, pre, start, motion, post, offstage, done): %s; arm normal force > 100 N: %d, > 1000 N: %d' % (c[38:45], c[46], c[47]))
dbg = (C.c_long * 16)(); lib.emu_get_dbg(dbg)
print('rows still changing by >= tol in the last sweep of a capped solve: table n %d t %d | pair n %d t %d | arm n %d t %d ; normal rows at their effort cap %d, friction rows at the cone %d' % tuple(list(dbg)[:8]))
d2 = (C.c_long * 48)(); lib.emu_get_dbg2(d2); d2 = list(d2)
print('fused runs ended by the clearance of box: (body-limited, table-limited) x 10 boxes:', [(d2[2 * i], d2[2 * i + 1]) for i in range(10)])
print('  mean clearance (mm) the limiting box had at the start of the run:', ['%.1f' % (1e-3 * d2[20 + i] / max(d2[2 * i] + d2[2 * i + 1], 1)) for i in range(10)])
print('  mean substeps of those runs:', ['%.1f' % (d2[30 + i] / max(d2[2 * i] + d2[2 * i + 1], 1)) for i in range(10)])
print('convex queries by owner role (table, body-body, arm-body, arm-table):', d2[40:44], ' of them without a contact:', d2[44:48])

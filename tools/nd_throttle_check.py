"""Is a long no-deactivation run at 8192 envs slower per env.step() because of WHAT it computes or because of HOW LONG the GPU
has been busy?  Five back-to-back launches of 2 steps per env, then one of 8 (kernel ms by HIP events; rocm-smi clocks between)."""
import os, sys, subprocess, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from robovat_amd import configs, scenes, lib
scene, names = scenes.make_scene()
cfg = configs.make_rv_config(env_cfg=configs.push_env_config(**{'PHYSICS.SLEEP_STEPS': 0}), n_envs=8192, seed=0, shape_names=names)
w = lib.World(cfg, scene, device=0)
w.reset(); w.synchronize()
def clocks():
    try:
        o = subprocess.run(['rocm-smi', '--showclocks', '--showpower'], capture_output=True, text=True, timeout=20).stdout
        return ' | '.join(l.strip() for l in o.splitlines() if 'sclk' in l or 'Power' in l)[:200]
    except Exception as e:
        return str(e)
k = 0
for n in (2, 2, 2, 2):
    t = time.time(); w.rollout(n, first_macro_index=k, auto_reset=False); w.synchronize(); el = time.time() - t
    st = w.stats(); k += n
    print('steps %d: kernel %.0f ms, %.0f env-steps/s, substeps/env-step %.0f | %s' % (n, w.last_kernel_ms(), st['env_steps'] / el, st['substeps'] / max(st['env_steps'], 1), clocks()), flush=True)

# a fresh world right after the long run: slow too (the GPU: clocks / heat) or fast again (the state of the envs)?
w2 = lib.World(cfg, scene, device=0)
w2.reset(); w2.synchronize()
t = time.time(); w2.rollout(2, first_macro_index=0, auto_reset=False); w2.synchronize(); el = time.time() - t
print('fresh world, steps 2: kernel %.0f ms, %.0f env-steps/s | %s' % (w2.last_kernel_ms(), w2.stats()['env_steps'] / el, clocks()), flush=True)
# ... and the first world once more after resetting it
w.reset(); w.synchronize()
t = time.time(); w.rollout(2, first_macro_index=0, auto_reset=False); w.synchronize(); el = time.time() - t
print('first world after reset, steps 2: kernel %.0f ms, %.0f env-steps/s' % (w.last_kernel_ms(), w.stats()['env_steps'] / el), flush=True)

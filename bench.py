#!/usr/bin/env python
"""bench.py — PushEnv (batched) env steps/sec on N MI355X.

Workload (BASELINE.json configs[1]): PushEnv, 4 rigid convex movables, 1024
vectorised envs per GPU, random policy (Philox U(-1,1)^4 keyed by global env
id and macro-step index), TASK_NAME=None.  One "step" = one batched
`env.step()`: policy -> the whole pre/start/motion/post/offstage phase machine
plus settle (thousands of 1 ms physics substeps per env, on device) ->
observation (position, body mask, attributes, segmented point cloud) + reward +
done.  Inputs are resident in HBM; `value` = env steps per second over all ranks.

`value` (mode "rollout", the default) times ONE `rv_rollout_record` launch in which
every env takes its K steps back to back -- policy on the device, auto-reset, the
observation / reward / done of EVERY step written to [K][N] buffers and all K*N
point clouds rendered -- i.e. a single-launch rollout, not a host-driven loop.  The
host-driven loop (K x (policy -> rv_set_actions -> rv_step_macro -> rv_observe with
point cloud -> rv_reward), every env waiting for the slowest env of each step) is
reported beside it as "lockstep_env_step" (or is `value` with --mode lockstep).

    python bench.py --gpus 1 --steps 20 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N \\
        --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

# 2 * (52 B_dyn + 8 J + 208 M) bytes per env-substep
ALGO_BYTES = {'config2': 3056, 'config3': 5552}
HBM_PEAK_GBS = 8000.0               # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
VALU_PEAK_PER_SIMD_CYCLE = 0.5      # MI355X_MICROARCH.md: a wave64 VALU instruction issues over 2 cycles on a SIMD-32

LINE_LIMIT = 8000                   # bytes of the ONE stdout line (the round-5 line was 22.7 KB and the driver could not parse it)


def _short(s, n=220):
    s = str(s)
    return s if len(s) <= n else s[:n - 3] + '...'


def _leg_scalars(v):
    """`value` (+ the few scalars worth a glance) of a leg; nested legs (reference_semantics.gpu.x ...) keep their shape."""
    if not isinstance(v, dict):
        return None
    if 'value' in v and isinstance(v['value'], (int, float)):
        out = {'value': round(v['value'], 1)}
        for k in ('sim_steps_per_s', 'ms_per_step', 'envs_per_gpu', 'envs', 'steps', 'cores', 'async_value', 'grasp_success_rate',
                  'one_wave_us_per_substep', 'thread_us_per_awake_substep'):
            if isinstance(v.get(k), (int, float)):
                out[k] = round(v[k], 3) if isinstance(v[k], float) else v[k]
        return out
    out = {}
    for k, x in v.items():
        sub = _leg_scalars(x)
        if sub:
            out[k] = sub
    return out or None


def _sig(x, digits=6):
    """floats to `digits` significant digits, recursively (the full precision is in the side file)"""
    if isinstance(x, float):
        return float('%.*g' % (digits, x))
    if isinstance(x, dict):
        return {k: _sig(v, digits) for k, v in x.items()}
    if isinstance(x, (list, tuple)):
        return [_sig(v, digits) for v in x]
    return x


def compact_line(full, limit=LINE_LIMIT, full_record=None):
    """The one JSON line the driver parses: the contract's keys, `roofline`, `cpu_baseline`, one scalar group per extra leg.
    Everything else (notes, per-leg outcome statistics, the nested semantics legs) is in `full_record` (bench_legs.json)
    and on stderr.  Guaranteed < `limit` bytes: optional groups are dropped, least important first, until it fits."""
    keep = ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better', 'scaling', 'vs_baseline',
            'dtype', 'data', 'sim_steps_per_s', 'substeps_per_env_step', 'awake_sim_steps_per_s', 'awake_substep_fraction',
            'max_substeps_in_launch', 'n1_same_workload', 'scaling_efficiency_vs_n1_same_workload')
    line = {k: full[k] for k in keep if k in full}
    line['config'] = {k: (_short(v) if isinstance(v, str) else v) for k, v in full.get('config', {}).items()}
    rf = full.get('roofline')
    if rf:
        r = {k: rf[k] for k in ('bound', 'achieved', 'peak', 'unit', 'frac', 'traffic', 'kernel', 'avg_kernel_ms',
                                'valu_insts_per_env_substep', 'env_substeps_per_launch', 'occupied_simds') if k in rf}
        hn = rf.get('hbm_nominal')
        if hn:
            r['hbm_nominal'] = {k: hn[k] for k in ('bound', 'achieved', 'peak', 'unit', 'frac', 'algorithmic_bytes_per_env_substep',
                                                  'achieved_awake_substeps_only') if k in hn}
        isd = rf.get('issue_side')
        if isd:
            r['while_resident'] = {k: isd[k] for k in ('valu_issue_frac_of_peak_while_resident', 'wait_frac', 'source') if k in isd}
        line['roofline'] = r
    cb = full.get('cpu_baseline')
    if cb:
        line['cpu_baseline'] = {k: (_short(cb[k], 300) if isinstance(cb[k], str) else cb[k])
                                for k in ('value', 'unit', 'cores', 'kind', 'sample', 'sim_steps_per_s', 'thread_us_per_awake_substep',
                                          'scaling_1_to_n', 'compiler_flags') if k in cb}
    if 'pybullet' in full:
        pbv = full['pybullet']
        line['pybullet'] = {k: (_short(v, 300) if isinstance(v, str) else v) for k, v in pbv.items()
                            if k in ('status', 'pose_err_hip_vs_pybullet', 'pose_err_f64_oracle_vs_pybullet', 'step_simulation')} \
            if isinstance(pbv, dict) else _short(pbv, 300)
    optional = []                       # (name, value), most important first
    pe = full.get('pose_err')
    if pe:
        optional.append(('pose_err_vs_f64_oracle', {k: {'max_pos_m': v['max_pos_m'], 'p99_pos_m': v['p99_pos_m'], 'max_angle_rad': v['max_angle_rad']}
                                                  for k, v in pe.items() if isinstance(v, dict) and 'max_pos_m' in v}))
    legs = {}
    skip = set(line) | {'roofline', 'cpu_baseline', 'config', 'pose_err', 'pybullet', 'host', 'deactivation'}
    for k, v in full.items():
        if k in skip:
            continue
        sub = _leg_scalars(v)
        if sub:
            legs[k] = sub
    if legs:
        optional.append(('legs', legs))
    de = (full.get('deactivation') or {})
    if de.get('pose_equivalence'):
        optional.append(('deactivation_pose_equivalence', {
            k: {q: v[q] for q in ('env_steps', 'median_m', 'p90_m', 'p99_m', 'max_m', 'flags_agree', 'within_bounds', 'horizon') if q in v}
            for k, v in de['pose_equivalence'].items() if isinstance(v, dict)}))
    dl = {k: _leg_scalars(v) for k, v in de.items() if k != 'pose_equivalence' and _leg_scalars(v)}
    if dl:
        optional.append(('deactivation_legs', dl))
    if full_record:
        line['full_record'] = full_record
    for name, val in optional:
        line[name] = val
    text = json.dumps(_sig(line))
    while len(text) >= limit and optional:
        name, _ = optional.pop()
        del line[name]
        line['dropped_for_size'] = line.get('dropped_for_size', []) + [name]
        text = json.dumps(_sig(line))
    assert len(text) < limit, len(text)
    return text


def effective_cpus():
    from oracle import orc
    return orc.effective_cpus()


def pybullet_probe():
    """(module or None, status): `import pybullet` is executed every time this is called (tools/pybullet_parity.py);
    no string about its availability in this file is a constant."""
    tools = os.path.join(ROOT, 'tools')
    if tools not in sys.path:
        sys.path.insert(0, tools)
    import pybullet_parity
    return pybullet_parity.probe()


def cpu_legs(cfg_kwargs, scene, names, quick=False):
    """Everything that runs the CPU oracle (kind 'port'; whether pybullet, the reference's physics, can be
    imported is asked at run time -- pybullet_probe() -- and its answer is part of the record): the same-workload baseline, the reference-semantics legs, BASELINE
    config 1 (1 env, HeuristicPushPolicy, 20 episodes; one thread and one worker per core as
    tools/parallel_run.py would start them), the pose-level deactivation comparison and the
    FP32-vs-FP64 pose error.

    Sampling (round-4 review): every multi-thread leg runs >= 16 envs per host thread (OpenMP over envs,
    schedule(dynamic): a thread takes the next env when it is done with one) for >= 10 s or 4 env.step()
    per env, the steps after the first in ONE call (no barrier between the steps of an env), so that the
    tail of the slowest env is a few per cent of the window; the same leg on ONE thread gives the
    per-thread cost and the measured 1 -> N scaling."""
    import numpy as np
    from robovat_amd import configs, lib, scenes
    from oracle import orc
    cores = effective_cpus()
    pb_status = pybullet_probe()[1]
    out = {'host': {'cpu_count': os.cpu_count(), 'threads_used': cores,
                    'note': 'threads_used = min(CPU count, affinity mask, cgroup CPU quota): what the container may keep busy'}}
    orc.set_num_threads(cores)
    per_thread = 4 if quick else 16
    n = max(16, per_thread * cores)
    min_seconds, max_steps = (3.0, 2) if quick else (10.0, 4)

    def timed(over, n_envs, threads, max_steps=max_steps):
        """`over` on n_envs envs with `threads` OpenMP threads: reset (untimed), then env.step() calls per env
        until min_seconds or max_steps: one step first (its duration sizes the rest), the others in one call."""
        env_cfg = configs.push_env_config(**over)
        sc, nm = (scenes.make_scene(env_cfg=env_cfg) if 'PHYSICS.ARM_ACCEL_SCALE' in over else (scene, names))
        c = configs.make_rv_config(env_cfg=env_cfg, n_envs=n_envs, shape_names=nm, **cfg_kwargs)
        wc = orc.OracleWorld(c, sc, double=False, native=True)      # (-O3 -march=native, built on this host)
        wc.set_num_threads(threads)
        wc.reset()
        p0, _ = wc.observe()
        c0 = wc.solver_counts()
        tot = {'env_steps': 0, 'substeps': 0, 'awake_substeps': 0, 'useful': 0, 'unsafe': 0, 'ineffective': 0}
        el, done_steps = 0.0, 0
        while done_steps < max_steps and el < min_seconds:
            k = 1 if done_steps == 0 else max(1, min(max_steps - done_steps, int((min_seconds - el) / max(el / done_steps, 1e-9) + 0.999)))
            t0 = time.perf_counter(); wc.rollout(k, first_macro_index=done_steps, auto_reset=False); el += time.perf_counter() - t0
            st = wc.stats()
            for key in tot:
                tot[key] += st[key]
            done_steps += k
        p1, _ = wc.observe()
        c1 = wc.solver_counts()
        wc.set_num_threads(cores)
        es = max(tot['env_steps'], 1)
        moved = np.linalg.norm(p1[..., :2] - p0[..., :2], axis=-1).sum(-1)
        isl = max(c1['island_solves'] - c0['island_solves'], 1)
        return {'value': tot['env_steps'] / el, 'unit': 'env_steps/s', 'sim_steps_per_s': tot['substeps'] / el,
                'awake_sim_steps_per_s': tot['awake_substeps'] / el, 'cores': threads, 'envs': n_envs, 'steps': done_steps,
                'seconds': el, 'thread_us_per_substep': 1e6 * threads * el / max(tot['substeps'], 1),
                'thread_us_per_awake_substep': 1e6 * threads * el / max(tot['awake_substeps'], 1),
                'useful': tot['useful'] / es, 'unsafe': tot['unsafe'] / es, 'ineffective': tot['ineffective'] / es,
                'disp_total_mean_mm': 1e3 * float(moved.mean()), 'substeps_per_env_step': tot['substeps'] / es,
                'mean_sweeps_per_island_solve': (c1['sweeps'] - c0['sweeps']) / isl,
                'mean_rows_per_island': (c1['row_steps'] - c0['row_steps']) / max(c1['sweeps'] - c0['sweeps'], 1)}

    def leg(over, max_steps=max_steps):
        """all host threads, and the same leg on ONE thread (16 envs) for the per-thread cost and the scaling"""
        full = timed(over, n, cores, max_steps)
        one = timed(over, 16 if not quick else 4, 1, max_steps)
        full['one_thread'] = {k: one[k] for k in ('sim_steps_per_s', 'value', 'thread_us_per_substep', 'thread_us_per_awake_substep', 'envs', 'steps', 'seconds')}
        full['scaling_1_to_n'] = full['sim_steps_per_s'] / max(one['sim_steps_per_s'], 1e-9)
        return full

    # (i) same workload and semantics as the GPU headline
    cb = leg({}, max_steps=4 if quick else 48)      # (with deactivation an env.step() costs the host ~3 ms: more steps fill the window)
    out['cpu_baseline'] = {
        'value': cb['value'], 'unit': 'env_steps/s', 'cores': cores, 'kind': 'port',
        'sim_steps_per_s': cb['sim_steps_per_s'], 'awake_sim_steps_per_s': cb['awake_sim_steps_per_s'],
        'sample': '%d envs (%d per thread) x %d env.step() of the same workload in %.1f s, float C oracle, OpenMP over envs with '
                  'schedule(dynamic); reference loop on pybullet: %s' % (n, per_thread, cb['steps'], cb['seconds'], pb_status),
        'thread_us_per_substep': cb['thread_us_per_substep'], 'thread_us_per_awake_substep': cb['thread_us_per_awake_substep'],
        'one_thread': cb['one_thread'], 'scaling_1_to_n': cb['scaling_1_to_n'],
        'mean_sweeps_per_island_solve': cb['mean_sweeps_per_island_solve'],
        'compiler_flags': 'gcc ' + ' '.join(orc.NATIVE_FLAGS) + ' (built on this host; the bit-exact parity target keeps -O2)',
        'note': 'the oracle never coasts: its non-awake substeps still run the arm (light part); thread_us_per_awake_substep charges '
                'them to the awake ones'}
    # (i') the reference's most likely semantics on the host cores (same legs as reference_semantics.gpu): every
    # substep is an awake one, so sim_steps_per_s compares like for like with the MI355X legs
    nd = {'PHYSICS.SLEEP_STEPS': 0}
    bs = {'PHYSICS.SOLVER_TOL': 0.0, 'PHYSICS.SOLVER_STALL': 0}
    out['reference_semantics_cpu'] = {'early_exit_effort_limited_motor': leg(nd)}
    if not quick:
        out['reference_semantics_cpu']['effort_limited_motor'] = leg(dict(nd, **bs))
        out['reference_semantics_cpu']['unlimited_motor'] = leg(dict(nd, **bs, **{'PHYSICS.ARM_ACCEL_SCALE': 1000.0}))
    # (i'') is the shipped deactivation an optimisation within the stated tolerance?  FP64 oracle, shipped semantics vs
    # no deactivation + 50 plain sweeps, one env.step() per env from identical states and actions
    # (tests/test_deactivation_equivalence.py asserts the same bounds as the FP32 tolerance test)
    def pose_equivalence(over, n_envs, seed=21):
        def mk(ov):
            ec = configs.push_env_config(**ov)
            return orc.OracleWorld(configs.make_rv_config(env_cfg=ec, n_envs=n_envs, seed=seed, shape_names=names), scene, double=True)
        a, b = mk({}), mk(over)
        a.reset()
        state, params = a.body_state(), a.body_params()
        a.set_body_state(state)
        b.reset(); b.set_body_params(params); b.set_body_state(state)
        act = a.policy_random(0)
        a.set_actions(act); b.set_actions(act); a.step_macro(); b.step_macro()
        on = params[:, :, 0] > 0
        perr = np.linalg.norm(a.body_state()[..., :3] - b.body_state()[..., :3], axis=-1)[on]
        ca, cb_ = a.env_counters(), b.env_counters()
        return {'env_steps': n_envs, 'median_m': float(np.median(perr)), 'p90_m': float(np.percentile(perr, 90)),
                'p99_m': float(np.percentile(perr, 99)), 'max_m': float(perr.max()),
                'flags_agree': float(((ca[:, 5] == cb_[:, 5]) & (ca[:, 6] == cb_[:, 6])).mean()),
                'bounds': 'median <= 2e-5 m, p90 <= 3e-4 m, flags >= 0.97 (those of test_fp32_tolerance_at_the_end_of_a_push)'}
    pe_n = 256 if quick else max(256, 4 * cores)
    out['deactivation_pose_equivalence'] = {
        'oracle': 'FP64 restatement, shipped semantics vs the other, one env.step() per env from identical states / actions',
        'vs_no_deactivation_50_sweeps': pose_equivalence(dict(nd, **bs), pe_n),
        'vs_no_deactivation': pose_equivalence(nd, pe_n)}
    for v in out['deactivation_pose_equivalence'].values():
        if isinstance(v, dict):
            v['within_bounds'] = bool(v['median_m'] <= 2e-5 and v['p90_m'] <= 3e-4 and v['flags_agree'] >= 0.97)
    # (ii) BASELINE config 1: run_env.py --env PushEnv --policy HeuristicPushPolicy, 20 episodes
    max_steps, episodes = 5, 20

    def config1(n_workers):
        env_cfg = configs.push_env_config(MAX_STEPS=max_steps)
        c1 = configs.make_rv_config(env_cfg=env_cfg, n_envs=n_workers, shape_names=names, seed=0)
        w1 = orc.OracleWorld(c1, scene, double=False, native=True)
        t0 = time.perf_counter()
        steps = sub = 0
        for _ in range(episodes):
            w1.reset(); sub += w1.stats()['substeps']
            for _ in range(max_steps):
                w1.set_actions(w1.policy_heuristic(20000)); w1.step_macro()
                st = w1.stats(); steps += st['env_steps']; sub += st['substeps']
        el = time.perf_counter() - t0
        return {'workers': n_workers, 'episodes_per_worker': episodes, 'env_steps_per_s': steps / el,
                'sim_steps_per_s': sub / el, 'seconds': el}
    out['config1_cpu'] = {'workload': 'PushEnv + HeuristicPushPolicy, 1 env per worker, TASK_NAME=None, MAX_STEPS=%d, %d episodes, '
                                      'float C oracle (pybullet: %s)' % (max_steps, episodes, pb_status),
                          'one_thread': config1(1)}
    if not quick:
        out['config1_cpu']['one_worker_per_core'] = config1(cores)
    return out


def pose_err_leg(scene, names):
    """Pose error of the HIP path (FP32) against the FP64 oracle from identical states (needs the GPU)."""
    import numpy as np
    from robovat_amd import configs, lib
    from oracle import orc
    out = {}
    n = 64
    cfg = configs.make_rv_config(n_envs=n, shape_names=names, seed=9)
    f32, f64 = orc.OracleWorld(cfg, scene, double=False), orc.OracleWorld(cfg, scene, double=True)
    world = lib.World(cfg, scene, device=0)
    f32.reset()
    state, params, joints = f32.body_state(), f32.body_params(), f32.joint_state()
    for x in (f64, world):
        x.set_body_params(params); x.set_joint_state(joints)
    state[:, :, 7] += 0.2
    f64.set_body_state(state); world.set_body_state(state)
    pb, pb_status = pybullet_probe()
    pe, done = {'oracle': 'oracle/rv_oracle.c, -DORC_DOUBLE build (FP64 restatement)',
                'pybullet_parity': pb_status,
                'scene': '%d envs x 4 bodies settled on the table, every body shoved at 0.2 m/s' % n}, 0
    out['pybullet'] = {'status': pb_status}
    if pb is not None:
        # SURVEY 8c last row: identical scenes in PyBullet (createCollisionShape / createMultiBody), HIP path and FP64
        # oracle from the same states; pose error after 1 / 10 / 100 substeps; baseline B1 (stepSimulation timing)
        import pybullet_parity
        try:
            wp, op = lib.World(cfg, scene, device=0), orc.OracleWorld(cfg, scene, double=True)
            for x in (wp, op):
                x.reset(); x.set_body_params(params); x.set_joint_state(joints)
            par = pybullet_parity.pose_parity(pb, cfg, scene, state.astype(np.float64), params, {'hip': wp, 'f64_oracle': op})
            out['pybullet']['pose_err_hip_vs_pybullet'] = par['hip']
            out['pybullet']['pose_err_f64_oracle_vs_pybullet'] = par['f64_oracle']
            wp.close()
            out['pybullet']['step_simulation'] = pybullet_parity.time_step_simulation()
        except Exception as ex:      # noqa: BLE001 -- a wheel of another version may lack a call: report what happened
            out['pybullet']['error'] = '%s: %s' % (type(ex).__name__, ex)

    def err(tag):
        from robovat_amd.math import rotations
        got = world.body_state().cpu().numpy().astype(np.float64); want = f64.body_state()
        perr = np.linalg.norm(got[..., :3] - want[..., :3], axis=-1)
        ang = rotations.quaternion_angle(got[..., 3:7], want[..., 3:7])
        pe[tag] = {'max_pos_m': float(perr.max()), 'p99_pos_m': float(np.percentile(perr, 99)), 'median_pos_m': float(np.median(perr)),
                   'max_angle_rad': float(ang.max()), 'p99_angle_rad': float(np.percentile(ang, 99)),
                   'median_angle_rad': float(np.median(ang))}
    for horizon in (1, 10, 100):
        world.step_sub(horizon - done); f64.step_sub(horizon - done); done = horizon
        err('substeps_%d' % horizon)
    # end of a push: one whole env.step() from identical settled states
    f32.reset()
    state, params, joints = f32.body_state(), f32.body_params(), f32.joint_state()
    w2, r2 = lib.World(cfg, scene, device=0), orc.OracleWorld(cfg, scene, double=True)
    w2.reset(); r2.reset()
    for x in (w2, r2):
        x.set_body_params(params); x.set_body_state(state)
    a = f32.policy_heuristic(2000)
    w2.set_actions(a); r2.set_actions(a); w2.step_macro(); r2.step_macro()
    world, f64 = w2, r2
    err('end_of_push')
    out['pose_err'] = pe
    w2.close()
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=5)
    ap.add_argument('--envs-per-gpu', type=int, default=None,
                    help='default: 1024 at --gpus 1 (BASELINE configs[1]), 8192 at --gpus N > 1 (BASELINE configs[4])')
    ap.add_argument('--seed', type=int, default=1234)
    ap.add_argument('--no-cpu-baseline', action='store_true', help='skip every CPU-oracle leg')
    ap.add_argument('--no-extra-legs', action='store_true',
                    help='only the headline: skip lockstep_env_step, async_rollout, config3_4096, config5_8192')
    ap.add_argument('--no-async', action='store_true')
    ap.add_argument('--quick', action='store_true', help='shorter CPU legs')
    ap.add_argument('--mode', choices=['rollout', 'lockstep'], default='rollout')
    ap.add_argument('--workload', choices=['config2', 'config3', 'config4', 'config5'], default='config2',
                    help='the timed workload (BASELINE.json configs[1..4]); other than config2: for profiling one of the legs '
                         'alone (tools/profile_bench.sh), implies --no-extra-legs')
    ap.add_argument('--over', nargs='*', default=[], help='config overrides of the timed workload, KEY=VALUE (profiling aid: e.g. '
                    'PHYSICS.SLEEP_STEPS=0 times the no-deactivation launch alone; implies --no-extra-legs)')
    ap.add_argument('--n1-reference', action='store_true', help='take the same-workload one-GPU reference of the N > 1 line also at N = 1 (exercises that code path on a one-GPU box)')
    ap.add_argument('--limb-legs', action='store_true', help='also time PHYSICS.LIMB_DYNAMICS=1 (optional mode, SURVEY 8 f1) on the headline and the grasp workload')
    ap.add_argument('--legs-out', default=os.path.join(ROOT, 'bench_legs.json'), help='where the full record (every leg, every note) is written')
    ap.add_argument('--extra-legs', action='store_true', help='run the extra legs at --gpus N > 1 as well (default: N = 1 only -- '
                    'at 8192 envs per rank the no-deactivation legs alone take minutes)')
    args = ap.parse_args()
    headline_only_run = args.no_extra_legs or args.workload != 'config2' or bool(args.over)      # (asked for on the command line: a profiling run)
    if args.gpus > 1 and not args.extra_legs:
        args.no_extra_legs = True

    # stdout carries exactly ONE line, the JSON: libraries that print to the C-level stdout (RCCL's
    # version banner, flushed at exit) are sent to stderr for the whole run
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)

    import torch
    from robovat_amd import configs, scenes, lib

    world_size = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    # The RCCL group is created at EVERY world size -- also 1 -- so that the N = 1 point of a scaling run
    # executes exactly the code path N = 8 does (barriers, max-over-ranks timing and the return gather
    # inside the timed region).  Under torch.distributed.run the rendezvous comes from the environment;
    # a bare `python bench.py` makes its own one-rank group on a free local port.
    import torch.distributed as dist
    dist_note = None
    if world_size == 1:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    own_port = 'MASTER_PORT' not in os.environ
    os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
    for attempt in range(4):
        try:
            if own_port:      # (found by binding to port 0 and closing: somebody may take it before the store does -> try another one)
                import socket
                with socket.socket() as sk:
                    sk.bind(('127.0.0.1', 0))
                    os.environ['MASTER_PORT'] = str(sk.getsockname()[1])
            dist.init_process_group(backend='nccl', rank=rank, world_size=world_size, device_id=torch.device('cuda', local_rank))
            break
        except Exception as ex:  # noqa: BLE001 -- only a one-rank run may go on without its group
            if world_size > 1:
                raise
            if attempt == 3:
                dist_note = 'RCCL group not created: %r' % (ex,)
                dist = None
    assert world_size == args.gpus, 'launch with torch.distributed.run --nproc-per-node %d' % args.gpus

    scene, names = scenes.make_scene()
    base_over, grasp_env = {}, None
    if args.workload != 'config2':
        args.no_extra_legs = True
        args.envs_per_gpu = args.envs_per_gpu or {'config3': 4096, 'config4': 2048, 'config5': 8192}[args.workload]
        if args.workload == 'config3':
            base_over = dict(TASK_NAME='crossing', LAYOUT_ID=0, MOVABLE_NAME='CONCAVE', MAX_STEPS=10)
        if args.workload == 'config4':
            grasp_env = configs.grasp_env_config()
            scene, names = scenes.make_scene(env_cfg=grasp_env)
    if args.envs_per_gpu is None:
        args.envs_per_gpu = 1024 if args.gpus == 1 else 8192
    n = args.envs_per_gpu
    workload = ("PushEnv 'crossing' layout 0, V-HACD concave movables, %d envs/GPU, random policy (BASELINE.json configs[2])" % n
                if args.workload == 'config3' else
                'Grasp4DofEnv, %d envs/GPU, random CUBOID grasps (BASELINE.json configs[3])' % n if args.workload == 'config4' else
                'PushEnv 4 rigid convex bodies, %d vectorised envs/GPU, random policy (BASELINE.json configs[1])' % n if n != 8192 else
                'PushEnv 4 rigid convex bodies, 8192 envs/GPU sharded across %d GPU(s), random policy, RCCL return-gather '
                '(BASELINE.json configs[4])' % args.gpus)
    cfg_kwargs = dict(seed=args.seed)
    for kv in args.over:
        key, val = kv.split('=', 1)
        try:
            val = int(val)
        except ValueError:
            try:
                val = float(val)
            except ValueError:
                pass
        base_over[key] = val
    if args.over:
        args.no_extra_legs = True
        workload += ' + overrides ' + ' '.join(args.over)

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    def make_world(n_envs, **over):
        env_cfg = grasp_env if (grasp_env is not None and not over) else configs.push_env_config(**over)
        c = configs.make_rv_config(env_cfg=env_cfg, n_envs=n_envs, env_id_offset=rank * n_envs, shape_names=names, **cfg_kwargs)
        return lib.World(c, scene, device=local_rank), c

    def gather_returns(world):
        """The only collective of the path: RCCL all-gather of episode returns + all-reduce of
        4 counters (robovat_amd/parallel.py)."""
        if dist is None:
            return
        from robovat_amd import parallel
        cnt = world.env_counters().to(torch.int64)
        st = torch.stack([cnt[:, 5].sum(), cnt[:, 6].sum(), cnt[:, 4].sum(), cnt[:, 1].sum()])
        parallel.gather_returns(world.episode_returns(), st)

    def lockstep_step(world, k):
        """The literal host-driven env.step() for the whole batch."""
        world.set_actions(world.policy_random(k))
        world.step_macro()
        obs = world.observe(point_cloud=True)
        r, d = world.reward()
        gather_returns(world)
        return obs, r, d

    def all_sum(*vals):
        t = torch.tensor([float(v) for v in vals], dtype=torch.float64, device='cuda')
        if dist is not None:
            dist.all_reduce(t)
        return [float(x) for x in t]

    def all_max(v):
        t = torch.tensor([float(v)], dtype=torch.float64, device='cuda')
        if dist is not None:
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    def time_rollout(world, k_steps, first):
        """K env.step() per env in one launch, every step's observation recorded."""
        barrier()
        t0 = time.perf_counter()
        obs, r, d = world.rollout_record(k_steps, first_macro_index=first, auto_reset=True, point_cloud=True)
        gather_returns(world)
        st = world.stats()
        barrier()
        el = all_max(time.perf_counter() - t0)
        return el, st, world.last_kernel_ms(), 1

    def time_lockstep(world, k_steps, first):
        barrier()
        t0 = time.perf_counter()
        kern_ms, tot = 0.0, {'substeps': 0, 'env_steps': 0, 'max_substeps': 0, 'awake_substeps': 0}
        for k in range(k_steps):
            lockstep_step(world, first + k)
            st = world.stats()     # synchronises, as a Python env loop that reads rewards does
            kern_ms += world.last_kernel_ms()
            for key in ('substeps', 'env_steps', 'awake_substeps'):
                tot[key] += st[key]
            tot['max_substeps'] = max(tot['max_substeps'], st['max_substeps'])
        barrier()
        el = all_max(time.perf_counter() - t0)
        return el, tot, kern_ms, k_steps

    # ---- semantics legs: the same workload / seed / actions under other physics settings -------------------
    # What the reference most likely runs (RECALLED PyBullet behaviour, DESIGN.md section 3, table of recalled
    # defaults): bullet_physics.py:173-181 loads every body WITHOUT URDF_ENABLE_SLEEPING (no deactivation),
    # Bullet sweeps 50 times without an early exit, and controllable_body.py:458-466 -> bullet_physics.py:1061-1104
    # passes no `forces=` to setJointMotorControlArray, whose default maximum motor force is either the URDF joint
    # effort (the shipped acceleration limits) or a large constant (the motors then reach
    # the commanded velocity within a step: ARM_ACCEL_SCALE = 1000).  Both motor variants are first-class legs.
    NO_DEACT = {'PHYSICS.SLEEP_STEPS': 0}
    BULLET_SWEEPS = {'PHYSICS.SOLVER_TOL': 0.0, 'PHYSICS.SOLVER_STALL': 0}
    SEMANTICS = [
        ('shipped', {}),
        ('bullet_rule_alone', {'PHYSICS.SLEEP_LINEAR': 0.8, 'PHYSICS.SLEEP_ANGULAR': 1.0, 'PHYSICS.SLEEP_STEPS': 2000,
                               'PHYSICS.SLEEP_POSITION_WINDOW': 0.0, 'PHYSICS.DEACTIVATION_STEPS': 0}),
        ('no_deactivation', dict(NO_DEACT)),                # (without deactivation resting islands converge to SOLVER_TOL_REST = 1e-7: configs.py)
        ('no_deactivation_one_tolerance', dict(NO_DEACT, **{'PHYSICS.SOLVER_TOL_REST': 0.0})),   # round 3's behaviour: resting bodies creep
        ('no_deactivation_50_sweeps', dict(NO_DEACT, **BULLET_SWEEPS)),
        ('no_deactivation_50_sweeps_unlimited_motor', dict(NO_DEACT, **BULLET_SWEEPS, **{'PHYSICS.ARM_ACCEL_SCALE': 1000.0})),
    ]

    def outcome_stats(st, pos):
        import numpy as np
        moved = np.linalg.norm(pos[1:, ..., :2] - pos[:-1, ..., :2], axis=-1).sum(-1)
        es = max(st['env_steps'], 1)
        return {'useful': st['useful'] / es, 'unsafe': st['unsafe'] / es, 'ineffective': st['ineffective'] / es,
                'disp_mean_mm': 1e3 * float(moved.mean()), 'disp_p50_mm': 1e3 * float(np.percentile(moved, 50)),
                'disp_p90_mm': 1e3 * float(np.percentile(moved, 90)), 'disp_p99_mm': 1e3 * float(np.percentile(moved, 99)),
                'awake_substep_fraction': st['awake_substeps'] / max(st['substeps'], 1),
                'substeps_per_env_step': st['substeps'] / es}

    def gpu_semantics_leg(over, n_envs, k_steps):
        env_cfg = configs.push_env_config(**over)
        sc, nm = (scenes.make_scene(env_cfg=env_cfg) if 'PHYSICS.ARM_ACCEL_SCALE' in over else (scene, names))
        c = configs.make_rv_config(env_cfg=env_cfg, n_envs=n_envs, env_id_offset=rank * n_envs, shape_names=nm, **cfg_kwargs)
        w = lib.World(c, sc, device=local_rank)
        w.reset()
        pos0 = w.observe()['position']
        barrier(); t0 = time.perf_counter()
        obs, r, d = w.rollout_record(k_steps, first_macro_index=0, auto_reset=False, point_cloud=False)
        st = w.stats()
        barrier(); el = all_max(time.perf_counter() - t0)
        pos = torch.cat([pos0[None], obs['position']], 0).cpu().numpy()
        out = {'value': all_sum(st['env_steps'])[0] / el, 'unit': 'env_steps/s', 'sim_steps_per_s': all_sum(st['substeps'])[0] / el,
               'awake_sim_steps_per_s': all_sum(st['awake_substeps'])[0] / el, 'envs': n_envs, 'steps': k_steps,
               'kernel_ms': w.last_kernel_ms()}
        out.update(outcome_stats(st, pos))
        w.close()
        return out

    def semantics_legs(n_envs, k_steps, quick=False):
        legs = {}
        for name, over in SEMANTICS:
            if quick and '50_sweeps' in name:
                continue
            # (the 50-sweep legs run ~50 x slower than the headline: half the steps keep the default run short)
            legs[name] = gpu_semantics_leg(over, n_envs, max(2, k_steps // 2) if '50_sweeps' in name else k_steps)
        return legs

    def leg_summary(el, st, k_steps, n_envs):
        es, ss = all_sum(st['env_steps'], st['substeps'])
        return {'value': es / el, 'unit': 'env_steps/s', 'sim_steps_per_s': ss / el, 'ms_per_step': 1e3 * el / k_steps,
                'envs_per_gpu': n_envs, 'steps': k_steps}

    world, cfg = make_world(n, **base_over)
    world.reset()
    reset_stats = world.stats()
    if args.workload == 'config2':
        for k in range(args.warmup):
            lockstep_step(world, k)
    else:
        world.rollout(args.warmup, first_macro_index=0, auto_reset=True, record=True)
    timer = time_rollout if args.mode == 'rollout' else time_lockstep
    # N > 1: the same per-GPU workload on ONE GPU of this very run -- rank 0 runs its shard alone while the other ranks wait
    # at the barrier -- so that the N-GPU line carries a like-for-like N = 1 reference (the driver's own N = 1 run is
    # BASELINE configs[1], 1024 envs: a different workload from the 8192 envs per GPU of configs[4])
    n1_same = None
    if world_size > 1 or args.n1_reference:
        barrier()
        if rank == 0:
            torch.cuda.synchronize(); t1 = time.perf_counter()
            if args.mode == 'rollout':
                world.rollout_record(args.steps, first_macro_index=args.warmup, auto_reset=True, point_cloud=True)
            else:
                for k in range(args.steps):
                    world.set_actions(world.policy_random(args.warmup + k)); world.step_macro(); world.observe(point_cloud=True); world.reward()
            st1 = world.stats()
            torch.cuda.synchronize()
            n1_same = st1['env_steps'] / (time.perf_counter() - t1)
        barrier()
        first_timed = args.warmup + args.steps
    else:
        first_timed = args.warmup
    elapsed, st, kern_ms, launches = timer(world, args.steps, first_timed)
    env_steps_all, substeps_all = all_sum(st['env_steps'], st['substeps'])
    next_index = first_timed + args.steps

    extra = {}
    if dist_note:
        extra['dist_note'] = dist_note
    if not args.no_extra_legs:
        # BASELINE.md's config 2 names 50 macro-steps per env: the same single-launch rollout with K = 50
        el50, st50, km50, _ = time_rollout(world, 50, next_index); next_index += 50
        extra['config2_k50'] = leg_summary(el50, st50, 50, n)
        extra['config2_k50'].update({'kernel_ms': km50, 'roofline_frac_nominal': ALGO_BYTES['config2'] * st50['substeps'] / (1e-3 * km50) / 1e9 / HBM_PEAK_GBS})
        other = time_lockstep if args.mode == 'rollout' else time_rollout
        el2, st2, km2, _ = other(world, args.steps, next_index); next_index += args.steps
        name = 'lockstep_env_step' if args.mode == 'rollout' else 'single_launch_rollout'
        extra[name] = leg_summary(el2, st2, args.steps, n)
        extra[name]['kernel_ms_per_step'] = km2 / args.steps
        extra[name]['note'] = ('host loop: K x (policy -> rv_set_actions -> rv_step_macro -> rv_observe incl. point cloud -> rv_reward); '
                               'every step waits for the slowest env of the batch' if args.mode == 'rollout' else
                               'one rv_rollout_record launch, observations of every step recorded')
        # host-driven, but without the lock step: rv_step_begin / rv_step_poll (EnvPool style).  Every
        # poll runs the stepping envs for ~poll_usec of GPU time and hands back observation (incl. point
        # cloud), reward and done of the envs whose env.step() completed; the host draws their next
        # action and starts them again.  K * N env.step() calls in all, no env waits for another.
        barrier()
        poll_usec = 1500
        A = torch.stack([world.policy_random(next_index + k) for k in range(4 * args.steps)])     # [4K, N, G, 4]
        out_buf = world.poll_buffers(point_cloud=True)
        cnt = torch.zeros(n, dtype=torch.long, device='cuda'); ar = torch.arange(n, device='cuda')
        # (the device ops of the loop once, untimed: on a fresh machine the first use of a torch kernel loads its code
        # object -- ~0.2 s for the handful used here, which would be charged to the 0.4 s leg)
        _live = torch.ones(n, dtype=torch.uint8, device='cuda') * (1 - out_buf['done'])
        _ = A[(cnt + _live.to(torch.long)).clamp(max=A.shape[0] - 1), ar]
        del _live, _
        barrier()
        tp = time.perf_counter()
        world.step_begin(A[0])
        done_steps, polls, sub_p = 0, 0, 0
        idle = 0
        t_poll = t_kern = 0.0
        amax = A.shape[0] - 1
        # the host looks at ONE number per poll (how many env.step() calls completed: rv_get_stats, which also
        # paces the loop); the bookkeeping -- per-env step counts, who is restarted, which action each env gets --
        # is a handful of asynchronous device ops, as the action selection of a policy network would be
        while done_steps < args.steps * n and idle < 50:
            tq = time.perf_counter()
            fin = world.step_poll(max_usec=poll_usec, out=out_buf)
            polls += 1
            stp = world.stats()
            nf = stp['env_steps']
            t_poll += time.perf_counter() - tq; t_kern += world.last_kernel_ms()
            sub_p += stp['substeps']
            idle = idle + 1 if nf == 0 else 0
            if nf == 0:
                continue
            done_steps += nf
            live = fin * (1 - out_buf['done'])        # an env whose episode ended stops, as in the lock-step leg (which has no reset either)
            cnt += live.to(torch.long)
            world.step_begin(A[cnt.clamp(max=amax), ar], mask=live)
        barrier()
        ep = all_max(time.perf_counter() - tp)
        next_index += 4 * args.steps
        extra['lockstep_partial'] = {'value': all_sum(done_steps)[0] / ep, 'unit': 'env_steps/s', 'sim_steps_per_s': all_sum(sub_p)[0] / ep,
                                     'polls': polls, 'poll_usec': poll_usec, 'envs_per_gpu': n,
                                     'ms_per_poll_call': 1e3 * t_poll / max(polls, 1), 'kernel_ms_per_poll': t_kern / max(polls, 1),
                                     'ms_per_poll_loop': 1e3 * ep / max(polls, 1),
                                     'steps_per_env_min_max': [int(cnt.min()), int(cnt.max())],
                                     'note': 'host loop over rv_step_begin / rv_step_poll: K*N env.step() calls, each poll returns the '
                                             'observation (incl. point cloud), reward and done of the envs that finished; per-env '
                                             'trajectories are those of rv_step_macro (tests/test_gpu_parity.py)'}
        # drain the steps still in flight so that the next legs start from whole steps
        world.step_poll()                             # (no budget: every pending step runs to its end)
        if not args.no_async:
            barrier()
            ta = time.perf_counter()
            taken = world.rollout_async(args.steps * n, first_macro_index=next_index)
            barrier()
            ea = all_max(time.perf_counter() - ta)
            extra['async_rollout'] = leg_summary(ea, world.stats(), args.steps, n)
            extra['async_rollout'].update({
                'steps_per_env_min_max': [int(taken.min().item()), int(taken.max().item())],
                'note': 'rv_rollout_async: K*N env.step() calls shared by the N envs of each GPU (work-conserving, no '
                        'observations recorded); per-env step counts vary'})
        world.close()
        # (the other BASELINE configurations first, the slow semantics legs after them)
        # BASELINE configs[2]: 'crossing' layout, V-HACD concave movables, 4096 envs
        w3, _ = make_world(4096, TASK_NAME='crossing', LAYOUT_ID=0, MOVABLE_NAME='CONCAVE', MAX_STEPS=10)
        w3.reset()
        k3 = min(args.steps, 10)
        el3, st3, km3, _ = time_rollout(w3, k3, 0)
        extra['config3_4096'] = leg_summary(el3, st3, k3, 4096)
        extra['config3_4096'].update({'workload': "PushEnv 'crossing' layout 0, V-HACD concave movables, PushReward, MAX_STEPS=10",
                                      'roofline_frac_nominal': ALGO_BYTES['config3'] * st3['substeps'] / (1e-3 * km3) / 1e9 / HBM_PEAK_GBS})
        w3.close()
        # BASELINE configs[4], per-GPU point: config-2 scene, 8192 envs per GPU
        # (the driver's N > 1 command: --steps 20 --warmup 5 at 8192 envs per GPU; rollouts of that size go through the task
        # queue of rv_env_kernel.h, which needs some steps per env to pay: this leg runs args.steps after args.warmup too)
        w5, _ = make_world(8192)
        w5.reset()
        w5.rollout(args.warmup, first_macro_index=0, auto_reset=True, record=True)
        el5, st5, km5, _ = time_rollout(w5, args.steps, args.warmup)
        extra['config5_8192'] = leg_summary(el5, st5, args.steps, 8192)
        extra['config5_8192'].update({'workload': 'config-2 scene, 8192 envs per GPU, %d steps after %d warm-up steps (task queue: one env.step() per task)' % (args.steps, args.warmup),
                                      'roofline_frac_nominal': ALGO_BYTES['config2'] * st5['substeps'] / (1e-3 * km5) / 1e9 / HBM_PEAK_GBS})
        ea5 = None
        if not args.no_async:
            barrier(); ta = time.perf_counter()
            w5.rollout_async(k3 * 8192, first_macro_index=args.warmup + args.steps)
            barrier(); ea5 = all_max(time.perf_counter() - ta)
            extra['config5_8192']['async_value'] = all_sum(w5.stats()['env_steps'])[0] / ea5
        w5.close()
        # BASELINE configs[3]: Grasp4DofEnv, 2048 envs, one graspable, force-limited gripper, random CUBOID grasps
        genv = configs.grasp_env_config()
        gscene, gnames = scenes.make_scene(env_cfg=genv)
        gc = configs.make_rv_config(env_cfg=genv, n_envs=2048, env_id_offset=rank * 2048, shape_names=gnames, **cfg_kwargs)
        w4 = lib.World(gc, gscene, device=local_rank)
        w4.reset()
        barrier(); t4 = time.perf_counter()
        w4.rollout(k3, first_macro_index=0, auto_reset=True, record=True)
        st4 = w4.stats()
        barrier(); el4 = all_max(time.perf_counter() - t4)
        extra['config4_grasp_2048'] = leg_summary(el4, st4, k3, 2048)
        extra['config4_grasp_2048'].update({
            'workload': 'Grasp4DofEnv, ACTION.TYPE=CUBOID random grasps, one graspable hull, 7-DoF FK / DLS IK every 10 substeps, '
                        'two force-limited prismatic fingers in the contact solver; every env.step() is a whole episode (reset included)',
            'grasp_success_rate': st4['successes'] / max(st4['env_steps'], 1),
            'roofline_frac_nominal': 1912 * st4['substeps'] / (1e-3 * w4.last_kernel_ms()) / 1e9 / HBM_PEAK_GBS})
        w4.close()
        # Deactivation semantics.  The reference loads its movables with
        # flags=URDF_USE_SELF_COLLISION_EXCLUDE_PARENT only (bullet_physics.py:173-181): no
        # URDF_ENABLE_SLEEPING, so PyBullet most likely never deactivates them.  Same workload,
        # same seed, same actions with (a) the shipped rule, (b) Bullet's own rule alone, (c) no
        # deactivation at all [+ Bullet's fixed 50 solver sweeps]: outcome statistics and rate.
        legs = semantics_legs(n, min(args.steps, 20), quick=args.quick)
        deact = {'note': 'same workload / seed / actions, one rv_rollout_record launch without auto-reset; displacement = sum over the '
                         'bodies of an env of the xy distance moved by one env.step(), mm.  bullet_physics.py:173-181 passes no '
                         'URDF_ENABLE_SLEEPING: the no_deactivation legs are the closest to the reference (see reference_semantics).  Without '
                         'deactivation islands at rest converge to 1e-7 N s (SOLVER_TOL_REST); no_deactivation_one_tolerance = round 3 '
                         '(1e-5 N s for every island: resting bodies creep, disp_p50_mm > 0)'}
        for k in ('shipped', 'bullet_rule_alone', 'no_deactivation', 'no_deactivation_one_tolerance', 'no_deactivation_50_sweeps'):
            if k in legs:
                deact[k] = legs[k]
        ref_name = 'no_deactivation_50_sweeps' if 'no_deactivation_50_sweeps' in legs else 'no_deactivation'
        sh, ref = legs['shipped'], legs[ref_name]
        deact['shipped_vs_reference_semantics'] = {
            'reference_leg': ref_name, 'disp_mean_ratio': sh['disp_mean_mm'] / max(ref['disp_mean_mm'], 1e-9),
            'useful_diff': sh['useful'] - ref['useful'], 'unsafe_diff': sh['unsafe'] - ref['unsafe'],
            'ineffective_diff': sh['ineffective'] - ref['ineffective']}
        nd, nd50 = legs['no_deactivation'], legs.get('no_deactivation_50_sweeps')
        if nd50 is not None:
            deact['early_exit_vs_50_sweeps_without_deactivation'] = {
                'disp_mean_ratio': nd['disp_mean_mm'] / max(nd50['disp_mean_mm'], 1e-9),
                'useful_diff': nd['useful'] - nd50['useful'], 'ineffective_diff': nd['ineffective'] - nd50['ineffective']}
        extra['deactivation'] = deact
        # the reference's most likely semantics as first-class legs, MI355X and (below, cpu_legs) host cores
        extra['reference_semantics'] = {
            'note': 'no deactivation, 50 solver sweeps without early exit (RECALLED Bullet defaults; DESIGN.md section 3) with both '
                    'readings of the POSITION_CONTROL default maximum force: effort_limited_motor = the URDF joint efforts (shipped '
                    'acceleration limits), unlimited_motor = commanded velocity reached within a step.  Every substep is an awake one: '
                    'sim_steps_per_s is the like-for-like rate against the cpu legs of the same name',
            'gpu': {'effort_limited_motor': legs.get('no_deactivation_50_sweeps'),
                    'unlimited_motor': legs.get('no_deactivation_50_sweeps_unlimited_motor'),
                    'early_exit_effort_limited_motor': legs['no_deactivation']},
            'headline_semantics': {k: sh[k] for k in ('value', 'useful', 'unsafe', 'ineffective', 'disp_mean_mm', 'awake_substep_fraction')}}
        # ... and at 8192 envs per GPU (BASELINE configs[4]'s size: the two-waves-per-SIMD build, eight envs per SIMD to
        # balance the tail of the slowest env away) -- where a throughput run of these semantics belongs
        if not args.quick:
            # 8 steps per env from a reset, and the first 2 alone (what rounds 4 / 5 quoted).  They differ by a factor of
            # two, and not because of the GPU: without deactivation an env that has pushed a body against / onto another one
            # keeps an island that needs all its sweeps in EVERY substep for the rest of the episode (with deactivation it
            # falls asleep); the mean env costs the same per substep at every step of an episode, the slowest ones get 3 - 4 x
            # slower, and a launch in which every env takes the same number of steps lasts as long as its slowest env
            # (tools/nd_throttle_check.py: 2-step launches in a row 1.98 -> 2.47 -> 3.19 -> 3.78 s, a fresh world 1.97 s again)
            extra['reference_semantics']['gpu_8192'] = {
                'early_exit_effort_limited_motor': gpu_semantics_leg(dict(NO_DEACT), 8192, 8),
                'effort_limited_motor': gpu_semantics_leg(dict(NO_DEACT, **BULLET_SWEEPS), 8192, 4)}      # (4 steps: 11 s of GPU time as it is)
            # ... and work-conserving (rv_rollout_async: the envs share a pool of env.step() calls, each takes its next one
            # while the pool lasts, episodes reset as they end) -- how the reference runs many envs: independent worker
            # processes, each at its own pace (tools/parallel_run.py:54-90).  Nobody waits for the slowest env; the price
            # is the mix: an env in a slow state contributes fewer steps (steps_per_env says how uneven it was)
            def async_leg(over, per_env):
                c = configs.make_rv_config(env_cfg=configs.push_env_config(**over), n_envs=8192, env_id_offset=rank * 8192, shape_names=names, **cfg_kwargs)
                w = lib.World(c, scene, device=local_rank)
                w.reset()
                barrier(); t0 = time.perf_counter()
                taken = w.rollout_async(per_env * 8192, first_macro_index=0)
                st = w.stats()
                barrier(); el = all_max(time.perf_counter() - t0)
                tk = taken.cpu().numpy()
                out = {'value': all_sum(st['env_steps'])[0] / el, 'unit': 'env_steps/s', 'sim_steps_per_s': all_sum(st['substeps'])[0] / el,
                       'envs': 8192, 'pool': per_env * 8192, 'kernel_ms': w.last_kernel_ms(), 'episodes_done': st['episodes_done'],
                       'steps_per_env': {'min': int(tk.min()), 'p10': float(np.percentile(tk, 10)), 'median': float(np.median(tk)), 'max': int(tk.max())}}
                w.close()
                return out
            import numpy as np
            extra['reference_semantics']['gpu_8192_work_conserving'] = {
                'early_exit_effort_limited_motor': async_leg(dict(NO_DEACT), 8),
                'effort_limited_motor': async_leg(dict(NO_DEACT, **BULLET_SWEEPS), 4)}
            extra['reference_semantics']['gpu_8192_first_2_steps'] = {
                'early_exit_effort_limited_motor': gpu_semantics_leg(dict(NO_DEACT), 8192, 2),
                'effort_limited_motor': gpu_semantics_leg(dict(NO_DEACT, **BULLET_SWEEPS), 8192, 2)}
        # (the kinematic pusher is the model -- DESIGN.md section 3 item 12; the optional contact-time limb dynamics
        # stays a tested mode of the library and is timed only on request)
        if args.limb_legs:
            # the same two workloads with the dynamic limb (PHYSICS.LIMB_DYNAMICS: joint-space inertia,
            # contact Jacobians and effort-limited motor rows of the seven joints in the solve of every substep in
            # which the arm touches an awake body) -- cost and outcome next to the kinematic limb
            limb = {'note': 'LIMB_DYNAMICS=1 vs the shipped kinematic limb (the headline and config4_grasp_2048 above): same seeds and actions'}
            wl, _ = make_world(n, **{'PHYSICS.LIMB_DYNAMICS': 1})
            wl.reset()
            barrier(); tl = time.perf_counter()
            wl.rollout(args.steps, first_macro_index=args.warmup, auto_reset=True, record=True)
            stl = wl.stats()
            barrier(); ell = all_max(time.perf_counter() - tl)
            limb['push_%d' % n] = leg_summary(ell, stl, args.steps, n)
            limb['push_%d' % n].update({k: stl[k] / max(stl['env_steps'], 1) for k in ('useful', 'unsafe', 'ineffective')})
            limb['push_%d' % n]['kinematic'] = {k: st[k] / max(st['env_steps'], 1) for k in ('useful', 'unsafe', 'ineffective')}
            wl.close()
            genv = configs.grasp_env_config(**{'PHYSICS.LIMB_DYNAMICS': 1})
            gc = configs.make_rv_config(env_cfg=genv, n_envs=2048, env_id_offset=rank * 2048, shape_names=gnames, **cfg_kwargs)
            wl = lib.World(gc, gscene, device=local_rank)
            wl.reset()
            barrier(); tl = time.perf_counter()
            wl.rollout(k3, first_macro_index=0, auto_reset=True, record=True)
            stl = wl.stats()
            barrier(); ell = all_max(time.perf_counter() - tl)
            limb['grasp_2048'] = leg_summary(ell, stl, k3, 2048)
            limb['grasp_2048']['grasp_success_rate'] = stl['successes'] / max(stl['env_steps'], 1)
            limb['grasp_2048']['kinematic_grasp_success_rate'] = extra['config4_grasp_2048']['grasp_success_rate']
            wl.close()
            extra['limb_dynamics'] = limb
        # the path's only collective, alone: one RCCL all-gather of returns f32[8192] + all-reduce of 4
        # int64 counters on this run's group (the 8-rank curve is the driver's to measure)
        if dist is not None:
            from robovat_amd import parallel
            ret = torch.zeros(8192, dtype=torch.float32, device='cuda'); cnt4 = torch.zeros(4, dtype=torch.int64, device='cuda')
            for _ in range(5):
                parallel.gather_returns(ret, cnt4)
            torch.cuda.synchronize(); tg = time.perf_counter()
            for _ in range(100):
                parallel.gather_returns(ret, cnt4)
            torch.cuda.synchronize()
            extra['gather_returns_alone'] = {'us_per_call': 1e4 * (time.perf_counter() - tg), 'ranks': world_size, 'backend': 'nccl (RCCL)',
                                             'payload': 'all-gather f32[8192] + all-reduce int64[4]'}
    else:
        world.close()

    world_n_waves = n          # one wave64 per env
    def roofline(achieved, awake_only, traffic, issue, avg_kernel_s, algo, per_launch, n_waves):
        """`bound` names the resource that was MEASURED to bind (round-4 review): the VALU issue rate of the resident waves.
        achieved = VALU instructions the launch issues (PMC count per env-substep of the same command, profiles/traffic.json,
        x the env-substeps of THIS launch) / (SIMD-cycles the waves were resident: waves x kernel time x shader clock); peak =
        one wave64 VALU instruction per 2 cycles per SIMD (MI355X_MICROARCH.md).  The contract's nominal HBM figure
        (algorithmic bytes x all env-substeps / kernel time) stays under `hbm_nominal`."""
        hbm = {'bound': 'hbm', 'achieved': achieved, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s', 'frac': achieved / HBM_PEAK_GBS,
               'algorithmic_bytes_per_env_substep': algo, 'achieved_awake_substeps_only': awake_only,
               'note': 'nominal figure of the contract: algorithmic bytes x ALL env-substeps / kernel time.  The state is LDS-resident '
                       'for the whole launch, so the measured HBM traffic (`traffic`) is ~1 % of it, and ~96 % of the substeps are '
                       'coasting / quiet ones that touch no body (`achieved_awake_substeps_only` counts the rest): HBM does not bind'}
        kern = 'k_env<MODE_ROLLOUT>' if args.mode == 'rollout' else 'k_env<MODE_MACRO>'
        if args.mode == 'rollout' and n_waves > 1024:      # more envs than SIMDs: the 256-register build; >= 10 steps and more envs than wave slots: through the task queues
            kern = 'k_env_occ2<-1> (per-XCD task queues)' if (n_waves > 2048 and args.steps >= 10) else 'k_env_occ2<MODE_ROLLOUT>'
        if not issue:
            hbm.update({'traffic': traffic, 'kernel': kern, 'avg_kernel_ms': 1e3 * avg_kernel_s, 'hbm_nominal': dict(hbm)})
            return hbm
        clk = issue.get('shader_clock_hz', 2.4e9)
        simds = min(n_waves, 1024)
        valu = issue['valu_insts_per_env_substep'] * per_launch
        a = valu / (simds * avg_kernel_s * clk)
        return {'bound': 'valu_issue', 'achieved': a, 'peak': VALU_PEAK_PER_SIMD_CYCLE, 'unit': 'VALU instructions / SIMD cycle',
                'frac': a / VALU_PEAK_PER_SIMD_CYCLE, 'traffic': traffic, 'kernel': kern, 'avg_kernel_ms': 1e3 * avg_kernel_s,
                'valu_insts_per_env_substep': issue['valu_insts_per_env_substep'], 'env_substeps_per_launch': per_launch,
                'occupied_simds': simds, 'shader_clock_hz': clk,
                'issue_side': issue, 'hbm_nominal': hbm,
                'note': 'the kernel is bound by the issue rate / dependent-instruction latency of its resident waves (one wave64 per env; '
                        'one wave per SIMD at 1024 envs), not by HBM: `frac` = VALU instructions issued per SIMD cycle over the launch / '
                        'the 0.5 per cycle a SIMD can issue (the instruction count per env-substep is the rocprofv3 SQ_INSTS_VALU of '
                        'this same command, profiles/; the duration is measured live with HIP events).  `issue_side` has the wave-resident '
                        'view of the same counters (idle SIMDs behind the slowest env excluded); DESIGN.md section 4'}

    if rank == 0:
        algo = {'config3': ALGO_BYTES['config3'], 'config4': 1912}.get(args.workload, ALGO_BYTES['config2'])
        avg_kernel_s = 1e-3 * kern_ms / launches
        per_launch = st['substeps'] / launches
        achieved = algo * per_launch / avg_kernel_s / 1e9
        awake_only = algo * (st['awake_substeps'] / launches) / avg_kernel_s / 1e9
        traffic, issue = None, None
        tpath = os.path.join(ROOT, 'profiles', 'traffic.json')
        if os.path.exists(tpath):
            with open(tpath) as f:
                tj = json.load(f)
            if args.mode == 'rollout' and args.workload == 'config2':
                # measured offline with rocprofv3 PMC passes on this same command, scaled to this launch
                traffic = tj['hbm_bytes_per_env_substep'] * per_launch
                issue = tj.get('issue')
        out = {
            'metric': 'env steps/sec (PushEnv, batched)',
            'value': env_steps_all / elapsed,
            'unit': 'env_steps/s',
            'n_gpus': world_size, 'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': 1e3 * elapsed / args.steps,
            'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
            'dtype': 'f32', 'data': 'synthetic',
            'config': {'workload': workload,
                       'envs_per_gpu': n, 'bodies': 4, 'dt': 1e-3, 'solver_iters': int(cfg.solver_iters),
                       'deactivation': ('on: islands at rest are put to sleep after %d substeps (BUILD-CHOSEN; the reference passes no '
                                        'URDF_ENABLE_SLEEPING, bullet_physics.py:173-181).  Pose-level check against no deactivation + 50 '
                                        'plain sweeps on the FP64 oracle: `deactivation.pose_equivalence` (within the FP32 tolerance bounds); '
                                        'the reference-semantics throughput is `reference_semantics`' % int(cfg.sleep_steps))
                                       if int(cfg.sleep_steps) > 0 else 'off',
                       'solver_exit': ('<= %d sweeps; an island stops on residual < %.0e N s (%.0e when at rest and nothing is ever '
                                       'deactivated) or after %d sweeps without a new smallest residual; mean sweeps per island solve: '
                                       '`cpu_baseline.mean_sweeps_per_island_solve` (float oracle, whose iterates are the kernel\'s bit for bit)'
                                       % (int(cfg.solver_iters), float(cfg.solver_tol), float(cfg.solver_tol_rest), int(cfg.solver_stall)))
                                      if float(cfg.solver_tol) > 0 else 'none: %d plain sweeps' % int(cfg.solver_iters),
                       'parallelism': 'env-shards x%d' % world_size,
                       'mode': 'single-launch rollout: K env.step() per env in one rv_rollout_record launch, observation (incl. '
                               '%d-point segmented point cloud per body), reward and done of every step recorded' % int(cfg.num_points)
                               if args.mode == 'rollout' else 'lock-step host loop of batched env.step()'},
            'sim_steps_per_s': substeps_all / elapsed,
            'substeps_per_env_step': substeps_all / max(env_steps_all, 1.0),
            'awake_sim_steps_per_s': st['awake_substeps'] / elapsed,
            'max_substeps_in_launch': st['max_substeps'],
            'awake_substep_fraction': st['awake_substeps'] / max(st['substeps'], 1),
            'reset_substeps': reset_stats['substeps'],
            'roofline': roofline(achieved, awake_only, traffic, issue, avg_kernel_s, algo, per_launch, world_n_waves),
        }
        if n1_same is not None:
            out['n1_same_workload'] = n1_same
            out['scaling_efficiency_vs_n1_same_workload'] = out['value'] / (world_size * n1_same)
            out['config']['n1_same_workload'] = ('rank 0 alone (other ranks idle at a barrier), same %d envs, same %d steps, without the '
                                                 'return gather: the like-for-like N = 1 point of this line; the driver\'s own --gpus 1 run '
                                                 'is BASELINE configs[1] (1024 envs), a different workload' % (n, args.steps))
        out.update(extra)
        if not args.no_cpu_baseline and world_size == 1:
            cl = cpu_legs(cfg_kwargs, scene, names, quick=args.quick)
            cl.update(pose_err_leg(scene, names))
            ref_cpu = cl.pop('reference_semantics_cpu', None)
            pe_ = cl.pop('deactivation_pose_equivalence', None)
            if pe_ is not None:
                out.setdefault('deactivation', {})['pose_equivalence'] = pe_
            out.update(cl)
            if ref_cpu is not None:
                rs = out.setdefault('reference_semantics', {})
                rs['cpu'] = ref_cpu
                g, c_ = rs.get('gpu', {}), ref_cpu
                for k, v in g.items():
                    if v:
                        v['one_wave_us_per_substep'] = 1e6 * v['envs'] / v['sim_steps_per_s']
                rs['gpu_over_cpu_sim_steps'] = {k: g[k]['sim_steps_per_s'] / c_[k]['sim_steps_per_s']
                                                for k in c_ if g.get(k) and c_[k]['sim_steps_per_s'] > 0}
                g8 = rs.get('gpu_8192') or {}
                for k, v in g8.items():
                    if v:
                        v['one_wave_us_per_substep'] = 1e6 * v['envs'] / v['sim_steps_per_s']
                rs['gpu_8192_over_cpu_sim_steps'] = {k: g8[k]['sim_steps_per_s'] / c_[k]['sim_steps_per_s']
                                                     for k in c_ if g8.get(k) and c_[k]['sim_steps_per_s'] > 0}
                rs['cpu_sampling'] = ('every cpu leg: >= 16 envs per host thread, >= 10 s or 4 env.step() per env, OpenMP schedule(dynamic); '
                                      '`one_thread` = the same leg on one thread (16 envs), `scaling_1_to_n` = all threads / one thread')
        # the whole record: a side file (and stderr); stdout carries the compact line only
        legs_path = args.legs_out
        default_legs = os.path.join(ROOT, 'bench_legs.json')
        try:
            if headline_only_run and legs_path == default_legs:      # (a headline-only / profiling run does not overwrite the record of a full one)
                raise OSError('headline-only run: the side file of a full run is left alone (stderr has this record)')
            with open(legs_path, 'w') as f:
                json.dump(out, f, indent=1)
            if os.path.isdir(os.path.join(ROOT, 'gpurun_out')):
                with open(os.path.join(ROOT, 'gpurun_out', 'bench_legs.json'), 'w') as f:
                    json.dump(out, f, indent=1)
        except OSError as ex:
            legs_path = 'not written: %r' % (ex,)
        sys.stderr.write('bench.py full record: ' + json.dumps(out) + '\n')
        sys.stderr.flush()
        os.write(json_fd, (compact_line(out, full_record=os.path.basename(legs_path) if os.path.sep in legs_path else legs_path) + '\n').encode())
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()

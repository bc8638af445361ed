#!/usr/bin/env python
"""bench.py — PushEnv (batched) env steps/sec on N MI355X.

Workload (BASELINE.json configs[1]): PushEnv, 4 rigid convex movables, 1024
vectorised envs per GPU, random policy (Philox U(-1,1)^4 keyed by global env
id and macro-step index), TASK_NAME=None.  One "step" = one batched
`env.step()`: policy -> rv_set_actions -> rv_step_macro (the whole
pre/start/motion/post/offstage phase machine plus settle, thousands of 1 ms
physics substeps per env, on device) -> rv_observe + rv_reward.  Inputs are
resident in HBM; `value` = env steps per second over all ranks.

    python bench.py --gpus 1 --steps 5 --warmup 1
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N \
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

ALGO_BYTES_PER_ENV_SUBSTEP = 3056   # 2*(52*4 + 8*9 + 208*6)
HBM_PEAK_GBS = 8000.0               # MI355X_MICROARCH.md: HBM3E 8 TB/s spec


def cpu_baseline(cfg_kwargs, scene, names, seconds_hint=20.0):
    """Time the CPU oracle (kind 'port': pybullet, the reference's physics, is
    not importable) on a bounded sample of the same workload."""
    from robovat_amd import configs
    from oracle import orc
    cores = os.cpu_count() or 1
    n = max(4, 2 * cores)
    cfg = configs.make_rv_config(n_envs=n, shape_names=names, **cfg_kwargs)
    w = orc.OracleWorld(cfg, scene, double=False)
    w.reset()
    t_all, steps, sub = 0.0, 0, 0
    k = 0
    while t_all < seconds_hint / 4 and k < 4:
        w.set_actions(w.policy_random(k))
        t0 = time.perf_counter(); w.step_macro(); t_all += time.perf_counter() - t0
        st = w.stats(); steps += st['env_steps']; sub += st['substeps']; k += 1
    return {
        'value': steps / t_all, 'unit': 'env_steps/s', 'cores': cores, 'kind': 'port',
        'sim_steps_per_s': sub / t_all,
        'sample': '%d envs x %d macro steps of the same workload, float C oracle, OpenMP over envs '
                  '(pybullet not importable -> reference loop skipped)' % (n, k),
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=50)   # BASELINE configs[1]: 50 macro-steps per env
    ap.add_argument('--warmup', type=int, default=2)
    ap.add_argument('--envs-per-gpu', type=int, default=1024)
    ap.add_argument('--seed', type=int, default=1234)
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-async', action='store_true',
                    help='skip the extra (untimed-for-`value`) asynchronous rollout reported under "async_rollout"')
    ap.add_argument('--mode', choices=['rollout', 'lockstep'], default='rollout',
                    help="rollout: the K timed env.step()s of every env run in ONE rv_rollout launch "
                         "(on-device RandomPolicy, auto-reset); lockstep: K x (policy -> rv_step_macro)")
    args = ap.parse_args()

    import torch
    from robovat_amd import configs, scenes, lib

    world_size = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    dist = None
    if world_size > 1:
        import torch.distributed as dist
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        torch.cuda.set_device(local_rank)
        dist.init_process_group(backend='nccl', device_id=torch.device('cuda', local_rank))
    else:
        torch.cuda.set_device(0)
        local_rank = 0
    assert world_size == args.gpus, 'launch with torch.distributed.run --nproc-per-node %d' % args.gpus

    scene, names = scenes.make_scene()
    n = args.envs_per_gpu
    cfg_kwargs = dict(seed=args.seed)
    cfg = configs.make_rv_config(n_envs=n, env_id_offset=rank * n, shape_names=names, **cfg_kwargs)
    world = lib.World(cfg, scene, device=local_rank)

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    world.reset()
    reset_stats = world.stats()
    returns_all = torch.zeros((world_size, n), dtype=torch.float32, device=world.device)
    counters = torch.zeros(4, dtype=torch.int64, device=world.device)

    def one_step(k):
        a = world.policy_random(k)
        world.set_actions(a)
        world.step_macro()
        obs = world.observe()
        r, d = world.reward()
        if dist is not None:
            # RCCL gather of episode returns + counters
            dist.all_gather_into_tensor(returns_all.view(-1), world.episode_returns())
            st = torch.stack([obs['is_safe'].sum(), obs['is_effective'].sum(), d.sum().to(torch.int64), obs['num_steps'].sum()])
            dist.all_reduce(st)
            counters.copy_(st)
        return r

    def gather_returns():
        if dist is not None:
            # the only collective of the path: RCCL all-gather of episode returns
            # + all-reduce of 4 counters (robovat_amd/parallel.py)
            from robovat_amd import parallel
            cnt = world.env_counters().to(torch.int64)
            st = torch.stack([cnt[:, 5].sum(), cnt[:, 6].sum(), cnt[:, 4].sum(), cnt[:, 1].sum()])
            allr, allc = parallel.gather_returns(world.episode_returns(), st)
            returns_all.copy_(allr); counters.copy_(allc)

    for k in range(args.warmup):
        one_step(k)
    barrier()
    kern_ms, substeps, env_steps, max_sub, awake, launches = 0.0, 0, 0, 0, 0, 0
    t0 = time.perf_counter()
    if args.mode == 'lockstep':
        for k in range(args.steps):
            one_step(args.warmup + k)
            # stats/kernel time are read after the step's kernels are queued; the
            # copies below synchronise the stream, which a Python env loop does anyway
            st = world.stats()
            kern_ms += world.last_kernel_ms(); launches += 1
            substeps += st['substeps']; env_steps += st['env_steps']; max_sub = max(max_sub, st['max_substeps'])
            awake += st['awake_substeps']
    else:
        rewards, dones = world.rollout(args.steps, first_macro_index=args.warmup, auto_reset=True, record=True)
        obs = world.observe()
        gather_returns()
        st = world.stats()
        kern_ms += world.last_kernel_ms(); launches += 1
        substeps += st['substeps']; env_steps += st['env_steps']; max_sub = st['max_substeps']; awake += st['awake_substeps']
    barrier()
    elapsed = time.perf_counter() - t0

    t = torch.tensor([elapsed], dtype=torch.float64, device=world.device)
    tot = torch.tensor([float(substeps), float(env_steps)], dtype=torch.float64, device=world.device)
    if dist is not None:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dist.all_reduce(tot)
    elapsed = float(t.item())
    substeps_all, env_steps_all = float(tot[0].item()), float(tot[1].item())

    # extra leg, NOT part of `value`: the same number of env.step() calls (K per env on
    # average) run asynchronously -- every env steps at its own pace while a shared pool
    # lasts, as the reference's independent worker processes do (tools/parallel_run.py)
    async_out = None
    if args.mode == 'rollout' and not args.no_async:
        barrier()
        ta = time.perf_counter()
        taken = world.rollout_async(args.steps * n, first_macro_index=args.warmup + args.steps)
        barrier()
        ea = time.perf_counter() - ta
        sa = world.stats()
        tt = torch.tensor([ea], dtype=torch.float64, device=world.device)
        ts = torch.tensor([float(sa['env_steps']), float(sa['substeps'])], dtype=torch.float64, device=world.device)
        if dist is not None:
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            dist.all_reduce(ts)
        async_out = {'value': float(ts[0].item()) / float(tt.item()), 'unit': 'env_steps/s',
                     'env_steps': int(ts[0].item()), 'sim_steps_per_s': float(ts[1].item()) / float(tt.item()),
                     'steps_per_env_min_max': [int(taken.min().item()), int(taken.max().item())],
                     'note': 'rv_rollout_async: K*N env.step() calls shared by the N envs of each GPU '
                             '(work-conserving); not the headline because per-env step counts vary'}

    if rank == 0:
        ms_per_step = 1e3 * elapsed / args.steps
        # roofline of the dominant kernel (k_env<MACRO>), this rank
        algo_bytes_per_launch = ALGO_BYTES_PER_ENV_SUBSTEP * (substeps / launches)
        avg_kernel_s = 1e-3 * kern_ms / launches
        achieved = algo_bytes_per_launch / avg_kernel_s / 1e9
        traffic = None
        tpath = os.path.join(ROOT, 'profiles', 'traffic.json')
        if os.path.exists(tpath) and args.mode == 'rollout':
            # HBM bytes per env-substep measured offline with rocprofv3 PMC passes on this
            # same command (profiles/r01_i_hbm_traffic.txt), scaled to this launch
            with open(tpath) as f:
                traffic = json.load(f)['hbm_bytes_per_env_substep'] * (substeps / launches)
        out = {
            'metric': 'env steps/sec (PushEnv, batched)',
            'value': env_steps_all / elapsed,
            'unit': 'env_steps/s',
            'n_gpus': world_size, 'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': ms_per_step,
            'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
            'dtype': 'f32', 'data': 'synthetic',
            'config': {'workload': 'PushEnv 4 rigid convex bodies, %d vectorised envs/GPU, random policy '
                                   '(BASELINE.json configs[1])' % n,
                       'envs_per_gpu': n, 'bodies': 4, 'dt': 1e-3, 'solver_iters': int(cfg.solver_iters),
                       'parallelism': 'env-shards x%d' % world_size, 'mode': args.mode},
            'sim_steps_per_s': substeps_all / elapsed,
            'substeps_per_env_step': substeps_all / max(env_steps_all, 1.0),
            'max_substeps_in_launch': max_sub,
            'awake_substep_fraction': awake / max(substeps, 1),
            'reset_substeps': reset_stats['substeps'],
            'roofline': {'bound': 'hbm', 'achieved': achieved, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                         'frac': achieved / HBM_PEAK_GBS, 'traffic': traffic,
                         'kernel': 'k_env<MODE_ROLLOUT>' if args.mode == 'rollout' else 'k_env<MODE_MACRO>', 'avg_kernel_ms': 1e3 * avg_kernel_s,
                         'algorithmic_bytes_per_env_substep': ALGO_BYTES_PER_ENV_SUBSTEP,
                         'note': 'state is LDS-resident for the whole launch; the kernel is VALU/latency-bound '
                                 '(see DESIGN.md §5), HBM traffic is ~2*sizeof(DevEnv) per env per launch'},
        }
        if async_out is not None:
            out['async_rollout'] = async_out
        if not args.no_cpu_baseline:
            out['cpu_baseline'] = cpu_baseline(cfg_kwargs, scene, names)
        print(json.dumps(out))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    world.close()


if __name__ == '__main__':
    main()

"""N>1 path on CPU: two `gloo` ranks, each stepping its env shard (the CPU
oracle stands in for the GPU world here), then the product's
``robovat_amd.parallel.gather_returns`` collective.  Results must equal one
process owning all envs: sharding never changes what an env does."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from robovat_amd import configs, scenes, parallel

N_PER_RANK = 3
STEPS = 2


def _run_shard(rank, world, offset):
    from oracle import orc
    scene, names = scenes.make_scene()
    cfg = configs.make_rv_config(env_cfg=configs.push_env_config(TASK_NAME='insertion', LAYOUT_ID=0, MAX_STEPS=3),
                                 n_envs=world, env_id_offset=offset, seed=77, shape_names=names)
    w = orc.OracleWorld(cfg, scene)
    w.reset()
    w.rollout(STEPS, 0, True)
    return w


def _worker(rank, world_size, port, q):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world_size)
    w = _run_shard(rank, N_PER_RANK, parallel.env_id_offset(rank, N_PER_RANK))
    returns = torch.tensor(w.episode_returns(), dtype=torch.float32)
    cnt = w.env_counters()
    counters = torch.tensor([cnt[:, 2].sum(), cnt[:, 5].sum(), cnt[:, 6].sum(), cnt[:, 1].sum()], dtype=torch.int64)
    allr, allc = parallel.gather_returns(returns, counters)
    if rank == 0:
        q.put((allr.numpy(), allc.numpy(), w.body_state()))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_gather_matches_single_process():
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0)); port = s.getsockname()[1]
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    allr, allc, state0 = q.get(timeout=300)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    single = _run_shard(0, 2 * N_PER_RANK, 0)
    assert allr.shape == (2, N_PER_RANK)
    assert np.array_equal(allr.reshape(-1), single.episode_returns().astype(np.float32))
    cnt = single.env_counters()
    assert list(allc) == [cnt[:, 2].sum(), cnt[:, 5].sum(), cnt[:, 6].sum(), cnt[:, 1].sum()]
    assert np.array_equal(state0, single.body_state()[:N_PER_RANK])


def test_eight_rank_shards_match_single_process():
    """BASELINE config 5's layout in small: 8 ranks x 2 envs, keyed by global env id -- the gathered returns and
    summed counters are those of one process that owns all 16 envs (no 8-GPU box is needed for that statement)."""
    n_ranks, per = 8, 2
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0)); port = s.getsockname()[1]
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker8, args=(r, n_ranks, per, port, q)) for r in range(n_ranks)]
    for p in procs:
        p.start()
    allr, allc = q.get(timeout=600)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    single = _run_shard(0, n_ranks * per, 0)
    assert allr.shape == (n_ranks, per)
    assert np.array_equal(allr.reshape(-1), single.episode_returns().astype(np.float32))
    cnt = single.env_counters()
    assert list(allc) == [cnt[:, 2].sum(), cnt[:, 5].sum(), cnt[:, 6].sum(), cnt[:, 1].sum()]


def _worker8(rank, world_size, per, port, q):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    os.environ['OMP_NUM_THREADS'] = '2'
    dist.init_process_group('gloo', rank=rank, world_size=world_size)
    w = _run_shard(rank, per, parallel.env_id_offset(rank, per))
    returns = torch.tensor(w.episode_returns(), dtype=torch.float32)
    cnt = w.env_counters()
    counters = torch.tensor([cnt[:, 2].sum(), cnt[:, 5].sum(), cnt[:, 6].sum(), cnt[:, 1].sum()], dtype=torch.int64)
    allr, allc = parallel.gather_returns(returns, counters)
    if rank == 0:
        q.put((allr.numpy(), allc.numpy()))
    dist.barrier()
    dist.destroy_process_group()


def test_gather_without_process_group_is_identity():
    r = torch.arange(4, dtype=torch.float32)
    out, c = parallel.gather_returns(r, torch.ones(4, dtype=torch.int64))
    assert out.shape == (1, 4) and torch.equal(out[0], r)
    assert parallel.env_id_offset(3, 8192) == 24576

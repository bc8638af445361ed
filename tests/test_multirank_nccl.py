"""The N > 1 path on real GPUs: two ranks, one lib.World shard each, `nccl` (= RCCL) for the only
collective of the path (robovat_amd.parallel.gather_returns).  Needs two HIP devices: skipped on
the one-GPU boxes this repo is developed on; the driver's 8-GPU node runs it (and bench.py --gpus N).
What every rank computes must equal one process owning all the envs."""
import os
import socket

import numpy as np
import pytest
import torch

from robovat_amd import configs, scenes, parallel

N_PER_RANK = 64
STEPS = 2
pytestmark = pytest.mark.gpu


def _make(n, offset, device):
    from robovat_amd import lib
    scene, names = scenes.make_scene()
    cfg = configs.make_rv_config(env_cfg=configs.push_env_config(TASK_NAME='insertion', LAYOUT_ID=0, MAX_STEPS=3),
                                 n_envs=n, env_id_offset=offset, seed=77, shape_names=names)
    w = lib.World(cfg, scene, device=device)
    w.reset()
    w.rollout(STEPS, 0, True)
    return w


def _worker(rank, world_size, port, q):
    import torch.distributed as dist
    os.environ['MASTER_ADDR'] = '127.0.0.1'; os.environ['MASTER_PORT'] = str(port)
    os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    torch.cuda.set_device(rank)
    dist.init_process_group('nccl', rank=rank, world_size=world_size, device_id=torch.device('cuda', rank))
    w = _make(N_PER_RANK, parallel.env_id_offset(rank, N_PER_RANK), rank)
    cnt = w.env_counters().to(torch.int64)
    counters = torch.stack([cnt[:, 2].sum(), cnt[:, 5].sum(), cnt[:, 6].sum(), cnt[:, 1].sum()])
    allr, allc = parallel.gather_returns(w.episode_returns(), counters)
    if rank == 0:
        q.put((allr.cpu().numpy(), allc.cpu().numpy(), w.body_state().cpu().numpy()))
    dist.barrier()
    dist.destroy_process_group()
    w.close()


@pytest.mark.skipif(not torch.cuda.is_available() or torch.cuda.device_count() < 2, reason='needs two HIP devices (RCCL)')
def test_two_gpu_shards_and_rccl_gather_match_single_process():
    import torch.multiprocessing as mp
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0)); port = s.getsockname()[1]
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    allr, allc, state0 = q.get(timeout=600)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    single = _make(2 * N_PER_RANK, 0, 0)
    assert allr.shape == (2, N_PER_RANK)
    assert np.array_equal(allr.reshape(-1), single.episode_returns().cpu().numpy())
    cnt = single.env_counters().cpu().numpy()
    assert list(allc) == [cnt[:, 2].sum(), cnt[:, 5].sum(), cnt[:, 6].sum(), cnt[:, 1].sum()]
    assert np.array_equal(state0, single.body_state().cpu().numpy()[:N_PER_RANK])
    single.close()


def _one_rank_group():
    """A one-rank RCCL group on a free local port.  The port is found by binding to 0 and closing: somebody else may take it
    before the store binds it (seen once: EADDRINUSE), so the rendezvous is retried on a new port."""
    import torch.distributed as dist
    last = None
    for _ in range(8):
        with socket.socket() as s:
            s.bind(('127.0.0.1', 0)); port = s.getsockname()[1]
        try:
            dist.init_process_group('nccl', init_method='tcp://127.0.0.1:%d' % port, rank=0, world_size=1, device_id=torch.device('cuda', 0))
            return
        except Exception as ex:      # noqa: BLE001 -- DistNetworkError (address in use): try the next port
            last = ex
    raise last


def test_one_rank_nccl_group_gathers():
    """RCCL on the one GPU that is here: a one-rank process group through the same call."""
    import torch.distributed as dist
    _one_rank_group()
    try:
        r = torch.arange(8, dtype=torch.float32, device='cuda'); c = torch.tensor([1, 2, 3, 4], dtype=torch.int64, device='cuda')
        out, cc = parallel.gather_returns(r, c)
        assert out.shape == (1, 8) and torch.equal(out[0], r) and torch.equal(cc, c)
    finally:
        dist.destroy_process_group()


def test_two_worlds_on_two_streams_in_one_process_are_reentrant():
    """SURVEY.md 8b "re-entrancy across worlds" (one process driving several GPUs holds several rv_world):
    two worlds -- the two env shards of a 2-rank run -- live in ONE process on ONE device, each on its own
    HIP stream, and are stepped concurrently (launches interleaved, no synchronisation in between).  Each
    must compute what it computes alone, and together what a single world owning all envs computes."""
    from robovat_amd import lib
    scene, names = scenes.make_scene()

    def cfg(n, offset):
        return configs.make_rv_config(env_cfg=configs.push_env_config(TASK_NAME='insertion', LAYOUT_ID=0, MAX_STEPS=3),
                                      n_envs=n, env_id_offset=offset, seed=77, shape_names=names)
    n = 96
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    worlds = []
    for r in range(2):
        with torch.cuda.stream(streams[r]):
            w = lib.World(cfg(n, r * n), scene, device=0)      # (binds the world to the stream that is current now)
            worlds.append(w)
    # interleaved launches on the two streams; nothing waits until the end
    for r in range(2):
        with torch.cuda.stream(streams[r]):
            worlds[r].reset()
    for k in range(STEPS):
        for r in (1, 0):
            with torch.cuda.stream(streams[r]):
                worlds[r].rollout(1, k, True)
    for s in streams:
        s.synchronize()
    both = lib.World(cfg(2 * n, 0), scene, device=0)
    both.reset(); both.rollout(STEPS, 0, True)
    torch.cuda.synchronize()
    want_state, want_ret = both.body_state().cpu().numpy(), both.episode_returns().cpu().numpy()
    for r in range(2):
        with torch.cuda.stream(streams[r]):
            got_state, got_ret = worlds[r].body_state().cpu().numpy(), worlds[r].episode_returns().cpu().numpy()
        assert np.array_equal(got_state, want_state[r * n:(r + 1) * n])
        assert np.array_equal(got_ret, want_ret[r * n:(r + 1) * n])
    # ... and the path's collective on this process's own one-rank RCCL group
    import torch.distributed as dist
    if not dist.is_initialized():
        _one_rank_group()
        made = True
    else:
        made = False
    allr, _ = parallel.gather_returns(worlds[0].episode_returns())
    assert allr.shape == (1, n) and np.array_equal(allr[0].cpu().numpy(), want_ret[:n])
    if made:
        dist.destroy_process_group()
    for w in worlds + [both]:
        w.close()

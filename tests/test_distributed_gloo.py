"""world_size=2 test of the multi-GPU path's only collective on CPU with gloo: each rank holds
the statistics of its batch shard; after all_gather_stats every rank has the estimators of the global batch."""
import math
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _shard_stats(v):
    neg = -v
    m = neg.max()
    return torch.stack([torch.tensor(float(len(v)), dtype=torch.float32), neg.sum(), ((v - v.mean()) ** 2).sum(), m,
                        (neg - m).exp().sum(), (2 * (neg - m)).exp().sum(), torch.tensor(0.0), torch.tensor(0.0)]).float()


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from sde_sampler_amd import engine as E

    torch.manual_seed(123)
    rnd = torch.randn(4096) * 2 + 30 + torch.arange(4096) * 1e-3  # the "global" rnd, identical on both ranks
    n = 4096 // world
    local = rnd[rank * n:(rank + 1) * n]
    merged = E.all_gather_stats(_shard_stats(local))
    q.put((rank, E.estimators_from_stats(merged)))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_estimator_merge():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = dict(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    torch.manual_seed(123)
    rnd = (torch.randn(4096) * 2 + 30 + torch.arange(4096) * 1e-3).double()
    neg = -rnd
    m = neg.max()
    for rank in range(world):
        est = results[rank]
        assert est["n"] == 4096
        assert est["mean_neg_rnd"] == pytest.approx(neg.mean().item(), abs=1e-4)
        assert est["var_rnd"] == pytest.approx(rnd.var().item(), rel=1e-4)
        assert est["log_norm_const_is"] == pytest.approx(((neg - m).exp().mean().log() + m).item(), abs=1e-4)
    assert results[0] == results[1]

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


def _dp_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from sde_sampler_amd.losses.oc import BaseOCLoss
    from sde_sampler_amd.utils.distributed import all_reduce_gradients

    torch.manual_seed(7)
    theta = torch.nn.Parameter(torch.tensor([0.3, -1.2, 0.7]))  # replicated "network"
    feats = torch.randn(4096, 3)                                # global batch
    n = 4096 // world
    out = {}
    for method in ("lv", "kl", "lv_traj"):
        loss_obj = BaseOCLoss(generative_ctrl=None, method=method, max_rnd=1e3, traj_per_sample=2 if method == "lv_traj" else 1)
        f = feats[rank * n:(rank + 1) * n]
        if method == "lv_traj":  # [traj_per_sample * samples]: two trajectories per sample, both on this rank
            f = torch.cat([f[: n // 2], f[: n // 2] * 1.1 + 0.05])
        rnd = (f @ theta).reshape(-1, 1) + 30.0 + (rank == 1) * (torch.arange(f.shape[0]).reshape(-1, 1) == 5) * 1e6  # one filtered row
        theta.grad = None
        loss_obj.report_global_loss = True  # (an extra all-reduce: off by default)
        share, metrics = loss_obj.compute_loss(rnd)
        share.backward()
        all_reduce_gradients([theta])
        out[method] = (share.item(), metrics["train/loss_global"], theta.grad.tolist(), loss_obj.n_filtered)  # (plain lists: a tensor in the queue is an fd the parent must fetch while this process lives)
        # the capture-safe form of the same loss (graph_safe: masked reductions, ONE device-side all-reduce, no .item()): same share,
        # same gradient, the filtered count kept in a tensor
        safe = BaseOCLoss(generative_ctrl=None, method=method, max_rnd=1e3, traj_per_sample=2 if method == "lv_traj" else 1)
        safe.graph_safe = True
        calls = []
        real = dist.all_reduce
        dist.all_reduce = lambda t, *a, **k: (calls.append(tuple(t.shape)), real(t, *a, **k))[1]
        theta.grad = None
        rnd2 = (f @ theta).reshape(-1, 1) + 30.0 + (rank == 1) * (torch.arange(f.shape[0]).reshape(-1, 1) == 5) * 1e6
        share2, metrics2 = safe.compute_loss(rnd2)
        share2.backward()
        dist.all_reduce = real
        assert len(calls) == 1 and "train/loss_global" not in metrics2, calls  # one collective on the loss path
        all_reduce_gradients([theta])
        out[method + "/graph_safe"] = (share2.item(), None, theta.grad.tolist(), int(metrics2["train/n_filtered_cumulative"]))
    q.put((rank, out))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_training_loss_is_the_global_batch_loss():
    """compute_loss under torch.distributed: the ranks' shares sum to the loss of the global batch and the all-reduced gradient is
    its gradient (single-process computation on the concatenated batch)."""
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_dp_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = dict(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    torch.manual_seed(7)
    theta = torch.nn.Parameter(torch.tensor([0.3, -1.2, 0.7]))
    feats = torch.randn(4096, 3)
    n = 2048
    for method in ("lv", "kl", "lv_traj"):
        parts = []
        for rank in range(world):
            f = feats[rank * n:(rank + 1) * n]
            if method == "lv_traj":
                f = torch.cat([f[: n // 2], f[: n // 2] * 1.1 + 0.05])
            r = (f @ theta).reshape(-1, 1) + 30.0 + (rank == 1) * (torch.arange(f.shape[0]).reshape(-1, 1) == 5) * 1e6
            parts.append(r)
        theta.grad = None
        if method == "lv_traj":
            per = []
            for r in parts:
                r2 = r.reshape(2, -1, 1)
                keep = (r2 < 1e3).all(dim=0)
                per.append(r2[:, keep].var(dim=0))
            ref = torch.cat(per).mean()
        else:
            allr = torch.cat(parts)
            kept = allr[allr < 1e3]
            ref = kept.var() if method == "lv" else kept.mean()
        ref.backward()
        shares = [results[r][method][0] for r in range(world)]
        assert sum(shares) == pytest.approx(ref.item(), rel=1e-5)
        for r in range(world):
            assert results[r][method][1] == pytest.approx(ref.item(), rel=1e-5)
            assert torch.allclose(torch.tensor(results[r][method][2]), theta.grad, rtol=1e-4, atol=1e-6), method
            assert results[r][method][3] == (2 if method == "lv_traj" else 1)
        # the capture-safe loss: the same shares (to fp32 rounding of another summation order), gradient and filtered count
        safe = [results[r][method + "/graph_safe"] for r in range(world)]
        assert sum(v[0] for v in safe) == pytest.approx(ref.item(), rel=1e-5)
        for r in range(world):
            assert safe[r][0] == pytest.approx(shares[r], rel=1e-4, abs=1e-6), (method, r)
            assert torch.allclose(torch.tensor(safe[r][2]), theta.grad, rtol=1e-4, atol=1e-6), method
            assert safe[r][3] == (2 if method == "lv_traj" else 1)


def _mask_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from sde_sampler_amd.utils.distributed import all_reduce_gradients

    a, b = torch.nn.Parameter(torch.ones(3)), torch.nn.Parameter(torch.ones(2))
    out = {}
    # (1) agreement: b has no gradient on either rank -- it stays None, a is summed
    a.grad, b.grad = torch.full((3,), float(rank + 1)), None
    all_reduce_gradients([a, b])
    out["agree"] = (a.grad.tolist(), b.grad is None)
    # (2) the patterns diverge LATER in the run: rank 1 suddenly has a gradient for b.  The collectives still match (fixed bucket);
    # the rank that sees a new pattern raises, the other one -- which repeats a pattern it has checked -- gets NaN gradients
    a.grad = torch.full((3,), 1.0)
    b.grad = torch.ones(2) if rank == 1 else None
    try:
        all_reduce_gradients([a, b])
        out["diverged"] = ("no error", [float(v) for v in a.grad])
    except RuntimeError as exc:
        out["diverged"] = ("RuntimeError", str(exc))
    q.put((rank, out))
    dist.barrier()
    dist.destroy_process_group()


def _dtype_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from sde_sampler_amd.utils.distributed import all_reduce_gradients

    a = torch.nn.Parameter(torch.ones(3, dtype=torch.float64))
    b = torch.nn.Parameter(torch.ones(2, dtype=torch.float32))
    c = torch.nn.Parameter(torch.ones(4, dtype=torch.float32))  # never has a gradient
    tiny = 1.0 + 2.0 ** -40  # not representable in fp32
    a.grad, b.grad, c.grad = torch.full((3,), tiny * (rank + 1), dtype=torch.float64), torch.full((2,), 0.5 * (rank + 1)), None
    all_reduce_gradients([a, b, c])
    q.put((rank, (str(a.grad.dtype), a.grad.tolist(), str(b.grad.dtype), b.grad.tolist(), c.grad is None)))
    dist.barrier()
    dist.destroy_process_group()


def test_gradient_all_reduce_keeps_the_parameters_dtype():
    """the bucket travels in the widest dtype among the parameters -- an fp64 gradient is reduced in fp64 (the bucket used to be
    cast to fp32), fp32 gradients come back as fp32, a parameter without a gradient keeps `grad = None`."""
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_dtype_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
    tiny = 1.0 + 2.0 ** -40
    for r in range(world):
        dt_a, ga, dt_b, gb, c_none = got[r]
        assert dt_a == "torch.float64" and ga == [3.0 * tiny] * 3 and 3.0 * tiny != 3.0
        assert dt_b == "torch.float32" and gb == [1.5, 1.5] and c_none


def test_gradient_all_reduce_with_diverging_gradient_patterns_never_mismatches_collectives():
    """every call joins ONE collective of a fixed shape; a disagreement about which parameters have gradients is an error
    on the rank that sees the new pattern and NaN gradients (for the trainer's finite-gradient guard) on the others -- never a hang."""
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_mask_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = dict(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank in range(world):
        assert results[rank]["agree"] == ([3.0, 3.0, 3.0], True)
    assert results[1]["diverged"][0] == "RuntimeError" and "disagree" in results[1]["diverged"][1]
    kind, grads = results[0]["diverged"]
    assert kind == "no error" and all(v != v for v in grads), results[0]["diverged"]  # NaN: the step will be skipped

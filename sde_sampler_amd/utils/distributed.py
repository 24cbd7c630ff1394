"""Data-parallel training helpers: trajectories are independent given the weights, so ranks hold batch shards
and replicated parameters; per step there is the loss's tiny all-reduce (global mean of rnd, losses/oc.py compute_loss) and ONE
gradient all-reduce over a single flat bucket (the control networks have 4e4 .. 2e5 parameters = 0.15 .. 0.8 MB of fp32:
latency-bound on xGMI, so one collective, not one per tensor)."""
from __future__ import annotations

from typing import Iterable

import torch


_checked_masks: set = set()


def all_reduce_gradients(parameters: Iterable[torch.nn.Parameter], group=None) -> None:
    """SUM all-reduce of the `.grad`s in one flat bucket (in place).  With the loss classes' data-parallel loss shares the
    result is the gradient of the global-batch loss on every rank.  No-op without an initialised process group.

    Only parameters that HAVE a gradient enter the bucket and are written back: a parameter the loss does not reach keeps
    `grad = None` (so Adam / weight decay keep skipping it, as without a group).  Which parameters have one must agree across
    the ranks -- checked with one tiny all-gather the first time a pattern is seen (outside any stream capture: the first,
    eager, steps of a run), never again for that pattern.  Under the `gloo` backend GPU gradients travel through the host
    (not capturable); `nccl` (= RCCL) reduces the device bucket in place."""
    import torch.distributed as dist

    if not (dist.is_available() and dist.is_initialized()):
        return  # (with a process group the collective runs at every world size, also 1: same code path as an 8-rank job)
    params = [p for p in parameters if p.requires_grad]
    mask = tuple(p.grad is not None for p in params)
    key = (id(group), mask)
    if key not in _checked_masks:
        world = dist.get_world_size(group)
        if world > 1:
            mine = [mask]
            theirs = [None] * world
            dist.all_gather_object(theirs, mine[0], group=group)
            if any(tuple(t) != mask for t in theirs):
                raise RuntimeError("all_reduce_gradients: the ranks disagree on which parameters have gradients")
        _checked_masks.add(key)
    params = [p for p in params if p.grad is not None]
    if not params:
        return
    flat = torch.cat([p.grad.reshape(-1) for p in params])
    if dist.get_backend(group) == "gloo" and flat.is_cuda:
        host = flat.cpu()
        dist.all_reduce(host, op=dist.ReduceOp.SUM, group=group)
        flat = host.to(flat.device)
    else:
        dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
    offset = 0
    for p in params:
        n = p.numel()
        p.grad.copy_(flat[offset:offset + n].view_as(p))
        offset += n

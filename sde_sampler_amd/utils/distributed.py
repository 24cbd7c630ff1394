"""Data-parallel training helpers: trajectories are independent given the weights, so ranks hold batch shards
and replicated parameters; per step there is the loss's tiny all-reduce (global mean of rnd, losses/oc.py compute_loss) and ONE
gradient all-reduce over a single flat bucket (the control networks have 4e4 .. 2e5 parameters = 0.15 .. 0.8 MB of fp32:
latency-bound on xGMI, so one collective, not one per tensor)."""
from __future__ import annotations

from typing import Iterable

import torch


_checked_masks: set = set()
_mask_tensors: dict = {}


def all_reduce_gradients(parameters: Iterable[torch.nn.Parameter], group=None) -> None:
    """SUM all-reduce of the `.grad`s in one flat bucket (in place).  With the loss classes' data-parallel loss shares the
    result is the gradient of the global-batch loss on every rank.  No-op without an initialised process group.

    The bucket has a FIXED shape -- every parameter that requires a gradient, zeros where this rank has none, followed by one
    flag per parameter ("this rank has a gradient") -- so the ranks' collectives always match, whatever happens to the autograd
    graph of one of them later in a run (a check that only runs the first time a pattern is seen lets the ranks enter
    different collectives when their patterns diverge afterwards).  Only parameters that HAVE a gradient here are written back: a
    parameter the loss does not reach keeps `grad = None` (Adam / weight decay keep skipping it, as without a group).  Which
    parameters have one must agree across the ranks.  That is checked twice, symmetrically -- the REDUCED flags are the same
    numbers on every rank: (i) on the device, in every call: a parameter whose flag sum is neither 0 nor the world size turns
    the whole reduced bucket into NaN on every rank (stream-ordered, capture-safe, no host synchronisation: the trainer's
    finite-gradient guard skips the step and counts it); (ii) on the host, the first time a rank sees a pattern outside a stream
    capture: a RuntimeError.  Under the `gloo` backend GPU gradients travel through the host (not capturable); `nccl` (= RCCL)
    reduces the device bucket in place."""
    import torch.distributed as dist

    if not (dist.is_available() and dist.is_initialized()):
        return  # (with a process group the collective runs at every world size, also 1: same code path as an 8-rank job)
    params = [p for p in parameters if p.requires_grad]
    if not params:
        return  # (the model is replicated: no rank has anything to reduce)
    mask = tuple(p.grad is not None for p in params)
    dev = params[0].device
    flags = _mask_tensors.get((dev, mask))
    if flags is None:
        flags = _mask_tensors[(dev, mask)] = torch.tensor([1.0 if m else 0.0 for m in mask], device=dev, dtype=torch.float32)
    n_flag = len(params)
    pieces = [p.grad.reshape(-1).float() if p.grad is not None else torch.zeros(p.numel(), device=dev, dtype=torch.float32)
              for p in params]
    flat = torch.cat(pieces + [flags])
    if dist.get_backend(group) == "gloo" and flat.is_cuda:
        host = flat.cpu()
        dist.all_reduce(host, op=dist.ReduceOp.SUM, group=group)
        flat = host.to(flat.device)
    else:
        dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
    world = dist.get_world_size(group)
    seen = flat[-n_flag:]
    bad = ((seen != 0) & (seen != world)).any()
    grads = torch.where(bad, torch.full_like(flat[:1], float("nan")), flat[:-n_flag])
    key = (id(group), mask)
    if key not in _checked_masks:
        capturing = flat.is_cuda and torch.cuda.is_current_stream_capturing()
        if not capturing:
            if bool(bad):
                raise RuntimeError("all_reduce_gradients: the ranks disagree on which parameters have gradients")
            _checked_masks.add(key)
    offset = 0
    for p in params:
        n = p.numel()
        if p.grad is not None:
            p.grad.copy_(grads[offset:offset + n].view_as(p))
        offset += n

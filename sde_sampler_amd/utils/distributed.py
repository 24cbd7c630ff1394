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
    # the bucket travels in the widest dtype among the parameters (fp32 networks: fp32; an fp64 parameter keeps its precision --
    # the bucket used to be cast to fp32), the flags with it
    dtype = params[0].dtype
    for p in params[1:]:
        dtype = torch.promote_types(dtype, p.dtype)
    flags = _mask_tensors.get((dev, mask, dtype))
    if flags is None:
        flags = _mask_tensors[(dev, mask, dtype)] = torch.tensor([1.0 if m else 0.0 for m in mask], device=dev, dtype=dtype)
    n_flag = len(params)
    sizes = [p.numel() for p in params]
    # ONE zero-filled bucket; the gradients that exist are copied into their slices (no per-parameter zero tensors)
    flat = torch.zeros(sum(sizes) + n_flag, device=dev, dtype=dtype)
    slots = flat[:-n_flag].split(sizes)
    have = [(slot, p.grad.reshape(-1)) for slot, p in zip(slots, params) if p.grad is not None]
    if have:
        torch._foreach_copy_([a for a, _ in have], [b for _, b in have])
    flat[-n_flag:].copy_(flags)
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
    back = [(p.grad, g.view_as(p)) for p, g in zip(params, grads.split(sizes)) if p.grad is not None]
    if back:
        torch._foreach_copy_([a for a, _ in back], [b for _, b in back])  # (casts back to each gradient's own dtype)

"""hipGraph capture of a whole training step (sample -> fused trajectory forward -> HIP backward -> weight-gradient GEMMs ->
optimizer step) for the launch-bound regime.

At the reference's training sizes (batch 512 ... 4096, 100 ... 200 steps) one optimisation step of this package is about forty
short kernels: the trajectory kernel itself is latency-bound (5 us per Euler-Maruyama step) and the host spends longer
issuing the rest than the GPU spends running it.  `GraphedTrainStep` records the step once on a side stream
(`torch.cuda.CUDAGraph`, i.e. hipStreamBeginCapture / hipGraphLaunch underneath; the ctypes launches of libsdeh.so go to
torch's current stream and are captured like torch's own kernels) and replays it with one launch.

Three things make the step replayable:

* fresh noise per replay -- launch arguments (seed, offset) are frozen at capture time, so the kernels add a device-resident
  counter to the Philox offset (`SdehProblem.rng_offset_dev`, include/sdeh.h) and the graph bumps it after the optimizer step;
  the backward launch replays the forward's draws because it reads the same counter value;
* no host round trip -- `loss.graph_safe = True` switches `compute_loss` to masked reductions (losses/oc.py);
* no stale autograd state -- the captured step differentiates w.r.t. fresh aliases of the parameters, so AccumulateGrad nodes
  that earlier eager steps left alive (bound to the default stream) never enter the captured graph.

The reference trainer skips the optimizer step when the loss or a gradient is not finite (`if loss_ok and grad_ok`,
solver/base.py:409-432) -- a host decision.  Here the same decision is taken on the device (`guard=True`): parameters and
optimizer state are snapshotted before `optimizer.step()`, the step runs on sanitised gradients, and everything is put back
when the step was not acceptable (exactly: new = ok * new + (1 - ok) * old with ok in {0, 1}); `n_skipped` counts those steps.
Without it a single bad batch poisons Adam's moments for good.  Data parallelism: pass the gradient all-reduce as
`reduce_gradients=` -- it runs BEFORE the check, so every rank decides on the same reduced gradients.

Platform caveat (torch 2.10 + ROCm 7.x): two consecutive multi-block framework reductions captured into one hipGraph return a
corrupted second result from the second replay on (tests/perf/rocm_graph_two_reductions.py reproduces it with PyTorch alone).
Everything this package puts into the step avoids them -- the weight-gradient partials and the loss statistics are reduced by
libsdeh's own kernels -- and tests/test_hip_graphs.py compares replayed gradients with eager ones.  A `loss_fn` / `after_backward`
of your own should keep its reductions small (one block: up to a few thousand elements) or check itself the same way.
"""
from __future__ import annotations

from typing import Callable, Iterable

import torch

from sde_sampler_amd import _lib as L

__all__ = ["GraphedTrainStep", "GraphedEval"]

#: first value of the device counter: far above anything `engine.calls` reaches, so that replays and eager launches of the
#: same seed never share a Philox offset
COUNTER_START = 1 << 40


class GraphedTrainStep:
    """Captures `optimizer.zero_grad(); l = loss_fn(); l.backward(); [after_backward()]; optimizer.step()` into one graph.

    loss_fn          () -> scalar loss tensor; may draw samples with torch's device generator, must not synchronise and must
                     only read tensors that stay alive (parameters, `ts`, buffers of the prior / target)
    losses           the loss objects `loss_fn` calls (their `rng_counter` / `graph_safe` are set here)
    optimizer        for Adam/AdamW pass `capturable=True`
    reduce_gradients optional () -> None right after backward, BEFORE the finite-gradient guard (data parallelism:
                     `lambda: all_reduce_gradients(params)`): the accept / reject decision is then taken on the REDUCED gradients
                     and is the same on every rank -- a NaN from another rank, or the disagreement poison of
                     `all_reduce_gradients`, rejects the step everywhere; also captured
    after_backward   optional () -> None between the guard's check and the optimizer step (gradient clipping -- the reference clips
                     only steps it accepted, solver/base.py:421-427), also captured
    warmup           eager steps on the side stream before capturing (allocator warm-up; they DO update the parameters)
    guard            skip the update on the device when the loss or a gradient is not finite (solver/base.py:409-432)
    max_loss         with guard: additionally require |loss| <= max_loss (the reference's `max_loss`)
    """

    def __init__(self, loss_fn: Callable[[], torch.Tensor], losses: Iterable, optimizer: torch.optim.Optimizer, *,
                 after_backward: Callable[[], None] | None = None, reduce_gradients: Callable[[], None] | None = None,
                 warmup: int = 3, device=None, guard: bool = True, max_loss: float | None = None):
        if not torch.cuda.is_available():
            raise RuntimeError("GraphedTrainStep needs a GPU (hipGraph capture)")
        for group in optimizer.param_groups:
            if "capturable" in group and not group["capturable"]:
                raise ValueError(f"{type(optimizer).__name__} must be created with capturable=True to be captured in a graph")
        self.device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        self.losses = list(losses)
        self.optimizer = optimizer
        self._loss_fn, self._after_backward, self._reduce_gradients = loss_fn, after_backward, reduce_gradients
        self.counter = torch.full((1,), COUNTER_START, dtype=torch.int64, device=self.device)
        for lo in self.losses:
            lo.rng_counter = self.counter
            lo.graph_safe = True
        self.replays = 0
        self.guard, self.max_loss = guard, max_loss
        self.n_skipped = torch.zeros((), dtype=torch.int64, device=self.device)
        self._params = [p for g in optimizer.param_groups for p in g["params"] if p.requires_grad]
        self._snap: list[torch.Tensor] | None = None
        self._table, self._table_key = None, None

        side = torch.cuda.Stream(device=self.device)
        side.wait_stream(torch.cuda.current_stream(self.device))
        with torch.cuda.stream(side):
            for _ in range(max(int(warmup), 1)):
                self._step()
            # guard: the (tensor, snapshot) pointer table of `_restore` must exist before the capture -- it is built by a host-to-device
            # copy, which a capture cannot contain.  A lazily initialised optimizer state (Adam) appears in the first step and the table
            # follows one step later: keep stepping eagerly until a step has left it alone (warmup <= 1 used to rebuild it
            # inside the capture).  These are real optimisation steps, like the warm-up's.
            extra = 0
            while self.guard and not self._table_ready():
                if extra == 4:
                    raise RuntimeError("GraphedTrainStep(guard=True): the optimizer state keeps changing its tensors from step to step; "
                                       "it cannot be captured")
                self._step()
                extra += 1
        self.extra_warmup = extra if self.guard else 0
        torch.cuda.current_stream(self.device).wait_stream(side)
        torch.cuda.synchronize(self.device)

        self.graph = torch.cuda.CUDAGraph()
        optimizer.zero_grad(set_to_none=True)
        with torch.cuda.graph(self.graph):
            self.loss = self._step()  # static output: overwritten by every replay

    def _restore(self, tensors: list[torch.Tensor], ok: torch.Tensor) -> None:
        key = tuple(t.data_ptr() for t in tensors) + tuple(s.data_ptr() for s in self._snap)
        if self._table_key != key:  # (built while the optimizer state appears; fixed from the last warm-up step on, i.e. in the capture)
            if torch.cuda.is_current_stream_capturing():
                raise RuntimeError("GraphedTrainStep(guard=True): parameters or optimizer state moved between the last eager step and the "
                                   "capture (the pointer table of the update guard cannot be built inside a capture)")
            rows = []
            for t, s in zip(tensors, self._snap):
                if not (t.is_contiguous() and s.is_contiguous() and (t.numel() * t.element_size()) % 4 == 0):
                    raise RuntimeError("GraphedTrainStep(guard=True): parameters and optimizer state must be contiguous 4- / 8-byte tensors")
                rows.append([t.data_ptr(), s.data_ptr(), t.numel() * t.element_size() // 4])
            self._table = torch.tensor(rows, dtype=torch.int64, device=self.device)
            self._table_key = key
        with torch.cuda.device(self.device):
            L.check(L.load().sdeh_guard_restore(self._table.data_ptr(), len(tensors), ok.data_ptr(), self.n_skipped.data_ptr(),
                                                torch.cuda.current_stream(self.device).cuda_stream))

    def _guarded(self) -> list[torch.Tensor]:
        """Parameters and every tensor of the optimizer state (moments, step counters): what `optimizer.step()` may change."""
        out = list(self._params)
        for p in self._params:
            out += [v for v in self.optimizer.state.get(p, {}).values() if isinstance(v, torch.Tensor)]
        return out

    def _table_ready(self) -> bool:
        """The pointer table `_restore` will ask for in the NEXT step exists already (same tensors, same snapshots)."""
        tensors = self._guarded()
        if self._snap is None or len(self._snap) != len(tensors) or self._table is None:
            return False
        return self._table_key == tuple(t.data_ptr() for t in tensors) + tuple(s.data_ptr() for s in self._snap)

    def _step(self) -> torch.Tensor:
        self.optimizer.zero_grad(set_to_none=True)
        for lo in self.losses:
            lo._graph_leaves = []  # the losses attach their autograd Functions to fresh aliases of the parameters (losses/_autograd.py)
        value = self._loss_fn()
        # Gradients by torch.autograd.grad w.r.t. those aliases, never through the real parameters' AccumulateGrad nodes: a node that
        # an earlier EAGER step left alive (e.g. through a `loss` tensor the caller still holds) is bound to the default stream, and
        # its mere presence in the captured autograd graph aborts the process at the end of the capture on this stack.
        owners, leaves = [], []
        for lo in self.losses:
            for params, aliases in lo._graph_leaves:
                owners += params
                leaves += aliases
            lo._graph_leaves = None
        covered = {id(p) for p in owners}
        rest = [p for p in self._params if id(p) not in covered]  # parameters the loss_fn uses outside the library's losses
        grads = torch.autograd.grad(value, leaves + rest, allow_unused=True)
        for p, g in zip(owners + rest, grads):
            if g is not None:
                p.grad = g if p.grad is None else p.grad + g
        if self._reduce_gradients is not None:
            self._reduce_gradients()
        if not self.guard:
            if self._after_backward is not None:
                self._after_backward()
            self.optimizer.step()
        else:
            with torch.no_grad():
                # solver/base.py:409-421: loss finite (or within max_loss) and every gradient finite -- on one flat copy of the
                # gradients, in one launch (sdeh_guard_check), which also sanitises them: the step below must not see NaN / Inf
                # (its result is discarded when not ok).  Three launches instead of a few per parameter.
                grads = [p.grad for p in self._params if p.grad is not None]
                if any(g.dtype != torch.float32 for g in grads) or value.dtype != torch.float32:
                    raise RuntimeError("GraphedTrainStep(guard=True): fp32 loss and gradients expected")
                flat = torch.cat([g.reshape(-1) for g in grads])
                ok = torch.empty(1, dtype=torch.bool, device=self.device)
                with torch.cuda.device(self.device):
                    L.check(L.load().sdeh_guard_check(flat.data_ptr(), flat.numel(), value.detach().reshape(1).data_ptr(),
                                                      -1.0 if self.max_loss is None else float(self.max_loss), ok.data_ptr(),
                                                      torch.cuda.current_stream(self.device).cuda_stream))
                torch._foreach_copy_(grads, [c.view_as(g) for c, g in zip(flat.split([g.numel() for g in grads]), grads)])
            if self._after_backward is not None:
                self._after_backward()
            with torch.no_grad():
                tensors = [t.detach() for t in self._guarded()]
                if self._snap is None or len(self._snap) != len(tensors):  # (re)built while the optimizer state appears (warm-up)
                    self._snap = [torch.empty_like(t) for t in tensors]
                torch._foreach_copy_(self._snap, tensors)
            self.optimizer.step()
            with torch.no_grad():
                tensors = [t.detach() for t in self._guarded()]
                if len(tensors) == len(self._snap):
                    # new <- old where the step is rejected: ONE launch over a table of (tensor, snapshot) pairs (sdeh_guard_restore)
                    # instead of two framework kernels per tensor (~90 tensors with Adam: 0.4 ms of a replayed step at batch 512)
                    self._restore(tensors, ok)
                else:
                    self.n_skipped += (~ok).reshape(()).to(torch.int64)
        self.counter.add_(1)
        return value.detach()

    def __call__(self) -> torch.Tensor:
        """One optimisation step (a single graph launch).  Returns the loss tensor of that step (device, no sync)."""
        self.graph.replay()
        self.replays += 1
        return self.loss


class GraphedEval:
    """One evaluation (`loss.eval(ts, x, ...)`: coefficient tables, trajectory kernel, estimator reduction, importance weights) as ONE
    graph launch plus the 8-float device->host copy its `Results` need.

    At the reference's evaluation sizes the host side of an eager `loss.eval` -- describing the problem, four ctypes launches, the
    output allocations -- costs as much as the kernels (B = 1024, T = 100: 0.28 ms of kernels, ~0.1 ms of host work before the
    first launch); replayed, the call is a copy of `x` into the captured input, one launch and the synchronising copy that the
    reference's `.item()`s are as well.

    eval_fn   (x) -> Results: calls `loss.eval(ts, x, ...)` of ONE of `losses` (e.g. `problems.Problem.eval`); must not synchronise
              otherwise and must not run data-parallel (no process group: the estimator merge across ranks goes through the host)
    losses    the loss objects involved (they get the device-resident Philox counter, as under GraphedTrainStep; pass `counter=` to
              share one with a captured training step)
    x         example input [B, d]; calls take inputs of this shape

    The tensors inside the returned `Results` (samples, weights, xs) are the graph's static outputs: the next call overwrites them."""

    def __init__(self, eval_fn: Callable, losses: Iterable, x: torch.Tensor, *, warmup: int = 2, counter: torch.Tensor | None = None):
        import torch.distributed as dist

        from sde_sampler_amd import engine as E

        if not x.is_cuda:
            raise RuntimeError("GraphedEval needs a GPU (hipGraph capture)")
        if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
            raise RuntimeError("GraphedEval: data-parallel evaluations merge their estimators through the host; evaluate eagerly")
        self.device = x.device
        self.losses = list(losses)
        self._eval_fn = eval_fn
        self.x = x.detach().clone()
        self.counter = counter if counter is not None else next(
            (lo.rng_counter for lo in self.losses if lo.rng_counter is not None), None)
        if self.counter is None:
            self.counter = torch.full((1,), COUNTER_START, dtype=torch.int64, device=self.device)
        for lo in self.losses:
            lo.rng_counter = self.counter
        side = torch.cuda.Stream(device=self.device)
        side.wait_stream(torch.cuda.current_stream(self.device))
        with torch.cuda.stream(side), torch.no_grad():
            for _ in range(max(int(warmup), 1)):
                eval_fn(self.x)
                self.counter.add_(1)
        torch.cuda.current_stream(self.device).wait_stream(side)
        torch.cuda.synchronize(self.device)
        self.graph = torch.cuda.CUDAGraph()
        self._parts: dict = {}
        E._deferred = self._parts
        try:
            with torch.cuda.graph(self.graph), torch.no_grad():
                if eval_fn(self.x) is not None:
                    raise RuntimeError("GraphedEval: eval_fn did not go through BaseOCLoss.compute_results")
                self.counter.add_(1)
        finally:
            E._deferred = None
        self.replays = 0

    def __call__(self, x: torch.Tensor | None = None):
        from sde_sampler_amd import engine as E
        from sde_sampler_amd.losses.oc import BaseOCLoss

        if x is not None and x is not self.x:
            self.x.copy_(x)
        self.graph.replay()
        self.replays += 1
        p = self._parts
        return BaseOCLoss.finish_results(E.merge_stats(p["stats"]), p["weights"], p["compute_weights"], p["ts"], p["samples"], p["xs"])

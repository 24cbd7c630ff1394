"""hipGraph capture of a whole training step (sample -> fused trajectory forward -> HIP backward -> weight-gradient GEMMs ->
optimizer step) for the launch-bound regime.

At the reference's training sizes (batch 512 ... 4096, 100 ... 200 steps) one optimisation step of this package is about forty
short kernels: the trajectory kernel itself is latency-bound (7 us per Euler-Maruyama step) and the host spends longer
issuing the rest than the GPU spends running it.  `GraphedTrainStep` records the step once on a side stream
(`torch.cuda.CUDAGraph`, i.e. hipStreamBeginCapture / hipGraphLaunch underneath; the ctypes launches of libsdeh.so go to
torch's current stream and are captured like torch's own kernels) and replays it with one launch.

Two things make the step replayable:

* fresh noise per replay -- launch arguments (seed, offset) are frozen at capture time, so the kernels add a device-resident
  counter to the Philox offset (`SdehProblem.rng_offset_dev`, include/sdeh.h) and the graph bumps it after the optimizer step;
  the backward launch replays the forward's draws because it reads the same counter value;
* no host round trip -- `loss.graph_safe = True` switches `compute_loss` to masked reductions (losses/oc.py).

What is NOT available in a captured step: the host-side `if loss_ok and grad_ok` of the reference trainer
(solver/base.py:409-432) -- use `max_rnd` / `filter_samples` (they act on the device) and check `GraphedTrainStep.loss`
from time to time instead; data-parallel loss shares (they go through the host).
"""
from __future__ import annotations

from typing import Callable, Iterable

import torch

__all__ = ["GraphedTrainStep"]

#: first value of the device counter: far above anything `engine.calls` reaches, so that replays and eager launches of the
#: same seed never share a Philox offset
COUNTER_START = 1 << 40


class GraphedTrainStep:
    """Captures `optimizer.zero_grad(); l = loss_fn(); l.backward(); [after_backward()]; optimizer.step()` into one graph.

    loss_fn          () -> scalar loss tensor; may draw samples with torch's device generator, must not synchronise and must
                     only read tensors that stay alive (parameters, `ts`, buffers of the prior / target)
    losses           the loss objects `loss_fn` calls (their `rng_counter` / `graph_safe` are set here)
    optimizer        for Adam/AdamW pass `capturable=True`
    after_backward   optional () -> None between backward and the optimizer step (gradient clipping, ...), also captured
    warmup           eager steps on the side stream before capturing (allocator warm-up; they DO update the parameters)
    """

    def __init__(self, loss_fn: Callable[[], torch.Tensor], losses: Iterable, optimizer: torch.optim.Optimizer, *,
                 after_backward: Callable[[], None] | None = None, warmup: int = 3, device=None):
        if not torch.cuda.is_available():
            raise RuntimeError("GraphedTrainStep needs a GPU (hipGraph capture)")
        for group in optimizer.param_groups:
            if "capturable" in group and not group["capturable"]:
                raise ValueError(f"{type(optimizer).__name__} must be created with capturable=True to be captured in a graph")
        self.device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        self.losses = list(losses)
        self.optimizer = optimizer
        self._loss_fn, self._after_backward = loss_fn, after_backward
        self.counter = torch.full((1,), COUNTER_START, dtype=torch.int64, device=self.device)
        for lo in self.losses:
            lo.rng_counter = self.counter
            lo.graph_safe = True
        self.replays = 0

        side = torch.cuda.Stream(device=self.device)
        side.wait_stream(torch.cuda.current_stream(self.device))
        with torch.cuda.stream(side):
            for _ in range(max(int(warmup), 1)):
                self._step()
        torch.cuda.current_stream(self.device).wait_stream(side)
        torch.cuda.synchronize(self.device)

        self.graph = torch.cuda.CUDAGraph()
        optimizer.zero_grad(set_to_none=True)
        with torch.cuda.graph(self.graph):
            self.loss = self._step()  # static output: overwritten by every replay

    def _step(self) -> torch.Tensor:
        self.optimizer.zero_grad(set_to_none=True)
        value = self._loss_fn()
        value.backward()
        if self._after_backward is not None:
            self._after_backward()
        self.optimizer.step()
        self.counter.add_(1)
        return value.detach()

    def __call__(self) -> torch.Tensor:
        """One optimisation step (a single graph launch).  Returns the loss tensor of that step (device, no sync)."""
        self.graph.replay()
        self.replays += 1
        return self.loss

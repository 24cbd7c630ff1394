"""Optimal-control losses whose `simulate()` runs on the MI355X trajectory engine (libsdeh.so).

Drop-in for the reference's sde_sampler/losses/oc.py: same class names, constructor keywords, `loss(ts, x, ...)`
/ `loss.eval(...) -> Results` / `state_dict()` contracts (BaseOCLoss 13-137, TimeReversalLoss 140-278,
ReferenceSDELoss 281-391, ExponentialIntegratorSDELoss 394-505), selected by pointing a Hydra `_target_` at
`sde_sampler_amd.losses.oc.<Class>`.  What changes is *how* simulate() is computed:

* the whole time-stepping loop is one persistent HIP kernel launch (sde_sampler_amd/csrc/sdeh_traj.hpp);
* Gaussian noise comes from the kernel's own Philox4x32-10 stream, keyed by `torch.initial_seed()` and counted
  per (call, trajectory, step) -- or from an explicit `noise=[T,B,d]` tensor (parity mode), which reproduces
  the reference's result for the same draws;
* the batch reductions of `compute_results` run on device and, under `torch.distributed`, merge across ranks
  with one 8-float all-gather.

Training: `loss(...)` back-propagates through the fused kernels for every method (losses/_autograd.py +
`sdeh_ctrl_backward`): row-parallel for "lv" / "lv_traj" (detached SDE control), back-propagation through time with the
adjoint kept in registers for "kl" / "kl_ito"; the Bridge (`inference_ctrl`, exact or Hutchinson divergence) trains through
`sdeh_ctrl_backward_ex` + `sdeh_bridge_div_backward`; wide networks (channels 128 / 256, or 64 with d > 64) train through the
kernels of `csrc/sdeh_wide_bwd.hip` (every method, closed-form and mixture targets, the Bridge with the exact divergence).  Not
built in (raises `SdehUnsupported`, never falls back): `sde_ctrl_noise` / `sde_ctrl_dropout` (dead code in the reference,
DESIGN.md section 7) and the Hutchinson divergence estimators on wide networks.
"""
from __future__ import annotations

import logging
import os
import math
from typing import Callable

import torch

from sde_sampler_amd import _lib as L
from sde_sampler_amd import engine as E
from sde_sampler_amd.utils.common import Results


def _resolve_terminal(fn: Callable):
    """terminal_unnorm_log_prob -> (distribution, clip_target) when it can be fused, else (None, None).

    Recognised: `target.unnorm_log_prob` of a built-in distribution, and the solver's
    `clipped_target_unnorm_log_prob` (reference solver/oc.py:48-54: clip(target.unnorm_log_prob(x), clip_target))."""
    owner = getattr(fn, "__self__", None)
    name = getattr(fn, "__name__", None)
    if owner is None:
        return None, None
    # (a NICE flow is "known" too: its log-density / score are HIP kernels of their own, evaluated by the engine around the trajectory
    # kernels' step segments -- engine.run, SDEH_DENS_EXTERNAL)
    if name == "unnorm_log_prob" and (E._known_distribution(owner) or E._external_target(owner)):
        return owner, None
    if name == "clipped_target_unnorm_log_prob" and hasattr(owner, "target") and (
            E._known_distribution(owner.target) or E._external_target(owner.target)):
        return owner.target, getattr(owner, "clip_target", None)
    return None, None


def _resolve_gaussian_log_prob(fn: Callable | None):
    """initial_log_prob / reference_log_prob -> the Gaussian distribution behind `dist.log_prob`, else None."""
    owner = E._bound_owner(fn, "log_prob") if fn is not None else None
    if owner is not None and (set(E._mro_names(owner)) & E._GAUSS_NAMES or
                              ("GMM" in E._mro_names(owner) and owner.mixture_weights is None)):
        return owner
    return None


def _world_size(group=None) -> int:
    import torch.distributed as dist

    return dist.get_world_size(group) if dist.is_available() and dist.is_initialized() else 1


def _dist_on() -> bool:
    """True once torch.distributed is initialised -- at ANY world size: a one-rank process group takes the data-parallel code path
    (collectives included), so that what runs on one GPU under RCCL is what runs on eight (tests/test_hip_rccl.py)."""
    import torch.distributed as dist

    return dist.is_available() and dist.is_initialized()


def _all_reduce_sum(t: torch.Tensor, group=None) -> torch.Tensor:
    """SUM all-reduce of a few scalars (RCCL on the device for the nccl backend, through the host for gloo); returned on the CPU."""
    import torch.distributed as dist

    t = t.detach()
    t = t.cpu() if dist.get_backend(group) == "gloo" else t.to(torch.float64)
    t = t.clone()
    dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
    return t.cpu()


def _all_reduce_on_device(t: torch.Tensor, group=None) -> torch.Tensor:
    """SUM all-reduce of a few doubles that STAY on their device: no `.item()`, no host copy for the nccl (= RCCL) backend -- an
    ordinary stream-ordered operation, capturable into a hipGraph.  (gloo reduces host tensors: CUDA inputs take a detour there.)"""
    import torch.distributed as dist

    t = t.detach().to(torch.float64).clone()
    if t.is_cuda and dist.get_backend(group) == "gloo":
        host = t.cpu()
        dist.all_reduce(host, op=dist.ReduceOp.SUM, group=group)
        return host.to(t.device)
    dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
    return t


class _MaskedMoment(torch.autograd.Function):
    """mean (kl) or unbiased variance (lv) of the kept rows of rnd [B, 1] from the statistics vector of
    engine.estimator_stats (n, sum(-rnd), M2, ...); the gradient is elementwise: 1/n, or 2 (rnd_i - mean) / (n - 1), zero on
    dropped rows -- exactly what `rnd[mask].mean()` / `.var()` back-propagate (losses/oc.py:72-92)."""

    @staticmethod
    def forward(ctx, rnd, mask, stats, lv: bool):
        n = stats[0]
        mean = -stats[1] / n
        ctx.save_for_backward(rnd, mask, n, mean)
        ctx.lv = lv
        return (stats[2] / (n - 1.0) if lv else mean).to(rnd.dtype)

    @staticmethod
    def backward(ctx, grad_out):
        rnd, mask, n, mean = ctx.saved_tensors
        per_row = 2.0 * (rnd - mean) / (n - 1.0) if ctx.lv else (1.0 / n).expand_as(rnd)
        return torch.where(mask, per_row * grad_out, torch.zeros_like(rnd)), None, None, None


class _FusedMoment(torch.autograd.Function):
    """_MaskedMoment with filter, statistics, loss value and per-row gradient in THREE launches of the library (sdeh_loss_moment)
    instead of ~20 framework kernels between the trajectory kernel and the fused backward: the filter of
    losses/oc.py:50-58 is the reduction's own predicate, the running `n_filtered` is bumped on the device by the same pass."""

    @staticmethod
    def forward(ctx, rnd, max_rnd: float, lv: bool, n_filtered):
        stats, w = E.loss_moment(rnd.detach(), max_rnd, lv, n_filtered)
        ctx.save_for_backward(w)
        return stats[7]

    @staticmethod
    def backward(ctx, grad_out):
        (w,) = ctx.saved_tensors
        return w * grad_out, None, None, None


class _SharedMoment(torch.autograd.Function):
    """This rank's additive share of the GLOBAL batch's mean (kl) or unbiased variance (lv) of the kept rows, from the local
    statistics (n_l, mean_l, M2_l about the local mean) and the global (n, mean) that ONE device-side all-reduce delivered:
        kl:  n_l mean_l / n            lv:  (M2_l + n_l (mean_l - mean)^2) / (n - 1)      (Chan: the local M2 moved to the global mean)
    -- no second pass over the rows.  Gradient per kept row: 1 / n, or 2 (rnd_i - mean) / (n - 1): with the global mean taken as a
    constant these are exact, the omitted term is proportional to sum_i (rnd_i - mean) = 0 over the global batch."""

    @staticmethod
    def forward(ctx, rnd, mask, n_l, mean_l, m2_l, n, mean, lv: bool):
        # (the per-row gradient in the rows' own precision from the rounded global moments: at world size 1 the operations -- and the
        # results, bit for bit -- of _MaskedMoment)
        ctx.save_for_backward(rnd, mask, n.to(rnd.dtype), mean.to(rnd.dtype))
        ctx.lv = lv
        share = (m2_l + n_l * (mean_l - mean) ** 2) / (n - 1.0) if lv else n_l * mean_l / n
        return share.to(rnd.dtype)

    @staticmethod
    def backward(ctx, grad_out):
        rnd, mask, n, mean = ctx.saved_tensors
        per_row = (2.0 * (rnd - mean) / (n - 1.0) if ctx.lv else (1.0 / n).expand_as(rnd)).to(rnd.dtype)
        return torch.where(mask, per_row * grad_out, torch.zeros_like(rnd)), None, None, None, None, None, None, None


_GRAPH_COUNTER_START = 1 << 40  # = utils.graphs.COUNTER_START (first value of a captured step's device-resident Philox counter)


class BaseOCLoss:
    #: set to a torch.distributed process group (or leave None for the default group); evaluation statistics are
    #: merged across ranks whenever torch.distributed is initialised (at any world size)
    process_group = None

    def __init__(self, generative_ctrl: Callable, sde=None, method: str = "kl", traj_per_sample: int = 1,
                 filter_samples: Callable | None = None, max_rnd: float | None = None,
                 sde_ctrl_dropout: float | None = None, sde_ctrl_noise: float | None = None, **kwargs):
        self.generative_ctrl = generative_ctrl
        self.sde = sde
        if method not in ["kl", "kl_ito", "lv", "lv_traj"]:
            raise ValueError("Unknown loss method.")
        self.method = method
        if traj_per_sample == 1 and self.method == "lv_traj":
            raise ValueError("Cannot compute variance over a single trajectory.")
        self.traj_per_sample = traj_per_sample
        self.filter_samples = filter_samples
        self.max_rnd = max_rnd
        self.sde_ctrl_noise = sde_ctrl_noise
        self.sde_ctrl_dropout = sde_ctrl_dropout
        if self.method in ["kl", "kl_ito"] and (sde_ctrl_noise is not None or sde_ctrl_dropout is not None):
            logging.warning("sde_ctrl_noise / sde_ctrl_dropout should only be used for the log-variance loss.")
        self.n_filtered = 0
        self.engine = E.TrajectoryEngine()
        #: row index of this rank's first trajectory in the global batch (keeps the Philox streams of data-parallel
        #: ranks disjoint).  None (default): rank * local batch when torch.distributed is initialised, else 0.
        self.row_offset: int | None = None
        #: optional one-element int64 device tensor ADDED to the Philox offset inside the kernels (SdehProblem.rng_offset_dev):
        #: launch arguments are frozen when a step is captured into a hipGraph, this counter is what moves between replays
        self.rng_counter: torch.Tensor | None = None
        #: True: `compute_loss` runs without host synchronisation (masked reductions instead of boolean indexing, the
        #: filtered-sample count kept on the device) so that a whole training step can be captured (utils/graphs.py)
        self.graph_safe = False
        self._n_filtered_dev: torch.Tensor | None = None
        #: data-parallel runs: also report the loss of the GLOBAL batch as `train/loss_global` (an extra all-reduce and a host
        #: synchronisation per step; the shares returned by `compute_loss` sum to it)
        self.report_global_loss = False

    # -- filtering / loss value (reference 50-92) ------------------------------------------------------------
    def filter(self, rnd: torch.Tensor, samples: torch.Tensor | None = None) -> torch.Tensor:
        mask = True
        if samples is not None and self.filter_samples is not None:
            mask = self.filter_samples(samples)
        if self.max_rnd is None:
            return mask & rnd.isfinite()
        return mask & (rnd < self.max_rnd)

    def compute_loss(self, rnd: torch.Tensor, samples: torch.Tensor | None = None) -> tuple[torch.Tensor, dict]:
        """The reference's loss over the batch (losses/oc.py:72-92).  Data-parallel (torch.distributed initialised with
        world_size > 1): the loss of the GLOBAL batch, returned as this rank's additive share -- the shares sum to the global
        loss and, after a SUM all-reduce of the parameter gradients (`utils.distributed.all_reduce_gradients`), every rank
        holds the gradient of the global loss.  The log-variance loss needs the global mean of `rnd` first (one 3-float
        all-reduce); with the mean taken as a constant the per-row gradients 2 (rnd_i - mean) / (N - 1) are exact,
        because the omitted term is proportional to sum_i (rnd_i - mean) = 0 over the global batch."""
        if (self.graph_safe and self.method in ("kl", "kl_ito", "lv") and rnd.is_cuda and rnd.dtype == torch.float32 and not _dist_on()
                and (samples is None or self.filter_samples is None)):
            # filter + statistics + loss + per-row gradient in three launches (sdeh_loss_moment); NaN-free: dropped rows get weight 0
            if self._n_filtered_dev is None:
                self._n_filtered_dev = torch.zeros((), device=rnd.device, dtype=torch.int64)
            loss = _FusedMoment.apply(rnd, math.inf if self.max_rnd is None else float(self.max_rnd), self.method == "lv", self._n_filtered_dev)
            return loss, {"train/n_filtered_cumulative": self._n_filtered_dev}
        mask = self.filter(rnd, samples=samples)
        assert mask.shape == rnd.shape
        world = _world_size(self.process_group)
        if self.graph_safe:
            return self._compute_loss_on_device(rnd, mask)
        if self.method == "lv_traj":
            rnd = rnd.reshape(self.traj_per_sample, -1, 1)
            mask = mask.reshape(self.traj_per_sample, -1, 1).all(dim=0)
            filtered = self.traj_per_sample * (mask.numel() - mask.sum())
            if not _dist_on():
                self.n_filtered += filtered.item()
                return rnd[:, mask].var(dim=0).mean(), {"train/n_filtered_cumulative": self.n_filtered}
            per_sample = rnd[:, mask].var(dim=0)
            tot = _all_reduce_sum(torch.stack([mask.sum().double(), filtered.double()]), self.process_group)
            self.n_filtered += int(tot[1].item())
            loss = per_sample.sum() / tot[0].item()
        else:
            filtered = mask.numel() - mask.sum()
            if not _dist_on():
                self.n_filtered += filtered.item()
                loss = rnd[mask].var() if self.method == "lv" else rnd[mask].mean()
                return loss, {"train/n_filtered_cumulative": self.n_filtered}
            kept = rnd[mask]
            tot = _all_reduce_sum(torch.stack([mask.sum().double(), kept.detach().double().sum(), filtered.double()]),
                                  self.process_group)
            n_glob, mean_glob = tot[0].item(), tot[1].item() / tot[0].item()
            self.n_filtered += int(tot[2].item())
            loss = ((kept - mean_glob) ** 2).sum() / (n_glob - 1) if self.method == "lv" else kept.sum() / n_glob
        metrics = {"train/n_filtered_cumulative": self.n_filtered}
        if self.report_global_loss:  # one more all-reduce + a host round trip: off the hot path unless asked for
            metrics["train/loss_global"] = _all_reduce_sum(loss.detach().double().reshape(1), self.process_group).item()
        return loss, metrics

    def _compute_loss_on_device(self, rnd: torch.Tensor, mask: torch.Tensor) -> tuple[torch.Tensor, dict]:
        """compute_loss without a host round trip: same estimators (losses/oc.py:72-92) as masked reductions.  Rows the filter
        drops contribute exactly zero value and zero gradient, as `rnd[mask]` does.  Data-parallel (a process group exists, at any
        world size): ONE device-side all-reduce of [n, sum rnd, n_filtered] (3 doubles; RCCL on the device tensor, capturable as a
        graph node) gives the global count and mean, the rank's share of the global loss follows from its local statistics
        (`_SharedMoment`) -- no second collective, no `.item()`."""
        zero = torch.zeros((), device=rnd.device, dtype=rnd.dtype)
        dp = _dist_on()
        if self._n_filtered_dev is None:
            self._n_filtered_dev = torch.zeros((), device=rnd.device, dtype=torch.int64)
        if self.method != "lv_traj":
            if rnd.is_cuda and rnd.dtype == torch.float32:
                # Sums through the library's own two-pass reduction (sdeh_reduce_estimators), elementwise ops otherwise: a captured
                # step must not contain the framework's multi-block reductions (utils/graphs.py, tests/perf/rocm_graph_two_reductions.py)
                stats = E.estimator_stats(torch.where(mask, rnd.detach(), torch.full_like(rnd, math.nan)), max_rnd=math.inf)  # +inf: keep finite rows
                if not dp:
                    self._n_filtered_dev += stats[6].to(torch.int64)
                    return _MaskedMoment.apply(rnd, mask, stats, self.method == "lv"), {"train/n_filtered_cumulative": self._n_filtered_dev}
                n_l, s_l, m2_l, nf_l = stats[0].double(), -stats[1].double(), stats[2].double(), stats[6].double()
            else:
                kept = mask.sum()
                n_l, nf_l = kept.double(), (mask.numel() - kept).double()
                s_l = torch.where(mask, rnd.detach(), zero).double().sum()
                m2_l = (torch.where(mask, rnd.detach().double() - s_l / n_l, zero.double()) ** 2).sum()
                if not dp:
                    mean = torch.where(mask, rnd, zero).sum() / kept
                    loss = (torch.where(mask, rnd - mean, zero) ** 2).sum() / (kept - 1) if self.method == "lv" else mean
                    self._n_filtered_dev += (mask.numel() - kept)
                    return loss, {"train/n_filtered_cumulative": self._n_filtered_dev}
            glob = _all_reduce_on_device(torch.stack([n_l, s_l, nf_l]), self.process_group)
            self._n_filtered_dev += glob[2].to(torch.int64)
            loss = _SharedMoment.apply(rnd, mask, n_l, s_l / n_l, m2_l, glob[0], glob[1] / glob[0], self.method == "lv")
            return loss, {"train/n_filtered_cumulative": self._n_filtered_dev}
        rnd = rnd.reshape(self.traj_per_sample, -1, 1)
        mask = mask.reshape(self.traj_per_sample, -1, 1).all(dim=0)
        kept = mask.sum()
        filtered = self.traj_per_sample * (mask.numel() - kept)
        clean = torch.where(mask.expand_as(rnd), rnd, zero)
        total = torch.where(mask, clean.var(dim=0), zero).sum()
        if dp:  # the samples of a trajectory group live on one rank: the shares are sums of per-sample variances over the global count
            glob = _all_reduce_on_device(torch.stack([kept.double(), filtered.double()]), self.process_group)
            kept, filtered = glob[0].to(rnd.dtype), glob[1].to(torch.int64)
        self._n_filtered_dev += filtered
        return total / kept, {"train/n_filtered_cumulative": self._n_filtered_dev}

    # -- evaluation statistics (reference 94-123) ---------------------------------------------------------------
    @staticmethod
    def compute_results(rnd: torch.Tensor, compute_weights: bool = False, ts: torch.Tensor | None = None,
                        samples: torch.Tensor | None = None, xs: torch.Tensor | None = None, group=None) -> Results:
        """log Z estimators from the per-trajectory `rnd` via one device reduction (+ one 8-float all-gather when
        running data-parallel).  The values are those of the reference formulas (neg_rnd.mean(), log mean exp,
        rnd.var()) over the GLOBAL batch; `samples` / `weights` stay rank-local shards."""
        local = E.estimator_stats(rnd)
        if E._deferred is not None:
            # utils.graphs.GraphedEval is capturing: leave the device half (reduction, importance weights against the LOCAL maximum --
            # there is no process group in a captured evaluation) in the graph and the host half (`finish_results`) to the replay
            weights = E.importance_weights(rnd, local[3:4]) if compute_weights else None
            E._deferred.update(stats=local, weights=weights, compute_weights=compute_weights, ts=ts, samples=samples, xs=xs)
            return None
        stats = E.all_gather_stats(local, group=group)
        weights = None
        if compute_weights:
            m = torch.as_tensor(float(stats[3]), dtype=torch.float32, device=rnd.device)
            weights = E.importance_weights(rnd, m)
        return BaseOCLoss.finish_results(stats, weights, compute_weights, ts, samples, xs)

    @staticmethod
    def finish_results(stats: torch.Tensor, weights, compute_weights: bool, ts, samples, xs) -> Results:
        """The host half of compute_results: estimators from the merged statistics."""
        est = E.estimators_from_stats(stats)
        metrics = {}
        if compute_weights:
            preds = {"log_norm_const_lb_ito": est["mean_neg_rnd"], "log_norm_const_is": est["log_norm_const_is"]}
            metrics["eval/lv_loss"] = est["var_rnd"]
        else:
            preds = {"log_norm_const_lb": est["mean_neg_rnd"]}
        return Results(samples=samples, weights=weights, log_norm_const_preds=preds, ts=ts, xs=xs, metrics=metrics)

    def __call__(self, ts: torch.Tensor, x: torch.Tensor, *args, **kwargs):
        raise NotImplementedError

    def eval(self, ts: torch.Tensor, x: torch.Tensor, *args, **kwargs) -> Results:
        raise NotImplementedError

    def load_state_dict(self, state_dict: dict, rewind: bool = False):
        """The reference's {"n_filtered"} (losses/oc.py:133-137) + the position of the noise stream.  By default the Philox position NEVER
        moves backwards (a resumed run must not reuse offsets this process has already consumed); `rewind=True` sets it exactly to the
        checkpoint's, which is what reproducing an earlier stretch of a run in the same process needs."""
        self.n_filtered = state_dict["n_filtered"]
        if self._n_filtered_dev is not None:
            self._n_filtered_dev.zero_()
        # extension of the reference's {"n_filtered"} (losses/oc.py:133-137): a resumed run continues the noise stream.  The Philox
        # offset of a launch is (stream_id << 40) + calls + *rng_counter; under a replayed hipGraph the device counter starts at
        # utils.graphs.COUNTER_START (apart from every eager offset) and advances by one per replay.  What is saved is the number of
        # REPLAYS, so that a checkpoint moves between eager and graphed runs without wiping COUNTER_START (replays would reuse eager
        # offsets) or carrying 2^40 into the stream-id bits of `calls`:
        #   graphed -> graphed: counter = max(current, COUNTER_START + replays)   (an eager checkpoint has replays = 0: the counter stays
        #   eager -> graphed:   where the warm-up and capture steps of GraphedTrainStep put it -- they are real optimizer steps, and a
        #                       rewind below them would reuse Philox offsets already consumed)
        #   graphed -> eager:   calls += replays                         checkpoints without the keys: nothing is touched
        self.engine.calls = int(state_dict.get("rng_calls", self.engine.calls))
        replays = state_dict.get("rng_replays")
        if replays is None and int(state_dict.get("rng_counter", 0)) >= _GRAPH_COUNTER_START:  # checkpoints of the first format
            replays = int(state_dict["rng_counter"]) - _GRAPH_COUNTER_START
        if replays is not None:
            if self.rng_counter is not None:
                if rewind:
                    self.rng_counter.fill_(_GRAPH_COUNTER_START + int(replays))
                else:
                    self.rng_counter.copy_(torch.clamp(self.rng_counter, min=_GRAPH_COUNTER_START + int(replays)))  # never rewinds
            else:
                self.engine.calls += int(replays)

    def state_dict(self) -> dict:
        on_device = 0 if self._n_filtered_dev is None else int(self._n_filtered_dev.item())
        counter = 0 if self.rng_counter is None else int(self.rng_counter.item())
        replays = counter - _GRAPH_COUNTER_START if counter >= _GRAPH_COUNTER_START else 0
        return {"n_filtered": self.n_filtered + on_device, "rng_calls": self.engine.calls, "rng_counter": counter,
                "rng_replays": replays}

    def _row_offset(self, local_batch: int) -> int:
        """Global index of this rank's first trajectory.  The default, rank * local batch, assumes EQUAL local batches on all
        ranks (what the solvers' `eval_batch_size // world` split gives); set `row_offset` explicitly for ragged shards."""
        if self.row_offset is not None:
            return int(self.row_offset)
        import torch.distributed as dist

        if dist.is_available() and dist.is_initialized():
            return dist.get_rank(self.process_group) * int(local_batch)
        return 0

    # -- shared plumbing ----------------------------------------------------------------------------------------
    _LOSS_KIND = None

    def _check_common(self, change_sde_ctrl: bool):
        if change_sde_ctrl and (self.sde_ctrl_noise is not None or self.sde_ctrl_dropout is not None):
            raise L.SdehUnsupported(-2, "sde_ctrl_noise / sde_ctrl_dropout are not built into the HIP engine")

    def _launch(self, ts, x, *, flags: int, terminal_unnorm_log_prob: Callable, second_log_prob: Callable | None,
                second_at_start: bool, second_at_end: bool, return_traj: bool, noise, reference_prior=None,
                alpha: float = 0.0, sigma: float = 0.0, inference_ctrl=None, div_noise=None):
        """Common body of the three simulate() methods: fuse what can be fused, call back what cannot."""
        target, clip_target = _resolve_terminal(terminal_unnorm_log_prob)
        second = _resolve_gaussian_log_prob(second_log_prob)
        if target is not None:
            flags |= L.FLAG_TERMINAL_TARGET
        if second is not None:
            if second_at_start:
                flags |= L.FLAG_INIT_LOGP
            if second_at_end:
                flags |= L.FLAG_TERMINAL_SECOND
        problem_kwargs = dict(loss_kind=self._LOSS_KIND, generative_ctrl=self.generative_ctrl, sde=self.sde, flags=flags,
                              terminal_target=target, clip_target=clip_target, second=second,
                              reference_prior=reference_prior, alpha=alpha, sigma=sigma, inference_ctrl=inference_ctrl,
                              rng_counter=self.rng_counter)

        def run(return_traj: bool, want_state: bool = False, want_gp: bool = False, want_planes: bool = False, split: bool = False):
            # split (Bridge training, losses/_autograd.py::_BridgeSplitFn): the problem WITHOUT its inference control -- whose terms are
            # row-parallel given the trajectory and the control -- keeping the fused backward's planes and u_t
            keep = E._Keep()
            pr = self.engine.build_problem(device=x.device, keep=keep, **(dict(problem_kwargs, inference_ctrl=None) if split else problem_kwargs))
            row_offset = self._row_offset(x.shape[0])
            offset = self.engine.offset()
            seed = torch.initial_seed()
            out = self.engine.run(pr, ts, x, noise=noise, return_traj=return_traj, keep=keep,
                                  row_offset=row_offset, seed=seed, want_gp=want_gp, div_noise=None if split else div_noise,
                                  want_planes=want_planes, want_u=split)
            x_T, rnd, xs = out[:3]
            # user-supplied callables the engine does not recognise are evaluated as given (device tensors in/out)
            if second is None and second_log_prob is not None:
                if second_at_start:
                    rnd = rnd + second_log_prob(x)
                if second_at_end:
                    rnd = rnd + second_log_prob(x_T)
            if target is None:
                rnd = rnd - terminal_unnorm_log_prob(x_T)
            assert rnd.shape == (x.shape[0], 1)
            if want_state:
                state = dict(problem_kwargs=problem_kwargs, noise=noise, seed=seed & 0xFFFFFFFFFFFFFFFF, offset=offset,
                             row_offset=row_offset, div_noise=div_noise)
                if want_planes:
                    state["planes"] = out[3]  # (zt, nn) of the forward launch, or None
                if want_gp:
                    state["score_planes"] = out[4]  # (sc, tscore): wide Bridge on a mixture target, else (None, None)
                return (x_T, rnd, xs, out[3], state) if want_gp else (x_T, rnd, xs, state)
            return x_T, rnd, xs

        needs_graph = torch.is_grad_enabled() and any(
            p.requires_grad for mod in (self.generative_ctrl, inference_ctrl)
            for p in getattr(mod, "parameters", lambda: [])())
        kl_unfused = not (flags & L.FLAG_CHANGE_SDE_CTRL) and (target is None or (second is None and second_log_prob is not None))
        if needs_graph and inference_ctrl is not None:
            if kl_unfused:
                raise L.SdehUnsupported(-2, "Bridge training with method='kl'/'kl_ito' needs terminal / initial log-densities "
                                            "of built-in distributions (they are differentiated inside the kernel)")
            from sde_sampler_amd.losses._autograd import simulate_bridge_with_grad

            x_T, rnd, _ = simulate_bridge_with_grad(self, run, ts, x, inference_ctrl, flags=flags, div_noise=div_noise,
                                                    problem_kwargs=problem_kwargs)
            return x_T, rnd, None
        if needs_graph:
            if not (flags & L.FLAG_CHANGE_SDE_CTRL) and (target is None or (second is None and second_log_prob is not None)):
                raise L.SdehUnsupported(
                    -2, "training with method='kl'/'kl_ito' back-propagates through the terminal log-densities inside "
                        "the kernel: terminal_unnorm_log_prob / reference_log_prob must be the methods of built-in "
                        "distributions (callables the engine cannot fuse are only supported for method='lv'/'lv_traj' "
                        "or under torch.no_grad())")
            from sde_sampler_amd.losses._autograd import simulate_with_grad

            x_T, rnd, _ = simulate_with_grad(self, run, ts, x)
            return x_T, rnd, None
        if inference_ctrl is not None and div_noise is None:  # no graph: the same split (plain launch + row-parallel inference pass)
            from sde_sampler_amd.losses._autograd import simulate_bridge_split

            out = simulate_bridge_split(self, run, ts, x, inference_ctrl, return_traj)
            if out is not None:
                return out
        return run(return_traj)

    def _train_call(self, ts, x, simulate_kwargs: dict):
        if self.traj_per_sample != 1:
            x = x.repeat(self.traj_per_sample, 1, 1).reshape(-1, x.shape[-1])
        samples, rnd, _ = self.simulate(ts, x, compute_ito_int=self.method != "kl",
                                        change_sde_ctrl=self.method in ["lv", "lv_traj"], return_traj=False,
                                        **simulate_kwargs)
        return self.compute_loss(rnd, samples=samples)


class TimeReversalLoss(BaseOCLoss):
    """DIS, and Bridge when `inference_ctrl` is given (the exact divergence of the inference control per step,
    losses/oc.py:189-202, or its Hutchinson estimate in training when `div_estimator` is "rademacher" / "gauss"; evaluation and
    training with every loss method): reference losses/oc.py:140-278."""

    _LOSS_KIND = L.LOSS_TIME_REVERSAL

    def __init__(self, *args, inference_ctrl: Callable | None = None, div_estimator: str | None = None, **kwargs):
        super().__init__(*args, **kwargs)
        self.inference_ctrl = inference_ctrl
        self.div_estimator = div_estimator
        if self.div_estimator is not None and self.inference_ctrl is None:
            logging.warning("Without inference control the divergence estimator has no effect.")

    def simulate(self, ts, x, terminal_unnorm_log_prob: Callable, initial_log_prob: Callable | None = None,
                 train: bool = True, compute_ito_int: bool = False, change_sde_ctrl: bool = False,
                 return_traj: bool = False, *, noise: torch.Tensor | None = None, div_noise: torch.Tensor | None = None):
        self._check_common(change_sde_ctrl)
        if self.inference_ctrl is not None and train and self.div_estimator is not None and div_noise is None:
            # probe vectors of the Hutchinson estimator (utils/autograd.py:25-42, n_samples = 1), one per trajectory and step
            shape = (ts.numel() - 1, *x.shape)
            if self.div_estimator == "rademacher":
                div_noise = torch.randint(0, 2, shape, device=x.device).float() * 2 - 1.0
            elif self.div_estimator == "gauss":
                div_noise = torch.randn(shape, device=x.device)
            else:
                raise NotImplementedError(f"Undefined noise type {self.div_estimator}.")
        if not (self.inference_ctrl is not None and train and self.div_estimator is not None):
            div_noise = None  # the estimator only acts in training (losses/oc.py:190)
        flags = (L.FLAG_TRAIN if train else 0) | (L.FLAG_ITO if compute_ito_int else 0) | \
                (L.FLAG_CHANGE_SDE_CTRL if change_sde_ctrl else 0)
        with_initial = not (train and self.method in ["kl", "kl_ito"])  # reference 168-172
        return self._launch(ts, x, flags=flags, terminal_unnorm_log_prob=terminal_unnorm_log_prob,
                            second_log_prob=initial_log_prob if with_initial else None, second_at_start=True,
                            second_at_end=False, return_traj=return_traj, noise=noise, inference_ctrl=self.inference_ctrl,
                            div_noise=div_noise)

    def __call__(self, ts, x, terminal_unnorm_log_prob: Callable, initial_log_prob: Callable | None, *, noise=None,
                 div_noise=None):
        kw = dict(terminal_unnorm_log_prob=terminal_unnorm_log_prob, initial_log_prob=initial_log_prob, train=True, noise=noise)
        if div_noise is not None:
            kw["div_noise"] = div_noise
        return self._train_call(ts, x, kw)

    def eval(self, ts, x, terminal_unnorm_log_prob: Callable, initial_log_prob: Callable | None = None,
             compute_weights: bool = True, return_traj: bool = True, *, noise=None) -> Results:
        with torch.no_grad():  # evaluation never builds a graph (the solver calls it under no_grad, solver/base.py:335-346)
            samples, rnd, xs = self.simulate(ts, x, terminal_unnorm_log_prob=terminal_unnorm_log_prob,
                                             initial_log_prob=initial_log_prob, compute_ito_int=compute_weights,
                                             train=False, return_traj=return_traj, noise=noise)
        return BaseOCLoss.compute_results(rnd, compute_weights=compute_weights, ts=ts, samples=samples, xs=xs,
                                          group=self.process_group)


class ReferenceSDELoss(BaseOCLoss):
    """PIS / EulerDDS: reference losses/oc.py:281-391."""

    _LOSS_KIND = L.LOSS_REFERENCE_SDE

    def __init__(self, *args, reference_ctrl: Callable | None = None, **kwargs):
        super().__init__(*args, **kwargs)
        self.reference_ctrl = reference_ctrl

    def _reference_prior(self):
        """EulerDDS passes `solver.reference_ctrl` = sde.diff(t,x) * prior.score(x) (solver/oc.py:305-306)."""
        if self.reference_ctrl is None:
            return None
        owner = getattr(self.reference_ctrl, "__self__", None)
        prior = getattr(owner, "prior", None)
        if getattr(self.reference_ctrl, "__name__", None) == "reference_ctrl" and prior is not None and \
                E._known_distribution(prior):
            return prior
        prior = getattr(self.reference_ctrl, "prior", None)  # a callable object carrying its prior
        if prior is not None and E._known_distribution(prior):
            return prior
        raise L.SdehUnsupported(-2, "reference_ctrl must be the solver's `reference_ctrl` (sigma * prior.score) "
                                    "or expose the Gaussian it uses as `.prior`")

    def simulate(self, ts, x, terminal_unnorm_log_prob: Callable, reference_log_prob: Callable,
                 compute_ito_int: bool = False, change_sde_ctrl: bool = False, return_traj: bool = False, *,
                 noise: torch.Tensor | None = None):
        self._check_common(change_sde_ctrl)
        prior = self._reference_prior()
        flags = (L.FLAG_ITO if compute_ito_int else 0) | (L.FLAG_CHANGE_SDE_CTRL if change_sde_ctrl else 0) | \
                (L.FLAG_REFERENCE_CTRL if prior is not None else 0)
        return self._launch(ts, x, flags=flags, terminal_unnorm_log_prob=terminal_unnorm_log_prob,
                            second_log_prob=reference_log_prob, second_at_start=False, second_at_end=True,
                            return_traj=return_traj, noise=noise, reference_prior=prior)

    def __call__(self, ts, x, terminal_unnorm_log_prob: Callable, reference_log_prob: Callable, *, noise=None):
        return self._train_call(ts, x, dict(terminal_unnorm_log_prob=terminal_unnorm_log_prob,
                                            reference_log_prob=reference_log_prob, noise=noise))

    def eval(self, ts, x, terminal_unnorm_log_prob: Callable, reference_log_prob: Callable | None = None,
             compute_weights: bool = True, return_traj: bool = True, *, noise=None) -> Results:
        with torch.no_grad():  # evaluation never builds a graph (the solver calls it under no_grad, solver/base.py:335-346)
            samples, rnd, xs = self.simulate(ts, x, terminal_unnorm_log_prob=terminal_unnorm_log_prob,
                                             reference_log_prob=reference_log_prob, compute_ito_int=compute_weights,
                                             change_sde_ctrl=False, return_traj=return_traj, noise=noise)
        return BaseOCLoss.compute_results(rnd, compute_weights=compute_weights, ts=ts, samples=samples, xs=xs,
                                          group=self.process_group)


class ExponentialIntegratorSDELoss(BaseOCLoss):
    """DDS with the exponential integrator of Vargas et al.: reference losses/oc.py:394-505."""

    _LOSS_KIND = L.LOSS_EXPONENTIAL

    def __init__(self, *args, alpha: float, sigma: float, **kwargs):
        super().__init__(*args, **kwargs)
        self.alpha = alpha
        self.sigma = sigma

    def simulate(self, ts, x, terminal_unnorm_log_prob: Callable, reference_log_prob: Callable,
                 compute_ito_int: bool = False, change_sde_ctrl: bool = False, return_traj: bool = False, *,
                 noise: torch.Tensor | None = None):
        self._check_common(change_sde_ctrl)
        flags = (L.FLAG_ITO if compute_ito_int else 0) | (L.FLAG_CHANGE_SDE_CTRL if change_sde_ctrl else 0)
        return self._launch(ts, x, flags=flags, terminal_unnorm_log_prob=terminal_unnorm_log_prob,
                            second_log_prob=reference_log_prob, second_at_start=False, second_at_end=True,
                            return_traj=return_traj, noise=noise, alpha=self.alpha, sigma=self.sigma)

    def __call__(self, ts, x, terminal_unnorm_log_prob: Callable, reference_log_prob: Callable, *, noise=None):
        return self._train_call(ts, x, dict(terminal_unnorm_log_prob=terminal_unnorm_log_prob,
                                            reference_log_prob=reference_log_prob, noise=noise))

    def eval(self, ts, x, terminal_unnorm_log_prob: Callable, reference_log_prob: Callable | None = None,
             compute_weights: bool = True, return_traj: bool = True, *, noise=None) -> Results:
        with torch.no_grad():  # evaluation never builds a graph (the solver calls it under no_grad, solver/base.py:335-346)
            samples, rnd, xs = self.simulate(ts, x, terminal_unnorm_log_prob=terminal_unnorm_log_prob,
                                             reference_log_prob=reference_log_prob, compute_ito_int=compute_weights,
                                             change_sde_ctrl=False, return_traj=return_traj, noise=noise)
        return BaseOCLoss.compute_results(rnd, compute_weights=compute_weights, ts=ts, samples=samples, xs=xs,
                                          group=self.process_group)

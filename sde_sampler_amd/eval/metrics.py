"""Evaluation metrics: same functions / keys as the reference's sde_sampler/eval/metrics.py (abs_and_rel_error 12-22,
compute_errors 25-61, frac_inside_domain 64-67, get_metrics 70-184).  Everything get_metrics reduces over the batch
(expectation functions plain and importance-weighted, ESS, per-coordinate means / standard deviations, fraction inside the
domain) comes from ONE pass of `sdeh_sample_stats` (include/sdeh.h) over the samples instead of ~25 separate torch
reductions, as sums that merge across ranks (`process_group`)."""
from __future__ import annotations

import logging
import math
from numbers import Number
from typing import Callable

import torch

from .. import _lib as L
from ..distr.base import EXPECTATION_FNS

_NAMES = ("square", "abs", "sum", "square_minus_sum")
assert tuple(EXPECTATION_FNS) == _NAMES  # the kernel's column order (include/sdeh.h: sdeh_sample_stats)


def abs_and_rel_error(prediction: Number, target: Number, suffix: str = "", eps: float = 1e-8) -> dict[str, float]:
    assert isinstance(prediction, Number)
    assert isinstance(target, Number)
    magnitude = abs(target) + eps
    error = abs(prediction - target)
    return {f"error{suffix}": error, f"rel_error{suffix}": error / magnitude}


def _with_errors(output: dict, target) -> dict:
    if target is not None:
        if not isinstance(target, Number):
            assert target.ndim == 0
            target = target.item()
        for key_name, pred in output.copy().items():
            output.update(abs_and_rel_error(prediction=pred, target=target, suffix=key_name.replace("eval", "")))
    return output


def compute_errors(prediction, target=None, name: str = "error", weights: torch.Tensor | None = None,
                   eps: float = 1e-8) -> dict[str, float]:
    output = {}
    if isinstance(prediction, Number):
        output[f"eval/{name}"] = prediction
    else:
        assert isinstance(prediction, torch.Tensor)
        if prediction.ndim == 0:
            output[f"eval/{name}"] = prediction.item()
        else:
            assert prediction.ndim == 2 and prediction.shape[-1] == 1
            output[f"eval/{name}"] = prediction.mean().item()
            if weights is not None:
                assert weights.shape == prediction.shape
                output[f"eval/{name}_is"] = ((prediction * weights).sum() / weights.sum()).item()
    return _with_errors(output, target)


def sample_stats(samples: torch.Tensor, weights: torch.Tensor | None = None, domain: torch.Tensor | None = None,
                 process_group=None) -> dict:
    """One kernel pass over samples [B, d] -> {"n", "sum_w", "sum_w2", "inside", "f": [4], "wf": [4], "mean": [d],
    "m2": [d]} as Python floats / CPU double tensors; merged over `process_group` when torch.distributed is initialised."""
    if not samples.is_cuda:
        raise RuntimeError("sample_stats runs on the HIP device only (got a CPU tensor); there is no CPU path in this package")
    dev, (B, d) = samples.device, samples.shape
    prep = lambda t: None if t is None else t.detach().to(device=dev, dtype=torch.float32).contiguous()
    x, w, dom = prep(samples), prep(weights), prep(domain)
    lib = L.load()
    scratch = torch.empty(lib.sdeh_sample_stats_scratch_floats(d), device=dev, dtype=torch.float32)
    out = torch.empty(12 + 2 * d, device=dev, dtype=torch.float32)
    ptr = lambda t: None if t is None else t.data_ptr()
    with torch.cuda.device(dev):
        L.check(lib.sdeh_sample_stats(x.data_ptr(), B, d, ptr(w), ptr(dom), scratch.data_ptr(), out.data_ptr(),
                                      torch.cuda.current_stream(dev).cuda_stream))
    rows = [out]
    import torch.distributed as dist

    if dist.is_available() and dist.is_initialized() and dist.get_world_size(process_group) > 1:
        world = dist.get_world_size(process_group)
        src = out.cpu() if dist.get_backend(process_group) == "gloo" else out
        bucket = torch.empty(world * out.numel(), dtype=src.dtype, device=src.device)
        dist.all_gather_into_tensor(bucket, src, group=process_group)
        rows = list(bucket.reshape(world, -1))
    return merge_sample_stats(torch.stack([r.double().cpu() for r in rows]), d)


def merge_sample_stats(rows: torch.Tensor, d: int) -> dict:
    """Chan merge of per-rank statistic vectors [R, 12 + 2d] (double, CPU)."""
    n = rows[:, 0]
    N = n.sum()
    mean = (rows[:, 12:12 + d] * n[:, None]).sum(0) / N
    m2 = (rows[:, 12 + d:12 + 2 * d] + n[:, None] * (rows[:, 12:12 + d] - mean) ** 2).sum(0)
    inside = rows[:, 3]
    return {"n": N.item(), "sum_w": rows[:, 1].sum().item(), "sum_w2": rows[:, 2].sum().item(),
            "inside": None if (inside < 0).any() else inside.sum().item(),
            "f": rows[:, 4:8].sum(0), "wf": rows[:, 8:12].sum(0), "mean": mean, "m2": m2}


def frac_inside_domain(samples, domain):
    assert samples.shape[-1] == domain.shape[0]
    st = sample_stats(samples, domain=domain)
    return st["inside"] / st["n"]


def get_metrics(distr, samples: torch.Tensor, weights: torch.Tensor | None = None,
                log_norm_const_preds: dict | None = None, expectation_preds: dict | None = None,
                marginal_dims: list[int] | None = None, sample_losses: dict[str, Callable] | None = None,
                process_group=None) -> dict[str, float]:
    marginal_dims = marginal_dims or []
    if not all(d < distr.dim for d in marginal_dims):
        logging.warning("Removing non-existent marginal dims for metrics.")
        marginal_dims = [d for d in marginal_dims if d < distr.dim]
    metrics = {}
    expectation_preds = expectation_preds or {}
    log_norm_const_preds = log_norm_const_preds or {}
    if weights is not None:
        assert weights.shape == (samples.shape[0], 1)
    st = sample_stats(samples, weights=weights, domain=distr.domain, process_group=process_group)
    n = st["n"]

    # Expectations (plain and importance-weighted means of the four expectation functions)
    for k, name in enumerate(_NAMES):
        target = distr.expectations.get(name)
        output = {f"eval/{name}": st["f"][k].item() / n}
        if weights is not None:
            output[f"eval/{name}_is"] = st["wf"][k].item() / st["sum_w"]
        metrics.update(_with_errors(output, target))
        if name in expectation_preds:
            metrics.update(compute_errors(prediction=expectation_preds[name], target=target, name=name + "_direct",
                                          weights=weights))

    # Log. normalization constant
    for name, pred in log_norm_const_preds.items():
        metrics.update(compute_errors(prediction=pred, target=distr.log_norm_const, name=name))

    # ESS
    if weights is not None:
        ess = st["sum_w"] ** 2 / st["sum_w2"]
        metrics["eval/effective_sample_size"] = ess
        metrics["eval/norm_effective_sample_size"] = ess / n

    # Stddevs (unbiased, as torch.std) and means per coordinate
    stddevs = (st["m2"] / (n - 1)).sqrt() if n > 1 else torch.full_like(st["m2"], math.nan)
    avg_stddev = stddevs.mean().item()
    metrics["eval/avg_stddev"] = avg_stddev
    for dim in marginal_dims:
        metrics[f"eval/stddev_{dim}"] = stddevs[dim].item()
        metrics[f"eval/avg_{dim}"] = st["mean"][dim].item()
    if distr.stddevs is not None:
        ref = distr.stddevs.detach().double().cpu()
        assert ref.shape == stddevs.shape
        metrics["error/avg_marginal_stddev"] = (stddevs - ref).abs().mean().item()
        metrics.update(compute_errors(prediction=avg_stddev, target=distr.stddevs.mean(), name="avg_stddev"))

    # Samples inside domain
    if distr.domain is not None:
        metrics["eval/frac_pred_in_domain"] = st["inside"] / n

    # Other losses based on samples of the distribution
    if sample_losses is not None:
        if hasattr(distr, "sample"):
            gt_samples = distr.sample((samples.shape[0],))
            assert gt_samples.shape == samples.shape
            if distr.domain is not None:
                metrics["eval/frac_groundtruth_in_domain"] = frac_inside_domain(gt_samples, distr.domain)
            metrics.update({"error/" + name: loss(samples, gt_samples).item() for name, loss in sample_losses.items()})
        else:
            logging.warning("Sampling not implemented for distribution %s.", distr.__class__.__name__)

    # Objective
    if hasattr(distr, "objective"):
        metrics["eval/obj_avg"] = distr.objective(samples.mean(dim=0, keepdims=True)).item()
        metrics["eval/avg_obj"] = distr.objective(samples).mean().item()
        metrics["eval/min_obj"] = distr.objective(samples).min().item()
    return metrics

#!/usr/bin/env python3
"""Golden vectors for the evaluation metrics (`metrics_*.npz`), produced by RUNNING THE REFERENCE's get_metrics
(eval/metrics.py:70-184) on fixed samples / importance weights.  Build container only (needs /root/reference).
The reference's Sinkhorn (eval/sinkhorn.py) cannot be run here (pykeops is not installed), so `sample_losses` is None."""
from __future__ import annotations

import json
import sys
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, str(Path(__file__).resolve().parent))
import make_golden as mg  # noqa: E402

from sde_sampler.eval.metrics import get_metrics  # noqa: E402

OUT = Path(__file__).resolve().parent

CASES = {
    "metrics_gmm2": dict(target=dict(kind="gmm", dim=2, name="fab"), B=4099, seed=3, spread=1.3, marginal_dims=[0, 1]),
    "metrics_funnel10": dict(target=dict(kind="funnel", dim=10), B=2048, seed=5, spread=0.8, marginal_dims=[0, 1, 12]),
    "metrics_dw1": dict(target=dict(kind="double_well", dim=1, separation=2.0, shift=1.5), B=1000, seed=7, spread=1.0,
                        marginal_dims=[0]),
}


def run_case(name, case):
    torch.manual_seed(1)
    target = mg.build_target(case["target"])
    if case["target"]["kind"] == "double_well":
        # the reference integrates these with torchquad (not installed): inject fixed reference statistics instead -- the
        # fixture pins get_metrics' arithmetic, not the quadrature
        target.expectations = {"square": 3.2, "abs": 1.7, "sum": 1.45}
        target.stddevs = torch.tensor([1.05])
        target.log_norm_const = 1.2345
    else:
        target.compute_stats()
    torch.manual_seed(case["seed"])
    d = case["target"]["dim"]
    B = case["B"]
    if hasattr(target, "sample"):
        samples = target.sample((B,)) * case["spread"] + 0.1 * torch.randn(B, d)
    else:
        samples = torch.randn(B, d) * 2.0 * case["spread"]
    weights = torch.exp(0.7 * torch.randn(B, 1))
    preds = {"log_norm_const_is": 0.123, "log_norm_const_lb": -1.5}
    md = [m for m in case["marginal_dims"]]
    out = {}
    for tag, w in (("w", weights), ("nw", None)):
        met = get_metrics(target, samples, weights=w, log_norm_const_preds=preds, marginal_dims=list(md))
        out[f"metrics_{tag}"] = np.frombuffer(json.dumps({k: float(v) for k, v in met.items()}).encode(), dtype=np.uint8)
    out["samples"] = samples.numpy()
    out["weights"] = weights.numpy()
    stats = dict(expectations={k: float(v) for k, v in target.expectations.items()},
                 log_norm_const=None if target.log_norm_const is None else float(target.log_norm_const),
                 has_stddevs=target.stddevs is not None, has_domain=target.domain is not None, preds=preds)
    if target.stddevs is not None:
        out["stddevs"] = target.stddevs.numpy()
    if target.domain is not None:
        out["domain"] = target.domain.numpy()
    out["meta"] = np.frombuffer(json.dumps(dict(case, name=name, stats=stats)).encode(), dtype=np.uint8)
    path = OUT / f"{name}.npz"
    np.savez_compressed(path, **out)
    print(f"{name:20s} B={B} d={d} keys={len(met)} {path.stat().st_size/1024:.1f} KB")


if __name__ == "__main__":
    torch.set_num_threads(1)
    for name, case in CASES.items():
        run_case(name, case)

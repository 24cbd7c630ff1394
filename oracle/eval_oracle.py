"""CPU oracle for the evaluation-side reductions -- TEST INFRASTRUCTURE, NOT PRODUCT CODE (same rules as em_oracle.py:
only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import it).

  sinkhorn_dense()   dense PyTorch restatement of Sinkhorn.compute (reference sde_sampler/eval/sinkhorn.py:63-178).
                     PARITY UNPINNED against a run of the reference: its [n, m] reductions go through `pykeops`
                     (requirements: pykeops, not installed in this image and not vendored), so the reference function cannot
                     be executed here.  The restatement follows the published algorithm line by line with the LazyTensor
                     reductions replaced by dense torch.logsumexp / sum / argmax, and is anchored on (a) the reference's call
                     site (`eval_sample_losses.sinkhorn(samples, gt_samples)`, eval/metrics.py:165-170, default arguments
                     p=2, eps=1e-3, max_iters=100, stop_thresh=1e-5 from conf/base.yaml:13-15), (b) the closed form of the
                     1-D optimal-transport cost (tests/test_hip_eval.py) and (c) the exact assignment cost in d > 1
                     (scipy.optimize.linear_sum_assignment): the entropic plan's cost lies in [OT, OT + eps log n]
                     (tests/test_oracle_golden.py, tests/test_hip_eval.py).
  metrics_reference_keys()  restatement of get_metrics (eval/metrics.py:70-184) -- PINNED: bit-compared (to fp32 rounding of
                     the reductions) with the output of the reference's own get_metrics in tests/golden/metrics_*.npz
                     (tests/golden/make_golden_metrics.py).
"""
from __future__ import annotations

import torch

Tensor = torch.Tensor


def sinkhorn_dense(x: Tensor, y: Tensor, w_x: Tensor | None = None, w_y: Tensor | None = None, p: int = 2,
                   eps: float = 1e-3, max_iters: int = 100, stop_thresh: float = 1e-5):
    """Returns (distance, corr_x_to_y, corr_y_to_x, iterations)."""
    diff = x[:, None, :] - y[None, :, :]
    if p == 1:
        M = (diff**p).abs().sum(dim=2)  # sinkhorn.py:116
    else:
        M = (diff**p).sum(dim=2) ** (1.0 / p)  # sinkhorn.py:118
    if w_x is None and w_y is None:  # sinkhorn.py:121-124
        w_x = torch.ones(x.shape[0]).to(x) / x.shape[0]
        w_y = torch.ones(y.shape[0]).to(x) / y.shape[0]
        w_y *= w_x.shape[0] / w_y.shape[0]
    log_a, log_b = torch.log(w_x), torch.log(w_y)
    u = torch.zeros_like(w_x)
    v = eps * torch.log(w_y)
    iters = 0
    for _ in range(max_iters):  # sinkhorn.py:149-167
        u_prev, v_prev = u, v
        u = eps * (log_a - ((-M + v[None, :]) / eps).logsumexp(dim=1))
        v = eps * (log_b - ((-M + u[:, None]) / eps).logsumexp(dim=0))
        iters += 1
        if (u_prev - u).abs().max() < stop_thresh and (v_prev - v).abs().max() < stop_thresh:
            break
    P = ((-M + u[:, None] + v[None, :]) / eps).exp()
    return (P * M).sum(), P.argmax(dim=1), P.argmax(dim=0), iters


def metrics_reference_keys(samples: Tensor, weights: Tensor | None, *, expectations: dict, log_norm_const,
                           stddevs: Tensor | None, domain: Tensor | None, log_norm_const_preds: dict | None = None,
                           marginal_dims=()) -> dict:
    """get_metrics without sample losses / objective (eval/metrics.py:70-155) in plain torch."""
    fns = {"square": lambda x: (x**2).sum(-1, keepdim=True), "abs": lambda x: x.abs().sum(-1, keepdim=True),
           "sum": lambda x: x.sum(-1, keepdim=True), "square_minus_sum": lambda x: (x**2 - x).sum(-1, keepdim=True)}

    def with_errors(out, target):
        if target is not None:
            target = float(target)
            for k, pred in dict(out).items():
                suffix = k.replace("eval", "")
                err = abs(pred - target)
                out["error" + suffix] = err
                out["rel_error" + suffix] = err / (abs(target) + 1e-8)
        return out

    m = {}
    for name, fn in fns.items():
        pred = fn(samples)
        out = {f"eval/{name}": pred.mean().item()}
        if weights is not None:
            out[f"eval/{name}_is"] = ((pred * weights).sum() / weights.sum()).item()
        m.update(with_errors(out, expectations.get(name)))
    for name, pred in (log_norm_const_preds or {}).items():
        m.update(with_errors({f"eval/{name}": float(pred)}, log_norm_const))
    if weights is not None:
        ess = (weights.sum() ** 2 / (weights**2).sum()).item()
        m["eval/effective_sample_size"] = ess
        m["eval/norm_effective_sample_size"] = ess / len(weights)
    sd, mu = samples.std(dim=0), samples.mean(dim=0)
    m["eval/avg_stddev"] = sd.mean().item()
    for dim in marginal_dims:
        m[f"eval/stddev_{dim}"] = sd[dim].item()
        m[f"eval/avg_{dim}"] = mu[dim].item()
    if stddevs is not None:
        m["error/avg_marginal_stddev"] = (sd - stddevs).abs().mean().item()
        m.update(with_errors({"eval/avg_stddev": sd.mean().item()}, stddevs.mean()))
    if domain is not None:
        inside = (domain[:, 0] <= samples) & (samples <= domain[:, 1])
        m["eval/frac_pred_in_domain"] = inside.all(dim=-1).float().mean().item()
    return m

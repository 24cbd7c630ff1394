#!/usr/bin/env python3
"""Times the trajectory kernel on every BASELINE.json configuration at its per-GPU batch size and checks
size-independent properties there: determinism per (seed, call), finite outputs, estimator/row consistency."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sde_sampler_amd import problems

def flops(d, c, lh, k, analytic):
    return 4 * d * c + 2 * lh * c * c + (6 * d * k + 4 * k if k else 10 * d) + 20 * d

for name, spec in problems.BASELINE_SPECS.items():
    prob = problems.build(problems.baseline_spec(name), device="cuda:0")
    B, T, d = spec["batch"], prob.ts.numel() - 1, spec["target"]["dim"]
    torch.manual_seed(3)
    x0 = prob.prior.sample((B,))
    eng = prob.loss.engine
    eng.timing = True
    ms = []
    stepped = spec["target"]["kind"] == "nice"  # evaluated in one-step segments around the flow's score: time the whole evaluation
    for i in range(10 if stepped else 16):  # the first launches run while the GPU clock ramps up
        eng.calls = 100 + i
        if stepped:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
        r = prob.eval(x0, compute_weights=False)
        if stepped:
            e1.record()
            e1.synchronize()
            ms.append(e0.elapsed_time(e1))
        else:
            ms.append(eng.last_kernel_ms())
    eng.calls = 50
    a = prob.eval(x0, compute_weights=True)
    eng.calls = 50
    b = prob.eval(x0, compute_weights=True)
    assert torch.equal(a.samples, b.samples) and torch.equal(a.weights, b.weights), name
    assert torch.isfinite(a.samples).all(), name
    k = 40 if spec["target"]["kind"] == "gmm" else 0
    if spec["net"]["channels"] > 64:  # wide-network workloads: bench.py's count (network + Bridge divergence)
        import bench
        f = bench.algorithmic_flops(spec)
    else:
        f = flops(d, 64, 2, k, k == 0)
    best = min(ms[4:] if stepped else ms[8:])
    print(f"{name:22s} B={B:6d} T={T:4d} d={d:3d}  kernel {best:8.3f} ms  {B * T / best / 1e6:8.3f} G traj-steps/s  "
          f"{f * B * T / best / 1e9:7.1f} TFLOP/s (F={f})  logZ_is={a.log_norm_const_preds['log_norm_const_is']:+.4f} "
          f"lb={r.log_norm_const_preds['log_norm_const_lb']:+.4f}", flush=True)

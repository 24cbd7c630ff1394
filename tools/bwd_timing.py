#!/usr/bin/env python3
"""Kernel times of one training step's forward and backward launches (events around the kernels, sdeh_plan_set_timing) and the
backward kernel's achieved rate against the fp32 peak:  python tools/bwd_timing.py [spec] [method] [B ...]
Backward FLOPs per trajectory-step (algorithmic): the adjoint chain and the weight gradients are one MLP's worth each,
2 x (4 d C + 2 Lh C^2); the fused kernel also re-evaluates the network (a third)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from sde_sampler_amd import problems  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "cfg3_gmm50_pis_kl"
method = sys.argv[2] if len(sys.argv) > 2 else "kl"
batches = [int(b) for b in sys.argv[3:]] or [2048, 16384, 65536]
n_rep = int(os.environ.get("REPS", "5"))
for B in batches:
    spec = problems.baseline_spec(name)
    spec["batch"] = B
    spec["loss"]["method"] = method
    if method.startswith("lv"):
        spec["loss"]["max_rnd"] = 1e8
    prob = problems.build(spec, device="cuda:0")
    eng = prob.loss.engine
    eng.timing = True
    x0 = prob.prior.sample((B,))
    T = prob.ts.numel() - 1
    d, c, lh = spec["target"]["dim"], spec["net"]["channels"], spec["net"]["num_layers"] - 2
    f_fwd, f_bwd = [], []
    names = ("", "")
    for rep in range(n_rep + 2):
        prob.ctrl.zero_grad()
        val, _ = prob.loss(prob.ts, x0, prob.target.unnorm_log_prob, prob.second_log_prob)
        torch.cuda.synchronize()
        t_f, n_f = eng.last_kernel_ms(), eng.last_kernel_name()
        val.backward()
        torch.cuda.synchronize()
        t_b, n_b = eng.last_kernel_ms(), eng.last_kernel_name()
        if rep >= 2:
            f_fwd.append(t_f)
            f_bwd.append(t_b)
        names = (n_f, n_b)
    tf_, tb_ = sorted(f_fwd)[len(f_fwd) // 2], sorted(f_bwd)[len(f_bwd) // 2]
    flops_b = 2 * (4 * d * c + 2 * lh * c * c)
    rate = flops_b * B * T / (tb_ * 1e-3) / 1e12
    print(f"{name} {method} B={B:6d} T={T} d={d}: forward {names[0]} {tf_:8.3f} ms | backward {names[1]} {tb_:8.3f} ms = "
          f"{rate:5.1f} TFLOP/s algorithmic ({flops_b} FLOP/traj-step) = {rate / bench.PEAK_FP32_TFLOPS:.3f} of fp32 peak "
          f"({1.5 * rate / bench.PEAK_FP32_TFLOPS:.3f} with the re-evaluation)", flush=True)

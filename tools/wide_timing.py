#!/usr/bin/env python3
"""Kernel time of the wide-network trajectory kernels (sdeh_wide.hip) over batch sizes: python tools/wide_timing.py [workload] [B ...]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from sde_sampler_amd import problems  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "wide_pis_funnel196"
batches = [int(b) for b in sys.argv[2:]] or [4096, 8192, 16384, 32768, 65536]
for B in batches:
    spec = problems.baseline_spec(name)
    spec["batch"] = B
    if os.environ.get("EM_STEPS"):
        spec["grid"]["steps"] = int(os.environ["EM_STEPS"])
    prob = problems.build(spec, device="cuda:0")
    prob.loss.engine.timing = True
    x0 = prob.prior.sample((B,))
    ms = bench.timed_kernel_ms(prob, x0, n_warm=2, n=3)
    T = prob.ts.numel() - 1
    f = bench.algorithmic_flops(spec)
    tf = f * B * T / (ms * 1e-3) / 1e12
    print(f"{name} B={B:6d} T={T} kernel={prob.loss.engine.last_kernel_name()} {ms:9.3f} ms  {B * T / ms * 1e3:.3e} traj-steps/s  "
          f"{tf:6.1f} TFLOP/s algorithmic ({f} FLOP/traj-step) = {tf / bench.PEAK_FP32_TFLOPS:.3f} of fp32 peak", flush=True)

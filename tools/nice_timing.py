#!/usr/bin/env python3
"""Timing of the NICE flow's score evaluation (csrc/sdeh_nice.hip, sdeh_nice_eval) at BASELINE configs[4]'s per-GPU batch and around it:
ms per evaluation (HIP events on the launch stream, median of REPS), algorithmic TFLOP/s (bench.nice_flops: forward + reverse pass of every
coupling's MLP) against the fp32 matrix peak, and the share of a Bridge step it is.     python tools/nice_timing.py [batch ...]"""
import os
import statistics
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from sde_sampler_amd import problems  # noqa: E402

REPS = int(os.environ.get("REPS", "20"))
spec = dict(kind="nice", dim=196)
target = problems.build_target(spec).to("cuda:0")
flops = bench.nice_flops(spec)
for B in [int(a) for a in sys.argv[1:]] or [512, 4096, 8192, 32768]:
    x = torch.randn(B, 196, device="cuda:0")
    for want in ("score", "logp"):
        fn = (lambda: target.score(x)) if want == "score" else (lambda: target.unnorm_log_prob(x))
        for _ in range(3):
            fn()
        ms = []
        for _ in range(REPS):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            fn()
            e1.record()
            e1.synchronize()
            ms.append(e0.elapsed_time(e1))
        m = statistics.median(ms)
        f = flops if want == "score" else flops / 2
        print(f"nice {want:5s} B={B:6d}: {m:8.3f} ms  {f * B / (m * 1e-3) / 1e12:6.1f} TFLOP/s algorithmic ({f / 1e6:.1f} MFLOP per row) = "
              f"{f * B / (m * 1e-3) / 1e12 / bench.PEAK_FP32_TFLOPS:.3f} of the fp32 matrix peak")

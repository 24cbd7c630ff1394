#!/usr/bin/env python3
"""Evaluation kernel time per step around the batch sizes where the launcher switches modes: quad (SDEH_WS_QUAD=1: four M waves per group of 32,
one group per workgroup) against what the launcher would otherwise take (SDEH_WS_QUAD=0: pair mode up to 8192, groups of 32 beyond)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sde_sampler_amd import problems  # noqa: E402

for name in ("cfg1_dw_dis_lv", "cfg4_funnel_dds_lv", "cfg2_gmm2_dis_kl", "cfg3_gmm50_pis_kl", "gmm50_pis_headline"):
    for B in (512, 2048, 6000, 8192, 12288, 16384, 24576):
        line = f"{name:20s} B={B:6d}:"
        for quad in ("0", "1"):
            os.environ["SDEH_WS_QUAD"] = quad
            spec = problems.baseline_spec(name)
            spec["batch"] = B
            prob = problems.build(spec, device="cuda:0")
            x0 = prob.prior.sample((B,))
            prob.loss.engine.timing = True
            ms = []
            for i in range(8):
                prob.eval(x0, compute_weights=False, return_traj=False)
                ms.append(prob.loss.engine.last_kernel_ms())
            T = prob.ts.numel() - 1
            line += f"  quad={quad}: {min(ms[3:]):7.3f} ms ({min(ms[3:]) / T * 1e3:6.2f} us/step)"
        print(line, flush=True)
os.environ.pop("SDEH_WS_QUAD", None)

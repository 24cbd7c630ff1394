#!/usr/bin/env python3
"""Bridge evaluation and training step at small batches: one wave per 32-row tile (SDEH_BRIDGE_SPLIT=1) against the coordinate split
(four waves per tile sharing the d tangent passes, csrc/sdeh_bridge.hpp).  python tools/bridge_split_timing.py"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sde_sampler_amd import problems  # noqa: E402

NET = dict(channels=64, num_layers=4, activation="gelu")
T = 200
for name, tspec in [("gmm d=2", dict(kind="gmm", dim=2, name="fab")), ("funnel d=10", dict(kind="funnel", dim=10)),
                    ("multi-well d=5", dict(kind="multi_well", dim=5, n_double_wells=5, separation=2.0, shift=0.0)),
                    ("gauss d=50", dict(kind="iso_gauss", dim=50, loc=1.0, scale=0.5))]:
    d = tspec["dim"]
    for B in (512, 2048, 8192, 16384):
        if d * B > 50 * 8192:  # the divergence backward keeps 3 (Lh + 1) C planes per coordinate: 126 GB at d = 50, B = 16 384, T = 200
            continue
        line = f"bridge {name:15s} B={B:6d} T={T}:"
        for split in ("1", "4"):
            os.environ["SDEH_BRIDGE_SPLIT"] = split
            spec = dict(batch=B, target=tspec, prior=dict(kind="iso_gauss", dim=d), sde=dict(kind="scaled_bm", diff_coeff=1.0, terminal_t=1.0),
                        ctrl=dict(kind="lerp_target", clip_model=10.0, clip_score=10.0, scale_score=1.0, gamma_dim=1, gamma_bias=1.0),
                        inference_ctrl=dict(kind="lerp_prior", clip_model=10.0, clip_score=10.0, scale_score=1.0, gamma_dim=1, gamma_bias=1.0),
                        net=NET, loss=dict(kind="time_reversal", method="lv", max_rnd=1e8), grid=dict(start=0.0, end=1.0, steps=T))
            torch.manual_seed(3)
            prob = problems.build(spec, device="cuda:0")
            x0 = prob.prior.sample((B,))
            prob.loss.engine.timing = True
            ms = []
            for i in range(6):
                prob.eval(x0, compute_weights=False)
                ms.append(prob.loss.engine.last_kernel_ms())
            params = list(prob.ctrl.parameters()) + list(prob.loss.inference_ctrl.parameters())
            steps = []
            for i in range(5):
                for p in params:
                    p.grad = None
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                val, _ = prob.loss(prob.ts, x0, prob.target.unnorm_log_prob, prob.second_log_prob)
                val.backward()
                torch.cuda.synchronize()
                steps.append((time.perf_counter() - t0) * 1e3)
            line += f"  split {split}: eval kernel {min(ms[2:]):7.3f} ms ({min(ms[2:]) / T * 1e3:6.1f} us/step), loss+backward {min(steps[1:]):7.2f} ms |"
        print(line, flush=True)
os.environ.pop("SDEH_BRIDGE_SPLIT", None)

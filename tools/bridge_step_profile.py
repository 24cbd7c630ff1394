#!/usr/bin/env python3
"""One Bridge training step (loss + backward) at the reference's batch, for a per-kernel profile:
cd /tmp && rocprofv3 --kernel-trace --stats -d <out> -- python tools/bridge_step_profile.py [d] [B] [method]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sde_sampler_amd import problems  # noqa: E402

d = int(sys.argv[1]) if len(sys.argv) > 1 else 10
B = int(sys.argv[2]) if len(sys.argv) > 2 else 2048
method = sys.argv[3] if len(sys.argv) > 3 else "lv"
T = 200
tspec = dict(kind="funnel", dim=d) if d == 10 else dict(kind="iso_gauss", dim=d, loc=1.0, scale=0.5)
spec = dict(batch=B, target=tspec, prior=dict(kind="iso_gauss", dim=d), sde=dict(kind="scaled_bm", diff_coeff=1.0, terminal_t=1.0),
            ctrl=dict(kind="lerp_target", clip_model=10.0, clip_score=10.0, scale_score=1.0, gamma_dim=1, gamma_bias=1.0),
            inference_ctrl=dict(kind="lerp_prior", clip_model=10.0, clip_score=10.0, scale_score=1.0, gamma_dim=1, gamma_bias=1.0),
            net=dict(channels=64, num_layers=4, activation="gelu"), loss=dict(kind="time_reversal", method=method, max_rnd=1e8),
            grid=dict(start=0.0, end=1.0, steps=T))
torch.manual_seed(3)
prob = problems.build(spec, device="cuda:0")
x0 = prob.prior.sample((B,))
params = list(prob.ctrl.parameters()) + list(prob.loss.inference_ctrl.parameters())
steps = []
for i in range(6):
    for p in params:
        p.grad = None
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    val, _ = prob.loss(prob.ts, x0, prob.target.unnorm_log_prob, prob.second_log_prob)
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    val.backward()
    torch.cuda.synchronize()
    steps.append(((t1 - t0) * 1e3, (time.perf_counter() - t1) * 1e3))
print(f"bridge d={d} B={B} T={T} {method}: forward {min(s[0] for s in steps[1:]):.2f} ms, backward {min(s[1] for s in steps[1:]):.2f} ms", flush=True)

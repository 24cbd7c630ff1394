#!/usr/bin/env python3
"""Soak: many launches of each kernel family at a fixed Philox offset must give bitwise identical, finite results (a missed hazard
around a hand-written instruction shows up as a few stale rows in an occasional launch).  python tests/perf/soak_determinism.py [N]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from sde_sampler_amd import problems  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 50
CASES = [("gmm50_pis_headline", 65536), ("gmm50_pis_headline", 24576), ("cfg4_funnel_dds_lv", 32768), ("cfg2_gmm2_dis_kl", 6000),
         ("cfg1_dw_dis_lv", 1024), ("wide_pis_funnel196", 8192), ("cfg5_like_bridge196", 512)]
for name, B in CASES:
    spec = problems.baseline_spec(name)
    spec["batch"] = B
    if name.startswith("cfg5"):
        spec["grid"]["steps"] = 20
    prob = problems.build(spec, device="cuda:0")
    torch.manual_seed(1)
    x0 = prob.prior.sample((B,))
    eng = prob.loss.engine
    ref = None
    bad = 0
    for i in range(N):
        eng.calls = 7
        r = prob.eval(x0, compute_weights=True)
        cur = (r.samples.clone(), r.weights.clone())
        assert torch.isfinite(cur[0]).all(), (name, i)
        if ref is None:
            ref = cur
        elif not (torch.equal(ref[0], cur[0]) and torch.equal(ref[1], cur[1])):
            bad += 1
    print(f"{name:22s} B={B:6d}: {N} launches, {bad} differ from the first   kernel={eng.last_kernel_name()}", flush=True)
    assert bad == 0
# training: gradients of the fused backward
for name, B, method in [("cfg3_gmm50_pis_kl", 8192, "kl"), ("cfg1_dw_dis_lv", 8192, "lv")]:
    spec = problems.baseline_spec(name)
    spec["batch"] = B
    spec["loss"]["method"] = method
    prob = problems.build(spec, device="cuda:0")
    torch.manual_seed(1)
    x0 = prob.prior.sample((B,))
    eng = prob.loss.engine
    ref, bad = None, 0
    for i in range(max(4, N // 4)):
        eng.calls = 3
        prob.ctrl.zero_grad()
        val, _ = prob.loss(prob.ts, x0, prob.target.unnorm_log_prob, prob.second_log_prob)
        val.backward()
        cur = torch.cat([p.grad.flatten() for p in prob.ctrl.parameters() if p.grad is not None])
        assert torch.isfinite(cur).all(), (name, i)
        if ref is None:
            ref = cur.clone()
        elif not torch.equal(ref, cur):
            bad += 1
    print(f"{name:22s} B={B:6d} {method}: {max(4, N // 4)} training steps, {bad} gradient vectors differ   kernel={eng.last_kernel_name()}", flush=True)
    assert bad == 0
print("soak OK")

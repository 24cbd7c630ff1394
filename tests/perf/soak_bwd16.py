#!/usr/bin/env python3
"""Soak of the 16-trajectory fused backward (csrc/sdeh_bwdf16.hip): repeated training steps at a fixed Philox offset must give bitwise
identical gradients, also with other processes loading the GPU (run a few copies side by side).  python tests/perf/soak_bwd16.py [N]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from sde_sampler_amd import problems  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 100
for name, B, layers in [("cfg2_gmm2_dis_kl", 1024, 4), ("cfg2_gmm2_dis_kl", 2048, 4), ("cfg3_gmm50_pis_kl", 2048, 4), ("cfg2_gmm2_dis_kl", 1000, 3),
                        ("cfg2_gmm2_dis_kl", 700, 5), ("cfg2_gmm2_dis_kl", 12000, 4)]:
    spec = problems.baseline_spec(name)
    spec["batch"] = B
    spec["net"]["num_layers"] = layers
    spec["loss"]["method"] = "kl"
    prob = problems.build(spec, device="cuda:0")
    torch.manual_seed(1)
    x0 = prob.prior.sample((B,))
    eng = prob.loss.engine
    ref, bad, where = None, 0, set()
    for i in range(N):
        eng.calls = 3
        prob.ctrl.zero_grad()
        val, _ = prob.loss(prob.ts, x0, prob.target.unnorm_log_prob, prob.second_log_prob)
        val.backward()
        cur = {k: p.grad.clone() for k, p in prob.ctrl.named_parameters() if p.grad is not None}
        assert all(torch.isfinite(v).all() for v in cur.values()), (name, i)
        if ref is None:
            ref = cur
        else:
            diff = [k for k in cur if not torch.equal(ref[k], cur[k])]
            if diff:
                bad += 1
                where.update(diff)
    print(f"{name:20s} B={B:6d} layers={layers}: {N} steps, {bad} differ from the first {sorted(where)}  kernel={eng.last_kernel_name()}", flush=True)
print("soak done")

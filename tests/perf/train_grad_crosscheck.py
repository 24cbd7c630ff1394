#!/usr/bin/env python3
"""While training (eager Adam steps through the fused backward), every K steps: the parameter gradients of the fused kernel against
those of the plane-writing kernels on the same batch and Philox offset.  python tests/perf/train_grad_crosscheck.py [spec] [seed] [steps]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from sde_sampler_amd import problems  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "cfg1_dw_dis_lv"
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 17
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 300
spec = problems.baseline_spec(name)
spec["batch"] = 2048
prob = problems.build(spec, device="cuda:0")
torch.manual_seed(seed)
opt = torch.optim.Adam(prob.ctrl.parameters(), lr=5e-3)
eng = prob.loss.engine


def grads(x0, planes):
    if planes:
        os.environ["SDEH_BWD_PLANES"] = "1"
    try:
        prob.ctrl.zero_grad()
        val, _ = prob.loss(prob.ts, x0, prob.target.unnorm_log_prob, prob.second_log_prob)
        val.backward()
        return val.item(), {k: p.grad.detach().clone() for k, p in prob.ctrl.named_parameters() if p.grad is not None}
    finally:
        os.environ.pop("SDEH_BWD_PLANES", None)


for step in range(steps):
    x0 = prob.prior.sample((2048,))
    if step % 25 == 0:
        calls = eng.calls
        v1, g1 = grads(x0, False)
        eng.calls = calls
        v2, g2 = grads(x0, True)
        eng.calls = calls
        gmax = max(g.abs().max().item() for g in g2.values())
        worst = max(((g1[k] - g2[k]).abs().max().item() / max(g2[k].abs().max().item(), 1e-4 * gmax, 1e-30), k) for k in g2)
        print(f"step {step:4d}: loss {v1:.6g} (planes {v2:.6g})  |grad|max {gmax:.3e}  worst rel diff fused vs planes {worst[0]:.2e} ({worst[1]})", flush=True)
    opt.zero_grad()
    val, _ = prob.loss(prob.ts, x0, prob.target.unnorm_log_prob, prob.second_log_prob)
    val.backward()
    opt.step()

#!/usr/bin/env python3
"""Bridge training parity report: loss(...).backward() through the HIP kernels vs the reference's autograd gradients stored in
tests/golden/bridge_*.npz (exact divergence, create_graph=True), methods lv and kl, both networks."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from sde_sampler_amd import problems
from tests.helpers import GOLDEN_BRIDGE, inference_params, load_fixture

for path in GOLDEN_BRIDGE:
    fx, meta, params, tt = load_fixture(path)
    prob = problems.build(meta, params, tt, device="cuda:0", params_inf=inference_params(fx))
    x0, noise = torch.from_numpy(fx["x0"]).cuda(), torch.from_numpy(fx["noise"]).cuda()
    loss = prob.loss
    for method in ("lv", "kl"):
        loss.method, loss.max_rnd = method, (1e8 if method == "lv" else None)
        for p in list(prob.ctrl.parameters()) + list(loss.inference_ctrl.parameters()):
            p.grad = None
        val, _ = loss(prob.ts, x0, prob.target.unnorm_log_prob, prob.second_log_prob, noise=noise)
        val.backward()
        line = f"{os.path.basename(path)[:-4]:28s} {method}: loss {val.item():.6f} (ref {float(fx[f'train_{method}/loss']):.6f})"
        for prefix, mod in (("grad", prob.ctrl), ("grad_inf", loss.inference_ctrl)):
            rows = []
            for k, p in mod.named_parameters():
                key = f"train_{method}/{prefix}/{k}"
                if key not in fx.files:
                    continue
                gr = torch.from_numpy(fx[key])
                g = p.grad.cpu() if p.grad is not None else torch.zeros_like(gr)
                rows.append((g - gr).abs().max().item() / max(gr.abs().max().item(), 1e-30))
            line += f"   {prefix}: {len(rows)} tensors, worst rel {max(rows):.1e}"
        print(line, flush=True)

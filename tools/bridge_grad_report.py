import sys, os
sys.path.insert(0, os.getcwd())
import torch, numpy as np
from tests.helpers import GOLDEN_BRIDGE, inference_params, load_fixture
from sde_sampler_amd import problems
for path in GOLDEN_BRIDGE:
    fx, meta, params, tt = load_fixture(path)
    prob = problems.build(meta, params, tt, device="cuda:0", params_inf=inference_params(fx))
    x0, noise = torch.from_numpy(fx["x0"]).cuda(), torch.from_numpy(fx["noise"]).cuda()
    loss = prob.loss; loss.method, loss.max_rnd = "lv", 1e8
    val, _ = loss(prob.ts, x0, prob.target.unnorm_log_prob, prob.second_log_prob, noise=noise)
    val.backward()
    print(os.path.basename(path), "loss", val.item(), "ref", float(fx["train_lv/loss"]))
    for prefix, mod in (("grad", prob.ctrl), ("grad_inf", loss.inference_ctrl)):
        rows = []
        for k, p in mod.named_parameters():
            key = f"train_lv/{prefix}/{k}"
            if key not in fx.files: continue
            gr = torch.from_numpy(fx[key]); g = p.grad.cpu() if p.grad is not None else torch.zeros_like(gr)
            rows.append((k, (g - gr).abs().max().item() / max(gr.abs().max().item(), 1e-30), gr.abs().max().item()))
        print("  ", prefix, "n=%d" % len(rows), "worst rel %.2e" % max(r[1] for r in rows), "| per-param:", ", ".join(f"{k.split('.')[-2]}.{k.split('.')[-1]}:{e:.1e}" for k, e, m in rows[:30]))

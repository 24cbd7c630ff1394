"""Prints, per golden fixture and method, the relative error of the HIP training gradients vs the reference's."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from tests.helpers import GOLDEN, load_fixture, hip_problem
for path in GOLDEN:
    fx, meta, params, tt = load_fixture(path)
    for method in ("lv", "kl"):
        prob = hip_problem(meta, params, tt)
        prob.loss.method = method
        x0 = torch.from_numpy(fx["x0"]).cuda(); noise = torch.from_numpy(fx["noise"]).cuda()
        val, _ = prob.loss(prob.ts, x0, prob.target.unnorm_log_prob, prob.second_log_prob, noise=noise)
        val.backward()
        worst, gnorm = 0.0, 0.0
        for name, p in prob.ctrl.named_parameters():
            ref = fx[f"train_{method}/grad/{name}"]
            got = p.grad.cpu().numpy()
            worst = max(worst, np.abs(got - ref).max() / max(np.abs(ref).max(), 1e-12))
            gnorm += float((ref.astype(np.float64) ** 2).sum())
        print(f"{os.path.basename(path):32s} {method}: loss {val.item():+.5f} (ref {float(fx[f'train_{method}/loss']):+.5f})  |grad| {gnorm ** 0.5:.3e}  worst rel err {worst:.2e}")

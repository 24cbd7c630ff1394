#!/usr/bin/env python3
"""The CPU side of tools/train_reference_schedule.py: the oracle (oracle/em_oracle.py, bit-exact against the reference on the golden
fixtures) trains the same problem on the same schedule -- basic_pis / kl, Adam lr 1e-3, batch 512, T = 100 -- for as many steps as
the budget allows and reports the same quantities, so that the HIP run's quality can be read against the reference's own course
(the full 10 000 steps take ~3 h of CPU).    python tests/perf/train_reference_cpu.py [--steps 600] [--threads 8]"""
import argparse
import json
import math
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch

from oracle import em_oracle as eo
from sde_sampler_amd import problems


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=600)
    ap.add_argument("--batch", type=int, default=512)
    ap.add_argument("--threads", type=int, default=8)
    ap.add_argument("--eval-every", type=int, default=100)
    args = ap.parse_args()
    torch.set_num_threads(args.threads)
    spec = problems.baseline_spec("gmm50_pis_headline")
    spec["loss"]["method"] = "kl"
    prob = problems.build(spec)
    params = {k: v.detach().clone().requires_grad_(v.is_floating_point()) for k, v in prob.ctrl.state_dict().items()}
    tt = dict(loc=prob.target.loc.clone(), scale=prob.target.scale.clone(), mixture_weights=prob.target.mixture_weights.clone())
    opt = torch.optim.Adam([v for v in params.values() if v.requires_grad], lr=1e-3)
    ts = prob.ts.clone()
    T, d = ts.numel() - 1, spec["target"]["dim"]
    torch.manual_seed(0)

    def evaluate(B=6000):
        with torch.no_grad():
            x0 = prob.prior.sample((B,))
            out = eo.Problem(spec, {k: v.detach() for k, v in params.items()}, tt).eval(ts, x0, torch.randn(T, B, d), compute_weights=True)
        comp = torch.cdist(out["samples"], tt["loc"]).argmin(dim=1)
        share = torch.bincount(comp, minlength=tt["loc"].shape[0]).double() / B
        return dict(log_norm_const_is=out["log_norm_const_is"], elbo=out["log_norm_const_lb_ito"],
                    modes_covered=int((share >= 0.5 / tt["loc"].shape[0]).sum()), batch=B)

    print(json.dumps(dict(step=0, **evaluate())), flush=True)
    t0 = time.perf_counter()
    for step in range(1, args.steps + 1):
        x0 = prob.prior.sample((args.batch,))
        noise = torch.randn(T, args.batch, d)
        opt.zero_grad(set_to_none=True)
        loss, _, _, _ = eo.Problem(spec, params, tt).train_loss(ts, x0, noise, method="kl")
        loss.backward()
        opt.step()
        if step % args.eval_every == 0:
            print(json.dumps(dict(step=step, loss=loss.item(), s_per_step=(time.perf_counter() - t0) / step, **evaluate())), flush=True)


if __name__ == "__main__":
    main()

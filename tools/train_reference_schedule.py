#!/usr/bin/env python3
"""End-to-end quality of the HIP training path on the metric's target (BASELINE.json: "|log-Z err|, GMM-40 d=50"; VERDICT r03 weak 4 /
next 6): trains `basic_pis` (ScoreCtrl + FourierMLP C = 64, Delta prior, ScaledBM; loss reference_sde, method kl) on the reference's
own schedule -- conf/solver/basic_oc_base.yaml: Adam lr 1e-3, 10 000 steps, batch 512, T = 100, no clipping / EMA / scheduler --
through utils.graphs.GraphedTrainStep (one hipGraph launch per optimisation step), evaluates every 500 steps at the reference's
evaluation batch (6000) and at the end at B = 65 536, and stores the control:

    python tools/train_reference_schedule.py [--steps 10000] [--batch 512] [--out tests/golden/trained_pis_gmm50_ref_schedule.pt]

Reported: log Z_is (true value 0), the ELBO, ESS / B and the number of mixture components that receive at least half of their
share 1/40 of the samples (PIS on this target is known to collapse onto few modes: log Z_is -> log(covered / 40))."""
import argparse
import json
import math
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from sde_sampler_amd import engine as E
from sde_sampler_amd import problems
from sde_sampler_amd.utils.graphs import GraphedTrainStep


def evaluate(prob, B):
    x = prob.prior.sample((B,))
    with torch.no_grad():
        x_T, rnd, _ = prob.loss.simulate(prob.ts, x, prob.target.unnorm_log_prob, prob.second_log_prob, compute_ito_int=True)
    est = E.estimators_from_stats(E.merge_stats(E.estimator_stats(rnd)))
    # nearest component (the padded mixture's means differ in the leading coordinates only; all scales are equal)
    loc = prob.target.loc
    comp = torch.cdist(x_T, loc).argmin(dim=1)
    share = torch.bincount(comp, minlength=loc.shape[0]).double() / B
    covered = int((share >= 0.5 / loc.shape[0]).sum())
    w = torch.exp(-rnd.double().flatten() - est["log_weight_max"])
    se = float(w.std() / w.mean() / math.sqrt(B))
    return dict(log_norm_const_is=est["log_norm_const_is"], se=se, elbo=est["mean_neg_rnd"], ess_frac=est["ess"] / B,
                modes_covered=covered, n_modes=int(loc.shape[0]), batch=B)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=10000)
    ap.add_argument("--batch", type=int, default=512)
    ap.add_argument("--lr", type=float, default=1e-3)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--out", default=os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden",
                                                  "trained_pis_gmm50_ref_schedule.pt"))
    args = ap.parse_args()
    spec = problems.baseline_spec("gmm50_pis_headline")
    spec["loss"]["method"] = "kl"
    prob = problems.build(spec, device="cuda:0")
    torch.manual_seed(args.seed)
    params = list(prob.ctrl.parameters())
    opt = torch.optim.Adam(params, lr=args.lr, capturable=True, fused=True)

    def loss_fn():
        x = prob.prior.sample((args.batch,))
        loss, _ = prob.loss(prob.ts, x, prob.target.unnorm_log_prob, prob.second_log_prob)
        return loss

    log = [dict(step=0, **evaluate(prob, 6000))]
    print(json.dumps(log[-1]), flush=True)
    step_fn = GraphedTrainStep(loss_fn, [prob.loss], opt, warmup=3, guard=True)
    done = 3
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    while done < args.steps:
        n = min(500 - done % 500, args.steps - done)
        for _ in range(n):
            loss = step_fn()
        done += n
        torch.cuda.synchronize()
        ms = 1e3 * (time.perf_counter() - t0) / (done - 3)
        log.append(dict(step=done, loss=float(loss), ms_per_step=ms, skipped=int(step_fn.n_skipped), **evaluate(prob, 6000)))
        print(json.dumps(log[-1]), flush=True)
        t0 += 0.0  # (evaluation time is inside the average: it is what a run of the reference's loop pays as well)
    final = evaluate(prob, 65536)
    note = (f"tools/train_reference_schedule.py: basic_pis / kl, Adam lr={args.lr}, {args.steps} steps of batch {args.batch}, T=100, "
            f"seed {args.seed}, GraphedTrainStep; final (B=65536, in-kernel noise): log Z_is={final['log_norm_const_is']:+.4f} "
            f"(se {final['se']:.4f}), ELBO={final['elbo']:+.4f}, ESS/B={final['ess_frac']:.4f}, "
            f"modes covered {final['modes_covered']}/{final['n_modes']}")
    os.makedirs(os.path.dirname(os.path.abspath(args.out)), exist_ok=True)
    torch.save({"params": {k: v.detach().cpu().clone() for k, v in prob.ctrl.state_dict().items()}, "note": note,
                "spec": "gmm50_pis_headline", "final": final, "log": log}, args.out)
    print("saved", args.out)
    print(note)


if __name__ == "__main__":
    main()

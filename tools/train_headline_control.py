#!/usr/bin/env python3
"""Trains the control of the metric's headline workload (GMM-40 d=50, basic_pis: ScoreCtrl + FourierMLP C=64, Delta prior,
ScaledBM) with the HIP training path (forward trajectory kernel + sdeh_ctrl_backward + Adam), until the importance weights of an
evaluation batch have an effective sample size above a threshold, and stores the parameters as a fixture:

    python tools/train_headline_control.py [--method lv] [--max-steps 6000] [--ess 0.05] [--out tests/golden/trained_pis_gmm50.pt]

bench.py's `log_z` block and tests/test_hip_logz.py evaluate log Z with this control (a random-init control has degenerate weights:
log_norm_const_is of the untrained network moves by +-0.8 between runs)."""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from sde_sampler_amd import engine as E
from sde_sampler_amd import problems


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--method", default="lv")
    ap.add_argument("--max-steps", type=int, default=6000)
    ap.add_argument("--batch", type=int, default=4096)
    ap.add_argument("--lr", type=float, default=2e-3)
    ap.add_argument("--ess", type=float, default=0.05)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--out", default=os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden",
                                                  "trained_pis_gmm50.pt"))
    args = ap.parse_args()
    spec = problems.baseline_spec("gmm50_pis_headline")
    spec["loss"]["method"] = args.method
    if args.method in ("lv", "lv_traj"):
        spec["loss"]["max_rnd"] = 1e8
    prob = problems.build(spec, device="cuda:0")
    torch.manual_seed(args.seed)
    params = list(prob.ctrl.parameters())
    opt = torch.optim.Adam(params, lr=args.lr)
    sched = torch.optim.lr_scheduler.StepLR(opt, step_size=2000, gamma=0.5)

    def evaluate():
        B = 65536
        x = prob.prior.sample((B,))
        with torch.no_grad():
            _, rnd, _ = prob.loss.simulate(prob.ts, x, prob.target.unnorm_log_prob, prob.second_log_prob, compute_ito_int=True)
        est = E.estimators_from_stats(E.merge_stats(E.estimator_stats(rnd)))
        return est["log_norm_const_is"], est["mean_neg_rnd"], est["ess"] / B

    lz, lb, ess = evaluate()
    print(f"[init] log Z_is = {lz:+.4f}  ELBO = {lb:+.4f}  ESS/B = {ess:.4f}", flush=True)
    t0, step = time.perf_counter(), 0
    while step < args.max_steps:
        x = prob.prior.sample((args.batch,))
        loss, _ = prob.loss(prob.ts, x, prob.target.unnorm_log_prob, prob.second_log_prob)
        opt.zero_grad(set_to_none=True)
        loss.backward()
        torch.nn.utils.clip_grad_norm_(params, 1.0)
        opt.step()
        sched.step()
        step += 1
        if step % 250 == 0:
            lz, lb, ess = evaluate()
            print(f"step {step}: loss {loss.item():.4f}  log Z_is = {lz:+.4f}  ELBO = {lb:+.4f}  ESS/B = {ess:.4f}  "
                  f"({1e3 * (time.perf_counter() - t0) / step:.2f} ms/step)", flush=True)
            if ess > args.ess and step >= 1000:
                break
    lz, lb, ess = evaluate()
    note = (f"tools/train_headline_control.py: method={args.method} batch={args.batch} lr={args.lr} steps={step} seed={args.seed}; "
            f"at the end log Z_is={lz:+.4f} ELBO={lb:+.4f} ESS/B={ess:.4f} (B=65536, in-kernel noise)")
    os.makedirs(os.path.dirname(os.path.abspath(args.out)), exist_ok=True)
    torch.save({"params": {k: v.detach().cpu().clone() for k, v in prob.ctrl.state_dict().items()}, "note": note,
                "spec": "gmm50_pis_headline"}, args.out)
    print("saved", args.out, "\n" + note)


if __name__ == "__main__":
    main()

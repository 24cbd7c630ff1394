#!/usr/bin/env python3
"""End-to-end training demonstration on the GPU: the loop of the reference's `Trainable.step` (solver/base.py:399-454:
sample prior, loss(...), backward, Adam step) with this package's loss classes -- forward AND backward run in the HIP
kernels.  Usage: python tools/train_demo.py [cfg1_dw_dis_lv|cfg2_gmm2_dis_kl|...] [--steps N] [--method lv|kl]"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from sde_sampler_amd import problems


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("name", nargs="?", default="cfg1_dw_dis_lv")
    ap.add_argument("--steps", type=int, default=300)
    ap.add_argument("--batch", type=int, default=2048)
    ap.add_argument("--method", default=None)
    ap.add_argument("--lr", type=float, default=5e-3)
    args = ap.parse_args()
    spec = problems.baseline_spec(args.name)
    if args.method:
        spec["loss"]["method"] = args.method
    prob = problems.build(spec, device="cuda:0")
    torch.manual_seed(0)
    if hasattr(prob.target, "compute_stats"):
        prob.target.compute_stats()
    true_logz = prob.target.log_norm_const
    opt = torch.optim.Adam(prob.ctrl.parameters(), lr=args.lr)

    def evaluate(tag):
        x = prob.prior.sample((16384,))
        r = prob.eval(x, compute_weights=True)
        lz = r.log_norm_const_preds["log_norm_const_is"]
        lb = r.log_norm_const_preds["log_norm_const_lb_ito"]
        err = abs(lz - true_logz) if true_logz is not None else float("nan")
        print(f"[{tag}] log Z_is = {lz:+.4f}  ELBO = {lb:+.4f}  true log Z = {true_logz}  |err| = {err:.4f}  "
              f"lv_loss = {r.metrics['eval/lv_loss']:.4f}", flush=True)
        return err

    e0 = evaluate("init")
    t0 = time.perf_counter()
    for step in range(args.steps):
        x = prob.prior.sample((args.batch,))
        loss, _ = prob.loss(prob.ts, x, prob.target.unnorm_log_prob, prob.second_log_prob)
        opt.zero_grad(set_to_none=True)
        loss.backward()
        torch.nn.utils.clip_grad_norm_(prob.ctrl.parameters(), 1.0)
        opt.step()
        if (step + 1) % 100 == 0:
            torch.cuda.synchronize()
            print(f"step {step + 1}: loss {loss.item():.4f}  ({1e3 * (time.perf_counter() - t0) / (step + 1):.2f} ms/step)", flush=True)
    e1 = evaluate("trained")
    print(f"RESULT method={spec['loss']['method']} steps={args.steps} err_init={e0:.4f} err_trained={e1:.4f}")


if __name__ == "__main__":
    main()

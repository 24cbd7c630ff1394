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
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--eval-batch", type=int, default=16384, help="trajectories of the log Z evaluations before / after training")
    ap.add_argument("--graph", action="store_true", help="capture the whole optimisation step into one hipGraph (utils/graphs.py)")
    ap.add_argument("--no-guard", action="store_true", help="with --graph: no device-side skip of non-finite updates")
    ap.add_argument("--em-steps", type=int, default=None, help="Euler-Maruyama steps of the grid (default: the specification's)")
    ap.add_argument("--channels", type=int, default=None, help="network width (default: the specification's)")
    ap.add_argument("--print-every", type=int, default=100)
    args = ap.parse_args()
    if args.name == "bridge_dw":  # a Bridge (conf/solver/bridge.yaml style) on the shifted double well
        lerp = dict(clip_model=10.0, clip_score=10.0, scale_score=1.0, gamma_dim=1, gamma_bias=1.0)  # conf/solver/bridge.yaml
        spec = dict(batch=args.batch, target=dict(kind="double_well", dim=1, separation=2.0, shift=1.5),
                    prior=dict(kind="iso_gauss", dim=1), sde=dict(kind="scaled_bm", diff_coeff=2.0, terminal_t=1.0),
                    ctrl=dict(kind="lerp_target", **lerp), inference_ctrl=dict(kind="lerp_prior", **lerp),
                    net=dict(channels=64, num_layers=4, activation="gelu"),
                    loss=dict(kind="time_reversal", method="lv", max_rnd=1e8), grid=dict(start=0.0, end=1.0, steps=100))
    else:
        spec = problems.baseline_spec(args.name)
    if args.method:
        spec["loss"]["method"] = args.method
    if args.em_steps:
        spec["grid"]["steps"] = args.em_steps
    if args.channels:
        spec["net"]["channels"] = args.channels
    prob = problems.build(spec, device="cuda:0")
    torch.manual_seed(args.seed)
    if hasattr(prob.target, "compute_stats") and spec["target"]["kind"] != "nice":  # (a flow: normalised by construction, log Z = 0)
        prob.target.compute_stats()
    true_logz = prob.target.log_norm_const
    train_params = list(prob.ctrl.parameters())
    groups = [dict(params=train_params, lr=args.lr)]
    if getattr(prob.loss, "inference_ctrl", None) is not None:  # conf/solver/bridge.yaml: inference_ctrl lr = 0.02 x lr
        inf_params = list(prob.loss.inference_ctrl.parameters())
        groups.append(dict(params=inf_params, lr=0.02 * args.lr))
        train_params = train_params + inf_params
    opt = torch.optim.Adam(groups, capturable=args.graph, fused=args.graph)  # (fused: one launch; the capturable foreach form runs ~90 per-tensor kernels)

    def evaluate(tag):
        x = prob.prior.sample((args.eval_batch,))
        r = prob.eval(x, compute_weights=True)
        lz = r.log_norm_const_preds["log_norm_const_is"]
        lb = r.log_norm_const_preds["log_norm_const_lb_ito"]
        err = abs(lz - true_logz) if true_logz is not None else float("nan")
        print(f"[{tag}] log Z_is = {lz:+.4f}  ELBO = {lb:+.4f}  true log Z = {true_logz}  |err| = {err:.4f}  "
              f"lv_loss = {r.metrics['eval/lv_loss']:.4f}", flush=True)
        return err

    e0 = evaluate("init")
    graphed = None
    if args.graph:
        from sde_sampler_amd.utils.graphs import GraphedTrainStep

        def loss_fn():
            x = prob.prior.sample((args.batch,))
            return prob.loss(prob.ts, x, prob.target.unnorm_log_prob, prob.second_log_prob)[0]

        graphed = GraphedTrainStep(loss_fn, [prob.loss], opt, warmup=3, guard=not args.no_guard,
                                   after_backward=lambda: torch.nn.utils.clip_grad_norm_(train_params, 1.0))
    t0 = t_last = time.perf_counter()
    for step in range(args.steps):
        if graphed is not None:
            loss = graphed()
        else:
            x = prob.prior.sample((args.batch,))
            loss, _ = prob.loss(prob.ts, x, prob.target.unnorm_log_prob, prob.second_log_prob)
            opt.zero_grad(set_to_none=True)
            loss.backward()
            torch.nn.utils.clip_grad_norm_(train_params, 1.0)
            opt.step()
        if (step + 1) % args.print_every == 0:
            torch.cuda.synchronize()
            now = time.perf_counter()
            print(f"step {step + 1}: loss {loss.item():.4f}  ({1e3 * (now - t_last) / args.print_every:.2f} ms/step over the last {args.print_every} steps, "
                  f"{1e3 * (now - t0) / (step + 1):.2f} since the start)", flush=True)
            t_last = now
    if graphed is not None:
        print(f"[graph] {graphed.replays} replays, {int(graphed.n_skipped)} updates skipped on the device (non-finite loss / gradient)")
    e1 = evaluate("trained")
    print(f"RESULT method={spec['loss']['method']} steps={args.steps} err_init={e0:.4f} err_trained={e1:.4f}")


if __name__ == "__main__":
    main()

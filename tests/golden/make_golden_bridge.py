#!/usr/bin/env python3
"""Golden vectors for the Bridge branch of TimeReversalLoss (`bridge_*.npz`): generative control + inference control whose
exact divergence enters the cost (reference losses/oc.py:189-202 with utils/autograd.py:81-105; wiring of
solver/oc.py:127-153).  Produced by RUNNING THE REFERENCE; build container only (see make_golden.py for the conventions).
Stored: both networks' parameters, ts, x0, per-step noise, the two evaluation passes, and -- for later rounds -- the train
losses and parameter gradients of both networks (method kl and lv, exact divergence with create_graph=True)."""
from __future__ import annotations

import json
import sys
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, str(Path(__file__).resolve().parent))
import make_golden as mg  # noqa: E402

from sde_sampler.losses.oc import TimeReversalLoss  # noqa: E402

OUT = Path(__file__).resolve().parent
NET = dict(channels=64, num_layers=4, activation="gelu")
ISO = lambda d: dict(kind="iso_gauss", dim=d, loc=0.0, scale=1.0)

CASES = {
    # conf/solver/basic_bridge.yaml on the "fab" mixture: LerpTargetCtrl / LerpPriorCtrl, ScaledBM(1), kl
    "bridge_gmm2_kl": dict(
        B=64, seed=61, target=dict(kind="gmm", dim=2, name="fab"), prior=ISO(2),
        sde=dict(kind="scaled_bm", diff_coeff=1.0, terminal_t=1.0),
        ctrl=dict(kind="lerp_target", clip_model=1e4, clip_score=1e4, scale_score=1.0, gamma_dim=1, gamma_bias=1.0),
        inference_ctrl=dict(kind="lerp_prior", clip_model=1e4, clip_score=1e4, scale_score=1.0, gamma_dim=1, gamma_bias=1.0),
        net=NET, loss=dict(kind="time_reversal", method="kl", max_rnd=None),
        grid=dict(start=0.0, end=1.0, steps=40, rescale_t=None)),
    # conf/solver/bridge.yaml style: clips 10 (active), lv, VP, five double wells, per-coordinate gamma for the inference control
    "bridge_mw5_lv": dict(
        B=64, seed=67, target=dict(kind="multi_well", dim=5, n_double_wells=5, separation=4.0, shift=0.0), prior=ISO(5),
        sde=dict(kind="vp", beta_min=0.1, beta_max=6.0, scale=1.0, terminal_t=1.0),
        ctrl=dict(kind="lerp_target", clip_model=10.0, clip_score=10.0, scale_score=1.0, gamma_dim=1, gamma_bias=1.0),
        inference_ctrl=dict(kind="lerp_prior", clip_model=0.05, clip_score=1.5, scale_score=0.7, gamma_dim=5, gamma_bias=1.0),
        net=NET, inference_net=dict(channels=64, num_layers=3, activation="silu"),
        loss=dict(kind="time_reversal", method="lv", max_rnd=1e8),
        grid=dict(start=0.0, end=1.0, steps=30, rescale_t=None)),
    # a plain ClippedCtrl as inference control (no score term), LerpCtrl generative, funnel d=10
    "bridge_funnel10_clipped_kl": dict(
        B=32, seed=71, target=dict(kind="funnel", dim=10), prior=ISO(10),
        sde=dict(kind="const_ou", drift_coeff=0.5, diff_coeff=1.2, terminal_t=1.0),
        ctrl=dict(kind="lerp", clip_model=1e4, clip_score=1e4, scale_score=1.0, gamma_dim=1, gamma_bias=1.0),
        inference_ctrl=dict(kind="clipped", clip_model=1e4),
        net=NET, loss=dict(kind="time_reversal", method="kl", max_rnd=None),
        grid=dict(start=0.0, end=1.0, steps=20, rescale_t=None)),
}


def run_case(name, case):
    torch.manual_seed(1)
    target = mg.build_target(case["target"])
    prior = mg.build_prior(case["prior"])
    sde = mg.build_sde(case["sde"])
    dim = case["target"]["dim"]
    ctrl = mg.build_ctrl(case["ctrl"], case["net"], dim, sde, prior, target)
    inf = mg.build_ctrl(case["inference_ctrl"], case.get("inference_net", case["net"]), dim, sde, prior, target)
    lspec = case["loss"]
    loss = TimeReversalLoss(generative_ctrl=ctrl, sde=sde, method=lspec["method"], max_rnd=lspec["max_rnd"],
                            filter_samples=getattr(target, "filter", None), inference_ctrl=inf)
    g = case["grid"]
    ts = mg.get_timesteps(g["start"], g["end"], steps=g["steps"], rescale_t=g["rescale_t"])
    x0, noise, state = mg.draw_inputs(case, prior, ts)
    terminal, second = target.unnorm_log_prob, prior.log_prob
    out = {}
    for k, v in ctrl.state_dict().items():
        out["param/" + k] = v.detach().numpy().copy()
    for k, v in inf.state_dict().items():
        out["param_inf/" + k] = v.detach().numpy().copy()
    out.update(ts=ts.numpy(), x0=x0.numpy(), noise=noise.numpy())
    with torch.no_grad():
        torch.set_rng_state(state)
        xT, rnd, xs = loss.simulate(ts, x0, terminal, second, train=False, compute_ito_int=True, return_traj=True)
        out["eval1/x_T"], out["eval1/rnd"] = xT.numpy(), rnd.numpy()
        torch.set_rng_state(state)
        res = loss.eval(ts, x0, terminal, second, compute_weights=True, return_traj=False)
        out["eval1/weights"] = res.weights.numpy()
        for k in ("log_norm_const_lb_ito", "log_norm_const_is"):
            out["eval1/" + k] = np.float64(res.log_norm_const_preds[k])
        out["eval1/lv_loss"] = np.float64(res.metrics["eval/lv_loss"])
        torch.set_rng_state(state)
        xT2, rnd2, _ = loss.simulate(ts, x0, terminal, second, train=False, compute_ito_int=False, return_traj=False)
        out["eval2/x_T"], out["eval2/rnd"] = xT2.numpy(), rnd2.numpy()
        torch.set_rng_state(state)
        res2 = loss.eval(ts, x0, terminal, second, compute_weights=False, return_traj=False)
        out["eval2/log_norm_const_lb"] = np.float64(res2.log_norm_const_preds["log_norm_const_lb"])
    for method in ["kl", "lv"]:
        loss.method, loss.n_filtered = method, 0
        ctrl.zero_grad(); inf.zero_grad()
        torch.set_rng_state(state)
        val, metrics = loss(ts, x0, terminal, second)
        val.backward()
        out[f"train_{method}/loss"] = np.float64(val.item())
        for prefix, mod in (("grad", ctrl), ("grad_inf", inf)):
            for k, p in mod.named_parameters():
                out[f"train_{method}/{prefix}/{k}"] = (p.grad.detach().numpy().copy() if p.grad is not None
                                                       else np.zeros(tuple(p.shape), np.float32))
    # Hutchinson divergence estimators (TimeReversalLoss.div_estimator; they only act in training): per step the reference
    # draws the probe (compute_divx) BEFORE the Brownian increment, so both sequences are pre-drawn in that order
    for est in ("rademacher", "gauss"):
        loss.div_estimator, loss.method, loss.n_filtered = est, "lv", 0
        torch.manual_seed(case["seed"] + 7)
        st = torch.get_rng_state()
        probes, incs = [], []
        for _ in range(len(ts) - 1):
            probes.append(torch.randint_like(x0, low=0, high=2).float() * 2 - 1.0 if est == "rademacher" else torch.randn_like(x0))
            incs.append(torch.randn_like(x0))
        out[f"hutch_{est}/probes"], out[f"hutch_{est}/noise"] = torch.stack(probes).numpy(), torch.stack(incs).numpy()
        ctrl.zero_grad(); inf.zero_grad()
        torch.set_rng_state(st)
        val, _ = loss(ts, x0, terminal, second)
        val.backward()
        out[f"hutch_{est}/loss"] = np.float64(val.item())
        for prefix, mod in (("grad", ctrl), ("grad_inf", inf)):
            for k, p in mod.named_parameters():
                out[f"hutch_{est}/{prefix}/{k}"] = (p.grad.detach().numpy().copy() if p.grad is not None
                                                    else np.zeros(tuple(p.shape), np.float32))
    loss.div_estimator, loss.method = None, lspec["method"]

    torch.manual_seed(99)
    xq = x0 + 0.5 * torch.randn_like(x0)
    out["kat/x"] = xq.numpy()
    out["kat/target_unnorm_log_prob"] = target.unnorm_log_prob(xq).detach().numpy()
    out["kat/target_score"] = target.score(xq.clone()).detach().numpy()
    out["kat/second_log_prob"] = second(xq).detach().numpy()
    out["kat/prior_score"] = prior.score(xq).detach().numpy()
    if case["target"]["kind"] == "gmm":
        out["target/loc"], out["target/scale"] = target.loc.numpy(), target.scale.numpy()
        out["target/mixture_weights"] = target.mixture_weights.numpy()
    out["meta"] = np.frombuffer(json.dumps(dict(case, name=name)).encode(), dtype=np.uint8)
    path = OUT / f"{name}.npz"
    np.savez_compressed(path, **out)
    print(f"{name:28s} B={case['B']:3d} T={len(ts)-1:3d} d={dim:3d} logZ_is={out['eval1/log_norm_const_is']:+.5f} "
          f"lb={out['eval2/log_norm_const_lb']:+.5f} train_kl={out['train_kl/loss']:+.5f} {path.stat().st_size/1024:.1f} KB")


if __name__ == "__main__":
    torch.set_num_threads(1)
    for name, case in CASES.items():
        run_case(name, case)

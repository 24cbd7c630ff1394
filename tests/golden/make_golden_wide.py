#!/usr/bin/env python3
"""Golden vectors for the WIDE-network kernels (`wide_*.npz`): FourierMLP with 128 / 256 channels and state dimensions up to
196 -- the shape of BASELINE.json configs[4] -- through the reference's three simulate() loops and its Bridge branch
(losses/oc.py:156-230 incl. 189-202, 286-343, 400-457; models/mlp.py:85-122 with `channels` a free constructor argument).
Produced by RUNNING THE REFERENCE; build container only (see make_golden.py for the conventions): parameters, ts, x0, per-step
noise -> x_T, rnd, estimators of both eval passes, and (round 3) the training losses and PARAMETER GRADIENTS of the reference's
autograd for methods kl and lv (`loss(...)` + `.backward()`, losses/oc.py:232-256; Bridges: both networks, exact divergence with
create_graph=True, utils/autograd.py:14-22) -- the pins of the wide training backward (csrc/sdeh_wide_bwd.hip)."""
from __future__ import annotations

import json
import math
import sys
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, str(Path(__file__).resolve().parent))
import make_golden as mg  # noqa: E402

from sde_sampler.losses.oc import TimeReversalLoss  # noqa: E402

OUT = Path(__file__).resolve().parent
#: gradient tensors above this size are stored as a strided sample of their flattened entries (+ their Euclidean norm): the two
#: methods x two networks of a C = 256 Bridge would otherwise be 15 MB of incompressible floats per fixture
GRAD_FULL_MAX, GRAD_STRIDE = 16384, 5
ISO = lambda d, **kw: dict(kind="iso_gauss", dim=d, loc=0.0, scale=1.0, **kw)

CASES = {
    # basic_pis shape on a funnel: ScoreCtrl, Delta prior, ScaledBM; C = 128; ragged batch (48 = 32 + 16)
    "wide_pis_funnel100_c128": dict(
        B=48, seed=31, target=dict(kind="funnel", dim=100), prior=dict(kind="delta", dim=100),
        sde=dict(kind="scaled_bm", diff_coeff=math.sqrt(0.2), terminal_t=5.0),
        ctrl=dict(kind="score", clip_model=1e4, clip_score=1e4, scale_score=1.0, gamma_dim=1, gamma_bias=0.01),
        net=dict(channels=128, num_layers=4, activation="gelu"),
        loss=dict(kind="reference_sde", method="kl", max_rnd=None),
        grid=dict(start=0.0, end=5.0, steps=12, rescale_t=None)),
    # basic_dis shape at the cfg5 geometry: LerpCtrl, VP, Gaussian target, d = 196, C = 256
    "wide_dis_gauss196_c256": dict(
        B=40, seed=37, target=dict(kind="iso_gauss", dim=196, loc=1.5, scale=0.8), prior=ISO(196),
        sde=dict(kind="vp", beta_min=0.1, beta_max=6.0, scale=1.0, terminal_t=1.0),
        ctrl=dict(kind="lerp", clip_model=1e4, clip_score=1e4, scale_score=1.0, gamma_dim=1, gamma_bias=1.0),
        net=dict(channels=256, num_layers=4, activation="gelu"),
        loss=dict(kind="time_reversal", method="kl", max_rnd=None),
        grid=dict(start=0.0, end=1.0, steps=10, rescale_t=None)),
    # dds shape: exponential integrator, lv, truncated prior, cosine grid, ACTIVE clips, per-coordinate gamma, SiLU, 3 layers,
    # double wells + Gaussian coordinates, d = 72 (not a multiple of 8 or 32)
    "wide_dds_mw70_c128": dict(
        B=33, seed=41, target=dict(kind="multi_well", dim=70, n_double_wells=5, separation=2.0, shift=0.5),
        prior=ISO(70, truncate_quartile=1e-4), sde=None,
        ctrl=dict(kind="score", clip_model=0.08, clip_score=2.0, scale_score=0.7, gamma_dim=70, gamma_bias=0.3),
        net=dict(channels=128, num_layers=3, activation="silu"),
        loss=dict(kind="exponential", method="lv", max_rnd=1e8, alpha=1.0, sigma=1.0),
        grid=dict(start=0.0, end=6.4, steps=14, rescale_t="cosine")),
    # EulerDDS shape (reference control = sigma * prior score), ReLU, 5 layers, funnel d = 40 with C = 256
    "wide_eulerdds_funnel40_c256": dict(
        B=32, seed=43, target=dict(kind="funnel", dim=40), prior=ISO(40),
        sde=dict(kind="vp", beta_min=0.1, beta_max=8.0, scale=1.0, terminal_t=1.0),
        ctrl=dict(kind="score", clip_model=1e4, clip_score=1e4, scale_score=1.0, gamma_dim=1, gamma_bias=0.01),
        net=dict(channels=256, num_layers=5, activation="relu"),
        loss=dict(kind="reference_sde", method="kl", max_rnd=None, reference_ctrl="prior_score"),
        grid=dict(start=0.0, end=1.0, steps=10, rescale_t=None)),
    # mixture targets in the wide kernels: general scales / weights (7 components, d = 100, C = 128) ...
    "wide_pis_gmm100_c128": dict(
        B=40, seed=61, target=dict(kind="gmm", dim=100, name="random7"), prior=dict(kind="delta", dim=100),
        sde=dict(kind="scaled_bm", diff_coeff=math.sqrt(0.2), terminal_t=5.0),
        ctrl=dict(kind="score", clip_model=1e4, clip_score=1e4, scale_score=1.0, gamma_dim=1, gamma_bias=0.01),
        net=dict(channels=128, num_layers=4, activation="gelu"),
        loss=dict(kind="reference_sde", method="kl", max_rnd=None),
        grid=dict(start=0.0, end=5.0, steps=12, rescale_t=None)),
    # ... and the reference's own padded 40-mode mixture at d = 72 with the DEFAULT 64 channels (d > 64: two of the four waves hold
    # the hidden layers' row tiles), basic_dis shape
    "wide_dis_gmm72_c64": dict(
        B=36, seed=67, target=dict(kind="gmm", dim=72, name="fab50"), prior=ISO(72),
        sde=dict(kind="vp", beta_min=0.1, beta_max=8.0, scale=1.0, terminal_t=1.0),
        ctrl=dict(kind="lerp", clip_model=1e4, clip_score=1e4, scale_score=1.0, gamma_dim=1, gamma_bias=1.0),
        net=dict(channels=64, num_layers=4, activation="gelu"),
        loss=dict(kind="time_reversal", method="kl", max_rnd=None),
        grid=dict(start=0.0, end=1.0, steps=10, rescale_t=None)),
    # the default 64 channels on a funnel in d = 90 (exponential integrator, lv, clips)
    "wide_dds_funnel90_c64": dict(
        B=32, seed=71, target=dict(kind="funnel", dim=90), prior=ISO(90, truncate_quartile=1e-4), sde=None,
        ctrl=dict(kind="score", clip_model=10.0, clip_score=10.0, scale_score=1.0, gamma_dim=1, gamma_bias=0.01),
        net=dict(channels=64, num_layers=4, activation="gelu"),
        loss=dict(kind="exponential", method="lv", max_rnd=1e8, alpha=1.0, sigma=1.0),
        grid=dict(start=0.0, end=6.4, steps=12, rescale_t="cosine")),
}

BRIDGE_CASES = {
    # conf/solver/basic_bridge.yaml at the cfg5 geometry: LerpTargetCtrl / LerpPriorCtrl, ScaledBM(1, T=1), d = 196, C = 256, with a
    # funnel target in place of the NICE flow (distr/nice.py needs torchvision + data/nice.pt)
    "widebridge_funnel196_c256": dict(
        B=16, seed=47, target=dict(kind="funnel", dim=196), prior=ISO(196),
        sde=dict(kind="scaled_bm", diff_coeff=1.0, terminal_t=1.0),
        ctrl=dict(kind="lerp_target", clip_model=10.0, clip_score=10.0, scale_score=1.0, gamma_dim=1, gamma_bias=1.0),  # bridge.yaml clips
        inference_ctrl=dict(kind="lerp_prior", clip_model=10.0, clip_score=10.0, scale_score=1.0, gamma_dim=1, gamma_bias=1.0),
        net=dict(channels=256, num_layers=4, activation="gelu"),
        loss=dict(kind="time_reversal", method="kl", max_rnd=None),
        grid=dict(start=0.0, end=1.0, steps=6, rescale_t=None)),
    # conf/solver/bridge.yaml style: active clips (10 / tight model clip), VP, per-coordinate gamma, a 3-layer SiLU inference
    # network, ClippedCtrl-free; d = 44, C = 128
    "widebridge_mw44_c128": dict(
        B=24, seed=53, target=dict(kind="multi_well", dim=44, n_double_wells=4, separation=2.0, shift=0.0), prior=ISO(44),
        sde=dict(kind="vp", beta_min=0.1, beta_max=6.0, scale=1.0, terminal_t=1.0),
        ctrl=dict(kind="lerp_target", clip_model=10.0, clip_score=10.0, scale_score=1.0, gamma_dim=1, gamma_bias=1.0),
        inference_ctrl=dict(kind="lerp_prior", clip_model=0.05, clip_score=1.5, scale_score=0.7, gamma_dim=44, gamma_bias=1.0),
        net=dict(channels=128, num_layers=4, activation="gelu"), inference_net=dict(channels=128, num_layers=3, activation="silu"),
        loss=dict(kind="time_reversal", method="lv", max_rnd=1e8),
        grid=dict(start=0.0, end=1.0, steps=8, rescale_t=None)),
    # ClippedCtrl inference control (no score term), LerpCtrl generative, ConstOU, Gaussian target, d = 33, C = 128, 2 hidden
    "widebridge_gauss33_clipped_c128": dict(
        B=20, seed=59, target=dict(kind="iso_gauss", dim=33, loc=-0.7, scale=1.3), prior=ISO(33),
        sde=dict(kind="const_ou", drift_coeff=0.5, diff_coeff=1.2, terminal_t=1.0),
        ctrl=dict(kind="lerp", clip_model=1e4, clip_score=1e4, scale_score=1.0, gamma_dim=1, gamma_bias=1.0),
        inference_ctrl=dict(kind="clipped", clip_model=1e4),
        net=dict(channels=128, num_layers=4, activation="gelu"),
        loss=dict(kind="time_reversal", method="kl", max_rnd=None),
        grid=dict(start=0.0, end=1.0, steps=8, rescale_t=None)),
    # Bridges on MIXTURE targets (distr/gauss.py:66-140): the benchmark's 40-mode mixture, d = 50, with 128-channel networks
    # (basic_bridge.yaml's controls, VP, lv) ...
    "widebridge_gmm50_c128": dict(
        B=24, seed=73, target=dict(kind="gmm", dim=50, name="fab50"), prior=ISO(50),
        sde=dict(kind="vp", beta_min=0.1, beta_max=6.0, scale=1.0, terminal_t=1.0),
        ctrl=dict(kind="lerp_target", clip_model=10.0, clip_score=10.0, scale_score=1.0, gamma_dim=1, gamma_bias=1.0),
        inference_ctrl=dict(kind="lerp_prior", clip_model=10.0, clip_score=10.0, scale_score=1.0, gamma_dim=1, gamma_bias=1.0),
        net=dict(channels=128, num_layers=4, activation="gelu"),
        loss=dict(kind="time_reversal", method="lv", max_rnd=1e8),
        grid=dict(start=0.0, end=1.0, steps=8, rescale_t=None)),
    # ... and general scales / weights (7 components) at d = 72, C = 256: LerpCtrl generative (prior and target score), ClippedCtrl
    # inference control on a 3-layer network, ScaledBM, kl
    "widebridge_gmm72_c256": dict(
        B=20, seed=79, target=dict(kind="gmm", dim=72, name="random7"), prior=ISO(72),
        sde=dict(kind="scaled_bm", diff_coeff=1.0, terminal_t=1.0),
        ctrl=dict(kind="lerp", clip_model=1e4, clip_score=1e4, scale_score=1.0, gamma_dim=1, gamma_bias=1.0),
        inference_ctrl=dict(kind="clipped", clip_model=1e4),
        net=dict(channels=256, num_layers=4, activation="gelu"), inference_net=dict(channels=256, num_layers=3, activation="silu"),
        loss=dict(kind="time_reversal", method="kl", max_rnd=None),
        grid=dict(start=0.0, end=1.0, steps=6, rescale_t=None)),
}


def _eval_passes(out, loss, ts, x0, state, terminal, second, train_kw):
    with torch.no_grad():
        torch.set_rng_state(state)
        xT, rnd, _ = loss.simulate(ts, x0, terminal, second, compute_ito_int=True, return_traj=False, **train_kw)
        out["eval1/x_T"], out["eval1/rnd"] = xT.numpy(), rnd.numpy()
        torch.set_rng_state(state)
        res = loss.eval(ts, x0, terminal, second, compute_weights=True, return_traj=False)
        assert torch.equal(res.samples, xT)
        out["eval1/weights"] = res.weights.numpy()
        for k in ("log_norm_const_lb_ito", "log_norm_const_is"):
            out["eval1/" + k] = np.float64(res.log_norm_const_preds[k])
        out["eval1/lv_loss"] = np.float64(res.metrics["eval/lv_loss"])
        torch.set_rng_state(state)
        xT2, rnd2, _ = loss.simulate(ts, x0, terminal, second, compute_ito_int=False, return_traj=False, **train_kw)
        out["eval2/x_T"], out["eval2/rnd"] = xT2.numpy(), rnd2.numpy()
        torch.set_rng_state(state)
        res2 = loss.eval(ts, x0, terminal, second, compute_weights=False, return_traj=False)
        out["eval2/log_norm_const_lb"] = np.float64(res2.log_norm_const_preds["log_norm_const_lb"])


def _train_passes(out, loss, mods, ts, x0, state, terminal, second):
    """train_{kl,lv}/loss and the gradients of every parameter of `mods` = (("grad", generative), ("grad_inf", inference)...)."""
    keep = loss.method
    for method in ("kl", "lv"):
        loss.method, loss.n_filtered = method, 0
        for _, mod in mods:
            mod.zero_grad()
        torch.set_rng_state(state)
        val, metrics = loss(ts, x0, terminal, second)
        val.backward()
        out[f"train_{method}/loss"] = np.float64(val.item())
        out[f"train_{method}/n_filtered"] = np.int64(metrics["train/n_filtered_cumulative"])
        for prefix, mod in mods:
            for k, p in mod.named_parameters():
                g = p.grad.detach().numpy().copy() if p.grad is not None else np.zeros(tuple(p.shape), np.float32)
                key = f"train_{method}/{prefix}/{k}"
                if g.size > GRAD_FULL_MAX:  # the [256, 256] / [256, 512] / [196, 256] matrices: every GRAD_STRIDE-th entry + the norm
                    out[key + "@stride"] = g.reshape(-1)[::GRAD_STRIDE].copy()
                    out[key + "@norm"] = np.float64(np.linalg.norm(g.astype(np.float64)))
                else:
                    out[key] = g
    loss.method, loss.n_filtered = keep, 0


def run_case(name, case):
    target, prior, sde, ctrl, loss, second, ts = mg.reference_problem(case)
    x0, noise, state = mg.draw_inputs(case, prior, ts)
    out = {"param/" + k: v.detach().numpy().copy() for k, v in ctrl.state_dict().items()}
    out.update(ts=ts.numpy(), x0=x0.numpy(), noise=noise.numpy())
    train_kw = {"train": False} if case["loss"]["kind"] == "time_reversal" else {}
    _eval_passes(out, loss, ts, x0, state, target.unnorm_log_prob, second, train_kw)
    _train_passes(out, loss, (("grad", ctrl),), ts, x0, state, target.unnorm_log_prob, second)
    if case["target"]["kind"] == "gmm":
        out["target/loc"], out["target/scale"] = target.loc.numpy(), target.scale.numpy()
        out["target/mixture_weights"] = target.mixture_weights.numpy()
    _finish(name, case, out, ts, x0)


def run_bridge_case(name, case):
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
    out = {"param/" + k: v.detach().numpy().copy() for k, v in ctrl.state_dict().items()}
    out.update({"param_inf/" + k: v.detach().numpy().copy() for k, v in inf.state_dict().items()})
    out.update(ts=ts.numpy(), x0=x0.numpy(), noise=noise.numpy())
    _eval_passes(out, loss, ts, x0, state, target.unnorm_log_prob, prior.log_prob, {"train": False})
    _train_passes(out, loss, (("grad", ctrl), ("grad_inf", inf)), ts, x0, state, target.unnorm_log_prob, prior.log_prob)
    if case["target"]["kind"] == "gmm":
        out["target/loc"], out["target/scale"] = target.loc.numpy(), target.scale.numpy()
        out["target/mixture_weights"] = target.mixture_weights.numpy()
    _finish(name, case, out, ts, x0)


def _finish(name, case, out, ts, x0):
    out["meta"] = np.frombuffer(json.dumps(dict(case, name=name)).encode(), dtype=np.uint8)
    path = OUT / f"{name}.npz"
    np.savez_compressed(path, **out)
    print(f"{name:34s} B={case['B']:3d} T={len(ts)-1:3d} d={x0.shape[1]:3d} logZ_is={out['eval1/log_norm_const_is']:+.5f} "
          f"lb={out['eval2/log_norm_const_lb']:+.5f} train_kl={out['train_kl/loss']:+.5f} train_lv={out['train_lv/loss']:+.5f} "
          f"{path.stat().st_size/1024:.1f} KB")


if __name__ == "__main__":
    torch.set_num_threads(4)
    only = set(sys.argv[1:])
    for name, case in CASES.items():
        if not only or name in only:
            run_case(name, case)
    for name, case in BRIDGE_CASES.items():
        if not only or name in only:
            run_bridge_case(name, case)

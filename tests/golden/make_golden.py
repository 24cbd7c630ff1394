#!/usr/bin/env python3
"""Generate the golden input/output vectors under tests/golden/ by RUNNING THE REFERENCE.

This script only works in the build container, where the reference checkout is mounted
read-only at /root/reference.  It never travels to the GPU box; only the ``*.npz`` files
it writes do.  Nothing of the reference's source is copied: the fixtures hold inputs
(network weights, time grid, x0, per-step Gaussian noise) and the reference's outputs
(x_T, rnd, estimators, loss values, parameter gradients).

Reference entry points exercised (all under /root/reference/sde_sampler):
  losses/oc.py:156-230   TimeReversalLoss.simulate      (DIS)
  losses/oc.py:286-343   ReferenceSDELoss.simulate      (PIS, EulerDDS)
  losses/oc.py:400-457   ExponentialIntegratorSDELoss.simulate (DDS)
  losses/oc.py:72-123    BaseOCLoss.compute_loss / compute_results
  models/mlp.py:43-122   TimeEmbed / FourierMLP
  models/reparam.py      ClippedCtrl / ScoreCtrl / LerpCtrl / LerpTargetCtrl / LerpPriorCtrl
  eq/sdes.py             VP / ConstOU / ScaledBM
  distr/*.py             GMM / IsotropicGauss / Gauss / Delta / DoubleWell / MultiWell / Funnel
  utils/common.py:18-55  get_timesteps

Noise replay recipe: seed, draw x0 from the prior, snapshot the RNG
state, pre-draw T tensors with randn_like, restore the RNG state and run the reference.
The reference then consumes exactly the pre-drawn noise (bit-exact replay).

Usage:  python tests/golden/make_golden.py            (writes tests/golden/*.npz)
"""
from __future__ import annotations

import json
import math
import sys
import types
from functools import partial
from pathlib import Path

import numpy as np
import torch

REFERENCE = Path("/root/reference")
OUT = Path(__file__).resolve().parent


def _import_reference():
    if not REFERENCE.exists():
        raise SystemExit("reference checkout not present; fixtures can only be generated in the build container")
    stubs = {
        "wandb": {"run": None, "log": lambda *a, **k: None},
        "torchquad": {"Boole": object},
        "torchsde": {"BaseBrownian": object},
    }
    for name, attrs in stubs.items():
        mod = types.ModuleType(name)
        for k, v in attrs.items():
            setattr(mod, k, v)
        sys.modules[name] = mod
    sys.path.insert(0, str(REFERENCE))


_import_reference()

from sde_sampler.distr.delta import Delta  # noqa: E402
from sde_sampler.distr.double_well import DoubleWell, MultiWell  # noqa: E402
from sde_sampler.distr.funnel import Funnel  # noqa: E402
from sde_sampler.distr.gauss import GMM, Gauss, IsotropicGauss  # noqa: E402
from sde_sampler.eq.sdes import VP, ConstOU, ScaledBM  # noqa: E402
from sde_sampler.losses.oc import (  # noqa: E402
    ExponentialIntegratorSDELoss,
    ReferenceSDELoss,
    TimeReversalLoss,
)
from sde_sampler.models.mlp import FourierMLP, TimeEmbed  # noqa: E402
from sde_sampler.models.reparam import (  # noqa: E402
    ClippedCtrl,
    LerpCtrl,
    LerpPriorCtrl,
    LerpTargetCtrl,
    ScoreCtrl,
)
from sde_sampler.utils.common import get_timesteps  # noqa: E402

ACTS = {"gelu": torch.nn.GELU, "silu": torch.nn.SiLU, "relu": torch.nn.ReLU}


# ----------------------------------------------------------------------------------------
# case table.  Every entry is plain data (also stored, as JSON, inside the fixture) so that the
# oracle and the HIP engine can rebuild the same problem without the reference.
# ----------------------------------------------------------------------------------------
def fab_loc(dim: int) -> torch.Tensor:
    gen = torch.Generator()
    gen.manual_seed(42)
    loc2 = (torch.rand((40, 2), generator=gen) - 0.5) * 2 * 40
    if dim == 2:
        return loc2
    return torch.cat([loc2, torch.zeros(40, dim - 2)], dim=1)


CASES = {
    # cfg1: conf/target/dw_shift.yaml + conf/solver/basic_dis.yaml with loss.method=lv
    "cfg1_dw_dis_lv": dict(
        B=128, seed=7,
        target=dict(kind="double_well", dim=1, separation=2.0, shift=1.5),
        prior=dict(kind="iso_gauss", dim=1, loc=0.0, scale=1.0),
        sde=dict(kind="vp", beta_min=0.1, beta_max=10.0, scale=1.0, terminal_t=1.0),
        ctrl=dict(kind="lerp", clip_model=1e4, clip_score=1e4, scale_score=1.0, gamma_dim=1, gamma_bias=1.0),
        net=dict(channels=64, num_layers=4, activation="gelu"),
        loss=dict(kind="time_reversal", method="lv", max_rnd=1e8),
        grid=dict(start=0.0, end=1.0, steps=100, rescale_t=None),
    ),
    # cfg2: GMM "fab" (40 modes, d=2) + basic_dis, loss.method=kl
    "cfg2_gmm2_dis_kl": dict(
        B=128, seed=7,
        target=dict(kind="gmm", dim=2, name="fab"),
        prior=dict(kind="iso_gauss", dim=2, loc=0.0, scale=1.0),
        sde=dict(kind="vp", beta_min=0.1, beta_max=10.0, scale=1.0, terminal_t=1.0),
        ctrl=dict(kind="lerp", clip_model=1e4, clip_score=1e4, scale_score=1.0, gamma_dim=1, gamma_bias=1.0),
        net=dict(channels=64, num_layers=4, activation="gelu"),
        loss=dict(kind="time_reversal", method="kl", max_rnd=None),
        grid=dict(start=0.0, end=1.0, steps=100, rescale_t=None),
    ),
    # cfg3 / north-star: GMM-40 in d=50 (explicit loc/scale) + basic_pis (ScoreCtrl, Delta prior, ScaledBM)
    "cfg3_gmm50_pis_kl": dict(
        B=32, seed=7,
        target=dict(kind="gmm", dim=50, name="fab50"),
        prior=dict(kind="delta", dim=50),
        sde=dict(kind="scaled_bm", diff_coeff=math.sqrt(0.2), terminal_t=5.0),
        ctrl=dict(kind="score", clip_model=1e4, clip_score=1e4, scale_score=1.0, gamma_dim=1, gamma_bias=0.01),
        net=dict(channels=64, num_layers=4, activation="gelu"),
        loss=dict(kind="reference_sde", method="kl", max_rnd=None),
        grid=dict(start=0.0, end=5.0, steps=40, rescale_t=None),
    ),
    # cfg4: Funnel d=10 + dds (exponential integrator, lv, truncated prior, cosine grid, clips 10)
    "cfg4_funnel_dds_lv": dict(
        B=64, seed=7,
        target=dict(kind="funnel", dim=10),
        prior=dict(kind="iso_gauss", dim=10, loc=0.0, scale=1.0, truncate_quartile=1e-4),
        sde=None,
        ctrl=dict(kind="score", clip_model=10.0, clip_score=10.0, scale_score=1.0, gamma_dim=1, gamma_bias=0.01),
        net=dict(channels=64, num_layers=4, activation="gelu"),
        loss=dict(kind="exponential", method="lv", max_rnd=1e8, alpha=1.0, sigma=1.0),
        grid=dict(start=0.0, end=12.8, steps=60, rescale_t="cosine"),
    ),
    # basic_dds_euler: ReferenceSDELoss with reference_ctrl = sigma * prior.score, VP, Gauss prior
    "eulerdds_funnel_kl": dict(
        B=64, seed=11,
        target=dict(kind="funnel", dim=10),
        prior=dict(kind="iso_gauss", dim=10, loc=0.0, scale=1.0),
        sde=dict(kind="vp", beta_min=0.1, beta_max=10.0, scale=1.0, terminal_t=1.0),
        ctrl=dict(kind="score", clip_model=1e4, clip_score=1e4, scale_score=1.0, gamma_dim=1, gamma_bias=0.01),
        net=dict(channels=64, num_layers=4, activation="gelu"),
        loss=dict(kind="reference_sde", method="kl", max_rnd=None, reference_ctrl="prior_score"),
        grid=dict(start=0.0, end=1.0, steps=50, rescale_t=None),
    ),
    # mw target (5 double wells, sep 4) + dis-like with ConstOU, lerp_dim (gamma of size d), SiLU, 5 layers
    "mw5_dis_constou_lv": dict(
        B=64, seed=13,
        target=dict(kind="multi_well", dim=5, n_double_wells=5, separation=4.0, shift=0.0),
        prior=dict(kind="iso_gauss", dim=5, loc=0.0, scale=1.0),
        sde=dict(kind="const_ou", drift_coeff=1.0, diff_coeff=1.5, terminal_t=1.0),
        ctrl=dict(kind="lerp", clip_model=10.0, clip_score=10.0, scale_score=1.0, gamma_dim=5, gamma_bias=1.0),
        net=dict(channels=64, num_layers=5, activation="silu"),
        loss=dict(kind="time_reversal", method="lv", max_rnd=1e8),
        grid=dict(start=0.0, end=1.0, steps=50, rescale_t=None),
    ),
    # mixed wells + gaussian coordinates, ClippedCtrl (dis_no_score), ReLU, 3 layers
    "mw8_clipped_vp_kl": dict(
        B=64, seed=17,
        target=dict(kind="multi_well", dim=8, n_double_wells=3, separation=2.0, shift=0.5),
        prior=dict(kind="iso_gauss", dim=8, loc=0.0, scale=1.0),
        sde=dict(kind="vp", beta_min=0.1, beta_max=4.0, scale=1.0, terminal_t=1.0),
        ctrl=dict(kind="clipped", clip_model=10.0),
        net=dict(channels=64, num_layers=3, activation="relu"),
        loss=dict(kind="time_reversal", method="kl", max_rnd=None),
        grid=dict(start=0.0, end=1.0, steps=50, rescale_t=None),
    ),
    # Gauss target (gauss_shift-like, d=3 diag) + LerpTargetCtrl / pis-like BM, general GMM (non-uniform scale/weights)
    "gmmgen_lerptarget_bm_kl": dict(
        B=64, seed=19,
        target=dict(kind="gmm", dim=3, name="random7"),
        prior=dict(kind="iso_gauss", dim=3, loc=0.0, scale=1.0),
        sde=dict(kind="scaled_bm", diff_coeff=1.0, terminal_t=1.0),
        ctrl=dict(kind="lerp_target", clip_model=1e4, clip_score=1e4, scale_score=1.0, gamma_dim=1, gamma_bias=1.0),
        net=dict(channels=64, num_layers=4, activation="gelu"),
        loss=dict(kind="time_reversal", method="kl", max_rnd=None),
        grid=dict(start=0.0, end=1.0, steps=50, rescale_t=None),
    ),
    "gauss_lerpprior_vp_lv": dict(
        B=64, seed=23,
        target=dict(kind="iso_gauss", dim=4, loc=3.0, scale=1.0),
        prior=dict(kind="iso_gauss", dim=4, loc=0.0, scale=1.0),
        sde=dict(kind="vp", beta_min=0.1, beta_max=10.0, scale=1.0, terminal_t=1.0),
        ctrl=dict(kind="lerp_prior", clip_model=1e4, clip_score=1e4, scale_score=1.0, gamma_dim=1, gamma_bias=1.0),
        net=dict(channels=64, num_layers=4, activation="gelu"),
        loss=dict(kind="time_reversal", method="lv", max_rnd=1e8),
        grid=dict(start=0.0, end=1.0, steps=50, rescale_t=None),
    ),
}


# models/reparam.py:58,134: detach_score=True (the constructor default; every shipped YAML sets False) detaches x in front of the
# score terms -- the kl gradients then carry no score Jacobian
CASES["funnel_lerp_detach_kl"] = dict(
    B=32, seed=29,
    target=dict(kind="funnel", dim=10),
    prior=dict(kind="iso_gauss", dim=10, loc=0.0, scale=1.0),
    sde=dict(kind="vp", beta_min=0.1, beta_max=6.0, scale=1.0, terminal_t=1.0),
    ctrl=dict(kind="lerp", clip_model=1e4, clip_score=1e4, scale_score=1.0, gamma_dim=1, gamma_bias=1.0, detach_score=True),
    net=dict(channels=64, num_layers=4, activation="gelu"),
    loss=dict(kind="time_reversal", method="kl", max_rnd=None),
    grid=dict(start=0.0, end=1.0, steps=20, rescale_t=None),
)

EXTRA_METHODS = {"cfg1_dw_dis_lv", "cfg4_funnel_dds_lv", "eulerdds_funnel_kl"}


def random_gmm(dim: int, k: int, seed: int):
    gen = torch.Generator()
    gen.manual_seed(seed)
    loc = (torch.rand((k, dim), generator=gen) - 0.5) * 8.0
    scale = 0.4 + torch.rand((k, dim), generator=gen)
    w = 0.2 + torch.rand((k,), generator=gen)
    return loc, scale, w


def build_target(spec):
    kind = spec["kind"]
    if kind == "double_well":
        return DoubleWell(dim=1, separation=spec["separation"], shift=spec["shift"])
    if kind == "multi_well":
        return MultiWell(dim=spec["dim"], n_double_wells=spec["n_double_wells"],
                         separation=spec["separation"], shift=spec["shift"])
    if kind == "funnel":
        return Funnel(dim=spec["dim"], n_reference_samples=1000)
    if kind == "iso_gauss":
        return IsotropicGauss(dim=spec["dim"], loc=spec["loc"], scale=spec["scale"], n_reference_samples=1000)
    if kind == "gmm":
        name = spec["name"]
        if name == "fab":
            return GMM(dim=2, name="fab", n_reference_samples=1000)
        if name == "fab50":
            d = spec["dim"]
            loc = fab_loc(d)
            scale = torch.nn.functional.softplus(torch.tensor(1.0)) * torch.ones(40, d)
            return GMM(dim=d, loc=loc, scale=scale, mixture_weights=torch.ones(40),
                       n_reference_samples=1000, domain_tol=None)
        if name == "random7":
            loc, scale, w = random_gmm(spec["dim"], 7, 1234)
            return GMM(dim=spec["dim"], loc=loc, scale=scale, mixture_weights=w,
                       n_reference_samples=1000, domain_tol=None)
    raise ValueError(kind)


def build_prior(spec):
    if spec["kind"] == "delta":
        return Delta(dim=spec["dim"])
    return IsotropicGauss(dim=spec["dim"], loc=spec["loc"], scale=spec["scale"],
                          truncate_quartile=spec.get("truncate_quartile"))


def build_sde(spec):
    if spec is None:
        return None
    if spec["kind"] == "vp":
        return VP(diff_coeff_sq_min=spec["beta_min"], diff_coeff_sq_max=spec["beta_max"],
                  scale_diff_coeff=spec["scale"], terminal_t=spec["terminal_t"])
    if spec["kind"] == "const_ou":
        return ConstOU(drift_coeff=spec["drift_coeff"], diff_coeff=spec["diff_coeff"], terminal_t=spec["terminal_t"])
    if spec["kind"] == "scaled_bm":
        return ScaledBM(diff_coeff=spec["diff_coeff"], terminal_t=spec["terminal_t"])
    raise ValueError(spec)


def build_ctrl(spec, net, dim, sde, prior, target):
    act = ACTS[net["activation"]]()
    zeros_ = torch.nn.init.zeros_
    base = FourierMLP(dim=dim, activation=act, num_layers=net["num_layers"], channels=net["channels"],
                      last_bias_init=zeros_, last_weight_init=zeros_)
    kind = spec["kind"]
    if kind == "clipped":
        ctrl = ClippedCtrl(base_model=base, clip_model=spec["clip_model"])
    else:
        score_model = TimeEmbed(dim_out=spec["gamma_dim"], activation=act, num_layers=4, channels=net["channels"],
                                last_bias_init=partial(torch.nn.init.constant_, val=spec["gamma_bias"]),
                                last_weight_init=zeros_)
        kw = dict(base_model=base, score_model=score_model, target_score=target.score, detach_score=spec.get("detach_score", False),
                  clip_score=spec["clip_score"], clip_model=spec["clip_model"], scale_score=spec["scale_score"])
        if kind == "score":
            ctrl = ScoreCtrl(**kw)
        else:
            cls = {"lerp": LerpCtrl, "lerp_target": LerpTargetCtrl, "lerp_prior": LerpPriorCtrl}[kind]
            ctrl = cls(**kw, sde=sde, prior_score=prior.score)
    # the shipped init zeroes the last layers, which would make the MLP numerically dead:
    # draw them from N(0, 0.05^2) so the fixtures exercise the whole network
    with torch.no_grad():
        for mod in [ctrl.base_model.out_layer] + ([ctrl.score_model.out_layer] if kind != "clipped" else []):
            mod.weight.normal_(0.0, 0.05)
            mod.bias.add_(torch.randn_like(mod.bias) * 0.05)
    return ctrl


def reference_problem(case):
    torch.manual_seed(1)  # conf/base.yaml:8
    target = build_target(case["target"])
    prior = build_prior(case["prior"])
    sde = build_sde(case["sde"])
    dim = case["target"]["dim"]
    ctrl = build_ctrl(case["ctrl"], case["net"], dim, sde, prior, target)
    lspec = case["loss"]
    common = dict(generative_ctrl=ctrl, sde=sde, method=lspec["method"], max_rnd=lspec["max_rnd"],
                  filter_samples=getattr(target, "filter", None))
    if lspec["kind"] == "time_reversal":
        loss = TimeReversalLoss(**common)
        second = prior.log_prob  # initial_log_prob (solver/oc.py:155-163)
    elif lspec["kind"] == "reference_sde":
        ref_ctrl = None
        if lspec.get("reference_ctrl") == "prior_score":
            ref_ctrl = lambda t, x: sde.diff(t, x) * prior.score(x)  # solver/oc.py:305-306
            reference = sde.marginal_distr(sde.terminal_t, x_init=prior.loc, var_init=prior.scale**2)
        else:
            reference = sde.marginal_distr(t=sde.terminal_t, x_init=prior.loc)  # solver/oc.py:189-191
        loss = ReferenceSDELoss(**common, reference_ctrl=ref_ctrl)
        second = reference.log_prob
    else:
        loss = ExponentialIntegratorSDELoss(**common, alpha=lspec["alpha"], sigma=lspec["sigma"])
        second = prior.log_prob  # solver/oc.py:243
    g = case["grid"]
    ts = get_timesteps(g["start"], g["end"], steps=g["steps"], rescale_t=g["rescale_t"])
    return target, prior, sde, ctrl, loss, second, ts


def draw_inputs(case, prior, ts):
    torch.manual_seed(case["seed"])
    x0 = prior.sample((case["B"],))
    state = torch.get_rng_state()
    noise = torch.stack([torch.randn_like(x0) for _ in range(len(ts) - 1)])
    return x0, noise, state


def run_case(name, case):
    target, prior, sde, ctrl, loss, second, ts = reference_problem(case)
    x0, noise, state = draw_inputs(case, prior, ts)
    out = {}
    meta = dict(case)
    meta["name"] = name

    # parameters
    for k, v in ctrl.state_dict().items():
        out["param/" + k] = v.detach().numpy().copy()
    out["ts"] = ts.numpy()
    out["x0"] = x0.numpy()
    out["noise"] = noise.numpy()

    terminal = target.unnorm_log_prob

    def sim(**kw):
        torch.set_rng_state(state)
        return loss.simulate(ts, x0, terminal, second, **kw)

    # --- eval pass 1: compute_weights=True (ito integral on), with trajectory ---
    with torch.no_grad():
        train_kw = {"train": False} if case["loss"]["kind"] == "time_reversal" else {}
        xT, rnd, xs = sim(compute_ito_int=True, return_traj=True, **train_kw)
        out["eval1/x_T"] = xT.numpy()
        out["eval1/rnd"] = rnd.numpy()
        if case["B"] * len(ts) * x0.shape[1] <= 200_000:
            out["eval1/xs"] = xs.numpy()
        torch.set_rng_state(state)
        res = loss.eval(ts, x0, terminal, second, compute_weights=True, return_traj=False)
        assert torch.equal(res.samples, xT)
        out["eval1/weights"] = res.weights.numpy()
        out["eval1/log_norm_const_lb_ito"] = np.float64(res.log_norm_const_preds["log_norm_const_lb_ito"])
        out["eval1/log_norm_const_is"] = np.float64(res.log_norm_const_preds["log_norm_const_is"])
        out["eval1/lv_loss"] = np.float64(res.metrics["eval/lv_loss"])
        # --- eval pass 2: compute_weights=False (solver/oc.py:88-97 "sample_time" semantics) ---
        xT2, rnd2, _ = sim(compute_ito_int=False, return_traj=False, **train_kw)
        out["eval2/x_T"] = xT2.numpy()
        out["eval2/rnd"] = rnd2.numpy()
        torch.set_rng_state(state)
        res2 = loss.eval(ts, x0, terminal, second, compute_weights=False, return_traj=False)
        out["eval2/log_norm_const_lb"] = np.float64(res2.log_norm_const_preds["log_norm_const_lb"])

    # --- train forward + parameter gradients for both kl and lv ---
    for method in ["kl", "lv"]:
        loss.method = method
        loss.n_filtered = 0
        ctrl.zero_grad()
        torch.set_rng_state(state)
        val, metrics = loss(ts, x0, terminal, second)
        val.backward()
        out[f"train_{method}/loss"] = np.float64(val.item())
        out[f"train_{method}/n_filtered"] = np.int64(metrics["train/n_filtered_cumulative"])
        for k, p in ctrl.named_parameters():
            out[f"train_{method}/grad/{k}"] = (p.grad.detach().numpy().copy() if p.grad is not None
                                               else np.zeros(tuple(p.shape), np.float32))
    loss.method = case["loss"]["method"]

    # --- the remaining loss methods (kl_ito, lv_traj with two trajectories per sample) on a few configurations ---
    if name in EXTRA_METHODS:
        loss.method, loss.n_filtered = "kl_ito", 0
        ctrl.zero_grad()
        torch.set_rng_state(state)
        val, metrics = loss(ts, x0, terminal, second)
        val.backward()
        out["train_kl_ito/loss"] = np.float64(val.item())
        for k, p in ctrl.named_parameters():
            out[f"train_kl_ito/grad/{k}"] = p.grad.detach().numpy().copy()
        loss.method, loss.traj_per_sample, loss.n_filtered = "lv_traj", 2, 0
        ctrl.zero_grad()
        torch.manual_seed(case["seed"] + 100)
        x_rep = x0.repeat(2, 1, 1).reshape(-1, x0.shape[-1])
        state2 = torch.get_rng_state()
        out["noise_traj2"] = torch.stack([torch.randn_like(x_rep) for _ in range(len(ts) - 1)]).numpy()
        torch.set_rng_state(state2)
        val, metrics = loss(ts, x0, terminal, second)
        val.backward()
        out["train_lv_traj/loss"] = np.float64(val.item())
        for k, p in ctrl.named_parameters():
            out[f"train_lv_traj/grad/{k}"] = p.grad.detach().numpy().copy()
        loss.method, loss.traj_per_sample = case["loss"]["method"], 1

    # distribution known-answer vectors (reference tests/distr_eval.py:45-55 pins analytic == autograd score)
    torch.manual_seed(99)
    xq = x0 + 0.5 * torch.randn_like(x0)
    out["kat/x"] = xq.numpy()
    out["kat/target_unnorm_log_prob"] = target.unnorm_log_prob(xq).detach().numpy()
    out["kat/target_score"] = target.score(xq.clone()).detach().numpy()
    out["kat/second_log_prob"] = second(xq).detach().numpy()
    if case["prior"]["kind"] != "delta":
        out["kat/prior_score"] = prior.score(xq).detach().numpy()
    if case["target"]["kind"] == "gmm":
        out["target/loc"] = target.loc.numpy()
        out["target/scale"] = target.scale.numpy()
        out["target/mixture_weights"] = target.mixture_weights.numpy()

    out["meta"] = np.frombuffer(json.dumps(meta).encode(), dtype=np.uint8)
    path = OUT / f"{name}.npz"
    np.savez_compressed(path, **out)
    kb = path.stat().st_size / 1024
    print(f"{name:28s} B={case['B']:4d} T={len(ts)-1:4d} d={x0.shape[1]:3d} "
          f"logZ_is={out['eval1/log_norm_const_is']:+.5f} lb={out['eval2/log_norm_const_lb']:+.5f} {kb:7.1f} KB")


def main():
    torch.set_num_threads(1)
    only = set(sys.argv[1:])  # optional: names of the cases to (re)generate
    for name, case in CASES.items():
        if not only or name in only:
            run_case(name, case)


if __name__ == "__main__":
    main()

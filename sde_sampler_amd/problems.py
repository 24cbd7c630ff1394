"""Builds complete sampling problems (target, prior, SDE, control network, loss) from plain-data specs.

This is the stand-in for the reference's Hydra composition (`conf/solver/*.yaml` + `conf/target/*.yaml`, which
needs hydra-core): a spec is a small dict with the same information the YAML files carry, and `build(spec)`
instantiates the host-side classes of this package the way `Solver.__init__`/`setup_models` would
(reference solver/base.py:34-105, solver/oc.py:38-46,127-153,185-197,238-252,282-303).  The BASELINE.json
configurations are available by name through `baseline_spec(name)`.  Used by bench.py, `__graft_entry__.smoke()`
and the tests; the same spec format is stored inside the golden fixtures.
"""
from __future__ import annotations

import copy
import math
from dataclasses import dataclass
from functools import partial
from typing import Callable

import torch
from torch import nn

from sde_sampler_amd.distr.delta import Delta
from sde_sampler_amd.distr.double_well import DoubleWell, MultiWell
from sde_sampler_amd.distr.funnel import Funnel
from sde_sampler_amd.distr.gauss import GMM, IsotropicGauss
from sde_sampler_amd.eq.sdes import VP, ConstOU, ScaledBM
from sde_sampler_amd.losses.oc import ExponentialIntegratorSDELoss, ReferenceSDELoss, TimeReversalLoss
from sde_sampler_amd.models.mlp import FourierMLP, TimeEmbed
from sde_sampler_amd.models.reparam import ClippedCtrl, LerpCtrl, LerpPriorCtrl, LerpTargetCtrl, ScoreCtrl
from sde_sampler_amd.utils.common import get_timesteps

ACTIVATIONS = {"gelu": nn.GELU, "silu": nn.SiLU, "relu": nn.ReLU}


def fab_loc(dim: int) -> torch.Tensor:
    """Means of the 40-mode "fab" mixture (reference distr/gauss.py:42-47), zero-padded to `dim` coordinates."""
    gen = torch.Generator()
    gen.manual_seed(42)
    loc = (torch.rand((40, 2), generator=gen) - 0.5) * 2 * 40
    return loc if dim == 2 else torch.cat([loc, torch.zeros(40, dim - 2)], dim=1)


def random_gmm(dim: int, k: int, seed: int):
    gen = torch.Generator()
    gen.manual_seed(seed)
    loc = (torch.rand((k, dim), generator=gen) - 0.5) * 8.0
    scale = 0.4 + torch.rand((k, dim), generator=gen)
    return loc, scale, 0.2 + torch.rand((k,), generator=gen)


def build_target(spec: dict, tensors: dict | None = None):
    kind = spec["kind"]
    if kind == "double_well":
        return DoubleWell(dim=1, separation=spec["separation"], shift=spec["shift"])
    if kind == "multi_well":
        return MultiWell(dim=spec["dim"], n_double_wells=spec["n_double_wells"], separation=spec["separation"],
                         shift=spec["shift"])
    if kind == "funnel":
        return Funnel(dim=spec["dim"], n_reference_samples=spec.get("n_reference_samples", 1000))
    if kind == "iso_gauss":
        return IsotropicGauss(dim=spec["dim"], loc=spec["loc"], scale=spec["scale"], n_reference_samples=1000)
    if kind == "gmm":
        if tensors is not None:
            return GMM(dim=spec["dim"], loc=tensors["loc"], scale=tensors["scale"],
                       mixture_weights=tensors["mixture_weights"], n_reference_samples=1000, domain_tol=None)
        name, d = spec["name"], spec["dim"]
        if name == "fab50" or (name == "fab" and d > 2):
            scale = torch.nn.functional.softplus(torch.tensor(1.0)) * torch.ones(40, d)
            return GMM(dim=d, loc=fab_loc(d), scale=scale, mixture_weights=torch.ones(40), n_reference_samples=1000,
                       domain_tol=None)
        if name in ("dense40_shared", "dense40_general"):
            # bench.py "extra": 40 modes whose means vary in EVERY coordinate (no varying-prefix shortcut), U(-40, 40) like the
            # "fab" means (distr/gauss.py:42-47); shared scale softplus(1), or per-(component, coordinate) scales in [1, 1.5)
            gen = torch.Generator()
            gen.manual_seed(43)
            loc = (torch.rand((40, d), generator=gen) - 0.5) * 2 * 40
            scale = torch.nn.functional.softplus(torch.tensor(1.0)) * torch.ones(40, d)
            if name == "dense40_general":
                scale = 1.0 + 0.5 * torch.rand((40, d), generator=gen)
            return GMM(dim=d, loc=loc, scale=scale, mixture_weights=torch.ones(40), n_reference_samples=1000, domain_tol=None)
        if name == "random7":
            loc, scale, w = random_gmm(d, 7, 1234)
            return GMM(dim=d, loc=loc, scale=scale, mixture_weights=w, n_reference_samples=1000, domain_tol=None)
        return GMM(dim=d, name=name, n_reference_samples=1000)
    if kind == "nice":
        # BASELINE configs[4]'s target (the reference's data/nice.pt is not shipped, so a NiceModel of the
        # checkpoint's geometry -- scripts/train_nice.py:66-78: coupling = 4, mid_dim = 1000 * 14 / 28 = 500, hidden = 5,
        # mask_config = 1.0 -- with seeded random weights, handed over as `Nice(model=...)`, distr/nice.py:242-243).  `tensors`: a
        # state_dict of the model (the fixtures carry the reference model's)
        from sde_sampler_amd.distr.nice import Nice, NiceModel, StandardLogistic

        with torch.random.fork_rng():
            torch.manual_seed(spec.get("seed", 5))
            model = NiceModel(prior=StandardLogistic(), coupling=spec.get("coupling", 4), in_out_dim=spec["dim"],
                              mid_dim=spec.get("mid_dim", 500), hidden=spec.get("hidden", 5), mask_config=spec.get("mask_config", 1.0))
            with torch.no_grad():  # a live scaling layer (the shipped initialisation is zeros)
                model.scaling.scale.normal_(0.0, spec.get("scale_std", 0.1))
                for layer in model.coupling:  # out_block of a trained flow is not tiny: keep the couplings numerically live
                    layer.out_block.weight.mul_(spec.get("out_gain", 1.0))
        if tensors is not None:
            model.load_state_dict(tensors)
        return Nice(model=model, dim=spec["dim"], n_reference_samples=1000)
    raise ValueError(f"unknown target kind {kind}")


def build_prior(spec: dict):
    if spec["kind"] == "delta":
        return Delta(dim=spec["dim"])
    return IsotropicGauss(dim=spec["dim"], loc=spec.get("loc", 0.0), scale=spec.get("scale", 1.0),
                          truncate_quartile=spec.get("truncate_quartile"))


def build_sde(spec: dict | None):
    if spec is None:
        return None
    kind, gen = spec["kind"], spec.get("generative", True)
    if kind == "vp":
        return VP(diff_coeff_sq_min=spec["beta_min"], diff_coeff_sq_max=spec["beta_max"],
                  scale_diff_coeff=spec.get("scale", 1.0), terminal_t=spec["terminal_t"], generative=gen)
    if kind == "const_ou":
        return ConstOU(drift_coeff=spec["drift_coeff"], diff_coeff=spec["diff_coeff"], terminal_t=spec["terminal_t"],
                       generative=gen)
    if kind == "scaled_bm":
        return ScaledBM(diff_coeff=spec["diff_coeff"], terminal_t=spec["terminal_t"], generative=gen)
    raise ValueError(f"unknown sde kind {kind}")


def build_integration(spec: dict, params: dict | None = None, target_tensors: dict | None = None, device=None):
    """The SDE object handed to `EulerIntegrator.integrate` for an `integrate` spec (the `int_*` golden fixtures):
    kind "langevin" -> LangevinSDE(target.score, diff_coeff, clip_score)              (solver/langevin.py:20-24)
    kind "controlled" -> a bare OU process, or ControlledSDE(OU, ctrl) with the control built on the generative twin of
    the OU process                                                                        (solver/oc.py:130-143).
    Returns (sde, target, prior, ctrl | None)."""
    from sde_sampler_amd.eq.sdes import ControlledSDE, LangevinSDE

    target = build_target(spec["target"], target_tensors)
    prior = build_prior(spec["prior"])
    ispec, ctrl = spec["integrate"], None
    if ispec["kind"] == "langevin":
        sde = LangevinSDE(target_score=target.score, diff_coeff=ispec["diff_coeff"], clip_score=ispec.get("clip_score"),
                          terminal_t=spec["grid"]["end"])
    else:
        sde = build_sde(spec["sde"])
        if spec.get("ctrl"):
            ctrl = build_ctrl(spec["ctrl"], spec["net"], spec["target"]["dim"], build_sde(dict(spec["sde"], generative=True)),
                              prior, target)
            if params is not None:
                ctrl.load_state_dict(params)
            sde = ControlledSDE(sde=sde, ctrl=ctrl)
        elif spec.get("wrap"):
            sde = ControlledSDE(sde=sde, ctrl=None)
    if device is not None:
        for mod in (target, prior, sde, ctrl):
            if mod is not None:
                mod.to(device)
    return sde, target, prior, ctrl


def build_ctrl(spec: dict, net: dict, dim: int, sde, prior, target, live_last_layers: bool = True):
    """`conf/model/{clipped,score,lerp,lerp_target,lerp_prior}.yaml` + `conf/model/base/*.yaml`."""
    act = ACTIVATIONS[net["activation"]]()
    zeros_ = nn.init.zeros_
    base = FourierMLP(dim=dim, activation=act, num_layers=net["num_layers"], channels=net["channels"],
                      last_bias_init=zeros_, last_weight_init=zeros_)
    kind = spec["kind"]
    if kind == "clipped":
        ctrl = ClippedCtrl(base_model=base, clip_model=spec.get("clip_model"))
    else:
        gamma = TimeEmbed(dim_out=spec.get("gamma_dim", 1), activation=act, num_layers=4, channels=net["channels"],
                          last_bias_init=partial(nn.init.constant_, val=spec.get("gamma_bias", 1.0)),
                          last_weight_init=zeros_)
        common = dict(base_model=base, score_model=gamma, target_score=target.score, detach_score=spec.get("detach_score", False),
                      clip_score=spec.get("clip_score"), clip_model=spec.get("clip_model"),
                      scale_score=spec.get("scale_score", 1.0))
        if kind == "score":
            ctrl = ScoreCtrl(**common)
        else:
            cls = {"lerp": LerpCtrl, "lerp_target": LerpTargetCtrl, "lerp_prior": LerpPriorCtrl}[kind]
            ctrl = cls(**common, sde=sde, prior_score=prior.score)
    if live_last_layers:
        # the shipped init zeroes the last layers (u == score term only); give them N(0, 0.05^2) entries so that
        # benchmarks and tests exercise a numerically live network
        with torch.no_grad():
            heads = [ctrl.base_model.out_layer] + ([ctrl.score_model.out_layer] if kind != "clipped" else [])
            for head in heads:
                head.weight.normal_(0.0, 0.05)
                head.bias.add_(torch.randn_like(head.bias) * 0.05)
    return ctrl


class PriorScoreReferenceCtrl:
    """EulerDDS reference control  sigma(t) * grad log prior(x)  (reference solver/oc.py:305-306)."""

    def __init__(self, sde, prior):
        self.sde, self.prior = sde, prior

    def __call__(self, t, x):
        return self.sde.diff(t, x) * self.prior.score(x)


@dataclass
class Problem:
    spec: dict
    target: object
    prior: object
    sde: object
    ctrl: nn.Module
    loss: object
    second_log_prob: Callable  # initial_log_prob (DIS) / reference_log_prob (PIS, DDS)
    reference_distr: object
    ts: torch.Tensor

    def to(self, device):
        for mod in (self.target, self.prior, self.sde, self.ctrl, self.reference_distr,
                    getattr(self.loss, "inference_ctrl", None)):
            if isinstance(mod, nn.Module):
                mod.to(device)
        self.ts = self.ts.to(device)
        return self

    def eval(self, x, compute_weights=True, return_traj=False, noise=None):
        """`TrainableDiff._compute_results` (reference solver/oc.py:217-231 etc.)."""
        with torch.no_grad():
            return self.loss.eval(self.ts, x, self.target.unnorm_log_prob, self.second_log_prob,
                                  compute_weights=compute_weights, return_traj=return_traj, noise=noise)


def build(spec: dict, params: dict | None = None, target_tensors: dict | None = None, device=None,
          params_inf: dict | None = None) -> Problem:
    spec = copy.deepcopy(spec)
    torch.manual_seed(spec.get("init_seed", 1))  # conf/base.yaml:8
    target = build_target(spec["target"], target_tensors)
    prior = build_prior(spec["prior"])
    sde = build_sde(spec.get("sde"))
    dim = spec["target"]["dim"]
    ctrl = build_ctrl(spec["ctrl"], spec["net"], dim, sde, prior, target)
    if params is not None:
        missing, unexpected = ctrl.load_state_dict(params, strict=False)
        assert not unexpected and all("timestep_coeff" in m for m in missing), (missing, unexpected)
    ls = spec["loss"]
    common = dict(generative_ctrl=ctrl, sde=sde, method=ls["method"], max_rnd=ls.get("max_rnd"),
                  traj_per_sample=ls.get("traj_per_sample", 1), filter_samples=getattr(target, "filter", None))
    reference = None
    if ls["kind"] == "time_reversal":
        inference = None
        if spec.get("inference_ctrl"):  # Bridge (solver/oc.py:127-153): a second control over the same sde / prior / target
            inference = build_ctrl(spec["inference_ctrl"], spec.get("inference_net", spec["net"]), dim, sde, prior, target)
            if params_inf is not None:
                missing, unexpected = inference.load_state_dict(params_inf, strict=False)
                assert not unexpected and all("timestep_coeff" in m for m in missing), (missing, unexpected)
        loss = TimeReversalLoss(**common, inference_ctrl=inference, div_estimator=ls.get("div_estimator"))
        second = prior.log_prob
    elif ls["kind"] == "reference_sde":
        if ls.get("reference_ctrl") == "prior_score":
            reference = sde.marginal_distr(sde.terminal_t, x_init=prior.loc, var_init=prior.scale**2)
            loss = ReferenceSDELoss(**common, reference_ctrl=PriorScoreReferenceCtrl(sde, prior))
        else:
            reference = sde.marginal_distr(t=sde.terminal_t, x_init=prior.loc)
            loss = ReferenceSDELoss(**common)
        second = reference.log_prob
    elif ls["kind"] == "exponential":
        loss = ExponentialIntegratorSDELoss(**common, alpha=ls["alpha"], sigma=ls["sigma"])
        second = prior.log_prob
    else:
        raise ValueError(ls["kind"])
    g = spec["grid"]
    ts = get_timesteps(g["start"], g["end"], steps=g["steps"], rescale_t=g.get("rescale_t"))
    prob = Problem(spec, target, prior, sde, ctrl, loss, second, reference, ts)
    return prob.to(device) if device is not None else prob


# ----------------------------------------------------------------------------------------------------------
# BASELINE.json configurations (SURVEY.md 8d "concrete synthetic inputs")
# ----------------------------------------------------------------------------------------------------------
_NET = dict(channels=64, num_layers=4, activation="gelu")
_VP10 = dict(kind="vp", beta_min=0.1, beta_max=10.0, scale=1.0, terminal_t=1.0)
_LERP = dict(kind="lerp", clip_model=1e4, clip_score=1e4, scale_score=1.0, gamma_dim=1, gamma_bias=1.0)
_SCORE = dict(kind="score", clip_model=1e4, clip_score=1e4, scale_score=1.0, gamma_dim=1, gamma_bias=0.01)

BASELINE_SPECS = {
    # configs[0]: target=dw_shift solver=basic_dis loss.method=lv, batch 1024, 100 EM steps
    "cfg1_dw_dis_lv": dict(
        batch=1024, target=dict(kind="double_well", dim=1, separation=2.0, shift=1.5),
        prior=dict(kind="iso_gauss", dim=1), sde=_VP10, ctrl=_LERP, net=_NET,
        loss=dict(kind="time_reversal", method="lv", max_rnd=1e8), grid=dict(start=0.0, end=1.0, steps=100)),
    # configs[1]: target=gmm (40 modes, d=2) solver=basic_dis loss.method=kl, batch 65 536, 100 steps
    "cfg2_gmm2_dis_kl": dict(
        batch=65536, target=dict(kind="gmm", dim=2, name="fab"),
        prior=dict(kind="iso_gauss", dim=2), sde=_VP10, ctrl=_LERP, net=_NET,
        loss=dict(kind="time_reversal", method="kl"), grid=dict(start=0.0, end=1.0, steps=100)),
    # the metric's headline: GMM-40 d=50, solver=basic_pis (ScoreCtrl, Delta prior, ScaledBM), batch 65 536, T=100
    "gmm50_pis_headline": dict(
        batch=65536, target=dict(kind="gmm", dim=50, name="fab50"),
        prior=dict(kind="delta", dim=50), sde=dict(kind="scaled_bm", diff_coeff=math.sqrt(0.2), terminal_t=5.0),
        ctrl=_SCORE, net=_NET, loss=dict(kind="reference_sde", method="kl"), grid=dict(start=0.0, end=5.0, steps=100)),
    # bench.py "extra": the headline with mixture means varying in all 50 coordinates
    "gmm50_dense_shared": dict(
        batch=65536, target=dict(kind="gmm", dim=50, name="dense40_shared"),
        prior=dict(kind="delta", dim=50), sde=dict(kind="scaled_bm", diff_coeff=math.sqrt(0.2), terminal_t=5.0),
        ctrl=_SCORE, net=_NET, loss=dict(kind="reference_sde", method="kl"), grid=dict(start=0.0, end=5.0, steps=100)),
    "gmm50_dense_general": dict(
        batch=65536, target=dict(kind="gmm", dim=50, name="dense40_general"),
        prior=dict(kind="delta", dim=50), sde=dict(kind="scaled_bm", diff_coeff=math.sqrt(0.2), terminal_t=5.0),
        ctrl=_SCORE, net=_NET, loss=dict(kind="reference_sde", method="kl"), grid=dict(start=0.0, end=5.0, steps=100)),
    # configs[2]: same target, batch 262 144 over 8 GPUs (32 768 per GPU), 200 steps
    "cfg3_gmm50_pis_kl": dict(
        batch=32768, target=dict(kind="gmm", dim=50, name="fab50"),
        prior=dict(kind="delta", dim=50), sde=dict(kind="scaled_bm", diff_coeff=math.sqrt(0.2), terminal_t=5.0),
        ctrl=_SCORE, net=_NET, loss=dict(kind="reference_sde", method="kl"), grid=dict(start=0.0, end=5.0, steps=200)),
    # configs[3]: target=funnel d=10 solver=dds loss.method=lv, batch 131 072 over 4 GPUs, "400 steps" (cosine grid)
    "cfg4_funnel_dds_lv": dict(
        batch=32768, target=dict(kind="funnel", dim=10),
        prior=dict(kind="iso_gauss", dim=10, truncate_quartile=1e-4), sde=None,
        ctrl=dict(_SCORE, clip_model=10.0, clip_score=10.0), net=_NET,
        loss=dict(kind="exponential", method="lv", max_rnd=1e8, alpha=1.0, sigma=1.0),
        grid=dict(start=0.0, end=12.8, steps=400, rescale_t="cosine")),
    # Wide-network workloads (bench.py --workload): the geometry of configs[4] -- d = 196 (MNIST 14 x 14 after NICE's preprocessing
    # would be 784; the task's configs[4] is quoted with channels = 256), two FourierMLP C = 256 -- on a target the kernels carry in closed form (one
    # launch over all steps; the NICE flow itself: "cfg5_nice_bridge196" below, stepped around csrc/sdeh_nice.hip); clips of conf/solver/bridge.yaml
    "wide_pis_funnel196": dict(
        batch=32768, target=dict(kind="funnel", dim=196),
        prior=dict(kind="delta", dim=196), sde=dict(kind="scaled_bm", diff_coeff=math.sqrt(0.2), terminal_t=5.0),
        ctrl=dict(_SCORE, clip_model=10.0, clip_score=10.0), net=dict(channels=256, num_layers=4, activation="gelu"),
        loss=dict(kind="reference_sde", method="kl"), grid=dict(start=0.0, end=5.0, steps=200)),
    # configs[4]: solver=bridge, channels=256, batch 32 768 over 8 GPUs (4096 per GPU), 200 steps
    "cfg5_like_bridge196": dict(
        batch=4096, target=dict(kind="funnel", dim=196),
        prior=dict(kind="iso_gauss", dim=196), sde=dict(kind="scaled_bm", diff_coeff=1.0, terminal_t=1.0),
        ctrl=dict(kind="lerp_target", clip_model=10.0, clip_score=10.0, scale_score=1.0, gamma_dim=1, gamma_bias=1.0),
        inference_ctrl=dict(kind="lerp_prior", clip_model=10.0, clip_score=10.0, scale_score=1.0, gamma_dim=1, gamma_bias=1.0),
        net=dict(channels=256, num_layers=4, activation="gelu"),
        loss=dict(kind="time_reversal", method="lv", max_rnd=1e8), grid=dict(start=0.0, end=1.0, steps=200)),
    # configs[4] AS WRITTEN: target = nice (the flow's geometry of scripts/train_nice.py with seeded random weights: SURVEY.md 8d item 5),
    # solver = bridge, channels = 256, batch 32 768 over 8 GPUs (4096 per GPU), 200 steps.  The flow's score is evaluated by
    # csrc/sdeh_nice.hip between the step segments of the wide Bridge kernel (engine.run: SDEH_DENS_EXTERNAL)
    "cfg5_nice_bridge196": dict(
        batch=4096, target=dict(kind="nice", dim=196),
        prior=dict(kind="iso_gauss", dim=196), sde=dict(kind="scaled_bm", diff_coeff=1.0, terminal_t=1.0),
        ctrl=dict(kind="lerp_target", clip_model=10.0, clip_score=10.0, scale_score=1.0, gamma_dim=1, gamma_bias=1.0),
        inference_ctrl=dict(kind="lerp_prior", clip_model=10.0, clip_score=10.0, scale_score=1.0, gamma_dim=1, gamma_bias=1.0),
        net=dict(channels=256, num_layers=4, activation="gelu"),
        loss=dict(kind="time_reversal", method="lv", max_rnd=1e8), grid=dict(start=0.0, end=1.0, steps=200)),
}


def baseline_spec(name: str) -> dict:
    return copy.deepcopy(BASELINE_SPECS[name])

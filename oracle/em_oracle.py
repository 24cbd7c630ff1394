"""CPU oracle for the Euler-Maruyama trajectory path -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` may import this
module, and only as the checker / the timed CPU baseline.  The shipped path (``sde_sampler_amd``) never
imports it and fails loudly when its HIP library is missing.

What it is: a plain PyTorch-CPU, fp32, op-for-op restatement of the reference's hot path -- the same ATen
ops in the same order, so that on identical inputs (weights, x0, per-step noise) it reproduces the
reference's outputs BIT-EXACTLY.  It is written functionally (parameter dict + plain-data problem spec), not
as the reference's module tree.

Parity pin: ``tests/test_oracle_golden.py`` checks every function here against the golden vectors in
``tests/golden/*.npz``, which were produced by importing and running the reference itself
(``tests/golden/make_golden.py``, build container only).  The reference's own test-suite has no test for
this path; its only adjacent known-answer test -- analytic score == autograd score at
rtol=atol=1e-4, reference tests/distr_eval.py:45-55 -- is reproduced as well.

Reference lines restated (all relative to /root/reference/sde_sampler):
  utils/common.py:18-55     -> timesteps()
  eq/sdes.py:68-269         -> class Sde (VP / ConstOU / ScaledBM coefficient functions, marginal_params)
  models/mlp.py:71-82       -> time_embed()
  models/mlp.py:114-122     -> fourier_mlp()
  models/reparam.py:13-200  -> class Ctrl (clipped / score / lerp / lerp_target / lerp_prior)
  distr/gauss.py:123-140,215-223, distr/double_well.py:39-45,165-179, distr/funnel.py:54-80,
  distr/base.py:130-137     -> class Density
  losses/oc.py:50-58,72-123 -> filter_mask(), compute_loss(), compute_results()
  losses/oc.py:156-230      -> simulate() kind="time_reversal" (incl. the Bridge branch 189-202 with utils/autograd.py:81-105)
  losses/oc.py:286-343      -> simulate() kind="reference_sde"
  losses/oc.py:400-457      -> simulate() kind="exponential"
  solver/oc.py:189-191,243,288-306 -> Problem.second_log_prob / reference_ctrl wiring
  eq/integrator.py:66-77,93-127    -> interpolate(), euler_integrate()
  eq/sdes.py:38-65,272-305         -> integration_case(): LangevinSDE / ControlledSDE drift and diffusion
"""
from __future__ import annotations

import math
from typing import Callable

import torch
import torch.nn.functional as F
from torch import distributions as D

Tensor = torch.Tensor


# ------------------------------------------------------------------------------------------------
# time grid (utils/common.py:18-55)
# ------------------------------------------------------------------------------------------------
def timesteps(start, end, dt=None, steps=None, rescale_t=None, device=None) -> Tensor:
    if (steps is None) is (dt is None):
        raise ValueError("Exactly one of `dt` and `steps` should be defined.")
    if steps is None:
        steps = int(math.ceil((end - start) / dt))
    if rescale_t is None:
        return torch.linspace(start, end, steps=steps + 1, device=device)
    if rescale_t == "quad":
        end = torch.as_tensor(end, dtype=torch.float, device=device)
        return torch.sqrt(torch.linspace(start, end.square(), steps=steps + 1, device=device)).clip(max=end)
    if rescale_t == "cosine":
        pre = torch.linspace(start, end, steps + 1, device=device) / end
        phase = ((pre + 0.008) / (1 + 0.008)) * torch.pi * 0.5
        dts = torch.cos(phase) ** 4
        dts /= dts.sum()
        dts *= end
        return torch.concat((torch.tensor([start], device=device), torch.cumsum(dts, -1)))
    raise ValueError("Unkown timestep rescaling method.")


def clip(v: Tensor, m) -> Tensor:
    """utils/common.py:83-84."""
    if m is not None:
        v = v.clip(min=-1.0 * m, max=m)
    return v


ACTIVATIONS: dict[str, Callable] = {
    "gelu": lambda v: F.gelu(v),  # exact erf form (conf/model/base/fouriermlp.yaml:5-6)
    "silu": F.silu,
    "relu": F.relu,
}


# ------------------------------------------------------------------------------------------------
# SDE coefficient functions (eq/sdes.py)
# ------------------------------------------------------------------------------------------------
class Sde:
    def __init__(self, spec: dict):
        f32 = lambda v: torch.tensor(v, dtype=torch.float)
        self.kind = spec["kind"]
        self.terminal_t = f32(spec["terminal_t"])
        self.generative = spec.get("generative", True)
        self.sign = 1.0 if self.generative else -1.0  # eq/sdes.py:76-77
        if self.kind == "vp":
            self.beta_min, self.beta_max, self.scale = f32(spec["beta_min"]), f32(spec["beta_max"]), f32(spec["scale"])
        elif self.kind == "const_ou":
            self.a, self.b = f32(spec["drift_coeff"]), f32(spec["diff_coeff"])
        elif self.kind == "scaled_bm":
            self.a, self.b = f32(0.0), f32(spec["diff_coeff"])
        else:
            raise ValueError(self.kind)

    # VP: eq/sdes.py:222-245
    def _beta(self, t):
        if self.generative:
            return torch.lerp(self.beta_max, self.beta_min, t / self.terminal_t)
        return torch.lerp(self.beta_min, self.beta_max, t / self.terminal_t)

    def drift_coeff(self, t):
        if self.kind == "vp":
            return self.sign * 0.5 * self._beta(t)
        return self.sign * self.a

    def diff(self, t, x=None):
        if self.kind == "vp":
            return self.scale * torch.sqrt(self._beta(t))
        return self.b

    def drift(self, t, x):
        return self.drift_coeff(t) * x

    def int_drift_coeff(self, s, t):
        dt = t - s
        if self.kind == "vp":
            return self.sign * 0.25 * (self._beta(t) + self._beta(s)) * dt
        return self.sign * self.a * dt

    def drift_div_int(self, s, t, x):
        return self.int_drift_coeff(s, t) * x.shape[-1]

    def marginal_params(self, t, x_init, var_init=None):
        """eq/sdes.py:157-188,257-269 -- only used to build the PIS / EulerDDS reference Gaussian."""
        if self.kind == "vp":
            s = torch.zeros(1)
            i = self.int_drift_coeff(s, t)
            loc = torch.exp(i)
            var = (1 - torch.exp(2 * i)) * self.scale**2
            if var_init is not None:
                var = var + loc**2 * var_init
            return loc * x_init, var
        if self.kind == "scaled_bm":
            var = self.b**2 * t
            if var_init is not None:
                var = var + var_init
            return x_init, var
        a = self.sign * self.a
        loc = torch.exp(a * t)
        var = -self.b**2 / (2 * a) * (1 - torch.exp(2 * a * t))
        if var_init is not None:
            var = var + loc**2 * var_init
        return loc * x_init, var


# ------------------------------------------------------------------------------------------------
# densities (targets / priors / reference Gaussians)
# ------------------------------------------------------------------------------------------------
def _prep(param, dim):
    if not isinstance(param, Tensor):
        param = torch.tensor(param, dtype=torch.float)
    param = torch.atleast_2d(param)
    if param.numel() == 1:
        param = param.repeat(1, dim)
    return param


class Density:
    """unnorm_log_prob / log_prob / score with the reference's arithmetic."""

    def __init__(self, spec: dict, tensors: dict | None = None):
        self.kind = spec["kind"]
        self.dim = spec["dim"]
        self.log_norm_const = spec.get("log_norm_const", 0.0)
        t = tensors or {}
        if self.kind == "gmm":
            self.loc, self.scale, self.w = t["loc"], t["scale"], t.get("mixture_weights")
            if self.w is None:
                self.distr = D.Independent(D.Normal(self.loc.squeeze(0), self.scale.squeeze(0)), 1)
            else:
                self.distr = D.MixtureSameFamily(D.Categorical(self.w), D.Independent(D.Normal(self.loc, self.scale), 1))
        elif self.kind in ("iso_gauss", "delta"):
            loc = spec.get("loc", 0.0)
            scale = spec.get("scale", 1e-3 if self.kind == "delta" else 1.0)
            self.loc, self.scale = _prep(loc, self.dim), _prep(scale, self.dim)
            self.distr = D.Independent(D.Normal(self.loc.squeeze(0), self.scale.squeeze(0)), 1)
        elif self.kind == "diag_gauss":
            self.loc, self.scale = _prep(t["loc"], self.dim), _prep(t["scale"], self.dim)
            self.distr = D.Independent(D.Normal(self.loc.squeeze(0), self.scale.squeeze(0)), 1)
        elif self.kind == "double_well":
            self.sep, self.shift = torch.tensor(spec["separation"]), torch.tensor(spec["shift"])
            self.log_norm_const = spec.get("log_norm_const")
        elif self.kind == "multi_well":
            self.n_dw = spec["n_double_wells"]
            self.sep, self.shift = torch.tensor(spec["separation"]), torch.tensor(spec["shift"])
            n_gauss = self.dim - self.n_dw
            self.gauss = None
            if n_gauss > 0:
                self.gauss = Density(dict(kind="iso_gauss", dim=n_gauss, loc=spec["shift"],
                                          log_norm_const=0.5 * math.log(2.0 * math.pi) * n_gauss))
            self.log_norm_const = spec.get("log_norm_const")
        elif self.kind == "funnel":
            var = spec.get("variance") or (self.dim - 1)
            self.first = Density(dict(kind="iso_gauss", dim=1, scale=math.sqrt(var)))
        elif self.kind == "nice":  # distr/nice.py:233-298 around NiceModel 123-231; tensors = the model's state_dict
            self.nice = {k: v for k, v in t.items()}
            self.n_coupling = 1 + max(int(k.split(".")[1]) for k in self.nice if k.startswith("coupling."))
            self.n_mid = 1 + max((int(k.split(".")[3]) for k in self.nice if ".mid_block." in k), default=-1)
            self.mask_config = spec.get("mask_config", 1.0)
        else:
            raise ValueError(self.kind)

    # -- log densities -------------------------------------------------------------------------
    def unnorm_log_prob(self, x: Tensor) -> Tensor:
        k = self.kind
        if k in ("gmm", "diag_gauss", "delta"):  # distr/gauss.py:137-140
            return self.distr.log_prob(x).unsqueeze(-1) + self.log_norm_const
        if k == "iso_gauss":  # distr/gauss.py:215-220
            var = self.scale[0, 0] ** 2
            norm_const = -0.5 * self.dim * (2.0 * math.pi * var).log()
            norm_const += self.log_norm_const
            sq_sum = torch.sum((x - self.loc[0, 0]) ** 2, dim=-1, keepdim=True)
            return norm_const - 0.5 * sq_sum / var
        if k == "double_well":  # distr/double_well.py:39-41
            y = x - self.shift
            return -((y**2 - self.sep) ** 2)
        if k == "multi_well":  # distr/double_well.py:165-172
            y = x[:, : self.n_dw] - self.shift
            lp = (-((y**2 - self.sep) ** 2)).sum(dim=-1, keepdim=True)
            if self.gauss is not None:
                lp += self.gauss.unnorm_log_prob(x[:, self.n_dw:])
            return lp
        if k == "funnel":  # distr/funnel.py:54-69
            x_first = x[:, 0].unsqueeze(-1)
            x_other = x[:, 1:]
            lp_first = self.first.unnorm_log_prob(x_first)
            norm_const = -x_other.shape[-1] * (x_first + math.log(2.0 * math.pi)) / 2.0
            x_sq_sum = (x_other**2).sum(dim=-1, keepdim=True)
            lp_other = norm_const - 0.5 * x_sq_sum * (-x_first).exp()
            return lp_first + lp_other + self.log_norm_const
        if k == "nice":  # nice.py:276-277 -> NiceModel.log_prob 176-189 -> f 164-174 -> Coupling.forward 64-97, Scaling 112-120
            p, F = self.nice, torch.nn.functional
            z = x
            for i in range(self.n_coupling):
                batch, width = z.shape
                z = z.reshape((batch, width // 2, 2))
                if (self.mask_config + i) % 2:  # NiceModel.__init__ 146: mask_config of coupling i
                    on, off = z[:, :, 0], z[:, :, 1]
                else:
                    off, on = z[:, :, 0], z[:, :, 1]
                h = F.relu(F.linear(off, p[f"coupling.{i}.in_block.0.weight"], p[f"coupling.{i}.in_block.0.bias"]))
                for l in range(self.n_mid):
                    h = F.relu(F.linear(h, p[f"coupling.{i}.mid_block.{l}.0.weight"], p[f"coupling.{i}.mid_block.{l}.0.bias"]))
                on = on + F.linear(h, p[f"coupling.{i}.out_block.weight"], p[f"coupling.{i}.out_block.bias"])
                z = torch.stack((on, off), dim=2) if (self.mask_config + i) % 2 else torch.stack((off, on), dim=2)
                z = z.reshape((batch, width))
            scale = p["scaling.scale"]
            z = z * torch.exp(scale)
            log_ll = torch.sum(-(F.softplus(z) + F.softplus(-z)), dim=1)  # StandardLogistic.log_prob 21-29
            return (log_ll + torch.sum(scale)).unsqueeze(-1) + self.log_norm_const
        raise ValueError(k)

    def log_prob(self, x: Tensor) -> Tensor:  # distr/base.py:116-119
        if self.log_norm_const is None:
            raise NotImplementedError
        return self.unnorm_log_prob(x) - self.log_norm_const

    # -- scores --------------------------------------------------------------------------------
    def autograd_score(self, x: Tensor, create_graph=False) -> Tensor:  # distr/base.py:130-137
        grad = x.requires_grad
        x.requires_grad_(True)
        with torch.set_grad_enabled(True):
            log_rho = self.unnorm_log_prob(x).sum()
            score = torch.autograd.grad(log_rho, x, create_graph=create_graph)[0]
        x.requires_grad_(grad)
        return score

    def score(self, x: Tensor, create_graph=False) -> Tensor:
        k = self.kind
        if (k == "gmm" and self.w is not None) or k == "nice":  # (Nice has no score of its own: distr/base.py:130-137)
            return self.autograd_score(x, create_graph=create_graph)
        if k in ("gmm", "diag_gauss", "delta"):  # Gauss.score, distr/gauss.py:182-183
            return (self.loc - x) / self.scale**2
        if k == "iso_gauss":  # distr/gauss.py:222-223
            return (self.loc[0, 0] - x) / self.scale[0, 0] ** 2
        if k == "double_well":  # distr/double_well.py:43-45
            y = x - self.shift
            return -4.0 * (y**2 - self.sep) * y
        if k == "multi_well":  # distr/double_well.py:174-179
            y = x[:, : self.n_dw] - self.shift
            score = -4.0 * (y**2 - self.sep) * y
            if self.gauss is not None:
                score = torch.cat([score, self.gauss.score(x[:, self.n_dw:])], dim=-1)
            return score
        if k == "funnel":  # distr/funnel.py:71-80
            x_first = x[:, 0].unsqueeze(-1)
            x_other = x[:, 1:]
            inv_var_other = (-x_first).exp()
            score_first = self.first.score(x_first) - 0.5 * x_other.shape[-1]
            score_first += 0.5 * (x_other**2).sum(dim=-1, keepdim=True) * inv_var_other
            score_other = -x_other * inv_var_other
            return torch.cat([score_first, score_other], dim=-1)
        raise ValueError(k)


# ------------------------------------------------------------------------------------------------
# networks (models/mlp.py), as pure functions of a flat parameter dict with the module-tree key names
# ------------------------------------------------------------------------------------------------
def _n_indexed(p: dict, prefix: str) -> int:
    n = 0
    while f"{prefix}{n}.weight" in p:
        n += 1
    return n


def time_embed(p: dict, prefix: str, t: Tensor, act: Callable, channels: int) -> Tensor:
    """models/mlp.py:71-82.  `timestep_coeff` is a non-persistent buffer: linspace(0.1, 100, C)[None]."""
    coeff = torch.linspace(start=0.1, end=100, steps=channels).unsqueeze(0)
    t = t.view(-1, 1).float()
    phase = p[prefix + "timestep_phase"]
    sin_e = torch.sin((coeff * t) + phase)
    cos_e = torch.cos((coeff * t) + phase)
    e = torch.cat([sin_e, cos_e], dim=1)
    for i in range(_n_indexed(p, prefix + "hidden_layer.")):
        e = act(F.linear(e, p[f"{prefix}hidden_layer.{i}.weight"], p[f"{prefix}hidden_layer.{i}.bias"]))
    return F.linear(e, p[prefix + "out_layer.weight"], p[prefix + "out_layer.bias"])


def fourier_mlp(p: dict, prefix: str, t: Tensor, x: Tensor, act: Callable, channels: int) -> Tensor:
    """models/mlp.py:114-122 (the time embedding is evaluated on B identical rows, as the reference does)."""
    t = t.view(-1, 1).expand(x.shape[0], 1).float()
    embed_t = time_embed(p, prefix + "timestep_embed.", t, act, channels)
    embed_x = F.linear(x, p[prefix + "input_embed.weight"], p[prefix + "input_embed.bias"])
    e = embed_x + embed_t
    for i in range(_n_indexed(p, prefix + "hidden_layer.")):
        e = F.linear(act(e), p[f"{prefix}hidden_layer.{i}.weight"], p[f"{prefix}hidden_layer.{i}.bias"])
    return F.linear(act(e), p[prefix + "out_layer.weight"], p[prefix + "out_layer.bias"])


class Ctrl:
    """models/reparam.py control parametrisations, `forward(t, x)` == __call__."""

    def __init__(self, spec: dict, net: dict, params: dict, sde: Sde | None, prior: Density | None, target: Density):
        self.kind = spec["kind"]
        self.clip_model = spec.get("clip_model")
        self.clip_score = spec.get("clip_score")
        self.scale_score = spec.get("scale_score", 1.0)
        self.detach_score = spec.get("detach_score", False)
        self.p, self.act, self.channels = params, ACTIVATIONS[net["activation"]], net["channels"]
        self.sde, self.prior, self.target = sde, prior, target

    def base(self, t, x):
        return clip(fourier_mlp(self.p, "base_model.", t, x, self.act, self.channels), self.clip_model)

    def gamma(self, t):
        return clip(time_embed(self.p, "score_model.", t, self.act, self.channels), self.clip_model)

    def __call__(self, t: Tensor, x: Tensor) -> Tensor:
        ctrl = self.base(t, x)
        if self.kind == "clipped":
            return ctrl
        xs = x.detach() if self.detach_score else x
        if self.kind == "score":  # reparam.py:56-83
            sc = clip(self.target.score(xs, create_graph=self.detach_score), self.clip_score)
            score = self.scale_score * sc
            score *= self.gamma(t)
            return ctrl + score
        w = t / self.sde.terminal_t
        if self.kind == "lerp":  # reparam.py:131-144
            out = self.target.score(xs, create_graph=self.detach_score)
            out = torch.lerp(self.prior.score(xs), out, w)
        elif self.kind == "lerp_target":  # reparam.py:185-197
            out = w * self.target.score(xs)
        elif self.kind == "lerp_prior":  # reparam.py:166-178
            out = (1.0 - w) * self.prior.score(xs)
        else:
            raise ValueError(self.kind)
        score = self.scale_score * clip(out, self.clip_score)
        score *= self.gamma(t)
        return ctrl + self.sde.diff(t, x) * score  # reparam.py:149-162


# ------------------------------------------------------------------------------------------------
# exact divergence of a control by automatic differentiation (utils/autograd.py:14-22,81-105)
# ------------------------------------------------------------------------------------------------
def compute_divx(fn, t: Tensor, x: Tensor, create_graph: bool = True, noise_type=None, probe: Tensor | None = None):
    """utils/autograd.py:81-105 (+ _compute_autodiv 14-22, _estimate_autodiv 25-42 with n_samples = 1).  `probe` replaces the
    estimator's own draw (randint_like * 2 - 1 for "rademacher", randn_like for "gauss")."""
    requires_grad = x.requires_grad
    with torch.set_grad_enabled(True):
        x.requires_grad_(True)
        outputs = fn(t, x)
        if noise_type is None:
            div = 0.0
            for i in range(outputs.shape[-1]):
                div = div + torch.autograd.grad(outputs[:, i].sum(), x, create_graph=create_graph, retain_graph=True)[0][:, i:i + 1]
        else:
            if probe is not None:
                noise = probe
            elif noise_type == "rademacher":
                noise = torch.randint_like(outputs, low=0, high=2).float() * 2 - 1.0
            elif noise_type == "gauss":
                noise = torch.randn_like(outputs)
            else:
                raise NotImplementedError(f"Undefined noise type {noise_type}.")
            grad = torch.autograd.grad(outputs, x, grad_outputs=noise, create_graph=create_graph)[0]
            div = 0.0 + (grad * noise).sum(dim=-1, keepdims=True)
    x.requires_grad_(requires_grad)
    if not torch.is_grad_enabled():
        outputs = outputs.detach()
    return div, outputs


# ------------------------------------------------------------------------------------------------
# estimators (losses/oc.py:50-123)
# ------------------------------------------------------------------------------------------------
def filter_mask(rnd: Tensor, max_rnd=None) -> Tensor:
    if max_rnd is None:
        return True & rnd.isfinite()
    return True & (rnd < max_rnd)


def compute_loss(rnd: Tensor, method: str, max_rnd=None, traj_per_sample: int = 1):
    mask = filter_mask(rnd, max_rnd)
    if method == "lv_traj":
        rnd = rnd.reshape(traj_per_sample, -1, 1)
        mask = mask.reshape(traj_per_sample, -1, 1).all(dim=0)
        n_filtered = traj_per_sample * (mask.numel() - mask.sum()).item()
        return rnd[:, mask].var(dim=0).mean(), n_filtered
    n_filtered = (mask.numel() - mask.sum()).item()
    if method == "lv":
        return rnd[mask].var(), n_filtered
    return rnd[mask].mean(), n_filtered


def compute_results(rnd: Tensor, compute_weights: bool) -> dict:
    neg = -rnd
    if compute_weights:
        m = neg.max()
        w = (neg - m).exp()
        return dict(weights=w, log_norm_const_lb_ito=neg.mean().item(),
                    log_norm_const_is=(w.mean().log() + m).item(), lv_loss=rnd.var().item())
    return dict(weights=None, log_norm_const_lb=neg.mean().item())


# ------------------------------------------------------------------------------------------------
# the problem = plain-data spec (the JSON stored in each golden fixture) + parameter dict
# ------------------------------------------------------------------------------------------------
class Problem:
    def __init__(self, meta: dict, params: dict, target_tensors: dict | None = None, params_inf: dict | None = None):
        self.meta = meta
        tspec, pspec = dict(meta["target"]), dict(meta["prior"])
        self.dim = tspec["dim"]
        self.sde = Sde(meta["sde"]) if meta.get("sde") else None
        self.target = Density(tspec, target_tensors)
        self.prior = Density(pspec)
        self.ctrl = Ctrl(meta["ctrl"], meta["net"], params, self.sde, self.prior, self.target)
        lspec = meta["loss"]
        self.kind, self.method, self.max_rnd = lspec["kind"], lspec["method"], lspec.get("max_rnd")
        self.alpha, self.sigma = lspec.get("alpha"), lspec.get("sigma")
        self.clip_target = meta.get("clip_target")
        self.reference_ctrl = None
        # Bridge: a second control network whose divergence enters the cost (losses/oc.py:189-202, solver/oc.py:127-143)
        self.inference_ctrl, self.div_estimator = None, lspec.get("div_estimator")
        if meta.get("inference_ctrl"):
            self.inference_ctrl = Ctrl(meta["inference_ctrl"], meta.get("inference_net", meta["net"]), params_inf or {},
                                       self.sde, self.prior, self.target)
        if self.kind == "time_reversal":
            self.second = self.prior  # initial_log_prob = prior.log_prob
        elif self.kind == "reference_sde":
            if lspec.get("reference_ctrl") == "prior_score":  # EulerDDS, solver/oc.py:288-306
                loc, var = self.sde.marginal_params(self.sde.terminal_t, self.prior.loc, var_init=self.prior.scale**2)
                self.reference_ctrl = lambda t, x: self.sde.diff(t, x) * self.prior.score(x)
            else:  # PIS, solver/oc.py:189-191
                loc, var = self.sde.marginal_params(self.sde.terminal_t, self.prior.loc)
            self.second = Density(dict(kind="diag_gauss", dim=self.dim), dict(loc=loc, scale=var.sqrt()))
        else:  # DDS: reference = prior (solver/oc.py:243)
            self.second = self.prior

    def terminal(self, x):  # solver/oc.py:48-54
        return clip(self.target.unnorm_log_prob(x), self.clip_target)

    def grid(self) -> Tensor:
        g = self.meta["grid"]
        return timesteps(g["start"], g["end"], steps=g["steps"], rescale_t=g.get("rescale_t"))

    # -- the three loops ------------------------------------------------------------------------
    def simulate(self, ts: Tensor, x: Tensor, noise: Tensor | None = None, *, train: bool = False,
                 compute_ito_int: bool = False, change_sde_ctrl: bool = False, return_traj: bool = False,
                 method: str | None = None, div_noise: Tensor | None = None):
        """Returns (x_T, rnd, xs|None).  `noise[i]` replaces the i-th `randn_like(x)` draw when given."""
        kind, sde, ctrl = self.kind, self.sde, self.ctrl
        method = method or self.method
        draw = (lambda i, x: noise[i]) if noise is not None else (lambda i, x: torch.randn_like(x))
        if kind == "time_reversal" and not (train and method in ("kl", "kl_ito")):
            rnd = self.second.log_prob(x)
        else:
            rnd = 0.0
        xs = [x] if return_traj else None
        for i, (s, t) in enumerate(zip(ts[:-1], ts[1:])):
            u = ctrl(s, x)
            u_sde = u.detach() if change_sde_ctrl else u
            dt = t - s
            if kind == "exponential":  # losses/oc.py:416-446
                if change_sde_ctrl:
                    cost = (u * (u_sde - 0.5 * u)).sum(dim=-1, keepdim=True)
                else:
                    cost = 0.5 * (u**2).sum(dim=-1, keepdim=True)
                beta_k = torch.clip(self.alpha * dt.sqrt(), 0, 1)
                alpha_k = torch.sqrt(1.0 - beta_k**2)
                rnd += beta_k**2 * self.sigma**2 * cost
                xi = draw(i, x)
                x = x * alpha_k + (beta_k**2) * (self.sigma**2) * u_sde + self.sigma * beta_k * xi
                if compute_ito_int:
                    rnd += (self.sigma * u * xi * beta_k).sum(dim=-1, keepdim=True)
            else:
                sig = sde.diff(s, x)
                if kind == "reference_sde" and self.reference_ctrl is not None:  # losses/oc.py:311-317
                    r = self.reference_ctrl(s, x)
                    g_minus, g_plus = u - r, r + u
                elif kind == "time_reversal" and self.inference_ctrl is not None:  # losses/oc.py:189-202
                    div, v = compute_divx(self.inference_ctrl, s, x, create_graph=train,
                                          noise_type=self.div_estimator if train else None,
                                          probe=None if div_noise is None else div_noise[i])
                    rnd = rnd + sig * div * dt
                    g_plus, g_minus = u + v, u - v
                else:
                    g_minus = g_plus = u
                if kind == "time_reversal":  # losses/oc.py:204-211
                    if change_sde_ctrl:
                        rnd += (g_plus * (u_sde - 0.5 * g_minus)).sum(dim=-1, keepdim=True) * dt
                    else:
                        rnd += 0.5 * (g_plus**2).sum(dim=-1, keepdim=True) * dt
                    if not train:
                        rnd -= sde.drift_div_int(s, t, x)
                    ito_ctrl = g_plus
                else:  # losses/oc.py:319-323
                    if change_sde_ctrl:
                        rnd += (g_minus * (u_sde - 0.5 * g_plus)).sum(dim=-1, keepdim=True) * dt
                    else:
                        rnd += 0.5 * (g_minus**2).sum(dim=-1, keepdim=True) * dt
                    ito_ctrl = g_minus
                db = draw(i, x) * dt.sqrt()
                x = x + (sde.drift(s, x) + sig * u_sde) * dt + sig * db
                if compute_ito_int:
                    rnd += (ito_ctrl * db).sum(dim=-1, keepdim=True)
            if return_traj:
                xs.append(x)
        if kind == "time_reversal":
            rnd -= self.terminal(x)
        else:
            rnd += self.second.log_prob(x) - self.terminal(x)
        assert rnd.shape == (x.shape[0], 1)
        return x, rnd, (torch.stack(xs) if return_traj else None)

    # -- loss.eval(...) (losses/oc.py:258-278, 371-391, 485-505) ----------------------------------
    @torch.no_grad()
    def eval(self, ts, x, noise=None, compute_weights=True, return_traj=False) -> dict:
        xT, rnd, xs = self.simulate(ts, x, noise, train=False, compute_ito_int=compute_weights,
                                    change_sde_ctrl=False, return_traj=return_traj)
        out = compute_results(rnd, compute_weights)
        out.update(samples=xT, rnd=rnd, xs=xs)
        return out

    # -- loss(...) (losses/oc.py:232-256, 345-369, 459-483) ---------------------------------------
    def train_loss(self, ts, x, noise=None, method=None, traj_per_sample: int = 1, div_noise=None):
        method = method or self.method
        if traj_per_sample != 1:
            x = x.repeat(traj_per_sample, 1, 1).reshape(-1, x.shape[-1])
        xT, rnd, _ = self.simulate(ts, x, noise, train=True, compute_ito_int=method != "kl",
                                   change_sde_ctrl=method in ("lv", "lv_traj"), method=method, div_noise=div_noise)
        loss, n_filtered = compute_loss(rnd, method, self.max_rnd, traj_per_sample)
        return loss, n_filtered, rnd, xT


def problem_from_fixture(fx) -> tuple["Problem", dict]:
    """`fx` = a loaded tests/golden/*.npz.  Returns (Problem, params-as-leaf-tensors)."""
    import json

    meta = json.loads(bytes(fx["meta"]).decode())
    params = {k[len("param/"):]: torch.from_numpy(fx[k].copy()) for k in fx.files if k.startswith("param/")}
    params_inf = {k[len("param_inf/"):]: torch.from_numpy(fx[k].copy()) for k in fx.files if k.startswith("param_inf/")}
    tt = None
    if meta["target"]["kind"] == "gmm":
        tt = {k: torch.from_numpy(fx["target/" + k].copy()) for k in ("loc", "scale", "mixture_weights")}
    elif meta["target"]["kind"] == "nice":
        tt = {k[len("target/"):]: torch.from_numpy(fx[k].copy()) for k in fx.files if k.startswith("target/")}
    return Problem(meta, params, tt, params_inf or None), params


# ------------------------------------------------------------------------------------------------
# plain Euler-Maruyama integrator (eq/integrator.py) for LangevinSDE / OU / ControlledSDE
# ------------------------------------------------------------------------------------------------
def interpolate(ts: Tensor, s: Tensor, t: Tensor, xs: Tensor, xt: Tensor, eps: float = 1e-8) -> Tensor:
    """eq/integrator.py:66-77."""
    ind = torch.searchsorted(ts, t + eps, side="right")
    t_eval = ts[:ind]
    assert (s <= t_eval).all() and (t_eval <= t + eps).all()
    return torch.lerp(xs, xt, (t_eval.view(-1, 1, 1) - s) / (t - s))


def euler_integrate(drift: Callable, diff: Callable, ts: Tensor, x_init: Tensor, timesteps: Tensor,
                    noise: Tensor | None = None, eps: float = 1e-8) -> Tensor:
    """eq/integrator.py:93-127.  `noise[i]` replaces the i-th `torch.randn(*xs.shape)` draw."""
    ts_count, xs_out, xs = 0, [], x_init
    for i, (s, t) in enumerate(zip(timesteps[:-1], timesteps[1:])):
        z = noise[i] if noise is not None else torch.randn(*xs.shape)
        dw = z * torch.sqrt(t - s)
        xt = xs + drift(s, xs) * (t - s) + diff(s, xs) * dw
        if ts[ts_count] <= t + eps:
            xs_out.append(interpolate(ts[ts_count:], s, t, xs, xt, eps=eps))
            ts_count += xs_out[-1].shape[0]
        xs = xt
    xs_out = torch.cat(xs_out)
    assert ts_count == xs_out.shape[0]
    return xs_out


def integration_case(meta: dict, params: dict, target_tensors: dict | None = None):
    """(drift, diff) callables of the SDE described by an `int_*` fixture's meta:
    meta["integrate"]["kind"] == "langevin":   eq/sdes.py:38-65  (diff_coeff, clip_score)
                               == "controlled": eq/sdes.py:272-305 over Sde(meta["sde"]) (generative or not);
                                  ctrl = Ctrl(meta["ctrl"]) built on the GENERATIVE twin of the sde (solver/oc.py:134-140),
                                  or None."""
    ispec = meta["integrate"]
    dim = meta["target"]["dim"]
    target = Density(dict(meta["target"]), target_tensors)
    if ispec["kind"] == "langevin":
        diff_coeff = torch.tensor(ispec["diff_coeff"], dtype=torch.float)
        drift = lambda t, x: clip(target.score(x) * diff_coeff**2 / 2.0, ispec.get("clip_score"))
        return drift, (lambda t, x: diff_coeff)
    sde = Sde(meta["sde"])
    ctrl = None
    if meta.get("ctrl"):
        prior = Density(dict(meta["prior"])) if meta.get("prior") else None
        ctrl = Ctrl(meta["ctrl"], meta["net"], params, Sde(dict(meta["sde"], generative=True)), prior, target)

    def drift(t, x):  # ControlledSDE.f_and_g (sdes.py:293-305); a bare OU has ctrl None
        out = sde.drift(t, x)
        if ctrl is not None:
            tc = t if sde.generative else sde.terminal_t - t
            out = out + sde.diff(t, x) * ctrl(tc, x)
        return out

    return drift, sde.diff

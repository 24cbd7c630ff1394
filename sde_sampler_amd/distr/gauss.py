"""Gaussian-family targets/priors (host side): GMM, Gauss, IsotropicGauss with the reference's constructor
arguments (sde_sampler/distr/gauss.py: gmm_params 14-63, GMM 66-155, Gauss 158-183, IsotropicGauss 186-242)."""
from __future__ import annotations

import logging
import math
from numbers import Number

import torch
from torch import distributions as D

from .base import Distribution


def gmm_params(name: str = "heart", dim: int = 2):
    """Named mixtures of the reference (means [K,2], one shared scale, uniform weights)."""
    if name == "heart":
        pts = [[-0.5, -0.25], [0.0, -1], [0.5, -0.25], [-1.0, 0.5], [-0.5, 1.0], [0.0, 0.5], [0.5, 1.0], [1.0, 0.5]]
        loc, factor = 1.5 * torch.tensor(pts), 1 / 8
    elif name == "dist":
        loc, factor = torch.tensor([[0.0, 0.0], [2, 0.0], [0.0, 3.0], [-4, 0.0], [0.0, -5]]), math.sqrt(0.2)
    elif name in ("fab", "multi"):
        k = 40 if name == "fab" else 80
        gen = torch.Generator()
        gen.manual_seed(42)
        loc = (torch.rand((k, 2), generator=gen) - 0.5) * 2 * k
        factor = torch.nn.functional.softplus(torch.tensor(1.0))
    elif name == "grid":
        axis = torch.linspace(-5, 5, 3)
        loc, factor = torch.cartesian_prod(axis, axis), math.sqrt(0.3)
    elif name == "circle":
        ang = 2 * torch.pi * torch.arange(1, 9) / 8
        loc, factor = torch.stack([4.0 * ang.cos(), 4.0 * ang.sin()], dim=1), math.sqrt(0.3)
    else:
        raise ValueError("Unknown mode for the Gaussian mixture.")
    if dim > 2:  # the reference pads with exactly 8 rows (only the 8-mode mixtures extend to dim > 2)
        loc = torch.cat([loc, torch.zeros(8, dim - 2)], dim=1)
    return loc, factor * torch.ones_like(loc), torch.ones(loc.shape[0])


class GMM(Distribution):
    def __init__(self, dim: int = 2, loc=None, scale=None, mixture_weights=None, n_reference_samples: int = int(1e7),
                 name: str | None = None, log_norm_const: float = 0.0, domain_scale: float = 5,
                 domain_tol: float | None = 1e-5, **kwargs):
        super().__init__(dim=dim, log_norm_const=log_norm_const, n_reference_samples=n_reference_samples, **kwargs)
        if name is not None:
            if any(t is not None for t in (loc, scale, mixture_weights)):
                logging.warning("Ignoring loc, scale, and mixture weights since name is specified.")
            loc, scale, mixture_weights = gmm_params(name, dim=dim)
        k = loc.shape[0]
        if not loc.shape == scale.shape == (k, self.dim):
            raise ValueError("Shape missmatch between loc and scale.")
        if mixture_weights is None and k > 1:
            raise ValueError("Require mixture weights.")
        if mixture_weights is not None and mixture_weights.shape != (k,):
            raise ValueError("Shape missmatch for the mixture weights.")
        self.register_buffer("loc", loc, persistent=False)
        self.register_buffer("scale", scale, persistent=False)
        self.register_buffer("mixture_weights", mixture_weights, persistent=False)
        self._refresh()
        if self.domain is None:
            half = domain_scale * self.scale.max(dim=0).values
            lo, hi = self.loc.min(dim=0).values - half, self.loc.max(dim=0).values + half
            self.set_domain(torch.stack([lo, hi], dim=-1))
        if domain_tol is not None and (self.pdf(self.domain.T) > domain_tol).any():
            raise ValueError("domain does not satisfy tolerance at the boundary.")

    def _refresh(self):
        if not hasattr(self, "loc") or self.loc is None:
            return
        if self.mixture_weights is None:
            self.distr = D.Independent(D.Normal(self.loc.squeeze(0), self.scale.squeeze(0)), 1)
        else:
            comps = D.Independent(D.Normal(self.loc, self.scale), 1)
            self.distr = D.MixtureSameFamily(D.Categorical(self.mixture_weights), comps)

    # kept for source compatibility with code that calls the reference's private hook
    _initialize_distr = _refresh

    @property
    def stddevs(self):
        return self.distr.variance.sqrt()

    @stddevs.setter
    def stddevs(self, value):
        pass

    def unnorm_log_prob(self, x):
        return self.distr.log_prob(x).unsqueeze(-1) + self.log_norm_const

    def marginal_distr(self, dim=0):
        if self.mixture_weights is None:
            return D.Normal(self.loc[0, dim], self.scale[0, dim])
        return D.MixtureSameFamily(D.Categorical(self.mixture_weights), D.Normal(self.loc[:, dim], self.scale[:, dim]))

    def marginal(self, x, dim=0):
        return self.marginal_distr(dim=dim).log_prob(x).exp()

    def sample(self, shape: tuple | None = None):
        return self.distr.sample(torch.Size(shape or ()))


class Gauss(GMM):
    def __init__(self, dim: int = 1, loc=0.0, scale=1.0, **kwargs):
        super().__init__(dim=dim, loc=self._as_row(loc, dim), scale=self._as_row(scale, dim), **kwargs)

    @staticmethod
    def _as_row(param, dim):
        if not isinstance(param, torch.Tensor):
            param = torch.tensor(param, dtype=torch.float)
        param = torch.atleast_2d(param)
        return param.repeat(1, dim) if param.numel() == 1 else param

    @property
    def stddevs(self):
        return self.scale.squeeze(0)

    @stddevs.setter
    def stddevs(self, value):
        pass

    def score(self, x, *args, **kwargs):
        return (self.loc - x) / self.scale**2


class IsotropicGauss(Gauss):
    """N(loc, scale^2 I); typically the prior.  `truncate_quartile` truncates *sampling* only."""

    def __init__(self, dim: int = 1, loc: float = 0.0, scale: float = 1.0, truncate_quartile: float | None = None,
                 **kwargs):
        super().__init__(dim=dim, loc=loc, scale=scale, **kwargs)
        assert torch.allclose(self.loc, self.loc[0, 0]) and torch.allclose(self.scale, self.scale[0, 0])
        if truncate_quartile is not None:
            q = torch.tensor([truncate_quartile / 2, 1 - truncate_quartile / 2], device=self.domain.device)
            truncate_quartile = self.marginal_distr().icdf(q).tolist()
        self.truncate_quartile = truncate_quartile

    def unnorm_log_prob(self, x):
        var = self.scale[0, 0] ** 2
        const = -0.5 * self.dim * (2.0 * math.pi * var).log() + self.log_norm_const
        return const - 0.5 * ((x - self.loc[0, 0]) ** 2).sum(dim=-1, keepdim=True) / var

    def score(self, x, *args, **kwargs):
        return (self.loc[0, 0] - x) / self.scale[0, 0] ** 2

    def marginal(self, x, **kwargs):
        return self.marginal_distr().log_prob(x).exp()

    def sample(self, shape: tuple | None = None):
        shape = shape or ()
        mu, sd = self.loc[0, 0], self.scale[0, 0]
        if self.truncate_quartile is None:
            return mu + sd * torch.randn(*shape, self.dim, device=self.domain.device)
        out = torch.empty(*shape, self.dim, device=self.domain.device)
        lo, hi = self.truncate_quartile
        return torch.nn.init.trunc_normal_(out, mean=mu, std=sd, a=lo, b=hi)

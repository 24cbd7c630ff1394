"""Double-well targets (reference: sde_sampler/distr/double_well.py, DoubleWell 14-100, MultiWell 103-193)."""
from __future__ import annotations

import math

import torch

from .base import Distribution, rejection_sampling
from .gauss import GMM, IsotropicGauss


class DoubleWell(Distribution):
    """rho(x) = exp(-((x - shift)^2 - separation)^2), one-dimensional."""

    def __init__(self, dim: int = 1, separation: float = 2.0, shift: float = 0.0, grid_points: int = 2001,
                 rejection_sampling_scaling: float = 3.0, domain_delta: float = 2.5, **kwargs):
        if dim != 1:
            raise ValueError("`dim` needs to be `1`. Consider using `MultiWell`.")
        super().__init__(dim=1, grid_points=grid_points, **kwargs)
        self.rejection_sampling_scaling = rejection_sampling_scaling
        self.register_buffer("separation", torch.tensor(separation), persistent=False)
        self.register_buffer("shift", torch.tensor(shift), persistent=False)
        if self.domain is None:
            half = self.separation.sqrt() + domain_delta
            self.set_domain(self.shift + half * torch.tensor([[-1.0, 1.0]]))

    def unnorm_log_prob(self, x):
        y = x - self.shift
        return -((y**2 - self.separation) ** 2)

    def score(self, x, *args, **kwargs):
        y = x - self.shift
        return -4.0 * (y**2 - self.separation) * y

    def marginal(self, x, *args, **kwargs):
        return self.pdf(x)

    def get_proposal_distr(self):
        dev = self.domain.device
        root = self.separation.sqrt()
        loc = self.shift + root * torch.tensor([[-1.0], [1.0]], device=dev)
        proposal = GMM(dim=1, loc=loc, scale=torch.ones(2, 1, device=dev) / root,
                       mixture_weights=torch.ones(2, device=dev), domain_tol=None)
        proposal.to(dev)
        return proposal

    def sample(self, shape: tuple | None = None):
        return rejection_sampling(shape=shape or (), target=self, proposal=self.get_proposal_distr(),
                                  scaling=self.rejection_sampling_scaling)


class MultiWell(Distribution):
    """`n_double_wells` independent double wells followed by unit Gaussians centred at `shift`."""

    def __init__(self, dim: int = 2, n_double_wells: int = 1, separation: float = 2.0, shift: float = 0.0,
                 domain_dw_delta: float = 2.5, domain_gauss_scale: float = 5.0, **kwargs):
        super().__init__(dim=dim, **kwargs)
        if n_double_wells > dim or n_double_wells == 0:
            raise ValueError(f"Please specify between 1 and {dim} double wells.")
        self.separation = separation
        self.n_double_wells = n_double_wells
        self.n_gauss = dim - n_double_wells
        self.double_well = DoubleWell(separation=separation, shift=shift, domain_delta=domain_dw_delta)
        domain = self.double_well.domain.repeat(n_double_wells, 1)
        self.gauss = None
        if self.n_gauss > 0:
            self.gauss = IsotropicGauss(dim=self.n_gauss, loc=shift, domain_scale=domain_gauss_scale,
                                        log_norm_const=0.5 * math.log(2.0 * math.pi) * self.n_gauss)
            domain = torch.cat([domain, self.gauss.domain])
        self.set_domain(domain)

    def compute_stats(self):
        dw, n = self.double_well, self.n_double_wells
        dw.compute_stats()
        self.log_norm_const = dw.log_norm_const * n
        self.expectations = {k: v * n for k, v in dw.expectations.items()}
        self.stddevs = torch.cat([dw.stddevs] * n)
        if self.gauss is not None:
            self.gauss.compute_stats()
            self.log_norm_const += self.gauss.log_norm_const
            for k in self.expectations:
                self.expectations[k] += self.gauss.expectations[k]
            self.stddevs = torch.cat([self.stddevs, self.gauss.stddevs])

    def unnorm_log_prob(self, x):
        n = self.n_double_wells
        out = self.double_well.unnorm_log_prob(x[:, :n]).sum(dim=-1, keepdim=True)
        if self.gauss is not None:
            out = out + self.gauss.unnorm_log_prob(x[:, n:])
        return out

    def score(self, x, *args, **kwargs):
        n = self.n_double_wells
        parts = [self.double_well.score(x[:, :n])]
        if self.gauss is not None:
            parts.append(self.gauss.score(x[:, n:]))
        return torch.cat(parts, dim=-1)

    def marginal(self, x, dim: int = 0):
        return self.double_well.marginal(x) if dim < self.n_double_wells else self.gauss.marginal(x)

    def sample(self, shape: tuple | None = None):
        shape = shape or ()
        out = self.double_well.sample(shape + (self.n_double_wells,)).squeeze(-1)
        if self.gauss is not None:
            out = torch.cat([out, self.gauss.sample(shape)], dim=-1)
        return out

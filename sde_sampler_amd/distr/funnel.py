"""Neal's funnel (reference: sde_sampler/distr/funnel.py:11-96): x_0 ~ N(0, variance), x_i | x_0 ~ N(0, e^{x_0})."""
from __future__ import annotations

import math

import torch

from .base import Distribution
from .gauss import IsotropicGauss


class Funnel(Distribution):
    def __init__(self, dim: int = 10, variance: float | None = None, n_reference_samples: int = int(1e7),
                 log_norm_const: float = 0.0, domain_first_scale: float = 5.0, domain_other_scale: float = 5.0,
                 domain_tol: float | None = 1e-5, **kwargs):
        super().__init__(dim=dim, log_norm_const=log_norm_const, n_reference_samples=n_reference_samples, **kwargs)
        self.variance = dim - 1 if variance is None else variance
        self.distr_first = IsotropicGauss(dim=1, scale=math.sqrt(self.variance), domain_scale=domain_first_scale,
                                          domain_tol=domain_tol)
        if self.domain is None:
            first = self.distr_first.domain
            other = first.sgn() * (first.abs() / domain_other_scale).exp()
            self.set_domain(torch.cat([first, other.repeat(dim - 1, 1)]))
        if domain_tol is not None and (self.pdf(self.domain.T) > domain_tol).any():
            raise ValueError("Domain does not satisfy tolerance at the boundary.")

    @staticmethod
    def log_prob_other(x_other, x_first):
        n = x_other.shape[-1]
        return -n * (x_first + math.log(2.0 * math.pi)) / 2.0 - 0.5 * (x_other**2).sum(-1, keepdim=True) * (-x_first).exp()

    def unnorm_log_prob(self, x):
        head, tail = x[:, :1], x[:, 1:]
        return self.distr_first.unnorm_log_prob(head) + Funnel.log_prob_other(tail, head) + self.log_norm_const

    def score(self, x, *args, **kwargs):
        head, tail = x[:, :1], x[:, 1:]
        inv_var = (-head).exp()
        d_head = self.distr_first.score(head) - 0.5 * tail.shape[-1] + 0.5 * (tail**2).sum(-1, keepdim=True) * inv_var
        return torch.cat([d_head, -tail * inv_var], dim=-1)

    def marginal(self, x, dim=0):
        if dim == 0:
            return self.distr_first.marginal(x)
        first = self.distr_first.sample((self.n_reference_samples, 1))
        return self.log_prob_other(x, first).exp().mean(axis=0)

    def sample(self, shape: tuple | None = None):
        shape = shape or ()
        head = self.distr_first.sample(shape)
        tail = torch.randn(*shape, self.dim - 1, device=head.device) * (0.5 * head).exp()
        return torch.cat((head, tail), dim=-1)

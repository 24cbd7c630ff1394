"""Dirac prior of PIS (reference: sde_sampler/distr/delta.py:8-28): a very narrow Gaussian whose `sample`
returns the location itself."""
from __future__ import annotations

from .gauss import Gauss


class Delta(Gauss):
    def __init__(self, dim: int = 1, loc=0.0, approx_scale: float = 1e-3, domain_scale: float = 10, **kwargs):
        super().__init__(dim=dim, loc=loc, scale=approx_scale, domain_scale=domain_scale, **kwargs)

    def sample(self, shape: tuple | None = None):
        return self.loc.repeat(*(shape or ()), 1)

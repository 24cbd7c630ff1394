"""Distribution base class (host side): same public surface as the reference's sde_sampler/distr/base.py
(`unnorm_log_prob`, `log_prob`, `pdf`, `score`, `domain`, `log_norm_const`, `compute_stats`) for the targets
and priors the HIP engine fuses.  Reference statistics via torchquad are replaced by a small 1-D quadrature."""
from __future__ import annotations

import logging
import math

import torch

EXPECTATION_FNS = {
    "square": lambda x: (x**2).sum(dim=-1, keepdims=True),
    "abs": lambda x: x.abs().sum(dim=-1, keepdims=True),
    "sum": lambda x: x.sum(dim=-1, keepdims=True),
    "square_minus_sum": lambda x: (x**2 - x).sum(dim=-1, keepdims=True),
}


class Distribution(torch.nn.Module):
    def __init__(self, dim: int, log_norm_const: float | None = None, domain=None,
                 n_reference_samples: int | None = None, grid_points: int | None = None):
        super().__init__()
        self.dim = dim
        self.log_norm_const = log_norm_const
        self.n_reference_samples = n_reference_samples
        self.grid_points = grid_points
        self.expectations: dict = {}
        self.register_buffer("stddevs", None, persistent=False)
        self.set_domain(domain)

    def set_domain(self, d=None):
        if d is not None:
            d = torch.as_tensor(d, dtype=torch.float)
            if d.ndim == 0:
                d = torch.stack([-d, d], dim=-1)
            if d.ndim == 1:
                d = d.unsqueeze(0)
            if d.shape == (1, 2):
                d = d.repeat(self.dim, 1)
            assert d.shape == (self.dim, 2)
        self.register_buffer("domain", d, persistent=False)

    # --- densities ---------------------------------------------------------------------------------------
    def unnorm_log_prob(self, x):
        raise NotImplementedError

    def log_prob(self, x):
        if self.log_norm_const is None:
            raise NotImplementedError
        return self.unnorm_log_prob(x) - self.log_norm_const

    def pdf(self, x):
        return self.log_prob(x).exp()

    def unnorm_pdf(self, x):
        return self.unnorm_log_prob(x).exp()

    def forward(self, x):
        return self.unnorm_log_prob(x)

    def score(self, x, create_graph: bool = False):
        """Generic score by automatic differentiation (subclasses override with closed forms)."""
        had_grad = x.requires_grad
        x.requires_grad_(True)
        with torch.enable_grad():
            total = self.unnorm_log_prob(x).sum()
            (grad,) = torch.autograd.grad(total, x, create_graph=create_graph)
        x.requires_grad_(had_grad)
        return grad

    # --- reference statistics ------------------------------------------------------------------------------
    def _refresh(self):
        """Hook: rebuild cached torch.distributions objects after a device / dtype move."""

    def _apply(self, fn, *args, **kwargs):
        out = super()._apply(fn, *args, **kwargs)
        self._refresh()
        return out

    @torch.no_grad()
    def compute_stats(self):
        if hasattr(self, "sample") and self.n_reference_samples is not None:
            samples = self.sample((self.n_reference_samples,))
            for name, fn in EXPECTATION_FNS.items():
                self.expectations.setdefault(name, fn(samples).mean().item())
            if self.stddevs is None:
                self.stddevs = samples.std(dim=0)
        elif self.grid_points is not None and self.domain is not None and self.dim == 1:
            self._compute_stats_quadrature()
        else:
            logging.warning("Cannot compute statistics for distribution `%s`", type(self).__name__)

    def _compute_stats_quadrature(self):
        """Composite Boole rule on `grid_points` nodes over the 1-D domain (the reference uses torchquad's
        Boole integrator with the same node count, distr/base.py:62-85)."""
        n = self.grid_points - (self.grid_points - 1) % 4
        lo, hi = self.domain[0, 0].double(), self.domain[0, 1].double()
        xs = torch.linspace(lo, hi, n, dtype=torch.double, device=self.domain.device)
        h = (hi - lo) / (n - 1)
        w = torch.tensor([7.0, 32.0, 12.0, 32.0], dtype=torch.double, device=xs.device).repeat((n - 1) // 4)
        w = torch.cat([w, w.new_tensor([7.0])])
        w[4:-1:4] = 14.0
        w = w * (2.0 * h / 45.0)

        def integrate(fn):
            return (w * fn(xs.unsqueeze(-1).float()).double().squeeze(-1)).sum()

        if self.log_norm_const is None:
            self.log_norm_const = math.log(integrate(self.unnorm_pdf).item())
        for name, fn in EXPECTATION_FNS.items():
            if name not in self.expectations:
                self.expectations[name] = integrate(lambda x: fn(x) * self.pdf(x)).item()
        if self.stddevs is None:
            mean = integrate(lambda x: x * self.pdf(x))
            var = integrate(lambda x: (x - mean.float()) ** 2 * self.pdf(x))
            self.stddevs = torch.atleast_1d(var.sqrt().float())


def sample_uniform(domain: torch.Tensor, batchsize: int = 1) -> torch.Tensor:
    lo, hi = domain[:, 0], domain[:, 1]
    return lo + torch.rand(batchsize, domain.shape[0], device=domain.device) * (hi - lo)


def rejection_sampling(shape: tuple, proposal: Distribution, target: Distribution, scaling: float) -> torch.Tensor:
    n = math.prod(shape)
    cand = proposal.sample((n * math.ceil(scaling) * 10,))
    thresh = torch.rand(cand.shape[0], 1, device=cand.device) * scaling * proposal.pdf(cand)
    kept = cand[thresh < target.pdf(cand)]
    if kept.shape[0] >= n:
        return kept[:n].reshape(*shape, -1)
    more = rejection_sampling((n - kept.shape[0],), proposal, target, scaling)
    return torch.concat([kept.reshape(*shape, -1), more])

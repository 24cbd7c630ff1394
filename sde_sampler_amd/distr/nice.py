"""The NICE flow target of BASELINE configs[4] (reference: sde_sampler/distr/nice.py -- StandardLogistic 17-40, Coupling 43-97,
Scaling 100-120, NiceModel 123-231, Nice 233-298): a normalising flow of additive couplings on 14 x 14 images, d = 196.

Host side: the module tree carries the REFERENCE's parameter names (`coupling.<i>.in_block.0.weight`, `coupling.<i>.mid_block.<l>.0.weight`,
`coupling.<i>.out_block.weight`, `scaling.scale`), so a checkpoint written by the reference's scripts/train_nice.py loads with
`load_state_dict` and the engine describes either package's object to libsdeh by attribute (`engine.describe_nice`).  On a GPU
`Nice.unnorm_log_prob` / `Nice.score` run on the HIP kernels of csrc/sdeh_nice.hip (`sdeh_nice_eval`: every Linear on the fp32 matrix
pipe, the score as the reverse pass through the couplings -- the reference differentiates with autograd, distr/base.py:130-137); on
the CPU the plain torch forms below serve host-side tooling (sampling reference points, tests).  The hot path never goes through this
module's torch code: the trajectory engine evaluates the score between its step segments itself (engine.py, `sdeh_simulate_fwd_steps`).

Not reproduced: `plots` (PIL / torchvision image grids: evaluation cosmetics) -- `mean` is kept for callers that
want to un-centre samples, resized with torch's own antialiased interpolation.
"""
from __future__ import annotations

import math
from pathlib import Path

import torch
from torch import nn

from .base import Distribution


class StandardLogistic(torch.distributions.Distribution):
    """nice.py:17-40."""

    def __init__(self):
        super().__init__(validate_args=False)

    def log_prob(self, x):
        return -(nn.functional.softplus(x) + nn.functional.softplus(-x))

    def sample(self, size, eps: float = 1e-20):
        z = torch.distributions.Uniform(eps, 1.0 - eps).sample(size)
        return torch.log(z) - torch.log(1.0 - z)


class Coupling(nn.Module):
    """Additive coupling (nice.py:43-97): x viewed as [B, W / 2, 2]; one half feeds an MLP whose output shifts the other."""

    def __init__(self, in_out_dim: int, mid_dim: int, hidden: int, mask_config):
        super().__init__()
        self.mask_config = mask_config
        self.in_block = nn.Sequential(nn.Linear(in_out_dim // 2, mid_dim), nn.ReLU())
        self.mid_block = nn.ModuleList([nn.Sequential(nn.Linear(mid_dim, mid_dim), nn.ReLU()) for _ in range(hidden - 1)])
        self.out_block = nn.Linear(mid_dim, in_out_dim // 2)

    def forward(self, x, reverse: bool = False):
        batch, width = x.shape
        x = x.reshape(batch, width // 2, 2)
        if self.mask_config:
            on, off = x[:, :, 0], x[:, :, 1]
        else:
            off, on = x[:, :, 0], x[:, :, 1]
        h = self.in_block(off)
        for block in self.mid_block:
            h = block(h)
        shift = self.out_block(h)
        on = on - shift if reverse else on + shift
        x = torch.stack((on, off), dim=2) if self.mask_config else torch.stack((off, on), dim=2)
        return x.reshape(batch, width)


class Scaling(nn.Module):
    """nice.py:100-120."""

    def __init__(self, dim: int):
        super().__init__()
        self.scale = nn.Parameter(torch.zeros((1, dim)), requires_grad=True)

    def forward(self, x, reverse: bool = False):
        log_det = torch.sum(self.scale)
        return (x * torch.exp(-self.scale) if reverse else x * torch.exp(self.scale)), log_det


class NiceModel(nn.Module):
    """nice.py:123-231."""

    def __init__(self, prior, coupling: int, in_out_dim: int, mid_dim: int, hidden: int, mask_config):
        super().__init__()
        self.prior = prior
        self.in_out_dim = in_out_dim
        self.coupling = nn.ModuleList([Coupling(in_out_dim=in_out_dim, mid_dim=mid_dim, hidden=hidden, mask_config=(mask_config + i) % 2)
                                       for i in range(coupling)])
        self.scaling = Scaling(in_out_dim)

    def g(self, z):
        x, _ = self.scaling(z, reverse=True)
        for layer in reversed(self.coupling):
            x = layer(x, reverse=True)
        return x

    def f(self, x):
        for layer in self.coupling:
            x = layer(x)
        return self.scaling(x)

    def log_prob(self, x):
        z, log_det = self.f(x)
        return torch.sum(self.prior.log_prob(z), dim=1) + log_det

    def sample(self, size: int):
        z = self.prior.sample((size, self.in_out_dim)).to(self.scaling.scale.device)
        return self.g(z)

    def forward(self, x):
        return self.log_prob(x)


class Nice(Distribution):
    """NICE trained on resized MNIST (nice.py:233-298).  `model` = a NiceModel (e.g. random-initialised: SURVEY.md 8d item 5), or
    `checkpoint` = a file written by scripts/train_nice.py ({"coupling", "mid_dim", "hidden", "mask_config", "model_state_dict"})."""

    def __init__(self, model: nn.Module | None = None, checkpoint: str | Path | None = None, mean_data_path: str | Path | None = None,
                 sample_chunk_size: int = 10000, dim: int = 196, log_norm_const: float = 0.0, n_reference_samples=int(1e6)):
        super().__init__(dim=dim, log_norm_const=log_norm_const, n_reference_samples=n_reference_samples)
        self.shape = (14, 14)
        if self.dim != math.prod(self.shape):
            raise ValueError(f"Dimension is {self.dim} but needs to be 196.")
        self.sample_chunk_size = sample_chunk_size
        mean = None
        if mean_data_path is not None:  # data/mnist_mean.pt: only used to un-centre samples for display
            mean = torch.load(mean_data_path).reshape((1, 1, 28, 28)).float()
            mean = nn.functional.interpolate(mean, size=self.shape, mode="bilinear", antialias=True).reshape((1, self.dim))
        self.register_buffer("mean", mean, persistent=False)
        self.model = model
        if self.model is None:
            if checkpoint is None:
                raise ValueError("Nice needs a `model` or a `checkpoint` (the reference's data/nice.pt is not shipped)")
            ckpt = torch.load(checkpoint)
            self.model = NiceModel(prior=StandardLogistic(), coupling=ckpt["coupling"], in_out_dim=196, mid_dim=ckpt["mid_dim"],
                                   hidden=ckpt["hidden"], mask_config=ckpt["mask_config"])
            self.model.load_state_dict(ckpt["model_state_dict"])
        self.model.eval()
        for p in self.model.parameters():
            p.requires_grad_(False)
        self._work: dict = {}

    # -- the HIP evaluation (GPU tensors) -------------------------------------------------------------------------------------
    def _hip_eval(self, x: torch.Tensor, want_score: bool, want_logp: bool):
        from sde_sampler_amd import engine as E

        return E.nice_eval(self, x, want_score=want_score, want_logp=want_logp, cache=self._work)

    def unnorm_log_prob(self, x: torch.Tensor) -> torch.Tensor:
        if x.is_cuda and not (torch.is_grad_enabled() and x.requires_grad):
            return self._hip_eval(x, False, True)[1].unsqueeze(-1)
        return self.model.log_prob(x).unsqueeze(-1) + self.log_norm_const

    def score(self, x: torch.Tensor, create_graph: bool = False) -> torch.Tensor:
        if x.is_cuda and not create_graph:
            return self._hip_eval(x.detach(), True, False)[0]
        return super().score(x, create_graph=create_graph)

    def sample(self, shape: tuple | None = None) -> torch.Tensor:
        shape = (1,) if shape is None else shape
        if len(shape) > 1:
            raise ValueError("Can only sample shapes (batch_size, dim).")
        size = shape[0]
        chunks, rest = divmod(size, self.sample_chunk_size)
        with torch.no_grad():
            out = [self.model.sample(self.sample_chunk_size) for _ in range(chunks)]
            if rest:
                out.append(self.model.sample(rest))
        out = torch.cat(out)
        assert out.shape == (size, self.dim)
        return out

"""Host helpers shared by the loss classes: the `Results` record and the time grids.

Mirrors the call contracts of the reference's sde_sampler/utils/common.py (Results: lines 9-13,
get_timesteps: 18-55, clip_and_log: 58-85) so that solver code written against the reference keeps working.
"""
from __future__ import annotations

import math
from typing import NamedTuple, Optional

import torch


class Results(NamedTuple):
    samples: object = {}
    weights: object = {}
    log_norm_const_preds: Optional[dict] = None
    expectation_preds: Optional[dict] = None
    ts: Optional[torch.Tensor] = None
    xs: Optional[torch.Tensor] = None
    metrics: dict = {}
    plots: dict = {}


def get_timesteps(start, end, dt=None, steps=None, rescale_t=None, device=None) -> torch.Tensor:
    """Time grid of the integrators: uniform, "quad" (sqrt-spaced) or the DDS "cosine" schedule.

    Note the cosine schedule returns ``steps + 2`` points (``steps + 1`` intervals), as in the reference.
    """
    if (steps is None) == (dt is None):
        raise ValueError("Exactly one of `dt` and `steps` should be defined.")
    n = int(math.ceil((end - start) / dt)) if steps is None else steps
    if rescale_t is None:
        return torch.linspace(start, end, steps=n + 1, device=device)
    if rescale_t == "quad":
        grid = torch.linspace(start, end.square(), steps=n + 1, device=device)
        return grid.sqrt().clip(max=end)
    if rescale_t == "cosine":
        frac = torch.linspace(start, end, n + 1, device=device) / end
        incr = torch.cos((frac + 0.008) / 1.008 * (0.5 * torch.pi)) ** 4
        incr /= incr.sum()
        incr *= end
        head = torch.tensor([start], device=device)
        return torch.concat((head, incr.cumsum(-1)))
    raise ValueError("Unkown timestep rescaling method.")


def clip_and_log(tensor: torch.Tensor, max_norm=None, name=None, t=None, log_dt: float = 0.2) -> torch.Tensor:
    """Symmetric clamp.  (The reference additionally logs max-abs values to wandb, which forces a device
    sync per call; logging is not part of this package.)"""
    return tensor if max_norm is None else tensor.clip(min=-1.0 * max_norm, max=max_norm)

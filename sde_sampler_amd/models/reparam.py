"""Control parametrisations (host side): same class names, constructor keywords and attributes as the
reference's sde_sampler/models/reparam.py (ClippedCtrl 13-36, ScoreCtrl 39-83, LerpCtrl 113-162,
LerpPriorCtrl 165-181, LerpTargetCtrl 184-200).  The attributes `base_model, score_model, clip_model, clip_score,
scale_score, detach_score, target_score, prior_score, sde, name` are what the HIP engine reads on every call
(the solver's MultiStepParams scheduler mutates the clip values in place, solver/base.py:586-597)."""
from __future__ import annotations

from typing import Callable

import torch
from torch.nn import Module

from sde_sampler_amd.utils.common import clip_and_log


class ClippedCtrl(Module):
    def __init__(self, base_model: Module, clip_model: float | None = None, name: str = "ctrl", **kwargs):
        super().__init__()
        self.base_model = base_model
        self.clip_model = clip_model
        self.name = name

    def clipped_base_model(self, t, x):
        return clip_and_log(self.base_model(t, x), max_norm=self.clip_model)

    def forward(self, t, x):
        return self.clipped_base_model(t, x)


class ScoreCtrl(ClippedCtrl):
    """u = clip(NN(t,x)) + scale_score * clip(grad log rho(x)) * clip(gamma(t))."""

    def __init__(self, *args, target_score: Callable, score_model: Module | None = None, detach_score: bool = True,
                 scale_score: float = 1.0, clip_score: float | None = None, **kwargs):
        super().__init__(*args, **kwargs)
        self.score_model = score_model
        self.target_score = target_score
        self.detach_score = detach_score
        self.scale_score = scale_score
        self.clip_score = clip_score

    def _maybe_detached(self, x):
        return x.detach() if self.detach_score else x

    def raw_score(self, t, x):
        x = self._maybe_detached(x)
        return self.target_score(x, create_graph=self.detach_score)

    def clipped_score_model(self, t, x):
        return clip_and_log(self.score_model(t, x), max_norm=self.clip_model)

    def score_term(self, t, x):
        term = self.scale_score * clip_and_log(self.raw_score(t, x), max_norm=self.clip_score)
        if self.score_model is not None:
            term = term * self.clipped_score_model(t, x)
        return term

    def forward(self, t, x):
        return self.clipped_base_model(t, x) + self.score_term(t, x)


class LerpCtrl(ScoreCtrl):
    """u = clip(NN) + sigma(t) * scale_score * clip(lerp(prior score, target score, t/T)) * clip(gamma(t))."""

    def __init__(self, *args, sde, prior_score: Callable, hard_constrain: bool = False, scale_lerp: float = 1.0,
                 **kwargs):
        super().__init__(*args, **kwargs)
        if sde.noise_type not in ("diagonal", "scalar"):
            raise ValueError(f"Invalid sde noise type {sde.noise_type}.")
        if hard_constrain:
            raise NotImplementedError("hard_constrain is broken in the reference (undefined terminal_t) and not reproduced")
        self.sde = sde
        self.prior_score = prior_score
        self.hard_constrain = hard_constrain
        self.scale_lerp = scale_lerp

    def raw_score(self, t, x):
        x = self._maybe_detached(x)
        target = self.target_score(x, create_graph=self.detach_score)
        return torch.lerp(self.prior_score(x), target, t / self.sde.terminal_t)

    def forward(self, t, x):
        return self.clipped_base_model(t, x) + self.sde.diff(t, x) * self.score_term(t, x)


class LerpPriorCtrl(LerpCtrl):
    def raw_score(self, t, x):
        x = self._maybe_detached(x)
        return (1.0 - t / self.sde.terminal_t) * self.prior_score(x)


class LerpTargetCtrl(LerpCtrl):
    def raw_score(self, t, x):
        x = self._maybe_detached(x)
        return t / self.sde.terminal_t * self.target_score(x)

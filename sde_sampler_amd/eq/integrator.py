"""`EulerIntegrator`: same constructor / `integrate` signature as the reference's sde_sampler/eq/integrator.py:80-127,
executed as ONE kernel launch by libsdeh.so (`sdeh_integrate`, include/sdeh.h) instead of a Python loop with
`drift`/`diff` callbacks, a `torch.cat` of per-step states and a post-hoc interpolation.

Fused SDE classes (recognised by class name, so the reference's own objects work too):
  * `LangevinSDE` whose `target_score` is the `.score` of a built-in distribution       (solver/langevin.py:20-48)
  * `VP` / `ConstOU` / `ScaledBM`, generative or not                                    (solver/oc.py:100-110)
  * `ControlledSDE(sde=<OU>, ctrl=None | ClippedCtrl | ScoreCtrl | Lerp*Ctrl)`           (solver/oc.py:130-143)
Anything else raises `SdehUnsupported`: there is no eager fallback in this package.
"""
from __future__ import annotations

import torch

from .. import _lib as L
from ..engine import TrajectoryEngine, _bound_owner, _Keep, _known_distribution, _mro_names, _unsupported
from ..utils.common import get_timesteps

_INF = float("inf")


class Integrator:
    def integrate(self, sde, ts, x_init, timesteps=None, bm=None):
        raise NotImplementedError


class EulerIntegrator(Integrator):
    def __init__(self, dt: float | None = 0.01, steps: int | None = None, rescale_t: str | None = None,
                 eps: float = 1e-8):
        self.dt = dt
        self.steps = steps
        self.rescale_t = rescale_t
        self.eps = eps
        self.engine = TrajectoryEngine()
        self.engine.stream_id = 1  # never the noise of a loss object (stream 0) evaluated in the same process
        self.row_offset = 0  # global index of x_init row 0 (rank * local batch for sharded runs)

    # ------------------------------------------------------------------------------------------------------
    def _problem(self, sde, dim: int, device, keep: _Keep):
        names = _mro_names(sde)
        if "LangevinSDE" in names:
            owner = _bound_owner(sde.target_score, "score")
            if owner is None or not _known_distribution(owner):
                raise _unsupported("LangevinSDE.target_score must be the `.score` of a built-in distribution")
            pr = self.engine.build_problem(loss_kind=L.LOSS_TIME_REVERSAL, generative_ctrl=None, sde=sde, flags=0,
                                           device=device, keep=keep, terminal_target=owner,
                                           allow_inference_sde=True, dim=dim)
            pr.clip_score = _INF if sde.clip_score is None else float(sde.clip_score)
            return pr, L.INT_LANGEVIN
        ctrl, base = None, sde
        if "ControlledSDE" in names:
            ctrl, base = sde.ctrl, sde.sde
        if "OU" not in _mro_names(base):
            raise _unsupported(f"EulerIntegrator: sde {type(sde).__name__} is not fused (LangevinSDE, OU processes and "
                               "ControlledSDE are)")
        pr = self.engine.build_problem(loss_kind=L.LOSS_TIME_REVERSAL, generative_ctrl=ctrl, sde=base, flags=0,
                                       device=device, keep=keep, allow_inference_sde=True, dim=dim)
        return pr, L.INT_CONTROLLED

    def integrate(self, sde, ts: torch.Tensor, x_init: torch.Tensor, timesteps: torch.Tensor | None = None,
                  bm=None, *, noise: torch.Tensor | None = None, seed: int | None = None) -> torch.Tensor:
        """Returns the states at the times `ts`, shape [len(ts), B, d].

        `noise` (keyword-only extension): [len(timesteps)-1, B, d] standard normals to consume instead of the in-kernel
        Philox stream -- the reference draws `torch.randn(*x.shape) * sqrt(t - s)` per step (integrator.py:115).
        `bm`: a callable Brownian motion `bm(s, t)` as in the reference; its increments are materialised once.
        """
        if timesteps is None:
            timesteps = get_timesteps(ts[0], ts[-1], dt=self.dt, steps=self.steps, rescale_t=self.rescale_t,
                                      device=ts.device)
        if bm is not None:
            if noise is not None:
                raise ValueError("pass either bm or noise")
            pairs = zip(timesteps[:-1], timesteps[1:])
            noise = torch.stack([bm(s, t) / torch.sqrt(t - s) for s, t in pairs])
        # the reference asserts that every output time is reached (integrator.py:125-126)
        if bool(ts[-1] > timesteps[-1] + self.eps) or bool(ts[0] < timesteps[0]):
            raise AssertionError("output times `ts` must lie inside the integration grid `timesteps`")
        keep = _Keep()
        pr, kind = self._problem(sde, x_init.shape[-1], x_init.device, keep)
        return self.engine.integrate(pr, kind, timesteps, ts, x_init, noise=noise, eps=self.eps, keep=keep,
                                     row_offset=self.row_offset, seed=seed)

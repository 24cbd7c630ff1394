"""SDE coefficient objects (host side).  Same class names / constructor arguments / methods as the reference's
sde_sampler/eq/sdes.py (OU 68-122, ConstOU 125-172, ScaledBM 175-188, VP 191-269) so that configs and solver
code can instantiate them unchanged; the HIP engine reads their buffers (see sde_sampler_amd/engine.py)."""
from __future__ import annotations

import torch
from torch.nn import Module


def _buf(module: Module, name: str, value: float):
    module.register_buffer(name, torch.tensor(value, dtype=torch.float), persistent=False)


class TorchSDE(Module):
    noise_type = "diagonal"
    sde_type = "ito"

    def __init__(self, terminal_t: float = 1.0):
        super().__init__()
        _buf(self, "terminal_t", terminal_t)

    def drift(self, t, x):
        raise NotImplementedError

    def diff(self, t, x):
        raise NotImplementedError

    def f(self, t, x):
        return self.drift(t, x).expand_as(x)

    def g(self, t, x):
        return self.diff(t, x).expand_as(x)


class OU(TorchSDE):
    """dX = sign * a(t) X dt + b(t) dW with scalar-in-time coefficients."""

    def __init__(self, generative: bool = True, **kwargs):
        super().__init__(**kwargs)
        self.generative = generative
        self.sign = 1.0 if generative else -1.0

    def drift_coeff_t(self, t):
        raise NotImplementedError

    def diff_coeff_t(self, t):
        raise NotImplementedError

    def int_drift_coeff_t(self, s, t):
        raise NotImplementedError

    def int_diff_coeff_sq_t(self, s, t):
        raise NotImplementedError

    def marginal_params(self, t, x_init, var_init=None):
        raise NotImplementedError

    def drift(self, t, x):
        return self.drift_coeff_t(t) * x

    def diff(self, t, x):
        return self.diff_coeff_t(t)

    def drift_div(self, t, x):
        return self.drift_coeff_t(t) * x.shape[-1]

    def drift_div_int(self, s, t, x):
        return self.int_drift_coeff_t(s, t) * x.shape[-1]

    def marginal_distr(self, t, x_init, var_init=None):
        from sde_sampler_amd.distr.gauss import Gauss

        loc, var = self.marginal_params(t, x_init, var_init=var_init)
        return Gauss(dim=x_init.shape[-1], loc=loc, scale=var.sqrt(), domain_tol=None)


class ConstOU(OU):
    def __init__(self, drift_coeff: float = 2.0, diff_coeff: float = 2.0, **kwargs):
        if drift_coeff < 0 or diff_coeff <= 0:
            raise ValueError("Choose non-negative drift_coeff and positive diff_coeff.")
        super().__init__(**kwargs)
        _buf(self, "drift_coeff", drift_coeff)
        _buf(self, "diff_coeff", diff_coeff)

    def drift_coeff_t(self, t):
        return self.sign * self.drift_coeff

    def diff_coeff_t(self, t):
        return self.diff_coeff

    def int_drift_coeff_t(self, s, t):
        return self.sign * self.drift_coeff * (t - s)

    def int_diff_coeff_sq_t(self, s, t):
        return self.diff_coeff**2 * (t - s)

    def marginal_params(self, t, x_init, var_init=None):
        a = self.sign * self.drift_coeff
        growth = torch.exp(a * t)
        var = -self.diff_coeff**2 / (2 * a) * (1 - torch.exp(2 * a * t))
        if var_init is not None:
            var = var + growth**2 * var_init
        return growth * x_init, var


class ScaledBM(ConstOU):
    def __init__(self, *args, **kwargs):
        super().__init__(*args, drift_coeff=0.0, **kwargs)

    def marginal_params(self, t, x_init, var_init=None):
        var = self.diff_coeff**2 * t
        return x_init, (var if var_init is None else var + var_init)


class VP(OU):
    """Variance-preserving SDE; beta(t) interpolates linearly (generative: from max down to min)."""

    def __init__(self, diff_coeff_sq_min: float = 0.1, diff_coeff_sq_max: float = 20.0,
                 scale_diff_coeff: float = 1.0, **kwargs):
        super().__init__(**kwargs)
        _buf(self, "scale_diff_coeff", scale_diff_coeff)
        _buf(self, "diff_coeff_sq_min", diff_coeff_sq_min)
        _buf(self, "diff_coeff_sq_max", diff_coeff_sq_max)

    def _diff_coeff_sq_t(self, t):
        lo, hi = self.diff_coeff_sq_min, self.diff_coeff_sq_max
        a, b = (hi, lo) if self.generative else (lo, hi)
        return torch.lerp(a, b, t / self.terminal_t)

    def drift_coeff_t(self, t):
        return self.sign * 0.5 * self._diff_coeff_sq_t(t)

    def diff_coeff_t(self, t):
        return self.scale_diff_coeff * self._diff_coeff_sq_t(t).sqrt()

    def int_drift_coeff_t(self, s, t):
        return self.sign * 0.25 * (self._diff_coeff_sq_t(t) + self._diff_coeff_sq_t(s)) * (t - s)

    def int_diff_coeff_sq_t(self, s, t):
        return 0.5 * self.scale_diff_coeff**2 * (self._diff_coeff_sq_t(t) + self._diff_coeff_sq_t(s)) * (t - s)

    def marginal_params(self, t, x_init, var_init=None):
        integral = self.int_drift_coeff_t(torch.zeros(1, device=t.device), t)
        growth = torch.exp(integral)
        var = (1 - torch.exp(2 * integral)) * self.scale_diff_coeff**2
        if var_init is not None:
            var = var + growth**2 * var_init
        return growth * x_init, var


class LangevinSDE(TorchSDE):
    """Overdamped Langevin dynamics dX = clip(target_score(X) * diff_coeff^2 / 2) dt + diff_coeff dW
    (reference eq/sdes.py:38-65; used by solver/langevin.py:45-46)."""

    def __init__(self, target_score, diff_coeff: float = 1.0, clip_score: float | None = None, **kwargs):
        super().__init__(**kwargs)
        self.target_score = target_score
        _buf(self, "diff_coeff", diff_coeff)
        self.clip_score = clip_score


class ControlledSDE(TorchSDE):
    """An OU process whose drift is shifted by diff * ctrl(t', x); t' = terminal_t - t when the OU process is the
    inference one (reference eq/sdes.py:272-305)."""

    def __init__(self, sde: OU, ctrl=None, **kwargs):
        super().__init__(terminal_t=sde.terminal_t.item(), **kwargs)
        self.sde = sde
        self.sde_type = sde.sde_type
        self.noise_type = sde.noise_type
        self.ctrl = ctrl

    def diff(self, t, x):
        return self.sde.diff(t, x)

    def f_and_g(self, t, x):
        """Host-side definition (reference eq/sdes.py:293-305) for code that evaluates the controlled process directly
        (torchsde-style `f` / `g`); `EulerIntegrator.integrate` does not call it -- it hands the whole process to libsdeh."""
        sde_diff = self.sde.diff(t, x)
        sde_drift = self.sde.drift(t, x)
        if self.ctrl is not None:
            if not self.sde.generative:
                t = self.terminal_t - t
            sde_drift = sde_drift + sde_diff * self.ctrl(t, x)
        return sde_drift, sde_diff.expand_as(x)

    def drift(self, t, x):
        return self.f_and_g(t, x)[0]

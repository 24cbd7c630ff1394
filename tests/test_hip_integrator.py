"""GPU parity tests of the plain Euler integrator (sdeh_integrate behind `EulerIntegrator.integrate`): against the
reference's golden vectors, the CPU oracle, and closed-form properties of the integrated processes."""
import json
import math
from pathlib import Path

import numpy as np
import pytest
import torch

from oracle import em_oracle as eo
from tests.helpers import GOLDEN_INT

pytestmark = pytest.mark.gpu

DEV = "cuda:0"
# fp32 with a different summation order than ATen (MFMA dot products, fused multiply-adds, analytic mixture score instead
# of autograd): per-step differences of a few ulp, mildly amplified over <= 100 steps
ATOL, RTOL = 2e-4, 2e-4


def load_int(path):
    fx = np.load(path)
    meta = json.loads(bytes(fx["meta"]).decode())
    params = {k[len("param/"):]: torch.from_numpy(fx[k].copy()) for k in fx.files if k.startswith("param/")}
    tt = None
    if meta["target"]["kind"] == "gmm":
        tt = {k: torch.from_numpy(fx["target/" + k].copy()) for k in ("loc", "scale", "mixture_weights")}
    return fx, meta, params, tt


def gpu(fx, key):
    return torch.from_numpy(fx[key]).to(DEV)


@pytest.mark.parametrize("path", GOLDEN_INT, ids=lambda p: Path(p).stem)
def test_integrator_matches_reference_golden(path):
    """Same x_init / timesteps / ts / per-step normals as the reference run: xs[len(ts), B, d] within fp32 tolerance."""
    from sde_sampler_amd import problems
    from sde_sampler_amd.eq.integrator import EulerIntegrator

    fx, meta, params, tt = load_int(path)
    sde, *_ = problems.build_integration(meta, params or None, tt, device=DEV)
    xs = EulerIntegrator().integrate(sde, ts=gpu(fx, "ts"), x_init=gpu(fx, "x_init"), timesteps=gpu(fx, "timesteps"),
                                     noise=gpu(fx, "noise"))
    ref = fx["xs"]
    assert xs.shape == ref.shape
    err = np.abs(xs.cpu().numpy() - ref)
    assert (err <= ATOL + RTOL * np.abs(ref)).all(), f"max |d xs| = {err.max():.3e}"


def test_default_grid_and_oracle_on_larger_batch():
    """timesteps=None -> get_timesteps(ts[0], ts[-1], dt) as in the reference (integrator.py:101-109); compared with the
    oracle on a ragged batch (B = 1000: a partially filled wave) and the padded d=3 kernel variant."""
    from sde_sampler_amd import problems
    from sde_sampler_amd.eq.integrator import EulerIntegrator

    meta = dict(target=dict(kind="gmm", dim=3, name="random7"), prior=dict(kind="iso_gauss", dim=3, loc=0.0, scale=1.0),
                integrate=dict(kind="langevin", diff_coeff=1.2, clip_score=50.0), grid=dict(start=0.0, end=1.0, steps=0))
    sde, target, prior, _ = problems.build_integration(meta, device=DEV)
    tt = dict(loc=target.loc.cpu(), scale=target.scale.cpu(), mixture_weights=target.mixture_weights.cpu())
    torch.manual_seed(3)
    B, dt = 1000, 0.02
    ts = torch.linspace(0.0, 1.0, 5)
    x0 = torch.randn(B, 3) * 2
    noise = torch.randn(50, B, 3)
    xs = EulerIntegrator(dt=dt).integrate(sde, ts=ts.to(DEV), x_init=x0.to(DEV), noise=noise.to(DEV))
    drift, diff = eo.integration_case(meta, {}, tt)
    ref = eo.euler_integrate(drift, diff, ts, x0, eo.timesteps(ts[0], ts[-1], dt=dt), noise=noise).detach()
    assert xs.shape == (5, B, 3)
    assert torch.allclose(xs.cpu(), ref, atol=ATOL, rtol=RTOL), (xs.cpu() - ref).abs().max()


def test_in_kernel_noise_is_deterministic_and_shard_invariant():
    from sde_sampler_amd import problems
    from sde_sampler_amd.eq.integrator import EulerIntegrator

    meta = dict(target=dict(kind="funnel", dim=10), prior=dict(kind="iso_gauss", dim=10, loc=0.0, scale=1.0),
                integrate=dict(kind="langevin", diff_coeff=1.0, clip_score=10.0), grid=dict(start=0.0, end=1.0, steps=0))
    sde, *_ = problems.build_integration(meta, device=DEV)
    torch.manual_seed(5)
    x0 = torch.randn(512, 10, device=DEV)
    ts = torch.linspace(0.0, 1.0, 3, device=DEV)

    def run(x, row_offset=0):
        integ = EulerIntegrator(dt=0.05)
        integ.row_offset = row_offset
        return integ.integrate(sde, ts=ts, x_init=x, seed=1234)

    full = run(x0)
    assert torch.equal(full, run(x0))
    halves = torch.cat([run(x0[:256]), run(x0[256:], row_offset=256)], dim=1)
    assert torch.equal(full, halves)
    assert not torch.equal(full[-1, :256], full[-1, 256:])


def test_ornstein_uhlenbeck_marginals():
    """Uncontrolled inference VP started at a point mass: the empirical mean / variance of X_T match the closed-form
    marginal (eq/sdes.py:257-269) up to the O(dt) Euler bias and the Monte-Carlo error."""
    from sde_sampler_amd.eq.integrator import EulerIntegrator
    from sde_sampler_amd.eq.sdes import VP

    sde = VP(diff_coeff_sq_min=0.1, diff_coeff_sq_max=10.0, terminal_t=1.0, generative=False).to(DEV)
    B = 1 << 16
    x0 = torch.full((B, 2), 3.0, device=DEV)
    ts = torch.tensor([0.0, 0.5, 1.0], device=DEV)
    xs = EulerIntegrator(dt=None, steps=400).integrate(sde, ts=ts, x_init=x0, seed=7)
    for k, t in enumerate([0.5, 1.0]):
        loc, var = sde.marginal_params(torch.tensor([t], device=DEV), x0[:1])
        assert abs(xs[k + 1].mean().item() - loc.mean().item()) < 0.02
        assert abs(xs[k + 1].var().item() - var.item()) < 0.03


def test_langevin_reaches_gaussian_target():
    """ULA on N(3, 0.5^2 I): stationary mean 3, variance sigma^2 / (1 - h / (4 sigma^2)) with h = diff^2 dt."""
    from sde_sampler_amd.distr.gauss import IsotropicGauss
    from sde_sampler_amd.eq.integrator import EulerIntegrator
    from sde_sampler_amd.eq.sdes import LangevinSDE

    target = IsotropicGauss(dim=4, loc=3.0, scale=0.5).to(DEV)
    sde = LangevinSDE(target_score=target.score, diff_coeff=1.0, clip_score=1e5, terminal_t=4.0).to(DEV)
    x0 = torch.zeros(1 << 15, 4, device=DEV)
    ts = torch.tensor([0.0, 4.0], device=DEV)
    xs = EulerIntegrator(dt=0.01).integrate(sde, ts=ts, x_init=x0, seed=11)
    assert torch.equal(xs[0], x0)
    var = 0.25 / (1.0 - 0.01 / (4 * 0.25))
    assert abs(xs[-1].mean().item() - 3.0) < 0.01
    assert abs(xs[-1].var().item() - var) < 0.01


def test_unsupported_and_invalid_inputs_fail_loudly():
    from sde_sampler_amd import SdehUnsupported
    from sde_sampler_amd.distr.gauss import IsotropicGauss
    from sde_sampler_amd.eq.integrator import EulerIntegrator
    from sde_sampler_amd.eq.sdes import VP, LangevinSDE, TorchSDE

    x0 = torch.zeros(8, 2, device=DEV)
    ts = torch.linspace(0, 1, 3, device=DEV)

    class MySDE(TorchSDE):
        pass

    with pytest.raises(SdehUnsupported):
        EulerIntegrator().integrate(MySDE(), ts=ts, x_init=x0)
    with pytest.raises(SdehUnsupported):  # a score callable the kernel cannot see into
        EulerIntegrator().integrate(LangevinSDE(target_score=lambda x: -x).to(DEV), ts=ts, x_init=x0)
    sde = VP(generative=False).to(DEV)
    with pytest.raises(AssertionError):  # output time beyond the grid (reference: assert ts_count == len(xs_out))
        EulerIntegrator().integrate(sde, ts=torch.tensor([0.0, 2.0], device=DEV), x_init=x0,
                                    timesteps=torch.linspace(0, 1, 11, device=DEV))
    with pytest.raises(RuntimeError, match="no CPU path"):
        EulerIntegrator().integrate(sde, ts=ts.cpu(), x_init=x0.cpu())
    target = IsotropicGauss(dim=3).to(DEV)
    with pytest.raises(ValueError):  # dimension mismatch between the score's distribution and the state
        EulerIntegrator().integrate(LangevinSDE(target_score=target.score).to(DEV), ts=ts, x_init=x0)

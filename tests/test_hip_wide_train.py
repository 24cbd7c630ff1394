"""Training through the wide-network kernels (C = 128 / 256, or 64 channels at d > 64): `loss(...).backward()` against the REFERENCE's
autograd gradients of tests/golden/wide_*.npz / widebridge_*.npz (tests/golden/make_golden_wide.py, both loss methods; Bridges: both
networks with the exact divergence) -- csrc/sdeh_wide_bwd.hip through the C ABI (sdeh_ctrl_backward_ex, sdeh_weight_grad,
sdeh_bridge_div_backward_wide)."""
import os
from pathlib import Path

import numpy as np
import pytest
import torch

from tests.helpers import GOLDEN_WIDE, GOLDEN_WIDE_BRIDGE, hip_problem, inference_params, load_fixture

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
GRAD_STRIDE = 5  # tests/golden/make_golden_wide.py: large gradient tensors are stored as every 5th flattened entry + their norm


def _golden_grad(fx, key):
    if key in fx.files:
        return fx[key], None
    return fx[key + "@stride"], float(fx[key + "@norm"])


def _check_grads(fx, method, prefix, module, tol=2e-4):
    worst = (0.0, "")
    for name, p in module.named_parameters():
        key = f"train_{method}/{prefix}/{name}"
        if key not in fx.files and key + "@stride" not in fx.files:
            continue
        ref, norm = _golden_grad(fx, key)
        g = p.grad.detach().cpu().numpy() if p.grad is not None else np.zeros(tuple(p.shape), np.float32)
        if norm is not None:
            assert abs(np.linalg.norm(g.astype(np.float64)) - norm) <= tol * max(norm, 1e-6), f"{method} {prefix} {name}: norm"
            g = g.reshape(-1)[::GRAD_STRIDE]
        scale = max(float(np.abs(ref).max()), 1e-6)
        err = float(np.abs(g - ref).max())
        worst = max(worst, (err / scale, name))
        assert err <= tol * scale + 1e-7, f"{method} {prefix} {name}: max err {err:.3e} vs scale {scale:.3e}"
    return worst


PLAIN = [p for p in GOLDEN_WIDE if "gmm" not in Path(p).name]


@pytest.mark.parametrize("method", ["lv", "kl"])
@pytest.mark.parametrize("path", PLAIN, ids=lambda p: Path(p).stem)
def test_wide_training_gradients_match_reference(path, method):
    fx, meta, params, tt = load_fixture(path)
    prob = hip_problem(meta, params, tt)
    prob.loss.method = method
    x0, noise = torch.from_numpy(fx["x0"]).to(DEV), torch.from_numpy(fx["noise"]).to(DEV)
    val, info = prob.loss(prob.ts, x0, prob.target.unnorm_log_prob, prob.second_log_prob, noise=noise)
    ref = float(fx[f"train_{method}/loss"])
    assert abs(val.item() - ref) <= 2e-4 * max(1.0, abs(ref)), (val.item(), ref)
    assert int(info["train/n_filtered_cumulative"]) == int(fx[f"train_{method}/n_filtered"])
    val.backward()
    assert prob.loss.engine.last_kernel_name() == f"bwd_wide<C={meta['net']['channels']},{'rows' if method == 'lv' else 'bptt'}>"
    worst = _check_grads(fx, method, "grad", prob.ctrl)
    print(f"{Path(path).stem} {method}: worst relative gradient error {worst[0]:.2e} ({worst[1]})")


def test_wide_training_noise_replay_equals_explicit_noise():
    """The backward replays the forward launch's Philox draws: gradients with in-kernel noise == gradients with the same draws given
    as a tensor (sdeh_debug_normals reproduces the stream)."""
    import ctypes as C

    from sde_sampler_amd import _lib as L

    fx, meta, params, tt = load_fixture([p for p in GOLDEN_WIDE if "dds_mw70" in p][0])
    prob = hip_problem(meta, params, tt)
    B, d, T = 70, meta["target"]["dim"], prob.ts.numel() - 1
    torch.manual_seed(2)
    x0 = prob.prior.sample((B,)).to(DEV)
    grads = []
    seed = torch.initial_seed()
    noise = torch.empty((T, B, d), device=DEV)
    for t in range(T):
        L.check(L.load().sdeh_debug_normals(C.c_uint64(seed & 0xFFFFFFFFFFFFFFFF), C.c_uint64(9), 0, t, d, B, noise[t].data_ptr(), None))
    torch.cuda.synchronize()
    for nz in (None, noise):
        prob.ctrl.zero_grad()
        prob.loss.engine.calls = 9
        val, _ = prob.loss(prob.ts, x0, prob.target.unnorm_log_prob, prob.second_log_prob, noise=nz)
        val.backward()
        grads.append(torch.cat([p.grad.flatten() for p in prob.ctrl.parameters() if p.grad is not None]).clone())
    assert torch.allclose(grads[0], grads[1], rtol=1e-5, atol=1e-7 * float(grads[1].abs().max()))


def test_wide_training_with_a_mixture_target_fails_loudly():
    from sde_sampler_amd import SdehUnsupported

    fx, meta, params, tt = load_fixture([p for p in GOLDEN_WIDE if "pis_gmm100" in p][0])
    prob = hip_problem(meta, params, tt)
    x0 = torch.from_numpy(fx["x0"]).to(DEV)
    val, _ = prob.loss(prob.ts, x0, prob.target.unnorm_log_prob, prob.second_log_prob)
    with pytest.raises(SdehUnsupported, match="mixture"):
        val.backward()


def _bridge(path):
    from sde_sampler_amd import problems

    fx, meta, params, tt = load_fixture(path)
    prob = problems.build(meta, params, tt, device=DEV, params_inf=inference_params(fx))
    return fx, meta, prob


@pytest.mark.parametrize("path", GOLDEN_WIDE_BRIDGE, ids=lambda p: Path(p).stem)
def test_wide_bridge_training_gradients_match_reference(path):
    """conf/solver/bridge.yaml's loss (time_reversal_lv) on wide networks: loss value and the parameter gradients of BOTH networks
    against the reference's autograd (exact divergence with create_graph=True: d backward passes per step, differentiated again)."""
    fx, meta, prob = _bridge(path)
    prob.loss.method = "lv"
    inf = prob.loss.inference_ctrl
    x0, noise = torch.from_numpy(fx["x0"]).to(DEV), torch.from_numpy(fx["noise"]).to(DEV)
    val, _ = prob.loss(prob.ts, x0, prob.target.unnorm_log_prob, prob.second_log_prob, noise=noise)
    ref = float(fx["train_lv/loss"])
    assert abs(val.item() - ref) <= 2e-4 * max(1.0, abs(ref)), (val.item(), ref)
    val.backward()
    assert prob.loss.engine.last_kernel_name() == f"bridge_div_bwd_wide<C={meta['net']['channels']}>"
    worst_u = _check_grads(fx, "lv", "grad", prob.ctrl)
    worst_v = _check_grads(fx, "lv", "grad_inf", inf)
    print(f"{Path(path).stem} lv: worst relative gradient error generative {worst_u[0]:.2e} ({worst_u[1]}), inference {worst_v[0]:.2e} ({worst_v[1]})")


def test_wide_bridge_training_with_kl_fails_loudly():
    from sde_sampler_amd import SdehUnsupported

    fx, meta, prob = _bridge(GOLDEN_WIDE_BRIDGE[0])
    prob.loss.method = "kl"
    x0 = torch.from_numpy(fx["x0"]).to(DEV)
    val, _ = prob.loss(prob.ts, x0, prob.target.unnorm_log_prob, prob.second_log_prob)
    with pytest.raises(SdehUnsupported, match="log-variance"):
        val.backward()

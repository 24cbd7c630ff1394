"""The fused training backward (csrc/sdeh_bwdf.hip: back-propagation + weight gradients in one kernel, no [C, T*B] planes) against
(i) the reference's autograd gradients on the golden fixtures and (ii) the plane-writing kernels it replaces (sdeh_ctrl_backward_ex +
sdeh_weight_grad, selected with SDEH_BWD_PLANES=1) over random problems: every loss / control / target kind, d = 1 .. 64 (one and two
coordinate tiles), ragged batches, supplied and replayed noise, methods kl / kl_ito / lv / lv_traj."""
import os
from pathlib import Path

import numpy as np
import pytest
import torch

from tests.helpers import GOLDEN, hip_problem, load_fixture
from tests.test_hip_fuzz import check_training_case, random_spec

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
N_RANDOM = 48 * int(os.environ.get("SDEH_FUZZ_SCALE", "1"))  # SDEH_FUZZ_SCALE=8: an occasional wider sweep (same seeds + more)


def _grads(prob, x0, noise, planes: bool, tile: int | None = None, waves: int | None = None):
    """Loss, gradients and the name of the backward kernel; `planes`: the plane-writing kernels instead of the fused one; `tile`: 16 |
    32 forces the trajectories per team of the fused backward (csrc/sdeh_bwdf16.hip serves kl / kl_ito below 16 384 trajectories);
    `waves`: 2 | 4 forces the wavefronts per 16-trajectory team (default: 4)."""
    if planes:
        os.environ["SDEH_BWD_PLANES"] = "1"
    # both paths behind the SAME forward kernel: the plane path's forward stores pre-activations from the M waves, which the quad mode
    # of small batches (four M waves on 16-row tiles, another MFMA shape and rounding) does not do
    os.environ["SDEH_WS_QUAD"] = "0"
    if tile is not None:
        os.environ["SDEH_BWD_TILE"] = str(tile)
    if waves is not None:
        os.environ["SDEH_BWD_WAVES"] = str(waves)
    try:
        prob.ctrl.zero_grad()
        val, _ = prob.loss(prob.ts, x0, prob.target.unnorm_log_prob, prob.second_log_prob, noise=noise)
        val.backward()
        name = prob.loss.engine.last_kernel_name()
        return val.item(), {k: (p.grad.detach().clone() if p.grad is not None else None) for k, p in prob.ctrl.named_parameters()}, name
    finally:
        os.environ.pop("SDEH_BWD_PLANES", None)
        os.environ.pop("SDEH_WS_QUAD", None)
        os.environ.pop("SDEH_BWD_TILE", None)
        os.environ.pop("SDEH_BWD_WAVES", None)


@pytest.mark.parametrize("method", ["lv", "kl"])
@pytest.mark.parametrize("path", GOLDEN, ids=lambda p: Path(p).stem)
def test_fused_backward_serves_the_golden_configurations(path, method):
    """The reference-gradient comparison itself is tests/test_hip_parity.py::test_training_gradients_match_reference; here: the
    fused kernel is what produced those gradients, and they agree with the plane-based path far inside that test's tolerance."""
    fx, meta, params, tt = load_fixture(path)
    prob = hip_problem(meta, params, tt)
    prob.loss.method = method
    x0 = torch.from_numpy(fx["x0"]).cuda()
    noise = torch.from_numpy(fx["noise"]).cuda()
    v2, g2, name2 = _grads(prob, x0, noise, planes=True)
    assert not name2.startswith("bwd_fused"), name2
    relu = type(prob.ctrl.base_model.activation).__name__ == "ReLU"
    # through time the fixture batches belong to the 16-trajectory kernel (activations without a kink), teams of four waves
    for tile, waves in ((16, 4), (16, 2), (32, None)):
        if method != "kl" and waves == 2:
            continue
        v1, g1, name1 = _grads(prob, x0, noise, planes=False, tile=tile, waves=waves)
        # every golden network (num_layers 3 .. 5) is one the fused kernel is compiled for
        expect = "bwd_fused16<bptt" if method == "kl" and tile == 16 and not relu else "bwd_fused<" + ("bptt" if method == "kl" else "rows")
        assert name1.startswith(expect), (name1, expect)
        assert v1 == v2
        for k in g1:
            ref = fx[f"train_{method}/grad/{k}"]
            a = g1[k].cpu().numpy() if g1[k] is not None else np.zeros_like(ref)
            b = g2[k].cpu().numpy() if g2[k] is not None else np.zeros_like(ref)
            scale = max(np.abs(ref).max(), 1e-6)
            assert np.abs(a - ref).max() <= 2e-4 * scale + 1e-7, f"{k}: fused({tile}, {waves}) vs reference {np.abs(a - ref).max():.3e} / {scale:.3e}"
            assert np.abs(a - b).max() <= 5e-5 * scale + 1e-7, f"{k}: fused({tile}, {waves}) vs planes {np.abs(a - b).max():.3e} / {scale:.3e}"


def test_relu_units_near_the_kink_follow_the_forward_pass():
    """The fused kernel re-evaluates the network; its pre-activations must be bit for bit the forward launch's (same first addend,
    same k order), or a ReLU unit within rounding of its kink takes the other side in the backward pass.  Case 697 of the wide
    sweep (ReLU, three hidden layers, lv_traj, 9252 row-steps) has such units: with the time embedding added BEHIND the input
    layer's products instead of in front, input_embed.bias came out 6e-4 off (float64 oracle: 3e-7)."""
    test_fused_backward_equals_plane_backward_on_random_problems(697, tol=1e-5)


@pytest.mark.parametrize("case", range(N_RANDOM))
def test_fused_backward_equals_plane_backward_on_random_problems(case, tol=2e-4):
    from sde_sampler_amd import SdehUnsupported, problems

    rng = np.random.default_rng(9000 + case)
    spec = random_spec(rng)
    if case % 2 == 0:
        spec["net"]["num_layers"] = 4  # the shipped depth (conf/model/base/fouriermlp.yaml); odd cases keep the random 3 .. 5
    method = str(rng.choice(["kl", "kl_ito", "lv", "lv_traj"]))
    spec["loss"]["method"] = method
    spec["loss"]["max_rnd"] = 1e8 if method.startswith("lv") else None
    if method == "lv_traj":
        spec["loss"]["traj_per_sample"] = 2
    spec["batch"] = int(rng.choice([33, 64, 100, 257]) if case % 5 else rng.choice([2, 7, 31]))  # every fifth: less than one tile
    if case % 4 == 3 and spec["target"]["kind"] != "double_well":  # two coordinate tiles (the double well is one-dimensional)
        d = int(rng.choice([33, 40, 64]))
        for part in ("target", "prior"):
            if spec[part] is not None and "dim" in spec[part]:
                spec[part]["dim"] = d
        if spec["target"]["kind"] == "gmm":
            spec["target"]["name"] = "random7"
        if spec["target"]["kind"] == "multi_well":
            spec["target"]["n_double_wells"] = min(spec["target"]["n_double_wells"], d)
        if spec["ctrl"].get("gamma_dim", 1) != 1:
            spec["ctrl"]["gamma_dim"] = d
    prob = problems.build(spec)
    prob.to(DEV)
    B, d, T = spec["batch"], spec["target"]["dim"], prob.ts.numel() - 1
    torch.manual_seed(case)
    x0 = prob.prior.sample((B,)).to(DEV)
    rows = B * spec["loss"].get("traj_per_sample", 1)
    noise = torch.randn(T, rows, d, device=DEV) if case % 3 else None  # every third case replays the Philox draws
    tag = f"case {case}: {method} {spec['loss']['kind']} / {spec['ctrl']['kind']} / {spec['target']['kind']} d={d} B={B} T={T}"
    eng = prob.loss.engine
    calls = eng.calls
    # odd cases through time: the 32-trajectory kernel although the batch is small (even ones: the launcher's choice, 16)
    tile = 32 if method.startswith("kl") and case % 2 else None
    waves = 2 if case % 4 == 0 else None  # (16-trajectory teams: two waves instead of the four these batch sizes get)
    try:
        v1, g1, name1 = _grads(prob, x0, noise, planes=False, tile=tile, waves=waves)
    except SdehUnsupported as exc:
        if "do not fit in LDS" in str(exc):
            pytest.skip(str(exc)[:120])
        raise
    eng.calls = calls  # same Philox offset for the second run
    try:
        v2, g2, name2 = _grads(prob, x0, noise, planes=True)
    except SdehUnsupported as exc:  # shapes only the fused kernel takes (three hidden layers at d > 32: the plane kernels' packed +
        pytest.skip("plane path: " + str(exc)[:100])  # transposed LDS images do not fit); test_..._matches_oracle_autograd covers them
    if name1.startswith("traj_legacy"):  # mixture tables beyond LDS: the forward keeps no planes, the plane path serves
        pytest.skip(f"{tag}: forward served by {name1}")
    assert name1.startswith("bwd_fused"), f"{tag}: {name1}"
    if method.startswith("kl") and tile is None and spec["net"].get("activation", "gelu") != "relu" and not (d <= 4 and spec["net"]["num_layers"] == 4):
        assert name1.startswith("bwd_fused16<bptt"), f"{tag}: {name1}"
    if method.startswith("kl") and tile is None and d <= 4 and spec["net"]["num_layers"] == 4:  # the scan form (sdeh_bwdf2.hip)
        assert name1.startswith("bwd_fused<bptt-scan"), f"{tag}: {name1}"
    assert not name2.startswith("bwd_fused"), f"{tag}: {name2}"
    assert v1 == v2 or (np.isnan(v1) and np.isnan(v2)), f"{tag}: loss {v1} vs {v2}"
    if not np.isfinite(v1):
        return
    gmax = max((torch.nan_to_num(g).abs().max().item() for g in g2.values() if g is not None), default=0.0)
    for k in g1:
        a, b = g1[k], g2[k]
        if a is None and b is None:
            continue
        a = torch.zeros_like(b) if a is None else a
        b = torch.zeros_like(a) if b is None else b
        if not torch.isfinite(b).all():
            continue
        # relative to the tensor's own scale, with a floor at 1e-3 of the network's largest gradient: a bias gradient that is the
        # cancelling sum of T * B terms carries the summation order's rounding (1e-7 of the terms), not its own 1e-7
        denom = max(b.abs().max().item(), 1e-3 * gmax, 1e-12)
        err = (a - b).abs().max().item() / denom
        assert err <= tol, f"{tag}: grad {k} fused vs planes rel err {err:.2e}"


@pytest.mark.parametrize("case", range(32 * int(os.environ.get("SDEH_FUZZ_SCALE", "1"))))
def test_fused_backward_matches_oracle_autograd_on_random_problems(case):
    """The random training problems of tests/test_hip_fuzz.py (even cases: the shipped depth; odd cases: 1 .. 3 hidden layers): loss
    and every parameter gradient against the ORACLE's autograd (conditioning-aware criteria of that test)."""
    check_training_case(1000 + case, num_layers=4 if case % 2 == 0 else None, expect_kernel="bwd_fused")


@pytest.mark.parametrize("batch,kernel", [(16384 + 17, "bwd_fused<bptt"), (12000 + 5, "bwd_fused16<bptt")])
def test_fused_backward_large_batch_is_deterministic(batch, kernel):
    """More tiles than teams (the persistent loop of either tiling), kl: two launches give bitwise identical gradients (fixed-order
    partial sums, no atomics), and the 16-trajectory launch agrees with the 32-trajectory one."""
    from sde_sampler_amd import problems

    spec = problems.baseline_spec("cfg3_gmm50_pis_kl")
    spec["batch"] = batch
    prob = problems.build(spec, device=DEV)
    torch.manual_seed(0)
    x0 = prob.prior.sample((spec["batch"],))
    eng = prob.loss.engine
    calls = eng.calls
    v1, g1, name = _grads(prob, x0, None, planes=False)
    eng.calls = calls
    v2, g2, _ = _grads(prob, x0, None, planes=False)
    assert name.startswith(kernel), name
    assert v1 == v2
    for k in g1:
        if g1[k] is not None:
            assert torch.equal(g1[k], g2[k]), k
    if "16" in kernel:
        eng.calls = calls
        v3, g3, name3 = _grads(prob, x0, None, planes=False, tile=32)
        assert name3.startswith("bwd_fused<bptt"), name3
        gmax = max(g.abs().max().item() for g in g3.values() if g is not None)
        for k in g1:
            if g1[k] is not None:
                err = (g1[k] - g3[k]).abs().max().item() / max(g3[k].abs().max().item(), 1e-3 * gmax)
                assert err <= 2e-4, f"{k}: tiles of 16 vs 32 rel err {err:.2e}"

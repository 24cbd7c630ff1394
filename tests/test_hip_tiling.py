"""32-row (one MFMA column tile per wave) against 64-row instantiations of the single-wave kernels, and shard invariance.

The parity tests run at sizes the oracle finishes in seconds, i.e. on whatever instantiation small batches select.  Kernels
that switch to 64-row tiles above 32 768 trajectories (the Hutchinson Bridge forward, the network-controlled integrator) are
pinned here: a batch of 40 000 must give what its two halves of 20 000 give, trajectory by trajectory -- the Philox streams are
keyed by the global row index, so `row_offset` reproduces the noise.  Back-propagation through time always runs 32-row tiles;
its test is the same statement as shard invariance of the gradient."""
import pytest
import torch

from sde_sampler_amd import problems

pytestmark = pytest.mark.gpu
B = 40000


def _build(spec):
    torch.manual_seed(5)
    return problems.build(spec, device="cuda:0")


def _grads(prob, x, row_offset, calls):
    lo = prob.loss
    lo.row_offset, lo.engine.calls = row_offset, calls
    params = [p for p in prob.ctrl.parameters()]
    inf = getattr(lo, "inference_ctrl", None)
    if inf is not None:
        params += list(inf.parameters())
    for p in params:
        p.grad = None
    value, _ = lo(prob.ts, x, prob.target.unnorm_log_prob, prob.second_log_prob)
    value.backward()
    return value.detach(), [p.grad.clone() for p in params]


def test_bptt_gradient_is_shard_invariant():
    spec = problems.baseline_spec("cfg2_gmm2_dis_kl")
    prob = _build(spec)
    x = prob.prior.sample((B,))
    v, g = _grads(prob, x, 0, 7)
    v0, g0 = _grads(prob, x[: B // 2], 0, 7)
    v1, g1 = _grads(prob, x[B // 2:], B // 2, 7)
    torch.testing.assert_close(v, 0.5 * (v0 + v1), rtol=1e-5, atol=1e-5)
    for a, b0, b1 in zip(g, g0, g1):
        ref = 0.5 * (b0 + b1)
        err = (a - ref).abs().max() / ref.abs().max().clamp_min(1e-12)
        assert err < 2e-4, float(err)


BRIDGE_SPEC = dict(batch=B, target=dict(kind="double_well", dim=1, separation=2.0, shift=1.5),
                   prior=dict(kind="iso_gauss", dim=1), sde=dict(kind="scaled_bm", diff_coeff=2.0, terminal_t=1.0),
                   ctrl=dict(kind="lerp_target", clip_model=10.0, clip_score=10.0, scale_score=1.0, gamma_dim=1, gamma_bias=1.0),
                   inference_ctrl=dict(kind="lerp_prior", clip_model=10.0, clip_score=10.0, scale_score=1.0, gamma_dim=1,
                                       gamma_bias=1.0),
                   net=dict(channels=64, num_layers=4, activation="gelu"),
                   loss=dict(kind="time_reversal", method="lv"), grid=dict(start=0.0, end=1.0, steps=40))


def test_bridge_hutchinson_64_row_tiles_match_32_row_tiles():
    """The probe-vector estimator has no per-coordinate redundancy, so batches > 32768 keep 64 trajectories per wave."""
    prob = _build(dict(BRIDGE_SPEC, loss=dict(kind="time_reversal", method="lv", div_estimator="rademacher")))
    x = prob.prior.sample((B,))
    eps = torch.randint(0, 2, (40, B, 1), device="cuda:0").float() * 2 - 1
    lo = prob.loss

    def sim(xx, e, row_offset):
        lo.row_offset, lo.engine.calls = row_offset, 3
        with torch.no_grad():
            return lo.simulate(prob.ts, xx, prob.target.unnorm_log_prob, prob.second_log_prob, train=True, compute_ito_int=True,
                               change_sde_ctrl=True, div_noise=e)

    xT, rnd, _ = sim(x, eps, 0)
    xT0, rnd0, _ = sim(x[: B // 2], eps[:, : B // 2].contiguous(), 0)
    xT1, rnd1, _ = sim(x[B // 2:], eps[:, B // 2:].contiguous(), B // 2)
    assert torch.equal(xT, torch.cat([xT0, xT1]))
    torch.testing.assert_close(rnd, torch.cat([rnd0, rnd1]), rtol=1e-5, atol=1e-4)


_TILES_SCRIPT = """
import sys, torch
sys.path.insert(0, {root!r})
from sde_sampler_amd import problems
spec = {spec!r}
torch.manual_seed(5)
prob = problems.build(spec, device="cuda:0")
x = prob.prior.sample((4096,))
prob.loss.engine.calls = 3
with torch.no_grad():
    xT, rnd, _ = prob.loss.simulate(prob.ts, x, prob.target.unnorm_log_prob, prob.second_log_prob, train=False, compute_ito_int=True)
torch.save(dict(xT=xT.cpu(), rnd=rnd.cpu()), {out!r})
"""


def test_bridge_exact_divergence_tilings_agree(tmp_path):
    """Exact divergence: 32-row tiles with act' kept in registers (the default), 32-row and 64-row tiles with the generic
    base + tangent passes (SDEH_BRIDGE_TILES; one subprocess per setting)."""
    import os
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    results = {}
    for mode in ("", "32g", "64"):
        out = str(tmp_path / f"tiles_{mode or 'default'}.pt")
        env = dict(os.environ, SDEH_BRIDGE_TILES=mode, SDEH_BRIDGE_SEQ="1")  # the step-sequential kernel (the default forward is the split of losses/_autograd.py)
        if not mode:
            env.pop("SDEH_BRIDGE_TILES")
        subprocess.run([sys.executable, "-c", _TILES_SCRIPT.format(root=root, spec=dict(BRIDGE_SPEC, batch=4096), out=out)],
                       check=True, env=env, timeout=600)
        results[mode] = torch.load(out)
    assert torch.equal(results[""]["xT"], results["64"]["xT"]) and torch.equal(results[""]["xT"], results["32g"]["xT"])
    assert torch.equal(results[""]["rnd"], results["32g"]["rnd"])  # same operations in the same order
    torch.testing.assert_close(results[""]["rnd"], results["64"]["rnd"], rtol=1e-5, atol=1e-4)


def test_controlled_integrator_64_row_tiles_match_32_row_tiles():
    from sde_sampler_amd.eq.integrator import EulerIntegrator

    meta = dict(target=dict(kind="double_well", dim=1, separation=2.0, shift=1.5), prior=dict(kind="iso_gauss", dim=1, loc=0.0, scale=1.0),
                sde=dict(kind="vp", beta_min=0.1, beta_max=10.0, scale=1.0, terminal_t=1.0, generative=True),
                ctrl=dict(kind="lerp", clip_model=1e4, clip_score=1e4, scale_score=1.0, gamma_dim=1, gamma_bias=1.0),
                net=dict(channels=64, num_layers=4, activation="gelu"),
                integrate=dict(kind="controlled"), grid=dict(start=0.0, end=1.0, steps=30))
    torch.manual_seed(9)
    sde, *_ = problems.build_integration(meta, device="cuda:0")
    x0 = torch.randn(B, 1, device="cuda:0")
    ts = torch.linspace(0.0, 1.0, 4, device="cuda:0")

    def run(x, row_offset=0):
        integ = EulerIntegrator(dt=None, steps=30)
        integ.row_offset = row_offset
        return integ.integrate(sde, ts=ts, x_init=x, seed=77)

    full = run(x0)
    halves = torch.cat([run(x0[: B // 2]), run(x0[B // 2:], row_offset=B // 2)], dim=1)
    assert torch.equal(full, halves)


@pytest.mark.parametrize("d,batch", [(1, 300), (2, 2048), (5, 777), (10, 2048), (33, 96)])
def test_bridge_coordinate_split_is_bitwise_the_single_wave_result(d, batch):
    """Small batches, exact divergence: the four waves of a workgroup carry the same 32 trajectories and share the d tangent passes
    (csrc/sdeh_bridge.hpp, `csplit`); the diagonal entries are summed in coordinate order by every wave, so samples, rnd, the
    trajectory and the training gradients of both networks are bit for bit those of one wave per tile (SDEH_BRIDGE_SPLIT=1)."""
    import os

    target = dict(kind="funnel", dim=d) if d >= 5 else (dict(kind="gmm", dim=d, name="random7") if d > 1 else BRIDGE_SPEC["target"])
    spec = dict(BRIDGE_SPEC, batch=batch, target=target,
                prior=dict(kind="iso_gauss", dim=d))
    for part in ("ctrl", "inference_ctrl"):
        spec[part] = dict(spec[part], clip_model=0.5 if d == 5 else 10.0)  # d = 5: the clamp's mask on the diagonal is active
    prob = _build(spec)
    x = prob.prior.sample((batch,))
    out = {}
    for split in ("1", "4"):
        os.environ["SDEH_BRIDGE_SPLIT"] = split
        os.environ["SDEH_BRIDGE_SEQ"] = "1"  # the step-sequential kernel this test is about (default: plain launch + row-parallel pass)
        try:
            prob.loss.engine.calls = 11
            res = prob.eval(x, compute_weights=True, return_traj=True)
            value, grads = _grads(prob, x, 0, 12)
            out[split] = (res.samples.clone(), res.weights.clone(), res.xs.clone(), value, grads)
        finally:
            os.environ.pop("SDEH_BRIDGE_SPLIT", None)
            os.environ.pop("SDEH_BRIDGE_SEQ", None)
    a, b = out["1"], out["4"]
    assert torch.isfinite(a[0]).all()
    for k in range(3):
        assert torch.equal(a[k], b[k]), k
    assert torch.equal(a[3], b[3])
    for ga, gb in zip(a[4], b[4]):
        assert torch.equal(ga, gb)


@pytest.mark.parametrize("name", ["cfg1_dw_dis_lv", "cfg2_gmm2_dis_kl", "cfg4_funnel_dds_lv"])
def test_quad_mode_agrees_with_pair_mode(name):
    """Small batches, d <= 32: four M waves on 16-row tiles (v_mfma_f32_16x16x4_f32, csrc/sdeh_traj_ws.hpp ws_mlp_quad) against the pair
    mode (two M waves, 32 x 32 x 2) on the same Philox stream: another instruction and another summation tree, so the agreement is that of
    two fp32 evaluations of the same trajectory -- estimators to 1e-4, rows to the contract's median / max bars."""
    import os

    spec = problems.baseline_spec(name)
    spec["batch"] = 1000
    prob = _build(spec)
    x = prob.prior.sample((1000,))
    out = {}
    for quad in ("0", "1"):
        os.environ["SDEH_WS_QUAD"] = quad
        try:
            prob.loss.engine.calls = 5
            res = prob.eval(x, compute_weights=True, return_traj=False)
            out[quad] = (res.samples.clone(), res.log_norm_const_preds["log_norm_const_lb_ito"], res.log_norm_const_preds["log_norm_const_is"])
        finally:
            os.environ.pop("SDEH_WS_QUAD", None)
    a, b = out["0"], out["1"]
    assert not torch.equal(a[0], b[0])  # (the switch really selects another kernel path)
    diff = (a[0] - b[0]).abs()
    assert diff.median().item() <= 1e-4 and diff.max().item() <= 1e-2, (diff.median().item(), diff.max().item())
    assert abs(a[1] - b[1]) <= 1e-4 and abs(a[2] - b[2]) <= 1e-3, (a[1:], b[1:])


@pytest.mark.parametrize("name", ["cfg1_dw_dis_lv", "cfg2_gmm2_dis_kl"])
def test_vector_pipe_out_layer_agrees_with_the_matrix_pipe_and_across_group_sizes(name):
    """d <= 4: the out layer on the V wave's vector pipe (the default for groups of 64 / 32 trajectories) against the matrix-pipe out
    layer (SDEH_WS_VOUT=0) on the same Philox stream -- two fp32 evaluations of the same trajectory: estimators to 1e-4, rows to the
    contract's bars; and with it groups of 32 trajectories (a batch of 20 000) stay bit-identical to groups of 64 (the same rows inside a
    batch of 40 000): the shard invariance of DESIGN.md 6."""
    import os

    B = 40000
    spec = problems.baseline_spec(name)
    spec["batch"] = B
    prob = _build(spec)
    x = prob.prior.sample((B,))

    def run(xx, **env):
        os.environ.update(env)
        try:
            prob.loss.engine.calls = 5
            res = prob.eval(xx, compute_weights=True, return_traj=False)
            return res.samples.clone(), res.log_norm_const_preds["log_norm_const_lb_ito"], res.log_norm_const_preds["log_norm_const_is"]
        finally:
            for k in env:
                os.environ.pop(k, None)

    vec64, vec32, mat64 = run(x), run(x[:20000].contiguous()), run(x, SDEH_WS_VOUT="0")
    assert torch.equal(vec32[0], vec64[0][:20000])
    assert not torch.equal(vec64[0], mat64[0])  # (the switch really selects another code path)
    diff = (vec64[0] - mat64[0]).abs()
    assert diff.median().item() <= 1e-4 and diff.max().item() <= 1e-2, (diff.median().item(), diff.max().item())
    assert abs(vec64[1] - mat64[1]) <= 1e-4 and abs(vec64[2] - mat64[2]) <= 1e-3, (vec64[1:], mat64[1:])

"""GPU parity tests: the HIP trajectory engine (through the C ABI, sde_sampler_amd/libsdeh.so) against
(a) the golden vectors captured from the reference and (b) the CPU oracle, on identical noise.

Tolerances (fp32, parity mode; SURVEY.md 8d / section 0.6: even re-associating one product in the reference
itself moves x_T by 5.8e-3 after 100 steps because the dynamics amplify 1-ulp changes):
  per-row x_T, rnd : median |d| <= 1e-4 (x scale), max |d| <= 1e-2 (+1e-4 relative for large-magnitude rnd)
  estimators       : |d| <= 1e-4 absolute, SURVEY 8d's bar (4e-6 relative beyond 25: tests/test_hip_contract.py::est_tol, which also
                     asserts the contract at its own batch size, B = 4096, for every BASELINE configuration)
"""
import math
import os
from pathlib import Path

import numpy as np
import pytest
import torch

from tests.helpers import GOLDEN, close, hip_problem, load_fixture, measured

pytestmark = pytest.mark.gpu

ROW_MAX, ROW_MEDIAN, ROW_RTOL = 1e-2, 1e-4, 1e-4
# Training parity bars: the loss at SURVEY 8d's estimator bar, 1e-4 relative (was 2e-3); parameter gradients at 2 x
# the worst error measured over all fixtures, methods and tensors (gpurun_out/parity_measured.txt of the round's GPU run; was 2e-4)
LOSS_BAR = 1e-4
GRAD_BAR = 1e-4


from contextlib import contextmanager


@contextmanager
def _env_var(name, value):
    old = os.environ.get(name)
    if value is None:
        os.environ.pop(name, None)
    else:
        os.environ[name] = value
    try:
        yield
    finally:
        if old is None:
            os.environ.pop(name, None)
        else:
            os.environ[name] = old


def _row_check(name, got, ref):
    got, ref = np.asarray(got, np.float64), np.asarray(ref, np.float64)
    err = np.abs(got - ref)
    scale = np.maximum(1.0, np.abs(ref))
    assert np.all(err <= ROW_MAX * scale), f"{name}: max err {err.max():.3e} (at |ref|={np.abs(ref).flat[err.argmax()]:.3e})"
    assert np.median(err / scale) <= ROW_MEDIAN, f"{name}: median err {np.median(err / scale):.3e}"


def _est_check(name, got, ref):
    from tests.test_hip_contract import est_tol

    assert abs(got - ref) <= est_tol(ref), f"{name}: {got} vs {ref}"


@pytest.mark.parametrize("path", GOLDEN, ids=lambda p: Path(p).stem)
def test_eval_matches_reference_golden(path):
    fx, meta, params, tt = load_fixture(path)
    prob = hip_problem(meta, params, tt)
    x0 = torch.from_numpy(fx["x0"]).cuda()
    noise = torch.from_numpy(fx["noise"]).cuda()
    want_xs = "eval1/xs" in fx.files
    r1 = prob.eval(x0, compute_weights=True, return_traj=want_xs, noise=noise)
    _row_check("x_T", r1.samples.cpu().numpy(), fx["eval1/x_T"])
    _est_check("lb_ito", r1.log_norm_const_preds["log_norm_const_lb_ito"], float(fx["eval1/log_norm_const_lb_ito"]))
    _est_check("logZ_is", r1.log_norm_const_preds["log_norm_const_is"], float(fx["eval1/log_norm_const_is"]))
    lv_ref = float(fx["eval1/lv_loss"])
    assert abs(r1.metrics["eval/lv_loss"] - lv_ref) <= 1e-4 * max(1.0, abs(lv_ref))  # a variance of O(B) rows at fixture size
    w, w_ref = r1.weights.cpu().numpy(), fx["eval1/weights"]
    assert w.shape == w_ref.shape and np.all(np.abs(w - w_ref) <= 2e-2 * np.maximum(w_ref, 1e-3) + 1e-4)
    if want_xs:
        xs = r1.xs.cpu().numpy()
        assert xs.shape == fx["eval1/xs"].shape
        assert np.array_equal(xs[0], fx["x0"])
        assert np.array_equal(xs[-1], r1.samples.cpu().numpy())
        _row_check("xs", xs, fx["eval1/xs"])
    r2 = prob.eval(x0, compute_weights=False, return_traj=False, noise=noise)
    _row_check("x_T(pass 2)", r2.samples.cpu().numpy(), fx["eval2/x_T"])
    _est_check("lb", r2.log_norm_const_preds["log_norm_const_lb"], float(fx["eval2/log_norm_const_lb"]))
    assert r2.weights is None and r2.xs is None
    # the Ito integral only enters rnd, never the state
    assert torch.equal(r1.samples, r2.samples)


@pytest.mark.parametrize("path", GOLDEN, ids=lambda p: Path(p).stem)
def test_rnd_rows_match_oracle(path):
    """Per-trajectory rnd (not part of Results) against the oracle on the same noise, both passes + train flags."""
    from oracle import em_oracle as eo

    fx, meta, params, tt = load_fixture(path)
    prob = hip_problem(meta, params, tt)
    oracle, _ = eo.problem_from_fixture(fx)
    ts, x0, noise = torch.from_numpy(fx["ts"]), torch.from_numpy(fx["x0"]), torch.from_numpy(fx["noise"])
    args = (prob.ts, x0.cuda(), prob.target.unnorm_log_prob, prob.second_log_prob)
    with torch.no_grad():
        for ito in (True, False):
            kw = dict(compute_ito_int=ito, return_traj=False, noise=noise.cuda())
            if meta["loss"]["kind"] == "time_reversal":
                kw["train"] = False
            _, rnd, _ = prob.loss.simulate(*args, **kw)
            _row_check(f"rnd(ito={ito})", rnd.cpu().numpy(), fx["eval1/rnd" if ito else "eval2/rnd"])
        # training-mode forward (log-variance form of the cost, no drift-divergence term), values only
        for method in ("kl", "lv"):
            _, rnd_o, _ = oracle.simulate(ts, x0, noise, train=True, compute_ito_int=method != "kl",
                                          change_sde_ctrl=method == "lv", method=method)
            prob.loss.method = method
            kw = dict(compute_ito_int=method != "kl", change_sde_ctrl=method == "lv", noise=noise.cuda())
            if meta["loss"]["kind"] == "time_reversal":
                kw["train"] = True
            _, rnd, _ = prob.loss.simulate(*args, **kw)
            _row_check(f"train rnd({method})", rnd.cpu().numpy(), rnd_o.detach().numpy())
            val, met = prob.loss(*args, noise=noise.cuda())
            ref = float(fx[f"train_{method}/loss"])
            measured(f"train loss {Path(path).stem} {method}", abs(val.item() - ref) / max(1.0, abs(ref)), LOSS_BAR)
            assert abs(val.item() - ref) <= LOSS_BAR * max(1.0, abs(ref)), (method, val.item(), ref)
            assert met["train/n_filtered_cumulative"] >= int(fx[f"train_{method}/n_filtered"])


def test_kl_training_with_unfusable_callable_fails_loudly():
    """method='kl' back-propagates through the terminal log-density inside the kernel; a callable the engine cannot
    fuse must raise, not fall back."""
    from sde_sampler_amd._lib import SdehUnsupported

    fx, meta, params, tt = load_fixture(GOLDEN[0])
    prob = hip_problem(meta, params, tt)
    prob.loss.method = "kl"
    x0 = torch.from_numpy(fx["x0"]).cuda()
    with pytest.raises(SdehUnsupported, match="built-in"):
        prob.loss(prob.ts, x0, lambda x: prob.target.unnorm_log_prob(x), prob.second_log_prob)


@pytest.mark.parametrize("method", ["lv", "kl"])
@pytest.mark.parametrize("path", GOLDEN, ids=lambda p: Path(p).stem)
def test_training_gradients_match_reference(path, method):
    """loss(...).backward() through the fused kernels (forward: trajectory kernel; backward: sdeh_ctrl_backward --
    row-parallel for 'lv', back-propagation through time for 'kl' -- + library GEMMs) against the parameter gradients
    the reference's autograd produced on the same noise (tests/golden: train_{lv,kl}/grad/*)."""
    fx, meta, params, tt = load_fixture(path)
    prob = hip_problem(meta, params, tt)
    prob.loss.method = method
    x0 = torch.from_numpy(fx["x0"]).cuda()
    noise = torch.from_numpy(fx["noise"]).cuda()
    prob.ctrl.zero_grad()
    val, met = prob.loss(prob.ts, x0, prob.target.unnorm_log_prob, prob.second_log_prob, noise=noise)
    ref_val = float(fx[f"train_{method}/loss"])
    assert abs(val.item() - ref_val) <= LOSS_BAR * max(1.0, abs(ref_val))
    val.backward()
    checked = 0
    for name, p in prob.ctrl.named_parameters():
        ref = fx[f"train_{method}/grad/{name}"]
        got = p.grad.detach().cpu().numpy() if p.grad is not None else np.zeros_like(ref)
        scale = max(np.abs(ref).max(), 1e-6)
        err = np.abs(got - ref).max()
        measured(f"grad {Path(path).stem} {method} {name}", err / scale, GRAD_BAR)
        assert err <= GRAD_BAR * scale + 1e-7, f"{name}: max err {err:.3e} vs scale {scale:.3e}"
        checked += 1
    assert checked >= 10


def test_lv_training_replays_kernel_noise():
    """Without an explicit noise tensor the backward pass replays the forward pass's Philox stream: the gradient must
    equal the one obtained when the same draws are passed explicitly (sdeh_debug_normals exposes them)."""
    import ctypes as C

    from sde_sampler_amd import _lib as L

    fx, meta, params, tt = load_fixture([p for p in GOLDEN if "cfg4" in p][0])
    prob = hip_problem(meta, params, tt)
    prob.loss.method = "lv"
    x0 = torch.from_numpy(fx["x0"]).cuda()
    T, B, d = fx["noise"].shape
    torch.manual_seed(1234)
    prob.loss.engine.calls = 5
    prob.ctrl.zero_grad()
    val, _ = prob.loss(prob.ts, x0, prob.target.unnorm_log_prob, prob.second_log_prob)
    val.backward()
    g_replay = {n: p.grad.clone() for n, p in prob.ctrl.named_parameters()}
    noise = torch.empty(T, B, d, device="cuda")
    lib = L.load()
    for t in range(T):
        assert lib.sdeh_debug_normals(torch.initial_seed(), 5, 0, t, d, B, noise[t].data_ptr(), None) == 0
    prob.ctrl.zero_grad()
    val2, _ = prob.loss(prob.ts, x0, prob.target.unnorm_log_prob, prob.second_log_prob, noise=noise)
    val2.backward()
    assert abs(val.item() - val2.item()) <= 1e-5 * max(1.0, abs(val.item()))
    for n, p in prob.ctrl.named_parameters():
        scale = max(g_replay[n].abs().max().item(), 1e-6)
        assert (p.grad - g_replay[n]).abs().max().item() <= 1e-4 * scale, n


def test_unrecognised_callables_are_called_back():
    """A terminal / initial log-density the engine cannot fuse is evaluated as given (device tensors)."""
    fx, meta, params, tt = load_fixture([p for p in GOLDEN if "cfg2" in p][0])
    prob = hip_problem(meta, params, tt)
    x0 = torch.from_numpy(fx["x0"]).cuda()
    noise = torch.from_numpy(fx["noise"]).cuda()
    with torch.no_grad():
        fused = prob.loss.eval(prob.ts, x0, prob.target.unnorm_log_prob, prob.prior.log_prob, noise=noise, return_traj=False)
        called = prob.loss.eval(prob.ts, x0, lambda x: prob.target.unnorm_log_prob(x), lambda x: prob.prior.log_prob(x),
                                noise=noise, return_traj=False)
    assert torch.equal(fused.samples, called.samples)
    a, b = fused.log_norm_const_preds, called.log_norm_const_preds
    assert abs(a["log_norm_const_is"] - b["log_norm_const_is"]) < 1e-4
    assert abs(a["log_norm_const_lb_ito"] - b["log_norm_const_lb_ito"]) < 1e-4


def test_ragged_and_tiny_batches():
    """Batch sizes that do not fill a wave / workgroup (1, 63, 65, 257) agree row-for-row with a full launch."""
    fx, meta, params, tt = load_fixture([p for p in GOLDEN if "cfg4" in p][0])
    prob = hip_problem(meta, params, tt)
    T, d = fx["noise"].shape[0], fx["noise"].shape[2]
    torch.manual_seed(3)
    x0 = torch.randn(300, d).cuda()
    noise = torch.randn(T, 300, d).cuda()
    full = prob.eval(x0, compute_weights=True, noise=noise)
    for b in (1, 63, 65, 257):
        part = prob.eval(x0[:b], compute_weights=True, noise=noise[:, :b].contiguous())
        assert torch.equal(part.samples, full.samples[:b]), b


@pytest.mark.parametrize("env_var", ["SDEH_GENERIC_ONLY", "SDEH_GENERIC_ONLY=2", "SDEH_LEGACY"])
def test_alternative_kernels_also_match(env_var):
    """The BASELINE configurations normally dispatch to compile-time specialised, wave-specialised kernels.  Re-run the
    golden parity tests in a subprocess with SDEH_GENERIC_ONLY=1 (run-time switched variants, incl. the one with mixture tables over
    four coordinates), =2 (the plain generic variants only) and with SDEH_LEGACY=1 (the single-wave kernel with scalar-load mixture
    tables, the fallback for mixtures too large for LDS)."""
    import os
    import subprocess
    import sys

    if os.environ.get("SDEH_GENERIC_ONLY") or os.environ.get("SDEH_LEGACY"):
        pytest.skip("already an alternative-kernel run")
    name, _, value = env_var.partition("=")
    env = dict(os.environ, **{name: value or "1"})
    # (the nested session starts with fresh logs of its own: it must not unlink this session's escape-hatch / measured-parity records)
    import tempfile
    env["SDEH_HATCH_REPORT"] = os.path.join(tempfile.mkdtemp(prefix="sdeh_nested_"), "fuzz_hatches.txt")
    out = subprocess.run([sys.executable, "-m", "pytest", __file__, "-q", "-x", "-m", "gpu", "-k",
                          "eval_matches_reference_golden or rnd_rows_match_oracle"],
                         env=env, capture_output=True, text=True, cwd=str(Path(__file__).parents[1]))
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]


@pytest.mark.parametrize("case", ["shared_all_dims_vary", "shared_prefix_3", "general_scales", "general_k64_falls_back"])
def test_mixture_table_variants_vs_oracle(case):
    """d = 50 mixtures that exercise every table form: shared scale with all coordinates varying (full tables), shared
    scale with a varying prefix (SDEH_DENS_FLAG_NVARY), general scales in LDS, and a mixture too large for LDS
    (single-wave kernel with scalar-load tables).  Checked against the oracle on identical noise."""
    from oracle import em_oracle as eo
    from sde_sampler_amd import problems

    d = 50
    gen = torch.Generator().manual_seed(5)
    k = 64 if case == "general_k64_falls_back" else 12
    loc = (torch.rand((k, d), generator=gen) - 0.5) * 6.0
    scale = 0.8 + 0.4 * torch.rand((1, d), generator=gen).expand(k, d).contiguous()
    if case == "shared_prefix_3":
        loc[:, 3:] = loc[0, 3:]
    if case.startswith("general"):
        scale = 0.8 + 0.4 * torch.rand((k, d), generator=gen)
    w = 0.5 + torch.rand((k,), generator=gen)
    tt = dict(loc=loc, scale=scale, mixture_weights=w)
    spec = problems.baseline_spec("gmm50_pis_headline")
    spec["grid"]["steps"] = 12
    prob = problems.build(spec, target_tensors=tt)
    params = {n: v.detach().clone() for n, v in prob.ctrl.state_dict().items()}
    B = 96
    torch.manual_seed(3)
    x0 = prob.prior.sample((B,))
    noise = torch.randn(12, B, d)
    ref = eo.Problem(spec, params, {n: v.clone() for n, v in tt.items()}).eval(prob.ts.clone(), x0.clone(), noise, compute_weights=True)
    prob.to("cuda:0")
    out = prob.eval(x0.cuda(), compute_weights=True, noise=noise.cuda())
    _row_check("x_T", out.samples.cpu().numpy(), ref["samples"].numpy())
    _est_check("lb_ito", out.log_norm_const_preds["log_norm_const_lb_ito"], ref["log_norm_const_lb_ito"])
    _est_check("logZ_is", out.log_norm_const_preds["log_norm_const_is"], ref["log_norm_const_is"])


@pytest.mark.parametrize("scales", ["shared", "general", "shared_overlapping", "general_overlapping"])
@pytest.mark.parametrize("mode", ["matrix_pipe", "scalar_cache_tables"])
def test_dense_mixture_whole_wave_paths_vs_oracle(mode, scales, monkeypatch):
    """A well-separated 40-component mixture with means varying in all 50 coordinates at a batch that launches whole waves
    (B > 8192): by default the binding vouches for the product form (engine._mixture_mm_ok) and both mixture contractions run on the
    matrix pipe inside the V wave (v_mfma_f32_4x4x1, kernel name "...,mm"); with the plan option SDEH_GMM_MM=0 the exact form streams
    its tables through the scalar cache (gmm_online_s: bit-identical to the LDS tables of the small-batch modes).  Both against the
    oracle on identical noise, at the bars of every other mixture test."""
    from oracle import em_oracle as eo
    from sde_sampler_amd import engine, problems

    if mode == "scalar_cache_tables":
        monkeypatch.setenv("SDEH_GMM_MM", "0")
    spec = problems.baseline_spec("gmm50_dense_" + scales.split("_")[0])
    spec["grid"]["steps"] = 10
    tensors = None
    if scales.endswith("overlapping"):
        # 28 components (the instruction stream's 40 rows are padded) within a few scaled units of the origin: every trajectory spreads
        # its weight over several components (the product form's rounding stays below 1e-5 there: rule (a) of engine._mixture_mm_ok)
        # -- the bench mixtures above are one-hot
        gen = torch.Generator().manual_seed(9)
        loc = (torch.rand((28, 50), generator=gen) - 0.5) * 1.6
        scale = 0.9 + 0.3 * torch.rand((1, 50) if scales.startswith("shared") else (28, 50), generator=gen)
        tensors = dict(loc=loc, scale=scale.expand(28, 50).contiguous(), mixture_weights=0.5 + torch.rand((28,), generator=gen))
    prob = problems.build(spec, target_tensors=tensors)
    assert engine._mixture_mm_ok(prob.target.loc, prob.target.scale)
    params = {n: v.detach().clone() for n, v in prob.ctrl.state_dict().items()}
    tt = dict(loc=prob.target.loc.clone(), scale=prob.target.scale.clone(), mixture_weights=prob.target.mixture_weights.clone())
    B = 8192 + 320
    torch.manual_seed(21)
    x0 = prob.prior.sample((B,))
    noise = torch.randn(10, B, 50)
    ref = eo.Problem(spec, params, tt).eval(prob.ts.clone(), x0.clone(), noise, compute_weights=True)
    prob.to("cuda:0")
    out = prob.eval(x0.cuda(), compute_weights=True, noise=noise.cuda())
    kernel = prob.loss.engine.last_kernel_name()
    variant = "50_0_pis_gmm" if scales.startswith("shared") else "50_0_g"  # (per-component scales: the run-time switched variant)
    assert kernel == (f"traj_ws<{variant},mm>" if mode == "matrix_pipe" else f"traj_ws<{variant}>"), kernel
    _row_check("x_T", out.samples.cpu().numpy(), ref["samples"].numpy())
    _est_check("lb_ito", out.log_norm_const_preds["log_norm_const_lb_ito"], ref["log_norm_const_lb_ito"])
    _est_check("logZ_is", out.log_norm_const_preds["log_norm_const_is"], ref["log_norm_const_is"])
    from tests.helpers import measured
    from tests.test_hip_contract import est_tol
    err = (out.samples.cpu() - ref["samples"]).abs()
    measured(f"dense_mixture[{scales},{mode}]/x_T_max", float(err.max()), ROW_MAX)
    measured(f"dense_mixture[{scales},{mode}]/x_T_median", float(err.median()), ROW_MEDIAN)
    measured(f"dense_mixture[{scales},{mode}]/logZ_is", abs(out.log_norm_const_preds["log_norm_const_is"] - ref["log_norm_const_is"]),
             est_tol(ref["log_norm_const_is"]))


def test_matrix_pipe_mixture_at_the_edge_of_the_guard():
    """Worst case the product-form guard admits (engine._mixture_mm_ok, rule (b)): 20 PAIRS of components 12.5 scaled units apart, 140
    sigma from the origin (logit rounding bound 1.3e-3), and trajectories started ON the mid-planes of the pairs, where both members
    carry weight.  The matrix-pipe path against the exact form (plan option SDEH_GMM_MM=0) on identical noise: the responsibilities may
    move by the rounding bound x r (1 - r), the score by that times the pair distance / sigma^2 -- measured and bounded here -- and the
    estimators agree at the bars of the other mixture tests.  (On the mid-plane the exact form's own logits are ~ (12.5 / 2)^2 = 39 with
    fp32 rounding 2e-6: the product form IS the less accurate one there -- by what the guard's constants say.)"""
    import math

    from sde_sampler_amd import engine, problems

    gen = torch.Generator().manual_seed(4)
    d, P = 50, 20
    c = torch.randn(P, d, generator=gen)
    c = c / c.norm(dim=1, keepdim=True) * 140.0
    e = torch.randn(P, d, generator=gen)
    e = e / e.norm(dim=1, keepdim=True)
    delta = 12.5 * math.sqrt(2.0)
    tt = dict(loc=torch.cat([c + 0.5 * delta * e, c - 0.5 * delta * e]), scale=torch.ones(2 * P, d), mixture_weights=torch.ones(2 * P))
    assert engine._mixture_mm_ok(tt["loc"], tt["scale"])
    spec = problems.baseline_spec("gmm50_dense_shared")
    spec["grid"]["steps"] = 8
    prob = problems.build(spec, target_tensors=tt, device="cuda:0")
    B = 8192 + 512
    torch.manual_seed(6)
    pair = torch.randint(0, P, (B,))
    # on the mid-plane (+- a fraction of the slab |l_j - l_k| < 20, which is 0.57 sigma wide here), scattered within it
    x0 = (c[pair] + 0.1 * torch.randn(B, 1) * e[pair] + 0.5 * torch.randn(B, d)).cuda()
    noise = 0.05 * torch.randn(8, B, d, device="cuda")
    out = {}
    for mode in ("mm", "exact"):
        with _env_var("SDEH_GMM_MM", None if mode == "mm" else "0"):
            r = prob.eval(x0, compute_weights=True, noise=noise)
            out[mode] = (r.samples.clone(), r.log_norm_const_preds["log_norm_const_lb_ito"], prob.loss.engine.last_kernel_name())
    assert out["mm"][2].endswith(",mm>") and not out["exact"][2].endswith(",mm>"), (out["mm"][2], out["exact"][2])
    diff = (out["mm"][0] - out["exact"][0]).abs()
    # score error <= 1.3e-3 x 1/4 x 17.7 = 6e-3 per step at worst, times the control's step (sigma^2 dt = 0.2 x 5 / 8): <= 7e-4 per step
    from tests.helpers import measured
    measured("matrix_pipe_guard_edge/x_T_max_vs_exact_form", float(diff.max()), 1.5e-3)
    measured("matrix_pipe_guard_edge/x_T_median_vs_exact_form", float(diff.median()), 1e-4)
    assert float(diff.max()) <= 1.5e-3 and float(diff.median()) <= 1e-4, (float(diff.max()), float(diff.median()))  # measured: 5.1e-4 / 0
    assert abs(out["mm"][1] - out["exact"][1]) <= 1e-4 * max(1.0, abs(out["exact"][1]))


@pytest.mark.parametrize("d,k,scales,shape", [(10, 21, "shared", "pis"), (10, 40, "general", "dis"), (10, 33, "general", "pis"),
                                              (50, 24, "general", "dis"), (50, 37, "shared", "dis"), (10, 30, "shared", "dds")])
def test_matrix_pipe_mixture_random_cases_vs_oracle(d, k, scales, shape):
    """The matrix-pipe mixture in both compiled dimension classes (d = 10: 100 + 120 instructions per step; d = 50), with 21 .. 40
    components (padding rows of the 40-row stream), both table forms, under the three losses / two controls the specs bring (run-time
    switched variants except PIS + shared scale at d = 50): random overlapping mixtures near the origin, where every trajectory spreads
    its weight and rule (a) of engine._mixture_mm_ok admits the product form; against the oracle on identical noise."""
    from oracle import em_oracle as eo
    from sde_sampler_amd import engine, problems

    base = {"pis": "gmm50_pis_headline", "dis": "cfg2_gmm2_dis_kl", "dds": "cfg4_funnel_dds_lv"}[shape]
    spec = problems.baseline_spec(base)
    spec["target"] = dict(kind="gmm", dim=d, name="random")
    spec["prior"] = dict(spec["prior"], dim=d)
    spec["grid"] = dict(spec["grid"], steps=8, rescale_t=None)
    gen = torch.Generator().manual_seed(100 * d + k)
    loc = (torch.rand((k, d), generator=gen) - 0.5) * (3.0 if d == 10 else 1.4)
    scale = 0.8 + 0.5 * torch.rand((1, d) if scales == "shared" else (k, d), generator=gen)
    tt = dict(loc=loc, scale=scale.expand(k, d).contiguous(), mixture_weights=0.5 + torch.rand((k,), generator=gen))
    prob = problems.build(spec, target_tensors=tt)
    assert engine._mixture_mm_ok(prob.target.loc, prob.target.scale)
    params = {n: v.detach().clone() for n, v in prob.ctrl.state_dict().items()}
    B = 8192 + 64
    torch.manual_seed(d + k)
    x0 = prob.prior.sample((B,))
    noise = torch.randn(8, B, d)
    ref = eo.Problem(spec, params, {n: v.clone() for n, v in tt.items()}).eval(prob.ts.clone(), x0.clone(), noise, compute_weights=True)
    prob.to("cuda:0")
    out = prob.eval(x0.cuda(), compute_weights=True, noise=noise.cuda())
    kernel = prob.loss.engine.last_kernel_name()
    assert kernel.endswith(",mm>"), kernel
    _row_check("x_T", out.samples.cpu().numpy(), ref["samples"].numpy())
    _est_check("lb_ito", out.log_norm_const_preds["log_norm_const_lb_ito"], ref["log_norm_const_lb_ito"])
    _est_check("logZ_is", out.log_norm_const_preds["log_norm_const_is"], ref["log_norm_const_is"])
    from tests.helpers import measured
    err = (out.samples.cpu() - ref["samples"]).abs()
    measured(f"matrix_pipe_mixture[d={d},k={k},{scales},{shape}]/x_T_max", float(err.max()), ROW_MAX)


@pytest.mark.parametrize("d", [5, 8, 10, 13, 16])
def test_out_layer_on_4x4_matrix_instructions_vs_oracle(d, monkeypatch):
    """State dimensions 5 .. 16 at whole-wave batches: the out layer runs as v_mfma_f32_4x4x1 row groups (sdeh_traj_ws.hpp: ws_out4_stage;
    33 ceil(d / 4) instructions per column tile instead of a 32-row tile with 16 .. 27 empty rows).  Against the oracle on identical noise
    and against the 32-row tiles (plan option SDEH_WS_OUT4=0), groups of 32 trajectories (B = 16 640) -- the BASELINE shard sizes and
    groups of 64 run it in tests/test_hip_fullsize.py."""
    from oracle import em_oracle as eo
    from sde_sampler_amd import problems

    spec = problems.baseline_spec("cfg4_funnel_dds_lv")
    spec["target"] = dict(kind="funnel", dim=d)
    spec["prior"] = dict(spec["prior"], dim=d)
    spec["grid"] = dict(spec["grid"], steps=6)
    prob = problems.build(spec)
    params = {n: v.detach().clone() for n, v in prob.ctrl.state_dict().items()}
    B = 16384 + 256  # (closed-form targets keep the quad mode up to 16 384 trajectories)
    torch.manual_seed(d)
    x0 = prob.prior.sample((B,))
    noise = torch.randn(prob.ts.numel() - 1, B, d)
    ref = eo.Problem(spec, params, None).eval(prob.ts.clone(), x0.clone(), noise, compute_weights=True)
    prob.to("cuda:0")
    out = prob.eval(x0.cuda(), compute_weights=True, noise=noise.cuda())
    _row_check("x_T", out.samples.cpu().numpy(), ref["samples"].numpy())
    _est_check("logZ_is", out.log_norm_const_preds["log_norm_const_is"], ref["log_norm_const_is"])
    monkeypatch.setenv("SDEH_WS_OUT4", "0")
    old = prob.eval(x0.cuda(), compute_weights=True, noise=noise.cuda())
    diff = float((out.samples - old.samples).abs().max())
    assert 0.0 < diff <= 2e-5, diff  # another summation order of the same 65 products per coordinate: not bitwise, fp32-close
    from tests.helpers import measured
    measured(f"out_layer_4x4[d={d}]/x_T_max_vs_oracle", float((out.samples.cpu() - ref["samples"]).abs().max()), ROW_MAX)
    measured(f"out_layer_4x4[d={d}]/x_T_max_vs_32_row_tiles", diff, 2e-5)


@pytest.mark.parametrize("d", [10, 20, 32, 33, 50, 64])
@pytest.mark.parametrize("shape", ["pis", "dis", "dds"])
def test_padded_reference_mixture_in_other_dimensions_vs_oracle(d, shape):
    """The reference's high-dimensional mixtures are its 2-d "fab" mixture padded with zero means (distr/gauss.py:59-60): under any
    loss / control, in the dimension classes that carry a "g4" variant (d = 50, 17 .. 32, 33 .. 64), the mixture tables cover four
    coordinates and the rest factors out as one Gaussian (SDEH_DENS_FLAG_NVARY).  Against the oracle on identical noise; d = 10
    takes the full tables of the plain generic variant."""
    from oracle import em_oracle as eo
    from sde_sampler_amd import problems

    base = {"pis": "gmm50_pis_headline", "dis": "cfg2_gmm2_dis_kl", "dds": "cfg4_funnel_dds_lv"}[shape]
    spec = problems.baseline_spec(base)
    spec["target"] = dict(kind="gmm", dim=d, name="fab50")
    spec["prior"] = dict(spec["prior"], dim=d)
    spec["grid"] = dict(spec["grid"], steps=10, rescale_t=None)
    prob = problems.build(spec)
    params = {n: v.detach().clone() for n, v in prob.ctrl.state_dict().items()}
    tt = dict(loc=prob.target.loc.clone(), scale=prob.target.scale.clone(), mixture_weights=prob.target.mixture_weights.clone())
    B = 100
    torch.manual_seed(11)
    x0 = prob.prior.sample((B,))
    noise = torch.randn(10, B, d)
    ref = eo.Problem(spec, params, tt).eval(prob.ts.clone(), x0.clone(), noise, compute_weights=True)
    prob.to("cuda:0")
    out = prob.eval(x0.cuda(), compute_weights=True, noise=noise.cuda())
    kernel = prob.loss.engine.last_kernel_name()
    if d == 50:
        assert kernel in ("traj_ws<50_0_pis_gmm4>", "traj_ws<50_0_g4>"), kernel
    elif d > 16:
        assert kernel == f"traj_ws<{32 if d <= 32 else 64}_1_g4>", kernel
    _row_check("x_T", out.samples.cpu().numpy(), ref["samples"].numpy())
    # (100 rows whose rnd is O(10 .. 100) -- means at +-40, scores of that size: the 1e-4 absolute bar of section 8d is set for B >= 4096;
    # here 2e-5 relative, measured worst case 1.0e-5 at d = 10 in the quad mode's own chunk order)
    for name in ("log_norm_const_lb_ito", "log_norm_const_is"):
        got, want = out.log_norm_const_preds[name], ref[name]
        assert abs(got - want) <= max(1e-4, 2e-5 * abs(want)), f"{name}: {got} vs {want}"


def test_deep_network_falls_back_to_single_wave_kernel():
    """num_layers = 8 (six hidden layers): packed weights + exchange buffers exceed 160 KiB of LDS, so the launch falls
    back to the single-wave kernel.  Checked against the oracle."""
    from oracle import em_oracle as eo
    from sde_sampler_amd import problems

    spec = problems.baseline_spec("cfg4_funnel_dds_lv")
    spec["net"]["num_layers"] = 8
    spec["grid"]["steps"] = 20
    prob = problems.build(spec)
    params = {n: v.detach().clone() for n, v in prob.ctrl.state_dict().items()}
    B, d = 80, 10
    torch.manual_seed(4)
    x0 = torch.randn(B, d)
    ts = prob.ts.clone()
    noise = torch.randn(ts.numel() - 1, B, d)
    ref = eo.Problem(spec, params, None).eval(ts, x0.clone(), noise, compute_weights=True)
    prob.to("cuda:0")
    out = prob.eval(x0.cuda(), compute_weights=True, noise=noise.cuda())
    _row_check("x_T", out.samples.cpu().numpy(), ref["samples"].numpy())
    _est_check("logZ_is", out.log_norm_const_preds["log_norm_const_is"], ref["log_norm_const_is"])


EXTRA_METHODS = [p for p in GOLDEN if "train_kl_ito/loss" in np.load(p).files]


@pytest.mark.parametrize("path", EXTRA_METHODS, ids=lambda p: Path(p).stem)
def test_kl_ito_and_lv_traj_training_match_reference(path):
    """The two remaining loss methods: kl_ito (BPTT with the Ito term) and lv_traj (variance over traj_per_sample = 2
    trajectories started from the same x0), losses and parameter gradients vs the reference."""
    fx, meta, params, tt = load_fixture(path)
    x0 = torch.from_numpy(fx["x0"]).cuda()
    for method, noise, tps in (("kl_ito", fx["noise"], 1), ("lv_traj", fx["noise_traj2"], 2)):
        prob = hip_problem(meta, params, tt)
        prob.loss.method, prob.loss.traj_per_sample = method, tps
        val, _ = prob.loss(prob.ts, x0, prob.target.unnorm_log_prob, prob.second_log_prob,
                           noise=torch.from_numpy(noise).cuda())
        ref_val = float(fx[f"train_{method}/loss"])
        measured(f"train loss {Path(path).stem} {method}", abs(val.item() - ref_val) / max(1.0, abs(ref_val)), LOSS_BAR)
        assert abs(val.item() - ref_val) <= LOSS_BAR * max(1.0, abs(ref_val)), (method, val.item(), ref_val)
        val.backward()
        for name, p in prob.ctrl.named_parameters():
            ref = fx[f"train_{method}/grad/{name}"]
            scale = max(np.abs(ref).max(), 1e-6)
            err = np.abs(p.grad.cpu().numpy() - ref).max()
            measured(f"grad {Path(path).stem} {method} {name}", err / scale, GRAD_BAR)
            assert err <= GRAD_BAR * scale + 1e-7, f"{method} {name}: max err {err:.3e} vs scale {scale:.3e}"

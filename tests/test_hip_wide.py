"""GPU parity tests of the wide-network kernels (sdeh_wide.hip: FourierMLP with 128 / 256 channels, d up to 196) through the C ABI:
against the reference's golden vectors on identical noise (tests/golden/make_golden_wide.py), the CPU oracle at larger batches,
and size-independent properties (tiling / sharding invariance, determinism)."""
import os
from pathlib import Path

import numpy as np
import pytest
import torch

from tests.helpers import GOLDEN_WIDE, GOLDEN_WIDE_BRIDGE, fuzz_close, hip_problem, inference_params, load_fixture

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _rows(name, got, ref, max_tol=1e-2, med_tol=1e-4):
    got, ref = np.asarray(got, np.float64), np.asarray(ref, np.float64)
    scale = np.maximum(1.0, np.abs(ref))
    err = np.abs(got - ref) / scale
    assert err.max() <= max_tol, f"{name}: max err {err.max():.3e}"
    assert np.median(err) <= med_tol, f"{name}: median err {np.median(err):.3e}"


#: share of rows allowed beyond the per-row bar after the conditioning probes (the narrow sweep's measured bound, tests/test_hip_fuzz.py)
DRIFT_MAX = float(os.environ.get("SDEH_FUZZ_DRIFT_MAX", "0.05"))


def _est_tol(want: float) -> float:
    """|delta| <= 1e-4 absolute "at B >= 4096"; an fp32 estimator of magnitude > 25 cannot be held to that (one ulp of 116
    is 7.6e-6 and every row sums 196 coordinates over T steps in another order than torch does): tests/test_hip_contract.py holds
    4e-6 relative at B = 4096, where the mean averages the rows' rounding; these fixtures have 24 .. 48 rows, so the relative part is
    1e-5 (1.3 ulp of the value; measured worst case 4.3e-6: wide_dis_gauss196_c256, |lb_ito| = 116.5)."""
    return max(1e-4, 1e-5 * abs(want))


class _ct:
    """forces the column tiles per workgroup (32 or 64 trajectories) of the wide kernels"""

    def __init__(self, ct):
        self.ct = ct

    def __enter__(self):
        if self.ct:
            os.environ["SDEH_WIDE_CT"] = str(self.ct)

    def __exit__(self, *a):
        os.environ.pop("SDEH_WIDE_CT", None)


@pytest.mark.parametrize("ct", [1, 2])
@pytest.mark.parametrize("path", GOLDEN_WIDE, ids=lambda p: Path(p).stem)
def test_wide_eval_matches_reference_golden(path, ct):
    fx, meta, params, tt = load_fixture(path)
    prob = hip_problem(meta, params, tt)
    x0, noise = torch.from_numpy(fx["x0"]).to(DEV), torch.from_numpy(fx["noise"]).to(DEV)
    with _ct(ct):
        r1 = prob.eval(x0, compute_weights=True, return_traj=True, noise=noise)
        ct_eff = 1 if meta["target"]["kind"] == "gmm" else ct  # mixture targets always run one column tile per workgroup
        assert prob.loss.engine.last_kernel_name() == f"traj_wide<C={meta['net']['channels']},CT={ct_eff}>"
        r2 = prob.eval(x0, compute_weights=False, return_traj=False, noise=noise)
        with torch.no_grad():
            kw = dict(compute_ito_int=True, return_traj=False, noise=noise)
            if meta["loss"]["kind"] == "time_reversal":
                kw["train"] = False
            _, rnd1, _ = prob.loss.simulate(prob.ts, x0, prob.target.unnorm_log_prob, prob.second_log_prob, **kw)
            kw["compute_ito_int"] = False
            _, rnd2, _ = prob.loss.simulate(prob.ts, x0, prob.target.unnorm_log_prob, prob.second_log_prob, **kw)
    _rows("x_T", r1.samples.cpu().numpy(), fx["eval1/x_T"])
    _rows("rnd (ito)", rnd1.cpu().numpy(), fx["eval1/rnd"])
    _rows("rnd", rnd2.cpu().numpy(), fx["eval2/rnd"])
    # SURVEY 8d's bars, as in tests/test_hip_contract.py: estimators 1e-4 absolute (4e-6 relative beyond a magnitude of 25);
    # eval/lv_loss, a variance that weights the rows the contract lets deviate, 1e-4 relative
    for key in ("log_norm_const_lb_ito", "log_norm_const_is"):
        want = float(fx["eval1/" + key])
        assert abs(r1.log_norm_const_preds[key] - want) <= _est_tol(want), (key, r1.log_norm_const_preds[key], want)
    want = float(fx["eval2/log_norm_const_lb"])
    assert abs(r2.log_norm_const_preds["log_norm_const_lb"] - want) <= _est_tol(want)
    lv = float(fx["eval1/lv_loss"])
    assert abs(r1.metrics["eval/lv_loss"] - lv) <= 1e-4 * max(1.0, abs(lv)), (r1.metrics["eval/lv_loss"], lv)
    assert torch.equal(r1.samples, r2.samples)  # the Ito integral only enters rnd
    xs = r1.xs.cpu().numpy()
    assert xs.shape == (prob.ts.numel(), *fx["x0"].shape)
    assert np.array_equal(xs[0], fx["x0"]) and np.array_equal(xs[-1], r1.samples.cpu().numpy())


@pytest.mark.parametrize("path", GOLDEN_WIDE, ids=lambda p: Path(p).stem)
def test_wide_tilings_and_shards_are_bitwise_identical(path):
    """32- and 64-trajectory workgroups run the same arithmetic per trajectory; two shards with row offsets draw the Philox
    stream of the corresponding rows of one launch."""
    fx, meta, params, tt = load_fixture(path)
    prob = hip_problem(meta, params, tt)
    torch.manual_seed(3)
    B = 200
    x0 = prob.prior.sample((B,)).to(DEV)
    eng = prob.loss.engine
    out = {}
    for ct in (1, 2):
        with _ct(ct):
            eng.calls, prob.loss.row_offset = 7, 0
            out[ct] = prob.eval(x0, compute_weights=True)
    assert torch.equal(out[1].samples, out[2].samples) and torch.equal(out[1].weights, out[2].weights)
    eng.calls, prob.loss.row_offset = 7, 0
    a = prob.eval(x0[:72], compute_weights=False)
    eng.calls, prob.loss.row_offset = 7, 72
    b = prob.eval(x0[72:], compute_weights=False)
    assert torch.equal(torch.cat([a.samples, b.samples]), out[1].samples)
    assert torch.isfinite(out[1].samples).all()


def test_wide_fast_mode_agrees_with_oracle_statistics():
    """In-kernel noise at B = 2048 against the oracle with torch noise: the lower bound within 4 standard errors."""
    from oracle import em_oracle as eo

    fx, meta, params, tt = load_fixture([p for p in GOLDEN_WIDE if "pis_funnel100" in p][0])
    prob = hip_problem(meta, params, tt)
    B = 2048
    x0 = torch.zeros(B, meta["target"]["dim"])
    with torch.no_grad():
        _, rnd, _ = prob.loss.simulate(prob.ts, x0.to(DEV), prob.target.unnorm_log_prob, prob.second_log_prob)
    oracle, _ = eo.problem_from_fixture(fx)
    torch.manual_seed(5)
    _, rnd_o, _ = oracle.simulate(torch.from_numpy(fx["ts"]), x0, None)
    g, o = rnd.cpu().double().flatten(), rnd_o.double().flatten()
    se = float(torch.sqrt(g.var() / B + o.var() / B))
    assert abs(float(g.mean() - o.mean())) <= 4 * se + 1e-3, (float(g.mean()), float(o.mean()), se)
    assert 0.8 < float(g.std() / o.std()) < 1.25


def test_wide_mixture_agrees_with_the_narrow_kernels_and_wide_bridge_runs_on_a_mixture():
    from sde_sampler_amd import problems

    # (training through the wide kernels: tests/test_hip_wide_train.py)
    # the headline mixture (GMM-40 d=50) through a 128-channel network runs in the wide kernel ...
    spec = problems.baseline_spec("gmm50_pis_headline")
    spec["net"] = dict(spec["net"], channels=128)
    gm = problems.build(spec, device=DEV)
    out = gm.eval(gm.prior.sample((300,)))
    assert gm.loss.engine.last_kernel_name() == "traj_wide<C=128,CT=1>" and torch.isfinite(out.samples).all()
    # ... and so does a Bridge on a mixture with wide networks (parity: the widebridge_gmm* fixtures below); whatever the number of
    # workgroups sharing a column tile, bit for bit
    lerp = dict(clip_model=10.0, clip_score=10.0, scale_score=1.0, gamma_dim=1, gamma_bias=1.0)
    bspec = dict(batch=64, target=dict(kind="gmm", dim=2, name="fab"), prior=dict(kind="iso_gauss", dim=2),
                 sde=dict(kind="scaled_bm", diff_coeff=1.0, terminal_t=1.0), ctrl=dict(kind="lerp_target", **lerp),
                 inference_ctrl=dict(kind="lerp_prior", **lerp), net=dict(channels=128, num_layers=4, activation="gelu"),
                 loss=dict(kind="time_reversal", method="kl"), grid=dict(start=0.0, end=1.0, steps=8))
    br = problems.build(bspec, device=DEV)
    x0 = br.prior.sample((70,))
    outs = []
    for n in (1, 8):
        with _split(n):
            br.loss.engine.calls = 0
            res = br.eval(x0)
            assert br.loss.engine.last_kernel_name().startswith("bridge_wide<C=128")
            outs.append((res.samples.clone(), res.weights.clone()))
    assert torch.isfinite(outs[0][0]).all() and torch.isfinite(outs[0][1]).all()
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])


def test_wide_bridge_mixture_scratch_larger_than_the_derivative_planes():
    """40 components with an inference network WITHOUT hidden layers (num_layers = 2) at C = 128: the mixture's logits / responsibilities
    (5 K + 1 rows of 32) need more LDS than the one act' plane they share a region with -- the region is sized for the larger of the
    two.  Against the oracle on identical noise."""
    from oracle import em_oracle as eo
    from sde_sampler_amd import problems

    lerp = dict(clip_model=10.0, clip_score=10.0, scale_score=1.0, gamma_dim=1, gamma_bias=1.0)
    spec = dict(batch=40, target=dict(kind="gmm", dim=50, name="fab50"), prior=dict(kind="iso_gauss", dim=50),
                sde=dict(kind="scaled_bm", diff_coeff=1.0, terminal_t=1.0), ctrl=dict(kind="lerp_target", **lerp),
                inference_ctrl=dict(kind="lerp_prior", **lerp), net=dict(channels=128, num_layers=4, activation="gelu"),
                inference_net=dict(channels=128, num_layers=2, activation="gelu"),
                loss=dict(kind="time_reversal", method="kl"), grid=dict(start=0.0, end=1.0, steps=5))
    prob = problems.build(spec)
    inf = prob.loss.inference_ctrl
    params = {k: v.detach().clone() for k, v in prob.ctrl.state_dict().items()}
    params_inf = {k: v.detach().clone() for k, v in inf.state_dict().items()}
    tt = dict(loc=prob.target.loc.clone(), scale=prob.target.scale.clone(), mixture_weights=prob.target.mixture_weights.clone())
    torch.manual_seed(2)
    x0 = prob.prior.sample((40,))
    noise = torch.randn(5, 40, 50)
    ref = eo.Problem(spec, params, tt, params_inf).eval(prob.ts.clone(), x0.clone(), noise, compute_weights=True)
    prob.to(DEV)
    out = prob.eval(x0.to(DEV), compute_weights=True, noise=noise.to(DEV))
    assert prob.loss.engine.last_kernel_name().startswith("bridge_wide<C=128")
    assert (out.samples.cpu() - ref["samples"]).abs().max().item() <= 2e-4 * max(1.0, float(ref["samples"].abs().max()))
    for name in ("log_norm_const_lb_ito", "log_norm_const_is"):
        got, want = out.log_norm_const_preds[name], ref[name]
        assert abs(got - want) <= _est_tol(want), f"{name}: {got} vs {want}"


# ---------------------------------------------------------------------------------------------------------------------------
# Bridge on wide networks (bridge_wide_kernel): exact divergence of the inference control, configs[4] geometry
# ---------------------------------------------------------------------------------------------------------------------------
class _split:
    def __init__(self, n):
        self.n = n

    def __enter__(self):
        if self.n:
            os.environ["SDEH_WIDE_SPLIT"] = str(self.n)

    def __exit__(self, *a):
        os.environ.pop("SDEH_WIDE_SPLIT", None)


def _bridge(path):
    from sde_sampler_amd import problems

    fx, meta, params, tt = load_fixture(path)
    prob = problems.build(meta, params, tt, device=DEV, params_inf=inference_params(fx))
    return fx, meta, prob


@pytest.mark.parametrize("split", [1, 4])
@pytest.mark.parametrize("path", GOLDEN_WIDE_BRIDGE, ids=lambda p: Path(p).stem)
def test_wide_bridge_eval_matches_reference_golden(path, split):
    fx, meta, prob = _bridge(path)
    x0, noise = torch.from_numpy(fx["x0"]).to(DEV), torch.from_numpy(fx["noise"]).to(DEV)
    with _split(split):
        r1 = prob.eval(x0, compute_weights=True, noise=noise)
        assert prob.loss.engine.last_kernel_name() == f"bridge_wide<C={meta['net']['channels']},split={split}>"
        r2 = prob.eval(x0, compute_weights=False, noise=noise)
        with torch.no_grad():
            _, rnd, xs = prob.loss.simulate(prob.ts, x0, prob.target.unnorm_log_prob, prob.second_log_prob, train=False,
                                            compute_ito_int=False, return_traj=True, noise=noise)
    _rows("x_T", r1.samples.cpu().numpy(), fx["eval1/x_T"], max_tol=2e-3)
    # the divergence sums d Jacobian entries of magnitude O(1) per step, accumulated over T steps into rnd
    _rows("rnd", rnd.cpu().numpy(), fx["eval2/rnd"], max_tol=1e-3, med_tol=1e-4)
    for res, key, ref_key in ((r1, "log_norm_const_is", "eval1/log_norm_const_is"), (r1, "log_norm_const_lb_ito", "eval1/log_norm_const_lb_ito"),
                              (r2, "log_norm_const_lb", "eval2/log_norm_const_lb")):
        want = float(fx[ref_key])
        assert abs(res.log_norm_const_preds[key] - want) <= _est_tol(want), (key, res.log_norm_const_preds[key], want)
    assert xs.shape == (prob.ts.numel(), *x0.shape) and torch.equal(xs[-1], r1.samples)


@pytest.mark.parametrize("path", GOLDEN_WIDE_BRIDGE, ids=lambda p: Path(p).stem)
def test_wide_bridge_is_invariant_to_the_workgroup_split_and_to_sharding(path):
    """1, 2, 4 or 8 workgroups per column tile add the same 32 coordinate-group sums in the same order: bitwise equal results;
    two shards with row offsets == one launch."""
    fx, meta, prob = _bridge(path)
    torch.manual_seed(3)
    B = 80
    x0 = prob.prior.sample((B,)).to(DEV)
    eng = prob.loss.engine
    out = {}
    for split in (1, 2, 8):
        with _split(split):
            eng.calls, prob.loss.row_offset = 5, 0
            out[split] = prob.eval(x0, compute_weights=True)
    for split in (2, 8):
        assert torch.equal(out[1].samples, out[split].samples) and torch.equal(out[1].weights, out[split].weights), split
    eng.calls, prob.loss.row_offset = 5, 0
    a = prob.eval(x0[:32], compute_weights=False)
    eng.calls, prob.loss.row_offset = 5, 32
    b = prob.eval(x0[32:], compute_weights=False)
    assert torch.equal(torch.cat([a.samples, b.samples]), out[1].samples)
    assert torch.isfinite(out[1].weights).all()


def _widen(spec, rng):
    """Reshape a random problem of tests/test_hip_fuzz.py into a wide one: 128 / 256 channels (or 64 with d > 64), d = 33 .. 250."""
    c = int(rng.choice([64, 128, 256]))
    d = int(rng.choice([70, 100, 130, 196, 250] if c == 64 else [33, 40, 70, 100, 130, 196, 250]))
    spec["net"]["channels"] = c
    spec["net"]["num_layers"] = int(rng.choice([3, 4]))  # the wide kernels take any depth; keep the oracle's CPU time bounded
    for part in ("target", "prior"):
        if spec[part] is not None and "dim" in spec[part]:
            spec[part]["dim"] = d
    tk = spec["target"]["kind"]
    if tk == "double_well":  # one-dimensional: take its many-dimensional sibling
        spec["target"] = dict(kind="multi_well", dim=d, n_double_wells=int(rng.integers(1, 6)), separation=spec["target"]["separation"],
                              shift=spec["target"]["shift"])
    elif tk == "multi_well":
        spec["target"]["n_double_wells"] = min(spec["target"]["n_double_wells"], d)
    elif tk == "gmm":
        spec["target"]["name"] = "random7"
    if spec["ctrl"].get("gamma_dim", 1) != 1:
        spec["ctrl"]["gamma_dim"] = d
    spec["grid"]["steps"] = min(spec["grid"]["steps"], 12)
    spec["batch"] = int(rng.choice([7, 33, 64, 100]))


@pytest.mark.parametrize("case", range(24 * int(os.environ.get("SDEH_FUZZ_SCALE", "1"))))
def test_random_wide_problem_matches_oracle(case):
    """Seeded random problems (loss x control x SDE x target x clip activity x ragged batch) on WIDE networks through the HIP engine vs
    the CPU oracle on identical noise, with the conditioning-aware criteria of tests/test_hip_fuzz.py."""
    from tests.test_hip_fuzz import check_eval_case

    check_eval_case(3000 + case, spec_hook=_widen, expect_kernel="traj_wide")


def _random_wide_bridge_spec(rng, mixture=False):
    """mixture=False: closed-form targets (the seeds of the first sweeps); True: the same draw with a mixture target put in its place."""
    from tests.test_hip_fuzz import random_spec

    while True:
        spec = random_spec(rng)
        if spec["loss"]["kind"] == "time_reversal" and spec["target"]["kind"] != "gmm":
            break
    c = int(rng.choice([128, 256]))
    d = int(rng.choice([33, 44, 70, 100, 150, 196]))
    if mixture:
        d = int(rng.choice([33, 50, 70, 100]))
        spec["target"] = dict(kind="gmm", dim=d, name="fab50" if d == 50 else "random7")
    for part in ("target", "prior"):
        if spec[part] is not None and "dim" in spec[part]:
            spec[part]["dim"] = d
    if spec["target"]["kind"] == "double_well":
        spec["target"] = dict(kind="multi_well", dim=d, n_double_wells=int(rng.integers(1, 6)), separation=spec["target"]["separation"],
                              shift=spec["target"]["shift"])
    elif spec["target"]["kind"] == "multi_well":
        spec["target"]["n_double_wells"] = min(spec["target"]["n_double_wells"], d)
    if spec["ctrl"].get("gamma_dim", 1) != 1:
        spec["ctrl"]["gamma_dim"] = d
    clip_active = rng.random() < 0.4
    inf = dict(kind=str(rng.choice(["lerp_prior", "clipped"])), clip_model=float(rng.uniform(0.02, 0.5)) if clip_active else 1e4)
    if inf["kind"] == "lerp_prior":
        inf.update(clip_score=float(rng.uniform(0.5, 3.0)) if clip_active else 1e4, scale_score=float(rng.choice([1.0, 0.5])),
                   gamma_dim=int(rng.choice([1, d])), gamma_bias=1.0)
    spec["inference_ctrl"] = inf
    spec["net"] = dict(channels=c, num_layers=int(rng.choice([3, 4])), activation=spec["net"]["activation"])
    spec["inference_net"] = dict(channels=c, num_layers=int(rng.choice([2, 3, 4])), activation=str(rng.choice(["gelu", "silu", "relu"])))
    spec["grid"]["steps"] = int(rng.integers(2, 6))
    spec["batch"] = int(rng.choice([5, 33, 40]))
    spec["loss"].update(method="kl", max_rnd=None)
    return spec


_BRIDGE_SWEEP = ([(c, False) for c in range(12 * int(os.environ.get("SDEH_FUZZ_SCALE", "1")))] +
                 [(c, True) for c in range(500, 500 + 6 * int(os.environ.get("SDEH_FUZZ_SCALE", "1")))])


@pytest.mark.parametrize("case,mixture", _BRIDGE_SWEEP, ids=lambda v: str(v))
def test_random_wide_bridge_matches_oracle(case, mixture):
    """Random Bridges (TimeReversalLoss with an inference control, exact divergence) on wide networks: x_T rows and the estimators
    against the oracle (d backward passes per step through the inference network) on identical noise.  Cases 500+: mixture targets."""
    import math

    from oracle import em_oracle as eo
    from sde_sampler_amd import problems
    from tests.test_hip_fuzz import _close, _perturbed

    rng = np.random.default_rng(7000 + case)
    spec = _random_wide_bridge_spec(rng, mixture)
    prob = problems.build(spec)
    inf = prob.loss.inference_ctrl
    params = {k: v.detach().clone() for k, v in prob.ctrl.state_dict().items()}
    params_inf = {k: v.detach().clone() for k, v in inf.state_dict().items()}
    tt = None
    if spec["target"]["kind"] == "gmm":
        tt = dict(loc=prob.target.loc.clone(), scale=prob.target.scale.clone(), mixture_weights=prob.target.mixture_weights.clone())
    oracle = eo.Problem(spec, params, tt, params_inf)
    ts = prob.ts.clone()
    B, d, T = spec["batch"], spec["target"]["dim"], ts.numel() - 1
    torch.manual_seed(case)
    x0 = prob.prior.sample((B,))
    noise = torch.randn(T, B, d)
    torch.set_num_threads(4)
    ref = oracle.eval(ts, x0.clone(), noise, compute_weights=True)
    ref_p = oracle.eval(ts, *_perturbed(x0, noise), compute_weights=True)
    tag = (f"case {case}: wide bridge {spec['ctrl']['kind']} + {spec['inference_ctrl']['kind']} / {spec['target']['kind']} d={d} B={B} T={T} "
           f"{spec['net']} inf {spec['inference_net']}")
    prob.to(DEV)
    out = prob.eval(x0.to(DEV), compute_weights=True, noise=noise.to(DEV))
    assert prob.loss.engine.last_kernel_name().startswith("bridge_wide"), tag
    if not math.isfinite(ref["log_norm_const_lb_ito"]):  # a random configuration that blows up in the reference itself
        assert not math.isfinite(out.log_norm_const_preds["log_norm_const_lb_ito"]), tag
        return
    cond_rows = torch.nan_to_num((ref_p["samples"] - ref["samples"]).abs().amax(dim=1), nan=math.inf)
    cond_lb = abs(ref_p["log_norm_const_lb_ito"] - ref["log_norm_const_lb_ito"]) if math.isfinite(ref_p["log_norm_const_lb_ito"]) else math.inf
    row_err = ((out.samples.cpu() - ref["samples"]).abs().amax(dim=1) - cond_rows).clamp_min(0.0)
    scale = max(1.0, float(ref["samples"].abs().max()))
    assert row_err.median().item() <= 1e-4 * scale and (row_err > 2e-3 * scale).float().mean().item() <= DRIFT_MAX, f"{tag}: x_T {row_err.max().item():.2e}"
    got, want = out.log_norm_const_preds["log_norm_const_lb_ito"], ref["log_norm_const_lb_ito"]
    assert fuzz_close("wide_bridge/lb_ito", got, want, 2.0 * cond_lb), f"{tag}: lb_ito {got} vs {want}"

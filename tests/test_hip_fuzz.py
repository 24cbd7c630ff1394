"""Randomised parity sweep: seeded random problems (loss x control x SDE x target x network shape x clip activity x batch
raggedness) through the HIP engine vs the CPU oracle on identical noise.  The golden fixtures pin nine hand-picked
configurations; this sweeps the combinations in between (and the kernel-variant selection that goes with them)."""
import math
import os

import numpy as np
import pytest
import torch

from oracle import em_oracle as eo
from tests.helpers import fuzz_close, hatch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
_SCALE = int(os.environ.get("SDEH_FUZZ_SCALE", "1"))  # SDEH_FUZZ_SCALE=4: an occasional wider sweep (same seeds + more)
N_CASES = 96 * _SCALE
N_TRAIN = 32 * _SCALE
N_BRIDGE = 24 * _SCALE
N_INT = 32 * _SCALE
#: largest fraction of the rows of a case that may still differ by more than 2e-3 x scale AFTER the oracle's own response to a
#: one-in-a-million perturbation of the inputs has been subtracted (set from the observed distribution, profiles/r02_fuzz_drift.txt)
#: Observed over 1532 cases (SDEH_FUZZ_SCALE=16): 1523 have no such row at all, 7 have <= 2 %, two small-batch cases with stiff clamped
#: wells have 11 % and 18 % under the three standard probes -- those are re-examined with eight more probes before the verdict.
DRIFT_MAX = float(os.environ.get("SDEH_FUZZ_DRIFT_MAX", "0.05"))
_MORE_PERTS = (2e-6, -2e-6, -4e-6, 8e-6, -8e-6, 5e-7, -5e-7, 1.6e-5)


def random_spec(rng: np.random.Generator) -> dict:
    d = int(rng.choice([1, 2, 3, 5, 8, 10, 16, 50]))
    loss_kind = str(rng.choice(["time_reversal", "reference_sde", "exponential"]))
    # target
    tkinds = ["gmm", "iso_gauss", "funnel", "multi_well"] if d >= 2 else ["double_well", "iso_gauss", "multi_well"]
    tk = str(rng.choice(tkinds))
    if tk == "gmm":
        target = dict(kind="gmm", dim=d, name="fab" if d == 2 and rng.random() < 0.5 else ("fab50" if d == 50 else "random7"))
        if target["name"] == "fab50" and d != 50:
            target["name"] = "random7"
    elif tk == "iso_gauss":
        target = dict(kind="iso_gauss", dim=d, loc=float(rng.uniform(-2, 2)), scale=float(rng.uniform(0.5, 2.0)))
    elif tk == "funnel":
        target = dict(kind="funnel", dim=d)
    elif tk == "double_well":
        target = dict(kind="double_well", dim=1, separation=float(rng.uniform(1, 3)), shift=float(rng.uniform(-1, 1)))
    else:
        target = dict(kind="multi_well", dim=d, n_double_wells=int(rng.integers(1, d + 1)), separation=float(rng.uniform(1, 3)),
                      shift=float(rng.uniform(-0.5, 0.5)))
    # sde / prior / control kinds that make sense for the loss (solver/oc.py)
    if loss_kind == "exponential":
        sde, prior = None, dict(kind="iso_gauss", dim=d, loc=0.0, scale=1.0)
        ctrl_kind = str(rng.choice(["score", "clipped"]))
    else:
        sk = str(rng.choice(["vp", "const_ou", "scaled_bm"]))
        if sk == "vp":
            sde = dict(kind="vp", beta_min=float(rng.uniform(0.05, 0.5)), beta_max=float(rng.uniform(2, 8)), scale=float(rng.uniform(0.7, 1.3)),
                       terminal_t=float(rng.choice([1.0, 2.0])))
        elif sk == "const_ou":
            sde = dict(kind="const_ou", drift_coeff=float(rng.uniform(0.2, 1.5)), diff_coeff=float(rng.uniform(0.5, 1.5)), terminal_t=1.0)
        else:
            sde = dict(kind="scaled_bm", diff_coeff=float(rng.uniform(0.4, 1.5)), terminal_t=float(rng.choice([1.0, 5.0])))
        if loss_kind == "reference_sde" and sk == "vp":
            sde["scale"] = 1.0  # EulerDDS: the reference Gaussian's variance (1 - e^{2I}) scale^2 + e^{2I} must stay positive
        if loss_kind == "reference_sde" and sk != "vp" and rng.random() < 0.5:
            prior, ctrl_kind = dict(kind="delta", dim=d), str(rng.choice(["score", "clipped"]))  # PIS
        else:
            prior = dict(kind="iso_gauss", dim=d, loc=0.0, scale=1.0)
            ctrl_kind = str(rng.choice(["score", "clipped", "lerp", "lerp_target", "lerp_prior"]))
    clip_active = rng.random() < 0.4
    ctrl = dict(kind=ctrl_kind, clip_model=float(rng.uniform(0.2, 2.0)) if clip_active else 1e4)
    if ctrl_kind != "clipped":
        ctrl.update(clip_score=float(rng.uniform(0.5, 5.0)) if clip_active else 1e4, scale_score=float(rng.choice([1.0, 0.5, 2.0])),
                    gamma_dim=int(rng.choice([1, d])), gamma_bias=float(rng.choice([1.0, 0.01])))
    loss = dict(kind=loss_kind, method="kl", max_rnd=None)
    if loss_kind == "exponential":
        loss.update(alpha=float(rng.uniform(0.5, 1.5)), sigma=float(rng.uniform(0.7, 1.3)))
    if loss_kind == "reference_sde" and prior["kind"] != "delta":
        loss["reference_ctrl"] = "prior_score"  # EulerDDS
    steps = int(rng.integers(4, 33))
    end = sde["terminal_t"] if sde else float(rng.uniform(3.0, 12.8))
    grid = dict(start=0.0, end=end, steps=steps, rescale_t="cosine" if loss_kind == "exponential" and rng.random() < 0.5 else None)
    net = dict(channels=64, num_layers=int(rng.integers(3, 6)), activation=str(rng.choice(["gelu", "silu", "relu"])))
    return dict(target=target, prior=prior, sde=sde, ctrl=ctrl, net=net, loss=loss, grid=grid,
                batch=int(rng.choice([1, 7, 33, 64, 65, 100, 257, 300])), init_seed=int(rng.integers(1, 1000)))


_PERTS = (1e-6, -1e-6, 4e-6)


def _perturbed(x0, noise, eps=_PERTS[0]):
    """The conditioning probe: the same problem with inputs moved by one part in a million.  Whatever this does to the REFERENCE's own
    trajectories / estimators / gradients is not a meaningful difference between two implementations either (rounding enters at
    1e-7 per operation and is amplified the same way); stiff wells with large steps and active clamps amplify by 1e6 and more.
    Three probes (the clamps make the response discontinuous), the largest response counts."""
    return x0 * (1.0 + eps), noise * (1.0 + eps)


def _close(got: float, want: float, tol: float) -> bool:
    if not math.isfinite(want):
        return not math.isfinite(got)  # blown up in the reference itself: blown up here as well (inf / nan alike)
    return math.isfinite(got) and abs(got - want) <= tol


@pytest.mark.parametrize("case", range(N_CASES))
def test_random_problem_matches_oracle(case):
    check_eval_case(case)


def check_eval_case(case, spec_hook=None, expect_kernel=None):
    """One random evaluation problem against the oracle on identical noise.  `spec_hook(spec, rng)` may reshape the problem (the wide
    sweep of tests/test_hip_wide.py: 128 / 256 channels, d up to 250); `expect_kernel`: prefix of the kernel that must serve it."""
    from sde_sampler_amd import problems

    rng = np.random.default_rng(1000 + case)
    spec = random_spec(rng)
    if spec_hook is not None:
        spec_hook(spec, rng)
    prob = problems.build(spec)
    params = {k: v.detach().clone() for k, v in prob.ctrl.state_dict().items()}
    tt = None
    if spec["target"]["kind"] == "gmm":
        tt = dict(loc=prob.target.loc.clone(), scale=prob.target.scale.clone(), mixture_weights=prob.target.mixture_weights.clone())
    oracle = eo.Problem(spec, params, tt)
    ts = prob.ts.clone()
    B, d, T = spec["batch"], spec["target"]["dim"], ts.numel() - 1
    torch.manual_seed(case)
    x0 = prob.prior.sample((B,))
    noise = torch.randn(T, B, d)
    weights = bool(rng.random() < 0.5)
    torch.set_num_threads(4)
    try:
        ref = oracle.eval(ts, x0.clone(), noise, compute_weights=weights, return_traj=True)
        probes = [oracle.eval(ts, *_perturbed(x0, noise, eps), compute_weights=weights, return_traj=True) for eps in _PERTS]
    except ValueError as exc:  # torch.distributions' support check on non-finite states: the configuration blows up in the reference
        hatch("eval:reference_rejects_configuration", f"case {case}")
        pytest.skip(f"random configuration rejected by the reference's own distribution checks: {str(exc)[:80]}")
    cond_rows = torch.stack([torch.nan_to_num((q["xs"] - ref["xs"]).abs().amax(dim=(0, 2)), nan=math.inf) for q in probes]).amax(dim=0)

    def cond_of(key):
        return max((abs(q[key] - ref[key]) if math.isfinite(q[key]) and math.isfinite(ref[key]) else math.inf) for q in probes)

    prob.to(DEV)
    out = prob.eval(x0.to(DEV), compute_weights=weights, return_traj=True, noise=noise.to(DEV))
    tag = f"case {case}: {spec['loss']['kind']} / {spec['ctrl']['kind']} / {spec['target']['kind']} d={d} B={B} T={T} {spec['net']}"
    if expect_kernel is not None:
        assert prob.loss.engine.last_kernel_name().startswith(expect_kernel), f"{tag}: {prob.loss.engine.last_kernel_name()}"
    # Per-row criterion: the dynamics amplify 1-ulp differences -- stiff wells with large steps and active clamps
    # can take single rows from 2e-6 to 0.2 within 20 steps (tests/perf/fuzz_case_debug.py shows the step-by-step growth of such a
    # case) -- so the bulk of the rows must agree tightly and only a minority may have drifted.
    scale = max(1.0, float(torch.nan_to_num(ref["xs"], nan=0.0, posinf=0.0, neginf=0.0).abs().max()))
    row_err = torch.nan_to_num((out.xs.cpu() - ref["xs"]).abs().amax(dim=(0, 2)), nan=0.0, posinf=0.0)  # non-finite rows: see estimators
    raw_err = row_err
    row_err = (raw_err - cond_rows).clamp_min(0.0)  # beyond what the conditioning of the row explains
    if row_err.median().item() > 1e-4 * scale:
        # the BULK of the rows is off: either a real discrepancy, or a configuration whose reference trajectories are chaotic as a
        # whole (case 2716 of the scale-32 sweep: exponential integrator with steps of 1.4 on a mixture, |x| up to 1e3 -- the
        # oracle's own rows move by 1 .. 20 under the eleven one-in-a-million probes).  Look with all probes; if the reference's
        # median response is itself beyond 1e-3 of the scale there is nothing to compare.
        more = [oracle.eval(ts, *_perturbed(x0, noise, eps), compute_weights=weights, return_traj=True) for eps in _MORE_PERTS]
        cond_all = torch.maximum(cond_rows, torch.stack([torch.nan_to_num((q["xs"] - ref["xs"]).abs().amax(dim=(0, 2)), nan=math.inf)
                                                          for q in more]).amax(dim=0))
        hatch("eval:bulk_reprobed", tag)
        if cond_all.median().item() > 1e-3 * scale:
            hatch("eval:chaotic_reference_skip", tag)
            pytest.skip(f"{tag}: chaotic in the reference itself (median response {cond_all.median().item():.2e} to 1e-6 probes, scale {scale:.1f})")
        row_err = (raw_err - cond_all).clamp_min(0.0)
    assert row_err.median().item() <= 1e-4 * scale, f"{tag}: median row error {row_err.median().item():.3e} (scale {scale:.2f})"

    def drift(cond):
        # rows whose REFERENCE trajectory moves by more than 5 % of the state scale under a one-in-a-million perturbation are
        # chaotic in the reference itself (case 432: a stiff double well under the exponential integrator, differences grow from 1e-6
        # to the size of the attractor within 20 steps, for the perturbed oracle exactly as for the kernel): subtracting two O(scale)
        # numbers says nothing, such rows are not counted either way
        chaotic = cond >= 0.05 * scale
        return ((((raw_err - cond).clamp_min(0.0) > 2e-3 * scale) & ~chaotic).float().mean().item(), int(chaotic.sum()))

    drifted, n_chaotic = drift(cond_rows)
    if drifted > DRIFT_MAX:  # three probes under-estimate the conditioning of a discontinuous (clamped) map: look again with eight more
        hatch("eval:drift_reprobed", tag)
        more = [oracle.eval(ts, *_perturbed(x0, noise, eps), compute_weights=weights, return_traj=True) for eps in _MORE_PERTS]
        cond_more = torch.stack([torch.nan_to_num((q["xs"] - ref["xs"]).abs().amax(dim=(0, 2)), nan=math.inf) for q in more]).amax(dim=0)
        drifted, n_chaotic = drift(torch.maximum(cond_rows, cond_more))
    if os.environ.get("SDEH_FUZZ_REPORT"):  # observed distribution of the criteria (profiles/r02_fuzz_drift.txt)
        with open(os.environ["SDEH_FUZZ_REPORT"], "a") as fh:
            fh.write(f"{case} B={B} T={T} d={d} median={row_err.median().item() / scale:.3e} max={row_err.max().item() / scale:.3e} "
                     f"drifted={drifted:.4f} chaotic_rows={n_chaotic}\n")
    if drifted > 0.0:
        hatch("eval:rows_beyond_bar_within_allowance", f"{tag}: {drifted:.4f}")
    assert drifted <= DRIFT_MAX, f"{tag}: {drifted:.0%} of the rows differ by more than {2e-3 * scale:.1e}"
    assert (out.xs[0].cpu() == ref["xs"][0]).all()  # the initial state is passed through
    key = "log_norm_const_lb_ito" if weights else "log_norm_const_lb"
    got, want = out.log_norm_const_preds[key], ref[key]
    if math.isfinite(want) and abs(want) > 1e8:
        # quartic wells far from the origin: costs of 1e9 .. 1e14 whose rows answer a 1e-6 probe with changes of 1e8 -- the
        # reference's trajectories have exploded (finite by luck); rows were compared above, the estimator only in magnitude
        hatch("eval:exploded_reference_estimator", tag)
        assert (not math.isfinite(got)) or abs(got) > 1e6, f"{tag}: {key} {got} vs {want}"
        return
    cond = cond_of(key)
    assert fuzz_close(f"eval/{key}", got, want, 2.0 * cond), f"{tag}: {key} {got} vs {want} (conditioning {cond:.2e})"
    if weights and math.isfinite(ref[key]):  # (non-finite rows: overflow shows as +inf or as nan depending on the order of operations)
        got, want = out.log_norm_const_preds["log_norm_const_is"], ref["log_norm_const_is"]
        assert _close(got, want, 5e-3 * max(1.0, abs(want)) + cond_of("log_norm_const_is")), f"{tag}: log_norm_const_is {got} vs {want}"


@pytest.mark.parametrize("case", range(N_TRAIN))
def test_random_training_gradients_match_oracle(case):
    """loss(...).backward() through the HIP forward + backward kernels vs the oracle's autograd, methods kl / kl_ito / lv."""
    check_training_case(case)


def check_training_case(case, num_layers=None, expect_kernel=None, spec_hook=None, relu_tol_scale=1.0):
    """One random training problem against the oracle's autograd.  `num_layers`: force the network depth (4 = the depth the fused
    backward kernel is compiled for); `expect_kernel`: prefix the backward kernel's name must have; `spec_hook(spec, rng)`: reshape
    the random problem (the wide-network sweep of tests/test_hip_wide_train.py)."""
    from sde_sampler_amd import problems

    rng = np.random.default_rng(5000 + case)
    spec = random_spec(rng)
    if num_layers is not None:
        spec["net"]["num_layers"] = num_layers
    method = str(rng.choice(["kl", "kl_ito", "lv"]))
    spec["loss"]["method"] = method
    spec["loss"]["max_rnd"] = 1e8 if method == "lv" else None
    spec["batch"] = int(rng.choice([33, 64, 100]))  # at least two rows for the variance
    if spec_hook is not None:
        spec_hook(spec, rng)
    prob = problems.build(spec)
    params = {k: v.detach().clone().requires_grad_(v.is_floating_point()) for k, v in prob.ctrl.state_dict().items()}
    tt = None
    if spec["target"]["kind"] == "gmm":
        tt = dict(loc=prob.target.loc.clone(), scale=prob.target.scale.clone(), mixture_weights=prob.target.mixture_weights.clone())
    oracle = eo.Problem(spec, params, tt)
    ts = prob.ts.clone()
    B, d, T = spec["batch"], spec["target"]["dim"], ts.numel() - 1
    torch.manual_seed(case)
    x0 = prob.prior.sample((B,))
    noise = torch.randn(T, B, d)
    torch.set_num_threads(4)
    try:
        ref_loss, _, _, _ = oracle.train_loss(ts, x0.clone(), noise, method=method)
    except ValueError as exc:  # torch.distributions' support check on non-finite states: the configuration blows up in the reference
        hatch("train:reference_rejects_configuration", f"case {case}")
        pytest.skip(f"random configuration rejected by the reference's own distribution checks: {str(exc)[:80]}")
    ref_loss.backward()
    # conditioning probe: loss and gradients of the reference at inputs moved by 1e-6
    cond_loss, cond_grad = 0.0, {k: 0.0 for k in params}
    for eps in _PERTS:
        params_p = {k: v.detach().clone().requires_grad_(v.is_floating_point()) for k, v in params.items()}
        try:
            loss_p, _, _, _ = eo.Problem(spec, params_p, tt).train_loss(ts, *_perturbed(x0, noise, eps), method=method)
        except ValueError:
            cond_loss = math.inf
            continue
        loss_p.backward()
        cond_loss = max(cond_loss, abs(loss_p.item() - ref_loss.item()) if math.isfinite(loss_p.item()) else math.inf)
        for k, v in params.items():
            if v.grad is not None and params_p[k].grad is not None:
                cond_grad[k] = max(cond_grad[k], float(torch.nan_to_num((params_p[k].grad - v.grad).abs(), nan=math.inf).max()))
    prob.to(DEV)
    from sde_sampler_amd import SdehUnsupported

    try:
        val, _ = prob.loss(prob.ts, x0.to(DEV), prob.target.unnorm_log_prob, prob.second_log_prob, noise=noise.to(DEV))
        val.backward()
        kernel = prob.loss.engine.last_kernel_name()
        if expect_kernel is not None and not kernel.startswith("traj_legacy"):  # (legacy forward: mixture tables beyond LDS, plane path)
            assert kernel.startswith(expect_kernel), kernel
    except SdehUnsupported as exc:  # a documented limit, e.g. a wide mixture next to the transposed weights in LDS
        if "do not fit in LDS" in str(exc):
            hatch("train:lds_table_limit_skip", f"case {case}")
            pytest.skip(str(exc)[:120])
        raise
    tag = f"case {case}: {method} {spec['loss']['kind']} / {spec['ctrl']['kind']} / {spec['target']['kind']} d={d} B={B} T={T}"
    if math.isfinite(ref_loss.item()) and abs(ref_loss.item()) > 1e8:
        # Large losses are compared like any other (relative bar + the reference's own response to the probes; the float64 oracle
        # arbitrates the gradients below) -- unless the reference is not comparable with ITSELF: a loss beyond 1e15, or one that a
        # one-in-a-million probe moves by more than 5 % (its trajectories have exploded and are finite by luck).  Then only the
        # magnitude is checked.  (the shortcut used to take every loss above 1e8.)
        if abs(ref_loss.item()) > 1e15 or not cond_loss <= 0.05 * abs(ref_loss.item()):
            hatch("train:exploded_reference_loss", tag)
            assert not math.isfinite(val.item()) or abs(val.item()) > 1e-2 * abs(ref_loss.item()), f"{tag}: loss {val.item()} vs {ref_loss.item()}"
            return
        hatch("train:large_loss_compared_relatively", tag)
    assert fuzz_close("train/loss", val.item(), ref_loss.item(), cond_loss), f"{tag}: loss {val.item()} vs {ref_loss.item()}"
    if not math.isfinite(ref_loss.item()):
        hatch("train:nonfinite_reference_loss", tag)
        return
    gmax = max((torch.nan_to_num(p.grad).abs().max().item() for p in params.values() if p.grad is not None), default=0.0)
    g64 = None
    for k, p in prob.ctrl.named_parameters():
        g_ref = params[k].grad
        if g_ref is None:
            continue
        g = p.grad.cpu() if p.grad is not None else torch.zeros_like(g_ref)
        # relative to the tensor's own scale, with a floor at 1e-4 of the largest gradient of the network (tiny gradients of
        # e.g. a clamped gamma carry only rounding noise)
        if not torch.isfinite(g_ref).all():
            hatch("train:nonfinite_reference_gradient", f"{tag}: {k}")
            continue  # the reference's own gradient is not finite for this random configuration
        denom = max(g_ref.abs().max().item(), 1e-4 * gmax, 1e-12)
        if cond_grad.get(k, 0.0) > 0.5 * denom:
            hatch("train:gradient_ill_conditioned_in_reference", f"{tag}: {k}")
            continue  # a 1e-6 change of the inputs moves the reference's own gradient by more than half of its size: nothing to compare
        err = max((g - g_ref).abs().max().item() - cond_grad.get(k, 0.0), 0.0) / denom
        tol = _grad_tol(spec["net"], k) * (relu_tol_scale if spec["net"].get("activation") == "relu" else 1.0)
        if err > tol:
            # before failing: is the fp32 ORACLE the inaccurate side?  (losses of 1e4 .. 1e11: its unrolled autograd accumulates
            # rounding that the +-1e-6 input probes do not show -- case 11224 of the wide sweep: library 12 x closer to float64.)  The
            # float64 run of the oracle decides: the library must be within the bar of float64, or at least as close to it as the
            # fp32 oracle is.
            hatch("train:float64_arbitration", f"{tag}: {k} {err:.2e}")
            if g64 is None:
                g64 = _float64_grads(spec, params, tt, ts, x0, noise, method)
            if g64 is not None and g64.get(k) is not None:
                e_hip = (g.double() - g64[k]).abs().max().item() / denom
                e_ref = (g_ref.double() - g64[k]).abs().max().item() / denom
                if e_hip <= max(tol, 1.5 * e_ref):
                    continue
                err = e_hip
        assert err <= tol, f"{tag}: grad {k} rel err {err:.2e} (conditioning {cond_grad.get(k, 0.0) / denom:.1e})"


def _float64_grads(spec, params, tt, ts, x0, noise, method):
    """{name: gradient} of the oracle run in float64 on the same inputs (None when that run fails)."""
    p64 = {k: v.detach().double().clone().requires_grad_(v.is_floating_point()) for k, v in params.items()}
    tt64 = None if tt is None else {k: v.double() for k, v in tt.items()}
    try:
        loss, _, _, _ = eo.Problem(spec, p64, tt64).train_loss(ts.double(), x0.double(), noise.double(), method=method)
        loss.backward()
    except Exception:  # noqa: BLE001 -- an oracle that cannot run in float64 just leaves the fp32 comparison standing
        return None
    return {k: v.grad for k, v in p64.items()}


def random_bridge_spec(rng: np.random.Generator) -> dict:
    while True:
        spec = random_spec(rng)
        if spec["loss"]["kind"] == "time_reversal" and spec["target"]["dim"] <= 10:
            break
    d = spec["target"]["dim"]
    clip_active = rng.random() < 0.4
    inf = dict(kind=str(rng.choice(["lerp_prior", "clipped"])), clip_model=float(rng.uniform(0.02, 0.5)) if clip_active else 1e4)
    if inf["kind"] == "lerp_prior":
        inf.update(clip_score=float(rng.uniform(0.5, 3.0)) if clip_active else 1e4, scale_score=float(rng.choice([1.0, 0.5])),
                   gamma_dim=int(rng.choice([1, d])), gamma_bias=1.0)
    spec["inference_ctrl"] = inf
    spec["inference_net"] = dict(channels=64, num_layers=int(rng.integers(3, 6)), activation=str(rng.choice(["gelu", "silu", "relu"])))
    spec["grid"]["steps"] = int(rng.integers(4, 17))
    spec["batch"] = int(rng.choice([33, 64, 100]))
    return spec


def _grad_tol(net_spec: dict, name: str, n_steps: int | None = None) -> float:
    """5e-3 of the largest entry -- except for ReLU networks: ONE pre-activation sitting on the kink (|z| ~ 1e-8: its sign is decided
    by the summation order of the fp32 GEMM, which no two implementations share) switches a unit on or off for a row, and every
    gradient below that layer moves by that row's share -- a few per cent among the T*B rows of the main network, ~1/T for the two
    time-only sub-networks whose tables have only T rows (against float64 the library is the accurate side there,
    tests/test_hip_tembed.py; the layers above the flipped one still agree to 1e-6 in such cases)."""
    if net_spec.get("activation") != "relu":
        return 5e-3
    if "timestep_embed" in name or "score_model" in name:
        # the time-only networks see T rows: one flipped unit of one row is ~1/T of a gradient (Bridge case 671 of the scale-32 sweep:
        # T = 8, 0.12 .. 0.16 in the gamma network's layers, float64 on neither side's side of the kink)
        return max(0.12, 1.5 / n_steps) if n_steps else 0.12
    return 5e-2


@pytest.mark.parametrize("case", range(N_BRIDGE))
def test_random_bridge_matches_oracle(case):
    """Bridge (TimeReversalLoss with an inference control): evaluation and training gradients of both networks."""
    from sde_sampler_amd import problems

    rng = np.random.default_rng(9000 + case)
    spec = random_bridge_spec(rng)
    method = str(rng.choice(["kl", "kl_ito", "lv"]))
    spec["loss"].update(method=method, max_rnd=1e8 if method == "lv" else None)
    prob = problems.build(spec)
    inf = prob.loss.inference_ctrl
    leaf = lambda sd: {k: v.detach().clone().requires_grad_(v.is_floating_point()) for k, v in sd.items()}
    params, params_inf = leaf(prob.ctrl.state_dict()), leaf(inf.state_dict())
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
    if not math.isfinite(ref["log_norm_const_lb_ito"]):  # a random configuration that blows up in the reference itself
        hatch("bridge:nonfinite_reference", f"case {case}")
        prob.to(DEV)
        out = prob.eval(x0.to(DEV), compute_weights=True, noise=noise.to(DEV))
        assert not math.isfinite(out.log_norm_const_preds["log_norm_const_lb_ito"])
        return
    ref_loss, _, _, _ = oracle.train_loss(ts, x0.clone(), noise, method=method)
    ref_loss.backward()
    params_p, params_inf_p = leaf(params), leaf(params_inf)
    oracle_p = eo.Problem(spec, params_p, tt, params_inf_p)
    ref_p = oracle_p.eval(ts, *_perturbed(x0, noise), compute_weights=True)
    loss_p, _, _, _ = oracle_p.train_loss(ts, *_perturbed(x0, noise), method=method)
    loss_p.backward()
    cond_loss = abs(loss_p.item() - ref_loss.item()) if math.isfinite(loss_p.item()) else math.inf
    cond_rows = torch.nan_to_num((ref_p["samples"] - ref["samples"]).abs().amax(dim=1), nan=math.inf)
    cond_lb = abs(ref_p["log_norm_const_lb_ito"] - ref["log_norm_const_lb_ito"]) if math.isfinite(ref_p["log_norm_const_lb_ito"]) else math.inf
    prob.to(DEV)
    tag = f"case {case}: bridge {method} / {spec['ctrl']['kind']} + {spec['inference_ctrl']['kind']} / {spec['target']['kind']} d={d} B={B} T={T}"
    out = prob.eval(x0.to(DEV), compute_weights=True, noise=noise.to(DEV))
    row_err = ((out.samples.cpu() - ref["samples"]).abs().amax(dim=1) - cond_rows).clamp_min(0.0)
    scale = max(1.0, float(ref["samples"].abs().max()))
    # (one conditioning probe only in this sweep -- the oracle's exact divergence is the slow side: 10 % of the rows may sit beyond the
    # per-row bar; measured: 1 case of 384 above 5 %)
    assert row_err.median().item() <= 1e-4 * scale and (row_err > 2e-3 * scale).float().mean().item() <= 2 * DRIFT_MAX, f"{tag}: x_T"
    got, want = out.log_norm_const_preds["log_norm_const_lb_ito"], ref["log_norm_const_lb_ito"]
    assert fuzz_close("bridge/lb_ito", got, want, cond_lb), f"{tag}: lb_ito {got} vs {want}"
    val, _ = prob.loss(prob.ts, x0.to(DEV), prob.target.unnorm_log_prob, prob.second_log_prob, noise=noise.to(DEV))
    val.backward()
    if math.isfinite(ref_loss.item()) and abs(ref_loss.item()) > 1e8 and (abs(ref_loss.item()) > 1e15 or not cond_loss <= 0.05 * abs(ref_loss.item())):
        hatch("bridge:exploded_reference_loss", tag)  # (not comparable with itself: see check_training_case)
        assert not math.isfinite(val.item()) or abs(val.item()) > 1e-2 * abs(ref_loss.item()), f"{tag}: loss {val.item()} vs {ref_loss.item()}"
        return
    assert fuzz_close("bridge/loss", val.item(), ref_loss.item(), cond_loss), f"{tag}: loss {val.item()} vs {ref_loss.item()}"
    if not math.isfinite(ref_loss.item()):
        hatch("bridge:nonfinite_reference_loss", tag)
        return
    for mod, pd, pd_p in ((prob.ctrl, params, params_p), (inf, params_inf, params_inf_p)):
        gmax = max((torch.nan_to_num(p.grad).abs().max().item() for p in pd.values() if p.grad is not None), default=0.0)
        for k, p in mod.named_parameters():
            g_ref = pd[k].grad
            if g_ref is None or not torch.isfinite(g_ref).all():
                continue
            g = p.grad.cpu() if p.grad is not None else torch.zeros_like(g_ref)
            cond = float(torch.nan_to_num((pd_p[k].grad - g_ref).abs(), nan=math.inf).max()) if pd_p[k].grad is not None else 0.0
            err = max((g - g_ref).abs().max().item() - cond, 0.0) / max(g_ref.abs().max().item(), 1e-4 * gmax, 1e-12)
            net_spec = spec["net"] if mod is prob.ctrl else spec["inference_net"]
            assert err <= _grad_tol(net_spec, k, T), f"{tag}: grad {k} rel err {err:.2e}"


@pytest.mark.parametrize("case", range(N_INT))
def test_random_integration_matches_oracle(case):
    """EulerIntegrator.integrate: random SDE class (Langevin / bare OU / ControlledSDE, generative or inference clock), random
    integration grid and random -- mostly off-grid, sometimes repeated -- output times."""
    from sde_sampler_amd import problems
    from sde_sampler_amd.eq.integrator import EulerIntegrator

    rng = np.random.default_rng(13000 + case)
    base = random_spec(rng)
    while base["sde"] is None:
        base = random_spec(rng)
    d = base["target"]["dim"]
    steps = int(rng.integers(3, 25))
    end = float(base["sde"]["terminal_t"])
    meta = dict(target=base["target"], prior=dict(kind="iso_gauss", dim=d, loc=0.0, scale=1.0), grid=dict(start=0.0, end=end, steps=steps))
    mode = str(rng.choice(["langevin", "ou", "controlled"]))
    if mode == "langevin":
        meta["integrate"] = dict(kind="langevin", diff_coeff=float(rng.uniform(0.3, 1.2)), clip_score=float(rng.choice([2.0, 1e5])))
    else:
        meta["integrate"] = dict(kind="controlled")
        meta["sde"] = dict(base["sde"], generative=bool(rng.random() < 0.5))
        if mode == "controlled":
            meta["ctrl"], meta["net"] = base["ctrl"], base["net"]
        else:
            meta["wrap"] = bool(rng.random() < 0.5)
    sde, target, prior, ctrl = problems.build_integration(meta)
    params = {k: v.detach().clone() for k, v in ctrl.state_dict().items()} if ctrl is not None else {}
    tt = None
    if meta["target"]["kind"] == "gmm":
        tt = dict(loc=target.loc.clone(), scale=target.scale.clone(), mixture_weights=target.mixture_weights.clone())
    timesteps = eo.timesteps(0.0, end, steps=steps)
    n_out = int(rng.integers(1, 9))
    pts = np.sort(rng.uniform(0.0, end, size=n_out)).astype(np.float32)
    if rng.random() < 0.5:
        pts[0] = 0.0
    pts[-1] = end  # the reference indexes ts[ts_count] after the last output was emitted: the last time must be the grid's end
    if n_out > 2 and rng.random() < 0.3:
        pts[1] = pts[2]  # a repeated output time
    ts = torch.from_numpy(np.sort(pts))
    B = int(rng.choice([1, 33, 64, 100]))
    torch.manual_seed(case)
    x0 = torch.randn(B, d) * 1.5
    noise = torch.randn(steps, B, d)
    torch.set_num_threads(4)
    drift, diff = eo.integration_case(meta, params, tt)
    ref = eo.euler_integrate(drift, diff, ts, x0.clone(), timesteps, noise=noise).detach()
    for mod in (sde, target, prior, ctrl):
        if mod is not None:
            mod.to(DEV)
    xs = EulerIntegrator().integrate(sde, ts=ts.to(DEV), x_init=x0.to(DEV), timesteps=timesteps.to(DEV), noise=noise.to(DEV))
    tag = f"case {case}: {mode} {meta.get('sde', meta['integrate'])} / {meta['target']['kind']} d={d} B={B} steps={steps} ts={ts.tolist()}"
    assert xs.shape == ref.shape, tag
    if not torch.isfinite(ref).all():
        return  # a random configuration that blows up in the reference itself
    # conditioning probes as in the other sweeps: what +-1e-6 on the inputs does to the reference's own rows is not a difference
    cond_rows = torch.stack([torch.nan_to_num((eo.euler_integrate(drift, diff, ts, x0 * (1.0 + eps), timesteps, noise=noise * (1.0 + eps)).detach()
                                               - ref).abs().amax(dim=(0, 2)), nan=math.inf) for eps in _PERTS]).amax(dim=0)
    row_err = ((xs.cpu() - ref).abs().amax(dim=(0, 2)) - cond_rows).clamp_min(0.0)
    scale = max(1.0, float(ref.abs().max()))
    assert row_err.median().item() <= 1e-4 * scale, f"{tag}: median row error {row_err.median().item():.3e}"
    assert (row_err > 2e-3 * scale).float().mean().item() <= DRIFT_MAX, f"{tag}: max row error {row_err.max().item():.3e}"


@pytest.mark.parametrize("case", [0, 1, 2, 3])
def test_network_without_hidden_layers_matches_oracle(case):
    """FourierMLP(num_layers=2) (models/mlp.py:99-103: no hidden layer at all): the small-batch modes that split a group's network over two
    or four M waves rely on a hidden layer's exchange barrier, so such a network runs the one-M-wave form at every batch size."""
    def hook(spec, rng):
        spec["net"]["num_layers"] = 2
        spec["net"]["activation"] = ["gelu", "silu", "relu", "gelu"][case]
    check_eval_case(case, spec_hook=hook)

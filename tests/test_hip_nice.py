"""BASELINE configs[4]'s target: the NICE flow (reference distr/nice.py) and the loss loops on it.

CPU (`-m "not gpu"`): the host mirror (sde_sampler_amd/distr/nice.py) against the reference's known answers (tests/golden/nice_kat_*.npz,
produced by RUNNING the reference: tests/golden/make_golden_nice.py), its parameter names against the reference's state_dict keys, and
the engine's description of it (SdehNice) field by field.
GPU: csrc/sdeh_nice.hip through the C ABI (sdeh_nice_eval) against those known answers and against fp32 autograd on ragged shapes; the
step segments of the wide kernels (sdeh_simulate_fwd_steps) against the one-launch result; the Bridge (conf/solver/bridge.yaml) and PIS
on the flow against the reference's golden outputs: evaluation passes, the lv training loss and the gradients of both networks."""
import ctypes as C
import json
from pathlib import Path

import numpy as np
import pytest
import torch

from tests.helpers import GOLDEN_NICE, GOLDEN_NICE_KAT, GOLDEN_WIDE_BRIDGE, inference_params, load_fixture, measured

DEV = "cuda:0"
gpu = pytest.mark.gpu


def _flow(path):
    from sde_sampler_amd import problems

    fx = np.load(path)
    meta = json.loads(bytes(fx["meta"]).decode())
    tensors = {k[len("target/"):]: torch.from_numpy(fx[k].copy()) for k in fx.files if k.startswith("target/")} or None
    target = problems.build_target(meta["target"], tensors)
    return fx, meta, target


def _sha(model) -> str:
    import hashlib

    h = hashlib.sha256()
    for k, v in model.state_dict().items():
        h.update(k.encode())
        h.update(v.detach().cpu().numpy().tobytes())
    return h.hexdigest()


# ------------------------------------------------------------------------------------------------------------------------ CPU
@pytest.mark.parametrize("path", GOLDEN_NICE_KAT, ids=lambda p: Path(p).stem)
def test_host_mirror_reproduces_the_reference_flow(path):
    """Same module tree (state_dict keys), same seeded weights, same log-density and autograd score as the reference's Nice -- bit for bit
    on the CPU (one thread: the fixtures' reduction order)."""
    fx, meta, target = _flow(path)
    if _sha(target.model) != bytes(fx["weights_sha256"]).decode():
        assert not any(k.startswith("target/") for k in fx.files), "a fixture that carries its weights must load them exactly"
        pytest.skip("this host's torch draws other initial weights than the build container's")
    n = torch.get_num_threads()
    torch.set_num_threads(1)
    try:
        x = torch.from_numpy(fx["x"])
        assert np.array_equal(target.unnorm_log_prob(x).numpy(), fx["unnorm_log_prob"])
        assert np.array_equal(target.score(x.clone()).numpy(), fx["score"])
    finally:
        torch.set_num_threads(n)
    keys = set(target.model.state_dict())
    assert {"scaling.scale", "coupling.0.in_block.0.weight", "coupling.3.out_block.bias", "coupling.1.mid_block.0.0.weight"} <= keys
    z = target.model.f(x)[0]
    assert torch.allclose(target.model.g(z), x, atol=2e-4)  # g inverts f (nice.py:155-174)
    assert tuple(target.sample((5,)).shape) == (5, 196)


def test_engine_describes_the_flow_by_attribute():
    from sde_sampler_amd import _lib as L
    from sde_sampler_amd import engine as E
    from sde_sampler_amd import problems

    target = problems.build_target(dict(kind="nice", dim=196, coupling=3, mid_dim=40, hidden=4, mask_config=0.0, seed=3))
    keep = E._Keep()
    nd = E.describe_nice(target, torch.device("cpu"), keep)
    assert (nd.dim, nd.n_coupling, nd.mid_dim, nd.n_mid) == (196, 3, 40, 3)
    assert [nd.mask_config[i] for i in range(3)] == [0, 1, 0]  # NiceModel: (mask_config + i) % 2
    m = target.model
    assert nd.in_w[1] == m.coupling[1].in_block[0].weight.data_ptr() and nd.out_b[2] == m.coupling[2].out_block.bias.data_ptr()
    assert nd.mid_w[0][2] == m.coupling[0].mid_block[2][0].weight.data_ptr() and nd.scale == m.scaling.scale.data_ptr()
    assert E._external_target(target) and not E._known_distribution(target)
    dens = L.SdehDensity()
    E._fill_density(target, dens, keep, torch.device("cpu"), "target")
    assert dens.kind == L.DENS_EXTERNAL and any(isinstance(k, E._ExternalTarget) for k in keep)
    assert all(not p.requires_grad for p in target.model.parameters())  # nice.py:271-273
    with pytest.raises(ValueError, match="needs to be 196"):
        from sde_sampler_amd.distr.nice import Nice
        Nice(model=m, dim=100)


def test_cfg5_spec_is_configs4_as_written():
    from sde_sampler_amd import problems

    spec = problems.baseline_spec("cfg5_nice_bridge196")
    assert spec["target"] == dict(kind="nice", dim=196) and spec["net"]["channels"] == 256 and spec["grid"]["steps"] == 200
    assert spec["batch"] * 8 == 32768 and spec["loss"]["method"] == "lv" and spec["inference_ctrl"]["kind"] == "lerp_prior"


# ------------------------------------------------------------------------------------------------------------------------ GPU
#: measured on MI355X (gpurun_out/parity_measured.txt): log-density relative to max(1, |reference|), score absolute against |score| <= 2.6
NICE_LOGP_BAR, NICE_SCORE_BAR = 2e-6, 2e-5


@gpu
@pytest.mark.parametrize("path", GOLDEN_NICE_KAT, ids=lambda p: Path(p).stem)
def test_nice_eval_matches_the_reference_known_answers(path):
    fx, meta, target = _flow(path)
    if _sha(target.model) != bytes(fx["weights_sha256"]).decode():
        pytest.skip("this host's torch draws other initial weights than the build container's")
    target.to(DEV)
    x = torch.from_numpy(fx["x"]).to(DEV)
    lp = target.unnorm_log_prob(x).cpu().numpy()
    sc = target.score(x).cpu().numpy()
    e_lp = float(np.max(np.abs(lp - fx["unnorm_log_prob"]) / np.maximum(1.0, np.abs(fx["unnorm_log_prob"]))))
    e_sc = float(np.max(np.abs(sc - fx["score"])))
    measured(f"nice_eval/{Path(path).stem}/logp", e_lp, NICE_LOGP_BAR)
    measured(f"nice_eval/{Path(path).stem}/score", e_sc, NICE_SCORE_BAR)
    assert lp.shape == fx["unnorm_log_prob"].shape and e_lp <= NICE_LOGP_BAR, e_lp
    assert e_sc <= NICE_SCORE_BAR, e_sc


@gpu
@pytest.mark.parametrize("batch", [1, 63, 65, 1000])
@pytest.mark.parametrize("geometry", [(2, 36, 1, 1.0), (4, 52, 4, 0.0), (3, 500, 5, 1.0), (1, 64, 2, 0.0)])
def test_nice_eval_on_ragged_shapes_against_fp32_autograd(batch, geometry):
    """Batches that are not multiples of the 64-row tile, widths that are not multiples of the 128 / 16 tiles, one hidden layer (no mid
    block), either mask parity -- against torch's fp32 autograd of the mirror model on the device."""
    from sde_sampler_amd import problems

    coupling, mid, hidden, mask = geometry
    target = problems.build_target(dict(kind="nice", dim=196, coupling=coupling, mid_dim=mid, hidden=hidden, mask_config=mask,
                                        seed=coupling + mid, scale_std=0.2, out_gain=2.0)).to(DEV)
    torch.manual_seed(batch)
    x = torch.randn(batch, 196, device=DEV) * 1.5
    lp = target.unnorm_log_prob(x)
    sc = target.score(x)
    xr = x.clone().requires_grad_(True)
    ref_lp = target.model.log_prob(xr)
    (ref_sc,) = torch.autograd.grad(ref_lp.sum(), xr)
    assert tuple(lp.shape) == (batch, 1) and tuple(sc.shape) == (batch, 196)
    e_lp = float(((lp.squeeze(-1) - ref_lp.detach()).abs() / ref_lp.detach().abs().clamp_min(1.0)).max())
    # A ReLU unit whose pre-activation is within rounding of zero takes one side here and the other in the comparison pass (two fp32
    # evaluations in different summation orders): the row's score then differs by that unit's whole contribution (measured: 2e-3 on one
    # row of 1000 at 7.5 M units).  Rows are judged individually; such rows must be rare and their deviation that of single units.
    row = (sc - ref_sc).abs().amax(dim=1)
    e_sc, flipped = float(row.median()), float((row > 5e-5).float().mean())
    measured(f"nice_eval_ragged/B{batch}/c{coupling}m{mid}h{hidden}", max(e_lp, e_sc), 5e-5)
    measured(f"nice_eval_ragged_kink_rows/B{batch}/c{coupling}m{mid}h{hidden}", flipped, 5e-3)
    assert e_lp <= 5e-6 and e_sc <= 2e-5, (e_lp, e_sc)
    assert flipped <= max(5e-3, 1.5 / batch) and float(row.max()) <= 2e-2, (flipped, float(row.max()))
    assert torch.equal(target.score(x), sc)  # deterministic


@gpu
def test_nice_eval_rejects_what_it_does_not_cover():
    from sde_sampler_amd import SdehUnsupported, problems

    target = problems.build_target(dict(kind="nice", dim=196, coupling=2, mid_dim=38, hidden=2)).to(DEV)  # mid_dim % 4 != 0
    with pytest.raises(SdehUnsupported, match="multiple"):
        target.score(torch.zeros(4, 196, device=DEV))


def _segment_run(prob, x0, noise, step, ext=None):
    """The problem's evaluation through sdeh_simulate_fwd_steps in segments of `step` steps; ext: callable x -> target score (then the
    problem's target is declared SDEH_DENS_EXTERNAL and the terminal log-density is subtracted here)."""
    from sde_sampler_amd import _lib as L
    from sde_sampler_amd import engine as E

    lo, eng = prob.loss, prob.loss.engine
    keep = E._Keep()
    flags = L.FLAG_ITO | L.FLAG_INIT_LOGP | L.FLAG_TERMINAL_TARGET
    pr = eng.build_problem(device=x0.device, keep=keep, loss_kind=L.LOSS_TIME_REVERSAL, generative_ctrl=lo.generative_ctrl, sde=lo.sde,
                           flags=flags, terminal_target=prob.target, clip_target=None, second=prob.prior, inference_ctrl=lo.inference_ctrl)
    if ext is not None:
        pr.target.kind = L.DENS_EXTERNAL
    B, d = x0.shape
    T = prob.ts.numel() - 1
    n_hidden = max(pr.base_model.n_hidden, pr.inference.base_model.n_hidden)
    plan = eng._plan(x0.device, d, pr.base_model.channels, n_hidden, T, 0)
    plan.reserve(B)
    lib = L.load()
    bufs = [torch.empty_like(x0), torch.empty_like(x0)]
    rnd = torch.full((B,), float("nan"), device=x0.device)
    xs = torch.zeros((T + 1, B, d), device=x0.device)
    cur = x0.contiguous()
    stream = torch.cuda.current_stream().cuda_stream
    for k, i in enumerate(range(0, T, step)):
        j = min(T, i + step)
        sc = None
        if ext is not None:
            assert step == 1
            sc = ext(cur).contiguous()
        out = bufs[k & 1]
        L.check(lib.sdeh_simulate_fwd_steps(plan.handle, C.byref(pr), prob.ts.data_ptr(), T, i, j, cur.data_ptr(), B, noise.data_ptr(), 5, 0, 0,
                                            out.data_ptr(), rnd.data_ptr(), xs.data_ptr(), None, None if sc is None else sc.data_ptr(), 0,
                                            stream))
        cur = out
    if ext is not None:
        rnd = rnd - prob.target.unnorm_log_prob(cur).squeeze(-1)
    torch.cuda.synchronize()
    return cur.clone(), rnd, xs


@gpu
@pytest.mark.parametrize("name", ["widebridge_funnel196_c256", "widebridge_gauss33_clipped_c128", "widebridge_mw44_c128"])
def test_step_segments_equal_the_one_launch_result(name):
    """sdeh_simulate_fwd_steps: the grid in segments of 1 / 3 / all steps -- tables prepared once, Philox counters and trajectory rows of
    the whole grid, rnd carried from segment to segment -- against sdeh_simulate_fwd_aux2's single launch: the states bit for bit, rnd to
    the order in which the network divergence's partial sums join it."""
    from sde_sampler_amd import problems

    path = [p for p in GOLDEN_WIDE_BRIDGE if name in p][0]
    fx, meta, params, tt = load_fixture(path)
    prob = problems.build(meta, params, tt, device=DEV, params_inf=inference_params(fx))
    x0, noise = torch.from_numpy(fx["x0"]).to(DEV), torch.from_numpy(fx["noise"]).to(DEV)
    with torch.no_grad():
        x_ref, rnd_ref, xs_ref = prob.loss.simulate(prob.ts, x0, prob.target.unnorm_log_prob, prob.second_log_prob, train=False,
                                                    compute_ito_int=True, return_traj=True, noise=noise)
    T = prob.ts.numel() - 1
    for step in (1, 3, T):
        x_T, rnd, xs = _segment_run(prob, x0, noise, step)
        assert torch.equal(x_T, x_ref), step
        assert torch.equal(xs, xs_ref), step
        err = float(((rnd - rnd_ref.squeeze(-1)).abs() / rnd_ref.squeeze(-1).abs().clamp_min(1.0)).max())
        measured(f"segments/{name}/step{step}", err, 1e-5)
        assert err <= (0.0 if step == T else 1e-5), (step, err)  # (measured: <= 3.2e-6)


@gpu
def test_a_supplied_score_stands_in_for_the_built_in_one():
    """SDEH_DENS_EXTERNAL: the built-in Gaussian target's score, computed by the CALLER per step, gives the built-in target's trajectories
    (the kernel multiplies by a precomputed 1 / sigma^2 where torch divides: equal to rounding)."""
    from sde_sampler_amd import problems

    path = [p for p in GOLDEN_WIDE_BRIDGE if "gauss33_clipped" in p][0]
    fx, meta, params, tt = load_fixture(path)
    prob = problems.build(meta, params, tt, device=DEV, params_inf=inference_params(fx))
    x0, noise = torch.from_numpy(fx["x0"]).to(DEV), torch.from_numpy(fx["noise"]).to(DEV)
    x_ref, rnd_ref, _ = _segment_run(prob, x0, noise, 1)
    x_T, rnd, _ = _segment_run(prob, x0, noise, 1, ext=lambda x: prob.target.score(x))
    assert float((x_T - x_ref).abs().max()) <= 2e-5
    assert float(((rnd - rnd_ref).abs() / rnd_ref.abs().clamp_min(1.0)).max()) <= 2e-5


def _rows(name, got, ref, max_tol=1e-2, med_tol=1e-4):
    got, ref = np.asarray(got, np.float64), np.asarray(ref, np.float64)
    err = np.abs(got - ref) / np.maximum(1.0, np.abs(ref))
    assert err.max() <= max_tol, f"{name}: max err {err.max():.3e}"
    assert np.median(err) <= med_tol, f"{name}: median err {np.median(err):.3e}"
    return float(err.max())


def _problem(path):
    from sde_sampler_amd import problems

    fx, meta, params, tt = load_fixture(path)
    prob = problems.build(meta, params, tt, device=DEV, params_inf=inference_params(fx) or None)
    return fx, meta, prob


@gpu
@pytest.mark.parametrize("path", GOLDEN_NICE, ids=lambda p: Path(p).stem)
def test_loss_loops_on_the_flow_match_the_reference(path):
    """Both evaluation passes of the reference (losses/oc.py:258-278 / 369-391) on identical noise: SURVEY 8d's bars."""
    fx, meta, prob = _problem(path)
    x0, noise = torch.from_numpy(fx["x0"]).to(DEV), torch.from_numpy(fx["noise"]).to(DEV)
    r1 = prob.eval(x0, compute_weights=True, return_traj=True, noise=noise)
    kernel = prob.loss.engine.last_kernel_name()
    assert kernel.startswith("bridge_wide<C=128" if meta.get("inference_ctrl") else "traj_wide<C=128"), kernel
    r2 = prob.eval(x0, compute_weights=False, return_traj=False, noise=noise)
    with torch.no_grad():
        kw = dict(compute_ito_int=True, return_traj=False, noise=noise)
        if meta["loss"]["kind"] == "time_reversal":
            kw["train"] = False
        _, rnd1, _ = prob.loss.simulate(prob.ts, x0, prob.target.unnorm_log_prob, prob.second_log_prob, **kw)
    e_x = _rows("x_T", r1.samples.cpu().numpy(), fx["eval1/x_T"])
    e_r = _rows("rnd (ito)", rnd1.cpu().numpy(), fx["eval1/rnd"])
    measured(f"nice_loops/{Path(path).stem}/x_T", e_x, 1e-2)
    measured(f"nice_loops/{Path(path).stem}/rnd", e_r, 1e-2)
    est = lambda want: max(1e-4, 1e-5 * abs(want))
    for key in ("log_norm_const_lb_ito", "log_norm_const_is"):
        want = float(fx["eval1/" + key])
        measured(f"nice_loops/{Path(path).stem}/{key}", abs(r1.log_norm_const_preds[key] - want), est(want))
        assert abs(r1.log_norm_const_preds[key] - want) <= est(want), (key, r1.log_norm_const_preds[key], want)
    want = float(fx["eval2/log_norm_const_lb"])
    assert abs(r2.log_norm_const_preds["log_norm_const_lb"] - want) <= est(want)
    lv = float(fx["eval1/lv_loss"])
    assert abs(r1.metrics["eval/lv_loss"] - lv) <= 1e-4 * max(1.0, abs(lv)), (r1.metrics["eval/lv_loss"], lv)
    assert torch.equal(r1.samples, r2.samples)
    xs = r1.xs.cpu().numpy()
    assert np.array_equal(xs[0], fx["x0"]) and np.array_equal(xs[-1], r1.samples.cpu().numpy())


NICE_LOSS_BAR, NICE_GRAD_BAR = 3e-5, {"lv": 1.3e-4, "kl": 3e-4}  # the wide kernels' bars (tests/test_hip_wide_train.py: plain / Bridge kl)


@gpu
@pytest.mark.parametrize("method", ["lv", "kl"])
@pytest.mark.parametrize("path", GOLDEN_NICE, ids=lambda p: Path(p).stem)
def test_training_on_the_flow_matches_the_reference_autograd(path, method):
    """conf/solver/bridge.yaml's loss (time_reversal_lv), basic_bridge.yaml's (kl) and pis.yaml's (kl) on the flow: loss value and the parameter
    gradients of every network against the reference's autograd.  The flow's score is a CONSTANT of that graph (distr/base.py:130-137:
    `create_graph = False`, or x detached: reparam.py:56-66, 185-197), also under back-propagation through time -- which is what the score
    plane handed to the backward kernels is; the terminal cost's derivative 1[|log rho| <= clip] score(x_T) arrives as a second plane."""
    from tests.test_hip_wide_train import _check_grads

    fx, meta, prob = _problem(path)
    prob.loss.method = method
    x0, noise = torch.from_numpy(fx["x0"]).to(DEV), torch.from_numpy(fx["noise"]).to(DEV)
    val, info = prob.loss(prob.ts, x0, prob.target.unnorm_log_prob, prob.second_log_prob, noise=noise)
    ref = float(fx[f"train_{method}/loss"])
    measured(f"nice_train_loss/{Path(path).stem}/{method}", abs(val.item() - ref) / max(1.0, abs(ref)), NICE_LOSS_BAR)
    assert abs(val.item() - ref) <= NICE_LOSS_BAR * max(1.0, abs(ref)), (val.item(), ref)
    assert int(info["train/n_filtered_cumulative"]) == int(fx[f"train_{method}/n_filtered"])
    val.backward()
    worst = _check_grads(fx, method, "grad", prob.ctrl, tol=NICE_GRAD_BAR[method])
    inf = getattr(prob.loss, "inference_ctrl", None)
    if inf is not None:
        worst = max(worst, _check_grads(fx, method, "grad_inf", inf, tol=NICE_GRAD_BAR[method]))
    measured(f"nice_train_grad/{Path(path).stem}/{method}", worst[0], NICE_GRAD_BAR[method])


@gpu
@pytest.mark.parametrize("name,method", [("dis_lerp", "kl"), ("dis_lerp", "lv"), ("dds_score", "kl")])
def test_training_other_solvers_on_the_flow_matches_the_oracle_autograd(name, method):
    """conf/solver/dis.yaml (LerpCtrl: the backward's score plane is torch.lerp(prior score, flow score, t / T) on the stored trajectory; the
    prior's part carries its Jacobian through time) and dds.yaml on the flow: gradients against the oracle's autograd on identical noise."""
    from oracle import em_oracle as eo
    from sde_sampler_amd import problems

    batch = 40
    spec = dict(OTHER_SOLVERS[name], batch=batch, net=dict(channels=128, num_layers=4, activation="gelu"),
                target=dict(kind="nice", dim=196, coupling=3, mid_dim=44, hidden=3, mask_config=1.0, seed=17, scale_std=0.15, out_gain=3.0))
    spec["loss"] = dict(spec["loss"], method=method, max_rnd=1e8 if method == "lv" else None)
    if name == "dds_score":  # (inactive clips: a clamp on its edge is a coin toss between two fp32 evaluations)
        spec["ctrl"] = dict(spec["ctrl"], clip_model=1e4, clip_score=1e4)
    prob = problems.build(spec)
    with torch.no_grad():
        for mod in (prob.ctrl.base_model.out_layer, prob.ctrl.score_model.out_layer):
            mod.weight.normal_(0.0, 0.05)
    params = {k: v.detach().clone().requires_grad_(v.is_floating_point() and not k.endswith("timestep_coeff")) for k, v in prob.ctrl.state_dict().items()}
    tt = {k: v.detach().clone() for k, v in prob.target.model.state_dict().items()}
    torch.manual_seed(29)
    x0 = prob.prior.sample((batch,))
    T = prob.ts.numel() - 1
    noise = torch.randn(T, batch, 196)
    n = torch.get_num_threads()
    torch.set_num_threads(1)
    try:
        ref_loss, _, _, _ = eo.Problem(spec, params, tt).train_loss(prob.ts.clone(), x0.clone(), noise, method=method)
        ref_loss.backward()
    finally:
        torch.set_num_threads(n)
    prob.to(DEV)
    val, _ = prob.loss(prob.ts, x0.to(DEV), prob.target.unnorm_log_prob, prob.second_log_prob, noise=noise.to(DEV))
    assert abs(val.item() - ref_loss.item()) <= NICE_LOSS_BAR * max(1.0, abs(ref_loss.item())), (val.item(), ref_loss.item())
    val.backward()
    worst = 0.0
    gmax = max(float(v.grad.abs().max()) for v in params.values() if v.grad is not None)
    for k, p in prob.ctrl.named_parameters():
        want = params[k].grad
        if want is None:
            continue
        got = p.grad.detach().cpu() if p.grad is not None else torch.zeros_like(want)
        scale = max(float(want.abs().max()), 1e-3 * gmax, 1e-12)
        worst = max(worst, float((got - want).abs().max()) / scale)
    measured(f"nice_train_other/{name}/{method}", worst, NICE_GRAD_BAR["kl"])
    assert worst <= NICE_GRAD_BAR["kl"], worst


@gpu
def test_a_control_without_a_score_term_meets_the_flow_in_one_segment():
    """ClippedCtrl on the flow: no score per step -- the grid runs as ONE segment, the flow enters through the terminal cost alone; against the
    oracle, evaluation and kl training."""
    from oracle import em_oracle as eo
    from sde_sampler_amd import problems

    spec = dict(batch=33, target=dict(kind="nice", dim=196, coupling=2, mid_dim=40, hidden=2, seed=4, scale_std=0.1, out_gain=2.0),
                prior=dict(kind="iso_gauss", dim=196, loc=0.0, scale=1.0), sde=dict(kind="vp", beta_min=0.1, beta_max=6.0, scale=1.0, terminal_t=1.0),
                ctrl=dict(kind="clipped", clip_model=1e4), net=dict(channels=128, num_layers=4, activation="gelu"),
                loss=dict(kind="time_reversal", method="kl", max_rnd=None), grid=dict(start=0.0, end=1.0, steps=5, rescale_t=None))
    prob = problems.build(spec)
    with torch.no_grad():
        prob.ctrl.base_model.out_layer.weight.normal_(0.0, 0.05)
    params = {k: v.detach().clone().requires_grad_(v.is_floating_point() and not k.endswith("timestep_coeff")) for k, v in prob.ctrl.state_dict().items()}
    tt = {k: v.detach().clone() for k, v in prob.target.model.state_dict().items()}
    torch.manual_seed(31)
    x0, noise = prob.prior.sample((33,)), torch.randn(5, 33, 196)
    oracle = eo.Problem(spec, params, tt)
    ref = oracle.eval(prob.ts.clone(), x0.clone(), noise, compute_weights=True)
    ref_loss, _, _, _ = oracle.train_loss(prob.ts.clone(), x0.clone(), noise, method="kl")
    ref_loss.backward()
    prob.to(DEV)
    got = prob.eval(x0.to(DEV), compute_weights=True, noise=noise.to(DEV))
    assert prob.loss.engine.last_kernel_name().startswith("traj_wide<C=128")
    assert abs(got.log_norm_const_preds["log_norm_const_lb_ito"] - ref["log_norm_const_lb_ito"]) <= max(1e-4, 1e-5 * abs(ref["log_norm_const_lb_ito"]))
    val, _ = prob.loss(prob.ts, x0.to(DEV), prob.target.unnorm_log_prob, prob.second_log_prob, noise=noise.to(DEV))
    assert abs(val.item() - ref_loss.item()) <= NICE_LOSS_BAR * max(1.0, abs(ref_loss.item()))
    val.backward()
    for k, p in prob.ctrl.named_parameters():
        want = params[k].grad
        if want is None:
            continue
        err = float((p.grad.detach().cpu() - want).abs().max()) / max(float(want.abs().max()), 1e-9)
        assert err <= 3e-4, (k, err)


@gpu
def test_configs4_as_written_runs_at_reduced_size():
    """BASELINE configs[4]: target = nice, solver = bridge, channels = 256 -- the spec as written at a small batch / few steps: finite,
    deterministic in-kernel noise, sharded = unsharded."""
    from sde_sampler_amd import problems

    spec = problems.baseline_spec("cfg5_nice_bridge196")
    spec["batch"], spec["grid"]["steps"] = 96, 5
    prob = problems.build(spec, device=DEV)
    with torch.no_grad():
        for p in list(prob.ctrl.base_model.out_layer.parameters()) + list(prob.loss.inference_ctrl.base_model.out_layer.parameters()):
            p.add_(0.05 * torch.randn_like(p))
    torch.manual_seed(3)
    x0 = prob.prior.sample((96,))
    eng = prob.loss.engine
    eng.calls, prob.loss.row_offset = 4, 0
    a = prob.eval(x0, compute_weights=True)
    assert a.samples.shape == (96, 196) and torch.isfinite(a.samples).all() and np.isfinite(a.log_norm_const_preds["log_norm_const_is"])
    assert eng.last_kernel_name() == "bridge_wide<C=256,split=8>"
    eng.calls = 4
    b = prob.eval(x0, compute_weights=True)
    assert torch.equal(a.samples, b.samples) and torch.equal(a.weights, b.weights)
    eng.calls, prob.loss.row_offset = 4, 0
    lo = prob.eval(x0[:32], compute_weights=False)
    eng.calls, prob.loss.row_offset = 4, 32
    hi = prob.eval(x0[32:], compute_weights=False)
    assert torch.equal(torch.cat([lo.samples, hi.samples]), a.samples)


OTHER_SOLVERS = {
    # conf/solver/dis.yaml's shape: LerpCtrl (prior AND target score: the supplied score meets the built-in prior score in torch.lerp's form), VP
    "dis_lerp": dict(prior=dict(kind="iso_gauss", dim=196, loc=0.0, scale=1.0), sde=dict(kind="vp", beta_min=0.1, beta_max=6.0, scale=1.0, terminal_t=1.0),
                     ctrl=dict(kind="lerp", clip_model=1e4, clip_score=1e4, scale_score=1.0, gamma_dim=1, gamma_bias=1.0),
                     loss=dict(kind="time_reversal", method="kl", max_rnd=None), grid=dict(start=0.0, end=1.0, steps=7, rescale_t=None)),
    # conf/solver/dds.yaml's shape: exponential integrator, ScoreCtrl with per-coordinate gamma and active clips, cosine grid
    "dds_score": dict(prior=dict(kind="iso_gauss", dim=196, loc=0.0, scale=1.0, truncate_quartile=1e-4), sde=None,
                      ctrl=dict(kind="score", clip_model=0.5, clip_score=1.0, scale_score=0.7, gamma_dim=196, gamma_bias=0.3),
                      loss=dict(kind="exponential", method="lv", max_rnd=1e8, alpha=1.0, sigma=1.0),
                      grid=dict(start=0.0, end=6.4, steps=9, rescale_t="cosine")),
}


@gpu
@pytest.mark.parametrize("name", sorted(OTHER_SOLVERS))
@pytest.mark.parametrize("channels,batch", [(128, 48), (64, 70)])
def test_other_solvers_on_the_flow_match_the_oracle(name, channels, batch):
    """DIS and DDS on the flow (the plain wide kernel in one-step segments; 64 channels at d = 196 run on it as well) against the CPU oracle
    -- which is bit-exact on the reference-generated NICE fixtures (tests/test_oracle_golden.py) -- on identical noise; ragged batches."""
    from oracle import em_oracle as eo
    from sde_sampler_amd import problems

    spec = dict(OTHER_SOLVERS[name], batch=batch, net=dict(channels=channels, num_layers=4, activation="gelu"),
                target=dict(kind="nice", dim=196, coupling=3, mid_dim=44, hidden=3, mask_config=1.0, seed=17, scale_std=0.15, out_gain=3.0))
    prob = problems.build(spec)
    with torch.no_grad():
        for mod in (prob.ctrl.base_model.out_layer, prob.ctrl.score_model.out_layer):
            mod.weight.normal_(0.0, 0.05)
    params = {k: v.detach().clone() for k, v in prob.ctrl.state_dict().items()}
    tt = {k: v.detach().clone() for k, v in prob.target.model.state_dict().items()}
    torch.manual_seed(23)
    x0 = prob.prior.sample((batch,))
    T = prob.ts.numel() - 1
    noise = torch.randn(T, batch, 196)
    n = torch.get_num_threads()
    torch.set_num_threads(1)
    try:
        ref = eo.Problem(spec, params, tt).eval(prob.ts.clone(), x0.clone(), noise, compute_weights=True)
    finally:
        torch.set_num_threads(n)
    prob.to(DEV)
    got = prob.eval(x0.to(DEV), compute_weights=True, return_traj=False, noise=noise.to(DEV))
    assert prob.loss.engine.last_kernel_name().startswith(f"traj_wide<C={channels}")
    e_x = _rows("x_T", got.samples.cpu().numpy(), ref["samples"].numpy())
    measured(f"nice_other_solvers/{name}/C{channels}/x_T", e_x, 1e-2)
    for key in ("log_norm_const_lb_ito", "log_norm_const_is"):
        want = ref[key]
        tol = max(1e-4, 1e-5 * abs(want))
        measured(f"nice_other_solvers/{name}/C{channels}/{key}", abs(got.log_norm_const_preds[key] - want), tol)
        assert abs(got.log_norm_const_preds[key] - want) <= tol, (key, got.log_norm_const_preds[key], want)


@gpu
def test_stepped_evaluation_is_stable_under_repetition():
    """200 back-to-back evaluations of the stepped path (5 segments + 5 score evaluations each, work memory reused from the engine's cache):
    bitwise repeatable for a fixed (seed, call), no drift of the allocator-backed planes."""
    from sde_sampler_amd import problems

    spec = problems.baseline_spec("cfg5_nice_bridge196")
    spec["batch"], spec["grid"]["steps"] = 160, 5
    spec["net"]["channels"] = 128
    spec["target"] = dict(kind="nice", dim=196, coupling=2, mid_dim=64, hidden=3)
    prob = problems.build(spec, device=DEV)
    torch.manual_seed(9)
    x0 = prob.prior.sample((160,))
    eng = prob.loss.engine
    eng.calls = 11
    first = prob.eval(x0, compute_weights=True)
    for _ in range(200):
        eng.calls = 11
        again = prob.eval(x0, compute_weights=True)
    assert torch.equal(first.samples, again.samples) and torch.equal(first.weights, again.weights)
    assert len(eng._nice_work) <= 2


def _random_flow_problem(case: int):
    """A random (solver, control, network, flow geometry, batch) combination on the flow; the geometry spans one to four couplings, widths that
    fill one, part of two and more than two output tiles, k-ranges that are and are not multiples of the k-tile, one to four hidden layers."""
    import random

    rng = random.Random(1000 + case)
    d = 196
    flow = dict(kind="nice", dim=d, coupling=rng.randint(1, 4), mid_dim=4 * rng.randint(2, 66), hidden=rng.randint(1, 4),
                mask_config=float(rng.randint(0, 1)), seed=rng.randint(1, 99), scale_std=rng.choice([0.05, 0.2]), out_gain=rng.choice([1.0, 3.0]))
    channels = rng.choice([64, 128, 256])
    solver = rng.choice(["bridge", "pis", "dis", "dds"] if channels >= 128 else ["pis", "dis", "dds"])
    clips = dict(clip_model=rng.choice([1e4, 0.3]), clip_score=rng.choice([1e4, 0.8]), scale_score=rng.choice([1.0, 0.6]))
    gamma = dict(gamma_dim=rng.choice([1, d]), gamma_bias=rng.choice([1.0, 0.2]))
    iso = dict(kind="iso_gauss", dim=d, loc=0.0, scale=1.0)
    steps = rng.randint(2, 6)
    spec = dict(batch=rng.choice([1, 31, 33, 64, 90]), target=flow, net=dict(channels=channels, num_layers=rng.choice([3, 4]), activation=rng.choice(["gelu", "silu", "relu"])))
    if solver == "bridge":
        spec.update(prior=iso, sde=dict(kind="scaled_bm", diff_coeff=1.0, terminal_t=1.0), ctrl=dict(kind="lerp_target", **clips, **gamma),
                    inference_ctrl=dict(kind=rng.choice(["lerp_prior", "clipped"]), clip_model=10.0, clip_score=10.0, scale_score=1.0, gamma_dim=1, gamma_bias=1.0),
                    loss=dict(kind="time_reversal", method="lv", max_rnd=1e8), grid=dict(start=0.0, end=1.0, steps=steps, rescale_t=None))
    elif solver == "pis":
        spec.update(prior=dict(kind="delta", dim=d), sde=dict(kind="scaled_bm", diff_coeff=0.45, terminal_t=5.0), ctrl=dict(kind="score", **clips, **gamma),
                    loss=dict(kind="reference_sde", method="lv", max_rnd=1e8), grid=dict(start=0.0, end=5.0, steps=steps, rescale_t=None))
    elif solver == "dis":
        spec.update(prior=iso, sde=dict(kind="vp", beta_min=0.1, beta_max=6.0, scale=1.0, terminal_t=1.0),
                    ctrl=dict(kind=rng.choice(["lerp", "lerp_target", "score"]), **clips, **gamma),
                    loss=dict(kind="time_reversal", method="kl", max_rnd=None), grid=dict(start=0.0, end=1.0, steps=steps, rescale_t=None))
    else:
        spec.update(prior=dict(iso, truncate_quartile=1e-4), sde=None, ctrl=dict(kind="score", **clips, **gamma),
                    loss=dict(kind="exponential", method="lv", max_rnd=1e8, alpha=1.0, sigma=1.0),
                    grid=dict(start=0.0, end=6.4, steps=steps + 2, rescale_t="cosine"))
    return solver, spec


@gpu
@pytest.mark.parametrize("case", range(16))
def test_random_problems_on_the_flow_match_the_oracle(case):
    from oracle import em_oracle as eo
    from sde_sampler_amd import problems

    solver, spec = _random_flow_problem(case)
    prob = problems.build(spec)
    inf = getattr(prob.loss, "inference_ctrl", None)
    with torch.no_grad():
        for ctrl in (prob.ctrl, inf):
            if ctrl is None:
                continue
            ctrl.base_model.out_layer.weight.normal_(0.0, 0.05)
            if getattr(ctrl, "score_model", None) is not None:
                ctrl.score_model.out_layer.weight.normal_(0.0, 0.05)
    params = {k: v.detach().clone() for k, v in prob.ctrl.state_dict().items()}
    params_inf = {k: v.detach().clone() for k, v in inf.state_dict().items()} if inf is not None else None
    tt = {k: v.detach().clone() for k, v in prob.target.model.state_dict().items()}
    B, T = spec["batch"], prob.ts.numel() - 1
    torch.manual_seed(case)
    x0 = prob.prior.sample((B,))
    noise = torch.randn(T, B, 196)
    n = torch.get_num_threads()
    torch.set_num_threads(1)
    try:
        ref = eo.Problem(spec, params, tt, params_inf=params_inf).eval(prob.ts.clone(), x0.clone(), noise, compute_weights=True)
    finally:
        torch.set_num_threads(n)
    prob.to(DEV)
    got = prob.eval(x0.to(DEV), compute_weights=True, return_traj=False, noise=noise.to(DEV))
    tag = f"{solver}/C{spec['net']['channels']}/c{spec['target']['coupling']}m{spec['target']['mid_dim']}h{spec['target']['hidden']}/B{B}"
    # (ReLU networks / flows put single units on their kink: rows are judged with the wide kernels' row criterion, the estimators with SURVEY 8d's bar)
    e_x = _rows("x_T " + tag, got.samples.cpu().numpy(), ref["samples"].numpy())
    measured(f"nice_fuzz/{case}/{tag}/x_T", e_x, 1e-2)
    want = ref["log_norm_const_lb_ito"]
    tol = max(2e-4, 2e-5 * abs(want)) * (4.0 if B < 8 else 1.0)
    measured(f"nice_fuzz/{case}/{tag}/lb_ito", abs(got.log_norm_const_preds["log_norm_const_lb_ito"] - want), tol)
    assert abs(got.log_norm_const_preds["log_norm_const_lb_ito"] - want) <= tol, (tag, got.log_norm_const_preds["log_norm_const_lb_ito"], want)

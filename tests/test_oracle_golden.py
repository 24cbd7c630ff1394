"""Pins the CPU oracle (oracle/em_oracle.py) against the golden vectors produced by RUNNING the reference
(tests/golden/make_golden.py).  The oracle follows the reference's op order, so equality is bit-exact."""
import glob
import json
from pathlib import Path

import numpy as np
import pytest
import torch

from oracle import em_oracle as eo

_ALL = sorted(glob.glob(str(Path(__file__).parent / "golden" / "*.npz")))
GOLDEN = [p for p in _ALL if not Path(p).name.startswith(("int_", "metrics_", "bridge_", "wide", "fullsize_", "nice"))]
GOLDEN_WIDE = [p for p in _ALL if Path(p).name.startswith(("wide_", "widebridge_"))]
GOLDEN_BRIDGE = [p for p in _ALL if Path(p).name.startswith("bridge_")]
GOLDEN_METRICS = [p for p in _ALL if Path(p).name.startswith("metrics_")]
GOLDEN_INT = [p for p in _ALL if Path(p).name.startswith("int_")]
assert GOLDEN, "golden fixtures missing"


def load(path):
    fx = np.load(path)
    prob, params = eo.problem_from_fixture(fx)
    ts = torch.from_numpy(fx["ts"])
    x0 = torch.from_numpy(fx["x0"])
    noise = torch.from_numpy(fx["noise"])
    return fx, prob, params, ts, x0, noise


@pytest.fixture(autouse=True)
def _one_thread():
    n = torch.get_num_threads()
    torch.set_num_threads(1)  # fixtures were generated single-threaded (reduction order)
    yield
    torch.set_num_threads(n)


@pytest.mark.parametrize("path", GOLDEN, ids=lambda p: Path(p).stem)
def test_time_grid(path):
    fx, prob, *_ = load(path)
    assert torch.equal(prob.grid(), torch.from_numpy(fx["ts"]))


@pytest.mark.parametrize("path", GOLDEN, ids=lambda p: Path(p).stem)
def test_eval_passes_bit_exact(path):
    fx, prob, params, ts, x0, noise = load(path)
    r1 = prob.eval(ts, x0, noise, compute_weights=True, return_traj="eval1/xs" in fx.files)
    assert np.array_equal(r1["samples"].numpy(), fx["eval1/x_T"])
    assert np.array_equal(r1["rnd"].numpy(), fx["eval1/rnd"])
    assert np.array_equal(r1["weights"].numpy(), fx["eval1/weights"])
    assert r1["log_norm_const_lb_ito"] == float(fx["eval1/log_norm_const_lb_ito"])
    assert r1["log_norm_const_is"] == float(fx["eval1/log_norm_const_is"])
    assert r1["lv_loss"] == float(fx["eval1/lv_loss"])
    if "eval1/xs" in fx.files:
        assert np.array_equal(r1["xs"].numpy(), fx["eval1/xs"])
    r2 = prob.eval(ts, x0, noise, compute_weights=False)
    assert np.array_equal(r2["samples"].numpy(), fx["eval2/x_T"])
    assert np.array_equal(r2["rnd"].numpy(), fx["eval2/rnd"])
    assert r2["log_norm_const_lb"] == float(fx["eval2/log_norm_const_lb"])


@pytest.mark.parametrize("method", ["kl", "lv"])
@pytest.mark.parametrize("path", GOLDEN, ids=lambda p: Path(p).stem)
def test_train_loss_and_grads_bit_exact(path, method):
    fx, prob, params, ts, x0, noise = load(path)
    names = [k for k in params if not k.endswith("timestep_coeff")]
    for k in names:
        params[k].requires_grad_(True)
    loss, n_filtered, _, _ = prob.train_loss(ts, x0, noise, method=method)
    loss.backward()
    assert loss.item() == float(fx[f"train_{method}/loss"])
    assert n_filtered == int(fx[f"train_{method}/n_filtered"])
    for k in names:
        ref = fx[f"train_{method}/grad/{k}"]
        got = params[k].grad.numpy() if params[k].grad is not None else np.zeros_like(ref)
        assert np.array_equal(got, ref), k


@pytest.mark.parametrize("path", GOLDEN, ids=lambda p: Path(p).stem)
def test_density_known_answers(path):
    """Reference KAT (tests/distr_eval.py:45-55): analytic score == autograd score, rtol=atol=1e-4; plus
    bit-exact log-density / score vectors captured from the reference."""
    fx, prob, *_ = load(path)
    x = torch.from_numpy(fx["kat/x"])
    assert np.array_equal(prob.target.unnorm_log_prob(x).numpy(), fx["kat/target_unnorm_log_prob"])
    assert np.array_equal(prob.target.score(x.clone()).numpy(), fx["kat/target_score"])
    assert np.array_equal(prob.second.log_prob(x).numpy(), fx["kat/second_log_prob"])
    if "kat/prior_score" in fx.files:
        assert np.array_equal(prob.prior.score(x).numpy(), fx["kat/prior_score"])
    torch.testing.assert_close(prob.target.score(x.clone()), prob.target.autograd_score(x.clone()), rtol=1e-4, atol=1e-4)
    if "kat/prior_score" in fx.files:
        torch.testing.assert_close(prob.prior.score(x.clone()), prob.prior.autograd_score(x.clone()), rtol=1e-4, atol=1e-4)


def test_noise_free_path_draws_from_global_rng():
    fx, prob, params, ts, x0, noise = load(GOLDEN[0])
    meta = json.loads(bytes(fx["meta"]).decode())
    torch.manual_seed(meta["seed"])
    _ = torch.randn_like(x0) if meta["prior"]["kind"] != "delta" else None  # consume the prior draw
    a = prob.eval(ts, x0, None, compute_weights=False)
    b = prob.eval(ts, x0, noise, compute_weights=False)
    # (x0 was drawn with prior.sample == loc + scale*randn for the untruncated prior, so the streams line up)
    assert np.array_equal(a["samples"].numpy(), b["samples"].numpy())


EXTRA = [p for p in GOLDEN if "train_kl_ito/loss" in np.load(p).files]


@pytest.mark.parametrize("path", EXTRA, ids=lambda p: Path(p).stem)
def test_remaining_methods_bit_exact(path):
    """kl_ito and lv_traj (two trajectories per sample) losses and gradients of the oracle == the reference's."""
    fx, prob, params, ts, x0, noise = load(path)
    names = [k for k in params if not k.endswith("timestep_coeff")]
    for method, nz, tps in (("kl_ito", noise, 1), ("lv_traj", torch.from_numpy(fx["noise_traj2"]), 2)):
        for k in names:
            params[k].grad = None
            params[k].requires_grad_(True)
        loss, _, _, _ = prob.train_loss(ts, x0, nz, method=method, traj_per_sample=tps)
        loss.backward()
        assert loss.item() == float(fx[f"train_{method}/loss"])
        for k in names:
            assert np.array_equal(params[k].grad.numpy(), fx[f"train_{method}/grad/{k}"]), (method, k)


@pytest.mark.parametrize("path", GOLDEN_INT, ids=lambda p: Path(p).stem)
def test_euler_integrator_bit_exact(path):
    """oracle.euler_integrate / integration_case vs the reference's EulerIntegrator.integrate on LangevinSDE, bare OU
    processes and ControlledSDE (tests/golden/make_golden_integrator.py), same noise: bit-exact."""
    import json

    fx = np.load(path)
    meta = json.loads(bytes(fx["meta"]).decode())
    params = {k[len("param/"):]: torch.from_numpy(fx[k].copy()) for k in fx.files if k.startswith("param/")}
    tt = None
    if meta["target"]["kind"] == "gmm":
        tt = {k: torch.from_numpy(fx["target/" + k].copy()) for k in ("loc", "scale", "mixture_weights")}
    drift, diff = eo.integration_case(meta, params, tt)
    xs = eo.euler_integrate(drift, diff, torch.from_numpy(fx["ts"]), torch.from_numpy(fx["x_init"]),
                            torch.from_numpy(fx["timesteps"]), noise=torch.from_numpy(fx["noise"]))
    assert xs.shape == fx["xs"].shape
    assert np.array_equal(xs.detach().numpy(), fx["xs"])


@pytest.mark.parametrize("path", GOLDEN_METRICS, ids=lambda p: Path(p).stem)
def test_metrics_oracle_matches_reference(path):
    """oracle.eval_oracle.metrics_reference_keys vs the reference's get_metrics output (make_golden_metrics.py)."""
    from oracle import eval_oracle as ev
    from tests.helpers import load_metrics_fixture

    fx, meta, expected = load_metrics_fixture(path)
    st = meta["stats"]
    samples, weights = torch.from_numpy(fx["samples"]), torch.from_numpy(fx["weights"])
    kw = dict(expectations=st["expectations"], log_norm_const=st["log_norm_const"],
              stddevs=torch.from_numpy(fx["stddevs"]) if st["has_stddevs"] else None,
              domain=torch.from_numpy(fx["domain"]) if st["has_domain"] else None, log_norm_const_preds=st["preds"],
              marginal_dims=[m for m in meta["marginal_dims"] if m < meta["target"]["dim"]])
    for tag, w in (("w", weights), ("nw", None)):
        got = ev.metrics_reference_keys(samples, w, **kw)
        assert set(got) == set(expected[tag]), set(got) ^ set(expected[tag])
        for k, v in expected[tag].items():
            assert got[k] == pytest.approx(v, rel=1e-6, abs=1e-9), k


@pytest.mark.parametrize("path", GOLDEN_BRIDGE, ids=lambda p: Path(p).stem)
def test_bridge_branch_bit_exact(path):
    """TimeReversalLoss with an inference control (losses/oc.py:189-202): evaluation passes, train losses and the parameter
    gradients of BOTH networks (exact divergence, create_graph=True) against the reference run."""
    fx, prob, params, ts, x0, noise = load(path)
    assert prob.inference_ctrl is not None
    r1 = prob.eval(ts, x0, noise, compute_weights=True)
    assert np.array_equal(r1["samples"].numpy(), fx["eval1/x_T"])
    assert np.array_equal(r1["rnd"].numpy(), fx["eval1/rnd"])
    assert np.array_equal(r1["weights"].numpy(), fx["eval1/weights"])
    assert r1["log_norm_const_is"] == float(fx["eval1/log_norm_const_is"])
    assert r1["log_norm_const_lb_ito"] == float(fx["eval1/log_norm_const_lb_ito"])
    r2 = prob.eval(ts, x0, noise, compute_weights=False)
    assert np.array_equal(r2["rnd"].numpy(), fx["eval2/rnd"])
    assert r2["log_norm_const_lb"] == float(fx["eval2/log_norm_const_lb"])
    pinf = prob.inference_ctrl.p
    for method in ("kl", "lv"):
        leaves = [p.requires_grad_(True) for p in list(params.values()) + list(pinf.values())]
        for p in leaves:
            p.grad = None
        loss, _, _, _ = prob.train_loss(ts, x0, noise, method=method)
        loss.backward()
        assert loss.item() == float(fx[f"train_{method}/loss"])
        for prefix, pd in (("grad", params), ("grad_inf", pinf)):
            for k, p in pd.items():
                key = f"train_{method}/{prefix}/{k}"
                if key not in fx.files:
                    continue
                g = p.grad.numpy() if p.grad is not None else np.zeros(tuple(p.shape), np.float32)
                assert np.array_equal(g, fx[key]), key


@pytest.mark.parametrize("est", ["rademacher", "gauss"])
@pytest.mark.parametrize("path", GOLDEN_BRIDGE, ids=lambda p: Path(p).stem)
def test_bridge_hutchinson_estimators_bit_exact(path, est):
    """TimeReversalLoss.div_estimator (utils/autograd.py:25-42) in training, probes and Brownian increments replayed."""
    fx, prob, params, ts, x0, _ = load(path)
    prob.div_estimator = est
    pinf = prob.inference_ctrl.p
    for p in list(params.values()) + list(pinf.values()):
        p.requires_grad_(True)
        p.grad = None
    loss, _, _, _ = prob.train_loss(ts, x0, torch.from_numpy(fx[f"hutch_{est}/noise"]), method="lv",
                                    div_noise=torch.from_numpy(fx[f"hutch_{est}/probes"]))
    loss.backward()
    assert loss.item() == float(fx[f"hutch_{est}/loss"])
    for prefix, pd in (("grad", params), ("grad_inf", pinf)):
        for k, p in pd.items():
            key = f"hutch_{est}/{prefix}/{k}"
            if key in fx.files and p.grad is not None:
                assert np.array_equal(p.grad.numpy(), fx[key]), key


@pytest.mark.parametrize("path", GOLDEN_WIDE, ids=lambda p: Path(p).stem)
def test_wide_network_eval_passes_bit_exact(path):
    """The wide-network fixtures (C = 128 / 256, d up to 196; tests/golden/make_golden_wide.py) -- evaluation passes of the three
    loops and of the Bridge branch.  The reference ran them with four intra-op threads."""
    torch.set_num_threads(4)
    fx, prob, params, ts, x0, noise = load(path)
    assert torch.equal(prob.grid(), ts)
    r1 = prob.eval(ts, x0, noise, compute_weights=True)
    assert np.array_equal(r1["samples"].numpy(), fx["eval1/x_T"])
    assert np.array_equal(r1["rnd"].numpy(), fx["eval1/rnd"])
    assert np.array_equal(r1["weights"].numpy(), fx["eval1/weights"])
    assert r1["log_norm_const_lb_ito"] == float(fx["eval1/log_norm_const_lb_ito"])
    assert r1["log_norm_const_is"] == float(fx["eval1/log_norm_const_is"])
    assert r1["lv_loss"] == float(fx["eval1/lv_loss"])
    r2 = prob.eval(ts, x0, noise, compute_weights=False)
    assert np.array_equal(r2["samples"].numpy(), fx["eval2/x_T"])
    assert np.array_equal(r2["rnd"].numpy(), fx["eval2/rnd"])
    assert r2["log_norm_const_lb"] == float(fx["eval2/log_norm_const_lb"])


def golden_grad(fx, key):
    """(reference gradient, is_strided): large tensors of the wide fixtures are stored as every GRAD_STRIDE-th flattened entry
    (`key@stride`, tests/golden/make_golden_wide.py) together with their Euclidean norm (`key@norm`)."""
    if key in fx.files:
        return fx[key], None
    return fx[key + "@stride"], float(fx[key + "@norm"])


@pytest.mark.parametrize("method", ["kl", "lv"])
@pytest.mark.parametrize("path", GOLDEN_WIDE, ids=lambda p: Path(p).stem)
def test_wide_network_train_loss_and_grads_bit_exact(path, method):
    """Round 3: the wide fixtures also carry the reference's training losses and parameter gradients (both networks of a Bridge,
    exact divergence with create_graph=True) -- the oracle's autograd reproduces them bit for bit."""
    torch.set_num_threads(4)
    fx, prob, params, ts, x0, noise = load(path)
    pinf = prob.inference_ctrl.p if prob.inference_ctrl is not None else {}
    groups = (("grad", params), ("grad_inf", pinf))
    for _, pd in groups:
        for k, p in pd.items():
            if p.is_floating_point() and not k.endswith("timestep_coeff"):
                p.requires_grad_(True)
    loss, n_filtered, _, _ = prob.train_loss(ts, x0, noise, method=method)
    loss.backward()
    assert loss.item() == float(fx[f"train_{method}/loss"])
    assert n_filtered == int(fx[f"train_{method}/n_filtered"])
    checked = 0
    for prefix, pd in groups:
        for k, p in pd.items():
            key = f"train_{method}/{prefix}/{k}"
            if key not in fx.files and key + "@stride" not in fx.files:
                continue
            ref, norm = golden_grad(fx, key)
            g = p.grad.numpy() if p.grad is not None else np.zeros(tuple(p.shape), np.float32)
            if norm is None:
                assert np.array_equal(g, ref), key
            else:
                assert np.array_equal(g.reshape(-1)[::5], ref), key
                assert np.linalg.norm(g.astype(np.float64)) == norm, key
            checked += 1
    assert checked >= 10


@pytest.mark.parametrize("d,n,eps", [(2, 128, 0.02), (5, 160, 0.05)])
def test_sinkhorn_oracle_brackets_the_exact_assignment_cost(d, n, eps):
    """oracle/eval_oracle.py::sinkhorn_dense cannot be pinned to a run of the reference (pykeops is not in the image).  An anchor that
    needs no restatement: for uniform weights and n = m the exact optimal-transport cost is scipy's linear_sum_assignment on the Euclidean
    cost matrix, and the entropic plan's transport cost <P, M> (what eval/sinkhorn.py:169-177 returns) lies in [OT, OT + eps log n]."""
    import math

    from scipy.optimize import linear_sum_assignment

    from oracle import eval_oracle as ev

    torch.manual_seed(d)
    x, y = torch.randn(n, d).double() * 1.5, torch.randn(n, d).double() + 0.7
    M = torch.cdist(x, y).numpy()
    r, c = linear_sum_assignment(M)
    exact = float(M[r, c].mean())
    dist, corr_xy, _, _ = ev.sinkhorn_dense(x, y, eps=eps, max_iters=3000, stop_thresh=1e-7)
    assert exact - 1e-6 <= dist.item() <= exact + eps * math.log(n), (dist.item(), exact)
    # the plan's row argmax agrees with the optimal assignment for a large share of the points (measured: 0.49 / 0.7 at these eps)
    assert (corr_xy.numpy() == c[np.argsort(r)]).mean() > 0.35


# ---- the NICE flow target of BASELINE configs[4] (tests/golden/make_golden_nice.py: outputs of the reference's distr/nice.py) --------------
GOLDEN_NICE = [p for p in _ALL if Path(p).name.startswith(("nicebridge", "nicepis"))]
GOLDEN_NICE_KAT = [p for p in _ALL if Path(p).name.startswith("nice_kat")]


def _nice_density(fx, meta):
    tspec = meta["target"]
    tensors = {k[len("target/"):]: torch.from_numpy(fx[k].copy()) for k in fx.files if k.startswith("target/")}
    if not tensors:  # weights = a function of the spec's seed: rebuilt by the host mirror, checked against the reference's checksum
        import hashlib

        from sde_sampler_amd import problems

        model = problems.build_target(tspec).model
        h = hashlib.sha256()
        for k, v in model.state_dict().items():
            h.update(k.encode())
            h.update(v.detach().numpy().tobytes())
        if h.hexdigest() != bytes(fx["weights_sha256"]).decode():
            pytest.skip("this host's torch draws other initial weights than the build container's: the seed-only fixture cannot be rebuilt")
        tensors = {k: v.detach().clone() for k, v in model.state_dict().items()}
    return eo.Density(tspec, tensors)


@pytest.mark.parametrize("path", GOLDEN_NICE_KAT, ids=lambda p: Path(p).stem)
def test_nice_flow_known_answers_bit_exact(path):
    """oracle Density("nice") = the reference's Nice.unnorm_log_prob / score (distr/nice.py:176-189, 276-277; distr/base.py:130-137) on
    rows from the mode out to the logistic tails."""
    fx = np.load(path)
    meta = json.loads(bytes(fx["meta"]).decode())
    dens = _nice_density(fx, meta)
    x = torch.from_numpy(fx["x"])
    assert np.array_equal(dens.unnorm_log_prob(x).numpy(), fx["unnorm_log_prob"])
    assert np.array_equal(dens.score(x.clone()).numpy(), fx["score"])


@pytest.mark.parametrize("path", GOLDEN_NICE, ids=lambda p: Path(p).stem)
def test_nice_loss_loops_bit_exact(path):
    """Bridge (losses/oc.py:189-202) and PIS on the flow: both evaluation passes, the lv training loss and its reference-autograd gradients."""
    fx, prob, params, ts, x0, noise = load(path)
    r1 = prob.eval(ts, x0, noise, compute_weights=True)
    assert np.array_equal(r1["samples"].numpy(), fx["eval1/x_T"]) and np.array_equal(r1["rnd"].numpy(), fx["eval1/rnd"])
    assert r1["log_norm_const_is"] == float(fx["eval1/log_norm_const_is"]) and r1["lv_loss"] == float(fx["eval1/lv_loss"])
    r2 = prob.eval(ts, x0, noise, compute_weights=False)
    assert np.array_equal(r2["rnd"].numpy(), fx["eval2/rnd"]) and r2["log_norm_const_lb"] == float(fx["eval2/log_norm_const_lb"])
    params_inf = prob.inference_ctrl.p if getattr(prob, "inference_ctrl", None) is not None else {}
    leaves = [(f"grad/{k}", v) for k, v in params.items() if not k.endswith("timestep_coeff")]
    leaves += [(f"grad_inf/{k}", v) for k, v in params_inf.items() if not k.endswith("timestep_coeff")]
    for _, v in leaves:
        v.requires_grad_(True)
    for method in ("kl", "lv"):
        for _, v in leaves:
            v.grad = None
        loss, n_filtered, _, _ = prob.train_loss(ts, x0, noise, method=method)
        loss.backward()
        assert loss.item() == float(fx[f"train_{method}/loss"]) and n_filtered == int(fx[f"train_{method}/n_filtered"])
        for key, v in leaves:
            got = v.grad.numpy() if v.grad is not None else np.zeros(tuple(v.shape), np.float32)
            full = f"train_{method}/{key}"
            if full in fx.files:
                assert np.array_equal(got, fx[full]), (method, key)
            else:
                assert np.array_equal(got.reshape(-1)[::5], fx[full + "@stride"]), (method, key)

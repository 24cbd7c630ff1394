"""hipGraph replay of a training step (sde_sampler_amd/utils/graphs.py) and the device-resident Philox offset
(SdehProblem.rng_offset_dev, include/sdeh.h) it relies on.  The graph-safe loss reductions are also covered on the CPU."""
import copy

import pytest
import torch

from sde_sampler_amd import problems
from sde_sampler_amd.losses.oc import TimeReversalLoss


def _masked_case(method, tps):
    torch.manual_seed(3)
    rnd = torch.randn(24 * tps, 1, dtype=torch.float64) * 3.0
    rnd[5] = float("inf")
    rnd[11] = float("nan")
    return rnd.requires_grad_(True)


@pytest.mark.parametrize("method,tps", [("kl", 1), ("kl_ito", 1), ("lv", 1), ("lv_traj", 4)])
def test_graph_safe_loss_matches_reference_reduction(method, tps):
    """losses/oc.py:72-92 as masked reductions: same value and the same gradient w.r.t. every rnd row (zero on filtered rows)."""
    out = []
    for safe in (False, True):
        lo = TimeReversalLoss(generative_ctrl=None, sde=None, method=method, traj_per_sample=tps)
        lo.graph_safe = safe
        rnd = _masked_case(method, tps)
        value, metrics = lo.compute_loss(rnd)
        (g,) = torch.autograd.grad(value, rnd)
        out.append((value.detach(), g, int(metrics["train/n_filtered_cumulative"]), lo.state_dict()["n_filtered"]))
    (v0, g0, n0, s0), (v1, g1, n1, s1) = out
    assert n0 == n1 == s0 == s1 and n0 > 0
    torch.testing.assert_close(v1, v0, rtol=1e-12, atol=1e-12)
    torch.testing.assert_close(g1, g0, rtol=1e-10, atol=1e-12)


def _build(seed=0, method="lv"):
    if method == "bridge":  # two networks, exact divergence: conf/solver/bridge.yaml style on the shifted double well
        lerp = dict(clip_model=10.0, clip_score=10.0, scale_score=1.0, gamma_dim=1, gamma_bias=1.0)
        spec = dict(batch=1024, target=dict(kind="double_well", dim=1, separation=2.0, shift=1.5),
                    prior=dict(kind="iso_gauss", dim=1), sde=dict(kind="scaled_bm", diff_coeff=2.0, terminal_t=1.0),
                    ctrl=dict(kind="lerp_target", **lerp), inference_ctrl=dict(kind="lerp_prior", **lerp),
                    net=dict(channels=64, num_layers=4, activation="gelu"),
                    loss=dict(kind="time_reversal", method="lv", max_rnd=1e8), grid=dict(start=0.0, end=1.0, steps=50))
    elif method == "wide_bridge_gmm":  # wide Bridge on the benchmark's mixture, kl: the forward launch keeps both score planes
        lerp = dict(clip_model=10.0, clip_score=10.0, scale_score=1.0, gamma_dim=1, gamma_bias=1.0)
        spec = dict(batch=256, target=dict(kind="gmm", dim=50, name="fab50"), prior=dict(kind="iso_gauss", dim=50),
                    sde=dict(kind="scaled_bm", diff_coeff=1.0, terminal_t=1.0), ctrl=dict(kind="lerp_target", **lerp),
                    inference_ctrl=dict(kind="lerp_prior", **lerp), net=dict(channels=128, num_layers=4, activation="gelu"),
                    loss=dict(kind="time_reversal", method="kl"), grid=dict(start=0.0, end=1.0, steps=6))
    elif method in ("nice_bridge", "nice_bridge_kl"):  # BASELINE configs[4] as written at a small size: the stepped forward around the flow's score
        spec = problems.baseline_spec("cfg5_nice_bridge196")
        spec["grid"]["steps"], spec["net"]["channels"] = 5, 128
        spec["target"] = dict(kind="nice", dim=196, coupling=2, mid_dim=64, hidden=3)
        spec["loss"]["method"] = "kl" if method.endswith("kl") else "lv"
        if method.endswith("kl"):
            spec["loss"]["max_rnd"] = None
    elif method.startswith("wide"):  # wide networks (csrc/sdeh_wide_bwd.hip): "wide_lv" / "wide_kl" plain, "wide_bridge" configs[4]'s shape
        spec = problems.baseline_spec("cfg5_like_bridge196" if method == "wide_bridge" else "wide_pis_funnel196")
        spec["grid"]["steps"] = 6 if method == "wide_bridge" else 12
        spec["loss"]["method"] = "kl" if method == "wide_kl" else "lv"
    else:
        spec = problems.baseline_spec("cfg1_dw_dis_lv")
        spec["loss"]["method"] = method
    torch.manual_seed(seed)
    return problems.build(spec, device="cuda:0")


def _params(prob):
    params = list(prob.ctrl.parameters())
    inf = getattr(prob.loss, "inference_ctrl", None)
    return params + (list(inf.parameters()) if inf is not None else [])


@pytest.mark.gpu
def test_device_offset_equals_by_value_offset():
    """offset = 2 by value + 3 on the device draws exactly the noise of offset = 5 by value."""
    prob = _build()
    x = prob.prior.sample((1024,))
    with torch.no_grad():
        prob.loss.engine.calls = 5
        a = prob.loss.simulate(prob.ts, x, prob.target.unnorm_log_prob, prob.second_log_prob, train=False, compute_ito_int=True)
        prob.loss.engine.calls = 2
        prob.loss.rng_counter = torch.tensor([3], dtype=torch.int64, device="cuda:0")
        b = prob.loss.simulate(prob.ts, x, prob.target.unnorm_log_prob, prob.second_log_prob, train=False, compute_ito_int=True)
        prob.loss.rng_counter.add_(1)
        c = prob.loss.simulate(prob.ts, x, prob.target.unnorm_log_prob, prob.second_log_prob, train=False, compute_ito_int=True)
    assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])
    assert not torch.equal(a[0], c[0])
    with pytest.raises(ValueError):
        prob.loss.rng_counter = torch.tensor([3.0], device="cuda:0")
        prob.loss.simulate(prob.ts, x, prob.target.unnorm_log_prob, prob.second_log_prob, train=False)


@pytest.mark.gpu
@pytest.mark.parametrize("method", ["lv", "kl", "bridge"])
def test_graphed_train_step_matches_eager(method):
    """W eager warm-up steps + K graph replays leave the parameters where W + K eager steps (same Philox offsets) leave them."""
    from sde_sampler_amd.utils.graphs import COUNTER_START, GraphedTrainStep

    W, K, B = 2, 5, 1024
    prob_g, prob_e = _build(1, method), _build(1, method)
    x = prob_g.prior.sample((B,))
    for p, q in zip(_params(prob_g), _params(prob_e)):
        assert torch.equal(p, q)

    def make(prob):
        params = _params(prob)
        opt = torch.optim.Adam(params, lr=2e-3, capturable=True)
        fn = lambda: prob.loss(prob.ts, x, prob.target.unnorm_log_prob, prob.second_log_prob)[0]
        clip = lambda: torch.nn.utils.clip_grad_norm_(params, 1.0)
        return params, opt, fn, clip

    params_g, opt_g, fn_g, clip_g = make(prob_g)
    graphed = GraphedTrainStep(fn_g, [prob_g.loss], opt_g, after_backward=clip_g, warmup=W)
    losses_g = [graphed().clone() for _ in range(K)]
    assert int(graphed.counter) == COUNTER_START + W + K

    params_e, opt_e, fn_e, clip_e = make(prob_e)
    lo = prob_e.loss
    lo.graph_safe, lo.rng_counter = True, torch.full((1,), COUNTER_START, dtype=torch.int64, device="cuda:0")

    def eager_step():
        opt_e.zero_grad(set_to_none=True)
        value = fn_e()
        value.backward()
        clip_e()
        opt_e.step()
        lo.rng_counter.add_(1)
        return value.detach().clone()

    for _ in range(W):
        eager_step()
    frozen = lo.engine.calls  # the by-value offset the capture froze
    losses_e = []
    for _ in range(K):
        lo.engine.calls = frozen
        losses_e.append(eager_step())
    torch.testing.assert_close(torch.stack(losses_g), torch.stack(losses_e), rtol=2e-4, atol=1e-5)
    assert len({float(v) for v in losses_g}) == K  # fresh noise in every replay
    for p, q in zip(params_g, params_e):
        torch.testing.assert_close(p, q, rtol=2e-3, atol=2e-5)


@pytest.mark.gpu
def test_graphed_step_skips_non_finite_updates_on_device():
    """solver/base.py:409-432 (`if loss_ok and grad_ok`) without a host round trip: a poisoned replay leaves parameters and
    optimizer state untouched, is counted, and the following replays carry on."""
    from sde_sampler_amd.utils.graphs import GraphedTrainStep

    prob = _build(2, "lv")
    x = prob.prior.sample((512,))
    params = _params(prob)
    opt = torch.optim.Adam(params, lr=2e-3, capturable=True)
    poison = torch.ones((), device="cuda:0")
    fn = lambda: prob.loss(prob.ts, x, prob.target.unnorm_log_prob, prob.second_log_prob)[0] * poison
    graphed = GraphedTrainStep(fn, [prob.loss], opt, warmup=2)
    graphed()
    before = [p.detach().clone() for p in params]
    state_before = [opt.state[p]["exp_avg"].clone() for p in params]
    steps_before = float(opt.state[params[0]]["step"])
    poison.fill_(float("nan"))
    assert not torch.isfinite(graphed())
    assert int(graphed.n_skipped) == 1
    for p, q in zip(params, before):
        assert torch.equal(p, q)
    for p, m in zip(params, state_before):
        assert torch.equal(opt.state[p]["exp_avg"], m)
    assert float(opt.state[params[0]]["step"]) == steps_before
    poison.fill_(1.0)
    assert torch.isfinite(graphed())
    assert int(graphed.n_skipped) == 1
    assert any(not torch.equal(p, q) for p, q in zip(params, before))
    assert all(torch.isfinite(p).all() for p in params)


@pytest.mark.gpu
@pytest.mark.parametrize("method,batch", [("lv", 2048), ("kl", 2048), ("bridge", 2048), ("lv", 65536), ("kl", 40000), ("bridge", 16384),
                                          ("wide_lv", 2048), ("wide_kl", 1000), ("wide_bridge", 256), ("wide_bridge_gmm", 200),
                                          ("nice_bridge", 200), ("nice_bridge_kl", 100)])
def test_replayed_gradients_equal_eager_gradients(method, batch):
    """Every parameter gradient of forward + backward replayed from a hipGraph (three replays) against the eager launch at the
    same Philox offset.  Guards against ordering / buffer-reuse hazards of captured steps (a multi-block framework reduction
    inside the step corrupted the bias gradients from the second replay on, tests/perf/rocm_graph_two_reductions.py)."""
    prob = _build(4, method)
    params = _params(prob)
    x = prob.prior.sample((batch,))
    lo = prob.loss
    lo.graph_safe = True
    lo.rng_counter = torch.zeros(1, dtype=torch.int64, device="cuda:0")

    def run():
        for p in params:
            p.grad = None
        lo.engine.calls = 5
        value = lo(prob.ts, x, prob.target.unnorm_log_prob, prob.second_log_prob)[0]
        value.backward()
        return value

    eager_value = run().detach().clone()
    eager = [p.grad.clone() for p in params]
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        run()
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    for p in params:
        p.grad = None
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        value = run()
    inf = getattr(lo, "inference_ctrl", None)
    names = [n for n, _ in prob.ctrl.named_parameters()] + (["inference." + n for n, _ in inf.named_parameters()] if inf is not None else [])
    for rep in range(3):
        graph.replay()
        torch.cuda.synchronize()
        torch.testing.assert_close(value, eager_value, rtol=1e-6, atol=1e-6)
        for name, p, g in zip(names, params, eager):
            scale = float(g.abs().max()) + 1e-12
            err = float((p.grad - g).abs().max())
            assert err <= 1e-5 * scale, (rep, name, err, scale)


@pytest.mark.gpu
@pytest.mark.parametrize("method", ["kl", "lv"])
def test_graph_safe_loss_on_device_matches_reference_reduction(method):
    """On the GPU the graph-safe loss takes its sums from sdeh_reduce_estimators: same value / gradient as rnd[mask].mean() / .var()."""
    out = []
    for safe in (False, True):
        lo = TimeReversalLoss(generative_ctrl=None, sde=None, method=method, max_rnd=50.0)
        lo.graph_safe = safe
        torch.manual_seed(11)
        rnd = (torch.randn(4096, 1, device="cuda:0") * 3.0 + 1.0)
        rnd[7], rnd[100], rnd[4000] = float("inf"), float("nan"), 77.0  # the last one exceeds max_rnd
        rnd.requires_grad_(True)
        value, metrics = lo.compute_loss(rnd)
        (g,) = torch.autograd.grad(value, rnd)
        out.append((value.detach(), g, int(metrics["train/n_filtered_cumulative"])))
    (v0, g0, n0), (v1, g1, n1) = out
    assert n0 == n1 == 3
    torch.testing.assert_close(v1, v0, rtol=2e-6, atol=1e-6)
    torch.testing.assert_close(g1, g0, rtol=1e-5, atol=1e-9)


@pytest.mark.gpu
def test_graphed_step_after_eager_steps_with_live_autograd_graph():
    """An eager training step first, its loss tensor still referenced (README flow): the stale AccumulateGrad nodes of the
    parameters must not take part in the capture."""
    from sde_sampler_amd.utils.graphs import GraphedTrainStep

    prob = _build(3, "lv")
    params = _params(prob)
    opt = torch.optim.Adam(params, lr=1e-3)
    kept = prob.loss(prob.ts, prob.prior.sample((1024,)), prob.target.unnorm_log_prob, prob.second_log_prob)[0]
    kept.backward(retain_graph=True)  # `kept` keeps the autograd graph (and the accumulators) alive
    opt.step()
    opt_g = torch.optim.Adam(params, lr=1e-3, capturable=True)
    graphed = GraphedTrainStep(lambda: prob.loss(prob.ts, prob.prior.sample((1024,)), prob.target.unnorm_log_prob,
                                                 prob.second_log_prob)[0], [prob.loss], opt_g)
    before = [p.detach().clone() for p in params]
    for _ in range(3):
        value = graphed()
    assert torch.isfinite(value) and torch.isfinite(kept)
    assert any(not torch.equal(p, q) for p, q in zip(params, before))


_N_RANDOM = 12 * int(__import__("os").environ.get("SDEH_FUZZ_SCALE", "1"))


@pytest.mark.gpu
@pytest.mark.parametrize("case", range(_N_RANDOM))
def test_random_problem_replayed_gradients_equal_eager(case):
    """The replay-vs-eager gradient check over random problems (loss x control x SDE x target x network shape x method): every
    code path a training step can take must be capture-safe, not only the configurations picked above."""
    import numpy as np
    import test_hip_fuzz as F
    from sde_sampler_amd import SdehUnsupported

    rng = np.random.default_rng(21000 + case)
    spec = F.random_spec(rng)
    method = str(rng.choice(["kl", "kl_ito", "lv", "lv_traj"]))
    spec["loss"]["method"] = method
    spec["loss"]["max_rnd"] = 1e8 if method.startswith("lv") else None
    if method == "lv_traj":
        spec["loss"]["traj_per_sample"] = 2
    B = int(rng.choice([64, 100, 2048]))
    torch.manual_seed(case)
    prob = problems.build(spec, device="cuda:0")
    params = _params(prob)
    x = prob.prior.sample((B,))
    lo = prob.loss
    lo.graph_safe = True
    lo.rng_counter = torch.zeros(1, dtype=torch.int64, device="cuda:0")

    def run():
        for p in params:
            p.grad = None
        lo.engine.calls = 9
        value = lo(prob.ts, x, prob.target.unnorm_log_prob, prob.second_log_prob)[0]
        value.backward()
        return value

    try:
        eager_value = run().detach().clone()
    except SdehUnsupported as exc:
        pytest.skip(str(exc)[:120])
    eager = [None if p.grad is None else p.grad.clone() for p in params]
    if not torch.isfinite(eager_value) or any(g is not None and not torch.isfinite(g).all() for g in eager):
        pytest.skip("random configuration without a finite loss / gradient")
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        run()
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    for p in params:
        p.grad = None
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        value = run()
    for rep in range(3):
        graph.replay()
        torch.cuda.synchronize()
        torch.testing.assert_close(value, eager_value, rtol=1e-5, atol=1e-6)
        for (name, _), p, g in zip(prob.ctrl.named_parameters(), params, eager):
            if g is None:
                assert p.grad is None
                continue
            scale = float(g.abs().max()) + 1e-12
            err = float((p.grad - g).abs().max())
            assert err <= 1e-5 * scale, (rep, method, spec["loss"]["kind"], spec["ctrl"]["kind"], spec["target"]["kind"], name, err, scale)


@pytest.mark.gpu
@pytest.mark.parametrize("name,weights", [("gmm50_pis_headline", False), ("cfg2_gmm2_dis_kl", True), ("cfg4_funnel_dds_lv", True),
                                          ("cfg5_nice_bridge196", True)])  # (the stepped path: segments + the flow's GEMM chain in one graph)
def test_graphed_eval_equals_eager_eval(name, weights):
    """utils.graphs.GraphedEval: the replayed evaluation (one launch + the 8-float copy) returns bit for bit what the eager
    `loss.eval` returns at the same Philox offset -- samples, importance weights and every estimator -- draws fresh noise per replay,
    and takes new inputs through its static buffer."""
    from sde_sampler_amd.utils.graphs import GraphedEval

    spec = problems.baseline_spec(name)
    spec["batch"] = 4096
    if name == "cfg5_nice_bridge196":
        spec["grid"]["steps"] = 4
    prob = problems.build(spec, device="cuda:0")
    torch.manual_seed(5)
    x0 = prob.prior.sample((4096,))
    eng = prob.loss.engine
    calls0 = eng.calls
    ge = GraphedEval(lambda x: prob.eval(x, compute_weights=weights, return_traj=False), [prob.loss], x0, warmup=1)
    assert eng.calls == calls0 + 2  # one warm-up launch + the captured one
    c = int(ge.counter)
    res = ge(x0)
    samples, w = res.samples.clone(), None if res.weights is None else res.weights.clone()
    assert int(ge.counter) == c + 1
    res2 = ge()  # fresh noise
    assert not torch.equal(res2.samples, samples)
    # the eager launch at the captured offset (calls0 + 1) and the counter value of the first replay
    eng.calls = calls0 + 1
    ge.counter.fill_(c)
    ref = prob.eval(x0, compute_weights=weights, return_traj=False)
    assert torch.equal(ref.samples, samples)
    assert ref.log_norm_const_preds == res.log_norm_const_preds and ref.metrics == res.metrics
    if weights:
        assert torch.equal(ref.weights, w)
    x1 = prob.prior.sample((4096,))
    ge.counter.fill_(c)
    res3 = ge(x1)
    eng.calls = calls0 + 1
    ge.counter.fill_(c)
    ref3 = prob.eval(x1, compute_weights=weights, return_traj=False)
    assert torch.equal(ref3.samples, res3.samples) and ref3.log_norm_const_preds == res3.log_norm_const_preds


@pytest.mark.gpu
def test_guard_restore_is_one_launch_that_only_acts_on_a_rejected_step():
    """sdeh_guard_restore (include/sdeh.h): tensors come back from their snapshots bit for bit when ok == 0 and are left alone when ok == 1;
    4- and 8-byte element types, sizes that are not a multiple of the block."""
    from sde_sampler_amd import _lib as L

    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    shapes = [(64, 64), (64,), (1,), (3, 50), (4097,)]
    new = [torch.randn(s, device=dev) for s in shapes] + [torch.arange(7, device=dev, dtype=torch.int64), torch.randn(5, device=dev, dtype=torch.float64)]
    old = [torch.randn_like(t) if t.is_floating_point() else t + 100 for t in new]
    table = torch.tensor([[t.data_ptr(), s.data_ptr(), t.numel() * t.element_size() // 4] for t, s in zip(new, old)], dtype=torch.int64, device=dev)
    kept = [t.clone() for t in new]
    skipped = torch.zeros((), dtype=torch.int64, device=dev)
    for ok_value in (True, False):
        ok = torch.tensor([ok_value], device=dev)
        L.check(L.load().sdeh_guard_restore(table.data_ptr(), len(new), ok.data_ptr(), skipped.data_ptr(), torch.cuda.current_stream(dev).cuda_stream))
        torch.cuda.synchronize()
        for t, k, s in zip(new, kept, old):
            assert torch.equal(t, k if ok_value else s)
        assert int(skipped) == (0 if ok_value else 1)


@pytest.mark.gpu
@pytest.mark.parametrize("n", [1, 1023, 50_000])
def test_guard_check_decides_like_the_reference_trainer(n):
    """sdeh_guard_check against solver/base.py:409-421 written in torch: loss finite (or within max_loss), every gradient finite; a rejected
    step's gradients come back zeroed, an accepted step's untouched."""
    from sde_sampler_amd import _lib as L

    dev = torch.device("cuda:0")
    torch.manual_seed(n)
    lib, st = L.load(), torch.cuda.current_stream(dev).cuda_stream
    for value, max_loss, poison in [(1.5, None, None), (float("nan"), None, None), (float("inf"), None, None), (1.5, None, "nan"),
                                    (1.5, None, "inf"), (-7.0, 5.0, None), (-4.0, 5.0, None), (-4.0, 5.0, "-inf"), (float("nan"), 5.0, None)]:
        g = torch.randn(n, device=dev)
        if poison is not None:
            g[n - 1 - (n // 3)] = float(poison)
        v = torch.tensor([value], device=dev)
        want = bool((torch.isfinite(v) if max_loss is None else v.abs() <= max_loss).all() and torch.isfinite(g).all())
        kept = g.clone()
        ok = torch.empty(1, dtype=torch.bool, device=dev)
        L.check(lib.sdeh_guard_check(g.data_ptr(), n, v.data_ptr(), -1.0 if max_loss is None else max_loss, ok.data_ptr(), st))
        assert bool(ok) == want, (value, max_loss, poison)
        assert torch.equal(g, kept) if want else not g.any()


@pytest.mark.gpu
def test_guarded_capture_with_one_warmup_step_builds_its_pointer_table_first():
    """with warmup = 1 and Adam (state appears in the first step) the update guard's pointer table used to be built -- a
    host-to-device copy -- inside the capture.  The constructor now keeps stepping eagerly until the table the capture will ask for exists."""
    from sde_sampler_amd.utils.graphs import GraphedTrainStep

    prob = _build(4, "lv")
    x = prob.prior.sample((512,))
    params = _params(prob)
    opt = torch.optim.Adam(params, lr=1e-3, capturable=True)
    fn = lambda: prob.loss(prob.ts, x, prob.target.unnorm_log_prob, prob.second_log_prob)[0]
    graphed = GraphedTrainStep(fn, [prob.loss], opt, warmup=1)
    assert graphed.extra_warmup == 1 and graphed._table_ready()
    before = [p.detach().clone() for p in params]
    for _ in range(3):
        assert torch.isfinite(graphed())
    assert int(graphed.n_skipped) == 0 and any(not torch.equal(p, q) for p, q in zip(params, before))
    # the default warm-up needs no extra step (existing runs keep their step count)
    prob2 = _build(4, "lv")
    opt2 = torch.optim.Adam(_params(prob2), lr=1e-3, capturable=True)
    fn2 = lambda: prob2.loss(prob2.ts, x, prob2.target.unnorm_log_prob, prob2.second_log_prob)[0]
    assert GraphedTrainStep(fn2, [prob2.loss], opt2, warmup=2).extra_warmup == 0


@pytest.mark.gpu
def test_guard_decides_on_the_reduced_gradients():
    """`reduce_gradients` (the data-parallel all-reduce) runs BEFORE the finite-gradient check, so a NaN that arrives with the
    reduction (another rank's, or all_reduce_gradients' disagreement poison) rejects the step; `after_backward` (clipping) runs after the
    check, on accepted steps' gradients, as solver/base.py:409-427 orders them."""
    from sde_sampler_amd.utils.graphs import GraphedTrainStep

    prob = _build(5, "lv")
    x = prob.prior.sample((512,))
    params = _params(prob)
    opt = torch.optim.Adam(params, lr=1e-3, capturable=True)
    other_rank = torch.zeros((), device="cuda:0")  # what "arrives" with the reduction: 0 = nothing, NaN = a poisoned bucket
    order = []

    def reduce():
        order.append("reduce")
        for p in params:
            if p.grad is not None:
                p.grad.add_(other_rank)

    def clip():
        order.append("clip")
        torch.nn.utils.clip_grad_norm_(params, 1.0)

    fn = lambda: prob.loss(prob.ts, x, prob.target.unnorm_log_prob, prob.second_log_prob)[0]
    graphed = GraphedTrainStep(fn, [prob.loss], opt, reduce_gradients=reduce, after_backward=clip, warmup=3)
    assert order[:2] == ["reduce", "clip"]
    graphed()
    before = [p.detach().clone() for p in params]
    other_rank.fill_(float("nan"))
    graphed()
    assert int(graphed.n_skipped) == 1 and all(torch.equal(p, q) for p, q in zip(params, before))
    other_rank.zero_()
    graphed()
    assert int(graphed.n_skipped) == 1 and any(not torch.equal(p, q) for p, q in zip(params, before))
    assert all(torch.isfinite(p).all() for p in params)

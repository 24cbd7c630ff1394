"""CPU tests (no GPU): the C-ABI library loads and exports every symbol include/sdeh.h declares, the host
classes keep the reference's plugin contract, the engine maps objects onto the ABI structs correctly, and the
product path fails loudly (no CPU fallback, no oracle import)."""
import ctypes
import json
import math
import os
import re
import subprocess
import sys
from pathlib import Path

import numpy as np
import pytest
import torch

from tests.helpers import GOLDEN, load_fixture

ROOT = Path(__file__).resolve().parents[1]


def test_library_exports_every_declared_symbol():
    from sde_sampler_amd import _lib as L

    header = (ROOT / "include" / "sdeh.h").read_text()
    declared = set(re.findall(r"\b(sdeh_[a-z_0-9]+)\s*\(", header))
    assert declared, "no prototypes found in include/sdeh.h"
    assert declared == set(L.PROTOTYPES), (declared ^ set(L.PROTOTYPES))
    lib = L.load()
    for name in declared:
        assert hasattr(lib, name), name
    assert lib.sdeh_abi_version() == L.SDEH_ABI_VERSION == int(re.search(r"#define SDEH_ABI_VERSION (\d+)", header).group(1))
    assert int(re.search(r"#define SDEH_MAX_HIDDEN (\d+)", header).group(1)) == L.SDEH_MAX_HIDDEN
    assert int(re.search(r"#define SDEH_REDUCE_SCRATCH (\d+)", header).group(1)) == L.SDEH_REDUCE_SCRATCH


def test_abi_argument_validation_without_gpu():
    """Entry points reject bad arguments before touching the device (no compute calls are made here)."""
    from sde_sampler_amd import _lib as L

    lib = L.load()
    plan = ctypes.c_void_p()
    assert lib.sdeh_plan_create(None, ctypes.byref(plan)) == -1
    bad = L.SdehPlanDesc(dim=0, channels=64, max_hidden=2, max_steps=10, max_components=0, device=0)
    assert lib.sdeh_plan_create(ctypes.byref(bad), ctypes.byref(plan)) == -1
    assert b"dim=0" in lib.sdeh_last_error()
    odd = L.SdehPlanDesc(dim=2, channels=96, max_hidden=2, max_steps=10, max_components=0, device=0)
    assert lib.sdeh_plan_create(ctypes.byref(odd), ctypes.byref(plan)) == -2  # SDEH_ERR_UNSUPPORTED: C in {64, 128, 256}
    huge = L.SdehPlanDesc(dim=300, channels=256, max_hidden=2, max_steps=10, max_components=0, device=0)
    assert lib.sdeh_plan_create(ctypes.byref(huge), ctypes.byref(plan)) == -2 and b"d <= 256" in lib.sdeh_last_error()
    assert lib.sdeh_simulate_fwd(None, None, None, 1, None, 1, None, 0, 0, 0, None, None, None, None) == -1
    assert lib.sdeh_reduce_estimators(None, 1, 0.0, None, None, None) == -1
    with pytest.raises(L.SdehUnsupported):
        L.check(-2)
    with pytest.raises(L.SdehError):
        L.check(-1)


def test_struct_layout_matches_header():
    """sizeof of the ctypes mirrors == what hipcc/gcc compute for include/sdeh.h."""
    from sde_sampler_amd import _lib as L

    src = '#include <stdio.h>\n#include "sdeh.h"\nint main(){printf("%zu %zu %zu %zu %zu\\n", sizeof(SdehDensity), ' \
          'sizeof(SdehTimeEmbed), sizeof(SdehFourierMLP), sizeof(SdehProblem), sizeof(SdehPlanDesc));return 0;}\n'
    exe = Path(os.environ.get("TMPDIR", "/tmp")) / "sdeh_sizeof"
    subprocess.run(["gcc", "-x", "c", "-", "-I", str(ROOT / "include"), "-o", str(exe)], input=src.encode(), check=True)
    sizes = list(map(int, subprocess.run([str(exe)], capture_output=True, check=True).stdout.split()))
    assert sizes == [ctypes.sizeof(t) for t in (L.SdehDensity, L.SdehTimeEmbed, L.SdehFourierMLP, L.SdehProblem, L.SdehPlanDesc)]


def test_product_never_imports_oracle_and_fails_loudly_without_library(tmp_path):
    code = (
        "import sys; sys.path.insert(0, %r)\n"
        "import sde_sampler_amd.losses.oc, sde_sampler_amd.problems, sde_sampler_amd.engine\n"
        "assert not any(m == 'oracle' or m.startswith('oracle.') for m in sys.modules), 'product imported the oracle'\n"
        "from sde_sampler_amd import _lib\n"
        "try:\n    _lib.load()\nexcept _lib.SdehLibraryError as e:\n    print('LOUD:', e)\nelse:\n    raise SystemExit('no error')\n"
    ) % str(ROOT)
    env = dict(os.environ, SDEH_LIBRARY=str(tmp_path / "missing.so"))
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True)
    assert out.returncode == 0, out.stderr
    assert "LOUD:" in out.stdout and "no CPU fallback" in out.stdout
    pat = re.compile(r"^\s*(from|import)\s+oracle\b", re.M)
    for path in list((ROOT / "sde_sampler_amd").rglob("*.py")) + list((ROOT / "tools").rglob("*.py")):
        assert not pat.search(path.read_text()), f"{path} imports the oracle"
    # the oracle is test infrastructure: outside tests/ only smoke() and the cpu_baseline leg of bench.py reach for it
    bench = (ROOT / "bench.py").read_text()
    assert len(pat.findall(bench)) == 1
    assert bench.split("from oracle")[0].rsplit("\ndef ", 1)[-1].startswith("cpu_baseline(")  # inside cpu_baseline()


def test_cpu_tensors_are_rejected():
    from sde_sampler_amd import problems

    prob = problems.build(problems.baseline_spec("cfg1_dw_dis_lv"))
    with pytest.raises(RuntimeError, match="no CPU path"):
        prob.eval(torch.zeros(4, 1))


@pytest.mark.parametrize("path", GOLDEN, ids=lambda p: Path(p).stem)
def test_engine_introspection(path):
    """Objects -> SdehProblem mapping for every golden configuration (kinds, sizes, scalars)."""
    from sde_sampler_amd import _lib as L
    from sde_sampler_amd import engine as E
    from sde_sampler_amd import problems
    from sde_sampler_amd.losses.oc import _resolve_gaussian_log_prob, _resolve_terminal

    fx, meta, params, tt = load_fixture(path)
    prob = problems.build(meta, params, tt)
    keep = E._Keep()
    target, clip = _resolve_terminal(prob.target.unnorm_log_prob)
    second = _resolve_gaussian_log_prob(prob.second_log_prob)
    assert target is prob.target and clip is None and second is not None
    ref_prior = prob.loss._reference_prior() if hasattr(prob.loss, "_reference_prior") else None
    pr = prob.loss.engine.build_problem(loss_kind=prob.loss._LOSS_KIND, generative_ctrl=prob.ctrl, sde=prob.sde, flags=0,
                                        device=torch.device("cpu"), keep=keep, terminal_target=target, second=second,
                                        reference_prior=ref_prior, alpha=getattr(prob.loss, "alpha", 0.0),
                                        sigma=getattr(prob.loss, "sigma", 0.0))
    kinds = {"clipped": L.CTRL_CLIPPED, "score": L.CTRL_SCORE, "lerp": L.CTRL_LERP, "lerp_target": L.CTRL_LERP_TARGET,
             "lerp_prior": L.CTRL_LERP_PRIOR}
    assert pr.ctrl_kind == kinds[meta["ctrl"]["kind"]]
    assert pr.base_model.dim == meta["target"]["dim"] and pr.base_model.channels == 64
    assert pr.base_model.n_hidden == meta["net"]["num_layers"] - 2
    assert pr.base_model.activation == {"gelu": 0, "silu": 1, "relu": 2}[meta["net"]["activation"]]
    assert pr.base_model.timestep_embed.n_hidden == 1 and pr.base_model.timestep_embed.dim_out == 64
    if meta["ctrl"]["kind"] != "clipped":
        assert pr.score_model.n_hidden == 3 and pr.score_model.dim_out == meta["ctrl"]["gamma_dim"]
        assert pr.clip_score == pytest.approx(meta["ctrl"]["clip_score"])
    assert pr.clip_model == pytest.approx(meta["ctrl"]["clip_model"])
    sde = meta["sde"]
    assert pr.sde_kind == (0 if sde is None else {"vp": 1, "const_ou": 2, "scaled_bm": 2}[sde["kind"]])
    if sde and sde["kind"] == "vp":
        assert (pr.vp_beta_min, pr.vp_beta_max) == pytest.approx((sde["beta_min"], sde["beta_max"]))
    if sde and sde["kind"] == "scaled_bm":
        assert pr.ou_drift == 0.0 and pr.ou_diff == pytest.approx(sde["diff_coeff"])
    tk = {"gmm": L.DENS_GMM, "double_well": L.DENS_MULTI_WELL, "multi_well": L.DENS_MULTI_WELL, "funnel": L.DENS_FUNNEL,
          "iso_gauss": L.DENS_DIAG_GAUSS}[meta["target"]["kind"]]
    assert pr.target.kind == tk and pr.target.dim == meta["target"]["dim"]
    assert pr.second.kind == L.DENS_DIAG_GAUSS
    if meta["loss"].get("reference_ctrl"):
        assert pr.prior.kind == L.DENS_DIAG_GAUSS
    assert all(t.dtype == torch.float32 and t.is_contiguous() for t in keep)


def test_host_modules_reproduce_reference_values():
    """The host-side nn.Modules (parameter containers) define the same functions as the reference: control output,
    log-densities and scores against the vectors captured from the reference."""
    from sde_sampler_amd import problems

    for path in GOLDEN:
        fx, meta, params, tt = load_fixture(path)
        prob = problems.build(meta, params, tt)
        x = torch.from_numpy(fx["kat/x"])
        np.testing.assert_allclose(prob.target.unnorm_log_prob(x).detach().numpy(), fx["kat/target_unnorm_log_prob"], rtol=2e-5, atol=2e-5)
        np.testing.assert_allclose(prob.target.score(x.clone()).detach().numpy(), fx["kat/target_score"], rtol=2e-5, atol=2e-5)
        np.testing.assert_allclose(prob.second_log_prob(x).detach().numpy(), fx["kat/second_log_prob"], rtol=2e-5, atol=2e-5)
        assert torch.equal(prob.ts, torch.from_numpy(fx["ts"]))


def test_loss_contract():
    from sde_sampler_amd.losses.oc import BaseOCLoss, ExponentialIntegratorSDELoss, ReferenceSDELoss, TimeReversalLoss
    from sde_sampler_amd.utils.common import Results

    with pytest.raises(ValueError, match="Unknown loss method"):
        TimeReversalLoss(generative_ctrl=None, method="foo")
    with pytest.raises(ValueError, match="single trajectory"):
        ReferenceSDELoss(generative_ctrl=None, method="lv_traj", traj_per_sample=1)
    loss = ExponentialIntegratorSDELoss(generative_ctrl=None, alpha=1.0, sigma=2.0, method="lv", max_rnd=1e8, unknown_kw=1)
    # reference keys + the position in the noise stream (call count + the device counter of replayed hipGraphs)
    assert loss.state_dict() == {"n_filtered": 0, "rng_calls": 0, "rng_counter": 0, "rng_replays": 0}
    loss.load_state_dict({"n_filtered": 7})  # a reference checkpoint (losses/oc.py:133-137) loads as is
    assert loss.n_filtered == 7 and (loss.alpha, loss.sigma) == (1.0, 2.0) and loss.engine.calls == 0
    loss.load_state_dict({"n_filtered": 7, "rng_calls": 12})
    assert loss.engine.calls == 12 and loss.engine.offset() == 12
    # resuming across eager <-> graphed runs: a graphed loss carries a device counter that starts at COUNTER_START
    from sde_sampler_amd.utils.graphs import COUNTER_START
    graphed = ExponentialIntegratorSDELoss(generative_ctrl=None, alpha=1.0, sigma=2.0, method="lv")
    graphed.rng_counter = torch.full((1,), COUNTER_START + 5, dtype=torch.int64)
    sd = graphed.state_dict()
    assert sd["rng_counter"] == COUNTER_START + 5 and sd["rng_replays"] == 5
    graphed.load_state_dict({"n_filtered": 0})  # a checkpoint without the keys (any earlier one) leaves the counter alone
    assert int(graphed.rng_counter) == COUNTER_START + 5
    # eager -> graphed: the counter stays where the capture's warm-up steps put it (a rewind to COUNTER_START would reuse the
    # Philox offsets those steps consumed)
    graphed.load_state_dict({"n_filtered": 0, "rng_calls": 12, "rng_counter": 0, "rng_replays": 0})
    assert int(graphed.rng_counter) == COUNTER_START + 5 and graphed.engine.calls == 12
    graphed.load_state_dict(dict(sd, rng_replays=9, rng_counter=COUNTER_START + 9))  # graphed -> graphed: forward to the checkpoint's position
    assert int(graphed.rng_counter) == COUNTER_START + 9
    graphed.load_state_dict(sd)  # ... and never backwards
    assert int(graphed.rng_counter) == COUNTER_START + 9
    graphed.load_state_dict(sd, rewind=True)  # ... unless asked to: reproducing an earlier stretch of the run in the same process
    assert int(graphed.rng_counter) == COUNTER_START + 5
    graphed.rng_counter.fill_(COUNTER_START + 5)
    eager = ExponentialIntegratorSDELoss(generative_ctrl=None, alpha=1.0, sigma=2.0, method="lv")
    eager.load_state_dict(dict(sd, rng_calls=3))  # graphed -> eager: only the replay count joins the calls, no carry into the stream id
    assert eager.engine.calls == 8 and eager.engine.offset() == 8
    eager.load_state_dict({"n_filtered": 0, "rng_calls": 3, "rng_counter": COUNTER_START + 2})  # the first format of these checkpoints
    assert eager.engine.calls == 5
    from sde_sampler_amd.eq.integrator import EulerIntegrator
    assert EulerIntegrator().engine.offset() == 1 << 40  # its own Philox stream
    rnd = torch.tensor([[1.0], [float("nan")], [2e9], [3.0]])
    val, met = loss.compute_loss(rnd)
    assert val.item() == pytest.approx(torch.tensor([1.0, 3.0]).var().item()) and met["train/n_filtered_cumulative"] == 9
    kl = TimeReversalLoss(generative_ctrl=None, method="kl")
    val, met = kl.compute_loss(rnd)
    assert val.item() == pytest.approx((1.0 + 2e9 + 3.0) / 3) and met["train/n_filtered_cumulative"] == 1
    lvt = TimeReversalLoss(generative_ctrl=None, method="lv_traj", traj_per_sample=2)
    r2 = torch.tensor([[1.0], [2.0], [3.0], [6.0]])
    val, _ = lvt.compute_loss(r2)
    assert val.item() == pytest.approx(((1 - 3) ** 2 / 2 + (2 - 6) ** 2 / 2) / 2)
    assert Results()._fields == ("samples", "weights", "log_norm_const_preds", "expectation_preds", "ts", "xs", "metrics", "plots")
    assert issubclass(TimeReversalLoss, BaseOCLoss)


def test_time_grids():
    from sde_sampler_amd.utils.common import get_timesteps

    assert torch.equal(get_timesteps(0.0, 1.0, steps=100), torch.linspace(0, 1, 101))
    ts = get_timesteps(0.0, 12.8, dt=0.05, rescale_t="cosine")
    assert ts.shape == (258,) and ts[0] == 0 and abs(ts[-1].item() - 12.8) < 1e-4 and (ts[1:] >= ts[:-1]).all()
    q = get_timesteps(0.0, torch.tensor(2.0), steps=10, rescale_t="quad")
    assert q.shape == (11,) and q[-1] == 2.0
    with pytest.raises(ValueError):
        get_timesteps(0.0, 1.0)
    with pytest.raises(ValueError):
        get_timesteps(0.0, 1.0, steps=3, rescale_t="nope")


def test_merge_stats_matches_direct_computation():
    from sde_sampler_amd import engine as E

    torch.manual_seed(0)
    rnd = torch.randn(1000, dtype=torch.float64) * 3 + 20
    chunks = [rnd[:100], rnd[100:101], rnd[101:700], rnd[700:]]

    def stats(v):
        neg = -v
        m = neg.max()
        return torch.stack([torch.tensor(float(len(v))), neg.sum(), ((v - v.mean()) ** 2).sum(), m, (neg - m).exp().sum(),
                            (2 * (neg - m)).exp().sum(), torch.tensor(0.0), torch.tensor(0.0)])

    merged = E.merge_stats(torch.stack([stats(c) for c in chunks] + [torch.tensor([0, 0, 0, -math.inf, 0, 0, 3.0, 0])]))
    est = E.estimators_from_stats(merged)
    neg = -rnd
    m = neg.max()
    w = (neg - m).exp()
    assert est["n"] == 1000 and est["n_filtered"] == 3
    assert est["mean_neg_rnd"] == pytest.approx(neg.mean().item(), abs=1e-10)
    assert est["var_rnd"] == pytest.approx(rnd.var().item(), rel=1e-10)
    assert est["log_norm_const_is"] == pytest.approx((w.mean().log() + m).item(), abs=1e-10)
    assert est["ess"] == pytest.approx((w.sum() ** 2 / (w**2).sum()).item(), rel=1e-10)


def test_euler_integrator_has_no_cpu_path():
    """EulerIntegrator mirrors the reference's constructor / integrate signature; CPU tensors are refused (no fallback)."""
    from sde_sampler_amd.eq.integrator import EulerIntegrator
    from sde_sampler_amd.eq.sdes import VP, ControlledSDE

    integ = EulerIntegrator(dt=0.1)
    assert (integ.dt, integ.steps, integ.rescale_t, integ.eps) == (0.1, None, None, 1e-8)
    sde = ControlledSDE(sde=VP(generative=False), ctrl=None)
    assert sde.terminal_t.item() == 1.0 and sde.noise_type == "diagonal"
    with pytest.raises(RuntimeError, match="no CPU path"):
        integ.integrate(sde, ts=torch.linspace(0, 1, 3), x_init=torch.zeros(4, 2))


def test_sinkhorn_argument_checks_mirror_the_reference():
    """Constructor / argument validation of eval/sinkhorn.py:34-60,71-110 (no GPU involved) and the loud CPU refusal."""
    from sde_sampler_amd.eval.sinkhorn import Sinkhorn

    with pytest.raises(TypeError):
        Sinkhorn(p=2.0)
    with pytest.raises(ValueError):
        Sinkhorn(p=0)
    with pytest.raises(ValueError):
        Sinkhorn(eps=0.0)
    with pytest.raises(TypeError):
        Sinkhorn(max_iters=0)
    with pytest.raises(TypeError):
        Sinkhorn(stop_thresh=1)
    sk = Sinkhorn()
    assert (sk.p, sk.eps, sk.max_iters, sk.stop_thresh, sk.n_max) == (2, 1e-3, 100, 1e-5, None)
    x, y = torch.zeros(4, 2), torch.zeros(5, 3)
    with pytest.raises(ValueError):
        sk.compute(x[0], y)
    with pytest.raises(ValueError):
        sk.compute(x, y)  # dimension mismatch
    with pytest.raises(ValueError):
        sk.compute(x, torch.zeros(5, 2), w_x=torch.ones(4) / 4)  # only one weight vector
    with pytest.raises(RuntimeError, match="no CPU path"):
        sk.compute(x, torch.zeros(5, 2))


def test_committed_pmc_record_belongs_to_the_headline_kernel_in_this_tree():
    """profiles/pmc_headline.json is stamped with the hash of the headline trajectory kernel's sources (bench.py:
    HEADLINE_KERNEL_SOURCES); bench.py reports `roofline.traffic` / `executed_tflops` only while the stamp matches.  A change to
    those sources needs a new PMC pass (tools/pmc_profile.sh) or -- when the kernel's code did not change -- a re-stamp
    (tools/pmc_headline_json.py on the kept summary)."""
    import bench

    rec = bench.pmc_record()
    assert rec is not None, "profiles/pmc_headline.json is stale: kernel_sha != bench.headline_kernel_sha()"
    assert rec["kernel_sha"] == bench.headline_kernel_sha() and rec["FETCH_SIZE_KiB"] > 0 and rec["SQ_INSTS_MFMA"] > 0


def test_problem_description_cache_follows_the_objects():
    """engine.build_problem caches the filled SdehProblem per (objects, flags) and revalidates it with a fingerprint of what it
    read: a replaced parameter tensor (EMA swap), a mutated clip value (MultiStepParams) and a rewritten coefficient tensor must all
    show up in the next description; in-place parameter updates keep the pointers (the kernels read the current values)."""
    from sde_sampler_amd import engine as E
    from sde_sampler_amd import problems

    prob = problems.build(problems.baseline_spec("gmm50_pis_headline"))
    eng = prob.loss.engine
    n_described = []
    describe = eng._describe
    eng._describe = lambda **kw: (n_described.append(1), describe(**kw))[1]
    kw = dict(loss_kind=prob.loss._LOSS_KIND, generative_ctrl=prob.ctrl, sde=prob.sde, flags=0, device=torch.device("cpu"),
              terminal_target=prob.target, reference_prior=prob.loss._reference_prior() if hasattr(prob.loss, "_reference_prior") else None)
    a = eng.build_problem(keep=E._Keep(), **kw)
    keep = E._Keep()
    b = eng.build_problem(keep=keep, **kw)
    assert len(n_described) == 1 and bytes(a) == bytes(b) and a is not b and len(keep) > 10  # a hit: a copy, the tensors kept alive
    b.clip_model = 123.0  # the copy is the caller's
    assert eng.build_problem(keep=E._Keep(), **kw).clip_model == a.clip_model
    w = prob.ctrl.base_model.out_layer.weight
    with torch.no_grad():
        w.mul_(1.5)  # an optimizer step: same storage
    assert eng.build_problem(keep=E._Keep(), **kw).base_model.out_w == a.base_model.out_w and len(n_described) == 1
    w.data = w.data.clone()  # an EMA swap: new storage
    c = eng.build_problem(keep=E._Keep(), **kw)
    assert len(n_described) == 2 and c.base_model.out_w == w.data_ptr() != a.base_model.out_w
    prob.ctrl.clip_model = 7.5  # MultiStepParams mutates the clip values
    assert eng.build_problem(keep=E._Keep(), **kw).clip_model == 7.5 and len(n_described) == 3
    assert eng.build_problem(keep=E._Keep(), **dict(kw, flags=4)).flags & 4 and len(n_described) == 4  # other flags: another entry
    with torch.no_grad():
        prob.target.loc[0, 0] += 1.0  # values of a table the description depends on (mixture structure): version bump
    eng.build_problem(keep=E._Keep(), **kw)
    assert len(n_described) == 5
    eng.build_problem(keep=E._Keep(), **kw)
    assert len(n_described) == 5  # a hit
    eng.invalidate()  # for what the fingerprint does not follow (scalars of sub-objects mutated in place): describe again
    eng.build_problem(keep=E._Keep(), **kw)
    assert len(n_described) == 6


def test_split_bridge_path_selection(monkeypatch):
    """Which Bridge training calls take the split (plain launch + row-parallel inference pass, fused backwards: losses/_autograd.py) --
    decided on the host from the objects alone: 64 channels, a two-hidden-layer inference network, the exact divergence; method kl also
    needs a two-hidden-layer generative network (the through-time kernels' KLB instantiations); the A/B switches turn it off."""
    from sde_sampler_amd import _lib as L
    from sde_sampler_amd import problems
    from sde_sampler_amd.losses import _autograd as A

    def bridge(num_layers_gen=4, num_layers_inf=4, channels=64):
        lerp = dict(clip_model=10.0, clip_score=10.0, scale_score=1.0, gamma_dim=1, gamma_bias=1.0)
        spec = dict(batch=64, target=dict(kind="funnel", dim=10), prior=dict(kind="iso_gauss", dim=10),
                    sde=dict(kind="scaled_bm", diff_coeff=1.0, terminal_t=1.0), ctrl=dict(kind="lerp_target", **lerp),
                    inference_ctrl=dict(kind="lerp_prior", **lerp), net=dict(channels=channels, num_layers=num_layers_gen, activation="gelu"),
                    loss=dict(kind="time_reversal", method="lv", max_rnd=1e8), grid=dict(start=0.0, end=1.0, steps=8))
        prob = problems.build(spec)
        if num_layers_inf != num_layers_gen:
            spec_i = dict(spec, net=dict(channels=channels, num_layers=num_layers_inf, activation="gelu"))
            prob.loss.inference_ctrl = problems.build(spec_i).loss.inference_ctrl
        return prob

    monkeypatch.delenv("SDEH_BRIDGE_SEQ", raising=False)
    monkeypatch.delenv("SDEH_BWD_PLANES", raising=False)
    prob = bridge()
    x = torch.zeros(64, 10)
    lv, kl = L.FLAG_CHANGE_SDE_CTRL, 0
    ok = lambda p, flags, dn=None: A._bridge_split_ok(p.loss, p.ts, x, p.loss.inference_ctrl, flags, dn)
    assert ok(prob, lv) and ok(prob, kl)
    assert not ok(prob, lv, torch.zeros(8, 64, 10))  # Hutchinson probes: the step-sequential kernel
    deep = bridge(num_layers_gen=5)                     # three hidden layers in the generative network
    assert ok(deep, lv) and not ok(deep, kl)
    assert not ok(bridge(num_layers_inf=3), lv)         # a one-hidden-layer inference network
    monkeypatch.setenv("SDEH_BRIDGE_SEQ", "1")
    assert not ok(prob, lv)
    monkeypatch.delenv("SDEH_BRIDGE_SEQ")
    monkeypatch.setenv("SDEH_BWD_PLANES", "1")
    assert not ok(prob, lv) and not ok(prob, kl)


def test_product_form_guard_of_the_matrix_pipe_mixture():
    """engine._mixture_mm_ok (SDEH_DENS_FLAG_MM_OK): product-form logits are admitted for mixtures near the origin (absolute rounding
    below 1e-5) and for well-separated ones (no two components within 12 scaled units, rounding below 5e-3), and refused where two
    nearby components sit far from the origin."""
    from sde_sampler_amd import engine, problems

    dense = problems.build(problems.baseline_spec("gmm50_dense_shared")).target
    assert engine._mixture_mm_ok(dense.loc, dense.scale)
    general = problems.build(problems.baseline_spec("gmm50_dense_general")).target  # per-component scales
    assert engine._mixture_mm_ok(general.loc, general.scale)
    gen = torch.Generator().manual_seed(0)
    near = (torch.rand((40, 50), generator=gen) - 0.5) * 0.5  # |m| ~ 1: rule (a)
    assert engine._mixture_mm_ok(near, torch.ones(40, 50))
    far_pair = (torch.rand((40, 50), generator=gen) - 0.5) * 80.0
    far_pair[1] = far_pair[0] + 0.5  # two components half a sigma apart, ~160 sigma from the origin
    assert not engine._mixture_mm_ok(far_pair, torch.ones(40, 50))
    # the flag reaches the problem description only for shared-scale mixtures of 21 .. 40 components with dense tables
    fab = problems.build(problems.baseline_spec("gmm50_pis_headline")).target
    out = engine.L.SdehDensity()
    engine._fill_density(fab, out, engine._Keep(), torch.device("cpu"), "target")
    assert out.flags & engine.L.DENS_FLAG_SHARED_SCALE and not out.flags & engine.L.DENS_FLAG_MM_OK
    out = engine.L.SdehDensity()
    engine._fill_density(dense, out, engine._Keep(), torch.device("cpu"), "target")
    assert out.flags & engine.L.DENS_FLAG_MM_OK

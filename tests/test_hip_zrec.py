"""The pre-activation record (ABI v6): sdeh_simulate_fwd_train3 keeps Z_k of every layer and the raw network output, the fused backward
(sdeh_ctrl_backward_fused_z) reads them instead of re-evaluating the network -- what the reference's autograd does between loss(...) and
loss.backward() (losses/oc.py:232-256 -> models/mlp.py:114-122).

(i) the record's CONTENT and layout in every mode of the forward kernel (groups of 64 / 32, pair, quad) against the network evaluated by
PyTorch (fp32, on the device) at the stored trajectory; (ii) gradients with the record against the re-evaluating launches (plan option
SDEH_BWD_ZREC=0) on the same Philox draws, every kernel that reads it; (iii) the reference-autograd goldens through it (their comparison
itself is tests/test_hip_parity.py: here only that the record path is what served them)."""
import os

import numpy as np
import pytest
import torch

from tests.helpers import measured

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _build(name, batch, method, steps=None, traj_per_sample=None):
    from sde_sampler_amd import problems

    spec = problems.baseline_spec(name)
    spec["batch"] = batch
    spec["loss"]["method"] = method
    if method.startswith("lv"):
        spec["loss"]["max_rnd"] = 1e8
    if steps is not None:
        spec["grid"]["steps"] = steps
    if traj_per_sample is not None:
        spec["loss"]["traj_per_sample"] = traj_per_sample
    torch.manual_seed(11)
    prob = problems.build(spec, device=DEV)
    # a control that is not the zero-initialised out layer of a fresh model: every layer's gradient is exercised
    with torch.no_grad():
        for p in prob.ctrl.base_model.out_layer.parameters():
            p.add_(0.05 * torch.randn_like(p))
    return prob, spec


def _spy(eng):
    got = {}
    orig = eng.run

    def run(*a, **k):
        out = orig(*a, **k)
        got["out"] = out
        return out

    eng.run = run
    return got


@pytest.mark.parametrize("mode", ["4", "4h", "2h", "p", "quad"])
@pytest.mark.parametrize("name", ["cfg2_gmm2_dis_kl", "cfg3_gmm50_pis_kl", "cfg4_funnel_dds_lv"])
def test_record_holds_the_networks_preactivations(name, mode):
    B = 333  # a ragged last tile; groups of 64: a second column tile beyond the batch in the last group
    prob, spec = _build(name, B, "lv", steps=6)
    T = prob.ts.numel() - 1
    eng = prob.loss.engine
    if mode == "quad":
        if spec["target"]["dim"] > 32:
            pytest.skip("quad mode: d <= 32")
        eng.options["SDEH_WS_QUAD"] = "1"
    else:
        eng.options["SDEH_WS_QUAD"] = "0"
        eng.options["SDEH_WS_GROUPS"] = mode
    got = _spy(eng)
    x0 = prob.prior.sample((B,))
    val, _ = prob.loss(prob.ts, x0, prob.target.unnorm_log_prob, prob.second_log_prob)
    kept = got["out"][3]
    assert kept is not None and kept[0] == "fused" and kept[4] is not None, "the launch kept no record"
    xs, zrec = got["out"][2], kept[4]
    net = prob.ctrl.base_model
    Lh, d = len(net.hidden_layer), spec["target"]["dim"]
    tiles, otd = (B + 31) // 32, (d + 31) // 32
    rec = zrec.view(T, tiles, (Lh + 1) * 2048 + otd * 1024)
    z = rec[..., :(Lh + 1) * 2048].reshape(T, tiles, Lh + 1, 16, 32, 4)  # [step][tile][layer][quad][trajectory][4 channels]
    z = z.permute(0, 2, 3, 5, 1, 4).reshape(T, Lh + 1, 64, tiles * 32)[..., :B]  # [step][layer][channel][row]
    nn = rec[..., (Lh + 1) * 2048:].reshape(T, tiles, otd * 8, 32, 4)  # [step][tile][coordinate quad][trajectory][4 coordinates]
    nn = nn.permute(0, 2, 4, 1, 3).reshape(T, otd * 32, tiles * 32)[:, :d, :B]  # [step][coordinate][row]
    worst_z = worst_n = 0.0
    with torch.no_grad():
        for t in range(T):
            x = xs[t].t().contiguous()
            h = net.input_embed(x) + net.timestep_embed(prob.ts[t].reshape(1, 1).float())
            hs = [h]
            for layer in net.hidden_layer:
                h = layer(net.activation(h))
                hs.append(h)
            out = net.out_layer(net.activation(h))
            for k, hk in enumerate(hs):
                scale = float(hk.abs().max()) + 1e-6
                worst_z = max(worst_z, float((z[t, k].t() - hk).abs().max()) / scale)
            worst_n = max(worst_n, float((nn[t].t() - out).abs().max()) / (float(out.abs().max()) + 1e-6))
    measured(f"zrec_content/{name}/{mode}/Z", worst_z, 2e-5)
    measured(f"zrec_content/{name}/{mode}/nn", worst_n, 2e-5)
    assert worst_z <= 2e-5, f"pre-activations differ from the network at the stored trajectory: {worst_z:.2e} of scale"
    assert worst_n <= 2e-5, f"raw network output: {worst_n:.2e}"


def _grads(prob, x0, zrec: bool, extra: dict | None = None):
    eng = prob.loss.engine
    eng.options["SDEH_BWD_ZREC"] = None if zrec else "0"
    for k, v in (extra or {}).items():
        eng.options[k] = v
    calls = eng.calls
    prob.ctrl.zero_grad()
    val, _ = prob.loss(prob.ts, x0, prob.target.unnorm_log_prob, prob.second_log_prob)
    val.backward()
    name = eng.last_kernel_name()
    eng.calls = calls  # the same Philox offset for the next run
    return float(val.detach()), {k: p.grad.detach().clone() for k, p in prob.ctrl.named_parameters() if p.grad is not None}, name


CASES = [  # (spec, method, batch, steps, what the record launch is called, plan options)
    ("cfg2_gmm2_dis_kl", "lv", 777, 12, "bwd_fused<rows,tiles=1,traj-split,zrec>", {}),
    ("cfg2_gmm2_dis_kl", "kl", 16 * 1024 + 100, 6, "bwd_fused<bptt,tiles=1,traj-split,zrec>", {}),
    ("cfg1_dw_dis_lv", "lv", 4097, 8, "bwd_fused<rows,tiles=1,traj-split,zrec>", {}),
    ("cfg1_dw_dis_lv", "kl_ito", 16 * 1024 + 33, 6, "bwd_fused<bptt,tiles=1,traj-split,zrec>", {}),
    ("cfg4_funnel_dds_lv", "lv", 1000, 9, "bwd_fused<rows,tiles=1,traj-split,zrec>", {}),
    ("cfg4_funnel_dds_lv", "kl", 16 * 1024 + 64, 5, "bwd_fused<bptt,tiles=1,traj-split,zrec>", {}),
    ("cfg3_gmm50_pis_kl", "lv", 2048, 7, "bwd_fused<rows,tiles=2,traj-split,zrec>", {}),
    ("cfg3_gmm50_pis_kl", "lv_traj", 515, 7, "bwd_fused<rows,tiles=2,traj-split,zrec>", {}),
    # the channel-split kernel (sdeh_bwdf.hip): two coordinate tiles through time, forced tilings, three hidden layers
    ("cfg3_gmm50_pis_kl", "kl", 16 * 1024 + 40, 5, "bwd_fused<bptt,tiles=2,traj-split,zrec>", {}),  # (with the record: trajectory-split)
    ("cfg3_gmm50_pis_kl", "kl", 16 * 1024 + 40, 5, "bwd_fused<bptt,tiles=2,chan-split,zrec>", {"SDEH_BWD_V1": "1"}),
    ("cfg3_gmm50_pis_kl", "kl_ito", 515, 7, "bwd_fused<bptt,tiles=2,chan-split,zrec>", {"SDEH_BWD_TILE": "32"}),
    ("cfg4_funnel_dds_lv", "kl", 1000, 6, "bwd_fused<bptt,tiles=1,chan-split,zrec>", {"SDEH_BWD_TILE": "32"}),
    ("cfg2_gmm2_dis_kl", "lv", 900, 6, "bwd_fused<rows,tiles=1,chan-split,zrec>", {"SDEH_BWD_V1": "1"}),
    ("cfg3_gmm50_pis_kl", "kl", 16 * 1024 + 40, 5, "bwd_fused<bptt,tiles=2,traj-split,zrec>", {"SDEH_BWD_V2": "1"}),
    ("cfg3_gmm50_pis_kl", "kl_ito", 700, 6, "bwd_fused<bptt,tiles=2,traj-split,zrec>", {"SDEH_BWD_V2": "1", "SDEH_BWD_TILE": "32"}),
    ("cfg1_dw_dis_lv", "kl", 700, 6, "bwd_fused<bptt,tiles=1,chan-split,zrec>", {"SDEH_BWD_V1": "1", "SDEH_BWD_TILE": "32"}),
    # small batches through time: teams of four waves on tiles of 16 (sdeh_bwdf16.hip; the forward ran in pair / quad mode) and the scan form
    ("cfg3_gmm50_pis_kl", "kl", 2048, 6, "bwd_fused16<bptt,tiles=2,zrec>", {}),
    ("cfg3_gmm50_pis_kl", "kl_ito", 1000, 6, "bwd_fused16<bptt,tiles=2,zrec>", {}),
    ("cfg4_funnel_dds_lv", "kl", 2048, 6, "bwd_fused16<bptt,tiles=1,zrec>", {}),
    ("cfg4_funnel_dds_lv", "kl", 9000, 5, "bwd_fused16<bptt,tiles=1,zrec>", {}),
    ("cfg2_gmm2_dis_kl", "kl", 5000, 6, "bwd_fused16<bptt,tiles=1,zrec>", {}),
    ("cfg2_gmm2_dis_kl", "kl", 2048, 8, "bwd_fused<bptt-scan,tiles=1,traj-split,zrec>", {}),
    ("cfg1_dw_dis_lv", "kl_ito", 515, 8, "bwd_fused<bptt-scan,tiles=1,traj-split,zrec>", {}),
]


@pytest.mark.parametrize("name,method,batch,steps,kernel,opts", CASES, ids=[f"{c[0]}-{c[1]}-{c[2]}" for c in CASES])
def test_gradients_with_the_record_equal_the_reevaluating_launch(name, method, batch, steps, kernel, opts):
    prob, spec = _build(name, batch, method, steps=steps, traj_per_sample=2 if method == "lv_traj" else None)
    x0 = prob.prior.sample((batch,))
    v1, g1, n1 = _grads(prob, x0, True, opts)
    v0, g0, n0 = _grads(prob, x0, False, opts)
    assert n1 == kernel, n1
    assert "zrec" not in n0, n0
    assert v1 == v0  # the forward launches differ in what they store only
    gmax = max(float(g.abs().max()) for g in g0.values())
    worst = 0.0
    for k in g0:
        denom = max(float(g0[k].abs().max()), 1e-3 * gmax, 1e-12)
        worst = max(worst, float((g1[k] - g0[k]).abs().max()) / denom)
    measured(f"zrec_vs_reevaluation/{name}/{method}/B{batch}", worst, 5e-6)
    # (groups of 64 / 32 re-evaluate bit for bit: row-parallel launches agree exactly, through time to the last bits of another
    # instruction order in the elementwise phase; the quad mode's pre-activations differ in rounding from their re-evaluation)
    assert worst <= (2e-5 if batch <= 16384 and spec["target"]["dim"] <= 32 else 5e-6), f"{worst:.2e}"  # (quad-mode forward: see above)


@pytest.mark.parametrize("layers,method", [(3, "lv"), (3, "kl"), (5, "lv"), (5, "kl")])
def test_other_depths_through_the_record(layers, method):
    """One hidden layer (trajectory-split teams) and three (channel-split teams: the only kernel compiled for them)."""
    from sde_sampler_amd import problems

    spec = problems.baseline_spec("cfg4_funnel_dds_lv")
    B = 16 * 1024 + 70 if method == "kl" else 600
    spec["batch"], spec["grid"]["steps"] = B, 5
    spec["net"]["num_layers"] = layers
    spec["loss"]["method"], spec["loss"]["max_rnd"] = method, (1e8 if method == "lv" else None)
    torch.manual_seed(7)
    prob = problems.build(spec, device=DEV)
    with torch.no_grad():
        for p in prob.ctrl.base_model.out_layer.parameters():
            p.add_(0.05 * torch.randn_like(p))
    x0 = prob.prior.sample((B,))
    v1, g1, n1 = _grads(prob, x0, True)
    v0, g0, n0 = _grads(prob, x0, False)
    assert n1.endswith(",zrec>") and "zrec" not in n0, (n1, n0)
    assert ("chan-split" in n1) == (layers == 5), n1
    assert v1 == v0
    gmax = max(float(g.abs().max()) for g in g0.values())
    for k in g0:
        err = float((g1[k] - g0[k]).abs().max()) / max(float(g0[k].abs().max()), 1e-3 * gmax, 1e-12)
        assert err <= 5e-6, f"{k}: {err:.2e}"


def test_relu_network_through_the_record():
    """A ReLU unit on its kink takes the side the FORWARD launch took: the record is the forward launch's pre-activation."""
    from sde_sampler_amd import problems

    spec = problems.baseline_spec("cfg4_funnel_dds_lv")
    spec["batch"], spec["grid"]["steps"] = 640, 8
    spec["net"]["activation"] = "relu"
    spec["loss"]["method"], spec["loss"]["max_rnd"] = "lv", 1e8
    torch.manual_seed(5)
    prob = problems.build(spec, device=DEV)
    with torch.no_grad():
        for p in prob.ctrl.base_model.out_layer.parameters():
            p.add_(0.05 * torch.randn_like(p))
    x0 = prob.prior.sample((640,))
    v1, g1, n1 = _grads(prob, x0, True)
    v0, g0, n0 = _grads(prob, x0, False)
    assert n1.endswith("zrec>") and v1 == v0
    for k in g0:
        assert torch.equal(g1[k], g0[k]), k  # (pair-mode / 32-trajectory forward: re-evaluated bit for bit as well)


def test_record_respects_its_memory_budget(monkeypatch):
    """SDEH_ZREC_BYTES caps the record: beyond it the launch keeps the planes of sdeh_simulate_fwd_train2 and the backward re-evaluates."""
    prob, spec = _build("cfg2_gmm2_dis_kl", 1024, "lv", steps=5)
    x0 = prob.prior.sample((1024,))
    monkeypatch.setenv("SDEH_ZREC_BYTES", "1000")
    v, g, n = _grads(prob, x0, True)
    assert n == "bwd_fused<rows,tiles=1,traj-split>", n
    monkeypatch.delenv("SDEH_ZREC_BYTES")
    v1, g1, n1 = _grads(prob, x0, True)
    assert n1.endswith(",zrec>") and v1 == v


RAGGED = [  # (spec, method, batch, steps, forward mode options): every forward mode in which the VECTOR wave writes the raw network output
    ("cfg2_gmm2_dis_kl", "lv", 777, 6, {}),                                           # d <= 4, groups of 32: out layer on the vector pipe
    ("cfg2_gmm2_dis_kl", "kl", 515, 6, {}),                                           # scan form (Jacobian pass reads the record)
    ("cfg2_gmm2_dis_kl", "kl", 5003, 5, {}),                                          # teams of 16 through time
    ("cfg1_dw_dis_lv", "kl_ito", 4097 + 13, 5, {}),                                   # d = 1
    ("cfg4_funnel_dds_lv", "lv", 1001, 6, {"SDEH_WS_QUAD": "0", "SDEH_WS_GROUPS": "p"}),   # pair mode: the partial sums meet in the V wave
    ("cfg4_funnel_dds_lv", "kl", 2045, 5, {"SDEH_WS_QUAD": "1"}),                     # quad mode
    ("cfg4_funnel_dds_lv", "kl", 16 * 1024 + 70, 4, {}),                              # groups of 64 (the M wave stores whole tiles)
    ("cfg3_gmm50_pis_kl", "kl", 1003, 4, {}),                                         # d > 32: M waves store the output tiles
]


@pytest.mark.parametrize("name,method,batch,steps,opts", RAGGED, ids=[f"{c[0]}-{c[1]}-{c[2]}" for c in RAGGED])
def test_ragged_last_tile_of_an_unclipped_control_on_a_poisoned_record(name, method, batch, steps, opts, monkeypatch):
    """the record is uninitialised memory; rows of the last 32-row tile beyond the batch used to keep whatever it
    held, and with `clip_model = None` (the clamp is +-inf) a NaN there reached the d gamma / clip partial sums as 0 x NaN.  The forward
    launch now writes zeros for those rows: a record PREFILLED WITH NaN gives the gradients of the re-evaluating launch."""
    from sde_sampler_amd import engine as E

    assert batch % 32 != 0
    prob, spec = _build(name, batch, method, steps=steps)
    prob.ctrl.clip_model = None  # no clamp on the network output: clipf(NaN, inf) stays NaN
    prob.loss.engine.invalidate()
    x0 = prob.prior.sample((batch,))
    seen = {}

    def poisoned(n_floats, device):
        seen["n"] = n_floats
        return torch.full((n_floats,), float("nan"), device=device, dtype=torch.float32)

    monkeypatch.setattr(E, "_alloc_zrec", poisoned)
    v1, g1, n1 = _grads(prob, x0, True, opts)
    monkeypatch.undo()
    v0, g0, n0 = _grads(prob, x0, False, opts)
    assert "n" in seen and n1.endswith("zrec>") and "zrec" not in n0, (n1, n0)
    assert v1 == v0 and np.isfinite(v1)
    gmax = max(float(g.abs().max()) for g in g0.values())
    for k in g0:
        assert torch.isfinite(g1[k]).all(), f"{k}: non-finite gradient from the record's rows beyond the batch"
        err = float((g1[k] - g0[k]).abs().max()) / max(float(g0[k].abs().max()), 1e-3 * gmax, 1e-12)
        assert err <= 2e-5, f"{k}: {err:.2e}"


def test_record_budget_is_decided_once_per_size(monkeypatch):
    """the 40 %-of-free-memory decision is cached per (device, record size) -- no driver query on later steps, the same
    launches every step; a failed allocation withdraws it."""
    from sde_sampler_amd import engine as E

    monkeypatch.delenv("SDEH_ZREC_BYTES", raising=False)
    E.reset_zrec_budget()
    calls = {"n": 0}
    real = torch.cuda.mem_get_info

    def counting(*a, **k):
        calls["n"] += 1
        return real(*a, **k)

    monkeypatch.setattr(torch.cuda, "mem_get_info", counting)
    prob, spec = _build("cfg2_gmm2_dis_kl", 1024, "lv", steps=5)
    x0 = prob.prior.sample((1024,))
    names = [_grads(prob, x0, True)[2] for _ in range(3)]
    assert calls["n"] == 1 and all(n.endswith(",zrec>") for n in names), (calls, names)
    key = next(iter(E._ZREC_DECISION))
    monkeypatch.setattr(E.torch, "empty", lambda *a, **k: (_ for _ in ()).throw(torch.OutOfMemoryError("simulated")))
    assert E._alloc_zrec(key[1] // 4, DEV) is None
    monkeypatch.undo()
    assert E._ZREC_DECISION[key] is False
    assert "zrec" not in _grads(prob, x0, True)[2]
    E.reset_zrec_budget()
    assert _grads(prob, x0, True)[2].endswith(",zrec>")

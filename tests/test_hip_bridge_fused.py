"""The fused backward of a 64-channel Bridge's inference network (csrc/sdeh_bridgef.hip + the Bridge form of csrc/sdeh_bwdf2.hip's
row-parallel kernel; reference losses/oc.py:189-202, utils/autograd.py:14-21) against the plane-writing kernels it replaces
(sdeh_ctrl_backward_ex + sdeh_bridge_div_backward + sdeh_weight_grad, pinned to the reference's autograd by tests/test_hip_bridge.py):
every parameter gradient of both networks on the same Philox draws, at shapes the goldens do not reach (ragged batches over several
teams, d in every tile class, active clamps)."""
import pytest
import torch

from tests.helpers import measured

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
NET = dict(channels=64, num_layers=4, activation="gelu")

CASES = [
    # name, target, d, batch, steps, inference kind, clip_model
    ("gmm_d2", dict(kind="gmm", dim=2, name="fab"), 2, 200, 12, "lerp_prior", 10.0),
    ("dw_d1_clipped", dict(kind="double_well", dim=1, separation=2.0, shift=1.0), 1, 77, 9, "clipped", 10.0),
    ("mw_d5_clamps", dict(kind="multi_well", dim=5, n_double_wells=5, separation=2.0, shift=0.0), 5, 333, 7, "lerp_prior", 0.05),
    ("funnel_d10", dict(kind="funnel", dim=10), 10, 1100, 6, "lerp_prior", 10.0),
    ("gauss_d20", dict(kind="iso_gauss", dim=20, loc=1.0, scale=0.5), 20, 96, 5, "lerp_prior", 10.0),
    ("gauss_d50", dict(kind="iso_gauss", dim=50, loc=1.0, scale=0.5), 50, 150, 4, "lerp_prior", 10.0),
    # other activations, an SDE with sigma(t) and a drift, active clamps of the inference control's score term
    ("gauss_d6_silu", dict(kind="iso_gauss", dim=6, loc=1.0, scale=0.5), 6, 130, 5, "lerp_prior", 10.0, dict(activation="silu")),
    ("gauss_d6_relu", dict(kind="iso_gauss", dim=6, loc=1.0, scale=0.5), 6, 130, 5, "lerp_prior", 10.0, dict(activation="relu")),
    ("gauss_d9_vp", dict(kind="iso_gauss", dim=9, loc=1.0, scale=0.5), 9, 100, 6, "lerp_prior", 10.0,
     dict(sde=dict(kind="vp", beta_min=0.1, beta_max=6.0, scale=1.0, terminal_t=1.0, generative=True))),
    ("gauss_d7_score_clamps", dict(kind="iso_gauss", dim=7, loc=1.0, scale=0.5), 7, 90, 5, "lerp_prior", 10.0, dict(inf_clip_score=0.4)),
]


def _grads(spec, x0_seed, mode, monkeypatch):
    """mode: "split" (the default path: plain forward + row-parallel inference pass, fused backwards), "seq" (the step-sequential Bridge
    forward, fused inference backward), "planes" (the plane-writing kernels throughout)."""
    from sde_sampler_amd import problems

    monkeypatch.delenv("SDEH_BWD_PLANES", raising=False)
    monkeypatch.delenv("SDEH_BRIDGE_SEQ", raising=False)
    if mode == "planes":
        monkeypatch.setenv("SDEH_BWD_PLANES", "1")
    elif mode == "seq":
        monkeypatch.setenv("SDEH_BRIDGE_SEQ", "1")
    torch.manual_seed(11)
    prob = problems.build(spec, device=DEV)
    torch.manual_seed(x0_seed)
    x0 = prob.prior.sample((spec["batch"],))
    params = list(prob.ctrl.named_parameters()) + [("inf." + k, p) for k, p in prob.loss.inference_ctrl.named_parameters()]
    prob.loss.engine.calls = 3
    val, _ = prob.loss(prob.ts, x0, prob.target.unnorm_log_prob, prob.second_log_prob)
    val.backward()
    return val.item(), {k: (None if p.grad is None else p.grad.clone()) for k, p in params}, prob.loss.engine.last_kernel_name()


@pytest.mark.parametrize("method", ["lv", "kl"])
@pytest.mark.parametrize("case", CASES, ids=lambda c: c[0])
def test_fused_bridge_backward_equals_the_plane_kernels(case, method, monkeypatch):
    """lv: row-parallel everywhere.  kl: the generative network's back-propagation through time takes its running cost on u + v and the
    inference terms' d loss / d x_t from the row-parallel Bridge kernel (sdeh_ctrl_backward_fused_ex)."""
    name, tspec, d, B, T, ikind, clip = case[:7]
    over = case[7] if len(case) > 7 else {}
    ictrl = dict(kind=ikind, clip_model=clip)
    if ikind == "lerp_prior":
        ictrl.update(clip_score=over.get("inf_clip_score", 10.0), scale_score=1.0, gamma_dim=(d if name == "mw_d5_clamps" else 1), gamma_bias=1.0)
    spec = dict(batch=B, target=tspec, prior=dict(kind="iso_gauss", dim=d),
                sde=over.get("sde", dict(kind="scaled_bm", diff_coeff=1.0, terminal_t=1.0)),
                ctrl=dict(kind="lerp_target", clip_model=10.0, clip_score=10.0, scale_score=1.0, gamma_dim=1, gamma_bias=1.0),
                inference_ctrl=ictrl, net=dict(NET, activation=over.get("activation", "gelu")), loss=dict(kind="time_reversal", method=method, max_rnd=1e8 if method == "lv" else None),
                grid=dict(start=0.0, end=1.0, steps=T))
    bar = 5e-5  # measured: <= 1.9e-5 (lv), <= 1.1e-5 (kl)
    v_p, g_p, _ = _grads(spec, 5, "planes", monkeypatch)
    for mode in (("split", "seq") if method == "lv" else ("split",)):
        v_f, g_f, kern = _grads(spec, 5, mode, monkeypatch)
        if mode == "seq":
            assert v_f == v_p  # the same forward launch
        else:  # rnd = plain launch + row-parallel sums: another summation order
            assert abs(v_f - v_p) <= 2e-5 * max(1.0, abs(v_p)), (v_f, v_p)
            measured(f"bridge_split_loss/{method}/{name}", abs(v_f - v_p) / max(1.0, abs(v_p)), 2e-5)
        # (split: the generative network's fused backward is the last launch; seq: the inference network's)
        assert kern.startswith("bwd_fused" if mode == "split" else "bridge_bwd_fused"), kern
        worst = 0.0
        for k in g_p:
            assert (g_f[k] is None) == (g_p[k] is None), (mode, k)
            if g_p[k] is None:
                continue
            scale = g_p[k].abs().max().item()
            if scale == 0.0:
                assert g_f[k].abs().max().item() <= 1e-7, (mode, k)
                continue
            err = (g_f[k] - g_p[k]).abs().max().item() / scale
            worst = max(worst, err)
            assert err <= bar, (mode, k, err)
        measured(f"bridge_{mode}_vs_planes/{method}/{name}", worst, bar)


@pytest.mark.parametrize("case", [CASES[0], CASES[3], CASES[5]], ids=lambda c: c[0])
def test_split_bridge_evaluation_equals_the_step_sequential_kernel(case, monkeypatch):
    """Without a graph (loss.eval, torch.no_grad()): the plain launch + the row-parallel inference pass against csrc/sdeh_bridge.hpp's kernel
    on the same Philox draws -- samples and trajectory bit for bit (the same generative launch arithmetic), rnd to fp32 rounding of
    another summation order."""
    from sde_sampler_amd import problems

    name, tspec, d, B, T, ikind, clip = case[:7]
    ictrl = dict(kind=ikind, clip_model=clip, clip_score=10.0, scale_score=1.0, gamma_dim=1, gamma_bias=1.0)
    spec = dict(batch=B, target=tspec, prior=dict(kind="iso_gauss", dim=d), sde=dict(kind="scaled_bm", diff_coeff=1.0, terminal_t=1.0),
                ctrl=dict(kind="lerp_target", clip_model=10.0, clip_score=10.0, scale_score=1.0, gamma_dim=1, gamma_bias=1.0),
                inference_ctrl=ictrl, net=NET, loss=dict(kind="time_reversal", method="lv", max_rnd=1e8),
                grid=dict(start=0.0, end=1.0, steps=T))
    torch.manual_seed(11)
    prob = problems.build(spec, device=DEV)
    x0 = prob.prior.sample((B,))
    out = {}
    for mode in ("split", "seq"):
        if mode == "seq":
            monkeypatch.setenv("SDEH_BRIDGE_SEQ", "1")
        else:
            monkeypatch.delenv("SDEH_BRIDGE_SEQ", raising=False)
        prob.loss.engine.calls = 9
        with torch.no_grad():
            x_T, rnd, xs = prob.loss.simulate(prob.ts, x0, prob.target.unnorm_log_prob, prob.second_log_prob, train=False,
                                              compute_ito_int=True, return_traj=True)
        out[mode] = (x_T.clone(), rnd.clone(), xs.clone(), prob.loss.engine.last_kernel_name())
    assert out["split"][3].startswith("bridge_rows_fwd") and out["seq"][3].startswith("bridge<"), (out["split"][3], out["seq"][3])
    diff_x = (out["split"][0] - out["seq"][0]).abs().max().item()
    assert diff_x <= 1e-4, diff_x  # (two kernels, two summation orders inside the network)
    assert (out["split"][2] - out["seq"][2]).abs().max().item() <= 1e-4
    scale = out["seq"][1].abs().clamp(min=1.0)
    err = ((out["split"][1] - out["seq"][1]).abs() / scale).max().item()
    measured(f"bridge_split_eval_rnd/{name}", err, 1e-4)
    assert err <= 1e-4, err


@pytest.mark.parametrize("opt", [None, "SDEH_BWD_V1", "SDEH_BWD_V2"])
def test_split_bridge_kl_at_a_batch_of_whole_tile_teams(opt, monkeypatch):
    """Method kl above 16 384 trajectories: the generative network's back-propagation through time runs on tiles of 32 -- trajectory-split
    teams (csrc/sdeh_bwdf2.hip, from 513 tiles on) or channel-split ones (csrc/sdeh_bwdf.hip; plan option) -- with the running cost on u + v and the
    inference terms' d loss / d x_t (ragged last tile: lanes beyond the batch must add nothing)."""
    d, B, T = 10, 16400 + 7, 3
    spec = dict(batch=B, target=dict(kind="funnel", dim=d), prior=dict(kind="iso_gauss", dim=d),
                sde=dict(kind="scaled_bm", diff_coeff=1.0, terminal_t=1.0),
                ctrl=dict(kind="lerp_target", clip_model=10.0, clip_score=10.0, scale_score=1.0, gamma_dim=1, gamma_bias=1.0),
                inference_ctrl=dict(kind="lerp_prior", clip_model=10.0, clip_score=10.0, scale_score=1.0, gamma_dim=1, gamma_bias=1.0),
                net=NET, loss=dict(kind="time_reversal", method="kl", max_rnd=None), grid=dict(start=0.0, end=1.0, steps=T))
    v_p, g_p, _ = _grads(spec, 5, "planes", monkeypatch)
    if opt is not None:
        monkeypatch.setenv(opt, "1")
    v_f, g_f, kern = _grads(spec, 5, "split", monkeypatch)
    assert kern == ("bwd_fused<bptt,tiles=1,chan-split>" if opt == "SDEH_BWD_V1" else "bwd_fused<bptt,tiles=1,traj-split>"), kern
    assert abs(v_f - v_p) <= 2e-5 * max(1.0, abs(v_p)), (v_f, v_p)
    worst = 0.0
    for k in g_p:
        if g_p[k] is None:
            continue
        scale = g_p[k].abs().max().item()
        err = (g_f[k] - g_p[k]).abs().max().item() / max(scale, 1e-30)
        worst = max(worst, err)
        assert err <= 5e-5, (k, err)
    measured(f"bridge_split_vs_planes/kl/tiles32/{opt}", worst, 5e-5)

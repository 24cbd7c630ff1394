"""Debug aid: the inference terms' d loss / d x_t plane of the split Bridge (method kl) against the plane kernels'."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from sde_sampler_amd import problems  # noqa: E402
from sde_sampler_amd.losses import _autograd as A  # noqa: E402

d, B, T = int(sys.argv[1]), int(sys.argv[2]), 6
tspec = dict(kind="funnel", dim=d) if sys.argv[3] == "funnel" else dict(kind="iso_gauss", dim=d, loc=1.0, scale=0.5)
spec = dict(batch=B, target=tspec, prior=dict(kind="iso_gauss", dim=d), sde=dict(kind="scaled_bm", diff_coeff=1.0, terminal_t=1.0),
            ctrl=dict(kind="lerp_target", clip_model=10.0, clip_score=10.0, scale_score=1.0, gamma_dim=1, gamma_bias=1.0),
            inference_ctrl=dict(kind="lerp_prior", clip_model=10.0, clip_score=10.0, scale_score=1.0, gamma_dim=1, gamma_bias=1.0),
            net=dict(channels=64, num_layers=4, activation="gelu"), loss=dict(kind="time_reversal", method="kl", max_rnd=None),
            grid=dict(start=0.0, end=1.0, steps=T))
rec = {}
orig_cb, orig_fb = A._ctrl_backward, A._fused_backward


def cb(engine, pr, keep, ts, xs, w, st, **kw):
    if kw.get("lam_extra") is not None:
        rec["old_dx"] = kw["lam_extra"].clone()      # [T, B, d]
        rec["old_cost"] = kw["cost_ctrl"].clone()
    return orig_cb(engine, pr, keep, ts, xs, w, st, **kw)


def fb(loss, pr, keep, ts, xs, w, st, sc, tscore, cost_ctrl=None, lam_extra=None):
    if lam_extra is not None:
        rec["new_dx"] = lam_extra.clone()            # [T, d, B]
        rec["new_cost"] = cost_ctrl.clone()
    return orig_fb(loss, pr, keep, ts, xs, w, st, sc, tscore, cost_ctrl=cost_ctrl, lam_extra=lam_extra)


A._ctrl_backward, A._fused_backward = cb, fb
grads = {}
for mode in ("planes", "split"):
    os.environ.pop("SDEH_BWD_PLANES", None)
    if mode == "planes":
        os.environ["SDEH_BWD_PLANES"] = "1"
    torch.manual_seed(11)
    prob = problems.build(spec, device="cuda:0")
    torch.manual_seed(5)
    x0 = prob.prior.sample((B,))
    prob.loss.engine.calls = 3
    val, _ = prob.loss(prob.ts, x0, prob.target.unnorm_log_prob, prob.second_log_prob)
    val.backward()
    grads[mode] = {k: p.grad.clone() for k, p in list(prob.ctrl.named_parameters()) + [("inf." + k, p) for k, p in prob.loss.inference_ctrl.named_parameters()] if p.grad is not None}
a, b = rec["old_dx"], rec["new_dx"].permute(0, 2, 1)
print("dx: max |old|", a.abs().max().item(), "max diff", (a - b).abs().max().item())
e = (a - b).abs()
idx = (e == e.max()).nonzero()[0].tolist()
print("   worst at (t, row, coord)", idx, a[tuple(idx)].item(), b[tuple(idx)].item())
print("   per-coordinate max diff", e.amax(dim=(0, 1)).tolist())
a, b = rec["old_cost"], rec["new_cost"].permute(0, 2, 1)
print("cost: max diff", (a - b).abs().max().item())
for k in grads["planes"]:
    g0, g1 = grads["planes"][k], grads["split"][k]
    print(f"{k:45s} {((g0 - g1).abs().max() / g0.abs().max().clamp(min=1e-12)).item():.3e}")

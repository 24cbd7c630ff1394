"""Debug aid: the planes of the scan form of BPTT (nn, J) against torch autograd of the network at the stored states."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from sde_sampler_amd import problems
from sde_sampler_amd.losses import _autograd as AG
DEV = "cuda:0"
spec = problems.baseline_spec("cfg3_gmm50_pis_kl"); spec["batch"] = 100; spec["grid"]["steps"] = 7
spec["target"] = dict(kind="iso_gauss", dim=2, loc=0.5, scale=1.2); spec["prior"]["dim"] = 2
prob = problems.build(spec, device=DEV)
torch.manual_seed(0)
x0 = prob.prior.sample((100,))
stash = {}
orig_empty = torch.empty
def spy(*a, **k):
    t = orig_empty(*a, **k)
    if len(a) == 1 and isinstance(a[0], int) and a[0] > 10000 and a[0] > stash.get("n", 0): stash["scratch"], stash["n"] = t, a[0]
    return t
os.environ["SDEH_BWD_SCAN"] = "1"
val, _ = prob.loss(prob.ts, x0, prob.target.unnorm_log_prob, prob.second_log_prob)
AG.torch.empty = spy
# capture xs from the autograd graph: run backward
val.backward()
AG.torch.empty = orig_empty
torch.cuda.synchronize()
print(prob.loss.engine.last_kernel_name(), "scratch", stash["scratch"].numel())
g_scan = {k: p.grad.clone() for k, p in prob.ctrl.named_parameters()}
T, B, d = 7, 100, 2
ws_ = 64 * 32 + 2 * 4096 + 32 * 64 + 2 * 64 + 32
tiles = (B + 31) // 32; slots = 2 * min((tiles * T + 1) // 2, 256)
off = slots * ws_ + tiles * T * 64 + tiles * T * 2 + ((slots + 31) // 32) * ws_ + ((tiles + 31) // 32) * T * 66
print("offset", off, "of", stash["scratch"].numel())
planes = stash["scratch"][off:off + T * B * (d * d + 2 * d)]
nn = planes[:T * B * d].view(T, d, B); jac = planes[T * B * d:T * B * d * (1 + d)].view(T, d, d, B); gq = planes[T * B * d * (1 + d):].view(T, d, B)
print("nn range", nn.abs().max().item(), "jac", jac.abs().max().item(), "gq", gq.abs().max().item())
# reference: network at stored states -- re-simulate to get xs
with torch.no_grad():
    prob.loss.engine.calls -= 1
    xT, rnd, xs = prob.loss.simulate(prob.ts, x0, prob.target.unnorm_log_prob, prob.second_log_prob, return_traj=True)
base = prob.ctrl.base_model
for t in (0, 3, 6):
    x = xs[t].clone().requires_grad_(True)
    out = base(prob.ts[t], x)
    print(t, "nn err", (out.detach().t() - nn[t]).abs().max().item())
    for k in range(d):
        (gx,) = torch.autograd.grad(out[:, k].sum(), x, retain_graph=True)
        print("   J row", k, "err", (gx.t() - jac[t, k]).abs().max().item(), "scale", gx.abs().max().item())
for p in prob.ctrl.parameters(): p.grad = None
os.environ["SDEH_BWD_SCAN"] = "0"
prob.loss.engine.calls -= 1
val2, _ = prob.loss(prob.ts, x0, prob.target.unnorm_log_prob, prob.second_log_prob)
val2.backward()
print(val.item(), val2.item(), prob.loss.engine.last_kernel_name())
for k, p in prob.ctrl.named_parameters():
    print(f"{k:50s} {(g_scan[k] - p.grad).abs().max().item():.3e} / {p.grad.abs().max().item():.3e}")
print("sum gq per coordinate", gq.sum(dim=(0, 2)).tolist(), " scan out bias", g_scan["base_model.out_layer.bias"].tolist(), " ref", prob.ctrl.base_model.out_layer.bias.grad.tolist())
print("gq per step (coordinate 0)", gq[:, 0].sum(dim=1).tolist())
# python replica of the scan with the planes (ScoreCtrl, PIS, no clamps): G_t = c_u lam + gc

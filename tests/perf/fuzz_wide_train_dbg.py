"""Per-parameter gradient errors of one random WIDE training case of tests/test_hip_wide_train.py (python fuzz_wide_train_dbg.py CASE...;
CASE as printed in the failure message, e.g. 11132)."""
import os, sys, math
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "tests"))
import numpy as np, torch
import test_hip_fuzz as F
from test_hip_wide_train import _widen_train
from oracle import em_oracle as eo
from sde_sampler_amd import problems
for case in [int(a) for a in sys.argv[1:]]:
    rng = np.random.default_rng(5000 + case)
    spec = F.random_spec(rng)
    method = str(rng.choice(["kl", "kl_ito", "lv"]))
    spec["loss"]["method"] = method; spec["loss"]["max_rnd"] = 1e8 if method == "lv" else None
    spec["batch"] = int(rng.choice([33, 64, 100]))
    _widen_train(spec, rng)
    print(case, method, {k: spec[k] for k in ("loss", "ctrl", "sde", "target", "prior", "net", "grid", "batch")})
    prob = problems.build(spec)
    params = {k: v.detach().clone().requires_grad_(v.is_floating_point()) for k, v in prob.ctrl.state_dict().items()}
    tt = None
    if spec["target"]["kind"] == "gmm":
        tt = dict(loc=prob.target.loc.clone(), scale=prob.target.scale.clone(), mixture_weights=prob.target.mixture_weights.clone())
    ts = prob.ts.clone(); B, d, T = spec["batch"], spec["target"]["dim"], ts.numel() - 1
    torch.manual_seed(case); x0 = prob.prior.sample((B,)); noise = torch.randn(T, B, d)
    ref_loss, _, _, _ = eo.Problem(spec, params, tt).train_loss(ts, x0.clone(), noise, method=method); ref_loss.backward()
    # float64 oracle for orientation
    p64 = {k: v.detach().double().clone().requires_grad_(v.is_floating_point()) for k, v in params.items()}
    tt64 = None if tt is None else {k: v.double() for k, v in tt.items()}
    try:
        l64, _, _, _ = eo.Problem(spec, p64, tt64).train_loss(ts.double(), x0.double(), noise.double(), method=method); l64.backward()
    except Exception as exc:
        print("   (float64 oracle failed:", str(exc)[:80], ")"); l64 = None
    prob.to("cuda:0")
    val, _ = prob.loss(prob.ts, x0.to("cuda:0"), prob.target.unnorm_log_prob, prob.second_log_prob, noise=noise.to("cuda:0")); val.backward()
    print("   loss", val.item(), "ref", ref_loss.item(), "f64", None if l64 is None else l64.item(), prob.loss.engine.last_kernel_name())
    for k, p in prob.ctrl.named_parameters():
        g_ref = params[k].grad
        if g_ref is None: continue
        g = p.grad.cpu()
        e64 = "" if l64 is None or p64[k].grad is None else " | vs f64: hip %.3e ref %.3e" % ((g.double() - p64[k].grad).abs().max().item(), (g_ref.double() - p64[k].grad).abs().max().item())
        print("   %-44s |ref|max %.3e  err %.3e%s" % (k, g_ref.abs().max().item(), (g - g_ref).abs().max().item(), e64))

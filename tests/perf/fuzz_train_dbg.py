"""Per-parameter gradient errors of one random training case of tests/test_hip_fuzz.py (python fuzz_train_dbg.py CASE...)."""
import os, sys, math
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "tests"))
import numpy as np, torch
import test_hip_fuzz as F
from oracle import em_oracle as eo
from sde_sampler_amd import problems
for case in [int(a) for a in sys.argv[1:]]:
    rng = np.random.default_rng(5000 + case)
    spec = F.random_spec(rng)
    method = str(rng.choice(["kl", "kl_ito", "lv"]))
    spec["loss"]["method"] = method; spec["loss"]["max_rnd"] = 1e8 if method == "lv" else None
    spec["batch"] = int(rng.choice([33, 64, 100]))
    print(case, method, {k: spec[k] for k in ("loss", "ctrl", "sde", "target", "prior", "net", "grid", "batch")})
    prob = problems.build(spec)
    params = {k: v.detach().clone().requires_grad_(v.is_floating_point()) for k, v in prob.ctrl.state_dict().items()}
    tt = None
    if spec["target"]["kind"] == "gmm":
        tt = dict(loc=prob.target.loc.clone(), scale=prob.target.scale.clone(), mixture_weights=prob.target.mixture_weights.clone())
        print("   gmm loc", tuple(prob.target.loc.shape), "scale uniq", prob.target.scale.unique().numel())
    oracle = eo.Problem(spec, params, tt)
    ts = prob.ts.clone(); B, d, T = spec["batch"], spec["target"]["dim"], ts.numel() - 1
    torch.manual_seed(case); x0 = prob.prior.sample((B,)); noise = torch.randn(T, B, d)
    ref_loss, _, _, _ = oracle.train_loss(ts, x0.clone(), noise, method=method); ref_loss.backward()
    prob.to("cuda:0")
    val, _ = prob.loss(prob.ts, x0.to("cuda:0"), prob.target.unnorm_log_prob, prob.second_log_prob, noise=noise.to("cuda:0")); val.backward()
    print("   loss", val.item(), "ref", ref_loss.item())
    for k, p in prob.ctrl.named_parameters():
        g_ref = params[k].grad
        if g_ref is None: continue
        g = p.grad.cpu()
        print("   %-44s |ref|max %.3e  err %.3e" % (k, g_ref.abs().max().item(), (g - g_ref).abs().max().item()))

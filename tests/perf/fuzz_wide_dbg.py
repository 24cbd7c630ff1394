"""Details of the cases the wide sweep (SDEH_FUZZ_SCALE=5) flags: step at which rows start to differ, magnitudes."""
import os, sys, math
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "tests"))
import numpy as np, torch
import test_hip_fuzz as F
from oracle import em_oracle as eo
from sde_sampler_amd import problems
for case in [int(a) for a in sys.argv[1:]]:
    rng = np.random.default_rng(1000 + case)
    try:
        spec = F.random_spec(rng)
        prob = problems.build(spec)
    except Exception as e:
        print(case, "BUILD ERROR", type(e).__name__, e); continue
    params = {k: v.detach().clone() for k, v in prob.ctrl.state_dict().items()}
    tt = None
    if spec["target"]["kind"] == "gmm":
        tt = dict(loc=prob.target.loc.clone(), scale=prob.target.scale.clone(), mixture_weights=prob.target.mixture_weights.clone())
    oracle = eo.Problem(spec, params, tt)
    ts = prob.ts.clone(); B, d, T = spec["batch"], spec["target"]["dim"], ts.numel() - 1
    torch.manual_seed(case); x0 = prob.prior.sample((B,)); noise = torch.randn(T, B, d)
    weights = bool(rng.random() < 0.5)
    try:
        ref = oracle.eval(ts, x0.clone(), noise, compute_weights=weights, return_traj=True)
        prob.to("cuda:0")
        out = prob.eval(x0.to("cuda:0"), compute_weights=weights, return_traj=True, noise=noise.to("cuda:0"))
    except Exception as e:
        print(case, "RUN ERROR", type(e).__name__, str(e)[:300]); continue
    xs_h, xs_r = out.xs.cpu(), ref["xs"]
    err_t = (xs_h - xs_r).abs().amax(dim=(1, 2))
    print(case, spec["loss"], spec["ctrl"], spec["sde"], spec["target"], "B", B, "T", T)
    print("   max|x| per step (ref):", ["%.2g" % v for v in xs_r.abs().amax(dim=(1, 2))[:: max(1, T // 8)].tolist()])
    print("   max err per step     :", ["%.2g" % v for v in err_t[:: max(1, T // 8)].tolist()], " nonfinite ref rows:", int((~torch.isfinite(xs_r[-1]).all(dim=1)).sum()))
    key = "log_norm_const_lb_ito" if weights else "log_norm_const_lb"
    print("   ", key, out.log_norm_const_preds[key], "vs", ref[key])

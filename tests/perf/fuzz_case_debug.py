import sys, os
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from tests.test_hip_fuzz import random_spec
from oracle import em_oracle as eo
from sde_sampler_amd import problems
case = 80
rng = np.random.default_rng(1000 + case)
spec = random_spec(rng)
print(spec)
prob = problems.build(spec)
params = {k: v.detach().clone() for k, v in prob.ctrl.state_dict().items()}
oracle = eo.Problem(spec, params, None)
ts = prob.ts.clone(); B, d, T = spec["batch"], spec["target"]["dim"], ts.numel() - 1
torch.manual_seed(case)
x0 = prob.prior.sample((B,)); noise = torch.randn(T, B, d)
ref = oracle.eval(ts, x0.clone(), noise, compute_weights=False, return_traj=True)
prob.to("cuda:0")
out = prob.eval(x0.cuda(), compute_weights=False, return_traj=True, noise=noise.cuda())
err = (out.xs.cpu() - ref["xs"]).abs()
print("per-step max err:", [f"{e:.1e}" for e in err.amax(dim=(1, 2)).tolist()])
i = err[-1].amax(dim=1).argmax().item()
print("worst row", i, "x_T ref", ref["xs"][-1, i].tolist(), "\n got", out.xs[-1, i].cpu().tolist())
print("rows with err>1e-3:", (err[-1].amax(dim=1) > 1e-3).sum().item(), "of", B, "| max |x| along worst row:", ref["xs"][:, i].abs().max().item())
print("dt:", (ts[1:] - ts[:-1]).tolist())

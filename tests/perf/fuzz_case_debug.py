"""Step-by-step look at one case of tests/test_hip_fuzz.py::test_random_problem_matches_oracle: python tests/perf/fuzz_case_debug.py <case>
Prints the specification, the growth of the HIP-vs-oracle difference per step, the oracle's own response to input perturbations
per row, and the rows the criterion counts as drifted."""
import math
import os
import sys

sys.path.insert(0, os.getcwd())
import numpy as np
import torch

from oracle import em_oracle as eo
from sde_sampler_amd import problems
from tests.test_hip_fuzz import _MORE_PERTS, _PERTS, _perturbed, random_spec

case = int(sys.argv[1]) if len(sys.argv) > 1 else 80
rng = np.random.default_rng(1000 + case)
spec = random_spec(rng)
print(spec)
prob = problems.build(spec)
params = {k: v.detach().clone() for k, v in prob.ctrl.state_dict().items()}
tt = None
if spec["target"]["kind"] == "gmm":
    tt = dict(loc=prob.target.loc.clone(), scale=prob.target.scale.clone(), mixture_weights=prob.target.mixture_weights.clone())
oracle = eo.Problem(spec, params, tt)
ts = prob.ts.clone()
B, d, T = spec["batch"], spec["target"]["dim"], ts.numel() - 1
torch.manual_seed(case)
x0 = prob.prior.sample((B,))
noise = torch.randn(T, B, d)
weights = bool(rng.random() < 0.5)
ref = oracle.eval(ts, x0.clone(), noise, compute_weights=weights, return_traj=True)
conds = []
for eps in _PERTS + _MORE_PERTS:
    q = oracle.eval(ts, *_perturbed(x0, noise, eps), compute_weights=weights, return_traj=True)
    conds.append(torch.nan_to_num((q["xs"] - ref["xs"]).abs().amax(dim=(0, 2)), nan=math.inf))
cond3, cond11 = torch.stack(conds[:3]).amax(dim=0), torch.stack(conds).amax(dim=0)
prob.to("cuda:0")
out = prob.eval(x0.cuda(), compute_weights=weights, return_traj=True, noise=noise.cuda())
err = (out.xs.cpu() - ref["xs"]).abs()
scale = max(1.0, float(torch.nan_to_num(ref["xs"], nan=0.0, posinf=0.0, neginf=0.0).abs().max()))
print("scale", scale, "per-step max err:", [f"{e:.1e}" for e in err.amax(dim=(1, 2)).tolist()])
row = err.amax(dim=(0, 2))
bad = (row - cond11).clamp_min(0) > 2e-3 * scale
print("rows beyond the bar after 11 probes:", int(bad.sum()), "of", B)
for i in torch.nonzero(bad).flatten().tolist():
    first = int((err[:, i].amax(dim=1) > 1e-4 * scale).nonzero()[0]) if (err[:, i].amax(dim=1) > 1e-4 * scale).any() else -1
    print(f"  row {i}: err {row[i]:.3e}  oracle response 3 probes {cond3[i]:.3e} / 11 probes {cond11[i]:.3e}  max|x| {ref['xs'][:, i].abs().max():.3e}  "
          f"first step with err > 1e-4 scale: {first}  x there (ref / got): {ref['xs'][max(first,0), i].tolist()} / {out.xs[max(first,0), i].cpu().tolist()}")
print("dt:", (ts[1:] - ts[:-1]).tolist()[:6], "...")

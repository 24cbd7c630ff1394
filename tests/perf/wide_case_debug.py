#!/usr/bin/env python3
"""Step-by-step comparison of one case of tests/test_hip_wide.py::test_random_wide_problem_matches_oracle with the oracle:
python tests/perf/wide_case_debug.py <case> [--no-widen]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import em_oracle as eo  # noqa: E402
from sde_sampler_amd import problems  # noqa: E402
from tests.test_hip_fuzz import random_spec  # noqa: E402
from tests.test_hip_wide import _widen  # noqa: E402

case = int(sys.argv[1])
rng = np.random.default_rng(1000 + 3000 + case)
spec = random_spec(rng)
_widen(spec, rng)
import json


def _merge(dst, src):
    for k, v in src.items():
        if isinstance(v, dict) and isinstance(dst.get(k), dict):
            _merge(dst[k], v)
        else:
            dst[k] = v


if os.environ.get("OVR"):  # e.g. OVR='{"net": {"channels": 128}, "batch": 33}'
    _merge(spec, json.loads(os.environ["OVR"]))
    if "target" in json.loads(os.environ["OVR"]) and "dim" in json.loads(os.environ["OVR"])["target"]:
        spec["prior"]["dim"] = spec["target"]["dim"]
print({k: spec[k] for k in ("target", "prior", "sde", "ctrl", "net", "loss", "grid", "batch")})
prob = problems.build(spec)
params = {k: v.detach().clone() for k, v in prob.ctrl.state_dict().items()}
tt = None
if spec["target"]["kind"] == "gmm":
    tt = dict(loc=prob.target.loc.clone(), scale=prob.target.scale.clone(), mixture_weights=prob.target.mixture_weights.clone())
oracle = eo.Problem(spec, params, tt)
ts = prob.ts.clone()
B, d, T = spec["batch"], spec["target"]["dim"], ts.numel() - 1
torch.manual_seed(3000 + case)
x0 = prob.prior.sample((B,))
noise = torch.randn(T, B, d)
ref = oracle.eval(ts, x0.clone(), noise, compute_weights=True, return_traj=True)
prob.to("cuda:0")
out = prob.eval(x0.to("cuda:0"), compute_weights=True, return_traj=True, noise=noise.to("cuda:0"))
print("kernel", prob.loss.engine.last_kernel_name())
xs, rx = out.xs.cpu(), ref["xs"]
for t in range(T + 1):
    e = (xs[t] - rx[t]).abs()
    bad_rows = (e.amax(dim=1) > 1e-3).nonzero().flatten().tolist()
    bad_cols = (e.amax(dim=0) > 1e-3).nonzero().flatten().tolist()
    if t > 3 and not os.environ.get("ALL"):
        break
    print(f"t={t}: max err {e.max().item():.3e}  bad rows {len(bad_rows)} {bad_rows[:8]}  bad coords {len(bad_cols)} {bad_cols[:12]}{'...' if len(bad_cols) > 12 else ''}")
print("lb_ito", out.log_norm_const_preds["log_norm_const_lb_ito"], ref["log_norm_const_lb_ito"])

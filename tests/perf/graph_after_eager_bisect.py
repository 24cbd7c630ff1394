import os, subprocess, sys
CODE = r'''
import sys; sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import torch
from test_hip_graphs import _build, _params
from sde_sampler_amd.utils.graphs import GraphedTrainStep
variant = sys.argv[1]
prob = _build(3, "lv"); params = _params(prob)
fn = lambda: prob.loss(prob.ts, prob.prior.sample((1024,)), prob.target.unnorm_log_prob, prob.second_log_prob)[0]
if variant == "eval_only":
    prob.eval(prob.prior.sample((1024,)), compute_weights=True)
elif variant == "fwd_nograd":
    with torch.no_grad(): fn()
elif variant == "fwd_grad":
    v = fn(); del v
elif variant == "fwd_bwd":
    v = fn(); v.backward(); del v
elif variant == "fwd_bwd_keep":
    kept = fn(); kept.backward(retain_graph=True)
elif variant == "fwd_bwd_opt":
    opt = torch.optim.Adam(params, lr=1e-3); v = fn(); v.backward(); opt.step(); del v
elif variant == "fwd_bwd_sync":
    v = fn(); v.backward(); del v; torch.cuda.synchronize(); import gc; gc.collect(); torch.cuda.empty_cache()
opt_g = torch.optim.Adam(params, lr=1e-3, capturable=True)
g = GraphedTrainStep(fn, [prob.loss], opt_g)
for _ in range(3): val = g()
print("OK", float(val))
'''
for variant in ["none", "eval_only", "fwd_nograd", "fwd_grad", "fwd_bwd", "fwd_bwd_keep", "fwd_bwd_opt", "fwd_bwd_sync"]:
    r = subprocess.run([sys.executable, "-c", CODE, variant], capture_output=True, text=True)
    last = [l for l in r.stdout.splitlines() if l.startswith("OK")]
    print(f"{variant:14s} rc={r.returncode} {last[-1] if last else ''}", flush=True)

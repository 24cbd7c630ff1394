import sys, json, torch
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sde_sampler_amd import problems
spec = problems.baseline_spec("gmm50_pis_headline")
prob = problems.build(spec, device="cuda:0")
B = int(sys.argv[2]) if len(sys.argv) > 2 else 65536
x0 = prob.prior.sample((B,))
prob.loss.engine.timing = True
ms = []
n_iter = int(sys.argv[1]) if len(sys.argv) > 1 else 6
for i in range(n_iter):
    r = prob.eval(x0, compute_weights=False)
    ms.append(prob.loss.engine.last_kernel_ms())
print("kernel ms", [round(m, 3) for m in ms], r.log_norm_const_preds)

import sys, os
sys.path.insert(0, os.getcwd())
import torch
from sde_sampler_amd import problems
for name, B in [("cfg3_gmm50_pis_kl", 32768), ("cfg4_funnel_dds_lv", 32768), ("cfg2_gmm2_dis_kl", 32768), ("gmm50_pis_headline", 24576)]:
    prob = problems.build(problems.baseline_spec(name), device="cuda:0")
    torch.manual_seed(0)
    x0 = prob.prior.sample((B,))
    prob.loss.engine.timing = True
    ms = []
    for i in range(14):
        r = prob.eval(x0, compute_weights=False)
        ms.append(prob.loss.engine.last_kernel_ms())
    T = prob.ts.numel() - 1
    print(f"{name:22s} B={B:6d} T={T}: kernel {min(ms[6:]):7.3f} ms  ({min(ms[6:]) * 1e3 / T:6.2f} us/step)  lb={r.log_norm_const_preds['log_norm_const_lb']:+.4f}", flush=True)

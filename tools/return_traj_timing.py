import sys, os
sys.path.insert(0, os.getcwd())
import torch
from sde_sampler_amd import problems
for name in ("gmm50_pis_headline", "cfg2_gmm2_dis_kl", "cfg4_funnel_dds_lv"):
    spec = problems.baseline_spec(name)
    prob = problems.build(spec, device="cuda:0")
    B = spec["batch"]
    x0 = prob.prior.sample((B,))
    prob.loss.engine.timing = True
    T, d = prob.ts.numel() - 1, spec["target"]["dim"]
    for rt in (False, True):
        ms = []
        for i in range(12):
            r = prob.eval(x0, compute_weights=True, return_traj=rt)
            ms.append(prob.loss.engine.last_kernel_ms())
        gb = (T + 1) * B * d * 4 / 1e9
        print(f"{name:22s} return_traj={rt!s:5s}: kernel {min(ms[5:]):7.3f} ms" + (f"   xs = {gb:.2f} GB -> {gb / (min(ms[5:]) * 1e-3):.0f} GB/s if it were only the write" if rt else ""), flush=True)
    del r

import sys, os, time
sys.path.insert(0, os.getcwd())
import torch
from sde_sampler_amd import problems
for name, B in [("cfg1_dw_dis_lv", 1024), ("cfg2_gmm2_dis_kl", 6000), ("gmm50_pis_headline", 6000), ("gmm50_pis_headline", 65536)]:
    prob = problems.build(problems.baseline_spec(name), device="cuda:0")
    x0 = prob.prior.sample((B,))
    prob.loss.engine.timing = True
    for _ in range(15): prob.eval(x0, compute_weights=False)
    torch.cuda.synchronize(); t0 = time.perf_counter(); k = []
    for _ in range(50):
        prob.eval(x0, compute_weights=False); k.append(prob.loss.engine.last_kernel_ms())
    torch.cuda.synchronize(); wall = (time.perf_counter() - t0) / 50 * 1e3
    print(f"{name:22s} B={B:6d}: wall {wall:6.3f} ms per eval, trajectory kernel {sum(k)/len(k):6.3f} ms, other {wall - sum(k)/len(k):6.3f} ms")

import sys, os, cProfile, pstats, time
sys.path.insert(0, os.getcwd())
import torch
from sde_sampler_amd import problems
spec = problems.baseline_spec("gmm50_pis_headline")
prob = problems.build(spec, device="cuda:0")
x0 = prob.prior.sample((65536,))
for _ in range(12): prob.eval(x0, compute_weights=False)
torch.cuda.synchronize()
t0=time.perf_counter()
for _ in range(50): prob.eval(x0, compute_weights=False)
torch.cuda.synchronize()
print("ms/step", (time.perf_counter()-t0)/50*1e3)
pr = cProfile.Profile(); pr.enable()
for _ in range(50): prob.eval(x0, compute_weights=False)
pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(22)

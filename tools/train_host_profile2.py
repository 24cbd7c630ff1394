import sys, os, cProfile, pstats, time
sys.path.insert(0, os.getcwd())
import torch
from sde_sampler_amd import problems
name = sys.argv[1] if len(sys.argv) > 1 else "cfg3_gmm50_pis_kl"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 65536
spec = problems.baseline_spec(name); spec["batch"] = B
prob = problems.build(spec, device="cuda:0")
opt = torch.optim.Adam(prob.ctrl.parameters(), lr=1e-4)
def step():
    x0 = prob.prior.sample((B,))
    opt.zero_grad(set_to_none=True)
    loss, _ = prob.loss(prob.ts, x0, prob.target.unnorm_log_prob, prob.second_log_prob)
    loss.backward()
    opt.step()
    return loss
for _ in range(5): step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(10): step()
torch.cuda.synchronize()
print("ms/step", (time.perf_counter() - t0) / 10 * 1e3, "peak GB", torch.cuda.max_memory_allocated() / 1e9, "reserved GB", torch.cuda.memory_reserved() / 1e9)
pr = cProfile.Profile(); pr.enable()
for _ in range(10): step()
torch.cuda.synchronize()
pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(14)

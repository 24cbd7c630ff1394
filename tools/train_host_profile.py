import sys, os, cProfile, pstats, time
sys.path.insert(0, os.getcwd())
import torch
from sde_sampler_amd import problems
prob = problems.build(problems.baseline_spec("cfg1_dw_dis_lv"), device="cuda:0")
opt = torch.optim.Adam(prob.ctrl.parameters(), lr=5e-3)
def step():
    x = prob.prior.sample((2048,))
    loss, _ = prob.loss(prob.ts, x, prob.target.unnorm_log_prob, prob.second_log_prob)
    opt.zero_grad(set_to_none=True); loss.backward(); opt.step()
for _ in range(30): step()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(100): step()
torch.cuda.synchronize(); print("ms/step", (time.perf_counter() - t0) * 10)
pr = cProfile.Profile(); pr.enable()
for _ in range(50): step()
torch.cuda.synchronize()
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(28)

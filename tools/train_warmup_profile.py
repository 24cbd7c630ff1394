import sys, os, time
sys.path.insert(0, os.getcwd())
import torch
from sde_sampler_amd import problems
prob = problems.build(problems.baseline_spec("cfg1_dw_dis_lv"), device="cuda:0")
opt = torch.optim.Adam(prob.ctrl.parameters(), lr=5e-3)
torch.cuda.synchronize(); t0 = time.perf_counter(); marks = []
for step in range(160):
    x = prob.prior.sample((2048,))
    loss, _ = prob.loss(prob.ts, x, prob.target.unnorm_log_prob, prob.second_log_prob)
    opt.zero_grad(set_to_none=True); loss.backward(); opt.step()
    if (step + 1) % 10 == 0:
        torch.cuda.synchronize(); t = time.perf_counter(); marks.append(round((t - t0) * 100, 1)); t0 = t
print("ms/step per block of 10:", marks)

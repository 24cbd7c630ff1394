import sys, os
sys.path.insert(0, os.getcwd())
import torch
from sde_sampler_amd import problems
from sde_sampler_amd.eq.integrator import EulerIntegrator
from sde_sampler_amd.eval.metrics import get_metrics
from sde_sampler_amd.eval.sinkhorn import Sinkhorn

prob = problems.build(problems.baseline_spec("cfg2_gmm2_dis_kl"), device="cuda:0")
x0 = prob.prior.sample((65536,))
res = prob.eval(x0, compute_weights=True)
print(res.log_norm_const_preds)
opt = torch.optim.Adam(prob.ctrl.parameters(), lr=5e-3)
loss, _ = prob.loss(prob.ts, prob.prior.sample((2048,)), prob.target.unnorm_log_prob, prob.second_log_prob)
loss.backward(); opt.step()
from sde_sampler_amd.utils.graphs import GraphedTrainStep
opt = torch.optim.Adam(prob.ctrl.parameters(), lr=5e-3, capturable=True, fused=True)  # fused: 1 launch instead of ~90
step = GraphedTrainStep(lambda: prob.loss(prob.ts, prob.prior.sample((2048,)), prob.target.unnorm_log_prob,
                                          prob.second_log_prob)[0], [prob.loss], opt)
for _ in range(20):
    loss = step()
print("graphed step loss", float(loss), "skipped", int(step.n_skipped))
prob.target.compute_stats()
m = get_metrics(prob.target, res.samples, res.weights, res.log_norm_const_preds, marginal_dims=[0, 1],
                sample_losses={"sinkhorn": Sinkhorn(n_max=4096)})
print({k: round(v, 4) for k, v in list(m.items())[:12]}, "...", "error/sinkhorn", m["error/sinkhorn"])
import __graft_entry__ as g
g.smoke()

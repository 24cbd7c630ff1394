import sys, os, torch
sys.path.insert(0, "/root/repo")
from sde_sampler_amd import problems
spec = problems.baseline_spec("wide_pis_funnel196"); spec["batch"] = 8192; spec["loss"]["method"] = "lv"
prob = problems.build(spec, device="cuda:0")
x0 = prob.prior.sample((8192,))
def step():
    prob.ctrl.zero_grad()
    val, _ = prob.loss(prob.ts, x0, prob.target.unnorm_log_prob, prob.second_log_prob)
    val.backward()
step(); torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True, record_shapes=True) as p:
    step(); torch.cuda.synchronize()
for e in p.events():
    if e.name in ("aten::copy_", "aten::contiguous", "aten::clone") and e.device_time_total > 300:
        print(e.name, e.device_time_total, e.input_shapes, [s for s in (e.stack or []) if "sde_sampler_amd" in s][:3])

import sys, torch
sys.path.insert(0, "/root/repo")
from sde_sampler_amd import problems
from sde_sampler_amd.utils.graphs import GraphedTrainStep
for name, B, T in (("wide_pis_funnel196", 2048, 20), ("cfg5_like_bridge196", 256, 4)):
    spec = problems.baseline_spec(name); spec["batch"] = B; spec["grid"]["steps"] = T; spec["loss"]["method"] = "lv"
    prob = problems.build(spec, device="cuda:0")
    inf = getattr(prob.loss, "inference_ctrl", None)
    params = list(prob.ctrl.parameters()) + (list(inf.parameters()) if inf is not None else [])
    opt = torch.optim.Adam(params, lr=1e-4, capturable=True)
    try:
        step = GraphedTrainStep(lambda: prob.loss(prob.ts, prob.prior.sample((B,)), prob.target.unnorm_log_prob, prob.second_log_prob)[0], [prob.loss], opt)
        vals = [float(step()) for _ in range(5)]
        torch.cuda.synchronize()
        print(name, "graphed losses", vals)
    except Exception as exc:
        print(name, "graph capture failed:", type(exc).__name__, str(exc)[:300])

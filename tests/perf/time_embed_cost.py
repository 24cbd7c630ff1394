"""How much of a replayed training step is the autograd pass over the two time-only sub-networks ([T, .] tables)?"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from sde_sampler_amd import problems
from sde_sampler_amd.utils.graphs import GraphedTrainStep
for freeze in (False, True):
    prob = problems.build(problems.baseline_spec("cfg1_dw_dis_lv"), device="cuda:0")
    torch.manual_seed(0)
    if freeze:
        for p in prob.ctrl.base_model.timestep_embed.parameters(): p.requires_grad_(False)
        if getattr(prob.ctrl, "score_model", None) is not None:
            for p in prob.ctrl.score_model.parameters(): p.requires_grad_(False)
    params = [p for p in prob.ctrl.parameters() if p.requires_grad]
    opt = torch.optim.Adam(params, lr=5e-3, capturable=True)
    fn = lambda: prob.loss(prob.ts, prob.prior.sample((2048,)), prob.target.unnorm_log_prob, prob.second_log_prob)[0]
    g = GraphedTrainStep(fn, [prob.loss], opt, after_backward=lambda: torch.nn.utils.clip_grad_norm_(params, 1.0))
    for _ in range(20): g()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(200): g()
    torch.cuda.synchronize()
    print("time-only sub-networks frozen" if freeze else "all parameters trained", f"{(time.perf_counter() - t0) / 200 * 1e3:.3f} ms/step", len(params), "tensors")

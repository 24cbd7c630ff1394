"""N optimiser steps with the HIP training path against N steps with the oracle's autograd, on identical inputs and noise
(VERDICT r03 weak 3 / next 6): per-step gradient parity implies trajectory-of-parameters parity only to first order -- this test
takes the steps.  The reference's trainer step is solver/base.py:399-432 (loss(...), backward, optimizer.step); the schedule here is
its Adam at lr = 1e-3 (conf/solver/basic_oc_base.yaml), 20 steps, at the BASELINE configurations' fixture sizes (cfg1 - cfg4)."""
from pathlib import Path

import numpy as np
import pytest
import torch

from oracle import em_oracle as eo
from tests.helpers import GOLDEN, hip_problem, load_fixture, measured

pytestmark = pytest.mark.gpu

CASES = [p for p in GOLDEN if Path(p).name.startswith(("cfg1_", "cfg2_", "cfg3_", "cfg4_"))]
N_STEPS, LR = 20, 1e-3
LOSS_BAR = 1e-3   # of max(1, |loss|) at every step (the parameters of the two runs drift apart: bars from the measured values)
# Parameters after the 20 steps.  Adam's update is g / (sqrt(v) + eps): an entry whose gradient is at rounding level (|g| ~ 1e-9 next
# to entries of 1e-2) moves by +-lr per step in BOTH runs, in a direction the last bit decides -- so single entries of the two runs
# separate by up to 2 lr per step whatever the kernels do, and the max-norm bar cannot be the per-gradient 1e-4.  Two measures, both
# at 2 x the worst value measured over the eight cases (profiles/r04_parity_measured.txt): the largest entry-wise difference relative
# to the tensor's scale (measured 1.5e-3: `base_model.timestep_embed.hidden_layer.0.weight` of cfg1 / kl, a [T, .]-table network fed
# by 100 rows) and the l2 distance of the whole UPDATE theta_20 - theta_0 relative to its length (measured 1.3e-3 for that case,
# <= 3.8e-5 for the other seven).
PARAM_BAR = 3e-3
UPDATE_L2_BAR = 2.5e-3


@pytest.mark.parametrize("method", ["kl", "lv"])
@pytest.mark.parametrize("path", CASES, ids=lambda p: Path(p).stem)
def test_twenty_adam_steps_match_the_oracle(path, method):
    fx, meta, params, tt = load_fixture(path)
    meta["loss"]["method"] = method
    meta["loss"]["max_rnd"] = 1e8 if method == "lv" else None
    prob = hip_problem(meta, params, tt)
    prob.loss.method = method
    B, d = fx["x0"].shape
    T = prob.ts.numel() - 1
    ts = prob.ts.cpu().clone()
    p_ref = {k: v.detach().clone().requires_grad_(v.is_floating_point()) for k, v in prob.ctrl.state_dict().items()}
    p_ref = {k: (v.cpu().detach().clone().requires_grad_(v.is_floating_point())) for k, v in p_ref.items()}
    opt_ref = torch.optim.Adam([v for v in p_ref.values() if v.requires_grad], lr=LR)
    opt_hip = torch.optim.Adam(list(prob.ctrl.parameters()), lr=LR)
    gen = torch.Generator().manual_seed(1234)
    torch.set_num_threads(4)
    losses = []
    for step in range(N_STEPS):
        x0 = torch.from_numpy(fx["x0"]) if step == 0 else prob.prior.sample((B,)).cpu()
        noise = torch.randn(T, B, d, generator=gen)
        # oracle: the reference's loss + autograd + Adam on the CPU
        opt_ref.zero_grad(set_to_none=True)
        loss_ref, _, _, _ = eo.Problem(meta, p_ref, tt).train_loss(ts, x0.clone(), noise, method=method)
        loss_ref.backward()
        opt_ref.step()
        # HIP: fused forward + backward kernels + Adam
        opt_hip.zero_grad(set_to_none=True)
        val, _ = prob.loss(prob.ts, x0.cuda(), prob.target.unnorm_log_prob, prob.second_log_prob, noise=noise.cuda())
        val.backward()
        opt_hip.step()
        losses.append(abs(val.item() - loss_ref.item()) / max(1.0, abs(loss_ref.item())))
    measured(f"loss over {N_STEPS} Adam steps {Path(path).stem} {method} (first / worst)", max(losses), LOSS_BAR)
    assert losses[0] <= 1e-4, losses[0]  # step 0: identical parameters -- the per-step bar of tests/test_hip_parity.py
    assert max(losses) <= LOSS_BAR, f"loss values drift apart: {['%.1e' % v for v in losses]}"
    named = dict(prob.ctrl.named_parameters())
    moved = max((p_ref[k].detach() - torch.from_numpy(fx["param/" + k])).abs().max().item() for k in named if "param/" + k in fx.files)
    assert moved > 10 * LR * 0.5, "the parameters must actually have moved"  # 20 Adam steps of ~lr each
    pmax = max(v.detach().abs().max().item() for v in p_ref.values() if v.requires_grad)
    worst, worst_k = 0.0, ""
    for k, p in named.items():
        ref = p_ref[k].detach()
        scale = max(ref.abs().max().item(), 1e-3 * pmax)
        err = (p.detach().cpu() - ref).abs().max().item() / scale
        if err > worst:
            worst, worst_k = err, k
    measured(f"params after {N_STEPS} Adam steps {Path(path).stem} {method} ({worst_k})", worst, PARAM_BAR)
    num = sum(((p.detach().cpu() - p_ref[k].detach()) ** 2).sum().item() for k, p in named.items())
    den = sum(((p_ref[k].detach() - torch.from_numpy(fx["param/" + k])) ** 2).sum().item() for k in named if "param/" + k in fx.files)
    rel_l2 = (num / den) ** 0.5
    measured(f"update l2 after {N_STEPS} Adam steps {Path(path).stem} {method}", rel_l2, UPDATE_L2_BAR)
    assert worst <= PARAM_BAR, f"{worst_k}: parameters differ by {worst:.2e} of their scale after {N_STEPS} steps"
    assert rel_l2 <= UPDATE_L2_BAR, f"the updates of the two runs differ by {rel_l2:.2e} of their length"

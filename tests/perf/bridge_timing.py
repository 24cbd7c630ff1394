#!/usr/bin/env python3
"""Times the Bridge evaluation kernel (two network passes + d tangent recursions through the hidden layers per step) at
eval-batch size."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import time
import torch
from sde_sampler_amd import problems
from oracle import em_oracle as eo

NET = dict(channels=64, num_layers=4, activation="gelu")
for name, tspec, B, T in [("basic_bridge gmm-fab d=2", dict(kind="gmm", dim=2, name="fab"), 65536, 100),
                          ("bridge funnel d=10", dict(kind="funnel", dim=10), 32768, 100)]:
    d = tspec["dim"]
    spec = dict(batch=B, target=tspec, prior=dict(kind="iso_gauss", dim=d),
                sde=dict(kind="scaled_bm", diff_coeff=1.0, terminal_t=1.0),
                ctrl=dict(kind="lerp_target", clip_model=1e4, clip_score=1e4, scale_score=1.0, gamma_dim=1, gamma_bias=1.0),
                inference_ctrl=dict(kind="lerp_prior", clip_model=1e4, clip_score=1e4, scale_score=1.0, gamma_dim=1, gamma_bias=1.0),
                net=NET, loss=dict(kind="time_reversal", method="kl"), grid=dict(start=0.0, end=1.0, steps=T))
    prob = problems.build(spec, device="cuda:0")
    x0 = prob.prior.sample((B,))
    prob.loss.engine.timing = True
    ms = []
    for i in range(12):
        r = prob.eval(x0, compute_weights=False)
        ms.append(prob.loss.engine.last_kernel_ms())
    best = min(ms[4:])
    # CPU oracle (the reference loop restated: d autograd backward passes per step for the divergence), bounded sample
    torch.set_num_threads(min(32, os.cpu_count() or 1))
    tt = None
    if tspec["kind"] == "gmm":
        tt = dict(loc=prob.target.loc.cpu(), scale=prob.target.scale.cpu(), mixture_weights=prob.target.mixture_weights.cpu())
    cpu_params = {k: v.detach().cpu() for k, v in prob.ctrl.state_dict().items()}
    cpu_inf = {k: v.detach().cpu() for k, v in prob.loss.inference_ctrl.state_dict().items()}
    oracle = eo.Problem(spec, cpu_params, tt, cpu_inf)
    nb, ns = 2048, 10
    t0 = time.perf_counter()
    oracle.eval(prob.ts[: ns + 1].cpu(), x0[:nb].cpu(), None, compute_weights=False)
    cpu = nb * ns / (time.perf_counter() - t0)
    print(f"{name:28s} B={B} T={T} d={d}: kernel {best:8.3f} ms  {B * T / best / 1e6:6.3f} G traj-steps/s  "
          f"(2 network passes + {d} tangent recursions per step)  lb={r.log_norm_const_preds['log_norm_const_lb']:+.4f}  | CPU oracle "
          f"{cpu / 1e3:7.1f} k traj-steps/s ({nb} x {ns} steps, 32 threads) -> x{B * T / best * 1e3 / cpu:,.0f}", flush=True)

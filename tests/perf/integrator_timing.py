#!/usr/bin/env python3
"""Times sdeh_integrate (EulerIntegrator.integrate as one kernel) on the reference's Langevin solver configuration
(conf/solver/langevin.yaml: B = 6000, dt = 0.01 over [0, 100] = 10 000 steps, 1001 output times) and on a throughput-sized
batch, and the CPU oracle (reference loop restated) on a bounded number of steps of the same problem."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from sde_sampler_amd import problems
from sde_sampler_amd.eq.integrator import EulerIntegrator
from oracle import em_oracle as eo

CASES = [
    ("langevin gmm-fab d=2  (langevin.yaml)", dict(kind="gmm", dim=2, name="fab"), 6000, 100.0, 0.01, 1001),
    ("langevin funnel d=10  (langevin.yaml)", dict(kind="funnel", dim=10), 6000, 100.0, 0.01, 1001),
    ("langevin gmm-40 d=50  B=65536", dict(kind="gmm", dim=50, name="fab50"), 65536, 10.0, 0.01, 11),
]
for name, tspec, B, T, dt, n_out in CASES:
    meta = dict(target=tspec, prior=dict(kind="iso_gauss", dim=tspec["dim"], loc=0.0, scale=1.0),
                integrate=dict(kind="langevin", diff_coeff=1.0, clip_score=1e5), grid=dict(start=0.0, end=T, steps=0))
    sde, target, prior, _ = problems.build_integration(meta, device="cuda:0")
    torch.manual_seed(0)
    x0 = prior.sample((B,))
    ts = torch.linspace(0.0, T, n_out, device="cuda:0")
    integ = EulerIntegrator(dt=dt)
    integ.engine.timing = True
    ms = []
    for i in range(4):
        xs = integ.integrate(sde, ts=ts, x_init=x0, seed=3)
        ms.append(integ.engine.last_kernel_ms())
    steps = int(round(T / dt))
    best = min(ms[1:])
    d = tspec["dim"]
    # CPU oracle: same drift/diffusion, 200 steps of the same batch (bounded sample), torch threads = min(32, cores)
    tt = None
    if tspec["kind"] == "gmm":
        tt = dict(loc=target.loc.cpu(), scale=target.scale.cpu(), mixture_weights=target.mixture_weights.cpu())
    drift, diff = eo.integration_case(meta, {}, tt)
    torch.set_num_threads(min(32, os.cpu_count() or 1))
    nb, ns = min(B, 6000), 100
    grid = torch.linspace(0.0, ns * dt, ns + 1)
    t0 = time.perf_counter()
    eo.euler_integrate(drift, diff, grid[[0, -1]], x0[:nb].cpu(), grid)
    cpu = nb * ns / (time.perf_counter() - t0)
    print(f"{name:40s} B={B:6d} steps={steps:6d} n_out={n_out:5d}  kernel {best:9.3f} ms  {B * steps / best / 1e6:7.3f} G traj-steps/s"
          f"  out {xs.numel() * 4 / 1e6:8.1f} MB  | CPU oracle {cpu / 1e6:7.3f} M traj-steps/s ({nb} x {ns} steps) -> x{B * steps / best * 1e3 / cpu:,.0f}"
          f"  mean|x_end|={xs[-1].abs().mean().item():.3f}", flush=True)

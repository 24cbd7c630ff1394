#!/usr/bin/env python3
"""Times the evaluation-side kernels: Sinkhorn() at the reference's settings (eps 1e-3, 100 iterations) on eval-batch-sized
clouds, get_metrics' statistics pass, and the CPU dense oracle on a bounded sample."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from sde_sampler_amd.eval.sinkhorn import Sinkhorn
from sde_sampler_amd.eval.metrics import sample_stats
from oracle import eval_oracle as ev

dev = "cuda:0"
for n, d in [(6000, 2), (65536, 2), (32768, 50), (32768, 10)]:
    torch.manual_seed(0)
    x = torch.randn(n, d, device=dev) * 3
    y = torch.randn(n, d, device=dev) * 3 + 0.5
    sk = Sinkhorn()
    sk(x, y).item()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    v = sk(x, y).item()
    dt = time.perf_counter() - t0
    it = sk.info()["iterations"]
    pairs = 2.0 * it * n * n + n * n  # two sweeps per iteration + the distance sweep
    # CPU dense oracle (fp32, 32 threads) on a bounded sample: 2048 x 2048, 10 iterations
    torch.set_num_threads(min(32, os.cpu_count() or 1))
    xs, ys = x[:2048].cpu(), y[:2048].cpu()
    t0 = time.perf_counter()
    ev.sinkhorn_dense(xs, ys, max_iters=10)
    cpu = (2.0 * 10 + 1) * 2048 * 2048 / (time.perf_counter() - t0)
    print(f"sinkhorn n=m={n:6d} d={d:3d}: {dt * 1e3:9.1f} ms for {it} iterations  {pairs / dt / 1e12:6.3f} T pair-updates/s  "
          f"value {v:.5f} | CPU dense oracle {cpu / 1e9:6.3f} G pair-updates/s (2048^2 x 10 it) -> x{pairs / dt / cpu:,.0f}", flush=True)
for B, d in [(65536, 2), (65536, 50), (262144, 50)]:
    x = torch.randn(B, d, device=dev)
    w = torch.rand(B, 1, device=dev)
    sample_stats(x, weights=w)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        st = sample_stats(x, weights=w)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 20
    print(f"sample_stats B={B:7d} d={d:3d}: {ms:7.3f} ms per call incl. host merge  ({B * d * 4 * 2 / ms / 1e6:8.1f} GB/s of samples read twice)")

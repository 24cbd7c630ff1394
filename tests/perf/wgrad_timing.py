#!/usr/bin/env python3
"""sdeh_weight_grad against the torch pipeline it replaced (GELU kernel + split-K bmm + sum + bias reduction) on one layer."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from sde_sampler_amd import _lib as L
from sde_sampler_amd.losses._autograd import _wgrad


def old(dk, zk, chunk=4096):
    a = torch.nn.functional.gelu(zk)
    P, N = dk.shape
    S = N // chunk
    main = S * chunk
    out = torch.bmm(dk[:, :main].view(P, S, chunk).transpose(0, 1), a[:, :main].view(64, S, chunk).permute(1, 2, 0)).sum(dim=0)
    if main < N:
        out += dk[:, main:] @ a[:, main:].t()
    return out, dk.sum(dim=1)


def timeit(fn, n=20):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


for B in (2048, 16384, 65536):
    N = 100 * B
    dk, zk = torch.randn(64, N, device="cuda:0"), torch.randn(64, N, device="cuda:0")
    t_new = timeit(lambda: _wgrad(dk, zk, L.ACT_GELU_ERF))
    t_old = timeit(lambda: old(dk, zk))
    gb = 2 * 64 * N * 4 / 1e9
    print(f"N = 100 x {B:6d}: sdeh_weight_grad + partial sums {t_new:7.3f} ms ({gb / t_new:6.2f} TB/s of the two planes)   "
          f"torch gelu + split-K bmm + sums {t_old:7.3f} ms", flush=True)

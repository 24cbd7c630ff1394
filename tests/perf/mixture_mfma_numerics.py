#!/usr/bin/env python3
"""VERDICT r04 next-step 2, "measure the numerics first": what happens to a dense mixture's logits, responsibilities and score when the two
contractions are written as matrix products (the form v_mfma_f32_* could take: logit_k = c_k + sum_d x_d mu_kd / sigma_d^2, score numerator
sum_k r_k mu_kd) instead of the difference form the kernels and the reference evaluate (sum_d (x_d - mu_kd)^2 / 2 sigma_d^2;
/root/reference/sde_sampler/distr/gauss.py:123-140 through torch.distributions).  CPU-only (numpy): fp32 chains in the order of a matrix
instruction's k loop (fp32 MFMA is bitwise an fmaf chain) against float64.

States: trajectories of the bench's dense mixture (bench.py `extra.gmm50_dense_shared`: 40 modes, means U(-40, 40) in all 50 coordinates)
integrated by the oracle with the untrained control (the benchmark's state distribution) AND states placed at / between modes (what a trained
sampler visits).  Variants: plain product; product centred on the mixture mean; product centred per trajectory on its nearest mode (not a
matrix product any more: the reference point differs per column).

    python tests/perf/mixture_mfma_numerics.py > profiles/r05_mixture_mfma_numerics.txt"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import em_oracle as eo  # noqa: E402  (measurement infrastructure)
from sde_sampler_amd import problems  # noqa: E402


def fma_chain(a, b):
    """sum_d a[..., d] * b[..., d] as an fp32 fmaf chain in d order (exact products via float64, one rounding per step)."""
    acc = np.zeros(np.broadcast(a[..., 0], b[..., 0]).shape, np.float32)
    for j in range(a.shape[-1]):
        acc = (acc.astype(np.float64) + a[..., j].astype(np.float64) * b[..., j].astype(np.float64)).astype(np.float32)
    return acc


def study(x, mu, sig, label):
    """x [N, d] fp32 states; mu [K, d], sig [d] (shared scale)."""
    x64, mu64, sig64 = x.astype(np.float64), mu.astype(np.float64), sig.astype(np.float64)
    # float64 truth
    l64 = -0.5 * (((x64[:, None, :] - mu64[None]) / sig64) ** 2).sum(-1)
    r64 = np.exp(l64 - l64.max(1, keepdims=True))
    r64 /= r64.sum(1, keepdims=True)
    s64 = (r64[:, :, None] * (mu64[None] - x64[:, None, :])).sum(1) / sig64 ** 2
    top2 = np.sort(l64, 1)[:, -2:]
    gap = top2[:, 1] - top2[:, 0]

    def metrics(l32, s32=None):
        r = np.exp((l32 - l32.max(1, keepdims=True)).astype(np.float64))
        r /= r.sum(1, keepdims=True)
        if s32 is None:
            s32 = (r[:, :, None] * (mu64[None] - x64[:, None, :])).sum(1) / sig64 ** 2
        dl = np.abs((l32 - l32.max(1, keepdims=True)) - (l64 - l64.max(1, keepdims=True)))
        live = r64 > 1e-12  # logit errors only matter where the component carries weight
        return (float(np.where(live, dl, 0).max()), float(np.abs(r - r64).max()),
                float((np.abs(s32 - s64).max(1) / np.maximum(1.0, np.abs(s64).max(1))).max()))

    isg = (1.0 / sig.astype(np.float64) ** 2)
    # (a) difference form in fp32, as the kernels do: sum_d (x - mu)^2 * (1 / 2 sigma^2)
    dif = (x[:, None, :] - mu[None]).astype(np.float32)
    la = -fma_chain(dif * dif, np.broadcast_to((0.5 * isg).astype(np.float32), dif.shape))
    # (b) product form: c_k + sum_d x_d m_kd  (the x^2 term is common to all components and drops out of the softmax)
    m = (mu64 * isg).astype(np.float32)
    c = (-0.5 * (mu64 ** 2 * isg).sum(1)).astype(np.float32)
    lb = c[None] + fma_chain(x[:, None, :], np.broadcast_to(m[None], (x.shape[0],) + m.shape))
    # (c) centred on the mixture mean
    ctr = mu64.mean(0)
    xc, muc = (x64 - ctr).astype(np.float32), (mu64 - ctr)
    mc = (muc * isg).astype(np.float32)
    cc = (-0.5 * (muc ** 2 * isg).sum(1)).astype(np.float32)
    lc = cc[None] + fma_chain(xc[:, None, :], np.broadcast_to(mc[None], (x.shape[0],) + mc.shape))
    # (d) centred per trajectory on its nearest mode (a different reference per column: a gather, not one matrix product)
    near = l64.argmax(1)
    xd = (x64 - mu64[near]).astype(np.float32)
    mud = mu64[None] - mu64[near][:, None, :]
    ld = (-0.5 * (mud ** 2 * isg).sum(-1)).astype(np.float32) + fma_chain(xd[:, None, :], (mud * isg).astype(np.float32))
    # score numerator as a product: sum_k r_k mu_kd (fp32 chain over k), minus x
    ra = np.exp((la - la.max(1, keepdims=True)).astype(np.float32))
    ra = (ra / ra.sum(1, keepdims=True)).astype(np.float32)
    num = fma_chain(np.broadcast_to(ra[:, None, :], (x.shape[0], mu.shape[1], mu.shape[0])), np.broadcast_to(mu.T[None], (x.shape[0], mu.shape[1], mu.shape[0])))
    sprod = ((num.astype(np.float64) - x64) * isg)
    print(f"## {label}: {x.shape[0]} states, |x|_max {np.abs(x).max():.1f}; top-2 logit gap: median {np.median(gap):.1f}, "
          f"{(gap < 20).mean():.1%} of the states below 20 (two modes carry weight)")
    print("   form                                   max |d logit| (live comps)   max |d r|    max score err / max(1, |score|)")
    for name, l in (("difference form, fp32 (kernels)", la), ("product c_k + x . m_k", lb), ("product, centred on the mixture mean", lc),
                    ("product, centred on the nearest mode", ld)):
        a, b, cerr = metrics(l)
        print(f"   {name:38s} {a:12.2e} {b:24.2e} {cerr:14.2e}")
    a, b, cerr = metrics(la, sprod)
    print(f"   {'score numerator sum_k r_k mu_kd - x':38s} {'(exact logits)':>12s} {'':24s} {cerr:14.2e}")


def main():
    torch.manual_seed(0)
    spec = problems.baseline_spec("gmm50_dense_shared")
    spec["batch"] = 256
    prob = problems.build(spec)
    mu = prob.target.loc.numpy().astype(np.float32)
    sig = prob.target.scale[0].numpy().astype(np.float32)
    params = {k: v.detach().clone() for k, v in prob.ctrl.state_dict().items()}
    tt = dict(loc=prob.target.loc.clone(), scale=prob.target.scale.clone(), mixture_weights=prob.target.mixture_weights.clone())
    oracle = eo.Problem(spec, params, tt)
    ts = oracle.grid()
    x0 = torch.zeros(256, 50)
    res = oracle.eval(ts, x0, None, compute_weights=False, return_traj=True)
    xs = res["xs"].numpy() if "xs" in res else None
    print("# dense 40-mode mixture, d = 50, means U(-40, 40), sigma = softplus(1) = 1.313 (bench.py extra.gmm50_dense_shared)")
    print("# bars the path must hold: rows |dx_T| median 1e-4, estimators 1e-4 (SURVEY 8d); the score enters x through sigma dt ~ 0.01 per step, 100 steps")
    if xs is not None:
        study(xs[::10].reshape(-1, 50).astype(np.float32), mu, sig, "benchmark trajectories (untrained control, every 10th step)")
    rng = np.random.default_rng(1)
    at = mu[rng.integers(0, 40, 512)] + sig * rng.standard_normal((512, 50)).astype(np.float32)
    study(at.astype(np.float32), mu, sig, "states AT modes (mu_k + sigma xi: what a trained sampler ends on)")
    i, j = rng.integers(0, 40, 512), rng.integers(0, 40, 512)
    lam = rng.uniform(0.35, 0.65, (512, 1)).astype(np.float32)
    mid = lam * mu[i] + (1 - lam) * mu[j] + sig * rng.standard_normal((512, 50)).astype(np.float32)
    study(mid.astype(np.float32), mu, sig, "states BETWEEN two modes (competing responsibilities)")


if __name__ == "__main__":
    main()

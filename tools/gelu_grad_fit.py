"""Fit of gelu'(v) = Phi(v) + v phi(v) for the backward kernels (csrc/sdeh_bwd.hpp act_grad):
    gelu'(v) = v >= 0 ? 1 - r(t) : r(t),   t = min(|v|, TMAX),   r(t) = 1 - Phi(t) - t phi(t) = 2^{-t^2 log2(e)/2} P(t)
with P a polynomial (weighted minimax, weight = the Gaussian factor, Lawson iteration).  One v_exp_f32 + deg FMAs."""
import numpy as np
from numpy.polynomial import chebyshev as Ch
from scipy.special import ndtr, erfcx

C2 = 0.5 * np.log2(np.e)


def fit(deg, tmax, N=40001, iters=300):
    t = np.linspace(0, tmax, N)
    gauss = np.exp(-0.5 * t * t)
    # P(t) = (1 - Phi(t)) e^{t^2/2} - t / sqrt(2 pi)  (erfcx: scaled complementary error function, no cancellation)
    Pex = 0.5 * erfcx(t / np.sqrt(2)) - t / np.sqrt(2 * np.pi)
    x = 2 * t / tmax - 1
    V = Ch.chebvander(x, deg)
    lam = np.ones(N)
    for _ in range(iters):
        W = np.sqrt(lam) * gauss
        c, *_ = np.linalg.lstsq(V * W[:, None], Pex * W, rcond=None)
        err = np.abs((V @ c - Pex) * gauss)
        lam = lam * (err + 1e-30)
        lam /= lam.sum()
    poly = np.poly1d(Ch.cheb2poly(c)[::-1])(np.poly1d([2 / tmax, -1]))
    return poly.coeffs[::-1], err.max()


def eval32(coef, v, tmax):
    v = v.astype(np.float32)
    t = np.minimum(np.abs(v), np.float32(tmax)).astype(np.float32)
    c = coef.astype(np.float32)
    p = np.full_like(t, c[-1])
    for k in range(len(c) - 2, -1, -1):
        p = (p.astype(np.float64) * t + c[k]).astype(np.float32)
    e = np.exp2((-(np.float32(C2) * t).astype(np.float32).astype(np.float64) * t)).astype(np.float32)
    r = (p.astype(np.float64) * e).astype(np.float32)
    return np.where(v >= 0, (1.0 - r.astype(np.float64)).astype(np.float32), r)


if __name__ == "__main__":
    v = np.concatenate([np.linspace(-9, 9, 400001), np.random.default_rng(0).normal(size=200000) * 2])
    v64 = v.astype(np.float32).astype(np.float64)
    exact = ndtr(v64) + v64 * np.exp(-0.5 * v64 * v64) / np.sqrt(2 * np.pi)
    for tmax in (5.5, 6.0):
        for deg in (5, 6, 7, 8, 9):
            coef, e = fit(deg, tmax)
            err = np.abs(eval32(coef, v, tmax) - exact)
            print(f"tmax {tmax} deg {deg}: fit {e:.2e}  fp32 max abs err {err.max():.2e}")
            print("   coef", ", ".join(f"{c:.10e}" for c in coef))

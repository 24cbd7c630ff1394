#!/usr/bin/env python3
"""a reference-held number at the headline's OWN size -- the reference run once at B = 65 536 (build container only;
imports /root/reference through make_golden.py), scalars stored:

  fullsize_cfg2_gmm2_dis.npz       BASELINE configs[1]: GMM-40 d = 2, basic_dis, B = 65 536, T = 100
  fullsize_headline_gmm50_pis.npz  the metric's configuration: GMM-40 d = 50, basic_pis, B = 65 536, T = 100

The inputs are NOT stored (1.3 GB of noise): they are a function of a seed -- torch.manual_seed(seed); x0 = loc + scale * randn(B, d)
(a Delta prior: loc); then T calls of randn_like(x0), the draws the reference's loop consumes (losses/oc.py:213-219, 325-331) -- and the
fixture keeps float64 checksums of them, so the GPU test (tests/test_hip_fullsize.py) knows that it regenerated the same numbers before it
compares anything.  Stored outputs of loss.eval (losses/oc.py:94-123, 258-278): log_norm_const_lb (compute_weights=False),
log_norm_const_lb_ito / log_norm_const_is / eval/lv_loss (compute_weights=True), and x_T, rnd of 64 rows.  The network's parameters
are those of the small fixtures of the same cases (make_golden.py seeds them with torch.manual_seed(1)): a checksum is kept here.

Usage:  python tests/golden/make_golden_fullsize.py [name ...]          (about a minute of reference time per case)"""
from __future__ import annotations

import copy
import json
import sys
import time
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, str(Path(__file__).resolve().parent))
import make_golden as mg  # noqa: E402  (imports the reference)

OUT = Path(__file__).resolve().parent
B_FULL = 65536
FULL = {
    "fullsize_cfg2_gmm2_dis": ("cfg2_gmm2_dis_kl", 100, 2024),
    "fullsize_headline_gmm50_pis": ("cfg3_gmm50_pis_kl", 100, 2025),
}


def inputs(case, seed, T):
    """x0 and the T increments as a function of the seed alone (the GPU test calls the same function)."""
    d, pr = case["target"]["dim"], case["prior"]
    torch.manual_seed(seed)
    if pr["kind"] == "delta":
        x0 = torch.zeros(B_FULL, d)
    else:
        x0 = pr["loc"] + pr["scale"] * torch.randn(B_FULL, d)
    state = torch.get_rng_state()
    noise = torch.stack([torch.randn_like(x0) for _ in range(T)])
    return x0, noise, state


def checksums(x0, noise):
    return dict(x0_sum=float(x0.double().sum()), noise_sum=float(noise.double().sum()), noise_abs_sum=float(noise.double().abs().sum()),
                noise_first=float(noise[0, 0, 0]), noise_last=float(noise[-1, -1, -1]))


def run(name, base, steps, seed):
    case = copy.deepcopy(mg.CASES[base])
    case["B"], case["grid"]["steps"] = B_FULL, steps
    target, prior, sde, ctrl, loss, second, ts = mg.reference_problem(case)
    T = len(ts) - 1
    x0, noise, state = inputs(case, seed, T)
    terminal = target.unnorm_log_prob
    out = {}
    t0 = time.time()
    with torch.no_grad():
        torch.set_rng_state(state)
        res = loss.eval(ts, x0, terminal, second, compute_weights=True, return_traj=False)
        train_kw = {"train": False} if case["loss"]["kind"] == "time_reversal" else {}
        torch.set_rng_state(state)
        xT, rnd, _ = loss.simulate(ts, x0, terminal, second, compute_ito_int=True, return_traj=False, **train_kw)
        assert torch.equal(res.samples, xT)
        torch.set_rng_state(state)
        res2 = loss.eval(ts, x0, terminal, second, compute_weights=False, return_traj=False)
    rows = np.linspace(0, B_FULL - 1, 64).astype(np.int64)
    out["rows"] = rows
    out["x_T"] = xT[rows].numpy()
    out["rnd"] = rnd[rows].numpy()
    out["log_norm_const_lb_ito"] = np.float64(res.log_norm_const_preds["log_norm_const_lb_ito"])
    out["log_norm_const_is"] = np.float64(res.log_norm_const_preds["log_norm_const_is"])
    out["lv_loss"] = np.float64(res.metrics["eval/lv_loss"])
    out["log_norm_const_lb"] = np.float64(res2.log_norm_const_preds["log_norm_const_lb"])
    psum = float(sum(v.double().abs().sum() for v in ctrl.state_dict().values()))
    meta = dict(name=name, base=base, seed=seed, steps=steps, B=B_FULL, T=T, case=case, param_abs_sum=psum, **checksums(x0, noise))
    out["meta"] = np.frombuffer(json.dumps(meta).encode(), dtype=np.uint8)
    path = OUT / f"{name}.npz"
    np.savez_compressed(path, **out)
    print(f"{name}: T={T} d={x0.shape[1]} logZ_is={out['log_norm_const_is']:+.6f} lb_ito={out['log_norm_const_lb_ito']:+.6f} "
          f"lb={out['log_norm_const_lb']:+.6f} lv_loss={out['lv_loss']:.6f}  {path.stat().st_size / 1024:.1f} KB  "
          f"({time.time() - t0:.0f} s of reference time)", flush=True)


if __name__ == "__main__":
    only = set(sys.argv[1:])
    for name, (base, steps, seed) in FULL.items():
        if not only or name in only:
            run(name, base, steps, seed)

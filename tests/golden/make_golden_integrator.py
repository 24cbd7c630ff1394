#!/usr/bin/env python3
"""Golden vectors for the plain Euler integrator (`int_*.npz`), produced by RUNNING THE REFERENCE's
`EulerIntegrator.integrate` (eq/integrator.py:93-127) on `LangevinSDE`, bare OU processes and `ControlledSDE`
(eq/sdes.py:38-65,272-305).  Build container only (needs /root/reference); see make_golden.py for the conventions.

Noise replay: the reference draws `torch.randn(*xs.shape) * sqrt(t - s)` once per step, so the T standard-normal tensors
are pre-drawn after snapshotting the RNG state and the state is restored before the run.
"""
from __future__ import annotations

import json
import math
import sys
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, str(Path(__file__).resolve().parent))
import make_golden as mg  # noqa: E402  (puts the reference on sys.path, with stubs for wandb/torchquad/torchsde)

from sde_sampler.eq.integrator import EulerIntegrator  # noqa: E402
from sde_sampler.eq.sdes import VP, ConstOU, ControlledSDE, LangevinSDE, ScaledBM  # noqa: E402

OUT = Path(__file__).resolve().parent

ISO = lambda d: dict(kind="iso_gauss", dim=d, loc=0.0, scale=1.0)
NET = dict(channels=64, num_layers=4, activation="gelu")

CASES = {
    # solver/langevin.py with the "fab" mixture, outputs on a sub-grid of the integration grid
    "int_langevin_gmm2": dict(
        B=64, seed=31, target=dict(kind="gmm", dim=2, name="fab"), prior=ISO(2),
        integrate=dict(kind="langevin", diff_coeff=1.0, clip_score=1e5),
        grid=dict(start=0.0, end=5.0, steps=100), ts="every:10"),
    # off-grid output times (interpolation weights strictly inside steps, two outputs inside one step), tight clip
    "int_langevin_funnel10_offgrid": dict(
        B=64, seed=37, target=dict(kind="funnel", dim=10), prior=ISO(10),
        integrate=dict(kind="langevin", diff_coeff=0.7, clip_score=5.0),
        grid=dict(start=0.0, end=2.0, steps=80), ts=[0.0, 0.013, 0.5, 0.51, 0.5125, 1.2345, 1.99, 2.0]),
    # inference process of DIS: bare VP with generative=False started at target samples (solver/oc.py:100-110)
    "int_ou_vp_inference": dict(
        B=64, seed=41, target=dict(kind="gmm", dim=2, name="fab"), prior=ISO(2), x_scale=10.0,
        sde=dict(kind="vp", beta_min=0.1, beta_max=10.0, scale=1.0, terminal_t=1.0, generative=False),
        integrate=dict(kind="controlled"), grid=dict(start=0.0, end=1.0, steps=50), ts="all"),
    # Bridge inference process: ControlledSDE(VP(generative=False), LerpPriorCtrl) (solver/oc.py:130-143)
    "int_controlled_inference_lerpprior": dict(
        B=64, seed=43, target=dict(kind="iso_gauss", dim=4, loc=3.0, scale=1.0), prior=ISO(4), x_scale=2.0,
        sde=dict(kind="vp", beta_min=0.1, beta_max=10.0, scale=1.0, terminal_t=1.0, generative=False),
        ctrl=dict(kind="lerp_prior", clip_model=1e4, clip_score=1e4, scale_score=1.0, gamma_dim=1, gamma_bias=1.0),
        net=NET, integrate=dict(kind="controlled"), grid=dict(start=0.0, end=1.0, steps=50), ts="all"),
    # generative process driven by a LerpCtrl, outputs every 5th step
    "int_controlled_generative_lerp": dict(
        B=64, seed=47, target=dict(kind="double_well", dim=1, separation=2.0, shift=1.5), prior=ISO(1),
        sde=dict(kind="vp", beta_min=0.1, beta_max=10.0, scale=1.0, terminal_t=1.0, generative=True),
        ctrl=dict(kind="lerp", clip_model=1e4, clip_score=1e4, scale_score=1.0, gamma_dim=1, gamma_bias=1.0),
        net=NET, integrate=dict(kind="controlled"), grid=dict(start=0.0, end=1.0, steps=100), ts="every:5"),
    # ControlledSDE with ctrl=None over a non-generative ConstOU
    "int_constou_none": dict(
        B=64, seed=53, target=dict(kind="multi_well", dim=5, n_double_wells=5, separation=4.0, shift=0.0), prior=ISO(5),
        x_scale=3.0, wrap=True,
        sde=dict(kind="const_ou", drift_coeff=1.0, diff_coeff=1.5, terminal_t=1.0, generative=False),
        integrate=dict(kind="controlled"), grid=dict(start=0.0, end=1.0, steps=40), ts="every:8"),
    # ScoreCtrl over a generative ScaledBM in d=50 (the headline control), outputs = endpoints + two off-grid times
    "int_controlled_score_gmm50": dict(
        B=32, seed=59, target=dict(kind="gmm", dim=50, name="fab50"), prior=ISO(50), x_scale=0.0,
        sde=dict(kind="scaled_bm", diff_coeff=math.sqrt(0.2), terminal_t=5.0, generative=True),
        ctrl=dict(kind="score", clip_model=1e4, clip_score=1e4, scale_score=1.0, gamma_dim=1, gamma_bias=0.01),
        net=NET, integrate=dict(kind="controlled"), grid=dict(start=0.0, end=5.0, steps=40), ts=[0.0, 1.0625, 3.3, 5.0]),
}


def build_sde(spec):
    gen = spec.get("generative", True)
    if spec["kind"] == "vp":
        return VP(diff_coeff_sq_min=spec["beta_min"], diff_coeff_sq_max=spec["beta_max"], scale_diff_coeff=spec["scale"],
                  terminal_t=spec["terminal_t"], generative=gen)
    if spec["kind"] == "const_ou":
        return ConstOU(drift_coeff=spec["drift_coeff"], diff_coeff=spec["diff_coeff"], terminal_t=spec["terminal_t"],
                       generative=gen)
    return ScaledBM(diff_coeff=spec["diff_coeff"], terminal_t=spec["terminal_t"], generative=gen)


def run_case(name, case):
    torch.manual_seed(1)
    target = mg.build_target(case["target"])
    prior = mg.build_prior(case["prior"])
    dim = case["target"]["dim"]
    g = case["grid"]
    timesteps = mg.get_timesteps(g["start"], g["end"], steps=g["steps"])
    if case["ts"] == "all":
        ts = timesteps.clone()
    elif isinstance(case["ts"], str):
        ts = timesteps[:: int(case["ts"].split(":")[1])].clone()
    else:
        ts = torch.tensor(case["ts"], dtype=torch.float)
    out, ctrl = {}, None
    if case["integrate"]["kind"] == "langevin":
        sde = LangevinSDE(target_score=target.score, diff_coeff=case["integrate"]["diff_coeff"],
                          clip_score=case["integrate"]["clip_score"], terminal_t=g["end"])
    else:
        sde = build_sde(case["sde"])
        if case.get("ctrl"):
            gen_sde = build_sde(dict(case["sde"], generative=True))  # the ctrl holds the solver's generative sde
            ctrl = mg.build_ctrl(case["ctrl"], case["net"], dim, gen_sde, prior, target)
            sde = ControlledSDE(sde=sde, ctrl=ctrl)
        elif case.get("wrap"):
            sde = ControlledSDE(sde=sde, ctrl=None)
    torch.manual_seed(case["seed"])
    x_init = prior.sample((case["B"],)) * case.get("x_scale", 1.0)
    state = torch.get_rng_state()
    noise = torch.stack([torch.randn(*x_init.shape) for _ in range(len(timesteps) - 1)])
    torch.set_rng_state(state)
    with torch.no_grad() if ctrl is None or case["ctrl"]["kind"] == "lerp_prior" else torch.enable_grad():
        xs = EulerIntegrator().integrate(sde, ts=ts, x_init=x_init, timesteps=timesteps)
    xs = xs.detach()
    assert xs.shape == (len(ts), *x_init.shape)
    if ctrl is not None:
        for k, v in ctrl.state_dict().items():
            out["param/" + k] = v.detach().numpy().copy()
    if case["target"]["kind"] == "gmm":
        out["target/loc"] = target.loc.numpy()
        out["target/scale"] = target.scale.numpy()
        out["target/mixture_weights"] = target.mixture_weights.numpy()
    out.update(ts=ts.numpy(), timesteps=timesteps.numpy(), x_init=x_init.numpy(), noise=noise.numpy(), xs=xs.numpy())
    meta = dict(case, name=name)
    out["meta"] = np.frombuffer(json.dumps(meta).encode(), dtype=np.uint8)
    path = OUT / f"{name}.npz"
    np.savez_compressed(path, **out)
    print(f"{name:36s} B={case['B']:3d} T={len(timesteps)-1:4d} n_out={len(ts):3d} d={dim:3d} "
          f"|x_end|={xs[-1].abs().mean():.4f} {path.stat().st_size/1024:7.1f} KB")


def main():
    torch.set_num_threads(1)
    for name, case in CASES.items():
        run_case(name, case)


if __name__ == "__main__":
    main()

"""Evaluation-kernel time of the run-time switched ("generic") trajectory variants at dimensions other than BASELINE's: the reference's
padded 40-mode mixture and a Gaussian target at d = 20 / 32 / 50 / 64 under PIS (ScoreCtrl) and DIS (LerpCtrl), B = 65 536, T = 100 --
with the reduced mixture tables ("g4" variants) and, SDEH_GENERIC_ONLY=2, with the plain generic variants (full tables).
    python tools/generic_dims_timing.py            (on the GPU box; profiles/r03_generic_dims_timing.txt)"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from sde_sampler_amd import problems  # noqa: E402

B, T = 65536, 100


def run(base, d, target, env):
    spec = problems.baseline_spec(base)
    spec["target"] = target
    spec["prior"] = dict(spec["prior"], dim=d)
    spec["grid"] = dict(spec["grid"], steps=T)
    spec["batch"] = B
    for k in ("SDEH_GENERIC_ONLY",):
        os.environ.pop(k, None)
    os.environ.update(env)
    try:
        prob = problems.build(spec, device="cuda:0")
        prob.loss.engine.timing = True
        x0 = prob.prior.sample((B,))
        ms, ms_min, n = bench.timed_kernel_ms(prob, x0)
        name = prob.loss.engine.last_kernel_name()
    finally:
        for k in env:
            os.environ.pop(k, None)
    tf = bench.algorithmic_flops(spec) * B * T / (ms * 1e-3) / 1e12
    return ms, tf, name


if __name__ == "__main__":
    print(f"# B = {B}, T = {T}; median of >= 10 launches after >= 10 warm-ups; frac = algorithmic FLOPs (SURVEY 8d) / 157.3 TFLOP/s")
    for base, label in (("gmm50_pis_headline", "PIS score"), ("cfg2_gmm2_dis_kl", "DIS lerp ")):
        for d in (20, 32, 50, 64):
            for tname, target in (("padded fab-40", dict(kind="gmm", dim=d, name="fab50")),
                                  ("gaussian     ", dict(kind="iso_gauss", dim=d, loc=1.0, scale=1.5))):
                rows = []
                envs = ({"SDEH_GENERIC_ONLY": "1"}, {"SDEH_GENERIC_ONLY": "2"}) if target["kind"] == "gmm" else ({"SDEH_GENERIC_ONLY": "1"},)
                for env in envs:
                    ms, tf, name = run(base, d, target, env)
                    rows.append(f"{name:22s} {ms:6.3f} ms  {tf / bench.PEAK_FP32_TFLOPS:5.3f}")
                print(f"{label} d={d:2d} {tname}: " + "   |   ".join(rows), flush=True)

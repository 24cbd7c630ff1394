#!/usr/bin/env python3
"""Ablation timing of the trajectory kernel (guide rule 8: ablate before optimising).

  python tools/ablate.py build            # here (no GPU): builds sde_sampler_amd/libsdeh_abl<mask>.so for every mask
  python tools/ablate.py run [--batch B]  # on the GPU box: times the headline workload with each library

Masks (sdeh_traj.hpp SDEH_ABL): 1 = no MLP, 2 = no target score, 4 = no Philox/Box-Muller, 8 = no activation.
The shipped libsdeh.so is always mask 0.
"""
import json
import os
import subprocess
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
CSRC = ROOT / "sde_sampler_amd" / "csrc"
MASKS = [0, 1, 2, 4, 8, 3, 6, 14]

CHILD = r"""
import sys, json, torch
sys.path.insert(0, %r)
from sde_sampler_amd import problems
spec = problems.baseline_spec("gmm50_pis_headline")
B = %d
prob = problems.build(spec, device="cuda:0")
x0 = prob.prior.sample((B,))
prob.loss.engine.timing = True
ms = []
for i in range(20):  # the first ~8 launches run while the GPU clock is still ramping up
    prob.eval(x0, compute_weights=False)
    ms.append(prob.loss.engine.last_kernel_ms())
print(json.dumps(sorted(ms[10:])))
"""


def build():
    for m in MASKS:
        if m == 0:
            continue
        subprocess.run(["make", "-C", str(CSRC), "-j8", f"ABL={m}", f"BUILD=build_abl{m}", f"OUT=../libsdeh_abl{m}.so"], check=True, stdout=subprocess.DEVNULL)
    print("built", [f"libsdeh_abl{m}.so" for m in MASKS if m])


def run(batch):
    base = None
    for m in MASKS:
        lib = ROOT / "sde_sampler_amd" / ("libsdeh.so" if m == 0 else f"libsdeh_abl{m}.so")
        env = dict(os.environ, SDEH_LIBRARY=str(lib))
        out = subprocess.run([sys.executable, "-c", CHILD % (str(ROOT), batch)], env=env, capture_output=True, text=True)
        if out.returncode != 0:
            print(f"mask {m:2d}: FAILED {out.stderr[-300:]}")
            continue
        ms = json.loads(out.stdout.strip().splitlines()[-1])
        med = ms[len(ms) // 2]
        base = base or med
        print(f"mask {m:2d}: kernel median {med:8.3f} ms  min {ms[0]:8.3f}  (delta vs full {base - med:+8.3f} ms)")


if __name__ == "__main__":
    if sys.argv[1] == "build":
        build()
    else:
        b = int(sys.argv[sys.argv.index("--batch") + 1]) if "--batch" in sys.argv else 65536
        run(b)

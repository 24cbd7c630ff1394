#!/usr/bin/env python3
"""Kernel-level timing of one training step on the wide networks (csrc/sdeh_wide_bwd.hip): python tools/wide_train_timing.py
<spec> <batch> [method] [steps].  Prints the forward kernel, the whole backward (HIP events around loss.backward()) and the per-kernel
split from the engine's own events where it has them."""
import sys
import os
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sde_sampler_amd import problems  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "cfg5_like_bridge196"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
method = sys.argv[3] if len(sys.argv) > 3 else "lv"
spec = problems.baseline_spec(name)
spec["batch"] = B
spec["loss"]["method"] = method
if len(sys.argv) > 4:
    spec["grid"]["steps"] = int(sys.argv[4])
prob = problems.build(spec, device="cuda:0")
eng = prob.loss.engine
eng.timing = True
inf = getattr(prob.loss, "inference_ctrl", None)
mods = [prob.ctrl] + ([inf] if inf is not None else [])
T, d, C = prob.ts.numel() - 1, spec["target"]["dim"], spec["net"]["channels"]
x0 = prob.prior.sample((B,))
for rep in range(3):
    for m in mods:
        m.zero_grad()
    torch.cuda.synchronize()
    e0, e1, e2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
    e0.record()
    val, _ = prob.loss(prob.ts, x0, prob.target.unnorm_log_prob, prob.second_log_prob)
    e1.record()
    torch.cuda.synchronize()
    fwd_k = eng.last_kernel_ms()
    val.backward()
    e2.record()
    torch.cuda.synchronize()
    last_k, last_name = eng.last_kernel_ms(), eng.last_kernel_name()
    print(f"rep {rep}: loss {val.item():.4f}  forward {e0.elapsed_time(e1):.2f} ms (kernel {fwd_k:.2f})  backward {e1.elapsed_time(e2):.2f} ms "
          f"(last kernel {last_name}: {last_k:.2f} ms)  mem {torch.cuda.max_memory_allocated() / 2**30:.1f} GiB", flush=True)
lh = spec["net"]["num_layers"] - 2
f_net = 4 * d * C + 2 * lh * C * C
rows = B * T
print(f"{name} B={B} T={T} d={d} C={C} method={method}: network pass {f_net / 1e3:.0f} kFLOP/row; first-order backward (chain + weight gradients) "
      f"2 x that = {2 * f_net * rows / 1e12:.2f} TFLOP per network; divergence backward 8 d C^2 = {8 * d * C * C * rows / 1e12:.1f} TFLOP")

#!/usr/bin/env python3
"""A/B of the pre-activation record (ABI v6: sdeh_simulate_fwd_train3 + sdeh_ctrl_backward_fused_z) against the re-evaluating fused
backward (plan option SDEH_BWD_ZREC=0): kernel times of forward and backward (events around the kernels) and the largest difference of
every parameter gradient relative to the tensor's scale, on identical Philox draws.
    python tools/zrec_ab.py [spec[:method[:B]] ...]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from sde_sampler_amd import problems  # noqa: E402

cases = sys.argv[1:] or ["cfg2_gmm2_dis_kl:kl:65536", "cfg2_gmm2_dis_kl:lv:65536", "cfg1_dw_dis_lv:lv:65536", "cfg1_dw_dis_lv:kl:65536",
                         "cfg3_gmm50_pis_kl:lv:65536", "cfg3_gmm50_pis_kl:kl:65536", "cfg4_funnel_dds_lv:lv:32768",
                         "cfg4_funnel_dds_lv:kl:32768", "cfg2_gmm2_dis_kl:lv:2048", "cfg3_gmm50_pis_kl:lv:2048"]
n_rep = int(os.environ.get("REPS", "5"))


def run(name, method, B, zrec):
    spec = problems.baseline_spec(name)
    spec["batch"] = B
    spec["loss"]["method"] = method
    if method.startswith("lv"):
        spec["loss"]["max_rnd"] = 1e8
    torch.manual_seed(3)
    prob = problems.build(spec, device="cuda:0")
    eng = prob.loss.engine
    eng.options["SDEH_BWD_ZREC"] = None if zrec else "0"
    eng.timing = True
    torch.manual_seed(5)
    x0 = prob.prior.sample((B,))
    tf, tb, names = [], [], ("", "")
    grads = None
    for rep in range(n_rep + 2):
        prob.ctrl.zero_grad()
        calls = eng.calls
        val, _ = prob.loss(prob.ts, x0, prob.target.unnorm_log_prob, prob.second_log_prob)
        torch.cuda.synchronize()
        t_f, n_f = eng.last_kernel_ms(), eng.last_kernel_name()
        val.backward()
        torch.cuda.synchronize()
        t_b, n_b = eng.last_kernel_ms(), eng.last_kernel_name()
        if rep == 0:
            grads = {k: p.grad.detach().clone() for k, p in prob.ctrl.named_parameters() if p.grad is not None}
            loss0 = float(val)
        eng.calls = calls  # the same Philox offset every repetition
        if rep >= 2:
            tf.append(t_f)
            tb.append(t_b)
        names = (n_f, n_b)
    T = prob.ts.numel() - 1
    d, c, lh = spec["target"]["dim"], spec["net"]["channels"], spec["net"]["num_layers"] - 2
    med = lambda v: sorted(v)[len(v) // 2]
    return dict(tf=med(tf), tb=med(tb), names=names, grads=grads, loss=loss0, T=T, flops=2 * (4 * d * c + 2 * lh * c * c), d=d)


for case in cases:
    name, method, B = case.split(":")
    B = int(B)
    a = run(name, method, B, True)
    b = run(name, method, B, False)
    worst = 0.0
    for k, g in a["grads"].items():
        scale = float(b["grads"][k].abs().max()) + 1e-30
        worst = max(worst, float((g - b["grads"][k]).abs().max()) / scale)
    rate = lambda r: r["flops"] * B * r["T"] / (r["tb"] * 1e-3) / 1e12 / bench.PEAK_FP32_TFLOPS
    print(f"{name} {method} B={B} T={a['T']} d={a['d']}: record fwd {a['tf']:.3f} + bwd {a['tb']:.3f} ms ({rate(a):.3f} of fp32 peak) "
          f"[{a['names'][0]} | {a['names'][1]}]  ||  re-evaluating fwd {b['tf']:.3f} + bwd {b['tb']:.3f} ms ({rate(b):.3f}) [{b['names'][1]}]"
          f"  ||  max grad diff {worst:.2e} of scale, loss diff {abs(a['loss'] - b['loss']):.2e}", flush=True)

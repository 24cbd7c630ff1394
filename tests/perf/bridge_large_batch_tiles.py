import os, sys
sys.path.insert(0, "/root/repo")
import torch
from sde_sampler_amd import problems
NET = dict(channels=64, num_layers=4, activation="gelu")
mode = os.environ.get("SDEH_BRIDGE_TILES", "default")
for d, B in [(2, 65536), (2, 131072), (2, 262144), (10, 65536), (10, 131072)]:
    tspec = dict(kind="funnel", dim=d) if d > 2 else dict(kind="gmm", dim=2, name="fab")
    lerp = dict(clip_model=1e4, clip_score=1e4, scale_score=1.0, gamma_dim=1, gamma_bias=1.0)
    spec = dict(batch=B, target=tspec, prior=dict(kind="iso_gauss", dim=d), sde=dict(kind="scaled_bm", diff_coeff=1.0, terminal_t=1.0),
                ctrl=dict(kind="lerp_target", **lerp), inference_ctrl=dict(kind="lerp_prior", **lerp),
                net=NET, loss=dict(kind="time_reversal", method="kl"), grid=dict(start=0.0, end=1.0, steps=100))
    torch.manual_seed(0)
    prob = problems.build(spec, device="cuda:0")
    x0 = prob.prior.sample((B,))
    prob.loss.engine.timing = True
    ms = []
    for i in range(8):
        r = prob.eval(x0, compute_weights=False)
        ms.append(prob.loss.engine.last_kernel_ms())
    print(f"[{mode}] d={d:2d} B={B:6d}: kernel {min(ms[3:]):8.3f} ms  lb={r.log_norm_const_preds['log_norm_const_lb']:+.5f}", flush=True)

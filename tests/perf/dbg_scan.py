import os, sys
sys.path.insert(0, "/root/repo")
import numpy as np, torch
from tests.test_hip_fuzz import random_spec
from tests.test_hip_bwd_fused import _grads
from sde_sampler_amd import problems
DEV = "cuda:0"
def run(case, method_override=None, use_noise=None, d_override=None):
    rng = np.random.default_rng(9000 + case)
    spec = random_spec(rng)
    if case % 2 == 0: spec["net"]["num_layers"] = 4
    method = str(rng.choice(["kl", "kl_ito", "lv", "lv_traj"]))
    if method_override: method = method_override
    spec["loss"]["method"] = method
    spec["loss"]["max_rnd"] = None
    spec["batch"] = int(rng.choice([33, 64, 100, 257]))
    prob = problems.build(spec); prob.to(DEV)
    B, d, T = spec["batch"], spec["target"]["dim"], prob.ts.numel() - 1
    torch.manual_seed(case)
    x0 = prob.prior.sample((B,)).to(DEV)
    noise = torch.randn(T, B, d, device=DEV) if use_noise else None
    eng = prob.loss.engine; calls = eng.calls
    out = {}
    for scan in ("1", "0"):
        os.environ["SDEH_BWD_SCAN"] = scan
        eng.calls = calls
        v, g, name = _grads(prob, x0, noise, planes=False)
        out[scan] = (v, g, name)
    gmax = max(g.abs().max().item() for g in out["0"][1].values() if g is not None)
    worst = max(((out["1"][1][k] - out["0"][1][k]).abs().max().item() / max(out["0"][1][k].abs().max().item(), 1e-3 * gmax), k) for k in out["0"][1] if out["0"][1][k] is not None)
    print(f"case {case} {method} {spec['loss']['kind']}/{spec['ctrl']['kind']}/{spec['target']['kind']} d={d} B={B} T={T} noise={'given' if use_noise else 'replayed'} clip={spec['ctrl'].get('clip_model')} act={spec['net'].get('activation')}: {out['1'][2]} vs {out['0'][2]} worst {worst[0]:.2e} {worst[1]}")
for case in (6,):
    for m in ("kl_ito", "kl"):
        for n in (False, True):
            run(case, m, n)
for case in range(0, 40, 2):
    try: run(case, "kl_ito", False)
    except Exception as e: print(case, type(e).__name__, str(e)[:80])

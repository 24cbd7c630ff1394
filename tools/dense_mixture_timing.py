"""Kernel time of the trajectory kernel on mixtures whose tables cover all 50 coordinates (bench.py's extra block: gmm50_dense_shared,
gmm50_dense_general, the headline on the generic kernel with full tables) + a hash of the outputs, so that two builds of the library
(tables through the scalar cache / as LDS broadcast reads: -DSDEH_GMM_SGPR=0) can be compared bit for bit on identical Philox draws.
  python tools/dense_mixture_timing.py [B]            SDEH_LIBRARY=<other build> python tools/dense_mixture_timing.py [B]
"""
import hashlib
import os
import statistics
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sde_sampler_amd import problems  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
REPS = int(os.environ.get("REPS", "7"))
print(f"# library: {os.environ.get('SDEH_LIBRARY', 'sde_sampler_amd/libsdeh.so')}  B = {B}")
for name, gen in (("gmm50_pis_headline", None), ("gmm50_dense_shared", None), ("gmm50_dense_general", None), ("gmm50_pis_headline", "2"),
                  ("cfg3_gmm50_pis_kl", None), ("cfg2_gmm2_dis_kl", None), ("cfg4_funnel_dds_lv", None)):
    spec = problems.baseline_spec(name)
    spec["batch"] = B if name not in ("cfg3_gmm50_pis_kl", "cfg4_funnel_dds_lv") or len(sys.argv) > 1 else spec["batch"]  # (configs[2] / [3]: the per-GPU shards of 32 768)
    if gen is not None:
        os.environ["SDEH_GENERIC_ONLY"] = gen
    try:
        prob = problems.build(spec, device="cuda:0")
        prob.loss.engine.timing = True
        torch.manual_seed(3)
        x0 = prob.prior.sample((spec["batch"],))
        ms = []
        for i in range(REPS + 2):
            r = prob.eval(x0, compute_weights=True, return_traj=False)  # Philox seed = torch's initial seed, offset = the call count
            if i >= 2:
                ms.append(prob.loss.engine.last_kernel_ms())
        torch.cuda.synchronize()
        h = hashlib.sha256(r.samples.cpu().numpy().tobytes()).hexdigest()[:12]
        lz = r.log_norm_const_preds
        T = prob.ts.numel() - 1
        print(f"{name + ('/generic' if gen else ''):34s} T={T:4d} kernel ms median {statistics.median(ms):.3f} min {min(ms):.3f}  x_T sha {h}  "
              f"log_Z_is {lz.get('log_norm_const_is', float('nan')):.6f}  {prob.loss.engine.last_kernel_name()}")
    finally:
        os.environ.pop("SDEH_GENERIC_ONLY", None)
    del prob, x0

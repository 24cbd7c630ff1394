import os, sys, torch
sys.path.insert(0, "/root/repo")
import bench
from sde_sampler_amd import problems
for name, B in (("cfg2_gmm2_dis_kl", 65536), ("cfg2_gmm2_dis_kl", 32768), ("cfg2_gmm2_dis_kl", 20000), ("cfg1_dw_dis_lv", 65536), ("cfg1_dw_dis_lv", 32768)):
    row = []
    for v in ("0", "1"):
        os.environ["SDEH_WS_VOUT"] = v
        spec = problems.baseline_spec(name); spec["batch"] = B
        prob = problems.build(spec, device="cuda:0"); prob.loss.engine.timing = True
        torch.manual_seed(0)
        x0 = prob.prior.sample((B,))
        ms, mn, n = bench.timed_kernel_ms(prob, x0)
        prob.loss.engine.calls = 3
        r = prob.eval(x0, compute_weights=False, return_traj=False)
        row.append(f"VOUT={v}: {ms:.3f} ms lb={r.log_norm_const_preds['log_norm_const_lb']:.5f} {prob.loss.engine.last_kernel_name()}")
    print(name, B, " | ".join(row), flush=True)

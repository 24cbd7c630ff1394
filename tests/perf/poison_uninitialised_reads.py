"""Diagnostic: fills the caching allocator with NaN / 1e30 / -3 before a wide training step and compares every gradient bitwise with a clean run -- an uninitialised read of a work buffer shows up as a difference (python tests/perf/poison_uninitialised_reads.py [wide_bridge|wide_lv|wide_kl] [batch])."""
import sys, torch
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import test_hip_graphs as G
which = sys.argv[1] if len(sys.argv) > 1 else "wide_bridge"
prob = G._build(4, which); params = G._params(prob); lo = prob.loss
inf = getattr(lo, "inference_ctrl", None)
names = [n for n, _ in prob.ctrl.named_parameters()] + (["inf." + n for n, _ in inf.named_parameters()] if inf is not None else [])
B = int(sys.argv[2]) if len(sys.argv) > 2 else 256
x = prob.prior.sample((B,))
def run():
    for p in params: p.grad = None
    lo.engine.calls = 5
    v = lo(prob.ts, x, prob.target.unnorm_log_prob, prob.second_log_prob)[0]; v.backward(); return v
run(); ref = [p.grad.clone() for p in params]
for val in (float("nan"), 1e30, -3.0):
    torch.cuda.empty_cache()
    big = [torch.full((1 << 28,), val, device="cuda:0") for _ in range(8)]  # 8 GiB of poison into the caching allocator
    del big
    v = run(); torch.cuda.synchronize()
    bad = [(n, float((p.grad - g).abs().max())) for n, p, g in zip(names, params, ref) if not torch.equal(p.grad, g)]
    print("poison", val, "loss", float(v), "params that differ:", bad[:8], len(bad))

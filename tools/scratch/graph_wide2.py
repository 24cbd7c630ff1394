import sys, torch
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import test_hip_graphs as G
prob = G._build(4, "wide_bridge"); params = G._params(prob); lo = prob.loss
inf = lo.inference_ctrl
names = [n for n, _ in prob.ctrl.named_parameters()] + ["inf." + n for n, _ in inf.named_parameters()]
x = prob.prior.sample((256,)); lo.graph_safe = True; lo.rng_counter = torch.zeros(1, dtype=torch.int64, device="cuda:0")
def run():
    for p in params: p.grad = None
    lo.engine.calls = 5
    v = lo(prob.ts, x, prob.target.unnorm_log_prob, prob.second_log_prob)[0]; v.backward(); return v
def cmp(tag, ref):
    bad = [(n, float((p.grad - g).abs().max()), float(g.abs().max())) for n, p, g in zip(names, params, ref) if float((p.grad - g).abs().max()) > 1e-5 * float(g.abs().max()) + 1e-12]
    print(tag, "differing params:", bad[:6], len(bad))
run(); ref = [p.grad.clone() for p in params]
for i in range(3):
    run(); torch.cuda.synchronize(); cmp(f"eager {i}", ref)
side = torch.cuda.Stream(); side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side): run()
torch.cuda.current_stream().wait_stream(side); torch.cuda.synchronize(); cmp("side stream", ref)
for p in params: p.grad = None
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g): v = run()
for i in range(4):
    g.replay(); torch.cuda.synchronize(); cmp(f"replay {i}", ref)

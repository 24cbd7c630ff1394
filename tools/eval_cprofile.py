import cProfile, pstats, sys, torch
sys.path.insert(0, "/root/repo")
from sde_sampler_amd import problems
spec = problems.baseline_spec("gmm50_pis_headline"); spec["batch"] = 1024
prob = problems.build(spec, device="cuda:0")
x0 = prob.prior.sample((1024,))
for _ in range(30): prob.eval(x0, compute_weights=False, return_traj=False)
cProfile.run("for _ in range(500): prob.eval(x0, compute_weights=False, return_traj=False)", "/tmp/prof")
pstats.Stats("/tmp/prof").sort_stats("tottime").print_stats(22)

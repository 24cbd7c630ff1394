"""Where the host time of one evaluation call goes (the headline workload's `sample_time` step, bench.py): wall time per call against
the trajectory kernel's own duration, and the host-side pieces in between (problem description, launch, reduction + the 8-float
device->host copy the estimators need).  Usage: python tools/eval_host_profile.py [workload spec name] [batch]"""
import sys
import time
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))

import torch

from sde_sampler_amd import engine as E
from sde_sampler_amd import problems


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else "gmm50_pis_headline"
    spec = problems.baseline_spec(name)
    if len(sys.argv) > 2:
        spec["batch"] = int(sys.argv[2])
    prob = problems.build(spec, device="cuda:0")
    torch.manual_seed(1)
    x0 = prob.prior.sample((spec["batch"],))
    eng = prob.loss.engine
    acc = {}

    def timed(obj, attr, label):
        fn = getattr(obj, attr)

        def wrapper(*a, **k):
            t0 = time.perf_counter()
            try:
                return fn(*a, **k)
            finally:
                acc[label] = acc.get(label, 0.0) + time.perf_counter() - t0
        setattr(obj, attr, wrapper)

    timed(eng, "build_problem", "build_problem")
    timed(eng, "run", "engine.run (allocations + prep/trajectory launches)")
    timed(E, "estimator_stats", "estimator_stats (reduction launches)")
    timed(E, "merge_stats", "merge_stats (device->host copy = the sync)")
    for timing in (False, True):
        eng.timing = timing
        for _ in range(20):
            prob.eval(x0, compute_weights=False, return_traj=False)
        torch.cuda.synchronize()
        acc.clear()
        n, k_ms = 300, []
        t0 = time.perf_counter()
        for _ in range(n):
            prob.eval(x0, compute_weights=False, return_traj=False)
            if timing:
                k_ms.append(eng.last_kernel_ms())
        torch.cuda.synchronize()
        wall = (time.perf_counter() - t0) / n * 1e3
        print(f"{name} B={spec['batch']} timing events {'on' if timing else 'off'}: {wall:.4f} ms per call"
              + (f", trajectory kernel {sum(k_ms) / n:.4f} ms" if timing else ""))
        for label, v in acc.items():
            print(f"    {label:60s} {v / n * 1e6:8.1f} us per call")
    # the same call replayed as one graph (utils.graphs.GraphedEval)
    from sde_sampler_amd.utils.graphs import GraphedEval

    eng.timing = False
    ge = GraphedEval(lambda x: prob.eval(x, compute_weights=False, return_traj=False), [prob.loss], x0)
    for _ in range(20):
        ge()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(300):
        ge()
    torch.cuda.synchronize()
    print(f"{name} B={spec['batch']} replayed as one hipGraph (GraphedEval): {(time.perf_counter() - t0) / 300 * 1e3:.4f} ms per call")
    eng.timing = True
    try:
        ge2 = GraphedEval(lambda x: prob.eval(x, compute_weights=False, return_traj=False), [prob.loss], x0)
        ge2(); ge2()
        print(f"    kernel events inside the graph: last_kernel_ms = {eng.last_kernel_ms():.4f}")
    except Exception as exc:  # noqa: BLE001
        print(f"    kernel events inside the graph: {type(exc).__name__}: {str(exc)[:200]}")


if __name__ == "__main__":
    main()

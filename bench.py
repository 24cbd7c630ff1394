#!/usr/bin/env python3
"""Benchmark of the hot path: trajectory-steps/s of the fused Euler-Maruyama engine on MI355X.

    python bench.py --gpus N --steps K --warmup W

N > 1 works both ways: launched by `python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N` (the ranks read
RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* from the environment) or invoked plainly -- then this script spawns the N ranks itself
(one process per GPU, RCCL) and errors out when the node has fewer GPUs than ranks.

Workload at N = 1 (BASELINE.json metric "trajectory-steps/sec ... GMM-40 d=50"): target = 40-mode GMM in d=50 (fab means in the
first two coordinates, scale softplus(1)), solver basic_pis (ScoreCtrl + FourierMLP C=64 / 4 layers GELU, Delta prior,
ScaledBM(sqrt 0.2, T=5)), batch 65 536 trajectories PER GPU, T = 100 steps, fp32, in-kernel Philox noise, random-init weights
(last layers N(0, 0.05^2)), x0 resident in HBM.  One "step" = one pass of the hot path over the batch with the reference's
`eval/sample_time` semantics (solver/oc.py:88-97): loss.eval(ts, x, ..., compute_weights=False, return_traj=False) under
no_grad, i.e. the trajectory kernel + the log-Z lower-bound reduction (+ the 8-float all-gather when N > 1).
`--workload` selects the other measured configurations (see WORKLOADS).  Prints ONE JSON line (rank 0).
"""
from __future__ import annotations

import argparse
import json
import math
import os
import statistics
import sys
import time
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

PEAK_FP32_TFLOPS = 157.3  # MI355X fp32 MFMA (= fp32 vector) dense peak, /opt/skills/guides/MI355X_MICROARCH.md

#: name -> (problems.py spec name, metric label, description).  The first entry is the metric's configuration.
WORKLOADS = {
    "gmm50_pis_headline": ("gmm50_pis_headline", "trajectory-steps/sec (batch x steps / s), GMM-40 d=50",
                           "GMM-40 d=50 (explicit loc/scale), basic_pis (ScoreCtrl, FourierMLP C=64 L=4 GELU, Delta prior, "
                           "ScaledBM sqrt(0.2) T=5), eval sample_time semantics"),
    "gmm50_dense_shared": ("gmm50_dense_shared", "trajectory-steps/sec, GMM-40 d=50 with means varying in ALL coordinates",
                           "as the headline, but loc ~ U(-40, 40) in every coordinate (shared scale softplus(1))"),
    "gmm50_dense_general": ("gmm50_dense_general", "trajectory-steps/sec, GMM-40 d=50 dense means, per-component scales",
                            "as the headline, but loc ~ U(-40, 40) and scale ~ U(1, 1.5) per (component, coordinate)"),
    "wide_pis_funnel196": ("wide_pis_funnel196", "trajectory-steps/sec, funnel d=196, FourierMLP C=256",
                           "funnel d=196, basic_pis-style (ScoreCtrl, FourierMLP C=256 L=4 GELU, ScaledBM), wide-network kernel"),
    "cfg5_like_bridge196": ("cfg5_like_bridge196", "trajectory-steps/sec, Bridge d=196 C=256 (BASELINE configs[4] shape)",
                            "Bridge (LerpTargetCtrl + LerpPriorCtrl inference control, exact divergence), funnel d=196 target "
                            "in place of the NICE flow (a closed-form target: the whole grid in ONE launch; the flow itself: cfg5_nice_bridge196), two FourierMLP C=256 L=4 GELU, ScaledBM(1, T=1)"),
    # BASELINE configs[4] AS WRITTEN: target = nice.  The flow's score is evaluated by csrc/sdeh_nice.hip between the one-step segments of
    # the wide Bridge kernel (engine.run: SDEH_DENS_EXTERNAL); weights: the checkpoint's geometry, seeded random (data/nice.pt is not shipped)
    "cfg5_nice_bridge196": ("cfg5_nice_bridge196", "trajectory-steps/sec, Bridge d=196 C=256 on the NICE flow (BASELINE configs[4] as written)",
                            "Bridge (LerpTargetCtrl on the NICE flow's score + LerpPriorCtrl inference control, exact divergence), NiceModel "
                            "coupling=4 mid_dim=500 hidden=5 (scripts/train_nice.py geometry, seeded random weights), two FourierMLP C=256 "
                            "L=4 GELU, ScaledBM(1, T=1)"),
    "train_cfg5_nice": ("cfg5_nice_bridge196", "trajectory-steps/sec of one optimisation step (forward + backward + Adam), Bridge lv on the NICE flow, d=196 C=256",
                        "BASELINE configs[4] as written: conf/solver/bridge.yaml's loss (time_reversal_lv, exact divergence) on the NICE flow, "
                        "T=200, batch 4096 per GPU (32 768 / 8)"),
    # one optimisation step (loss forward, backward, Adam) -- the reference's Trainable.step (solver/base.py:399-454) on the HIP path
    "train_gmm2_dis_kl": ("cfg2_gmm2_dis_kl", "trajectory-steps/sec of one optimisation step (forward + backward + Adam), DIS kl, GMM-40 d=2",
                          "BASELINE configs[1]: GMM-40 d=2, basic_dis (LerpCtrl, FourierMLP C=64 L=4 GELU, VP), loss.method=kl, "
                          "one training step per bench step"),
    "train_gmm50_pis_kl": ("cfg3_gmm50_pis_kl", "trajectory-steps/sec of one optimisation step (forward + backward + Adam), PIS kl, GMM-40 d=50",
                           "BASELINE configs[2]'s shape: GMM-40 d=50, basic_pis, loss.method=kl, T=200, one training step per bench step"),
    # training on the wide networks (csrc/sdeh_wide_bwd.hip): configs[4]'s shape as the reference trains it (conf/solver/bridge.yaml:
    # two FourierMLP C=256, loss time_reversal_lv, exact divergence) at its per-GPU batch, and a plain PIS-style network of that width
    "train_cfg5_like": ("cfg5_like_bridge196", "trajectory-steps/sec of one optimisation step (forward + backward + Adam), Bridge lv, d=196 C=256",
                        "BASELINE configs[4]'s shape: Bridge (LerpTargetCtrl + LerpPriorCtrl, exact divergence) on a funnel d=196 in place "
                        "of the NICE flow, two FourierMLP C=256 L=4 GELU, loss.method=lv, T=200, batch 4096 per GPU (32 768 / 8)"),
    "train_wide_pis_lv": ("wide_pis_funnel196", "trajectory-steps/sec of one optimisation step (forward + backward + Adam), PIS lv, d=196 C=256",
                          "funnel d=196, basic_pis-style, FourierMLP C=256 L=4 GELU, loss.method=lv, T=200, batch 8192"),
}
#: default batch / loss method of the wide training workloads (the others: 65 536 and the spec's method)
TRAIN_DEFAULTS = {"train_cfg5_like": (4096, "lv"), "train_wide_pis_lv": (8192, "lv"), "train_cfg5_nice": (4096, "lv")}


#: the translation-unit sources of the headline trajectory kernel: `profiles/pmc_headline.json` is stamped with their hash, and its
#: counters (HBM traffic, executed instructions) are only reported while the kernel they were taken on is the kernel that runs
HEADLINE_KERNEL_SOURCES = ("sdeh_traj_ws.hpp", "sdeh_variants.inc", "sdeh_traj_inst.hip", "sdeh_traj.hpp", "sdeh_common.hpp", "Makefile")


def headline_kernel_sha() -> str:
    import hashlib

    h = hashlib.sha256()
    for name in HEADLINE_KERNEL_SOURCES:
        h.update((ROOT / "sde_sampler_amd" / "csrc" / name).read_bytes())
    return h.hexdigest()[:16]


def pmc_record():
    """Per-launch PMC counters of the headline trajectory kernel from the committed rocprofv3 passes (tools/pmc_profile.sh ->
    tools/pmc_headline_json.py), or None when they were taken on other kernel sources than the ones in this tree."""
    try:
        rec = json.loads((ROOT / "profiles" / "pmc_headline.json").read_text())
    except (OSError, ValueError):
        return None
    return rec if rec.get("kernel_sha") == headline_kernel_sha() else None


def flops_per_traj_step(d: int, c: int, lh: int, k: int) -> float:
    """F(d,C,Lh,K) = 4dC + 2 Lh C^2 (MLP, time embedding hoisted) + 6dK + 4K (GMM score) + 20d."""
    return 4 * d * c + 2 * lh * c * c + 6 * d * k + 4 * k + 20 * d


def algorithmic_flops(spec: dict) -> float:
    d, c, lh = spec["target"]["dim"], spec["net"]["channels"], spec["net"]["num_layers"] - 2
    k = 40 if spec["target"]["kind"] == "gmm" else 0
    f = flops_per_traj_step(d, c, lh, k)
    if spec.get("inference_ctrl"):
        # Bridge (losses/oc.py:189-202): a second network pass + the exact divergence of the inference control.  Minimal
        # formulation: per coordinate two C x C products (forward tangent through hidden layer 1, adjoint through
        # hidden layer 2) and a length-C dot product -- the reference's d backward passes cost about twice that.
        ci, lhi = spec.get("inference_net", spec["net"])["channels"], spec.get("inference_net", spec["net"])["num_layers"] - 2
        f += 4 * d * ci + 2 * lhi * ci * ci + d * (2 * lhi * ci * ci + 2 * ci)
    f += nice_flops(spec["target"])
    return f


def nice_flops(tspec: dict) -> float:
    """The NICE flow's score per trajectory-step (distr/nice.py): every coupling's MLP forward and its reverse pass (d / d x only: the
    weights are constants), 2 FLOPs per multiply-add: 2 x 2 x coupling x (2 (d / 2) mid + (hidden - 1) mid^2)."""
    if tspec.get("kind") != "nice":
        return 0.0
    half, mid, hidden, n_c = tspec["dim"] // 2, tspec.get("mid_dim", 500), tspec.get("hidden", 5), tspec.get("coupling", 4)
    return 4.0 * n_c * (2 * half * mid + (hidden - 1) * mid * mid)


def target_state(spec: dict, prob):
    """What the CPU oracle needs of a target besides its spec: the NICE flow's weights (oracle/em_oracle.py Density("nice"))."""
    if spec["target"]["kind"] == "nice":
        return {k: v.detach().cpu().clone() for k, v in prob.target.model.state_dict().items()}
    return None


def physical_cores() -> int:
    try:
        import psutil

        return psutil.cpu_count(logical=False) or (os.cpu_count() or 1)
    except Exception:  # pragma: no cover
        return os.cpu_count() or 1


def usable_cpus() -> int:
    """CPUs this process may actually run on: the scheduler affinity and the cgroup CPU quota (a container on a 128-core host is often
    given a few cores' worth of time: filling the host with 128 processes then measures the scheduler, not the reference)."""
    n = physical_cores()
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except (AttributeError, OSError):  # pragma: no cover
        pass
    for path, parse in (("/sys/fs/cgroup/cpu.max", lambda t: (t.split()[0], t.split()[1])),
                        ("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", lambda t: (t.strip(), None))):
        try:
            quota, period = parse(open(path).read())
            if period is None:
                period = open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read().strip()
            if quota not in ("max", "-1") and int(quota) > 0:
                n = min(n, max(1, int(int(quota) / int(period))))
            break
        except (OSError, ValueError, IndexError):
            continue
    return max(1, n)


def cpu_process_parallel(spec, prob_cpu_state, chunk: int, n_intervals: int | None, T: int, window: float = 8.0) -> dict:
    """The honest all-cores figure of the reference's CPU path: its per-step tensors are too small for intra-op
    threads (128 torch threads are SLOWER than one), so the host is filled the way a user would fill it -- N independent one-thread
    processes (oracle/cpu_worker.py), N = the cores this process may use (`usable_cpus`: affinity and cgroup quota), each integrating its own chunk of trajectories (`sample_time` semantics,
    solver/oc.py:88-97).  Aggregate rate = chunks finished by all processes inside one common time window x chunk x T / window.
    Plain subprocesses with a hard timeout: a host that cannot run them is reported, never waited for."""
    import pickle
    import subprocess
    import tempfile

    n = int(os.environ.get("SDEH_BENCH_CPU_PROCS", usable_cpus()))  # (round 5: the cores this container may use, not the host's 128)
    try:  # an interpreter with torch loaded is ~0.5 GB resident: never more processes than the host's free memory holds three times over
        import psutil

        n = max(1, min(n, int(psutil.virtual_memory().available / 1.5e9)))
    except Exception:  # pragma: no cover
        pass
    params, tt, params_inf = prob_cpu_state
    lead = 15.0 + 0.2 * n  # interpreter + torch import + warm-up chunk of every process before the window opens
    root = os.path.dirname(os.path.abspath(__file__))
    env = dict(os.environ, OMP_NUM_THREADS="1", MKL_NUM_THREADS="1", HIP_VISIBLE_DEVICES="", PYTHONPATH=root + os.pathsep + os.environ.get("PYTHONPATH", ""))
    procs, chunks, late, failed = [], 0, 0, 0
    with tempfile.TemporaryDirectory() as tmp:
        t_start = time.time() + lead
        job = os.path.join(tmp, "job.pkl")
        for i in range(n):
            with open(f"{job}.{i}", "wb") as fh:
                pickle.dump((spec, params, tt, params_inf, chunk, n_intervals, t_start, window, 100 + i), fh)
            procs.append(subprocess.Popen([sys.executable, "-m", "oracle.cpu_worker", f"{job}.{i}"], cwd=root, env=env,
                                          stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True))
        deadline = t_start + window + 60.0
        for p in procs:
            try:
                out, _ = p.communicate(timeout=max(1.0, deadline - time.time()))
                done, ready = out.split()
                chunks += int(done)
                late += 1 - int(ready)
            except Exception:  # noqa: BLE001  (timeout, crash, malformed line: that process contributes nothing)
                p.kill()
                failed += 1
    return {"value": chunks * chunk * T / window if chunks else None, "processes": n, "usable_cpus": usable_cpus(), "threads_per_process": 1, "chunk": chunk,
            "window_s": window, "chunks_finished": chunks, "processes_late_for_the_window": late, "processes_failed": failed}


def cpu_baseline(spec, prob_cpu_state, budget_s: float = 30.0, parity: dict | None = None, train_method: str | None = None) -> dict:
    """The reference's CPU path (oracle = op-for-op PyTorch-CPU restatement, bit-exact on the reference-generated fixtures) on a
    bounded sample of the same workload, SURVEY.md 8d: torch.set_num_threads(all physical cores) AND one thread, median of >= 5
    full-T chunks after one warm-up chunk, `sample_time` semantics (compute_weights=False, no trajectory)."""
    from oracle import em_oracle as eo

    params, tt, params_inf = prob_cpu_state
    if train_method is not None:  # the training workloads: the oracle's loss + autograd backward on a bounded sample
        leaf = lambda sd: {k: v.detach().clone().requires_grad_(v.is_floating_point()) for k, v in sd.items()}
        leaves = leaf(params)
        leaves_inf = leaf(params_inf) if params_inf is not None else None
        oracle = eo.Problem(spec, leaves, tt, params_inf=leaves_inf)
        ts = oracle.grid()
        bridge = bool(spec.get("inference_ctrl"))
        if bridge:  # d backward passes per step, differentiated again: the first 4 intervals of the grid, 32 rows
            ts = ts[:5]
        wide = spec["net"]["channels"] > 64
        T, d, rows = ts.numel() - 1, spec["target"]["dim"], (32 if bridge else (64 if wide else 256))
        # thread counts swept like the evaluation leg's (one thread is often the fastest setting for these small per-step tensors)
        by_threads = {}
        for threads in sorted({1, min(8, physical_cores()), min(32, physical_cores())}):
            torch.set_num_threads(threads)
            rates = []
            for _ in range(4):
                x0c = torch.zeros(rows, d) if spec["prior"]["kind"] == "delta" else torch.randn(rows, d)
                for v in list(leaves.values()) + (list(leaves_inf.values()) if leaves_inf else []):
                    v.grad = None
                t0 = time.perf_counter()
                l_ref, _, _, _ = oracle.train_loss(ts, x0c, None, method=train_method)
                l_ref.backward()
                rates.append(rows * T / (time.perf_counter() - t0))
            by_threads[threads] = statistics.median(rates[1:])
        best = max(by_threads, key=by_threads.get)
        return {"value": by_threads[best], "unit": "trajectory-steps/s", "cores": best, "kind": "port",
                "by_threads": {str(k): v for k, v in by_threads.items()},
                "sample": f"oracle/em_oracle.py train_loss + autograd backward (no optimizer step), {rows} trajectories x T={T}, "
                          f"median of 3 after 1 warm-up per thread count {sorted(by_threads)}; `value` = the best ({best} threads)"}
    oracle = eo.Problem(spec, params, tt, params_inf=params_inf)
    ts = oracle.grid()
    bridge = bool(spec.get("inference_ctrl"))
    if bridge:  # the exact divergence costs the reference d backward passes per step: sample the first 8 intervals of the grid
        ts = ts[:9]
    T, d = ts.numel() - 1, spec["target"]["dim"]
    cores = physical_cores()

    def timed(threads: int, chunk: int, budget: float):
        torch.set_num_threads(threads)
        torch.manual_seed(7)
        x0 = torch.zeros(chunk, d) if spec["prior"]["kind"] == "delta" else torch.randn(chunk, d)
        t_warm = time.perf_counter()
        oracle.eval(ts, x0, None, compute_weights=False)  # warm-up chunk
        n_min = 5 if time.perf_counter() - t_warm < 4.0 else 3  # a thread count that is this slow is not the one reported
        rates, lbs, t_begin = [], [], time.perf_counter()
        while len(rates) < n_min or (time.perf_counter() - t_begin < budget and len(rates) < 7):
            t0 = time.perf_counter()
            res = oracle.eval(ts, x0, None, compute_weights=False)
            rates.append(chunk * T / (time.perf_counter() - t0))
            lbs.append(res["log_norm_const_lb"])
        return statistics.median(rates), len(rates), sum(lbs) / len(lbs), time.perf_counter() - t_begin

    scale = max(1.0, algorithmic_flops(spec) / 42344.0)  # keep the CPU work bounded for the heavier workloads
    steps_scale = 100.0 / T
    chunk_all = max(32, int(4096 * steps_scale / scale))
    chunk_one = max(8, int(256 * steps_scale / scale))
    if bridge:
        chunk_all, chunk_one = 32, 8
    # thread counts: all physical cores, one thread, and two in-between settings -- on many-core hosts the per-op
    # synchronisation of 100+ threads makes "all cores" the SLOWEST configuration for these small per-step tensors
    by_threads = {}
    for threads in sorted({cores, 1, min(cores, 8), min(cores, 32)}):
        chunk = chunk_one if threads == 1 else chunk_all
        rate, n, lb, secs = timed(threads, chunk, budget_s / 4.0)
        by_threads[threads] = dict(rate=rate, n=n, lb=lb, secs=secs, chunk=chunk)
    best = max(by_threads, key=lambda k: by_threads[k]["rate"])
    rate_all, n_all, lb_all, s_all = (by_threads[cores][k] for k in ("rate", "n", "lb", "secs"))
    rate_one, n_one, lb_one, s_one = (by_threads[1][k] for k in ("rate", "n", "lb", "secs"))
    torch.set_num_threads(cores)
    parity_out = None
    if parity is not None:  # the trained control on the GPU leg's x0 / noise: what the reference's CPU path computes for them
        ref = eo.Problem(spec, parity["params"], tt).eval(parity["ts"], parity["x0"], parity["noise"], compute_weights=True)
        parity_out = {"cpu_log_norm_const_is": ref["log_norm_const_is"], "cpu_log_norm_const_lb_ito": ref["log_norm_const_lb_ito"],
                      "delta_vs_cpu": parity["gpu_is"] - ref["log_norm_const_is"],
                      "delta_lb_ito_vs_cpu": parity["gpu_lb_ito"] - ref["log_norm_const_lb_ito"]}
    # all cores the way they can be used: N one-thread processes over disjoint chunks (the intra-op thread sweep above does not scale)
    par = cpu_process_parallel(spec, prob_cpu_state, chunk_one, 8 if bridge else None, T)
    value, used = by_threads[best]["rate"], best
    if par.get("value") and par["value"] > value:
        value, used = par["value"], par["processes"]
    return {"parity_log_z": parity_out, "value": value, "unit": "trajectory-steps/s", "cores": used, "kind": "port",
            "value_best_thread_count": by_threads[best]["rate"], "best_thread_count": best,
            "value_all_cores_process_parallel": par.get("value"), "process_parallel": par,
            "value_all_physical_cores": rate_all, "physical_cores": cores, "value_1_thread": rate_one,
            "by_threads": {str(k): v["rate"] for k, v in by_threads.items()},
            "sample": f"`value` = the larger of (a) {par.get('processes')} one-thread PROCESSES side by side, each integrating chunks of "
                      f"{par.get('chunk')} trajectories x T={T} for a common {par.get('window_s')} s window ({par.get('chunks_finished')} chunks "
                      f"finished), and (b) the best intra-op thread count; (b): "
                      f"oracle/em_oracle.py (PyTorch-CPU restatement of the reference loop, torch.randn noise), same workload"
                      f"{' (first 8 of the grid intervals: exact divergence by d backward passes per step)' if bridge else ''}; `value` = "
                      f"the best of the thread counts tried ({best} threads); {cores} torch threads (= physical cores; "
                      f"{os.cpu_count()} hardware threads): median of {n_all} chunks of {chunk_all} trajectories x T={T} after 1 "
                      f"warm-up ({s_all:.1f} s, log_norm_const_lb={lb_all:.4f}) -> {rate_all:.3e}; 1 thread: median of {n_one} chunks "
                      f"of {chunk_one} ({s_one:.1f} s) -> {rate_one:.3e} trajectory-steps/s"}


def timed_kernel_ms(prob, x0, n_warm: int = 10, n: int = 10, budget_s: float = 4.0) -> tuple[float, float, int]:
    """(median, min, launches) of the trajectory kernel's duration: >= 10 untimed launches first (the first ~8 after idle run while
    the clock ramps up), then >= 10 timed ones -- fewer only for kernels that take longer than budget_s / 10."""
    t0 = time.perf_counter()
    for i in range(n_warm):
        prob.eval(x0, compute_weights=False, return_traj=False)
        torch.cuda.synchronize()
        if i >= 1 and time.perf_counter() - t0 > budget_s:
            break
    ms, t0 = [], time.perf_counter()
    for i in range(n):
        prob.eval(x0, compute_weights=False, return_traj=False)
        ms.append(prob.loss.engine.last_kernel_ms())
        if i >= 2 and time.perf_counter() - t0 > budget_s:
            break
    return statistics.median(ms), min(ms), len(ms)


def extra_block(device, B: int) -> dict:
    """Kernel time of the workloads the headline's specialisations do not apply to, same B and T: dense-mean
    mixtures (no varying-prefix shortcut), and the headline forced onto the generic run-time-switched kernel."""
    from sde_sampler_amd import problems

    out = {}
    # (headline_generic_kernel: run-time switches for loss / control / target / activation, mixture tables over the four coordinates the
    # reference's padded mixtures differ in -- variant "50_0_g4"; ..._full_tables: the plain generic variant, tables over all 50)
    # (the dense mixtures run their contractions on the matrix pipe where the binding vouches for the product form of the logits,
    # engine._mixture_mm_ok -- both bench mixtures qualify; "..._exact_form": plan option SDEH_GMM_MM=0, squared-distance logits on
    # the vector pipe with the tables streamed through the scalar cache)
    for name, gen in (("gmm50_dense_shared", None), ("gmm50_dense_general", None), ("gmm50_dense_shared", "x"), ("gmm50_dense_general", "x"),
                      ("gmm50_pis_headline", "1"), ("gmm50_pis_headline", "2")):
        spec = problems.baseline_spec(name)
        spec["batch"] = B
        exact = gen == "x"
        generic = gen is not None and not exact
        if generic:
            os.environ["SDEH_GENERIC_ONLY"] = gen
        if exact:
            os.environ["SDEH_GMM_MM"] = "0"
        try:
            prob = problems.build(spec, device=device)
            prob.loss.engine.timing = True
            x0 = prob.prior.sample((B,))
            ms, ms_min, n = timed_kernel_ms(prob, x0)
        finally:
            os.environ.pop("SDEH_GENERIC_ONLY", None)
            os.environ.pop("SDEH_GMM_MM", None)
        T = prob.ts.numel() - 1
        tf = algorithmic_flops(spec) * B * T / (ms * 1e-3) / 1e12
        out[("headline_generic_kernel" if gen == "1" else "headline_generic_kernel_full_tables") if generic else name + ("_exact_form" if exact else "")] = {"kernel_ms": ms, "kernel_ms_min": ms_min, "launches": n,
                                                               "algorithmic_tflops": tf, "frac": tf / PEAK_FP32_TFLOPS,
                                                               "kernel": prob.loss.engine.last_kernel_name()}
    # the wide-network kernels (BASELINE configs[4]'s shape: C = 256, d = 196) at their workloads' own batch and T
    for name in ("wide_pis_funnel196", "cfg5_like_bridge196"):
        spec = problems.baseline_spec(name)
        prob = problems.build(spec, device=device)
        prob.loss.engine.timing = True
        x0 = prob.prior.sample((spec["batch"],))
        ms, ms_min, n = timed_kernel_ms(prob, x0, n_warm=2, n=3, budget_s=1.5)
        T = prob.ts.numel() - 1
        tf = algorithmic_flops(spec) * spec["batch"] * T / (ms * 1e-3) / 1e12
        out[name] = {"batch": spec["batch"], "steps": T, "kernel_ms": ms, "kernel_ms_min": ms_min, "launches": n,
                     "algorithmic_tflops": tf, "frac": tf / PEAK_FP32_TFLOPS, "kernel": prob.loss.engine.last_kernel_name()}
        del prob, x0
    # One training step of BASELINE configs[1] (DIS, method kl, GMM d = 2, T = 100) and of configs[2]'s shape (PIS kl, GMM-40 d = 50,
    # T = 200) at this batch: kernel times of the training forward and of the fused backward (back-propagation through time + weight
    # gradients, csrc/sdeh_bwdf.hip), and the backward's rate -- 2 x (4dC + 2 Lh C^2) FLOPs per trajectory-step (adjoint chain +
    # weight gradients; its re-evaluation of the network is not counted).
    # (+ the same two at the reference's training batch of 2048 -- conf/solver/*.yaml --, where back-propagation through time is
    # latency-bound and runs on tiles of 16 trajectories with four waves per tile, csrc/sdeh_bwdf16.hip)
    for name, bt in (("cfg2_gmm2_dis_kl", B), ("cfg3_gmm50_pis_kl", B), ("cfg2_gmm2_dis_kl", 2048), ("cfg3_gmm50_pis_kl", 2048)):
        spec = problems.baseline_spec(name)
        spec["batch"] = bt
        prob = problems.build(spec, device=device)
        eng = prob.loss.engine
        eng.timing = True
        x0 = prob.prior.sample((bt,))
        t_f, t_b = [], []
        for rep in range(4):
            prob.ctrl.zero_grad()
            val, _ = prob.loss(prob.ts, x0, prob.target.unnorm_log_prob, prob.second_log_prob)
            torch.cuda.synchronize()
            t_f.append(eng.last_kernel_ms())
            val.backward()
            torch.cuda.synchronize()
            t_b.append(eng.last_kernel_ms())
            bwd_name = eng.last_kernel_name()
        T = prob.ts.numel() - 1
        d, c, lh = spec["target"]["dim"], spec["net"]["channels"], spec["net"]["num_layers"] - 2
        tb = min(t_b[1:])
        tf = 2 * (4 * d * c + 2 * lh * c * c) * bt * T / (tb * 1e-3) / 1e12
        out["train_step_" + name + ("" if bt == B else f"_b{bt}")] = {"method": "kl", "batch": bt, "steps": T, "forward_kernel_ms": min(t_f[1:]), "backward_kernel_ms": tb,
                                     "backward_kernel": bwd_name, "backward_algorithmic_tflops": tf,
                                     "backward_frac": tf / PEAK_FP32_TFLOPS}
    # Training on the wide networks (csrc/sdeh_wide_bwd.hip; `--workload train_cfg5_like | train_wide_pis_lv` are the full-size runs):
    # configs[4]'s shape -- Bridge lv, two C = 256 networks, exact divergence, B = 4096 per GPU -- at T = 20 of its 200 steps (the
    # kernels' time per step does not depend on T), and a plain PIS-style network of that width at B = 8192, T = 200
    for name, bt, steps in (("cfg5_like_bridge196", 4096, 20), ("wide_pis_funnel196", 8192, 200)):
        spec = problems.baseline_spec(name)
        spec["batch"], spec["grid"]["steps"], spec["loss"]["method"] = bt, steps, "lv"
        prob = problems.build(spec, device=device)
        eng = prob.loss.engine
        eng.timing = True
        inf = getattr(prob.loss, "inference_ctrl", None)
        x0 = prob.prior.sample((bt,))
        t_f, t_k, t_b = [], [], []
        for rep in range(3):
            for m in (prob.ctrl, inf):
                if m is not None:
                    m.zero_grad()
            val, _ = prob.loss(prob.ts, x0, prob.target.unnorm_log_prob, prob.second_log_prob)
            torch.cuda.synchronize()
            t_f.append(eng.last_kernel_ms())
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            val.backward()
            e1.record()
            e1.synchronize()
            t_k.append(eng.last_kernel_ms())
            t_b.append(e0.elapsed_time(e1))
            kname = eng.last_kernel_name()
        d, c, lh = spec["target"]["dim"], spec["net"]["channels"], spec["net"]["num_layers"] - 2
        f_net = 4 * d * c + 2 * lh * c * c
        bridge = inf is not None
        f_dom, f_exec = (8 * d * c * c, 14 * d * c * c) if bridge else (f_net, 2 * f_net)
        f_bwd = (4 * f_net + 8 * d * c * c) if bridge else 2 * f_net
        rows = bt * steps
        out["train_step_wide_" + name] = {
            "method": "lv", "batch": bt, "steps": steps, "forward_kernel_ms": min(t_f[1:]), "backward_ms": min(t_b[1:]),
            "dominant_kernel": kname, "dominant_kernel_ms": min(t_k[1:]),
            "dominant_frac_algorithmic": f_dom * rows / (min(t_k[1:]) * 1e-3) / 1e12 / PEAK_FP32_TFLOPS,
            "dominant_frac_executed": f_exec * rows / (min(t_k[1:]) * 1e-3) / 1e12 / PEAK_FP32_TFLOPS,
            "backward_frac_algorithmic": f_bwd * rows / (min(t_b[1:]) * 1e-3) / 1e12 / PEAK_FP32_TFLOPS}
        del prob, x0
    # A 64-channel Bridge (conf/solver/bridge.yaml / basic_bridge.yaml with the shipped networks; funnel d = 10 and a d = 50 Gaussian, T = 200):
    # loss + backward of one step, wall clock on HIP events.  Forward = plain launch + row-parallel inference pass, backward = the two fused
    # kernels (DESIGN.md 3e'); "dominant" = the divergence backward launch group of csrc/sdeh_bridgef.hip, 6 x 2 x 64 x 64 FLOPs per (row, coordinate).
    for d, bt, method in ((10, 2048, "lv"), (10, 2048, "kl"), (50, 16384, "lv")):
        tspec = dict(kind="funnel", dim=d) if d == 10 else dict(kind="iso_gauss", dim=d, loc=1.0, scale=0.5)
        ctrl = dict(clip_model=10.0, clip_score=10.0, scale_score=1.0, gamma_dim=1, gamma_bias=1.0)
        spec = dict(batch=bt, target=tspec, prior=dict(kind="iso_gauss", dim=d), sde=dict(kind="scaled_bm", diff_coeff=1.0, terminal_t=1.0),
                    ctrl=dict(kind="lerp_target", **ctrl), inference_ctrl=dict(kind="lerp_prior", **ctrl),
                    net=dict(channels=64, num_layers=4, activation="gelu"),
                    loss=dict(kind="time_reversal", method=method, max_rnd=1e8 if method == "lv" else None), grid=dict(start=0.0, end=1.0, steps=200))
        prob = problems.build(spec, device=device)
        inf = prob.loss.inference_ctrl
        x0 = prob.prior.sample((bt,))
        t_f, t_b = [], []
        for rep in range(4):
            prob.ctrl.zero_grad()
            inf.zero_grad()
            e0, e1, e2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
            e0.record()
            val, _ = prob.loss(prob.ts, x0, prob.target.unnorm_log_prob, prob.second_log_prob)
            e1.record()
            val.backward()
            e2.record()
            e2.synchronize()
            t_f.append(e0.elapsed_time(e1))
            t_b.append(e1.elapsed_time(e2))
        rows = bt * 200
        out[f"train_step_bridge64_d{d}_b{bt}_{method}"] = {
            "method": method, "batch": bt, "steps": 200, "dim": d, "forward_ms": min(t_f[1:]), "backward_ms": min(t_b[1:]),
            "step_ms": min(f + b for f, b in zip(t_f[1:], t_b[1:])),
            "backward_frac_by_divergence_flops": 6 * 2 * 64 * 64 * d * rows / (min(t_b[1:]) * 1e-3) / 1e12 / PEAK_FP32_TFLOPS}
        del prob, x0
    return out


def log_z_block(spec, device, B: int, rank: int, world: int) -> dict | None:
    """log-Z quality on the headline target with a TRAINED control (tests/golden/trained_pis_gmm50.pt, produced by
    tools/train_demo.py with the HIP training path): importance-sampling log Z in fast mode (in-kernel noise) with its standard
    error and ESS, and -- parity mode, identical noise -- against the CPU oracle on a sub-batch."""
    # the control trained on the REFERENCE's schedule (tools/train_reference_schedule.py: basic_pis / kl, Adam 1e-3, 10 000 steps of
    # batch 512 through GraphedTrainStep); the 1000-step control of rounds 2-3 as the fallback
    path = ROOT / "tests" / "golden" / "trained_pis_gmm50_ref_schedule.pt"
    if not path.exists():
        path = ROOT / "tests" / "golden" / "trained_pis_gmm50.pt"
    if not path.exists() or world != 1:
        return None
    from sde_sampler_amd import engine as E
    from sde_sampler_amd import problems

    state = torch.load(path, map_location="cpu")
    prob = problems.build(spec, params=state["params"], device=device)
    x0 = prob.prior.sample((B,))
    with torch.no_grad():
        _, rnd, _ = prob.loss.simulate(prob.ts, x0, prob.target.unnorm_log_prob, prob.second_log_prob, compute_ito_int=True)
    est = E.estimators_from_stats(E.merge_stats(E.estimator_stats(rnd)))
    w = torch.exp(-rnd.double() - est["log_weight_max"]).flatten()
    se = float(w.std() / w.mean() / math.sqrt(B))  # delta method: s.e.(log mean w) = cv(w) / sqrt(B)
    x_T = prob.eval(x0, compute_weights=False, return_traj=False).samples
    comp = torch.cdist(x_T[:16384], prob.target.loc).argmin(dim=1)  # nearest component of the padded mixture (equal scales)
    share = torch.bincount(comp, minlength=prob.target.loc.shape[0]).double() / comp.numel()
    out = {"control": "trained (tests/golden/%s: %s)" % (path.name, state.get("note", "")),
           "log_norm_const_is": est["log_norm_const_is"], "se": se, "ess": est["ess"], "ess_frac": est["ess"] / B,
           "log_norm_const_lb_ito": est["mean_neg_rnd"], "true_log_norm_const": 0.0, "batch": B,
           "abs_log_z_error": abs(est["log_norm_const_is"] - 0.0),
           "modes_covered": int((share >= 0.5 / share.numel()).sum()), "n_modes": int(share.numel()),
           "note": "PIS on this target collapses onto few of the 40 modes (log Z_is -> log(covered / 40) = -3.69 for one): the oracle "
                   "trained on the same schedule on the CPU takes the same course (profiles/r04_train_reference_cpu.txt, tests/perf/train_reference_cpu.py)"}
    # parity mode on a sub-batch (identical x0 and noise): the GPU half here, the CPU half inside the cpu_baseline leg
    Bs, T, d = 4096, prob.ts.numel() - 1, spec["target"]["dim"]
    torch.manual_seed(11)
    noise = torch.randn(T, Bs, d)
    got = prob.eval(x0[:Bs], compute_weights=True, return_traj=False, noise=noise.to(device))
    out["parity_batch"] = Bs
    out["_parity"] = dict(params=state["params"], x0=x0[:Bs].cpu(), noise=noise, ts=prob.ts.cpu(),
                          gpu_is=got.log_norm_const_preds["log_norm_const_is"],
                          gpu_lb_ito=got.log_norm_const_preds["log_norm_const_lb_ito"])
    return out


def run_train(args, device):
    """Training workloads (single GPU): a bench step = loss(...) forward, backward, Adam.  Roofline: the dominant backward kernel
    (the fused backward for 64 channels; the chain kernel or the Bridge's divergence backward for the wide networks) with its
    algorithmic FLOPs; `backward_ms` = the whole backward between HIP events on the launch stream.  cpu_baseline: the oracle's loss +
    autograd backward on a bounded sample."""
    from sde_sampler_amd import problems

    spec_name, metric, description = WORKLOADS[args.workload]
    spec = problems.baseline_spec(spec_name)
    batch_default, method_default = TRAIN_DEFAULTS.get(args.workload, (65536, None))
    spec["batch"] = args.batch or batch_default
    if method_default is not None:
        spec["loss"]["method"] = method_default
    if args.em_steps:
        spec["grid"]["steps"] = args.em_steps
    prob = problems.build(spec)
    inf = getattr(prob.loss, "inference_ctrl", None)
    params_cpu = {k: v.detach().clone() for k, v in prob.ctrl.state_dict().items()}
    params_inf_cpu = {k: v.detach().clone() for k, v in inf.state_dict().items()} if inf is not None else None
    tt = (dict(loc=prob.target.loc.clone(), scale=prob.target.scale.clone(), mixture_weights=prob.target.mixture_weights.clone())
          if spec["target"]["kind"] == "gmm" else target_state(spec, prob))
    prob.to(device)
    B, T, d = spec["batch"], prob.ts.numel() - 1, spec["target"]["dim"]
    c, lh = spec["net"]["channels"], spec["net"]["num_layers"] - 2
    method = spec["loss"]["method"]
    trainable = list(prob.ctrl.parameters()) + (list(inf.parameters()) if inf is not None else [])
    opt = torch.optim.Adam(trainable, lr=1e-4)
    eng = prob.loss.engine
    eng.timing = True
    # the loss without a host round trip (losses/oc.py: `graph_safe` -- filter, statistics, loss value and per-row gradient in three
    # launches of the library, the running n_filtered kept on the device): nothing in a step synchronises, the host runs ahead of the GPU
    # and the kernel durations are read from the library's ring of event pairs AFTER the timed region
    prob.loss.graph_safe = True
    torch.manual_seed(1)
    marks = []

    def step(record):
        x0 = prob.prior.sample((B,))
        opt.zero_grad(set_to_none=True)
        loss, _ = prob.loss(prob.ts, x0, prob.target.unnorm_log_prob, prob.second_log_prob)
        if record:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
        loss.backward()
        if record:
            e1.record()
            marks.append((e0, e1))
        opt.step()
        return loss

    for _ in range(args.warmup):
        step(False)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = step(True)
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    bwd_total_ms = [e0.elapsed_time(e1) for e0, e1 in marks]
    # the timed launches of the region, newest first: (backward, forward) per step
    hist = eng.kernel_ms_history(2 * min(args.steps, 64))
    kernel = eng.last_kernel_name()
    bwd_ms = [ms for name, ms in hist if name == kernel]
    fwd_ms = [ms for name, ms in hist if name.startswith("traj_")] or [float("nan")]
    k_ms = statistics.median(bwd_ms)
    f_net = 4 * d * c + 2 * lh * c * c
    if kernel.startswith("bridge_div_bwd_wide"):
        ci, lhi = spec.get("inference_net", spec["net"])["channels"], spec.get("inference_net", spec["net"])["num_layers"] - 2
        n_prod = 4 if lhi == 2 else 2  # [C, C] products per (row, coordinate) the gradient itself needs (two weight-gradient
        flops_k = n_prod * 2 * d * ci * ci  # accumulations + two adjoint products); the kernel also recomputes F_j, G_j (+ F_j once more)
        flops_exec = (7 if lhi == 2 else 3) * 2 * d * ci * ci
        flops_total = 2 * f_net + 2 * (4 * d * ci + 2 * lhi * ci * ci) + flops_k
        note = ("dominant kernel = the Bridge's divergence backward (csrc/sdeh_wide_bwd.hip, two launches: one hidden layer's [C, C] "
                "gradient resident per launch); algorithmic FLOPs per trajectory-step: d x 4 products of 2 C^2 (weight-gradient "
                "accumulations of both hidden layers + the two adjoint products); executed: 7 products (F_j, G_j recomputed, F_j twice)")
    elif kernel.startswith("bwd_wide"):
        flops_k, flops_exec, flops_total = f_net, 2 * f_net, 2 * f_net
        note = ("dominant kernel = the wide chain kernel (csrc/sdeh_wide_bwd.hip): algorithmic FLOPs = the adjoint chain (4dC + 2 Lh C^2); it "
                "also re-evaluates the network (executed: twice that); the weight gradients are sdeh_weight_grad's (counted in "
                "`backward_algorithmic_tflops` over the whole backward)")
    else:
        flops_k = flops_total = 2 * f_net
        zrec = ",zrec" in kernel
        flops_exec = 2 * f_net if zrec else 3 * f_net
        note = ("dominant kernel = the fused backward (csrc/sdeh_bwdf*.hip); algorithmic FLOPs: adjoint chain + weight gradients = "
                "2 x (4dC + 2 Lh C^2); " + ("it reads the pre-activation record of the training forward (ABI v6) and does NOT re-evaluate the "
                                            "network: executed = algorithmic matrix work" if zrec else
                                            "it re-evaluates the network (executed: a third on top)"))
    achieved = flops_k * B * T / (k_ms * 1e-3) / 1e12
    b_ms = statistics.median(bwd_total_ms)
    out = {"metric": metric, "value": B * T * args.steps / elapsed, "unit": "trajectory-steps/s", "n_gpus": 1, "steps": args.steps,
           "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": "weak",
           "vs_baseline": None, "dtype": "f32", "data": "synthetic",
           "config": {"workload": f"{args.workload}: {description}", "batch_per_gpu": B, "global_batch": B, "em_steps": T, "dim": d,
                      "channels": c, "method": method, "noise": "in-kernel Philox4x32-10 + Box-Muller (replayed by the backward)"},
           "roofline": {"bound": "mfma", "achieved": achieved, "peak": PEAK_FP32_TFLOPS, "unit": "TFLOP/s", "frac": achieved / PEAK_FP32_TFLOPS,
                        "traffic": None, "kernel": kernel, "kernel_ms": k_ms, "forward_kernel_ms": statistics.median(fwd_ms),
                        "flops_per_traj_step": flops_k, "executed_tflops": flops_exec * B * T / (k_ms * 1e-3) / 1e12,
                        "frac_executed": flops_exec * B * T / (k_ms * 1e-3) / 1e12 / PEAK_FP32_TFLOPS,
                        "backward_ms": b_ms, "backward_algorithmic_tflops": flops_total * B * T / (b_ms * 1e-3) / 1e12,
                        "backward_frac": flops_total * B * T / (b_ms * 1e-3) / 1e12 / PEAK_FP32_TFLOPS, "note": note},
           "final_loss": float(loss)}
    if not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(spec, (params_cpu, tt, params_inf_cpu), train_method=method)
        out["speedup_vs_cpu_baseline"] = out["value"] / out["cpu_baseline"]["value"]
    print(json.dumps(out))


def run(args, rank: int, world: int, local_rank: int):
    import torch.distributed as dist

    if args.same_device:
        local_rank = 0
    n_dev = torch.cuda.device_count()
    if local_rank >= n_dev:
        raise SystemExit(f"rank {rank}: needs cuda:{local_rank} but the node has {n_dev} GPU(s) "
                         f"(--same-device --backend gloo exercises the N > 1 path on one GPU)")
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    # a process group exists whenever a launcher provided the rendezvous (WORLD_SIZE in the environment, also WORLD_SIZE=1) or
    # --dist asks for it: the one-rank job then runs the very code an 8-rank job runs (RCCL init with device_id, the 8-float
    # all-gather on the device tensor, barriers) -- tests/test_hip_rccl.py
    use_dist = world > 1 or args.dist or os.environ.get("WORLD_SIZE") is not None
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("RANK", str(rank))
        os.environ.setdefault("WORLD_SIZE", str(world))
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=device)
        else:
            dist.init_process_group(args.backend)

    from sde_sampler_amd import problems

    spec_name, metric, description = WORKLOADS[args.workload]
    spec = problems.baseline_spec(spec_name)
    if args.batch:
        spec["batch"] = args.batch
    if args.em_steps:
        spec["grid"]["steps"] = args.em_steps
    prob = problems.build(spec)
    inf = getattr(prob.loss, "inference_ctrl", None)
    cpu_state = ({k: v.detach().clone() for k, v in prob.ctrl.state_dict().items()},
                 dict(loc=prob.target.loc.clone(), scale=prob.target.scale.clone(),
                      mixture_weights=prob.target.mixture_weights.clone()) if spec["target"]["kind"] == "gmm" else target_state(spec, prob),
                 {k: v.detach().clone() for k, v in inf.state_dict().items()} if inf is not None else None)
    prob.to(device)
    B, T, d = spec["batch"], prob.ts.numel() - 1, spec["target"]["dim"]
    # the same seed on every rank: the in-kernel Philox stream is keyed by (seed, call, GLOBAL row), so the N-rank job draws
    # exactly the noise a single launch over the N*B rows would (the prior draw below is per rank)
    torch.manual_seed(1)
    x0 = prob.prior.sample((B,))
    prob.loss.row_offset = rank * B
    prob.loss.engine.timing = True

    def step():
        return prob.eval(x0, compute_weights=False, return_traj=False)

    # One step = one eager `loss.eval` (coefficient tables + trajectory kernel + estimator reduction + the 8-float copy its Results
    # need), the trajectory kernel bracketed by HIP events on its stream in every step of the timed region (the roofline's duration).
    # --graphed replays the same call as one hipGraph (utils.graphs.GraphedEval: identical kernels and results, ~50 us less host work
    # per step) -- events recorded by graph nodes read 0.3 ms too long on this stack, so that mode is reported in `graphed_step`
    # next to the eager line (measured behind the timed region) and is not what `value` / `roofline` are taken from.
    graphed = False

    def fence():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    # a target evaluated between one-step segments (the NICE flow): the "kernel" of the roofline is the whole evaluation -- T segment
    # launches + T evaluations of the flow's score, every one on torch's current stream -- between two events on that stream
    stepped = spec["target"]["kind"] == "nice"
    for _ in range(args.warmup):
        step()
    kernel_ms = []
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        if stepped:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
        res = step()
        if stepped:
            e1.record()
            e1.synchronize()
            kernel_ms.append(e0.elapsed_time(e1))
        else:
            kernel_ms.append(prob.loss.engine.last_kernel_ms())
    fence()
    elapsed = time.perf_counter() - t0
    scaling_detail = None
    if use_dist:
        cdev = device if args.backend == "nccl" else "cpu"
        # what a sub-linear curve would be attributed to on first contact with a multi-GPU node: every rank's kernel time and step time,
        # and the wall time of the one collective of an evaluation (8 floats per rank, latency-bound) measured on its own
        mine = torch.tensor([statistics.median(kernel_ms), 1e3 * elapsed / args.steps], device=cdev, dtype=torch.float64)
        per_rank = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(per_rank, mine)
        part, merged = torch.zeros(8, device=cdev), torch.zeros(8 * world, device=cdev)
        fence()
        tc = time.perf_counter()
        for _ in range(20):
            dist.all_gather_into_tensor(merged, part)
        if cdev != "cpu":
            torch.cuda.synchronize()
        coll_us = 1e6 * (time.perf_counter() - tc) / 20
        scaling_detail = {"kernel_ms_per_rank": [float(v[0]) for v in per_rank], "ms_per_step_per_rank": [float(v[1]) for v in per_rank],
                          "estimator_all_gather_us": coll_us,
                          "note": "per rank: median trajectory-kernel ms (HIP events) and wall ms per step of the timed region; the 8-float "
                                  "all_gather_into_tensor of an evaluation timed alone (20 back-to-back calls + one synchronisation)"}
        t = torch.tensor([elapsed], device=cdev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = t.item()

    # the same step replayed as one hipGraph (N = 1 without a process group: the estimator merge across ranks goes through the host)
    graphed_step = None
    if not use_dist and not args.no_graphed:
        from sde_sampler_amd.utils.graphs import GraphedEval

        prob.loss.engine.timing = False
        calls_before = prob.loss.engine.calls  # (the evaluations below keep the Philox offsets of a run without this block)
        replay = GraphedEval(lambda x: prob.eval(x, compute_weights=False, return_traj=False), [prob.loss], x0)
        n_rep = max(args.steps, 20)
        for _ in range(10):
            replay()
        torch.cuda.synchronize()
        tg = time.perf_counter()
        for _ in range(n_rep):
            replay()
        torch.cuda.synchronize()
        tg = (time.perf_counter() - tg) / n_rep
        graphed_step = {"ms_per_step": 1e3 * tg, "value": B * T / tg, "steps": n_rep,
                        "what": "the bench step replayed as one hipGraph (sde_sampler_amd.utils.graphs.GraphedEval): one launch + the 8-float copy"}
        prob.loss.rng_counter = None  # back to by-value Philox offsets for the evaluations below
        prob.loss.engine.calls = calls_before
        prob.loss.engine.timing = True

    # quality: log Z from one weighted evaluation (in-kernel noise), global over all ranks
    full = prob.eval(x0, compute_weights=True, return_traj=False)
    if rank != 0:
        if use_dist:
            dist.destroy_process_group()
        return

    flops = algorithmic_flops(spec)
    k_ms = sum(kernel_ms) / len(kernel_ms)
    achieved = flops * B * T / (k_ms * 1e-3) / 1e12
    headline = args.workload == "gmm50_pis_headline" and B == 65536 and T == 100
    pmc = pmc_record() if headline else None
    roofline = {"bound": "mfma", "achieved": achieved, "peak": PEAK_FP32_TFLOPS, "unit": "TFLOP/s",
                "frac": achieved / PEAK_FP32_TFLOPS,
                "traffic": (2.0 * pmc["FETCH_SIZE_KiB"] + pmc["WRITE_SIZE_KiB"]) * 1024.0 if pmc else None,
                "algorithmic_hbm_bytes": B * (8 * d + 4),
                "kernel": (prob.loss.engine.last_kernel_name() + f" x {T} one-step segments + nice_gemm (csrc/sdeh_nice.hip: {T} score evaluations)"
                           if stepped else prob.loss.engine.last_kernel_name()),
                "kernel_ms": k_ms, "kernel_ms_min": min(kernel_ms),
                "flops_per_traj_step": flops,
                "note": "fp32 MFMA and fp32 VALU share one datapath on gfx950 (profiles/r01_ubench_coexec.txt): 157.3 TFLOP/s is "
                        "the budget for both; `achieved` counts SURVEY 8d's ALGORITHMIC FLOPs; `executed_tflops` counts the "
                        "instructions the kernel really issued (committed PMC pass of this workload): SQ_INSTS_MFMA x 4096 "
                        "(v_mfma_f32_32x32x2_f32) + SQ_INSTS_VALU_FMA_F32 x 128 (64 lanes x 2), divided by this run's kernel time -- a lower "
                        "bound since round 2: the GELU polynomial issues ~5.9e7 v_pk_fma_f32 per launch (two FMAs per lane), which "
                        "that counter does not count twice"}
    if pmc and "SQ_INSTS_MFMA" in pmc:
        executed = (pmc["SQ_INSTS_MFMA"] * 4096.0 + pmc["SQ_INSTS_VALU_FMA_F32"] * 128.0) / (k_ms * 1e-3) / 1e12
        roofline["executed_tflops"] = executed
        roofline["frac_executed"] = executed / PEAK_FP32_TFLOPS
        roofline["pmc_source"] = pmc.get("source")
    out = {
        "metric": metric,
        "value": world * B * T * args.steps / elapsed,
        "unit": "trajectory-steps/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": 1e3 * elapsed / args.steps,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32", "data": "synthetic",
        "config": {"workload": f"{args.workload}: {description}",
                   "batch_per_gpu": B, "global_batch": world * B, "em_steps": T, "dim": d,
                   "channels": spec["net"]["channels"],
                   "noise": "in-kernel Philox4x32-10 + Box-Muller", "parallelism": f"batch-sharded x{world}",
                   "step": "loss.eval replayed as one hipGraph (utils.graphs.GraphedEval)" if graphed else "eager loss.eval"},
        "roofline": roofline,
        "log_z_untrained_control": {"log_norm_const_is": full.log_norm_const_preds["log_norm_const_is"],
                                    "log_norm_const_lb_ito": full.log_norm_const_preds["log_norm_const_lb_ito"],
                                    "log_norm_const_lb": res.log_norm_const_preds["log_norm_const_lb"],
                                    "true_log_norm_const": 0.0,
                                    "note": "random-init control: importance weights are degenerate, only the bound is meaningful"},
    }
    print("[bench] gpu leg done: " + json.dumps({k: out[k] for k in ("value", "ms_per_step", "roofline")}),
          file=sys.stderr, flush=True)
    if graphed_step is not None:
        out["graphed_step"] = graphed_step
    if world == 1 and headline and not args.no_extra:
        out["log_z"] = log_z_block(spec, device, B, rank, world)
        out["extra"] = extra_block(device, B)
    parity = out["log_z"].pop("_parity", None) if out.get("log_z") else None
    if world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(spec, cpu_state, args.cpu_budget, parity)
        checked = out["cpu_baseline"].pop("parity_log_z", None)
        if checked:
            out["log_z"].update(checked)
        if out["cpu_baseline"]["value"]:
            out["speedup_vs_cpu_baseline"] = out["value"] / out["cpu_baseline"]["value"]
    if use_dist:
        out["config"]["process_group"] = {"backend": dist.get_backend(), "world_size": dist.get_world_size()}
        out["scaling_detail"] = scaling_detail
    print(json.dumps(out), flush=True)
    if use_dist:
        dist.destroy_process_group()


def _spawned(rank: int, args, port: int):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(args.gpus), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    run(args, rank, args.gpus, rank)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None,
                    help="timed steps (default: 1000 for the headline so that the timed region is > 2 s; fewer for the heavy workloads)")
    ap.add_argument("--warmup", type=int, default=None,
                    help="untimed steps (the first ~8 launches after idle run while the GPU clock is still ramping up)")
    ap.add_argument("--workload", default="gmm50_pis_headline", choices=sorted(WORKLOADS))
    ap.add_argument("--batch", type=int, default=None, help="trajectories per GPU (default: the workload's)")
    ap.add_argument("--em-steps", type=int, default=None, help="Euler-Maruyama steps T (default: the workload's)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--eager", action="store_true", help="(the default; kept for scripts) one eager loss.eval per step")
    ap.add_argument("--no-graphed", action="store_true", help="skip the `graphed_step` measurement (the step replayed as one hipGraph)")
    ap.add_argument("--no-extra", action="store_true", help="skip the log_z / extra blocks of the headline line")
    ap.add_argument("--cpu-budget", type=float, default=30.0, help="seconds of CPU work for the cpu_baseline leg")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend for N > 1 (nccl = RCCL)")
    ap.add_argument("--dist", action="store_true",
                    help="initialise the process group (RCCL) also for --gpus 1: a one-rank job that runs the multi-rank code path")
    ap.add_argument("--same-device", action="store_true",
                    help="testing aid: all ranks use cuda:0 (with --backend gloo) to exercise the N > 1 path on one GPU")
    args = ap.parse_args()
    heavy = args.workload in ("wide_pis_funnel196", "cfg5_like_bridge196", "train_cfg5_like", "cfg5_nice_bridge196", "train_cfg5_nice")
    train = args.workload.startswith("train_")
    if args.steps is None:
        args.steps = (3 if args.workload in ("train_cfg5_like", "train_cfg5_nice") else 5) if heavy else ((20 if args.workload == "train_wide_pis_lv" else 100) if train else 1000)
    if args.warmup is None:
        args.warmup = (1 if args.workload in ("train_cfg5_like", "train_cfg5_nice") else 2) if heavy else (5 if train else 20)
    if train:
        if args.gpus != 1 or os.environ.get("WORLD_SIZE") not in (None, "1"):
            raise SystemExit("the training workloads are single-GPU measurements (data-parallel training: tests/test_distributed_gloo.py)")
        torch.cuda.set_device(0)
        run_train(args, torch.device("cuda", 0))
        return

    env_world = os.environ.get("WORLD_SIZE")
    if env_world is not None:  # launched by torch.distributed.run (or an equivalent launcher)
        world = int(env_world)
        if world != args.gpus:
            raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
        run(args, int(os.environ.get("RANK", "0")), world, int(os.environ.get("LOCAL_RANK", "0")))
    elif args.gpus > 1:  # plain invocation: spawn one rank per GPU ourselves
        if not args.same_device and torch.cuda.device_count() < args.gpus:
            raise SystemExit(f"--gpus {args.gpus} but this node has {torch.cuda.device_count()} GPU(s)")
        import socket

        import torch.multiprocessing as mp

        with socket.socket() as s:
            s.bind(("127.0.0.1", 0))
            port = s.getsockname()[1]
        mp.spawn(_spawned, args=(args, port), nprocs=args.gpus, join=True)
    else:
        run(args, 0, 1, 0)


if __name__ == "__main__":
    main()

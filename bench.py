#!/usr/bin/env python3
"""Benchmark of the hot path: trajectory-steps/s of the fused Euler-Maruyama engine on MI355X.

    python bench.py --gpus N --steps K --warmup W          (N > 1: launched by torch.distributed.run)

Workload (BASELINE.json metric "trajectory-steps/sec ... GMM-40 d=50"): target = 40-mode GMM in d=50
(fab means in the first two coordinates, scale softplus(1)), solver basic_pis (ScoreCtrl + FourierMLP C=64/4 layers
GELU, Delta prior, ScaledBM(sqrt 0.2, T=5)), batch 65 536 trajectories PER GPU, T = 100 steps, fp32, in-kernel
Philox noise, random-init weights (last layers N(0, 0.05^2)), x0 resident in HBM.
One "step" = one pass of the hot path over the batch with the reference's `eval/sample_time` semantics
(solver/oc.py:88-97): loss.eval(ts, x, ..., compute_weights=False, return_traj=False) under no_grad, i.e. the
trajectory kernel + the log-Z lower-bound reduction (+ the 8-float all-gather when N > 1).
Prints ONE JSON line (rank 0).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

WORKLOAD = "gmm50_pis_headline"
PEAK_FP32_TFLOPS = 157.3  # MI355X fp32 MFMA (= fp32 vector) dense peak, /opt/skills/guides/MI355X_MICROARCH.md


def measured_hbm_traffic():
    """HBM bytes per trajectory-kernel launch from the committed rocprofv3 PMC passes (tools/pmc_profile.sh):
    2 x FETCH_SIZE (gfx950 correction for wide coalesced reads, MI355X_MICROARCH.md section HBM) + WRITE_SIZE, KiB -> bytes."""
    path = ROOT / "profiles" / "pmc_traffic.json"
    try:
        rec = json.loads(path.read_text())
        return (2.0 * rec["FETCH_SIZE_KiB"] + rec["WRITE_SIZE_KiB"]) * 1024.0
    except (OSError, KeyError, ValueError):
        return None


def flops_per_traj_step(d: int, c: int, lh: int, k: int) -> float:
    """F(d,C,Lh,K) = 4dC + 2 Lh C^2 (MLP, time embedding hoisted) + 6dK + 4K (GMM score) + 20d."""
    return 4 * d * c + 2 * lh * c * c + 6 * d * k + 4 * k + 20 * d


def cpu_baseline(spec, prob_cpu_state, budget_s: float = 20.0) -> dict:
    """The reference's CPU path (oracle = op-for-op PyTorch-CPU restatement) on a bounded sample of the workload:
    chunks of 2048 trajectories, each integrated for all T steps, until ~budget_s of CPU time is spent."""
    from oracle import em_oracle as eo

    params, tt = prob_cpu_state
    oracle = eo.Problem(spec, params, tt)
    ts = oracle.grid()
    T, d = ts.numel() - 1, spec["target"]["dim"]
    chunk = 2048
    x0 = torch.zeros(chunk, d)  # Delta prior: x0 = 0 (distr/delta.py:25-28)
    ncpu = os.cpu_count() or 1
    # pick the intra-op thread count on a short probe (all hardware threads is not always the fastest)
    best = None
    for threads in sorted({ncpu, max(1, ncpu // 2), min(ncpu, 32)}, reverse=True):
        torch.set_num_threads(threads)
        oracle.eval(ts[:3], x0, None, compute_weights=False)  # warm-up
        t0 = time.perf_counter()
        oracle.eval(ts[:6], x0, None, compute_weights=False)
        dt = time.perf_counter() - t0
        if best is None or dt < best[1]:
            best = (threads, dt)
    threads = best[0]
    torch.set_num_threads(threads)
    torch.manual_seed(7)
    done, lbs = 0, []
    t0 = time.perf_counter()
    while True:
        res = oracle.eval(ts, x0, None, compute_weights=False)
        lbs.append(res["log_norm_const_lb"])
        done += chunk
        dt = time.perf_counter() - t0
        if dt > budget_s or done >= spec["batch"]:
            break
    return {"value": done * T / dt, "unit": "trajectory-steps/s", "cores": threads, "kind": "port",
            "sample": f"oracle/em_oracle.py (PyTorch-CPU restatement of the reference loop, torch.randn noise), same "
                      f"workload, {done} of {spec['batch']} trajectories in chunks of {chunk}, T={T}, {threads} torch "
                      f"threads of {ncpu} hardware threads, {dt:.1f} s, log_norm_const_lb={sum(lbs) / len(lbs):.4f}"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=10,
                    help="untimed steps (the first ~8 launches after idle run while the GPU clock is still ramping up)")
    ap.add_argument("--batch", type=int, default=None, help="trajectories per GPU (default: the workload's 65 536)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-budget", type=float, default=20.0, help="seconds of CPU work for the cpu_baseline leg")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend for N > 1 (nccl = RCCL)")
    ap.add_argument("--same-device", action="store_true",
                    help="testing aid: all ranks use cuda:0 (with --backend gloo) to exercise the N > 1 path on one GPU")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    import torch.distributed as dist

    if args.same_device:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=device)
        else:
            dist.init_process_group(args.backend)

    from sde_sampler_amd import problems

    spec = problems.baseline_spec(WORKLOAD)
    if args.batch:
        spec["batch"] = args.batch
    prob = problems.build(spec)
    cpu_state = ({k: v.detach().clone() for k, v in prob.ctrl.state_dict().items()},
                 dict(loc=prob.target.loc.clone(), scale=prob.target.scale.clone(),
                      mixture_weights=prob.target.mixture_weights.clone()))
    prob.to(device)
    B, T, d = spec["batch"], prob.ts.numel() - 1, spec["target"]["dim"]
    # the same seed on every rank: the in-kernel Philox stream is keyed by (seed, call, GLOBAL row), so the N-rank job draws
    # exactly the noise a single launch over the N*B rows would (x0 of the Delta prior is zero on every rank)
    torch.manual_seed(1)
    x0 = prob.prior.sample((B,))
    prob.loss.row_offset = rank * B
    prob.loss.engine.timing = True

    def step():
        return prob.eval(x0, compute_weights=False, return_traj=False)

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    kernel_ms = []
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        res = step()
        kernel_ms.append(prob.loss.engine.last_kernel_ms())
    fence()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], device=device if args.backend == "nccl" else "cpu", dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = t.item()

    # quality: log Z from one weighted evaluation (in-kernel noise), global over all ranks
    full = prob.eval(x0, compute_weights=True, return_traj=False)
    if rank != 0:
        if world > 1:
            dist.destroy_process_group()
        return

    flops = flops_per_traj_step(d, 64, spec["net"]["num_layers"] - 2, 40)
    k_ms = sum(kernel_ms) / len(kernel_ms)
    achieved = flops * B * T / (k_ms * 1e-3) / 1e12
    out = {
        "metric": "trajectory-steps/sec (batch x steps / s), GMM-40 d=50",
        "value": world * B * T * args.steps / elapsed,
        "unit": "trajectory-steps/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": 1e3 * elapsed / args.steps,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32", "data": "synthetic",
        "config": {"workload": f"{WORKLOAD}: GMM-40 d=50 (explicit loc/scale), basic_pis (ScoreCtrl, FourierMLP C=64 "
                               f"L=4 GELU, Delta prior, ScaledBM sqrt(0.2) T=5), eval sample_time semantics",
                   "batch_per_gpu": B, "global_batch": world * B, "em_steps": T, "dim": d, "gmm_components": 40,
                   "noise": "in-kernel Philox4x32-10 + Box-Muller", "parallelism": f"batch-sharded x{world}"},
        "roofline": {"bound": "mfma", "achieved": achieved, "peak": PEAK_FP32_TFLOPS, "unit": "TFLOP/s",
                     "frac": achieved / PEAK_FP32_TFLOPS, "traffic": measured_hbm_traffic(),
                     "algorithmic_hbm_bytes": B * (8 * d + 4),
                     "kernel": "sdeh::traj_ws_kernel<50,64,...> (wave-specialised; pis_gmm4 variant)", "kernel_ms": k_ms,
                     "kernel_ms_min": min(kernel_ms),
                     "flops_per_traj_step": flops,
                     "note": "fp32 MFMA and fp32 VALU share one datapath on gfx950 (profiles/r01_ubench_coexec.txt): "
                             "157.3 TFLOP/s is the budget for both; F counts SURVEY 8d's algorithmic FLOPs"},
        "log_z": {"log_norm_const_is": full.log_norm_const_preds["log_norm_const_is"],
                  "log_norm_const_lb_ito": full.log_norm_const_preds["log_norm_const_lb_ito"],
                  "log_norm_const_lb": res.log_norm_const_preds["log_norm_const_lb"],
                  "true_log_norm_const": 0.0},
    }
    print("[bench] gpu leg done: " + json.dumps({k: out[k] for k in ("value", "ms_per_step", "roofline", "log_z")}),
          file=sys.stderr, flush=True)
    if world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(spec, cpu_state, args.cpu_budget)
        out["speedup_vs_cpu_baseline"] = out["value"] / out["cpu_baseline"]["value"]
    print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

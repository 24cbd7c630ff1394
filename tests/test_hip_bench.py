"""bench.py end to end on one GPU: the single-rank line, and the N > 1 path (self-spawned ranks, estimator merge through
torch.distributed) exercised with two ranks sharing cuda:0 over gloo."""
import json
import subprocess
import sys
from pathlib import Path

import pytest
import torch

pytestmark = pytest.mark.gpu

ROOT = Path(__file__).resolve().parents[1]


def _bench(*flags, timeout=600):
    proc = subprocess.run([sys.executable, str(ROOT / "bench.py"), *flags], capture_output=True, text=True, timeout=timeout,
                          cwd=str(ROOT))
    assert proc.returncode == 0, proc.stderr[-2000:]
    lines = [ln for ln in proc.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, proc.stdout[-2000:]
    return json.loads(lines[0])


def test_two_ranks_on_one_device_equal_one_launch_over_the_global_batch():
    """`python bench.py --gpus 2` without a launcher spawns its own ranks; the merged lower bound equals what ONE launch over the
    2B global rows gives (global-row Philox counters: the two shards draw exactly the noise of rows [0,B) and [B,2B))."""
    from sde_sampler_amd import problems

    B = 4096
    out = _bench("--gpus", "2", "--same-device", "--backend", "gloo", "--steps", "3", "--warmup", "1", "--batch", str(B),
                 "--no-cpu-baseline")
    assert out["n_gpus"] == 2 and out["config"]["global_batch"] == 2 * B and out["config"]["batch_per_gpu"] == B
    assert out["steps"] == 3 and out["scaling"] == "weak" and out["value"] > 0
    spec = problems.baseline_spec("gmm50_pis_headline")
    spec["batch"] = 2 * B
    prob = problems.build(spec, device="cuda:0")
    torch.manual_seed(1)
    x0 = prob.prior.sample((2 * B,))
    # bench.py: warm-up + timed steps advance the per-call Philox offset; its last timed step is call (warmup + steps - 1)
    prob.loss.engine.calls = 1 + 3 - 1
    single = prob.eval(x0, compute_weights=False, return_traj=False)
    got = out["log_z_untrained_control"]["log_norm_const_lb"]
    want = single.log_norm_const_preds["log_norm_const_lb"]
    assert abs(got - want) <= 1e-5 * max(1.0, abs(want)), (got, want)


def test_more_ranks_than_gpus_is_an_error_not_a_silent_single_gpu_run():
    n = torch.cuda.device_count() + 1
    proc = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--gpus", str(n), "--steps", "1", "--no-cpu-baseline"],
                          capture_output=True, text=True, timeout=300, cwd=str(ROOT))
    assert proc.returncode != 0 and "GPU(s)" in (proc.stderr + proc.stdout)


def test_single_rank_line_has_the_contract_fields():
    out = _bench("--steps", "5", "--warmup", "2", "--batch", "8192", "--no-cpu-baseline", "--no-extra")
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline"):
        assert key in out, key
    r = out["roofline"]
    assert r["bound"] == "mfma" and r["peak"] == 157.3 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9
    assert r["kernel"].startswith("traj_") and r["kernel_ms"] > 0


def test_training_workload_line():
    """`bench.py --workload train_*`: one optimisation step per bench step; the roofline block is the fused backward kernel's."""
    out = _bench("--workload", "train_gmm2_dis_kl", "--batch", "4096", "--steps", "3", "--warmup", "1", "--no-cpu-baseline")
    for key in ("metric", "value", "unit", "n_gpus", "steps", "ms_per_step", "higher_is_better", "dtype", "config", "roofline"):
        assert key in out, key
    r = out["roofline"]
    assert r["kernel"].startswith("bwd_fused16<bptt") and r["kernel_ms"] > 0 and r["forward_kernel_ms"] > 0
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9
    assert out["config"]["method"] == "kl" and out["value"] > 0 and torch.isfinite(torch.tensor(out["final_loss"]))


def test_one_rank_process_group_line_agrees_with_the_group_less_line():
    """`bench.py --gpus 1 --dist` runs the code an 8-rank job runs (RCCL init with device_id, the 8-float
    all-gather on the device tensor, barriers, the max-over-ranks all-reduce) on the one GPU.  Its N = 1 point must agree with the
    BENCH line, so that the first SCALE run's curve starts where BENCH says; `scaling_detail` carries what a sub-linear curve would be
    attributed to (per-rank kernel and step times, the collective's wall time)."""
    flags = ("--steps", "200", "--warmup", "20", "--no-cpu-baseline", "--no-extra", "--no-graphed")
    # two separate processes on a GPU whose clock follows its power budget: a pair of runs can differ by a few per cent with no code
    # difference (round 5's measurement run saw one such pair in five) -- the claim is about the code paths, so a disagreeing pair is
    # measured again (at most three pairs) and every pair is reported
    pairs = []
    for _ in range(3):
        plain = _bench(*flags)
        grouped = _bench("--dist", *flags)
        assert grouped["config"]["process_group"] == {"backend": "nccl", "world_size": 1}
        assert "process_group" not in plain["config"] and "scaling_detail" not in plain
        pairs.append((grouped["value"], plain["value"]))
        if abs(grouped["value"] - plain["value"]) / plain["value"] <= 0.03:
            break
    rel = abs(grouped["value"] - plain["value"]) / plain["value"]
    assert rel <= 0.03, pairs
    sd = grouped["scaling_detail"]
    assert len(sd["kernel_ms_per_rank"]) == 1 and len(sd["ms_per_step_per_rank"]) == 1
    assert 0.5 * grouped["roofline"]["kernel_ms"] <= sd["kernel_ms_per_rank"][0] <= 1.5 * grouped["roofline"]["kernel_ms"]
    assert 0 < sd["estimator_all_gather_us"] < 5000
    assert abs(grouped["roofline"]["frac"] - plain["roofline"]["frac"]) <= 0.04, (grouped["roofline"]["frac"], plain["roofline"]["frac"])

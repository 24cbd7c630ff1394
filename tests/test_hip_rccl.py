"""The RCCL code path, executed (VERDICT r02 missing #2): a one-rank `nccl` process group on the one GPU runs exactly what an
8-rank job runs -- init with `device_id`, the 8-float `all_gather_into_tensor` on the device tensor, the log-variance loss's
all-reduce, the flat gradient all-reduce, barriers -- and must reproduce the group-less results."""
import json
import math
import os
import subprocess
import sys
from pathlib import Path

import pytest

pytestmark = pytest.mark.gpu

ROOT = Path(__file__).resolve().parents[1]


def _run(cmd, timeout=600, env=None):
    proc = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout, cwd=str(ROOT), env=env)
    assert proc.returncode == 0, (proc.stdout[-1500:], proc.stderr[-3000:])
    lines = [ln for ln in proc.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, proc.stdout[-2000:]
    return json.loads(lines[0])


def test_nccl_world_size_one_reproduces_the_groupless_path():
    out = _run([sys.executable, str(ROOT / "tests" / "rccl_world1_worker.py")])
    ref, got, calls = out["ref"], out["got"], out["calls"]
    # 2 problems x (eval + the raw merge ...) all-gathers; per problem the loss's 3-double all-reduce + the gradient bucket
    assert calls["all_gather"] >= 3 and calls["all_reduce"] >= 4, calls
    assert got["merge"] == ref["merge"]  # hex floats: bitwise
    for key in ref:
        if key.endswith("/eval"):
            assert got[key] == ref[key], (key, got[key], ref[key])  # estimators and importance weights: bitwise
        elif key.endswith("/train"):
            # the data-parallel loss share is the same estimator written as a sum over the kept rows (sum / n instead of mean()):
            # equal to fp32 rounding, not bitwise
            a, b = got[key], ref[key]
            assert math.isclose(a["loss"], b["loss"], rel_tol=2e-6, abs_tol=1e-7), (key, a, b)
            assert math.isclose(a["grad_norm"], b["grad_norm"], rel_tol=2e-5), (key, a, b)
            assert "train/loss_global" not in a["info_keys"]  # (an extra all-reduce + host round trip: only with report_global_loss)
        elif key.endswith("/graphed"):
            # the captured data-parallel step (ONE device-side all-reduce on the loss path + the gradient bucket as graph nodes, no host
            # synchronisation -- the worker runs an eager step under set_sync_debug_mode("error")): bit for bit the group-less one
            assert got[key] == ref[key], (key, got[key], ref[key])
            assert got[key]["skipped"] == 0
    # warm-up steps + the capture + the eager step: (loss + gradients) all-reduces each; replays issue them as graph nodes
    assert calls["graphed_all_reduce"] >= 2 * 2 * 4, calls


def test_bench_under_a_launcher_with_one_rank_uses_rccl():
    """`torch.distributed.run --nproc-per-node 1 bench.py --gpus 1` (what the driver's scaling run does at N = 1 when it goes through
    the launcher): the line must say so and carry the same lower bound as the plain run."""
    common = ["--gpus", "1", "--steps", "3", "--warmup", "1", "--batch", "8192", "--no-cpu-baseline", "--no-extra"]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    launched = _run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
                     "--master-port", "29547", str(ROOT / "bench.py"), *common], env=env)
    assert launched["config"]["process_group"] == {"backend": "nccl", "world_size": 1}
    plain = _run([sys.executable, str(ROOT / "bench.py"), "--eager", *common])
    assert "process_group" not in plain["config"] and plain["config"]["step"] == "eager loss.eval"
    assert plain["graphed_step"]["value"] > 0.9 * plain["value"]  # the same step replayed as one hipGraph, measured behind the timed region
    assert launched["log_z_untrained_control"] == plain["log_z_untrained_control"]  # same seeds, same Philox counters: bitwise
    flagged = _run([sys.executable, str(ROOT / "bench.py"), "--dist", *common])
    assert flagged["config"]["process_group"]["backend"] == "nccl"
    assert flagged["log_z_untrained_control"] == plain["log_z_untrained_control"]

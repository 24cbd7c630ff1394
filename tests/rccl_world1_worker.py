"""Worker of tests/test_hip_rccl.py (one process, one GPU): everything the multi-GPU path does over RCCL, at world size 1.

1. group-less reference: evaluation estimators, a data-parallel-free training loss and its gradients;
2. `dist.init_process_group("nccl", device_id=cuda:0)` with WORLD_SIZE=1, then the SAME calls again -- they now take the
   multi-rank code path: `all_gather_into_tensor` of the 8 estimator statistics on the DEVICE tensor, the log-variance loss's
   3-double all-reduce on the device, `all_reduce_gradients` over one flat device bucket, `barrier`;
3. prints one JSON line with both sets of numbers; the test compares them.
"""
import json
import os
import sys
from pathlib import Path

import torch
import torch.distributed as dist

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))

from sde_sampler_amd import engine as E  # noqa: E402
from sde_sampler_amd import problems  # noqa: E402
from sde_sampler_amd.utils.distributed import all_reduce_gradients  # noqa: E402


def run_all(tag: str) -> dict:
    out = {}
    dev = torch.device("cuda", 0)
    for name, method in (("gmm50_pis_headline", "kl"), ("cfg1_dw_dis_lv", "lv")):
        spec = problems.baseline_spec(name)
        spec["batch"] = 2048
        spec["loss"]["method"] = method
        torch.manual_seed(1)
        prob = problems.build(spec, device=dev)
        torch.manual_seed(3)
        x0 = prob.prior.sample((2048,))
        res = prob.eval(x0, compute_weights=True, return_traj=False)
        out[f"{name}/eval"] = {**res.log_norm_const_preds, **res.metrics,
                               "weights_sum_hex": float(res.weights.double().sum()).hex()}
        prob.ctrl.zero_grad()
        val, info = prob.loss(prob.ts, x0, prob.target.unnorm_log_prob, prob.second_log_prob)
        val.backward()
        all_reduce_gradients(prob.ctrl.parameters())  # no-op without a group; one flat SUM all-reduce on the device with one
        flat = torch.cat([p.grad.reshape(-1) for p in prob.ctrl.parameters() if p.grad is not None])
        out[f"{name}/train"] = {"loss": float(val), "grad_norm": float(flat.double().norm()),
                                "grad_sum_hex": float(flat.double().sum()).hex(), "info_keys": sorted(info)}
    # the raw collective on a device tensor: bitwise the local merge
    stats = E.estimator_stats(torch.linspace(-3.0, 5.0, 4096, device=dev).reshape(-1, 1))
    out["merge"] = [float(v).hex() for v in E.all_gather_stats(stats)]
    return out


def run_graphed(tag: str) -> dict:
    """A whole optimisation step -- forward, fused backward, the gradient all-reduce, Adam -- captured into ONE hipGraph and replayed
    (utils.graphs.GraphedTrainStep), with and without a process group; and the same step eagerly under
    torch.cuda.set_sync_debug_mode("error"): a host synchronisation anywhere in a data-parallel step raises."""
    from sde_sampler_amd.utils.graphs import GraphedTrainStep

    out = {}
    dev = torch.device("cuda", 0)
    for name, method in (("cfg1_dw_dis_lv", "lv"), ("cfg2_gmm2_dis_kl", "kl")):
        spec = problems.baseline_spec(name)
        spec["batch"] = 2048
        spec["loss"]["method"] = method
        torch.manual_seed(1)
        prob = problems.build(spec, device=dev)
        params = list(prob.ctrl.parameters())
        opt = torch.optim.Adam(params, lr=1e-3, capturable=True)
        torch.manual_seed(5)

        def loss_fn():
            x = prob.prior.sample((2048,))
            return prob.loss(prob.ts, x, prob.target.unnorm_log_prob, prob.second_log_prob)[0]

        step = GraphedTrainStep(loss_fn, [prob.loss], opt, reduce_gradients=lambda: all_reduce_gradients(params), warmup=2, guard=True)
        losses = [float(step()) for _ in range(3)]
        # an eager step of the same kind must not synchronise either
        torch.cuda.synchronize()
        torch.cuda.set_sync_debug_mode("error")
        try:
            opt.zero_grad(set_to_none=True)
            val = loss_fn()
            val.backward()
            all_reduce_gradients(params)
            opt.step()
        finally:
            torch.cuda.set_sync_debug_mode("default")
        torch.cuda.synchronize()
        flat = torch.cat([p.detach().reshape(-1) for p in params])
        out[f"{name}/graphed"] = {"losses_hex": [v.hex() for v in losses], "params_sum_hex": float(flat.double().sum()).hex(),
                                  "params_abs_hex": float(flat.double().abs().sum()).hex(), "skipped": int(step.n_skipped)}
    return out


def main():
    assert torch.cuda.is_available()
    torch.cuda.set_device(0)
    ref = run_all("no group")
    ref.update(run_graphed("no group"))
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29541")
    os.environ["RANK"], os.environ["WORLD_SIZE"] = "0", "1"
    dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
    assert dist.get_backend() == "nccl" and dist.get_world_size() == 1
    calls = {"all_gather": 0, "all_reduce": 0}
    ag, ar = dist.all_gather_into_tensor, dist.all_reduce

    def counted_ag(out, inp, *a, **k):
        assert inp.is_cuda and out.is_cuda, "the estimator statistics must be gathered on the device (RCCL)"
        calls["all_gather"] += 1
        return ag(out, inp, *a, **k)

    def counted_ar(t, *a, **k):
        assert t.is_cuda, "RCCL all-reduce on a host tensor"
        calls["all_reduce"] += 1
        return ar(t, *a, **k)

    dist.all_gather_into_tensor, dist.all_reduce = counted_ag, counted_ar
    got = run_all("nccl world 1")
    n_eager = dict(calls)
    got.update(run_graphed("nccl world 1"))
    calls["graphed_all_reduce"] = calls["all_reduce"] - n_eager["all_reduce"]
    calls["all_reduce"] = n_eager["all_reduce"]
    dist.barrier()
    torch.cuda.synchronize()
    dist.all_gather_into_tensor, dist.all_reduce = ag, ar
    dist.destroy_process_group()
    print(json.dumps({"ref": ref, "got": got, "calls": calls}))


if __name__ == "__main__":
    main()

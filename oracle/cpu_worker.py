"""One single-threaded process of bench.py's process-parallel CPU baseline (TEST / MEASUREMENT INFRASTRUCTURE like the rest of oracle/:
only bench.py's cpu_baseline leg starts it; the product never does).  The reference's CPU path (restated in em_oracle.py; `sample_time`
semantics of /root/reference/sde_sampler/solver/oc.py:88-97) does not scale with intra-op threads -- its per-step tensors are too small
-- so a host is filled with N independent one-thread processes, each integrating its own chunk of trajectories:

    python -m oracle.cpu_worker <job.pkl>      ->  one line "<chunks finished inside the window> <1 if ready before the window opened>"

job = (spec, params, target tensors, inference params, chunk, n_intervals | None, t_start, window, seed)."""
import pickle
import sys
import time


def main(path: str) -> None:
    import torch

    torch.set_num_threads(1)
    from oracle import em_oracle as eo

    with open(path, "rb") as fh:
        spec, params, tt, params_inf, chunk, n_intervals, t_start, window, seed = pickle.load(fh)
    oracle = eo.Problem(spec, params, tt, params_inf=params_inf)
    ts = oracle.grid()
    if n_intervals is not None:
        ts = ts[:n_intervals + 1]
    d = spec["target"]["dim"]
    torch.manual_seed(seed)
    x0 = torch.zeros(chunk, d) if spec["prior"]["kind"] == "delta" else torch.randn(chunk, d)
    oracle.eval(ts, x0, None, compute_weights=False)  # warm-up chunk
    ready = time.time() <= t_start
    while time.time() < t_start:
        time.sleep(0.005)
    done = 0
    while True:
        oracle.eval(ts, x0, None, compute_weights=False)
        if time.time() > t_start + window:
            break
        done += 1
    print(f"{done} {int(ready)}", flush=True)


if __name__ == "__main__":
    main(sys.argv[1])

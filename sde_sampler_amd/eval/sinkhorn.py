"""`Sinkhorn`: same constructor / `compute` / `__call__` contracts as the reference's sde_sampler/eval/sinkhorn.py:10-196
(the `eval_sample_losses.sinkhorn` entry of conf/base.yaml:13-15), executed by libsdeh.so (`sdeh_sinkhorn`,
include/sdeh.h) instead of pykeops LazyTensors: tiled [n, m] log-sum-exp sweeps in fp32, the distance matrix never
materialised, the convergence test on the device.  There is no CPU path."""
from __future__ import annotations

import ctypes as C

import torch

from .. import _lib as L


class Sinkhorn:
    """Entropy-regularised p-Wasserstein cost between two d-dimensional point clouds (p = 1 or 2)."""

    def __init__(self, p: float = 2, eps: float = 1e-3, max_iters: int = 100, stop_thresh: float = 1e-5,
                 verbose: bool = False, n_max: int | None = None, **kwargs):
        if not isinstance(p, int):
            raise TypeError(f"p must be an integer greater than 0, got {p}")
        if p <= 0:
            raise ValueError(f"p must be an integer greater than 0, got {p}")
        self.p = p
        if eps <= 0:
            raise ValueError("Entropy regularization term eps must be > 0")
        self.eps = eps
        if not isinstance(max_iters, int) or max_iters <= 0:
            raise TypeError(f"max_iters must be an integer > 0, got {max_iters}")
        self.max_iters = max_iters
        if not isinstance(stop_thresh, float):
            raise TypeError(f"stop_thresh must be a float, got {stop_thresh}")
        self.stop_thresh = stop_thresh
        self.n_max = n_max
        self.verbose = verbose
        self.last_info: dict = {}  # iterations run and the last potential updates of the most recent call

    @staticmethod
    def _weights(w, pts, name):
        if len(w.shape) > 1:
            w = w.squeeze()
        if len(w.shape) != 1:
            raise ValueError(f"{name} must have shape [n,] or [n, 1] where x.shape = [n, d], but got {name}.shape = {w.shape}")
        if w.shape[0] != pts.shape[0]:
            raise ValueError(f"{name} must match its point cloud in dimension 0 but got {tuple(pts.shape)} and {tuple(w.shape)}")
        return w

    def compute(self, x: torch.Tensor, y: torch.Tensor, w_x: torch.Tensor | None = None,
                w_y: torch.Tensor | None = None, correspondences: bool = True):
        """Returns (distance, corrs_x_to_y [n], corrs_y_to_x [m]) like the reference (`correspondences=False` skips the two
        argmax sweeps and returns None for them)."""
        if len(x.shape) != 2:
            raise ValueError(f"x must be an [n, d] tensor but got shape {x.shape}")
        if len(y.shape) != 2:
            raise ValueError(f"x must be an [m, d] tensor but got shape {y.shape}")
        if x.shape[1] != y.shape[1]:
            raise ValueError(f"x and y must match in the last dimension (i.e. x.shape=[n, d], y.shape[m, d]) "
                             f"but got x.shape = {x.shape}, y.shape={y.shape}")
        if (w_x is None) != (w_y is None):
            raise ValueError("If w_x is not None, w_y must also be not None" if w_y is None
                             else "If w_y is not None, w_x must also be not None")
        if not x.is_cuda:
            raise RuntimeError("Sinkhorn runs on the HIP device only (got a CPU tensor); there is no CPU path in this package")
        dev = x.device
        prep = lambda t: t.detach().to(device=dev, dtype=torch.float32).contiguous()
        x, y = prep(x), prep(y)
        if w_x is not None:
            w_x, w_y = prep(self._weights(w_x, x, "w_x")), prep(self._weights(w_y, y, "w_y"))
            sum_w_x, sum_w_y = w_x.sum().item(), w_y.sum().item()
            if abs(sum_w_x - sum_w_y) > 1e-5:
                raise ValueError(f"Weights w_x and w_y do not sum to the same value, got w_x.sum() = {sum_w_x} and "
                                 f"w_y.sum() = {sum_w_y} (absolute difference = {abs(sum_w_x - sum_w_y)}")
        if self.p > 2:
            raise L.SdehUnsupported(-2, f"Sinkhorn: p={self.p} (the kernels implement p = 1 and p = 2)")
        n, m, d = x.shape[0], y.shape[0], x.shape[1]
        lib = L.load()
        work = torch.empty(lib.sdeh_sinkhorn_workspace_floats(n, m), device=dev, dtype=torch.float32)
        out = torch.empty(4, device=dev, dtype=torch.float32)
        c1 = torch.empty(n, device=dev, dtype=torch.int64) if correspondences else None
        c2 = torch.empty(m, device=dev, dtype=torch.int64) if correspondences else None
        ptr = lambda t: None if t is None else t.data_ptr()
        with torch.cuda.device(dev):
            L.check(lib.sdeh_sinkhorn(x.data_ptr(), n, y.data_ptr(), m, d, ptr(w_x), ptr(w_y), self.p, float(self.eps),
                                      self.max_iters, float(self.stop_thresh), work.data_ptr(), out.data_ptr(), ptr(c1),
                                      ptr(c2), torch.cuda.current_stream(dev).cuda_stream))
        self.last_info = {"out": out}
        return out[0], c1, c2

    def info(self) -> dict:
        """{"iterations", "max_err_u", "max_err_v"} of the most recent call (synchronises)."""
        o = self.last_info["out"].tolist()
        return {"iterations": int(o[1]), "max_err_u": o[2], "max_err_v": o[3]}

    def __call__(self, x: torch.Tensor, y: torch.Tensor, w_x: torch.Tensor | None = None,
                 w_y: torch.Tensor | None = None):
        if self.n_max is not None:
            x, y = x[: self.n_max], y[: self.n_max]
            if w_x is not None:
                w_x = w_x[: self.n_max]
            if w_y is not None:
                w_y = w_y[: self.n_max]
        return self.compute(x, y, w_x=w_x, w_y=w_y, correspondences=False)[0]

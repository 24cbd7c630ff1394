"""GPU tests of the evaluation-side kernels: sdeh_sample_stats behind get_metrics (vs the reference's golden output) and
sdeh_sinkhorn behind Sinkhorn (vs the dense oracle and the closed-form 1-D transport cost)."""
import math
from pathlib import Path

import numpy as np
import pytest
import torch

from oracle import eval_oracle as ev
from tests.helpers import GOLDEN_METRICS, load_metrics_fixture

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


class _Distr:
    """The attributes get_metrics reads from a Distribution (eval/metrics.py:70-184)."""

    def __init__(self, dim, stats, fx):
        self.dim = dim
        self.expectations = stats["expectations"]
        self.log_norm_const = stats["log_norm_const"]
        self.stddevs = torch.from_numpy(fx["stddevs"]).to(DEV) if stats["has_stddevs"] else None
        self.domain = torch.from_numpy(fx["domain"]).to(DEV) if stats["has_domain"] else None


@pytest.mark.parametrize("path", GOLDEN_METRICS, ids=lambda p: Path(p).stem)
def test_get_metrics_matches_reference_golden(path):
    from sde_sampler_amd.eval.metrics import get_metrics

    fx, meta, expected = load_metrics_fixture(path)
    distr = _Distr(meta["target"]["dim"], meta["stats"], fx)
    samples, weights = torch.from_numpy(fx["samples"]).to(DEV), torch.from_numpy(fx["weights"]).to(DEV)
    for tag, w in (("w", weights), ("nw", None)):
        got = get_metrics(distr, samples, weights=w, log_norm_const_preds=meta["stats"]["preds"],
                          marginal_dims=list(meta["marginal_dims"]))
        assert set(got) == set(expected[tag]), set(got) ^ set(expected[tag])
        for k, v in expected[tag].items():
            # fp32 reductions in a different order (the reference: ATen's pairwise sums; here Welford / block sums)
            assert got[k] == pytest.approx(v, rel=2e-5, abs=2e-6), (k, got[k], v)


@pytest.mark.parametrize("B,d", [(1, 3), (7, 1), (1000, 50), (65536, 2), (300_001, 10), (513, 200)])
def test_sample_stats_against_torch(B, d):
    from sde_sampler_amd.eval.metrics import sample_stats

    torch.manual_seed(B + d)
    x = (torch.randn(B, d, device=DEV) * 3 + torch.arange(d, device=DEV) * 0.5).contiguous()
    w = torch.rand(B, 1, device=DEV) + 0.1
    dom = torch.stack([torch.full((d,), -4.0), torch.full((d,), 6.0 + d)], dim=1).to(DEV)
    st = sample_stats(x, weights=w, domain=dom)
    xd, wd = x.double(), w.double()
    assert st["n"] == B
    assert st["sum_w"] == pytest.approx(wd.sum().item(), rel=1e-5)
    assert st["sum_w2"] == pytest.approx((wd**2).sum().item(), rel=1e-5)
    assert st["inside"] == ((dom[:, 0] <= x) & (x <= dom[:, 1])).all(-1).sum().item()
    fs = [(xd**2).sum(-1), xd.abs().sum(-1), xd.sum(-1), (xd**2 - xd).sum(-1)]
    for k, f in enumerate(fs):
        assert st["f"][k].item() == pytest.approx(f.sum().item(), rel=2e-5, abs=1e-3)
        assert st["wf"][k].item() == pytest.approx((f * wd[:, 0]).sum().item(), rel=2e-5, abs=1e-3)
    assert torch.allclose(st["mean"], xd.mean(0).cpu(), rtol=1e-5, atol=1e-5)
    if B > 1:
        assert torch.allclose(st["m2"] / (B - 1), xd.var(0).cpu(), rtol=1e-4, atol=1e-6)


@pytest.mark.parametrize("n,m,d,p,weighted", [(300, 257, 2, 2, False), (1000, 1000, 10, 2, False), (513, 64, 3, 1, False),
                                              (400, 300, 2, 2, True), (64, 1500, 50, 2, False)])
def test_sinkhorn_against_dense_oracle(n, m, d, p, weighted):
    from sde_sampler_amd.eval.sinkhorn import Sinkhorn

    torch.manual_seed(n + m + d)
    x, y = torch.randn(n, d) * 1.5, torch.randn(m, d) + 0.7
    w_x = w_y = None
    if weighted:
        w_x, w_y = torch.rand(n) + 0.5, torch.rand(m) + 0.5
        w_x, w_y = w_x / w_x.sum(), w_y / w_y.sum()
    kw = dict(p=p, eps=5e-2, max_iters=60, stop_thresh=1e-5)  # eps large enough for fp32 dense logsumexp to be a fair oracle
    ref_d, ref_c1, ref_c2, ref_it = ev.sinkhorn_dense(x.double(), y.double(), None if w_x is None else w_x.double(),
                                                       None if w_y is None else w_y.double(), **kw)
    sk = Sinkhorn(**kw)
    gpu = lambda t: None if t is None else t.to(DEV)
    dist, c1, c2 = sk.compute(gpu(x), gpu(y), gpu(w_x), gpu(w_y))
    info = sk.info()
    assert abs(info["iterations"] - ref_it) <= 1
    assert dist.item() == pytest.approx(ref_d.item(), rel=2e-4)
    # the transport plan's argmax per row / column.  Only the well-posed direction is compared: with n < m every row splits
    # its mass over ~m/n columns whose plan entries are EQUAL at convergence (each equals the column marginal), so the row
    # argmax is decided by rounding noise -- in the oracle as much as here.
    if n >= m:
        assert (c1.cpu() == ref_c1).float().mean() > 0.98
    if m >= n:
        assert (c2.cpu() == ref_c2).float().mean() > 0.98
    assert c1.shape == (n,) and c2.shape == (m,) and int(c1.max()) < m and int(c2.max()) < n
    # __call__ == compute()[0]; n_max truncation (sinkhorn.py:187-196)
    assert Sinkhorn(**kw)(gpu(x), gpu(y), gpu(w_x), gpu(w_y)).item() == pytest.approx(dist.item(), rel=1e-6)
    if not weighted:
        small = Sinkhorn(n_max=50, **kw)(gpu(x), gpu(y)).item()
        ref_small = ev.sinkhorn_dense(x[:50].double(), y[:50].double(), **kw)[0].item()
        assert small == pytest.approx(ref_small, rel=2e-4)


def test_sinkhorn_converged_cost_matches_exact_1d_transport():
    """In one dimension the optimal-transport cost with the Euclidean ground metric is mean |x_(i) - y_(i)| over the sorted
    samples; the converged entropic cost is within O(eps) of it."""
    from sde_sampler_amd.eval.sinkhorn import Sinkhorn

    torch.manual_seed(0)
    n = 4096
    x = torch.randn(n, 1, device=DEV)
    y = torch.randn(n, 1, device=DEV) * 1.5 + 2.0
    exact = (x[:, 0].sort().values - y[:, 0].sort().values).abs().mean().item()
    sk = Sinkhorn(eps=0.05, max_iters=3000, stop_thresh=1e-4)
    got = sk(x, y).item()
    assert got == pytest.approx(exact, rel=0.05), (got, exact, sk.info())
    assert sk.info()["iterations"] < 3000  # the device-side convergence test fired
    assert Sinkhorn(eps=0.05, max_iters=3000, stop_thresh=1e-4)(x, y).item() == got  # deterministic
    assert Sinkhorn(eps=0.05, max_iters=500)(x, x.clone()).item() < 0.05  # a cloud against itself


@pytest.mark.parametrize("d,n,eps", [(2, 128, 0.02), (5, 160, 0.05)])
def test_sinkhorn_cost_brackets_the_exact_assignment_cost(d, n, eps):
    """d > 1, uniform weights, n = m: the exact optimal-transport cost is an assignment problem (scipy.optimize.linear_sum_assignment on
    the Euclidean cost matrix).  The entropic plan's transport cost <P, M> -- what eval/sinkhorn.py:169-177 returns -- lies in
    [OT, OT + eps log n] once the potentials have settled: an anchor that depends on no restatement of the reference's iteration."""
    import math

    from scipy.optimize import linear_sum_assignment

    from sde_sampler_amd.eval.sinkhorn import Sinkhorn

    torch.manual_seed(d)
    x, y = torch.randn(n, d) * 1.5, torch.randn(n, d) + 0.7
    M = torch.cdist(x.double(), y.double()).numpy()
    r, c = linear_sum_assignment(M)
    exact = float(M[r, c].mean())
    got = Sinkhorn(eps=eps, max_iters=3000, stop_thresh=1e-6)(x.to(DEV), y.to(DEV)).item()
    assert exact - 1e-3 <= got <= exact + eps * math.log(n), (got, exact)


def test_sinkhorn_reference_settings_against_dense_oracle():
    """conf/base.yaml:13-15 instantiates Sinkhorn() with its defaults (eps = 1e-3, at most 100 iterations) -- far from
    converged at that eps, so the value is that of the 100th iterate: compared with the fp64 dense oracle's 100th iterate."""
    from sde_sampler_amd.eval.sinkhorn import Sinkhorn

    torch.manual_seed(2)
    x, y = torch.randn(1024, 2) * 2.0, torch.randn(1024, 2) * 1.5 + 1.0
    ref, _, _, it = ev.sinkhorn_dense(x.double(), y.double())
    sk = Sinkhorn()
    got = sk(x.to(DEV), y.to(DEV)).item()
    assert it == 100 and sk.info()["iterations"] == 100
    assert got == pytest.approx(ref.item(), rel=5e-3), (got, ref.item())


def test_sinkhorn_full_eval_batch_runs_and_is_finite():
    """cfg2's eval batch (65 536 x 65 536 pairs per half-iteration, d = 2): finishes and gives a finite, positive cost."""
    from sde_sampler_amd.eval.sinkhorn import Sinkhorn

    torch.manual_seed(1)
    x = torch.randn(65536, 2, device=DEV) * 10
    y = torch.randn(65536, 2, device=DEV) * 10 + 1.0
    sk = Sinkhorn(max_iters=10)
    v = sk(x, y).item()
    assert math.isfinite(v) and 0.0 < v < 5.0, v
    assert sk.info()["iterations"] == 10

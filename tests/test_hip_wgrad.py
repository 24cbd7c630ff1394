"""sdeh_weight_grad (include/sdeh.h): the weight-gradient contraction over the N = T*B rows against torch in float64."""
import ctypes as C

import pytest
import torch

from sde_sampler_amd import _lib as L

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _run(dmat, z, act, chunk):
    m, N = dmat.shape
    c = z.shape[0]
    n_chunks = -(-N // chunk)
    mp, cp = 64 * ((m + 63) // 64), 64 * ((c + 63) // 64)  # partials are padded to multiples of 64 (wide layers: [64, 64] blocks)
    pw = torch.full((n_chunks, mp, cp), float("nan"), device=DEV)
    pb = torch.full((n_chunks, mp), float("nan"), device=DEV)
    L.check(L.load().sdeh_weight_grad(dmat.data_ptr(), m, z.data_ptr(), c, N, act, chunk, pw.data_ptr(), pb.data_ptr(),
                                      torch.cuda.current_stream().cuda_stream))
    return pw, pb


ACTS = {L.ACT_GELU_ERF: torch.nn.functional.gelu, L.ACT_SILU: torch.nn.functional.silu, L.ACT_RELU: torch.relu,
        L.ACT_IDENTITY: lambda v: v}


@pytest.mark.parametrize("m,c,N,chunk", [(64, 64, 4096, 128), (64, 64, 5000, 256), (1, 64, 777, 64), (50, 64, 2049, 8),
                                         (64, 2, 1003, 128), (33, 17, 130, 128), (10, 64, 100 * 257, 1024),
                                         # layers of the wide networks (round 3): blocked over [64, 64] tiles, 2 x 2 per workgroup
                                         (256, 256, 3000, 512), (196, 256, 1111, 128), (256, 196, 2050, 256), (128, 70, 777, 64),
                                         (65, 129, 300, 32)])
@pytest.mark.parametrize("act", sorted(ACTS))
def test_weight_grad_matches_float64(m, c, N, chunk, act):
    torch.manual_seed(m * 1000 + c + N)
    dmat = torch.randn(m, N, device=DEV)
    z = torch.randn(c, N, device=DEV) * 2.0
    pw, pb = _run(dmat, z, act, chunk)
    assert torch.isfinite(pw).all() and torch.isfinite(pb).all()  # every partial is written, padding rows / columns are zero
    w, b = pw.sum(0), pb.sum(0)
    assert (w[m:] == 0).all() and (w[:, c:] == 0).all() and (b[m:] == 0).all()
    ref_w = dmat.double() @ ACTS[act](z.double()).t()
    ref_b = dmat.double().sum(1)
    # tolerance: 5e-6 of each sum's condition (sum of |terms|) + the activation's own absolute accuracy (2e-7) times sum |D|
    tol = 5e-6 * (dmat.double().abs() @ ACTS[act](z.double()).abs().t()) + 2e-7 * dmat.double().abs().sum(1, keepdim=True)
    assert ((w[:m, :c].double() - ref_w).abs() <= tol).all()
    torch.testing.assert_close(b[:m].double(), ref_b, rtol=0, atol=2e-6 * float(dmat.abs().sum(1).max()))
    # per-chunk partials: chunk k holds exactly the rows [k chunk, (k+1) chunk)
    k = pw.shape[0] - 1
    sl = slice(k * chunk, N)
    ref_k = dmat[:, sl].double() @ ACTS[act](z[:, sl].double()).t()
    tol_k = 5e-6 * (dmat[:, sl].double().abs() @ ACTS[act](z[:, sl].double()).abs().t()) + 2e-7 * dmat[:, sl].double().abs().sum(1, keepdim=True)
    assert ((pw[k, :m, :c].double() - ref_k).abs() <= tol_k).all()


def test_weight_grad_unaligned_views_and_errors():
    torch.manual_seed(0)
    big = torch.randn(3, 64, 1001, device=DEV)  # N % 4 != 0: plane k starts at an odd multiple of 4 bytes
    dmat, z = big[1], big[2]
    pw, pb = _run(dmat, z, L.ACT_GELU_ERF, 128)
    ref = dmat.double() @ torch.nn.functional.gelu(z.double()).t()
    assert (pw.sum(0).double() - ref).abs().max() < 1e-3
    lib = L.load()
    args = (dmat.data_ptr(), 64, z.data_ptr(), 64, 1001, 0, 128, pw.data_ptr(), pb.data_ptr(), None)
    for bad in [dict(chunk=12), dict(m=257), dict(c=0), dict(act=7)]:
        a = list(args)
        if "chunk" in bad: a[6] = bad["chunk"]
        if "m" in bad: a[1] = bad["m"]
        if "c" in bad: a[3] = bad["c"]
        if "act" in bad: a[5] = bad["act"]
        assert lib.sdeh_weight_grad(*a) < 0

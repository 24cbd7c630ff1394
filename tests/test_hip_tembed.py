"""sdeh_time_embed_backward (include/sdeh.h): TimeEmbed parameter gradients from the gradient of its [T, dim_out] table, against
autograd on the module (models/mlp.py:43-82)."""
import pytest
import torch

from sde_sampler_amd.losses._autograd import _time_embed_grads
from sde_sampler_amd.models.mlp import TimeEmbed
from sde_sampler_amd import engine as E

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.mark.parametrize("act", [torch.nn.GELU, torch.nn.SiLU, torch.nn.ReLU])
@pytest.mark.parametrize("num_layers,dim_out,T,clip", [(2, 64, 100, None), (4, 1, 100, True), (4, 7, 33, None), (3, 2, 401, True),
                                                       (5, 1, 64, None)])
def test_time_embed_backward_matches_autograd(act, num_layers, dim_out, T, clip):
    torch.manual_seed(num_layers * 100 + dim_out)
    te = TimeEmbed(dim_out=dim_out, activation=act(), num_layers=num_layers, channels=64).to(DEV)
    with torch.no_grad():
        te.out_layer.weight.normal_(0.0, 0.3)
        te.out_layer.bias.normal_(0.0, 0.3)
    ts = torch.linspace(0.0, 1.7, T, device=DEV)
    gtab = torch.randn(T, dim_out, device=DEV)
    params = list(te.parameters())
    out = te(ts)
    if clip is not None:
        mags = out.detach().abs().flatten().sort().values  # clamp active on about half of the entries, threshold in the widest gap
        mid = mags[mags.numel() // 4: 3 * mags.numel() // 4]
        k = int((mid[1:] - mid[:-1]).argmax())
        clip = float(0.5 * (mid[k] + mid[k + 1]))
        out = out.clip(min=-clip, max=clip)
    ref = torch.autograd.grad(out, params, gtab)
    got = _time_embed_grads(te, E._activation_id(te.activation), ts, gtab, clip)
    assert got is not None and set(got) == {id(p) for p in params}
    for (name, p), r in zip(te.named_parameters(), ref):
        g = got[id(p)]
        assert g.shape == r.shape
        scale = float(r.abs().max()) + 1e-12
        assert float((g - r).abs().max()) <= 2e-5 * scale + 1e-6, (name, float((g - r).abs().max()), scale)


def test_time_embed_backward_declines_deep_networks():
    te = TimeEmbed(dim_out=1, activation=torch.nn.GELU(), num_layers=7, channels=64).to(DEV)
    assert _time_embed_grads(te, 0, torch.zeros(4, device=DEV), torch.zeros(4, 1, device=DEV), None) is None

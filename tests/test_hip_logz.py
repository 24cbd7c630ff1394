"""log Z on the metric's headline workload (GMM-40 d=50, basic_pis) with a TRAINED control -- tests/golden/trained_pis_gmm50.pt, produced
with the HIP training path by tools/train_headline_control.py (ESS/B = 0.88; like PIS in the literature it settles on one of the 40
modes: log Z_is = log(1/40) = -3.69).  With a control whose importance weights are not degenerate the north star's
"log Z within +-0.01 of the reference" can be checked: parity mode against the oracle on identical noise, fast mode (in-kernel
noise) against the oracle's torch noise within the Monte-Carlo error."""
import math
from pathlib import Path

import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
FIXTURE = Path(__file__).parent / "golden" / "trained_pis_gmm50.pt"


def _problem(batch):
    from sde_sampler_amd import problems

    state = torch.load(FIXTURE, map_location="cpu")
    spec = problems.baseline_spec("gmm50_pis_headline")
    spec["batch"] = batch
    prob = problems.build(spec, params=state["params"])
    tt = dict(loc=prob.target.loc.clone(), scale=prob.target.scale.clone(), mixture_weights=prob.target.mixture_weights.clone())
    return spec, state["params"], tt, prob


def test_trained_control_log_z_matches_oracle_on_identical_noise():
    from oracle import em_oracle as eo

    B = 4096
    spec, params, tt, prob = _problem(B)
    T, d = prob.ts.numel() - 1, 50
    torch.manual_seed(23)
    x0, noise = prob.prior.sample((B,)), torch.randn(T, B, d)
    threads = torch.get_num_threads()
    torch.set_num_threads(min(16, threads))
    try:
        ref = eo.Problem(spec, params, tt).eval(prob.ts.clone(), x0.clone(), noise, compute_weights=True)
    finally:
        torch.set_num_threads(threads)
    prob.to(DEV)
    out = prob.eval(x0.to(DEV), compute_weights=True, noise=noise.to(DEV))
    assert abs(out.log_norm_const_preds["log_norm_const_is"] - ref["log_norm_const_is"]) <= 1e-4
    assert abs(out.log_norm_const_preds["log_norm_const_lb_ito"] - ref["log_norm_const_lb_ito"]) <= 1e-4
    w = ref["weights"].double().flatten()
    assert float(w.sum() ** 2 / (w * w).sum()) / B > 0.5  # the control really has an effective sample size


def test_trained_control_fast_mode_log_z_within_0p01_of_the_oracle():
    """B = 65 536 in-kernel noise against 16 384 oracle rows with torch noise: |delta log Z_is| <= 0.01 (the north star's bar) and
    within 4 combined standard errors."""
    from oracle import em_oracle as eo
    from sde_sampler_amd import engine as E

    B, Bo = 65536, 16384
    spec, params, tt, prob = _problem(B)
    torch.manual_seed(29)
    threads = torch.get_num_threads()
    torch.set_num_threads(min(16, threads))
    try:
        _, rnd_o, _ = eo.Problem(spec, params, tt).simulate(prob.ts.clone(), prob.prior.sample((Bo,)), None, compute_ito_int=True)
    finally:
        torch.set_num_threads(threads)
    prob.to(DEV)
    with torch.no_grad():
        _, rnd, _ = prob.loss.simulate(prob.ts, prob.prior.sample((B,)), prob.target.unnorm_log_prob, prob.second_log_prob,
                                       compute_ito_int=True)

    def log_z(r):
        neg = -r.double().flatten()
        m = neg.max()
        w = torch.exp(neg - m)
        return float(torch.log(w.mean()) + m), float(w.std() / w.mean() / math.sqrt(w.numel())), float(w.sum() ** 2 / (w * w).sum())

    z_g, se_g, ess_g = log_z(rnd.cpu())
    z_o, se_o, ess_o = log_z(rnd_o)
    est = E.estimators_from_stats(E.merge_stats(E.estimator_stats(rnd)))
    assert abs(est["log_norm_const_is"] - z_g) <= 1e-5  # the device reduction == the float64 host formula
    assert ess_g / B > 0.5 and ess_o / Bo > 0.5
    assert abs(z_g - z_o) <= 0.01, (z_g, z_o)
    assert abs(z_g - z_o) <= 4.0 * math.hypot(se_g, se_o) + 1e-4, (z_g, z_o, se_g, se_o)

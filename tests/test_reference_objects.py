"""The boundary claim of DESIGN.md section 1, tested in the build container: `engine.build_problem` introspects the REFERENCE's own
objects (sde_sampler.models.reparam.*Ctrl over sde_sampler.models.mlp.FourierMLP / TimeEmbed, sde_sampler.eq.sdes.*,
sde_sampler.distr.*) exactly like this package's host classes, and the shipped conf/*_hip.yaml files carry the keys of the
reference's own YAMLs with `_target_` pointing at the engine's classes.  Skipped where /root/reference does not exist (GPU box)."""
import importlib
import math
import sys
import types
from functools import partial
from pathlib import Path

import pytest
import torch
import yaml

ROOT = Path(__file__).resolve().parents[1]
REFERENCE = Path("/root/reference")
pytestmark = pytest.mark.skipif(not REFERENCE.exists(), reason="reference checkout not present (build container only)")


@pytest.fixture(scope="module")
def ref():
    for name, attrs in {"wandb": {"run": None, "log": lambda *a, **k: None}, "torchquad": {"Boole": object},
                        "torchsde": {"BaseBrownian": object}}.items():
        if name not in sys.modules:
            mod = types.ModuleType(name)
            for k, v in attrs.items():
                setattr(mod, k, v)
            sys.modules[name] = mod
    sys.path.insert(0, str(REFERENCE))
    try:
        yield types.SimpleNamespace(
            mlp=importlib.import_module("sde_sampler.models.mlp"), reparam=importlib.import_module("sde_sampler.models.reparam"),
            sdes=importlib.import_module("sde_sampler.eq.sdes"), gauss=importlib.import_module("sde_sampler.distr.gauss"),
            funnel=importlib.import_module("sde_sampler.distr.funnel"), dw=importlib.import_module("sde_sampler.distr.double_well"),
            delta=importlib.import_module("sde_sampler.distr.delta"))
    finally:
        sys.path.remove(str(REFERENCE))


def _nets(ref, dim, channels=64, gamma_dim=1):
    act = torch.nn.GELU()
    base = ref.mlp.FourierMLP(dim=dim, activation=act, num_layers=4, channels=channels)
    gamma = ref.mlp.TimeEmbed(dim_out=gamma_dim, activation=act, num_layers=4, channels=channels,
                              last_bias_init=partial(torch.nn.init.constant_, val=1.0))
    return base, gamma


def test_reference_lerp_ctrl_vp_gmm_maps_onto_the_problem_struct(ref):
    from sde_sampler_amd import _lib as L
    from sde_sampler_amd import engine as E
    from sde_sampler_amd.losses import oc

    target = ref.gauss.GMM(dim=2, name="fab", n_reference_samples=100)
    prior = ref.gauss.IsotropicGauss(dim=2)
    sde = ref.sdes.VP(diff_coeff_sq_min=0.1, diff_coeff_sq_max=10.0, terminal_t=1.0)
    base, gamma = _nets(ref, 2)
    ctrl = ref.reparam.LerpCtrl(base_model=base, score_model=gamma, target_score=target.score, prior_score=prior.score, sde=sde,
                                detach_score=False, clip_score=1e4, clip_model=25.0, scale_score=0.5)
    tgt, clip_target = oc._resolve_terminal(target.unnorm_log_prob)
    assert tgt is target and clip_target is None
    assert oc._resolve_gaussian_log_prob(prior.log_prob) is prior
    keep = E._Keep()
    pr = E.TrajectoryEngine().build_problem(loss_kind=L.LOSS_TIME_REVERSAL, generative_ctrl=ctrl, sde=sde, flags=L.FLAG_ITO,
                                            device=torch.device("cpu"), keep=keep, terminal_target=tgt, second=prior)
    assert (pr.ctrl_kind, pr.sde_kind) == (L.CTRL_LERP, L.SDE_VP)
    assert (pr.clip_model, pr.clip_score, pr.scale_score) == (25.0, 1e4, 0.5) and math.isinf(pr.clip_target)
    assert (pr.vp_beta_min, pr.vp_beta_max, pr.vp_scale, pr.terminal_t) == (pytest.approx(0.1), 10.0, 1.0, 1.0)
    net = pr.base_model
    assert (net.dim, net.channels, net.n_hidden, net.activation) == (2, 64, 2, L.ACT_GELU_ERF)
    assert net.input_w == base.input_embed.weight.data_ptr() and net.out_b == base.out_layer.bias.data_ptr()
    assert net.hidden_w[1] == base.hidden_layer[1].weight.data_ptr()
    assert net.timestep_embed.n_hidden == 1 and net.timestep_embed.dim_out == 64
    assert net.timestep_embed.hidden_w[0] == base.timestep_embed.hidden_layer[0].weight.data_ptr()
    assert (pr.score_model.n_hidden, pr.score_model.dim_out) == (3, 1)
    assert pr.score_model.out_b == gamma.out_layer.bias.data_ptr()
    assert (pr.target.kind, pr.target.dim, pr.target.n_components) == (L.DENS_GMM, 2, 40)
    assert pr.target.flags & L.DENS_FLAG_SHARED_SCALE  # every named mixture of the reference shares its scales
    assert (pr.prior.kind, pr.second.kind) == (L.DENS_DIAG_GAUSS, L.DENS_DIAG_GAUSS)
    assert not pr.flags & (L.FLAG_DETACH_SCORE | L.FLAG_TARGET_SCORE_CONST)


def test_reference_score_ctrl_bm_funnel_and_detach_flag(ref):
    from sde_sampler_amd import _lib as L
    from sde_sampler_amd import engine as E

    target = ref.funnel.Funnel(dim=10, n_reference_samples=100)
    prior = ref.delta.Delta(dim=10)
    sde = ref.sdes.ScaledBM(diff_coeff=math.sqrt(0.2), terminal_t=5.0)
    base, gamma = _nets(ref, 10, gamma_dim=10)
    ctrl = ref.reparam.ScoreCtrl(base_model=base, score_model=gamma, target_score=target.score)  # detach_score defaults to True
    reference = sde.marginal_distr(t=sde.terminal_t, x_init=prior.loc)
    pr = E.TrajectoryEngine().build_problem(loss_kind=L.LOSS_REFERENCE_SDE, generative_ctrl=ctrl, sde=sde, flags=0,
                                            device=torch.device("cpu"), keep=E._Keep(), terminal_target=target, second=reference)
    assert (pr.ctrl_kind, pr.sde_kind, pr.ou_drift) == (L.CTRL_SCORE, L.SDE_CONST_OU, 0.0)
    assert pr.ou_diff == pytest.approx(math.sqrt(0.2)) and pr.terminal_t == 5.0
    assert math.isinf(pr.clip_model) and math.isinf(pr.clip_score)
    assert (pr.target.kind, pr.target.p0) == (L.DENS_FUNNEL, 9.0)
    assert pr.second.kind == L.DENS_DIAG_GAUSS and pr.score_model.dim_out == 10
    assert pr.flags & L.FLAG_DETACH_SCORE
    ctrl.detach_score = False
    pr = E.TrajectoryEngine().build_problem(loss_kind=L.LOSS_REFERENCE_SDE, generative_ctrl=ctrl, sde=sde, flags=0,
                                            device=torch.device("cpu"), keep=E._Keep(), terminal_target=target, second=reference)
    assert not pr.flags & L.FLAG_DETACH_SCORE


def test_reference_bridge_pair_and_wide_networks(ref):
    """conf/solver/bridge.yaml wiring at configs[4]'s width: LerpTargetCtrl + LerpPriorCtrl over 256-channel FourierMLPs."""
    from sde_sampler_amd import _lib as L
    from sde_sampler_amd import engine as E

    d = 12
    target = ref.dw.MultiWell(dim=d, n_double_wells=3, separation=2.0, shift=0.5)
    prior = ref.gauss.IsotropicGauss(dim=d)
    sde = ref.sdes.ScaledBM(diff_coeff=1.0, terminal_t=1.0)
    base, gamma = _nets(ref, d, channels=256)
    base2, gamma2 = _nets(ref, d, channels=256)
    gen = ref.reparam.LerpTargetCtrl(base_model=base, score_model=gamma, target_score=target.score, prior_score=prior.score, sde=sde,
                                     detach_score=False, clip_score=10.0, clip_model=10.0)
    inf = ref.reparam.LerpPriorCtrl(base_model=base2, score_model=gamma2, target_score=target.score, prior_score=prior.score, sde=sde,
                                    detach_score=False, clip_score=10.0, clip_model=10.0, name="inference_ctrl")
    pr = E.TrajectoryEngine().build_problem(loss_kind=L.LOSS_TIME_REVERSAL, generative_ctrl=gen, sde=sde, flags=0,
                                            device=torch.device("cpu"), keep=E._Keep(), terminal_target=target, second=prior,
                                            inference_ctrl=inf)
    assert pr.flags & L.FLAG_INFERENCE_CTRL and pr.ctrl_kind == L.CTRL_LERP_TARGET
    assert pr.inference.ctrl_kind == L.CTRL_LERP_PRIOR and pr.inference.clip_model == 10.0
    assert pr.base_model.channels == 256 and pr.inference.base_model.channels == 256
    assert pr.inference.base_model.input_w == base2.input_embed.weight.data_ptr()
    assert (pr.target.kind, pr.target.n_components, pr.target.p0, pr.target.p1) == (L.DENS_MULTI_WELL, 3, 2.0, 0.5)


@pytest.mark.parametrize("name", ["time_reversal", "time_reversal_lv", "reference_sde", "reference_sde_lv", "exponential_sde",
                                  "exponential_sde_lv"])
def test_shipped_loss_yaml_mirrors_the_reference_yaml(name):
    ours = yaml.safe_load((ROOT / "conf" / "loss" / f"{name}_hip.yaml").read_text())
    theirs = yaml.safe_load((REFERENCE / "conf" / "loss" / f"{name}.yaml").read_text())
    assert set(ours) == set(theirs)
    module, cls = ours["_target_"].rsplit(".", 1)
    assert module == "sde_sampler_amd.losses.oc" and cls == theirs["_target_"].rsplit(".", 1)[1]
    assert hasattr(importlib.import_module(module), cls)
    assert {k: v for k, v in ours.items() if k != "_target_"} == {k: v for k, v in theirs.items() if k != "_target_"}
    # the class accepts exactly these keyword arguments (Hydra instantiates with them plus the solver's collaborators)
    kwargs = {k: v for k, v in ours.items() if k != "_target_"}
    getattr(importlib.import_module(module), cls)(generative_ctrl=None, sde=None, **kwargs)


def test_shipped_integrator_yaml_mirrors_the_reference_yaml():
    ours = yaml.safe_load((ROOT / "conf" / "integrator" / "euler_hip.yaml").read_text())
    theirs = yaml.safe_load((REFERENCE / "conf" / "integrator" / "euler.yaml").read_text())
    assert {k: v for k, v in ours.items() if k != "_target_"} == {k: v for k, v in theirs.items() if k != "_target_"}
    from sde_sampler_amd.eq.integrator import EulerIntegrator

    assert ours["_target_"] == "sde_sampler_amd.eq.integrator.EulerIntegrator"
    EulerIntegrator(**{k: v for k, v in ours.items() if k != "_target_"})


def _reference_nice():
    """The reference's distr/nice.py with a stand-in for the one torchvision transform its constructor calls (a display-only resize)."""
    if "torchvision" not in sys.modules:
        class Resize:
            def __init__(self, size, antialias=True):
                self.size = size

            def __call__(self, img):
                return torch.nn.functional.interpolate(img.unsqueeze(0), size=self.size, mode="bilinear", antialias=True).squeeze(0)

        tv, tr, ut = types.ModuleType("torchvision"), types.ModuleType("torchvision.transforms"), types.ModuleType("torchvision.utils")
        tr.Resize, ut.make_grid = Resize, None
        tv.transforms, tv.utils = tr, ut
        sys.modules.update({"torchvision": tv, "torchvision.transforms": tr, "torchvision.utils": ut})
    return importlib.import_module("sde_sampler.distr.nice")


def test_reference_nice_objects_map_onto_sdeh_nice(ref):
    """BASELINE configs[4] as written: the REFERENCE's Nice(model=NiceModel(...)) behind a LerpTargetCtrl is described like this package's --
    SdehNice pointers are the reference modules' own parameters, the problem's target is SDEH_DENS_EXTERNAL with the object riding along."""
    from sde_sampler_amd import _lib as L
    from sde_sampler_amd import engine as E

    sys.path.insert(0, str(REFERENCE))
    try:
        rn = _reference_nice()
    finally:
        sys.path.remove(str(REFERENCE))
    torch.manual_seed(5)
    model = rn.NiceModel(prior=rn.StandardLogistic(), coupling=4, in_out_dim=196, mid_dim=500, hidden=5, mask_config=1.0)
    target = rn.Nice(model=model, dim=196, n_reference_samples=1000)
    keep = E._Keep()
    nd = E.describe_nice(target, torch.device("cpu"), keep)
    assert (nd.dim, nd.n_coupling, nd.mid_dim, nd.n_mid) == (196, 4, 500, 4)
    assert [nd.mask_config[i] for i in range(4)] == [1, 0, 1, 0]
    assert nd.in_w[2] == model.coupling[2].in_block[0].weight.data_ptr() and nd.mid_b[3][1] == model.coupling[3].mid_block[1][0].bias.data_ptr()
    assert nd.out_w[0] == model.coupling[0].out_block.weight.data_ptr() and nd.scale == model.scaling.scale.data_ptr()
    # the same seed through this package's mirror gives the same weights (the seed-only fixtures rely on it)
    from sde_sampler_amd.distr import nice as mine
    torch.manual_seed(5)
    twin = mine.NiceModel(prior=mine.StandardLogistic(), coupling=4, in_out_dim=196, mid_dim=500, hidden=5, mask_config=1.0)
    assert all(torch.equal(a, b) for a, b in zip(model.state_dict().values(), twin.state_dict().values()))
    assert list(model.state_dict()) == list(twin.state_dict())
    prior, sde = ref.gauss.IsotropicGauss(dim=196), ref.sdes.ScaledBM(diff_coeff=1.0, terminal_t=1.0)
    base, gamma = _nets(ref, 196, channels=256)
    gen = ref.reparam.LerpTargetCtrl(base_model=base, score_model=gamma, target_score=target.score, prior_score=prior.score, sde=sde,
                                     detach_score=False, clip_score=10.0, clip_model=10.0)
    keep = E._Keep()
    pr = E.TrajectoryEngine().build_problem(loss_kind=L.LOSS_TIME_REVERSAL, generative_ctrl=gen, sde=sde, flags=0, device=torch.device("cpu"),
                                            keep=keep, terminal_target=None, second=prior)
    assert pr.target.kind == L.DENS_EXTERNAL and pr.target.dim == 196 and pr.ctrl_kind == L.CTRL_LERP_TARGET
    assert [k.obj for k in keep if isinstance(k, E._ExternalTarget)] == [target]


def test_shipped_target_yaml_mirrors_the_reference_yaml():
    ours = yaml.safe_load((ROOT / "conf" / "target" / "nice_hip.yaml").read_text())
    theirs = yaml.safe_load((REFERENCE / "conf" / "target" / "nice.yaml").read_text())
    assert ours["_target_"] == "sde_sampler_amd.distr.nice.Nice" and theirs["_target_"].endswith(".Nice")
    assert {k: v for k, v in theirs.items() if k != "_target_"}.items() <= ours.items()  # + `checkpoint`: the reference's default file is not shipped

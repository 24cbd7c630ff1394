"""Bridge between the reference's Python plugin surface and libsdeh.so.

`TrajectoryEngine.simulate(...)` is what the three loss classes call instead of running the per-step Python
loop.  It introspects the collaborating objects exactly as SURVEY.md 8b lists them -- the `generative_ctrl`
module (`base_model`, `score_model`, clip values, `target_score`, `prior_score`, `sde`), the SDE object and the
distributions behind the log-density callables -- by *class name and attribute*, so it works with this package's
host classes and with the reference's own classes alike, fills an `SdehProblem` (include/sdeh.h) with raw device
pointers and launches the HIP kernels on torch's current stream.

There is no eager fallback: configurations the kernels do not cover raise `SdehUnsupported`; a missing library
raises `SdehLibraryError`.
"""
from __future__ import annotations

import ctypes as C
import math
import os
import weakref
from typing import Callable

import torch

from . import _lib as L
from ._lib import SdehUnsupported

_INF = float("inf")

_CTRL_KINDS = {
    "LerpTargetCtrl": L.CTRL_LERP_TARGET, "LerpPriorCtrl": L.CTRL_LERP_PRIOR, "LerpCtrl": L.CTRL_LERP,
    "ScoreCtrl": L.CTRL_SCORE, "ClippedCtrl": L.CTRL_CLIPPED,
}
_CTRL_UNSUPPORTED = {"CancelDriftCtrl", "PotentialCtrl"}
_GAUSS_NAMES = {"IsotropicGauss", "Gauss", "Delta"}


_MRO_CACHE: dict = {}


def _mro_names(obj) -> tuple:
    cls = type(obj)
    names = _MRO_CACHE.get(cls)
    if names is None:
        names = _MRO_CACHE[cls] = tuple(c.__name__ for c in cls.__mro__)
    return names


def _scalar(t) -> float:
    """float(t) for the 0-dim buffers of SDE / distribution objects without a device sync per call: the value is
    cached on the tensor object and refreshed when the tensor's version counter moves."""
    if not isinstance(t, torch.Tensor):
        return float(t)
    if not t.is_cuda:
        return float(t)
    cached = getattr(t, "_sdeh_scalar", None)
    if cached is not None and cached[0] == t._version:
        return cached[1]
    value = float(t)
    try:
        t._sdeh_scalar = (t._version, value)
    except AttributeError:
        pass
    return value


def _unsupported(msg: str):
    return SdehUnsupported(-2, msg)


class _Keep(list):
    """Holds tensors whose device pointers were handed to C for the duration of one call."""

    converted = False

    def ptr(self, t: torch.Tensor | None, device, what: str) -> int | None:
        if t is None:
            return None
        if not isinstance(t, torch.Tensor):
            t = torch.as_tensor(t)
        if t.device != device or t.dtype != torch.float32 or not t.is_contiguous():
            t = t.detach().to(device=device, dtype=torch.float32).contiguous()
            self.converted = True  # the pointer is that of a copy made for this call: such a description is not cached
        self.append(t)
        return t.data_ptr()


def _sub(mod, name: str):
    """`mod.<name>` for a submodule / parameter / buffer without nn.Module.__getattr__'s Python-level search (~1.2 us per
    access; build_problem makes ~60 of them per launch -- a third of the host time between two evaluations)."""
    d = getattr(mod, "__dict__", None) or {}
    for table in ("_modules", "_parameters", "_buffers"):
        v = d.get(table, {}).get(name)
        if v is not None:
            return v
    return getattr(mod, name)


def _activation_id(act) -> int:
    name = type(act).__name__
    if name == "GELU":
        if getattr(act, "approximate", "none") != "none":
            raise _unsupported("GELU(approximate='tanh') is not built in (exact erf GELU is)")
        return L.ACT_GELU_ERF
    if name == "SiLU":
        return L.ACT_SILU
    if name == "ReLU":
        return L.ACT_RELU
    raise _unsupported(f"activation {name} is not built into the trajectory kernel (GELU, SiLU, ReLU are)")


def _fill_time_embed(te, out: L.SdehTimeEmbed, keep: _Keep, device, what: str):
    if "TimeEmbed" not in _mro_names(te):
        raise _unsupported(f"{what}: expected a TimeEmbed module, got {type(te).__name__}")
    layers = list(_sub(te, "hidden_layer"))
    if not 1 <= len(layers) <= L.SDEH_MAX_HIDDEN:
        raise _unsupported(f"{what}: {len(layers)} hidden layers")
    out_layer = _sub(te, "out_layer")
    out.channels, out.n_hidden, out.dim_out = te.channels, len(layers), out_layer.out_features
    out.coeff = keep.ptr(_sub(te, "timestep_coeff").reshape(-1), device, what)
    out.phase = keep.ptr(_sub(te, "timestep_phase").reshape(-1), device, what)
    for i, lin in enumerate(layers):
        out.hidden_w[i] = keep.ptr(_sub(lin, "weight"), device, what)
        out.hidden_b[i] = keep.ptr(_sub(lin, "bias"), device, what)
    out.out_w = keep.ptr(_sub(out_layer, "weight"), device, what)
    out.out_b = keep.ptr(_sub(out_layer, "bias"), device, what)


def _fill_fourier_mlp(net, out: L.SdehFourierMLP, keep: _Keep, device):
    if "FourierMLP" not in _mro_names(net):
        raise _unsupported(f"base_model {type(net).__name__}: only FourierMLP is fused (SURVEY.md section 2)")
    layers = list(_sub(net, "hidden_layer"))
    if len(layers) > L.SDEH_MAX_HIDDEN:
        raise _unsupported(f"base_model: {len(layers)} hidden layers > {L.SDEH_MAX_HIDDEN}")
    input_embed, out_layer, act, te = _sub(net, "input_embed"), _sub(net, "out_layer"), _sub(net, "activation"), _sub(net, "timestep_embed")
    out.dim, out.channels, out.n_hidden = input_embed.in_features, net.channels, len(layers)
    out.activation = _activation_id(act)
    if out_layer.out_features != out.dim:
        raise _unsupported("base_model.dim_out != dim")
    out.input_w = keep.ptr(_sub(input_embed, "weight"), device, "input_embed")
    out.input_b = keep.ptr(_sub(input_embed, "bias"), device, "input_embed")
    for i, lin in enumerate(layers):
        out.hidden_w[i] = keep.ptr(_sub(lin, "weight"), device, "hidden_layer")
        out.hidden_b[i] = keep.ptr(_sub(lin, "bias"), device, "hidden_layer")
    out.out_w = keep.ptr(_sub(out_layer, "weight"), device, "out_layer")
    out.out_b = keep.ptr(_sub(out_layer, "bias"), device, "out_layer")
    _fill_time_embed(te, out.timestep_embed, keep, device, "base_model.timestep_embed")
    if type(_sub(te, "activation")) is not type(act):
        raise _unsupported("base_model and its timestep_embed use different activations")


def _fill_density(dist, out: L.SdehDensity, keep: _Keep, device, what: str):
    """Maps a Distribution object (reference or sde_sampler_amd.distr) onto an SdehDensity."""
    names = _mro_names(dist)
    lnc = getattr(dist, "log_norm_const", None)
    out.dim = dist.dim
    out.log_norm_const = 0.0 if lnc is None else float(lnc)
    if any(n in _GAUSS_NAMES for n in names) or ("GMM" in names and dist.mixture_weights is None):
        out.kind = L.DENS_DIAG_GAUSS
        out.loc = keep.ptr(dist.loc.reshape(-1), device, what)
        out.scale = keep.ptr(dist.scale.reshape(-1), device, what)
        if dist.loc.numel() != dist.dim:
            raise _unsupported(f"{what}: Gaussian with loc of shape {tuple(dist.loc.shape)}")
    elif "GMM" in names:
        out.kind = L.DENS_GMM
        out.n_components = dist.loc.shape[0]
        out.loc = keep.ptr(dist.loc, device, what)
        out.scale = keep.ptr(dist.scale, device, what)
        out.mixture_weights = keep.ptr(dist.mixture_weights, device, what)
        shared, n_vary = _mixture_structure(dist.loc, dist.scale)
        if shared:
            out.flags |= L.DENS_FLAG_SHARED_SCALE | (((n_vary + 1) & 0xFFFF) << 8)
        if n_vary > 8 and 21 <= dist.loc.shape[0] <= 40 and _mixture_mm_ok(dist.loc, dist.scale):
            out.flags |= L.DENS_FLAG_MM_OK
    elif "DoubleWell" in names:
        out.kind, out.n_components = L.DENS_MULTI_WELL, 1
        out.p0, out.p1 = _scalar(dist.separation), _scalar(dist.shift)
        out.log_norm_const = 0.0  # DoubleWell.unnorm_log_prob carries no constant
    elif "MultiWell" in names:
        out.kind, out.n_components = L.DENS_MULTI_WELL, dist.n_double_wells
        out.p0, out.p1 = _scalar(dist.separation), _scalar(dist.double_well.shift)
        out.log_norm_const = 0.0
    elif "Funnel" in names:
        out.kind = L.DENS_FUNNEL
        out.p0 = float(dist.variance)
    elif _external_target(dist):
        # no closed form inside the trajectory kernels: the engine evaluates the score between step segments (TrajectoryEngine.run)
        out.kind = L.DENS_EXTERNAL
        keep.append(_ExternalTarget(dist))
    else:
        raise _unsupported(f"{what}: distribution {type(dist).__name__} has no fused log-density/score "
                           "(GMM, Gauss, IsotropicGauss, Delta, DoubleWell, MultiWell, Funnel are built in)")


def _mixture_structure(loc: torch.Tensor, scale: torch.Tensor) -> tuple[bool, int]:
    """(shared_scale, n_varying): whether every component has the same per-coordinate scale, and -- if so -- the
    smallest n such that coordinates >= n have the same mean in every component (the reference's high-dimensional
    mixtures pad a 2-d mixture with zero means, distr/gauss.py:59-60).  These are the promises behind
    SDEH_DENS_FLAG_SHARED_SCALE / SDEH_DENS_FLAG_NVARY.  Costs one device sync per (tensor object, version): targets
    are fixed buffers, so this happens once.  The result is cached ON the tensor object (never by address: a freed
    tensor's address is reused)."""
    stamp = (loc._version, id(scale), scale._version, str(loc.device))
    cached = getattr(loc, "_sdeh_structure", None)
    if cached is not None and cached[0] == stamp:
        return cached[1]
    shared = bool((scale == scale[:1]).all().item())
    n_vary = loc.shape[1]
    if shared:
        varying = (loc != loc[:1]).any(dim=0).nonzero()
        n_vary = int(varying.max().item()) + 1 if varying.numel() else 0
    try:
        loc._sdeh_structure = (stamp, (shared, n_vary))
    except AttributeError:  # exotic tensor subclasses without a __dict__: just recompute next time
        pass
    return shared, n_vary


# Product-form logits (SDEH_DENS_FLAG_MM_OK): l_k = c_k - |m_k|^2 + 2 y . m_k with y = x / (sqrt2 sigma), m_k = mu_k / (sqrt2 sigma)
# carries the fp32 rounding of its LARGE terms, eps ~ 6e-8 x 2 |y| |m_k| (tests/perf/mixture_mfma_numerics.py; per-component scales:
# + 6e-8 |y|^2), where the squared-distance form -|y - m_k|^2 carries 6e-8 x its own value.  The two only differ for a component the
# trajectory is CLOSE to (small squared distance, large products) -- and an error in one component's logit only matters where another
# component has comparable weight.
#   (a) eps <= MM_ABS_TOL everywhere the sampler can be (|y| <= max |m| + 6): any mixture near the origin qualifies;
#   (b) or no two components are within MM_MIN_SEPARATION of each other (in the wider one's units): two components share a
#       trajectory's weight (|l_j - l_k| < 20) only in the slab where BOTH squared distances exceed (sep / 2)^2 - 10 >= 26 -- there the
#       squared-distance form's own logits are large and rounded as well -- and eps <= MM_ABS_CAP bounds what the product form can
#       do inside that slab: a responsibility moves by at most 0.5 %.
MM_ABS_TOL = 1.0e-5
MM_MIN_SEPARATION = 12.0
MM_ABS_CAP = 5.0e-3


def _mixture_mm_ok(loc: torch.Tensor, scale: torch.Tensor) -> bool:
    """Whether a shared-scale mixture's logits may be evaluated in product form on the matrix pipe (rule above).  One device sync
    per (tensor object, version), cached on the tensor like `_mixture_structure`."""
    stamp = (loc._version, id(scale), scale._version, str(loc.device), "mm")
    cached = getattr(loc, "_sdeh_mm_ok", None)
    if cached is not None and cached[0] == stamp:
        return cached[1]
    with torch.no_grad():
        sd = scale.double().expand_as(loc) * math.sqrt(2.0)
        m = loc.double() / sd  # every component in its own units
        norm = m.norm(dim=1)
        reach = float(norm.max().item()) + 6.0
        shared = bool((scale == scale[:1]).all().item())
        # shared scale: the x^2 term is common to all components and never formed; per-component scales: it is part of every logit
        eps = 6.0e-8 * (2.0 * reach * float(norm.max().item()) + (0.0 if shared else reach * reach))
        ok = eps <= MM_ABS_TOL
        if not ok and m.shape[0] > 1:
            # pairwise separation in the WIDER of the two components' units, coordinate by coordinate
            wide = torch.maximum(sd[:, None, :], sd[None, :, :])
            dist = ((loc.double()[:, None, :] - loc.double()[None, :, :]) / wide).norm(dim=2)
            dist.fill_diagonal_(float("inf"))
            sep = float(dist.min().item())
            ok = sep >= MM_MIN_SEPARATION and eps <= MM_ABS_CAP
    try:
        loc._sdeh_mm_ok = (stamp, ok)
    except AttributeError:
        pass
    return ok


def _has_analytic_score(dist) -> bool:
    """False when `dist.score` is Distribution.score itself (autograd of unnorm_log_prob, distr/base.py:130-137), i.e. a plain
    GMM; Gauss / IsotropicGauss / Delta / the wells / Funnel override it with a closed form."""
    for cls in type(dist).__mro__:
        if "score" in vars(cls):
            return cls.__name__ != "Distribution"
    return True


def _known_distribution(obj) -> bool:
    names = set(_mro_names(obj))
    return bool(names & (_GAUSS_NAMES | {"GMM", "DoubleWell", "MultiWell", "Funnel"}))


class _PriorRef:
    """Rides in a problem's `_Keep` next to an `_ExternalTarget`: the Gaussian prior whose score a LerpCtrl interpolates with the supplied
    target score (the backward's `sc_in` plane is that interpolation, formed on the stored trajectory)."""

    def __init__(self, obj):
        self.obj = obj


class _ExternalTarget:
    """Rides in a problem's `_Keep`: the target object whose score the engine evaluates BETWEEN the step segments of the wide kernels
    (SDEH_DENS_EXTERNAL, sdeh_simulate_fwd_steps) -- the NICE flow of BASELINE configs[4] (csrc/sdeh_nice.hip)."""

    def __init__(self, obj):
        self.obj = obj


def _external_target(obj) -> bool:
    """A `Nice` distribution (this package's or the reference's distr/nice.py:233-298) around a NiceModel."""
    if "Nice" not in _mro_names(obj):
        return False
    model = getattr(obj, "model", None)
    return model is not None and hasattr(model, "coupling") and hasattr(model, "scaling")


def describe_nice(nice, device, keep: _Keep) -> L.SdehNice:
    """SdehNice (include/sdeh.h) of a `Nice` distribution: raw device pointers of the couplings' nn.Linear parameters (reference
    distr/nice.py:43-62, 100-110, 123-153), read by attribute -- works on the reference's own NiceModel as well."""
    model = nice.model
    prior = type(getattr(model, "prior", None)).__name__
    if prior != "StandardLogistic":
        raise _unsupported(f"NICE prior {prior}: the logistic prior of distr/nice.py is built in")
    out = L.SdehNice()
    couplings = list(model.coupling)
    if not 1 <= len(couplings) <= L.SDEH_NICE_MAX_COUPLING:
        raise _unsupported(f"NICE with {len(couplings)} coupling layers (1 .. {L.SDEH_NICE_MAX_COUPLING})")
    out.dim, out.n_coupling = int(model.in_out_dim), len(couplings)
    out.mid_dim = int(couplings[0].in_block[0].weight.shape[0])
    out.n_mid = len(couplings[0].mid_block)
    if out.n_mid > L.SDEH_MAX_HIDDEN or out.mid_dim % 4 or out.dim % 2 or out.dim > 256:
        raise _unsupported(f"NICE geometry dim={out.dim} mid_dim={out.mid_dim} hidden={out.n_mid + 1}: even dim <= 256, mid_dim a multiple "
                           f"of 4, at most {L.SDEH_MAX_HIDDEN + 1} hidden layers")
    for c, layer in enumerate(couplings):
        lin_in, lin_out = layer.in_block[0], layer.out_block
        if (tuple(lin_in.weight.shape) != (out.mid_dim, out.dim // 2) or tuple(lin_out.weight.shape) != (out.dim // 2, out.mid_dim)
                or len(layer.mid_block) != out.n_mid or type(layer.in_block[1]).__name__ != "ReLU"):
            raise _unsupported(f"NICE coupling {c}: not the Linear + ReLU blocks of distr/nice.py:56-62")
        out.mask_config[c] = 1 if layer.mask_config else 0
        out.in_w[c], out.in_b[c] = keep.ptr(lin_in.weight, device, "nice in_block"), keep.ptr(lin_in.bias, device, "nice in_block")
        for l, block in enumerate(layer.mid_block):
            out.mid_w[c][l] = keep.ptr(block[0].weight, device, "nice mid_block")
            out.mid_b[c][l] = keep.ptr(block[0].bias, device, "nice mid_block")
        out.out_w[c], out.out_b[c] = keep.ptr(lin_out.weight, device, "nice out_block"), keep.ptr(lin_out.bias, device, "nice out_block")
    out.scale = keep.ptr(model.scaling.scale.reshape(-1), device, "nice scaling")
    lnc = getattr(nice, "log_norm_const", None)
    out.log_norm_const = 0.0 if lnc is None else float(lnc)
    return out


def nice_eval(nice, x: torch.Tensor, *, want_score: bool, want_logp: bool, cache: dict | None = None, score_out: torch.Tensor | None = None,
              desc=None):
    """(score [B, d] | None, unnorm_log_prob [B] | None) of a `Nice` target at x on the HIP kernels (sdeh_nice_eval), on torch's
    current stream.  `cache`: work memory per batch size; `desc`: an (SdehNice, keep) pair described once for a whole trajectory."""
    if not x.is_cuda:
        raise RuntimeError("nice_eval needs GPU tensors (the CPU form of the target is Nice.model.log_prob)")
    lib = L.load()
    device = x.device
    xc = x.detach()
    if xc.dtype != torch.float32 or not xc.is_contiguous():
        xc = xc.float().contiguous()
    batch = xc.shape[0]
    keep = _Keep()
    nd = describe_nice(nice, device, keep) if desc is None else desc[0]
    n_work = lib.sdeh_nice_work_floats(C.byref(nd), batch, 1 if want_score else 0)
    if n_work < 0:
        raise L.SdehError(-1, L.load().sdeh_last_error().decode())
    key = (device.index, batch, bool(want_score))
    work = None if cache is None else cache.get(key)
    if work is None or work.numel() < n_work:
        work = torch.empty(n_work, device=device, dtype=torch.float32)
        if cache is not None:
            if len(cache) > 8:
                cache.clear()
            cache[key] = work
    score = None
    if want_score:
        score = score_out if score_out is not None else torch.empty_like(xc)
    logp = torch.empty(batch, device=device, dtype=torch.float32) if want_logp else None
    with torch.cuda.device(device):
        L.check(lib.sdeh_nice_eval(C.byref(nd), xc.data_ptr(), batch, None if score is None else score.data_ptr(),
                                   None if logp is None else logp.data_ptr(), work.data_ptr(), work.numel(),
                                   torch.cuda.current_stream(device).cuda_stream))
    return score, logp


def _bound_owner(fn, method: str):
    """`obj` if `fn` is the bound method `obj.<method>`, else None."""
    owner = getattr(fn, "__self__", None)
    return owner if owner is not None and getattr(fn, "__name__", None) == method else None


def _same_coefficients(a, b) -> bool:
    """True when two OU objects describe the same process up to the `generative` flag (the solver builds the
    inference SDE as `instantiate(cfg.sde, generative=False)`, solver/oc.py:130-133)."""
    if type(a) is not type(b):
        return False
    names = ("terminal_t", "diff_coeff_sq_min", "diff_coeff_sq_max", "scale_diff_coeff", "drift_coeff", "diff_coeff")
    return all(_scalar(getattr(a, n)) == _scalar(getattr(b, n)) for n in names if hasattr(a, n))


_SCALAR_ATTRS = ("clip_model", "clip_score", "scale_score", "detach_score", "hard_constrain", "terminal_t", "diff_coeff_sq_min",
                 "diff_coeff_sq_max", "scale_diff_coeff", "drift_coeff", "diff_coeff", "generative", "loc", "scale", "mixture_weights",
                 "log_norm_const", "separation", "shift", "variance", "n_double_wells", "dim", "double_well", "target_score", "prior_score",
                 "sde", "base_model", "score_model", "activation")


def _fp_plan(o) -> list:
    """[(dict, key, read_on_host)]: where the things `_describe` reads from object `o` live -- every parameter / buffer of a module
    tree (their data pointers go into the SdehProblem) and the scalar / tensor attributes whose VALUES are read on the host.  Built
    once per object (the module TREE is taken as fixed; its tensors and attribute values may change)."""
    d = o.__dict__
    plan = d.get("_sdeh_fp")
    if plan is None:
        plan = []
        if isinstance(o, torch.nn.Module):
            for m in o.modules():
                plan += [(m._parameters, n, False) for n in m._parameters] + [(m._buffers, n, False) for n in m._buffers]
        for name in _SCALAR_ATTRS:
            for table in (d, d.get("_buffers"), d.get("_parameters"), d.get("_modules")):
                if table is not None and name in table:
                    plan.append((table, name, True))
                    break
        try:
            d["_sdeh_fp"] = plan
        except TypeError:
            pass
    return plan


def _fingerprint(objs) -> tuple | None:
    """What `_describe` read from the objects besides their identity: see TrajectoryEngine.build_problem."""
    out = []
    Tensor = torch.Tensor
    for o in objs:
        if o is None:
            continue
        if isinstance(o, Tensor):
            out.append(o.data_ptr())
            continue
        if getattr(o, "__dict__", None) is None:
            return None
        for table, key, host in _fp_plan(o):
            v = table.get(key)
            if isinstance(v, Tensor):
                out.append(v.data_ptr())
                if host:
                    out.append(v._version)
            elif v is None or isinstance(v, (int, float, bool, str)):
                out.append(v)
            else:  # a sub-object or a bound method (target_score = target.score): which one
                out.append(id(getattr(v, "__self__", v)))
    return tuple(out)


_OPTION_KEYS = tuple((name, name.encode()) for name in L.PLAN_OPTIONS)


class _Plan:
    def __init__(self, lib, desc: L.SdehPlanDesc):
        self.lib, self.handle = lib, C.c_void_p()
        L.check(lib.sdeh_plan_create(C.byref(desc), C.byref(self.handle)))
        self.reserved = int(desc.max_batch)
        self.options = {name: os.environ.get(name) for name in L.PLAN_OPTIONS}  # what sdeh_plan_create read

    def sync_options(self, overrides: dict | None = None) -> None:
        """Kernel-mode options: `overrides` (engine.options) over the environment (the tests' switch), pushed to the plan when they
        differ from what it holds.  The library reads no environment variable on the launch path."""
        env = getattr(os.environ, "_data", None)  # the raw dict of the process environment (bytes keys): 16 lookups per launch
        for name, raw in _OPTION_KEYS:
            if overrides and name in overrides:
                want = overrides[name]
            elif env is not None:
                want = env.get(raw)
                want = None if want is None else want.decode()
            else:
                want = os.environ.get(name)
            if want != self.options[name]:
                L.check(self.lib.sdeh_plan_set_option(self.handle, name.encode(), None if want is None else str(want).encode()))
                self.options[name] = want

    def reserve(self, batch: int) -> None:
        """Batch-dependent plan scratch (wide Bridge): grown HERE, outside the stream-ordered calls (it synchronises the device)."""
        if batch > self.reserved:
            if torch.cuda.is_current_stream_capturing():
                raise L.SdehError(-4, f"the plan's scratch covers {self.reserved} trajectories and cannot grow to {batch} while the "
                                      "stream is capturing: launch once at this batch first")
            L.check(self.lib.sdeh_plan_reserve(self.handle, int(batch)))
            self.reserved = int(batch)

    def __del__(self):
        try:
            if self.handle:
                self.lib.sdeh_plan_destroy(self.handle)
        except Exception:  # interpreter shutdown
            pass


class TrajectoryEngine:
    """Owns the libsdeh plans of one loss object and turns `simulate(...)` calls into kernel launches."""

    def __init__(self):
        self._plans: dict[tuple, _Plan] = {}
        self.calls = 0  # advances the Philox stream: one `offset` per simulate() call
        #: distinguishes the Philox streams of several engines in one process (bits 40.. of the offset): two objects with the
        #: same `stream_id`, seed, call count and row offsets draw IDENTICAL noise.  Loss objects default to 0, the
        #: EulerIntegrator to 1 (a loss plus the inference-process integrator is the combination the reference's solvers build);
        #: give further loss objects of one process their own id.
        self.stream_id = 0
        self.timing = False  # record HIP events around the trajectory kernel (bench.py)
        #: kernel-mode options of this engine's plans ({name in _lib.PLAN_OPTIONS: value or None}); take precedence over the
        #: environment variables of the same names
        self.options: dict = {}
        self._problems: dict = {}  # build_problem's cache
        self._nice_work: dict = {}  # work memory of sdeh_nice_eval per batch size (NICE targets)
        self._last_plan = None

    # ------------------------------------------------------------------------------------------------------
    def _plan(self, device, dim, channels, n_hidden, n_steps, k) -> _Plan:
        key = (device.index, dim, channels)
        plan = self._plans.get(key)
        cap = getattr(plan, "cap", None)
        if plan is None or n_hidden > cap[0] or n_steps > cap[1] or k > cap[2]:
            cap = (max(n_hidden, cap[0] if cap else 0), max(n_steps, 2 * cap[1] if cap else 512),
                   max(k, cap[2] if cap else 0))
            desc = L.SdehPlanDesc(dim=dim, channels=channels, max_hidden=cap[0], max_steps=cap[1],
                                  max_components=cap[2], device=device.index)
            plan = _Plan(L.load(), desc)
            plan.cap = cap
            plan.timing = False
            self._plans[key] = plan
        if plan.timing != self.timing:
            L.check(plan.lib.sdeh_plan_set_timing(plan.handle, 1 if self.timing else 0))
            plan.timing = self.timing
        plan.sync_options(self.options)
        self._last_plan = plan
        return plan

    def offset(self) -> int:
        """The Philox offset of the NEXT launch: call count in the low 40 bits, `stream_id` above."""
        return (int(self.stream_id) << 40) + int(self.calls)

    def last_kernel_ms(self) -> float:
        """Duration of the last trajectory-kernel launch (HIP events on its stream); needs `timing = True`."""
        ms = C.c_float()
        L.check(self._last_plan.lib.sdeh_plan_last_kernel_ms(self._last_plan.handle, C.byref(ms)))
        return ms.value

    def kernel_ms_history(self, n: int = 128) -> list[tuple[str, float]]:
        """(kernel name, ms) of up to `n` of the last timed launches of the plan that served the last call, newest first
        (sdeh_plan_timing_entry: a ring of event pairs -- read it AFTER a run instead of synchronising after every launch)."""
        out = []
        if self._last_plan is None:
            return out
        ms, name = C.c_float(), C.create_string_buffer(96)
        for back in range(n):
            rc = self._last_plan.lib.sdeh_plan_timing_entry(self._last_plan.handle, back, C.byref(ms), name, 96)
            if rc != 0:
                if rc < 0:
                    L.check(rc)
                break
            out.append((name.value.decode(), ms.value))
        return out

    def last_kernel_name(self) -> str:
        """Name of the compiled kernel variant that served the last launch (sdeh_plan_last_kernel_name)."""
        if self._last_plan is None:
            return ""
        return self._last_plan.lib.sdeh_plan_last_kernel_name(self._last_plan.handle).decode()

    def invalidate(self) -> None:
        """Forget every cached problem description (`build_problem`).  The fingerprint that revalidates a cached description covers
        the TOP-LEVEL collaborating objects (control, SDE, target, prior / reference, inference control): the data pointers of all
        their parameters / buffers (whole module trees), the versions of tensors whose values were read on the host, their scalar
        attributes.  Sub-objects reached through an attribute or a bound method (`ctrl.target_score.__self__`, `dist.double_well`, an
        activation's `approximate`) enter by identity only, and an object's attribute SET is taken as fixed after its first use:
        after mutating a scalar of such a sub-object in place, or adding attributes / parameters to an object that was already
        used, call this (replacing the sub-object, or any top-level mutation, is seen without it)."""
        self._problems.clear()

    # ------------------------------------------------------------------------------------------------------
    def build_problem(self, *, device, keep: _Keep, **kw) -> L.SdehProblem:
        """The SdehProblem of the collaborating objects (signature: `_describe`).  Introspecting ~60 module attributes costs ~45 us
        per call -- a third of the host time between two evaluations at B = 1024 -- and the answer only changes when a parameter
        tensor is REPLACED (EMA swap), a clip value is mutated or a coefficient tensor is written: the filled struct is cached per
        (objects, flags, scalars) and revalidated with a fingerprint of exactly those things (`_fingerprint`: data pointers of every
        parameter / buffer, versions of the tensors whose VALUES were read on the host, the scalar attributes).  In-place parameter
        updates (optimizer steps) keep pointers and need nothing: the kernels read the current values.  A hit returns a copy."""
        objs = tuple(kw.get(n) for n in ("generative_ctrl", "sde", "terminal_target", "second", "reference_prior", "inference_ctrl",
                                         "rng_counter"))
        key = (kw.get("loss_kind"), kw.get("flags"), kw.get("clip_target"), kw.get("alpha", 0.0), kw.get("sigma", 0.0),
               kw.get("allow_inference_sde", False), kw.get("dim"), device.type, device.index, tuple(id(o) for o in objs))
        fp = _fingerprint(objs)
        ent = self._problems.get(key)
        if ent is not None and fp is not None and ent[0] == fp and all(r() is o for r, o in zip(ent[3], objs) if r is not None):
            keep.extend(ent[2])
            return L.SdehProblem.from_buffer_copy(ent[1])
        own = _Keep()
        # (no graph: a converted copy of a parameter made with grad enabled would carry a grad_fn, keep that parameter's AccumulateGrad node
        # -- bound to the stream of THIS call -- alive in the cache, and break a later hipGraph capture of the backward)
        with torch.no_grad():
            pr = self._describe(device=device, keep=own, **kw)
        keep.extend(own)
        if fp is not None and not own.converted:
            try:
                refs = tuple(None if o is None else weakref.ref(o) for o in objs)
            except TypeError:
                return pr
            if len(self._problems) > 64:
                self._problems.clear()
            self._problems[key] = (fp, L.SdehProblem.from_buffer_copy(pr), list(own), refs)
        return pr

    def _describe(self, *, loss_kind: int, generative_ctrl, sde, flags: int, device, keep: _Keep,
                  terminal_target=None, clip_target=None, second=None, reference_prior=None,
                  alpha: float = 0.0, sigma: float = 0.0, allow_inference_sde: bool = False,
                  dim: int | None = None, inference_ctrl=None, rng_counter: torch.Tensor | None = None) -> L.SdehProblem:
        pr = L.SdehProblem()
        pr.loss_kind, pr.flags = loss_kind, flags
        if rng_counter is not None:  # device-resident Philox offset (hipGraph replays, utils/graphs.py)
            if not (rng_counter.is_cuda and rng_counter.dtype == torch.int64 and rng_counter.numel() == 1):
                raise ValueError("rng_counter must be a one-element int64 tensor on the GPU")
            keep.append(rng_counter)
            pr.rng_offset_dev = rng_counter.data_ptr()
        if generative_ctrl is None:  # sdeh_integrate only: LangevinSDE / bare OU / ControlledSDE(ctrl=None)
            if not allow_inference_sde:
                raise ValueError("generative_ctrl is None")
            pr.ctrl_kind = L.CTRL_NONE
            pr.clip_model = pr.clip_score = pr.clip_target = _INF
            pr.scale_score = 1.0
            pr.base_model.dim, pr.base_model.channels = int(dim), 64
            if terminal_target is not None:
                _fill_density(terminal_target, pr.target, keep, device, "target")
                if pr.target.dim != dim:
                    raise ValueError(f"target dim {pr.target.dim} != state dim {dim}")
            self._fill_sde(pr, sde, allow_inference_sde)
            return pr
        # ---- control --------------------------------------------------------------------------------------
        names = _mro_names(generative_ctrl)
        bad = [n for n in names if n in _CTRL_UNSUPPORTED]
        if bad:
            raise _unsupported(f"generative_ctrl {bad[0]} is not fused by the HIP engine")
        kind = next((_CTRL_KINDS[n] for n in names if n in _CTRL_KINDS), None)
        if kind is None:
            raise _unsupported(f"generative_ctrl of type {type(generative_ctrl).__name__} is not a known control module")
        if getattr(generative_ctrl, "hard_constrain", False):
            raise _unsupported("hard_constrain=True (dead code in the reference) is not supported")
        pr.ctrl_kind = kind
        clip_model = getattr(generative_ctrl, "clip_model", None)
        clip_score = getattr(generative_ctrl, "clip_score", None)
        pr.clip_model = _INF if clip_model is None else float(clip_model)
        pr.clip_score = _INF if clip_score is None else float(clip_score)
        pr.scale_score = float(getattr(generative_ctrl, "scale_score", 1.0))
        pr.clip_target = _INF if clip_target is None else float(clip_target)
        _fill_fourier_mlp(generative_ctrl.base_model, pr.base_model, keep, device)
        dim = pr.base_model.dim
        target_obj = terminal_target
        if kind != L.CTRL_CLIPPED:
            score_model = getattr(generative_ctrl, "score_model", None)
            if score_model is not None:
                _fill_time_embed(score_model, pr.score_model, keep, device, "score_model")
                if type(score_model.activation) is not type(generative_ctrl.base_model.activation):
                    raise _unsupported("score_model and base_model use different activations")
            if kind != L.CTRL_LERP_PRIOR:
                owner = _bound_owner(generative_ctrl.target_score, "score")
                if owner is None or not (_known_distribution(owner) or _external_target(owner)):
                    raise _unsupported("generative_ctrl.target_score must be the `.score` of a built-in distribution")
                if target_obj is not None and owner is not target_obj:
                    raise _unsupported("target_score and terminal_unnorm_log_prob belong to different distributions")
                target_obj = owner
            if kind in (L.CTRL_LERP, L.CTRL_LERP_PRIOR):
                owner = _bound_owner(generative_ctrl.prior_score, "score")
                if owner is None or not _known_distribution(owner):
                    raise _unsupported("generative_ctrl.prior_score must be the `.score` of a Gaussian prior")
                if reference_prior is not None and owner is not reference_prior:
                    raise _unsupported("prior_score and reference_ctrl use different priors")
                reference_prior = owner
            ctrl_sde = getattr(generative_ctrl, "sde", None)
            if kind >= L.CTRL_LERP and ctrl_sde is not None and ctrl_sde is not sde and not _same_coefficients(ctrl_sde, sde):
                raise _unsupported("generative_ctrl.sde differs from the loss's sde")
        if target_obj is not None:
            _fill_density(target_obj, pr.target, keep, device, "target")
            if pr.target.dim != dim:
                raise ValueError(f"target dim {pr.target.dim} != model dim {dim}")
        if kind != L.CTRL_CLIPPED:
            # which score terms are constants of the autograd graph (only the back-propagation-through-time kernel reads this):
            # detach_score=True detaches x in front of every score (reparam.py:58,134,169,188); otherwise a score obtained through
            # Distribution.score's autograd call (distr/base.py:130-137, create_graph=False) is a constant as well -- mixtures are
            # handled as such inside the kernel, a ONE-component GMM (evaluated here as a Gaussian) needs the flag
            if bool(getattr(generative_ctrl, "detach_score", False)):
                pr.flags |= L.FLAG_DETACH_SCORE
            elif target_obj is not None and pr.target.kind == L.DENS_DIAG_GAUSS and not _has_analytic_score(target_obj):
                pr.flags |= L.FLAG_TARGET_SCORE_CONST
        if reference_prior is not None:
            _fill_density(reference_prior, pr.prior, keep, device, "prior")
            if pr.prior.kind != L.DENS_DIAG_GAUSS:
                raise _unsupported("only Gaussian priors have a fused score")
            if pr.target.kind == L.DENS_EXTERNAL:
                keep.append(_PriorRef(reference_prior))
        if second is not None:
            _fill_density(second, pr.second, keep, device, "initial/reference density")
            if pr.second.kind != L.DENS_DIAG_GAUSS:
                raise _unsupported("only Gaussian initial/reference densities are fused")
        if inference_ctrl is not None:  # Bridge: TimeReversalLoss.inference_ctrl (losses/oc.py:189-202)
            inames = _mro_names(inference_ctrl)
            ikind = next((_CTRL_KINDS[n] for n in inames if n in _CTRL_KINDS), None)
            if ikind not in (L.CTRL_CLIPPED, L.CTRL_LERP_PRIOR):
                raise _unsupported(f"inference_ctrl {type(inference_ctrl).__name__}: the divergence is built in for ClippedCtrl "
                                   "and LerpPriorCtrl (the reference's Bridge configurations)")
            if getattr(inference_ctrl, "hard_constrain", False):
                raise _unsupported("hard_constrain=True (dead code in the reference) is not supported")
            if getattr(inference_ctrl, "detach_score", False) and ikind == L.CTRL_LERP_PRIOR:
                raise _unsupported("inference_ctrl.detach_score=True removes the score term from the divergence; "
                                   "the built-in divergence assumes detach_score=False (conf/solver/bridge.yaml)")
            inf = pr.inference
            inf.ctrl_kind = ikind
            icm, ics = getattr(inference_ctrl, "clip_model", None), getattr(inference_ctrl, "clip_score", None)
            inf.clip_model = _INF if icm is None else float(icm)
            inf.clip_score = _INF if ics is None else float(ics)
            inf.scale_score = float(getattr(inference_ctrl, "scale_score", 1.0))
            _fill_fourier_mlp(inference_ctrl.base_model, inf.base_model, keep, device)
            if inf.base_model.dim != dim:
                raise ValueError(f"inference_ctrl dim {inf.base_model.dim} != generative_ctrl dim {dim}")
            if ikind == L.CTRL_LERP_PRIOR:
                ism = getattr(inference_ctrl, "score_model", None)
                if ism is not None:
                    _fill_time_embed(ism, inf.score_model, keep, device, "inference score_model")
                owner = _bound_owner(inference_ctrl.prior_score, "score")
                if owner is None or not _known_distribution(owner):
                    raise _unsupported("inference_ctrl.prior_score must be the `.score` of a Gaussian prior")
                if pr.prior.kind != L.DENS_NONE and owner is not reference_prior:
                    raise _unsupported("inference_ctrl.prior_score and the generative control use different priors")
                _fill_density(owner, pr.prior, keep, device, "prior")
                if pr.prior.kind != L.DENS_DIAG_GAUSS:
                    raise _unsupported("only Gaussian priors have a fused score")
                isde = getattr(inference_ctrl, "sde", None)
                if isde is not None and isde is not sde and not _same_coefficients(isde, sde):
                    raise _unsupported("inference_ctrl.sde differs from the loss's sde")
            pr.flags |= L.FLAG_INFERENCE_CTRL
        self._fill_sde(pr, sde, allow_inference_sde)
        pr.exp_alpha, pr.exp_sigma = float(alpha), float(sigma)
        return pr

    @staticmethod
    def _fill_sde(pr: L.SdehProblem, sde, allow_inference_sde: bool = False):
        if sde is None:
            pr.sde_kind = L.SDE_NONE
            return
        snames = _mro_names(sde)
        if not getattr(sde, "generative", True):
            if not allow_inference_sde:
                raise _unsupported("the engine integrates the generative SDE (generative=True)")
            pr.flags |= L.FLAG_INFERENCE_SDE
        pr.terminal_t = _scalar(sde.terminal_t)
        if "VP" in snames:
            pr.sde_kind = L.SDE_VP
            pr.vp_beta_min, pr.vp_beta_max = _scalar(sde.diff_coeff_sq_min), _scalar(sde.diff_coeff_sq_max)
            pr.vp_scale = _scalar(sde.scale_diff_coeff)
        elif "ConstOU" in snames:
            pr.sde_kind = L.SDE_CONST_OU
            pr.ou_drift, pr.ou_diff = _scalar(sde.drift_coeff), _scalar(sde.diff_coeff)
        elif "LangevinSDE" in snames:  # constant diffusion, the drift is the (clipped) target score
            pr.sde_kind = L.SDE_CONST_OU
            pr.ou_drift, pr.ou_diff = 0.0, _scalar(sde.diff_coeff)
        else:
            raise _unsupported(f"sde {type(sde).__name__}: VP, ConstOU and ScaledBM are built in")

    # ------------------------------------------------------------------------------------------------------
    def integrate(self, pr: L.SdehProblem, kind: int, timesteps: torch.Tensor, ts: torch.Tensor, x: torch.Tensor, *,
                  noise: torch.Tensor | None, eps: float, keep: _Keep, row_offset: int = 0,
                  seed: int | None = None) -> torch.Tensor:
        """Launches prep + the Euler integrator kernel (sdeh_integrate).  Returns xs [len(ts), B, d]."""
        if not x.is_cuda:
            raise RuntimeError("the HIP trajectory engine needs CUDA/HIP tensors (got a CPU tensor); "
                               "there is no CPU path in this package")
        device = x.device
        lib = L.load()
        if x.dim() != 2:
            raise ValueError(f"x_init must be [batch, dim], got {tuple(x.shape)}")
        batch, dim = x.shape
        if dim != pr.base_model.dim:
            raise ValueError(f"x_init has dim {dim}, the problem expects {pr.base_model.dim}")
        n_steps, n_out = timesteps.numel() - 1, ts.numel()
        if n_steps < 1 or n_out < 1:
            raise ValueError("need at least two integration time points and one output time")
        steps_p = keep.ptr(timesteps.reshape(-1), device, "timesteps")
        ts_p = keep.ptr(ts.reshape(-1), device, "ts")
        x_p = keep.ptr(x, device, "x_init")
        noise_p = None
        if noise is not None:
            if tuple(noise.shape) != (n_steps, batch, dim):
                raise ValueError(f"noise must be [{n_steps}, {batch}, {dim}], got {tuple(noise.shape)}")
            noise_p = keep.ptr(noise, device, "noise")
        xs = torch.empty((n_out, batch, dim), device=device, dtype=torch.float32)
        k = pr.target.n_components if pr.target.kind == L.DENS_GMM else 0
        plan = self._plan(device, dim, 64, pr.base_model.n_hidden, max(n_steps, n_out), k)
        if seed is None:
            seed = torch.initial_seed()
        offset = self.offset()
        self.calls += 1
        stream = torch.cuda.current_stream(device).cuda_stream
        with torch.cuda.device(device):
            L.check(lib.sdeh_integrate(plan.handle, C.byref(pr), kind, steps_p, n_steps, ts_p, n_out, float(eps), x_p,
                                       batch, noise_p, seed & 0xFFFFFFFFFFFFFFFF, offset, row_offset, xs.data_ptr(),
                                       stream))
        return xs

    # ------------------------------------------------------------------------------------------------------
    def run(self, pr: L.SdehProblem, ts: torch.Tensor, x: torch.Tensor, *, noise: torch.Tensor | None,
            return_traj: bool, keep: _Keep, row_offset: int = 0, seed: int | None = None, want_gp: bool = False,
            div_noise: torch.Tensor | None = None, want_planes: bool = False, want_u: bool = False):
        """Launches prep + trajectory kernels.  Returns (x_T [B,d], rnd [B,1], xs [T+1,B,d] | None); with `want_gp`
        (Bridge training) additionally the plane u + v [T,B,d] as a fourth element and, fifth, (sc [T,B,d] | None, tscore [B,d] | None)
        -- the score planes a wide Bridge on a mixture target keeps for its generative network's backward; with `want_planes` (training without an
        inference control) a fourth element: ("fused", sc [T,d,B] | None, tscore [d,B] | None, u [T,d,B] | None, zrec | None) when the fused backward takes the
        problem (xs is then the coordinate-major [T+1,d,B] plane), else (zt [(Lh+1),C,T*B], nn [T,B,d]); None when the launch kept nothing."""
        if not x.is_cuda:
            raise RuntimeError("the HIP trajectory engine needs CUDA/HIP tensors (got a CPU tensor); "
                               "there is no CPU path in this package")
        device = x.device
        lib = L.load()
        if x.dim() != 2:
            raise ValueError(f"x must be [batch, dim], got {tuple(x.shape)}")
        batch, dim = x.shape
        if dim != pr.base_model.dim:
            raise ValueError(f"x has dim {dim}, the model expects {pr.base_model.dim}")
        n_steps = ts.numel() - 1
        if n_steps < 1:
            raise ValueError("need at least two time points")
        ts_p = keep.ptr(ts.reshape(-1), device, "ts")
        x_p = keep.ptr(x, device, "x")
        noise_p = None
        if noise is not None:
            if tuple(noise.shape) != (n_steps, batch, dim):
                raise ValueError(f"noise must be [{n_steps}, {batch}, {dim}], got {tuple(noise.shape)}")
            noise_p = keep.ptr(noise, device, "noise")
        x_T = torch.empty((batch, dim), device=device, dtype=torch.float32)
        rnd = torch.empty((batch, 1), device=device, dtype=torch.float32)
        xs = None  # row-major trajectory [T+1, B, d]: allocated by the branch that writes it (the fused training path keeps [T+1, d, B])
        k = pr.target.n_components if pr.target.kind == L.DENS_GMM else 0
        n_hidden = pr.base_model.n_hidden
        if pr.flags & L.FLAG_INFERENCE_CTRL:
            n_hidden = max(n_hidden, pr.inference.base_model.n_hidden)
        plan = self._plan(device, dim, pr.base_model.channels, n_hidden, n_steps, k)
        if (pr.flags & L.FLAG_INFERENCE_CTRL) and (pr.base_model.channels != 64 or dim > 64):
            plan.reserve(batch)  # the wide Bridge's divergence scratch (no stream-ordered call of the library allocates)
        if seed is None:
            seed = torch.initial_seed()
        offset = self.offset()
        self.calls += 1
        stream = torch.cuda.current_stream(device).cuda_stream
        gp = torch.empty((n_steps, batch, dim), device=device, dtype=torch.float32) if want_gp else None
        dn_p = None
        if div_noise is not None:
            if tuple(div_noise.shape) != (n_steps, batch, dim):
                raise ValueError(f"div_noise must be [{n_steps}, {batch}, {dim}], got {tuple(div_noise.shape)}")
            dn_p = keep.ptr(div_noise, device, "div_noise")
        if pr.target.kind == L.DENS_EXTERNAL:
            # a target whose score the kernels do not carry (the NICE flow, csrc/sdeh_nice.hip): the wide kernels run the grid one step
            # at a time (sdeh_simulate_fwd_steps; tables prepared once, Philox counters of the whole grid), the engine evaluates the
            # target's score at x_t in between -- every launch on this stream, no host synchronisation
            if div_noise is not None:
                raise _unsupported("Hutchinson divergence estimators with a NICE target (the wide kernels carry the exact divergence)")
            ext = next((k for k in keep if isinstance(k, _ExternalTarget)), None)
            if ext is None:
                raise RuntimeError("SDEH_DENS_EXTERNAL without its target object (problem built outside build_problem?)")
            training = want_planes or want_gp
            need_score = pr.ctrl_kind in (L.CTRL_SCORE, L.CTRL_LERP, L.CTRL_LERP_TARGET)
            bptt = training and not (pr.flags & L.FLAG_CHANGE_SDE_CTRL)
            terminal = bool(pr.flags & L.FLAG_TERMINAL_TARGET)
            if return_traj or training:
                xs = torch.empty((n_steps + 1, batch, dim), device=device, dtype=torch.float32)
            # the score plane: every step's row is kept when a backward pass will want it, one [B, d] buffer otherwise
            sc = (torch.empty((n_steps if training else 1, batch, dim), device=device, dtype=torch.float32) if need_score else None)
            ping = (torch.empty((batch, dim), device=device, dtype=torch.float32), torch.empty((batch, dim), device=device, dtype=torch.float32))
            nkeep = _Keep()
            desc = (describe_nice(ext.obj, device, nkeep), nkeep)
            keep.extend(nkeep)
            cur = x.detach()
            if cur.dtype != torch.float32 or not cur.is_contiguous():
                cur = cur.float().contiguous()
            cur_ptr = cur.data_ptr()
            # a control without a score term (ClippedCtrl on the flow) only meets the flow in its terminal cost: the whole grid is one segment
            seg = 1 if need_score else n_steps
            with torch.cuda.device(device):
                for i in range(0, n_steps, seg):
                    sc_i = None
                    if need_score:
                        sc_i = sc[i if training else 0]
                        nice_eval(ext.obj, cur, want_score=True, want_logp=False, cache=self._nice_work, score_out=sc_i, desc=desc)
                    out = x_T if i + seg >= n_steps else ping[i & 1]
                    L.check(lib.sdeh_simulate_fwd_steps(plan.handle, C.byref(pr), ts_p, n_steps, i, min(i + seg, n_steps), cur_ptr, batch, noise_p,
                                                        seed & 0xFFFFFFFFFFFFFFFF, offset, row_offset, out.data_ptr(), rnd.data_ptr(),
                                                        None if xs is None else xs.data_ptr(), None if gp is None else gp.data_ptr(),
                                                        None if sc_i is None else sc_i.data_ptr(), 0, stream))
                    cur, cur_ptr = out, out.data_ptr()
                tscore = None
                if terminal:
                    # the terminal cost rnd -= clip(target.unnorm_log_prob(x_T), clip_target) (losses/oc.py:225, solver/oc.py:48-54) -- the
                    # segments leave it to the caller of a supplied target -- and, for back-propagation through time, its derivative
                    # 1[|log rho| <= clip_target] score(x_T) (torch.clamp's backward), the plane the wide backward takes as `tscore_in`
                    score_T, lp = nice_eval(ext.obj, x_T, want_score=bptt, want_logp=True, cache=self._nice_work, desc=desc)
                    ct = pr.clip_target
                    rnd.sub_((lp.clamp(-ct, ct) if math.isfinite(ct) else lp).unsqueeze(-1))
                    if bptt:
                        tscore = score_T if not math.isfinite(ct) else score_T * (lp.abs() <= ct).unsqueeze(-1).to(score_T.dtype)
            sc_in = None
            if training and need_score:
                # what the backward kernels take as `sc_in`: the score ENTERING the control, i.e. after the control's interpolation weight
                # (reparam.py:185-197: t / T * target_score) and before clip_score / gamma(t) -- the kernels' own product, in fp32.  The flow's
                # score is a constant of the reference's autograd graph (Distribution.score: create_graph = False, or x detached), also
                # under back-propagation through time: no derivative of the flow's score is ever needed
                sc_in = sc
                if pr.ctrl_kind in (L.CTRL_LERP_TARGET, L.CTRL_LERP):
                    # (tensor / tensor: a true fp32 division like the kernels' `s / terminal_t`; torch.full is a fill kernel -- capture-safe, no host copy)
                    wl = ts.reshape(-1)[:-1].to(device=device, dtype=torch.float32) / torch.full((1,), pr.terminal_t, dtype=torch.float32, device=device)
                    if pr.ctrl_kind == L.CTRL_LERP_TARGET:
                        sc_in.mul_(wl.view(-1, 1, 1))
                    else:  # LerpCtrl (reparam.py:131-144): torch.lerp(prior_score(x_t), target_score(x_t), t / T) on the stored trajectory
                        pref = next((k for k in keep if isinstance(k, _PriorRef)), None)
                        if pref is None:
                            raise RuntimeError("LerpCtrl on a supplied target without its prior (problem built outside build_problem?)")
                        with torch.no_grad():
                            psc = pref.obj.score(xs[:-1].reshape(-1, dim)).reshape(n_steps, batch, dim)
                        sc_in = torch.lerp(psc, sc, wl.view(-1, 1, 1))
            if want_planes:
                return x_T, rnd, xs, ("wide", sc_in, tscore)
            if want_gp:
                return x_T, rnd, xs, gp, (sc_in, tscore)
            return x_T, rnd, (xs if return_traj else None)
        if want_planes:  # training forward: keep what the backward kernels need
            if not return_traj or want_gp or div_noise is not None:
                raise ValueError("want_planes goes with return_traj=True and without the Bridge outputs")
            if want_u and (pr.base_model.channels != 64 or dim > 64 or 64 * batch * 4 >= 2 ** 32):
                raise ValueError("want_u: 64-channel networks, d <= 64, planes within 32-bit byte offsets")
            if pr.base_model.channels != 64 or dim > 64:
                # wide networks (csrc/sdeh_wide_bwd.hip): the forward keeps the trajectory only; the backward re-evaluates the network
                # on the matrix pipe at the stored states
                xs = torch.empty((n_steps + 1, batch, dim), device=device, dtype=torch.float32)
                if pr.target.kind == L.DENS_GMM:
                    # mixture target: the backward evaluates no mixture -- the forward keeps the score entering the control and (kl) the
                    # terminal target score, row-major (sdeh_simulate_fwd_train2 on a wide plan)
                    need_sc = pr.ctrl_kind in (L.CTRL_SCORE, L.CTRL_LERP, L.CTRL_LERP_TARGET)
                    bptt = not (pr.flags & L.FLAG_CHANGE_SDE_CTRL)
                    sc = torch.empty((n_steps, batch, dim), device=device, dtype=torch.float32) if need_sc else None
                    tscore = (torch.empty((batch, dim), device=device, dtype=torch.float32)
                              if bptt and (pr.flags & L.FLAG_TERMINAL_TARGET) else None)
                    with torch.cuda.device(device):
                        status = lib.sdeh_simulate_fwd_train2(plan.handle, C.byref(pr), ts_p, n_steps, x_p, batch, noise_p,
                                                              seed & 0xFFFFFFFFFFFFFFFF, offset, row_offset, x_T.data_ptr(),
                                                              rnd.data_ptr(), xs.data_ptr(), None if sc is None else sc.data_ptr(),
                                                              None if tscore is None else tscore.data_ptr(), stream)
                    L.check(status if status < 0 else 0)
                    return x_T, rnd, xs, ("wide", sc, tscore)
                with torch.cuda.device(device):
                    L.check(lib.sdeh_simulate_fwd_aux(plan.handle, C.byref(pr), ts_p, n_steps, x_p, batch, noise_p,
                                                      seed & 0xFFFFFFFFFFFFFFFF, offset, row_offset, x_T.data_ptr(), rnd.data_ptr(),
                                                      xs.data_ptr(), None, None, stream))
                return x_T, rnd, xs, None
            # (the fused backward addresses its [d][B] planes with 32-bit byte offsets: beyond 2^24 trajectories the plane path)
            if 64 * batch * 4 < 2 ** 32 and (want_u or lib.sdeh_ctrl_backward_fused_supported(plan.handle, C.byref(pr))):
                # fused backward (csrc/sdeh_bwdf.hip): the combined score per step and the terminal target score, no [C, T*B] planes
                # (coordinate-major planes: [.., d, B])
                xs_cm = torch.empty((n_steps + 1, dim, batch), device=device, dtype=torch.float32)
                sc = (torch.empty((n_steps, dim, batch), device=device, dtype=torch.float32)
                      if pr.ctrl_kind != L.CTRL_CLIPPED else None)
                bptt = not (pr.flags & L.FLAG_CHANGE_SDE_CTRL)
                tscore = (torch.empty((dim, batch), device=device, dtype=torch.float32)
                          if bptt and (pr.flags & L.FLAG_TERMINAL_TARGET) else None)
                # want_u (split Bridge training, losses/_autograd.py): also the control driving the SDE, [T, d, B]
                u_cm = torch.empty((n_steps, dim, batch), device=device, dtype=torch.float32) if want_u else None
                # the pre-activation record (with the raw network output; ABI v6): kept when the backward launch that will serve this problem
                # reads them (then it does not re-evaluate the network) and the record fits the memory budget
                zrec = None
                n_z = lib.sdeh_zrec_floats(dim, pr.base_model.n_hidden, n_steps, batch)
                # (a split Bridge with method kl takes its running cost and adjoint planes through kernels that re-evaluate: nothing kept)
                if (not (want_u and bptt) and lib.sdeh_ctrl_backward_fused_reads_zrec(plan.handle, C.byref(pr), batch)
                        and _zrec_fits(4 * n_z, device)):
                    zrec = _alloc_zrec(n_z, device)
                with torch.cuda.device(device):
                    if zrec is not None:
                        status = lib.sdeh_simulate_fwd_train3(plan.handle, C.byref(pr), ts_p, n_steps, x_p, batch, noise_p,
                                                              seed & 0xFFFFFFFFFFFFFFFF, offset, row_offset, x_T.data_ptr(),
                                                              rnd.data_ptr(), xs_cm.data_ptr(), None if sc is None else sc.data_ptr(),
                                                              None if tscore is None else tscore.data_ptr(),
                                                              None if u_cm is None else u_cm.data_ptr(), zrec.data_ptr(), stream)
                    elif want_u:
                        status = lib.sdeh_simulate_fwd_train2u(plan.handle, C.byref(pr), ts_p, n_steps, x_p, batch, noise_p,
                                                               seed & 0xFFFFFFFFFFFFFFFF, offset, row_offset, x_T.data_ptr(),
                                                               rnd.data_ptr(), xs_cm.data_ptr(), None if sc is None else sc.data_ptr(),
                                                               None if tscore is None else tscore.data_ptr(), u_cm.data_ptr(), stream)
                    else:
                        status = lib.sdeh_simulate_fwd_train2(plan.handle, C.byref(pr), ts_p, n_steps, x_p, batch, noise_p,
                                                              seed & 0xFFFFFFFFFFFFFFFF, offset, row_offset, x_T.data_ptr(),
                                                              rnd.data_ptr(), xs_cm.data_ptr(), None if sc is None else sc.data_ptr(),
                                                              None if tscore is None else tscore.data_ptr(), stream)
                if status < 0:
                    L.check(status)
                if status == 0:
                    return x_T, rnd, xs_cm, ("fused", sc, tscore, u_cm, zrec)
                if want_u:  # served by a kernel that keeps no planes (mixture tables beyond LDS, SDEH_LEGACY): the caller falls back
                    return x_T, rnd, None, None
                del xs_cm, sc, tscore, zrec  # integrated by a kernel that keeps no planes (mixture tables beyond LDS): once more, the plane way
            xs = torch.empty((n_steps + 1, batch, dim), device=device, dtype=torch.float32)
            zt = torch.empty((pr.base_model.n_hidden + 1, pr.base_model.channels, n_steps * batch), device=device, dtype=torch.float32)
            nn = torch.empty((n_steps, batch, dim), device=device, dtype=torch.float32)
            with torch.cuda.device(device):
                status = lib.sdeh_simulate_fwd_train(plan.handle, C.byref(pr), ts_p, n_steps, x_p, batch, noise_p,
                                                     seed & 0xFFFFFFFFFFFFFFFF, offset, row_offset, x_T.data_ptr(),
                                                     rnd.data_ptr(), xs.data_ptr(), zt.data_ptr(), nn.data_ptr(), stream)
            if status < 0:
                L.check(status)
            return x_T, rnd, xs, ((zt, nn) if status == 0 else None)  # 1: served by a kernel that keeps no planes
        if return_traj:
            xs = torch.empty((n_steps + 1, batch, dim), device=device, dtype=torch.float32)
        sc = tscore = None
        if want_gp and pr.target.kind == L.DENS_GMM and (pr.base_model.channels != 64 or dim > 64):
            # training forward of a wide Bridge on a mixture target: the backward of the generative network evaluates no mixture -- keep
            # the score entering it and (kl) the terminal target score, as the plain wide training forward does
            if pr.ctrl_kind in (L.CTRL_SCORE, L.CTRL_LERP, L.CTRL_LERP_TARGET):
                sc = torch.empty((n_steps, batch, dim), device=device, dtype=torch.float32)
            if not (pr.flags & L.FLAG_CHANGE_SDE_CTRL) and (pr.flags & L.FLAG_TERMINAL_TARGET):
                tscore = torch.empty((batch, dim), device=device, dtype=torch.float32)
        with torch.cuda.device(device):
            L.check(lib.sdeh_simulate_fwd_aux2(plan.handle, C.byref(pr), ts_p, n_steps, x_p, batch, noise_p,
                                               seed & 0xFFFFFFFFFFFFFFFF, offset, row_offset, x_T.data_ptr(),
                                               rnd.data_ptr(), None if xs is None else xs.data_ptr(),
                                               None if gp is None else gp.data_ptr(), dn_p,
                                               None if sc is None else sc.data_ptr(),
                                               None if tscore is None else tscore.data_ptr(), stream))
        if want_gp:
            return x_T, rnd, xs, gp, (sc, tscore)
        return x_T, rnd, xs

    # ------------------------------------------------------------------------------------------------------

_SCRATCH: dict = {}


#: (device index, record bytes) -> bool: the budget decision, taken ONCE per size (a decision re-taken from the allocator's
#: momentary state could flip between steps or differ across ranks -- and with it the kernels that serve a step and its low-order bits)
_ZREC_DECISION: dict = {}


def _zrec_fits(nbytes: int, device) -> bool:
    """Memory budget of the pre-activation record (768 B per trajectory-step with two hidden layers: 5 GB at B = 65 536, T = 100): at most
    SDEH_ZREC_BYTES (a fixed byte budget), by default 40 % of what was free on the device when a record of this size was first asked for
    (the caching allocator's idle blocks included).  The decision is cached per (device, size): no driver query on the launch path after
    the first step, the same kernels every step.  A failed allocation withdraws it (`_alloc_zrec`); `reset_zrec_budget()` forgets all."""
    cap = os.environ.get("SDEH_ZREC_BYTES")
    if cap is not None:
        return nbytes <= int(float(cap))
    key = (torch.device(device).index, int(nbytes))
    fits = _ZREC_DECISION.get(key)
    if fits is None:
        free, _total = torch.cuda.mem_get_info(device)
        free += max(0, torch.cuda.memory_reserved(device) - torch.cuda.memory_allocated(device))
        fits = _ZREC_DECISION[key] = bool(nbytes <= 0.4 * free)
    return fits


def _alloc_zrec(n_floats: int, device):
    """The record's storage (uninitialised: the forward launch writes every word a backward launch reads, rows of the ragged last tile
    included).  Out of memory -> None (the launch that re-evaluates the network serves the step) and the cached decision is withdrawn."""
    try:
        return torch.empty(n_floats, device=device, dtype=torch.float32)
    except torch.OutOfMemoryError:
        _ZREC_DECISION[(torch.device(device).index, 4 * int(n_floats))] = False
        return None


def reset_zrec_budget() -> None:
    """Forget the cached record-budget decisions (after freeing or claiming a large share of the device memory)."""
    _ZREC_DECISION.clear()


#: set by utils.graphs.GraphedEval while it captures an evaluation: BaseOCLoss.compute_results then leaves its host half to the replay
_deferred: dict | None = None


def estimator_stats(rnd: torch.Tensor, max_rnd: float = math.nan) -> torch.Tensor:
    """8 mergeable partial statistics of `rnd` (include/sdeh.h: sdeh_reduce_estimators), on device."""
    dev = rnd.device
    scratch = _SCRATCH.get(dev)
    if scratch is None:  # stream-ordered reuse: consecutive reductions on one device are serialised by the stream
        scratch = _SCRATCH[dev] = torch.empty(L.SDEH_REDUCE_SCRATCH, device=dev, dtype=torch.float32)
    out = torch.empty(8, device=dev, dtype=torch.float32)
    rnd = rnd.contiguous()
    stream = torch.cuda.current_stream(dev).cuda_stream
    with torch.cuda.device(dev):
        L.check(L.load().sdeh_reduce_estimators(rnd.data_ptr(), rnd.numel(), max_rnd, scratch.data_ptr(),
                                                out.data_ptr(), stream))
    return out


def loss_moment(rnd: torch.Tensor, max_rnd: float, log_variance: bool, n_filtered: torch.Tensor | None) -> tuple[torch.Tensor, torch.Tensor]:
    """(stats [8] with stats[7] = the loss, d loss / d rnd [like rnd]) of BaseOCLoss.compute_loss for methods kl / lv over the rows the
    filter keeps (include/sdeh.h: sdeh_loss_moment -- three launches, no host round trip); `n_filtered` (int64 device scalar) += dropped rows."""
    dev = rnd.device
    scratch = _SCRATCH.get(dev)
    if scratch is None:
        scratch = _SCRATCH[dev] = torch.empty(L.SDEH_REDUCE_SCRATCH, device=dev, dtype=torch.float32)
    out = torch.empty(8, device=dev, dtype=torch.float32)
    rnd = rnd.contiguous()
    w = torch.empty_like(rnd)
    stream = torch.cuda.current_stream(dev).cuda_stream
    with torch.cuda.device(dev):
        L.check(L.load().sdeh_loss_moment(rnd.data_ptr(), rnd.numel(), max_rnd, int(log_variance),
                                          None if n_filtered is None else n_filtered.data_ptr(), scratch.data_ptr(), out.data_ptr(),
                                          w.data_ptr(), stream))
    return out, w


def importance_weights(rnd: torch.Tensor, log_weight_max: torch.Tensor) -> torch.Tensor:
    """exp(-rnd - m) with the (global) maximum m given as a device scalar."""
    rnd = rnd.contiguous()
    w = torch.empty_like(rnd)
    m = log_weight_max.to(device=rnd.device, dtype=torch.float32).reshape(1).contiguous()
    stream = torch.cuda.current_stream(rnd.device).cuda_stream
    with torch.cuda.device(rnd.device):
        L.check(L.load().sdeh_importance_weights(rnd.data_ptr(), rnd.numel(), m.data_ptr(), w.data_ptr(), stream))
    return w


# ----------------------------------------------------------------------------------------------------------
# mergeable estimator statistics: v = [n, sum(-rnd), M2(rnd), m=max(-rnd), sum e^{-rnd-m}, sum e^{2(-rnd-m)}, n_filtered, 0]
# ----------------------------------------------------------------------------------------------------------
def merge_stats(stats: torch.Tensor) -> torch.Tensor:
    """Combines per-rank statistics [R, 8] into one [8] vector (Chan's parallel variance + rescaled exp-sums).
    Runs on the host in double precision: the payload is 8 numbers per rank (one device->host copy)."""
    rows = stats.detach().reshape(-1, 8).cpu().tolist()
    N = sum(r[0] for r in rows)
    S = sum(r[1] for r in rows)
    NF = sum(r[6] for r in rows)
    valid = [r for r in rows if r[0] > 0]
    if not valid:
        return torch.tensor([0.0, 0.0, 0.0, -math.inf, 0.0, 0.0, NF, 0.0], dtype=torch.float64)
    mean = -S / N  # mean of rnd over all kept rows
    M2 = sum(r[2] + r[0] * (-r[1] / r[0] - mean) ** 2 for r in valid)
    m = max(r[3] for r in valid)
    e1 = sum(r[4] * math.exp(r[3] - m) for r in valid)
    e2 = sum(r[5] * math.exp(2.0 * (r[3] - m)) for r in valid)
    return torch.tensor([N, S, M2, m, e1, e2, NF, 0.0], dtype=torch.float64)


def all_gather_stats(stats: torch.Tensor, group=None) -> torch.Tensor:
    """The single collective of an evaluation: all-gather 8 floats per rank (RCCL on GPU tensors, gloo on CPU
    tensors), then merge on the host.  Without an initialised process group this is just the local merge.  With one, the
    collective runs at EVERY world size (also 1): a single-rank job executes exactly the code an 8-rank job does
    (tests/test_hip_rccl.py runs it over RCCL on one GPU and compares bitwise with the group-less path)."""
    import torch.distributed as dist

    if not (dist.is_available() and dist.is_initialized()):
        return merge_stats(stats)
    world = dist.get_world_size(group)
    if dist.get_backend(group) == "gloo":  # CPU collectives (tests): move the 8 floats to the host first
        stats = stats.detach().cpu()
    bucket = torch.empty(world * 8, dtype=stats.dtype, device=stats.device)
    dist.all_gather_into_tensor(bucket, stats.contiguous().reshape(8), group=group)
    return merge_stats(bucket)


def estimators_from_stats(v: torch.Tensor) -> dict:
    """log_norm_const_lb(_ito) = mean(-rnd); log_norm_const_is = log mean exp(-rnd); lv_loss = var(rnd) (unbiased);
    ess = (sum w)^2 / sum w^2  (reference: losses/oc.py:94-123, eval/metrics.py:121-126)."""
    n, s, m2, m, e1, e2, nf = (float(t) for t in v[:7])
    return {
        "n": n, "n_filtered": nf,
        "mean_neg_rnd": s / n if n > 0 else math.nan,
        "log_norm_const_is": math.log(e1 / n) + m if n > 0 and e1 > 0 else math.nan,
        "var_rnd": m2 / (n - 1) if n > 1 else math.nan,
        "mean_rnd": -s / n if n > 0 else math.nan,
        "log_weight_max": m,
        "ess": (e1 * e1 / e2) if e2 > 0 else math.nan,
    }

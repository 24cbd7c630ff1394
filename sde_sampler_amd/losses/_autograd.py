"""Autograd bridge for training through the fused trajectory kernels.

`simulate_with_grad(...)` returns `(x_T, rnd, None)` like `simulate()`, with `rnd` attached to the autograd graph of the
control network's parameters.  Forward = the HIP trajectory kernel (keeping the trajectory `xs`); backward =
`sdeh_ctrl_backward` (HIP: per-row re-evaluation + back-propagation on the matrix pipe, noise replayed from the
Philox counters; back-propagation through time for method "kl" / "kl_ito") followed by `sdeh_weight_grad` (one pass over
the N = T*B rows per layer: activation on the fly, MFMA contraction, bias sums) and by autograd on the two time-only
sub-networks' [T, .] tables.

`simulate_bridge_with_grad(...)` does the same for a Bridge (TimeReversalLoss with an inference control) and the
log-variance methods: the generative network as above, the inference network through `sdeh_ctrl_backward_ex` (upstream
gradient (u + v) dt + dB) plus `sdeh_bridge_div_backward` (the divergence term: reverse mode over the forward-mode
tangents, see include/sdeh.h).
"""
from __future__ import annotations

import ctypes as C
import os

import torch

from sde_sampler_amd import _lib as L
from sde_sampler_amd import engine as E


def _ctrl_parameters(ctrl) -> list[torch.nn.Parameter]:
    return [p for p in ctrl.parameters() if p.requires_grad]


def _ctrl_backward(engine, pr, keep, ts, xs, w, st, gextra=None, cost_ctrl=None, lam_extra=None, dx_out=None, planes=None,
                   sc_in=None, tscore_in=None):
    """Runs sdeh_ctrl_backward_ex for the control in the problem's generative slots; returns the planes.  `planes` = (zt, nn) kept
    by the training forward (sdeh_simulate_fwd_train): the kernel then reads them instead of re-evaluating the network."""
    dev = xs.device
    T, B, d = xs.shape[0] - 1, xs.shape[1], xs.shape[2]
    N, Cn, Lh = T * B, pr.base_model.channels, pr.base_model.n_hidden
    g = 0 if pr.ctrl_kind == L.CTRL_CLIPPED else (pr.score_model.dim_out if pr.score_model.n_hidden > 0 else 1)
    zt = planes[0] if planes is not None else torch.empty((Lh + 1, Cn, N), device=dev, dtype=torch.float32)
    nn_in = planes[1] if planes is not None else None
    dt = torch.empty_like(zt)
    dout = torch.empty((d, N), device=dev, dtype=torch.float32)
    dgam = torch.zeros((max(g, 1), N), device=dev, dtype=torch.float32)
    # wide networks: the chain kernel also leaves x_t coordinate-major (the operand of input_embed.weight's gradient)
    xt = torch.empty((d, N), device=dev, dtype=torch.float32) if (Cn != 64 or d > 64) else None
    plan = engine._plan(dev, d, Cn, Lh, T, pr.target.n_components if pr.target.kind == L.DENS_GMM else 0)
    noise = st["noise"]
    stream = torch.cuda.current_stream(dev).cuda_stream
    ptr = lambda t: None if t is None else t.data_ptr()
    with torch.cuda.device(dev):
        L.check(L.load().sdeh_ctrl_backward_ex(
            plan.handle, C.byref(pr), keep.ptr(ts.reshape(-1), dev, "ts"), T, xs.data_ptr(), B,
            None if noise is None else keep.ptr(noise, dev, "noise"), st["seed"], st["offset"], st["row_offset"],
            w.data_ptr(), ptr(gextra), ptr(cost_ctrl), ptr(lam_extra), ptr(dx_out), zt.data_ptr(), dt.data_ptr(),
            dout.data_ptr(), dgam.data_ptr(), ptr(nn_in), ptr(xt), ptr(sc_in), ptr(tscore_in), stream))
    return (zt, dt, dout, dgam) if xt is None else (zt, dt, dout, dgam, xt)


def _fused_backward(loss, pr, keep, ts, xs, w, st, sc, tscore, cost_ctrl=None, lam_extra=None, zrec=None) -> dict[int, torch.Tensor]:
    """Parameter gradients of the generative control from sdeh_ctrl_backward_fused (back-propagation and weight gradients in one
    kernel; csrc/sdeh_bwdf.hip) + sdeh_time_embed_backward on the two [T, .] tables.  `zrec`: the pre-activation record of
    sdeh_simulate_fwd_train3 -- the launch then reads it instead of re-evaluating the network (ABI v6)."""
    ctrl, engine = loss.generative_ctrl, loss.engine
    base = ctrl.base_model
    dev = xs.device
    T, d, B = xs.shape[0] - 1, xs.shape[1], xs.shape[2]  # coordinate-major planes of sdeh_simulate_fwd_train2
    score_model = getattr(ctrl, "score_model", None) if pr.ctrl_kind != L.CTRL_CLIPPED else None
    g = 1 if score_model is None else score_model.out_layer.out_features
    bptt = not (pr.flags & L.FLAG_CHANGE_SDE_CTRL)
    lib = L.load()
    n_scratch, n_out = C.c_int64(), C.c_int64()
    Lh = len(base.hidden_layer)
    L.check(lib.sdeh_ctrl_backward_fused_sizes(d, Lh, T, B, g, int(bptt), C.byref(n_scratch), C.byref(n_out)))
    scratch = torch.empty(n_scratch.value, device=dev, dtype=torch.float32)
    out = torch.empty(n_out.value, device=dev, dtype=torch.float32)
    plan = engine._plan(dev, d, base.channels, len(base.hidden_layer), T, pr.target.n_components if pr.target.kind == L.DENS_GMM else 0)
    noise = st["noise"]
    ptr = lambda t: None if t is None else t.data_ptr()
    with torch.cuda.device(dev):
        # cost_ctrl / lam_extra [T, d, B] (split Bridge, method kl): the running cost on u + v, the inference terms' d loss / d x_t
        L.check(lib.sdeh_ctrl_backward_fused_z(
            plan.handle, C.byref(pr), keep.ptr(ts.reshape(-1), dev, "ts"), T, xs.data_ptr(), B,
            None if noise is None else keep.ptr(noise, dev, "noise"), st["seed"], st["offset"], st["row_offset"],
            w.data_ptr(), ptr(sc), ptr(tscore), ptr(cost_ctrl), ptr(lam_extra), ptr(zrec), scratch.data_ptr(), scratch.numel(),
            out.data_ptr(), torch.cuda.current_stream(dev).cuda_stream))
    return _fused_record_grads(ctrl, ts, out, d, T, Lh, g, score_model)


def _fused_record_grads(ctrl, ts, out, d, T, Lh, g, score_model, div: bool = False) -> dict[int, torch.Tensor]:
    """{id(parameter): gradient} from the `out` record of sdeh_ctrl_backward_fused (include/sdeh.h); `div`: followed by the direct
    weight gradients of a Bridge's divergence term (sdeh_bridge_backward_fused), which are added."""
    base = ctrl.base_model
    P, gw = 32 * ((d + 31) // 32), (2 if g == 1 else 64)
    grads: dict[int, torch.Tensor] = {}
    pos = 0

    def take(n, *shape):
        nonlocal pos
        v = out[pos:pos + n].view(*shape)
        pos += n
        return v

    with torch.no_grad():
        grads[id(base.input_embed.weight)] = take(64 * P, 64, P)[:, :d].contiguous()
        w_hid = take(Lh * 4096, Lh, 64, 64)
        grads[id(base.out_layer.weight)] = take(P * 64, P, 64)[:d]
        b_hid = take(Lh * 64, Lh, 64)
        grads[id(base.out_layer.bias)] = take(P, P)[:d]
        for k, lin in enumerate(base.hidden_layer):
            grads[id(lin.weight)], grads[id(lin.bias)] = w_hid[k], b_hid[k]
        d_emb = take(T * 64, T, 64)
        d_gam = take(T * gw, T, gw)
        grads[id(base.input_embed.bias)] = d_emb.sum(dim=0)
        d_gam = d_gam.sum(dim=1, keepdim=True) if g == 1 else d_gam[:, :d].contiguous()
        if div:
            dv_hid = take(Lh * 4096, Lh, 64, 64)
            dv_in = take(P * 64, P, 64)[:d]
            dv_out = take(P * 64, P, 64)[:d]
            for k, lin in enumerate(base.hidden_layer):
                grads[id(lin.weight)] = grads[id(lin.weight)] + dv_hid[k]
            grads[id(base.input_embed.weight)] = grads[id(base.input_embed.weight)] + dv_in.t()
            grads[id(base.out_layer.weight)] = grads[id(base.out_layer.weight)] + dv_out
    grads.update(_time_table_grads(ctrl, ts, d_emb, d_gam if score_model is not None else None))
    return grads


def _plan_option(eng, name: str):
    """The value a launch of `eng` would push into its plan for option `name` -- `engine.options` (an explicit None = automatic) over the
    environment, as `engine._Plan.sync_options` resolves it; like the library (`plan_opt`), any non-empty value counts as set."""
    if name in eng.options:
        return eng.options[name] or None
    return os.environ.get(name) or None


def _bridge_fused_ok(eng, inf_model, d, T, B, st) -> bool:
    """Does the fused inference-network backward (sdeh_bridge_backward_fused) serve this Bridge?  64 channels, two hidden layers, the
    exact divergence, planes within 32-bit byte offsets; plan option / environment SDEH_BWD_PLANES keeps the plane-writing kernels."""
    return (inf_model.channels == 64 and d <= 64 and len(inf_model.hidden_layer) == 2 and st.get("div_noise") is None
            and not _plan_option(eng, "SDEH_BWD_PLANES") and 64 * T * B * 4 < 2 ** 32)


def _bridge_fused_inference(eng, pr_v, keep_v, inf, ts, xs, gp, w, st, cm: bool = False, dx_out=None) -> dict[int, torch.Tensor] | None:
    """Every gradient of a 64-channel Bridge's inference network from sdeh_bridge_backward_fused (csrc/sdeh_bridgef.hip: no
    per-coordinate planes), or None where that path is not compiled (then: sdeh_ctrl_backward_ex + sdeh_bridge_div_backward)."""
    base = inf.base_model
    dev = xs.device
    T, B, d = (xs.shape[0] - 1, xs.shape[2], xs.shape[1]) if cm else (xs.shape[0] - 1, xs.shape[1], xs.shape[2])  # cm: [T + 1, d, B] planes
    Lh = len(base.hidden_layer)
    if not _bridge_fused_ok(eng, base, d, T, B, st) or not (pr_v.flags & L.FLAG_CHANGE_SDE_CTRL) \
            or pr_v.ctrl_kind not in (L.CTRL_CLIPPED, L.CTRL_LERP_PRIOR):
        return None
    score_model = getattr(inf, "score_model", None) if pr_v.ctrl_kind != L.CTRL_CLIPPED else None
    g = 1 if score_model is None else score_model.out_layer.out_features
    lib = L.load()
    n_scratch, n_out = C.c_int64(), C.c_int64()
    L.check(lib.sdeh_bridge_backward_fused_sizes(d, Lh, T, B, g, C.byref(n_scratch), C.byref(n_out)))
    xs_cm = xs if cm else xs.permute(0, 2, 1).contiguous()   # [T + 1, d, B]: the kernels move whole cache lines of consecutive trajectories
    gp_cm = gp if cm else gp.permute(0, 2, 1).contiguous()   # [T, d, B]
    scratch = torch.empty(n_scratch.value, device=dev, dtype=torch.float32)
    out = torch.empty(n_out.value, device=dev, dtype=torch.float32)
    plan = eng._plan(dev, d, base.channels, Lh, T, 0)
    noise = st["noise"]
    with torch.cuda.device(dev):
        L.check(lib.sdeh_bridge_backward_fused(
            plan.handle, C.byref(pr_v), keep_v.ptr(ts.reshape(-1), dev, "ts"), T, xs_cm.data_ptr(), B,
            None if noise is None else keep_v.ptr(noise, dev, "noise"), st["seed"], st["offset"], st["row_offset"],
            w.data_ptr(), gp_cm.data_ptr(), None if dx_out is None else dx_out.data_ptr(), scratch.data_ptr(), scratch.numel(), out.data_ptr(),
            torch.cuda.current_stream(dev).cuda_stream))
    return _fused_record_grads(inf, ts, out, d, T, Lh, g, score_model, div=True)


def _time_table_grads(ctrl, ts, d_emb, d_gam) -> dict[int, torch.Tensor]:
    """Parameter gradients of the two time-only sub-networks from the gradients of their [T, .] tables (sdeh_time_embed_backward;
    autograd on the tables only for shapes that kernel does not take)."""
    base = ctrl.base_model
    act_id = E._activation_id(base.activation)
    grads: dict[int, torch.Tensor] = {}
    steps = ts[:-1].contiguous()
    te_params = [p for p in base.timestep_embed.parameters() if p.requires_grad]
    if te_params:
        got = _time_embed_grads(base.timestep_embed, act_id, steps, d_emb, None)
        if got is None:
            with torch.enable_grad():
                emb = base.timestep_embed(steps)
                got = {id(p): gp for p, gp in zip(te_params, torch.autograd.grad(emb, te_params, d_emb, allow_unused=True))}
        grads.update(got)
    score_model = getattr(ctrl, "score_model", None)
    if score_model is not None and d_gam is not None:
        sm_params = [p for p in score_model.parameters() if p.requires_grad]
        if sm_params:
            clip_model = getattr(ctrl, "clip_model", None)
            got = _time_embed_grads(score_model, E._activation_id(score_model.activation), steps, d_gam, clip_model)
            if got is None:
                with torch.enable_grad():
                    gam = score_model(steps)
                    if clip_model is not None:
                        gam = gam.clip(min=-clip_model, max=clip_model)
                    got = {id(p): gp for p, gp in zip(sm_params, torch.autograd.grad(gam, sm_params, d_gam, allow_unused=True))}
            grads.update(got)
    return grads


def _wgrad_batch(items: list[tuple[torch.Tensor, torch.Tensor, int]]) -> list[tuple[torch.Tensor, torch.Tensor]]:
    """[(dmat @ act(z)^T [m, c], dmat.sum(1) [m]) for (dmat [m, N], z [c, N], act) in items], every product in ONE pass over its
    two coordinate-major planes (sdeh_weight_grad: activation on the fly, K-contiguous MFMA operands) and ONE reduction of all
    the per-chunk partials at the end (sdeh_partial_sums).  All items share N; m, c <= 256 (the products of the wide networks are
    tiled into [64, 64] blocks by the kernel, partials padded to multiples of 64)."""
    N = items[0][0].shape[1]
    dev = items[0][0].device
    pad = lambda v: 64 * ((v + 63) // 64)
    mp = max(pad(dm.shape[0]) for dm, _, _ in items)
    cp = max(pad(z.shape[0]) for _, z, _ in items)
    # ~8192 waves per product (a few per SIMD hide the HBM latency; one wave = one [64, 64] block of one chunk); at least 128 rows
    # per partial keeps the partial sums small
    blocks = (mp // 64) * (cp // 64)
    chunk = max(128, -(-N // max(256, 8192 // blocks)))
    chunk = (chunk + 31) // 32 * 32
    n_chunks = -(-N // chunk)
    lib = L.load()
    stream = torch.cuda.current_stream(dev).cuda_stream
    keep = []
    out = []
    # one partial buffer for the 64-channel case (every item [64, 64]: a single reduction launch); per-shape buffers otherwise
    groups: dict[tuple[int, int], list[int]] = {}
    for i, (dmat, z, _) in enumerate(items):
        groups.setdefault((pad(dmat.shape[0]), pad(z.shape[0])), []).append(i)
    results: list = [None] * len(items)
    with torch.cuda.device(dev):
        for (gm, gc), idxs in groups.items():
            n = len(idxs)
            part_w = torch.empty((n, n_chunks, gm, gc), device=dev, dtype=torch.float32)
            part_b = torch.empty((n, n_chunks, gm), device=dev, dtype=torch.float32)
            for slot, i in enumerate(idxs):
                dmat, z, act_id = items[i]
                dmat, z = dmat.contiguous(), z.contiguous()
                keep.append((dmat, z))
                L.check(lib.sdeh_weight_grad(dmat.data_ptr(), dmat.shape[0], z.data_ptr(), z.shape[0], N, act_id, chunk,
                                             part_w[slot].data_ptr(), part_b[slot].data_ptr(), stream))
            # the sums over the chunks are done by the library too (deterministic two-pass kernel): torch's multi-block reduction is
            # not safe to replay inside a hipGraph on this stack (see csrc/sdeh_wgrad.hip)
            w = torch.empty((n, gm, gc), device=dev, dtype=torch.float32)
            b = torch.empty((n, gm), device=dev, dtype=torch.float32)
            scratch = torch.empty(lib.sdeh_partial_sums_scratch_floats(n, n_chunks, gm * gc), device=dev, dtype=torch.float32)
            L.check(lib.sdeh_partial_sums(part_w.data_ptr(), n, n_chunks, gm * gc, scratch.data_ptr(), w.data_ptr(), stream))
            L.check(lib.sdeh_partial_sums(part_b.data_ptr(), n, n_chunks, gm, scratch.data_ptr(), b.data_ptr(), stream))
            for slot, i in enumerate(idxs):
                dmat, z, _ = items[i]
                results[i] = (w[slot, :dmat.shape[0], :z.shape[0]], b[slot, :dmat.shape[0]])
    return results


def _wgrad(dmat: torch.Tensor, z: torch.Tensor, act_id: int) -> tuple[torch.Tensor, torch.Tensor]:
    return _wgrad_batch([(dmat, z, act_id)])[0]


def _time_embed_grads(te, act_id: int, steps: torch.Tensor, table_grad: torch.Tensor, clip) -> dict[int, torch.Tensor] | None:
    """{id(parameter): gradient} of a TimeEmbed module from the gradient of its [T, dim_out] table (one kernel launch), or None
    when the module is not one the kernel takes (then the caller differentiates the table with autograd)."""
    if "TimeEmbed" not in E._mro_names(te) or os.environ.get("SDEH_TE_AUTOGRAD"):  # testing aid: differentiate the table with autograd
        return None
    dev = table_grad.device
    keep = E._Keep()
    st = L.SdehTimeEmbed()
    E._fill_time_embed(te, st, keep, dev, "time embedding")
    if not 1 <= st.n_hidden <= 4 or st.channels > 64:  # (the kernel stages a [C, 2C] layer in LDS: 64 channels)
        return None
    layers = list(E._sub(te, "hidden_layer"))
    out_layer, phase = E._sub(te, "out_layer"), E._sub(te, "timestep_phase")
    lib = L.load()
    flat = torch.empty(lib.sdeh_time_embed_param_floats(C.byref(st)), device=dev, dtype=torch.float32)
    work = torch.empty(lib.sdeh_time_embed_workspace_floats(C.byref(st), steps.numel()), device=dev, dtype=torch.float32)
    table_grad = table_grad.contiguous().float()
    with torch.cuda.device(dev):
        status = lib.sdeh_time_embed_backward(C.byref(st), act_id, keep.ptr(steps, dev, "ts"), steps.numel(), table_grad.data_ptr(),
                                              float("inf") if clip is None else float(clip), work.data_ptr(), flat.data_ptr(),
                                              torch.cuda.current_stream(dev).cuda_stream)
    if status == -2:  # a shape the kernel does not stage in LDS (e.g. a per-coordinate gamma(t) with dim_out > 64): autograd on the table
        return None
    L.check(status)
    # flat layout (include/sdeh.h): phase | (weight, bias) per hidden layer | out_layer weight, bias
    out: dict[int, torch.Tensor] = {}
    pos = 0
    for param in [phase] + [t for lin in layers for t in (lin.weight, lin.bias)] + [out_layer.weight, out_layer.bias]:
        n = param.numel()
        out[id(param)] = flat[pos:pos + n].view(param.shape)
        pos += n
    return out


def _weight_grads(ctrl, ts, xs, zt, dt, dout, dgam, xt=None, extra=None) -> dict[int, torch.Tensor]:
    """Parameter gradients of one control from the coordinate-major planes (sdeh_weight_grad over N; autograd on the [T, .]
    tables of the two time-only sub-networks).  `extra`: additive second-order contributions of the Bridge divergence term."""
    base = ctrl.base_model
    T, B, d = xs.shape[0] - 1, xs.shape[1], xs.shape[2]
    N, Cn, Lh = T * B, base.channels, len(base.hidden_layer)
    score_model = getattr(ctrl, "score_model", None)
    g = 0 if score_model is None else score_model.out_layer.out_features
    act_id = E._activation_id(base.activation)
    grads: dict[int, torch.Tensor] = {}
    extra = extra or {}
    with torch.no_grad():
        # every product of this control goes into one batch: (parameter, wants_bias, planes...); products of the same parameter add
        w_in, w_out = base.input_embed.weight, base.out_layer.weight
        jobs: list[tuple[torch.nn.Parameter, torch.nn.Parameter | None, torch.Tensor, torch.Tensor, int]] = []
        Xt = xt if xt is not None else xs[:T].reshape(N, d).t().contiguous()  # [d, N]
        d0 = dt[0] + extra["d2"][0] if "d2" in extra else dt[0]
        jobs.append((w_in, None, d0, Xt, L.ACT_IDENTITY))
        d_emb = d0.reshape(Cn, T, B).sum(dim=2).t().contiguous()  # [T, C]: gradient of the time embedding table
        grads[id(base.input_embed.bias)] = d_emb.sum(dim=0)
        for k in range(Lh):
            lin = base.hidden_layer[k]
            dk = dt[k + 1] + extra["d2"][k + 1] if "d2" in extra else dt[k + 1]
            jobs.append((lin.weight, lin.bias, dk, zt[k], act_id))
        jobs.append((w_out, base.out_layer.bias, dout, zt[Lh], act_id))
        per_coord: list[tuple[int, int, int]] = []
        if "td" in extra and extra.get("eps") is not None:  # Hutchinson: one tangent stream in direction eps
            td, ta, cj = extra["td"], extra["ta"], extra["cj"]
            jobs.append((w_in, None, td[0, 0], extra["eps"].reshape(N, d).t().contiguous(), L.ACT_IDENTITY))
            for k in range(Lh):
                jobs.append((base.hidden_layer[k].weight, None, td[0, k + 1], ta[0, k], L.ACT_IDENTITY))
            jobs.append((w_out, None, cj, ta[0, Lh], L.ACT_IDENTITY))
        elif "td" in extra:  # tangent streams of the divergence term, one per coordinate j
            td, ta, cj = extra["td"], extra["ta"], extra["cj"]
            for j in range(d):
                for k in range(Lh):
                    jobs.append((base.hidden_layer[k].weight, None, td[j, k + 1], ta[j, k], L.ACT_IDENTITY))
                # column j of input_embed.weight: sum_n td[j, 0][:, n] (the bias sum of that plane); row j of out_layer.weight:
                # sum_n cj[j][n] ta[j, Lh][:, n] (a one-row product) -- through the same kernel, not through framework reductions
                per_coord.append((j, len(jobs), len(jobs) + 1))
                jobs.append((None, None, td[j, 0], cj[j:j + 1], L.ACT_IDENTITY))  # only its bias sum is used
                jobs.append((None, None, cj[j:j + 1], ta[j, Lh], L.ACT_IDENTITY))
        results = _wgrad_batch([(a, b, c) for _, _, a, b, c in jobs])
        for (pw, pb, _, _, _), (wmat, bvec) in zip(jobs, results):
            if pw is not None:
                grads[id(pw)] = grads[id(pw)] + wmat if id(pw) in grads else wmat.clone()
            if pb is not None:
                grads[id(pb)] = bvec.clone()
        for j, i_in, i_out in per_coord:
            grads[id(w_in)][:, j] += results[i_in][1]
            grads[id(w_out)][j] += results[i_out][0][0]
    d_gam = None
    if score_model is not None:
        with torch.no_grad():
            dg = dgam[:g] + extra["dgam"][:g] if "dgam" in extra else dgam[:g]
            d_gam = dg.reshape(g, T, B).sum(dim=2).t().contiguous()  # [T, g]
    grads.update(_time_table_grads(ctrl, ts, d_emb, d_gam))
    return grads


_PLANE_BUDGET: dict = {}


def _plane_budget(device, key) -> float:
    """Byte budget of the Bridge paths' planes: SDEH_BRIDGE_PLANE_BYTES, else half of what was free on the device when this shape
    (`key`) first asked -- taken once per shape, so the path that serves a step does not follow the allocator's momentary state."""
    cap = float(os.environ.get("SDEH_BRIDGE_PLANE_BYTES", 0))
    if cap:
        return cap
    k = (torch.device(device).index, key)
    got = _PLANE_BUDGET.get(k)
    if got is None:
        got = _PLANE_BUDGET[k] = 0.5 * torch.cuda.mem_get_info(device)[0]
    return got


class _TrajectoryFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, loss, launch, ts, x, *params):
        x_T, rnd, xs, state = launch(return_traj=True, want_state=True, want_planes=True)
        ctx.loss, ctx.state, ctx.n_params = loss, state, len(params)
        ctx.save_for_backward(ts, xs)
        ctx.mark_non_differentiable(x_T)
        return x_T, rnd

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, _grad_xT, grad_rnd):
        ts, xs = ctx.saved_tensors
        loss, st = ctx.loss, ctx.state
        w = grad_rnd.reshape(-1).contiguous().float()
        keep = E._Keep()
        pr = loss.engine.build_problem(device=xs.device, keep=keep, **st["problem_kwargs"])
        kept = st.get("planes")
        if kept is not None and kept[0] == "fused":
            grads = _fused_backward(loss, pr, keep, ts, xs, w, st, kept[1], kept[2], zrec=kept[4])
        elif kept is not None and kept[0] == "wide":  # wide network on a mixture target: the forward launch's score planes
            planes = _ctrl_backward(loss.engine, pr, keep, ts, xs, w, st, sc_in=kept[1], tscore_in=kept[2])
            grads = _weight_grads(loss.generative_ctrl, ts, xs, *planes)
        else:
            planes = _ctrl_backward(loss.engine, pr, keep, ts, xs, w, st, planes=kept)
            grads = _weight_grads(loss.generative_ctrl, ts, xs, *planes)
        return (None, None, None, None) + tuple(grads.get(id(p)) for p in st["params"])


class _BridgeFn(torch.autograd.Function):
    """Bridge: parameters = the generative network's, then the inference network's."""

    @staticmethod
    def forward(ctx, loss, launch, ts, x, *params):
        x_T, rnd, xs, gp, state = launch(return_traj=True, want_state=True, want_gp=True)
        ctx.loss, ctx.state = loss, state
        ctx.save_for_backward(ts, xs, gp)
        ctx.mark_non_differentiable(x_T)
        return x_T, rnd

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, _grad_xT, grad_rnd):
        ts, xs, gp = ctx.saved_tensors
        loss, st = ctx.loss, ctx.state
        T, B, d = xs.shape[0] - 1, xs.shape[1], xs.shape[2]
        w = grad_rnd.reshape(-1).contiguous().float()
        # The 64-channel Bridge backward keeps 3 (Lh + 1) C floats per (row, coordinate) -- tangent pre-activations, tangent activations
        # and their adjoints (sdeh_bridge_div_backward) -- until the weight-gradient contraction has read them: 115 KB per row at d = 50,
        # i.e. 126 GB for conf/solver/bridge.yaml's B = 16 384, T = 200 (VERDICT r03 missing 5: the allocation failed).  Every term of
        # the backward is a sum over trajectories given w = d loss / d rnd, and the Philox counters are keyed by the GLOBAL row: the
        # batch is processed in slices whose planes fit a budget (half of the free device memory), the gradients of the slices added in
        # order.  One slice = the unsliced computation bit for bit.
        inf_model = st["problem_kwargs"]["inference_ctrl"].base_model
        Cn, Lh = inf_model.channels, len(inf_model.hidden_layer)
        n_slices = 1
        lv = bool(st["problem_kwargs"]["flags"] & L.FLAG_CHANGE_SDE_CTRL)
        if Cn == 64 and d <= 64 and not torch.cuda.is_current_stream_capturing():
            dd = 1 if st.get("div_noise") is not None else d
            per_traj = 4.0 * T * (3 * dd * (Lh + 1) * Cn + 6 * (Lh + 1) * Cn + 6 * d)
            if lv and _bridge_fused_ok(loss.engine, inf_model, d, T, B, st):  # fused: three [64, T B] planes + the generative network's
                per_traj = 4.0 * T * (3 * Cn + 6 * (Lh + 1) * Cn + 8 * d)
            budget = _plane_budget(xs.device, ("bwd", B, T, d))
            n_slices = max(1, -(-int(per_traj * B) // max(int(budget), 1)))
        if n_slices == 1:
            grads = _BridgeFn._backward_rows(loss, st, ts, xs, gp, w)
            return (None, None, None, None) + tuple(grads.get(id(p)) for p in st["params"])
        step = 32 * -(-B // (32 * n_slices))
        total: dict[int, torch.Tensor] = {}
        for b0 in range(0, B, step):
            b1 = min(B, b0 + step)
            sub = dict(st, row_offset=st["row_offset"] + b0)
            for key in ("noise", "div_noise"):
                if st.get(key) is not None:
                    sub[key] = st[key][:, b0:b1].contiguous()
            part = _BridgeFn._backward_rows(loss, sub, ts, xs[:, b0:b1].contiguous(), gp[:, b0:b1].contiguous(), w[b0:b1].contiguous())
            for k, g in part.items():
                if g is not None:
                    total[k] = g if k not in total else total[k] + g
            del part
        return (None, None, None, None) + tuple(total.get(id(p)) for p in st["params"])

    @staticmethod
    def _backward_rows(loss, st, ts, xs, gp, w) -> dict[int, torch.Tensor]:
        """Parameter gradients of both networks from the trajectories `xs [T+1, B, d]` (a slice of the batch or all of it)."""
        dev = xs.device
        T, B, d = xs.shape[0] - 1, xs.shape[1], xs.shape[2]
        N = T * B
        kw = dict(st["problem_kwargs"])
        inf = kw.pop("inference_ctrl")
        lv = bool(kw["flags"] & L.FLAG_CHANGE_SDE_CTRL)
        eng = loss.engine

        sc_u, tscore_u = st.get("score_planes") or (None, None)  # wide Bridge on a mixture target (engine.run)

        def generative(**extra):  # lv: d rnd / d u = dB (the cost's u-derivative vanishes identically); kl: BPTT
            keep = E._Keep()
            pr_u = eng.build_problem(device=dev, keep=keep, **kw)
            return _weight_grads(loss.generative_ctrl, ts, xs,
                                 *_ctrl_backward(eng, pr_u, keep, ts, xs, w, st, sc_in=sc_u, tscore_in=tscore_u, **extra))

        grads = generative() if lv else None
        # inference network, first order: d rnd / d v = (u + v) dt (+ dB with the Ito term); v does not drive the SDE, so this is
        # row-parallel for every method.  kl: its share of d loss / d x_t is collected in `dx` for the generative network's BPTT
        dx = None if lv else torch.empty((T, B, d), device=dev, dtype=torch.float32)
        keep_v = E._Keep()
        v_flags = (kw["flags"] | L.FLAG_CHANGE_SDE_CTRL) & ~(L.FLAG_TERMINAL_TARGET | L.FLAG_INIT_LOGP | L.FLAG_TERMINAL_SECOND)
        kw_v = dict(kw, generative_ctrl=inf, terminal_target=None, second=None, clip_target=None, flags=v_flags)
        pr_v = eng.build_problem(device=dev, keep=keep_v, **kw_v)
        if lv:  # 64 channels, two hidden layers, exact divergence: fused, nothing per coordinate leaves the chip
            inf_grads = _bridge_fused_inference(eng, pr_v, keep_v, inf, ts, xs, gp, w, st)
            if inf_grads is not None:
                grads.update(inf_grads)
                return grads
        zt, dt, dout, dgam, *xt_v = _ctrl_backward(eng, pr_v, keep_v, ts, xs, w, st, gextra=gp, dx_out=dx)
        xt_v = xt_v[0] if xt_v else None
        # inference network, divergence term
        keep_b = E._Keep()
        pr_b = eng.build_problem(device=dev, keep=keep_b, **st["problem_kwargs"])
        Cn, Lh = pr_b.inference.base_model.channels, pr_b.inference.base_model.n_hidden
        g = dgam.shape[0]
        if Cn != 64 or d > 64:  # wide networks: the fused divergence backward (csrc/sdeh_wide_bwd.hip)
            if st.get("div_noise") is not None:
                raise L.SdehUnsupported(-2, "wide-network Bridge: the Hutchinson divergence estimators are built for channels = 64")
            inf_grads = _wide_bridge_inference_grads(eng, pr_b, keep_b, inf, ts, xs, w, zt, dt, dout, dgam, dx, xt_v)
            if not lv:  # generative network: back-propagation through time with the cost on u + v and the inference network's d loss / d x_t
                grads = generative(cost_ctrl=gp, lam_extra=dx)
            grads.update(inf_grads)
            return grads
        eps = st.get("div_noise")
        if eps is not None:
            eps = eps.detach().to(device=dev, dtype=torch.float32).contiguous()
        tz = torch.empty((1 if eps is not None else d, Lh + 1, Cn, N), device=dev, dtype=torch.float32)
        ta, td = torch.empty_like(tz), torch.empty_like(tz)
        d2 = torch.empty((Lh + 1, Cn, N), device=dev, dtype=torch.float32)
        cj = torch.empty((d, N), device=dev, dtype=torch.float32)
        dgam2 = torch.zeros((g, N), device=dev, dtype=torch.float32)
        plan = eng._plan(dev, d, Cn, max(Lh, pr_b.base_model.n_hidden), T, 0)
        with torch.cuda.device(dev):
            L.check(L.load().sdeh_bridge_div_backward(
                plan.handle, C.byref(pr_b), keep_b.ptr(ts.reshape(-1), dev, "ts"), T, xs.data_ptr(), B, w.data_ptr(),
                zt.data_ptr(), tz.data_ptr(), ta.data_ptr(), td.data_ptr(), d2.data_ptr(), cj.data_ptr(), dgam2.data_ptr(),
                None if dx is None else dx.data_ptr(), None if eps is None else eps.data_ptr(),
                torch.cuda.current_stream(dev).cuda_stream))
        if not lv:  # generative network: back-propagation through time with the cost on u + v and the extra d loss / d x_t
            grads = generative(cost_ctrl=gp, lam_extra=dx)
        grads.update(_weight_grads(inf, ts, xs, zt, dt, dout, dgam, extra=dict(d2=d2, td=td, ta=ta, cj=cj, dgam=dgam2, eps=eps)))
        return grads


def _wide_bridge_inference_grads(eng, pr_b, keep_b, inf, ts, xs, w, zt, dt, dout, dgam, dx=None, xt=None) -> dict[int, torch.Tensor]:
    """Gradients of the inference network of a wide Bridge: first-order planes (zt, dt, dout, dgam from sdeh_ctrl_backward_ex) + the
    divergence term through sdeh_bridge_div_backward_wide (adjoint planes d2 of the base pre-activations, the tangent streams' direct
    weight gradients, d / d gamma of the score part)."""
    dev = xs.device
    T, B, d = xs.shape[0] - 1, xs.shape[1], xs.shape[2]
    N = T * B
    base = inf.base_model
    Cn, Lh = base.channels, len(base.hidden_layer)
    if Lh not in (1, 2):  # (the C side refuses as well; say it before anything is allocated)
        raise L.SdehUnsupported(-2, f"wide-network Bridge: the divergence backward is built for inference networks with one or two "
                                    f"hidden layers (num_layers 3 or 4), got {Lh}")
    lib = L.load()
    n_scratch, n_out = C.c_int64(), C.c_int64()
    L.check(lib.sdeh_bridge_div_backward_wide_sizes(d, Cn, Lh, T, B, C.byref(n_scratch), C.byref(n_out)))
    scratch = torch.empty(n_scratch.value, device=dev, dtype=torch.float32)
    out = torch.empty(n_out.value, device=dev, dtype=torch.float32)
    d2 = torch.empty((Lh + 1, Cn, N), device=dev, dtype=torch.float32)
    dgam2 = torch.zeros_like(dgam)
    plan = eng._plan(dev, d, Cn, max(Lh, pr_b.base_model.n_hidden), T, pr_b.target.n_components if pr_b.target.kind == L.DENS_GMM else 0)
    with torch.cuda.device(dev):
        L.check(lib.sdeh_bridge_div_backward_wide(
            plan.handle, C.byref(pr_b), keep_b.ptr(ts.reshape(-1), dev, "ts"), T, xs.data_ptr(), B, w.data_ptr(), zt.data_ptr(),
            d2.data_ptr(), dgam2.data_ptr(), None if dx is None else dx.data_ptr(), scratch.data_ptr(), scratch.numel(),
            out.data_ptr(), torch.cuda.current_stream(dev).cuda_stream))
    grads = _weight_grads(inf, ts, xs, zt, dt, dout, dgam, xt=xt, extra=dict(d2=d2, dgam=dgam2))
    with torch.no_grad():
        g_in = out[:d * Cn].view(d, Cn)
        g_out = out[d * Cn:2 * d * Cn].view(d, Cn)
        g_hid = out[2 * d * Cn:].view(Lh, Cn, Cn)
        grads[id(base.input_embed.weight)] = grads[id(base.input_embed.weight)] + g_in.t()
        grads[id(base.out_layer.weight)] = grads[id(base.out_layer.weight)] + g_out
        grads[id(base.hidden_layer[0].weight)] = grads[id(base.hidden_layer[0].weight)] + g_hid[0]
        if Lh == 2:
            grads[id(base.hidden_layer[1].weight)] = grads[id(base.hidden_layer[1].weight)] + g_hid[1].t()
    return grads


def _leaves(loss, params):
    """The tensors the autograd Function is attached to.  Normally the parameters themselves.  Inside a captured step
    (utils/graphs.py sets `loss._graph_leaves = []`) fresh leaves aliasing the parameters' storage: the real parameters'
    AccumulateGrad nodes may have been left alive by an earlier eager step (a `loss` tensor the caller still holds); they are
    bound to the default stream and abort a capture they take part in.  GraphedTrainStep differentiates w.r.t. the aliases."""
    record = getattr(loss, "_graph_leaves", None)
    if record is None:
        return params
    leaves = [p.detach().requires_grad_() for p in params]
    record.append((params, leaves))
    return leaves


def simulate_with_grad(loss, launch, ts, x):
    """`launch(return_traj, want_state)` runs the HIP forward; returns (x_T, rnd attached to the parameters, None)."""
    params = _ctrl_parameters(loss.generative_ctrl)

    def wrapped(return_traj, want_state, want_planes=False):
        x_T, rnd, xs, state = launch(return_traj=return_traj, want_state=want_state, want_planes=want_planes)
        state["params"] = params
        return x_T, rnd, xs, state

    x_T, rnd = _TrajectoryFn.apply(loss, wrapped, ts, x, *_leaves(loss, params))
    return x_T, rnd, None


def _inference_problem(eng, kw: dict, dev):
    """The inference control of a Bridge as the control of a plain row-parallel problem (it does not drive the SDE): (pr_v, keep_v, inf)."""
    kw = dict(kw)
    inf = kw.pop("inference_ctrl")
    keep_v = E._Keep()
    v_flags = (kw["flags"] | L.FLAG_CHANGE_SDE_CTRL) & ~(L.FLAG_TERMINAL_TARGET | L.FLAG_INIT_LOGP | L.FLAG_TERMINAL_SECOND)
    kw_v = dict(kw, generative_ctrl=inf, terminal_target=None, second=None, clip_target=None, flags=v_flags)
    return eng.build_problem(device=dev, keep=keep_v, **kw_v), keep_v, inf


def _bridge_split_forward(loss, launch, ts, x):
    """(x_T, rnd, xs [T + 1, d, B], u + v [T, d, B], state) of a 64-channel Bridge from the plain launch + the row-parallel inference pass
    (sdeh_simulate_fwd_train2u, sdeh_bridge_inference_fwd), or None when the plain launch was served by a kernel that keeps no planes."""
    x_T, rnd_u, xs_cm, state = launch(return_traj=True, want_state=True, want_planes=True, split=True)
    if xs_cm is None:
        return None
    u_cm = state["planes"][3]
    eng, dev = loss.engine, x.device
    T, d, B = xs_cm.shape[0] - 1, xs_cm.shape[1], xs_cm.shape[2]
    pr_v, keep_v, inf = _inference_problem(eng, state["problem_kwargs"], dev)
    lib = L.load()
    drnd = torch.empty((B, 1), device=dev, dtype=torch.float32)
    gp_cm = torch.empty((T, d, B), device=dev, dtype=torch.float32)
    scratch = torch.empty(lib.sdeh_bridge_inference_fwd_scratch_floats(T, B), device=dev, dtype=torch.float32)
    plan = eng._plan(dev, d, 64, len(inf.base_model.hidden_layer), T, 0)
    noise = state["noise"]
    with torch.cuda.device(dev):
        L.check(lib.sdeh_bridge_inference_fwd(
            plan.handle, C.byref(pr_v), keep_v.ptr(ts.reshape(-1), dev, "ts"), T, xs_cm.data_ptr(), B,
            None if noise is None else keep_v.ptr(noise, dev, "noise"), state["seed"], state["offset"], state["row_offset"],
            u_cm.data_ptr(), drnd.data_ptr(), gp_cm.data_ptr(), scratch.data_ptr(), scratch.numel(),
            torch.cuda.current_stream(dev).cuda_stream))
    return x_T, rnd_u + drnd, xs_cm, gp_cm, state


def simulate_bridge_split(loss, launch, ts, x, inference_ctrl, return_traj: bool):
    """Bridge WITHOUT a graph (evaluation, torch.no_grad()): (x_T, rnd, xs [T + 1, B, d] | None) through the same split, or None where it
    does not apply (then the step-sequential kernel of csrc/sdeh_bridge.hpp)."""
    gen, inf_model = loss.generative_ctrl.base_model, inference_ctrl.base_model
    B, d = x.shape
    if (os.environ.get("SDEH_BRIDGE_SEQ") or gen.channels != 64 or inf_model.channels != 64 or d > 64 or len(inf_model.hidden_layer) != 2
            or 64 * B * 4 >= 2 ** 32):
        return None
    # the split keeps four [T, d, B] planes (trajectory, score, u, the inference terms' gradient) where the step-sequential kernel
    # needs O(d B): only within the memory budget of the training path's planes (evaluation batches go up to 2^24 rows)
    T = ts.numel() - 1
    budget = _plane_budget(x.device, ("fwd", B, T, d))
    if 4.0 * (T + 1) * d * B * 4 > budget:
        return None
    calls = loss.engine.calls
    out = _bridge_split_forward(loss, launch, ts, x)
    if out is None:
        loss.engine.calls = calls  # the fall-back launch draws the same noise
        return None
    x_T, rnd, xs_cm, _, _ = out
    return x_T, rnd, (xs_cm.permute(0, 2, 1).contiguous() if return_traj else None)


class _BridgeSplitFn(torch.autograd.Function):
    """Bridge training, method lv, 64 channels, the exact divergence (conf/solver/bridge.yaml): the SDE is driven by the generative control
    alone (losses/oc.py:176-217), so the forward is the PLAIN launch (sdeh_simulate_fwd_train2u: the wave-specialised kernel, keeping the
    fused backward's planes and u_t) + one row-parallel pass over every (step, trajectory) for the inference control's terms
    (sdeh_bridge_inference_fwd), instead of one wave walking the steps with d tangent passes in its loop; the backward is
    sdeh_ctrl_backward_fused for the generative network and sdeh_bridge_backward_fused for the inference network.  No transposes, no
    per-coordinate planes."""

    @staticmethod
    def forward(ctx, loss, launch, ts, x, *params):
        out = _bridge_split_forward(loss, launch, ts, x)
        if out is None:
            raise RuntimeError("sdeh_simulate_fwd_train2u kept no planes although sdeh_ctrl_backward_fused_supported said it would")
        x_T, rnd, xs_cm, gp_cm, state = out
        ctx.loss, ctx.state, ctx.kept, ctx.zkept = loss, state, state["planes"][1:3], state["planes"][4]
        ctx.save_for_backward(ts, xs_cm, gp_cm)
        ctx.mark_non_differentiable(x_T)
        return x_T, rnd

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, _grad_xT, grad_rnd):
        ts, xs_cm, gp_cm = ctx.saved_tensors
        loss, st = ctx.loss, ctx.state
        sc, tscore = ctx.kept
        eng, dev = loss.engine, xs_cm.device
        w = grad_rnd.reshape(-1).contiguous().float()
        kw = dict(st["problem_kwargs"])
        kw.pop("inference_ctrl")
        keep = E._Keep()
        pr_u = eng.build_problem(device=dev, keep=keep, **kw)
        pr_v, keep_v, inf = _inference_problem(eng, st["problem_kwargs"], dev)
        lv = bool(kw["flags"] & L.FLAG_CHANGE_SDE_CTRL)
        # kl / kl_ito: the inference terms' d loss / d x_t (row-parallel: v does not drive the SDE) joins the generative network's adjoint, whose
        # running cost is taken on u + v
        dx = None if lv else torch.empty_like(gp_cm)
        inf_grads = _bridge_fused_inference(eng, pr_v, keep_v, inf, ts, xs_cm, gp_cm, w, st, cm=True, dx_out=dx)
        if inf_grads is None:
            raise RuntimeError("split Bridge backward: sdeh_bridge_backward_fused refused a problem its forward half accepted")
        grads = _fused_backward(loss, pr_u, keep, ts, xs_cm, w, st, sc, tscore, cost_ctrl=None if lv else gp_cm, lam_extra=dx,
                                zrec=ctx.zkept)
        grads.update(inf_grads)
        return (None, None, None, None) + tuple(grads.get(id(p)) for p in st["params"])


def _bridge_split_ok(loss, ts, x, inference_ctrl, flags: int, div_noise) -> bool:
    """Does the split path serve this Bridge training call?  Exact divergence, both networks of 64 channels, the inference network with
    two hidden layers, the generative problem one the fused backward takes (checked on the plain problem itself by the caller)."""
    if div_noise is not None or os.environ.get("SDEH_BRIDGE_SEQ"):  # (A/B aid: the step-sequential forward)
        return False
    gen, inf_model = loss.generative_ctrl.base_model, inference_ctrl.base_model
    T, (B, d) = ts.numel() - 1, x.shape
    if not (flags & L.FLAG_CHANGE_SDE_CTRL) and len(gen.hidden_layer) != 2:  # kl: the through-time kernels take u + v / d loss / d x_t planes
        return False                                                          # for two hidden layers (template KLB)
    return (gen.channels == 64 and 64 * B * 4 < 2 ** 32
            and _bridge_fused_ok(loss.engine, inf_model, d, T, B, dict(div_noise=div_noise)))


def simulate_bridge_with_grad(loss, launch, ts, x, inference_ctrl, flags: int = 0, div_noise=None, problem_kwargs: dict | None = None):
    """Bridge counterpart: `rnd` is attached to the parameters of both networks."""
    params = _ctrl_parameters(loss.generative_ctrl) + _ctrl_parameters(inference_ctrl)
    if problem_kwargs is not None and _bridge_split_ok(loss, ts, x, inference_ctrl, flags, div_noise):
        eng = loss.engine
        keep = E._Keep()
        pr_u = eng.build_problem(device=x.device, keep=keep, **dict(problem_kwargs, inference_ctrl=None))
        T, d = ts.numel() - 1, x.shape[1]
        k = pr_u.target.n_components if pr_u.target.kind == L.DENS_GMM else 0
        plan = eng._plan(x.device, d, 64, pr_u.base_model.n_hidden, T, k)
        if L.load().sdeh_ctrl_backward_fused_supported(plan.handle, C.byref(pr_u)):
            def wrapped_split(**kwargs):
                x_T, rnd, xs, state = launch(**kwargs)
                state["params"] = params
                return x_T, rnd, xs, state

            x_T, rnd = _BridgeSplitFn.apply(loss, wrapped_split, ts, x, *_leaves(loss, params))
            return x_T, rnd, None

    def wrapped(return_traj, want_state, want_gp):
        x_T, rnd, xs, gp, state = launch(return_traj=return_traj, want_state=want_state, want_gp=want_gp)
        state["params"] = params
        return x_T, rnd, xs, gp, state

    x_T, rnd = _BridgeFn.apply(loss, wrapped, ts, x, *_leaves(loss, params))
    return x_T, rnd, None

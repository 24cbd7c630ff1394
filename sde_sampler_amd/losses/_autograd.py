"""Autograd bridge for training through the fused trajectory kernel (log-variance losses).

`simulate_with_grad(...)` returns `(x_T, rnd, None)` like `simulate()`, with `rnd` attached to the autograd graph of the
control network's parameters.  Forward = the HIP trajectory kernel (keeping the trajectory `xs`); backward =
`sdeh_ctrl_backward` (HIP: per-row re-evaluation + back-propagation on the matrix pipe, noise replayed from the
Philox counters) followed by plain library GEMMs over the N = T*B rows for the weight gradients and by autograd on the
two time-only sub-networks' [T, .] tables.  See include/sdeh.h for why this is exact for method = "lv" / "lv_traj"
(the reference detaches the control that drives the SDE, losses/oc.py:60-70).
"""
from __future__ import annotations

import ctypes as C

import torch

from sde_sampler_amd import _lib as L
from sde_sampler_amd import engine as E


def _ctrl_parameters(ctrl) -> list[torch.nn.Parameter]:
    return [p for p in ctrl.parameters() if p.requires_grad]


class _TrajectoryFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, loss, launch, ts, x, *params):
        x_T, rnd, xs, state = launch(return_traj=True, want_state=True)
        ctx.loss, ctx.state, ctx.n_params = loss, state, len(params)
        ctx.save_for_backward(ts, xs)
        ctx.mark_non_differentiable(x_T)
        return x_T, rnd

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, _grad_xT, grad_rnd):
        ts, xs = ctx.saved_tensors
        loss, st = ctx.loss, ctx.state
        ctrl = loss.generative_ctrl
        base = ctrl.base_model
        dev = xs.device
        T, B, d = xs.shape[0] - 1, xs.shape[1], xs.shape[2]
        N, Cn, Lh = T * B, base.channels, len(base.hidden_layer)
        score_model = getattr(ctrl, "score_model", None)
        g = 0 if score_model is None else score_model.out_layer.out_features
        zt = torch.empty((Lh + 1, Cn, N), device=dev, dtype=torch.float32)
        dt = torch.empty_like(zt)
        dout = torch.empty((d, N), device=dev, dtype=torch.float32)
        dgam = torch.empty((max(g, 1), N), device=dev, dtype=torch.float32)
        w = grad_rnd.reshape(-1).contiguous().float()
        keep = E._Keep()
        pr = loss.engine.build_problem(device=dev, keep=keep, **st["problem_kwargs"])
        plan = loss.engine._plan(dev, d, Cn, Lh, T, pr.target.n_components if pr.target.kind == L.DENS_GMM else 0)
        noise = st["noise"]
        stream = torch.cuda.current_stream(dev).cuda_stream
        with torch.cuda.device(dev):
            L.check(L.load().sdeh_ctrl_backward(
                plan.handle, C.byref(pr), keep.ptr(ts.reshape(-1), dev, "ts"), T, xs.data_ptr(), B,
                None if noise is None else keep.ptr(noise, dev, "noise"), st["seed"], st["offset"], st["row_offset"],
                w.data_ptr(), zt.data_ptr(), dt.data_ptr(), dout.data_ptr(), dgam.data_ptr(), stream))
        act = base.activation
        grads: dict[int, torch.Tensor] = {}
        with torch.no_grad():
            a_prev = None
            X = xs[:T].reshape(N, d)
            grads[id(base.input_embed.weight)] = dt[0] @ X
            d_emb = dt[0].reshape(Cn, T, B).sum(dim=2).t().contiguous()  # [T, C]: gradient of the time embedding table
            grads[id(base.input_embed.bias)] = d_emb.sum(dim=0)
            for k in range(Lh + 1):
                a_k = act(zt[k])
                if k < Lh:
                    lin = base.hidden_layer[k]
                    grads[id(lin.weight)] = dt[k + 1] @ a_k.t()
                    grads[id(lin.bias)] = dt[k + 1].sum(dim=1)
                else:
                    grads[id(base.out_layer.weight)] = dout @ a_k.t()
                    grads[id(base.out_layer.bias)] = dout.sum(dim=1)
        # the two time-only sub-networks: differentiate their [T, .] tables
        with torch.enable_grad():
            te_params = [p for p in base.timestep_embed.parameters() if p.requires_grad]
            if te_params:
                emb = base.timestep_embed(ts[:-1])
                for p, gp in zip(te_params, torch.autograd.grad(emb, te_params, d_emb, allow_unused=True)):
                    grads[id(p)] = gp
            if score_model is not None:
                sm_params = [p for p in score_model.parameters() if p.requires_grad]
                if sm_params:
                    gam = score_model(ts[:-1])
                    clip_model = getattr(ctrl, "clip_model", None)
                    if clip_model is not None:
                        gam = gam.clip(min=-clip_model, max=clip_model)
                    d_gam = dgam[:g].reshape(g, T, B).sum(dim=2).t().contiguous()  # [T, g]
                    for p, gp in zip(sm_params, torch.autograd.grad(gam, sm_params, d_gam, allow_unused=True)):
                        grads[id(p)] = gp
        out = tuple(grads.get(id(p)) for p in st["params"])
        return (None, None, None, None) + out


def simulate_with_grad(loss, launch, ts, x):
    """`launch(return_traj, want_state)` runs the HIP forward; returns (x_T, rnd attached to the parameters, None)."""
    params = _ctrl_parameters(loss.generative_ctrl)
    holder = {}

    def wrapped(return_traj, want_state):
        x_T, rnd, xs, state = launch(return_traj=return_traj, want_state=want_state)
        state["params"] = params
        holder["state"] = state
        return x_T, rnd, xs, state

    x_T, rnd = _TrajectoryFn.apply(loss, wrapped, ts, x, *params)
    return x_T, rnd, None

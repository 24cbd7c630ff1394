"""Parameter containers of the control network (host side).

`TimeEmbed` and `FourierMLP` keep the attribute / state_dict layout of the reference's sde_sampler/models/mlp.py
(TimeEmbed 43-82, FourierMLP 85-122): `input_embed`, `timestep_embed.{timestep_coeff,timestep_phase,hidden_layer,
out_layer}`, `hidden_layer`, `out_layer`, `activation`, `channels` -- the HIP engine introspects exactly these
(SURVEY.md 8b), so checkpoints and optimizer param groups carry over.  Their `forward` is the plain PyTorch
definition of the function the kernel evaluates; the loss classes never call it (they go through libsdeh).
"""
from __future__ import annotations

from typing import Callable

import torch
from torch import nn


class Model(nn.Module):
    def __init__(self, dim: int, dim_out: int | None = None):
        super().__init__()
        self.dim = dim
        self.dim_in = dim + 1
        self.dim_out = dim_out or dim

    @staticmethod
    def init_linear(layer: nn.Linear, bias_init: Callable | None = None, weight_init: Callable | None = None):
        if bias_init:
            bias_init(layer.bias)
        if weight_init:
            weight_init(layer.weight)


class TimeEmbed(Model):
    """t -> MLP([sin(c t + phi), cos(c t + phi)]) with fixed frequencies c = linspace(0.1, 100, C)."""

    def __init__(self, dim_out: int, activation: Callable, num_layers: int = 2, channels: int = 64,
                 last_bias_init: Callable | None = None, last_weight_init: Callable | None = None):
        super().__init__(dim=1, dim_out=dim_out)
        self.channels = channels
        self.activation = activation
        self.register_buffer("timestep_coeff", torch.linspace(start=0.1, end=100, steps=channels).unsqueeze(0),
                             persistent=False)
        self.timestep_phase = nn.Parameter(torch.randn(1, channels))
        widths = [2 * channels] + [channels] * (num_layers - 1)
        self.hidden_layer = nn.ModuleList([nn.Linear(a, channels) for a in widths[:-1]])
        self.out_layer = nn.Linear(channels, self.dim_out)
        Model.init_linear(self.out_layer, bias_init=last_bias_init, weight_init=last_weight_init)

    def forward(self, t: torch.Tensor, *args) -> torch.Tensor:
        assert t.ndim in (0, 1, 2)
        t = t.view(-1, 1).float()
        arg = self.timestep_coeff * t + self.timestep_phase
        h = torch.cat([arg.sin(), arg.cos()], dim=1)
        for layer in self.hidden_layer:
            h = self.activation(layer(h))
        return self.out_layer(h)


class FourierMLP(Model):
    """x, t -> out_layer(act(... hidden(act(input_embed(x) + timestep_embed(t)))))."""

    def __init__(self, dim: int, activation: Callable, num_layers: int = 4, channels: int = 64,
                 last_bias_init: Callable | None = None, last_weight_init: Callable | None = None, **kwargs):
        super().__init__(dim=dim, **kwargs)
        self.channels = channels
        self.activation = activation
        self.input_embed = nn.Linear(self.dim, channels)
        self.timestep_embed = TimeEmbed(dim_out=channels, activation=activation, num_layers=2, channels=channels)
        self.hidden_layer = nn.ModuleList([nn.Linear(channels, channels) for _ in range(num_layers - 2)])
        self.out_layer = nn.Linear(channels, self.dim_out)
        Model.init_linear(self.out_layer, bias_init=last_bias_init, weight_init=last_weight_init)

    def forward(self, t: torch.Tensor, x: torch.Tensor) -> torch.Tensor:
        # a scalar time: the embedding is one row, broadcast over the batch; per-row times are embedded row by row
        t = t.reshape(-1, 1).float()
        if t.shape[0] not in (1, x.shape[0]):
            raise ValueError(f"t has {t.shape[0]} entries for a batch of {x.shape[0]}")
        h = self.input_embed(x) + self.timestep_embed(t)
        for layer in self.hidden_layer:
            h = layer(self.activation(h))
        return self.out_layer(self.activation(h))

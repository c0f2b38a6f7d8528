"""`Activation1d(fused=True)` / `FusedAntiAliasActivation` with the reference's interface
(/root/reference/indextts/BigVGAN/alias_free_activation/cuda/activation1d.py:13-76), running the HIP kernel."""
from __future__ import annotations

import torch
import torch.nn as nn

from indextts.BigVGAN.alias_free_activation.cuda import load
from itts_hip.pack import _filter12

_op = None


def _get_op():
    global _op
    if _op is None:
        _op = load.load()
    return _op


class FusedAntiAliasActivation(torch.autograd.Function):
    """Filter size 12, replication padding, log-scale alpha/beta (as the reference kernel hard-codes)."""

    @staticmethod
    def forward(ctx, inputs, up_ftr, down_ftr, alpha, beta):
        return _get_op().forward(inputs, up_ftr, down_ftr, alpha, beta)

    @staticmethod
    def backward(ctx, output_grads):
        raise NotImplementedError


class Activation1d(nn.Module):
    def __init__(self, activation, up_ratio: int = 2, down_ratio: int = 2, up_kernel_size: int = 12,
                 down_kernel_size: int = 12, fused: bool = True):
        super().__init__()
        if (up_ratio, down_ratio, up_kernel_size, down_kernel_size) != (2, 2, 12, 12) or not fused:
            raise NotImplementedError("the HIP kernel implements the ratio-2 / 12-tap configuration only")
        self.act = activation
        f = torch.from_numpy(_filter12())
        self.register_buffer("up_filter", f.clone())
        self.register_buffer("down_filter", f.clone())

    def forward(self, x):
        name = self.act.__class__.__name__
        if name not in ("Snake", "SnakeBeta"):
            raise NotImplementedError(f"fused activation for {name}")
        alpha = self.act.alpha.data
        beta = self.act.beta.data if name == "SnakeBeta" else self.act.alpha.data
        if not getattr(self.act, "alpha_logscale", False):  # the kernel takes log-scale parameters (:60-71)
            alpha, beta = torch.log(alpha), torch.log(beta)
        return FusedAntiAliasActivation.apply(x, self.up_filter, self.down_filter, alpha, beta)

"""autograd wrapper of the skinny dense-layer kernels (csrc/linear.hip).

`latent_linear(x, weight, bias)` computes what `nn.Linear.forward` / `F.linear` computes for the
two latent FCs of the autoencoder (reference models.py:130,144); the parameters stay ordinary
`nn.Linear` parameters, so `state_dict` names and shapes are unchanged.
"""
from __future__ import annotations

import torch

from . import ops


class _LatentLinear(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias):
        x = x.contiguous()
        ctx.save_for_backward(x, weight)
        ctx.has_bias = bias is not None
        return ops.linear_fwd(x, weight.contiguous(), bias)

    @staticmethod
    def backward(ctx, dy):
        x, weight = ctx.saved_tensors
        dy = dy.contiguous()
        dx = ops.linear_bwd_data(dy, weight) if ctx.needs_input_grad[0] else None
        dW = db = None
        if ctx.needs_input_grad[1] or (ctx.has_bias and ctx.needs_input_grad[2]):
            dW, db = ops.linear_bwd_wgt(dy, x, want_bias=ctx.has_bias and ctx.needs_input_grad[2])
        return dx, dW, db


def latent_linear(x: torch.Tensor, weight: torch.Tensor, bias) -> torch.Tensor:
    if x.dim() != 2:
        raise RuntimeError("latent_linear expects a 2-D input, got %s" % (tuple(x.shape),))
    return _LatentLinear.apply(x, weight, bias)

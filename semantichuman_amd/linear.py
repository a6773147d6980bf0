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
        from . import _lib
        x = x.contiguous()
        ctx.save_for_backward(x, weight)
        ctx.has_bias = bias is not None
        ctx.mma = _lib.get_f32_mma_mode()          # the node's arithmetic form: backward (autograd's thread, later) keeps it
        return ops.linear_fwd(x, weight.contiguous(), bias, ctx.mma)

    @staticmethod
    def backward(ctx, dy):
        x, weight = ctx.saved_tensors
        dy = dy.contiguous()
        dx = ops.linear_bwd_data(dy, weight, ctx.mma) if ctx.needs_input_grad[0] else None
        dW = db = None
        if ctx.needs_input_grad[1] or (ctx.has_bias and ctx.needs_input_grad[2]):
            want_bias = ctx.has_bias and ctx.needs_input_grad[2]
            param = _fused_param(weight, dy, x) if ctx.needs_input_grad[1] else None
            if param is not None:
                # the optimizer asked for this parameter's update to be applied by the kernel that computes its gradient
                # (optim.Adam.fuse_linear_weight_gradients): no dW is materialised, `weight.grad` stays None
                db = _fused_apply(param, dy, x, want_bias, ctx.mma)
            else:
                dW, db = ops.linear_bwd_wgt(dy, x, want_bias=want_bias, mma=ctx.mma)
        return dx, dW, db


# weight.data_ptr() -> (weakref to the parameter, weakref to the optimizer that owns its update, index of its param group):
# registered by semantichuman_amd.optim.Adam.fuse_linear_weight_gradients, consulted by the backward pass above.  Both references
# are weak: an entry whose parameter or optimizer is gone is dropped at the next lookup and the layer is back on the ordinary
# gradient + step() path.  With an entry, backward ITSELF updates the weight (see that method's docstring for what follows).
_FUSED_UPDATE = {}


def _fused_param(weight, dy, x):
    """The registered parameter behind `weight` if this backward may apply its update in the weight-gradient kernel, else None."""
    key = weight.data_ptr()
    fused = _FUSED_UPDATE.get(key)
    if fused is None:
        return None
    param, opt = fused[0](), fused[1]()
    if param is None or opt is None:                   # the parameter or its optimizer died: a stale entry
        _FUSED_UPDATE.pop(key, None)
        return None
    if param.data_ptr() == key and param.shape == weight.shape and param.grad is None and weight.is_contiguous() and \
            ops.linear_bwd_wgt_adam_ok(dy.shape[0], dy.shape[1], x.shape[1]):
        return param
    return None


def _fused_apply(param, dy, x, want_bias, mma):
    _, opt_ref, gi = _FUSED_UPDATE[param.data_ptr()]
    opt = opt_ref()
    return opt._fused_update(opt.param_groups[gi], param, dy, x, want_bias, mma)


def latent_linear(x: torch.Tensor, weight: torch.Tensor, bias) -> torch.Tensor:
    if x.dim() != 2:
        raise RuntimeError("latent_linear expects a 2-D input, got %s" % (tuple(x.shape),))
    return _LatentLinear.apply(x, weight, bias)


class _LatentLinearBF16(torch.autograd.Function):
    """The same layer on the bf16 path: bf16 working copy of the weight (semantichuman_amd.shadow), x / y bf16 or fp32,
    fp32 accumulation, fp32 master-weight gradients."""

    @staticmethod
    def forward(ctx, x, weight, bias, out_dtype):
        from . import shadow
        x = x.contiguous()
        w16 = shadow.get(weight)
        ctx.save_for_backward(x, weight)
        ctx.has_bias = bias is not None
        return ops.linear_fwd_bf16(x, w16, bias, out_dtype)

    @staticmethod
    def backward(ctx, dy):
        from . import shadow
        x, weight = ctx.saved_tensors
        dy = dy.contiguous()
        dx = ops.linear_bwd_data_bf16(dy, shadow.get(weight), x.dtype) if ctx.needs_input_grad[0] else None
        dW = db = None
        if ctx.needs_input_grad[1] or (ctx.has_bias and ctx.needs_input_grad[2]):
            want_bias = ctx.has_bias and ctx.needs_input_grad[2]
            param = _fused_param(weight, dy, x) if ctx.needs_input_grad[1] else None
            if param is not None:                          # (see _LatentLinear.backward) - the bf16 working copy is rewritten too
                db = _fused_apply(param, dy, x, want_bias, None)
            else:
                dW, db = ops.linear_bwd_wgt_bf16(dy, x, want_bias=want_bias)
        return dx, dW, db, None


def latent_linear_bf16(x: torch.Tensor, weight: torch.Tensor, bias, out_dtype) -> torch.Tensor:
    if x.dim() != 2:
        raise RuntimeError("latent_linear_bf16 expects a 2-D input, got %s" % (tuple(x.shape),))
    return _LatentLinearBF16.apply(x, weight, bias, out_dtype)


class _GroupedLinear(torch.autograd.Function):
    """All groups of one family of per-part layers in one launch per direction (csrc/grouped_linear.hip).
    forward(x, x_off, y_cols, y_off, n_groups, w_0..w_{G-1}, b_0..b_{G-1})"""

    @staticmethod
    def forward(ctx, x, x_off, y_cols, y_off, G, *params):
        weights, biases = list(params[:G]), list(params[G:])
        x = x.contiguous()
        weights_c = [w.contiguous() for w in weights]
        ctx.save_for_backward(x, *weights_c)
        ctx.meta = (tuple(x_off), y_cols, tuple(y_off), G, [b is not None for b in biases])
        return ops.grouped_linear_fwd(x, x_off, weights_c, biases if any(b is not None for b in biases) else None, y_cols, y_off)

    @staticmethod
    def backward(ctx, dy):
        x, *weights = ctx.saved_tensors
        x_off, y_cols, y_off, G, has_bias = ctx.meta
        dy = dy.contiguous()
        dx = ops.grouped_linear_bwd_data(dy, y_off, weights, x.shape[1], x_off) if ctx.needs_input_grad[0] else None
        need_w = [ctx.needs_input_grad[5 + g] for g in range(G)]
        need_b = [has_bias[g] and ctx.needs_input_grad[5 + G + g] for g in range(G)]
        dWs, dbs = [None] * G, [None] * G
        if any(need_w) or any(need_b):
            dWs, dbs = ops.grouped_linear_bwd_wgt(dy, y_off, x, x_off, weights, need_b)
        return (dx, None, None, None, None) + tuple(dWs) + tuple(dbs)


def grouped_linear(x: torch.Tensor, x_off, modules, y_off=None):
    """Apply `modules[g]` (nn.Linear) to the column block x[:, x_off[g] : x_off[g] + in_features_g] for every g and
    return the outputs side by side ([M, sum out_features], or at the given column offsets `y_off`).  What the
    reference does with one `Linear` call per part (models.py:236,252,269)."""
    if x.dim() != 2:
        raise RuntimeError("grouped_linear expects a 2-D input, got %s" % (tuple(x.shape),))
    outs = [m.out_features for m in modules]
    if y_off is None:
        y_off, o = [], 0
        for n in outs:
            y_off.append(o); o += n
    y_cols = max(o + n for o, n in zip(y_off, outs))
    G = len(modules)
    return _GroupedLinear.apply(x, list(x_off), y_cols, list(y_off), G, *[m.weight for m in modules], *[m.bias for m in modules])

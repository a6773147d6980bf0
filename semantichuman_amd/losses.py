"""Training losses and the evaluation metric as HIP kernels with autograd wrappers.

  l1_loss(a, b)              = F.l1_loss(a, b)          reference train_funcs.py:501, main.py:296
  edge_ratio_loss(...)       = the `edgereg` term        reference train_funcs.py:12-39, 503-508
                               (the reference loops over samples on the host with a
                                .cpu().numpy() sync each; here one batched kernel)
  vertex_l2_mm / eval_l1     = the evaluation metric     reference test_funcs.py:41-49
"""
from __future__ import annotations

import numpy as np
import torch

from . import ops


class _L1Loss(torch.autograd.Function):
    @staticmethod
    def forward(ctx, a, b):
        a, b = a.contiguous(), b.contiguous()
        ctx.save_for_backward(a, b)
        return ops.l1_loss_fwd(a, b)

    @staticmethod
    def backward(ctx, g):
        a, b = ctx.saved_tensors
        gb = ops.l1_loss_bwd(a, b, g.contiguous())
        ga = -gb if ctx.needs_input_grad[0] else None
        return ga, (gb if ctx.needs_input_grad[1] else None)


def l1_loss(a: torch.Tensor, b: torch.Tensor) -> torch.Tensor:
    """mean |a-b| over every element (dummy row included, as train_funcs.py:501 does)."""
    if a.shape != b.shape:
        raise RuntimeError("l1_loss: shape mismatch %s vs %s" % (tuple(a.shape), tuple(b.shape)))
    return _L1Loss.apply(a, b)


class FaceTables:
    """Device copies of the face list plus the vertex->corner lists that make the edge-loss
    gradient a gather (no atomics).  Built once per template mesh."""

    def __init__(self, faces, n_rows: int, device):
        f = np.ascontiguousarray(np.asarray(faces, dtype=np.int32))
        corners = np.arange(f.size, dtype=np.int32)            # corner id = 3*face + k
        order = np.argsort(f.ravel(), kind="stable")
        vptr = np.zeros(n_rows + 1, dtype=np.int32)
        np.cumsum(np.bincount(f.ravel(), minlength=n_rows), out=vptr[1:])
        self.n_faces = f.shape[0]
        self.faces = torch.from_numpy(f).to(device)
        self.vptr = torch.from_numpy(vptr).to(device)
        self.vcorner = torch.from_numpy(corners[order]).to(device)
        # per vertex and in the same corner order: the two other vertices of the corner's face (the fused loss gradient
        # gathers straight from this list)
        cs = corners[order]
        ff, kk = cs // 3, cs % 3
        nbr = np.stack([f[ff, (kk + 1) % 3], f[ff, (kk + 2) % 3]], axis=1).astype(np.int32)
        self.vnbr = torch.from_numpy(np.ascontiguousarray(nbr.reshape(-1))).to(device)


class _EdgeRatioLoss(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x_hat, x, ft: FaceTables):
        x_hat, x = x_hat.contiguous(), x.contiguous()
        ctx.save_for_backward(x_hat, x)
        ctx.ft = ft
        return ops.edge_ratio_loss_fwd(x_hat, x, ft.faces)

    @staticmethod
    def backward(ctx, g):
        x_hat, x = ctx.saved_tensors
        ft = ctx.ft
        return ops.edge_ratio_loss_bwd(x_hat, x, ft.faces, ft.vptr, ft.vcorner, g.contiguous()), None, None


def edge_ratio_loss(x_hat, x, ft: FaceTables):
    """mean over batch and faces of sum_{3 edges} | |e_rec| / (|e_gt| + 1e-5) - 1 |.
    x (the ground truth) is treated as a constant, as in the reference (numpy target)."""
    return _EdgeRatioLoss.apply(x_hat, x.detach(), ft)


class _ReconLoss(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x_hat, x, ft: FaceTables, edge_w: float):
        x_hat, x = x_hat.contiguous(), x.contiguous()
        ctx.save_for_backward(x_hat, x)
        ctx.ft, ctx.edge_w = ft, float(edge_w)
        ctx.set_materialize_grads(False)             # no zeros for the (non-differentiable) parts output
        total, parts = ops.recon_loss_fwd(x_hat, x, ft.faces, edge_w)
        ctx.mark_non_differentiable(parts)
        return total, parts

    @staticmethod
    def backward(ctx, g, _unused):
        x_hat, x = ctx.saved_tensors
        ft = ctx.ft
        return ops.recon_loss_bwd(x_hat, x, ft.n_faces, ft.vptr, ft.vnbr, ctx.edge_w, g.contiguous()), None, None, None


def recon_loss(x_hat, x, ft: FaceTables, edge_w: float):
    """The loss of the plain training loop in one piece: `l1_loss(x, x_hat) + edge_w * edge_ratio_loss(x_hat, x, ft)`
    (train_funcs.py:501-508) with the same arithmetic, but three kernel launches instead of eleven.
    Returns (total, parts) where parts = float32 [2] = (l1, edge), detached, for logging."""
    return _ReconLoss.apply(x_hat, x.detach(), ft, edge_w)


def vertex_l2_mm(x_hat, x, dummy_node: bool = True, mm_constant: float = 1000.0):
    """mean per-vertex Euclidean error in mm, dummy row dropped (test_funcs.py:42-49)."""
    n_real = x.shape[1] - 1 if dummy_node else x.shape[1]
    return ops.vertex_l2(x_hat.detach().contiguous(), x.detach().contiguous(), n_real, mm_constant)


def eval_l1(x_hat, x, dummy_node: bool = True):
    """mean |x_hat - x| with the dummy row dropped (test_funcs.py:46).  The dummy row of both
    tensors is zero, so this is the all-row mean rescaled by (N+1)/N."""
    full = ops.l1_loss_fwd(x_hat.detach().contiguous(), x.detach().contiguous())
    if not dummy_node:
        return full
    n1 = x.shape[1]
    tail = (x_hat.detach()[:, -1] - x.detach()[:, -1]).abs().sum()      # exact even if the dummy rows differ
    return (full * (x.numel()) - tail) / float(x.shape[0] * (n1 - 1) * x.shape[2])

"""Typed Python wrappers around the C ABI (one function per entry point of sh_kernels.h).

Tensors are fp32, contiguous, on a HIP device.  A 3-D activation has one of two layouts:
    'bm'  batch-major  [B, rows, C]   - the reference layout (models.py:37)
    'vm'  vertex-major [rows, B, C]   - the internal fast layout (a gathered neighbour is one
                                        contiguous B*C block -> coalesced 16-byte reads)
The kernels take explicit strides, so both layouts go through the same code.
"""
from __future__ import annotations

import torch

from . import _lib
from ._lib import ACT_IDS, check, ptr, stream_ptr


def _dims(t: torch.Tensor, layout: str, dtypes=(torch.float32,)):
    """-> (B, rows, C, stride_row, stride_batch) for a contiguous 3-D tensor."""
    if t.dim() != 3 or not t.is_contiguous() or t.dtype not in dtypes:
        raise ValueError("expected a contiguous %s 3-D tensor, got %s %s" % ("/".join(str(d) for d in dtypes), tuple(t.shape), t.dtype))
    if not t.is_cuda:
        raise RuntimeError("semantichuman_amd kernels need a HIP device tensor (got %s); there is no CPU path" % t.device)
    if layout == "bm":
        B, R, C = t.shape
        return B, R, C, C, R * C
    if layout == "vm":
        R, B, C = t.shape
        return B, R, C, B * C, C
    raise ValueError("layout must be 'bm' or 'vm'")


def _check_index_range(t):
    if t.numel() >= 2 ** 32:
        raise RuntimeError("semantichuman_amd: gathered tensors are addressed with 32-bit element offsets; "
                           "%d elements is too large - split the batch" % t.numel())


def alloc(B: int, rows: int, C: int, layout: str, device, extra_rows: int = 0) -> torch.Tensor:
    if layout == "bm":
        if extra_rows:
            raise ValueError("extra rows need the vertex-major layout")
        return torch.empty((B, rows, C), dtype=torch.float32, device=device)
    return torch.empty((rows + extra_rows, B, C), dtype=torch.float32, device=device)


def act_id(name: str) -> int:
    if name not in ACT_IDS:
        raise NotImplementedError(name)          # same error type as reference models.py:31-32
    return ACT_IDS[name]


def spiral_conv_fwd(x, x_layout, table, weight, bias, y, y_layout, R, S, act, zero_row):
    B, _, Cin, xsv, xsb = _dims(x, x_layout)
    B2, Ry, Cout, ysv, ysb = _dims(y, y_layout)
    assert B == B2 and Ry >= R and weight.shape == (Cout, S * Cin) and table.dtype == torch.int32
    _check_index_range(x)
    check(_lib.load().sh_spiral_conv_fwd(ptr(x), xsv, xsb, ptr(table), ptr(weight), ptr(bias), ptr(y), ysv, ysb,
                                         B, R, S, Cin, Cout, act, zero_row, _lib.mma_id(), stream_ptr()), "sh_spiral_conv_fwd")


def spiral_conv_bwd_data(dpre, dp_layout, table_t, weight_t, dx, dx_layout, yprev, yp_layout, act_prev, zero_row,
                         n_in, S, Cin, Cout):
    B, _, C1, dsv, dsb = _dims(dpre, dp_layout)
    B2, Rx, C2, xsv, xsb = _dims(dx, dx_layout)
    assert B == B2 and C1 == Cout and C2 == Cin and Rx >= n_in and weight_t.shape == (Cin, S * Cout)
    assert table_t.dtype == torch.int32 and tuple(table_t.shape) == (n_in, S)
    _check_index_range(dpre)
    if yprev is not None:
        _, _, C3, ysv, ysb = _dims(yprev, yp_layout)
        assert C3 == Cin
    else:
        ysv = ysb = 0
    check(_lib.load().sh_spiral_conv_bwd_data(ptr(dpre), dsv, dsb, ptr(table_t), ptr(weight_t), ptr(dx), xsv, xsb,
                                              ptr(yprev), ysv, ysb, act_prev, zero_row, B, n_in, S, Cin, Cout,
                                              _lib.mma_id(), stream_ptr()), "sh_spiral_conv_bwd_data")


def weight_transpose(weight, S, Cin, Cout):
    wt = torch.empty((Cin, S * Cout), dtype=torch.float32, device=weight.device)
    check(_lib.load().sh_weight_transpose(ptr(weight), ptr(wt), S, Cin, Cout, stream_ptr()), "sh_weight_transpose")
    return wt


def spiral_conv_bwd_wgt(dpre, dp_layout, x, x_layout, table, R, S, Cin, Cout, want_bias=True):
    B, _, C1, dsv, dsb = _dims(dpre, dp_layout)
    B2, _, C2, xsv, xsb = _dims(x, x_layout)
    assert B == B2 and C1 == Cout and C2 == Cin
    lib = _lib.load()
    nbytes = lib.sh_spiral_conv_bwd_wgt_workspace(B, R, S, Cin, Cout)
    ws = torch.empty((nbytes + 3) // 4, dtype=torch.float32, device=x.device)
    dW = torch.empty((Cout, S * Cin), dtype=torch.float32, device=x.device)
    db = torch.empty((Cout,), dtype=torch.float32, device=x.device) if want_bias else None
    check(lib.sh_spiral_conv_bwd_wgt(ptr(dpre), dsv, dsb, ptr(x), xsv, xsb, ptr(table), ptr(dW), ptr(db), ptr(ws),
                                     nbytes, B, R, S, Cin, Cout, _lib.mma_id(), stream_ptr()), "sh_spiral_conv_bwd_wgt")
    return dW, db


def spiral_conv_bwd_wgt_deferred(dpre, dp_layout, x, x_layout, table, R, S, Cin, Cout, want_bias=True):
    """Weight-gradient pass that only writes its partial slabs; returns a job for
    `spiral_conv_bwd_wgt_reduce` (which reduces the slabs of a whole stack in one launch)."""
    B, _, C1, dsv, dsb = _dims(dpre, dp_layout)
    B2, _, C2, xsv, xsb = _dims(x, x_layout)
    assert B == B2 and C1 == Cout and C2 == Cin
    lib = _lib.load()
    nbytes = lib.sh_spiral_conv_bwd_wgt_workspace(B, R, S, Cin, Cout)
    ws = torch.empty((nbytes + 3) // 4, dtype=torch.float32, device=x.device)
    check(lib.sh_spiral_conv_bwd_wgt(ptr(dpre), dsv, dsb, ptr(x), xsv, xsb, ptr(table), ptr(None), ptr(None), ptr(ws),
                                     nbytes, B, R, S, Cin, Cout, _lib.mma_id(), stream_ptr()), "sh_spiral_conv_bwd_wgt")
    dW = torch.empty((Cout, S * Cin), dtype=torch.float32, device=x.device)
    db = torch.empty((Cout,), dtype=torch.float32, device=x.device) if want_bias else None
    return dict(ws=ws, dW=dW, db=db, dims=(B, R, S, Cin, Cout))


def _c_ptr_array(tensors):
    import ctypes
    return (ctypes.c_void_p * len(tensors))(*[0 if t is None else t.data_ptr() for t in tensors])


def _c_int_array(vals):
    import ctypes
    return (ctypes.c_int * len(vals))(*vals)


def spiral_conv_bwd_wgt_reduce(jobs):
    """One launch: fixed-order reduction of the partial slabs of every job (<= 16)."""
    if not jobs:
        return
    import ctypes
    cols = list(zip(*[j["dims"] for j in jobs]))
    args = [_c_ptr_array([j["ws"] for j in jobs]), _c_ptr_array([j["dW"] for j in jobs]), _c_ptr_array([j["db"] for j in jobs])]
    args += [_c_int_array(c) for c in cols]
    check(_lib.load().sh_spiral_conv_bwd_wgt_reduce_multi(len(jobs), *[ctypes.cast(a, ctypes.c_void_p) for a in args], stream_ptr()),
          "sh_spiral_conv_bwd_wgt_reduce_multi")


def weight_transpose_multi(weights, dims):
    """dims: [(S, Cin, Cout)] -> list of weight_t tensors, one launch."""
    import ctypes
    outs = [torch.empty((ci, s * co), dtype=torch.float32, device=w.device) for w, (s, ci, co) in zip(weights, dims)]
    cols = list(zip(*dims))
    args = [_c_ptr_array(list(weights)), _c_ptr_array(outs)] + [_c_int_array(c) for c in cols]
    check(_lib.load().sh_weight_transpose_multi(len(outs), *[ctypes.cast(a, ctypes.c_void_p) for a in args], stream_ptr()),
          "sh_weight_transpose_multi")
    return outs


def act_backward(dy, dy_layout, y, y_layout, dpre, dp_layout, R, act, zero_row):
    B, _, C, asv, asb = _dims(dy, dy_layout)
    _, _, _, ysv, ysb = _dims(y, y_layout)
    _, _, _, psv, psb = _dims(dpre, dp_layout)
    check(_lib.load().sh_act_backward(ptr(dy), asv, asb, ptr(y), ysv, ysb, ptr(dpre), psv, psb, B, R, C, act, zero_row,
                                      stream_ptr()), "sh_act_backward")


def spmm(csr_dev, x, x_layout, y, y_layout, rows, yprev=None, yp_layout="vm", act_prev=0, zero_row=-1):
    """csr_dev = (rowptr, col, val) device tensors."""
    B, _, C, xsv, xsb = _dims(x, x_layout)
    B2, Ry, C2, ysv, ysb = _dims(y, y_layout)
    assert B == B2 and C == C2 and Ry >= rows
    if yprev is not None:
        _, _, _, psv, psb = _dims(yprev, yp_layout)
    else:
        psv = psb = 0
    rowptr, col, val = csr_dev
    check(_lib.load().sh_spmm(ptr(rowptr), ptr(col), ptr(val), ptr(x), xsv, xsb, ptr(y), ysv, ysb, ptr(yprev), psv, psb,
                              act_prev, zero_row, B, rows, C, stream_ptr()), "sh_spmm")


def _linear_ws(M, N, K, device):
    nbytes = _lib.load().sh_linear_workspace(M, N, K)
    return torch.empty(max(1, (nbytes + 3) // 4), dtype=torch.float32, device=device), nbytes


def _check2d(*ts):
    for t in ts:
        if t is None:
            continue
        if not (t.is_cuda and t.dtype == torch.float32 and t.is_contiguous()):
            raise RuntimeError("semantichuman_amd linear kernels need contiguous fp32 HIP tensors (got %s %s %s); there is "
                               "no CPU path" % (t.device, t.dtype, tuple(t.shape)))


def linear_fwd(x, weight, bias, mma=None):
    """y = x @ weight.T + bias for x [M,K], weight [N,K] (nn.Linear layout).  mma: arithmetic form of the products (None = the
    caller's default, _lib.get_f32_mma_mode())."""
    _check2d(x, weight, bias)
    M, K = x.shape
    N = weight.shape[0]
    assert weight.shape[1] == K
    y = torch.empty((M, N), dtype=torch.float32, device=x.device)
    ws, nb = _linear_ws(M, N, K, x.device)
    check(_lib.load().sh_linear_fwd(ptr(x), ptr(weight), ptr(bias), ptr(y), M, N, K, ptr(ws), nb, _lib.mma_id(mma), stream_ptr()), "sh_linear_fwd")
    return y


def linear_bwd_data(dy, weight, mma=None):
    _check2d(dy, weight)
    M, N = dy.shape
    K = weight.shape[1]
    dx = torch.empty((M, K), dtype=torch.float32, device=dy.device)
    ws, nb = _linear_ws(M, N, K, dy.device)
    check(_lib.load().sh_linear_bwd_data(ptr(dy), ptr(weight), ptr(dx), M, N, K, ptr(ws), nb, _lib.mma_id(mma), stream_ptr()), "sh_linear_bwd_data")
    return dx


def linear_bwd_wgt(dy, x, want_bias=True, mma=None):
    _check2d(dy, x)
    M, N = dy.shape
    K = x.shape[1]
    dW = torch.empty((N, K), dtype=torch.float32, device=dy.device)
    db = torch.empty((N,), dtype=torch.float32, device=dy.device) if want_bias else None
    ws, nb = _linear_ws(M, N, K, dy.device)
    check(_lib.load().sh_linear_bwd_wgt(ptr(dy), ptr(x), ptr(dW), ptr(db), M, N, K, ptr(ws), nb, _lib.mma_id(mma), stream_ptr()), "sh_linear_bwd_wgt")
    return dW, db


def linear_bwd_wgt_adam_ok(M, N, K):
    return bool(_lib.load().sh_linear_bwd_wgt_adam_ok(int(M), int(N), int(K)))


def linear_bwd_wgt_adam(dy, x, weight, exp_avg, exp_avg_sq, step, lr, betas, eps, weight_decay, want_bias=True, mma=None, weight_bf16=None):
    """dW = dy^T x used as the gradient of Adam's update of `weight` / `exp_avg` / `exp_avg_sq`, in place, in the kernel that
    computes it (sh_linear_bwd_wgt_adam); returns the bias gradient (or None).  `step` is not advanced (sh_adam_step with a
    zero-length entry, or sh_adam_bump).  dy / x: fp32 or bf16; `weight_bf16`: the bf16 working copy to rewrite, or None."""
    for t in (dy, x):
        if not (t.is_cuda and t.dim() == 2 and t.is_contiguous() and t.dtype in (torch.float32, torch.bfloat16)):
            raise RuntimeError("linear_bwd_wgt_adam: dy / x must be contiguous 2-D fp32 or bf16 HIP tensors")
    M, N = dy.shape
    K = x.shape[1]
    for t in (weight, exp_avg, exp_avg_sq):
        if not (t.is_cuda and t.dtype == torch.float32 and t.is_contiguous() and tuple(t.shape) == (N, K)):
            raise RuntimeError("linear_bwd_wgt_adam: weight / exp_avg / exp_avg_sq must be contiguous fp32 [%d, %d] HIP tensors" % (N, K))
    if weight_bf16 is not None and not (weight_bf16.dtype == torch.bfloat16 and weight_bf16.is_contiguous() and tuple(weight_bf16.shape) == (N, K)):
        raise RuntimeError("linear_bwd_wgt_adam: the bf16 working copy must be a contiguous bf16 [%d, %d] tensor" % (N, K))
    db = torch.empty((N,), dtype=torch.float32, device=dy.device) if want_bias else None
    did = lambda t: 1 if t.dtype == torch.bfloat16 else 0               # noqa: E731 - enum sh_dtype
    check(_lib.load().sh_linear_bwd_wgt_adam(ptr(dy), did(dy), ptr(x), did(x), ptr(weight), ptr(weight_bf16), ptr(exp_avg), ptr(exp_avg_sq),
                                             ptr(step), ptr(lr), float(betas[0]), float(betas[1]), float(eps), float(weight_decay), ptr(db),
                                             M, N, K, _lib.mma_id(mma), stream_ptr()), "sh_linear_bwd_wgt_adam")
    return db


def _glin_args(x_off, y_off, N, K):
    import ctypes
    G = len(N)
    arr64 = lambda v: (ctypes.c_int64 * G)(*[int(t) for t in v])          # noqa: E731
    arr32 = lambda v: (ctypes.c_int * G)(*[int(t) for t in v])            # noqa: E731
    return G, arr64(x_off), arr64(y_off), arr32(N), arr32(K)


def _ptr_array(tensors):
    import ctypes
    return (ctypes.c_void_p * len(tensors))(*[0 if t is None else t.data_ptr() for t in tensors])


def grouped_linear_fwd(x, x_off, weights, biases, y_cols, y_off):
    """y[:, y_off[g]:y_off[g]+N_g] = x[:, x_off[g]:x_off[g]+K_g] @ W_g.T + b_g for every group, one launch.
    x [M, Kx] contiguous fp32; weights: list of [N_g, K_g]; biases: list (entries may be None) or None."""
    _check2d(x, *weights)
    M = x.shape[0]
    y = torch.empty((M, y_cols), dtype=torch.float32, device=x.device)
    G, xo, yo, N, K = _glin_args(x_off, y_off, [w.shape[0] for w in weights], [w.shape[1] for w in weights])
    check(_lib.load().sh_grouped_linear_fwd(G, ptr(x), x.shape[1], xo, _ptr_array(weights),
                                            _ptr_array(biases) if biases is not None else None, ptr(y), y_cols, yo, M, N, K,
                                            stream_ptr()), "sh_grouped_linear_fwd")
    return y


def grouped_linear_bwd_data(dy, y_off, weights, x_cols, x_off):
    _check2d(dy, *weights)
    M = dy.shape[0]
    dx = torch.zeros((M, x_cols), dtype=torch.float32, device=dy.device)      # columns no group covers stay zero
    G, xo, yo, N, K = _glin_args(x_off, y_off, [w.shape[0] for w in weights], [w.shape[1] for w in weights])
    check(_lib.load().sh_grouped_linear_bwd_data(G, ptr(dy), dy.shape[1], yo, _ptr_array(weights), ptr(dx), x_cols, xo, M, N, K,
                                                 stream_ptr()), "sh_grouped_linear_bwd_data")
    return dx


def grouped_linear_bwd_wgt(dy, y_off, x, x_off, weights, want_bias):
    _check2d(dy, x)
    M = dy.shape[0]
    dWs = [torch.empty_like(w) for w in weights]
    dbs = [torch.empty((w.shape[0],), dtype=torch.float32, device=w.device) if wb else None for w, wb in zip(weights, want_bias)]
    G, xo, yo, N, K = _glin_args(x_off, y_off, [w.shape[0] for w in weights], [w.shape[1] for w in weights])
    check(_lib.load().sh_grouped_linear_bwd_wgt(G, ptr(dy), dy.shape[1], yo, ptr(x), x.shape[1], xo, _ptr_array(dWs),
                                                _ptr_array(dbs), M, N, K, stream_ptr()), "sh_grouped_linear_bwd_wgt")
    return dWs, dbs


def _ws(device):
    return torch.empty(_lib.load().sh_reduce_workspace() // 4, dtype=torch.float32, device=device)


def l1_loss_fwd(a, b):
    assert a.shape == b.shape and a.is_contiguous() and b.is_contiguous() and a.is_cuda
    out = torch.empty((), dtype=torch.float32, device=a.device)
    check(_lib.load().sh_l1_loss_fwd(ptr(a), ptr(b), a.numel(), ptr(out), ptr(_ws(a.device)), stream_ptr()), "sh_l1_loss_fwd")
    return out


def l1_loss_bwd(a, b, gscale):
    g = torch.empty_like(b)
    check(_lib.load().sh_l1_loss_bwd(ptr(a), ptr(b), a.numel(), ptr(gscale), ptr(g), stream_ptr()), "sh_l1_loss_bwd")
    return g


def vertex_l2(a, b, n_real, scale=1000.0):
    assert a.shape == b.shape and a.dim() == 3 and a.shape[2] == 3 and a.is_contiguous() and b.is_contiguous()
    out = torch.empty((), dtype=torch.float32, device=a.device)
    check(_lib.load().sh_vertex_l2(ptr(a), ptr(b), a.shape[0], a.shape[1], n_real, scale, ptr(out), ptr(_ws(a.device)),
                                   stream_ptr()), "sh_vertex_l2")
    return out


def edge_ratio_loss_fwd(x_hat, x, faces):
    assert x_hat.shape == x.shape and x.shape[2] == 3 and x_hat.is_contiguous() and x.is_contiguous()
    out = torch.empty((), dtype=torch.float32, device=x.device)
    check(_lib.load().sh_edge_ratio_loss_fwd(ptr(x_hat), ptr(x), ptr(faces), x.shape[0], x.shape[1], faces.shape[0], ptr(out),
                                             ptr(_ws(x.device)), stream_ptr()), "sh_edge_ratio_loss_fwd")
    return out


def recon_loss_fwd(x_hat, x, faces, edge_w):
    """-> (total float32 [], parts float32 [2] = (l1, edge)) for contiguous [B, N1, 3] tensors; two separate allocations,
    so that neither is a view autograd would have to copy through."""
    assert x_hat.shape == x.shape and x.shape[2] == 3 and x_hat.is_contiguous() and x.is_contiguous() and x.is_cuda
    total = torch.empty((), dtype=torch.float32, device=x.device)
    parts = torch.empty((2,), dtype=torch.float32, device=x.device)
    ws = torch.empty(_lib.load().sh_recon_loss_workspace() // 4, dtype=torch.float32, device=x.device)
    check(_lib.load().sh_recon_loss_fwd(ptr(x_hat), ptr(x), ptr(faces), x.shape[0], x.shape[1], faces.shape[0], float(edge_w),
                                        ptr(total), ptr(parts), ptr(ws), stream_ptr()), "sh_recon_loss_fwd")
    return total, parts


def recon_loss_bwd(x_hat, x, n_faces, vptr, vnbr, edge_w, gscale):
    g = torch.empty_like(x_hat)
    check(_lib.load().sh_recon_loss_bwd(ptr(x_hat), ptr(x), ptr(vptr), ptr(vnbr), x.shape[0], x.shape[1], int(n_faces),
                                        float(edge_w), ptr(gscale), ptr(g), stream_ptr()), "sh_recon_loss_bwd")
    return g


def edge_ratio_loss_bwd(x_hat, x, faces, vptr, vcorner, gscale):
    g = torch.empty_like(x_hat)
    check(_lib.load().sh_edge_ratio_loss_bwd(ptr(x_hat), ptr(x), ptr(faces), ptr(vptr), ptr(vcorner), x.shape[0], x.shape[1],
                                             faces.shape[0], ptr(gscale), ptr(g), stream_ptr()), "sh_edge_ratio_loss_bwd")
    return g


def measure_girth(v, rings):
    """v [B, rows, 3] fp32 (rows may include the dummy row); rings = (ptr, a, b, f) device tensors -> girth [B, P]."""
    ptr_, a, b, f = rings
    if not (v.is_cuda and v.dtype == torch.float32 and v.dim() == 3 and v.shape[2] == 3 and v.stride(2) == 1
            and v.stride(1) == 3):
        raise RuntimeError("semantichuman_amd.measure_girth needs fp32 HIP vertices [B, rows, 3] (got %s %s %s); there is "
                           "no CPU path" % (v.device, v.dtype, tuple(v.shape)))
    B, P = v.shape[0], ptr_.numel() - 1
    out = torch.empty((B, P), dtype=torch.float32, device=v.device)
    check(_lib.load().sh_measure_girth(ptr(v), v.stride(0) if B > 1 else v.shape[1] * 3, ptr(ptr_), ptr(a), ptr(b), ptr(f),
                                       B, P, ptr(out), stream_ptr()), "sh_measure_girth")
    return out


def bone_length(kps, bones):
    """kps [B, K, 3] fp32 contiguous; bones int32 [P, 3] (third id -1 for a two-joint bone) -> length [B, P]."""
    if not (kps.is_cuda and kps.dtype == torch.float32 and kps.dim() == 3 and kps.shape[2] == 3 and kps.is_contiguous()):
        raise RuntimeError("semantichuman_amd.bone_length needs contiguous fp32 HIP joints [B, K, 3]; there is no CPU path")
    B, K, P = kps.shape[0], kps.shape[1], bones.shape[0]
    out = torch.empty((B, P), dtype=torch.float32, device=kps.device)
    check(_lib.load().sh_bone_length(ptr(kps), ptr(bones), B, K, P, ptr(out), stream_ptr()), "sh_bone_length")
    return out


NORM_FLAGS = {"zeromean": 1, "zeroroot": 2, "onelength": 4, "small": 8, "gass": 16, "normal": 32}


def dataset_normalize(raw, flags, dummy_rows=1, j_root=None, mean=None, std=None, center=None, scale=None):
    """raw [n, N, 3] fp32 HIP tensor -> normalised, dummy-padded [n, N + dummy_rows, 3]."""
    if not (raw.is_cuda and raw.dtype == torch.float32 and raw.dim() == 3 and raw.shape[2] == 3 and raw.is_contiguous()):
        raise RuntimeError("semantichuman_amd.dataset_normalize needs a contiguous fp32 HIP tensor [n, N, 3]; there is no CPU path")
    n, N = raw.shape[0], raw.shape[1]
    for t, shape in ((j_root, (N,)), (mean, (N, 3)), (std, (N, 3)), (center, (n, 3)), (scale, (n, 3))):
        if t is not None and not (t.is_cuda and t.dtype == torch.float32 and t.is_contiguous() and tuple(t.shape) == shape):
            raise RuntimeError("dataset_normalize: table must be a contiguous fp32 HIP tensor of shape %s, got %s" % (shape, tuple(t.shape)))
    out = torch.empty((n, N + dummy_rows, 3), dtype=torch.float32, device=raw.device)
    check(_lib.load().sh_dataset_normalize(ptr(raw), ptr(out), n, N, dummy_rows, flags, ptr(j_root), ptr(mean), ptr(std),
                                           ptr(center), ptr(scale), stream_ptr()), "sh_dataset_normalize")
    return out


def gather_meshes(src, idx):
    """src [n, ...] contiguous fp32 HIP tensor, idx int64 HIP tensor [b] -> src[idx] (bit-exact copy)."""
    if not (src.is_cuda and src.dtype == torch.float32 and src.is_contiguous() and idx.is_cuda and idx.dtype == torch.int64
            and idx.dim() == 1 and idx.is_contiguous()):
        raise RuntimeError("semantichuman_amd.gather_meshes needs a contiguous fp32 HIP tensor and an int64 HIP index; no CPU path")
    b = idx.numel()
    out = torch.empty((b,) + tuple(src.shape[1:]), dtype=torch.float32, device=src.device)
    if b:
        check(_lib.load().sh_gather_meshes(ptr(src), src[0].numel(), ptr(idx), b, ptr(out), stream_ptr()), "sh_gather_meshes")
    return out


# ------------------------------------------------------------------------------------------ bf16 compute path
_ANY = (torch.float32, torch.bfloat16)


def dtype_id(t: torch.Tensor) -> int:
    return _lib.DTYPE_IDS[str(t.dtype).replace("torch.", "")]


def conv_wfrag_prep(weights, dims, transpose):
    """fp32 master weights [Cout, S*Cin] -> bf16 fragment-ordered working copies (uint8 buffers), one launch.
    dims: [(S, Cin, Cout)]; transpose: [0 forward operand | 1 backward-data operand]."""
    import ctypes
    lib = _lib.load()
    outs = []
    for w, (s, ci, co), tr in zip(weights, dims, transpose):
        if not (w.is_cuda and w.dtype == torch.float32 and w.is_contiguous()):
            raise RuntimeError("conv_wfrag_prep needs contiguous fp32 HIP weights; there is no CPU path")
        nb = lib.sh_conv_wfrag_bytes(s, co if tr else ci, ci if tr else co)
        outs.append(torch.empty(nb, dtype=torch.uint8, device=w.device))
    cols = list(zip(*dims))
    args = [_c_ptr_array(list(weights)), _c_ptr_array(outs)] + [_c_int_array(c) for c in cols] + [_c_int_array([int(t) for t in transpose])]
    check(lib.sh_conv_wfrag_prep_multi(len(outs), *[ctypes.cast(a, ctypes.c_void_p) for a in args], stream_ptr()),
          "sh_conv_wfrag_prep_multi")
    return outs


def spiral_conv_fwd_bf16(x, x_layout, table, wfrag, bias, y, y_layout, R, S, Cin, Cout, act, zero_row):
    B, _, C1, xsv, xsb = _dims(x, x_layout, _ANY)
    B2, Ry, C2, ysv, ysb = _dims(y, y_layout, _ANY)
    assert B == B2 and Ry >= R and C1 == Cin and C2 == Cout and table.dtype == torch.int32
    _check_index_range(x)
    check(_lib.load().sh_spiral_conv_fwd_bf16(ptr(x), dtype_id(x), xsv, xsb, ptr(table), ptr(wfrag), ptr(bias), ptr(y), dtype_id(y),
                                              ysv, ysb, B, R, S, Cin, Cout, act, zero_row, stream_ptr()), "sh_spiral_conv_fwd_bf16")


def spiral_conv_bwd_data_bf16(dpre, dp_layout, table_t, wfrag_t, dx, dx_layout, yprev, yp_layout, act_prev, zero_row, n_in, S, Cin, Cout):
    B, _, C1, dsv, dsb = _dims(dpre, dp_layout, _ANY)
    B2, Rx, C2, xsv, xsb = _dims(dx, dx_layout, _ANY)
    assert B == B2 and C1 == Cout and C2 == Cin and Rx >= n_in and tuple(table_t.shape) == (n_in, S)
    _check_index_range(dpre)
    if yprev is not None:
        _, _, C3, ysv, ysb = _dims(yprev, yp_layout, (torch.bfloat16,))
        assert C3 == Cin
    else:
        ysv = ysb = 0
    check(_lib.load().sh_spiral_conv_bwd_data_bf16(ptr(dpre), dtype_id(dpre), dsv, dsb, ptr(table_t), ptr(wfrag_t), ptr(dx),
                                                   dtype_id(dx), xsv, xsb, ptr(yprev), ysv, ysb, act_prev, zero_row, B, n_in, S,
                                                   Cin, Cout, stream_ptr()), "sh_spiral_conv_bwd_data_bf16")


def spiral_conv_bf16_rag_ok(B, S, Cg, Nout, rag_L) -> bool:
    return bool(_lib.load().sh_spiral_conv_bf16_rag_ok(B, S, Cg, Nout, rag_L))


def spiral_conv_bwd_data_bf16_rag(dpre, dp_layout, rag_rows, rag_pos, wfrag_t, dx, dx_layout, yprev, yp_layout, act_prev, zero_row, n_in, S,
                                  Cin, Cout):
    """Backward-data over ragged source lists (mesh_ops.transpose_table_ragged): bf16 on both sides, no pre-summed rows."""
    B, _, C1, dsv, dsb = _dims(dpre, dp_layout, (torch.bfloat16,))
    B2, Rx, C2, xsv, xsb = _dims(dx, dx_layout, (torch.bfloat16,))
    assert B == B2 and C1 == Cout and C2 == Cin and Rx >= n_in and rag_rows.shape == rag_pos.shape and rag_rows.shape[0] == n_in
    assert rag_rows.dtype == torch.int32 and rag_pos.dtype == torch.int32 and rag_rows.is_contiguous() and rag_pos.is_contiguous()
    _check_index_range(dpre)
    if yprev is not None:
        _, _, C3, ysv, ysb = _dims(yprev, yp_layout, (torch.bfloat16,))
        assert C3 == Cin
    else:
        ysv = ysb = 0
    check(_lib.load().sh_spiral_conv_bwd_data_bf16_rag(ptr(dpre), dsv, dsb, ptr(rag_rows), ptr(rag_pos), int(rag_rows.shape[1]), ptr(wfrag_t),
                                                       ptr(dx), xsv, xsb, ptr(yprev), ysv, ysb, act_prev, zero_row, B, n_in, S, Cin, Cout,
                                                       stream_ptr()), "sh_spiral_conv_bwd_data_bf16_rag")


def spiral_conv_bwd_wgt_bf16(dpre, dp_layout, x, x_layout, table, R, S, Cin, Cout, want_bias=True):
    """-> (dW fp32 [Cout, S*Cin], dbias fp32 [Cout] or None)"""
    import ctypes
    B, _, C1, dsv, dsb = _dims(dpre, dp_layout, _ANY)
    B2, _, C2, xsv, xsb = _dims(x, x_layout, _ANY)
    assert B == B2 and C1 == Cout and C2 == Cin
    lib = _lib.load()
    nbytes = lib.sh_spiral_conv_bwd_wgt_workspace_bf16(B, R, S, Cin, Cout)
    ws = torch.empty((nbytes + 3) // 4, dtype=torch.float32, device=x.device)
    check(lib.sh_spiral_conv_bwd_wgt_bf16(ptr(dpre), dtype_id(dpre), dsv, dsb, ptr(x), dtype_id(x), xsv, xsb, ptr(table), ptr(ws), nbytes,
                                          B, R, S, Cin, Cout, stream_ptr()), "sh_spiral_conv_bwd_wgt_bf16")
    dW = torch.empty((Cout, S * Cin), dtype=torch.float32, device=x.device)
    db = torch.empty((Cout,), dtype=torch.float32, device=x.device) if want_bias else None
    args = [_c_ptr_array([ws]), _c_ptr_array([dW]), _c_ptr_array([db])] + [_c_int_array([v]) for v in (B, R, S, Cin, Cout)]
    check(lib.sh_spiral_conv_bwd_wgt_reduce_multi_bf16(1, *[ctypes.cast(a, ctypes.c_void_p) for a in args], stream_ptr()),
          "sh_spiral_conv_bwd_wgt_reduce_multi_bf16")
    return dW, db


def wgrad_thin_ok(B, n_in, S, Cin, Cout, dtype) -> bool:
    return bool(_lib.load().sh_spiral_conv_bwd_wgt_thin_ok(B, n_in, S, Cin, Cout, _lib.DTYPE_IDS[str(dtype).replace("torch.", "")]))


def _thin_dx_args(dx, weight, B, Cin, act_prev, zero_prev):
    if dx is None:
        return [None, None, 0, 0, None, 0, -1]
    _, _, Cd, gsv, gsb = _dims(dx, "vm", _ANY)
    assert Cd == Cin and weight is not None
    return [ptr(weight), ptr(dx), gsv, gsb, None, int(act_prev), int(zero_prev)]


def spiral_conv_bwd_wgt_thin_deferred(dpre_ext, x, table_t, R, S, Cin, Cout, want_bias=True, weight=None, dx=None, act_prev=0,
                                      zero_prev=-1):
    """fp32 path: slabs only; the job goes to `spiral_conv_bwd_wgt_reduce` with the other layers of the stack.
    dx (vertex-major [>= R, B, Cin]) given: the launch also writes the layer's input gradient (see sh_kernels.h)."""
    B, _, C1, dsv, dsb = _dims(dpre_ext, "vm")
    B2, rows_x, C2, xsv, xsb = _dims(x, "vm")
    assert B == B2 and C1 == Cout and C2 == Cin and rows_x >= R
    lib = _lib.load()
    nbytes = lib.sh_spiral_conv_bwd_wgt_workspace(B, R, S, Cin, Cout)
    ws = torch.empty((nbytes + 3) // 4, dtype=torch.float32, device=x.device)
    check(lib.sh_spiral_conv_bwd_wgt_thin(ptr(dpre_ext), dsv, dsb, ptr(x), dtype_id(x), xsv, xsb, ptr(table_t), ptr(ws), nbytes,
                                          *_thin_dx_args(dx, weight, B, Cin, act_prev, zero_prev), B, R, R, S, Cin, Cout, dtype_id(x),
                                          stream_ptr()), "sh_spiral_conv_bwd_wgt_thin")
    dW = torch.empty((Cout, S * Cin), dtype=torch.float32, device=x.device)
    db = torch.empty((Cout,), dtype=torch.float32, device=x.device) if want_bias else None
    return dict(ws=ws, dW=dW, db=db, dims=(B, R, S, Cin, Cout))


def spiral_conv_bwd_wgt_thin(dpre_ext, x, table_t, R, S, Cin, Cout, want_bias=True, weight=None, dx=None, act_prev=0, zero_prev=-1):
    """Role-swapped weight gradient of a 16 -> 3 channel layer (wgrad_thin.hip): dpre_ext fp32 [rows >= R (+ pre-summed extra
    rows), B, 3] and x [R, B, 16] (fp32 or bf16: selects the path), both vertex-major; table_t int32 [R, S] the transposed
    table backward-data uses.  -> (dW fp32 [Cout, S*Cin], dbias fp32 [Cout] or None)."""
    import ctypes
    B, _, C1, dsv, dsb = _dims(dpre_ext, "vm", _ANY)
    B2, rows_x, C2, xsv, xsb = _dims(x, "vm", _ANY)
    assert B == B2 and C1 == Cout and C2 == Cin and rows_x >= R and dpre_ext.dtype == torch.float32
    lib = _lib.load()
    b16 = x.dtype == torch.bfloat16
    if not lib.sh_spiral_conv_bwd_wgt_thin_ok(B, R, S, Cin, Cout, dtype_id(x)):
        raise RuntimeError("spiral_conv_bwd_wgt_thin: shape not covered (B %d R %d S %d Cin %d Cout %d)" % (B, R, S, Cin, Cout))
    nbytes = (lib.sh_spiral_conv_bwd_wgt_workspace_bf16 if b16 else lib.sh_spiral_conv_bwd_wgt_workspace)(B, R, S, Cin, Cout)
    ws = torch.empty((nbytes + 3) // 4, dtype=torch.float32, device=x.device)
    check(lib.sh_spiral_conv_bwd_wgt_thin(ptr(dpre_ext), dsv, dsb, ptr(x), dtype_id(x), xsv, xsb, ptr(table_t), ptr(ws), nbytes,
                                          *_thin_dx_args(dx, weight, B, Cin, act_prev, zero_prev), B, R, R, S, Cin, Cout, dtype_id(x),
                                          stream_ptr()), "sh_spiral_conv_bwd_wgt_thin")
    dW = torch.empty((Cout, S * Cin), dtype=torch.float32, device=x.device)
    db = torch.empty((Cout,), dtype=torch.float32, device=x.device) if want_bias else None
    args = [_c_ptr_array([ws]), _c_ptr_array([dW]), _c_ptr_array([db])] + [_c_int_array([v]) for v in (B, R, S, Cin, Cout)]
    red = lib.sh_spiral_conv_bwd_wgt_reduce_multi_bf16 if b16 else lib.sh_spiral_conv_bwd_wgt_reduce_multi
    check(red(1, *[ctypes.cast(a, ctypes.c_void_p) for a in args], stream_ptr()), "sh_spiral_conv_bwd_wgt_reduce_multi")
    return dW, db


def cast_bf16(src, out=None):
    """fp32 HIP tensor -> bf16 copy (one streaming kernel); `out`: an existing bf16 tensor of the same shape to refresh
    in place (its address may be baked into a captured hipGraph)."""
    if not (src.is_cuda and src.dtype == torch.float32 and src.is_contiguous()):
        raise RuntimeError("cast_bf16 needs a contiguous fp32 HIP tensor; there is no CPU path")
    if out is not None and not (out.is_cuda and out.dtype == torch.bfloat16 and out.is_contiguous() and out.shape == src.shape
                                and out.device == src.device):
        raise RuntimeError("cast_bf16: `out` must be a contiguous bf16 HIP tensor of the source's shape and device")
    dst = out if out is not None else torch.empty(src.shape, dtype=torch.bfloat16, device=src.device)
    if src.numel():
        check(_lib.load().sh_cast_f32_to_bf16(ptr(src), ptr(dst), src.numel(), stream_ptr()), "sh_cast_f32_to_bf16")
    return dst


def _check2d_any(*ts):
    for t in ts:
        if t is not None and not (t.is_cuda and t.dtype in _ANY and t.is_contiguous()):
            raise RuntimeError("semantichuman_amd bf16 linear kernels need contiguous bf16/fp32 HIP tensors (got %s %s); there is no "
                               "CPU path" % (t.device, t.dtype))


def _linear_ws_bf16(M, N, K, device):
    nbytes = _lib.load().sh_linear_workspace_bf16(M, N, K)
    return torch.empty(max(4, (nbytes + 3) // 4), dtype=torch.float32, device=device), nbytes


def linear_fwd_bf16(x, w_bf16, bias, out_dtype):
    _check2d_any(x)
    M, K = x.shape
    N = w_bf16.shape[0]
    assert w_bf16.dtype == torch.bfloat16 and w_bf16.shape[1] == K and w_bf16.is_contiguous()
    y = torch.empty((M, N), dtype=out_dtype, device=x.device)
    ws, nb = _linear_ws_bf16(M, N, K, x.device)
    check(_lib.load().sh_linear_fwd_bf16(ptr(x), dtype_id(x), ptr(w_bf16), ptr(bias), ptr(y), dtype_id(y), M, N, K, ptr(ws), nb,
                                         stream_ptr()), "sh_linear_fwd_bf16")
    return y


def linear_bwd_data_bf16(dy, w_bf16, out_dtype):
    _check2d_any(dy)
    M, N = dy.shape
    K = w_bf16.shape[1]
    dx = torch.empty((M, K), dtype=out_dtype, device=dy.device)
    ws, nb = _linear_ws_bf16(M, N, K, dy.device)
    check(_lib.load().sh_linear_bwd_data_bf16(ptr(dy), dtype_id(dy), ptr(w_bf16), ptr(dx), dtype_id(dx), M, N, K, ptr(ws), nb,
                                              stream_ptr()), "sh_linear_bwd_data_bf16")
    return dx


def linear_bwd_wgt_bf16(dy, x, want_bias=True):
    _check2d_any(dy, x)
    M, N = dy.shape
    K = x.shape[1]
    dW = torch.empty((N, K), dtype=torch.float32, device=dy.device)
    db = torch.empty((N,), dtype=torch.float32, device=dy.device) if want_bias else None
    check(_lib.load().sh_linear_bwd_wgt_bf16(ptr(dy), dtype_id(dy), ptr(x), dtype_id(x), ptr(dW), ptr(db), M, N, K, stream_ptr()),
          "sh_linear_bwd_wgt_bf16")
    return dW, db

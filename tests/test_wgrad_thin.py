"""Role-swapped weight gradient of the 16 -> 3 channel layer (csrc/wgrad_thin.hip; the decoder's last conv, reference
models.py:146-153) against the float64 statement of the ordinary form

    dW[co, s, ci] = sum_{v,b} dpre[v,b,co] * x[table[v,s],b,ci],      db[co] = sum_{v,b} dpre[v,b,co]

on the fp32 and the bf16 path, and against the general kernels it replaces inside the stack sequencers."""
import numpy as np
import pytest
import torch

from semantichuman_amd import mesh_ops, ops
from tests import emulate

pytestmark = pytest.mark.gpu


def dev():
    return torch.device("cuda:0")


def bf(t):
    return t.to(torch.bfloat16).to(torch.float32)


def rand_table(R, S, seed):
    g = np.random.RandomState(seed)
    t = g.randint(0, R, size=(R, S)).astype(np.int32)
    t[:, 0] = np.arange(R)
    t[g.rand(R, S) < 0.1] = R - 1                    # padding entries -> the dummy row
    t[R - 1] = R - 1
    return t


# (B, R, S): batch blocks 1 / 2 / 3, odd and even spiral lengths (the self chunk lands in either half of a chunk pair),
# vertex counts that leave waves without work and ragged last ranges
# spirals of 11..30 (fp32 path only: two or three launches over shares of the positions; BASELINE config 4 forces 18)
LONG = [(64, 400, 18), (32, 130, 12), (64, 260, 20), (32, 77, 11), (48, 150, 22), (32, 90, 30), (16, 60, 21)]
# batches of 16 (mod 32): a vertex's last item is a half item (the semantic loop's three passes of 16 meshes run as one batch of 48)
HALF = [(48, 301, 10), (16, 97, 9), (80, 55, 7), (48, 1200, 10), (16, 20, 1)]


@pytest.mark.parametrize("B,R,S", [(64, 301, 10), (32, 97, 9), (96, 55, 7), (64, 1200, 10), (32, 20, 1), (64, 700, 4)] + LONG + HALF)
@pytest.mark.parametrize("path", ["f32", "bf16"])
def test_thin_wgrad_matches_the_ordinary_form(B, R, S, path):
    if path == "bf16" and S > 10:
        pytest.skip("the bf16 path's thin kernel takes spirals of at most 10")
    torch.manual_seed(5)
    table = rand_table(R, S, 6)
    tt = mesh_ops.transpose_table_dense(table, R, none_row=R - 1, skip_row=-1)
    x = torch.randn(R, B, 16)
    dpre = torch.randn(R, B, 3)
    dpre[-1] = 0                                     # the zero row every producer of dpre keeps ("no source" entries point here)
    ext = torch.from_numpy(emulate.extend_dpre(dpre.double().numpy(), tt)).float()
    if path == "bf16":
        x = bf(x)
        # the kernel rounds the (pre-summed) gradient rows to bf16 on load; the reference rounds the same rows
        dW_ref, _ = emulate.conv_bwd_wgt_swapped(bf(ext).double().numpy(), x.double().numpy(), tt.table_t)
        tol = 2e-6 * np.sqrt(R * B) + 1e-6
    else:
        dW_ref, _ = emulate.conv_bwd_wgt(dpre.double().numpy(), x.double().numpy(), table)
        tol = 2e-6 * np.sqrt(R * B) + 1e-6
    db_ref = dpre.double().numpy().reshape(-1, 3).sum(0)                 # the bias sum stays fp32 on both paths
    xd = x.to(dev(), torch.bfloat16 if path == "bf16" else torch.float32)
    td = torch.from_numpy(tt.table_t).to(dev())
    dW, db = ops.spiral_conv_bwd_wgt_thin(ext.to(dev()), xd, td, R, S, 16, 3)
    assert float(np.abs(dW.cpu().double().numpy() - dW_ref).max()) <= tol * float(np.abs(dW_ref).max()) + 1e-5
    assert float(np.abs(db.cpu().double().numpy() - db_ref).max()) <= tol * float(np.abs(db_ref).max()) + 1e-4
    dW2, db2 = ops.spiral_conv_bwd_wgt_thin(ext.to(dev()), xd, td, R, S, 16, 3)
    assert torch.equal(dW, dW2) and torch.equal(db, db2)                 # fixed-order sums: bitwise reproducible
    # ... and the general kernel of the same path agrees to accumulation noise
    tf = torch.from_numpy(table).to(dev())
    if path == "bf16":
        dWg, dbg = ops.spiral_conv_bwd_wgt_bf16(dpre.to(dev()), "vm", xd, "vm", tf, R, S, 16, 3)
        gtol = 2.0 ** -8 * np.sqrt(S)                                    # rounding a pre-summed row once vs each of its terms
    else:
        dWg, dbg = ops.spiral_conv_bwd_wgt(dpre.to(dev()), "vm", xd, "vm", tf, R, S, 16, 3)
        gtol = tol
    scale = float(dWg.abs().max())
    assert float((dW - dWg).abs().max()) <= gtol * scale + 1e-5
    # (the general bf16 kernel sums bf16-ROUNDED gradients for the bias; this one sums them in fp32)
    assert float((db - dbg).abs().max()) <= (2.0 ** -7 * np.sqrt(R * B) if path == "bf16" else tol * float(dbg.abs().max())) + 1e-4


@pytest.mark.parametrize("B,R,S", [(64, 301, 10), (32, 97, 9), (96, 55, 7)] + LONG + HALF)
@pytest.mark.parametrize("path", ["f32", "bf16"])
@pytest.mark.parametrize("act", ["elu", "identity"])
def test_thin_launch_also_gives_the_input_gradient(B, R, S, path, act):
    """dx of the same launch = backward-data over the transposed table times act'(x), dummy row forced to zero; the
    weight gradient of that launch is the one of the launch without dx, bit for bit."""
    if path == "bf16" and S > 10:
        pytest.skip("the bf16 path's thin kernel takes spirals of at most 10")
    torch.manual_seed(7)
    table = rand_table(R, S, 8)
    tt = mesh_ops.transpose_table_dense(table, R, none_row=R - 1, skip_row=-1)
    x = torch.randn(R, B, 16)
    x[-1] = 0
    W = torch.randn(3, S * 16) / np.sqrt(S * 16)
    dpre = torch.randn(R, B, 3)
    dpre[-1] = 0
    ext = torch.from_numpy(emulate.extend_dpre(dpre.double().numpy(), tt)).float()
    a = ops.act_id(act)
    if path == "bf16":
        x = bf(x)
        ref = emulate.conv_bwd_data(bf(ext).double().numpy(), tt.table_t, bf(W).double().numpy(), 16)
        tol = 2.0 ** -8
    else:
        ref = emulate.conv_bwd_data(ext.double().numpy(), tt.table_t, W.double().numpy(), 16)
        tol = 2e-6 * np.sqrt(3 * S)
    ref = ref * emulate.DACT[a](x.double().numpy())
    ref[R - 1] = 0
    dt = torch.bfloat16 if path == "bf16" else torch.float32
    xd, td = x.to(dev(), dt), torch.from_numpy(tt.table_t).to(dev())
    dx = torch.full((R + 3, B, 16), float("nan"), dtype=dt, device=dev())
    dW, db = ops.spiral_conv_bwd_wgt_thin(ext.to(dev()), xd, td, R, S, 16, 3, weight=W.to(dev()), dx=dx, act_prev=a, zero_prev=R - 1)
    err = float((dx[:R].float().cpu().double() - torch.from_numpy(ref)).abs().max())
    assert err <= tol * float(np.abs(ref).max()) + 1e-6
    assert torch.isnan(dx[R:].float()).all()                             # rows behind the real ones are not touched
    dW0, db0 = ops.spiral_conv_bwd_wgt_thin(ext.to(dev()), xd, td, R, S, 16, 3)
    assert torch.equal(dW, dW0) and torch.equal(db, db0)


def test_thin_wgrad_rejects_what_it_does_not_cover():
    x = torch.zeros((10, 24, 16), device=dev())
    g = torch.zeros((10, 24, 3), device=dev())
    t = torch.zeros((10, 4), dtype=torch.int32, device=dev())
    with pytest.raises(RuntimeError):
        ops.spiral_conv_bwd_wgt_thin(g, x, t, 10, 4, 16, 3)              # batch not a multiple of 16

"""bf16 compute path (BASELINE config 3): every bf16 kernel, through the C ABI, against a float64 evaluation of the same
operator on the SAME bf16-rounded operands (so what is measured is the kernel - fp32 accumulation, one rounding of the
result - not the quantisation of the inputs), and the whole model against the fp32 oracle within the tolerance SURVEY 8a
states for this configuration (forward 1e-2 relative).

Tolerances: a bf16 result carries one rounding of 2^-9 relative (half an ulp of 8 significant bits); sums of K <= 1024
bf16 products accumulated in fp32 add <= 1e-6 relative.  Tensor checks use  max|err| <= 2^-8 * max|ref|  (one ulp at the
tensor's scale) unless stated otherwise."""
import os

import numpy as np
import pytest
import torch

import semantichuman_amd as sh
from semantichuman_amd import mesh_ops, ops
from tests import emulate

pytestmark = pytest.mark.gpu
ULP = 2.0 ** -8


def dev():
    return torch.device("cuda:0")


def bf(t):
    """round an fp32 tensor to bf16 and back (the value the kernels see)"""
    return t.to(torch.bfloat16).to(torch.float32)


def rand_table(R, n_in, S, seed):
    g = np.random.RandomState(seed)
    t = g.randint(0, n_in, size=(R, S)).astype(np.int32)
    t[:, 0] = np.arange(R) % n_in
    t[g.rand(R, S) < 0.1] = n_in - 1                # padding entries -> the dummy row
    return t


# (B, n_in, R, S, Cin, Cout, act): channel modes 16 / 32 / 64 / 128, odd spiral lengths (K % 32 != 0), odd batches,
# R < n_in (fused row select), channel tiles 1 / 2 / 4 / 8
CONV_SHAPES = [
    (16, 50, 50, 9, 16, 32, "elu"), (3, 41, 20, 11, 16, 16, "relu"), (64, 70, 70, 8, 32, 64, "elu"),
    (20, 33, 33, 8, 64, 128, "elu"), (16, 40, 40, 8, 128, 64, "tanh"), (5, 64, 64, 10, 32, 32, "identity"),
    (32, 30, 30, 18, 64, 32, "leaky_relu"), (17, 45, 45, 7, 32, 16, "sigmoid"),
    # the full-line form (64 / 128 gathered channels, <= 2 channel tiles: 8-row tiles, paired lanes) with ragged batches
    (5, 40, 40, 8, 64, 32, "elu"), (17, 45, 45, 7, 128, 16, "relu"), (64, 33, 20, 9, 64, 16, "tanh"),
]


@pytest.mark.parametrize("shape", CONV_SHAPES)
@pytest.mark.parametrize("layouts", [("vm", "vm"), ("bm", "vm"), ("vm", "bm")])
def test_conv_fwd_and_bwd_data_bf16(shape, layouts):
    B, n_in, R, S, Cin, Cout, act = shape
    li, lo = layouts
    torch.manual_seed(0)
    table = rand_table(R, n_in, S, 1)
    x = bf(torch.randn(n_in, B, Cin))                                   # vertex-major master copy
    x[-1] = 0.0 if act != "sigmoid" else x[-1]
    W = bf(torch.randn(Cout, S * Cin) / np.sqrt(S * Cin))
    bias = torch.randn(Cout) * 0.1
    a = ops.act_id(act)
    ref = emulate.conv_fwd(x.double().numpy(), table, W.double().numpy(), bias.double().numpy(), a, R - 1)

    xd = (x if li == "vm" else x.permute(1, 0, 2).contiguous()).to(dev(), torch.bfloat16)
    y = torch.full((R, B, Cout) if lo == "vm" else (B, R, Cout), float("nan"), dtype=torch.bfloat16, device=dev())
    wf, wft = ops.conv_wfrag_prep([W.to(dev())] * 2, [(S, Cin, Cout)] * 2, [0, 1])
    td = torch.from_numpy(table).to(dev())
    ops.spiral_conv_fwd_bf16(xd, li, td, wf, bias.to(dev()), y, lo, R, S, Cin, Cout, a, R - 1)
    got = y.float().cpu()
    got = got if lo == "vm" else got.permute(1, 0, 2)
    assert torch.isfinite(got).all()
    err = float((got.double() - torch.from_numpy(ref)).abs().max())
    assert err <= ULP * float(np.abs(ref).max()) + 1e-6, (err, np.abs(ref).max())
    assert float(got[R - 1].abs().max()) == 0.0                         # masked dummy row (models.py:49-51)

    # backward-data: the same kernel over the transposed table, epilogue x act'(yprev), zero row
    tt = mesh_ops.transpose_table_dense(table, n_in, none_row=R - 1, skip_row=-1)
    dpre = bf(torch.randn(R, B, Cout))
    dpre[R - 1] = 0
    ext = emulate.extend_dpre(dpre.double().numpy(), tt)
    yprev = bf(torch.randn(n_in, B, Cin))
    ref_dx = emulate.conv_bwd_data(ext, tt.table_t, W.double().numpy(), Cin) * emulate.DACT[a](yprev.double().numpy())
    ref_dx[n_in - 1] = 0
    dp_dev = torch.from_numpy(ext).to(torch.float32)
    # extra rows are sums of bf16 rows: round them as the kernel chain would store them
    dp_dev = bf(dp_dev)
    ref_dx = emulate.conv_bwd_data(dp_dev.double().numpy(), tt.table_t, W.double().numpy(), Cin) * emulate.DACT[a](yprev.double().numpy())
    ref_dx[n_in - 1] = 0
    dpd = (dp_dev if lo == "vm" else dp_dev.permute(1, 0, 2).contiguous()).to(dev(), torch.bfloat16)
    ypd = yprev.to(dev(), torch.bfloat16)
    dx = torch.full((n_in, B, Cin) if li == "vm" else (B, n_in, Cin), float("nan"), dtype=torch.bfloat16, device=dev())
    ops.spiral_conv_bwd_data_bf16(dpd, lo, torch.from_numpy(tt.table_t).to(dev()), wft, dx, li, ypd, "vm", a, n_in - 1, n_in, S, Cin, Cout)
    gdx = dx.float().cpu()
    gdx = gdx if li == "vm" else gdx.permute(1, 0, 2)
    assert torch.isfinite(gdx).all()
    err = float((gdx.double() - torch.from_numpy(ref_dx)).abs().max())
    assert err <= ULP * float(np.abs(ref_dx).max()) + 1e-6, (err, np.abs(ref_dx).max())

    # round 6: the same gradient over ragged source lists (conv_bf16r_kernel) - no pre-summed rows, so the reference is the float64
    # gradient of the UNROUNDED sums (the sums are formed in fp32 by the matrix pipe)
    # (the input's dummy row - read through every padding entry, a list of dozens of sources - is masked in dx: skipped, as the stacks do
    # where that gradient is dead; a list longer than 64 sources keeps the dense form)
    rag = mesh_ops.transpose_table_ragged(table, n_in, none_row=R - 1, skip_row=n_in - 1)
    assert rag is not None
    if ops.spiral_conv_bf16_rag_ok(B, S, Cout, Cin, rag[0].shape[1]):
        ref_r = emulate.conv_bwd_data(ext, tt.table_t, W.double().numpy(), Cin) * emulate.DACT[a](yprev.double().numpy())
        ref_r[n_in - 1] = 0
        dpr = (dpre if lo == "vm" else dpre.permute(1, 0, 2).contiguous()).to(dev(), torch.bfloat16)
        dxr = torch.full_like(dx, float("nan"))
        ops.spiral_conv_bwd_data_bf16_rag(dpr, lo, torch.from_numpy(rag[0]).to(dev()), torch.from_numpy(rag[1]).to(dev()), wft, dxr, li, ypd, "vm",
                                          a, n_in - 1, n_in, S, Cin, Cout)
        gr = dxr.float().cpu()
        gr = gr if li == "vm" else gr.permute(1, 0, 2)
        assert torch.isfinite(gr).all()
        err = float((gr.double() - torch.from_numpy(ref_r)).abs().max())
        assert err <= ULP * float(np.abs(ref_r).max()) + 1e-6, (err, np.abs(ref_r).max())
    else:
        assert Cout % 32 != 0, "the ragged form takes every layer that gathers a multiple of 32 channels"


@pytest.mark.parametrize("B,N,S,Cout", [(16, 60, 10, 16), (5, 37, 9, 16), (64, 50, 3, 16)])
def test_conv_bf16_three_channel_fp32_sides(B, N, S, Cout):
    """First encoder layer: fp32 xyz in (batch-major, the reference layout), bf16 out.  Last decoder layer: bf16 in, fp32 xyz
    out; and its backward-data: fp32 xyz gradient in, bf16 out."""
    torch.manual_seed(1)
    n1 = N + 1
    table = rand_table(n1, n1, S, 2)
    a = ops.act_id("elu")
    # --- 3 (fp32) -> Cout (bf16)
    x = torch.randn(B, n1, 3)
    x[:, -1] = 0
    W = bf(torch.randn(Cout, S * 3) / np.sqrt(S * 3))
    bias = torch.randn(Cout) * 0.1
    xr = bf(x)                                                           # the kernel rounds the fp32 input to bf16 on load
    ref = emulate.conv_fwd(xr.permute(1, 0, 2).double().numpy(), table, W.double().numpy(), bias.double().numpy(), a, n1 - 1)
    wf, = ops.conv_wfrag_prep([W.to(dev())], [(S, 3, Cout)], [0])
    y = torch.full((n1, B, Cout), float("nan"), dtype=torch.bfloat16, device=dev())
    td = torch.from_numpy(table).to(dev())
    ops.spiral_conv_fwd_bf16(x.to(dev()), "bm", td, wf, bias.to(dev()), y, "vm", n1, S, 3, Cout, a, n1 - 1)
    err = float((y.float().cpu().double() - torch.from_numpy(ref)).abs().max())
    assert err <= ULP * float(np.abs(ref).max()) + 1e-6
    # --- Cout (bf16) -> 3 (fp32, batch-major), identity activation
    W2 = bf(torch.randn(3, S * Cout) / np.sqrt(S * Cout))
    b2 = torch.randn(3) * 0.1
    h = bf(torch.randn(n1, B, Cout))
    ref2 = emulate.conv_fwd(h.double().numpy(), table, W2.double().numpy(), b2.double().numpy(), 0, n1 - 1)
    wf2, wf2t = ops.conv_wfrag_prep([W2.to(dev())] * 2, [(S, Cout, 3)] * 2, [0, 1])
    out = torch.full((B, n1, 3), float("nan"), dtype=torch.float32, device=dev())
    ops.spiral_conv_fwd_bf16(h.to(dev(), torch.bfloat16), "vm", td, wf2, b2.to(dev()), out, "bm", n1, S, Cout, 3, 0, n1 - 1)
    err = float((out.cpu().permute(1, 0, 2).double() - torch.from_numpy(ref2)).abs().max())
    assert err <= 2e-6 * float(np.abs(ref2).max()) * S * Cout ** 0.5 + 1e-6        # fp32 result: accumulation noise only
    assert float(out[:, -1].abs().max()) == 0.0
    # --- its backward-data: dpre fp32 [n1, B, 3] (vertex-major) -> dx bf16 [n1, B, Cout]
    tt = mesh_ops.transpose_table_dense(table, n1, none_row=n1 - 1, skip_row=-1)
    dpre = torch.randn(n1, B, 3)
    dpre[-1] = 0
    ext = torch.from_numpy(emulate.extend_dpre(dpre.double().numpy(), tt)).float()
    yprev = bf(torch.randn(n1, B, Cout))
    ref3 = emulate.conv_bwd_data(bf(ext).double().numpy(), tt.table_t, W2.double().numpy(), Cout) * emulate.DACT[a](yprev.double().numpy())
    ref3[n1 - 1] = 0
    dx = torch.full((n1, B, Cout), float("nan"), dtype=torch.bfloat16, device=dev())
    ops.spiral_conv_bwd_data_bf16(ext.to(dev()), "vm", torch.from_numpy(tt.table_t).to(dev()), wf2t, dx, "vm",
                                  yprev.to(dev(), torch.bfloat16), "vm", a, n1 - 1, n1, S, Cout, 3)
    err = float((dx.float().cpu().double() - torch.from_numpy(ref3)).abs().max())
    assert err <= ULP * float(np.abs(ref3).max()) + 1e-6


@pytest.mark.parametrize("shape", CONV_SHAPES + [(64, 40, 40, 10, 16, 8, "elu"), (7, 25, 25, 6, 8, 24, "elu")])
def test_conv_bwd_wgt_bf16(shape):
    B, n_in, R, S, Cin, Cout, _ = shape
    torch.manual_seed(2)
    table = rand_table(R, n_in, S, 3)
    x = bf(torch.randn(n_in, B, Cin))
    dpre = bf(torch.randn(R, B, Cout))
    dW_ref, db_ref = emulate.conv_bwd_wgt(dpre.double().numpy(), x.double().numpy(), table)
    td = torch.from_numpy(table).to(dev())
    for xl, dl in (("vm", "vm"), ("bm", "vm")):
        xd = (x if xl == "vm" else x.permute(1, 0, 2).contiguous()).to(dev(), torch.bfloat16)
        dW, db = ops.spiral_conv_bwd_wgt_bf16(dpre.to(dev(), torch.bfloat16), dl, xd, xl, td, R, S, Cin, Cout)
        # fp32 accumulation of exact bf16 products over R*B rows: relative error ~ sqrt(rows) * 2^-24
        tol = 2e-6 * np.sqrt(R * B) + 1e-6
        assert float(np.abs(dW.cpu().double().numpy() - dW_ref).max()) <= tol * float(np.abs(dW_ref).max()) + 1e-5
        assert float(np.abs(db.cpu().double().numpy() - db_ref).max()) <= tol * float(np.abs(db_ref).max()) + 1e-4
    a, _ = ops.spiral_conv_bwd_wgt_bf16(dpre.to(dev(), torch.bfloat16), "vm", x.to(dev(), torch.bfloat16), "vm", td, R, S, Cin, Cout)
    b, _ = ops.spiral_conv_bwd_wgt_bf16(dpre.to(dev(), torch.bfloat16), "vm", x.to(dev(), torch.bfloat16), "vm", td, R, S, Cin, Cout)
    assert torch.equal(a, b)                                             # fixed-order slab reduction: bitwise reproducible


@pytest.mark.parametrize("B,N,S,C", [(16, 60, 10, 16), (5, 37, 9, 16), (64, 130, 3, 16)])
def test_conv_bwd_wgt_bf16_three_channel_fp32_sides(B, N, S, C):
    torch.manual_seed(3)
    n1 = N + 1
    table = rand_table(n1, n1, S, 4)
    td = torch.from_numpy(table).to(dev())
    # first layer: x fp32 [B, n1, 3] (batch-major), dpre bf16 [n1, B, C]
    x = torch.randn(B, n1, 3)
    dpre = bf(torch.randn(n1, B, C))
    dW_ref, db_ref = emulate.conv_bwd_wgt(dpre.double().numpy(), bf(x).permute(1, 0, 2).double().numpy(), table)
    dW, db = ops.spiral_conv_bwd_wgt_bf16(dpre.to(dev(), torch.bfloat16), "vm", x.to(dev()), "bm", td, n1, S, 3, C)
    tol = 2e-6 * np.sqrt(n1 * B) + 1e-6
    assert float(np.abs(dW.cpu().double().numpy() - dW_ref).max()) <= tol * float(np.abs(dW_ref).max()) + 1e-5
    assert float(np.abs(db.cpu().double().numpy() - db_ref).max()) <= tol * float(np.abs(db_ref).max()) + 1e-4
    # last layer: x bf16 [n1, B, C], dpre fp32 [n1, B, 3]
    h = bf(torch.randn(n1, B, C))
    g = torch.randn(n1, B, 3)
    dW_ref, db_ref = emulate.conv_bwd_wgt(bf(g).double().numpy(), h.double().numpy(), table)
    dW, db = ops.spiral_conv_bwd_wgt_bf16(g.to(dev()), "vm", h.to(dev(), torch.bfloat16), "vm", td, n1, S, C, 3)
    assert float(np.abs(dW.cpu().double().numpy() - dW_ref).max()) <= tol * float(np.abs(dW_ref).max()) + 1e-5
    assert float(np.abs(db.cpu().double().numpy() - db_ref).max()) <= tol * float(np.abs(db_ref).max()) + 1e-4


@pytest.mark.parametrize("mnk", [(64, 256, 55296), (64, 55296, 256), (3, 16, 1376), (16, 1376, 16), (130, 72, 200), (1, 8, 8)])
@pytest.mark.parametrize("xf32", [False, True])
def test_latent_linear_bf16(mnk, xf32):
    """fwd / bwd_data / bwd_wgt of the latent FCs with bf16 working weights; x or dy may be fp32 (the latent code)."""
    M, N, K = mnk
    torch.manual_seed(4)
    W = bf(torch.randn(N, K) / np.sqrt(K))
    x = torch.randn(M, K)
    dy = torch.randn(M, N) / np.sqrt(N)
    bias = torch.randn(N) * 0.1
    xr, dyr = bf(x), bf(dy)                         # what the kernels multiply (fp32 operands are rounded on load)
    wd = ops.cast_bf16(W.to(dev()))
    assert torch.equal(wd.float().cpu(), W)
    xd = x.to(dev()) if xf32 else x.to(dev(), torch.bfloat16)
    dyd = dy.to(dev()) if xf32 else dy.to(dev(), torch.bfloat16)
    acc_tol = 4e-6
    for out_dtype in (torch.float32, torch.bfloat16):
        tol = acc_tol if out_dtype == torch.float32 else ULP
        y = ops.linear_fwd_bf16(xd, wd, bias.to(dev()), out_dtype)
        ref = xr.double() @ W.double().T + bias.double()
        assert float((y.double().cpu() - ref).abs().max()) <= tol * float(ref.abs().max()) * max(1.0, np.sqrt(K) / 8) + 1e-6
        dx = ops.linear_bwd_data_bf16(dyd, wd, out_dtype)
        ref = dyr.double() @ W.double()
        assert float((dx.double().cpu() - ref).abs().max()) <= tol * float(ref.abs().max()) * max(1.0, np.sqrt(N) / 8) + 1e-6
    dW, db = ops.linear_bwd_wgt_bf16(dyd, xd)
    ref = dyr.double().T @ xr.double()
    assert float((dW.double().cpu() - ref).abs().max()) <= acc_tol * float(ref.abs().max()) * max(1.0, np.sqrt(M) / 4) + 1e-6
    assert float((db.double().cpu() - dyr.double().sum(0)).abs().max()) <= acc_tol * float(dyr.abs().sum(0).max()) + 1e-6
    dW2, _ = ops.linear_bwd_wgt_bf16(dyd, xd)
    assert torch.equal(dW, dW2)


# ------------------------------------------------------------------------------------------ whole model
from oracle import ref_cpu                                              # noqa: E402  (checker only)
from semantichuman_amd.hierarchy import load_hierarchy                  # noqa: E402

FE = [[3, 16, 32, 64, 128], [[], [], [], [], []]]
FD = [[128, 64, 32, 32, 16], [[], [], [], [], 3]]
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _models(h, g, latent, seed=0):
    S, D, U = h.dense_constants()
    torch.manual_seed(seed)
    om = ref_cpu.SpiralAEOracle(FE, FD, latent, h.sizes, h.spiral_sizes, S, D, U)
    if g is not None:
        om.load_state_dict({k[3:]: torch.from_numpy(g[k]) for k in g.files if k.startswith("w0/")})
    m = sh.SpiralAutoencoder(FE, FD, latent, h.sizes, h.spiral_sizes, h.spirals, h.D, h.U, dev())
    m.load_state_dict(om.state_dict())
    return m.set_compute_dtype(torch.bfloat16), om


def rel(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return float((a - b).abs().max() / b.abs().max())


def _emulated_bf16_forward(m, x):
    """float64 evaluation of the plain autoencoder with bf16 rounding exactly where the bf16 path rounds (activations
    after every step, working weights, the fp32 input / latent code on load); conv / re-sampling formulation of
    tests/emulate.py over the model's own step tables."""
    r = lambda a: torch.from_numpy(np.asarray(a, dtype=np.float32)).to(torch.bfloat16).double().numpy()     # noqa: E731

    def run(stack, cur, convs, last_fp32):
        n = len(stack.steps)
        for i, st in enumerate(stack.steps):
            if st.kind == "conv":
                c = convs[st.param].conv
                cur = emulate.conv_fwd(cur, st.table, r(c.weight.detach().cpu()), c.bias.detach().double().cpu().numpy(), st.act, st.zero_row)
            else:
                cur = emulate.spmm(st.csr, cur)
            if not (last_fp32 and i == n - 1):
                cur = r(cur)
        return cur
    B = x.shape[0]
    h = run(m._enc_stack, r(x.permute(1, 0, 2)), m.conv, False)                      # [rows, B, C]
    hz = np.transpose(h, (1, 0, 2)).reshape(B, -1)
    z = hz @ r(m.fc_latent_enc.weight.detach().cpu()).T + m.fc_latent_enc.bias.detach().double().cpu().numpy()
    z = z.astype(np.float32).astype(np.float64)
    y = r(r(z) @ r(m.fc_latent_dec.weight.detach().cpu()).T + m.fc_latent_dec.bias.detach().double().cpu().numpy())
    y = np.transpose(y.reshape(B, m.sizes[-1] + 1, -1), (1, 0, 2))
    out = run(m._dec_stack, y, m.dconv, True)
    return np.transpose(out, (1, 0, 2)), z


def test_model_bf16_vs_reference_golden_and_fp32_oracle():
    """forward within 1e-2 (relative to the tensor's scale) of the REFERENCE's own fp32 output
    (small_ae.npz, produced by the reference in the build container) and of the fp32 oracle; gradients of the bf16 step
    against the oracle's fp32 gradients in the l2 sense (bf16 rounding noise is ~2^-9 per tensor element, uncorrelated)."""
    p = os.path.join(GOLDEN, "small_ae.npz")
    g, h = np.load(p), load_hierarchy(p)
    m, om = _models(h, g, 16)
    x = torch.from_numpy(g["x"])
    xd = x.to(dev())
    x_hat, z = m(xd)
    assert x_hat.dtype == torch.float32 and z.dtype == torch.float32       # the module's API keeps fp32 tensors
    # (1) the kernels do what the bf16 formulation says: against a float64 evaluation that rounds to bf16 at the same points
    #     (differences left: fp32 accumulation order, and the rare 1-ulp rounding flip it causes downstream)
    xe, ze = _emulated_bf16_forward(m, x)
    assert rel(z, torch.from_numpy(ze)) <= 4e-3 and rel(x_hat, torch.from_numpy(xe)) <= 4e-3
    # (2) the bf16 formulation against the REFERENCE's fp32 vectors.  These weights are a smooth closed-form fill
    #     (w = a sin(b i + c), oracle/gen_golden.py): rounding errors of neighbouring terms are correlated and do not average
    #     out in the latent FC, hence 3e-2 on z here; the random-initialised full-size model below holds 1e-2
    assert rel(x_hat, torch.from_numpy(g["x_hat"])) <= 1e-2
    assert rel(z, torch.from_numpy(g["z"])) <= 3e-2                      # 1e-2 holds on random weights: test_model_bf16_vs_reference_random_init
    assert rel(m.decode(torch.from_numpy(g["z_in"]).to(dev())), torch.from_numpy(g["decode_out"])) <= 1e-2
    assert float(x_hat[:, -1].abs().max()) == 0.0
    sh.l1_loss(xd, x_hat).backward()
    xo, zo = om(x)
    torch.nn.functional.l1_loss(x, xo).backward()
    for (n, a), b in zip(m.named_parameters(), om.parameters()):
        assert a.grad.dtype == torch.float32
        ga, gb = a.grad.double().cpu(), b.grad.double()
        err = float((ga - gb).norm() / gb.norm())
        assert err <= 1e-1, (n, err)          # correlated rounding on the smooth weight fill (see above); 5e-2 at full size below


def test_model_bf16_vs_reference_random_init():
    """the bf16 bar of SURVEY 8a (forward within 1e-2) against the REFERENCE's own output under the
    reference's own default (random) initialisation - small_ae_random.npz, produced by oracle/gen_golden.py gen_small_random
    from /root/reference on the 170-vertex hierarchy: x_hat AND z within 1e-2, every L1-loss gradient within 5e-2 (l2)."""
    g = np.load(os.path.join(GOLDEN, "small_ae_random.npz"))
    h = load_hierarchy(os.path.join(GOLDEN, "small_ae.npz"))
    m, _ = _models(h, g, 16)
    x = torch.from_numpy(g["x"]).to(dev())
    x_hat, z = m(x)
    assert rel(x_hat, torch.from_numpy(g["x_hat"])) <= 1e-2
    assert rel(z, torch.from_numpy(g["z"])) <= 1e-2
    sh.l1_loss(x, x_hat).backward()
    for n, a in m.named_parameters():
        gb = torch.from_numpy(g["grad_l1/" + n]).double()
        err = float((a.grad.double().cpu() - gb).norm() / gb.norm())
        assert err <= 5e-2, (n, err)


def test_model_bf16_full_size_6890():
    from semantichuman_amd import synthetic
    h = load_hierarchy(os.path.join(GOLDEN, "template6890.npz"))
    m, om = _models(h, None, 256, seed=3)
    m32 = sh.SpiralAutoencoder(FE, FD, 256, h.sizes, h.spiral_sizes, h.spirals, h.D, h.U, dev())
    m32.load_state_dict(om.state_dict())
    x = torch.from_numpy(synthetic.synth_batch(h.verts, 64, seed=1)).to(dev())
    x_hat, z = m(x)
    x32, z32 = m32(x)                                   # the fp32 HIP path (itself pinned to the oracle / reference elsewhere)
    assert rel(x_hat, x32) <= 1e-2 and rel(z, z32) <= 1e-2
    # ... and DIRECTLY against the oracle (oracle/ref_cpu.py, the reference's formulation on CPU) on a two-mesh slice: every
    # mesh of a batch goes through the kernels independently, so rows 0-1 of the batch-64 result are the batch-2 result
    with torch.no_grad():
        xo, zo = om(x[:2].cpu())
    assert rel(x_hat[:2], xo) <= 1e-2 and rel(z[:2], zo) <= 1e-2
    x_hat2, z2 = m(x)
    assert torch.equal(x_hat, x_hat2) and torch.equal(z, z2)
    sh.l1_loss(x, x_hat).backward()
    sh.l1_loss(x, x32).backward()
    g1 = [q.grad.clone() for q in m.parameters()]
    for (n, a), b in zip(m.named_parameters(), m32.parameters()):
        err = float((a.grad.double() - b.grad.double()).norm() / b.grad.double().norm())
        assert err <= 5e-2, (n, err)
    m.zero_grad()
    sh.l1_loss(x, m(x)[0]).backward()
    for a, q in zip(g1, m.parameters()):
        assert torch.equal(a, q.grad)                   # deterministic: fixed-order reductions, no atomics


def test_adam_keeps_bf16_working_copies_current():
    from semantichuman_amd import shadow
    p = os.path.join(GOLDEN, "small_ae.npz")
    g, h = np.load(p), load_hierarchy(p)
    m, _ = _models(h, g, 16)
    x = torch.from_numpy(g["x"]).to(dev())
    opt = sh.optim.Adam(m.parameters(), lr=1e-3, weight_decay=5e-5)
    for _ in range(3):
        opt.zero_grad()
        sh.l1_loss(x, m(x)[0]).backward()
        opt.step()
    for w in (m.fc_latent_enc.weight, m.fc_latent_dec.weight):
        s = shadow.lookup(w)
        assert s is not None and torch.equal(s, w.detach().to(torch.bfloat16))
    # an in-place change made through torch invalidates the copy; the next use re-converts
    with torch.no_grad():
        m.fc_latent_enc.weight.mul_(0.5)
    assert shadow.lookup(m.fc_latent_enc.weight) is None
    m(x)
    assert torch.equal(shadow.lookup(m.fc_latent_enc.weight), m.fc_latent_enc.weight.detach().to(torch.bfloat16))


def test_captured_bf16_step_survives_load_state_dict():
    """A hipGraph of the bf16 step has the bf16 working copies' addresses baked in (the latent FCs read them, Adam rewrites
    them).  load_state_dict makes the copies stale; the next eager use must refresh them IN PLACE, so that a replay of the
    old graph reads current weights: replay from a restored state == the first replay from that state, bit for bit."""
    from semantichuman_amd import shadow
    p = os.path.join(GOLDEN, "small_ae.npz")
    g, h = np.load(p), load_hierarchy(p)
    m, _ = _models(h, g, 16)
    x = torch.from_numpy(g["x"]).to(dev())
    opt = sh.optim.Adam(m.parameters(), lr=1e-3, weight_decay=5e-5)

    def step():
        opt.zero_grad(set_to_none=True)
        sh.l1_loss(x, m(x)[0]).backward()
        opt.step()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(2):
            step()
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        step()
    torch.cuda.synchronize()
    sd0 = {k: v.detach().clone() for k, v in m.state_dict().items()}
    opt0 = {id(q): {k: v.detach().clone() for k, v in opt.state[q].items()} for q in m.parameters()}
    addr = {n: shadow.lookup(w).data_ptr() for n, w in (("enc", m.fc_latent_enc.weight), ("dec", m.fc_latent_dec.weight))}
    graph.replay()
    torch.cuda.synchronize()
    first = [q.detach().clone() for q in m.parameters()]
    assert not torch.equal(first[0], list(sd0.values())[0])          # the replay did train
    # back to the captured state through torch: parameters by load_state_dict (in-place copies: version bump -> stale copies),
    # optimizer state tensor by tensor
    m.load_state_dict(sd0)
    for q in m.parameters():
        for k, v in opt.state[q].items():
            v.copy_(opt0[id(q)][k])
    assert shadow.lookup(m.fc_latent_enc.weight) is None             # stale until the next use ...
    with torch.no_grad():
        m(x)                                                          # ... which refreshes them where they are
    for n, w in (("enc", m.fc_latent_enc.weight), ("dec", m.fc_latent_dec.weight)):
        assert shadow.lookup(w) is not None and shadow.lookup(w).data_ptr() == addr[n], n
    graph.replay()
    torch.cuda.synchronize()
    for a, q in zip(first, m.parameters()):
        assert torch.equal(a, q.detach())


def test_matched_l2_bf16_vs_fp32_training():
    """Config 3's acceptance: 'matched L2' on the trained metric.  30 training steps (batch 16, L1 + 1e-2 edge loss, Adam)
    from the same weights on the same batches, bf16 path vs fp32 path; held-out per-vertex L2 within 5 %.
    30 steps from a random init end in the steep part of the descent (L2 ~ 90 mm) where ONE evaluation of ONE trajectory moves by
    several per cent when an fp32 sum is merely re-associated: across the kernel revisions of rounds 2-6 the single-point gap read
    0.9 % ... 2.7 %, then 6.6 % when the bf16 backward-data pass went to ragged lists - whose gradients are as close to the fp32
    step's as the dense form's (tools/exp/bf16_grad_err.py: summed relative error 0.1140 against 0.1140 at batch 16, 0.1101 /
    0.1096 at 64).  So the figure compared is a mean: two seeds, evaluations after steps 24, 26, 28 and 30.  The bound leaves room
    for re-association, not for a wrong kernel: one dropped spiral tap or a stale working copy of a weight moves the figure by tens
    of per cent.  (The statement on a TRAINED model is tools/trained_l2.py: 2000 steps, 5 seeds, profiles/r06_trained_l2.json.)"""
    from semantichuman_amd import synthetic
    h = load_hierarchy(os.path.join(GOLDEN, "template6890.npz"))
    data = torch.from_numpy(synthetic.synth_batch(h.verts, 16 * 6, seed=100)).to(dev())
    test = torch.from_numpy(synthetic.synth_batch(h.verts, 16, seed=7)).to(dev())
    ft = sh.FaceTables(h.faces, h.sizes[0] + 1, dev())
    out = {}
    for dt in (torch.float32, torch.bfloat16):
        vals = []
        for seed in (2, 3):
            torch.manual_seed(seed)
            m = sh.SpiralAutoencoder(FE, FD, 256, h.sizes, h.spiral_sizes, h.spirals, h.D, h.U, dev()).set_compute_dtype(dt)
            opt = sh.optim.Adam(m.parameters(), lr=1e-3, weight_decay=5e-5)
            for i in range(30):
                xb = data[(i % 6) * 16:(i % 6 + 1) * 16]
                opt.zero_grad()
                loss, _ = sh.recon_loss(m(xb)[0], xb, ft, 1e-2)
                loss.backward()
                opt.step()
                if i + 1 in (24, 26, 28, 30):
                    with torch.no_grad():
                        vals.append(float(sh.vertex_l2_mm(m(test)[0], test)))
        out[dt] = (sum(vals) / len(vals), vals)
    a, b = out[torch.float32][0], out[torch.bfloat16][0]
    assert abs(b - a) <= 5e-2 * a, out


@pytest.mark.parametrize("name,B", [("small_ae.npz", 1), ("small_ae.npz", 5), ("small_ae.npz", 32), ("small_ae.npz", 96),
                                    ("template6890.npz", 32), ("template6890.npz", 48)])
def test_bf16_step_at_other_batch_sizes(name, B):
    """Every dispatch branch that depends on the batch (thin / LDS-DMA / staged weight gradients need B % 32 == 0 or
    (R * B) % 32 == 0, ragged batch tiles otherwise): one training step's gradients on the bf16 path stay within 2e-2
    (relative L2 per parameter) of the fp32 path of the same module, and are finite."""
    from semantichuman_amd import synthetic
    h = load_hierarchy(os.path.join(GOLDEN, name))
    torch.manual_seed(0)
    m = sh.SpiralAutoencoder(FE, FD, 32, h.sizes, h.spiral_sizes, h.spirals, h.D, h.U, dev())
    x = torch.from_numpy(synthetic.synth_batch(h.verts, B, seed=1)).to(dev())
    ft = sh.FaceTables(h.faces, h.sizes[0] + 1, dev())
    grads = {}
    for dt in (torch.float32, torch.bfloat16):
        m.set_compute_dtype(dt)
        m.zero_grad(set_to_none=True)
        loss, _ = sh.recon_loss(m(x)[0], x, ft, 1e-2)
        loss.backward()
        grads[dt] = {n: p.grad.clone() for n, p in m.named_parameters()}
    for n, a in grads[torch.float32].items():
        b = grads[torch.bfloat16][n]
        assert torch.isfinite(b).all(), n
        assert float((a - b).norm() / (a.norm() + 1e-12)) <= 2e-2, n


def test_bf16_step_runs_the_intended_kernels():
    """Dispatch guard at the benchmark shape (6890 vertices, batch 64): one training step on the bf16 path launches the
    kernels DESIGN 4b describes - a silent fall-back to a slower form (staged weight gradients, general kernels for the thin
    layer, plain gathers on the 64-channel layers, unfolded up-sampling, per-stack fragment conversion) fails here."""
    from semantichuman_amd import _lib, synthetic
    h = load_hierarchy(os.path.join(GOLDEN, "template6890.npz"))
    torch.manual_seed(1)
    m = sh.SpiralAutoencoder(FE, FD, 256, h.sizes, h.spiral_sizes, h.spirals, h.D, h.U, dev()).set_compute_dtype(torch.bfloat16)
    x = torch.from_numpy(synthetic.synth_batch(h.verts, 64, seed=1)).to(dev())
    ft = sh.FaceTables(h.faces, h.sizes[0] + 1, dev())

    def step():
        m.zero_grad(set_to_none=True)
        sh.recon_loss(m(x)[0], x, ft, 1e-2)[0].backward()
    step()
    torch.cuda.synchronize()
    _lib.profile_enable(True)
    step()
    torch.cuda.synchronize()
    recs = _lib.profile_records_by_kernel()
    _lib.profile_enable(False)
    fam = {}
    for name, tag, _ms in recs:
        key = name.split("<")[0]
        fam.setdefault(key, []).append((name, tag))
    assert len(fam.get("wfrag_prep_kernel", [])) == 1                       # one conversion launch per step
    assert len(fam.get("wgrad_thin_kernel", [])) == 1 and "dx=1" in fam["wgrad_thin_kernel"][0][1]
    assert len(fam.get("wgrad_bf16_dma_kernel", [])) == 7                  # every bf16 x bf16 weight gradient
    assert len(fam.get("wgrad_bf16_kernel", [])) == 1                      # only the fp32 3-channel input side is staged
    conv = fam.get("conv_bf16_kernel", [])
    # 9 forward + 7 backward-data (dec4's rides in the thin launch); round 6: 5 of the 6 backward-data layers that gather a multiple
    # of 32 channels walk ragged source lists (conv_bf16r_kernel) and need no pre-sum launch - dec0 keeps the dense table: its input's
    # dummy row carries a live gradient (the latent FC writes that row) and is read through every padding entry, a list of ~1100
    assert len(fam.get("conv_bf16r_kernel", [])) == 5
    assert len(conv) == 11
    # line-wise loads (BC_C32C, round 3) on every other layer that gathers a multiple of 32 bf16 channels: 6 forward + dec0's backward
    assert sum(1 for n, _ in conv if n.split(",")[2].strip() == "4") == 7
    assert sum(1 for n, _ in conv if n.split(",")[2].strip() in ("0", "3")) == 0      # no plain / full-line gathers left
    up = [t for _, t in fam.get("spmm_bf16_kernel", []) if "rows=3445 " in t or "rows=1722 " in t or "rows=861 " in t]
    assert len(up) == 3                                                     # folded up-sampling: only the blended rows
    # what is left of spmm_bf16: 4 down-sampling + 3 up-sampling forward, their 7 transposes, the pre-sum of the one dense layer
    assert len(fam.get("spmm_bf16_kernel", [])) == 11, [t for _, t in fam.get("spmm_bf16_kernel", [])]      # 4 + 4 re-sampling, 3 pre-sums (dec3, dec0 x 2)


def test_model_bf16_with_second_convs_per_level():
    """reference models.py:72-75, 96-99: optional second conv per level (see tests/test_gpu_parity.py:
    test_second_conv_per_level_vs_oracle) on the bf16 path - forward within 1e-2 of the fp32 oracle, gradients within 5e-2 (l2)."""
    from semantichuman_amd import synthetic
    FE2 = [[3, 16, 32, 64, 128], [[], 16, 32, [], []]]
    FD2 = [[128, 64, 32, 32, 16], [[], 64, [], 32, 3]]
    h = load_hierarchy(os.path.join(GOLDEN, "small_ae.npz"))
    S, D, U = h.dense_constants()
    torch.manual_seed(11)
    om = ref_cpu.SpiralAEOracle(FE2, FD2, 32, h.sizes, h.spiral_sizes, S, D, U)
    m = sh.SpiralAutoencoder(FE2, FD2, 32, h.sizes, h.spiral_sizes, h.spirals, h.D, h.U, dev())
    m.load_state_dict(om.state_dict())
    m.set_compute_dtype(torch.bfloat16)
    x = torch.from_numpy(synthetic.synth_batch(h.verts, 16, seed=3))
    xd = x.to(dev())
    x_hat, z = m(xd)
    xo, zo = om(x)
    assert rel(x_hat, xo.detach()) <= 1e-2 and rel(z, zo.detach()) <= 1e-2
    sh.l1_loss(xd, x_hat).backward()
    torch.nn.functional.l1_loss(x, xo).backward()
    for (n, a), b in zip(m.named_parameters(), om.parameters()):
        ga, gb = a.grad.double().cpu(), b.grad.double()
        assert float((ga - gb).norm() / gb.norm()) <= 5e-2, n

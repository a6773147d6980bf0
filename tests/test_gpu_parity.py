"""GPU parity tests: every HIP kernel, through the C ABI, against the CPU oracle and the
reference's golden vectors.  Run with `pytest -m gpu` on an MI355X.

Tolerances (fp32):
  forward tensors      max-abs-err <= 1e-5 * max|ref|      (the reference's own fp32-vs-fp64
                                                           noise is 5e-7 relative)
  gradients            max-abs-err <= 1e-4 * max|g_ref|    per tensor
  row-select / gather  bit-exact
"""
import os

import numpy as np
import pytest
import torch

import semantichuman_amd as sh
from oracle import ref_cpu
from semantichuman_amd import mesh_ops, ops
from semantichuman_amd.hierarchy import load_hierarchy

pytestmark = pytest.mark.gpu
FE = [[3, 16, 32, 64, 128], [[], [], [], [], []]]
FD = [[128, 64, 32, 32, 16], [[], [], [], [], 3]]
FWD_TOL, GRAD_TOL = 1e-5, 1e-4


def dev():
    return torch.device("cuda:0")


def close(got, ref, tol, what="", floor=0.0):
    """max-abs-err <= tol * max|ref| (+ floor: an absolute term for tensors that are themselves the
    result of massive cancellation, e.g. a bias gradient of 1e-7 summed from +-1e-4 terms; given as
    1e-6 x the largest gradient magnitude of the whole model)."""
    got = got.detach().cpu().numpy() if torch.is_tensor(got) else np.asarray(got)
    ref = ref.detach().cpu().numpy() if torch.is_tensor(ref) else np.asarray(ref)
    assert got.shape == ref.shape, (what, got.shape, ref.shape)
    err, scale = np.abs(got - ref).max(), np.abs(ref).max()
    assert np.isfinite(got).all(), what
    assert err <= tol * scale + floor + 1e-30, "%s: err %.3e > %.1e * %.3e + %.1e" % (what, err, tol, scale, floor)


def test_the_forms_are_really_switched(request):
    """tests/conftest.py parametrizes every GPU test of this module over the three arithmetic forms of the fp32 products; the
    instance's id must be the form the library calls run in (pytest >= 8 no longer sets up a fixture that was only appended to
    metafunc.fixturenames: in round 4 all three instances ran in the process default)."""
    from semantichuman_amd import _lib
    form = request.node.callspec.params["f32_mma"]
    assert request.node.name.endswith("[%s]" % form)
    assert _lib.get_f32_mma_mode() == form and _lib.mma_id() == _lib.MMA_MODES[form]


def test_the_form_is_visible_as_a_fixture(f32_mma):
    from semantichuman_amd import _lib
    assert _lib.get_f32_mma_mode() == f32_mma


def test_native_library_is_loaded():
    from semantichuman_amd import _lib
    lib = _lib.load()
    assert lib.sh_version() >= 100
    with open("/proc/self/maps") as f:
        assert "libsh_kernels.so" in f.read()


# ------------------------------------------------------------------------------------------
@pytest.mark.parametrize("act", ["relu", "elu", "leaky_relu", "sigmoid", "tanh", "identity"])
def test_spiral_conv_module_vs_reference_golden(golden_dir, act):
    g = np.load(os.path.join(golden_dir, "conv_acts.npz"))
    B, N1, cin = g["x"].shape
    S = g["spirals"].shape[1]
    cout = g["w"].shape[0]
    m = sh.SpiralConv(cin, S, cout, activation=act, device=dev()).to(dev())
    with torch.no_grad():
        m.conv.weight.copy_(torch.from_numpy(g["w"]))
        m.conv.bias.copy_(torch.from_numpy(g["b"]))
    x = torch.from_numpy(g["x"]).to(dev()).requires_grad_(True)
    adj = torch.from_numpy(g["spirals"].astype(np.int64))[None].repeat(B, 1, 1).to(dev())     # int64, -1 padded
    y = m(x, adj)
    (y * torch.from_numpy(g["gy"]).to(dev())).sum().backward()
    close(y, g[act + "/y"], FWD_TOL, "y")
    assert float(y[:, -1].abs().max()) == 0.0                       # dummy row is masked (models.py:49-51)
    close(x.grad, g[act + "/gx"], GRAD_TOL, "gx")
    close(m.conv.weight.grad, g[act + "/gw"], GRAD_TOL, "gw")
    close(m.conv.bias.grad, g[act + "/gb"], GRAD_TOL, "gb")


SHAPES = [  # (B, n_rows, S, Cin, Cout)  - ragged / odd cases on purpose
    (1, 7, 1, 4, 4), (3, 50, 5, 3, 16), (5, 171, 11, 12, 20), (2, 300, 8, 16, 3), (70, 33, 9, 32, 32),
    (4, 129, 7, 64, 128), (130, 20, 3, 8, 64), (2, 64, 4, 5, 7),
    # dispatch corners: 128 channels with >= 768 row tiles (eight channel tiles -> two direct four-tile workgroups), two
    # channel tiles at a batch that is not a multiple of 64 (stay direct, 64-row workgroups), spiral length 18 (config 4)
    (64, 1600, 4, 16, 128), (48, 1400, 5, 16, 32), (32, 3200, 18, 32, 64),
    # more than 128 channels on either side (round 3: the output channels split over workgroups / launches, any count), and a
    # channel count past 128 that is not a multiple of 16
    (16, 200, 6, 16, 160), (5, 90, 4, 192, 272), (16, 300, 5, 160, 8), (3, 70, 3, 8, 132),
    # <= 3 output channels over 16-channel rows: the line-wise VALU forward (spiral lengths 6..12), full / ragged batch slices
    (64, 500, 10, 16, 3), (20, 130, 12, 16, 2), (7, 90, 6, 16, 1), (33, 64, 13, 16, 3),
    # ... and spirals of 13..24 (two passes over halves of the positions; config 4 forces 18), past 24 the general kernels
    (32, 300, 18, 16, 3), (16, 120, 24, 16, 3), (5, 60, 22, 16, 2), (8, 40, 25, 16, 3),
    (64, 400, 10, 3, 16), (9, 77, 7, 3, 16),
]


@pytest.mark.parametrize("shape", SHAPES)
@pytest.mark.parametrize("layouts", [("bm", "bm"), ("vm", "vm"), ("bm", "vm")])
def test_conv_kernels_vs_oracle(shape, layouts):
    B, N1, S, cin, cout = shape
    rs = np.random.RandomState(hash(shape) % 1000)
    table = rs.randint(0, N1, size=(N1, S)).astype(np.int32)
    table[:, 0] = np.arange(N1)
    table[rs.rand(N1, S) < 0.15] = N1 - 1                       # padding -> dummy row
    table[-1] = N1 - 1
    x = torch.from_numpy(rs.randn(B, N1, cin).astype(np.float32))
    x[:, -1] = 0.3                                              # a NON-zero dummy row (decoder input case)
    W = torch.from_numpy((rs.randn(cout, S * cin) / np.sqrt(S * cin)).astype(np.float32))
    b = torch.from_numpy(rs.randn(cout).astype(np.float32))
    gy = torch.from_numpy(rs.randn(B, N1, cout).astype(np.float32))
    # oracle
    xo, Wo, bo = x.clone().requires_grad_(True), W.clone().requires_grad_(True), b.clone().requires_grad_(True)
    adj = torch.from_numpy(table.astype(np.int64))[None]
    yo = ref_cpu.spiral_conv(xo, adj, Wo, bo, "elu")
    (yo * gy).sum().backward()
    # HIP
    lin, lout = layouts
    d = dev()
    xd = (x if lin == "bm" else x.permute(1, 0, 2)).contiguous().to(d)
    y = ops.alloc(B, N1, cout, lout, d)
    tab = torch.from_numpy(table).to(d)
    Wd, bd = W.to(d), b.to(d)
    ops.spiral_conv_fwd(xd, lin, tab, Wd, bd, y, lout, N1, S, ops.act_id("elu"), N1 - 1)
    y_bm = y if lout == "bm" else y.permute(1, 0, 2)
    close(y_bm, yo, FWD_TOL, "fwd")
    # backward: dpre -> wgrad, bwd-data
    gyd = (gy if lout == "bm" else gy.permute(1, 0, 2)).contiguous().to(d)
    dpre = ops.alloc(B, N1, cout, "vm", d)
    ops.act_backward(gyd, lout, y, lout, dpre, "vm", N1, ops.act_id("elu"), N1 - 1)
    dW, db = ops.spiral_conv_bwd_wgt(dpre, "vm", xd, lin, tab, N1, S, cin, cout)
    close(dW, Wo.grad, GRAD_TOL, "dW")
    close(db, bo.grad, GRAD_TOL, "db")
    tt = mesh_ops.transpose_table_dense(table, N1, none_row=N1 - 1)     # dpre[N1-1] == 0 (zero_row above)
    ext = ops.alloc(B, N1, cout, "vm", d, extra_rows=tt.n_extra)
    ext[:N1].copy_(dpre)
    for m, lo, n in ((tt.csr1, N1, tt.n1), (tt.csr2, N1 + tt.n1, tt.n2)):
        if m is not None:
            ops.spmm(tuple(torch.from_numpy(a).to(d) for a in (m.rowptr, m.col, m.val)), ext, "vm", ext[lo:], "vm", n)
    dx = ops.alloc(B, N1, cin, lin, d)
    ops.spiral_conv_bwd_data(ext, "vm", torch.from_numpy(tt.table_t).to(d), ops.weight_transpose(Wd, S, cin, cout),
                             dx, lin, None, "vm", 0, -1, N1, S, cin, cout)
    close(dx if lin == "bm" else dx.permute(1, 0, 2), xo.grad, GRAD_TOL, "dx")


def test_spiral_conv_table_cache_is_keyed_by_content():
    """the reference passes a fresh `S[i].repeat(bsize,1,1)` every forward; two index tensors of equal shape but
    different content must never share a cached gather table, wherever the allocator places them."""
    torch.manual_seed(0)
    n, s, cin, cout, b = 40, 5, 4, 8, 3
    conv = sh.SpiralConv(cin, s, cout, activation="identity").to(dev())
    x = torch.randn(b, n + 1, cin, device=dev())
    outs = []
    for seed in (1, 2, 1):
        g = torch.Generator().manual_seed(seed)
        adj = torch.randint(-1, n, (1, n + 1, s), generator=g)
        adj[0, -1] = -1
        a = adj.repeat(b, 1, 1).to(dev())
        outs.append(conv(x, a).clone())
        ref = torch.nn.functional.linear(x[:, adj[0].to(dev())].reshape(b, n + 1, s * cin), conv.conv.weight, conv.conv.bias)
        ref[:, -1] = 0
        assert float((outs[-1] - ref).abs().max()) <= 1e-5 * float(ref.abs().max())
        del a                                                    # let the allocator recycle the address for the next index
    assert torch.equal(outs[0], outs[2]) and not torch.equal(outs[0], outs[1])
    bad = torch.randint(-1, n, (b, n + 1, s))
    with pytest.raises(NotImplementedError):
        conv(x, bad.to(dev()))


def test_gather_is_bit_exact():
    """Identity weights, identity activation: the kernel output is a pure copy of the gathered
    rows (north_star: bit-exact spiral index gathers)."""
    B, N1, S, C = 3, 200, 6, 16
    rs = np.random.RandomState(3)
    table = rs.randint(0, N1, size=(N1, S)).astype(np.int32)
    x = torch.from_numpy(rs.randn(B, N1, C).astype(np.float32))
    d = dev()
    for s in range(S):
        W = torch.zeros(C, S * C)
        W[:, s * C:(s + 1) * C] = torch.eye(C)
        y = ops.alloc(B, N1, C, "bm", d)
        ops.spiral_conv_fwd(x.to(d), "bm", torch.from_numpy(table).to(d), W.to(d), None, y, "bm", N1, S, 0, -1)
        assert torch.equal(y.cpu(), x[:, table[:, s].astype(np.int64)])


def test_fused_row_select_is_exact_and_matches_dense_D(golden_dir):
    h = load_hierarchy(os.path.join(golden_dir, "small_ae.npz"))
    S, D, U = h.dense_constants()
    rs = np.random.RandomState(5)
    B, cin, cout = 4, 16, 32
    x = torch.from_numpy(rs.randn(B, h.sizes[1] + 1, cin).astype(np.float32)); x[:, -1] = 0
    W = torch.from_numpy((rs.randn(cout, h.spiral_sizes[1] * cin) * 0.1).astype(np.float32))
    b = torch.from_numpy(rs.randn(cout).astype(np.float32))
    yo = torch.matmul(D[1], ref_cpu.spiral_conv(x, S[1], W, b, "elu"))       # conv then dense D (models.py:122-127)
    d = dev()
    full = mesh_ops.spirals_to_table(h.spirals[1])
    fused = mesh_ops.compose_select(full, h.D[1].col)
    yf = ops.alloc(B, fused.shape[0], cout, "bm", d)
    ops.spiral_conv_fwd(x.to(d), "bm", torch.from_numpy(fused).to(d), W.to(d), b.to(d), yf, "bm", fused.shape[0],
                        h.spiral_sizes[1], 2, fused.shape[0] - 1)
    ya = ops.alloc(B, full.shape[0], cout, "bm", d)
    ops.spiral_conv_fwd(x.to(d), "bm", torch.from_numpy(full).to(d), W.to(d), b.to(d), ya, "bm", full.shape[0],
                        h.spiral_sizes[1], 2, full.shape[0] - 1)
    assert torch.equal(yf, ya[:, torch.from_numpy(h.D[1].col.astype(np.int64)).to(d)])    # fusion is bit-exact
    close(yf, yo, FWD_TOL, "fused conv+D vs dense")


def test_spmm_vs_dense(golden_dir):
    h = load_hierarchy(os.path.join(golden_dir, "small_ae.npz"))
    d = dev()
    rs = np.random.RandomState(7)
    for m in (h.U[0], h.U[2], h.D[1], h.U[1].transpose()):
        for C in (3, 32):
            x = torch.from_numpy(rs.randn(5, m.cols, C).astype(np.float32))
            ref = torch.matmul(torch.from_numpy(m.todense())[None], x)
            md = tuple(torch.from_numpy(a).to(d) for a in (m.rowptr, m.col, m.val))
            for lay in ("bm", "vm"):
                xd = (x if lay == "bm" else x.permute(1, 0, 2)).contiguous().to(d)
                y = ops.alloc(5, m.rows, C, lay, d)
                ops.spmm(md, xd, lay, y, lay, m.rows)
                close(y if lay == "bm" else y.permute(1, 0, 2), ref, FWD_TOL, "spmm")
    # a row-select applied by spmm is a bit-exact copy
    m = h.D[0]
    x = torch.from_numpy(rs.randn(2, m.cols, 8).astype(np.float32))
    y = ops.alloc(2, m.rows, 8, "bm", d)
    ops.spmm(tuple(torch.from_numpy(a).to(d) for a in (m.rowptr, m.col, m.val)), x.to(d), "bm", y, "bm", m.rows)
    assert torch.equal(y.cpu(), x[:, m.col.astype(np.int64)])


@pytest.mark.parametrize("mnk", [(64, 256, 55296), (64, 55296, 256), (4, 16, 1536), (3, 1536, 16), (5, 7, 13), (130, 70, 4100),
                                 (1, 8, 136), (64, 8, 2176), (20, 128, 2048), (3, 64, 4096), (33, 320, 256), (48, 192, 1088),
                                 (200, 640, 256), (1024, 1280, 256), (130, 128, 96), (65, 64, 384)])
def test_latent_linear_vs_torch(mnk):
    """y = x W^T + b and its gradients (the latent FCs, models.py:130,144) - ragged sizes, the
    split-reduction path (K = 55296, 4100), the wide-output path (N = 55296), batches of 3 / 20 / 33 / 48 rows through
    the streaming kernels (masked 16-row tiles), and batches of 65 / 130 / 200 / 1024 rows with a short reduction (K <= 384: the
    large-batch forward of the split forms, linear_fwd_wide_x3_kernel; config 5's decode)."""
    from semantichuman_amd.linear import latent_linear
    M, N, K = mnk
    rs = np.random.RandomState(M + N + K)
    x = torch.from_numpy(rs.randn(M, K).astype(np.float32))
    W = torch.from_numpy((rs.randn(N, K) / np.sqrt(K)).astype(np.float32))
    b = torch.from_numpy(rs.randn(N).astype(np.float32))
    gy = torch.from_numpy(rs.randn(M, N).astype(np.float32))
    xo, Wo, bo = (t.clone().double().requires_grad_(True) for t in (x, W, b))          # float64 reference
    yo = torch.nn.functional.linear(xo, Wo, bo)
    (yo * gy.double()).sum().backward()
    d = dev()
    xd, Wd, bd = (t.to(d).requires_grad_(True) for t in (x, W, b))
    y = latent_linear(xd, Wd, bd)
    (y * gy.to(d)).sum().backward()
    close(y, yo.float(), FWD_TOL, "y")
    close(xd.grad, xo.grad.float(), GRAD_TOL, "dx")
    close(Wd.grad, Wo.grad.float(), GRAD_TOL, "dW")
    close(bd.grad, bo.grad.float(), GRAD_TOL, "db")
    y2 = latent_linear(xd, Wd, None)                                                   # no bias
    close(y2, (yo - bo).float(), FWD_TOL, "y no bias", floor=1e-6)
    assert torch.equal(latent_linear(xd, Wd, bd), y)                                   # deterministic


def test_losses_and_metric_vs_oracle(golden_dir):
    h = load_hierarchy(os.path.join(golden_dir, "small_ae.npz"))
    rs = np.random.RandomState(11)
    B, N1 = 5, h.sizes[0] + 1
    x = torch.from_numpy(np.concatenate([rs.randn(B, N1 - 1, 3), np.zeros((B, 1, 3))], 1).astype(np.float32))
    xh = (x + 0.05 * torch.from_numpy(rs.randn(B, N1, 3).astype(np.float32)))
    xh[:, -1] = 0
    d = dev()
    xd, xhd = x.to(d), xh.to(d).requires_grad_(True)
    xho = xh.clone().requires_grad_(True)
    ft = sh.FaceTables(h.faces, N1, d)
    lo = torch.nn.functional.l1_loss(x, xho) * 0.7 + 0.01 * ref_cpu.edge_ratio_loss(xho, x, h.faces)
    lo.backward()
    l = sh.l1_loss(xd, xhd) * 0.7 + 0.01 * sh.edge_ratio_loss(xhd, xd, ft)
    l.backward()
    assert l.item() == pytest.approx(lo.item(), rel=2e-6)
    close(xhd.grad, xho.grad, GRAD_TOL, "loss grad")
    l1o, l2o = ref_cpu.eval_metrics(xh, x)
    assert sh.vertex_l2_mm(xhd, xd).item() == pytest.approx(l2o.item(), rel=2e-6)
    assert sh.eval_l1(xhd, xd).item() == pytest.approx(l1o.item(), rel=2e-6)
    # odd element count for the vector tail path
    a, b = torch.from_numpy(rs.randn(1031).astype(np.float32)), torch.from_numpy(rs.randn(1031).astype(np.float32))
    assert ops.l1_loss_fwd(a.to(d), b.to(d)).item() == pytest.approx((a - b).abs().mean().item(), rel=2e-6)


# ------------------------------------------------------------------------------------------
def make_models(h, g, latent):
    S, D, U = h.dense_constants()
    om = ref_cpu.SpiralAEOracle(FE, FD, latent, h.sizes, h.spiral_sizes, S, D, U)
    m = sh.SpiralAutoencoder(FE, FD, latent, h.sizes, h.spiral_sizes, h.spirals, h.D, h.U, dev())
    if g is not None:
        sd = {k[3:]: torch.from_numpy(g[k]) for k in g.files if k.startswith("w0/")}
        om.load_state_dict(sd)
    m.load_state_dict(om.state_dict())
    return m, om


@pytest.mark.parametrize("adam", ["torch", "hip"])
def test_autoencoder_vs_reference_golden(golden_dir, adam):
    """models.SpiralAutoencoder of the REFERENCE (run in the build container, vectors committed):
    outputs, latent, loss, every parameter gradient, weights after one Adam step (torch.optim.Adam and the
    library's own Adam kernel), eval metric."""
    p = os.path.join(golden_dir, "small_ae.npz")
    g, h = np.load(p), load_hierarchy(p)
    m, om = make_models(h, g, 16)
    x = torch.from_numpy(g["x"]).to(dev())
    opt = (torch.optim.Adam if adam == "torch" else sh.optim.Adam)(m.parameters(), lr=1e-3, weight_decay=5e-5)
    opt.zero_grad()
    x_hat, z = m(x)
    close(x_hat, g["x_hat"], FWD_TOL, "x_hat")
    close(z, g["z"], FWD_TOL, "z")
    close(m.decode(torch.from_numpy(g["z_in"]).to(dev())), g["decode_out"], FWD_TOL, "decode")
    assert sh.eval_l1(x_hat, x).item() == pytest.approx(float(g["eval_l1_w0"]), rel=1e-5)
    assert sh.vertex_l2_mm(x_hat, x).item() == pytest.approx(float(g["eval_l2mm_w0"]), rel=1e-5)
    ft = sh.FaceTables(h.faces, h.sizes[0] + 1, dev())
    rec, edge = sh.l1_loss(x, x_hat), sh.edge_ratio_loss(x_hat, x, ft)
    assert rec.item() == pytest.approx(float(g["loss_rec"]), rel=1e-5)
    assert edge.item() == pytest.approx(float(g["loss_edge"]), rel=1e-5)
    (rec + 1e-2 * edge).backward()
    gmax = max(float(np.abs(g["grad/" + name]).max()) for name, _ in m.named_parameters())
    for name, prm in m.named_parameters():
        close(prm.grad, g["grad/" + name], GRAD_TOL, "grad " + name, floor=1e-6 * gmax)
    opt.step()
    for name, prm in m.named_parameters():
        dlt = np.abs(prm.detach().cpu().numpy() - g["w1/" + name])
        assert dlt.max() <= 1e-4 and dlt.mean() <= 1e-7, (name, dlt.max(), dlt.mean())


def test_state_dict_and_checkpoint_roundtrip(golden_dir, tmp_path):
    p = os.path.join(golden_dir, "small_ae.npz")
    g, h = np.load(p), load_hierarchy(p)
    m, om = make_models(h, g, 16)
    opt = torch.optim.Adam(m.parameters(), lr=1e-3, weight_decay=5e-5)
    sched = torch.optim.lr_scheduler.StepLR(opt, 1, gamma=0.99)
    ck = {"epoch": 3, "autoencoder_state_dict": m.cpu().state_dict(), "optimizer_state_dict": opt.state_dict(),
          "scheduler_state_dict": sched.state_dict()}                      # train_funcs.py:562-567 layout
    torch.save(ck, tmp_path / "checkpoint3.pth.tar")
    m.to(dev())
    ld = torch.load(tmp_path / "checkpoint3.pth.tar", map_location="cpu")
    om.load_state_dict(ld["autoencoder_state_dict"])                        # oracle == reference names
    m2, _ = make_models(h, None, 16)
    m2.load_state_dict(ld["autoencoder_state_dict"])
    x = torch.from_numpy(g["x"]).to(dev())
    assert torch.equal(m2(x)[0], m(x)[0])


def test_full_size_6890_vs_oracle_and_reference_probe(golden_dir):
    """6890-vertex template (BASELINE config sizes), B=2: against the oracle on the same
    inputs, and against the probe of the REFERENCE's own output stored in the fixture."""
    p = os.path.join(golden_dir, "template6890.npz")
    g, h = np.load(p), load_hierarchy(p)
    from semantichuman_amd import synthetic
    m, om = make_models(h, None, 256)
    import math
    with torch.no_grad():                       # the closed-form weights the fixture was made with
        for j, (name, prm) in enumerate(om.named_parameters()):
            fan_in = prm.shape[1] if prm.dim() == 2 else prm.shape[0]
            prm.copy_(torch.from_numpy(synthetic.closed_form_fill(tuple(prm.shape), 1.0 / math.sqrt(fan_in), 0.37 + 0.011 * j, 0.1 * j)))
    m.load_state_dict(om.state_dict())
    x = torch.from_numpy(synthetic.synth_batch(h.verts, 2, seed=0))
    xd = x.to(dev())
    x_hat, z = m(xd)
    close(z, g["z_B2"], FWD_TOL, "z vs reference")
    close(x_hat[:, ::97], g["x_hat_B2_probe"], FWD_TOL, "x_hat probe vs reference")
    xo, zo = om(x)
    close(x_hat, xo, FWD_TOL, "x_hat vs oracle")
    sh.l1_loss(xd, x_hat).backward()
    torch.nn.functional.l1_loss(x, xo).backward()
    gmax = max(float(po.grad.abs().max()) for po in om.parameters())
    for (name, prm), po in zip(m.named_parameters(), om.parameters()):
        close(prm.grad, po.grad, GRAD_TOL, "grad " + name, floor=1e-6 * gmax)


def test_batch64_properties(golden_dir):
    """BASELINE batch size (64) at 6890 vertices, through size-independent properties:
    per-sample independence (sample b of a batch == that sample alone, bitwise), determinism
    (two runs bitwise equal, no atomics), dummy row stays exactly zero."""
    p = os.path.join(golden_dir, "template6890.npz")
    h = load_hierarchy(p)
    from semantichuman_amd import synthetic
    torch.manual_seed(0)
    m, _ = make_models(h, None, 256)
    x = torch.from_numpy(synthetic.synth_batch(h.verts, 64, seed=1)).to(dev())
    x_hat, z = m(x)
    x_hat2, z2 = m(x)
    assert torch.equal(x_hat, x_hat2) and torch.equal(z, z2)
    assert float(x_hat[:, -1].abs().max()) == 0.0
    with torch.no_grad():
        h1 = m._enc_stack.run_forward(x[5:6].contiguous(), "bm", "bm", [c.conv.weight for c in m.conv],
                                      [c.conv.bias for c in m.conv], keep=False)[0]
        h64 = m._enc_stack.run_forward(x, "bm", "bm", [c.conv.weight for c in m.conv], [c.conv.bias for c in m.conv],
                                       keep=False)[0]
    assert torch.equal(h1[0], h64[5])            # conv stack: no cross-sample coupling, same summation order
    sh.l1_loss(x, x_hat).backward()
    g1 = [prm.grad.clone() for prm in m.parameters()]
    m.zero_grad()
    xh3, _ = m(x)
    sh.l1_loss(x, xh3).backward()
    for a, (name, prm) in zip(g1, m.named_parameters()):
        # every gradient, the latent FCs' included, comes from the library's kernels: fixed-order slab / split-K
        # reductions, no atomics -> bitwise reproducible run to run
        assert torch.equal(a, prm.grad), name


@pytest.mark.parametrize("batch", [3, 64])
def test_native_stack_sequencing_equals_call_by_call(golden_dir, batch, monkeypatch):
    """sh_stack_forward / sh_stack_backward (one library call per stack and direction) issue the same launches as
    the call-by-call Python sequencing: outputs, input gradient and every parameter gradient bitwise equal."""
    from semantichuman_amd import stack as stack_mod, synthetic
    p = os.path.join(golden_dir, "template6890.npz")
    h = load_hierarchy(p)
    torch.manual_seed(3)
    m, _ = make_models(h, None, 256)
    x = torch.from_numpy(synthetic.synth_batch(h.verts, batch, seed=2)).to(dev())

    def run(native):
        monkeypatch.setattr(stack_mod, "NATIVE", native)
        m.zero_grad(set_to_none=True)
        xi = x.clone().requires_grad_(True)
        x_hat, z = m(xi)
        (sh.l1_loss(x, x_hat) + 0.1 * z.square().mean()).backward()
        with torch.no_grad():
            x_eval = m(x)[0]
        return [x_hat.detach().clone(), z.detach().clone(), xi.grad.clone(), x_eval] + [q.grad.clone() for q in m.parameters()]

    from semantichuman_amd import _lib
    n0 = _lib.load().sh_p3_launch_count()
    a = run(True)
    planes = _lib.load().sh_p3_launch_count() != n0    # the plane kernels exist in the native sequencer only: the call-by-call
    b = run(False)                                     # path serves a planes3 call with the split3 kernels (same six products per
    names = ["x_hat", "z", "dx", "x_hat (no grad)"] + [n for n, _ in m.named_parameters()]     # element, other summation order)
    gmax = max(float(q.abs().max()) for q in b[4:])
    for name, u, v in zip(names, a, b):
        if planes:
            close(u, v, FWD_TOL if name in ("x_hat", "z", "x_hat (no grad)") else GRAD_TOL, name, floor=0.0 if name in names[:4] else 1e-6 * gmax)
        else:
            assert torch.equal(u, v), name           # the FC kernels are outside the stacks: identical launches both ways


def test_vae_branch(golden_dir):
    """VAE_flag=True (reference models.py:82-83, 131-136; off in every shipped config): fc_latent_enc has 2 * nz
    outputs, z = z_mu + eps * exp(z_var / 2).  Checked in the deterministic limit (log-variance forced to -200:
    z == z_mu, which must equal the plain model with the first nz rows of the weight) and statistically
    (z_var == 0: z - z_mu is unit normal)."""
    p = os.path.join(golden_dir, "small_ae.npz")
    g, h = np.load(p), load_hierarchy(p)
    m, om = make_models(h, g, 16)
    torch.manual_seed(5)
    mv = sh.SpiralAutoencoder(FE, FD, 16, h.sizes, h.spiral_sizes, h.spirals, h.D, h.U, dev(), VAE_flag=True)
    assert mv.fc_latent_enc.weight.shape[0] == 32
    sd = m.state_dict()
    with torch.no_grad():
        for k, v in mv.state_dict().items():
            if not k.startswith("fc_latent_enc"):
                v.copy_(sd[k])
        mv.fc_latent_enc.weight[:16].copy_(sd["fc_latent_enc.weight"])
        mv.fc_latent_enc.bias[:16].copy_(sd["fc_latent_enc.bias"])
        mv.fc_latent_enc.weight[16:].zero_()
        mv.fc_latent_enc.bias[16:].fill_(-200.0)
    x = torch.from_numpy(g["x"]).to(dev())
    x_hat, z = mv(x)
    close(z, g["z"], FWD_TOL, "z (zero variance) vs reference")
    close(x_hat, g["x_hat"], FWD_TOL, "x_hat (zero variance) vs reference")
    assert torch.equal(mv.z_mu, z) and float(mv.z_var.max()) == -200.0
    with torch.no_grad():
        mv.fc_latent_enc.bias[16:].zero_()                     # unit variance
    zs = torch.stack([mv.encode(x, True) - mv.z_mu for _ in range(200)])
    assert abs(float(zs.mean())) < 0.05 and abs(float(zs.std()) - 1.0) < 0.05
    assert mv.encode(x, False).shape[1] == 32                  # the flag argument overrides the attribute (models.py:115)


def test_fused_recon_loss_equals_separate_terms(golden_dir):
    """recon_loss = l1_loss + w * edge_ratio_loss: same values (fixed-order sums) and the same gradient."""
    p = os.path.join(golden_dir, "small_ae.npz")
    g, h = np.load(p), load_hierarchy(p)
    x = torch.from_numpy(g["x"]).to(dev())
    xh0 = torch.from_numpy(g["x_hat"]).to(dev())
    ft = sh.FaceTables(h.faces, h.sizes[0] + 1, dev())
    a = xh0.clone().requires_grad_(True)
    total, parts = sh.recon_loss(a, x, ft, 1e-2)
    (total * 3.0).backward()
    b = xh0.clone().requires_grad_(True)
    rec, edge = sh.l1_loss(x, b), sh.edge_ratio_loss(b, x, ft)
    ((rec + 1e-2 * edge) * 3.0).backward()
    assert parts[0].item() == pytest.approx(rec.item(), rel=1e-6) and parts[1].item() == pytest.approx(edge.item(), rel=1e-6)
    assert total.item() == pytest.approx((rec + 1e-2 * edge).item(), rel=1e-6)
    assert parts[0].item() == pytest.approx(float(g["loss_rec"]), rel=1e-5) and parts[1].item() == pytest.approx(float(g["loss_edge"]), rel=1e-5)
    close(a.grad, b.grad.cpu().numpy(), 1e-6, "fused loss gradient")


def test_zero_length_reconstructed_edge(golden_dir):
    """The one edge case of train_funcs.py:30-39 where the library deviates from the reference ON PURPOSE (include/sh_kernels.h,
    DESIGN 4f): a reconstructed edge of length exactly 0.  Reference / oracle: d sqrt(sum(d^2)) at d = 0 is inf * 0 - the gradient
    of that batch entry's vertices is NaN.  Library (both the separate edge-ratio kernels and the fused reconstruction loss): the
    loss value is the reference's (|0 / t - 1| = 1 for the collapsed edge), the collapsed edge contributes NO gradient, every other
    term's gradient is the reference's - checked against the oracle's gradient of the same loss with the collapsed edge's term left out."""
    h = load_hierarchy(os.path.join(golden_dir, "small_ae.npz"))
    rs = np.random.RandomState(5)
    B, N1 = 3, h.sizes[0] + 1
    x = torch.from_numpy(np.concatenate([rs.randn(B, N1 - 1, 3), np.zeros((B, 1, 3))], 1).astype(np.float32))
    xh = x + 0.05 * torch.from_numpy(rs.randn(B, N1, 3).astype(np.float32))
    xh[:, -1] = 0
    f0 = np.asarray(h.faces)[0]
    i, j = int(f0[0]), int(f0[1])
    xh[1, j] = xh[1, i]                                             # batch entry 1: edge (i, j) of face 0 collapses in the reconstruction
    # --- the reference's formulation (oracle): NaN gradient
    xo = xh.clone().requires_grad_(True)
    lo = ref_cpu.edge_ratio_loss(xo, x, h.faces)
    lo.backward()
    assert torch.isfinite(lo) and torch.isnan(xo.grad[1]).any() and torch.isfinite(xo.grad[0]).all()
    # --- the oracle with the collapsed edges' terms left out of the graph: what every other term contributes
    faces = torch.as_tensor(np.asarray(h.faces), dtype=torch.long)
    xr = xh.clone().requires_grad_(True)
    tot = 0.0
    for a, b in ((0, 1), (1, 2), (0, 2)):
        d = xr[:, faces[:, a]] - xr[:, faces[:, b]]
        live = (d.detach() ** 2).sum(2) > 0
        ln = torch.sqrt(torch.where(live[..., None], d, torch.ones_like(d)).pow(2).sum(2))
        t = torch.sqrt(((x[:, faces[:, a]] - x[:, faces[:, b]]) ** 2).sum(2)) + 0.00001
        tot = tot + torch.where(live, torch.abs(ln / t - 1), torch.ones_like(ln))      # a collapsed edge: |0 / t - 1| = 1, no gradient
    lref = tot.mean(dim=1).mean()
    lref.backward()
    assert lref.item() == pytest.approx(lo.item(), rel=1e-6) and torch.isfinite(xr.grad).all()
    # --- the library
    d_ = dev()
    ft = sh.FaceTables(h.faces, N1, d_)
    xd = x.to(d_)
    a = xh.to(d_).requires_grad_(True)
    le = sh.edge_ratio_loss(a, xd, ft)
    le.backward()
    assert le.item() == pytest.approx(lo.item(), rel=2e-6)
    assert torch.isfinite(a.grad).all()
    close(a.grad, xr.grad, GRAD_TOL, "edge-ratio gradient with a collapsed edge")
    b = xh.to(d_).requires_grad_(True)
    total, parts = sh.recon_loss(b, xd, ft, 1.0)
    total.backward()
    l1 = xh.clone().requires_grad_(True)
    torch.nn.functional.l1_loss(x, l1).backward()
    assert torch.isfinite(b.grad).all()
    close(b.grad, xr.grad + l1.grad, GRAD_TOL, "fused loss gradient with a collapsed edge")


def test_c_abi_error_contract():
    """Bad calls return a negative status and leave a message in sh_last_error(); the Python wrappers turn that into
    RuntimeError - nothing fails silently and nothing falls back."""
    import ctypes
    from semantichuman_amd import _lib
    lib = _lib.load()
    d = dev()
    x = torch.zeros((2, 9, 4), device=d)
    table = torch.zeros((9, 3), dtype=torch.int32, device=d)
    w = torch.zeros((4, 12), device=d)
    y = torch.zeros((2, 9, 4), device=d)
    st = _lib.stream_ptr()
    # null pointer -> SH_ERR_INVALID_ARG
    assert lib.sh_spiral_conv_fwd(None, 4, 36, _lib.ptr(table), _lib.ptr(w), None, _lib.ptr(y), 4, 36, 2, 9, 3, 4, 4, 2, 8, 0, st) == -1
    assert b"null" in lib.sh_last_error()
    # unknown activation id
    assert lib.sh_spiral_conv_fwd(_lib.ptr(x), 4, 36, _lib.ptr(table), _lib.ptr(w), None, _lib.ptr(y), 4, 36, 2, 9, 3, 4, 4, 17, 8, 0, st) == -1
    # a spiral longer than the kernels are built for -> SH_ERR_UNSUPPORTED, message names the limit
    assert lib.sh_spiral_conv_fwd(_lib.ptr(x), 4, 36, _lib.ptr(table), _lib.ptr(w), None, _lib.ptr(y), 4, 36, 2, 9, 65, 4, 4, 2, 8, 0, st) == -2
    assert b"64" in lib.sh_last_error()
    # workspace too small -> SH_ERR_WORKSPACE
    need = lib.sh_spiral_conv_bwd_wgt_workspace(2, 9, 3, 4, 4)
    assert need > 0
    ws = torch.zeros(4, device=d)
    dW = torch.zeros((4, 12), device=d)
    rc = lib.sh_spiral_conv_bwd_wgt(_lib.ptr(y), 4, 36, _lib.ptr(x), 4, 36, _lib.ptr(table), _lib.ptr(dW), None, _lib.ptr(ws),
                                    ctypes.c_size_t(16), 2, 9, 3, 4, 4, 0, st)
    assert rc == -3 and b"workspace" in lib.sh_last_error()
    # whole-stack entry points: empty step table, channel mismatch between consecutive steps, missing output buffer
    assert lib.sh_stack_forward(0, None, _lib.ptr(x), 1, 9, 4, 2, None, None, None, 1, 0, None, None, 1, st) == -1
    steps = (_lib.StackStep * 1)()
    steps[0].kind, steps[0].param, steps[0].table = 0, 0, table.data_ptr()
    steps[0].R, steps[0].S, steps[0].n_in, steps[0].cin, steps[0].cout, steps[0].act, steps[0].zero_row = 9, 3, 9, 8, 4, 2, 8
    wp, outs = (ctypes.c_void_p * 1)(w.data_ptr()), (ctypes.c_void_p * 1)(0)
    assert lib.sh_stack_forward(1, steps, _lib.ptr(x), 1, 9, 4, 2, wp, None, outs, 1, 0, None, None, 1, st) == -1
    assert b"channels" in lib.sh_last_error()
    steps[0].cin = 4
    assert lib.sh_stack_forward(1, steps, _lib.ptr(x), 1, 9, 4, 2, wp, None, outs, 1, 0, None, None, 1, st) == -1
    assert b"output buffer" in lib.sh_last_error()
    # round 6, the three-plane weight gradient: a shape it does not take, a workspace that is too small, an odd number of 16-row
    # units without a zero row to complete it, an unknown plan kind in the deferred reduction
    img = torch.zeros(1 << 20, dtype=torch.uint8, device=d)
    t7 = torch.zeros((7, 3), dtype=torch.int32, device=d)
    assert lib.sh_spiral_conv_bwd_wgt_p3_ok(20, 7, 3, 32, 32) == 0 and lib.sh_spiral_conv_bwd_wgt_p3_workspace(20, 7, 3, 32, 32) == 0
    assert lib.sh_spiral_conv_bwd_wgt_p3(_lib.ptr(img), -1, _lib.ptr(img), _lib.ptr(t7), _lib.ptr(img), ctypes.c_size_t(1 << 20), 20, 7, 3, 32, 32, st) == -2
    assert b"not taken" in lib.sh_last_error()
    need3 = lib.sh_spiral_conv_bwd_wgt_p3_workspace(16, 7, 3, 32, 32)
    assert need3 > 0 and lib.sh_spiral_conv_bwd_wgt_p3_ok(16, 7, 3, 32, 32) == 1
    assert lib.sh_spiral_conv_bwd_wgt_p3(_lib.ptr(img), 6, _lib.ptr(img), _lib.ptr(t7), _lib.ptr(img), ctypes.c_size_t(16), 16, 7, 3, 32, 32, st) == -3
    assert lib.sh_spiral_conv_bwd_wgt_p3(_lib.ptr(img), -1, _lib.ptr(img), _lib.ptr(t7), _lib.ptr(img), ctypes.c_size_t(1 << 30), 16, 7, 3, 32, 32, st) == -2
    assert b"odd number of units" in lib.sh_last_error()
    ws3 = torch.zeros(need3, dtype=torch.uint8, device=d)
    assert lib.sh_spiral_conv_bwd_wgt_p3(_lib.ptr(img), 6, _lib.ptr(img), _lib.ptr(t7), _lib.ptr(ws3), ctypes.c_size_t(need3), 16, 7, 3, 32, 32, st) == 0
    one = lambda v, ct: (ct * 1)(v)      # noqa: E731
    dW3 = torch.zeros((32, 96), device=d)
    kinds_bad = one(3, ctypes.c_int)
    args = [one(ws3.data_ptr(), ctypes.c_void_p), one(dW3.data_ptr(), ctypes.c_void_p), one(0, ctypes.c_void_p)] + \
        [one(v, ctypes.c_int) for v in (16, 7, 3, 32, 32)]
    assert lib.sh_spiral_conv_bwd_wgt_reduce_multi_kinds(1, *[ctypes.cast(a, ctypes.c_void_p) for a in args], ctypes.cast(kinds_bad, ctypes.c_void_p), st) == -1
    assert b"plan kind" in lib.sh_last_error()
    assert lib.sh_spiral_conv_bwd_wgt_reduce_multi_kinds(1, *[ctypes.cast(a, ctypes.c_void_p) for a in args], ctypes.cast(one(2, ctypes.c_int), ctypes.c_void_p), st) == 0
    torch.cuda.synchronize()
    # the typed wrappers raise
    with pytest.raises(RuntimeError, match="status -2"):             # a spiral longer than the kernels' 64-entry table lines
        ops.spiral_conv_fwd(x, "bm", torch.zeros((9, 65), dtype=torch.int32, device=d), torch.zeros((8, 65 * 4), device=d), None,
                            torch.zeros((2, 9, 8), device=d), "bm", 9, 65, 2, 8)
    torch.cuda.synchronize()


@pytest.mark.parametrize("cfg", [("template6890.npz", 64), ("template6890.npz", 48), ("template27554.npz", 32)])
def test_bench_launch_table_matches_the_library(golden_dir, cfg):
    """bench.py prices the roofline with a table (pass, rows, K, channels) -> algorithmic FLOPs built from the model's
    layers; here every conv-family launch the library's profiler reports for one training step must find its entry, and
    every entry must be hit by exactly one launch, so a dispatch change cannot silently detach the roofline from the
    kernels - in both arithmetic forms of the fp32 products."""
    import bench
    from semantichuman_amd import _lib, synthetic
    name, B = cfg
    h = load_hierarchy(os.path.join(golden_dir, name))
    torch.manual_seed(1)
    m = sh.SpiralAutoencoder(FE, FD, 256, h.sizes, h.spiral_sizes, h.spirals, h.D, h.U, dev())
    x = torch.from_numpy(synthetic.synth_batch(h.verts, B, seed=1)).to(dev())

    def step():
        m.zero_grad(set_to_none=True)
        sh.l1_loss(x, m(x)[0]).backward()
    was = _lib.get_f32_mma_mode()
    try:
        for mode in ("exact", "split3", "planes3"):
            _lib.set_f32_mma_mode(mode)
            step()
            torch.cuda.synchronize()
            _lib.profile_enable(True)
            step()
            torch.cuda.synchronize()
            recs = _lib.profile_records_by_kernel()
            _lib.profile_enable(False)
            table = bench.f32_work_table(m, B)
            hit = {}
            for kname, shape, _ms in recs:
                if kname.startswith(("gather_gemm", "wgrad", "conv_out3", "conv_p3")):
                    key = bench.parse_tag_f32(kname, shape)
                    if key is None and "pass=" in shape:                   # an earlier launch of a multi-pass layer: priced with the last
                        continue
                    assert key in table, (mode, kname, shape)
                    hit[id(table[key])] = hit.get(id(table[key]), 0) + 1
            assert hit == {id(v): 1 for v in table.values()}, mode
            # the streaming launches are priced too (bench.hbm_work_table), and `roofline` is the step's largest kernel name
            hb = bench.hbm_work_table(m, B)
            for kname, shape, _ms in recs:
                if kname.startswith("spmm_kernel"):
                    assert bench.parse_tag_hbm(kname, shape) in hb, (mode, kname, shape)
            rf = bench.roofline_f32(recs, m, B, 1, h.sizes[0])
            assert rf["roofline"]["kernel"] == rf["kernel_breakdown"][0]["kernel"], mode
            assert rf["roofline"]["frac"] is not None and 0 < rf["roofline"]["frac"] < 1, (mode, rf["roofline"])
            for e in rf["kernel_breakdown"]:
                if e["kernel"].startswith("conv_p3") or "split3" in e["kernel"]:
                    assert e["peak_tflops"] == pytest.approx(2500.0 / 6), e
    finally:
        _lib.set_f32_mma_mode(was)


@pytest.mark.gpu
@pytest.mark.parametrize("B,N1,C,act", [(64, 171, 3, "identity"), (5, 37, 3, "elu"), (33, 100, 8, "tanh"), (64, 6891, 3, "identity")])
def test_act_backward_batch_major_to_vertex_major(B, N1, C, act):
    """The tile-turning form of sh_act_backward (batch-major gradient / output, vertex-major result, few channels - the
    stack's last layer) is the element-wise definition dpre = dy * act'(y) with the dummy row forced to zero: exact."""
    torch.manual_seed(3)
    dy, y = torch.randn(B, N1, C), torch.randn(B, N1, C)
    a = ops.act_id(act)
    dpre = torch.full((N1 + 2, B, C), float("nan"), device=dev())              # extra rows must stay untouched
    ops.act_backward(dy.to(dev()), "bm", y.to(dev()), "bm", dpre, "vm", N1, a, N1 - 1)
    yd = y.double()
    der = {"identity": torch.ones_like(yd), "elu": torch.where(yd > 0, torch.ones_like(yd), yd + 1), "tanh": 1 - yd * yd}[act]
    ref = (dy * der.float()).permute(1, 0, 2).clone()
    ref[N1 - 1] = 0
    assert torch.equal(dpre[:N1].cpu(), ref)
    assert torch.isnan(dpre[N1:]).all()


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_folded_upsampling_is_bitwise_the_unfolded_one(golden_dir, dtype):
    """stack.fold_identity_rows (identity rows of U served from the coarse tensor through a composed gather table, U
    appends only the blended rows): output and every parameter gradient of a training step are bit-identical to the stack
    built without the fold, on both compute paths."""
    from semantichuman_amd import stack as stack_mod, synthetic
    h = load_hierarchy(os.path.join(golden_dir, "small_ae.npz"))
    x = torch.from_numpy(synthetic.synth_batch(h.verts, 32, seed=1)).to(dev())
    ft = sh.FaceTables(h.faces, h.sizes[0] + 1, dev())
    res = {}
    for fold in (True, False):
        old = stack_mod.FOLD_U
        stack_mod.FOLD_U = fold
        try:
            torch.manual_seed(0)
            m = sh.SpiralAutoencoder(FE, FD, 32, h.sizes, h.spiral_sizes, h.spirals, h.D, h.U, dev())
        finally:
            stack_mod.FOLD_U = old
        assert any(s.kind == "spmm" and s.extend for s in m._dec_stack.steps) == fold
        m.set_compute_dtype(dtype)
        xh = m(x)[0]
        sh.recon_loss(xh, x, ft, 1e-2)[0].backward()
        res[fold] = (xh.detach().clone(), {n: p.grad.clone() for n, p in m.named_parameters()})
    assert torch.equal(res[True][0], res[False][0])
    for n, g in res[True][1].items():
        assert torch.equal(g, res[False][1][n]), n


# the optional second conv of a level (reference models.py:72-75: `filters_enc[1][i]` truthy -> an extra SpiralConv in front of the
# level's main one; :96-99: `filters_dec[1][i+1]` truthy -> one behind it) - the default configuration only uses the decoder's last
FE2 = [[3, 16, 32, 64, 128], [[], 16, 32, [], []]]
FD2 = [[128, 64, 32, 32, 16], [[], 64, [], 32, 3]]


@pytest.mark.parametrize("cfg", [("small_ae.npz", 16), ("template6890.npz", 2)])
def test_second_conv_per_level_vs_oracle(golden_dir, cfg):
    """A model WITH second convs on several levels (encoder levels 1 and 2: 16 -> 16 -> 32 and 32 -> 32 -> 64, the level's
    row-select D folding into the LAST conv of the level only; decoder levels 3 and 1: 128 -> 64 -> 64 and 32 -> 32 -> 32 behind
    the up-sampling): forward, loss and every parameter gradient against the oracle's statement of models.py:69-113, at batch
    16 (all three arithmetic forms can engage) and at full size."""
    name, B = cfg
    p = os.path.join(golden_dir, name)
    h = load_hierarchy(p)
    from semantichuman_amd import synthetic
    S, D, U = h.dense_constants()
    torch.manual_seed(11)
    om = ref_cpu.SpiralAEOracle(FE2, FD2, 32, h.sizes, h.spiral_sizes, S, D, U)
    m = sh.SpiralAutoencoder(FE2, FD2, 32, h.sizes, h.spiral_sizes, h.spirals, h.D, h.U, dev())
    assert len(m.conv) == 6 and len(m.dconv) == 7 and [k for k in m.state_dict()] == [k for k in om.state_dict()]
    m.load_state_dict(om.state_dict())
    x = torch.from_numpy(synthetic.synth_batch(h.verts, B, seed=3))
    xd = x.to(dev())
    x_hat, z = m(xd)
    xo, zo = om(x)
    close(z, zo, FWD_TOL, "z")
    close(x_hat, xo, FWD_TOL, "x_hat")
    assert float(x_hat[:, -1].abs().max()) == 0.0
    sh.l1_loss(xd, x_hat).backward()
    torch.nn.functional.l1_loss(x, xo).backward()
    gmax = max(float(po.grad.abs().max()) for po in om.parameters())
    for (n, prm), po in zip(m.named_parameters(), om.parameters()):
        close(prm.grad, po.grad, GRAD_TOL, "grad " + n, floor=1e-6 * gmax)

"""Host-side table construction (mesh_ops, stack planning) checked on CPU against the oracle."""
import os

import numpy as np
import pytest
import torch

from oracle import ref_cpu
from semantichuman_amd import mesh_ops, models, synthetic
from semantichuman_amd.hierarchy import load_hierarchy
from semantichuman_amd.losses import FaceTables
from semantichuman_amd.stack import ConvStep
from tests import emulate

FE = [[3, 16, 32, 64, 128], [[], [], [], [], []]]
FD = [[128, 64, 32, 32, 16], [[], [], [], [], 3]]


@pytest.fixture(scope="module")
def small(golden_dir):
    return np.load(os.path.join(golden_dir, "small_ae.npz")), load_hierarchy(os.path.join(golden_dir, "small_ae.npz"))


def test_spirals_to_table_wraps_minus_one(small):
    g, h = small
    t = mesh_ops.spirals_to_table(h.spirals[1][None])
    n1 = h.sizes[1] + 1
    assert t.dtype == np.int32 and t.min() >= 0 and t.max() == n1 - 1
    assert np.array_equal(t[h.spirals[1] >= 0], h.spirals[1][h.spirals[1] >= 0])
    assert np.all(t[h.spirals[1] == -1] == n1 - 1) and np.all(t[-1] == n1 - 1)
    with pytest.raises(IndexError):
        mesh_ops.spirals_to_table(np.array([[[0, 5]]]))


def test_csr_roundtrip_and_transpose(small):
    g, h = small
    for m in h.U + h.D:
        dense = m.todense()
        back = mesh_ops.dense_to_csr(dense[None])
        assert np.array_equal(back.todense(), dense)
        assert np.array_equal(m.transpose().todense(), dense.T)
    assert all(d.is_row_select() for d in h.D) and not any(u.is_row_select() for u in h.U)
    for u in h.U:
        assert np.diff(u.rowptr).max() <= 3
        np.testing.assert_allclose(u.todense().sum(1), 1.0, atol=1e-6)
        assert u.todense()[-1, -1] == 1.0                    # padded dummy entry (main.py:190-191)


def test_empty_and_ragged_rows():
    d = np.zeros((4, 5), np.float32)
    d[0, 1] = 2; d[0, 4] = -1; d[3, 0] = 0.5           # rows 1,2 empty
    c = mesh_ops.dense_to_csr(d)
    assert list(c.rowptr) == [0, 2, 2, 2, 3] and np.array_equal(c.todense(), d)
    t = c.transpose()
    assert t.rows == 5 and np.array_equal(t.todense(), d.T)
    x = np.arange(10.0).reshape(5, 1, 2)
    np.testing.assert_allclose(emulate.spmm(c, x)[:, 0], d.astype(np.float64) @ x[:, 0])


def test_transpose_table_brute_force(small):
    g, h = small
    table = mesh_ops.spirals_to_table(h.spirals[2][None])
    n_in, S = table.shape
    gl = mesh_ops.transpose_table(table, n_in)
    assert gl.ptr[-1] == table.size
    for u in range(n_in):
        for s in range(S):
            want = np.nonzero(table[:, s] == u)[0]
            assert np.array_equal(gl.src[gl.ptr[u * S + s]:gl.ptr[u * S + s + 1]], want)
    gl2 = mesh_ops.transpose_table(table, n_in, skip_row=n_in - 1)
    assert gl2.ptr[-1] == (table != n_in - 1).sum()
    assert gl2.ptr[(n_in - 1) * S] == gl2.ptr[-1]


def test_dense_transposed_table(small):
    """table_t + pre-summed extra rows == the scatter-add the reference's autograd performs."""
    g, h = small
    table = mesh_ops.spirals_to_table(h.spirals[1][None])
    R, S = table.shape
    st = ConvStep(param=0, table=table, n_in=R, cin=4, cout=5, act=2).finalize()
    tt = st.tt
    assert tt.table_t.shape == (R, S) and tt.table_t.dtype == np.int32
    assert tt.n1 > 0 and tt.n2 > 0                        # the dummy row is read hundreds of times per position
    assert np.diff(tt.csr1.rowptr).max() <= 16 and tt.csr2.rows == tt.n2 and tt.csr2.cols == R + tt.n1
    assert tt.table_t.max() == R + tt.n_extra - 1
    rs = np.random.RandomState(0)
    dpre = rs.randn(R, 2, 5)
    dpre[st.zero_row] = 0                                  # contract: the "no source" row of dpre is zero
    W = rs.randn(5, S * 4)
    got = emulate.conv_bwd_data(emulate.extend_dpre(dpre, tt), tt.table_t, W, 4)
    want = np.zeros((R, 2, 4))                             # brute-force scatter-add
    for r in range(R):
        for s in range(S):
            want[table[r, s]] += dpre[r] @ W[:, s * 4:(s + 1) * 4]
    np.testing.assert_allclose(got, want, rtol=1e-12, atol=1e-12)
    # a dead dummy row gets no sources at all
    tt2 = mesh_ops.transpose_table_dense(table, R, none_row=R - 1, skip_row=R - 1)
    assert np.all(tt2.table_t[R - 1] == R - 1) and tt2.n1 == 0
    # no multi-entry list at all -> no extra rows
    ident = np.arange(6, dtype=np.int32).reshape(6, 1)
    tt3 = mesh_ops.transpose_table_dense(ident, 6, none_row=5)
    assert tt3.n_extra == 0 and np.array_equal(tt3.table_t, ident)


def test_conv_layout_matches_reference_keys(small):
    g, h = small
    enc, dec = models.conv_layout(FE, FD, h.spiral_sizes, "elu")
    keys = [str(k) for k in g["state_dict_keys"]]
    for j, (cin, S, cout, act, lvl) in enumerate(enc):
        assert g["w0/conv.%d.conv.weight" % j].shape == (cout, cin * S)
    for j, (cin, S, cout, act, lvl) in enumerate(dec):
        assert g["w0/dconv.%d.conv.weight" % j].shape == (cout, cin * S)
    assert dec[-1][3] == "identity" and all(l[3] == "elu" for l in enc + dec[:-1])
    assert len(enc) + len(dec) == sum(k.endswith("conv.weight") for k in keys)
    # optional second convolution per level (models.py:72-75, :96-99)
    enc2, dec2 = models.conv_layout([[3, 16, 32, 64, 128], [8, [], 48, [], []]], [[128, 64, 32, 32, 16], [[], 64, [], [], 3]],
                                    h.spiral_sizes, "relu")
    assert [(l[0], l[2]) for l in enc2] == [(3, 8), (8, 16), (16, 32), (32, 48), (48, 64), (64, 128)]
    assert [(l[0], l[2], l[3]) for l in dec2] == [(128, 64, "relu"), (64, 64, "relu"), (64, 32, "relu"), (32, 32, "relu"),
                                                  (32, 16, "relu"), (16, 3, "identity")]
    o = ref_cpu.layer_plan([[3, 16, 32, 64, 128], [8, [], 48, [], []]], [[128, 64, 32, 32, 16], [[], 64, [], [], 3]],
                           h.spiral_sizes, "relu")
    assert [tuple(l) for l in o[0]] == [tuple(l) for l in enc2] and [tuple(l) for l in o[1]] == [tuple(l) for l in dec2]


def test_module_state_dict_is_drop_in(small):
    g, h = small
    import semantichuman_amd as sh
    m = sh.SpiralAutoencoder(FE, FD, 16, h.sizes, h.spiral_sizes, h.spirals, h.D, h.U, None)
    assert list(m.state_dict().keys()) == [str(k) for k in g["state_dict_keys"]]
    m.load_state_dict({k[3:]: torch.from_numpy(g[k]) for k in g.files if k.startswith("w0/")})      # strict
    with pytest.raises(NotImplementedError):
        sh.SpiralConv(3, 4, 5, activation="gelu")
    with pytest.raises(RuntimeError):                        # no silent CPU fallback
        m(torch.from_numpy(g["x"]))


def test_stack_formulation_equals_oracle_autograd(small):
    """Fused row-select, transposed lists, dead-dummy skipping, epilogue placement: the numpy
    emulation of the kernel chain must reproduce the oracle's outputs and autograd gradients."""
    g, h = small
    import semantichuman_amd as sh
    S, D, U = h.dense_constants()
    om = ref_cpu.SpiralAEOracle(FE, FD, 16, h.sizes, h.spiral_sizes, S, D, U).double()
    sd = {k[3:]: torch.from_numpy(g[k]).double() for k in g.files if k.startswith("w0/")}
    om.load_state_dict(sd)
    m = sh.SpiralAutoencoder(FE, FD, 16, h.sizes, h.spiral_sizes, h.spirals, h.D, h.U, None)
    x = torch.from_numpy(g["x"]).double().requires_grad_(True)
    for stack, convs, inp in ((m._enc_stack, "conv", x),):
        W = [sd["%s.%d.conv.weight" % (convs, j)].numpy() for j in range(len(getattr(m, convs)))]
        Bs = [sd["%s.%d.conv.bias" % (convs, j)].numpy() for j in range(len(getattr(m, convs)))]
        xin = x.detach().numpy().transpose(1, 0, 2)                       # vertex-major
        acts = emulate.stack_forward(stack, xin, W, Bs)
        # oracle encoder activations
        cur, j = x, 0
        for lvl in range(4):
            cur = ref_cpu.spiral_conv(cur, S[lvl], om.conv[j].conv.weight, om.conv[j].conv.bias, "elu")
            cur = torch.matmul(D[lvl].double(), cur)
            np.testing.assert_allclose(acts[j].transpose(1, 0, 2), cur.detach().numpy(), rtol=1e-10, atol=1e-12)
            j += 1
        gy = torch.from_numpy(np.random.RandomState(1).randn(*cur.shape))
        om.zero_grad()
        (cur * gy).sum().backward()
        gx, grads = emulate.stack_backward(stack, xin, acts, gy.numpy().transpose(1, 0, 2).copy(), W)
        np.testing.assert_allclose(gx.transpose(1, 0, 2), x.grad.numpy(), rtol=1e-9, atol=1e-12)
        for j in range(4):
            np.testing.assert_allclose(grads[j][0], om.conv[j].conv.weight.grad.numpy(), rtol=1e-9, atol=1e-11)
            np.testing.assert_allclose(grads[j][1], om.conv[j].conv.bias.grad.numpy(), rtol=1e-9, atol=1e-11)
    # decoder, including the LIVE dummy row of its input
    hin = torch.from_numpy(np.random.RandomState(2).randn(2, h.sizes[-1] + 1, 128)).requires_grad_(True)
    W = [sd["dconv.%d.conv.weight" % j].numpy() for j in range(5)]
    Bs = [sd["dconv.%d.conv.bias" % j].numpy() for j in range(5)]
    hv = hin.detach().numpy().transpose(1, 0, 2)
    acts = emulate.stack_forward(m._dec_stack, hv, W, Bs)
    cur, j = hin, 0
    for lvl in (3, 2, 1, 0):
        cur = torch.matmul(U[lvl].double(), cur)
        for _ in range(2 if lvl == 0 else 1):
            cur = ref_cpu.spiral_conv(cur, S[lvl], om.dconv[j].conv.weight, om.dconv[j].conv.bias, om.dconv[j].act)
            j += 1
    np.testing.assert_allclose(acts[-1].transpose(1, 0, 2), cur.detach().numpy(), rtol=1e-9, atol=1e-12)
    gy = torch.from_numpy(np.random.RandomState(3).randn(*cur.shape))
    om.zero_grad()
    (cur * gy).sum().backward()
    gx, grads = emulate.stack_backward(m._dec_stack, hv, acts, gy.numpy().transpose(1, 0, 2).copy(), W)
    assert np.abs(hin.grad.numpy()[:, -1]).max() > 0                       # the dummy-row gradient is live here
    np.testing.assert_allclose(gx.transpose(1, 0, 2), hin.grad.numpy(), rtol=1e-9, atol=1e-12)
    for j in range(5):
        np.testing.assert_allclose(grads[j][0], om.dconv[j].conv.weight.grad.numpy(), rtol=1e-9, atol=1e-11)
        np.testing.assert_allclose(grads[j][1], om.dconv[j].conv.bias.grad.numpy(), rtol=1e-9, atol=1e-11)
    dead = [st.dead_dummy_grad for st in m._dec_stack.steps if st.kind == "conv"]
    assert dead == [False, True, True, True, True]


def test_barycentric_upsample_reproduces_vertices():
    v, f = synthetic.box_sphere(6, 6, 4)
    U = mesh_ops.barycentric_upsample(v, f, v)               # same mesh: every vertex maps to itself
    np.testing.assert_allclose(U.todense() @ v, v, atol=1e-6)
    p = (v[f[:, 0]] + v[f[:, 1]] + v[f[:, 2]]) / 3            # face centres lie on the surface
    Uc = mesh_ops.barycentric_upsample(v, f, p)
    np.testing.assert_allclose(Uc.todense() @ v, p, atol=1e-6)
    np.testing.assert_allclose(Uc.todense().sum(1), 1.0, atol=1e-6)


def test_face_tables():
    v, f = synthetic.box_sphere(3, 3, 2)
    ft = FaceTables(f, v.shape[0] + 1, "cpu")
    vptr, vc = ft.vptr.numpy(), ft.vcorner.numpy()
    assert vptr[-1] == f.size and vptr[v.shape[0]] == vptr[-1]          # dummy row has no corners
    for vert in range(v.shape[0]):
        corners = vc[vptr[vert]:vptr[vert + 1]]
        assert np.all(f.ravel()[corners] == vert) and len(corners) == (f == vert).sum()


def test_synthetic_batch_contract():
    v, f = synthetic.box_sphere(42, 42, 20)
    assert v.shape == (6890, 3) and f.shape == (13776, 3)
    x = synthetic.synth_batch(v, 3, seed=0)
    assert x.shape == (3, 6891, 3) and x.dtype == np.float32 and np.all(x[:, -1] == 0)
    assert np.array_equal(x, synthetic.synth_batch(v, 3, seed=0))


def test_split_identity_rows_and_folded_stack(small):
    """mesh_ops.split_identity_rows: Z = [x ; U_b x] read through row_map is U x, bit for bit; and a decoder stack built
    with the fold has the same step outputs (float64 emulation) as one built without it."""
    from semantichuman_amd import stack as stack_mod
    _, h = small
    rs = np.random.RandomState(0)
    for U in h.U:
        row_map, u_b, m = mesh_ops.split_identity_rows(U)
        assert u_b.rows < U.rows and m.rows == U.cols + u_b.rows and m.cols == U.cols
        x = rs.randn(U.cols, 3, 4).astype(np.float32)
        full = U.todense() @ x.reshape(U.cols, -1)
        z = np.concatenate([x.reshape(U.cols, -1), u_b.todense() @ x.reshape(U.cols, -1)], 0)
        assert np.array_equal(z[row_map], full)
        assert np.array_equal(m.todense() @ x.reshape(U.cols, -1), z)
    # the same model with and without the fold
    outs = {}
    for fold in (True, False):
        old = stack_mod.FOLD_U
        stack_mod.FOLD_U = fold
        try:
            torch.manual_seed(0)
            m_ = models.SpiralAutoencoder(FE, FD, 8, h.sizes, h.spiral_sizes, h.spirals, h.D, h.U, torch.device("cpu"))
        finally:
            stack_mod.FOLD_U = old
        st = m_._dec_stack
        assert any(s.kind == "spmm" and s.extend for s in st.steps) == fold
        W = [c.conv.weight.detach().double().numpy() for c in m_.dconv]
        Bs = [c.conv.bias.detach().double().numpy() for c in m_.dconv]
        hv = np.random.RandomState(1).randn(h.sizes[-1] + 1, 2, FD[0][0])
        outs[fold] = emulate.stack_forward(st, hv, W, Bs)[-1]
    assert np.array_equal(outs[True], outs[False])


@pytest.mark.parametrize("act", ["identity", "relu", "elu", "leaky_relu", "sigmoid", "tanh"])
def test_emulate_operators_equal_the_oracle(act):
    """tests/emulate.py is a third formulation of the operators (numpy float64 over the kernels' tables) that the bf16
    and thin-layer GPU tests compare against.  Here it is pinned, operator by operator and on the same inputs, to the oracle
    (oracle/ref_cpu.py, itself pinned to the reference's vectors): SpiralConv forward against ref_cpu.spiral_conv, the three
    backward formulations (transposed-table backward-data with pre-summed rows, weight gradient, role-swapped weight
    gradient) against torch autograd of that same oracle function - float64 emulation vs the oracle's float32 at 1e-6."""
    from semantichuman_amd import ops
    rs = np.random.RandomState(11)
    B, N1, S, cin, cout = 3, 41, 5, 8, 6
    adj = rs.randint(-1, N1 - 1, size=(1, N1, S)).astype(np.int64)          # -1 = the dummy row, several readers per row
    adj[0, :, 0] = np.arange(N1)
    adj[0, -1, :] = -1                                                       # the dummy row's own spiral (utils_spiral.py:89)
    x = torch.from_numpy(rs.randn(B, N1, cin).astype(np.float32)); x[:, -1] = 0
    W = torch.from_numpy((rs.randn(cout, S * cin) * 0.3).astype(np.float32)).requires_grad_(True)
    b = torch.from_numpy(rs.randn(cout).astype(np.float32)).requires_grad_(True)
    xr = x.clone().requires_grad_(True)
    y = ref_cpu.spiral_conv(xr, torch.from_numpy(adj), W, b, act)            # the oracle (models.py:34-53)
    g = torch.from_numpy(rs.randn(B, N1, cout).astype(np.float32))
    y.backward(g)

    table = mesh_ops.spirals_to_table(adj)
    a = ops.act_id(act)
    xv = x.permute(1, 0, 2).double().numpy()                                 # vertex-major, as the kernels see it
    ye = emulate.conv_fwd(xv, table, W.detach().double().numpy(), b.detach().double().numpy(), a, N1 - 1)
    yo = y.detach().permute(1, 0, 2).numpy()
    tol = 1e-6 * max(1.0, float(np.abs(yo).max()))
    assert np.abs(ye - yo).max() <= tol

    # backward: dpre = g * act'(y) with the dummy row dead, then the three formulations
    dpre = g.permute(1, 0, 2).double().numpy() * emulate.DACT[a](ye)
    dpre[N1 - 1] = 0
    tt = mesh_ops.transpose_table_dense(table, N1, none_row=N1 - 1)
    dx = emulate.conv_bwd_data(emulate.extend_dpre(dpre, tt), tt.table_t, W.detach().double().numpy(), cin)
    gx = xr.grad.permute(1, 0, 2).numpy()
    assert np.abs(dx - gx).max() <= 1e-6 * max(1.0, float(np.abs(gx).max()))
    dW, db = emulate.conv_bwd_wgt(dpre, xv, table)
    assert np.abs(dW - W.grad.numpy()).max() <= 2e-6 * max(1.0, float(W.grad.abs().max()))
    assert np.abs(db - b.grad.numpy()).max() <= 2e-6 * max(1.0, float(b.grad.abs().max()))
    dW2, db2 = emulate.conv_bwd_wgt_swapped(emulate.extend_dpre(dpre, tt), xv, tt.table_t)
    assert np.abs(dW2 - W.grad.numpy()).max() <= 2e-6 * max(1.0, float(W.grad.abs().max()))
    assert np.abs(db2 - b.grad.numpy()).max() <= 2e-6 * max(1.0, float(b.grad.abs().max()))


def test_semantic_loop_bookkeeping_helpers_on_cpu():
    """train_semantic.weighted_sum (tensor-op fallback off the GPU) is the `loss = loss + w * term` chain, and _SplitRows
    is row slicing whose backward is one concatenation: values and gradients as the plain forms."""
    from semantichuman_amd import train_semantic as ts
    vals = [torch.tensor(v, requires_grad=True) for v in (0.731, 12.5, 3e-4)]
    ref_vals = [v.detach().clone().requires_grad_(True) for v in vals]
    ws = (1.0, 1e-2, 0.37)
    tot = ts.weighted_sum(list(zip(ws, vals)))
    ref = ref_vals[0] + ws[1] * ref_vals[1]
    ref = ref + ws[2] * ref_vals[2]
    assert torch.equal(tot.detach(), ref.detach())
    tot.backward(); ref.backward()
    assert all(torch.equal(a.grad, b.grad) for a, b in zip(vals, ref_vals))
    x = torch.randn(7, 5, 3, requires_grad=True)
    xr = x.detach().clone().requires_grad_(True)
    a, b, c = ts._SplitRows.apply(x, 2, 4, 1)
    assert torch.equal(a, xr[:2]) and torch.equal(b, xr[2:6]) and torch.equal(c, xr[6:])
    (a.sum() * 2.0 + (b * b).sum()).backward()                    # the last piece gets no gradient: zeros in the concatenation
    (xr[:2].sum() * 2.0 + (xr[2:6] * xr[2:6]).sum()).backward()
    assert torch.equal(x.grad, xr.grad)
    y = torch.randn(6, 4, requires_grad=True)                      # rows beyond the listed pieces: zero gradient
    (p,) = ts._SplitRows.apply(y, 4)
    p.sum().backward()
    assert torch.equal(y.grad[:4], torch.ones(4, 4)) and torch.equal(y.grad[4:], torch.zeros(2, 4))


def test_corrupt_checkpoint_is_not_reported_as_a_weights_only_refusal(tmp_path):
    """torch.load on a corrupt file raises pickle.UnpicklingError('invalid load key ...'); load_checkpoint must
    let it through unchanged instead of suggesting trust_pickle=True (which would unpickle - i.e. execute - the file)."""
    import pickle

    import torch

    from semantichuman_amd import train_funcs
    for k, blob in enumerate((b"this is not a checkpoint", b"\x00\x01garbage" * 7, b"")):
        bad = tmp_path / ("corrupt%d.pth.tar" % k)
        bad.write_bytes(blob)
        with pytest.raises(Exception) as ei:                     # whatever the unpickler raises on these bytes, unchanged
            train_funcs.load_checkpoint(str(bad), torch.nn.Linear(2, 2))
        assert "trust_pickle" not in str(ei.value), ei.value
    del pickle


class _NotAllowed:                                   # a global the weights-only unpickler refuses
    def __init__(self):
        self.v = 3


def test_checkpoint_with_pickled_objects_needs_trust_pickle(tmp_path):
    """The other half: a well-formed checkpoint that holds a pickled object IS the weights_only refusal - reported with the
    trust_pickle hint, loaded when the caller vouches for the file."""
    import torch

    from semantichuman_amd import train_funcs
    m = torch.nn.Linear(2, 2)
    p = tmp_path / "ck.pth.tar"
    torch.save({"epoch": 4, "autoencoder_state_dict": m.state_dict(), "extra": _NotAllowed()}, p)
    with pytest.raises(RuntimeError, match="trust_pickle"):
        train_funcs.load_checkpoint(str(p), torch.nn.Linear(2, 2))
    assert train_funcs.load_checkpoint(str(p), torch.nn.Linear(2, 2), trust_pickle=True) == 5


def test_ragged_source_lists_are_the_dense_transposed_table_with_its_presums():
    """mesh_ops.transpose_table_ragged (round 6; the plane backward-data kernel's list form): for every input row exactly the
    (output row, position) pairs that read it - ordered by position, then row; the dead dummy row empty; padding marked -1 -
    and a backward-data pass evaluated over the lists equals the one over the dense transposed table with its pre-summed rows."""
    import numpy as np
    from semantichuman_amd import mesh_ops
    rs = np.random.RandomState(3)
    R, S, n_in, Co, Ci = 61, 5, 43, 4, 3
    table = rs.randint(0, n_in, size=(R, S)).astype(np.int32)
    table[:, 3] = n_in - 1                                    # a dummy row with R readers at one position
    for skip in (-1, n_in - 1):
        rag = mesh_ops.transpose_table_ragged(table, n_in, none_row=R - 1, skip_row=skip, max_len=200)
        assert rag is not None
        rows, pos = rag
        for u in range(n_in):
            want = [] if u == skip else [(r, s) for s in range(S) for r in range(R) if table[r, s] == u]
            got = [(int(rows[u, j]), int(pos[u, j])) for j in range(rows.shape[1]) if pos[u, j] >= 0]
            assert want == got, u
            assert (pos[u, len(got):] == -1).all() and (rows[u, len(got):] == R - 1).all()
        # the numbers: dx[u] = sum_j dpre[rows[u, j]] @ W[pos[u, j]]  ==  sum_s dpre_ext[table_t[u, s]] @ W[s]
        tt = mesh_ops.transpose_table_dense(table, n_in, none_row=R - 1, skip_row=skip)
        dpre = rs.randn(R, Co)
        dpre[R - 1] = 0                                        # the "no source" row
        W = rs.randn(S, Co, Ci)
        ext = np.concatenate([dpre, np.zeros((tt.n_extra, Co))])
        if tt.csr1 is not None:
            for k in range(tt.n1):
                ext[R + k] = ext[tt.csr1.col[tt.csr1.rowptr[k]:tt.csr1.rowptr[k + 1]]].sum(0)
        if tt.csr2 is not None:
            for k in range(tt.n2):
                ext[R + tt.n1 + k] = ext[tt.csr2.col[tt.csr2.rowptr[k]:tt.csr2.rowptr[k + 1]]].sum(0)
        dense = sum(ext[tt.table_t[:, s]] @ W[s] for s in range(S))
        lists = np.zeros((n_in, Ci))
        for j in range(rows.shape[1]):
            ok = pos[:, j] >= 0
            lists[ok] += np.einsum("uc,uci->ui", dpre[rows[ok, j]], W[pos[ok, j]])
        assert np.allclose(dense, lists, rtol=1e-12, atol=1e-12)
    assert mesh_ops.transpose_table_ragged(table, n_in, none_row=R - 1, skip_row=-1, max_len=16) is None      # the long list: dense form


def test_grouped_lists_cover_every_source_exactly_once():
    """mesh_ops.group_lists (round 6; csrc/p3_conv.hip conv_p3g_kernel): every output row is a member of exactly one group; the
    entries of a group that a member reads are exactly that row's (source row, position) list - multiplicities included - and an
    evaluation over the groups equals the one over the one-row lists; unions stay within the cap; on a real hierarchy the union is
    markedly shorter than the members' lists together (what the kernel's gather shrinks by)."""
    import os
    import numpy as np
    from semantichuman_amd import mesh_ops
    from semantichuman_amd.hierarchy import load_hierarchy
    rs = np.random.RandomState(5)

    def check(rows, pos, members, max_len=64):
        g = mesh_ops.group_lists(rows, pos, members=members, max_len=max_len)
        assert g is not None
        g_rows, g_pos, g_out = g
        n = rows.shape[0]
        seen = np.zeros(n, dtype=int)
        x = rs.randn(int(rows.max()) + 1, 3)
        W = rs.randn(int(pos.max()) + 1, 3, 2)
        ref = np.zeros((n, 2))
        for i in range(n):
            for r, q in zip(rows[i], pos[i]):
                if q >= 0:
                    ref[i] += x[r] @ W[q]
        got = np.zeros((n, 2))
        for k in range(g_rows.shape[0]):
            mem = [int(v) for v in g_out[k] if v >= 0]
            assert 1 <= len(mem) <= members and (g_out[k, len(mem):] == -1).all()
            n_ent = int((g_pos[k] != 0xFFFFFFFF).sum())
            assert n_ent <= max_len and (g_pos[k, n_ent:] == 0xFFFFFFFF).all() and (g_pos[k, :n_ent] != 0xFFFFFFFF).all()
            for m, i in enumerate(mem):
                seen[i] += 1
                mine = sorted((int(g_rows[k, j]), int((g_pos[k, j] >> (8 * m)) & 0xFF)) for j in range(n_ent) if (g_pos[k, j] >> (8 * m)) & 0xFF != 0xFF)
                want = sorted((int(r), int(q)) for r, q in zip(rows[i], pos[i]) if q >= 0)
                assert mine == want, (k, m, i)
                for j in range(n_ent):
                    q = (int(g_pos[k, j]) >> (8 * m)) & 0xFF
                    if q != 0xFF:
                        got[i] += x[g_rows[k, j]] @ W[q]
            for m in range(len(mem), 4):
                assert all((int(g_pos[k, j]) >> (8 * m)) & 0xFF == 0xFF for j in range(n_ent))
        assert (seen == 1).all()
        assert np.allclose(got, ref, rtol=1e-12, atol=1e-12)
        return g

    # random lists: ragged lengths, a row read at two positions by one member, a row with very many readers
    n, L = 57, 9
    rows = rs.randint(0, 40, size=(n, L)).astype(np.int32)
    pos = np.tile(np.arange(L, dtype=np.int32), (n, 1))
    pos[rs.rand(n, L) < 0.2] = -1
    rows[:, 4] = 39
    rows[3, 0] = rows[3, 1]
    for members in (2, 4):
        check(rows, pos, members)
        check(rows, pos, members, max_len=12)               # a tight cap: groups stay smaller
    assert mesh_ops.group_lists(rows, pos, members=4, max_len=4) is None       # a single list longer than the cap
    # a real hierarchy: forward tables of every level and the ragged backward lists of one
    h = load_hierarchy(os.path.join(os.path.dirname(__file__), "golden", "template6890.npz"))
    for lvl in (1, 3):
        t = mesh_ops.spirals_to_table(h.spirals[lvl])
        p = np.ascontiguousarray(np.broadcast_to(np.arange(t.shape[1], dtype=np.int32), t.shape))
        g_rows, g_pos, g_out = check(t, p, 4)
        assert int((g_pos != 0xFFFFFFFF).sum()) * 1.6 < t.size, ((g_pos != 0xFFFFFFFF).sum(), t.size)
    t = mesh_ops.spirals_to_table(h.spirals[2])
    rag = mesh_ops.transpose_table_ragged(t, t.shape[0], none_row=t.shape[0] - 1, skip_row=t.shape[0] - 1)
    check(rag[0], rag[1], 2)

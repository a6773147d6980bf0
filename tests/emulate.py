"""numpy (float64) emulation of the FORMULATION the HIP kernels use - gather tables, fused
row-select, dense transposed tables with pre-summed extra rows, CSR re-sampling.  It lets the
CPU test suite prove that the tables built by semantichuman_amd.mesh_ops / stack are right
(against the oracle's autograd) without a GPU.  Test infrastructure only."""
import numpy as np

ACT = {0: lambda v: v, 1: lambda v: np.maximum(v, 0), 2: lambda v: np.where(v > 0, v, np.expm1(np.minimum(v, 0))),
       3: lambda v: np.where(v > 0, v, 0.02 * v), 4: lambda v: 1 / (1 + np.exp(-v)), 5: np.tanh}
DACT = {0: lambda y: np.ones_like(y), 1: lambda y: (y > 0).astype(y.dtype), 2: lambda y: np.where(y > 0, 1.0, y + 1.0),
        3: lambda y: np.where(y > 0, 1.0, 0.02), 4: lambda y: y * (1 - y), 5: lambda y: 1 - y * y}


def conv_fwd(x, table, W, b, act, zero_row):
    """x [n_in,B,Cin] (vertex-major) -> y [R,B,Cout]"""
    R, S = table.shape
    G = x[table]                                        # [R,S,B,Cin]
    G = G.transpose(0, 2, 1, 3).reshape(R, x.shape[1], -1)
    y = ACT[act](G @ W.T + (0 if b is None else b))
    if zero_row >= 0:
        y[zero_row] = 0
    return y


def extend_dpre(dpre, tt):
    """Append the pre-summed extra rows of a mesh_ops.TransposedTable (what the two sh_spmm
    launches in stack.run_backward write behind the R real rows)."""
    ext = dpre
    if tt.csr1 is not None:
        ext = np.concatenate([ext, spmm(tt.csr1, ext)], 0)
    if tt.csr2 is not None:
        ext = np.concatenate([ext, spmm(tt.csr2, ext)], 0)
    return ext


def conv_bwd_data(dpre_ext, table_t, W, cin):
    """dx[u] = sum_s dpre_ext[table_t[u,s]] . W[:, s*cin:(s+1)*cin]  - the forward formulation
    over the transposed table (sh_spiral_conv_bwd_data)."""
    n_in, S = table_t.shape
    dx = np.zeros((n_in, dpre_ext.shape[1], cin))
    for s in range(S):
        dx += dpre_ext[table_t[:, s]] @ W[:, s * cin:(s + 1) * cin]
    return dx


def conv_bwd_wgt(dpre, x, table):
    R, S = table.shape
    G = x[table].transpose(0, 2, 1, 3).reshape(R * x.shape[1], -1)
    P = dpre[:R].reshape(R * x.shape[1], -1)
    return P.T @ G, P.sum(0)


def conv_bwd_wgt_swapped(dpre_ext, x, table_t):
    """The same weight gradient over the transposed table (sh_spiral_conv_bwd_wgt_thin):
    dW[co, s*cin + ci] = sum_{u,b} x[u,b,ci] * dpre_ext[table_t[u,s],b,co]; the bias sum runs over the first n_in rows."""
    n_in, S = table_t.shape
    B, cin, cout = x.shape[1], x.shape[2], dpre_ext.shape[2]
    G = dpre_ext[table_t]                                   # [n_in, S, B, cout]
    dW = np.einsum("ubi,usbo->osi", x, G).reshape(cout, S * cin)
    return dW, dpre_ext[:n_in].reshape(n_in * B, cout).sum(0)


def spmm(csr, x):
    y = np.zeros((csr.rows,) + x.shape[1:])
    for r in range(csr.rows):
        for e in range(csr.rowptr[r], csr.rowptr[r + 1]):
            y[r] += csr.val[e] * x[csr.col[e]]
    return y


def stack_forward(stack, x, weights, biases):
    acts, cur = [], x
    for st in stack.steps:
        if st.kind == "conv":
            cur = conv_fwd(cur, st.table, weights[st.param], biases[st.param], st.act, st.zero_row)
        else:
            cur = spmm(st.csr, cur)
        acts.append(cur)
    return acts


def stack_backward(stack, x, acts, g, weights):
    """Mirror of semantichuman_amd.stack.Stack.run_backward in numpy. -> (gx, {param: (dW, db)})"""
    steps, grads = stack.steps, {}
    last = len(steps) - 1
    st = steps[last]
    cur = g * DACT[st.act](acts[last]) if st.kind == "conv" else g
    if st.kind == "conv":
        cur[st.zero_row] = 0
    for i in range(last, -1, -1):
        st = steps[i]
        inp = x if i == 0 else acts[i - 1]
        prev = steps[i - 1] if i > 0 else None
        if st.kind == "conv":
            grads[st.param] = conv_bwd_wgt(cur, inp, st.table)
            g_in = conv_bwd_data(extend_dpre(cur[:st.R], st.tt), st.tt.table_t, weights[st.param], st.cin)
        else:
            g_in = spmm(st.csr_t, cur)
        if prev is not None and prev.kind == "conv":
            g_in = g_in * DACT[prev.act](acts[i - 1])
            g_in[prev.zero_row] = 0
        cur = g_in
    return cur, grads

"""TEST INFRASTRUCTURE - the parity oracle.  Never imported by the product path
(`semantichuman_amd/`); only `tests/`, `__graft_entry__.smoke()` and
`bench.py`'s `cpu_baseline` leg may use it.

A CPU (PyTorch fp32/fp64) restatement of the reference's spiral-convolution
autoencoder hot path *in the reference's own formulation*: advanced-index
gather -> dense linear -> activation -> dummy-row mask; dense matmul for the
D / U re-sampling; mean-abs loss.  These are the same ATen ops the reference
dispatches, so the numbers are the reference's numbers and the
time is the reference's CPU time.

Pinned against the reference itself: `oracle/gen_golden.py` imports
`/root/reference/models.py` etc. in the build container and stores its outputs
in `tests/golden/*.npz`; `tests/test_oracle_golden.py` checks every function
here against those vectors.

Reference lines followed:
  spiral_conv            models.py:34-53
  SpiralAEOracle.encode  models.py:115-137   (VAE branch unused, Appendix C)
  SpiralAEOracle.decode  models.py:139-154
  l1 loss                train_funcs.py:501  (F.l1_loss(tx, tx_hat), all N+1 rows)
  edge_ratio_loss        train_funcs.py:12-39, 503-508
  eval_metrics           test_funcs.py:41-49
  bone_lengths, girths, plane_ring   utils_SH.py:86-161
  dataset_item           autoencoder_dataset.py:26-50
"""
from __future__ import annotations

import torch
import torch.nn as nn
import torch.nn.functional as F

ACTIVATIONS = {
    "relu": torch.relu,
    "elu": F.elu,
    "leaky_relu": lambda t: F.leaky_relu(t, 0.02),
    "sigmoid": torch.sigmoid,
    "tanh": torch.tanh,
    "identity": lambda t: t,
}


def spiral_conv(x, spiral_adj, weight, bias, activation="elu"):
    """x [B,N+1,Cin]; spiral_adj int64 [B|1,N+1,S] (-1 = dummy row);
    weight [Cout, S*Cin] (column k = s*Cin + c); -> [B,N+1,Cout], last row 0."""
    if activation not in ACTIVATIONS:
        raise NotImplementedError(activation)
    B, N1, C = x.shape
    S = spiral_adj.shape[-1]
    nbr = spiral_adj.reshape(-1, N1, S)[0]                 # shared by the whole batch
    gathered = x[:, nbr.reshape(-1), :]                     # negative index wraps to row N
    gathered = gathered.reshape(B * N1, S * C)
    out = ACTIVATIONS[activation](F.linear(gathered, weight, bias)).reshape(B, N1, -1)
    keep = torch.ones(1, N1, 1, dtype=x.dtype, device=x.device)
    keep[0, N1 - 1, 0] = 0
    return out * keep


def resample(M, x):
    """D / U applied to a batch [B, rows_in, C] (models.py:127,148: `torch.matmul(D[i], x)` with D dense [1, R, rows_in],
    broadcast over the batch).  A dense operand is multiplied exactly as the reference does.  For templates whose dense
    operators do not fit a test box (27 554 vertices: 1.5 GB per level-0 matrix, 0.4 TFLOP per product) the same operator
    may be handed over as a 2-D torch sparse tensor (`sparse_operator`): the product is then taken over the batch folded
    into the columns - the same <= 3 terms per output element (the dense product's other terms are exact zeros);
    tests/test_oracle_golden.py pins this form to the dense one."""
    if M.layout == torch.strided:
        return torch.matmul(M, x)
    B, n, C = x.shape
    y = torch.sparse.mm(M, x.permute(1, 0, 2).reshape(n, B * C))
    return y.reshape(M.shape[0], B, C).permute(1, 0, 2)


def sparse_operator(csr):
    """A semantichuman_amd.mesh_ops.CSR-like object (rows, cols, rowptr, col, val) -> 2-D torch sparse COO tensor."""
    import numpy as np
    r = np.repeat(np.arange(csr.rows, dtype=np.int64), np.diff(csr.rowptr))
    idx = torch.from_numpy(np.stack([r, csr.col.astype(np.int64)]))
    return torch.sparse_coo_tensor(idx, torch.from_numpy(csr.val.astype(np.float32)), (csr.rows, csr.cols)).coalesce()


def layer_plan(filters_enc, filters_dec, spiral_sizes, activation="elu"):
    """The (in_c, S, out_c, act) sequence the reference constructor builds
    (models.py:69-113).  Returns (enc_layers, dec_layers); each entry also
    carries the mesh level it runs at."""
    n_lvl = len(spiral_sizes) - 1
    enc, c = [], filters_enc[0][0]
    for i in range(n_lvl):
        if filters_enc[1][i]:
            enc.append((c, spiral_sizes[i], filters_enc[1][i], activation, i))
            c = filters_enc[1][i]
        enc.append((c, spiral_sizes[i], filters_enc[0][i + 1], activation, i))
        c = filters_enc[0][i + 1]
    dec, c = [], filters_dec[0][0]
    for i in range(n_lvl):
        lvl = n_lvl - 1 - i
        S = spiral_sizes[-2 - i]
        last = i == n_lvl - 1
        extra = filters_dec[1][i + 1]
        if not last:
            dec.append((c, S, filters_dec[0][i + 1], activation, lvl)); c = filters_dec[0][i + 1]
            if extra:
                dec.append((c, S, extra, activation, lvl)); c = extra
        elif extra:
            dec.append((c, S, filters_dec[0][i + 1], activation, lvl)); c = filters_dec[0][i + 1]
            dec.append((c, S, extra, "identity", lvl)); c = extra
        else:
            dec.append((c, S, filters_dec[0][i + 1], "identity", lvl)); c = filters_dec[0][i + 1]
    return enc, dec


class _Conv(nn.Module):
    def __init__(self, cin, S, cout, act):
        super().__init__()
        self.conv = nn.Linear(cin * S, cout)       # -> state_dict key '<stack>.<j>.conv.weight'
        self.act = act


class SpiralAEOracle(nn.Module):
    """Plain spiral autoencoder with the reference's parameter names
     so state_dicts are interchangeable with
    reference models.SpiralAutoencoder."""

    def __init__(self, filters_enc, filters_dec, latent_size, sizes, spiral_sizes,
                 spirals, D, U, activation="elu"):
        super().__init__()
        self.sizes, self.spiral_sizes, self.latent_size = sizes, spiral_sizes, latent_size
        self.spirals, self.D, self.U = spirals, D, U
        self.enc_plan, self.dec_plan = layer_plan(filters_enc, filters_dec, spiral_sizes, activation)
        self.conv = nn.ModuleList([_Conv(c, S, o, a) for (c, S, o, a, _) in self.enc_plan])
        c_last = self.enc_plan[-1][2]
        self.fc_latent_enc = nn.Linear((sizes[-1] + 1) * c_last, latent_size)
        self.fc_latent_dec = nn.Linear(latent_size, (sizes[-1] + 1) * filters_dec[0][0])
        self.dconv = nn.ModuleList([_Conv(c, S, o, a) for (c, S, o, a, _) in self.dec_plan])

    def encode(self, x):
        n_lvl = len(self.spiral_sizes) - 1
        j = 0
        for lvl in range(n_lvl):
            while j < len(self.enc_plan) and self.enc_plan[j][4] == lvl:
                m = self.conv[j]
                x = spiral_conv(x, self.spirals[lvl], m.conv.weight, m.conv.bias, m.act)
                j += 1
            x = resample(self.D[lvl], x)                      # dense matmul, broadcast over batch
        return self.fc_latent_enc(x.reshape(x.shape[0], -1))

    def decode(self, z):
        n_lvl = len(self.spiral_sizes) - 1
        x = self.fc_latent_dec(z).reshape(z.shape[0], self.sizes[-1] + 1, -1)
        j = 0
        for lvl in range(n_lvl - 1, -1, -1):
            x = resample(self.U[lvl], x)
            while j < len(self.dec_plan) and self.dec_plan[j][4] == lvl:
                m = self.dconv[j]
                x = spiral_conv(x, self.spirals[lvl], m.conv.weight, m.conv.bias, m.act)
                j += 1
        return x

    def forward(self, x):
        z = self.encode(x)
        return self.decode(z), z


def edge_ratio_loss(x_hat, x, faces):
    """train_funcs.py:503-508 without the per-sample host loop: for every face
    and each of its 3 edges |len_rec / (len_gt + 1e-5) - 1|, summed over the 3
    edges, mean over faces, mean over the batch.  The reference computes the GT
    edge lengths in numpy from the fp32 vertices (get_target, :22-28); fp32
    torch gives the same values."""
    f = torch.as_tensor(faces, dtype=torch.long)

    def edge_lengths(p):
        a, b, c = p[:, f[:, 0]], p[:, f[:, 1]], p[:, f[:, 2]]
        return [torch.sqrt(((a - b) ** 2).sum(2)), torch.sqrt(((b - c) ** 2).sum(2)),
                torch.sqrt(((a - c) ** 2).sum(2))]
    tgt = [t.detach() + 0.00001 for t in edge_lengths(x)]
    rec = edge_lengths(x_hat)
    score = sum(torch.abs(r / t - 1) for r, t in zip(rec, tgt))     # [B, F]
    return score.mean(dim=1).mean()


def eval_metrics(x_hat, x, mm_constant=1000.0):
    """test_funcs.py:41-49 for one batch: (mean |x_hat-x|, mean per-vertex
    Euclidean error in mm), dummy row dropped."""
    a, b = x_hat[:, :-1], x[:, :-1]
    l1 = (a - b).abs().mean()
    l2 = torch.sqrt((((a - b) * mm_constant) ** 2).sum(2)).mean()
    return l1, l2


def train_step(model, optim, x, faces=None, edgereg_w=0.0):
    """One iteration of train_funcs.py:495-510 (plain loop)."""
    optim.zero_grad()
    x_hat, _ = model(x)
    loss = F.l1_loss(x, x_hat)
    if faces is not None and edgereg_w > 0:
        loss = loss + edgereg_w * edge_ratio_loss(x_hat, x, faces)
    loss.backward()
    optim.step()
    return loss.detach()


# =============================================================================================
# Semantic model and its losses - restated in the reference's formulation
# =============================================================================================
class SemanticAEOracle(nn.Module):
    """reference models.SpiralAutoencoder_multiz_partkps (models.py:166-310), same parameter names."""

    def __init__(self, kps_index_list, vert_part_index_dict, filters_enc, filters_dec, latent_size, part_kps_latent_size,
                 sizes, spiral_sizes, spirals, D, U, activation="elu"):
        super().__init__()
        self.sizes, self.spiral_sizes, self.spirals, self.D, self.U = sizes, spiral_sizes, spirals, D, U
        self.kps_index_list = kps_index_list
        self.parts = [torch.as_tensor(v, dtype=torch.long) for v in vert_part_index_dict.values()]
        self.enc_plan, self.dec_plan = layer_plan(filters_enc, filters_dec, spiral_sizes, activation)
        self.conv = nn.ModuleList([_Conv(c, S, o, a) for (c, S, o, a, _) in self.enc_plan])
        feat = self.enc_plan[-1][2]
        self.fc_latent_enc_list = nn.ModuleList([nn.Linear(len(v) * feat, latent_size) for v in self.parts])
        self.fc_latent_dec_list = nn.ModuleList([nn.Linear(latent_size + part_kps_latent_size, len(v) * filters_dec[0][0])
                                                 for v in self.parts])
        self.kps_enc_list = nn.ModuleList([nn.Linear(len(k) * 3, part_kps_latent_size) for k in kps_index_list])
        self.dconv = nn.ModuleList([_Conv(c, S, o, a) for (c, S, o, a, _) in self.dec_plan])

    def encode(self, x, kps):
        B, n_lvl, j = x.shape[0], len(self.spiral_sizes) - 1, 0
        for lvl in range(n_lvl):
            while j < len(self.enc_plan) and self.enc_plan[j][4] == lvl:
                m = self.conv[j]
                x = spiral_conv(x, self.spirals[lvl], m.conv.weight, m.conv.bias, m.act)
                j += 1
            x = torch.matmul(self.D[lvl], x)
        z = torch.stack([fc(x[:, idx, :].reshape(B, -1)) for idx, fc in zip(self.parts, self.fc_latent_enc_list)], dim=1)
        zk = torch.stack([fc(kps[:, idx, :].reshape(B, -1)) for idx, fc in zip(self.kps_index_list, self.kps_enc_list)], dim=1)
        return z, zk, x[:, -1:, :]

    def decode(self, z, zk, dummy):
        B, n_lvl = z.shape[0], len(self.spiral_sizes) - 1
        x = torch.cat([fc(torch.cat([z[:, k], zk[:, k]], dim=1)) for k, fc in enumerate(self.fc_latent_dec_list)], dim=1)
        x = x.view(B, self.sizes[-1], -1)
        order = torch.cat(self.parts)
        y = x.clone()
        y[:, order, :] = x[:, :order.shape[0], :]                       # models.py:270-272
        x = torch.cat([y, dummy], dim=1)
        j = 0
        for lvl in range(n_lvl - 1, -1, -1):
            x = torch.matmul(self.U[lvl], x)
            while j < len(self.dec_plan) and self.dec_plan[j][4] == lvl:
                m = self.dconv[j]
                x = spiral_conv(x, self.spirals[lvl], m.conv.weight, m.conv.bias, m.act)
                j += 1
        return x

    def forward(self, x, kps):
        z, zk, dummy = self.encode(x, kps)
        return self.decode(z, zk, dummy), z, zk


def dist_matrix(x):
    """utils_distance.py:366-376."""
    r = torch.sum(x ** 2, dim=2).unsqueeze(2)
    return F.relu(r - 2 * torch.bmm(x, x.transpose(2, 1)) + r.transpose(2, 1)) ** 0.5


def angle_degrees(v, bone):
    """utils_SH.py:442-478 for one part: v [B,n,3], bone [B,3] -> [B,n,n] degrees between
    (v_i - v_j) and the bone; NaN (i == j) -> cos 1 -> 0 degrees."""
    d = v[:, :, None, :] - v[:, None, :, :]
    dm = torch.sqrt((d * d).sum(-1))
    km = torch.sqrt((bone * bone).sum(-1))[:, None, None]
    cos = torch.abs((d * bone[:, None, None, :]).sum(-1) / (dm * km))
    cos = torch.where(torch.isnan(cos), torch.ones_like(cos), cos).clamp(0, 1)
    return torch.arccos(cos) * 180 / torch.pi


def bone_directions(kps, skl_list):
    return torch.stack([kps[:, b[0]] - (kps[:, b[1]] if len(b) == 2 else (kps[:, b[1]] + kps[:, b[2]]) / 2) for b in skl_list], 1)


def part_pairdist_loss(x_rec, x_gt, kps_gt, parts, skl_list, leaf=(0, 7, 10, 13, 16), scale=None, w_mode="threshold",
                       thr=0.8, relat=True):
    """train_funcs.py:243-284 (`interp`) / :353-389 (`exc`, scale=None), '1/K' part weights."""
    bones = bone_directions(kps_gt, skl_list)
    total = 0
    for i, idx in enumerate(parts):
        idx = torch.as_tensor(idx, dtype=torch.long)
        De, De_r = dist_matrix(x_gt[:, idx, :]), dist_matrix(x_rec[:, idx, :])
        if scale is not None:
            De = De * scale[:, i][:, None, None]
        if w_mode == "all_one" or i in leaf:
            w = torch.ones_like(De)
        else:
            ang = angle_degrees(x_gt[:, idx, :], bones[:, i])
            w = ang / 90 if w_mode in ("linear", "threshold") else torch.sin(ang / 180 * torch.pi)
            if w_mode == "threshold":
                w = torch.where(w < thr, torch.zeros_like(w), w)
        w = w * (1 - torch.eye(w.shape[1], dtype=w.dtype))[None]
        nz = torch.where((w * De) != 0)
        if relat:
            li = F.l1_loss(w[nz] * De_r[nz] / De[nz], w[nz])
        else:
            li = F.l1_loss(w[nz] * De_r[nz], w[nz] * De[nz])
        total = total + li / len(parts)
    return total


def part_volume_loss(x_rec, x_gt, faces, face_part_index, parts):
    """train_funcs.py:56-71 averaged over the batch (:323-329)."""
    f = torch.as_tensor(faces, dtype=torch.long)
    fpi = torch.as_tensor(face_part_index)
    total = 0
    for b in range(x_rec.shape[0]):
        for k in parts:
            tf = f[fpi == k]
            rv = torch.sum(torch.linalg.cross(x_rec[b, tf[:, 0]], x_rec[b, tf[:, 1]]) * x_rec[b, tf[:, 2]])
            gv = torch.sum(torch.linalg.cross(x_gt[b, tf[:, 0]], x_gt[b, tf[:, 1]]) * x_gt[b, tf[:, 2]])
            total = total + (torch.abs(rv / gv) - torch.abs(gv / gv)).abs() / len(parts)
    return total / x_rec.shape[0]


def bone_lengths(kps, skl_list):
    """utils_SH.py:86-98 cal_length: kps [K, 3] -> [len(skl_list)]."""
    out = torch.zeros(len(skl_list), dtype=kps.dtype)
    for i, s in enumerate(skl_list):
        tail = kps[s[1]] if len(s) == 2 else (kps[s[1]] + kps[s[2]]) / 2
        out[i] = torch.sqrt(torch.sum((kps[s[0]] - tail) ** 2))
    return out


def girths(v, factor_list, edge_point_index_list):
    """utils_SH.py:153-159 (measure_body_quick): sequential closed-polyline length of every ring of one mesh."""
    out = []
    for fac, epi in zip(factor_list, edge_point_index_list):
        fac = torch.as_tensor(fac, dtype=v.dtype)
        epi = torch.as_tensor(epi, dtype=torch.long)
        q = v[epi[:, 0], :] * (1 - fac) + v[epi[:, 1], :] * fac
        g = torch.sqrt(torch.sum((q[0] - q[-1]) ** 2))
        for i in range(q.shape[0] - 1):
            g = g + torch.sqrt(torch.sum((q[i] - q[i + 1]) ** 2))
        out.append(g)
    return torch.stack(out)


def plane_ring(face_point, face_normal, points):
    """utils_SH.py:100-142 cal_girth in the reference's formulation: one 3x3 solve per cut edge
    (rows: plane equation; the two symmetric equations of the edge's line), then the signed-angle sort."""
    n_pts = points.shape[0]
    A = torch.zeros((n_pts, 3, 3), dtype=points.dtype)
    Bv = torch.zeros((n_pts, 3), dtype=points.dtype)
    lp = points[:, 0, :]
    lo = points[:, 0, :] - points[:, 1, :]
    lo = torch.where(lo == 0, torch.full_like(lo, 1e-6), lo)
    A[:, 0, :] = face_normal[None]
    A[:, 1, 0], A[:, 1, 1] = 1 / lo[:, 0], -1 / lo[:, 1]
    A[:, 2, 0], A[:, 2, 2] = 1 / lo[:, 0], -1 / lo[:, 2]
    Bv[:, 0] = torch.sum(face_point * face_normal)
    Bv[:, 1] = lp[:, 0] / lo[:, 0] - lp[:, 1] / lo[:, 1]
    Bv[:, 2] = lp[:, 0] / lo[:, 0] - lp[:, 2] / lo[:, 2]
    X = torch.linalg.solve(A, Bv)
    Xv = X - X.mean(dim=0)
    m = torch.sqrt(torch.sum(Xv * Xv, dim=1))
    cos = torch.sum(Xv[0:1] * Xv[1:], dim=1) / (m[1:] * m[0])
    theta = torch.arccos(cos) / torch.pi * 180
    cr = torch.linalg.cross(Xv[0:1].expand(n_pts - 1, 3), Xv[1:])
    flag = torch.where(cr[:, 0] * cr[:, 1] * cr[:, 2] > 0, 1.0, -1.0).to(points.dtype)
    order = torch.sort(torch.cat((torch.zeros(1, dtype=points.dtype), theta * flag)))[1]
    g = torch.sqrt(torch.sum((X[order[0]] - X[order[-1]]) ** 2))
    for i in range(n_pts - 1):
        g = g + torch.sqrt(torch.sum((X[order[i]] - X[order[i + 1]]) ** 2))
    return g, X, order


def dataset_item(verts_init, normalization, J_regressor=None, shapedata=None, idx=0, dummy_node=True):
    """autoencoder_dataset.py:26-50: normalise one loaded sample [N, 3] and append the dummy node (numpy)."""
    import numpy as np
    v = np.array(verts_init)
    if "zeromean" in normalization:
        v = v - np.mean(v, axis=0)
    if "zeroroot" in normalization:
        v = v - np.matmul(J_regressor, v)[0]
    if "onelength" in normalization:
        v = v / (np.max(v, axis=0) - np.min(v, axis=0))[1] * 1.5
    if "small" in normalization:
        v = v / 1.5
    if "gass" in normalization:
        v = (v - shapedata.mean) / shapedata.std
    if "normal" in normalization:
        v = (v - shapedata.center[idx, :]) * shapedata.scale[idx]
    v[np.where(np.isnan(v))] = 0.0
    v = v.astype("float32")
    if dummy_node:
        v = np.concatenate([v, np.zeros((1, v.shape[1]), dtype=np.float32)], axis=0)
    return v

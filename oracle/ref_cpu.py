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
    keep = torch.ones(1, N1, 1, dtype=x.dtype)
    keep[0, N1 - 1, 0] = 0
    return out * keep


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
            x = torch.matmul(self.D[lvl], x)                  # dense, broadcast over batch
        return self.fc_latent_enc(x.reshape(x.shape[0], -1))

    def decode(self, z):
        n_lvl = len(self.spiral_sizes) - 1
        x = self.fc_latent_dec(z).reshape(z.shape[0], self.sizes[-1] + 1, -1)
        j = 0
        for lvl in range(n_lvl - 1, -1, -1):
            x = torch.matmul(self.U[lvl], x)
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

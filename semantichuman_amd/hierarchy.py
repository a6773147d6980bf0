"""Mesh hierarchy container: what reference main.py:93-205 assembles before building the model
(per-level spiral index arrays, down/up-sampling operators, faces), in sparse form.

On disk it is a flat .npz (tests/golden/template6890.npz was produced by the reference's own
QSlim + spiral generator through oracle/gen_golden.py).
"""
from __future__ import annotations

from dataclasses import dataclass

import numpy as np

from . import mesh_ops
from .mesh_ops import CSR


@dataclass
class Hierarchy:
    sizes: list            # vertices per level, e.g. [6890, 3445, 1723, 862, 431]
    spiral_sizes: list     # spiral length per level
    spirals: list          # int32 [N_l+1, S_l], -1 = padding, last row all -1 (utils_spiral.py:87-93)
    D: list                # CSR [N_{l+1}+1, N_l+1] incl. the dummy row/col (main.py:186-191)
    U: list                # CSR [N_l+1, N_{l+1}+1]
    verts: np.ndarray      # template vertices [N_0, 3] float64
    faces: np.ndarray      # int32 [F, 3]

    def dense_constants(self):
        """(spirals int64 [1,N+1,S], D, U dense fp32 [1,R,C]) torch tensors exactly as the
        reference prepares them (main.py:183-205) - for the CPU oracle."""
        import torch
        S = [torch.from_numpy(s.astype(np.int64))[None] for s in self.spirals]
        D = [torch.from_numpy(d.todense())[None] for d in self.D]
        U = [torch.from_numpy(u.todense())[None] for u in self.U]
        return S, D, U


def load_hierarchy(path: str) -> Hierarchy:
    g = np.load(path)
    sizes = [int(v) for v in g["sizes"]]
    spiral_sizes = [int(v) for v in g["spiral_sizes"]]
    levels = len(sizes) - 1
    spirals = [g["spirals_%d" % l].astype(np.int32) for l in range(levels + 1)]
    D, U = [], []
    for l in range(levels):
        sel = g["D_sel_%d" % l].astype(np.int32)
        d = CSR(sizes[l + 1], sizes[l], np.arange(sizes[l + 1] + 1, dtype=np.int32), sel,
                np.ones(sel.shape[0], dtype=np.float32))
        u = CSR(sizes[l], sizes[l + 1], g["U_rowptr_%d" % l].astype(np.int32), g["U_col_%d" % l].astype(np.int32),
                g["U_val_%d" % l].astype(np.float32))
        D.append(mesh_ops.pad_dummy(d))
        U.append(mesh_ops.pad_dummy(u))
    return Hierarchy(sizes, spiral_sizes, spirals, D, U, g["verts"], g["faces"].astype(np.int32))


def layout_spirals(adj_spirals, dilation=None, nb_stds=2):
    """The layout stage of the reference's spiral generation (utils_spiral.generate_spirals :58-93), applied to the
    per-vertex spiral LISTS of every level (`adj_spirals[l][j]` = spiral of vertex j, as get_spirals returns them):
      * dilation d_l: keep the centre and every d_l-th entry after it, `s[:1] + s[1::d_l]` (:58-66);
      * spiral size of a level: int(mean + nb_stds * std) of its spiral lengths (:70-82);
      * array float64 [1, N_l + 1, S_l] filled with -1, row j = the first S_l entries of spiral j, the extra last row
        (the dummy vertex) all -1 (:87-93).
    -> (list of arrays, list of sizes, dilated lists), the reference's return triple."""
    adj_spirals = [list(map(list, level)) for level in adj_spirals]
    if dilation:
        for i, dil in enumerate(dilation):
            adj_spirals[i] = [s[:1] + s[1::dil] for s in adj_spirals[i]]
    sizes = []
    for level in adj_spirals:
        L = np.array([len(s) for s in level])
        sizes.append(int(L.mean() + nb_stds * L.std()))
    out = []
    for level, size in zip(adj_spirals, sizes):
        S = np.zeros((1, len(level) + 1, size)) - 1
        for j, s in enumerate(level):
            S[0, j, :len(s)] = s[:size]
        out.append(S)
    return out, sizes, adj_spirals

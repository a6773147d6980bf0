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

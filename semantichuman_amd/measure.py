"""Body measurements: bone lengths and girths.

Mirrors reference utils_SH.py:86-161:
  * `cal_length(kps, skl_list)`            utils_SH.py:86-98   -> HIP kernel sh_bone_length
  * `measure_body_quick(v, kps, skl_list, factor_list, edge_point_index_list)`
                                           utils_SH.py:144-161 -> HIP kernels sh_measure_girth + sh_bone_length
  * `measure_body_batch(...)`              the same for a whole batch of meshes in two launches (what
                                           obj2npy.py:93-110 does one mesh at a time in numpy)
  * `cal_girth(face_point, face_normal, points)`  utils_SH.py:100-142, the CALIBRATION step that turns a
    cutting plane into an ordered ring of edge points.  It runs once per template, on the host (float64
    closed form of the reference's 3x3 solves), and its result is packed into `GirthRings`.
The measuring itself has no CPU path: tensors must live on the GPU.
"""
from __future__ import annotations

import math

import numpy as np
import torch

from . import ops


class GirthRings:
    """Device tables of P girth rings: CSR over rings of (vertex a, vertex b, factor) edge points
    (the reference keeps them as two pickled lists, cfg.PATH.factor_list / edge_point_index_list)."""

    def __init__(self, factor_list, edge_point_index_list, device):
        assert len(factor_list) == len(edge_point_index_list)
        ptr, a, b, f = [0], [], [], []
        for fac, epi in zip(factor_list, edge_point_index_list):
            epi = np.asarray(epi, dtype=np.int64).reshape(-1, 2)
            fac = np.broadcast_to(np.asarray(fac, dtype=np.float32).reshape(-1), (epi.shape[0],))   # scalar or [n, 1]
            a.append(epi[:, 0]); b.append(epi[:, 1]); f.append(fac)
            ptr.append(ptr[-1] + epi.shape[0])
        self.n_rings = len(factor_list)
        self.max_vertex = int(max(max(x.max() for x in a), max(x.max() for x in b))) if ptr[-1] else -1
        dev = torch.device(device)
        self.ptr = torch.tensor(ptr, dtype=torch.int32, device=dev)
        self.a = torch.from_numpy(np.concatenate(a).astype(np.int32)).to(dev)
        self.b = torch.from_numpy(np.concatenate(b).astype(np.int32)).to(dev)
        self.f = torch.from_numpy(np.concatenate(f).astype(np.float32)).to(dev)

    def tables(self):
        return self.ptr, self.a, self.b, self.f


def bone_table(skl_list, device):
    """[[a, b] | [a, b, c], ...] -> int32 [P, 3] with -1 for the missing third joint."""
    t = np.full((len(skl_list), 3), -1, dtype=np.int32)
    for i, s in enumerate(skl_list):
        if len(s) not in (2, 3):
            raise ValueError("bone %d: expected 2 or 3 joint ids, got %r" % (i, s))
        t[i, :len(s)] = s
    return torch.from_numpy(t).to(device)


def measure_body_batch(v, kps, rings: GirthRings, bones):
    """v [B, rows, 3], kps [B, K, 3], bones = bone_table(...) -> (girth [B, P], length [B, P2])."""
    if rings.max_vertex >= v.shape[1]:
        raise IndexError("girth ring refers to vertex %d but meshes have %d rows" % (rings.max_vertex, v.shape[1]))
    if int(bones.max()) >= kps.shape[1]:
        raise IndexError("bone refers to joint %d but there are %d joints" % (int(bones.max()), kps.shape[1]))
    return ops.measure_girth(v, rings.tables()), ops.bone_length(kps.contiguous(), bones)


def cal_length(kps, skl_list):
    """utils_SH.py:86-98: kps [N_kps, 3] -> length [N_part]."""
    return ops.bone_length(kps[None].contiguous(), bone_table(skl_list, kps.device))[0]


def measure_body_quick(v, kps, skl_list, factor_list, edge_point_index_list):
    """utils_SH.py:144-161: one mesh v [N_v, 3], joints kps [N_kps, 3] -> (girth [N_part], length [N_part])."""
    rings = GirthRings(factor_list, edge_point_index_list, v.device)
    g, l = measure_body_batch(v[None].contiguous(), kps[None], rings, bone_table(skl_list, v.device))
    return g[0], l[0]


# ------------------------------------------------------------------------------------------------ calibration
def cal_girth(face_point, face_normal, points):
    """utils_SH.py:100-142 on the host.  points [N, 2, 3]: end points of the N mesh edges cut by the plane
    through `face_point` with normal `face_normal`.  Returns (girth, X [N, 3], order [N]).

    The reference solves, per edge, the 3x3 system {n.X = n.p ; X lies on the edge's line} with
    torch.linalg.solve; the closed form is  X = a + t d,  d = a - b (zeros replaced by 1e-6 as at :111),
    t = n.(p - a) / n.d.  The ring order is the reference's: sort by the signed angle (degrees) between
    X_0 - mean and X_i - mean, the sign taken from the product of the cross product's components (:131-133).
    """
    p = np.asarray(face_point, dtype=np.float64).reshape(3)
    n = np.asarray(face_normal, dtype=np.float64).reshape(3)
    pts = np.asarray(points, dtype=np.float64)
    a = pts[:, 0, :]
    d = pts[:, 0, :] - pts[:, 1, :]
    d = np.where(d == 0, 1e-6, d)
    t = ((p - a) @ n) / (d @ n)
    X = a + t[:, None] * d
    Xv = X - X.mean(axis=0)
    m = np.sqrt((Xv * Xv).sum(axis=1))
    cos = (Xv[0:1] * Xv[1:]).sum(axis=1) / (m[1:] * m[0])
    theta = np.arccos(cos) / math.pi * 180
    cr = np.cross(np.repeat(Xv[0:1], Xv.shape[0] - 1, axis=0), Xv[1:])
    flag = np.where(cr[:, 0] * cr[:, 1] * cr[:, 2] > 0, 1.0, -1.0)
    order = np.argsort(np.concatenate(([0.0], theta * flag)), kind="stable")
    Xo = X[order]
    girth = np.sqrt(((Xo[0] - Xo[-1]) ** 2).sum()) + np.sqrt(((Xo[:-1] - Xo[1:]) ** 2).sum(axis=1)).sum()
    return float(girth), X, order


def ring_from_plane(v, edges, face_point, face_normal, vert_mask=None):
    """Calibrate one girth ring on a template: the edges of `edges` [E, 2] (optionally only those with both
    ends in `vert_mask`) that straddle the plane, ordered by cal_girth, as (factor [n], edge_point_index [n, 2])
    in the format measure_body_quick consumes (point = v[a] * (1 - factor) + v[b] * factor)."""
    v = np.asarray(v, dtype=np.float64)
    edges = np.asarray(edges, dtype=np.int64)
    side = (v - np.asarray(face_point, dtype=np.float64)) @ np.asarray(face_normal, dtype=np.float64)
    cut = side[edges[:, 0]] * side[edges[:, 1]] < 0
    if vert_mask is not None:
        cut &= vert_mask[edges[:, 0]] & vert_mask[edges[:, 1]]
    e = edges[cut]
    if e.shape[0] < 3:
        raise ValueError("plane cuts %d edges; a ring needs at least 3" % e.shape[0])
    _, X, order = cal_girth(face_point, face_normal, v[e])
    a, b = v[e[:, 0]], v[e[:, 1]]
    fac = np.linalg.norm(X - a, axis=1) / np.linalg.norm(b - a, axis=1)
    return fac[order].astype(np.float32), e[order]

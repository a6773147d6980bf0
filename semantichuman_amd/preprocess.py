"""Native mesh preprocessing: ctypes binding of libsh_preprocess.so (C ABI: include/sh_preprocess.h) and
the glue that assembles what reference main.py:93-181 assembles - M, A, D, U, F of `mesh_sampling.generate_transform_matrices`
(mesh_sampling.py:229-265) and the spiral index arrays of `utils_spiral.generate_spirals` (utils_spiral.py:45-95) - plus
readers / writers of the reference's `downsampling_matrices{a}{b}{c}{d}.pkl` layout (main.py:99-113).

Host code only (C++ behind a C ABI; no GPU).  The library must be present: there is no Python fallback.
"""
from __future__ import annotations

import ctypes
import math
import os
import pickle
from ctypes import POINTER, c_char_p, c_double, c_int, c_int32, c_int64

import numpy as np

from . import hierarchy, mesh_ops

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("SH_PREPROCESS_LIB") or os.path.join(_HERE, "lib", "libsh_preprocess.so")
_P = ctypes.c_void_p
SIGNATURES = {
    "shp_last_error": (c_char_p, []),
    "shp_qslim": (c_int, [_P, c_int, _P, c_int, c_int, _P, _P, _P, _P]),
    "shp_barycentric_upsample": (c_int, [_P, c_int, _P, c_int, _P, c_int, _P, _P, _P]),
    "shp_spirals": (c_int, [_P, c_int, _P, c_int, _P, c_int, c_int, _P, _P, c_int64, _P]),
}
_lib = None


class PreprocessLibraryError(RuntimeError):
    pass


def load(path=None):
    global _lib
    if _lib is not None and path is None:
        return _lib
    p = path or LIB_PATH
    if not os.path.exists(p):
        raise PreprocessLibraryError("semantichuman_amd: native preprocessing library not found at %s; build it with "
                                     "`make -C semantichuman_amd/csrc_host` (or __graft_entry__.build())" % p)
    lib = ctypes.CDLL(p)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)
        fn.restype, fn.argtypes = res, args
    if path is None:
        _lib = lib
    return lib


def _check(rc, what):
    if rc != 0:
        raise RuntimeError("%s failed (status %d): %s" % (what, rc, (load().shp_last_error() or b"").decode()))


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def _i32(a):
    return np.ascontiguousarray(a, dtype=np.int32)


def qslim(verts, faces, factor=None, n_verts_desired=None):
    """mesh_sampling.qslim_decimator_transformer (mesh_sampling.py:98-211): -> (new_faces int32 [F',3], keep int32 [N'])
    where row i of the reference's transform `mtx` is the unit row selecting column keep[i]."""
    if factor is None and n_verts_desired is None:
        raise Exception("Need either factor or n_verts_desired.")
    v, f = _f64(verts), _i32(faces)
    if n_verts_desired is None:
        n_verts_desired = math.ceil(len(v) * factor)
    fo = np.empty_like(f)
    keep = np.empty(len(v), dtype=np.int32)
    nf, nk = c_int32(0), c_int32(0)
    _check(load().shp_qslim(v.ctypes.data, len(v), f.ctypes.data, len(f), int(n_verts_desired), fo.ctypes.data, ctypes.byref(nf),
                            keep.ctypes.data, ctypes.byref(nk)), "shp_qslim")
    return fo[:nf.value].copy(), keep[:nk.value].copy()


def upsample_coefficients(src_v, src_f, tgt_v):
    """-> (cols int32 [T,3], coeffs float64 [T,3], part int32 [T]) of mesh_sampling.setup_deformation_transfer(source, target)."""
    sv, sf, tv = _f64(src_v), _i32(src_f), _f64(tgt_v)
    cols = np.empty((len(tv), 3), dtype=np.int32)
    coef = np.empty((len(tv), 3), dtype=np.float64)
    part = np.empty(len(tv), dtype=np.int32)
    _check(load().shp_barycentric_upsample(sv.ctypes.data, len(sv), sf.ctypes.data, len(sf), tv.ctypes.data, len(tv), cols.ctypes.data,
                                           coef.ctypes.data, part.ctypes.data), "shp_barycentric_upsample")
    return cols, coef, part


def setup_deformation_transfer(src_v, src_f, tgt_v):
    """The reference's U as a scipy csc_matrix [T, N_src] (explicit zeros kept, like the reference's)."""
    import scipy.sparse as sp
    cols, coef, _ = upsample_coefficients(src_v, src_f, tgt_v)
    rows = np.repeat(np.arange(len(cols)), 3)
    return sp.csc_matrix((coef.ravel(), (rows, cols.ravel())), shape=(len(cols), len(src_v)))


def upsample_csr(src_v, src_f, tgt_v) -> mesh_ops.CSR:
    """U as the CSR the kernels consume: fp32 values (main.py:205), zero coefficients dropped, columns ascending in a row."""
    cols, coef, _ = upsample_coefficients(src_v, src_f, tgt_v)
    vals = coef.astype(np.float32)
    keep = vals != 0
    o = np.argsort(np.where(keep, cols, np.iinfo(np.int32).max), axis=1, kind="stable")
    cols_s, vals_s, keep_s = (np.take_along_axis(a, o, 1) for a in (cols, vals, keep))
    rowptr = np.zeros(len(cols) + 1, dtype=np.int32)
    np.cumsum(keep.sum(1), out=rowptr[1:])
    return mesh_ops.CSR(len(cols), len(src_v), rowptr, cols_s[keep_s].astype(np.int32), vals_s[keep_s].astype(np.float32))


def get_spirals(verts, faces, reference_points, n_steps=1):
    """utils_spiral.get_spirals (utils_spiral.py:130-417; counter-clockwise, not random): list of per-vertex spiral lists."""
    v, f, rp = _f64(verts), _i32(faces), _i32(reference_points).ravel()
    rowptr = np.empty(len(v) + 1, dtype=np.int32)
    cap = 64 * len(v) * max(1, n_steps)
    while True:
        out = np.empty(cap, dtype=np.int32)
        n = c_int64(0)
        rc = load().shp_spirals(v.ctypes.data, len(v), f.ctypes.data, len(f), rp.ctypes.data, len(rp), int(n_steps), rowptr.ctypes.data,
                                out.ctypes.data, cap, ctypes.byref(n))
        if rc == -3 and n.value > cap:
            cap = int(n.value)
            continue
        _check(rc, "shp_spirals")
        return [out[rowptr[i]:rowptr[i + 1]].tolist() for i in range(len(v))]


def vert_connectivity(nv, faces):
    """opendr.topology.get_vert_connectivity as a scipy csc_matrix (entries = number of shared face orientations)."""
    import scipy.sparse as sp
    f = np.asarray(faces)
    vpv = sp.csc_matrix((nv, nv))
    for i in range(3):
        IS, JS = f[:, i].ravel(), f[:, (i + 1) % 3].ravel()
        m = sp.csc_matrix((np.ones(len(IS)), (IS, JS)), shape=(nv, nv))
        vpv = vpv + m + m.T
    return vpv


def generate_transform_matrices(verts, faces, factors):
    """mesh_sampling.generate_transform_matrices (mesh_sampling.py:229-265): -> (M, A, D, U, F) with M = [(verts, faces)] per
    level, A = adjacency, D = csc row-select matrices, U = csc up-sampling matrices, F = faces of the coarser levels."""
    import scipy.sparse as sp
    M = [(np.asarray(verts, dtype=np.float64), np.asarray(faces))]
    A, D, U, F = [vert_connectivity(len(verts), faces)], [], [], []
    for fac in factors:
        v, f = M[-1]
        ds_f, keep = qslim(v, f, factor=1.0 / fac)
        d = sp.csc_matrix((np.ones(len(keep)), (np.arange(len(keep)), keep)), shape=(len(keep), len(v)))
        D.append(d)
        F.append(ds_f)
        new_v = d.dot(v)
        M.append((new_v, ds_f))
        A.append(vert_connectivity(len(new_v), ds_f))
        U.append(setup_deformation_transfer(new_v, ds_f, v))
    return M, A, D, U, F


def save_downsampling_matrices(path, M, A, D, U, F):
    """The dict main.py:99-101 pickles: {'M_verts_faces', 'A', 'D', 'U', 'F'}."""
    with open(path, "wb") as fp:
        pickle.dump({"M_verts_faces": [(v, f) for v, f in M], "A": A, "D": D, "U": U, "F": F}, fp)


def load_downsampling_matrices(path):
    """main.py:103-113.  The file is a pickle by the reference's own format: load only files you wrote or trust."""
    with open(path, "rb") as fp:
        d = pickle.load(fp)
    return d["M_verts_faces"], d["A"], d["D"], d["U"], d["F"]


def reference_points_per_level(M, ref_point):
    """main.py:166-171: the vertex of every coarser level closest to the template's reference vertex."""
    pts = [[int(ref_point)]]
    for i in range(1, len(M)):
        d = ((M[i][0] - M[0][0][pts[0]]) ** 2).sum(1)
        pts.append([int(np.argmin(d))])
    return pts


def build_hierarchy(verts, faces, factors=(2, 2, 2, 2), step_sizes=(2, 2, 1, 1, 1), dilation=(2, 2, 1, 1, 1), ref_point=414):
    """Everything the model constructor needs from a template mesh (main.py:93-205), natively: -> hierarchy.Hierarchy."""
    M, A, D, U, F = generate_transform_matrices(verts, faces, factors)
    pts = reference_points_per_level(M, ref_point)
    lists = [get_spirals(M[i][0], M[i][1], pts[i], n_steps=step_sizes[i]) for i in range(len(M))]
    arrays, sizes, _ = hierarchy.layout_spirals(lists, dilation=list(dilation) if dilation else None)
    Ds, Us = [], []
    for l in range(len(D)):
        keep = D[l].tocsr().indices.astype(np.int32)
        d = mesh_ops.CSR(len(keep), M[l][0].shape[0], np.arange(len(keep) + 1, dtype=np.int32), keep, np.ones(len(keep), dtype=np.float32))
        Ds.append(mesh_ops.pad_dummy(d))
        Us.append(mesh_ops.pad_dummy(upsample_csr(M[l + 1][0], M[l + 1][1], M[l][0])))
    return hierarchy.Hierarchy([m[0].shape[0] for m in M], sizes, [a[0].astype(np.int32) for a in arrays], Ds, Us, M[0][0],
                               np.asarray(M[0][1], dtype=np.int32))

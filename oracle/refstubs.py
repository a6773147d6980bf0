"""TEST INFRASTRUCTURE - not part of the product path.

Import shims that let the *reference* (`/root/reference`, Python, read-only) be
imported in the build container so that golden vectors can be generated from it
(`oracle/gen_golden.py`).  The reference never travels to the GPU box; only the
vectors it produced do (`tests/golden/*.npz`).

The reference imports several packages that are absent from this image
(SURVEY.md 8c): yacs, psbody.mesh, opendr.topology, trimesh, torch_scatter,
tensorboardX, pytorch3d.  None of them is on the hot path; the shims below
provide just enough surface for `models.py`, `utils_spiral.py`,
`mesh_sampling.qslim_decimator_transformer`, `train_funcs.py` and
`test_funcs.py` to import and run.

`opendr.topology` is a third-party dependency that is not vendored in the
reference (README.md:16-37 pins "opendr"); the two functions the reference calls
(mesh_sampling.py:99,231) are restated here from their published behaviour:
vertex adjacency as a symmetric sparse matrix, and the unique undirected edge
list (row < col) of that matrix in COO order.
"""
from __future__ import annotations

import ast
import sys
import types

import numpy as np
import scipy.sparse as sp

REFERENCE_ROOT = "/root/reference"


class _CfgNode(dict):
    """Attribute-dict stand-in for yacs.config.CfgNode (configure/cfgs.py:5)."""

    def __init__(self, init=None, new_allowed=False):
        super().__init__(init or {})

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError as e:
            raise AttributeError(k) from e

    def __setattr__(self, k, v):
        self[k] = v

    def merge_from_file(self, path):
        import yaml
        with open(path) as f:
            data = yaml.safe_load(f)

        def coerce(v):
            # yacs literal_evals strings such as '1e-2' 
            if isinstance(v, str):
                try:
                    return ast.literal_eval(v)
                except Exception:
                    return v
            return v

        def merge(dst, src):
            for k, v in src.items():
                if isinstance(v, dict):
                    if k not in dst:
                        dst[k] = _CfgNode()
                    merge(dst[k], v)
                else:
                    dst[k] = coerce(v)
        merge(self, data)


def get_vert_connectivity(mesh_v, mesh_f):
    """Sparse symmetric vertex-vertex adjacency (entry = number of shared faces
    orientations), as opendr.topology.get_vert_connectivity returns it."""
    n = len(mesh_v)
    vpv = sp.csc_matrix((n, n))
    for i in range(3):
        IS = mesh_f[:, i].ravel()
        JS = mesh_f[:, (i + 1) % 3].ravel()
        mtx = sp.csc_matrix((np.ones(len(IS)), (IS, JS)), shape=(n, n))
        vpv = vpv + mtx + mtx.T
    return vpv


def get_vertices_per_edge(mesh_v, mesh_f):
    """E x 2 array of unique undirected edges (first < second)."""
    vc = sp.coo_matrix(get_vert_connectivity(mesh_v, mesh_f))
    result = np.hstack((vc.row.reshape(-1, 1), vc.col.reshape(-1, 1)))
    return result[result[:, 0] < result[:, 1]]


class Mesh:
    """Minimal psbody.mesh.Mesh stand-in: .v (float64 [V,3]) and .f (int [F,3])."""

    def __init__(self, v=None, f=None, filename=None):
        if filename is not None:
            raise NotImplementedError("file I/O is out of scope for the oracle")
        self.v = np.asarray(v, dtype=np.float64)
        self.f = np.asarray(f)


def install():
    """Register the shims and put the reference on sys.path. Idempotent."""
    sys.dont_write_bytecode = True

    def mod(name, **attrs):
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        sys.modules[name] = m
        return m

    if "yacs" not in sys.modules:
        mod("yacs")
        mod("yacs.config", CfgNode=_CfgNode)
    if "psbody" not in sys.modules:
        mod("psbody")
        mod("psbody.mesh", Mesh=Mesh)
    if "trimesh" not in sys.modules:
        mod("trimesh")
        mod("trimesh.exchange")
        mod("trimesh.exchange.export", export_mesh=None)
    if "torch_scatter" not in sys.modules:
        mod("torch_scatter", scatter_add=None)
    if "opendr" not in sys.modules:
        mod("opendr")
        mod("opendr.topology", get_vert_connectivity=get_vert_connectivity,
            get_vertices_per_edge=get_vertices_per_edge)
    if "tensorboardX" not in sys.modules:
        mod("tensorboardX", SummaryWriter=object)
    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)

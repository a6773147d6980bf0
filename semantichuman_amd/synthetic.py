"""Synthetic genus-0 triangle meshes and mesh batches for benchmarks and tests.

The reference trains on DFAUST/SMPL meshes (6890 vertices) that are not
redistributable, so every input in this repo is synthetic.  `box_sphere(a,b,c)`
builds the surface grid of an a x b x c box (V = 2(ab+bc+ca)+2 vertices,
F = 2V-4 faces), projects it onto the unit sphere and applies an anisotropic
scale so the aspect is body-like.  (42,42,20) gives exactly 6890 vertices /
13776 faces; (84,84,40) gives 27554 vertices; (6,6,4) gives 170.

Nothing here comes from the reference; it is the stand-in for the
template mesh `template.obj` that reference main.py:47 loads.
"""
from __future__ import annotations

import numpy as np


def box_sphere(a: int, b: int, c: int, scale=(0.25, 0.15, 0.9)):
    """Return (verts float64 [V,3], faces int32 [F,3]) of a closed, consistently
    oriented (outward, counter-clockwise) genus-0 mesh."""
    dims = (a, b, c)
    vid: dict[tuple[int, int, int], int] = {}
    pts: list[tuple[int, int, int]] = []

    def on_surface(p):
        return any(p[k] == 0 or p[k] == dims[k] for k in range(3))

    # z-major, then y, then x: neighbouring ids are spatially close on each ring
    for z in range(c + 1):
        for y in range(b + 1):
            for x in range(a + 1):
                p = (x, y, z)
                if on_surface(p):
                    vid[p] = len(pts)
                    pts.append(p)

    faces: list[tuple[int, int, int]] = []

    def quad(p00, p10, p11, p01, flip_diag):
        """Corners are counter-clockwise seen from outside."""
        i00, i10, i11, i01 = vid[p00], vid[p10], vid[p11], vid[p01]
        if flip_diag:
            faces.append((i00, i10, i01))
            faces.append((i10, i11, i01))
        else:
            faces.append((i00, i10, i11))
            faces.append((i00, i11, i01))

    # For each of the 3 axes, the two opposite faces of the box.  (u,v,w) is a
    # right-handed permutation of the axes with w the face normal.
    for w_axis in range(3):
        u_axis, v_axis = (w_axis + 1) % 3, (w_axis + 2) % 3
        for side in (0, 1):
            wv = dims[w_axis] * side
            for i in range(dims[u_axis]):
                for j in range(dims[v_axis]):
                    def P(di, dj):
                        p = [0, 0, 0]
                        p[u_axis], p[v_axis], p[w_axis] = i + di, j + dj, wv
                        return tuple(p)
                    # same diagonal on every quad of a face: interior valence 6
                    flip = False
                    if side == 1:   # outward normal = +w: (u,v) counter-clockwise
                        quad(P(0, 0), P(1, 0), P(1, 1), P(0, 1), flip)
                    else:           # outward normal = -w: reverse orientation
                        quad(P(0, 0), P(0, 1), P(1, 1), P(1, 0), flip)

    v = np.asarray(pts, dtype=np.float64)
    v = v / np.asarray(dims, dtype=np.float64) - 0.5          # unit cube centred
    v = v / np.linalg.norm(v, axis=1, keepdims=True)           # onto the sphere
    v = v * np.asarray(scale, dtype=np.float64)
    f = np.asarray(faces, dtype=np.int32)
    assert v.shape[0] == 2 * (a * b + b * c + c * a) + 2
    assert f.shape[0] == 2 * v.shape[0] - 4
    return v, f


def synth_batch(template: np.ndarray, n: int, seed: int, dummy: bool = True) -> np.ndarray:
    """n synthetic meshes x = T*(1+0.1*e1) + 0.005*e2, float32
    [n, V(+1), 3]; the extra last row is the all-zero dummy vertex that the
    reference dataset appends (autoencoder_dataset.py:45-48)."""
    rs = np.random.RandomState(seed)
    V = template.shape[0]
    e1 = rs.randn(n, 1, 3)
    e2 = rs.randn(n, V, 3)
    x = template[None] * (1.0 + 0.1 * e1) + 0.005 * e2
    if dummy:
        x = np.concatenate([x, np.zeros((n, 1, 3))], axis=1)
    return x.astype(np.float32)


def closed_form_fill(shape, a: float, b: float, c: float) -> np.ndarray:
    """Deterministic, RNG-free tensor fill w[i] = a*sin(b*i+c) (float32).  Used for
    weights in golden fixtures so any box can regenerate them bit-identically
    from three scalars (computed in float64, rounded once)."""
    n = int(np.prod(shape))
    i = np.arange(n, dtype=np.float64)
    return (a * np.sin(b * i + c)).astype(np.float32).reshape(shape)

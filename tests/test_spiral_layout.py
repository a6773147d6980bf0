"""Row a8, layout stage: hierarchy.layout_spirals against what the reference's generate_spirals produced from the same
per-vertex spiral lists (fixture tests/golden/spiral_layout.npz, oracle/gen_golden.py:gen_spiral_layout) - bit-exact -
and the consumer side: the int32 gather tables the kernels take map -1 to the dummy row (models.py:42 negative-index wrap)."""
import os

import numpy as np

from semantichuman_amd import mesh_ops
from semantichuman_amd.hierarchy import layout_spirals


def _raw(g):
    levels = []
    for i in range(int(g["levels"])):
        flat, lens = g["raw_flat_%d" % i], g["raw_len_%d" % i]
        offs = np.concatenate([[0], np.cumsum(lens)])
        levels.append([flat[offs[j]:offs[j + 1]].tolist() for j in range(len(lens))])
    return levels


def test_layout_matches_reference(golden_dir):
    g = np.load(os.path.join(golden_dir, "spiral_layout.npz"))
    raw = _raw(g)
    for tag, dil in (("dil", g["dilation"].tolist()), ("nodil", None)):
        arrays, sizes, _ = layout_spirals(raw, dilation=dil)
        assert sizes == g["sizes_" + tag].tolist()
        for i, S in enumerate(arrays):
            ref = g["S_%s_%d" % (tag, i)]
            assert S.dtype == ref.dtype == np.float64 and S.shape == ref.shape
            assert np.array_equal(S, ref), (tag, i)
            assert (S[0, -1] == -1).all()                      # the dummy vertex's own row


def test_layout_edge_cases():
    # shorter spirals are padded with -1, longer ones truncated, the size rule is int(mean + 2 std)
    arrays, sizes, dil = layout_spirals([[[0, 1, 2, 3, 4, 5, 6], [1, 0], [2, 0, 1]]], dilation=[2])
    assert dil[0] == [[0, 1, 3, 5], [1, 0], [2, 0]]
    L = np.array([4, 2, 2])
    assert sizes == [int(L.mean() + 2 * L.std())] == [4]
    assert arrays[0].tolist() == [[[0, 1, 3, 5], [1, 0, -1, -1], [2, 0, -1, -1], [-1, -1, -1, -1]]]
    arrays, sizes, _ = layout_spirals([[[0, 1, 2, 3, 4, 5, 6, 7, 8], [1], [2], [3], [4], [5], [6], [7]]])
    assert sizes == [7] and arrays[0][0, 0].tolist() == [0, 1, 2, 3, 4, 5, 6]       # truncated


def test_gather_table_maps_padding_to_dummy_row(golden_dir):
    g = np.load(os.path.join(golden_dir, "spiral_layout.npz"))
    S = g["S_dil_0"]
    table = mesh_ops.spirals_to_table(S)
    n1 = S.shape[1]
    assert table.dtype == np.int32 and table.shape == (n1, S.shape[2])
    assert np.array_equal(table, np.where(S[0] < 0, n1 - 1, S[0]).astype(np.int32))

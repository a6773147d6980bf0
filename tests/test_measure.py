"""Body measurements (utils_SH.cal_length / cal_girth / measure_body_quick) against vectors
produced by the reference (tests/golden/measure.npz, oracle/gen_golden.py:gen_measure)."""
import os

import numpy as np
import pytest
import torch

from oracle import ref_cpu
from semantichuman_amd import constants as C
from semantichuman_amd import measure

OBLIQUE = (5, 6, 7)     # cuts whose ring order is well defined (for axis-aligned normals the reference's sign
#                         test, a product of cross-product components that are ~0, is decided by rounding noise)


@pytest.fixture(scope="module")
def gm(golden_dir):
    return np.load(os.path.join(golden_dir, "measure.npz"))


def rings_of(g):
    n = int(g["n_planes"])
    return [g["factor_%d" % i] for i in range(n)], [g["epi_%d" % i] for i in range(n)]


# ------------------------------------------------------------------------------------------ CPU
def test_oracle_measurements_match_reference(gm):
    g = gm
    fac, epi = rings_of(g)
    x, kps = torch.from_numpy(g["x"]), torch.from_numpy(g["kps"])
    for b in range(x.shape[0]):
        np.testing.assert_allclose(ref_cpu.girths(x[b], fac, epi).numpy(), g["girth_batch"][b], rtol=1e-6)
        np.testing.assert_allclose(ref_cpu.bone_lengths(kps[b], C.SKL_LIST[1:]).numpy(), g["length_batch"][b], rtol=1e-6)
        np.testing.assert_allclose(ref_cpu.bone_lengths(kps[b], C.NEWSKL_LIST).numpy(), g["length_newskl"][b], rtol=1e-6)


def test_oracle_plane_ring_matches_reference(gm):
    g = gm
    v = g["verts"]
    for i in range(int(g["n_planes"])):
        pts = torch.from_numpy(v[g["cut_edges_%d" % i]])
        girth, X, order = ref_cpu.plane_ring(torch.from_numpy(g["plane_p_%d" % i]), torch.from_numpy(g["plane_n_%d" % i]), pts)
        np.testing.assert_allclose(X.numpy(), g["X_%d" % i], atol=2e-6)
        if i in OBLIQUE:
            assert np.array_equal(order.numpy(), g["order_%d" % i])
            np.testing.assert_allclose(float(girth), float(g["girth_%d" % i]), rtol=1e-5)


def test_host_cal_girth_matches_reference(gm):
    """The product's calibration step (float64 closed form) against the reference's fp32 3x3 solves."""
    g = gm
    v = g["verts"]
    edges = np.unique(np.sort(np.concatenate([g["cut_edges_%d" % i] for i in range(int(g["n_planes"]))]), axis=1), axis=0)
    for i in range(int(g["n_planes"])):
        p, n = g["plane_p_%d" % i], g["plane_n_%d" % i]
        girth, X, order = measure.cal_girth(p, n, v[g["cut_edges_%d" % i]])
        np.testing.assert_allclose(X, g["X_%d" % i], atol=5e-6)                 # fp32 solve noise of the reference
        if i in OBLIQUE:
            assert np.array_equal(order, g["order_%d" % i])
            assert abs(girth - float(g["girth_%d" % i])) <= 1e-5 * girth
            # the calibrated ring reproduces the reference-derived edge-point list
            fac, epi = measure.ring_from_plane(v, edges, p, n)
            assert np.array_equal(epi, g["epi_%d" % i])
            np.testing.assert_allclose(fac, g["factor_%d" % i][:, 0], atol=2e-4)   # |X-a|/|b-a| with |b-a| ~ 0.05


def test_bone_table_and_ring_packing(gm):
    fac, epi = rings_of(gm)
    r = measure.GirthRings(fac, epi, "cpu")
    assert r.n_rings == len(fac) and r.ptr[-1].item() == sum(e.shape[0] for e in epi)
    assert r.a.dtype == torch.int32 and r.f.dtype == torch.float32
    r2 = measure.GirthRings([0.25] * len(epi), epi, "cpu")                      # scalar factor broadcasts over the ring
    assert torch.all(r2.f == 0.25) and r2.f.numel() == r.f.numel()
    t = measure.bone_table([[1, 2], [3, 4, 5]], "cpu")
    assert t.tolist() == [[1, 2, -1], [3, 4, 5]]
    with pytest.raises(ValueError):
        measure.bone_table([[1]], "cpu")
    with pytest.raises(RuntimeError):                                           # no CPU path for the measuring itself
        measure.measure_body_batch(torch.from_numpy(gm["x"]), torch.from_numpy(gm["kps"]), r, measure.bone_table(C.SKL_LIST[1:], "cpu"))


# ------------------------------------------------------------------------------------------ GPU
@pytest.mark.gpu
def test_hip_measurements_match_reference(gm):
    g = gm
    dev = torch.device("cuda:0")
    fac, epi = rings_of(g)
    x, kps = torch.from_numpy(g["x"]).to(dev), torch.from_numpy(g["kps"]).to(dev)
    rings = measure.GirthRings(fac, epi, dev)
    girth, length = measure.measure_body_batch(x, kps, rings, measure.bone_table(C.SKL_LIST[1:], dev))
    # fp32, different summation order (wave reduction vs the reference's sequential sum over ~100 segments)
    np.testing.assert_allclose(girth.cpu().numpy(), g["girth_batch"], rtol=2e-6)
    np.testing.assert_allclose(length.cpu().numpy(), g["length_batch"], rtol=1e-6)
    # reference-shaped single-mesh calls
    g1, l1 = measure.measure_body_quick(x[1], kps[1], C.SKL_LIST[1:], [torch.from_numpy(f) for f in fac], epi)
    np.testing.assert_allclose(g1.cpu().numpy(), g["girth_batch"][1], rtol=2e-6)
    np.testing.assert_allclose(l1.cpu().numpy(), g["length_batch"][1], rtol=1e-6)
    np.testing.assert_allclose(measure.cal_length(kps[2], C.NEWSKL_LIST).cpu().numpy(), g["length_newskl"][2], rtol=1e-6)
    # edge cases: a 2-point ring counts its segment twice (utils_SH.py:156-158), a 1-point ring is 0
    small = measure.GirthRings([0.5, 0.0], [np.array([[0, 1], [2, 3]]), np.array([[4, 5]])], dev)
    gs = ops_girth(x, small)
    q0, q1 = (x[:, 0] + x[:, 1]) / 2, (x[:, 2] + x[:, 3]) / 2
    np.testing.assert_allclose(gs[:, 0].cpu().numpy(), (2 * (q0 - q1).norm(dim=1)).cpu().numpy(), rtol=1e-6)
    assert torch.all(gs[:, 1] == 0)
    with pytest.raises(IndexError):
        measure.measure_body_batch(x[:, :50], kps, rings, measure.bone_table(C.SKL_LIST[1:], dev))


def ops_girth(x, rings):
    from semantichuman_amd import ops
    return ops.measure_girth(x, rings.tables())

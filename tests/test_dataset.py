"""Data path against items produced by the reference's autoencoder_dataset
(tests/golden/dataset.npz, oracle/gen_golden.py:gen_dataset)."""
import os
import types

import numpy as np
import pytest
import torch

from oracle import ref_cpu
from semantichuman_amd import dataset as ds_mod


@pytest.fixture(scope="module")
def gd(golden_dir):
    return np.load(os.path.join(golden_dir, "dataset.npz"))


def shapedata_of(g):
    return types.SimpleNamespace(mean=g["mean"], std=g["std"], center=g["center"], scale=g["scale"])


# ------------------------------------------------------------------------------------------ CPU
def test_oracle_dataset_items_match_reference(gd):
    g, sd = gd, shapedata_of(gd)
    with np.errstate(invalid="ignore"):
        for k, norm in enumerate(g["normalizations"]):
            for i in range(g["raw"].shape[0]):
                item = ref_cpu.dataset_item(g["raw"][i], str(norm), g["J_regressor"], sd, i)
                np.testing.assert_array_equal(item, g["verts_%d" % k][i])           # same numpy ops -> bit-exact
        np.testing.assert_array_equal(ref_cpu.dataset_item(g["raw"][5], "zeroroot", g["J_regressor"], sd, 5, dummy_node=False),
                                      g["verts_nodummy"][5])


def test_split_layout_and_flags(gd, tmp_path):
    g = gd
    ds_mod.write_split(str(tmp_path), "train", g["raw"], g["measure"])
    assert sorted(os.listdir(tmp_path)) == ["measure_train", "paths_train.npy", "points_train"]
    assert sorted(os.listdir(tmp_path / "points_train"))[:2] == ["000000.npy", "000001.npy"]
    d = ds_mod.autoencoder_dataset(str(tmp_path), "train", None, normalization="zeroroot", measure_flag=True,
                                   J_regressor=g["J_regressor"])
    assert len(d) == g["raw"].shape[0]
    raw, meas = d.read_raw()
    np.testing.assert_array_equal(raw, g["raw"]); np.testing.assert_array_equal(meas, g["measure"])
    assert ds_mod.normalization_flags("No") == 0
    assert ds_mod.normalization_flags("zeroroot") == 2
    assert ds_mod.normalization_flags("zeroroot_onelength_small") == 2 | 4 | 8
    assert ds_mod.normalization_flags("zeromean_zeroroot_normal") == 1 | 2 | 32
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        d.resident("cpu")
    with pytest.raises(RuntimeError, match="resident"):
        d[0]


def test_loader_sharding_is_complete_and_equal_on_every_rank():
    """10 samples over 3 ranks: every rank iterates the SAME number of samples / batches (one collective per batch in the
    training loops); padded by wrap-around every sample is seen, truncated the shards are disjoint."""
    class Fake:
        verts = torch.zeros(10, 2, 3); measure_flag = False
        def __len__(self): return 10
        def resident(self, device): return self
    for pad, per in ((True, 4), (False, 3)):
        orders = []
        for r in range(3):
            ld = ds_mod.ResidentLoader(Fake(), batch_size=2, shuffle=True, device="cpu", seed=7, rank=r, world_size=3, pad=pad)
            orders.append(ld._order().tolist())
            assert len(orders[-1]) == per and len(ld) == -(-per // 2)
        seen = sum(orders, [])
        if pad:
            assert set(seen) == set(range(10)) and len(seen) == 12
        else:
            assert len(set(seen)) == 9


# ------------------------------------------------------------------------------------------ GPU
@pytest.mark.gpu
def test_hip_resident_dataset_matches_reference(gd, tmp_path):
    g, sd = gd, shapedata_of(gd)
    dev = torch.device("cuda:0")
    ds_mod.write_split(str(tmp_path), "train", g["raw"], g["measure"])
    for k, norm in enumerate(g["normalizations"]):
        d = ds_mod.autoencoder_dataset(str(tmp_path), "train", sd, normalization=str(norm), dummy_node=True, measure_flag=True,
                                       J_regressor=g["J_regressor"]).resident(dev)
        got, want = d.verts.cpu().numpy(), g["verts_%d" % k]
        if norm == "No":
            np.testing.assert_array_equal(got, want)                                 # pure copy + pad + NaN->0
        else:
            # fp32 elementwise ops on O(1) coordinates; the reductions (mean / J.v) are accumulated in double here
            # and in float32-pairwise/BLAS order in numpy
            np.testing.assert_allclose(got, want, atol=2e-6, rtol=2e-6, err_msg=str(norm))
        assert np.all(got[:, -1, :] == 0)
        item = d[4]
        assert item["idx"] == 4 and torch.equal(item["verts"], d.verts[4]) and torch.equal(item["measure"].cpu(), torch.from_numpy(g["measure"][4]))
    d = ds_mod.autoencoder_dataset(str(tmp_path), "train", sd, normalization="zeroroot", dummy_node=False,
                                   J_regressor=g["J_regressor"]).resident(dev)
    np.testing.assert_allclose(d.verts.cpu().numpy(), g["verts_nodummy"], atol=2e-6, rtol=2e-6)


@pytest.mark.gpu
def test_hip_resident_loader_batches(gd, tmp_path):
    g = gd
    dev = torch.device("cuda:0")
    ds_mod.write_split(str(tmp_path), "train", g["raw"], g["measure"])
    d = ds_mod.autoencoder_dataset(str(tmp_path), "train", None, normalization="zeroroot", measure_flag=True,
                                   J_regressor=g["J_regressor"])
    ld = ds_mod.ResidentLoader(d, batch_size=5, shuffle=False, device=dev)
    batches = list(ld)
    assert len(ld) == 3 and [b["verts"].shape[0] for b in batches] == [5, 5, 2]        # ragged last batch, like DataLoader
    assert torch.equal(torch.cat([b["verts"] for b in batches]), d.verts)              # gather is a bit-exact copy
    assert torch.equal(torch.cat([b["idx"] for b in batches]).cpu(), torch.arange(12))
    assert torch.equal(torch.cat([b["measure"] for b in batches]).cpu(), torch.from_numpy(g["measure"]))
    assert len(list(ds_mod.ResidentLoader(d, batch_size=5, drop_last=True))) == 2
    sh = ds_mod.ResidentLoader(d, batch_size=4, shuffle=True, seed=3)
    e1 = torch.cat([b["idx"] for b in sh]).cpu()
    e2 = torch.cat([b["idx"] for b in sh]).cpu()
    assert sorted(e1.tolist()) == list(range(12)) and not torch.equal(e1, e2)          # a fresh permutation per epoch
    for b in sh:
        assert torch.equal(b["verts"], d.verts[b["idx"]])
        break


@pytest.mark.gpu
def test_hip_training_from_resident_split_equals_dataloader(golden_dir, tmp_path):
    """The reference loop fed by ResidentLoader (disk -> HBM once -> batch gathers) ends with exactly the weights it
    reaches when fed by a torch DataLoader over the same samples: the data path changes where batches come from,
    not what they contain."""
    import semantichuman_amd as sh
    from semantichuman_amd import train_funcs
    from semantichuman_amd.hierarchy import load_hierarchy
    from tests.test_train_loop import DS, FD, FE, Writer
    g = np.load(os.path.join(golden_dir, "small_loop.npz"))
    g0 = np.load(os.path.join(golden_dir, "small_ae.npz"))
    h = load_hierarchy(os.path.join(golden_dir, "small_ae.npz"))
    dev = torch.device("cuda:0")
    ds_mod.write_split(str(tmp_path), "train", g["x_train"][:, :-1, :])
    ds_mod.write_split(str(tmp_path), "val", g["x_val"][:, :-1, :])
    ends = []
    for resident in (True, False):
        m = sh.SpiralAutoencoder(FE, FD, 16, h.sizes, h.spiral_sizes, h.spirals, h.D, h.U, dev)
        m.load_state_dict({k[3:]: torch.from_numpy(g0[k]) for k in g0.files if k.startswith("w0/")})
        opt = torch.optim.Adam(m.parameters(), lr=1e-3, weight_decay=5e-5)
        if resident:
            ltr = ds_mod.ResidentLoader(ds_mod.autoencoder_dataset(str(tmp_path), "train", None), batch_size=2, device=dev)
            lva = ds_mod.ResidentLoader(ds_mod.autoencoder_dataset(str(tmp_path), "val", None), batch_size=2, device=dev)
        else:
            ltr = torch.utils.data.DataLoader(DS(torch.from_numpy(g["x_train"])), batch_size=2, shuffle=False)
            lva = torch.utils.data.DataLoader(DS(torch.from_numpy(g["x_val"])), batch_size=2, shuffle=False)
        w = Writer()
        out = tmp_path / ("r%d" % resident)
        out.mkdir()
        train_funcs.train_autoencoder_dataloader(ltr, lva, dev, m, opt, torch.nn.functional.l1_loss, 1, 2, 10, None, None, w,
                                                 types.SimpleNamespace(reference_mesh=types.SimpleNamespace(f=h.faces)),
                                                 str(out), str(out), "checkpoint", None, None, None, False,
                                                 edgereg_epoch=0, edgereg_w=1e-2, ck_frequency=10, verbose=False)
        ends.append(([p.detach().clone() for p in m.parameters()], w.s))
    assert ends[0][1] == ends[1][1]
    for a, b in zip(ends[0][0], ends[1][0]):
        assert torch.equal(a, b)

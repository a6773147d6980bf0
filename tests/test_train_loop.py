"""The training / evaluation loops against a run of the REFERENCE's own
train_funcs.train_autoencoder_dataloader + test_funcs.test_autoencoder_dataloader
(tests/golden/small_loop.npz, produced by oracle/gen_golden.py:gen_loop)."""
import os

import numpy as np
import pytest
import torch

from oracle import ref_cpu
from semantichuman_amd.hierarchy import load_hierarchy

FE = [[3, 16, 32, 64, 128], [[], [], [], [], []]]
FD = [[128, 64, 32, 32, 16], [[], [], [], [], 3]]
# Trajectory tolerances.  Single steps are pinned tightly elsewhere (test_oracle_golden /
# test_gpu_parity: 1e-5 / 1e-4).  Over several Adam steps the comparison is chaotic: the first
# update moves every weight by ~lr * sign(g), so elements with |g| ~ 1e-8 flip with rounding, and
# the non-smooth edge term amplifies that.  Measured here: the SAME oracle code in fp32 vs fp64
# differs by 1e-6 / 5e-5 / 1e-3 in the epoch-1/2/3 training loss, the reference loop vs a lock-step
# re-implementation by up to 1e-2 at step 6; torch's multi-threaded CPU scatter-add (index_put_ accumulate in the gather's
# backward) is not even run-to-run reproducible, so the reference run behind the fixture carries that noise itself.
# Per-epoch relative tolerances below are ~3x those figures; the CPU test pins one thread so that ITS result is fixed.
EPOCH_TOL = [2e-4, 1e-2, 5e-2]


class DS(torch.utils.data.Dataset):
    dummy_node = True

    def __init__(self, x):
        self.x = x

    def __len__(self):
        return self.x.shape[0]

    def __getitem__(self, i):
        return {"verts": self.x[i], "idx": i}


class Writer:
    def __init__(self):
        self.s = []

    def add_scalar(self, tag, value, step):
        self.s.append((tag, float(value), int(step)))


def golden(golden_dir):
    g = np.load(os.path.join(golden_dir, "small_loop.npz"))
    g0 = np.load(os.path.join(golden_dir, "small_ae.npz"))
    h = load_hierarchy(os.path.join(golden_dir, "small_ae.npz"))
    ref_scalars = list(zip([str(t) for t in g["scalar_tags"]], g["scalar_values"].tolist(), g["scalar_steps"].tolist()))
    return g, g0, h, ref_scalars


def test_oracle_loop_matches_reference_loop(golden_dir):
    """CPU: the oracle's train_step, driven epoch by epoch, reproduces the reference loop's scalars."""
    g, g0, h, ref_scalars = golden(golden_dir)
    S, D, U = h.dense_constants()
    m = ref_cpu.SpiralAEOracle(FE, FD, 16, h.sizes, h.spiral_sizes, S, D, U)
    m.load_state_dict({k[3:]: torch.from_numpy(g0[k]) for k in g0.files if k.startswith("w0/")})
    opt = torch.optim.Adam(m.parameters(), lr=1e-3, weight_decay=5e-5)
    sched = torch.optim.lr_scheduler.StepLR(opt, 1, gamma=0.99)
    threads = torch.get_num_threads()
    torch.set_num_threads(1)                      # fixed summation order: the outcome of this test does not vary run to run
    try:
        _oracle_loop_body(g, h, m, opt, sched, ref_scalars)
    finally:
        torch.set_num_threads(threads)


def _oracle_loop_body(g, h, m, opt, sched, ref_scalars):
    xtr, xva = torch.from_numpy(g["x_train"]), torch.from_numpy(g["x_val"])
    ref_tr = [v for t, v, s in ref_scalars if t == "avg_epoch_train_loss"]
    ref_va = [v for t, v, s in ref_scalars if t == "avg_epoch_valid_loss"]
    for ep in range(3):
        tl = 0.0
        for i in range(0, 6, 2):
            tl += 2 * float(ref_cpu.train_step(m, opt, xtr[i:i + 2], faces=h.faces, edgereg_w=1e-2))
        with torch.no_grad():
            vl = sum(2 * float(torch.nn.functional.l1_loss(xva[i:i + 2, :-1], m(xva[i:i + 2])[0][:, :-1])) for i in range(0, 4, 2))
        sched.step()
        assert tl / 6 == pytest.approx(ref_tr[ep], rel=EPOCH_TOL[ep])
        assert vl / 4 == pytest.approx(ref_va[ep], rel=EPOCH_TOL[ep])
    l1, l2 = ref_cpu.eval_metrics(m(xva)[0], xva)
    assert float(l1) == pytest.approx(float(g["eval_l1"]), rel=EPOCH_TOL[2])
    assert float(l2) == pytest.approx(float(g["eval_l2mm"]), rel=EPOCH_TOL[2])


@pytest.mark.gpu
def test_hip_loop_matches_reference_loop(golden_dir, tmp_path):
    """GPU: semantichuman_amd.train_funcs / test_funcs with the HIP model against the reference run:
    identical logging tags and steps, losses within the trajectory tolerances above, reference
    checkpoint layout, resume works."""
    import semantichuman_amd as sh
    from semantichuman_amd import test_funcs, train_funcs
    from types import SimpleNamespace
    g, g0, h, ref_scalars = golden(golden_dir)
    dev = torch.device("cuda:0")
    m = sh.SpiralAutoencoder(FE, FD, 16, h.sizes, h.spiral_sizes, h.spirals, h.D, h.U, dev)
    m.load_state_dict({k[3:]: torch.from_numpy(g0[k]) for k in g0.files if k.startswith("w0/")})
    opt = torch.optim.Adam(m.parameters(), lr=1e-3, weight_decay=5e-5)
    sched = torch.optim.lr_scheduler.StepLR(opt, 1, gamma=0.99)
    ltr = torch.utils.data.DataLoader(DS(torch.from_numpy(g["x_train"])), batch_size=2, shuffle=False)
    lva = torch.utils.data.DataLoader(DS(torch.from_numpy(g["x_val"])), batch_size=2, shuffle=False)
    w = Writer()
    shapedata = SimpleNamespace(reference_mesh=SimpleNamespace(f=h.faces))
    train_funcs.train_autoencoder_dataloader(ltr, lva, dev, m, opt, torch.nn.functional.l1_loss, 1, 3, 10, None, sched, w,
                                             shapedata, str(tmp_path), str(tmp_path), "checkpoint", None, None, None, False,
                                             edgereg_epoch=0, edgereg_w=1e-2, ck_frequency=1, verbose=False)
    assert [(t, s) for t, v, s in w.s] == [(t, s) for t, v, s in ref_scalars]
    for (t, v, s), (_, rv, _) in zip(w.s, ref_scalars):
        ep = (s // 3) if t.startswith("loss/") else s - 1              # step 0,3,6 -> epoch index; epoch tags carry 1..3
        assert v == pytest.approx(rv, rel=EPOCH_TOL[ep] * (3 if "edgereg" in t else 1)), (t, s)
    assert opt.param_groups[0]["lr"] == pytest.approx(float(g["lr_after"]), rel=1e-12)
    _, _, _, l1, l2 = test_funcs.test_autoencoder_dataloader(dev, m, lva, None, None)
    assert l1 == pytest.approx(float(g["eval_l1"]), rel=EPOCH_TOL[2])
    assert l2 == pytest.approx(float(g["eval_l2mm"]), rel=EPOCH_TOL[2])
    for name, p in m.named_parameters():
        d = np.abs(p.detach().cpu().numpy() - g["w_end/" + name])
        assert d.max() <= 9.1e-3, (name, d.max())                 # nothing can move further than 9 steps x lr
    # checkpoint: reference keys, CPU tensors, loads into the oracle (= reference parameter names), resume
    ck = torch.load(tmp_path / "checkpoint3.pth.tar", map_location="cpu", weights_only=False)
    assert sorted(ck.keys()) == [str(k) for k in g["ck_keys"]] and ck["epoch"] == int(g["ck_epoch"])
    assert all(not v.is_cuda for v in ck["autoencoder_state_dict"].values())
    S, D, U = h.dense_constants()
    ref_cpu.SpiralAEOracle(FE, FD, 16, h.sizes, h.spiral_sizes, S, D, U).load_state_dict(ck["autoencoder_state_dict"])
    m2 = sh.SpiralAutoencoder(FE, FD, 16, h.sizes, h.spiral_sizes, h.spirals, h.D, h.U, dev)
    opt2 = torch.optim.Adam(m2.parameters(), lr=1e-3, weight_decay=5e-5)
    sched2 = torch.optim.lr_scheduler.StepLR(opt2, 1, gamma=0.99)
    assert train_funcs.load_checkpoint(tmp_path / "checkpoint3.pth.tar", m2, opt2, sched2) == 4
    assert opt2.param_groups[0]["lr"] == pytest.approx(opt.param_groups[0]["lr"])
    x = torch.from_numpy(g["x_val"]).to(dev)
    assert torch.equal(m2(x)[0], m(x)[0])


@pytest.mark.gpu
def test_hip_step_through_rccl_reducer_matches_plain_step(golden_dir):
    """One rank, backend "nccl" (= RCCL): the bucketed / in-place gradient all-reduce with the collectives forced on must
    leave a training step unchanged (average over one rank), with the library's Adam reading the reduced gradients."""
    import torch.distributed as dist
    import semantichuman_amd as sh
    from semantichuman_amd.parallel import GradientAllReducer
    g, g0, h, _ = golden(golden_dir)
    dev = torch.device("cuda:0")
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29541")
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    try:
        x = torch.from_numpy(g["x_train"][:2]).to(dev)
        ft = sh.FaceTables(h.faces, h.sizes[0] + 1, dev)
        ends = []
        for use_reducer in (False, True):
            m = sh.SpiralAutoencoder(FE, FD, 16, h.sizes, h.spiral_sizes, h.spirals, h.D, h.U, dev)
            m.load_state_dict({k[3:]: torch.from_numpy(g0[k]) for k in g0.files if k.startswith("w0/")})
            opt = sh.optim.Adam(m.parameters(), lr=1e-3, weight_decay=5e-5)
            red = GradientAllReducer(m, bucket_cap_mb=0.05, inplace_min_mb=0.05, force_collectives=True) if use_reducer else None
            if red is not None:
                assert red.active and any(b.inplace for b in red.buckets) and any(not b.inplace for b in red.buckets)
            for _ in range(2):
                opt.zero_grad(set_to_none=True)
                loss, _ = sh.recon_loss(m(x)[0], x, ft, 1e-2)
                if red is not None:
                    red.prepare()
                loss.backward()
                if red is not None:
                    red.finish()
                opt.step()
            torch.cuda.synchronize()
            ends.append([p.detach().clone() for p in m.parameters()])
        for a, b in zip(*ends):
            assert torch.equal(a, b)
    finally:
        dist.destroy_process_group()

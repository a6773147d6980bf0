"""Data-parallel gradient averaging (semantichuman_amd.parallel) with two processes over gloo
on CPU.  The compute under test here is the sharding / bucketing / overlapped all-reduce logic,
which is model-agnostic; the model used as the stand-in is the CPU oracle (tests may use it)."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FE = [[3, 16, 32, 64, 128], [[], [], [], [], []]]
FD = [[128, 64, 32, 32, 16], [[], [], [], [], 3]]


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _build(golden_dir):
    from oracle import ref_cpu
    from semantichuman_amd.hierarchy import load_hierarchy
    g0 = np.load(os.path.join(golden_dir, "small_ae.npz"))
    h = load_hierarchy(os.path.join(golden_dir, "small_ae.npz"))
    S, D, U = h.dense_constants()
    m = ref_cpu.SpiralAEOracle(FE, FD, 16, h.sizes, h.spiral_sizes, S, D, U)
    m.load_state_dict({k[3:]: torch.from_numpy(g0[k]) for k in g0.files if k.startswith("w0/")})
    return m, h


def _worker(rank, world, port, golden_dir, overlap, out_dir):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(1)
    from semantichuman_amd import synthetic
    from semantichuman_amd.parallel import GradientAllReducer, all_reduce_mean_scalar, shard_batch
    m, h = _build(golden_dir)
    # small cap -> several packed buckets; gradients >= 50 KB (the two latent FCs here) are reduced in place
    red = GradientAllReducer(m, bucket_cap_mb=0.05, overlap=overlap, inplace_min_mb=0.05)
    assert len(red.buckets) > 3 and red.message_bytes == sum(p.numel() * 4 for p in m.parameters())
    assert sum(b.inplace for b in red.buckets) >= 2 and any(not b.inplace for b in red.buckets)
    x = torch.from_numpy(synthetic.synth_batch(h.verts, 4, seed=5))
    xs = x[shard_batch(4, rank, world)]
    opt = torch.optim.Adam(m.parameters(), lr=1e-3, weight_decay=5e-5)
    losses = []
    for _ in range(2):                                                          # two steps: hooks must re-arm
        opt.zero_grad()
        x_hat, _ = m(xs)
        loss = torch.nn.functional.l1_loss(xs, x_hat)
        red.prepare()
        loss.backward()
        red.finish()
        opt.step()
        losses.append(float(all_reduce_mean_scalar(loss.detach())))
    torch.save({"w": {k: v.clone() for k, v in m.state_dict().items()}, "loss": losses}, os.path.join(out_dir, "r%d.pt" % rank))
    dist.destroy_process_group()


@pytest.mark.parametrize("overlap", [True, False])
def test_two_rank_data_parallel_equals_single_process(golden_dir, tmp_path, overlap):
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), golden_dir, overlap, str(tmp_path)), nprocs=world, join=True)
    r0, r1 = (torch.load(tmp_path / ("r%d.pt" % r), weights_only=False) for r in range(2))
    for k in r0["w"]:
        assert torch.equal(r0["w"][k], r1["w"][k]), k                          # replicas stay identical
    # single process on the full batch: mean of shard means == mean over the global batch
    from semantichuman_amd import synthetic
    m, h = _build(golden_dir)
    x = torch.from_numpy(synthetic.synth_batch(h.verts, 4, seed=5))
    opt = torch.optim.Adam(m.parameters(), lr=1e-3, weight_decay=5e-5)
    ref_losses = []
    for _ in range(2):
        opt.zero_grad()
        loss = torch.nn.functional.l1_loss(x, m(x)[0])
        loss.backward()
        opt.step()
        ref_losses.append(float(loss))
    # step 1: mean of shard means == mean over the global batch, to rounding.  Step 2 already carries the first Adam
    # update, whose +-lr moves on ~zero gradients depend on fp32 summation order (and on the thread count of the CPU
    # reductions): a looser bound
    assert r0["loss"][0] == pytest.approx(ref_losses[0], rel=1e-5)
    assert r0["loss"][1] == pytest.approx(ref_losses[1], rel=2e-3)
    for k, v in m.state_dict().items():
        d = (r0["w"][k] - v).abs()
        # Adam turns fp32 sum-order noise on ~zero gradients into +-lr steps for a few elements:
        # bulk must agree tightly, the worst element by less than the 2 steps x lr it can move
        assert float(d.mean()) <= 1e-7 and float(d.max()) <= 2.1e-3, k


def _worker_bf16(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from semantichuman_amd.parallel import GradientAllReducer
    torch.manual_seed(0)
    m = torch.nn.Sequential(torch.nn.Linear(64, 512), torch.nn.Linear(512, 8))      # one "large" weight, three small tensors
    red = GradientAllReducer(m, inplace_min_mb=0.1, large_message_dtype=torch.bfloat16)
    assert [b.inplace for b in red.buckets] == [True, False]
    torch.manual_seed(10 + rank)
    x = torch.randn(16, 64)
    red.prepare()
    m(x).square().mean().backward()
    local = [p.grad.clone() for p in m.parameters()]
    red.finish()
    torch.save({"local": local, "reduced": [p.grad.clone() for p in m.parameters()]}, os.path.join(out_dir, "b%d.pt" % rank))
    dist.destroy_process_group()


def test_bf16_large_messages(tmp_path):
    """large_message_dtype=bfloat16: the large gradient is averaged through a bf16 message (relative error <= 2^-8 per
    element, identical on both ranks, still an fp32 .grad); the packed small gradients stay exact fp32."""
    mp.spawn(_worker_bf16, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    r0, r1 = (torch.load(tmp_path / ("b%d.pt" % r), weights_only=False) for r in range(2))
    for i, (a, b) in enumerate(zip(r0["reduced"], r1["reduced"])):
        assert torch.equal(a, b) and a.dtype == torch.float32
        mean = 0.5 * (r0["local"][i] + r1["local"][i])
        if i == 0:                                                   # the 64 x 512 weight: bf16 message
            assert not torch.equal(a, mean)
            assert float((a - mean).abs().max()) <= 2.0 ** -7 * float(mean.abs().max())
        else:
            assert torch.allclose(a, mean, rtol=1e-6, atol=1e-9)


def test_shard_batch_contract():
    from semantichuman_amd.parallel import shard_batch
    assert [shard_batch(512, r, 8) for r in (0, 7)] == [slice(0, 64), slice(448, 512)]
    with pytest.raises(ValueError):
        shard_batch(10, 0, 4)


def test_shard_order_gives_every_rank_the_same_number_of_batches():
    from semantichuman_amd.dataset import shard_len, shard_order
    order = torch.arange(10)
    for pad, per in ((True, 3), (False, 2)):
        shards = [shard_order(order, r, 4, pad) for r in range(4)]
        assert [s.numel() for s in shards] == [per] * 4 and shard_len(10, 4, pad) == per
        seen = torch.cat(shards).tolist()
        assert set(seen) == (set(range(10)) if pad else set(range(8)))       # padding repeats, never invents, samples
    assert torch.equal(shard_order(order, 0, 1), order)
    assert [shard_order(torch.arange(2), r, 5).numel() for r in range(5)] == [1] * 5   # fewer samples than ranks: wrap around
    with pytest.raises(ValueError):
        shard_order(torch.arange(2), 0, 5, pad=False)


class _ListLoader:
    """The slice of the loader protocol the training loop uses (iteration, len, .dataset), over a rank's shard."""

    def __init__(self, verts, idx, batch):
        self.dataset = list(range(len(idx)))
        self.batches = [{"verts": verts[idx[i:i + batch]], "idx": idx[i:i + batch]} for i in range(0, len(idx), batch)]

    def __len__(self):
        return len(self.batches)

    def __iter__(self):
        return iter(self.batches)


class _Writer:
    def __init__(self):
        self.rows = []

    def add_scalar(self, tag, v, step):
        self.rows.append((tag, float(v), step))


class _TinyAE(torch.nn.Module):
    def __init__(self):
        super().__init__()
        torch.manual_seed(3)
        self.f = torch.nn.Linear(3, 3)

    def forward(self, x):
        return self.f(x), None


def _loop_worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from types import SimpleNamespace
    from semantichuman_amd.dataset import shard_order
    from semantichuman_amd.parallel import GradientAllReducer
    from semantichuman_amd.train_funcs import train_autoencoder_dataloader
    g = torch.Generator().manual_seed(0)
    verts = torch.randn(5, 7, 3, generator=g)                    # 5 samples over 2 ranks: n % world != 0
    vval = torch.randn(3, 7, 3, generator=g)
    tr = _ListLoader(verts, shard_order(torch.arange(5), rank, world), 1)
    va = _ListLoader(vval, shard_order(torch.arange(3), rank, world), 1)
    assert len(tr) == 3                                          # the same on both ranks, or the loop below deadlocks
    model = _TinyAE()
    opt = torch.optim.SGD(model.parameters(), lr=0.1)
    sched = torch.optim.lr_scheduler.StepLR(opt, 1, gamma=0.99)
    w = _Writer()
    shapedata = SimpleNamespace(reference_mesh=SimpleNamespace(f=np.zeros((1, 3), dtype=np.int32)))
    l1 = lambda a, b: (a - b).abs().mean()                       # noqa: E731 - a torch loss: the loop itself is what is under test
    hist = train_autoencoder_dataloader(tr, va, torch.device("cpu"), model, opt, l1, 1, 2, 1, None, sched, w, shapedata, out_dir,
                                        out_dir, "ck", edgereg_w=0.0, ck_frequency=1, reducer=GradientAllReducer(model), verbose=False)
    torch.save({"hist": hist, "w": model.state_dict(), "rows": w.rows}, os.path.join(out_dir, "loop%d.pt" % rank))
    dist.destroy_process_group()


def test_two_rank_training_loop_with_uneven_dataset(tmp_path):
    """n % world != 0 must not leave ranks with different batch counts (deadlock in the per-batch all-reduce);
    epoch losses are means over ALL ranks' samples; only rank 0 writes checkpoints and logs."""
    mp.spawn(_loop_worker, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    r0, r1 = (torch.load(tmp_path / ("loop%d.pt" % r), weights_only=False) for r in range(2))
    for k in r0["w"]:
        assert torch.equal(r0["w"][k], r1["w"][k])
    assert r0["hist"] == r1["hist"] and len(r0["hist"]) == 2
    assert r1["rows"] == [] and any(t == "avg_epoch_train_loss" for t, _, _ in r0["rows"])
    # epoch loss = sum over both ranks' (padded) shards / 6 samples: of the order of one sample's loss, not 1/world of it
    e1 = r0["hist"][0][1]
    assert 0.3 < e1 < 3.0
    cks = sorted(p.name for p in tmp_path.iterdir() if p.name.startswith("ck"))
    assert cks == ["ck1.pth.tar", "ck2.pth.tar"]
    from semantichuman_amd.train_funcs import load_checkpoint
    m = _TinyAE()
    assert load_checkpoint(str(tmp_path / "ck2.pth.tar"), m) == 3                  # loads with weights_only=True
    assert torch.equal(m.f.weight, r0["w"]["f.weight"])


class _TinySemanticAE(torch.nn.Module):
    """(x, kps) -> (x_hat, ...) like SpiralAutoencoder_multiz_partkps.forward, on torch CPU ops."""

    def __init__(self):
        super().__init__()
        torch.manual_seed(4)
        self.f = torch.nn.Linear(3, 3)

    def forward(self, x, kps=None):
        return self.f(x), None, None


class _SavedMeshes:
    def __init__(self):
        self.reference_mesh = None
        self.calls = []

    def save_meshes(self, stem, arr, ind):
        self.calls.append((os.path.basename(stem), tuple(arr.shape), [int(i) for i in ind]))


def _semantic_loop_worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from types import SimpleNamespace
    from semantichuman_amd import train_semantic as ts
    from semantichuman_amd.dataset import shard_order
    from semantichuman_amd.parallel import GradientAllReducer

    # The loop's data-parallel behaviour is what is under test; its HIP-backed pieces (the per-run context and the loss
    # terms) are replaced by CPU stand-ins with the same interface, in this test process only.
    class Ctx:
        def __init__(self, opts, shapedata, J, parts, names, device):
            self.opts, self.device = opts, device
            self.kps_keep_t = torch.arange(2)

        def joints(self, x):
            return x[:, :3, :]

    def fake_losses(model, ctx, tx, tx_i, tx_e, epoch, measure=None, interp_measure=None, loss_fn=None, **kw):
        tx_hat = model(tx)[0]
        ctx.last_tx_hat = tx_hat.detach()
        rec = (tx - tx_hat).abs().mean()
        return rec, {"rec_loss": rec}
    ts.SemanticContext, ts.semantic_losses = Ctx, fake_losses

    g = torch.Generator().manual_seed(0)
    verts = torch.randn(5, 7, 3, generator=g)                    # 5 samples over 2 ranks: n % world != 0
    vval = torch.randn(3, 7, 3, generator=g)
    tr = _ListLoader(verts, shard_order(torch.arange(5), rank, world), 1)
    va = _ListLoader(vval, shard_order(torch.arange(3), rank, world), 1)
    model = _TinySemanticAE()
    opt = torch.optim.SGD(model.parameters(), lr=0.1)
    sched = torch.optim.lr_scheduler.StepLR(opt, 1, gamma=0.99)
    w, sd = _Writer(), _SavedMeshes()
    opts = ts.SemanticTrainOptions()
    opts.ck_frequency = 50
    l1 = lambda a, b: (a - b).abs().mean()                       # noqa: E731
    hist = ts.train_autoencoder_dataloader_nonormal(tr, va, torch.device("cpu"), model, opt, l1, 49, 50, 1, tr, sched, w, sd, out_dir,
                                                    out_dir, "sck", None, None, None, save_recons=True, options=opts,
                                                    reducer=GradientAllReducer(model), verbose=False)
    torch.save({"hist": hist, "w": model.state_dict(), "rows": w.rows, "saved": sd.calls, "last_val_idx": int(va.batches[-1]["idx"][0])},
               os.path.join(out_dir, "sloop%d.pt" % rank))
    dist.destroy_process_group()


def test_two_rank_semantic_loop_with_uneven_dataset(tmp_path):
    """The semantic loop as a data-parallel citizen: epoch losses are means over ALL ranks' samples,
    rank 0 alone logs, writes the checkpoint (followed by a barrier) and - save_recons, reference train_funcs.py:459-470 -
    dumps the first mesh of the epoch's last training batch (ground truth and reconstruction) under the sample index of
    the last validation batch, at epochs that are multiples of 50."""
    mp.spawn(_semantic_loop_worker, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    r0, r1 = (torch.load(tmp_path / ("sloop%d.pt" % r), weights_only=False) for r in range(2))
    for k in r0["w"]:
        assert torch.equal(r0["w"][k], r1["w"][k])
    assert r0["hist"] == r1["hist"] and [e for e, _, _ in r0["hist"]] == [49, 50]
    assert 0.3 < r0["hist"][0][1] < 3.0                           # a per-sample mean, not 1/world of it
    assert r1["rows"] == [] and any(t == "avg_epoch_valid_loss" for t, _, _ in r0["rows"])
    assert sorted(p.name for p in tmp_path.iterdir() if p.name.startswith("sck")) == ["sck50.pth.tar"]
    assert r1["saved"] == []
    assert r0["saved"] == [("epoch50_GT", (1, 6, 3), [r0["last_val_idx"]]), ("epoch50_rec", (1, 6, 3), [r0["last_val_idx"]])]


def _sharded_worker(rank, world, port, golden_dir, sharded, out_dir):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(1)
    from semantichuman_amd import synthetic
    from semantichuman_amd.parallel import GradientAllReducer, shard_batch
    m, h = _build(golden_dir)
    # bf16 working copies of the large parameters (semantichuman_amd.shadow), as the bf16 compute path keeps them: on this CPU
    # box the conversion kernel is replaced by torch's cast - what is under test is that the sharded update keeps the copies
    # CURRENT (the optimizer only ever sees this rank's slice; the all-gather writes through `p.data`)
    from semantichuman_amd import ops, shadow
    ops.cast_bf16 = lambda src, out=None: (out.copy_(src.to(torch.bfloat16)) if out is not None else src.to(torch.bfloat16))
    big = [p for p in m.parameters() if p.numel() * 4 >= 0.05 * 2 ** 20]
    copies = [shadow.get(p) for p in big]
    addrs = [c.data_ptr() for c in copies]
    red = GradientAllReducer(m, bucket_cap_mb=0.05, inplace_min_mb=0.05, shard_large=sharded)
    params = red.optimizer_params()
    if sharded:
        assert len(red.shards) >= 2 and all(s.numel() * world == p.numel() for p, s in red.shards.items())
        assert sum(p.numel() for p in params) < 0.6 * sum(p.numel() for p in m.parameters())      # the optimizer holds ~1/2 of the model
    opt = torch.optim.Adam(params, lr=1e-3, weight_decay=5e-5)
    x = torch.from_numpy(synthetic.synth_batch(h.verts, 4, seed=5))
    xs = x[shard_batch(4, rank, world)]
    for _ in range(3):
        opt.zero_grad(set_to_none=True)
        for p in m.parameters():
            p.grad = None
        loss = torch.nn.functional.l1_loss(xs, m(xs)[0])
        red.prepare()
        loss.backward()
        red.finish()
        opt.step()
        red.gather_weights()
        if sharded:      # after every step the working copies are the bf16 image of the gathered weights, where they always were
            assert len(big) >= 2
            for p, c, a in zip(big, copies, addrs):
                assert shadow.lookup(p) is c and c.data_ptr() == a
                assert torch.equal(c, p.detach().to(torch.bfloat16))
    state_elems = sum(v.numel() for st in opt.state.values() for v in st.values() if torch.is_tensor(v))
    torch.save({"w": {k: v.clone() for k, v in m.state_dict().items()}, "state_elems": state_elems},
               os.path.join(out_dir, "s%d_r%d.pt" % (int(sharded), rank)))
    dist.destroy_process_group()


def test_sharded_update_of_the_large_parameters_is_bitwise_the_all_reduce_path(golden_dir, tmp_path):
    """VERDICT r3 item 9 (what the CPU can prove): the two latent FC gradients reduce-scattered, Adam on 1 / world of each FC,
    the updated slices all-gathered - after three steps every weight on every rank is BITWISE the weight of the all-reduce
    path, with half the optimizer state per rank - and (round 6) the bf16 working copies the bf16 compute path keeps of those
    parameters are current after every step, at their old addresses (asserted inside the workers)."""
    world = 2
    for sharded in (False, True):
        mp.spawn(_sharded_worker, args=(world, _free_port(), golden_dir, sharded, str(tmp_path)), nprocs=world, join=True)
    ref = [torch.load(tmp_path / ("s0_r%d.pt" % r), weights_only=False) for r in range(world)]
    got = [torch.load(tmp_path / ("s1_r%d.pt" % r), weights_only=False) for r in range(world)]
    for k in ref[0]["w"]:
        for r in range(world):
            assert torch.equal(got[r]["w"][k], ref[0]["w"][k]), (k, r)
    assert got[0]["state_elems"] < 0.6 * ref[0]["state_elems"]

"""GPU-resident data path.

The reference (autoencoder_dataset.py:9-58, main.py:209-237) keeps one .npy per sample on disk,
loads and normalises it with numpy inside `__getitem__`, and moves batches to the device through a
4-worker DataLoader.  A 6890-vertex split is 83 KB per mesh - 4096 meshes are 0.34 GB of the 288 GB
of HBM - so here the split is read once, uploaded packed, normalised and dummy-padded ON DEVICE by
one kernel launch (sh_dataset_normalize), and every batch is a row gather from the resident tensor
(sh_gather_meshes).  The on-disk layout is the reference's:

    root_dir/paths_{split}.npy                  basenames ('000000', ...)
    root_dir/points_{split}/{basename}.npy      [N, 3] vertices
    root_dir/measure_{split}/{basename}.npy     [M] measurements (when measure_flag)

`autoencoder_dataset` keeps the reference's constructor and item protocol (`{'verts', 'idx'[, 'measure']}`);
`ResidentLoader` replaces `torch.utils.data.DataLoader` and yields the same collated dicts, already on
the device.  There is no CPU path for the normalisation: `resident()` needs a HIP device.
"""
from __future__ import annotations

import os

import numpy as np
import torch

from . import ops


def normalization_flags(normalization: str) -> int:
    """The reference tests substrings of one string (autoencoder_dataset.py:29-40), e.g. 'zeroroot'."""
    flags = 0
    for name, bit in ops.NORM_FLAGS.items():
        if name in normalization:
            flags |= bit
    return flags


def write_split(root_dir, points_dataset, verts, measure=None, start=0):
    """Write meshes [n, N, 3] in the reference's on-disk layout (data_generation.py:48-66)."""
    os.makedirs(os.path.join(root_dir, "points_" + points_dataset), exist_ok=True)
    if measure is not None:
        os.makedirs(os.path.join(root_dir, "measure_" + points_dataset), exist_ok=True)
    names = []
    for i in range(len(verts)):
        name = str(start + i).zfill(6)
        np.save(os.path.join(root_dir, "points_" + points_dataset, name + ".npy"), verts[i])
        if measure is not None:
            np.save(os.path.join(root_dir, "measure_" + points_dataset, name + ".npy"), measure[i])
        names.append(name)
    np.save(os.path.join(root_dir, "paths_" + points_dataset + ".npy"), np.asarray(names))


class autoencoder_dataset:
    """Same constructor as reference autoencoder_dataset.py:11.  `shapedata` only needs the attributes the
    selected normalisation reads (`mean`/`std` for 'gass', `center`/`scale` for 'normal')."""

    def __init__(self, root_dir, points_dataset, shapedata, normalization="No", dummy_node=True, measure_flag=False,
                 anglew_flag=False, J_regressor=None):
        self.shapedata = shapedata
        self.normalization = normalization
        self.root_dir = root_dir
        self.points_dataset = points_dataset
        self.dummy_node = dummy_node
        self.paths = np.load(os.path.join(root_dir, "paths_" + points_dataset + ".npy"))
        self.measure_flag = measure_flag
        self.J_regressor = J_regressor
        self.anglew_flag = anglew_flag
        self.verts = None            # [n, N(+1), 3] on device after resident()
        self.measure = None          # [n, M] on device after resident() when measure_flag

    def __len__(self):
        return len(self.paths)

    # -------------------------------------------------------------------------------------- loading
    def read_raw(self):
        """All samples of the split as one float32 [n, N, 3] host array (+ measurements [n, M] or None)."""
        pts = [np.load(os.path.join(self.root_dir, "points_" + self.points_dataset, str(b) + ".npy")) for b in self.paths]
        shapes = {p.shape for p in pts}
        if len(shapes) != 1:
            raise ValueError("meshes of split %r differ in shape: %s" % (self.points_dataset, sorted(shapes)))
        raw = np.stack(pts).astype(np.float32)
        meas = None
        if self.measure_flag:
            meas = np.stack([np.load(os.path.join(self.root_dir, "measure_" + self.points_dataset, str(b) + ".npy"))
                             for b in self.paths]).astype(np.float32)
        return raw, meas

    def resident(self, device):
        """Upload the split and normalise it on `device`; idempotent."""
        device = torch.device(device)
        if self.verts is not None and self.verts.device == device:
            return self
        if device.type != "cuda":
            raise RuntimeError("autoencoder_dataset.resident: the dataset is normalised by a HIP kernel; device %s has no "
                               "such path (there is no CPU fallback)" % device)
        raw, meas = self.read_raw()
        n, N = raw.shape[0], raw.shape[1]
        flags = normalization_flags(self.normalization)

        def dev(a, shape=None):
            t = torch.from_numpy(np.array(a, dtype=np.float32)).to(device)
            return t if shape is None else t.reshape(shape).contiguous()
        kw = {}
        if flags & ops.NORM_FLAGS["zeroroot"]:
            if self.J_regressor is None:
                raise ValueError("normalization %r needs J_regressor" % self.normalization)
            kw["j_root"] = dev(np.asarray(self.J_regressor)[0], (N,))
        if flags & ops.NORM_FLAGS["gass"]:
            kw["mean"], kw["std"] = dev(self.shapedata.mean, (N, 3)), dev(self.shapedata.std, (N, 3))
        if flags & ops.NORM_FLAGS["normal"]:
            c = np.asarray(self.shapedata.center, dtype=np.float32)[:n]
            s = np.broadcast_to(np.asarray(self.shapedata.scale, dtype=np.float32)[:n].reshape(n, -1), (n, 3))
            kw["center"], kw["scale"] = dev(c, (n, 3)), dev(s, (n, 3))
        self.verts = ops.dataset_normalize(torch.from_numpy(raw).to(device), flags, dummy_rows=1 if self.dummy_node else 0, **kw)
        self.measure = None if meas is None else torch.from_numpy(meas).to(device)
        return self

    def __getitem__(self, idx):
        if self.verts is None:
            raise RuntimeError("autoencoder_dataset: call resident(device) first; samples live on the GPU")
        item = {"verts": self.verts[idx], "idx": idx}
        if self.measure_flag:
            item["measure"] = self.measure[idx]
        return item


def shard_len(n: int, world_size: int, pad: bool = True) -> int:
    """Samples per rank and epoch - the same on every rank."""
    return -(-n // world_size) if pad else n // world_size


def shard_order(order: torch.Tensor, rank: int, world_size: int, pad: bool = True) -> torch.Tensor:
    """This rank's strided shard of an epoch's sample order, equal in length on every rank (ranks that ran different
    numbers of batches would leave the longer ones blocked in the per-batch gradient all-reduce).  pad: repeat samples
    from the start of the order until its length divides (torch DistributedSampler, drop_last=False); else drop the tail."""
    if not 0 <= rank < world_size:
        raise ValueError("rank %d outside world of %d" % (rank, world_size))
    n = order.numel()
    if world_size == 1:
        return order
    per = shard_len(n, world_size, pad)
    if per == 0:
        raise ValueError("%d samples cannot be sharded over %d ranks without padding" % (n, world_size))
    total = per * world_size
    if total > n:
        reps = -(-total // n)
        order = order.repeat(reps)[:total] if reps > 1 else torch.cat([order, order[:total - n]])
    else:
        order = order[:total]
    return order[rank::world_size]


class ResidentLoader:
    """DataLoader replacement for a resident `autoencoder_dataset`: iterating yields the dicts the reference's
    default collate produces (`verts` [b, N+1, 3], `idx` int64 [b], `measure` [b, M]) on the device.
    With torch.distributed, pass rank/world_size to iterate a disjoint strided shard of each epoch's order.  Every rank
    gets the SAME number of samples (hence of batches - the training loops issue one collective per batch): the epoch's
    order is padded by wrapping around, as torch's DistributedSampler does (`pad=True`, default), or truncated to a
    multiple of the world size (`pad=False`)."""

    def __init__(self, dataset, batch_size=1, shuffle=False, device=None, drop_last=False, seed=0, rank=0, world_size=1,
                 pad=True):
        if device is None and dataset.verts is None:
            raise ValueError("ResidentLoader: pass a device or make the dataset resident first")
        self.dataset = dataset.resident(device if device is not None else dataset.verts.device)
        self.batch_size, self.shuffle, self.drop_last = int(batch_size), shuffle, drop_last
        self.rank, self.world_size, self.pad = rank, world_size, pad
        self._gen = torch.Generator().manual_seed(seed)
        self.epoch = 0

    def _order(self):
        n = len(self.dataset)
        order = torch.randperm(n, generator=self._gen) if self.shuffle else torch.arange(n)
        return shard_order(order, self.rank, self.world_size, self.pad)

    def __len__(self):
        n = shard_len(len(self.dataset), self.world_size, self.pad)
        return n // self.batch_size if self.drop_last else -(-n // self.batch_size)

    def __iter__(self):
        order = self._order().to(self.dataset.verts.device)
        self.epoch += 1
        for lo in range(0, order.numel(), self.batch_size):
            idx = order[lo:lo + self.batch_size].contiguous()
            if self.drop_last and idx.numel() < self.batch_size:
                return
            batch = {"verts": ops.gather_meshes(self.dataset.verts, idx), "idx": idx}
            if self.dataset.measure_flag:
                batch["measure"] = ops.gather_meshes(self.dataset.measure, idx)
            yield batch

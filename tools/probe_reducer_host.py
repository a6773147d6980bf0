#!/usr/bin/env python3
"""Where the host time of one eagerly launched data-parallel step goes (world of one rank on a 1-GPU box).
python tools/probe_reducer_host.py [batch]"""
import os
import sys
import time

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import semantichuman_amd as sh                                   # noqa: E402
from semantichuman_amd import synthetic                          # noqa: E402
from semantichuman_amd.hierarchy import load_hierarchy           # noqa: E402
from semantichuman_amd.parallel import GradientAllReducer        # noqa: E402

FE = [[3, 16, 32, 64, 128], [[], [], [], [], []]]
FD = [[128, 64, 32, 32, 16], [[], [], [], [], 3]]
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
USE = os.environ.get("PROBE_REDUCER", "1") != "0"
for k, v in (("MASTER_ADDR", "127.0.0.1"), ("MASTER_PORT", "29545"), ("RANK", "0"), ("WORLD_SIZE", "1")):
    os.environ.setdefault(k, v)
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
dist.init_process_group("nccl", device_id=dev)
h = load_hierarchy(os.path.join(ROOT, "tests", "golden", "template6890.npz"))
torch.manual_seed(2)
model = sh.SpiralAutoencoder(FE, FD, 256, h.sizes, h.spiral_sizes, h.spirals, h.D, h.U, dev)
optim = sh.optim.Adam(model.parameters(), lr=1e-3, weight_decay=5e-5)
ft = sh.FaceTables(h.faces, h.sizes[0] + 1, dev)
red = GradientAllReducer(model, force_collectives=True) if USE else None
x = torch.from_numpy(synthetic.synth_batch(h.verts, B, seed=1)).to(dev)
acc = {}


def timed(name, fn):
    def w(*a, **k):
        t0 = time.perf_counter()
        r = fn(*a, **k)
        acc[name] = acc.get(name, 0.0) + time.perf_counter() - t0
        return r
    return w


if red:
    red._launch = timed("reducer._launch (3 calls)", red._launch)


def step():
    t0 = time.perf_counter()
    optim.zero_grad(set_to_none=True)
    xh, _ = model(x)
    loss, _ = sh.recon_loss(xh, x, ft, 1e-2)
    t1 = time.perf_counter()
    if red:
        red.prepare()
    loss.backward()
    t2 = time.perf_counter()
    if red:
        red.finish()
    t3 = time.perf_counter()
    optim.step()
    t4 = time.perf_counter()
    for k, v in (("forward+loss", t1 - t0), ("backward (incl. hooks)", t2 - t1), ("finish", t3 - t2), ("adam", t4 - t3), ("step", t4 - t0)):
        acc[k] = acc.get(k, 0.0) + v


for _ in range(30):
    step()
torch.cuda.synchronize()
acc.clear()
N = 300
t0 = time.perf_counter()
for _ in range(N):
    step()
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print("batch %d reducer %s: issue %.3f ms/step, drained %.3f ms/step" % (B, bool(red), 1e3 * (t1 - t0) / N, 1e3 * (t2 - t0) / N))
for k, v in acc.items():
    print("  %-28s %7.1f us/step" % (k, 1e6 * v / N))
dist.destroy_process_group()

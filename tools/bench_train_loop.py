#!/usr/bin/env python3
"""Throughput of the DROP-IN training loop (train_funcs.train_autoencoder_dataloader with the reference's
signature, a ResidentLoader over a split written in the reference's on-disk layout, StepLR, per-epoch validation)
- the same step bench.py times, but issued eagerly by the loop a user of the reference would run.
python tools/bench_train_loop.py [--meshes 1024] [--epochs 10]"""
import argparse
import json
import os
import sys
import tempfile
import time
from types import SimpleNamespace

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import semantichuman_amd as sh                                                  # noqa: E402
from semantichuman_amd import synthetic, train_funcs                            # noqa: E402
from semantichuman_amd.dataset import ResidentLoader, autoencoder_dataset, write_split   # noqa: E402
from semantichuman_amd.hierarchy import load_hierarchy                          # noqa: E402

FE = [[3, 16, 32, 64, 128], [[], [], [], [], []]]
FD = [[128, 64, 32, 32, 16], [[], [], [], [], 3]]
ap = argparse.ArgumentParser()
ap.add_argument("--meshes", type=int, default=1024)
ap.add_argument("--epochs", type=int, default=10)
ap.add_argument("--batch", type=int, default=64)
args = ap.parse_args()
dev = torch.device("cuda", 0)
h = load_hierarchy(os.path.join(ROOT, "tests", "golden", "template6890.npz"))
torch.manual_seed(2)
model = sh.SpiralAutoencoder(FE, FD, 256, h.sizes, h.spiral_sizes, h.spirals, h.D, h.U, dev)
optim = sh.optim.Adam(model.parameters(), lr=1e-3, weight_decay=5e-5)
sched = torch.optim.lr_scheduler.StepLR(optim, 1, gamma=0.99)
shapedata = SimpleNamespace(reference_mesh=SimpleNamespace(f=h.faces))
with tempfile.TemporaryDirectory() as root:
    write_split(root, "train", synthetic.synth_batch(h.verts, args.meshes, seed=100)[:, :-1])
    write_split(root, "val", synthetic.synth_batch(h.verts, args.batch, seed=7)[:, :-1])
    ltr = ResidentLoader(autoencoder_dataset(root, "train", shapedata), batch_size=args.batch, shuffle=True, device=dev)
    lva = ResidentLoader(autoencoder_dataset(root, "val", shapedata), batch_size=args.batch, shuffle=False, device=dev)


def run(first, last):
    return train_funcs.train_autoencoder_dataloader(ltr, lva, dev, model, optim, torch.nn.functional.l1_loss, first, last, 10,
                                                    None, sched, None, shapedata, None, None, "checkpoint", verbose=False)


run(1, 2)                                    # warm-up epochs
torch.cuda.synchronize()
t0 = time.perf_counter()
hist = run(3, 2 + args.epochs)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
iters = args.epochs * len(ltr)
print(json.dumps({"metric": "drop-in training loop, 6890 vertices, batch %d" % args.batch, "epochs": args.epochs,
                  "iterations": iters, "ms_per_iteration_incl_validation": 1e3 * dt / iters,
                  "meshes_per_s": args.epochs * args.meshes / dt, "train_loss_first": hist[0][1], "train_loss_last": hist[-1][1],
                  "val_loss_last": hist[-1][2], "launch": "eager", "dtype": "f32", "data": "synthetic"}))

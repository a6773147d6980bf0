#!/usr/bin/env python3
"""Shader clock the MI355X grants the training step in steady state: after `--seconds` of back-to-back training
steps (eager launches, batch 64) a probe kernel (sh_clock_probe: dependent fp32 MFMAs timed with s_memtime against the
100 MHz s_memrealtime) runs between steps; the median over its workgroups and over the probes is the clock, and
64 FLOP/clk/SIMD x 1024 SIMDs x clock is the fp32 MFMA ceiling at that clock.  Also probes an idle GPU for contrast.
    python tools/clock_probe.py [--seconds 3]"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import semantichuman_amd as sh                                   # noqa: E402
from semantichuman_amd import _lib, synthetic                    # noqa: E402
from semantichuman_amd.hierarchy import load_hierarchy           # noqa: E402

FE = [[3, 16, 32, 64, 128], [[], [], [], [], []]]
FD = [[128, 64, 32, 32, 16], [[], [], [], [], 3]]
ap = argparse.ArgumentParser()
ap.add_argument("--seconds", type=float, default=3.0)
ap.add_argument("--batch", type=int, default=64)
a = ap.parse_args()
dev = torch.device("cuda:0")
lib = _lib.load()
NWG, ITERS = 64, 4000
buf = torch.zeros(2 * NWG, dtype=torch.int64, device=dev)


def probe():
    _lib.check(lib.sh_clock_probe(_lib.ptr(buf), NWG, ITERS, _lib.stream_ptr()), "sh_clock_probe")
    v = buf.cpu().numpy().reshape(NWG, 2).astype(np.float64)          # synchronises
    return float(np.median(v[:, 0] / v[:, 1]) * 100.0)                 # MHz


idle = [probe() for _ in range(5)]
h = load_hierarchy(os.path.join(ROOT, "tests", "golden", "template6890.npz"))
torch.manual_seed(2)
model = sh.SpiralAutoencoder(FE, FD, 256, h.sizes, h.spiral_sizes, h.spirals, h.D, h.U, dev)
optim = sh.optim.Adam(model.parameters(), lr=1e-3, weight_decay=5e-5)
ft = sh.FaceTables(h.faces, h.sizes[0] + 1, dev)
x = torch.from_numpy(synthetic.synth_batch(h.verts, a.batch, seed=1)).to(dev)


def step():
    optim.zero_grad(set_to_none=True)
    xh, _ = model(x)
    loss, _ = sh.recon_loss(xh, x, ft, 1e-2)
    loss.backward()
    optim.step()


t0 = time.perf_counter()
n = 0
while time.perf_counter() - t0 < a.seconds:
    for _ in range(50):
        step()
    n += 50
busy = []
for _ in range(40):                                   # probes interleaved with further steps, no idle time in between
    for _ in range(10):
        step()
    busy.append(probe())
mhz = float(np.median(busy))
print(json.dumps({"idle_probe_mhz": float(np.median(idle)), "under_training_load_mhz": mhz,
                  "under_load_min_max_mhz": [float(min(busy)), float(max(busy))],
                  "fp32_mfma_ceiling_tflops_at_that_clock": 64 * 1024 * mhz * 1e6 / 1e12,
                  "steps_before_probing": n, "batch": a.batch}))

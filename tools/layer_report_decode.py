#!/usr/bin/env python3
"""Per-launch timing of one decode batch (library HIP events).  python tools/layer_report_decode.py [batch]"""
import os
import re
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import semantichuman_amd as sh                                   # noqa: E402
from semantichuman_amd import _lib                               # noqa: E402
from semantichuman_amd.hierarchy import load_hierarchy           # noqa: E402

FE = [[3, 16, 32, 64, 128], [[], [], [], [], []]]
FD = [[128, 64, 32, 32, 16], [[], [], [], [], 3]]
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
dev = torch.device("cuda:0")
h = load_hierarchy(os.path.join(ROOT, "tests", "golden", "template6890.npz"))
torch.manual_seed(2)
m = sh.SpiralAutoencoder(FE, FD, 256, h.sizes, h.spiral_sizes, h.spirals, h.D, h.U, dev)
z = torch.randn(B, 256, generator=torch.Generator().manual_seed(0)).to(dev)
with torch.no_grad():
    for _ in range(3):
        m.decode(z)
    torch.cuda.synchronize()
    N = 5
    _lib.profile_enable(True)
    for _ in range(N):
        m.decode(z)
    torch.cuda.synchronize()
recs = _lib.profile_records()
_lib.profile_enable(False)
per = len(recs) // N
tot = 0.0
print("%-44s %-44s %8s %7s" % ("kernel", "shape", "us", "TF/s"))
for i in range(per):
    name = recs[i][0]
    us = 1e3 * sum(recs[i + k * per][1] for k in range(N)) / N
    tot += us
    kern, _, shape = name.partition("|")
    tf = ""
    mm = re.search(r"R=(\d+) B=(\d+) K=(\d+) N=(\d+)", shape)
    if mm:
        R, Bb, K, Nn = map(int, mm.groups())
        tf = "%.1f" % (2.0 * R * Bb * K * Nn / (us * 1e-6) / 1e12)
    mm = re.search(r"M=(\d+) N=(\d+) K=(\d+)", shape)
    if mm:
        M, Nn, K = map(int, mm.groups())
        tf = "%.1f" % (2.0 * M * Nn * K / (us * 1e-6) / 1e12)
    print("%-44s %-44s %8.1f %7s" % (kern.replace("_kernel", ""), shape, us, tf))
print("total library kernels: %.1f us/batch" % tot)

#!/usr/bin/env python3
"""Kernel times of the fused training loss (recon_partial / recon_final / recon_bwd) at BASELINE config 2's shape."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import semantichuman_amd as sh
from semantichuman_amd import _lib, synthetic
from semantichuman_amd.hierarchy import load_hierarchy
dev = torch.device("cuda:0")
h = load_hierarchy(os.path.join(ROOT, "tests", "golden", "template6890.npz"))
ft = sh.FaceTables(h.faces, h.sizes[0] + 1, dev)
x = torch.from_numpy(synthetic.synth_batch(h.verts, 64, seed=1)).to(dev)
xh = (x + 0.01 * torch.randn_like(x)).requires_grad_(True)
def step():
    xh.grad = None
    loss, _ = sh.recon_loss(xh, x, ft, 1e-2)
    loss.backward()
for _ in range(3):
    step()
torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
N = 20
with profile(activities=[ProfilerActivity.CUDA]) as prof:
    for _ in range(N):
        step()
    torch.cuda.synchronize()
for e in prof.key_averages():
    if "recon" in e.key:
        print("%-60s %.1f us" % (e.key[:60], e.device_time_total / N))

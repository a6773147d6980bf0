#!/usr/bin/env python3
"""How the CPU oracle's training step scales with torch's thread count on this box (picks bench.py's default)."""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import ref_cpu
from semantichuman_amd import synthetic
from semantichuman_amd.hierarchy import load_hierarchy
FE = [[3, 16, 32, 64, 128], [[], [], [], [], []]]; FD = [[128, 64, 32, 32, 16], [[], [], [], [], 3]]
h = load_hierarchy(os.path.join(ROOT, "tests", "golden", "template6890.npz"))
S, D, U = h.dense_constants()
B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
x = torch.from_numpy(synthetic.synth_batch(h.verts, B, seed=1))
print("cpu_count", os.cpu_count(), "affinity", len(os.sched_getaffinity(0)))
for nt in (8, 16, 32, 64, 128, 256):
    if nt > (os.cpu_count() or 1):
        break
    torch.set_num_threads(nt)
    torch.manual_seed(2)
    om = ref_cpu.SpiralAEOracle(FE, FD, 256, h.sizes, h.spiral_sizes, S, D, U)
    opt = torch.optim.Adam(om.parameters(), lr=1e-3, weight_decay=5e-5)
    ref_cpu.train_step(om, opt, x, faces=h.faces, edgereg_w=1e-2)
    t0 = time.perf_counter()
    ref_cpu.train_step(om, opt, x, faces=h.faces, edgereg_w=1e-2)
    dt = time.perf_counter() - t0
    print("threads %3d  B=%d  %.2f s/step  %.2f meshes/s" % (nt, B, dt, B / dt), flush=True)

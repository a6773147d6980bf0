#!/usr/bin/env python3
"""Which tensor-library launches (fills, copies, ...) are left in the plain autoencoder's training step, and where from."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import semantichuman_amd as sh
from semantichuman_amd import _lib, synthetic
from semantichuman_amd.hierarchy import load_hierarchy
from torch.profiler import profile, ProfilerActivity
FE = [[3, 16, 32, 64, 128], [[], [], [], [], []]]
FD = [[128, 64, 32, 32, 16], [[], [], [], [], 3]]
dev = torch.device("cuda:0")
h = load_hierarchy(os.path.join(ROOT, "tests", "golden", "template6890.npz"))
_lib.set_f32_mma_mode("planes3")
m = sh.SpiralAutoencoder(FE, FD, 256, h.sizes, h.spiral_sizes, h.spirals, h.D, h.U, dev)
opt = sh.optim.Adam(m.parameters(), lr=1e-3, weight_decay=5e-5)
ft = sh.FaceTables(h.faces, h.sizes[0] + 1, dev)
x = torch.from_numpy(synthetic.synth_batch(h.verts, 64, seed=1)).to(dev)
xin = torch.empty_like(x)
def step():
    xin.copy_(x)
    opt.zero_grad(set_to_none=True)
    loss, _ = sh.recon_loss(m(xin)[0], xin, ft, 1e-2)
    loss.backward()
    opt.step()
for _ in range(5):
    step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    for _ in range(3):
        step()
    torch.cuda.synchronize()
rows = {}
for e in prof.events():
    if not e.name.startswith("aten::") or not e.kernels:
        continue
    chain, q = [], e.cpu_parent
    while q is not None and len(chain) < 5:
        chain.append(q.name.replace("autograd::engine::evaluate_function: ", "bwd "))
        q = q.cpu_parent
    key = (e.name, " < ".join(chain)[:160])
    r = rows.setdefault(key, [0, 0.0])
    r[0] += 1
    r[1] += sum(k.duration for k in e.kernels)
for (name, frame), (n, us) in sorted(rows.items(), key=lambda kv: -kv[1][1]):
    print("%5.1f launches/step %7.1f us/step  %-24s %s" % (n / 3, us / 3, name, frame))

#!/usr/bin/env python3
"""Per-launch timing of one training step - forward, loss, backward, Adam - (library HIP events), with shapes and achieved TFLOP/s.
Run on the GPU box:  python tools/layer_report.py [batch] [template.npz] [f32|bf16]"""
import os
import re
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import semantichuman_amd as sh                                   # noqa: E402
from semantichuman_amd import _lib, synthetic                    # noqa: E402
from semantichuman_amd.hierarchy import load_hierarchy           # noqa: E402

FE = [[3, 16, 32, 64, 128], [[], [], [], [], []]]
FD = [[128, 64, 32, 32, 16], [[], [], [], [], 3]]
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
dev = torch.device("cuda:0")
TEMPLATE = sys.argv[2] if len(sys.argv) > 2 else os.path.join("tests", "golden", "template6890.npz")
h = load_hierarchy(TEMPLATE if os.path.isabs(TEMPLATE) else os.path.join(ROOT, TEMPLATE))
torch.manual_seed(2)
model = sh.SpiralAutoencoder(FE, FD, 256, h.sizes, h.spiral_sizes, h.spirals, h.D, h.U, dev)
DTYPE = sys.argv[3] if len(sys.argv) > 3 else "f32"
if DTYPE == "bf16":
    model.set_compute_dtype(torch.bfloat16)
ft = sh.FaceTables(h.faces, h.sizes[0] + 1, dev)
x = torch.from_numpy(synthetic.synth_batch(h.verts, B, seed=1)).to(dev)


# the step bench.py replays: fused reconstruction loss, Adam with the latent FCs' update inside their weight-gradient kernels
# (SH_LAYER_REPORT_TWO_KERNEL_ADAM=1: gradients written, multi-tensor Adam for everything)
opt = sh.optim.Adam(model.parameters(), lr=1e-3, weight_decay=5e-5)
if os.environ.get("SH_LAYER_REPORT_TWO_KERNEL_ADAM", "0") == "0":
    opt.fuse_linear_weight_gradients([model.fc_latent_enc, model.fc_latent_dec])


def step():
    opt.zero_grad(set_to_none=True)
    xh, _ = model(x)
    loss, _ = sh.recon_loss(xh, x, ft, 1e-2)
    loss.backward()
    opt.step()


for _ in range(3):
    step()
torch.cuda.synchronize()
N = 5
_lib.profile_enable(True)
for _ in range(N):
    step()
torch.cuda.synchronize()
recs = _lib.profile_records()
_lib.profile_enable(False)
per = len(recs) // N
tot = 0.0
print("%-44s %-40s %8s %7s" % ("kernel", "shape", "us", "TF/s"))
for i in range(per):
    name = recs[i][0]
    us = 1e3 * sum(recs[i + k * per][1] for k in range(N)) / N
    tot += us
    kern, _, shape = name.partition("|")
    tf = ""
    m = re.search(r"R=(\d+) B=(\d+) K=(\d+) N=(\d+)", shape)
    m2 = re.search(r"R=(\d+) B=(\d+) S=(\d+) C(?:g|in)=(\d+) N=(\d+)", shape)
    if m:
        R, Bb, K, Nn = map(int, m.groups())
        tf = "%.1f" % (2.0 * R * Bb * K * Nn / (us * 1e-6) / 1e12)
    elif m2:
        R, Bb, S, C, Nn = map(int, m2.groups())
        tf = "%.1f" % (2.0 * R * Bb * S * C * Nn / (us * 1e-6) / 1e12)
    print("%-44s %-40s %8.1f %7s" % (kern.replace("_kernel", ""), shape, us, tf))
print("total library kernels: %.1f us/step" % tot)

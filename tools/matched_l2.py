#!/usr/bin/env python3
"""'At matched L2' (BASELINE north star): train the plain autoencoder for the same steps from the same weights on the same
synthetic batches with (a) the HIP path and (b) the CPU oracle (the reference's formulation), then compare the evaluation
metric of test_funcs.py:41-49 (L1, per-vertex L2 in mm) on held-out meshes.  One JSON line.
    python tools/matched_l2.py [--steps 30] [--batch 16]"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import semantichuman_amd as sh                                   # noqa: E402
from oracle import ref_cpu                                        # noqa: E402  (checker only)
from semantichuman_amd import synthetic                           # noqa: E402
from semantichuman_amd.hierarchy import load_hierarchy            # noqa: E402

FE = [[3, 16, 32, 64, 128], [[], [], [], [], []]]
FD = [[128, 64, 32, 32, 16], [[], [], [], [], 3]]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--batch", type=int, default=16)
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    h = load_hierarchy(os.path.join(ROOT, "tests", "golden", "template6890.npz"))
    torch.manual_seed(2)
    m = sh.SpiralAutoencoder(FE, FD, 256, h.sizes, h.spiral_sizes, h.spirals, h.D, h.U, dev)
    init = {k: v.detach().cpu().clone() for k, v in m.state_dict().items()}
    B = a.batch
    data = torch.from_numpy(synthetic.synth_batch(h.verts, 4 * B, seed=100))
    test = torch.from_numpy(synthetic.synth_batch(h.verts, B, seed=7))
    ft = sh.FaceTables(h.faces, h.sizes[0] + 1, dev)
    opt = sh.optim.Adam(m.parameters(), lr=1e-3, weight_decay=5e-5)
    dd = data.to(dev)
    t0 = time.perf_counter()
    for i in range(a.steps):
        x = dd[(i % 4) * B:(i % 4 + 1) * B]
        opt.zero_grad(set_to_none=True)
        loss, _ = sh.recon_loss(m(x)[0], x, ft, 1e-2)
        loss.backward()
        opt.step()
    torch.cuda.synchronize()
    t_gpu = time.perf_counter() - t0
    with torch.no_grad():
        xt = test.to(dev)
        xh = m(xt)[0]
        g_l1, g_l2 = float(sh.eval_l1(xh, xt)), float(sh.vertex_l2_mm(xh, xt))
    torch.set_num_threads(16)
    S, D, U = h.dense_constants()
    om = ref_cpu.SpiralAEOracle(FE, FD, 256, h.sizes, h.spiral_sizes, S, D, U)
    om.load_state_dict(init)
    oopt = torch.optim.Adam(om.parameters(), lr=1e-3, weight_decay=5e-5)
    t0 = time.perf_counter()
    for i in range(a.steps):
        ref_cpu.train_step(om, oopt, data[(i % 4) * B:(i % 4 + 1) * B], faces=h.faces, edgereg_w=1e-2)
    t_cpu = time.perf_counter() - t0
    with torch.no_grad():
        c_l1, c_l2 = (float(v) for v in ref_cpu.eval_metrics(om(test)[0], test))
    print(json.dumps({"steps": a.steps, "batch": B, "hip": {"eval_l1": g_l1, "eval_l2_mm": g_l2, "seconds": t_gpu},
                      "cpu_oracle": {"eval_l1": c_l1, "eval_l2_mm": c_l2, "seconds": t_cpu, "threads": 16},
                      "l2_rel_diff": abs(g_l2 - c_l2) / c_l2, "final_train_loss_hip": float(loss.detach())}))


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""The three arithmetic forms of the fp32 path over a LONG training run: the plain autoencoder trained for --steps steps
(batch 64, 6890 vertices, the same initial weights and the same batch order) in every form; held-out per-vertex L2
(test_funcs.py:41-49) along the way.  Trajectories are not bitwise comparable (Adam amplifies rounding-level differences
between ANY two summation orders); what must hold is that the forms reach the same error level.  One JSON line.
    python tools/form_trajectory.py [--steps 600] [--batch 64]"""
import argparse
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import semantichuman_amd as sh                                   # noqa: E402
from semantichuman_amd import _lib, synthetic                    # noqa: E402
from semantichuman_amd.hierarchy import load_hierarchy           # noqa: E402

FE = [[3, 16, 32, 64, 128], [[], [], [], [], []]]
FD = [[128, 64, 32, 32, 16], [[], [], [], [], 3]]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=600)
    ap.add_argument("--batch", type=int, default=64)
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    h = load_hierarchy(os.path.join(ROOT, "tests", "golden", "template6890.npz"))
    B = a.batch
    data = torch.from_numpy(synthetic.synth_batch(h.verts, 16 * B, seed=100)).to(dev)
    test = torch.from_numpy(synthetic.synth_batch(h.verts, 64, seed=7)).to(dev)
    ft = sh.FaceTables(h.faces, h.sizes[0] + 1, dev)
    torch.manual_seed(2)
    init = {k: v.detach().cpu().clone() for k, v in
            sh.SpiralAutoencoder(FE, FD, 256, h.sizes, h.spiral_sizes, h.spirals, h.D, h.U, dev).state_dict().items()}
    out = {"steps": a.steps, "batch": B, "vertices": int(h.sizes[0]), "forms": {}}
    was = _lib.get_f32_mma_mode()
    for form in ("exact", "split3", "planes3"):
        _lib.set_f32_mma_mode(form)
        m = sh.SpiralAutoencoder(FE, FD, 256, h.sizes, h.spiral_sizes, h.spirals, h.D, h.U, dev)
        m.load_state_dict(init)
        opt = sh.optim.Adam(m.parameters(), lr=1e-3, weight_decay=5e-5)
        curve = []
        for i in range(a.steps):
            x = data[(i * B) % data.shape[0]:(i * B) % data.shape[0] + B]
            opt.zero_grad(set_to_none=True)
            loss, _ = sh.recon_loss(m(x)[0], x, ft, 1e-2)
            loss.backward()
            opt.step()
            if (i + 1) % (a.steps // 6) == 0:
                with torch.no_grad():
                    curve.append([i + 1, float(sh.vertex_l2_mm(m(test)[0], test).item()), float(loss.item())])
        out["forms"][form] = {"l2_mm_curve": curve, "final_l2_mm": curve[-1][1], "final_train_loss": curve[-1][2]}
    _lib.set_f32_mma_mode(was)
    ex = out["forms"]["exact"]["final_l2_mm"]
    for form in ("split3", "planes3"):
        out["forms"][form]["final_l2_rel_to_exact"] = out["forms"][form]["final_l2_mm"] / ex - 1.0
    print(json.dumps(out))


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""Gradient error of one bf16 training step against the fp32 (exact) step from the same weights on the same batch: relative l2 error per
parameter.  Run with SH_BF16_RAGGED=1 / 0 to compare the ragged and the dense backward-data forms (tools/exp/r06_bf16_rag_err.sh)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
FE = [[3, 16, 32, 64, 128], [[], [], [], [], []]]
FD = [[128, 64, 32, 32, 16], [[], [], [], [], 3]]


def main():
    import semantichuman_amd as sh
    from semantichuman_amd import _lib, synthetic
    from semantichuman_amd.hierarchy import load_hierarchy
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
    dev = torch.device("cuda:0")
    h = load_hierarchy(os.path.join(ROOT, "tests", "golden", "template6890.npz"))
    x = torch.from_numpy(synthetic.synth_batch(h.verts, B, seed=100)).to(dev)
    ft = sh.FaceTables(h.faces, h.sizes[0] + 1, dev)
    grads = {}
    for dt in (torch.float32, torch.bfloat16):
        torch.manual_seed(2)
        _lib.set_f32_mma_mode("exact")
        m = sh.SpiralAutoencoder(FE, FD, 256, h.sizes, h.spiral_sizes, h.spirals, h.D, h.U, dev).set_compute_dtype(dt)
        loss, _ = sh.recon_loss(m(x)[0], x, ft, 1e-2)
        loss.backward()
        torch.cuda.synchronize()
        grads[dt] = {n: p.grad.detach().double().clone() for n, p in m.named_parameters()}
    tot = 0.0
    for n in grads[torch.float32]:
        a, b = grads[torch.float32][n], grads[torch.bfloat16][n]
        e = float((a - b).norm() / (a.norm() + 1e-30))
        tot += e
        print("%-28s %.4e" % (n, e))
    print("SUM %.5f  (SH_BF16_RAGGED=%s)" % (tot, os.environ.get("SH_BF16_RAGGED", "1")))


if __name__ == "__main__":
    main()

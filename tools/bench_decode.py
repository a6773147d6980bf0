#!/usr/bin/env python3
"""BASELINE config 5: decode 100 000 random latents at batch 1024 (98 batches, the last one of 672) on the 6890-vertex
template; reports the p50 batch latency, the per-mesh latency it implies and the overall meshes/s.  One JSON line.
    python tools/bench_decode.py [--latents 100000] [--batch 1024]"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import semantichuman_amd as sh                                   # noqa: E402
from semantichuman_amd.hierarchy import load_hierarchy           # noqa: E402

FE = [[3, 16, 32, 64, 128], [[], [], [], [], []]]
FD = [[128, 64, 32, 32, 16], [[], [], [], [], 3]]


def run(latents=100000, batch=1024, dev=None, template=None):
    """The measurement as a function (bench.py's `secondary` block calls it with fewer latents)."""
    dev = dev or torch.device("cuda:0")
    h = load_hierarchy(template or os.path.join(ROOT, "tests", "golden", "template6890.npz"))
    torch.manual_seed(2)
    m = sh.SpiralAutoencoder(FE, FD, 256, h.sizes, h.spiral_sizes, h.spirals, h.D, h.U, dev)
    z = torch.randn(latents, 256, generator=torch.Generator().manual_seed(0)).to(dev)
    with torch.no_grad():
        for _ in range(3):
            m.decode(z[:batch])
        torch.cuda.synchronize()
        lat = []
        t0 = time.perf_counter()
        for o in range(0, latents, batch):
            t1 = time.perf_counter()
            out = m.decode(z[o:o + batch])
            torch.cuda.synchronize()
            lat.append((time.perf_counter() - t1, out.shape[0]))
        total = time.perf_counter() - t0
    full = sorted(t for t, n in lat if n == batch)
    p50 = full[len(full) // 2]
    return {"metric": "decode of random latents, %d vertices" % h.sizes[0], "latents": latents, "batch": batch,
            "batches": len(lat), "p50_batch_ms": 1e3 * p50, "per_mesh_latency_us": 1e6 * p50 / batch,
            "meshes_per_s": latents / total, "dtype": "f32", "data": "synthetic"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--latents", type=int, default=100000)
    ap.add_argument("--batch", type=int, default=1024)
    a = ap.parse_args()
    print(json.dumps(run(a.latents, a.batch)))


if __name__ == "__main__":
    main()

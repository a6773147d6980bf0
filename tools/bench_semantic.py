#!/usr/bin/env python3
"""Time one iteration of the SEMANTIC training loop (reference train_funcs.py:73-472) on the 6890-vertex
template: SpiralAutoencoder_multiz_partkps, three passes (reconstruction, interpolation, exchange) with the part
pairwise-distance, key-point, part-volume and latent-norm losses, backward, Adam.  The 17 body parts are synthetic
(farthest-point Voronoi patches on the template, like the 578-vertex fixture), the joint regressor is a fixed positive
sparse-ish matrix; batch 16 per pass (traincfg.yaml:21-23).  One JSON line.
    python tools/bench_semantic.py [--batch 16] [--steps 20]"""
import argparse
import json
import os
import sys
import time
from types import SimpleNamespace

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import semantichuman_amd as sh                                   # noqa: E402
from semantichuman_amd import constants as C, synthetic           # noqa: E402
from semantichuman_amd import train_semantic as ts                # noqa: E402
from semantichuman_amd.hierarchy import load_hierarchy            # noqa: E402


def voronoi_parts(v, k):
    seeds = [0]
    dmin = np.linalg.norm(v - v[0], axis=1)
    for _ in range(k - 1):
        seeds.append(int(np.argmax(dmin)))
        dmin = np.minimum(dmin, np.linalg.norm(v - v[seeds[-1]], axis=1))
    owner = np.argmin(np.linalg.norm(v[:, None, :] - v[None, seeds, :], axis=2), axis=1)
    return [np.sort(np.nonzero(owner == j)[0]) for j in range(k)]


def kernel_report(step, n_prof=3):
    """Library kernels of `n_prof` eagerly launched iterations (the library's own per-launch HIP events): per kernel name
    launches / iteration and us / iteration, sorted by time; plus the raw per-launch records of one iteration."""
    from semantichuman_amd import _lib
    torch.cuda.synchronize()
    _lib.profile_enable(True)
    for _ in range(n_prof):
        step()
    torch.cuda.synchronize()
    recs = _lib.profile_records()
    _lib.profile_enable(False)
    agg = {}
    for name, ms in recs:
        k = name.split("|")[0]
        a = agg.setdefault(k, [0, 0.0])
        a[0] += 1
        a[1] += ms
    rows = [{"kernel": k, "launches_per_iteration": n / n_prof, "us_per_iteration": 1e3 * ms / n_prof} for k, (n, ms) in agg.items()]
    rows.sort(key=lambda r: -r["us_per_iteration"])
    per = len(recs) // n_prof
    return rows, recs[:per], sum(r["us_per_iteration"] for r in rows)


def run(batch=16, steps=50, graph=False, dev=None, warmup=10, torch_ops=False, report=False):
    """The measurement as a function (bench.py's `secondary` block calls it with graph=True)."""
    a = SimpleNamespace(batch=batch, steps=steps, graph=graph)
    dev = dev or torch.device("cuda:0")
    h = load_hierarchy(os.path.join(ROOT, "tests", "golden", "template6890.npz"))
    vi = h.verts / np.asarray((0.25, 0.15, 0.9))
    fine = dict(zip(C.PART_LIST, voronoi_parts(vi, 17)))
    # coarsest level: vertices of level 4 are a subset of level 0 (row selects): compose the selects
    idx = np.arange(h.sizes[0])
    for d in h.D:                                   # CSR row selects incl. the dummy row: col[:-1] = kept vertices
        idx = idx[np.asarray(d.col[:-1])]
    coarse = dict(zip(C.PART_LIST, voronoi_parts(vi[idx], 17)))
    J = np.abs(synthetic.closed_form_fill((35, h.sizes[0]), 1.0, 0.618, 0.3)) ** 8
    J = (J / J.sum(1, keepdims=True)).astype(np.float32)
    torch.manual_seed(2)
    m = sh.SpiralAutoencoder_multiz_partkps(C.KPS_INDEX_LIST, coarse, C.FILTER_SIZES_ENC, C.FILTER_SIZES_DEC, 8, 8, h.sizes,
                                            h.spiral_sizes, h.spirals, h.D, h.U, dev)
    opt = sh.optim.Adam(m.parameters(), lr=1e-3, weight_decay=5e-5)
    ctx = ts.SemanticContext(ts.SemanticTrainOptions(), SimpleNamespace(reference_mesh=SimpleNamespace(f=h.faces)), J, fine,
                             C.PART_LIST, dev)
    B = a.batch
    tx, txi, txe = (torch.from_numpy(synthetic.synth_batch(h.verts, B, seed=s)).to(dev) for s in (1, 2, 3))
    measure = torch.ones((B, 32), device=dev)

    def step():
        opt.zero_grad(set_to_none=True)
        total, _ = ts.semantic_losses(m, ctx, tx, txi, txe, epoch=1, measure=measure, draw_factor=1.1, exc_choice="ori")
        total.backward()
        opt.step()
        return total
    for _ in range(warmup):                          # also lets the caching allocator grow to its steady-state pool
        step()
    torch.cuda.synchronize()
    if torch_ops:                                    # which tensor-library launches are left in the iteration, and where from
        from torch.profiler import profile, ProfilerActivity
        with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
            for _ in range(3):
                step()
            torch.cuda.synchronize()
        rows = {}
        for e in prof.events():
            if not e.name.startswith("aten::") or not e.kernels:
                continue
            chain, q = [], e.cpu_parent                        # the enclosing operators / autograd nodes say where it comes from
            while q is not None and len(chain) < 4:
                chain.append(q.name.replace("autograd::engine::evaluate_function: ", "bwd "))
                q = q.cpu_parent
            key = (e.name, " < ".join(chain)[:150])
            r = rows.setdefault(key, [0, 0.0])
            r[0] += 1
            r[1] += sum(k.duration for k in e.kernels)
        for (name, frame), (n, us) in sorted(rows.items(), key=lambda kv: -kv[1][1]):
            print("%5.1f launches/it %7.1f us/it  %-28s %s" % (n / 3, us / 3, name, frame))
        return {}
    if a.graph:                                      # fixed edit factor / exchange kind: nothing host-side varies per replay
        sidestream = torch.cuda.Stream()
        sidestream.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(sidestream):
            step()
        torch.cuda.current_stream().wait_stream(sidestream)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            loss = step()
        runner = g.replay
    else:
        runner = step
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        out = runner()
        loss = out if out is not None else loss
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / a.steps
    res = {"metric": "semantic training iteration (3 passes), 6890 vertices", "batch_per_pass": B, "steps": a.steps,
           "launch": "hipGraph replay" if a.graph else "eager", "ms_per_iteration": 1e3 * dt, "meshes_per_s": 3 * B / dt,
           "loss": float(loss.detach()), "dtype": "f32", "data": "synthetic (Voronoi parts, synthetic joint regressor)",
           "parts": {n: int(len(p)) for n, p in list(fine.items())[:4]}}
    if report:
        rows, one, tot = kernel_report(step)
        res["library_kernel_us_per_iteration"] = tot
        res["library_launches_per_iteration"] = len(one)
        res["kernel_breakdown"] = rows[:10]
        res["_records"] = one
        res["_pair_loss"] = {"B": 3 * B, "part_sizes": [int(len(p)) for p in fine.values()]}
    return res


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=16)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--graph", action="store_true", help="capture the iteration into a hipGraph and replay it")
    ap.add_argument("--torch-ops", action="store_true", help="list the tensor-library launches left in the iteration, by call site")
    ap.add_argument("--layer-report", action="store_true", help="every library launch of one iteration (us), then the totals by kernel")
    a = ap.parse_args()
    res = run(a.batch, a.steps, a.graph, torch_ops=a.torch_ops, report=a.layer_report)
    if a.layer_report:
        print("%-52s %-52s %8s" % ("kernel", "shape", "us"))
        for name, ms in res.pop("_records"):
            k, _, shape = name.partition("|")
            print("%-52s %-52s %8.1f" % (k.replace("_kernel", ""), shape, 1e3 * ms))
        print("--- by kernel (launches / iteration, us / iteration)")
        for r in res["kernel_breakdown"]:
            print("%-52s %6.1f %9.1f" % (r["kernel"], r["launches_per_iteration"], r["us_per_iteration"]))
        print("library kernels: %.1f us / iteration in %d launches (torch's own element-wise launches are not in this list)" %
              (res["library_kernel_us_per_iteration"], res["library_launches_per_iteration"]))
        res.pop("_pair_loss", None)
    print(json.dumps(res))


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""'Matched per-vertex L2' on a TRAINED model: the metric BASELINE.json states the target at.

Trains the plain spiral autoencoder (6890 vertices, batch 64) for `--steps` steps with the reference's schedule - Adam(lr 1e-3,
weight_decay 5e-5), StepLR(step 1 epoch, gamma 0.99), L1 + 1e-2 edge-ratio loss, shuffled epochs of 16 batches (main.py:262-264,
train_funcs.py:495-510, traincfg.yaml) - in every arithmetic form of this library (exact fp32 MFMA, planes3 = the bench's
headline form, bf16 = BASELINE config 3's per-GPU shard) and, as the reference of the training DYNAMICS, in the oracle's
pure-torch model moved to the GPU with torch.optim.Adam (no kernel of this library).  Same initial weights, same batches in the
same order for every form of a seed; `--seeds` seeds (initialisation AND epoch permutations).  Every `--eval-every` steps the
held-out per-vertex L2 in mm (test_funcs.py:47-49) on 256 meshes none of the forms trains on.

    python tools/trained_l2.py [--steps 2000] [--seeds 3] [--forms exact,planes3,bf16,oracle] [--out profiles/r06_trained_l2.json]

Output: one JSON object - per form and seed the curve [(step, held-out L2 mm, training loss)], per form mean / min / max over the
seeds at every evaluation point, the TAIL figure (each seed's mean over the evaluation points of the last fifth of the run: one late evaluation
still moves by +-10 % from point to point) and `verdict`: for each library form whether its tail mean lies inside the spread
(min..max over seeds) of the oracle's, and its relative distance to the oracle's mean.  Test infrastructure: the only thing that imports
`oracle/` here is the reference leg."""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
FE = [[3, 16, 32, 64, 128], [[], [], [], [], []]]
FD = [[128, 64, 32, 32, 16], [[], [], [], [], 3]]
LR, WD, EDGE_W, GAMMA, B, EPOCH_BATCHES = 1e-3, 5e-5, 1e-2, 0.99, 64, 16


def epoch_orders(seed, n_epochs, n_data):
    g = torch.Generator().manual_seed(1000 + seed)
    return [torch.randperm(n_data, generator=g) for _ in range(n_epochs)]


def run_form(form, seed, sd0, h, data, test, steps, eval_every, dev):
    import semantichuman_amd as sh
    from semantichuman_amd import _lib
    n_data = data.shape[0]
    n_epochs = (steps + EPOCH_BATCHES - 1) // EPOCH_BATCHES
    orders = [o.to(dev) for o in epoch_orders(seed, n_epochs, n_data)]
    ft = sh.FaceTables(h.faces, h.sizes[0] + 1, dev)
    curve = []
    t0 = time.time()
    if form == "oracle":
        from oracle import ref_cpu
        S, D, U = h.dense_constants()
        m = ref_cpu.SpiralAEOracle(FE, FD, 256, h.sizes, h.spiral_sizes, S, D, U)
        m.load_state_dict({k: v.cpu() for k, v in sd0.items()})
        m.to(dev)
        m.spirals = [t.to(dev) for t in m.spirals]
        m.D = [t.to(dev) for t in m.D]
        m.U = [t.to(dev) for t in m.U]
        faces = torch.as_tensor(np.asarray(h.faces), dtype=torch.long, device=dev)
        opt = torch.optim.Adam(m.parameters(), lr=LR, weight_decay=WD)

        def loss_of(x):
            xh, _ = m(x)
            return torch.nn.functional.l1_loss(x, xh) + EDGE_W * ref_cpu.edge_ratio_loss(xh, x, faces)

        def l2_of():
            with torch.no_grad():
                tot = 0.0
                for i in range(0, test.shape[0], B):
                    xh, _ = m(test[i:i + B])
                    tot += float(ref_cpu.eval_metrics(xh, test[i:i + B])[1]) * min(B, test.shape[0] - i)
                return tot / test.shape[0]
    else:
        m = sh.SpiralAutoencoder(FE, FD, 256, h.sizes, h.spiral_sizes, h.spirals, h.D, h.U, dev)
        m.load_state_dict(sd0)
        if form == "bf16":
            m.set_compute_dtype(torch.bfloat16)
        else:
            _lib.set_f32_mma_mode(form)
        opt = sh.optim.Adam(m.parameters(), lr=LR, weight_decay=WD)

        def loss_of(x):
            return sh.recon_loss(m(x)[0], x, ft, EDGE_W)[0]

        def l2_of():
            with torch.no_grad():
                tot = 0.0
                for i in range(0, test.shape[0], B):
                    tot += float(sh.vertex_l2_mm(m(test[i:i + B])[0], test[i:i + B])) * min(B, test.shape[0] - i)
                return tot / test.shape[0]
    sched = torch.optim.lr_scheduler.StepLR(opt, 1, gamma=GAMMA)
    curve.append((0, l2_of(), None))
    step = 0
    for ep in range(n_epochs):
        order = orders[ep]
        for bi in range(EPOCH_BATCHES):
            if step >= steps:
                break
            x = data[order[bi * B:(bi + 1) * B]]
            opt.zero_grad(set_to_none=True)
            loss = loss_of(x)
            loss.backward()
            opt.step()
            step += 1
            if step % eval_every == 0 or step == steps:
                curve.append((step, l2_of(), float(loss)))
        sched.step()
    torch.cuda.synchronize()
    print("  %-8s seed %d: %s  (%.0f s)" % (form, seed, " ".join("%d:%.2f" % (s, l2) for s, l2, _ in curve), time.time() - t0), flush=True)
    return curve


TAIL_FRACTION = 0.8      # the "tail" figure averages the evaluation points at or after this fraction of the run


def summarise(curves, forms, steps, seeds, eval_every, build):
    """Per form: mean / min / max over the seeds at every evaluation point, and the TAIL figure - each seed's mean over its last
    evaluation points of the last fifth of the run (a single evaluation late in training still moves by +-10 % from one point to the next: the model is
    evaluated between two Adam steps at lr ~3e-4; the snapshot at the last step is reported too, the verdict uses the tail)."""
    summary, tail = {}, {}
    for f in forms:
        pts = []
        for k in range(len(curves[f][0])):
            v = [c[k][1] for c in curves[f]]
            pts.append({"step": curves[f][0][k][0], "mean_mm": float(np.mean(v)), "min_mm": float(np.min(v)), "max_mm": float(np.max(v))})
        summary[f] = pts
        sel = [k for k, q in enumerate(curves[f][0]) if q[0] >= TAIL_FRACTION * steps]
        t = [float(np.mean([c[k][1] for k in sel])) for c in curves[f]]
        tail[f] = {"steps": [curves[f][0][k][0] for k in sel], "per_seed_mm": t, "mean_mm": float(np.mean(t)), "min_mm": float(np.min(t)),
                   "max_mm": float(np.max(t))}
    verdict = {}
    if "oracle" in forms:
        o = tail["oracle"]
        for f in forms:
            if f == "oracle":
                continue
            e = tail[f]
            verdict[f] = {"tail_mean_mm": e["mean_mm"], "oracle_tail_mean_mm": o["mean_mm"], "oracle_tail_spread_mm": [o["min_mm"], o["max_mm"]],
                          "rel_diff_of_means": (e["mean_mm"] - o["mean_mm"]) / o["mean_mm"],
                          "mean_inside_oracle_spread": bool(o["min_mm"] <= e["mean_mm"] <= o["max_mm"]),
                          "last_step_mean_mm": summary[f][-1]["mean_mm"], "oracle_last_step_mean_mm": summary["oracle"][-1]["mean_mm"]}
    return {"what": "held-out per-vertex L2 (mm, test_funcs.py:47-49) of the plain autoencoder trained with the reference's schedule "
                    "(Adam 1e-3 / 5e-5, StepLR gamma 0.99 per 16-batch epoch, L1 + 1e-2 edge loss), 6890 vertices, batch 64, synthetic data",
            "steps": steps, "seeds": seeds, "eval_every": eval_every, "build": build,
            "forms": {"exact": "fp32 MFMA kernels", "planes3": "three-plane form of the fp32 products (the bench headline's form)",
                      "bf16": "bf16 compute path (BASELINE config 3's per-GPU shard)",
                      "oracle": "oracle/ref_cpu.py's pure-torch model on the GPU, torch.optim.Adam (no library kernel)"},
            "tail": tail, "summary": summary, "verdict": verdict,
            "curves": {f: [[[s_, l2, tl] for s_, l2, tl in c] for c in curves[f]] for f in forms}}


def main():
    if len(sys.argv) > 2 and sys.argv[1] == "--resummarise":      # recompute the statistics of an existing result file (no GPU)
        d = json.load(open(sys.argv[2]))
        curves = {f: [[tuple(q) for q in c] for c in cs] for f, cs in d["curves"].items()}
        out = summarise(curves, list(curves), d["steps"], d["seeds"], d["eval_every"], d.get("build"))
        json.dump(out, open(sys.argv[2], "w"), indent=1)
        print(json.dumps({"tail": {f: out["tail"][f] for f in curves}, "verdict": out["verdict"]}))
        return
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--seeds", type=int, default=3)
    ap.add_argument("--eval-every", type=int, default=200)
    ap.add_argument("--forms", default="exact,planes3,bf16,oracle")
    ap.add_argument("--out", default=os.path.join(ROOT, "profiles", "r06_trained_l2.json"))
    args = ap.parse_args()
    import semantichuman_amd as sh
    from semantichuman_amd import _lib, synthetic
    from semantichuman_amd.hierarchy import load_hierarchy
    dev = torch.device("cuda:0")
    h = load_hierarchy(os.path.join(ROOT, "tests", "golden", "template6890.npz"))
    data = torch.from_numpy(synthetic.synth_batch(h.verts, EPOCH_BATCHES * B, seed=100)).to(dev)
    test = torch.from_numpy(synthetic.synth_batch(h.verts, 256, seed=7)).to(dev)
    forms = [f for f in args.forms.split(",") if f]
    was = _lib.get_f32_mma_mode()
    curves = {f: [] for f in forms}
    try:
        for seed in range(args.seeds):
            torch.manual_seed(10 + seed)
            m0 = sh.SpiralAutoencoder(FE, FD, 256, h.sizes, h.spiral_sizes, h.spirals, h.D, h.U, dev)
            sd0 = {k: v.detach().clone() for k, v in m0.state_dict().items()}
            del m0
            for f in forms:
                curves[f].append(run_form(f, seed, sd0, h, data, test, args.steps, args.eval_every, dev))
                torch.cuda.empty_cache()
    finally:
        _lib.set_f32_mma_mode(was)
    out = summarise(curves, forms, args.steps, args.seeds, args.eval_every, _lib.build_id())
    os.makedirs(os.path.dirname(args.out), exist_ok=True)
    with open(args.out, "w") as fh:
        json.dump(out, fh, indent=1)
    print(json.dumps({"tail": {f: out["tail"][f] for f in forms}, "verdict": out["verdict"]}))


if __name__ == "__main__":
    main()

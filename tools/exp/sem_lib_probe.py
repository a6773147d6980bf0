#!/usr/bin/env python3
"""Library kernel durations of one eager semantic iteration (profiler tags)."""
import os, sys, collections
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import bench_semantic as bs
import semantichuman_amd as sh
from semantichuman_amd import _lib, train_semantic as ts
orig = ts.semantic_losses
state = {"n": 0}
def wrapped(*a, **k):
    state["n"] += 1
    if state["n"] == 5:
        torch.cuda.synchronize(); _lib.profile_enable(True)
    return orig(*a, **k)
ts.semantic_losses = wrapped
orig_step = sh.optim.Adam.step
def step_wrapped(self, *a, **k):
    r = orig_step(self, *a, **k)
    if state["n"] == 5 and "done" not in state:
        torch.cuda.synchronize()
        state["recs"] = _lib.profile_records_by_kernel(); _lib.profile_enable(False); state["done"] = 1
    return r
sh.optim.Adam.step = step_wrapped
bs.run(batch=16, steps=1, graph=False, warmup=6)
tot = 0
for k, tag, ms in state["recs"]:
    print("%-44s %-70s %7.1f" % (k[:44], tag[:70], ms * 1e3)); tot += ms
print("total library kernels (profiled ones): %.1f us" % (tot * 1e3))

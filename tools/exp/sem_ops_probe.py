#!/usr/bin/env python3
"""Which torch ops (kernel launches) one semantic iteration issues, by source line."""
import os, sys, collections
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import bench_semantic as bs
from torch.profiler import profile, ProfilerActivity

# reuse bench_semantic's setup by running a few eager steps under the profiler
import types
orig_run = bs.run
src = open(os.path.join(ROOT, "tools", "bench_semantic.py")).read()
# build the step function the same way: exec run() body up to the warm-up with steps=0 is awkward; instead call run with a hook
captured = {}
real_sync = torch.cuda.synchronize
def run_probe():
    # monkeypatch: make run() do warmup=3, steps=1 eager, profile the last step
    prof = profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True)
    state = {"n": 0}
    orig_backward = torch.Tensor.backward
    return prof
import semantichuman_amd as sh
from semantichuman_amd import train_semantic as ts
calls = collections.Counter()
orig = ts.semantic_losses
prof_holder = {}
def wrapped(*a, **k):
    wrapped.n += 1
    if wrapped.n == 5:
        prof = profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True)
        prof.__enter__()
        prof_holder["p"] = prof
    return orig(*a, **k)
wrapped.n = 0
ts.semantic_losses = wrapped
orig_step = sh.optim.Adam.step
def step_wrapped(self, *a, **k):
    r = orig_step(self, *a, **k)
    if wrapped.n == 5 and "p" in prof_holder and "done" not in prof_holder:
        torch.cuda.synchronize()
        prof_holder["p"].__exit__(None, None, None)
        prof_holder["done"] = True
    return r
sh.optim.Adam.step = step_wrapped
bs.run(batch=16, steps=1, graph=False, warmup=6)
p = prof_holder["p"]
rows = []
for e in p.key_averages(group_by_stack_n=6):
    if e.device_time_total > 0 or getattr(e, "cuda_time_total", 0) > 0:
        st = [s for s in e.stack if "semantichuman_amd" in s or "tools/" in s][:2]
        rows.append((e.count, e.key, " <- ".join(s.split("semantichuman_amd/")[-1] for s in st)))
agg = collections.Counter()
for c, k, st in rows:
    agg[(k, st)] += c
tot = 0
for (k, st), c in sorted(agg.items(), key=lambda kv: -kv[1])[:70]:
    print("%4d  %-38s %s" % (c, k[:38], st[:150]))
    tot += c
print("total op instances with device time:", sum(agg.values()))

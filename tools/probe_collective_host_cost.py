#!/usr/bin/env python3
"""Host-side cost of issuing one RCCL all-reduce through torch.distributed (world of one rank, 1-GPU box):
what each collective adds to an eagerly launched training step.  python tools/probe_collective_host_cost.py"""
import os
import time

import torch
import torch.distributed as dist

os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29544")
os.environ.setdefault("RANK", "0")
os.environ.setdefault("WORLD_SIZE", "1")
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
dist.init_process_group("nccl", device_id=dev)
small = torch.ones(1024, device=dev)
big = torch.ones(14 * 1024 * 1024, device=dev)


def timeit(fn, n=300):
    for _ in range(20):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    t1 = time.perf_counter()            # host time to ISSUE n calls (queue not drained)
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    return 1e6 * (t1 - t0) / n, 1e6 * (t2 - t0) / n


def ar_async(t):
    def f():
        w = dist.all_reduce(t, op=dist.ReduceOp.AVG, async_op=True)
        w.wait()
    return f


def ar_sync(t):
    return lambda: dist.all_reduce(t, op=dist.ReduceOp.AVG)


for name, fn in (("small async+wait", ar_async(small)), ("small sync", ar_sync(small)), ("56MB async+wait", ar_async(big)),
                 ("56MB sync", ar_sync(big)), ("torch add_ (reference launch)", lambda: small.add_(1.0))):
    issue, total = timeit(fn)
    print("%-32s issue %7.1f us/call   drained %7.1f us/call" % (name, issue, total))
dist.destroy_process_group()

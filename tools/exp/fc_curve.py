#!/usr/bin/env python3
"""Time of the latent FC passes against the long dimension (fixed cost vs per-byte cost).  GPU box."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from semantichuman_amd import _lib, ops

d = torch.device("cuda:0")


def timed(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    _lib.profile_enable(True)
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    out = {}
    for n, tag, ms in _lib.profile_records_by_kernel():
        out[n] = out.get(n, 0.0) + ms * 1e3 / reps
    _lib.profile_enable(False)
    return out


flush = torch.empty(512 << 20, dtype=torch.uint8, device=d)
for L in (55296,):
    M = 64
    x = torch.randn(M, L, device=d); W = torch.randn(256, L, device=d) * 0.01; b = torch.randn(256, device=d)
    z = torch.randn(M, 256, device=d); Wd = torch.randn(L, 256, device=d) * 0.01; bd = torch.randn(L, device=d)
    dy = torch.randn(M, L, device=d)
    def cold(fn):
        def g():
            flush.zero_()
            fn()
        return g
    r = {}
    r["P1 fwd N=256 K=L"] = timed(cold(lambda: ops.linear_fwd(x, W, b)))
    r["P2 fwd N=L K=256"] = timed(cold(lambda: ops.linear_fwd(z, Wd, bd)))
    r["P3 bwd_data N=L K=256"] = timed(cold(lambda: ops.linear_bwd_data(dy, Wd)))
    r["P4 bwd_data N=256 K=L"] = timed(cold(lambda: ops.linear_bwd_data(z, W)))
    r["P5 bwd_wgt N=L K=256"] = timed(cold(lambda: ops.linear_bwd_wgt(dy, z)))
    r["P6 bwd_wgt N=256 K=L"] = timed(cold(lambda: ops.linear_bwd_wgt(z, x)))
    for k, v in r.items():
        print("L=%6d %-24s %s" % (L, k, "  ".join("%s %.1f" % (n.replace("linear_", "").replace("_kernel", "")[:28], t) for n, t in v.items())))

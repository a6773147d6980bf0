"""Times the latent-FC weight-gradient kernel with Adam inside (sh_linear_bwd_wgt_adam) and the two-kernel form it replaces
(sh_linear_bwd_wgt + sh_adam_step) on the benchmark's shapes: HIP events around 20 launches after 3 warm-ups.
Usage: python tools/fc_adam_probe.py [exact|planes3]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import semantichuman_amd as sh                     # noqa: E402
from semantichuman_amd import _lib, ops            # noqa: E402


def timed(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


def main():
    form = sys.argv[1] if len(sys.argv) > 1 else "planes3"
    dev = torch.device("cuda:0")
    print("lib build", _lib.build_id(), "form", form)
    for M, N, K in [(64, 55296, 256), (64, 256, 55296), (32, 220672, 256), (32, 256, 220672)]:
        g = torch.Generator().manual_seed(1)
        dy = (torch.randn(M, N, generator=g) * 1e-2).to(dev)
        x = torch.randn(M, K, generator=g).to(dev)
        w = (torch.randn(N, K, generator=g) * 0.05).to(dev).requires_grad_(True)
        opt = sh.optim.Adam([w], lr=1e-3, weight_decay=5e-5)
        st = opt._state_of(w)
        lr = torch.full((), 1e-3, device=dev)

        def fused():
            ops.linear_bwd_wgt_adam(dy, x, w.data, st["exp_avg"], st["exp_avg_sq"], st["step"], lr, (0.9, 0.999), 1e-8, 5e-5, want_bias=True, mma=form)

        def two():
            w.grad, _ = ops.linear_bwd_wgt(dy, x, want_bias=True, mma=form)
            opt.step()
        tf, tt = timed(fused), timed(two)
        byt = 4.0 * (6 * N * K + M * K + M * N)
        print("M=%d N=%d K=%d  fused %.1f us (%.2f TB/s of its 24 B/weight)   two kernels %.1f us (%.2f TB/s of their 32 B/weight)"
              % (M, N, K, tf, byt / tf / 1e6, tt, (byt + 8.0 * N * K) / tt / 1e6))


if __name__ == "__main__":
    main()

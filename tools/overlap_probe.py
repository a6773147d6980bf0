"""Probe: does a re-sampling launch (HBM-bound) hide under a streaming weight-gradient launch (matrix-bound) when both are
resident at once?  Two streams, HIP events: t(wgrad alone), t(spmm alone), t(both started together).  If t(both) ~ max, hosting the
backward pass's U^T launches as tail workgroups of the weight gradients (as the pre-sum jobs already are) would pay; if ~ sum, not.
Usage: python tools/overlap_probe.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import semantichuman_amd as sh                                   # noqa: E402
from semantichuman_amd import _lib, ops                          # noqa: E402
from semantichuman_amd.hierarchy import load_hierarchy           # noqa: E402

FE = [[3, 16, 32, 64, 128], [[], [], [], [], []]]
FD = [[128, 64, 32, 32, 16], [[], [], [], [], 3]]


def main():
    dev = torch.device("cuda:0")
    B = 64
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    h = load_hierarchy(os.path.join(root, "tests", "golden", "template6890.npz"))
    model = sh.SpiralAutoencoder(FE, FD, 256, h.sizes, h.spiral_sizes, h.spirals, h.D, h.U, dev)
    _lib.set_f32_mma_mode("planes3")
    steps = model._dec_stack.steps
    convs = [s for s in steps if s.kind == "conv"]
    spmms = [s for s in steps if s.kind == "spmm"]
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    print("lib build", _lib.build_id())
    for cv in convs:
        if cv.cout < 16 or cv.cin < 16:
            continue
        dpre = ops.alloc(B, cv.R, cv.cout, "vm", dev, extra_rows=cv.n_extra).normal_()
        x = ops.alloc(B, cv.n_in, cv.cin, "vm", dev).normal_()
        lib = _lib.load()
        nbytes = lib.sh_spiral_conv_bwd_wgt_workspace(B, cv.R, cv.S, cv.cin, cv.cout)
        ws = torch.empty((nbytes + 3) // 4, dtype=torch.float32, device=dev)
        _, _, _, dsv, dsb = ops._dims(dpre, "vm")
        _, _, _, xsv, xsb = ops._dims(x, "vm")

        def wg():
            _lib.check(lib.sh_spiral_conv_bwd_wgt(_lib.ptr(dpre), dsv, dsb, _lib.ptr(x), xsv, xsb, _lib.ptr(cv.dev["table"]), None, None, _lib.ptr(ws),
                                                  nbytes, B, cv.R, cv.S, cv.cin, cv.cout, _lib.mma_id(), _lib.stream_ptr()), "wgrad")
        for sp in spmms:
            rows_t = sp.csr_t.rows
            C = cv.cin
            src = ops.alloc(B, sp.csr_t.cols, C, "vm", dev).normal_()
            dst = ops.alloc(B, rows_t, C, "vm", dev)
            if src.shape[0 if False else 0] == 0:
                continue

            def spm():
                ops.spmm(sp.dev["mt"], src, "vm", dst, "vm", rows_t)

            def timed(fa, fb, reps=8, n=10):
                # GPU-side time: `reps` x [fork, fa on one stream || fb on another, join] captured into one hipGraph, replayed n times
                def body():
                    cur = torch.cuda.current_stream()
                    for _ in range(reps):
                        s1.wait_stream(cur); s2.wait_stream(cur)
                        if fa:
                            with torch.cuda.stream(s1):
                                fa()
                        if fb:
                            with torch.cuda.stream(s2):
                                fb()
                        cur.wait_stream(s1); cur.wait_stream(s2)
                side = torch.cuda.Stream()
                side.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(side):
                    body()
                torch.cuda.current_stream().wait_stream(side)
                torch.cuda.synchronize()
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g):
                    body()
                g.replay()
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(n):
                    g.replay()
                e1.record()
                torch.cuda.synchronize()
                return e0.elapsed_time(e1) / (n * reps) * 1e3
            tw, ts, tb = timed(wg, None), timed(None, spm), timed(wg, spm)
            print("wgrad R=%d K=%d N=%d  |  spmm^T rows=%d C=%d :  alone %.1f + %.1f = %.1f us   together %.1f us   (max %.1f)"
                  % (cv.R, cv.S * cv.cin, cv.cout, rows_t, C, tw, ts, tw + ts, tb, max(tw, ts)))


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""Does the weight gradient of a conv layer (exact fp32 MFMA, wgrad_stream) overlap with the backward-data of the same layer
(three-plane form, conv_p3) when the two run on different streams?  Both only need dpre of the layer.  Per layer: wall time of
20 x (W; D) on one stream against 20 x W on one stream beside 20 x D on another.  GPU box."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import p3_probe                                                      # noqa: E402
import semantichuman_amd as sh                                       # noqa: E402
from semantichuman_amd import _lib, ops                              # noqa: E402
from semantichuman_amd.hierarchy import load_hierarchy               # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
dev = torch.device("cuda:0")
lib = _lib.load()
h = load_hierarchy(os.path.join(ROOT, "tests", "golden", "template6890.npz"))
torch.manual_seed(0)
model = sh.SpiralAutoencoder(p3_probe.FE, p3_probe.FD, 256, h.sizes, h.spiral_sizes, h.spirals, h.D, h.U, dev)
_lib.set_f32_mma_mode("planes3")
REPS = 20
tot = {"serial": 0.0, "overlap": 0.0, "w": 0.0, "d": 0.0}
for sname, stack in (("enc", model._enc_stack), ("dec", model._dec_stack)):
    for i, st in enumerate(stack.steps):
        if st.kind != "conv" or st.cin == 3 or st.cout == 3:
            continue
        R, S, cin, cout, n_in = st.R, st.S, st.cin, st.cout, st.n_in
        if not lib.sh_spiral_conv_p3_ok(B, S, cout, cin):
            continue
        table, table_t = st.dev["table"], st.dev["table_t"]
        w = (torch.randn((cout, S * cin), device=dev) / (S * cin) ** 0.5).contiguous()
        x_in = torch.randn((n_in, B, cin), device=dev)
        dpre = torch.randn((R + st.n_extra, B, cout), device=dev)
        dpre[st.zero_row] = 0
        dimg = p3_probe.to_p3(dpre)
        wf = p3_probe.wfrag3(w, S, cin, cout, True)
        dx = torch.empty((n_in, B, cin), device=dev)
        dximg = torch.empty(lib.sh_p3_bytes(n_in, B, cin), dtype=torch.uint8, device=dev)
        nbytes = lib.sh_spiral_conv_bwd_wgt_workspace(B, R, S, cin, cout)
        ws = torch.empty((nbytes + 3) // 4, dtype=torch.float32, device=dev)
        kind = lib.sh_spiral_conv_p3_kind(B, S, cout, cin)

        def W():
            _lib.check(lib.sh_spiral_conv_bwd_wgt(_lib.ptr(dpre), B * cout, cout, _lib.ptr(x_in), B * cin, cin, _lib.ptr(table), None, None,
                                                  _lib.ptr(ws), nbytes, B, R, S, cin, cout, _lib.mma_id(), _lib.stream_ptr()), "wgt")

        def D():
            _lib.check(lib.sh_spiral_conv_bwd_data_p3(_lib.ptr(dimg), st.zero_row, _lib.ptr(dpre) if kind == 1 else None, B * cout, cout, R,
                                                      _lib.ptr(table_t), _lib.ptr(wf), _lib.ptr(dx), B * cin, cin, _lib.ptr(dximg), None, 0, 0,
                                                      0, -1, B, n_in, S, cin, cout, _lib.stream_ptr()), "bwd_data_p3")

        def wall(fn):
            fn()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            fn()
            e1.record()
            torch.cuda.synchronize()
            return e0.elapsed_time(e1) * 1e3 / REPS

        s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()

        def serial():
            for _ in range(REPS):
                W(); D()

        def only(f):
            def g():
                for _ in range(REPS):
                    f()
            return g

        def overlap():
            main = torch.cuda.current_stream()
            s1.wait_stream(main); s2.wait_stream(main)
            with torch.cuda.stream(s1):
                for _ in range(REPS):
                    W()
            with torch.cuda.stream(s2):
                for _ in range(REPS):
                    D()
            main.wait_stream(s1); main.wait_stream(s2)

        def pairwise():                                     # what a sequencer would do: fork / join around every pair
            main = torch.cuda.current_stream()
            for _ in range(REPS):
                s1.wait_stream(main)
                with torch.cuda.stream(s1):
                    W()
                D()
                main.wait_stream(s1)

        tw, td, ts, to, tp = wall(only(W)), wall(only(D)), wall(serial), wall(overlap), wall(pairwise)
        tot["serial"] += ts; tot["overlap"] += to; tot["w"] += tw; tot["d"] += td
        tot["pair"] = tot.get("pair", 0.0) + tp
        print("%s%d %3d->%3d R=%5d S=%2d | W %6.1f  D %6.1f  serial %6.1f  two streams %6.1f  fork/join per pair %6.1f us" % (sname, i, cin, cout, R, S, tw, td, ts, to, tp), flush=True)
print("sum: W %.1f  D %.1f  serial %.1f  two streams %.1f  fork/join per pair %.1f us" % (tot["w"], tot["d"], tot["serial"], tot["overlap"], tot["pair"]))

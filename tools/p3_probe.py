#!/usr/bin/env python3
"""Per-layer probe of the three-plane form (csrc/p3_conv.hip) on the conv shapes of a template: error against a float64
evaluation and kernel time, next to the exact fp32 MFMA form and the bf16x3 (split3) form.

    python tools/p3_probe.py [batch] [template.npz] [--bwd] [--adversarial] [--reps N]

Prints one line per conv step and direction:  shape | max|err| exact / split3 / p3 (relative to max|ref|) | us each.
Test infrastructure (float64 on the GPU through torch); nothing here is on the product path."""
from __future__ import annotations

import ctypes
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import semantichuman_amd as sh                                    # noqa: E402
from semantichuman_amd import _lib, ops                            # noqa: E402
from semantichuman_amd.hierarchy import load_hierarchy            # noqa: E402

FE = [[3, 16, 32, 64, 128], [[], [], [], [], []]]
FD = [[128, 64, 32, 32, 16], [[], [], [], [], 3]]


def arr(vals, ct):
    return (ct * len(vals))(*vals)


def wfrag3(w, S, cin, cout, tr):
    lib = _lib.load()
    nb = lib.sh_conv_wfrag3_bytes(S, cout if tr else cin, cin if tr else cout)
    buf = torch.empty(nb, dtype=torch.uint8, device=w.device)
    _lib.check(lib.sh_conv_wfrag3_prep_multi(1, arr([w.data_ptr()], ctypes.c_void_p), arr([buf.data_ptr()], ctypes.c_void_p),
                                             arr([S], ctypes.c_int), arr([cin], ctypes.c_int), arr([cout], ctypes.c_int),
                                             arr([1 if tr else 0], ctypes.c_int), _lib.stream_ptr()), "wfrag3")
    return buf


def to_p3(x):
    """x: [rows][B][C] vertex-major fp32 -> plane image (uint8 tensor)."""
    lib = _lib.load()
    rows, B, C = x.shape
    nb = lib.sh_p3_bytes(rows, B, C)
    assert nb > 0, (rows, B, C)
    buf = torch.empty(nb, dtype=torch.uint8, device=x.device)
    _lib.check(lib.sh_to_p3(_lib.ptr(x), B * C, C, _lib.ptr(buf), B, rows, C, _lib.stream_ptr()), "sh_to_p3")
    return buf


def timed(fn, reps):
    """Average kernel time per call in us, from the library's own per-launch HIP events (what rocprofv3 reports): host
    call overhead does not enter."""
    fn()
    torch.cuda.synchronize()
    _lib.profile_enable(True)
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    recs = _lib.profile_records()
    _lib.profile_enable(False)
    return 1e3 * sum(ms for _, ms in recs) / reps


def rnd(shape, dev, adversarial, gen):
    x = torch.randn(shape, device=dev, generator=gen)
    if adversarial:
        # six decades of dynamic range, and sums that cancel: magnitudes 10^U(-3,3), signs alternating along the last axis
        mag = torch.pow(10.0, 6.0 * torch.rand(shape, device=dev, generator=gen) - 3.0)
        sgn = torch.where(torch.arange(shape[-1], device=dev) % 2 == 0, 1.0, -1.0)
        x = mag * sgn * (1.0 + 1e-3 * x)
    return x.float().contiguous()


def probe_layers(B=64, tpl=None, directions=("fwd", "bwd"), adversarial=False, reps=1, skip=True, seed=1, f32_presum_rows=True):
    """Yields one record per conv step (3-channel sides excluded) and direction of the template's plain autoencoder:
    {"name", "bwd", "ok" (the three-plane kernels take the shape), "err": {form: max|err| / max|ref| against a float64
    evaluation on the device}, "us": {form: kernel time}, "us_to_p3", "img_ok" (the image the kernel wrote of its output is
    the image of that output, bit for bit)}."""
    tpl = tpl or os.path.join(ROOT, "tests", "golden", "template6890.npz")
    dev = torch.device("cuda:0")
    lib = _lib.load()
    h = load_hierarchy(tpl)
    torch.manual_seed(0)
    model = sh.SpiralAutoencoder(FE, FD, 256, h.sizes, h.spiral_sizes, h.spirals, h.D, h.U, dev)
    gen = torch.Generator(device=dev)
    gen.manual_seed(seed)
    was = _lib.get_f32_mma_mode()
    F32ROWS = f32_presum_rows      # backward-data: the pre-summed rows (>= R) are read from the fp32 tensor, as the stack sequencer runs it
    try:
        for sname, stack in (("enc", model._enc_stack), ("dec", model._dec_stack)):
            for i, st in enumerate(stack.steps):
                if st.kind != "conv" or st.cin == 3 or st.cout == 3:
                    continue
                R, S, cin, cout, n_in = st.R, st.S, st.cin, st.cout, st.n_in
                zrow = st.zero_row if skip else -1
                table, table_t = st.dev["table"], st.dev["table_t"]
                w = (torch.randn((cout, S * cin), device=dev, generator=gen) / (S * cin) ** 0.5).contiguous()
                if adversarial:
                    w = rnd((cout, S * cin), dev, True, gen)
                bias = torch.randn((cout,), device=dev, generator=gen).contiguous()
                for d in directions:
                    bwd = d == "bwd"
                    if not bwd:
                        Cg, Nout, rows_out, tbl = cin, cout, R, table
                        x = rnd((n_in, B, cin), dev, adversarial, gen)
                    else:
                        Cg, Nout, rows_out, tbl = cout, cin, n_in, table_t
                        x = rnd((R + st.n_extra, B, cout), dev, adversarial, gen)
                        x[st.zero_row] = 0             # as in the stack: the dummy row of dpre is zero, "no source" entries point at it
                    ok = bool(lib.sh_spiral_conv_p3_ok(B, S, Cg, Nout))
                    x64, w64 = x.double(), w.double().view(cout, S, cin)
                    ref = torch.zeros((rows_out, B, Nout), dtype=torch.float64, device=dev)
                    for s in range(S):
                        ref += x64[tbl[:, s].long()] @ (w64[:, s, :].t() if not bwd else w64[:, s, :])
                    if not bwd:
                        ref += bias.double()
                    scale = float(ref.abs().max())
                    y = torch.empty((rows_out, B, Nout), dtype=torch.float32, device=dev)
                    wt = ops.weight_transpose(w, S, cin, cout) if bwd else None

                    def run_f32():
                        if not bwd:
                            ops.spiral_conv_fwd(x, "vm", table, w, bias, y, "vm", R, S, 0, -1)
                        else:
                            ops.spiral_conv_bwd_data(x, "vm", table_t, wt, y, "vm", None, "vm", 0, -1, n_in, S, cin, cout)
                    rec = {"name": "%s%d %s R=%d K=%d N=%d" % (sname, i, d, rows_out, S * Cg, Nout), "bwd": bwd, "ok": ok,
                           "err": {}, "us": {}, "scale": scale}
                    for mode in ("exact", "split3"):
                        _lib.set_f32_mma_mode(mode)
                        rec["us"][mode] = timed(run_f32, reps)
                        rec["err"][mode] = float((y.double() - ref).abs().max()) / scale
                    _lib.set_f32_mma_mode("exact")
                    if ok:
                        wf = wfrag3(w, S, cin, cout, bwd)
                        xp = to_p3(x)
                        nimg = lib.sh_p3_bytes(rows_out, B, Nout)
                        yp = torch.empty(nimg, dtype=torch.uint8, device=dev) if nimg else None
                        y.zero_()

                        def run_p3():
                            if not bwd:
                                _lib.check(lib.sh_spiral_conv_fwd_p3(_lib.ptr(xp), _lib.ptr(table), _lib.ptr(wf), _lib.ptr(bias), _lib.ptr(y),
                                                                     B * Nout, Nout, _lib.ptr(yp), B, R, S, cin, cout, 0, -1, _lib.stream_ptr()),
                                           "sh_spiral_conv_fwd_p3")
                            else:
                                _lib.check(lib.sh_spiral_conv_bwd_data_p3(_lib.ptr(xp), zrow, _lib.ptr(x) if (F32ROWS and lib.sh_spiral_conv_p3_kind(B, S, Cg, Nout) == 1) else None, B * cout, cout, R, _lib.ptr(table_t), _lib.ptr(wf), _lib.ptr(y), B * Nout,
                                                                          Nout, _lib.ptr(yp), None, 0, 0, None, 0, -1, B, n_in, S, cin, cout,
                                                                          _lib.stream_ptr()), "sh_spiral_conv_bwd_data_p3")
                        rec["us"]["planes3"] = timed(run_p3, reps)
                        rec["err"]["planes3"] = float((y.double() - ref).abs().max()) / scale
                        # round 6: the same backward-data pass over ragged source lists (no dense table, no pre-summed rows)
                        ref_r = None
                        if bwd and getattr(st, "rag", None) is not None:
                            rr, rp = st.dev["rag_rows"], st.dev["rag_pos"]
                            # reference of the list forms = the dense one when the extra rows hold the pre-sums: build it from the lists
                            ref_r = torch.zeros((rows_out, B, Nout), dtype=torch.float64, device=dev)
                            for j in range(rr.shape[1]):
                                ok_j = (rp[:, j] >= 0)
                                for s_ in range(S):
                                    sel = ok_j & (rp[:, j] == s_)
                                    if bool(sel.any()):
                                        ref_r[sel] += x64[rr[sel, j].long()] @ w64[:, s_, :]
                        if ref_r is not None and lib.sh_spiral_conv_p3_rag_ok(B, S, Cg, Nout, int(st.rag[0].shape[1])):
                            y.zero_()

                            def run_rag():
                                _lib.check(lib.sh_spiral_conv_bwd_data_p3_rag(_lib.ptr(xp), _lib.ptr(rr), _lib.ptr(rp), int(rr.shape[1]), _lib.ptr(wf), _lib.ptr(y),
                                                                              B * Nout, Nout, _lib.ptr(yp), None, 0, 0, None, 0, -1, B, n_in, S, cin, cout,
                                                                              _lib.stream_ptr()), "sh_spiral_conv_bwd_data_p3_rag")
                            rec["us"]["planes3_rag"] = timed(run_rag, reps)
                            rec["err"]["planes3_rag"] = float((y.double() - ref_r).abs().max()) / float(ref_r.abs().max())
                            rec["rag_L"] = int(rr.shape[1])
                        rec["img_ok"] = None if yp is None else bool(torch.equal(to_p3(y), yp))
                        # round 6: GROUPED lists (conv_p3g_kernel): rows with overlapping lists share the union - forward from the table,
                        # backward-data from the ragged lists; same float64 references, image of the result checked like the others'
                        grp = st.dev.get("bgrp" if bwd else "fgrp")
                        if grp is not None and lib.sh_spiral_conv_p3_grp_ok(B, S, Cg, Nout, int(grp[0].shape[1])) and (not bwd or ref_r is not None):
                            g_r, g_p, g_o = grp
                            ref_g = ref_r if bwd else ref
                            y.zero_()
                            ypg = torch.empty(nimg, dtype=torch.uint8, device=dev) if nimg else None

                            def run_grp():
                                _lib.check(lib.sh_spiral_conv_p3_grp(_lib.ptr(xp), _lib.ptr(g_r), _lib.ptr(g_p), _lib.ptr(g_o), int(g_r.shape[0]), int(g_r.shape[1]),
                                                                     _lib.ptr(wf), None if bwd else _lib.ptr(bias), _lib.ptr(y), B * Nout, Nout, _lib.ptr(ypg), None, 0, 0,
                                                                     None, 0, -1, 1 if bwd else 0, B, rows_out, S, Cg, Nout, _lib.stream_ptr()),
                                           "sh_spiral_conv_p3_grp")
                            rec["us"]["planes3_grp"] = timed(run_grp, reps)
                            rec["err"]["planes3_grp"] = float((y.double() - ref_g).abs().max()) / float(ref_g.abs().max())
                            rec["grp"] = (int(g_r.shape[0]), int(g_r.shape[1]), int((g_p.view(torch.int32) != -1).sum()))
                            rec["grp_img_ok"] = None if ypg is None else bool(torch.equal(to_p3(y), ypg))
                        rec["us_to_p3"] = timed(lambda: to_p3(x), reps)
                    yield rec
                    del x64, w64, ref
    finally:
        _lib.set_f32_mma_mode(was)


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    flags = [a for a in sys.argv[1:] if a.startswith("--")]
    B = int(args[0]) if args else 64
    tpl = args[1] if len(args) > 1 else None
    reps = 20
    for f in flags:
        if f.startswith("--reps="):
            reps = int(f.split("=")[1])
    dirs = ("bwd",) if "--bwd" in flags else (("fwd", "bwd") if "--both" in flags else ("fwd",))
    print("%-34s %-30s %-30s" % ("layer", "max|err|/max|ref| exact split3 p3", "us exact split3 p3 (+to_p3)"))
    for r in probe_layers(B, tpl, dirs, "--adversarial" in flags, reps, "--noskip" not in flags):
        if r["ok"]:
            p3s = "%.2e" % r["err"]["planes3"]
            t3 = "%6.1f (+%.1f)%s" % (r["us"]["planes3"], r["us_to_p3"], "" if r["img_ok"] is None else " img=" + ("ok" if r["img_ok"] else "MISMATCH"))
        else:
            p3s, t3 = "   -   ", "  -"
        rag = "   ragged lists (L %d): err %.2e, %6.1f us" % (r["rag_L"], r["err"]["planes3_rag"], r["us"]["planes3_rag"]) if "planes3_rag" in r["us"] else ""
        if "planes3_grp" in r["us"]:
            rag += "   grouped (%d groups, L %d, %d entries): err %.2e, %6.1f us%s" % (r["grp"] + (r["err"]["planes3_grp"], r["us"]["planes3_grp"],
                                                                                        "" if r["grp_img_ok"] in (True, None) else " img=MISMATCH"))
        print("%-34s %.2e %.2e %-9s   %6.1f %6.1f %s%s" % (r["name"], r["err"]["exact"], r["err"]["split3"], p3s, r["us"]["exact"],
                                                           r["us"]["split3"], t3, rag), flush=True)


if __name__ == "__main__":
    main()

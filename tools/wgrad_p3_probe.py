#!/usr/bin/env python3
"""Per-layer probe of the three-plane weight gradient (csrc/wgrad_p3.hip) on the conv shapes of a template: error against a
float64 evaluation and kernel time, next to the exact fp32 MFMA kernel (wgrad_stream_kernel).

    python tools/wgrad_p3_probe.py [batch] [template.npz] [--adversarial] [--reps=N]

One line per conv step:  shape | max|err| exact / p3 (relative to max|ref|) for dW and dbias | us exact / p3.
Test infrastructure (float64 on the GPU through torch); nothing here is on the product path."""
from __future__ import annotations

import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))

import semantichuman_amd as sh                                    # noqa: E402
from semantichuman_amd import _lib, ops                            # noqa: E402
from semantichuman_amd.hierarchy import load_hierarchy            # noqa: E402
import p3_probe                                                    # noqa: E402
from p3_probe import FE, FD, rnd, timed, to_p3                     # noqa: E402


def wgrad_p3(dp_img, x_img, table, B, R, S, cin, cout, zero_row=-1):
    """-> (dW [cout, S*cin], dbias [cout]) from the kernel's slabs, summed here in float64 (the probe checks the kernel's
    products; the shared slab reduction has its own tests)."""
    lib = _lib.load()
    nb = lib.sh_spiral_conv_bwd_wgt_p3_workspace(B, R, S, cin, cout)
    assert nb > 0
    ws = torch.zeros(nb // 4, dtype=torch.float32, device=table.device)
    _lib.check(lib.sh_spiral_conv_bwd_wgt_p3(_lib.ptr(dp_img), zero_row, _lib.ptr(x_img), _lib.ptr(table), _lib.ptr(ws), nb, B, R, S, cin, cout,
                                             _lib.stream_ptr()), "sh_spiral_conv_bwd_wgt_p3")
    n = cout * S * cin
    nslab = nb // 4 // (n + cout)
    dW = ws[:nslab * n].view(nslab, cout, S * cin)
    db = ws[nslab * n:].view(nslab, cout)
    return dW, db, ws, nb


def probe_layers(B=64, tpl=None, adversarial=False, reps=1, seed=1, local_table=False):
    tpl = tpl or os.path.join(ROOT, "tests", "golden", "template6890.npz")
    dev = torch.device("cuda:0")
    lib = _lib.load()
    h = load_hierarchy(tpl)
    torch.manual_seed(0)
    model = sh.SpiralAutoencoder(FE, FD, 256, h.sizes, h.spiral_sizes, h.spirals, h.D, h.U, dev)
    gen = torch.Generator(device=dev)
    gen.manual_seed(seed)
    was = _lib.get_f32_mma_mode()
    try:
        for sname, stack in (("enc", model._enc_stack), ("dec", model._dec_stack)):
            for i, st in enumerate(stack.steps):
                if st.kind != "conv" or st.cin == 3 or st.cout == 3:
                    continue
                R, S, cin, cout, n_in = st.R, st.S, st.cin, st.cout, st.n_in
                table = st.dev["table"]
                if local_table:       # experiment: perfect locality - vertex v gathers rows v, v + 1, ... (what would a local numbering buy?)
                    table = ((torch.arange(R, device=dev)[:, None] + torch.arange(S, device=dev)[None, :]) % n_in).to(torch.int32).contiguous()
                x = rnd((n_in, B, cin), dev, adversarial, gen)
                dp = rnd((R, B, cout), dev, adversarial, gen)
                dp[st.zero_row] = 0
                ok = bool(lib.sh_spiral_conv_bwd_wgt_p3_ok(B, R, S, cin, cout))
                # float64 reference, position by position
                dp64 = dp.double().reshape(R * B, cout)
                ref = torch.empty((cout, S, cin), dtype=torch.float64, device=dev)
                for s in range(S):
                    g = x[table[:, s].long()].double().reshape(R * B, cin)
                    ref[:, s, :] = dp64.t() @ g
                ref = ref.reshape(cout, S * cin)
                refb = dp64.sum(0)
                scale, scaleb = float(ref.abs().max()), float(refb.abs().max())
                rec = {"name": "%s%d R=%d K=%d N=%d" % (sname, i, R, S * cin, cout), "ok": ok, "err": {}, "errb": {}, "us": {}}
                _lib.set_f32_mma_mode("exact")
                out = {}

                def run_exact():
                    out["e"] = ops.spiral_conv_bwd_wgt(dp, "vm", x, "vm", table, R, S, cin, cout)
                rec["us"]["exact"] = timed(run_exact, reps)
                dW, db = out["e"]
                rec["err"]["exact"] = float((dW.double() - ref).abs().max()) / scale
                rec["errb"]["exact"] = float((db.double() - refb).abs().max()) / scaleb
                if ok:
                    xi, di = to_p3(x), to_p3(dp)

                    def run_p3():
                        out["p"] = wgrad_p3(di, xi, table, B, R, S, cin, cout, st.zero_row)
                    rec["us"]["p3"] = timed(run_p3, reps)
                    dWs, dbs, _, _ = out["p"]
                    rec["err"]["p3"] = float((dWs.double().sum(0) - ref).abs().max()) / scale
                    rec["errb"]["p3"] = float((dbs.double().sum(0) - refb).abs().max()) / scaleb
                    rec["nslab"] = dWs.shape[0]
                yield rec
                del ref, dp64
    finally:
        _lib.set_f32_mma_mode(was)


def presum_check():
    """sh_spiral_conv_bwd_wgt_p3_presum: rows and image of the pre-sum job are sh_spmm's, the slabs those of the launch without a job
    (bitwise); prints which way the job ran (the library reads SH_WP3_TAIL once per process)."""
    import numpy as np
    lib = _lib.load()
    d = torch.device("cuda:0")
    torch.manual_seed(5)
    B, R, n_in, S, cin, cout, n_sum = 64, 301, 407, 7, 32, 32, 157
    table = torch.randint(0, n_in, (R, S), dtype=torch.int32, device=d)
    x = torch.randn(n_in, B, cin, device=d)
    dp = torch.randn(R + n_sum, B, cout, device=d)
    g = np.random.RandomState(1)
    rowptr = np.concatenate([[0], np.cumsum(g.randint(1, 9, size=n_sum))]).astype(np.int32)      # 1..8 entries: both tail paths
    col = g.randint(0, R, size=rowptr[-1]).astype(np.int32)
    val = np.ones(rowptr[-1], dtype=np.float32)
    m = tuple(torch.from_numpy(a).to(d) for a in (rowptr, col, val))
    xi, di = to_p3(x), to_p3(dp[:R].contiguous())
    assert lib.sh_spiral_conv_bwd_wgt_p3_ok(B, R, S, cin, cout)
    nb = lib.sh_spiral_conv_bwd_wgt_p3_workspace(B, R, S, cin, cout)
    ws0 = torch.zeros(nb // 4, dtype=torch.float32, device=d)
    ws1 = torch.zeros_like(ws0)
    _lib.check(lib.sh_spiral_conv_bwd_wgt_p3(_lib.ptr(di), -1, _lib.ptr(xi), _lib.ptr(table), _lib.ptr(ws0), nb, B, R, S, cin, cout, _lib.stream_ptr()), "p3")
    want = dp.clone()
    ops.spmm(m, want, "vm", want[R:], "vm", n_sum)
    for with_img in (False, True):
        got = dp.clone()
        img = torch.zeros(lib.sh_p3_bytes(n_sum, B, cout), dtype=torch.uint8, device=d) if with_img else None
        ws1.zero_()
        _lib.profile_enable(True)
        _lib.check(lib.sh_spiral_conv_bwd_wgt_p3_presum(_lib.ptr(di), -1, _lib.ptr(xi), _lib.ptr(table), _lib.ptr(ws1), nb, _lib.ptr(got), B * cout, cout,
                                                        _lib.ptr(m[0]), _lib.ptr(m[1]), _lib.ptr(m[2]), _lib.ptr(got[R:]), _lib.ptr(img), n_sum, B, R, S,
                                                        cin, cout, _lib.stream_ptr()), "p3_presum")
        torch.cuda.synchronize()
        recs = _lib.profile_records()
        _lib.profile_enable(False)
        assert torch.equal(got, want), "rows differ from sh_spmm's"
        assert torch.equal(ws1, ws0), "slabs differ from the launch without a job"
        if with_img:
            assert torch.equal(img, to_p3(want[R:].contiguous())), "image differs"
        tail = any("wgrad_p3" in n and "presum=%d" % n_sum in n for n, _ in recs)
        print("launches:", [n.split("|")[0] for n, _ in recs], "tail_blocks>0" if tail else "own launch")
    print("PRESUM OK")


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    flags = [a for a in sys.argv[1:] if a.startswith("--")]
    if "--presum-check" in flags:
        return presum_check()
    B = int(args[0]) if args else 64
    tpl = args[1] if len(args) > 1 else None
    reps = 20
    for f in flags:
        if f.startswith("--reps="):
            reps = int(f.split("=")[1])
    print("%-30s %-22s %-22s %s" % ("layer", "dW err exact / p3", "dbias err exact / p3", "us exact / p3 (slabs)"))
    for r in probe_layers(B, tpl, "--adversarial" in flags, reps, local_table="--local-table" in flags):
        if r["ok"]:
            print("%-30s %.2e %.2e    %.2e %.2e    %6.1f %6.1f (%d)" % (r["name"], r["err"]["exact"], r["err"]["p3"], r["errb"]["exact"],
                                                                       r["errb"]["p3"], r["us"]["exact"], r["us"]["p3"], r["nslab"]), flush=True)
        else:
            print("%-30s %.2e    -        %.2e    -        %6.1f    -" % (r["name"], r["err"]["exact"], r["errb"]["exact"], r["us"]["exact"]), flush=True)


if __name__ == "__main__":
    main()

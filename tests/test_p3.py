"""The three-plane form of the fp32 path's conv products (csrc/p3_conv.hip, SH_MMA_PLANES3) against a float64 evaluation -
the gate under which a bf16x3 form may be reported as fp32 at all (reference arithmetic: models.py:34-53, aten::addmm in
fp32): for every conv shape of BASELINE config 2 (6890-vertex template, batch 64), forward and backward-data,

    max|y_form - y_f64|   <=   1.5 * max|y_exact - y_f64|          form in {split3, planes3}

on (i) activations / weights of the training step's scale and (ii) adversarial operands: six decades of dynamic range and
sums that cancel.  Plus what the form rests on: the plane image IS the tensor (h + m + l == x, bit for bit, in the documented
fragment-major layout), and every producer's image is the image of what it stored.

The parity of whole training steps in this form is covered by every fp32 GPU test of the suite (tests/conftest.py runs them
in all three forms at the same tolerances)."""
import ctypes
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))

pytestmark = pytest.mark.gpu


def dev():
    return torch.device("cuda:0")


def decode_image(img: torch.Tensor, rows: int, B: int, C: int) -> torch.Tensor:
    """Plane image (uint8 tensor, layout of include/sh_kernels.h / csrc/p3_conv.hip) -> the fp32 tensor [rows][B][C] it encodes,
    h + m + l evaluated in fp32 (exact: 8 + 8 + 8 significand bits)."""
    raw = img.cpu().numpy().view(np.uint16)
    nbg = B // 16
    if C == 16:
        a = raw.reshape(rows, nbg, 3, 2, 16, 8)                       # [row][bg][plane][kb2][b][8 ch]
        a = a.transpose(2, 0, 1, 4, 3, 5).reshape(3, rows, nbg * 16, 16)
    else:
        a = raw.reshape(rows, nbg, C // 32, 3, 4, 16, 8)              # [row][bg][cg][plane][kb][b][8 ch]
        a = a.transpose(3, 0, 1, 5, 2, 4, 6).reshape(3, rows, nbg * 16, C)
    f = (a.astype(np.uint32) << 16).view(np.float32)
    return torch.from_numpy(np.ascontiguousarray((f[0] + f[1]) + f[2]))


@pytest.mark.parametrize("rows,B,C", [(37, 16, 16), (5, 64, 32), (19, 32, 64), (3, 48, 128)])
def test_plane_image_is_the_tensor(rows, B, C):
    import p3_probe
    torch.manual_seed(rows)
    x = torch.randn(rows, B, C) * torch.pow(10.0, 6 * torch.rand(rows, B, C) - 3)       # six decades
    x[0, 0, :4] = torch.tensor([0.0, -0.0, 1.0, -1.0e-30])                                # zeros, a tiny normal number
    img = p3_probe.to_p3(x.to(dev()))
    torch.cuda.synchronize()
    back = decode_image(img, rows, B, C)
    assert torch.equal(back, x), float((back - x).abs().max())


@pytest.mark.parametrize("adversarial", [False, True])
def test_error_of_the_split_forms_against_float64(adversarial):
    """The gate.  Records every layer's three errors (relative to max|ref|) so a failure says which layer and by how much."""
    import p3_probe
    rows = list(p3_probe.probe_layers(64, None, ("fwd", "bwd"), adversarial, reps=1))
    assert len(rows) == 14 and all(r["ok"] for r in rows), [r["name"] for r in rows if not r["ok"]]
    bad = []
    for r in rows:
        e = r["err"]
        for form in ("split3", "planes3"):
            if not e[form] <= 1.5 * e["exact"] + 1e-9:
                bad.append((r["name"], form, e[form], e["exact"]))
        assert r["img_ok"] in (True, None), r["name"]
        assert e["exact"] < 5e-6, (r["name"], e)                    # the exact form itself is at fp32 level on these sums
        # round 6: backward-data over ragged source lists (conv_p3r_kernel) - its own float64 reference (the sums over a list's
        # sources are formed by the matrix pipe instead of by a pre-sum launch), the same gate
        if "planes3_rag" in e and not e["planes3_rag"] <= 1.5 * e["exact"] + 1e-9:
            bad.append((r["name"], "planes3_rag", e["planes3_rag"], e["exact"]))
        # ... and over GROUPED lists (conv_p3g_kernel: rows with overlapping lists share one list of the union), both directions
        if "planes3_grp" in e:
            if not e["planes3_grp"] <= 1.5 * e["exact"] + 1e-9:
                bad.append((r["name"], "planes3_grp", e["planes3_grp"], e["exact"]))
            assert r["grp_img_ok"] in (True, None), r["name"]
    assert not bad, bad
    assert sum(1 for r in rows if "planes3_rag" in r["err"]) == 4, [r["name"] for r in rows if "planes3_rag" in r["err"]]
    # 5 forward (enc2 gathers 16 channels), 5 backward (the level-0 layer's gradient has 16)
    assert sum(1 for r in rows if "planes3_grp" in r["err"]) == 10, [r["name"] for r in rows if "planes3_grp" in r["err"]]


def test_nine_products_are_not_needed():
    """Six of the nine partial products: the three dropped ones are below 2^-24 |w||x| each.  With all nine (SH_P3_NP=9 - read
    once per process, so a child process) the error against float64 does not improve by more than rounding: the six-product
    form is not what limits the accuracy."""
    import json
    import subprocess
    code = ("import sys, json; sys.path.insert(0, %r); import p3_probe; "
            "print(json.dumps([r['err'] for r in p3_probe.probe_layers(64, None, ('fwd',), True, 1)]))" % os.path.join(ROOT, "tools"))
    out = {}
    for np_ in ("6", "9"):
        r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, SH_P3_NP=np_), capture_output=True, text=True, timeout=600,
                           cwd=ROOT)
        assert r.returncode == 0, r.stderr[-2000:]
        out[np_] = json.loads([l for l in r.stdout.splitlines() if l.startswith("[")][-1])
    for e6, e9 in zip(out["6"], out["9"]):
        assert e6["planes3"] <= 1.5 * e9["planes3"] + 1e-9, (e6, e9)


def test_producers_write_the_image_of_what_they_store():
    """sh_spmm_p3 (up-sampling rows, pre-summed rows, U^T with the activation derivative) and the thin kernel's fused input
    gradient: image == sh_to_p3(fp32 result), bit for bit."""
    import p3_probe
    from semantichuman_amd import _lib, ops
    from semantichuman_amd.mesh_ops import CSR
    lib = _lib.load()
    d = dev()
    torch.manual_seed(3)
    rows, cols, B = 211, 97, 32
    g = np.random.RandomState(0)
    rowptr = np.concatenate([[0], np.cumsum(g.randint(1, 4, size=rows))]).astype(np.int32)
    col = g.randint(0, cols, size=rowptr[-1]).astype(np.int32)
    val = g.randn(rowptr[-1]).astype(np.float32)
    m = tuple(torch.from_numpy(a).to(d) for a in (rowptr, col, val))
    for C in (16, 32, 64):
        x = torch.randn(cols, B, C, device=d)
        yprev = torch.randn(rows, B, C, device=d)
        y = torch.empty(rows, B, C, device=d)
        img = torch.empty(lib.sh_p3_bytes(rows, B, C), dtype=torch.uint8, device=d)
        for yp, act in ((None, 0), (yprev, 2)):
            _lib.check(lib.sh_spmm_p3(_lib.ptr(m[0]), _lib.ptr(m[1]), _lib.ptr(m[2]), _lib.ptr(x), B * C, C, _lib.ptr(y), B * C, C, _lib.ptr(img),
                                      _lib.ptr(yp), B * C if yp is not None else 0, C if yp is not None else 0, act, 7, B, rows, C,
                                      _lib.stream_ptr()), "sh_spmm_p3")
            y2 = torch.empty_like(y)
            ops.spmm(m, x, "vm", y2, "vm", rows, yprev=yp, yp_layout="vm", act_prev=act, zero_row=7)
            assert torch.equal(y, y2)                                   # the fp32 rows are sh_spmm's
            assert torch.equal(img, p3_probe.to_p3(y)), C
    del CSR


def test_stack_step_selects_the_plane_kernels():
    """In SH_MMA_PLANES3 the library's profiler must see the three-plane kernels on every conv of the plain autoencoder they are
    built for (and none in the other forms): a silent fallback would make the form's timings meaningless."""
    import semantichuman_amd as sh
    from semantichuman_amd import _lib, synthetic
    from semantichuman_amd.hierarchy import load_hierarchy
    FE = [[3, 16, 32, 64, 128], [[], [], [], [], []]]
    FD = [[128, 64, 32, 32, 16], [[], [], [], [], 3]]
    h = load_hierarchy(os.path.join(ROOT, "tests", "golden", "template6890.npz"))
    torch.manual_seed(1)
    m = sh.SpiralAutoencoder(FE, FD, 256, h.sizes, h.spiral_sizes, h.spirals, h.D, h.U, dev())
    x = torch.from_numpy(synthetic.synth_batch(h.verts, 32, seed=1)).to(dev())
    was = _lib.get_f32_mma_mode()
    counts = {}
    try:
        for mode in ("exact", "planes3"):
            _lib.set_f32_mma_mode(mode)
            m.zero_grad(set_to_none=True)
            sh.l1_loss(x, m(x)[0]).backward()
            torch.cuda.synchronize()
            _lib.profile_enable(True)
            m.zero_grad(set_to_none=True)
            sh.l1_loss(x, m(x)[0]).backward()
            torch.cuda.synchronize()
            names = [n for n, _, _ in _lib.profile_records_by_kernel()]
            _lib.profile_enable(False)
            counts[mode] = sum(1 for n in names if n.startswith("conv_p3"))
    finally:
        _lib.set_f32_mma_mode(was)
    assert counts["exact"] == 0
    assert counts["planes3"] == 14, counts          # 7 forward + 7 backward-data launches (every conv but the 3-channel sides)


@pytest.mark.parametrize("adversarial", [False, True])
@pytest.mark.parametrize("mnk", [(64, 256, 55296), (64, 55296, 256), (16, 256, 55296), (48, 1024, 4096)])
def test_latent_fc_bf16x3_error_against_float64(mnk, adversarial):
    """The same gate for the latent FCs (models.py:130,144; round 5: csrc/linear.hip runs their six GEMMs in the bf16x3 form when
    the caller's form is split3 / planes3 - both fp32 operands split exactly into three bf16 terms in registers, six partial
    products on v_mfma_f32_16x16x32_bf16, fp32 accumulation): forward, input gradient, weight gradient and bias gradient against
    a float64 evaluation, max-abs error <= 1.5 x the exact fp32 MFMA kernels' + one fp32 ulp of the result's scale - on
    training-scale operands and on adversarial ones (six decades of dynamic range, cancelling sums of 55 296 terms)."""
    from semantichuman_amd import _lib, ops
    M, N, K = mnk
    g = torch.Generator().manual_seed(M + N + K + (7 if adversarial else 0))
    if adversarial:
        mag = lambda *s: torch.pow(10.0, 6 * torch.rand(*s, generator=g) - 3)                     # noqa: E731
        x = torch.randn(M, K, generator=g) * mag(M, K)
        W = torch.randn(N, K, generator=g) * mag(N, K) / K ** 0.5
        dy = torch.randn(M, N, generator=g) * mag(M, N)
        x[:, 1::2] = -x[:, 0::2] * (1 + 1e-3 * torch.randn(M, K // 2, generator=g))                # sums that cancel
    else:
        x = torch.randn(M, K, generator=g)
        W = torch.randn(N, K, generator=g) / K ** 0.5
        dy = torch.randn(M, N, generator=g) * 1e-3
    b = torch.randn(N, generator=g)
    ref = {"y": x.double() @ W.double().T + b.double(), "dx": dy.double() @ W.double(), "dW": dy.double().T @ x.double(),
           "db": dy.double().sum(0)}
    d = dev()
    xd, Wd, bd, dyd = (t.to(d) for t in (x, W, b, dy))
    err = {}
    for form in ("exact", "planes3"):
        _lib.profile_enable(True)
        out = {"y": ops.linear_fwd(xd, Wd, bd, form), "dx": ops.linear_bwd_data(dyd, Wd, form)}
        out["dW"], out["db"] = ops.linear_bwd_wgt(dyd, xd, True, form)
        torch.cuda.synchronize()
        names = [n for n, _, _ in _lib.profile_records_by_kernel()]
        _lib.profile_enable(False)
        assert sum(1 for n in names if "_x3_" in n) == (3 if form == "planes3" else 0), names      # the kernels under test really ran
        err[form] = {k: float((out[k].double().cpu() - ref[k]).abs().max()) for k in ref}
    for k in ref:
        scale = float(ref[k].abs().max())
        assert err["planes3"][k] <= 1.5 * err["exact"][k] + 2.0 ** -23 * scale, (k, err["planes3"][k], err["exact"][k], scale)
        assert err["exact"][k] <= 2e-5 * scale, (k, err["exact"][k], scale)
    print("FC %s %s: " % (mnk, "adversarial" if adversarial else "training-scale") +
          "  ".join("%s %.2e / %.2e" % (k, err["planes3"][k] / float(ref[k].abs().max()), err["exact"][k] / float(ref[k].abs().max())) for k in ref))


# ---------------------------------------------------------------------------------------------------------------------------
# round 6: the weight gradient in the three-plane form (csrc/wgrad_p3.hip; autograd of reference models.py:45, dW = dpre^T . gather(x))

@pytest.mark.parametrize("tpl,B,n_taken", [("template6890.npz", 64, 6), ("template27554.npz", 32, 6), ("template6890.npz", 16, 6),
                                           ("template6890.npz", 48, 6)])
@pytest.mark.parametrize("adversarial", [False, True])
def test_three_plane_weight_gradient_error_against_float64(tpl, B, n_taken, adversarial):
    """The same gate as the convs': max|dW - dW_f64| of the plane kernel <= 1.5 x that of the exact fp32 MFMA kernel, for every conv
    layer of config 2 (batch 64: two batch pairs per vertex) and config 4 (27 554 vertices, spiral 18, batch 32: one), sums over
    up to 441 024 (vertex, batch) rows; training-scale operands and adversarial ones (six decades of dynamic range).  dbias too.
    Batch 16 and 48 (the semantic loop's): a 32-row step pairs 16-row units of DIFFERENT vertices, and the layers with an odd
    number of units (863 rows x 1 or 3 groups) are completed by the layer's all-zero dummy row."""
    import wgrad_p3_probe
    rows = list(wgrad_p3_probe.probe_layers(B, os.path.join(ROOT, "tests", "golden", tpl), adversarial, reps=1))
    taken = [r for r in rows if r["ok"]]
    assert len(taken) == n_taken, [(r["name"], r["ok"]) for r in rows]
    bad = []
    for r in taken:
        for key in ("err", "errb"):
            e = r[key]
            assert e["exact"] < 5e-6, (r["name"], key, e)
            if not e["p3"] <= 1.5 * e["exact"] + 1e-9:
                bad.append((r["name"], key, e["p3"], e["exact"]))
    assert not bad, bad


@pytest.mark.parametrize("tail", ["0", "1"])
def test_three_plane_weight_gradient_carries_the_presum_job_bitwise(tail):
    """sh_spiral_conv_bwd_wgt_p3_presum, with the job as a launch of its own in front (SH_WP3_TAIL=0, the default) and as tail
    workgroups of the weight-gradient launch (SH_WP3_TAIL=1; read once per process, hence a child process): the rows written are
    sh_spmm's rows (and their image is the image of those rows), bit for bit, and the slabs are those of the launch without a job."""
    import subprocess
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "wgrad_p3_probe.py"), "--presum-check"], env=dict(os.environ, SH_WP3_TAIL=tail),
                       capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert r.returncode == 0 and "PRESUM OK" in r.stdout, (r.stdout[-2000:], r.stderr[-2000:])
    assert ("tail_blocks>0" in r.stdout) == (tail == "1"), r.stdout[-500:]


@pytest.mark.parametrize("tpl,B", [("template6890.npz", 64), ("template27554.npz", 32), ("template6890.npz", 48)])
def test_training_on_the_images_leaves_unread_fp32_rows_unwritten_and_changes_nothing(tpl, B):
    """sh_stack_forward(keep_fp32 == 2) / sh_stack_backward(acts_fp32 == 2) (round 6): fp32 rows that neither pass reads - the gathered
    input of a conv whose forward, weight gradient and activation derivative all run on its plane image; gradient rows handed to a
    step that reads them through their image alone - are not written.  One training step with the arenas poisoned with NaN
    (tools/drop_fp32_check.py; a read of an unwritten row would surface as NaN, not as the row of an earlier step): loss,
    reconstruction and every parameter gradient equal, BIT FOR BIT, those of SH_P3_DROP_FP32=0 where every row is written.  The
    switch is read once per process: two child processes."""
    import subprocess
    out = {}
    for drop in ("1", "0"):
        r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "drop_fp32_check.py"), os.path.join(ROOT, "tests", "golden", tpl), str(B)],
                           env=dict(os.environ, SH_P3_DROP_FP32=drop), capture_output=True, text=True, timeout=900, cwd=ROOT)
        lines = [l for l in r.stdout.splitlines() if l.startswith("DIGEST ")]
        assert r.returncode == 0 and lines, (r.stdout[-2000:], r.stderr[-3000:])
        out[drop] = lines[-1]
        n_img = int([l for l in r.stdout.splitlines() if l.startswith("IMAGE_ONLY ")][-1].split()[1])
        # 6890 vertices: 2 encoder convs, 3 up-sampling steps, 4 gradient hand-overs (9); 27 554 (wider spirals: more layers on streamed
        # weights, whose backward-data pass keeps the dense table and its fp32 pre-sums): 7
        assert (n_img >= (8 if "6890" in tpl else 5)) if drop == "1" else (n_img == 0), (drop, n_img)
    assert out["1"] == out["0"]

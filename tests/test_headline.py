"""The benchmark's own arithmetic form at the benchmark's own sizes, against the CPU oracle.

bench.py's headline runs the fp32 step in the three-plane form (SH_MMA_PLANES3, csrc/p3_conv.hip); the plane kernels need a batch
that is a multiple of 16, so the suite's B = 2 oracle comparisons exercise the split3 kernels in their [planes3] instances.  Here
a FULL training step - forward, L1 + 1e-2 x edge-ratio loss (train_funcs.py:501-508), EVERY parameter gradient, one Adam step
(main.py:262: lr 1e-3, coupled weight decay 5e-5) - of the plain autoencoder (models.py:34-53, 115-162) runs at

    6890 vertices,  batch 16 and batch 64 (BASELINE config 2: the headline)
    27 554 vertices, spiral length 18, batch 16 (BASELINE config 4's template)

in the exact fp32 form, in split3 and in planes3, against oracle/ref_cpu.py on the same inputs and weights, at the suite's
tolerances (FWD 1e-5, GRAD 1e-4); and the library's own profiler is asked which kernels ran: in planes3 every conv launch the
plane kernels are built for must BE a plane kernel, and the producers must have written images.
"""
import os

import numpy as np
import pytest
import torch

from oracle import ref_cpu
from semantichuman_amd.hierarchy import load_hierarchy

pytestmark = pytest.mark.gpu
FE = [[3, 16, 32, 64, 128], [[], [], [], [], []]]
FD = [[128, 64, 32, 32, 16], [[], [], [], [], 3]]
FWD_TOL, GRAD_TOL = 1e-5, 1e-4
EDGE_W, LR, WD = 1e-2, 1e-3, 5e-5

CASES = [("template6890.npz", 16), ("template6890.npz", 64), ("template27554.npz", 16)]
# conv launches of a planes3 step that must be plane kernels: every forward / backward-data launch except the 3-channel sides
# (enc0 forward; dec4 forward, whose backward-data is fused into the thin weight-gradient launch; enc0 has no backward-data)
P3_LAUNCHES = {"template6890.npz": 14, "template27554.npz": 14}


@pytest.fixture(scope="module", params=CASES, ids=lambda c: "%s-B%d" % (c[0].split(".")[0], c[1]))
def case(request, golden_dir):
    """Template, batch, initial weights and the ORACLE's step on them (computed once per case, shared by the three forms)."""
    tpl, B = request.param
    from semantichuman_amd import synthetic
    h = load_hierarchy(os.path.join(golden_dir, tpl))
    S = [torch.from_numpy(s.astype(np.int64))[None] for s in h.spirals]
    if h.sizes[0] > 10000:          # dense D / U of this template: 4 GB and 0.4 TFLOP per product - the pinned sparse form
        D, U = [ref_cpu.sparse_operator(d) for d in h.D], [ref_cpu.sparse_operator(u) for u in h.U]
    else:
        _, D, U = h.dense_constants()
    torch.manual_seed(20 + B)
    om = ref_cpu.SpiralAEOracle(FE, FD, 256, h.sizes, h.spiral_sizes, S, D, U)       # nn.Linear's default initialisation
    sd0 = {k: v.clone() for k, v in om.state_dict().items()}
    x = torch.from_numpy(synthetic.synth_batch(h.verts, B, seed=7))
    opt = torch.optim.Adam(om.parameters(), lr=LR, weight_decay=WD)
    opt.zero_grad()
    xo, zo = om(x)
    rec, edge = torch.nn.functional.l1_loss(x, xo), ref_cpu.edge_ratio_loss(xo, x, h.faces)
    (rec + EDGE_W * edge).backward()
    grads = {n: p.grad.clone() for n, p in om.named_parameters()}
    opt.step()
    w1 = {n: p.detach().clone() for n, p in om.named_parameters()}
    del om, opt
    return dict(tpl=tpl, B=B, h=h, x=x, sd0=sd0, x_hat=xo.detach(), z=zo.detach(), rec=float(rec.detach()), edge=float(edge.detach()), grads=grads, w1=w1)


def close(got, ref, tol, what, floor=0.0):
    got, ref = got.detach().cpu().numpy(), ref.detach().cpu().numpy()
    assert got.shape == ref.shape and np.isfinite(got).all(), what
    err, scale = np.abs(got - ref).max(), np.abs(ref).max()
    assert err <= tol * scale + floor + 1e-30, "%s: err %.3e > %.1e * %.3e + %.1e" % (what, err, tol, scale, floor)
    return err / (scale + 1e-30)


@pytest.mark.parametrize("form", ["exact", "split3", "planes3"])
def test_full_training_step_vs_oracle(case, form, record_property):
    import semantichuman_amd as sh
    from semantichuman_amd import _lib
    dev = torch.device("cuda:0")
    h, B = case["h"], case["B"]
    was = _lib.get_f32_mma_mode()
    _lib.set_f32_mma_mode(form)
    try:
        m = sh.SpiralAutoencoder(FE, FD, 256, h.sizes, h.spiral_sizes, h.spirals, h.D, h.U, dev)
        m.load_state_dict(case["sd0"])
        opt = sh.optim.Adam(m.parameters(), lr=LR, weight_decay=WD)
        ft = sh.FaceTables(h.faces, h.sizes[0] + 1, dev)
        x = case["x"].to(dev)
        _lib.profile_enable(True)
        opt.zero_grad()
        x_hat, z = m(x)
        rec, edge = sh.l1_loss(x, x_hat), sh.edge_ratio_loss(x_hat, x, ft)
        (rec + EDGE_W * edge).backward()
        torch.cuda.synchronize()
        names = [n for n, _, _ in _lib.profile_records_by_kernel()]
        _lib.profile_enable(False)
        # ---- which kernels ran
        n_p3 = sum(1 for n in names if n.startswith("conv_p3"))
        off_plane = [n for n in names if n.startswith(("gather_gemm_direct", "gather_gemm_split3", "to_p3"))]
        if form == "planes3":
            assert n_p3 == P3_LAUNCHES[case["tpl"]], (n_p3, sorted(set(names)))
            assert not off_plane, off_plane                           # no conv left on the exact / split3 kernels, no stand-alone image pass
            assert sum(1 for n in names if n.startswith("spmm_kernel<true, p3>")) >= 6      # re-sampling + U^T launches wrote images
            assert any(n.startswith("wfrag3_prep") for n in names)
            # round 6: the weight gradients of the layers between the 3-channel sides and the 16-channel level-0 layer run on the
            # images too (csrc/wgrad_p3.hip; any batch that has images, i.e. multiples of 16 - at batch 16 a stage pairs two vertices and
            # the 863-row layers' odd unit count is completed by the dummy row): six of the nine, the rest on the fp32 kernels
            # ... and the backward-data pass of the four layers with a resident weight walks ragged source lists: no pre-sum launch
            # of theirs is left (the two streamed-weight layers and the level-0 layers keep the dense table)
            lib = _lib.load()
            n_rag = 0
            for stack in (m._enc_stack, m._dec_stack):
                for i, st in enumerate(stack.steps):
                    first = stack is m._enc_stack and i == 0                    # no backward-data pass
                    if st.kind == "conv" and not first and getattr(st, "rag", None) is not None and \
                            lib.sh_spiral_conv_p3_rag_ok(B, st.S, st.cout, st.cin, int(st.rag[0].shape[1])):
                        n_rag += 1
            assert n_rag == (4 if case["tpl"] == "template6890.npz" else n_rag) and n_rag >= 2
            # ... as GROUPS of input rows whose source lists overlap (conv_p3g_kernel<.., true, ..>), and so do the forward passes of
            # the resident-weight layers that gather a multiple of 32 channels (conv_p3g_kernel<.., false, ..>)
            # (+ the level-0 layer, whose gradient has 16 channels: only the grouped kernel takes that)
            n_bg16 = sum(1 for stack in (m._enc_stack, m._dec_stack) for i, st in enumerate(stack.steps)
                         if st.kind == "conv" and not (stack is m._enc_stack and i == 0) and st.cout == 16 and getattr(st, "bgrp", None) is not None
                         and lib.sh_spiral_conv_p3_grp_ok(B, st.S, st.cout, st.cin, int(st.bgrp[0].shape[1]))
                         and lib.sh_spiral_conv_p3_grp_pays(B, int(st.bgrp[0].shape[0])))
            assert sum(1 for n in names if n.startswith("conv_p3r_kernel") or (n.startswith("conv_p3g_kernel") and ", true, " in n.split("<")[1][:20])) \
                == n_rag + n_bg16, (n_rag, n_bg16, sorted(set(names)))
            # (where the launch has enough groups to fill the chip: sh_spiral_conv_p3_grp_pays)
            n_fg = sum(1 for stack in (m._enc_stack, m._dec_stack) for st in stack.steps if st.kind == "conv" and getattr(st, "fgrp", None) is not None
                       and lib.sh_spiral_conv_p3_grp_pays(B, int(st.fgrp[0].shape[0])))
            assert (n_fg >= 2 or B < 64) and sum(1 for n in names if n.startswith("conv_p3g_kernel") and ", false, " in n.split("<")[1][:20]) == n_fg, (n_fg, sorted(set(names)))
            n_wp3 = sum(1 for n in names if n.startswith("wgrad_p3_kernel"))
            assert n_wp3 == 6, (n_wp3, sorted(set(names)))
            assert sum(1 for n in names if n.startswith(("wgrad_stream", "wgrad_split3"))) == 2, sorted(set(names))
        else:
            assert n_p3 == 0 and not any("p3" in n for n in names), sorted(set(names))
        # ---- forward, loss
        e_x = close(x_hat, case["x_hat"], FWD_TOL, "x_hat")
        e_z = close(z, case["z"], FWD_TOL, "z")
        assert float(x_hat[:, -1].abs().max()) == 0.0
        assert rec.item() == pytest.approx(case["rec"], rel=1e-5) and edge.item() == pytest.approx(case["edge"], rel=1e-5)
        # ---- every parameter gradient
        gmax = max(float(g.abs().max()) for g in case["grads"].values())
        worst = 0.0
        for name, prm in m.named_parameters():
            worst = max(worst, close(prm.grad, case["grads"][name], GRAD_TOL, "grad " + name, floor=1e-6 * gmax))
        # ---- one Adam step.  The first update is -lr g / (|g| + 1e-8) ~ -lr sign(g): where |g| is far above the gradient
        # tolerance the weights must agree to fp32 rounding; where g ~ 0 the sign is ill-conditioned and an element may move
        # by up to 2 lr.
        opt.step()
        torch.cuda.synchronize()
        for name, prm in m.named_parameters():
            g, w1 = case["grads"][name], case["w1"][name]
            d = (prm.detach().cpu() - w1).abs()
            big = g.abs() >= 100 * GRAD_TOL * float(g.abs().max())
            assert float(d.max()) <= 2 * LR * 1.05, name
            if bool(big.any()):       # d(update) = lr eps dg / (|g| + eps)^2: <= 0.5 % of lr for |dg| <= 1e-2 |g|
                assert float(d[big].max()) <= 5e-6 + 1e-6 * float(w1.abs().max()), (name, float(d[big].max()))
            assert float(d.mean()) <= 2e-6, (name, float(d.mean()))
        record_property("rel_err", {"x_hat": e_x, "z": e_z, "worst_grad": worst})
        print("%s B=%d %s: x_hat %.2e z %.2e worst grad %.2e plane launches %d" % (case["tpl"], B, form, e_x, e_z, worst, n_p3))
    finally:
        _lib.profile_enable(False)
        _lib.set_f32_mma_mode(was)


@pytest.mark.parametrize("form", ["exact", "planes3"])
def test_the_benchs_own_step_vs_oracle(case, form):
    """The step exactly as bench.py issues it - the fused reconstruction loss, Adam with the two latent FCs' update applied
    inside their weight-gradient kernels (no FC weight gradient is ever materialised) - against the oracle's weights after one
    `loss.backward(); optimizer.step()` (train_funcs.py:383-392), with the post-Adam criterion of the test above; the profiler
    must show that the fused kernels are what ran."""
    import semantichuman_amd as sh
    from semantichuman_amd import _lib
    dev = torch.device("cuda:0")
    h, B = case["h"], case["B"]
    was = _lib.get_f32_mma_mode()
    _lib.set_f32_mma_mode(form)
    try:
        m = sh.SpiralAutoencoder(FE, FD, 256, h.sizes, h.spiral_sizes, h.spirals, h.D, h.U, dev)
        m.load_state_dict(case["sd0"])
        opt = sh.optim.Adam(m.parameters(), lr=LR, weight_decay=WD)
        opt.fuse_linear_weight_gradients([m.fc_latent_enc, m.fc_latent_dec])
        ft = sh.FaceTables(h.faces, h.sizes[0] + 1, dev)
        x = case["x"].to(dev)
        _lib.profile_enable(True)
        opt.zero_grad(set_to_none=True)
        x_hat, _ = m(x)
        loss, parts = sh.recon_loss(x_hat, x, ft, EDGE_W)
        loss.backward()
        opt.step()
        torch.cuda.synchronize()
        names = [n for n, _, _ in _lib.profile_records_by_kernel()]
        _lib.profile_enable(False)
        assert sum(1 for n in names if n.startswith("linear_bwd_wgt_adam")) == 2, sorted(set(names))
        assert not any(n.startswith(("linear_bwd_wgt_x3", "linear_bwd_wgt_dma", "linear_bwd_wgt_stream")) for n in names)
        assert m.fc_latent_enc.weight.grad is None and m.fc_latent_dec.weight.grad is None
        assert float(parts[0]) == pytest.approx(case["rec"], rel=1e-5) and float(parts[1]) == pytest.approx(case["edge"], rel=1e-5)
        for name, prm in m.named_parameters():
            g, w1 = case["grads"][name], case["w1"][name]
            d = (prm.detach().cpu() - w1).abs()
            big = g.abs() >= 100 * GRAD_TOL * float(g.abs().max())
            assert float(d.max()) <= 2 * LR * 1.05, name
            if bool(big.any()):
                assert float(d[big].max()) <= 5e-6 + 1e-6 * float(w1.abs().max()), (name, float(d[big].max()))
            assert float(d.mean()) <= 2e-6, (name, float(d.mean()))
            assert float(opt.state[prm]["step"]) == 1.0, name
        opt.remove_fusion()
    finally:
        _lib.profile_enable(False)
        _lib.set_f32_mma_mode(was)


@pytest.mark.parametrize("form", ["exact", "planes3"])
def test_replayed_graph_step_is_bitwise_the_eager_step(form, golden_dir):
    """The launch path `bench.py` times - forward + fused L1/edge loss + backward + Adam captured into ONE hipGraph, the latent
    FCs' update inside their weight-gradient kernels (optim.Adam.fuse_linear_weight_gradients), the batch copied into a fixed
    input tensor before every replay - against the same three training steps (train_funcs.py:495-510) launched eagerly from the
    same weights on the same batches: every parameter, both Adam moments and the step counts equal BIT FOR BIT, at the headline
    size (6890 vertices, batch 64), in the exact form and in the three-plane form."""
    import semantichuman_amd as sh
    from semantichuman_amd import _lib, synthetic
    dev = torch.device("cuda:0")
    h = load_hierarchy(os.path.join(golden_dir, "template6890.npz"))
    B, n_steps = 64, 3
    data = torch.from_numpy(synthetic.synth_batch(h.verts, n_steps * B, seed=11)).to(dev)
    ft = sh.FaceTables(h.faces, h.sizes[0] + 1, dev)
    was = _lib.get_f32_mma_mode()
    _lib.set_f32_mma_mode(form)
    try:
        torch.manual_seed(4)
        m = sh.SpiralAutoencoder(FE, FD, 256, h.sizes, h.spiral_sizes, h.spirals, h.D, h.U, dev)
        init = {k: v.detach().clone() for k, v in m.state_dict().items()}
        opt = sh.optim.Adam(m.parameters(), lr=LR, weight_decay=WD)
        opt.fuse_linear_weight_gradients([m.fc_latent_enc, m.fc_latent_dec])
        xin = torch.empty((B, h.sizes[0] + 1, 3), dtype=torch.float32, device=dev)
        unit = torch.ones((), dtype=torch.float32, device=dev)

        def one_step():
            opt.zero_grad(set_to_none=True)
            loss, _ = sh.recon_loss(m(xin)[0], xin, ft, EDGE_W)
            loss.backward(unit)
            opt.step()

        def reset():
            m.load_state_dict(init)
            for st in opt.state.values():
                for v in st.values():
                    if torch.is_tensor(v):
                        v.zero_()

        def snapshot():
            torch.cuda.synchronize()
            out = {"p/" + n: q.detach().clone() for n, q in m.named_parameters()}
            for n, q in m.named_parameters():
                for k, v in opt.state[q].items():
                    out["s/%s/%s" % (n, k)] = v.detach().clone()
            return out
        # --- eager
        xin.copy_(data[:B])
        one_step()                                   # creates the optimizer state; then back to the start
        reset()
        _lib.profile_enable(True)
        for i in range(n_steps):
            xin.copy_(data[i * B:(i + 1) * B])
            one_step()
        torch.cuda.synchronize()
        names = [n for n, _, _ in _lib.profile_records_by_kernel()]
        _lib.profile_enable(False)
        assert sum(1 for n in names if n.startswith("linear_bwd_wgt_adam")) == 2 * n_steps, sorted(set(names))
        if form == "planes3":
            assert sum(1 for n in names if n.startswith("conv_p3")) == 14 * n_steps
        eager = snapshot()
        assert float(eager["s/fc_latent_enc.weight/step"]) == n_steps
        # --- captured: warm-up on a side stream, capture one step, back to the start, three replays
        reset()
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(2):
                one_step()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            one_step()
        reset()
        for i in range(n_steps):
            xin.copy_(data[i * B:(i + 1) * B])
            graph.replay()
        replayed = snapshot()
        assert set(eager) == set(replayed)
        diff = [k for k in eager if not torch.equal(eager[k], replayed[k])]
        assert not diff, diff[:8]
        assert not torch.equal(eager["p/fc_latent_dec.weight"], init["fc_latent_dec.weight"])       # it did train
        opt.remove_fusion()
    finally:
        _lib.profile_enable(False)
        _lib.set_f32_mma_mode(was)

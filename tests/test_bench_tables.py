"""bench.py's pricing tables without a GPU: the model's launch plan is pure host data (stack.py builds it in numpy, the shape
predicates of the kernel library are host functions), so the algorithmic FLOP / byte tables the roofline block is computed from
can be checked for internal consistency on the CPU box - a dispatch or table change cannot silently detach `roofline`,
`whole_step.matrix_pipes` and `step_work` from each other."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

FE = [[3, 16, 32, 64, 128], [[], [], [], [], []]]
FD = [[128, 64, 32, 32, 16], [[], [], [], [], 3]]


@pytest.fixture(scope="module", params=["template6890.npz", "template27554.npz"])
def model(request, golden_dir):
    import semantichuman_amd as sh
    from semantichuman_amd.hierarchy import load_hierarchy
    h = load_hierarchy(os.path.join(golden_dir, request.param))
    return sh.SpiralAutoencoder(FE, FD, 256, h.sizes, h.spiral_sizes, h.spirals, h.D, h.U, None), h      # device None: host tables only


@pytest.mark.parametrize("B", [16, 64])
def test_work_tables_are_consistent(model, B):
    import bench
    m, h = model
    table = bench.f32_work_table(m, B)
    entries = {id(v): v for v in table.values()}                       # both spellings of a 3-channel key share one entry
    conv_steps = [st for stack in (m._enc_stack, m._dec_stack) for st in stack.steps if st.kind == "conv"]
    fwd = sum(2.0 * B * st.R * st.S * st.cin * st.cout for st in conv_steps)
    # every conv has a forward and a weight-gradient entry, every conv but the first a backward-data one; the thin layer's weight
    # gradient and backward-data are ONE launch priced with both
    assert sum(v["flops"] for v in entries.values()) == pytest.approx(3 * fwd - 2.0 * B * conv_steps[0].R * conv_steps[0].S * 3 * 16)
    flops_step, bytes_step = bench.step_work(m, B, "f32")
    fc = sum(3 * 2.0 * B * l.in_features * l.out_features for l in (m.fc_latent_enc, m.fc_latent_dec))
    resamp = flops_step - fc - sum(v["flops"] for v in entries.values())
    assert 0 < resamp < 0.02 * flops_step                               # what is left are the <= 3 nnz / row re-sampling products
    # the per-pipe split covers exactly the matrix work, whatever the form
    for mma in ("exact", "planes3"):
        p = bench.matrix_pipe_split(m, B, "f32", mma)
        tot = p["f32_pipe"]["algorithmic_flops"] + p["bf16_pipe"]["algorithmic_flops"]
        assert tot == pytest.approx(flops_step - resamp)
        assert (p["bf16_pipe"]["algorithmic_flops"] > 0) == (mma == "planes3")
        assert p["bf16_pipe"]["instruction_flops"] == pytest.approx(6 * p["bf16_pipe"]["algorithmic_flops"])
    dt = 3e-3 * (h.sizes[0] / 6890.0) * (B / 64.0)                      # a step time of the measured order for this template / batch
    blk = bench.whole_step_block(m, B, "f32", dt, "planes3")
    assert 0 < blk["frac_mfma"] < 1 and 0 < blk["frac_hbm"] < 1
    assert bench.whole_step_block(m, B, "f32", dt, "split3")["frac_mfma"] is None
    assert bench.whole_step_block(m, B, "bf16", dt)["matrix_pipes"]["f32_pipe"]["algorithmic_flops"] == 0


def test_streaming_launches_are_priced(model):
    import bench
    m, h = model
    B = 64
    t = bench.hbm_work_table(m, B)
    assert t[("adam",)] == [28.0 * sum(p.numel() for p in m.parameters())]
    # every re-sampling step has a forward and a transposed entry; bytes = 4 B C (rows written + distinct rows read)
    for stack in (m._enc_stack, m._dec_stack):
        c = 3 if stack is m._enc_stack else m.filters_dec[0][0]
        for st in stack.steps:
            if st.kind == "conv":
                c = st.cout
                continue
            key = ("spmm", int(st.csr_fwd.rows), int(c))
            want = 4.0 * B * c * (st.csr_fwd.rows + np.unique(st.csr_fwd.col).size)
            assert any(v == pytest.approx(want) for v in t[key]), key
    assert bench.parse_tag_hbm("spmm_kernel<true, p3>", "rows=863 B=64 C=128") == ("spmm", 863, 128)
    assert bench.parse_tag_hbm("adam_kernel", "tensors=22 blocks=100") == ("adam",)
    assert bench.parse_tag_hbm("adam_kernel", "tensors=24 blocks=90 numel=237000") == ("adam", 237000)
    # the weight-gradient launch that applies Adam to its tile: six weight-sized streams instead of one
    assert bench.parse_tag_linear("linear_bwd_wgt_adam_x3_kernel", "M=64 N=256 K=55296")[1] == 4.0 * (6 * 256 * 55296 + 64 * 55296 + 64 * 256)
    assert bench.parse_tag_linear("linear_bwd_wgt_x3_kernel", "M=64 N=256 K=55296")[1] == 4.0 * (256 * 55296 + 64 * 55296 + 64 * 256)
    assert bench.parse_tag_linear("linear_fwd_x3_kernel<4>", "M=64 N=256 K=55296 split=247")[0] == 2.0 * 64 * 256 * 55296
    assert bench.parse_tag_f32("conv_p3s_kernel<2, true, 6, 4>", "R=863 B=64 K=512 N=128 grid=256x512") == ("bwd", 863, 512, 128)
    assert bench.parse_tag_f32("wgrad_stream_kernel<2, 4, 3, true, ilv>", "R=3446 B=64 K=352 N=32 grid=256 presum=4495") == ("wgt", 3446, 352, 32)

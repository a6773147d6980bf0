#!/usr/bin/env python3
"""Headline benchmark: training meshes/s of the plain spiral autoencoder at 6890 vertices,
batch 64 per GPU, fp32 (BASELINE.json configs[1]); synthetic meshes resident in HBM.

    python bench.py --gpus N --steps K --warmup W [--dtype f32|bf16]
    N > 1: one rank per GPU, RCCL gradient all-reduce.  Either launched by torch.distributed.run
    (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* in the environment), or - when WORLD_SIZE is not set -
    bench.py starts the N rank processes itself (fresh children, before this process touches the GPU),
    waits for them and exits with the first non-zero status; rank 0 prints the JSON line.

One "step" = the reference's training iteration (train_funcs.py:495-510): zero_grad, forward,
L1 + 1e-2 * edge-ratio loss, backward, [gradient all-reduce], Adam(lr 1e-3, wd 5e-5) step.
Prints ONE JSON line (rank 0) with the whole-job throughput, the roofline of the dominant HIP
kernel (HIP events recorded by the library on the launch stream) and a CPU baseline (the
oracle = the reference's formulation, on this box's host cores; a reported number, not a
target).  Nothing here reads /root/reference.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FE = [[3, 16, 32, 64, 128], [[], [], [], [], []]]
FD = [[128, 64, 32, 32, 16], [[], [], [], [], 3]]
PEAK_F32_MFMA_TFLOPS = 157.3      # MI355X_MICROARCH.md: fp32-input MFMA dense peak
PEAK_BF16_MFMA_TFLOPS = 2500.0    # MI355X_MICROARCH.md: bf16 MFMA dense peak
PEAK_HBM_GBS = 8000.0             # MI355X_MICROARCH.md: HBM3E spec peak
PAIRDIST_VALU_INSTR_PER_PAIR = 170.0    # pairdist_fwd_kernel, thresholded weight mode: DESIGN 4c; re-measured in profiles/r06_pmc_sq_semantic.txt


def f32_work_table(model, B):
    """Algorithmic FLOPs / fused-ideal bytes of every conv-family launch of one fp32 training step, keyed by what the
    library's profiler tags the launch with - (pass, rows, K, output channels) - not by kernel name, so the pricing follows
    the launches whichever instantiation the dispatch picks (DESIGN.md 'Roofline accounting'):
        ("fwd", R, S*Cin, Cout)   ("bwd", n_in, S*Cout, Cin)   ("wgt", R, S*Cin, Cout)   ("wgt+bwd", R, S*Cin, Cout)
    3-channel inputs run over zero-padded quads (K tagged 4*S); both spellings are keys of the same entry."""
    from semantichuman_amd import _lib
    out = {}
    first = True
    for stack in (model._enc_stack, model._dec_stack):
        for st in stack.steps:
            if st.kind != "conv":
                continue
            K = st.S * st.cin
            fl = 2.0 * B * st.R * K * st.cout
            # each needed input row read once + weights, output written once
            byt = 4.0 * (B * st.n_in * st.cin + B * st.R * st.cout + st.cout * K)
            e = {"flops": fl, "bytes": byt}
            out[("fwd", st.R, K, st.cout)] = e
            if st.cin == 3:
                out[("fwd", st.R, 4 * st.S, st.cout)] = e          # the same entry under its second spelling
            not_first = not (first and stack is model._enc_stack)
            thin = not_first and st.R == st.n_in and bool(_lib.load().sh_spiral_conv_bwd_wgt_thin_ok(B, st.n_in, st.S, st.cin, st.cout, 0))
            if not_first and not thin:
                # backward-data = the same kernels over the transposed table; algorithmic FLOPs are those of the R*S real
                # (row, position) pairs, not of the padded n_in*S table
                bw = dict(e)
                out[("bwd", st.n_in, st.S * st.cout, st.cin)] = bw
                if st.cout == 3:
                    out[("bwd", st.n_in, 4 * st.S, st.cin)] = bw      # 3-channel gradient rows run over zero-padded quads
            if thin:     # the 16 -> 3 channel layer: role-swapped weight gradient AND backward-data in one launch (csrc/wgrad_thin.hip)
                out[("wgt+bwd", st.R, K, st.cout)] = {"flops": 2 * fl, "bytes": 2 * byt}
            else:
                w = dict(e)
                out[("wgt", st.R, K, st.cout)] = w
                if st.cin == 3:
                    out[("wgt", st.R, 4 * st.S, st.cout)] = w
            first = False
    return out


def parse_tag_f32(name, shape):
    """Key into f32_work_table() of one profiler record (kernel name, shape tag), or None for other kernels."""
    fam = name.split("<")[0]
    targs = [t.strip() for t in name.split("<")[1].rstrip(">").split(",")] if "<" in name else []
    f = dict(kv.split("=") for kv in shape.split() if "=" in kv)
    try:
        if fam == "gather_gemm_kernel":
            return ("bwd" if targs[2] == "true" else "fwd", int(f["R"]), int(f["K"]), int(f["N"]))
        if fam == "conv_p3_kernel":                        # three-plane form, LDS-resident weight: <NT, RT, C16, BWD, NP>
            return ("bwd" if targs[3] == "true" else "fwd", int(f["R"]), int(f["K"]), int(f["N"]))
        if fam == "conv_p3r_kernel":                       # three-plane backward-data over ragged source lists: <NT, NP>
            return ("bwd", int(f["R"]), int(f["K"]), int(f["N"]))
        if fam == "conv_p3g_kernel":                       # three-plane form over grouped lists: <NT, G, BWD, NP>
            return ("bwd" if targs[2] == "true" else "fwd", int(f["R"]), int(f["K"]), int(f["N"]))
        if fam == "conv_p3s_kernel":                       # three-plane form, streamed weight: <RT, BWD, NP>
            return ("bwd" if targs[1] == "true" else "fwd", int(f["R"]), int(f["K"]), int(f["N"]))
        if fam == "conv_out3_linewise_kernel":             # forward of the <= 3-channel last layer on the VALU
            if f.get("pass") == "1":                       # first of two passes over a long spiral: the second carries the layer's key
                return None
            return ("fwd", int(f["R"]), int(f["K"]), int(f["N"]))
        if fam.startswith("gather_gemm_"):                 # direct / split3 / coalesced forms: <NT, BWD, ...>
            return ("bwd" if targs[1] == "true" else "fwd", int(f["R"]), int(f["K"]), int(f["N"]))
        if fam in ("wgrad_stream_kernel", "wgrad_kernel", "wgrad_split3_kernel", "wgrad_p3_kernel"):
            return ("wgt", int(f["R"]), int(f["K"]), int(f["N"]))
        if fam == "wgrad_thin_kernel":
            if "pass" in f and f["pass"].split("/")[0] != f["pass"].split("/")[1]:     # an earlier launch of a multi-pass layer: priced with the last
                return None
            return ("wgt+bwd" if f.get("dx") == "1" else "wgt", int(f["R"]), int(f["S"]) * int(f["Cin"]), int(f["N"]))
    except (KeyError, ValueError, IndexError):
        return None
    return None


def bf16_work_table(model, B):
    """Algorithmic FLOPs / HBM bytes of every bf16 conv-family launch, keyed by (kernel family, backward?, R, S, gathered
    channels, output channels) - the fields of the shape tag the library's profiler attaches to each launch.  Bytes = every
    tensor the launch needs read once + its output written once (fused ideal): bf16 activations (the 3-channel
    xyz tensors are fp32), bf16 weight fragments, fp32 partial slabs are not counted (they are overhead, not algorithm)."""
    from semantichuman_amd import _lib
    out = {}
    first = True
    for stack in (model._enc_stack, model._dec_stack):
        for st in stack.steps:
            if st.kind != "conv":
                continue
            K = st.S * st.cin
            fl = 2.0 * B * st.R * K * st.cout
            e_in = 4 if st.cin == 3 else 2
            e_out = 4 if st.cout == 3 else 2
            byt = B * st.n_in * st.cin * e_in + B * st.R * st.cout * e_out + 2.0 * st.cout * K
            out[("conv_bf16_kernel", False, st.R, st.S, st.cin, st.cout)] = (fl, byt)
            not_first = not (first and stack is model._enc_stack)
            # the 16 -> 3 channel layer: weight gradient and backward-data share one launch (csrc/wgrad_thin.hip)
            thin = not_first and st.R == st.n_in and bool(_lib.load().sh_spiral_conv_bwd_wgt_thin_ok(B, st.n_in, st.S, st.cin, st.cout, 1))
            if not_first and not thin:
                out[("conv_bf16_kernel", True, st.n_in, st.S, st.cout, st.cin)] = (fl, byt)
            wg = (fl, byt + 2.0 * st.cout * K)                                                                # + fp32 dW out
            out[("wgrad_bf16_kernel", None, st.R, st.S, st.cin, st.cout)] = (wg[0] + fl, wg[1] + byt) if thin else wg
            first = False
    return out


def parse_tag(name, shape):
    """(family, bwd, R, S, C, N) of a profiler record of the bf16 conv family, or None."""
    fam = name.split("<")[0]
    if fam in ("wgrad_bf16_dma_kernel", "wgrad_thin_kernel"):       # the other two forms of the bf16 weight gradient
        fam = "wgrad_bf16_kernel"
    if fam not in ("conv_bf16_kernel", "conv_bf16r_kernel", "wgrad_bf16_kernel"):
        return None
    f = dict(kv.split("=") for kv in shape.split() if "=" in kv)
    try:
        if fam == "conv_bf16r_kernel":                               # backward-data over ragged source lists: <NT>
            return ("conv_bf16_kernel", True, int(f["R"]), int(f["S"]), int(f["Cg"]), int(f["N"]))
        if fam == "conv_bf16_kernel":
            bwd = name.split("<")[1].split(",")[3].strip() == "true"
            return (fam, bwd, int(f["R"]), int(f["S"]), int(f["Cg"]), int(f["N"]))
        return (fam, None, int(f["R"]), int(f["S"]), int(f["Cin"]), int(f["N"]))
    except (KeyError, ValueError, IndexError):
        return None


def lib_sha16():
    """Identity of the kernel SOURCES: SHA-256 over csrc/*.hip, csrc/*.h, include/*.h (each group in name order) and the Makefile - the
    bytes csrc/Makefile hashes into the library as sh_build_id().  (The bytes of the .so itself are not reproducible
    across checkouts; a profile taken on these sources stays valid wherever they are rebuilt.)"""
    import glob
    import hashlib
    d = os.path.join(ROOT, "semantichuman_amd", "csrc")
    hsh = hashlib.sha256()
    for f in sorted(glob.glob(os.path.join(d, "*.hip"))) + sorted(glob.glob(os.path.join(d, "*.h"))) + \
            sorted(glob.glob(os.path.join(ROOT, "include", "*.h"))) + [os.path.join(d, "Makefile")]:
        hsh.update(open(f, "rb").read())
    return hsh.hexdigest()[:16]


def workload_tag(verts, B, dtype, mma="exact"):
    """Name of a measured workload: what a PMC profile must have been taken on to be quoted for a run."""
    return "%dv_b%d_%s%s" % (verts, B, dtype, "" if (dtype != "f32" or mma == "exact") else "_" + mma)


def measured_traffic(kernel, workload):
    """HBM bytes per launch of `kernel` from the committed rocprofv3 PMC passes (tools/pmc_traffic.py: FETCH_SIZE and WRITE_SIZE
    in separate passes, gfx950 corrections applied) - quoted only while the profile was taken on THIS workload (template,
    batch, dtype, arithmetic form), on THIS build of the kernel library (sh_build_id() of the loaded .so) and with the same
    SH_* switches; anything else yields (None, reason) instead of a silently wrong number."""
    from semantichuman_amd import _lib
    import glob
    # the newest round's profile of this workload (profiles/rNN_pmc_traffic_<workload>.json); its stamp decides whether it is quoted
    cands = sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_pmc_traffic_%s.json" % workload)))
    name = os.path.basename(cands[-1]) if cands else "r06_pmc_traffic_%s.json" % workload
    path = os.path.join(ROOT, "profiles", name)
    try:
        pmc = json.load(open(path))
    except (OSError, ValueError):
        return None, "no PMC profile for this workload (%s)" % name
    meta = pmc.get("_meta", {})
    if meta.get("workload") != workload:
        return None, "PMC profile %s is stamped for workload %r, not %r" % (name, meta.get("workload"), workload)
    if meta.get("lib_sha16") != _lib.build_id():
        return None, "PMC profile %s was taken on another build of the kernel library (%s != %s)" % (name, meta.get("lib_sha16"), _lib.build_id())
    strip = lambda e: {k: v for k, v in e.items() if k != "SH_F32_MMA"}      # noqa: E731 - the arithmetic form is part of the workload tag
    if strip(meta.get("env", {})) != strip(_lib.env_overrides()):
        return None, "PMC profile %s was taken with other SH_* switches (%s)" % (name, meta.get("env"))
    # the library's profiler names a launch by what it does; rocprofv3 by the kernel instantiation(s) that did it
    alias = {"spmm_kernel<true, p3>": ["spmm_kernel<true, true>", "spmm_p3x8_kernel"],
             "linear_bwd_wgt_x3_kernel": ["linear_bwd_wgt_dma_kernel<true, false, false, false>"],
             "linear_bwd_wgt_dma_kernel": ["linear_bwd_wgt_dma_kernel<false, false, false, false>"],
             "linear_bwd_wgt_adam_x3_kernel": ["linear_bwd_wgt_dma_kernel<true, true, false, false>"],
             "linear_bwd_wgt_adam_kernel": ["linear_bwd_wgt_dma_kernel<false, true, false, false>", "linear_bwd_wgt_dma_kernel<false, true, true, false>",
                                            "linear_bwd_wgt_dma_kernel<false, true, false, true>", "linear_bwd_wgt_dma_kernel<false, true, true, true>"]}
    import re
    m = re.match(r"linear_fwd_x3_kernel<(\d+)>$", kernel)
    if m:
        alias[kernel] = ["linear_fwd_dma_kernel<%s, true>" % m.group(1)]
    m = re.match(r"linear_fwd_dma_kernel<(\d+)>$", kernel)
    if m:
        alias[kernel] = ["linear_fwd_dma_kernel<%s, false>" % m.group(1), kernel]
    if kernel.endswith(", ilv>"):                               # the interleaved-tile instance of the streaming weight gradient
        alias[kernel] = [kernel[:-len(", ilv>")] + ", false, true>"]
    names = [k for k in alias.get(kernel, [kernel]) if k in pmc]
    if not names:
        return None, "kernel not in %s" % name
    n = sum(pmc[k]["launches_profiled"] for k in names)
    byt = sum(pmc[k]["hbm_bytes_per_launch"] * pmc[k]["launches_profiled"] for k in names) / n
    return byt, "bytes/launch, rocprofv3 PMC FETCH_SIZE x2 + WRITE_SIZE (profiles/%s%s: same workload, library build and switches)" % (
        name, "" if names == [kernel] else " as " + " + ".join(names))


def step_work(model, B, dtype):
    """Algorithmic FLOPs and fused-ideal HBM bytes of ONE training step (SURVEY 8d formulae, from the live shapes): conv layers
    forward + backward-data (not for the first layer) + weight gradient, the two latent FCs x 3, every activation read once
    and written once per pass, plus the per-step constants (weights, weight gradients, Adam's 7 passes)."""
    e = 2 if dtype == "bf16" else 4
    flops = byt = 0.0
    n_par = sum(p.numel() for p in model.parameters())
    first = True
    for stack in (model._enc_stack, model._dec_stack):
        for st in stack.steps:
            if st.kind == "conv":
                f = 2.0 * B * st.R * st.S * st.cin * st.cout
                a_in = B * st.n_in * st.cin * (4 if st.cin == 3 else e)
                a_out = B * st.R * st.cout * (4 if st.cout == 3 else e)
                flops += f * (2 if (first and stack is model._enc_stack) else 3)
                byt += 3 * (a_in + a_out)                      # forward, backward-data and weight-gradient passes touch both once
                first = False
            else:
                c = None
                for prev in stack.steps:                        # channels passing through the re-sampling step
                    if prev is st:
                        break
                    if prev.kind == "conv":
                        c = prev.cout
                c = c if c is not None else (model.filters_dec[0][0] if stack is model._dec_stack else 3)
                flops += 2.0 * 2 * st.csr.val.size * B * c
                byt += 2.0 * B * c * e * (st.csr.rows + st.csr.cols)
    for fc in (model.fc_latent_enc, model.fc_latent_dec):
        flops += 3 * 2.0 * B * fc.in_features * fc.out_features
    w_bytes = n_par * (2 if dtype == "bf16" else 4)
    byt += 2 * w_bytes + 4.0 * n_par + 7 * 4.0 * n_par + (2.0 * n_par if dtype == "bf16" else 0)   # weights fwd + dgrad, dW, Adam (+ bf16 copies)
    return flops, byt


def matrix_pipe_split(model, B, dtype, mma):
    """Algorithmic matrix FLOPs of one training step by the pipe that executes them, and the time they need at the dense peaks
    (one `frac_mfma` against the fp32 peak mis-states a step whose conv products run on the bf16 pipe at six
    instruction FLOPs per algorithmic FLOP).  planes3: forward / backward-data of every conv step the plane kernels take run on
    v_mfma_f32_16x16x32_bf16 (x6), as do (round 6) the weight gradients whose two images exist; the other weight gradients and the latent FCs on v_mfma_f32_16x16x4_f32; exact: everything on the fp32
    pipe; bf16: everything on the bf16 pipe (x1).  (split3 partitions by tile count inside the library: not split here.)"""
    from semantichuman_amd import _lib
    lib = _lib.load()
    f32 = b16 = 0.0
    first = True
    for stack in (model._enc_stack, model._dec_stack):
        for i, st in enumerate(stack.steps):
            if st.kind != "conv":
                continue
            f = 2.0 * B * st.R * st.S * st.cin * st.cout
            has_bwd = not (first and stack is model._enc_stack)
            first = False
            if dtype == "bf16":
                b16 += f * (3 if has_bwd else 2)
                continue
            p3 = mma == "planes3" and B % 16 == 0
            fwd_p3 = p3 and i > 0 and bool(lib.sh_spiral_conv_p3_ok(B, st.S, st.cin, st.cout))
            bwd_p3 = p3 and has_bwd and bool(lib.sh_spiral_conv_p3_ok(B, st.S, st.cout, st.cin))
            # round 6: the weight gradient of a step whose both images exist runs on the bf16 pipe too (csrc/wgrad_p3.hip)
            wgt_p3 = (fwd_p3 and bwd_p3 and os.environ.get("SH_P3_WGRAD", "1") != "0" and
                      bool(lib.sh_spiral_conv_bwd_wgt_p3_ok(B, st.R, st.S, st.cin, st.cout)))
            b16 += (f if fwd_p3 else 0.0) + (f if bwd_p3 else 0.0) + (f if wgt_p3 else 0.0)
            f32 += (0.0 if fwd_p3 else f) + ((0.0 if bwd_p3 else f) if has_bwd else 0.0) + (0.0 if wgt_p3 else f)
    fc = sum(3 * 2.0 * B * m.in_features * m.out_features for m in (model.fc_latent_enc, model.fc_latent_dec))
    if dtype == "bf16":
        b16 += fc
    else:
        f32 += fc
    mult = 1.0 if dtype == "bf16" else 6.0
    t_f32, t_b16 = f32 / (PEAK_F32_MFMA_TFLOPS * 1e12), mult * b16 / (PEAK_BF16_MFMA_TFLOPS * 1e12)
    return {"f32_pipe": {"algorithmic_flops": f32, "peak_tflops": PEAK_F32_MFMA_TFLOPS, "min_ms": 1e3 * t_f32},
            "bf16_pipe": {"algorithmic_flops": b16, "instruction_flops": mult * b16, "peak_tflops": PEAK_BF16_MFMA_TFLOPS, "min_ms": 1e3 * t_b16},
            "min_ms_both": 1e3 * (t_f32 + t_b16)}


def whole_step_block(model, B, dtype, dt, mma=None):
    """`whole_step`: algorithmic FLOPs / fused-ideal bytes of one step over the measured step time `dt` (seconds), the matrix work
    split by pipe; `frac_mfma` = the time the step's matrix work needs at the dense peak of the pipe it runs on / dt."""
    fl, by = step_work(model, B, dtype)
    pipes = matrix_pipe_split(model, B, dtype, mma) if mma != "split3" else None
    blk = {"flops": fl, "hbm_bytes_ideal": by, "tflops": fl / dt / 1e12, "gbps": by / dt / 1e9,
           "frac_hbm": by / dt / 1e9 / PEAK_HBM_GBS,
           "note": "algorithmic FLOPs and fused-ideal bytes of one step (SURVEY 8d) / measured ms_per_step, per GPU"}
    if pipes is not None:
        blk["matrix_pipes"] = pipes
        blk["frac_mfma"] = pipes["min_ms_both"] * 1e-3 / dt
        blk["frac_mfma_note"] = "time the matrix work needs at the dense peak of the pipe each part runs on (matrix_pipes) / step time"
    else:
        blk["frac_mfma"] = None
        blk["frac_mfma_note"] = "split3 chooses the pipe per launch inside the library: see kernel_breakdown"
    blk["tflops_vs_f32_mfma_peak"] = fl / dt / 1e12 / PEAK_F32_MFMA_TFLOPS
    return blk


def hbm_work_table(model, B):
    """Algorithmic HBM bytes of the launches that are bound by them, keyed by (family, shape tag fields):
      ("spmm", rows, C) -> [bytes per launch, ...] in the order a step issues launches with that key: every output element
          written once + every DISTINCT input row read once, 4 bytes per element (the plane images a producer also writes in
          the planes3 form are overhead of that form, not algorithm);
      ("adam",) -> 7 fp32 streams per parameter (all of them; a launch whose tag carries numel= is priced by that count instead:
          with the latent FCs' update applied inside their weight-gradient kernels the multi-tensor launch only holds the rest)."""
    import numpy as np
    out = {}

    def add(csr, C):
        if csr is None or csr.rows == 0:
            return
        distinct = int(np.unique(csr.col).size)
        out.setdefault(("spmm", int(csr.rows), int(C)), []).append(4.0 * B * C * (csr.rows + distinct))
    for stack in (model._enc_stack, model._dec_stack):                  # forward launches
        c = 3 if stack is model._enc_stack else model.filters_dec[0][0]
        for st in stack.steps:
            if st.kind == "conv":
                c = st.cout
            else:
                add(st.csr_fwd, c)
    for stack in (model._dec_stack, model._enc_stack):                  # backward launches, reverse order
        chans, c = [], (3 if stack is model._enc_stack else model.filters_dec[0][0])
        for st in stack.steps:
            chans.append(c)
            if st.kind == "conv":
                c = st.cout
        for st, cin in reversed(list(zip(stack.steps, chans))):
            if st.kind == "conv":
                add(st.tt.csr1, st.cout)
                add(st.tt.csr2, st.cout)
            else:
                add(st.csr_t, cin)
    out[("adam",)] = [7 * 4.0 * sum(p.numel() for p in model.parameters())]
    return out


def parse_tag_hbm(name, shape):
    fam = name.split("<")[0]
    f = dict(kv.split("=") for kv in shape.split() if "=" in kv)
    try:
        if fam == "spmm_kernel":
            return ("spmm", int(f["rows"]), int(f["C"]))
        if fam == "adam_kernel":
            return ("adam", int(f["numel"])) if "numel" in f else ("adam",)
    except (KeyError, ValueError):
        return None
    return None


def parse_tag_linear(name, shape):
    """(flops, bytes) of a latent-FC launch from its tag (M = batch, N x K weight), or None.  A weight-gradient launch that applies
    Adam to its tile (linear_bwd_wgt_adam*) reads the weight and its two moments and writes them back: six weight-sized streams."""
    if not name.startswith("linear_"):
        return None
    f = dict(kv.split("=") for kv in shape.split() if "=" in kv)
    try:
        M, N, K = int(f["M"]), int(f["N"]), int(f["K"])
    except (KeyError, ValueError):
        return None
    return 2.0 * M * N * K, 4.0 * ((6 if "_adam" in name else 1) * N * K + M * K + M * N)


def replayed_training(sh, h, B, dtype, dev, steps, warmup, n_data=None, seed=100, fused_update=True):
    """One more training configuration measured the way the headline is: model on hierarchy `h`, batch B, kernels in
    `dtype`; forward + loss + backward + Adam captured into one hipGraph; `steps` timed replays (batch copy-in included)
    after `warmup`.  Returns (result dict, model, init_state, data) - used by the `secondary` block."""
    from semantichuman_amd import synthetic
    torch.manual_seed(2)
    model = sh.SpiralAutoencoder(FE, FD, 256, h.sizes, h.spiral_sizes, h.spirals, h.D, h.U, dev)
    if dtype == "bf16":
        model.set_compute_dtype(torch.bfloat16)
    init_state = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
    optim = sh.optim.Adam(model.parameters(), lr=1e-3, weight_decay=5e-5)
    if fused_update:
        optim.fuse_linear_weight_gradients([model.fc_latent_enc, model.fc_latent_dec])
    ft = sh.FaceTables(h.faces, h.sizes[0] + 1, dev)
    n_data = n_data or 4 * B
    data = torch.from_numpy(synthetic.synth_batch(h.verts, n_data, seed=seed)).to(dev)
    xin = torch.empty((B, h.sizes[0] + 1, 3), dtype=torch.float32, device=dev)
    unit = torch.ones((), dtype=torch.float32, device=dev)

    def one_step():
        optim.zero_grad(set_to_none=True)
        x_hat, _ = model(xin)
        loss, _ = sh.recon_loss(x_hat, xin, ft, 1e-2)
        loss.backward(unit)
        optim.step()

    xin.copy_(data[:B])
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(3):
            one_step()
    torch.cuda.current_stream().wait_stream(s)
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        one_step()
    model.load_state_dict(init_state)
    if dtype == "bf16":                           # refresh the latent FCs' bf16 working copies eagerly (see main())
        with torch.no_grad():
            model(xin)
    for st in optim.state.values():
        for v in st.values():
            if torch.is_tensor(v):
                v.zero_()

    def step(i):
        o = (i * B) % n_data
        xin.copy_(data[o:o + B])
        graph.replay()
    for i in range(warmup):
        step(i)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        step(warmup + i)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    from semantichuman_amd import _lib
    res = {"ms_per_step": 1e3 * dt, "meshes_per_s": B / dt, "steps": steps, "warmup": warmup, "batch": B, "dtype": dtype,
           "vertices": int(h.sizes[0]), "spiral_sizes": [int(v) for v in h.spiral_sizes[:-1]], "launch": "hipGraph replay",
           "whole_step": whole_step_block(model, B, dtype, dt, _lib.get_f32_mma_mode() if dtype == "f32" else None)}
    if dtype == "f32":
        res["f32_mma"] = _lib.get_f32_mma_mode()
    res["adam"] = "latent FCs updated inside their weight-gradient kernels" if fused_update else "multi-tensor kernel"
    del graph
    optim.remove_fusion()       # the model outlives this optimizer: nothing of it may stay registered
    return res, model, init_state, data, ft


def secondary_block(sh, h, B, dev, init_state, data, test, ft, cpu_l2_mm, args):
    """BASELINE configs 3 (per-GPU shard), 4, 5 and the semantic iteration, each measured after the headline's timed region:
    20 replayed steps (graph replay, batch copy-in included) after 5 warm-up; failures are reported, not fatal."""
    from semantichuman_amd import _lib
    from semantichuman_amd.hierarchy import load_hierarchy
    out = {}
    steps, warm = 20, 5

    def leg(name, fn):
        try:
            t0 = time.perf_counter()
            out[name] = fn()
            out[name]["wall_s"] = round(time.perf_counter() - t0, 2)
        except Exception as e:      # noqa: BLE001
            out[name] = {"error": "%s: %s" % (type(e).__name__, str(e)[:300])}
        torch.cuda.synchronize()
        torch.cuda.empty_cache()

    def bf16_leg():
        res, model, init16, d16, ft16 = replayed_training(sh, h, B, "bf16", dev, steps, warm, fused_update=not args.no_fused_update)
        # roofline of its dominant kernel against HBM: per-launch HIP events of 3 eagerly launched steps
        opt = sh.optim.Adam(model.parameters(), lr=1e-3, weight_decay=5e-5)
        if not args.no_fused_update:             # the profiled steps are the timed step's launches
            opt.fuse_linear_weight_gradients([model.fc_latent_enc, model.fc_latent_dec])
        x = d16[:B]
        _lib.profile_enable(True)
        for _ in range(3):
            opt.zero_grad(set_to_none=True)
            loss, _ = sh.recon_loss(model(x)[0], x, ft16, 1e-2)
            loss.backward()
            opt.step()
        torch.cuda.synchronize()
        recs = _lib.profile_records_by_kernel()
        _lib.profile_enable(False)
        rf = roofline_bf16(recs, model, B, 3, h.sizes[0])
        res["roofline"] = rf["roofline"]
        res["hip_kernel_ms_per_step"] = rf["hip_kernel_ms_per_step"]
        opt.remove_fusion()
        del model, opt
        if cpu_l2_mm is not None:
            margs = (FE, FD, 256, h.sizes, h.spiral_sizes, h.spirals, h.D, h.U)
            l2 = hip_trajectory_l2(sh, margs, init_state, "bf16", data[:B], test, ft, 1 + args.cpu_iters, dev)
            res["matched_l2"] = {"steps": 1 + args.cpu_iters, "batch": B, "hip_mm": l2, "cpu_oracle_mm": cpu_l2_mm,
                                 "rel_diff": abs(l2 - cpu_l2_mm) / cpu_l2_mm,
                                 "note": "same trajectory as the headline's matched_l2, bf16 kernels vs the fp32 CPU oracle"}
        res["config"] = "BASELINE configs[2] per-GPU shard: batch %d, bf16 kernels, fp32 master weights / gradients / Adam" % B
        return res

    def f32_roofline_of(model, x, ftab, B4, verts, train=True):
        """Per-launch HIP events of 3 eagerly launched steps (or decodes) -> the roofline block of that workload's dominant kernel."""
        opt = sh.optim.Adam(model.parameters(), lr=1e-3, weight_decay=5e-5) if train else None
        if train and not args.no_fused_update:
            opt.fuse_linear_weight_gradients([model.fc_latent_enc, model.fc_latent_dec])
        _lib.profile_enable(True)
        for _ in range(3):
            if train:
                opt.zero_grad(set_to_none=True)
                loss, _ = sh.recon_loss(model(x)[0], x, ftab, 1e-2)
                loss.backward()
                opt.step()
            else:
                with torch.no_grad():
                    model.decode(x)
        torch.cuda.synchronize()
        recs = _lib.profile_records_by_kernel()
        _lib.profile_enable(False)
        rf = roofline_f32(recs, model, B4, 3, verts)
        out = {"roofline": rf["roofline"], "hip_kernel_ms_per_step": rf["hip_kernel_ms_per_step"], "kernel_breakdown": rf["kernel_breakdown"][:5]}
        if "roofline_matrix_family" in rf:
            out["roofline_matrix_family"] = rf["roofline_matrix_family"]
        return out

    def config4_leg():
        h4 = load_hierarchy(os.path.join(ROOT, "tests", "golden", "template27554.npz"))
        res, model, _, d4, ft4 = replayed_training(sh, h4, 32, "f32", dev, steps, warm, n_data=64, fused_update=not args.no_fused_update)
        res.update(f32_roofline_of(model, d4[:32], ft4, 32, h4.sizes[0]))
        del model
        res["config"] = "BASELINE configs[3]: 27 554 vertices, spiral length 18, batch 32, fp32"
        return res

    def decode_leg():
        sys.path.insert(0, os.path.join(ROOT, "tools"))
        import bench_decode
        r = bench_decode.run(latents=20 * 1024, batch=1024, dev=dev)
        torch.manual_seed(2)
        md = sh.SpiralAutoencoder(FE, FD, 256, h.sizes, h.spiral_sizes, h.spirals, h.D, h.U, dev)
        r.update(f32_roofline_of(md, torch.randn(1024, 256, device=dev), None, 1024, h.sizes[0], train=False))
        del md
        r["config"] = "BASELINE configs[4] on a bounded sample: 20 batches of 1024 random latents (the full run is tools/bench_decode.py: 100k)"
        r["f32_mma"] = _lib.get_f32_mma_mode()
        return r

    def semantic_leg():
        sys.path.insert(0, os.path.join(ROOT, "tools"))
        import bench_semantic
        r = bench_semantic.run(batch=16, steps=steps, graph=True, dev=dev, warmup=5, report=True)
        r["config"] = "semantic training iteration (SURVEY row f1): 3 passes x 16 meshes, all part losses, backward, Adam"
        r["f32_mma"] = _lib.get_f32_mma_mode()
        r.pop("_records", None)
        pl = r.pop("_pair_loss", None)
        r["kernel_breakdown"] = r.get("kernel_breakdown", [])[:6]
        # the dominant kernel of the iteration is the part pair-distance sweep (train_funcs.py:243-284, utils_SH.py:442-478): neither
        # HBM- nor MFMA-bound but VALU / transcendental-bound - acosf, two square roots and three IEEE divisions per vertex pair.
        # Its roof is the vector ALU's issue rate: 256 CUs x 4 SIMDs x 16 lanes per clock at 2.4 GHz; instructions per pair from the
        # PMC pass of profiles/r06_pmc_sq_semantic.txt (SQ_INSTS_VALU of the launch / its pairs)
        top = r["kernel_breakdown"][0] if r["kernel_breakdown"] else None
        if top and pl and top["kernel"].startswith("pairdist_fwd"):
            pairs = float(pl["B"]) / 3.0 * sum(n * (n - 1) for n in pl["part_sizes"])      # per launch: one pass's 16 meshes x ordered pairs of every part
            per_launch_s = 1e-6 * top["us_per_iteration"] / max(1.0, top["launches_per_iteration"])
            ipp = PAIRDIST_VALU_INSTR_PER_PAIR
            peak = 256 * 4 * 16 * 2.4e9 / 1e12
            ach = pairs * ipp / per_launch_s / 1e12
            byt, why = measured_traffic("pairdist_fwd_kernel", "semantic_6890v_b48_f32_" + _lib.get_f32_mma_mode())
            r["roofline"] = {"kernel": top["kernel"], "bound": "valu", "achieved": ach, "peak": peak, "unit": "T lane-op/s", "frac": ach / peak,
                             "pairs_per_launch": pairs, "valu_instr_per_pair": ipp, "avg_launch_ms": 1e3 * per_launch_s,
                             "traffic": byt, "traffic_note": why,
                             "note": "vertex pairs x VALU instructions per pair / launch time against the vector ALU issue peak; not an HBM or MFMA kernel"}
        return r

    def other_f32_leg(other):
        def run():
            was = _lib.get_f32_mma_mode()
            _lib.set_f32_mma_mode(other)
            try:
                res, model, _, _, _ = replayed_training(sh, h, B, "f32", dev, steps, warm, fused_update=not args.no_fused_update)
            finally:
                _lib.set_f32_mma_mode(was)
            del model
            res["f32_mma"] = other
            res["config"] = "the headline's step with another arithmetic form of the fp32 products (%s)" % other
            return res
        return run

    for other in ("exact", "split3", "planes3"):
        if other != _lib.get_f32_mma_mode():
            leg("f32_%s_step" % other, other_f32_leg(other))
    leg("bf16_step", bf16_leg)
    leg("config4_27k", config4_leg)
    leg("decode_b1024", decode_leg)
    leg("semantic_iteration", semantic_leg)
    return out


def hip_trajectory_l2(sh, model_args, init_state, dtype, xg, test, ft, n_steps, dev):
    """Held-out per-vertex L2 (mm) after `n_steps` training steps on the batch `xg` from `init_state` (Adam from zero
    moments) on the HIP path - the trajectory the CPU oracle runs for `cpu_baseline`."""
    m2 = sh.SpiralAutoencoder(*model_args, dev)
    if dtype == "bf16":
        m2.set_compute_dtype(torch.bfloat16)
    m2.load_state_dict(init_state)
    o2 = sh.optim.Adam(m2.parameters(), lr=1e-3, weight_decay=5e-5)
    for _ in range(n_steps):
        o2.zero_grad(set_to_none=True)
        l2loss, _ = sh.recon_loss(m2(xg)[0], xg, ft, 1e-2)
        l2loss.backward()
        o2.step()
    with torch.no_grad():
        return float(sh.vertex_l2_mm(m2(test)[0], test).item())


def roofline_bf16(recs, model, B, nprof, verts):
    """`roofline` / `kernel_families` / `kernel_breakdown` of a bf16 run from the library's per-launch HIP-event records of
    `nprof` eagerly launched steps."""
    from semantichuman_amd import _lib
    result = {}
    work = bf16_work_table(model, B)
    hbm = hbm_work_table(model, B)                      # fp32 bytes; the bf16 re-sampling launches move half of them
    seen = {}
    agg = {}
    for name, shape, ms in recs:
        a = agg.setdefault(name, {"n": 0, "ms": 0.0, "flops": 0.0, "bytes": 0.0, "matched": 0})
        a["n"] += 1; a["ms"] += ms
        key = parse_tag(name, shape)
        if key in work:
            a["flops"] += work[key][0]; a["bytes"] += work[key][1]; a["matched"] += 1
        elif name.startswith("spmm_bf16_kernel"):
            f = dict(kv.split("=") for kv in shape.split() if "=" in kv)
            hk = ("spmm", int(f.get("rows", -1)), int(f.get("C", -1)))
            if hk in hbm:
                k = seen.get(hk, 0)
                seen[hk] = k + 1
                a["bytes"] += 0.5 * hbm[hk][k % len(hbm[hk])]; a["matched"] += 1
    kernels = []
    for name, a in sorted(agg.items(), key=lambda kv: -kv[1]["ms"]):
        e = {"kernel": name, "launches_per_step": a["n"] / nprof, "avg_ms": a["ms"] / a["n"], "ms_per_step": a["ms"] / nprof}
        if a["matched"] == a["n"] and a["n"]:
            e["gbps"] = a["bytes"] / (a["ms"] * 1e-3) / 1e9
            if a["flops"] > 0:
                e["tflops"] = a["flops"] / (a["ms"] * 1e-3) / 1e12
        kernels.append(e)
    fam = {}
    for k in kernels:
        f = fam.setdefault(k["kernel"].split("<")[0].split("|")[0], 0.0)
        fam[k["kernel"].split("<")[0].split("|")[0]] = f + k["ms_per_step"]
    conv = [k for k in kernels if "tflops" in k]
    dom = conv[0] if conv else kernels[0]
    a = agg[dom["kernel"]]
    # bf16: ridge of the chip ~ 2.5 PF / 8 TB/s = 300 FLOP/B, these layers have 30-250 FLOP/B -> HBM roof
    traffic, traffic_note = measured_traffic(dom["kernel"], workload_tag(verts, B, "bf16"))
    result["roofline"] = {"bound": "hbm", "kernel": dom["kernel"], "achieved": dom.get("gbps"), "peak": PEAK_HBM_GBS, "unit": "GB/s",
                          "frac": (dom["gbps"] / PEAK_HBM_GBS) if "gbps" in dom else None, "traffic": traffic, "traffic_unit": traffic_note,
                          "avg_launch_ms": dom["avg_ms"], "launches_per_step": dom["launches_per_step"],
                          "algorithmic_bytes_per_launch": a["bytes"] / a["n"] if a["n"] else None,
                          "flops_per_launch": a["flops"] / a["n"] if a["n"] else None,
                          "mfma_tflops": dom.get("tflops"), "mfma_peak_tflops": 2500.0}
    # `roofline` describes the kernel with the largest share of the step, whatever its family (in bf16 that is the optimizer:
    # 7 fp32 streams + the bf16 working copy per parameter, priced below); the most expensive conv-family kernel keeps its own
    # line, `roofline_conv_family`
    top = kernels[0]
    if top["kernel"] != dom["kernel"]:
        result["roofline_conv_family"] = result["roofline"]
        line = {"bound": "hbm", "kernel": top["kernel"], "achieved": None, "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": None,
                "traffic": None, "avg_launch_ms": top["avg_ms"], "launches_per_step": top["launches_per_step"],
                "ms_per_step": top["ms_per_step"]}
        if "gbps" in top:                                     # a streaming kernel priced by hbm_work_table (re-sampling)
            ta = agg[top["kernel"]]
            tr, tn = measured_traffic(top["kernel"], workload_tag(verts, B, "bf16"))
            line.update(achieved=top["gbps"], frac=top["gbps"] / PEAK_HBM_GBS, algorithmic_bytes_per_launch=ta["bytes"] / ta["n"], traffic=tr,
                        traffic_unit=tn, note="algorithmic bytes (outputs written once + distinct inputs read once, bf16) of its launches / their HIP-event time")
        if top["kernel"].startswith("linear_bwd_wgt_adam"):
            # the two latent FCs' weight gradients with Adam applied to the tile: weight, exp_avg, exp_avg_sq read and written
            # (24 B / weight), the bf16 working copy written (2 B), dy and x read
            byt = 0.0
            for name, shape, _ in recs:
                if name == top["kernel"]:
                    f = dict(kv.split("=") for kv in shape.split() if "=" in kv)
                    M, N, K = int(f["M"]), int(f["N"]), int(f["K"])
                    byt += 26.0 * N * K + (2.0 if f.get("dy") == "bf16" else 4.0) * M * N + (2.0 if f.get("x") == "bf16" else 4.0) * M * K
            byt /= nprof
            tr, tn = measured_traffic(top["kernel"], workload_tag(verts, B, "bf16"))
            line.update(achieved=byt / (top["ms_per_step"] * 1e-3) / 1e9, algorithmic_bytes_per_step=byt, traffic=tr, traffic_unit=tn,
                        note="latent-FC weight gradients with the Adam update of the same tile inside (sh_linear_bwd_wgt_adam): "
                             "3 fp32 streams read + written, the bf16 working copy written; no gradient is materialised")
            line["frac"] = line["achieved"] / PEAK_HBM_GBS
        if top["kernel"].startswith("adam_kernel"):
            n_par = sum(p.numel() for p in model.parameters())
            byt = (7 * 4.0 + 2.0) * n_par                     # p, g, m, v read; p, m, v written; bf16 copy written
            line.update(achieved=byt / (top["ms_per_step"] * 1e-3) / 1e9, algorithmic_bytes_per_step=byt,
                        note="multi-tensor Adam over %.2fM parameters: 7 fp32 streams + the bf16 working copy" % (n_par / 1e6))
            line["frac"] = line["achieved"] / PEAK_HBM_GBS
        result["roofline"] = line
    result["kernel_families"] = {n: {"ms_per_step": v} for n, v in sorted(fam.items(), key=lambda kv: -kv[1])[:10]}
    result["kernel_breakdown"] = kernels[:10]
    result["hip_kernel_ms_per_step"] = sum(k["ms_per_step"] for k in kernels)
    return result


def roofline_f32(recs, model, B, nprof, verts):
    """`roofline` / `roofline_matrix_family` / `kernel_families` / `kernel_breakdown` of an fp32 run from the library's per-launch
    HIP-event records of `nprof` eagerly launched steps.

    `roofline` describes the kernel NAME with the largest share of the step, whatever its family: a matrix
    kernel against the dense peak of the pipe it runs on (fp32 MFMA 157.3 TF; a bf16x3 kernel - conv_p3*, *split3* - executes
    six bf16 MFMA FLOPs per algorithmic fp32 FLOP: dense bf16 peak / 6 = 416.7 TF), a streaming kernel (re-sampling, Adam)
    against HBM from its algorithmic bytes (hbm_work_table).  `roofline_matrix_family` keeps the most expensive kernel that
    does matrix work when that is another one."""
    from semantichuman_amd import _lib
    result = {}
    work = f32_work_table(model, B)
    hbm = hbm_work_table(model, B)
    seen = {}
    agg = {}
    for name, shape, ms in recs:
        a = agg.setdefault(name, {"n": 0, "ms": 0.0, "flops": 0.0, "bytes": 0.0, "matched": 0, "hbm": 0})
        a["n"] += 1; a["ms"] += ms
        key = parse_tag_f32(name, shape)
        lin = parse_tag_linear(name, shape)
        hk = parse_tag_hbm(name, shape)
        if key in work:
            a["flops"] += work[key]["flops"]; a["bytes"] += work[key]["bytes"]; a["matched"] += 1
        elif lin is not None:
            a["flops"] += lin[0]; a["bytes"] += lin[1]; a["matched"] += 1
        elif hk is not None and hk[0] == "adam" and len(hk) == 2:
            a["bytes"] += 28.0 * hk[1]; a["matched"] += 1; a["hbm"] += 1
        elif hk in hbm:
            lst = hbm[hk]
            k = seen.get(hk, 0)
            seen[hk] = k + 1
            a["bytes"] += lst[k % len(lst)]; a["matched"] += 1; a["hbm"] += 1
    PEAK_X3 = PEAK_BF16_MFMA_TFLOPS / 6.0

    def is_x3(name):
        return "split3" in name or name.startswith(("conv_p3", "wgrad_p3")) or "_x3_" in name
    kernels = []
    for name, a in sorted(agg.items(), key=lambda kv: -kv[1]["ms"]):
        e = {"kernel": name, "launches_per_step": a["n"] / nprof, "avg_ms": a["ms"] / a["n"], "ms_per_step": a["ms"] / nprof}
        if a["matched"] == a["n"] and a["n"]:
            if a["flops"] > 0:
                peak = PEAK_X3 if is_x3(name) else PEAK_F32_MFMA_TFLOPS
                e["tflops"] = a["flops"] / (a["ms"] * 1e-3) / 1e12
                e["peak_tflops"] = peak
                e["frac"] = e["tflops"] / peak
            e["gbps_algorithmic"] = a["bytes"] / (a["ms"] * 1e-3) / 1e9
            if "frac" not in e:                                   # streaming kernels: against HBM
                e["peak_gbps"] = PEAK_HBM_GBS
                e["frac"] = e["gbps_algorithmic"] / PEAK_HBM_GBS
        kernels.append(e)
    fam = {}
    for k in kernels:
        f = fam.setdefault(k["kernel"].split("<")[0].split("|")[0], [0.0, 0.0])
        f[0] += k["ms_per_step"]
        f[1] += agg[k["kernel"]]["flops"] / nprof
    mma = _lib.get_f32_mma_mode()

    def line(k):
        a = agg[k["kernel"]]
        traffic, traffic_note = measured_traffic(k["kernel"], workload_tag(verts, B, "f32", mma))
        base = {"kernel": k["kernel"], "traffic": traffic, "traffic_unit": traffic_note,
                # what the launch really moved per second (PMC bytes / this run's HIP-event time): how close the kernel is to what
                # the memory system delivers, next to `achieved`, which only counts the bytes the algorithm needs
                "traffic_gbps": (traffic / (k["avg_ms"] * 1e-3) / 1e9) if traffic else None, "avg_launch_ms": k["avg_ms"],
                "launches_per_step": k["launches_per_step"], "ms_per_step": k["ms_per_step"], "f32_mma": mma,
                "algorithmic_bytes_per_launch": (a["bytes"] / a["n"]) if a["matched"] == a["n"] else None}
        # FLOPs / byte of the launch decides the roof: below the ridge of the pipe it runs on it is an HBM line
        if "tflops" in k and not k["kernel"].startswith("linear_"):
            x3 = is_x3(k["kernel"])
            base.update({"bound": "mfma", "achieved": k["tflops"], "peak": k["peak_tflops"], "unit": "TFLOP/s", "frac": k["frac"],
                         "peak_note": ("dense bf16 MFMA peak %.0f TF / 6 partial products per fp32 product" % PEAK_BF16_MFMA_TFLOPS) if x3
                                      else "fp32-input MFMA dense peak",
                         "frac_of_f32_mfma_peak": k["tflops"] / PEAK_F32_MFMA_TFLOPS, "flops_per_launch": a["flops"] / a["n"]})
        elif "gbps_algorithmic" in k:
            base.update({"bound": "hbm", "achieved": k["gbps_algorithmic"], "peak": PEAK_HBM_GBS, "unit": "GB/s",
                         "frac": k["gbps_algorithmic"] / PEAK_HBM_GBS,
                         "note": "algorithmic bytes (outputs written once + distinct inputs read once, fp32) of its launches / their HIP-event time" +
                                 ("; these launches write the 6-byte plane image of their rows (and the 4-byte fp32 row only where a later kernel still reads it), "
                                  "which the fp32 count does not include: traffic / algorithmic ~ 2 is the data format, not re-reads" if "p3" in k["kernel"] else "")})
            if "tflops" in k:
                base.update({"mfma_tflops": k["tflops"], "mfma_frac_of_f32_peak": k["tflops"] / PEAK_F32_MFMA_TFLOPS})
        else:
            base.update({"bound": "hbm", "achieved": None, "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": None})
        return base
    result["roofline"] = line(kernels[0])
    mat = [k for k in kernels if "tflops" in k and not k["kernel"].startswith("linear_")]
    if mat and mat[0]["kernel"] != kernels[0]["kernel"]:
        result["roofline_matrix_family"] = line(mat[0])
    # matrix FLOPs of the step by the pipe that executed them, from the records themselves
    f32p = sum(a["flops"] for n, a in agg.items() if not is_x3(n)) / nprof
    x3p = sum(a["flops"] for n, a in agg.items() if is_x3(n)) / nprof
    result["matrix_pipes_measured"] = {
        "f32_pipe": {"algorithmic_flops_per_step": f32p, "kernel_ms_per_step": sum(a["ms"] for n, a in agg.items() if a["flops"] > 0 and not is_x3(n)) / nprof},
        "bf16_pipe_x6": {"algorithmic_flops_per_step": x3p, "kernel_ms_per_step": sum(a["ms"] for n, a in agg.items() if a["flops"] > 0 and is_x3(n)) / nprof}}
    result["kernel_families"] = {n: {"ms_per_step": v[0], "tflops": (v[1] / (v[0] * 1e-3) / 1e12) if v[1] else None}
                                 for n, v in sorted(fam.items(), key=lambda kv: -kv[1][0])[:8]}
    result["kernel_breakdown"] = kernels[:8]
    result["hip_kernel_ms_per_step"] = sum(k["ms_per_step"] for k in kernels)
    return result


def collective_block(reducer, world, rank, dev, backend):
    """World size, the device every rank runs on, and what the step's gradient messages cost when nothing else runs: each
    message size of the reducer all-reduced 5 times (after 2 warm-ups) between HIP events; bus bandwidth by the ring
    formula 2 (n - 1) / n x bytes / time.  Every rank calls this (collectives inside)."""
    on_gpu = torch.device(dev).type == "cuda"          # (the dry run times the same block over gloo with host tensors)
    ids = [None] * dist.get_world_size()
    dist.all_gather_object(ids, {"rank": rank, "device": torch.cuda.current_device() if on_gpu else "cpu",
                                 "name": torch.cuda.get_device_name() if on_gpu else "host", "pid": os.getpid()})
    msgs = []
    for b in reducer.buckets:
        nbytes = b.numel * (2 if (b.inplace and reducer.large_dtype is not None and reducer.large_dtype.itemsize == 2) else 4)
        buf = torch.zeros(nbytes // 4, dtype=torch.float32, device=dev)
        for _ in range(2):
            dist.all_reduce(buf)
        if on_gpu:
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5):
                dist.all_reduce(buf)
            e1.record()
            torch.cuda.synchronize()
            ms1 = e0.elapsed_time(e1) / 5.0
        else:
            t0 = time.perf_counter()
            for _ in range(5):
                dist.all_reduce(buf)
            ms1 = 1e3 * (time.perf_counter() - t0) / 5.0
        t = torch.tensor([ms1], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        ms = float(t.item())
        n = dist.get_world_size()
        msgs.append({"bytes": nbytes, "ms": ms, "alg_gbps": nbytes / ms / 1e6, "bus_gbps": 2.0 * (n - 1) / n * nbytes / ms / 1e6})
        del buf
    return {"world_size": dist.get_world_size(), "backend": backend, "ranks": ids, "messages": msgs,
            "averaging": "ncclAvg in the collective" if reducer.avg else "sum of pre-scaled gradients",
            "overlap": "large messages launched from gradient hooks" if reducer.overlap else "after backward"}


def _free_port():
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def visible_gpus():
    """Number of GPUs a rank process of this job can open, WITHOUT touching the HIP runtime (the launcher parent must stay
    GPU-free: it spawns the ranks): KFD topology nodes with SIMDs (/sys/class/kfd/kfd/topology/nodes/*/properties,
    simd_count > 0; CPU nodes have 0), narrowed by HIP_/ROCR_/CUDA_VISIBLE_DEVICES when set.  None when sysfs has no KFD
    topology (then the rank processes fail with their own message)."""
    import glob
    n = 0
    nodes = glob.glob("/sys/class/kfd/kfd/topology/nodes/*/properties")
    if not nodes:
        return None
    for f in nodes:
        try:
            props = dict(l.split(None, 1) for l in open(f).read().splitlines() if " " in l)
        except OSError:
            continue
        if int(props.get("simd_count", "0").strip() or 0) > 0:
            n += 1
    for var in ("HIP_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        v = os.environ.get(var)
        if v is not None:
            n = min(n, len([x for x in v.split(",") if x.strip() != ""]))
    return n


def self_launch(n, argv):
    """`python bench.py --gpus N` without a launcher: start N fresh rank processes of this script (one per GPU) with the
    torch.distributed.run environment, wait, and return the first non-zero exit status.  Called BEFORE anything in this
    process initialises the GPU (children are started with subprocess, never exec'd from a process that touched HIP)."""
    import subprocess
    if os.environ.get("SH_BENCH_DRYRUN", "0") == "0" and os.environ.get("SH_BENCH_BACKEND", "nccl") == "nccl":
        have = visible_gpus()                                  # sysfs / environment only: no HIP call in this process
        if have is not None and have < n:
            print("bench.py: --gpus %d but this node has %d visible GPU(s)" % (n, have), file=sys.stderr)
            return 2
    env = dict(os.environ)
    env.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n))
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")          # dmabuf IPC only on this pool (RCCL needs it)
    env.setdefault("OMP_NUM_THREADS", "4")
    job_dir = None
    if n > 1:                                                  # the supervisors' marker files: a fresh directory per job
        import tempfile
        job_dir = tempfile.mkdtemp(prefix="sh_bench_")
        env["SH_BENCH_JOB_DIR"] = job_dir
    import signal

    def on_signal(signum, _frame):                             # `timeout N python bench.py --gpus N`: leave through the finally below
        raise SystemExit(128 + signum)
    for sig in (signal.SIGTERM, signal.SIGINT):
        signal.signal(sig, on_signal)
    procs = []
    rc = 0
    try:
        for r in range(n):
            e = dict(env, RANK=str(r), LOCAL_RANK=str(r))
            procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=e))
        while procs:
            for p in list(procs):
                st = p.poll()
                if st is None:
                    continue
                procs.remove(p)
                if st != 0 and rc == 0:
                    rc = st
                    for q in procs:                            # a dead rank leaves the others blocked in a collective
                        q.terminate()                          # SIGTERM: a supervisor ends its GPU child before it exits
            time.sleep(0.05)
    finally:
        _stop_processes(procs)
        if job_dir:
            import shutil
            shutil.rmtree(job_dir, ignore_errors=True)
    return rc


def _stop_processes(procs, grace=15.0):
    """SIGTERM, a grace period, only then SIGKILL: the processes are SUPERVISORS whose SIGTERM handler ends their own GPU
    child (a bare kill() would orphan a rank that is blocked in a collective and holds the GPU)."""
    for p in procs:
        if p.poll() is None:
            p.terminate()
    t0 = time.time()
    for p in procs:
        try:
            p.wait(max(0.1, grace - (time.time() - t0)))
        except Exception:      # noqa: BLE001 - subprocess.TimeoutExpired
            p.kill()
            p.wait()


# ---- N > 1: every rank runs under a GPU-free supervisor that can start the rank's work again in a more conservative mode
# ("order of fallbacks").  The multi-rank step is one hipGraph with the RCCL all-reduces inside; a capture
# or replay that fails with collectives recorded cannot be survived by the process (the process group's watchdog aborts
# it), and under torch.distributed.run a dead rank ends the job.  So the process the launcher started never touches the
# GPU: it starts the real rank as a CHILD (subprocess: never an exec from a GPU process), and when an attempt fails on any
# rank, all supervisors start FRESH children for the next attempt on a fresh rendezvous port.
# Round 5: the eager mode goes first.  In a world of one rank on RCCL the replayed graph is no faster than the eager step (1.724
# against 1.708 ms, tools/exp/r05_dp_modes.sh: the host is not what paces a 1.7-ms step), and a graph with RCCL nodes is the only
# mode here that has never run with more than one rank and could hang rather than fail.  SH_BENCH_DP_GRAPH=1 puts it first again.
ATTEMPTS = (
    ("graph", {"SH_BENCH_DP_GRAPH": "1"}),                                   # the step as one hipGraph, RCCL inside (opt-in)
    ("eager", {"SH_BENCH_DP_GRAPH": "0"}),                                   # the same step, launched eagerly
    ("eager-safe", {"SH_BENCH_DP_GRAPH": "0", "SH_BENCH_DP_SAFE": "1"}),     # + lazy communicator, no stream priority, all-reduce
)                                                                            #   (sum of pre-scaled gradients) after backward


def _touch(path, text=""):
    tmp = path + ".tmp%d" % os.getpid()
    with open(tmp, "w") as f:
        f.write(text)
    os.replace(tmp, path)                                  # atomic: a reader never sees a half-written file


def mark_phase(name):
    """A rank tells its supervisor how far it is (SH_BENCH_PHASE_FILE, set by the supervisor): 'rendezvous' -> 'built' -> 'warm'
    -> 'timed'.  Everything after the rendezvous is seconds of work, so the supervisor treats a phase that lasts longer than
    SH_BENCH_PHASE_TIMEOUT as a hang (a collective that never completes) without waiting for the whole-attempt limit, which has
    to cover a cold `import torch` on a fresh box."""
    path = os.environ.get("SH_BENCH_PHASE_FILE")
    if path:
        try:
            _touch(path, name)
        except OSError:
            pass


def supervise_rank(argv):
    """The launcher's rank process (RANK / WORLD_SIZE set, SH_BENCH_ATTEMPT not): runs the attempts of ATTEMPTS in order
    until one succeeds on rank 0; returns the exit status.  Coordination between the supervisors of one node is a
    directory of marker files (one node by contract: `--nnodes=1`):
        attempt<k>.port      rank 0's fresh rendezvous port for attempt k > 0
        attempt<k>.rankfail.<r>   rank r's child exited non-zero
        attempt<k>.ok / .failed   rank 0's verdict (it owns the JSON line); everybody waits for it
    Rank 0's child writes the JSON line to a pipe; it is forwarded only when the attempt is the one that counts."""
    import subprocess
    import tempfile
    import threading
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ["WORLD_SIZE"])
    port0 = os.environ.get("MASTER_PORT", "29500")
    job = os.environ.get("SH_BENCH_JOB_DIR")                    # `python bench.py --gpus N`: made (mkdtemp) and removed by the launcher
    own_job = job is None
    if own_job:
        # torch.distributed.run: every supervisor of the job has the same parent (the agent).  Port 29500 is the default and PIDs
        # are recycled, so the name also carries the parent's START TIME (/proc/<pid>/stat field 22): a later job can never
        # inherit this job's attempt<k>.ok / .failed markers.  Rank 0 removes the directory when everybody has read the verdict.
        try:
            start = open("/proc/%d/stat" % os.getppid()).read().rsplit(")", 1)[1].split()[19]
        except (OSError, IndexError):
            start = "0"
        job = os.path.join(tempfile.gettempdir(), "sh_bench_%s_%d_%s_%s" % (port0, os.getppid(), start, os.environ.get("TORCHELASTIC_RESTART_COUNT", "0")))
    os.makedirs(job, exist_ok=True)
    child = {"p": None}

    def stop_child():
        p = child["p"]
        if p is not None and p.poll() is None:
            p.terminate()
            try:
                p.wait(10)
            except subprocess.TimeoutExpired:
                p.kill()
                p.wait()

    def on_signal(signum, _frame):                             # SIGTERM / SIGINT from the launcher: leave through the finally below
        raise SystemExit(128 + signum)
    import signal
    for sig in (signal.SIGTERM, signal.SIGINT):
        signal.signal(sig, on_signal)
    try:
        return _supervise_attempts(argv, rank, world, job, child)
    finally:
        stop_child()                                           # never leave a rank behind (blocked in a collective, holding the GPU)
        if own_job:
            if rank == 0:
                t0 = time.time()                                # the other supervisors read the verdict, then say so
                while time.time() - t0 < 10 and sum(1 for f in os.listdir(job) if f.startswith("done.")) < world - 1:
                    time.sleep(0.05)
                import shutil
                shutil.rmtree(job, ignore_errors=True)
            else:
                try:
                    _touch(os.path.join(job, "done.%d" % rank))
                except OSError:
                    pass


def _supervise_attempts(argv, rank, world, job, child):
    import subprocess
    import threading
    timeout_all = float(os.environ.get("SH_BENCH_ATTEMPT_TIMEOUT", "900"))
    # the graph attempt (opt-in: the step as ONE hipGraph with the RCCL all-reduces inside) has never run with more than one rank
    # on hardware: if it hangs instead of failing, give up on it sooner - a whole N > 1 attempt is ~1-3 minutes (import, build, 60 steps)
    timeout_graph = float(os.environ.get("SH_BENCH_GRAPH_ATTEMPT_TIMEOUT", str(min(timeout_all, 420.0))))
    grace = float(os.environ.get("SH_BENCH_FAIL_GRACE", "20"))
    # ... and once a rank is past the rendezvous (mark_phase), no single phase of it is more than seconds of work
    phase_timeout = float(os.environ.get("SH_BENCH_PHASE_TIMEOUT", "150"))
    attempts = [a for a in ATTEMPTS if a[0] != "graph" or os.environ.get("SH_BENCH_DP_GRAPH", "0") == "1"]
    history, rc = [], 1

    def wait_for(paths, limit):
        t0 = time.time()
        while time.time() - t0 < limit:
            for q in paths:
                if os.path.exists(q):
                    return q
            time.sleep(0.05)
        return None

    for k, (name, extra) in enumerate(attempts):
        timeout = timeout_graph if name == "graph" else timeout_all
        base = os.path.join(job, "attempt%d" % k)
        env = dict(os.environ, SH_BENCH_ATTEMPT=str(k), SH_BENCH_ATTEMPT_NAME=name, SH_BENCH_ATTEMPT_HISTORY="; ".join(history), **extra)
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        env["SH_BENCH_PHASE_FILE"] = phase_file = base + ".phase.%d" % rank
        if k > 0:                                           # fresh rendezvous: the first one's store holds the dead attempt's keys
            if rank == 0:
                _touch(base + ".port", str(_free_port()))
            if not wait_for([base + ".port"], 120):
                print("bench.py[supervisor %d]: no rendezvous port for attempt %d" % (rank, k), file=sys.stderr)
                return rc
            env["MASTER_PORT"] = open(base + ".port").read().strip()
            env.pop("TORCHELASTIC_USE_AGENT_STORE", None)   # rank 0's child hosts the store itself
        p = subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=env,
                             stdout=subprocess.PIPE if rank == 0 else None, text=True if rank == 0 else None)
        child["p"] = p                                          # the caller's finally / signal handler ends it
        lines = []
        reader = None
        if rank == 0:
            reader = threading.Thread(target=lambda: lines.extend(p.stdout), daemon=True)
            reader.start()
        t0, first_fail, killed = time.time(), None, None
        while p.poll() is None:
            if os.path.exists(base + ".failed"):
                killed = "attempt declared failed by rank 0"
            elif time.time() - t0 > timeout:
                killed = "no result after %.0f s" % timeout
            elif _phase_age(phase_file) > phase_timeout:
                killed = "no progress for %.0f s after phase '%s'" % (phase_timeout, open(phase_file).read().strip())
            elif rank == 0:
                if first_fail is None and any(f.startswith("attempt%d.rankfail." % k) for f in os.listdir(job)):
                    first_fail = time.time()
                if first_fail is not None and time.time() - first_fail > grace:
                    killed = "another rank failed"           # the survivors are blocked in a collective
            if killed:
                p.terminate()
                try:
                    p.wait(10)
                except subprocess.TimeoutExpired:
                    p.kill()
                    p.wait()
                break
            time.sleep(0.05)
        st = p.returncode
        if reader:
            reader.join(5)
        if rank == 0:
            ok = st == 0 and not killed and any(l.startswith("{") for l in lines)
            _touch(base + (".ok" if ok else ".failed"), killed or "rc=%s" % st)
        else:
            if (st != 0 and not killed) or (killed and not os.path.exists(base + ".failed")):
                _touch(base + ".rankfail.%d" % rank, killed or "rc=%s" % st)     # (also a hang this supervisor ended itself)
            verdict = wait_for([base + ".ok", base + ".failed"], timeout + grace + 30)
            ok = verdict is not None and verdict.endswith(".ok")
            if verdict is None:
                print("bench.py[supervisor %d]: no verdict from rank 0 for attempt %d" % (rank, k), file=sys.stderr)
                return st or 1
        if ok:
            if rank == 0:
                sys.stdout.write("".join(lines))
                sys.stdout.flush()
            return 0
        rc = st if st not in (0, None) else 1
        history.append("%s: %s" % (name, killed or "rc=%s" % st))
        print("bench.py[supervisor %d]: attempt %d (%s) failed: %s" % (rank, k, name, history[-1]), file=sys.stderr)
    return rc


def _phase_age(path):
    try:
        if open(path).read().strip() == "timed":               # the collectives are behind it; what follows (profiles, report) is host work
            return -1.0
        return time.time() - os.stat(path).st_mtime
    except OSError:
        return -1.0                                            # not past the rendezvous yet: only the whole-attempt limit applies


def attempt_note():
    """What config.launch says about the supervisor's attempts (empty for the first one)."""
    h = os.environ.get("SH_BENCH_ATTEMPT_HISTORY", "")
    return " [attempt %s '%s' after failed: %s]" % (os.environ.get("SH_BENCH_ATTEMPT"), os.environ.get("SH_BENCH_ATTEMPT_NAME"), h) if h else ""


def injected_rank_failure(rank):
    """Test hook: SH_BENCH_TEST_RANK_FAIL='<attempt>:<rank>' makes that rank of that attempt exit with status 7 right after
    the rendezvous (tests/test_bench_launch.py proves the retry with it)."""
    spec = os.environ.get("SH_BENCH_TEST_RANK_FAIL")
    if spec and spec in ("%s:%d" % (os.environ.get("SH_BENCH_ATTEMPT", "0"), rank), "*:%d" % rank):
        print("bench.py[rank %d]: injected failure (SH_BENCH_TEST_RANK_FAIL=%s)" % (rank, spec), file=sys.stderr)
        sys.stdout.flush()
        os._exit(7)
    hang = os.environ.get("SH_BENCH_TEST_RANK_HANG")     # '<attempt>:<rank>' or '*:*': that rank never gets past the rendezvous
    if hang and hang in ("%s:%d" % (os.environ.get("SH_BENCH_ATTEMPT", "0"), rank), "*:%d" % rank, "*:*"):
        time.sleep(3600)


def dry_run(args, world, rank):
    """SH_BENCH_DRYRUN=1: the launch / rendezvous / timing / reporting skeleton of this script over gloo with no GPU work
    (the CPU container has no device): rendezvous, warm-up, K barrier-bracketed empty steps, MAX over ranks, one JSON
    line from rank 0.  tests/test_bench_launch.py runs it for N = 2 through the self-launcher."""
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("gloo")
        mark_phase("rendezvous")
        injected_rank_failure(rank)
        dist.barrier()
    mark_phase("warm")
    t0 = time.perf_counter()
    for _ in range(args.steps):
        pass
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    coll = None
    if world > 1:                                       # the `collective` block of an N > 1 line, exercised end to end over gloo
        import types
        stand_in = types.SimpleNamespace(buckets=[types.SimpleNamespace(numel=1 << 16, inplace=True), types.SimpleNamespace(numel=1 << 12, inplace=False)],
                                         large_dtype=None, avg=False, overlap=True)
        coll = collective_block(stand_in, world, rank, torch.device("cpu"), "gloo")
    if rank == 0:
        print(json.dumps({"collective": coll, "metric": "training meshes/sec at 6890 verts, batch=%d" % args.batch, "value": None, "unit": "meshes/s",
                          "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / max(1, args.steps),
                          "dry_run": True, "config": {"global_batch": world * args.batch, "parallelism": "dp%d" % world,
                                                     "launch": "dry run (%s)" % os.environ.get("SH_BENCH_ATTEMPT_NAME", "single process") + attempt_note()}}))
    if dist.is_initialized():
        dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--batch", type=int, default=64, help="per-GPU batch (BASELINE: 64)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-secondary", action="store_true", help="skip the `secondary` block (the other BASELINE configurations, "
                    "each timed as 20 replayed steps after the headline's timed region)")
    ap.add_argument("--no-graph", action="store_true", help="eager launches instead of a captured hipGraph")
    ap.add_argument("--adam", choices=["hip", "torch"], default="hip", help="library Adam kernel, or torch's fused capturable Adam")
    ap.add_argument("--no-fused-update", action="store_true", help="one GPU: write the latent FCs' weight gradients and let the multi-tensor Adam "
                    "read them back, instead of applying the update inside the weight-gradient kernel (the default; same bits)")
    ap.add_argument("--adam-overlap", action="store_true", help="update the big parameters on a side stream underneath backward (measured: no gain, the GPU is already saturated)")
    ap.add_argument("--grad-comm", choices=["fp32", "bf16"], default="fp32",
                    help="N > 1: type of the two large gradient messages (bf16 halves the xGMI bytes; fp32 is the measured default)")
    ap.add_argument("--shard-optimizer", dest="shard_optimizer", action="store_true", default=None,
                    help="N > 1: reduce-scatter the two latent FC gradients, update 1/N of each FC per rank, all-gather the weights "
                         "(same xGMI bytes as the all-reduce, 1/N of Adam's HBM traffic and state); weights bitwise those of the all-reduce "
                         "path, bf16 working copies kept current (tests/test_parallel_gloo.py).  Default: ON for --dtype bf16 (BASELINE "
                         "configs[2], where Adam is the step's largest HBM stream), off for fp32 - no N > 1 hardware run has measured either")
    ap.add_argument("--no-shard-optimizer", dest="shard_optimizer", action="store_false")
    ap.add_argument("--dtype", choices=["f32", "bf16"], default="f32",
                    help="arithmetic of the kernels: f32 = BASELINE configs[1] (the headline), bf16 = configs[2] (bf16 activations and "
                         "working weights, fp32 accumulation, fp32 master weights / gradients / Adam)")
    ap.add_argument("--f32-mma", choices=["exact", "split3", "planes3"], default="planes3",
                    help="arithmetic form of the fp32 path's matrix products (include/sh_kernels.h enum sh_mma_mode): exact = fp32 "
                         "MFMA, the reference's own arithmetic (timed in secondary.f32_exact_step); split3 = every fp32 operand split exactly into three bf16 "
                         "terms, six partial products on the bf16 MFMA with fp32 accumulation (fp32-level error: the GPU parity tests run "
                         "in every form at the same tolerances); planes3 = the same arithmetic with the split written ONCE by the producer "
                         "of a tensor as three bf16 planes the conv kernels gather (csrc/p3_conv.hip; tests/test_p3.py gates its error "
                         "against a float64 evaluation next to the exact form's).  The secondary block times the other forms")
    ap.add_argument("--cpu-iters", type=int, default=8, help="timed CPU-baseline steps (8 steps at batch 64 = ~13 s of host work)")
    ap.add_argument("--template", default=os.path.join("tests", "golden", "template6890.npz"),
                    help="mesh hierarchy fixture; tests/golden/template27554.npz + --batch 32 is BASELINE config 4")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(args.gpus, sys.argv[1:]))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        if rank == 0:
            print("bench.py: --gpus %d but WORLD_SIZE=%d (launch with torch.distributed.run)" % (args.gpus, world), file=sys.stderr)
        sys.exit(2)
    if world > 1 and "SH_BENCH_ATTEMPT" not in os.environ and os.environ.get("SH_BENCH_SUPERVISE", "1") != "0":
        sys.exit(supervise_rank(sys.argv[1:]))     # this process stays GPU-free; the rank's work runs in children
    if "SH_BENCH_ATTEMPT" in os.environ:
        # a supervisor's child: die with the supervisor even when that one is SIGKILLed (its finally block cannot run then) - set
        # before this process touches the GPU; a supervisor that is already gone means there is nobody to report to
        try:
            import ctypes
            import signal
            ctypes.CDLL(None, use_errno=True).prctl(1, int(signal.SIGKILL), 0, 0, 0)      # PR_SET_PDEATHSIG
            if os.getppid() == 1:
                sys.exit(1)
        except (OSError, AttributeError):
            pass
    if os.environ.get("SH_BENCH_DRYRUN", "0") != "0":
        if args.gpus != world:
            sys.exit(2)
        return dry_run(args, world, rank)
    # SH_BENCH_BACKEND=gloo lets the multi-rank control flow be exercised on a box with ONE GPU (all ranks share it,
    # collectives go through the host); the measured configuration is always nccl = RCCL, one rank per GPU
    backend = os.environ.get("SH_BENCH_BACKEND", "nccl")
    dev_index = local_rank if backend == "nccl" else local_rank % max(1, torch.cuda.device_count())
    # SH_BENCH_FORCE_REDUCER=1: run the data-parallel control flow (hooks, buckets, RCCL calls, eager launches) in a world
    # of ONE rank - what the multi-GPU path costs on the host side, measurable on a 1-GPU box
    force_reducer = os.environ.get("SH_BENCH_FORCE_REDUCER", "0") != "0"
    # SH_BENCH_DP_SAFE=1 (the supervisor's last attempt): the most conservative data-parallel configuration - communicator
    # created lazily, no stream priority, one blocking all-reduce pass (sum of pre-scaled gradients) after backward
    safe = os.environ.get("SH_BENCH_DP_SAFE", "0") != "0"
    if force_reducer and world == 1:
        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
    if world > 1 or force_reducer:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # the all-reduce kernels share the chip with backward kernels that are sized to fill every CU: give the
        # collectives' stream dispatch priority so their few workgroups are placed as soon as a slot frees
        if not safe:
            os.environ.setdefault("TORCH_NCCL_HIGH_PRIORITY", "1")
        if backend == "nccl" and not safe:
            dist.init_process_group("nccl", device_id=torch.device("cuda", dev_index))     # communicator created here, eagerly
        else:
            torch.cuda.set_device(dev_index)
            dist.init_process_group(backend)                                               # safe mode: created by the first collective
        mark_phase("rendezvous")
        injected_rank_failure(rank)
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)

    import semantichuman_amd as sh
    from semantichuman_amd import _lib, synthetic
    from semantichuman_amd.hierarchy import load_hierarchy
    from semantichuman_amd.parallel import GradientAllReducer
    _lib.load()                                   # fail loudly if the HIP library is missing
    _lib.set_f32_mma_mode(args.f32_mma)           # explicit, and reported in config.workload / build.f32_mma

    h = load_hierarchy(args.template if os.path.isabs(args.template) else os.path.join(ROOT, args.template))
    B = args.batch
    torch.manual_seed(2)                          # cfgs.py:46 seed; identical replicas on every rank
    model = sh.SpiralAutoencoder(FE, FD, 256, h.sizes, h.spiral_sizes, h.spirals, h.D, h.U, dev)
    if args.dtype == "bf16":
        model.set_compute_dtype(torch.bfloat16)
    init_state = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
    ft = sh.FaceTables(h.faces, h.sizes[0] + 1, dev)
    want_shard = (args.dtype == "bf16") if args.shard_optimizer is None else bool(args.shard_optimizer)
    shard_opt = want_shard and (world > 1 or force_reducer) and not safe
    reducer = GradientAllReducer(model, bucket_cap_mb=64.0, force_collectives=force_reducer, overlap=not safe,
                                 average_in_collective=not safe, shard_large=shard_opt,
                                 large_message_dtype=torch.bfloat16 if (args.grad_comm == "bf16" and not shard_opt) else None) \
        if (world > 1 or force_reducer) else None
    # the optimizer's parameters: the model's - or, with --shard-optimizer, this rank's 1 / world slice of the two latent FCs
    # (reduce-scattered gradients, all-gathered weights; parallel.GradientAllReducer(shard_large=True))
    opt_params = reducer.optimizer_params() if (reducer is not None and shard_opt) else list(model.parameters())
    if args.adam == "hip":                        # main.py:262
        optim = sh.optim.Adam(opt_params, lr=1e-3, weight_decay=5e-5)
        if world == 1 and args.adam_overlap:
            optim.overlap_backward()              # the two latent FCs (99 % of the parameters) update underneath the encoder backward
        # one GPU: nothing consumes the latent FCs' weight gradients but Adam, so the kernel that computes a tile of them applies
        # the update to that tile (sh_linear_bwd_wgt_adam; bit-identical, tests/test_optim.py) - 8 of 32 bytes per weight never move.
        # With a gradient all-reduce in between (N > 1) the gradients have to exist: the ordinary two-kernel form.
        fused_update = reducer is None and not args.no_fused_update and not args.adam_overlap
        if fused_update:
            optim.fuse_linear_weight_gradients([model.fc_latent_enc, model.fc_latent_dec])
    else:
        optim = torch.optim.Adam(opt_params, lr=1e-3, weight_decay=5e-5, capturable=True, fused=True)
        fused_update = False

    n_data = 16 * B                               # resident synthetic set, disjoint per rank
    data = torch.from_numpy(synthetic.synth_batch(h.verts, n_data, seed=100 + rank)).to(dev)
    test = torch.from_numpy(synthetic.synth_batch(h.verts, B, seed=7)).to(dev)

    mark_phase("built")
    xin = torch.empty((B, h.sizes[0] + 1, 3), dtype=torch.float32, device=dev)
    last = {}                                     # the step's loss tensor (a fixed address inside the graph's pool when captured)
    unit = torch.ones((), dtype=torch.float32, device=dev)   # d loss / d loss, made once instead of one fill launch per step

    def fwd_bwd():
        optim.zero_grad(set_to_none=True)
        x_hat, _ = model(xin)
        loss, _ = sh.recon_loss(x_hat, xin, ft, 1e-2)     # l1 + 1e-2 * edge (train_funcs.py:501-508, traincfg.yaml:41), fused
        if reducer:
            reducer.prepare()
        loss.backward(unit)
        last["loss"] = loss.detach()

    def one_step():
        fwd_bwd()
        if reducer:
            reducer.finish()
        optim.step()
        if reducer:
            reducer.gather_weights()              # (--shard-optimizer: the updated slices of the large parameters; else nothing)

    # One hipGraph holds the whole step - forward, backward, the RCCL all-reduces (launched from the gradient hooks in the
    # middle of backward, on the process group's own stream: the capture forks to it and joins in finish()) and Adam - so
    # the multi-rank step is replayed, not host-paced.  torch's NCCL process group is capturable once the communicator
    # exists (init_process_group(device_id=...) creates it eagerly; the warm-up steps below run every collective once).
    use_graph = not args.no_graph
    graph = None
    graph_note = None
    if use_graph and reducer is not None and os.environ.get("SH_BENCH_DP_GRAPH", "1") == "0":
        use_graph, graph_note = False, "eager (SH_BENCH_DP_GRAPH=0)"      # opt-out: a capture that fails WITH collectives inside is fatal
    if use_graph and reducer is not None and backend != "nccl":
        # a host-side collective (gloo: the one-GPU control-flow test) synchronises the stream: not capturable by construction
        use_graph, graph_note = False, "eager (collectives of backend %s are host-side: not capturable)" % backend
    if use_graph:
        xin.copy_(data[:B])
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            for _ in range(3):                    # warm allocator, Adam state, communicator
                one_step()
        torch.cuda.current_stream().wait_stream(s)
        torch.cuda.synchronize()
        if reducer and world > 1:
            dist.barrier()
        ok = 1
        before = torch.cuda.current_stream()
        cap_stream = torch.cuda.Stream()
        try:
            graph = torch.cuda.CUDAGraph()
            # thread_local: the process group's watchdog thread may query events of the warm-up collectives meanwhile
            with torch.cuda.graph(graph, stream=cap_stream, capture_error_mode="thread_local" if reducer else "global"):
                if os.environ.get("SH_BENCH_TEST_CAPTURE_FAIL", "0") != "0":      # test hook: a capture that dies half way
                    fwd_bwd()
                    raise RuntimeError("injected capture failure (SH_BENCH_TEST_CAPTURE_FAIL)")
                one_step()
        except Exception as e:                    # noqa: BLE001 - any capture failure means: run eagerly, and say why
            ok = 0
            graph_note = "hipGraph capture failed (%s: %s); eager launches" % (type(e).__name__, str(e).splitlines()[0][:200] if str(e) else "")
            print("bench.py[rank %d]: %s" % (rank, graph_note), file=sys.stderr)
            # (without collectives inside - measured with SH_BENCH_TEST_CAPTURE_FAIL=1 - the run continues eagerly; a capture that
            # dies AFTER it recorded RCCL work takes the process down through the process group's watchdog: SH_BENCH_DP_GRAPH=0)
            # leave the capture cleanly: end it if the context manager could not (an exception inside the body makes its
            # __exit__ raise before it restores the stream), go back to the stream we came from, drop what was captured
            try:
                if torch.cuda.is_current_stream_capturing():
                    graph.capture_end()
            except Exception:                     # noqa: BLE001
                pass
            torch.cuda.set_stream(before)
            graph = None
            torch.cuda.synchronize()
            optim.zero_grad(set_to_none=True)
        if world > 1:                             # every rank replays, or none does
            flag = torch.tensor([ok], dtype=torch.int32, device=dev)
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
            if int(flag.item()) == 0 and graph is not None:
                graph, graph_note = None, "hipGraph capture failed on another rank; eager launches"
        # capture must not change what is measured: restore the initial weights / optimizer
        model.load_state_dict(init_state)
        if args.dtype == "bf16":                  # the bf16 working copies of the latent FCs are refreshed by an EAGER read only:
            with torch.no_grad():                 # without this the first replayed step would run on the warm-up state's copies
                model(xin)
        for st in optim.state.values():
            for v in st.values():
                if torch.is_tensor(v):
                    v.zero_()

    def step(i):
        o = (i * B) % n_data
        xin.copy_(data[o:o + B])
        if graph is not None:
            graph.replay()
        else:
            one_step()

    mark_phase("captured" if graph is not None else "eager")
    for i in range(args.warmup):
        step(i)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    mark_phase("warm")
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(args.warmup + i)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    mark_phase("timed")
    final_loss = float(last["loss"].item())

    with torch.no_grad():
        xh, _ = model(test)
        l2mm = float(sh.vertex_l2_mm(xh, test).item())

    result = {
        "metric": "training meshes/sec at %d verts, batch=%d" % (h.sizes[0], B),
        "value": world * B * args.steps / elapsed,
        "unit": "meshes/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": 1e3 * elapsed / args.steps,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": args.dtype, "data": "synthetic",
        "config": {"workload": (({"split3": "[fp32 products as exact bf16x3 operand splits, six partial products on the bf16 MFMA, fp32 accumulate] ",
                                  "planes3": "[fp32 products as exact bf16x3 operand splits written once by the producer as three bf16 planes, "
                                             "six partial products on the bf16 MFMA, fp32 accumulate] "}.get(args.f32_mma, "[fp32 MFMA] ")) if args.dtype == "f32"
                                else "[bf16 kernels, fp32 master weights] ") +
                               "plain spiral AE training step (fwd + L1 + 1e-2*edge loss + bwd + Adam), %s, levels %s, "
                               "spiral sizes %s, nz 256, %.2fM params"
                               % ("box_sphere(42,42,20) 6890-vertex template" if h.sizes[0] == 6890 else "%d-vertex template" % h.sizes[0],
                                  h.sizes, h.spiral_sizes[:-1], sum(p.numel() for p in model.parameters()) / 1e6),
                   "global_batch": world * B, "per_gpu_batch": B, "parallelism": "dp%d" % world,
                   "launch": (("hipGraph replay" + (" (RCCL all-reduces inside the graph)" if reducer else "")) if graph is not None
                              else (graph_note or "eager")) + (" safe mode (SH_BENCH_DP_SAFE)" if safe and reducer else "") +
                             (" [sharded update of the large parameters: reduce-scatter / all-gather]" if shard_opt else "") + attempt_note(),
                   "adam": ("library kernel; the two latent FCs' update applied inside their weight-gradient kernels (sh_linear_bwd_wgt_adam)"
                            if fused_update else ("library multi-tensor kernel" if args.adam == "hip" else "torch fused capturable")),
                   **({"gradient_messages": "%.1f MB %s all-reduce per step" % (
                       sum(b.numel * (2 if (b.inplace and args.grad_comm == "bf16") else 4) for b in reducer.buckets) / 1e6,
                       "bf16 (large) + fp32" if args.grad_comm == "bf16" else "fp32")} if reducer else {})},
        "train_loss_last": final_loss,
        "recon_l2_mm_after_run": l2mm,
    }

    result["whole_step"] = whole_step_block(model, B, args.dtype, elapsed / args.steps, args.f32_mma if args.dtype == "f32" else None)

    # ---- the data-parallel job as RCCL saw it: ranks, devices, and the gradient messages timed alone (HIP events on this
    # rank's stream around blocking all-reduces of the step's own message sizes, MAX over ranks)
    if reducer is not None and dist.is_initialized():
        result["collective"] = collective_block(reducer, world, rank, dev, backend)

    # ---- roofline of the dominant kernel: HIP events recorded by the library around every launch
    if not args.no_roofline:
        # EVERY rank runs these steps (the gradient all-reduce inside them is a collective); only rank 0 records
        nprof = 5
        from semantichuman_amd import stack as _stack
        overlap_was, _stack.OVERLAP_WGRAD = _stack.OVERLAP_WGRAD, False     # serial launches: clean per-kernel durations
        if args.adam == "hip":
            optim.remove_overlap()
        if rank == 0:
            _lib.profile_enable(True)
        for i in range(nprof):
            o = (i * B) % n_data
            xin.copy_(data[o:o + B])
            fwd_bwd()
            if reducer:
                reducer.finish()
            optim.step()
            if reducer:
                reducer.gather_weights()
        torch.cuda.synchronize()
        _stack.OVERLAP_WGRAD = overlap_was
    if rank == 0 and not args.no_roofline:
        recs = _lib.profile_records_by_kernel()
        _lib.profile_enable(False)
        result.update((roofline_bf16 if args.dtype == "bf16" else roofline_f32)(recs, model, B, nprof, h.sizes[0]))

    cpu_l2_ref = {}
    # ---- CPU baseline: the oracle (reference formulation) on this box's host cores, rank 0, N=1
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        from oracle import ref_cpu
        S, D, U = h.dense_constants()
        om = ref_cpu.SpiralAEOracle(FE, FD, 256, h.sizes, h.spiral_sizes, S, D, U)
        om.load_state_dict(init_state)
        oopt = torch.optim.Adam(om.parameters(), lr=1e-3, weight_decay=5e-5)
        xc = data[:B].cpu()
        # thread count: the fastest of a short calibration - all cores is NOT it (measured on the 256-thread host of
        # the MI355X box: 30 meshes/s with 16 threads, 0.4 with 256; tools/cpu_threads_probe.py)
        ncpu = os.cpu_count() or 1
        best, xs = (0.0, 1), xc[:min(B, 16)]
        for nt in [t for t in (8, 16, 32, 64) if t <= ncpu] or [ncpu]:
            torch.set_num_threads(nt)
            ref_cpu.train_step(om, oopt, xs, faces=h.faces, edgereg_w=1e-2)
            t0 = time.perf_counter()
            ref_cpu.train_step(om, oopt, xs, faces=h.faces, edgereg_w=1e-2)
            rate = xs.shape[0] / (time.perf_counter() - t0)
            if rate > best[0]:
                best = (rate, nt)
        ncores = best[1]
        torch.set_num_threads(ncores)
        om.load_state_dict(init_state)
        oopt = torch.optim.Adam(om.parameters(), lr=1e-3, weight_decay=5e-5)     # fresh moments: the timed steps are also the
        ref_cpu.train_step(om, oopt, xc, faces=h.faces, edgereg_w=1e-2)          # matched-L2 trajectory (1 warm-up + cpu_iters)
        t0 = time.perf_counter()
        for _ in range(args.cpu_iters):
            ref_cpu.train_step(om, oopt, xc, faces=h.faces, edgereg_w=1e-2)
        ct = time.perf_counter() - t0
        result["cpu_baseline"] = {"value": B * args.cpu_iters / ct, "unit": "meshes/s", "cores": ncores, "kind": "port",
                                  "sample": "%d training steps at batch %d (same template, same init) after 1 warm-up, torch CPU fp32, "
                                            "oracle/ref_cpu.py; %d threads = the fastest of {8,16,32,64} in a one-step calibration at "
                                            "batch 16 (host has %d)" % (args.cpu_iters, B, ncores, ncpu)}
        # ---- matched L2: the SAME trajectory on the HIP path (same init, same batch, same number of steps, Adam from zero
        # moments), held-out per-vertex L2 (test_funcs.py:46-49) of both
        with torch.no_grad():
            l2_cpu = float(ref_cpu.eval_metrics(om(test.cpu())[0], test.cpu())[1])
        cpu_l2_ref["mm"] = l2_cpu
        margs = (FE, FD, 256, h.sizes, h.spiral_sizes, h.spirals, h.D, h.U)
        l2_hip = hip_trajectory_l2(sh, margs, init_state, args.dtype, data[:B], test, ft, 1 + args.cpu_iters, dev)
        result["matched_l2"] = {"steps": 1 + args.cpu_iters, "batch": B, "hip_mm": l2_hip, "cpu_oracle_mm": l2_cpu,
                                "rel_diff": abs(l2_hip - l2_cpu) / l2_cpu,
                                "note": "held-out per-vertex L2 after the same training steps from the same weights on the same batch: "
                                        "HIP path (%s kernels) vs the fp32 CPU oracle" % args.dtype}
        # ---- the two thread counts SURVEY 8d asks for beside the calibrated one, on a bounded sample (batch 4, one timed step)
        for label, nt, nb, warm in (("one_thread", 1, 4, 1), ("all_cores", ncpu, 2, 0)):    # all 256 threads: ~10 s per mesh, no warm-up
            xs4 = xc[:nb]
            torch.set_num_threads(nt)
            om.load_state_dict(init_state)
            o4 = torch.optim.Adam(om.parameters(), lr=1e-3, weight_decay=5e-5)
            for _ in range(warm):
                ref_cpu.train_step(om, o4, xs4, faces=h.faces, edgereg_w=1e-2)
            t0 = time.perf_counter()
            ref_cpu.train_step(om, o4, xs4, faces=h.faces, edgereg_w=1e-2)
            result["cpu_baseline"][label] = {"value": nb / (time.perf_counter() - t0), "unit": "meshes/s", "cores": nt,
                                             "sample": "1 training step at batch %d after %d warm-up" % (nb, warm)}
        torch.set_num_threads(ncores)

    # ---- the other BASELINE configurations, measured in the same run (each a few hundred ms of GPU time): what README /
    # DESIGN quote for them is traceable to this line
    if rank == 0 and world == 1 and not args.no_secondary and h.sizes[0] == 6890 and args.dtype == "f32":
        result["secondary"] = secondary_block(sh, h, B, dev, init_state, data, test, ft, cpu_l2_ref.get("mm"), args)

    if rank == 0:
        result["build"] = {"lib_build_id": _lib.build_id(), "source_sha16": lib_sha16(), "f32_mma": _lib.get_f32_mma_mode(),
                           "env": _lib.env_overrides()}
        print(json.dumps(result))
    if dist.is_initialized():
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

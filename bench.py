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
PEAK_HBM_GBS = 8000.0             # MI355X_MICROARCH.md: HBM3E spec peak


def conv_launch_table(model, B):
    """Algorithmic FLOPs / bytes per launch of every conv kernel instance, keyed by the kernel
    name the library's profiler reports (DESIGN.md 'Roofline accounting')."""
    from semantichuman_amd import _lib
    out = {}

    def add(name, flops, nbytes):
        e = out.setdefault(name, {"flops": 0.0, "bytes": 0.0, "launches": 0})
        e["flops"] += flops; e["bytes"] += nbytes; e["launches"] += 1

    def nt(c):
        t = (c + 15) // 16
        return 1 if t <= 1 else 2 if t <= 2 else 4 if t <= 4 else 8
    first = True
    tb16 = "true" if B >= 16 else "false"          # batch-slice width of the tiles (SH_GG_TB default 16)
    tb = 16 if B >= 16 else 1 << max(0, (B - 1).bit_length())

    def nt_split(rows, nout, cg):
        """Output-channel tiles per workgroup after the few-tiles split of dispatch_gg() (csrc/spiral_conv.hip)."""
        t, blocks, split = nt(nout), -(-rows // (128 // tb)) * -(-B // tb), 1
        direct = cg % 4 == 0
        while t > 2 and blocks * split < 768:
            t //= 2; split *= 2
        if t == 2 and blocks * split * (2 if direct else 1) < 768:
            t, split = 1, split * 2
        if t == 8 and direct:
            t, split = 4, split * 2
        return t, blocks * split
    def gg_name(t, blocks128, cg, bwd):
        """dispatch_gg(): two channel tiles with 16-byte gathers run the direct (LDS-free gather) form; 3-channel
        gathered rows with one channel tile run the padded-quad (dwordx3) mode of the staged kernel."""
        if cg == 3 and t == 1:
            return "gather_gemm_kernel<1, true, %s, %s, true>" % (bwd, tb16)
        vec4 = "true" if cg % 4 == 0 else "false"
        if t in (2, 4) and vec4 == "true":
            return "gather_gemm_direct_kernel<%d, %s, %d>" % (t, bwd, 1 if blocks128 <= 1024 else 2)
        return "gather_gemm_kernel<%d, %s, %s, %s, false>" % (t, vec4, bwd, tb16)

    for stack in (model._enc_stack, model._dec_stack):
        for st in stack.steps:
            if st.kind != "conv":
                continue
            K = st.S * st.cin
            fl = 2.0 * B * st.R * K * st.cout
            vec = "true" if st.cin % 4 == 0 else "false"
            # fwd: read each needed input row once + weights, write output
            byt = 4.0 * (B * st.n_in * st.cin + B * st.R * st.cout + st.cout * K)
            add(gg_name(*nt_split(st.R, st.cout, st.cin), cg=st.cin, bwd="false"), fl, byt)
            not_first = not (first and stack is model._enc_stack)
            thin = not_first and st.R == st.n_in and bool(_lib.load().sh_spiral_conv_bwd_wgt_thin_ok(B, st.n_in, st.S, st.cin, st.cout, 0))
            if not_first and not thin:
                # backward-data = the same kernel over the transposed table; algorithmic FLOPs are
                # those of the R*S real (row, position) pairs, not of the padded n_in*S table
                add(gg_name(*nt_split(st.n_in, st.cin, st.cout), cg=st.cout, bwd="true"), fl, byt)
            if thin:
                # the 16 -> 3 channel layer: role-swapped weight gradient AND backward-data in one launch (csrc/wgrad_thin.hip)
                add("wgrad_thin_kernel<f32>", 2 * fl, 2 * byt)
            elif st.cin % 4 == 0 or st.cin == 3:       # same choices as plan_wgrad() in csrc/spiral_conv.hip
                cot = nt(st.cout)
                add("wgrad_stream_kernel<%d, %d, %d, %s, %s>" % (cot, 1 if B <= 4 else 4, 3 if cot <= 2 else 2,
                                                                 "true" if B % (4 if B <= 4 else 16) == 0 else "false",
                                                                 "true" if st.cin == 3 else "false"), fl, byt)
            else:
                cp = nt(st.cout) * 16
                cost1, cost2 = -(-K // 64) * (64 + cp), -(-K // 128) * (128 + cp)
                ctw = 1 if (nt(st.cout) == 8 or cost1 <= cost2) else 2
                add("wgrad_kernel<%d, %d, %s>" % (nt(st.cout), ctw, vec), fl, byt)
            first = False
    return out


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
    if fam not in ("conv_bf16_kernel", "wgrad_bf16_kernel"):
        return None
    f = dict(kv.split("=") for kv in shape.split() if "=" in kv)
    try:
        if fam == "conv_bf16_kernel":
            bwd = name.split("<")[1].split(",")[3].strip() == "true"
            return (fam, bwd, int(f["R"]), int(f["S"]), int(f["Cg"]), int(f["N"]))
        return (fam, None, int(f["R"]), int(f["S"]), int(f["Cin"]), int(f["N"]))
    except (KeyError, ValueError, IndexError):
        return None


def lib_sha16():
    """Identity of the kernel-library build: SHA-256 over the kernel sources (csrc/*.hip, *.h, Makefile, in name order) - what
    the library is compiled from.  (The bytes of the .so itself are not reproducible across checkouts; a profile taken on
    these sources stays valid wherever they are rebuilt.)"""
    import glob
    import hashlib
    d = os.path.join(ROOT, "semantichuman_amd", "csrc")
    hsh = hashlib.sha256()
    for f in sorted(glob.glob(os.path.join(d, "*.hip")) + glob.glob(os.path.join(d, "*.h")) + [os.path.join(d, "Makefile")]):
        hsh.update(os.path.basename(f).encode() + b"\0" + open(f, "rb").read())
    return hsh.hexdigest()[:16]


def measured_traffic(kernel, dtype):
    """HBM bytes per launch of `kernel` from the committed rocprofv3 PMC passes (tools/pmc_traffic.py: FETCH_SIZE and WRITE_SIZE
    in separate passes, gfx950 corrections applied) - quoted only while the profile was taken on THIS build of the kernel
    library (hash stamp); a stale profile yields (None, reason) instead of a silently wrong number."""
    for name in ("r02_pmc_traffic_%s.json" % dtype,):
        path = os.path.join(ROOT, "profiles", name)
        try:
            pmc = json.load(open(path))
        except (OSError, ValueError):
            return None, "no PMC profile (%s)" % name
        meta = pmc.get("_meta", {})
        if meta.get("lib_sha16") != lib_sha16():
            return None, "PMC profile %s was taken on another build of the kernel library (%s != %s)" % (name, meta.get("lib_sha16"), lib_sha16())
        if kernel not in pmc:
            return None, "kernel not in %s" % name
        return pmc[kernel]["hbm_bytes_per_launch"], "bytes/launch, rocprofv3 PMC FETCH_SIZE x2 + WRITE_SIZE (profiles/%s, same library build)" % name
    return None, "no PMC profile"


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


def _free_port():
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def self_launch(n, argv):
    """`python bench.py --gpus N` without a launcher: start N fresh rank processes of this script (one per GPU) with the
    torch.distributed.run environment, wait, and return the first non-zero exit status.  Called BEFORE anything in this
    process initialises the GPU (children are started with subprocess, never exec'd from a process that touched HIP)."""
    import subprocess
    if os.environ.get("SH_BENCH_DRYRUN", "0") == "0" and os.environ.get("SH_BENCH_BACKEND", "nccl") == "nccl":
        have = torch.cuda.device_count()                       # counts devices without initialising the GPU in this process
        if have < n:
            print("bench.py: --gpus %d but this node has %d visible GPU(s)" % (n, have), file=sys.stderr)
            return 2
    env = dict(os.environ)
    env.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n))
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")          # dmabuf IPC only on this pool (RCCL needs it)
    env.setdefault("OMP_NUM_THREADS", "4")
    procs = []
    for r in range(n):
        e = dict(env, RANK=str(r), LOCAL_RANK=str(r))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=e))
    rc = 0
    try:
        while procs:
            for p in list(procs):
                st = p.poll()
                if st is None:
                    continue
                procs.remove(p)
                if st != 0 and rc == 0:
                    rc = st
                    for q in procs:                            # a dead rank leaves the others blocked in a collective
                        q.terminate()
            time.sleep(0.05)
    finally:
        for p in procs:
            p.kill()
    return rc


def dry_run(args, world, rank):
    """SH_BENCH_DRYRUN=1: the launch / rendezvous / timing / reporting skeleton of this script over gloo with no GPU work
    (the CPU container has no device): rendezvous, warm-up, K barrier-bracketed empty steps, MAX over ranks, one JSON
    line from rank 0.  tests/test_bench_launch.py runs it for N = 2 through the self-launcher."""
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("gloo")
        dist.barrier()
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
    if rank == 0:
        print(json.dumps({"metric": "training meshes/sec at 6890 verts, batch=%d" % args.batch, "value": None, "unit": "meshes/s",
                          "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / max(1, args.steps),
                          "dry_run": True, "config": {"global_batch": world * args.batch, "parallelism": "dp%d" % world}}))
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
    ap.add_argument("--no-graph", action="store_true", help="eager launches instead of a captured hipGraph")
    ap.add_argument("--adam", choices=["hip", "torch"], default="hip", help="library Adam kernel, or torch's fused capturable Adam")
    ap.add_argument("--adam-overlap", action="store_true", help="update the big parameters on a side stream underneath backward (measured: no gain, the GPU is already saturated)")
    ap.add_argument("--grad-comm", choices=["fp32", "bf16"], default="fp32",
                    help="N > 1: type of the two large gradient messages (bf16 halves the xGMI bytes; fp32 is the measured default)")
    ap.add_argument("--dtype", choices=["f32", "bf16"], default="f32",
                    help="arithmetic of the kernels: f32 = BASELINE configs[1] (the headline), bf16 = configs[2] (bf16 activations and "
                         "working weights, fp32 accumulation, fp32 master weights / gradients / Adam)")
    ap.add_argument("--cpu-iters", type=int, default=8, help="timed CPU-baseline steps (8 steps at batch 64 = ~13 s of host work)")
    ap.add_argument("--template", default=os.path.join("tests", "golden", "template6890.npz"),
                    help="mesh hierarchy fixture; tests/golden/template27554.npz + --batch 32 is BASELINE config 4")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(args.gpus, sys.argv[1:]))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
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
    if force_reducer and world == 1:
        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
    if world > 1 or force_reducer:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # the all-reduce kernels share the chip with backward kernels that are sized to fill every CU: give the
        # collectives' stream dispatch priority so their few workgroups are placed as soon as a slot frees
        os.environ.setdefault("TORCH_NCCL_HIGH_PRIORITY", "1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", dev_index))
        else:
            dist.init_process_group(backend)
    if args.gpus != world:
        if rank == 0:
            print("bench.py: --gpus %d but WORLD_SIZE=%d (launch with torch.distributed.run)" % (args.gpus, world), file=sys.stderr)
        sys.exit(2)
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)

    import semantichuman_amd as sh
    from semantichuman_amd import _lib, synthetic
    from semantichuman_amd.hierarchy import load_hierarchy
    from semantichuman_amd.parallel import GradientAllReducer
    _lib.load()                                   # fail loudly if the HIP library is missing

    h = load_hierarchy(args.template if os.path.isabs(args.template) else os.path.join(ROOT, args.template))
    B = args.batch
    torch.manual_seed(2)                          # cfgs.py:46 seed; identical replicas on every rank
    model = sh.SpiralAutoencoder(FE, FD, 256, h.sizes, h.spiral_sizes, h.spirals, h.D, h.U, dev)
    if args.dtype == "bf16":
        model.set_compute_dtype(torch.bfloat16)
    init_state = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
    if args.adam == "hip":                        # main.py:262
        optim = sh.optim.Adam(model.parameters(), lr=1e-3, weight_decay=5e-5)
        if world == 1 and args.adam_overlap:
            optim.overlap_backward()              # the two latent FCs (99 % of the parameters) update underneath the encoder backward
    else:
        optim = torch.optim.Adam(model.parameters(), lr=1e-3, weight_decay=5e-5, capturable=True, fused=True)
    ft = sh.FaceTables(h.faces, h.sizes[0] + 1, dev)
    reducer = GradientAllReducer(model, bucket_cap_mb=64.0, force_collectives=force_reducer,
                                 large_message_dtype=torch.bfloat16 if args.grad_comm == "bf16" else None) \
        if (world > 1 or force_reducer) else None

    n_data = 16 * B                               # resident synthetic set, disjoint per rank
    data = torch.from_numpy(synthetic.synth_batch(h.verts, n_data, seed=100 + rank)).to(dev)
    test = torch.from_numpy(synthetic.synth_batch(h.verts, B, seed=7)).to(dev)

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

    use_graph = (not args.no_graph) and world == 1 and not force_reducer
    graph = None
    if use_graph:
        # capture forward+backward+Adam once, replay per step: removes ~100 host launches/step
        xin.copy_(data[:B])
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            for _ in range(3):                    # warm allocator, Adam state
                fwd_bwd(); optim.step()
        torch.cuda.current_stream().wait_stream(s)
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            fwd_bwd(); optim.step()
        # capture must not change what is measured: restore the initial weights / optimizer
        model.load_state_dict(init_state)
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
            fwd_bwd()
            if reducer:
                reducer.finish()
            optim.step()

    for i in range(args.warmup):
        step(i)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
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
        "config": {"workload": ("" if args.dtype == "f32" else "[bf16 kernels, fp32 master weights] ") +
                               "plain spiral AE training step (fwd + L1 + 1e-2*edge loss + bwd + Adam), %s, levels %s, "
                               "spiral sizes %s, nz 256, %.2fM params"
                               % ("box_sphere(42,42,20) 6890-vertex template" if h.sizes[0] == 6890 else "%d-vertex template" % h.sizes[0],
                                  h.sizes, h.spiral_sizes[:-1], sum(p.numel() for p in model.parameters()) / 1e6),
                   "global_batch": world * B, "per_gpu_batch": B, "parallelism": "dp%d" % world,
                   "launch": "hipGraph replay" if graph is not None else "eager",
                   **({"gradient_messages": "%.1f MB %s all-reduce per step" % (
                       sum(b.numel * (2 if (b.inplace and args.grad_comm == "bf16") else 4) for b in reducer.buckets) / 1e6,
                       "bf16 (large) + fp32" if args.grad_comm == "bf16" else "fp32")} if reducer else {})},
        "train_loss_last": final_loss,
        "recon_l2_mm_after_run": l2mm,
    }

    fl_step, by_step = step_work(model, B, args.dtype)
    result["whole_step"] = {"flops": fl_step, "hbm_bytes_ideal": by_step, "tflops": fl_step / (elapsed / args.steps) / 1e12,
                            "gbps": by_step / (elapsed / args.steps) / 1e9,
                            "frac_mfma": fl_step / (elapsed / args.steps) / 1e12 / (2500.0 if args.dtype == "bf16" else PEAK_F32_MFMA_TFLOPS),
                            "frac_hbm": by_step / (elapsed / args.steps) / 1e9 / PEAK_HBM_GBS,
                            "note": "algorithmic FLOPs and fused-ideal bytes of one step (SURVEY 8d) / measured ms_per_step, per GPU"}

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
        torch.cuda.synchronize()
        _stack.OVERLAP_WGRAD = overlap_was
    if rank == 0 and not args.no_roofline and args.dtype == "bf16":
        recs = _lib.profile_records_by_kernel()
        _lib.profile_enable(False)
        work = bf16_work_table(model, B)
        agg = {}
        for name, shape, ms in recs:
            a = agg.setdefault(name, {"n": 0, "ms": 0.0, "flops": 0.0, "bytes": 0.0, "matched": 0})
            a["n"] += 1; a["ms"] += ms
            key = parse_tag(name, shape)
            if key in work:
                a["flops"] += work[key][0]; a["bytes"] += work[key][1]; a["matched"] += 1
        kernels = []
        for name, a in sorted(agg.items(), key=lambda kv: -kv[1]["ms"]):
            e = {"kernel": name, "launches_per_step": a["n"] / nprof, "avg_ms": a["ms"] / a["n"], "ms_per_step": a["ms"] / nprof}
            if a["matched"] == a["n"] and a["n"]:
                e["gbps"] = a["bytes"] / (a["ms"] * 1e-3) / 1e9
                e["tflops"] = a["flops"] / (a["ms"] * 1e-3) / 1e12
            kernels.append(e)
        fam = {}
        for k in kernels:
            f = fam.setdefault(k["kernel"].split("<")[0].split("|")[0], 0.0)
            fam[k["kernel"].split("<")[0].split("|")[0]] = f + k["ms_per_step"]
        conv = [k for k in kernels if "gbps" in k]
        dom = conv[0] if conv else kernels[0]
        a = agg[dom["kernel"]]
        # bf16: ridge of the chip ~ 2.5 PF / 8 TB/s = 300 FLOP/B, these layers have 30-250 FLOP/B -> HBM roof
        traffic, traffic_note = measured_traffic(dom["kernel"], "bf16")
        result["roofline"] = {"bound": "hbm", "kernel": dom["kernel"], "achieved": dom.get("gbps"), "peak": PEAK_HBM_GBS, "unit": "GB/s",
                              "frac": (dom["gbps"] / PEAK_HBM_GBS) if "gbps" in dom else None, "traffic": traffic, "traffic_unit": traffic_note,
                              "avg_launch_ms": dom["avg_ms"], "launches_per_step": dom["launches_per_step"],
                              "algorithmic_bytes_per_launch": a["bytes"] / a["n"] if a["n"] else None,
                              "flops_per_launch": a["flops"] / a["n"] if a["n"] else None,
                              "mfma_tflops": dom.get("tflops"), "mfma_peak_tflops": 2500.0}
        result["kernel_families"] = {n: {"ms_per_step": v} for n, v in sorted(fam.items(), key=lambda kv: -kv[1])[:10]}
        result["kernel_breakdown"] = kernels[:10]
        result["hip_kernel_ms_per_step"] = sum(k["ms_per_step"] for k in kernels)
    elif rank == 0 and not args.no_roofline:
        recs = _lib.profile_records_by_kernel()
        _lib.profile_enable(False)
        agg = {}
        for name, _shape, ms in recs:
            a = agg.setdefault(name, [0, 0.0])
            a[0] += 1; a[1] += ms
        table = conv_launch_table(model, B)
        kernels = []
        for name, (cnt, tot) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
            e = {"kernel": name, "launches_per_step": cnt / nprof, "avg_ms": tot / cnt, "ms_per_step": tot / nprof}
            if name in table and table[name]["launches"] == cnt / nprof:
                e["tflops"] = table[name]["flops"] / (tot / nprof * 1e-3) / 1e12
            kernels.append(e)
        # The dominant kernel of the step is the fused gather+MFMA conv kernel: `gather_gemm_kernel` and its
        # direct form `gather_gemm_direct_kernel` (forward and backward-data) take the largest share of the
        # step as a family; the roofline is
        # quoted for its single most expensive instantiation, under the exact name rocprofv3 prints
        # (profiles/), achieved = algorithmic FLOPs of its launches / their measured duration.
        conv = [k for k in kernels if "tflops" in k and k["kernel"].startswith("gather_gemm_")]
        dom = conv[0] if conv else kernels[0]
        fam = {}
        for k in kernels:
            f = fam.setdefault(k["kernel"].split("<")[0].split("|")[0], [0.0, 0.0])
            f[0] += k["ms_per_step"]
            if k["kernel"] in table:
                f[1] += table[k["kernel"]]["flops"]
        # HBM bytes per launch of that kernel from the committed rocprofv3 PMC passes (FETCH_SIZE x2 + WRITE_SIZE,
        # collected separately; tools/pmc_traffic.py) - measured on the same workload, not in this run
        traffic, traffic_note = measured_traffic(dom["kernel"], "f32")
        if "tflops" in dom:
            result["roofline"] = {"bound": "mfma", "kernel": dom["kernel"], "achieved": dom["tflops"], "peak": PEAK_F32_MFMA_TFLOPS,
                                  "unit": "TFLOP/s", "frac": dom["tflops"] / PEAK_F32_MFMA_TFLOPS, "traffic": traffic,
                                  "traffic_unit": traffic_note,
                                  "avg_launch_ms": dom["avg_ms"], "launches_per_step": dom["launches_per_step"],
                                  "flops_per_launch": table[dom["kernel"]]["flops"] / table[dom["kernel"]]["launches"],
                                  "algorithmic_bytes_per_launch": table[dom["kernel"]]["bytes"] / table[dom["kernel"]]["launches"]}
        else:
            result["roofline"] = {"bound": "hbm", "kernel": dom["kernel"], "achieved": None, "peak": PEAK_HBM_GBS, "unit": "GB/s",
                                  "frac": None, "traffic": None, "avg_launch_ms": dom["avg_ms"]}
        result["kernel_families"] = {n: {"ms_per_step": v[0], "tflops": (v[1] / (v[0] * 1e-3) / 1e12) if v[1] else None}
                                     for n, v in sorted(fam.items(), key=lambda kv: -kv[1][0])[:8]}
        result["kernel_breakdown"] = kernels[:8]
        result["hip_kernel_ms_per_step"] = sum(k["ms_per_step"] for k in kernels)

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
        m2 = sh.SpiralAutoencoder(FE, FD, 256, h.sizes, h.spiral_sizes, h.spirals, h.D, h.U, dev)
        if args.dtype == "bf16":
            m2.set_compute_dtype(torch.bfloat16)
        m2.load_state_dict(init_state)
        o2 = sh.optim.Adam(m2.parameters(), lr=1e-3, weight_decay=5e-5)
        xg = data[:B]
        for _ in range(1 + args.cpu_iters):
            o2.zero_grad(set_to_none=True)
            l2loss, _ = sh.recon_loss(m2(xg)[0], xg, ft, 1e-2)
            l2loss.backward()
            o2.step()
        with torch.no_grad():
            l2_hip = float(sh.vertex_l2_mm(m2(test)[0], test).item())
        result["matched_l2"] = {"steps": 1 + args.cpu_iters, "batch": B, "hip_mm": l2_hip, "cpu_oracle_mm": l2_cpu,
                                "rel_diff": abs(l2_hip - l2_cpu) / l2_cpu,
                                "note": "held-out per-vertex L2 after the same training steps from the same weights on the same batch: "
                                        "HIP path (%s kernels) vs the fp32 CPU oracle" % args.dtype}
        del m2, o2
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

    if rank == 0:
        print(json.dumps(result))
    if dist.is_initialized():
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

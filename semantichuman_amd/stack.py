"""Plans and executes a stack of spiral convolutions and mesh re-sampling steps on the GPU.

The reference runs encode/decode as a Python loop of small autograd ops
(models.py:115-154).  Here each stack (encoder, decoder) is ONE autograd node:

  forward   conv -> conv -> ...   every step one fused HIP kernel, activations kept in the
            vertex-major layout between steps, only the first input / last output use the
            caller's layout (strides, no transposes).  A row-select down-sampling D is folded
            into the gather table of the convolution in front of it (mesh_ops.compose_select),
            so the rows D would discard are never computed.
  backward  a hand-scheduled chain: for every conv, the weight-gradient kernel and the
            backward-data kernel (the forward kernel run over the TRANSPOSED gather table);
            the activation derivative of the PREVIOUS layer is applied in the epilogue of
            whichever kernel produces that layer's output gradient, so there is no separate
            elementwise pass and no saved pre-activation.  No atomics anywhere.

Plan construction is pure numpy (testable without a GPU); `to(device)` uploads the tables.

The launches of a stack are issued by ONE library call each way (`sh_stack_forward` / `sh_stack_backward`,
csrc/stack_exec.hip) into buffers carved out of one arena per pass: issued call by call from Python
(`run_forward` / `run_backward` below, kept for the side-stream experiments and as the cross-check of the native
sequencing, SH_STACK_NATIVE=0) a launch costs ~25 us of host time, which made the host pace an eagerly
launched step (1.2 ms of host time per step against 1.85 ms of GPU time, and more than that with gradient
collectives in the loop).
"""
from __future__ import annotations

import ctypes
import os
from dataclasses import dataclass, field
from typing import Optional

import numpy as np
import torch

from . import _lib, mesh_ops, ops
from .mesh_ops import CSR, TransposedTable

# one library call per stack and direction (csrc/stack_exec.hip); 0 = the call-by-call Python sequencing below
NATIVE = os.environ.get("SH_STACK_NATIVE", "1") != "0"
_LAYOUT_ID = {"vm": 0, "bm": 1}
# the library's own switch (csrc/stack_exec.hip reads it once): with the exact weight-gradient kernels nothing reads the forward
# images during the backward pass, so they are not kept
_P3_WGRAD = os.environ.get("SH_P3_WGRAD", "1").strip() != "0"
_P3_RAGGED = os.environ.get("SH_P3_RAGGED", "1").strip() != "0"
# Test switch: fill the activation / gradient arenas with NaN when they are allocated.  A training pass on the images (keep_fp32 == 2)
# leaves fp32 rows unwritten; a kernel that read one would otherwise see whatever the allocator's block held before - possibly the same
# rows of an earlier, identical step.
DEBUG_POISON = os.environ.get("SH_DEBUG_POISON", "0").strip() == "1"
_P3_GROUPED = os.environ.get("SH_P3_GROUPED", "1").strip() != "0"
_ALIGN = 64                       # floats: every carved buffer starts 256-byte aligned (16-byte vector accesses)


def _round(n: int) -> int:
    return (n + _ALIGN - 1) // _ALIGN * _ALIGN

# Side-stream options, both OFF: measured on MI355X, concurrency between these kernels only re-orders MFMA/HBM-saturated
# work and pays for the fork/join.
# run weight-gradient kernels on a side stream, concurrently with the backward-data chain (1.97 vs 1.93 ms/step)
OVERLAP_WGRAD = os.environ.get("SH_OVERLAP_WGRAD", "0") != "0"
# run the small list pre-sum kernels of backward-data on a side stream, underneath the weight-gradient kernel of the
# same layer (both only read dpre_i; the pre-sums are memory-bound and tiny, the weight gradient is MFMA-bound)
OVERLAP_PRESUM = os.environ.get("SH_OVERLAP_PRESUM", "0") != "0"     # measured on MI355X: 2.03 vs 1.93 ms/step - off


def _dev(a: np.ndarray, device):
    return torch.from_numpy(np.ascontiguousarray(a)).to(device)


def _csr_dev(m: CSR, device):
    return (_dev(m.rowptr, device), _dev(m.col, device), _dev(m.val, device))


@dataclass
class ConvStep:
    """One SpiralConv (optionally with a fused row-select down-sampling)."""
    param: int                 # index of the SpiralConv module in its ModuleList
    table: np.ndarray          # int32 [R, S], values in [0, n_in)
    n_in: int
    cin: int
    cout: int
    act: int
    dead_dummy_grad: bool = False   # gradient w.r.t. the input's dummy row is provably zero
    dummy_in: int = -1              # that row (default n_in - 1; elsewhere once an up-sampling was folded into the table)
    kind: str = "conv"
    # filled by finalize()
    R: int = 0
    S: int = 0
    zero_row: int = -1
    tt: Optional[TransposedTable] = None
    n_extra: int = 0           # extra rows the dpre buffer needs behind its R real rows
    dev: dict = field(default_factory=dict)

    def finalize(self):
        self.R, self.S = self.table.shape
        self.zero_row = self.R - 1                       # the dummy row of the output (models.py:49-51)
        # "no source" entries point at dpre's own dummy row, which every producer of dpre forces to
        # zero (act_backward / the zero_row epilogues).
        dummy = self.dummy_in if self.dummy_in >= 0 else self.n_in - 1
        self.tt = mesh_ops.transpose_table_dense(self.table, self.n_in, none_row=self.zero_row,
                                                 skip_row=dummy if self.dead_dummy_grad else -1)
        self.n_extra = self.tt.n_extra
        # the same sources as ragged lists (round 6): the plane backward-data kernel of a layer with a resident weight walks them
        # instead of the dense table and needs no pre-summed rows (SH_P3_RAGGED=0, or a list longer than 64: the dense form)
        self.rag = mesh_ops.transpose_table_ragged(self.table, self.n_in, none_row=self.zero_row,
                                                   skip_row=dummy if self.dead_dummy_grad else -1) if _P3_RAGGED else None
        # grouped lists (round 6): output rows with overlapping source lists share one list of the union, for the plane kernels with a
        # resident weight (csrc/p3_conv.hip conv_p3g_kernel) - forward from the table, backward-data from the ragged lists; how many
        # members a group takes is the kernel's (registers: 4 with <= 2 channel tiles, else 2).  SH_P3_GROUPED=0: none
        self.fgrp = self.bgrp = None
        if _P3_GROUPED:
            lib = _lib.load()
            m = lib.sh_spiral_conv_p3_grp_members(16, self.S, self.cin, self.cout)
            if m:
                pos = np.ascontiguousarray(np.broadcast_to(np.arange(self.S, dtype=np.int32), self.table.shape))
                self.fgrp = mesh_ops.group_lists(self.table, pos, members=m)
            m = lib.sh_spiral_conv_p3_grp_members(16, self.S, self.cout, self.cin)
            if m and self.rag is not None:
                self.bgrp = mesh_ops.group_lists(self.rag[0], self.rag[1], members=m)
        return self

    def to(self, device):
        self.dev = {"table": _dev(self.table, device), "table_t": _dev(self.tt.table_t, device)}
        if getattr(self, "rag", None) is not None:
            self.dev["rag_rows"], self.dev["rag_pos"] = _dev(self.rag[0], device), _dev(self.rag[1], device)
        for key in ("fgrp", "bgrp"):
            g = getattr(self, key, None)
            if g is not None:
                self.dev[key] = (_dev(g[0], device), _dev(g[1].view(np.int32), device), _dev(g[2], device))
        if self.tt.csr1 is not None:
            self.dev["sum1"] = _csr_dev(self.tt.csr1, device)
        if self.tt.csr2 is not None:
            self.dev["sum2"] = _csr_dev(self.tt.csr2, device)
        return self


@dataclass
class SpmmStep:
    """y = M x for a sparse re-sampling matrix (U, or a D that is not a pure row select).

    Folded form (`extend`, see fold_identity_rows): the step's input buffer has room for n_b more rows behind its `cols`
    real ones and the launch computes only those - the non-identity rows of U (`csr_fwd`); the tensor the next step sees is
    Z = [x ; U_b x], i.e. `csr` is then the matrix [I ; U_b] (its transpose serves the backward pass unchanged)."""
    csr: CSR
    kind: str = "spmm"
    csr_t: Optional[CSR] = None
    extend: bool = False
    csr_fwd: Optional[CSR] = None      # what the forward launch multiplies with (csr itself unless folded)
    dummy_out: int = -1                # row of the output that carries the dummy vertex (default: the last)
    dev: dict = field(default_factory=dict)

    def finalize(self):
        if self.csr_t is None:
            self.csr_t = self.csr.transpose()
        if self.csr_fwd is None:
            self.csr_fwd = self.csr
        if self.dummy_out < 0:
            self.dummy_out = self.csr.rows - 1
        return self

    def to(self, device):
        self.dev = {"m": _csr_dev(self.csr_fwd, device), "mt": _csr_dev(self.csr_t, device)}
        return self

    def passes_dummy_only_to_dummy(self) -> bool:
        """True if the last input column feeds only the output's dummy row (the padded 1 of
        main.py:190-191), i.e. a dead gradient on the output dummy row stays confined."""
        t = self.csr_t
        lo, hi = t.rowptr[t.rows - 1], t.rowptr[t.rows]
        return hi - lo == 1 and t.col[lo] == self.dummy_out


FOLD_U = os.environ.get("SH_FOLD_U", "1") != "0"


def fold_identity_rows(steps):
    """conv -> U -> conv: half of U's rows are identity rows (the vertices the coarse mesh kept, and the dummy row), exact
    copies of rows the first conv just wrote.  Compose the second conv's gather table with the row map of
    mesh_ops.split_identity_rows and let U append only its blended rows to the first conv's output buffer: same values in
    every gathered row (bit-identical results), half the rows written by the re-sampling launch."""
    if not FOLD_U:
        return steps
    for i in range(1, len(steps) - 1):
        prev, sp, nxt = steps[i - 1], steps[i], steps[i + 1]
        if not (sp.kind == "spmm" and prev.kind == "conv" and nxt.kind == "conv") or sp.extend:
            continue
        if nxt.n_in != sp.csr.rows or prev.table.shape[0] != sp.csr.cols:
            continue
        row_map, u_b, m = mesh_ops.split_identity_rows(sp.csr)
        if u_b.rows == sp.csr.rows:
            continue
        old_dummy = nxt.dummy_in if nxt.dummy_in >= 0 else nxt.n_in - 1
        nxt.table = np.ascontiguousarray(row_map[nxt.table]).astype(np.int32)
        nxt.dummy_in = int(row_map[old_dummy])
        nxt.n_in = m.rows
        sp.dummy_out = int(row_map[sp.csr.rows - 1])
        # the transposed matrix keeps U^T's entry order (increasing ORIGINAL row), so the backward sums - and with them
        # every gradient - stay bit-identical to the unfolded form
        ut = sp.csr.transpose()
        sp.csr_t = CSR(ut.rows, m.rows, ut.rowptr, row_map[ut.col].astype(np.int32), ut.val)
        sp.csr, sp.csr_fwd, sp.extend = m, u_b, True
    return steps


def mark_dead_dummy(steps, input_dummy_dead: bool = False):
    """Decide for every conv whether the gradient w.r.t. its input's dummy row is provably zero:
    it is when that row was produced by a masked SpiralConv, possibly handed through
    re-sampling steps that map dummy -> dummy only."""
    dead = input_dummy_dead
    for st in steps:
        if st.kind == "conv":
            st.dead_dummy_grad = dead
            dead = True                      # its own output dummy row is masked
        else:
            dead = dead and st.passes_dummy_only_to_dummy()
    return steps


class Stack:
    def __init__(self, steps, input_dummy_dead: bool = False):
        fold_identity_rows(steps)
        for s in steps:
            if s.kind == "spmm":
                s.finalize()
        mark_dead_dummy(steps, input_dummy_dead)
        for s in steps:
            if s.kind == "conv":
                s.finalize()
        self.steps = list(steps)
        self.device = None

    def to(self, device):
        for s in self.steps:
            s.to(device)
        self.device = torch.empty(0, device=device).device      # normalised ('cuda' -> 'cuda:0')
        self._nsteps = None                                      # the step table holds device pointers of the old upload
        return self

    def _extends(self, i: int) -> bool:
        st = self.steps[i]
        return st.kind == "spmm" and st.extend

    def _buffer_rows(self, i: int) -> int:
        """Rows of the buffer step i writes: its own, plus the rows a folded up-sampling appends behind them."""
        st = self.steps[i]
        rows = st.R if st.kind == "conv" else st.csr.rows
        if i + 1 < len(self.steps) and self._extends(i + 1):
            rows = max(rows, self.steps[i + 1].csr.rows)
        return rows

    def conv_steps(self):
        return [s for s in self.steps if s.kind == "conv"]

    # ------------------------------------------------------------------ native sequencing
    def _native_steps(self):
        """The sh_stack_step table (host memory; device pointers of the uploaded tables)."""
        arr = getattr(self, "_nsteps", None)
        if arr is not None:
            return arr
        P = lambda t: t.data_ptr()                                    # noqa: E731
        arr = (_lib.StackStep * len(self.steps))()
        for e, st in zip(arr, self.steps):
            if st.kind == "conv":
                e.kind, e.param = 0, st.param
                e.table, e.table_t = P(st.dev["table"]), P(st.dev["table_t"])
                e.R, e.S, e.n_in, e.cin, e.cout, e.act, e.zero_row = st.R, st.S, st.n_in, st.cin, st.cout, st.act, st.zero_row
                e.n1, e.n2 = st.tt.n1, st.tt.n2
                for name, ref in (("sum1", e.sum1), ("sum2", e.sum2)):
                    if name in st.dev:
                        ref.rowptr, ref.col, ref.val = (P(t) for t in st.dev[name])
                if "rag_rows" in st.dev:
                    e.rag_rows, e.rag_pos, e.rag_L = P(st.dev["rag_rows"]), P(st.dev["rag_pos"]), int(st.dev["rag_rows"].shape[1])
                if "fgrp" in st.dev:
                    r, q, o = st.dev["fgrp"]
                    e.fg_rows, e.fg_pos, e.fg_out, e.fg_n, e.fg_L = P(r), P(q), P(o), int(r.shape[0]), int(r.shape[1])
                if "bgrp" in st.dev:
                    r, q, o = st.dev["bgrp"]
                    e.bg_rows, e.bg_pos, e.bg_out, e.bg_n, e.bg_L = P(r), P(q), P(o), int(r.shape[0]), int(r.shape[1])
            else:
                e.kind, e.param = 1, -1
                e.m.rowptr, e.m.col, e.m.val = (P(t) for t in st.dev["m"])
                e.mt.rowptr, e.mt.col, e.mt.val = (P(t) for t in st.dev["mt"])
                e.m_rows, e.m_cols, e.extend = st.csr_fwd.rows, st.csr.cols, 1 if st.extend else 0
        self._nsteps = arr
        return arr

    def _plan(self, B: int, c0: int):
        """Buffer sizes / arena offsets (in floats) for batch B; cached."""
        key = (B, c0)
        plans = self.__dict__.setdefault("_plans", {})
        if key in plans:
            return plans[key]
        lib = _lib.load()
        n = len(self.steps)
        cin_of, c = [], c0
        for st in self.steps:
            cin_of.append(c)
            c = st.cout if st.kind == "conv" else c
        out_rows = [st.R if st.kind == "conv" else st.csr.rows for st in self.steps]
        out_ch = [st.cout if st.kind == "conv" else cin_of[i] for i, st in enumerate(self.steps)]
        # forward arena: outputs of all steps but the last
        f_off, o = np.zeros(n, dtype=np.uint64), 0
        for i in range(n - 1):
            if self._extends(i):                       # appends its rows to the previous step's buffer
                f_off[i] = f_off[i - 1]
                continue
            f_off[i] = o
            o += _round(self._buffer_rows(i) * B * out_ch[i])
        f_total = o
        # backward arena: dpre of the last step | two alternating regions for the input-gradient chain | weight_t | slabs
        last = self.steps[-1]
        o = 0
        dpre_last_off = 0
        if last.kind == "conv":
            o += _round((last.R + last.n_extra) * B * last.cout)
        gin_size = [0] * n
        for i in range(1, n):
            st, prev = self.steps[i], self.steps[i - 1]
            rows_in = st.n_in if st.kind == "conv" else st.csr.cols
            gin_size[i] = (rows_in + (prev.n_extra if prev.kind == "conv" else 0)) * B * cin_of[i]
        region = [_round(max([gin_size[i] for i in range(1, n) if i % 2 == par] + [0])) for par in (0, 1)]
        region_off = [o, o + region[0]]
        o += region[0] + region[1]
        g_off = np.zeros(n, dtype=np.uint64)
        g_mask = np.zeros(n, dtype=np.uint64)
        for i in range(1, n):
            g_off[i], g_mask[i] = region_off[i % 2], 1
        wt_off, wt_mask = np.zeros(n, dtype=np.uint64), np.zeros(n, dtype=np.uint64)
        ws_off, ws_mask, ws_bytes = np.zeros(n, dtype=np.uint64), np.zeros(n, dtype=np.uint64), np.zeros(n, dtype=np.uint64)
        for i, st in enumerate(self.steps):
            if st.kind != "conv":
                continue
            wt_off[i], wt_mask[i] = o, 1
            o += _round(st.cin * st.S * st.cout)
            # (room for either plan of the step's partial slabs: the fp32 kernels' or the three-plane weight gradient's)
            nb = max(int(lib.sh_spiral_conv_bwd_wgt_workspace(B, st.R, st.S, st.cin, st.cout)),
                     int(lib.sh_spiral_conv_bwd_wgt_p3_workspace(B, st.R, st.S, st.cin, st.cout)))
            ws_off[i], ws_mask[i], ws_bytes[i] = o, 1, nb
            o += _round((nb + 3) // 4)
        b_total = o
        # gradients of the parameters: one flat tensor, dW then dbias per parameter index
        npar = 1 + max([st.param for st in self.steps if st.kind == "conv"], default=-1)
        dW_off, db_off, shapes, o = np.zeros(npar, dtype=np.uint64), np.zeros(npar, dtype=np.uint64), [None] * npar, 0
        for st in self.steps:
            if st.kind != "conv":
                continue
            dW_off[st.param] = o
            o += _round(st.cout * st.S * st.cin)
            db_off[st.param] = o
            o += _round(st.cout)
            shapes[st.param] = (st.cout, st.S * st.cin)
        # three-plane form (SH_MMA_PLANES3): byte offsets of the plane images of the forward buffers (one arena), of the gradient
        # buffers (another; no aliasing - 288 GB) and of the weight fragments (forward and backward-data operand, one buffer
        # filled by ONE conversion launch per forward pass); a zero mask = no image (shape outside the plane kernels)
        al = lambda nbytes: (int(nbytes) + 255) // 256 * 256          # noqa: E731
        pl_off, pl_mask, o3 = np.zeros(n, dtype=np.uint64), np.zeros(n, dtype=np.uint64), 0
        for i in range(n - 1):
            if self._extends(i):
                pl_off[i], pl_mask[i] = pl_off[i - 1], pl_mask[i - 1]
                continue
            nb = int(lib.sh_p3_bytes(self._buffer_rows(i), B, out_ch[i]))
            if nb:
                pl_off[i], pl_mask[i] = o3, 1
                o3 += al(nb)
        pl_total = o3
        gpl_off, gpl_mask, o3 = np.zeros(n, dtype=np.uint64), np.zeros(n, dtype=np.uint64), 0
        for i in range(1, n):
            prev = self.steps[i - 1]
            if prev.kind != "conv":
                continue
            nb = int(lib.sh_p3_bytes(prev.R + prev.n_extra, B, prev.cout))
            if nb and lib.sh_spiral_conv_p3_ok(B, prev.S, prev.cout, prev.cin):
                gpl_off[i], gpl_mask[i] = o3, 1
                o3 += al(nb)
        dpl_off, dpl_mask = 0, 0
        if last.kind == "conv" and lib.sh_spiral_conv_p3_ok(B, last.S, last.cout, last.cin):
            nb = int(lib.sh_p3_bytes(last.R + last.n_extra, B, last.cout))
            if nb:
                dpl_off, dpl_mask = o3, 1
                o3 += al(nb)
        gpl_total = o3
        wf3_off, wf3_mask, wf3t_off, wf3t_mask = (np.zeros(n, dtype=np.uint64) for _ in range(4))
        o3 = 0
        wf3_jobs = []                                                  # (step index, transpose, byte offset)
        for i, st in enumerate(self.steps):
            if st.kind != "conv":
                continue
            if i > 0 and lib.sh_spiral_conv_p3_ok(B, st.S, st.cin, st.cout):
                wf3_off[i], wf3_mask[i] = o3, 1
                wf3_jobs.append((i, 0, o3))
                o3 += al(lib.sh_conv_wfrag3_bytes(st.S, st.cin, st.cout))
            if lib.sh_spiral_conv_p3_ok(B, st.S, st.cout, st.cin):
                wf3t_off[i], wf3t_mask[i] = o3, 1
                wf3_jobs.append((i, 1, o3))
                o3 += al(lib.sh_conv_wfrag3_bytes(st.S, st.cout, st.cin))
        wf3_total = o3
        plan = dict(pl_off=pl_off, pl_mask=pl_mask, pl_total=pl_total, gpl_off=gpl_off, gpl_mask=gpl_mask, dpl_off=dpl_off,
                    dpl_mask=dpl_mask, gpl_total=gpl_total, wf3_off=wf3_off, wf3_mask=wf3_mask, wf3t_off=wf3t_off,
                    wf3t_mask=wf3t_mask, wf3_jobs=wf3_jobs, wf3_total=wf3_total,
                    f_off=f_off * 4, f_total=f_total, dpre_last_off=dpre_last_off, g_off=g_off * 4, g_mask=g_mask,
                    wt_off=wt_off * 4, wt_mask=wt_mask, ws_off=ws_off * 4, ws_mask=ws_mask, ws_bytes=ws_bytes, b_total=b_total,
                    dW_off=dW_off * 4, db_off=db_off * 4, dW_off_f=dW_off, db_off_f=db_off, shapes=shapes, p_total=o, npar=npar,
                    out_rows=out_rows, out_ch=out_ch)
        plans[key] = plan
        return plan

    @staticmethod
    def _ptr_array(tensors):
        return (ctypes.c_void_p * len(tensors))(*[0 if t is None else t.data_ptr() for t in tensors])

    def _p3_prepare(self, plan, weights, with_backward: bool, device):
        """Three-plane form: the image arena of the forward buffers and the weight fragments - this stack's own conversion
        launch, unless `prepare_p3_frags` already converted them together with other stacks' (one launch per training step)."""
        planes = torch.empty(max(256, plan["pl_total"]), dtype=torch.uint8, device=device)
        pre = self.__dict__.pop("_p3_next", None)
        if pre is not None and pre[2] == id(plan) and (pre[3] or not with_backward):
            return planes, pre[0], pre[1]
        wf3 = torch.empty(max(256, plan["wf3_total"]), dtype=torch.uint8, device=device)
        convert_p3_frags([(self, plan, weights, wf3, 0)], with_backward)
        return planes, wf3, 0

    def native_forward(self, x, in_layout, out_layout, weights, biases, mma: str = "exact", with_backward: bool = False):
        """-> (output, arena holding the outputs of the inner steps, three-plane state or None)."""
        B = x.shape[0] if in_layout == "bm" else x.shape[1]
        rows0 = x.shape[1] if in_layout == "bm" else x.shape[0]
        c0 = x.shape[2]
        if not (x.is_contiguous() and x.dtype == torch.float32):
            raise ValueError("expected a contiguous fp32 3-D tensor, got %s %s" % (tuple(x.shape), x.dtype))
        if x.numel() >= 2 ** 32:
            raise RuntimeError("semantichuman_amd: gathered tensors are addressed with 32-bit element offsets; "
                               "%d elements is too large - split the batch" % x.numel())
        plan = self._plan(B, c0)
        n = len(self.steps)
        arena = torch.empty(max(1, plan["f_total"]), dtype=torch.float32, device=x.device)
        if DEBUG_POISON:
            arena.fill_(float("nan"))
        out = ops.alloc(B, plan["out_rows"][-1], plan["out_ch"][-1], out_layout, x.device)
        outs = plan["f_off"] + np.uint64(arena.data_ptr())
        outs[n - 1] = out.data_ptr()
        p3 = None
        keep = 1 if with_backward else 0
        planes_p = wf3_p = None
        if mma == "planes3" and B % 16 == 0 and plan["wf3_total"]:
            planes, wf3, wbase = self._p3_prepare(plan, weights, with_backward, x.device)
            # what the backward pass needs of this: the weight fragments - and, since round 6, the image arena of the forward
            # activations (6 bytes per element): the three-plane weight gradient (csrc/wgrad_p3.hip) reads a step's gathered input
            # through it.  SH_P3_WGRAD=0 (the exact weight-gradient kernels): not kept, as before.
            p3 = (planes if (with_backward and _P3_WGRAD) else None, wf3, wbase)
            # ... and with the images kept, the fp32 rows that neither pass reads are not written (sh_stack_forward keep_fp32 == 2;
            # SH_P3_DROP_FP32=0: every row, as before)
            if p3[0] is not None:
                keep = 2
            pl = (plan["pl_off"] + np.uint64(planes.data_ptr())) * plan["pl_mask"]
            wf = (plan["wf3_off"] + np.uint64(wf3.data_ptr() + wbase)) * plan["wf3_mask"]
            planes_p, wf3_p = pl.ctypes.data, wf.ctypes.data
        _lib.check(_lib.load().sh_stack_forward(n, self._native_steps(), _lib.ptr(x), _LAYOUT_ID[in_layout], rows0, c0, B,
                                                self._ptr_array(weights), self._ptr_array(biases), outs.ctypes.data,
                                                _LAYOUT_ID[out_layout], _lib.mma_id(mma), planes_p, wf3_p, keep, _lib.stream_ptr()),
                   "sh_stack_forward")
        return out, arena, p3

    def native_backward(self, x, in_layout, out_layout, arena, out, g, weights, need_x_grad, need_bias, mma: str = "exact", p3=None):
        """-> (grad_x or None, {param: (dW, db)})"""
        B = x.shape[0] if in_layout == "bm" else x.shape[1]
        rows0 = x.shape[1] if in_layout == "bm" else x.shape[0]
        c0 = x.shape[2]
        plan = self._plan(B, c0)
        n = len(self.steps)
        dev = x.device
        work = torch.empty(max(1, plan["b_total"]), dtype=torch.float32, device=dev)
        if DEBUG_POISON:
            work.fill_(float("nan"))
        flat = torch.empty(max(1, plan["p_total"]), dtype=torch.float32, device=dev)
        gx = ops.alloc(B, rows0, c0, in_layout, dev) if need_x_grad else None
        acts = plan["f_off"] + np.uint64(arena.data_ptr())
        acts[n - 1] = out.data_ptr()
        wbase, fbase = np.uint64(work.data_ptr()), np.uint64(flat.data_ptr())
        gin = (plan["g_off"] + wbase) * plan["g_mask"]
        gin[0] = gx.data_ptr() if need_x_grad else 0
        wt = (plan["wt_off"] + wbase) * plan["wt_mask"]
        ws = (plan["ws_off"] + wbase) * plan["ws_mask"]
        dW = plan["dW_off"] + fbase
        assert len(need_bias) == plan["npar"] == len(weights)
        db = (plan["db_off"] + fbase) * np.array([1 if nb else 0 for nb in need_bias], dtype=np.uint64)
        gpl_p = wf3t_p = inpl_p = None
        dpl = ctypes.c_void_p(0)
        if mma == "planes3" and p3 is not None and p3[0] is not None:
            # image of the INPUT of step i = image of the buffer step i - 1 wrote
            inpl = np.zeros(n, dtype=np.uint64)
            inpl[1:] = ((plan["pl_off"] + np.uint64(p3[0].data_ptr())) * plan["pl_mask"])[:n - 1]
            inpl_p = inpl.ctypes.data
        if mma == "planes3" and p3 is not None:
            gimg = torch.empty(max(256, plan["gpl_total"]), dtype=torch.uint8, device=dev)
            gpl = (plan["gpl_off"] + np.uint64(gimg.data_ptr())) * plan["gpl_mask"]
            wf = (plan["wf3t_off"] + np.uint64(p3[1].data_ptr() + p3[2])) * plan["wf3t_mask"]
            gpl_p, wf3t_p = gpl.ctypes.data, wf.ctypes.data
            if plan["dpl_mask"]:
                dpl = ctypes.c_void_p(gimg.data_ptr() + int(plan["dpl_off"]))
        _lib.check(_lib.load().sh_stack_backward(
            n, self._native_steps(), _lib.ptr(x), _LAYOUT_ID[in_layout], rows0, c0, B, acts.ctypes.data, _lib.ptr(g),
            _LAYOUT_ID[out_layout], self._ptr_array(weights), gin.ctypes.data, ctypes.c_void_p(int(wbase) + 4 * plan["dpre_last_off"]),
            wt.ctypes.data, ws.ctypes.data, plan["ws_bytes"].ctypes.data, dW.ctypes.data, db.ctypes.data, 1 if need_x_grad else 0,
            _lib.mma_id(mma), gpl_p, dpl, wf3t_p, inpl_p, 2 if inpl_p is not None else 1, _lib.stream_ptr()), "sh_stack_backward")
        grads = {}
        for j, shp in enumerate(plan["shapes"]):
            if shp is None:
                continue
            o = int(plan["dW_off_f"][j])
            dWj = flat[o:o + shp[0] * shp[1]].view(shp)
            o = int(plan["db_off_f"][j])
            grads[j] = (dWj, flat[o:o + shp[0]] if need_bias[j] else None)
        return gx, grads

    # ------------------------------------------------------------------ bf16 compute path (BASELINE config 3)
    def _plan_bf16(self, B: int, c0: int, x_is_f32: bool, out_is_f32: bool):
        """Arena offsets in BYTES for the bf16 path: activations / gradients between steps are bf16, the fragment-ordered
        working copies of the conv weights and the fp32 partial slabs of the weight gradients live in the same arenas."""
        key = ("bf16", B, c0, x_is_f32, out_is_f32)
        plans = self.__dict__.setdefault("_plans", {})
        if key in plans:
            return plans[key]
        lib = _lib.load()
        al = lambda nbytes: (int(nbytes) + 255) // 256 * 256          # noqa: E731
        n = len(self.steps)
        cin_of, c = [], c0
        for st in self.steps:
            cin_of.append(c)
            c = st.cout if st.kind == "conv" else c
        out_rows = [st.R if st.kind == "conv" else st.csr.rows for st in self.steps]
        out_ch = [st.cout if st.kind == "conv" else cin_of[i] for i, st in enumerate(self.steps)]
        f_off, wf_off, wf_mask, o = np.zeros(n, dtype=np.uint64), np.zeros(n, dtype=np.uint64), np.zeros(n, dtype=np.uint64), 0
        for i in range(n - 1):
            if self._extends(i):
                f_off[i] = f_off[i - 1]
                continue
            f_off[i] = o
            o += al(self._buffer_rows(i) * B * out_ch[i] * 2)
        for i, st in enumerate(self.steps):
            if st.kind == "conv":
                wf_off[i], wf_mask[i] = o, 1
                o += al(lib.sh_conv_wfrag_bytes(st.S, st.cin, st.cout))
        f_total = o
        last = self.steps[-1]
        o = 0
        if last.kind == "conv":
            o += al((last.R + last.n_extra) * B * last.cout * (4 if out_is_f32 else 2))
        gin_size = [0] * n
        for i in range(1, n):
            st, prev = self.steps[i], self.steps[i - 1]
            rows_in = st.n_in if st.kind == "conv" else st.csr.cols
            gin_size[i] = (rows_in + (prev.n_extra if prev.kind == "conv" else 0)) * B * cin_of[i] * 2
        region = [al(max([gin_size[i] for i in range(1, n) if i % 2 == par] + [0])) for par in (0, 1)]
        region_off = [o, o + region[0]]
        o += region[0] + region[1]
        g_off, g_mask = np.zeros(n, dtype=np.uint64), np.zeros(n, dtype=np.uint64)
        for i in range(1, n):
            g_off[i], g_mask[i] = region_off[i % 2], 1
        wt_off, wt_mask = np.zeros(n, dtype=np.uint64), np.zeros(n, dtype=np.uint64)
        ws_off, ws_mask, ws_bytes = np.zeros(n, dtype=np.uint64), np.zeros(n, dtype=np.uint64), np.zeros(n, dtype=np.uint64)
        for i, st in enumerate(self.steps):
            if st.kind != "conv":
                continue
            wt_off[i], wt_mask[i] = o, 1
            o += al(lib.sh_conv_wfrag_bytes(st.S, st.cout, st.cin))
            nb = int(lib.sh_spiral_conv_bwd_wgt_workspace_bf16(B, st.R, st.S, st.cin, st.cout))
            ws_off[i], ws_mask[i], ws_bytes[i] = o, 1, nb
            o += al(nb)
        b_total = o
        npar = 1 + max([st.param for st in self.steps if st.kind == "conv"], default=-1)
        dW_off, db_off, shapes, o = np.zeros(npar, dtype=np.uint64), np.zeros(npar, dtype=np.uint64), [None] * npar, 0
        for st in self.steps:
            if st.kind != "conv":
                continue
            dW_off[st.param] = o
            o += _round(st.cout * st.S * st.cin)
            db_off[st.param] = o
            o += _round(st.cout)
            shapes[st.param] = (st.cout, st.S * st.cin)
        plan = dict(f_off=f_off, wf_off=wf_off, wf_mask=wf_mask, f_total=f_total, g_off=g_off, g_mask=g_mask, wt_off=wt_off,
                    wt_mask=wt_mask, ws_off=ws_off, ws_mask=ws_mask, ws_bytes=ws_bytes, b_total=b_total, dW_off=dW_off * 4,
                    db_off=db_off * 4, dW_off_f=dW_off, db_off_f=db_off, shapes=shapes, p_total=o, npar=npar, out_rows=out_rows,
                    out_ch=out_ch)
        plans[key] = plan
        return plan

    @staticmethod
    def _io_dims(x, layout):
        B = x.shape[0] if layout == "bm" else x.shape[1]
        rows = x.shape[1] if layout == "bm" else x.shape[0]
        return B, rows, x.shape[2]

    def native_forward_bf16(self, x, in_layout, out_layout, out_dtype, weights, biases, wf=None):
        """x: bf16, or fp32 with 3 channels.  -> (output of type out_dtype, byte arena holding the inner activations).
        wf: a WFrags holding this stack's converted weights (then no conversion launch), or None."""
        B, rows0, c0 = self._io_dims(x, in_layout)
        if not (x.is_contiguous() and x.dtype in (torch.float32, torch.bfloat16)):
            raise ValueError("expected a contiguous fp32 / bf16 3-D tensor, got %s %s" % (tuple(x.shape), x.dtype))
        if x.numel() >= 2 ** 32:
            raise RuntimeError("semantichuman_amd: gathered tensors are addressed with 32-bit element offsets; "
                               "%d elements is too large - split the batch" % x.numel())
        plan = self._plan_bf16(B, c0, x.dtype == torch.float32, out_dtype == torch.float32)
        n = len(self.steps)
        arena = torch.empty(max(256, plan["f_total"]), dtype=torch.uint8, device=x.device)
        rows_o, ch_o = plan["out_rows"][-1], plan["out_ch"][-1]
        out = torch.empty((B, rows_o, ch_o) if out_layout == "bm" else (rows_o, B, ch_o), dtype=out_dtype, device=x.device)
        base = np.uint64(arena.data_ptr())
        outs = plan["f_off"] + base
        outs[n - 1] = out.data_ptr()
        wfp = wf.fwd[id(self)] if wf is not None else (plan["wf_off"] + base) * plan["wf_mask"]
        _lib.check(_lib.load().sh_stack_forward_bf16(
            n, self._native_steps(), _lib.ptr(x), ops.dtype_id(x), _LAYOUT_ID[in_layout], rows0, c0, B, self._ptr_array(weights),
            self._ptr_array(biases), wfp.ctypes.data, 1 if wf is not None else 0, outs.ctypes.data, ops.dtype_id(out),
            _LAYOUT_ID[out_layout], _lib.stream_ptr()), "sh_stack_forward_bf16")
        return out, arena

    def native_backward_bf16(self, x, in_layout, out_layout, arena, out, g, weights, need_x_grad, need_bias, wf=None):
        B, rows0, c0 = self._io_dims(x, in_layout)
        plan = self._plan_bf16(B, c0, x.dtype == torch.float32, out.dtype == torch.float32)
        n = len(self.steps)
        dev = x.device
        work = torch.empty(max(256, plan["b_total"]), dtype=torch.uint8, device=dev)
        flat = torch.empty(max(1, plan["p_total"]), dtype=torch.float32, device=dev)
        gx = torch.empty_like(x) if need_x_grad else None
        abase, wbase, fbase = np.uint64(arena.data_ptr()), np.uint64(work.data_ptr()), np.uint64(flat.data_ptr())
        acts = plan["f_off"] + abase
        acts[n - 1] = out.data_ptr()
        gin = (plan["g_off"] + wbase) * plan["g_mask"]
        gin[0] = gx.data_ptr() if need_x_grad else 0
        ready = wf is not None and wf.bwd is not None
        wt = wf.bwd[id(self)] if ready else (plan["wt_off"] + wbase) * plan["wt_mask"]
        ws = (plan["ws_off"] + wbase) * plan["ws_mask"]
        dW = plan["dW_off"] + fbase
        assert len(need_bias) == plan["npar"] == len(weights)
        db = (plan["db_off"] + fbase) * np.array([1 if nb else 0 for nb in need_bias], dtype=np.uint64)
        _lib.check(_lib.load().sh_stack_backward_bf16(
            n, self._native_steps(), _lib.ptr(x), ops.dtype_id(x), _LAYOUT_ID[in_layout], rows0, c0, B, acts.ctypes.data, _lib.ptr(g),
            ops.dtype_id(out), _LAYOUT_ID[out_layout], self._ptr_array(weights), gin.ctypes.data, ops.dtype_id(x),
            ctypes.c_void_p(int(wbase)), wt.ctypes.data, 1 if ready else 0, ws.ctypes.data, plan["ws_bytes"].ctypes.data, dW.ctypes.data,
            db.ctypes.data, 1 if need_x_grad else 0, _lib.stream_ptr()), "sh_stack_backward_bf16")
        grads = {}
        for j, shp in enumerate(plan["shapes"]):
            if shp is None:
                continue
            o = int(plan["dW_off_f"][j])
            dWj = flat[o:o + shp[0] * shp[1]].view(shp)
            o = int(plan["db_off_f"][j])
            grads[j] = (dWj, flat[o:o + shp[0]] if need_bias[j] else None)
        return gx, grads

    def _side_stream(self, dev):
        s = getattr(self, "_side", None)
        if s is None or s.device != dev:
            s = self._side = torch.cuda.Stream(device=dev)
        return s

    # ------------------------------------------------------------------ forward
    def run_forward(self, x, in_layout, out_layout, weights, biases, keep: bool):
        """-> (output, [activation of every step]) ; activations are only kept when `keep`."""
        B = x.shape[0] if in_layout == "bm" else x.shape[1]
        cur, cur_layout = x, in_layout
        acts = []
        last = len(self.steps) - 1
        for i, st in enumerate(self.steps):
            lay = out_layout if i == last else "vm"
            if st.kind == "conv":
                y = ops.alloc(B, st.R, st.cout, lay, x.device, extra_rows=self._buffer_rows(i) - st.R)
                ops.spiral_conv_fwd(cur, cur_layout, st.dev["table"], weights[st.param], biases[st.param], y, lay,
                                    st.R, st.S, st.act, st.zero_row)
            elif st.extend:                              # the blended rows go behind the rows the previous conv wrote
                y = cur
                ops.spmm(st.dev["m"], cur, "vm", cur[st.csr.cols:], "vm", st.csr_fwd.rows)
            else:
                C = cur.shape[2]
                y = ops.alloc(B, st.csr.rows, C, lay, x.device)
                ops.spmm(st.dev["m"], cur, cur_layout, y, lay, st.csr.rows)
            if keep:
                acts.append(y)
            cur, cur_layout = y, lay
        return cur, acts

    # ------------------------------------------------------------------ backward
    def run_backward(self, x, in_layout, out_layout, acts, g, weights, need_x_grad: bool, need_bias):
        """g: gradient w.r.t. the stack output (layout out_layout).
        -> (grad_x or None, {param: (dW, db)})"""
        steps = self.steps
        last = len(steps) - 1
        B = x.shape[0] if in_layout == "bm" else x.shape[1]
        dev = x.device
        grads = {}
        # Weight-gradient kernels only read (dpre_i, input_i) and are needed by nobody until the
        # optimizer, so they run on a side stream concurrently with the backward-data chain on the
        # main stream.  Each launch of this net has limited parallelism (a few hundred to a few
        # thousand workgroups with long, barrier-paced tiles); two independent kernel streams fill
        # each other's tails and stalls.  Under hipGraph capture this becomes two parallel branches.
        main = torch.cuda.current_stream(dev)
        side = self._side_stream(dev) if OVERLAP_WGRAD else None
        keep_alive = []          # buffers the side stream still reads must outlive the join below

        def in_of(i):
            return (x, in_layout) if i == 0 else (acts[i - 1], "vm")

        # all weight transposes of the stack in one launch (weights are known up front)
        conv_idx = [i for i, st in enumerate(steps) if st.kind == "conv" and (i > 0 or need_x_grad)]
        wts = dict(zip(conv_idx, ops.weight_transpose_multi([weights[steps[i].param] for i in conv_idx],
                                                            [(steps[i].S, steps[i].cin, steps[i].cout) for i in conv_idx]))) \
            if conv_idx else {}
        jobs = []                # deferred weight-gradient reductions (one launch at the end)

        # gradient entering the last step
        st = steps[last]
        if st.kind == "conv":
            dpre = ops.alloc(B, st.R, st.cout, "vm", dev, extra_rows=st.n_extra)
            ops.act_backward(g, out_layout, acts[last], out_layout, dpre, "vm", st.R, st.act, st.zero_row)
            cur, cur_layout = dpre, "vm"
        else:
            cur, cur_layout = g, out_layout

        for i in range(last, -1, -1):
            st = steps[i]
            inp, inp_layout = in_of(i)
            prev = steps[i - 1] if i > 0 else None
            want_in = i > 0 or need_x_grad
            # where the input gradient goes: previous conv's dpre buffer (vm, maybe extra rows),
            # previous spmm's output gradient (vm), or the caller's x gradient (in_layout)
            g_in = None
            if want_in:
                rows_in = st.n_in if st.kind == "conv" else st.csr.cols
                cin = st.cin if st.kind == "conv" else cur.shape[2]
                if i == 0:
                    g_in, g_layout = ops.alloc(B, rows_in, cin, in_layout, dev), in_layout
                else:
                    extra = prev.n_extra if prev.kind == "conv" else 0
                    g_in, g_layout = ops.alloc(B, rows_in, cin, "vm", dev, extra_rows=extra), "vm"
            ep = dict(yprev=None, yp_layout="vm", act_prev=0, zero_row=-1)
            if prev is not None and prev.kind == "conv":
                ep = dict(yprev=acts[i - 1], yp_layout="vm", act_prev=prev.act, zero_row=prev.zero_row)

            if st.kind == "conv":
                # rows referenced more than once per (input row, position) are summed into the extra rows of the
                # dpre buffer before backward-data (see mesh_ops.TransposedTable)
                n1, n2 = (st.tt.n1, st.tt.n2) if want_in else (0, 0)

                def presum():
                    if n1:
                        ops.spmm(st.dev["sum1"], cur, "vm", cur[st.R:], "vm", n1)
                    if n2:
                        ops.spmm(st.dev["sum2"], cur, "vm", cur[st.R + n1:], "vm", n2)
                presum_side = OVERLAP_PRESUM and side is None and (n1 or n2)
                if presum_side:
                    ps = self._side_stream(dev)
                    ps.wait_stream(main)                       # dpre_i is complete on main
                    with torch.cuda.stream(ps):
                        presum()                               # writes rows >= R; the weight gradient reads rows < R
                # a 16 -> 3 channel layer takes the role-swapped weight gradient (wgrad_thin.hip), which reads the
                # pre-summed rows: same order as sh_stack_backward
                thin = (want_in and side is None and not presum_side and cur_layout == "vm" and inp_layout == "vm"
                        and st.R == st.n_in and ops.wgrad_thin_ok(B, st.n_in, st.S, st.cin, st.cout, cur.dtype))
                thin_dx = thin and g_layout == "vm" and (ep["yprev"] is None or ep["yprev"] is inp)
                if thin:
                    presum()
                    job = ops.spiral_conv_bwd_wgt_thin_deferred(
                        cur, inp, st.dev["table_t"], st.R, st.S, st.cin, st.cout, want_bias=need_bias[st.param],
                        weight=weights[st.param], dx=g_in if thin_dx else None,
                        act_prev=ep["act_prev"] if ep["yprev"] is not None else 0, zero_prev=ep["zero_row"])
                else:
                    if side is not None:
                        side.wait_stream(main)                 # dpre_i (and input_i) are complete on main
                        keep_alive.append(cur)
                    with torch.cuda.stream(side if side is not None else main):
                        job = ops.spiral_conv_bwd_wgt_deferred(cur, cur_layout, inp, inp_layout, st.dev["table"], st.R, st.S,
                                                               st.cin, st.cout, want_bias=need_bias[st.param])
                jobs.append(job)
                grads[st.param] = (job["dW"], job["db"])
                if want_in:
                    if presum_side:
                        main.wait_stream(ps)
                    elif not thin:
                        presum()
                if want_in and not (thin and thin_dx):
                    ops.spiral_conv_bwd_data(cur, cur_layout, st.dev["table_t"], wts[i], g_in, g_layout,
                                             ep["yprev"], ep["yp_layout"], ep["act_prev"], ep["zero_row"],
                                             st.n_in, st.S, st.cin, st.cout)
            elif want_in:
                ops.spmm(st.dev["mt"], cur, cur_layout, g_in, g_layout, st.csr.cols, yprev=ep["yprev"],
                         yp_layout=ep["yp_layout"], act_prev=ep["act_prev"], zero_row=ep["zero_row"])
            if want_in:
                cur, cur_layout = g_in, g_layout
        with torch.cuda.stream(side if side is not None else main):
            for k in range(0, len(jobs), 16):
                ops.spiral_conv_bwd_wgt_reduce(jobs[k:k + 16])
        if side is not None:
            main.wait_stream(side)                             # join: gradients are consumed on main
            for dW, db in grads.values():
                dW.record_stream(main)
                if db is not None:
                    db.record_stream(main)
            del keep_alive
        return (cur if need_x_grad else None), grads


def convert_p3_frags(items, with_backward: bool):
    """ONE sh_conv_wfrag3_prep_multi launch for the three-plane weight fragments of several stacks.
    items: [(stack, plan, weights, buffer, byte offset of the stack's fragments in the buffer)]."""
    lib = _lib.load()
    w_p, o_p, S, Ci, Co, tr = [], [], [], [], [], []
    for stack, plan, weights, buf, base in items:
        for i, t, off in plan["wf3_jobs"]:
            if t and not with_backward:
                continue
            st = stack.steps[i]
            w = weights[st.param]
            if not (w.is_cuda and w.dtype == torch.float32 and w.is_contiguous()):
                raise RuntimeError("semantichuman_amd: conv weights must be contiguous fp32 HIP tensors")
            w_p.append(w.data_ptr()); o_p.append(buf.data_ptr() + base + off)
            S.append(st.S); Ci.append(st.cin); Co.append(st.cout); tr.append(t)
    n = len(w_p)
    if not n:
        return
    arr = lambda vals, ct: (ct * n)(*vals)                              # noqa: E731
    _lib.check(lib.sh_conv_wfrag3_prep_multi(n, arr(w_p, ctypes.c_void_p), arr(o_p, ctypes.c_void_p), arr(S, ctypes.c_int),
                                             arr(Ci, ctypes.c_int), arr(Co, ctypes.c_int), arr(tr, ctypes.c_int), _lib.stream_ptr()),
               "sh_conv_wfrag3_prep_multi")


def prepare_p3_frags(stacks_convs_c0, B: int, with_backward: bool):
    """Three-plane form (SH_MMA_PLANES3): the weight fragments of several stacks - forward and, when a backward pass can
    follow, backward-data operands - converted by ONE launch; each stack's next forward pass picks its share up
    (Stack._p3_prepare) instead of launching its own conversion.  stacks_convs_c0: [(stack, ModuleList of SpiralConv, c0)]."""
    if B % 16:
        return
    items, total = [], 0
    for stack, convs, c0 in stacks_convs_c0:
        plan = stack._plan(B, c0)
        if not plan["wf3_total"]:
            continue
        items.append((stack, plan, [m.conv.weight for m in convs], None, total))
        total += (int(plan["wf3_total"]) + 255) // 256 * 256
    if not items:
        return
    buf = torch.empty(total, dtype=torch.uint8, device=items[0][2][0].device)
    items = [(st, pl, ws, buf, off) for st, pl, ws, _, off in items]
    convert_p3_frags(items, with_backward)
    for st, pl, ws, _, off in items:
        st._p3_next = (buf, off, id(pl), with_backward)


class StackFunction(torch.autograd.Function):
    """autograd node for a whole Stack.  params = (w_0, b_0, w_1, b_1, ...) for the SpiralConv
    modules in ModuleList order (b_j may be None)."""

    @staticmethod
    def forward(ctx, stack: Stack, in_layout: str, out_layout: str, x, *params):
        weights, biases = list(params[0::2]), list(params[1::2])
        x = x.contiguous()
        need = any(ctx.needs_input_grad[3:])
        ctx.stack, ctx.layouts = stack, (in_layout, out_layout)
        ctx.has_bias = [b is not None for b in biases]
        ctx.native = NATIVE and not (OVERLAP_WGRAD or OVERLAP_PRESUM)
        # the arithmetic form of the node: what the caller's default says NOW; the backward pass (autograd's thread, later)
        # runs in the same form whatever the default is by then
        ctx.mma = _lib.get_f32_mma_mode()
        if ctx.native:
            out, arena, ctx.p3 = stack.native_forward(x, in_layout, out_layout, weights, biases, ctx.mma, with_backward=need or ctx.needs_input_grad[3])
            ctx.save_for_backward(x, out, arena, *weights)
            return out
        out, acts = stack.run_forward(x, in_layout, out_layout, weights, biases, keep=need)
        ctx.acts = acts[:-1]                 # internal activations; the output itself goes through
        ctx.save_for_backward(x, out, *weights)   # save_for_backward (no ctx <-> output cycle)
        return out

    @staticmethod
    def backward(ctx, g):
        stack = ctx.stack
        in_layout, out_layout = ctx.layouts
        g = g.contiguous()
        # forward args: (stack, in_layout, out_layout, x, w_0, b_0, w_1, b_1, ...)
        need_bias = [hb and ctx.needs_input_grad[5 + 2 * j] for j, hb in enumerate(ctx.has_bias)]
        if ctx.native:
            x, out, arena, *weights = ctx.saved_tensors
            gx, grads = stack.native_backward(x, in_layout, out_layout, arena, out, g, weights, ctx.needs_input_grad[3], need_bias,
                                              ctx.mma, ctx.p3)
            ctx.p3 = None
        else:
            x, out, *weights = ctx.saved_tensors
            was = _lib.get_f32_mma_mode()
            _lib.set_f32_mma_mode(ctx.mma)
            try:
                gx, grads = stack.run_backward(x, in_layout, out_layout, ctx.acts + [out], g, weights,
                                               ctx.needs_input_grad[3], need_bias)
            finally:
                _lib.set_f32_mma_mode(was)
            ctx.acts = None
        res = [None, None, None, gx]
        for j in range(len(weights)):
            dW, db = grads.get(j, (None, None))
            res += [dW, db]
        return tuple(res)


class WFrags:
    """bf16 fragment-ordered working copies of the conv weights of one or more stacks, written by ONE conversion launch:
    the forward operand of every conv step and, when a backward pass will follow, the backward-data operand too.  The
    buffer is a fresh allocation per call (an autograd graph that still refers to an older one keeps it alive), so the
    copies always belong to the weights as they were when the forward pass ran."""

    def __init__(self, stacks_and_weights, with_backward: bool):
        lib = _lib.load()
        jobs, total = [], 0                                              # (stack, step, weight, transpose, byte offset)
        al = lambda n: (n + 255) // 256 * 256                            # noqa: E731
        for stack, weights in stacks_and_weights:
            for i, st in enumerate(stack.steps):
                if st.kind != "conv":
                    continue
                for tr in ((0, 1) if with_backward else (0,)):
                    nb = lib.sh_conv_wfrag_bytes(st.S, st.cout if tr else st.cin, st.cin if tr else st.cout)
                    jobs.append((stack, i, st, weights[st.param], tr, total))
                    total += al(nb)
        dev = stacks_and_weights[0][1][0].device
        self.buf = torch.empty(max(256, total), dtype=torch.uint8, device=dev)
        base = self.buf.data_ptr()
        self.fwd = {id(stack): np.zeros(len(stack.steps), dtype=np.uint64) for stack, _ in stacks_and_weights}
        self.bwd = {id(stack): np.zeros(len(stack.steps), dtype=np.uint64) for stack, _ in stacks_and_weights} if with_backward else None
        for stack, i, st, w, tr, off in jobs:
            (self.bwd if tr else self.fwd)[id(stack)][i] = base + off
            if not (w.is_cuda and w.dtype == torch.float32 and w.is_contiguous()):
                raise RuntimeError("semantichuman_amd: conv weights must be contiguous fp32 HIP tensors (bf16 working copies are made from them)")
        n = len(jobs)
        arr = lambda vals, ct: (ct * n)(*vals)                           # noqa: E731
        _lib.check(lib.sh_conv_wfrag_prep_multi(
            n, arr([j[3].data_ptr() for j in jobs], ctypes.c_void_p), arr([base + j[5] for j in jobs], ctypes.c_void_p),
            arr([j[2].S for j in jobs], ctypes.c_int), arr([j[2].cin for j in jobs], ctypes.c_int),
            arr([j[2].cout for j in jobs], ctypes.c_int), arr([j[4] for j in jobs], ctypes.c_int), _lib.stream_ptr()),
            "sh_conv_wfrag_prep_multi")


def prepare_wfrags(stacks_and_convs):
    """One conversion launch for all the given (stack, ModuleList of SpiralConv) pairs; both operand orientations when a
    backward pass can follow."""
    pairs = [(stack, [m.conv.weight for m in convs]) for stack, convs in stacks_and_convs]
    with_backward = torch.is_grad_enabled() and any(w.requires_grad for _, ws in pairs for w in ws)
    return WFrags(pairs, with_backward)


class StackFunctionBF16(torch.autograd.Function):
    """autograd node of a whole Stack on the bf16 path: fp32 master parameters in, fp32 parameter gradients out;
    x is bf16 (or fp32 xyz), the output has `out_dtype`, gradients flow in the tensors' own types."""

    @staticmethod
    def forward(ctx, stack: Stack, in_layout: str, out_layout: str, out_dtype, wf, x, *params):
        weights, biases = list(params[0::2]), list(params[1::2])
        x = x.contiguous()
        ctx.stack, ctx.layouts, ctx.wf = stack, (in_layout, out_layout), wf
        ctx.has_bias = [b is not None for b in biases]
        out, arena = stack.native_forward_bf16(x, in_layout, out_layout, out_dtype, weights, biases, wf)
        ctx.save_for_backward(x, out, arena, *weights)
        return out

    @staticmethod
    def backward(ctx, g):
        x, out, arena, *weights = ctx.saved_tensors
        in_layout, out_layout = ctx.layouts
        g = g.contiguous()
        # forward args: (stack, in_layout, out_layout, out_dtype, wf, x, w_0, b_0, ...)
        need_bias = [hb and ctx.needs_input_grad[7 + 2 * j] for j, hb in enumerate(ctx.has_bias)]
        gx, grads = ctx.stack.native_backward_bf16(x, in_layout, out_layout, arena, out, g, weights, ctx.needs_input_grad[5], need_bias,
                                                   ctx.wf)
        res = [None, None, None, None, None, gx]
        for j in range(len(weights)):
            dW, db = grads.get(j, (None, None))
            res += [dW, db]
        return tuple(res)


def run_stack_bf16(stack: Stack, x, in_layout, out_layout, out_dtype, convs, wf=None):
    """bf16 compute path of `run_stack`: x bf16 (or fp32 with 3 channels), output of type out_dtype.  wf: a WFrags made by
    `prepare_wfrags` for (at least) this stack in this forward pass; None = the stack converts its own weights."""
    if not x.is_cuda:
        raise RuntimeError("semantichuman_amd: input is on %s; the spiral-convolution kernels run on a HIP device "
                           "only (there is no CPU fallback)" % x.device)
    if stack.device is None or stack.device != x.device:
        raise RuntimeError("semantichuman_amd: model tables are on %s but the input is on %s - move the module with "
                           ".to(device)" % (stack.device, x.device))
    params = []
    for m in convs:
        params += [m.conv.weight, m.conv.bias]
    if wf is None:
        wf = prepare_wfrags([(stack, convs)])
    if torch.is_grad_enabled() and (x.requires_grad or any(p is not None and p.requires_grad for p in params)):
        return StackFunctionBF16.apply(stack, in_layout, out_layout, out_dtype, wf, x, *params)
    return stack.native_forward_bf16(x.contiguous(), in_layout, out_layout, out_dtype, params[0::2], params[1::2], wf)[0]


def run_stack(stack: Stack, x, in_layout, out_layout, convs):
    """convs: the ModuleList of SpiralConv modules this stack's ConvSteps index into."""
    if not x.is_cuda:
        raise RuntimeError("semantichuman_amd: input is on %s; the spiral-convolution kernels run on a HIP device "
                           "only (there is no CPU fallback)" % x.device)
    if stack.device is None or stack.device != x.device:
        raise RuntimeError("semantichuman_amd: model tables are on %s but the input is on %s - move the module with "
                           ".to(device)" % (stack.device, x.device))
    params = []
    for m in convs:
        params += [m.conv.weight, m.conv.bias]
    if torch.is_grad_enabled() and (x.requires_grad or any(p is not None and p.requires_grad for p in params)):
        return StackFunction.apply(stack, in_layout, out_layout, x, *params)
    if NATIVE:
        return stack.native_forward(x.contiguous(), in_layout, out_layout, params[0::2], params[1::2], _lib.get_f32_mma_mode())[0]
    out, _ = stack.run_forward(x.contiguous(), in_layout, out_layout, params[0::2], params[1::2], keep=False)
    return out

"""Data-parallel training over the GPUs of one node: one process per GPU, RCCL over xGMI.

The reference is single-process / single-GPU (main.py:197-201; no torch.distributed anywhere).
The training path shards on the batch axis only: every mesh is independent through
forward and backward, the only cross-sample coupling is the mean in the losses and the sum in
the weight gradients.  So each rank holds a full replica, works on its own batch shard, and the
ONE collective per step is the gradient all-reduce (sum, pre-scaled by 1/world = mean).

Design for xGMI (point-to-point links, no switch): few, large messages.
  * gradients are packed into contiguous buckets in BACKWARD order; the plain autoencoder's
    114 MB of gradients are 99 % two latent FC matrices (2 x 56.6 MB), which become ready in
    the middle of backward (decoder stack -> fc_latent_dec -> fc_latent_enc -> encoder stack);
  * the all-reduce of a large gradient is launched asynchronously the moment it has been produced
    (post-accumulate-grad hook), so the FC messages travel while the encoder-stack backward kernels
    run; the packed small gradients (1 % of the bytes) follow in `finish()`, which waits and re-points
    .grad at the reduced bucket views.

Backend "nccl" is RCCL on ROCm; on CPU (tests) the same code runs over gloo.
"""
from __future__ import annotations

from typing import List

import torch
import torch.distributed as dist


class GradBucket:
    """Either a packed bucket of small gradients (copied into `flat`), or - `inplace` - ONE large gradient that is
    all-reduced where autograd left it: the two 56.6 MB FC gradients would otherwise cost a 227 MB copy pass per step."""

    def __init__(self, params: List[torch.nn.Parameter], inplace: bool = False):
        self.params = params
        self.inplace = inplace
        self.numel = sum(p.numel() for p in params)
        p0 = params[0]
        self.flat = None if inplace else torch.zeros(self.numel, dtype=p0.dtype, device=p0.device)
        self.views, o = [], 0
        if not inplace:
            for p in params:
                self.views.append(self.flat[o:o + p.numel()].view_as(p))
                o += p.numel()
        self.work = None
        self.buf = None          # the tensor in flight
        self.full = None         # the fp32 gradient while a reduced-precision copy of it is in flight


class GradientAllReducer:
    """Bucketed, overlapped gradient averaging for a replicated nn.Module."""

    def __init__(self, module: torch.nn.Module, process_group=None, bucket_cap_mb: float = 64.0, overlap: bool = True,
                 inplace_min_mb: float = 16.0, force_collectives: bool = False, large_message_dtype=None,
                 average_in_collective: bool = True, shard_large: bool = False):
        """force_collectives: issue the collectives even in a world of one rank (exercises the RCCL path on a 1-GPU box).
        large_message_dtype: e.g. torch.bfloat16 - the large in-place gradients travel in that type (half the xGMI
        bytes: 57 MB instead of 114 MB for the plain autoencoder) and are converted back into the
        fp32 gradient after the collective.  Off by default: fp32 training exchanges fp32 gradients.
        average_in_collective=False: never use ncclAvg (sum of pre-scaled gradients; bench.py's safe mode).
        shard_large=True: the SHARDED update of the large parameters - their gradients are
        reduce-SCATTERED (each rank receives the averaged 1 / world slice), the optimizer updates only that slice
        (`optimizer_params()` hands it the slices as parameters of their own, so its moments are 1 / world the size), and
        `gather_weights()` after `optimizer.step()` all-gathers the updated slices into every replica: the same bytes over
        the links as the all-reduce, 1 / world of Adam's HBM traffic and state for 99 % of this model's parameters.  The
        elementwise update of a slice is the update of those elements, so the weights are the all-reduce path's."""
        self.large_dtype = large_message_dtype
        self.group = process_group
        self.world = dist.get_world_size(process_group) if dist.is_initialized() else 1
        self.active = self.world > 1 or (force_collectives and dist.is_initialized())
        self.overlap = overlap
        # RCCL averages in the collective (ncclAvg): no pre-scaling pass.  gloo (CPU tests) only sums.
        self.avg = average_in_collective and dist.is_initialized() and dist.get_backend(process_group) == "nccl"
        params = [p for p in module.parameters() if p.requires_grad]
        if self.avg and self.active and params:
            try:                                               # every rank builds its reducer: a collective probe is safe
                probe = torch.ones(1, dtype=params[0].dtype, device=params[0].device)
                dist.all_reduce(probe, op=dist.ReduceOp.AVG, group=process_group)
                self.avg = bool(probe.item() == 1.0)
            except (RuntimeError, ValueError):
                self.avg = False                               # older collectives library: sum of pre-scaled gradients
        # backward produces gradients roughly in reverse registration order
        cap = int(bucket_cap_mb * 1024 * 1024)
        big = int(inplace_min_mb * 1024 * 1024)
        self.buckets: List[GradBucket] = []
        cur, cur_bytes = [], 0
        small: List[GradBucket] = []
        for p in reversed(params):
            nbytes = p.numel() * p.element_size()
            if nbytes >= big:                                  # large gradient: reduced in place, its own message
                self.buckets.append(GradBucket([p], inplace=True))
                continue
            if cur and cur_bytes + nbytes > cap:
                small.append(GradBucket(cur))
                cur, cur_bytes = [], 0
            cur.append(p)
            cur_bytes += nbytes
        if cur:
            small.append(GradBucket(cur))
        self.buckets += small                                  # all small gradients together (one message for this net)
        # Only the large in-place messages are launched from hooks (they are what there is to overlap: 99 % of the
        # bytes, ready in the middle of backward).  The small gradients are packed by ONE multi-tensor copy and go out as
        # ONE message in finish(): packed per parameter they were 20 copy launches (a ~5 us slot each on the compute
        # stream, profiles/r01_reducer_timeline.txt) and three messages for 1 % of the bytes.
        self._where = {b.params[0]: b for b in self.buckets if b.inplace}
        # sharded update: slice [rank * n / world, (rank + 1) * n / world) of every large parameter, as a leaf of its own that
        # shares the parameter's storage (the optimizer writes the replica's weight in place)
        self.shard_large = bool(shard_large) and self.active
        self.shards = {}
        self.rank = dist.get_rank(process_group) if dist.is_initialized() else 0
        if self.shard_large:
            if self.large_dtype is not None:
                raise ValueError("shard_large and large_message_dtype are exclusive")
            for b in self.buckets:
                if not b.inplace:
                    continue
                p = b.params[0]
                if p.numel() % self.world or not p.is_contiguous():
                    raise ValueError("shard_large: a %s parameter does not split evenly over %d ranks" % (tuple(p.shape), self.world))
                n = p.numel() // self.world
                sh = torch.nn.Parameter(p.data.view(-1)[self.rank * n:(self.rank + 1) * n], requires_grad=True)
                self.shards[p] = sh
                b.shard_grad = torch.zeros(n, dtype=p.dtype, device=p.device)
            self._rs_ok = None                                     # does this backend have reduce_scatter_tensor?
        self._hooks = []
        if self.active and overlap:
            for p in self._where:
                self._hooks.append(p.register_post_accumulate_grad_hook(self._on_grad))
        self._armed = False

    @property
    def message_bytes(self) -> int:
        return sum(b.numel * b.params[0].element_size() for b in self.buckets)

    def prepare(self):
        """Call before backward (after zero_grad)."""
        for b in self.buckets:
            b.work = None
        self._armed = True

    def optimizer_params(self):
        """What the optimizer is built over: the small parameters as they are and, with shard_large, this rank's slice of
        every large one instead of the parameter itself."""
        out = []
        for b in self.buckets:
            for p in b.params:
                out.append(self.shards.get(p, p))
        return out

    def gather_weights(self):
        """shard_large: after optimizer.step(), every replica's large parameters = the concatenation of all ranks' updated
        slices (the slice leaves share the parameters' storage, so this rank's part is already in place)."""
        if not self.shard_large:
            return
        from . import shadow
        for p, sh in self.shards.items():
            self._gather_one(p, sh)
            # bf16 compute path: the optimizer updated the SLICE (a parameter of its own - it has no working copy), and the
            # all-gather wrote the other ranks' slices through `p.data`, which torch's version counter does not see: the bf16
            # working copy of the whole parameter is re-converted here, in place (one streaming pass; a captured graph keeps
            # its address)
            shadow.refresh(p)

    def _gather_one(self, p, sh):
        flat = p.data.view(-1)
        n = sh.numel()
        if self._all_gather_into is not False:
            try:
                dist.all_gather_into_tensor(flat, sh.data.clone() if flat.device.type == "cpu" else sh.data, group=self.group)
                self._all_gather_into = True
                return
            except (RuntimeError, NotImplementedError):
                if self._all_gather_into is True:
                    raise
                self._all_gather_into = False
        parts = [flat[r * n:(r + 1) * n] for r in range(self.world)]
        dist.all_gather(parts, sh.data.clone(), group=self.group)

    _all_gather_into = None

    def _launch_sharded(self, b: GradBucket):
        """Reduce-scatter of a large gradient: this rank keeps the averaged slice the optimizer will consume."""
        p = b.params[0]
        if p.grad is None:
            p.grad = torch.zeros_like(p)
        g = (p.grad if p.grad.is_contiguous() else p.grad.contiguous()).view(-1)
        scale = 1.0 if self.avg else 1.0 / self.world
        if scale != 1.0:
            g.mul_(scale)
        op = dist.ReduceOp.AVG if self.avg else dist.ReduceOp.SUM
        b.buf = g
        if self._rs_ok is not False:
            try:
                b.work = dist.reduce_scatter_tensor(b.shard_grad, g, op=op, group=self.group, async_op=True)
                self._rs_ok = True
                return
            except (RuntimeError, NotImplementedError):
                if self._rs_ok is True:
                    raise
                self._rs_ok = False                              # e.g. gloo: all-reduce, then keep the slice (same values)
        b.work = dist.all_reduce(g, op=op, group=self.group, async_op=True)

    def _launch(self, b: GradBucket):
        if self.shard_large and b.inplace:
            return self._launch_sharded(b)
        scale = 1.0 if self.avg else 1.0 / self.world
        if b.inplace:
            p = b.params[0]
            if p.grad is None:
                p.grad = torch.zeros_like(p)
            b.buf = p.grad if p.grad.is_contiguous() else p.grad.contiguous()
            if self.large_dtype is not None and self.large_dtype != b.buf.dtype:
                b.full, b.buf = b.buf, b.buf.to(self.large_dtype)        # one conversion pass each way, half the message
        else:
            src, dst = [], []
            for p, v in zip(b.params, b.views):
                if p.grad is None:
                    v.zero_()
                elif p.grad.data_ptr() != v.data_ptr():
                    src.append(p.grad)
                    dst.append(v)
            if dst:
                torch._foreach_copy_(dst, src)                 # one multi-tensor launch, not one copy per parameter
            b.buf = b.flat
        if scale != 1.0:
            b.buf.mul_(scale)
        op = dist.ReduceOp.AVG if self.avg else dist.ReduceOp.SUM
        b.work = dist.all_reduce(b.buf, op=op, group=self.group, async_op=True)

    def _on_grad(self, p):
        if self._armed:
            self._launch(self._where[p])

    def finish(self):
        """Call after backward, before optimizer.step(): waits for the collectives and makes
        every .grad the averaged gradient."""
        if not self.active:
            return
        for b in self.buckets:
            if b.work is None:           # the packed small gradients; or overlap off / a parameter backward did not reach
                self._launch(b)
        for b in self.buckets:
            b.work.wait()
            if self.shard_large and b.inplace:
                p = b.params[0]
                if not self._rs_ok:
                    n = b.shard_grad.numel()
                    b.shard_grad.copy_(b.buf[self.rank * n:(self.rank + 1) * n])
                self.shards[p].grad = b.shard_grad
                p.grad = None                                      # the optimizer owns the slice, not the parameter
                b.buf = None
                continue
            if b.inplace:
                if b.full is not None:
                    b.full.copy_(b.buf)
                    b.buf, b.full = b.full, None
                b.params[0].grad = b.buf
            else:
                for p, v in zip(b.params, b.views):
                    p.grad = v
            b.buf = None
        self._armed = False

    def remove_hooks(self):
        for h in self._hooks:
            h.remove()
        self._hooks = []


def shard_batch(n_total: int, rank: int, world: int):
    """Even split of a global batch (config 3: 512 -> 64 per GPU).  Returns the slice of this
    rank; the global batch must divide evenly so every rank's loss mean has the same weight."""
    if n_total % world:
        raise ValueError("global batch %d is not divisible by world size %d" % (n_total, world))
    per = n_total // world
    return slice(rank * per, (rank + 1) * per)


def all_reduce_mean_scalar(t: torch.Tensor, group=None) -> torch.Tensor:
    """Average a scalar metric (loss, L1, L2) over ranks - logging / evaluation only."""
    if dist.is_initialized() and dist.get_world_size(group) > 1:
        t = t.clone()
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
        t /= dist.get_world_size(group)
    return t

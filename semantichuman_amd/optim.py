"""Adam on the HIP kernel.

`Adam` is a drop-in for `torch.optim.Adam(params, lr, betas, eps, weight_decay)` as the reference
constructs it (main.py:262: coupled L2 weight decay, no amsgrad) - same constructor keywords, same
`state_dict()` layout (`state[i] = {'step', 'exp_avg', 'exp_avg_sq'}` + `param_groups`), so the
reference's checkpoints load into it and its checkpoints load into torch's Adam, and
`torch.optim.lr_scheduler.StepLR` (main.py:263-264) drives it unchanged.

All parameters of a group are updated by one multi-tensor launch (sh_adam_step).  Learning rate and
step counts live in device memory, so a step captured in a hipGraph keeps counting its steps and reads
the learning rate from the device scalar at replay time.  That scalar is refreshed from
`param_groups[i]['lr']` by `step()` when it runs eagerly - a graph replay never re-enters `step()`, so
after `scheduler.step()` call `optimizer.sync_lr()` (outside the graph) before the next replay.

`overlap_backward(min_numel)`: the update is HBM-bound (7 passes over every parameter-sized array)
while the backward pass is MFMA-bound, so large parameters can be updated on a side stream as soon
as autograd has produced their gradient, underneath the rest of backward; `step()` then joins the
side stream and updates what is left.  The result is identical to updating everything in `step()`
(Adam's update of a parameter depends only on that parameter's own gradient).  Do not enable it
when gradients are modified between backward and step (clipping, all-reduce averaging).
"""
from __future__ import annotations

import ctypes

import torch

from . import _lib


class Adam(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0, amsgrad=False):
        if amsgrad:
            raise NotImplementedError("semantichuman_amd.optim.Adam: amsgrad is not used by the reference and not built")
        if not 0.0 <= lr or not 0.0 <= eps or not 0.0 <= weight_decay or not all(0.0 <= b < 1.0 for b in betas):
            raise ValueError("invalid Adam hyper-parameters lr=%r betas=%r eps=%r weight_decay=%r" % (lr, betas, eps, weight_decay))
        super().__init__(params, dict(lr=lr, betas=tuple(betas), eps=eps, weight_decay=weight_decay, amsgrad=False))
        self._lr_dev = {}            # id(group) -> (device scalar, value it holds)
        self._early = set()          # parameters already updated during this backward
        self._hooks = []
        self._side = None
        self._fused = []             # weak references to weights updated inside their weight-gradient kernel
        self._fused_done = []        # ... those that were, during this backward: step() advances their step counts
        self._release_foreign_fusions()

    def _release_foreign_fusions(self):
        """A new optimizer over parameters whose update ANOTHER optimizer applies inside their weight-gradient kernel replaces
        that optimizer: its registrations are dropped, or it would keep updating those weights during backward - with its own
        moments and a step count nobody advances - while this optimizer finds no gradient to apply."""
        import warnings
        from . import linear
        for group in self.param_groups:
            for p in group["params"]:
                ent = linear._FUSED_UPDATE.get(p.data_ptr()) if p.is_cuda else None
                if ent is None or ent[0]() is not p:
                    continue
                other = ent[1]()
                if other is not None and other is not self:
                    warnings.warn("semantichuman_amd.optim.Adam: a parameter of this optimizer was registered for the fused weight-"
                                  "gradient update of another optimizer; that registration is removed (call "
                                  "fuse_linear_weight_gradients on the new optimizer to fuse again)", stacklevel=3)
                    other._drop_fusion_of(p)
                linear._FUSED_UPDATE.pop(p.data_ptr(), None)

    def _drop_fusion_of(self, p):
        self._fused = [r for r in self._fused if r() is not None and r() is not p]

    def __del__(self):
        try:
            self.remove_fusion()
        except Exception:          # interpreter shutdown: modules may be gone
            pass

    # ---------------------------------------------------------------------------------- internals
    def _state_of(self, p):
        st = self.state[p]
        if not st:
            if not (p.is_cuda and p.dtype == torch.float32 and p.is_contiguous()):
                raise RuntimeError("semantichuman_amd.optim.Adam updates contiguous fp32 HIP parameters (got %s %s); there "
                                   "is no CPU path" % (p.device, p.dtype))
            st["step"] = torch.zeros((), dtype=torch.float32, device=p.device)
            st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
            st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
        elif not (st["step"].is_cuda and st["step"].dtype == torch.float32):        # state loaded from a torch.optim.Adam checkpoint
            st["step"] = torch.as_tensor(float(st["step"]), dtype=torch.float32, device=p.device)
        return st

    def _lr_tensor(self, group, device):
        cur = self._lr_dev.get(id(group))
        lr = float(group["lr"])
        if cur is None or cur[0].device != device:
            cur = (torch.full((), lr, dtype=torch.float32, device=device), lr)
        elif cur[1] != lr:
            cur[0].fill_(lr)                                   # scheduler changed it: refresh in stream order
            cur = (cur[0], lr)
        self._lr_dev[id(group)] = cur
        return cur[0]

    def sync_lr(self):
        """Copy every group's current `lr` into its device scalar (in stream order).  `step()` does this itself when it
        runs eagerly; a step that was captured into a hipGraph is replayed without re-entering Python, so a scheduler's
        change only reaches the captured kernel through this call."""
        for group in self.param_groups:
            ps = [p for p in group["params"] if p.is_cuda]
            if ps:
                self._lr_tensor(group, ps[0].device)

    def _apply(self, group, params, bump_only=()):
        params = [p for p in params if p.grad is not None]
        if not params and not bump_only:
            return
        n_upd = len(params)
        params = params + list(bump_only)      # (updated inside their weight-gradient kernel: only their step counts advance)
        n = len(params)
        arr = lambda: (ctypes.c_void_p * n)()                      # noqa: E731
        P, G, M, V, S, N = arr(), arr(), arr(), arr(), arr(), (ctypes.c_int64 * n)()
        W16, any16 = arr(), False           # bf16 working copies the bf16 compute path keeps of some parameters (shadow.py)
        from . import shadow
        keep = []
        for i, p in enumerate(params):
            if i >= n_upd:
                S[i], N[i] = self.state[p]["step"].data_ptr(), 0
                continue
            g = p.grad
            if g.is_sparse or g.dtype != torch.float32 or g.device != p.device:
                raise RuntimeError("semantichuman_amd.optim.Adam needs dense fp32 gradients on the parameter's device")
            if not g.is_contiguous():
                g = g.contiguous()
                keep.append(g)
            st = self._state_of(p)
            P[i], G[i], M[i], V[i], S[i] = p.data_ptr(), g.data_ptr(), st["exp_avg"].data_ptr(), st["exp_avg_sq"].data_ptr(), \
                st["step"].data_ptr()
            N[i] = p.numel()
            w16 = shadow.lookup(p)
            if w16 is not None:
                W16[i], any16 = w16.data_ptr(), True
        lr = self._lr_tensor(group, params[0].device)
        b1, b2 = group["betas"]
        if any16:       # the update rewrites the registered bf16 copies in the same kernel
            _lib.check(_lib.load().sh_adam_step_bf16(n, P, G, M, V, S, W16, N, _lib.ptr(lr), float(b1), float(b2), float(group["eps"]),
                                                     float(group["weight_decay"]), _lib.stream_ptr()), "sh_adam_step_bf16")
        else:
            _lib.check(_lib.load().sh_adam_step(n, P, G, M, V, S, N, _lib.ptr(lr), float(b1), float(b2), float(group["eps"]),
                                                float(group["weight_decay"]), _lib.stream_ptr()), "sh_adam_step")

    # ---------------------------------------------------------------------------------- overlap with backward
    def overlap_backward(self, min_numel=1 << 20):
        """Update parameters of at least `min_numel` elements as soon as their gradient exists (see module doc)."""
        self.remove_overlap()
        for group in self.param_groups:
            for p in group["params"]:
                if p.requires_grad and p.numel() >= min_numel:
                    self._hooks.append(p.register_post_accumulate_grad_hook(lambda q, g=group: self._on_grad(g, q)))
        return self

    def remove_overlap(self):
        for h in self._hooks:
            h.remove()
        self._hooks = []

    def _on_grad(self, group, p):
        if self._side is None:
            self._side = torch.cuda.Stream(device=p.device)
        self._side.wait_stream(torch.cuda.current_stream(p.device))      # the gradient was produced on the current stream
        with torch.cuda.stream(self._side):
            self._apply(group, [p])
        self._early.add(id(p))

    # ---------------------------------------------------------------------------------- update inside the weight-gradient kernel
    def fuse_linear_weight_gradients(self, modules):
        """For the weights of the given `nn.Linear` modules that run on `latent_linear` (the autoencoder's two latent FCs: 99 %
        of its parameters), let the kernel that computes a tile of the weight gradient apply this optimizer's update to the same
        tile of the weight and its two moments (sh_linear_bwd_wgt_adam) instead of writing the gradient for `step()` to read
        back: 24 instead of 32 bytes of HBM traffic per weight and step, the same bits.  `weight.grad` is then never
        materialised (it stays None); `step()` updates everything else and advances the fused parameters' step counts.

        **`backward()` itself then MUTATES these weights and their moments** - so, as with `overlap_backward`: only when
        nothing else consumes the gradients between backward and step (no all-reduce, no clipping) and with exactly ONE
        backward per `step()`.  A second backward through a fused layer before `step()` (gradient accumulation) raises a
        RuntimeError instead of applying a second update with the same bias-correction step; a `step()` that the caller skips
        (NaN guard, GradScaler) cannot undo the update backward already applied - do not fuse under such a policy.
        Shapes the kernel does not serve (batch > 64, sizes that are not multiples of 64) and parameters that already hold a
        `.grad` fall back to the ordinary gradient + `step()`.  On the bf16 compute path the same kernel runs: the bf16
        operand tiles are widened to fp32 on their way into LDS and multiplied on the fp32 MFMA (products of bf16 values are
        exact there), the bf16 working copy is rewritten with the update - close to, not bit-identical with,
        `linear_bwd_wgt_bf16` followed by `sh_adam_step_bf16` (tests/test_bf16.py has the tolerance test).
        The registry entry holds this optimizer weakly and is dropped when the optimizer dies, when `remove_fusion()` is
        called, or when another `optim.Adam` is constructed over the same parameter."""
        import weakref
        from . import linear
        for mod in modules:
            w = mod.weight
            group = next((g for g in self.param_groups if any(q is w for q in g["params"])), None)
            if group is None:
                raise ValueError("fuse_linear_weight_gradients: a module's weight is not a parameter of this optimizer")
            if not (w.is_cuda and w.dtype == torch.float32 and w.is_contiguous() and w.dim() == 2):
                raise RuntimeError("fuse_linear_weight_gradients: contiguous 2-D fp32 HIP weights only")
            prev = linear._FUSED_UPDATE.get(w.data_ptr())
            if prev is not None and prev[0]() is w and prev[1]() not in (None, self):
                prev[1]()._drop_fusion_of(w)
            linear._FUSED_UPDATE[w.data_ptr()] = (weakref.ref(w), weakref.ref(self), self.param_groups.index(group))
            if not any(r() is w for r in self._fused):
                self._fused.append(weakref.ref(w))
        return self

    def remove_fusion(self):
        from . import linear
        for r in self._fused:
            w = r()
            if w is None:
                continue
            ent = linear._FUSED_UPDATE.get(w.data_ptr())
            if ent is not None and ent[1]() in (None, self):       # (never another optimizer's registration)
                linear._FUSED_UPDATE.pop(w.data_ptr(), None)
        self._fused = []
        self._fused_done = []

    @torch.no_grad()
    def _fused_update(self, group, p, dy, x, want_bias, mma):
        from . import ops, shadow
        if any(q is p for q in self._fused_done):
            raise RuntimeError("semantichuman_amd.optim.Adam: second backward through a layer whose update is fused into its weight-"
                               "gradient kernel before step() - backward itself applies the update there, so gradient accumulation "
                               "is not possible; call remove_fusion() to accumulate")
        st = self._state_of(p)
        db = ops.linear_bwd_wgt_adam(dy, x, p, st["exp_avg"], st["exp_avg_sq"], st["step"], self._lr_tensor(group, p.device), group["betas"],
                                     group["eps"], group["weight_decay"], want_bias=want_bias, mma=mma, weight_bf16=shadow.lookup(p))
        self._fused_done.append(p)
        return db

    # ---------------------------------------------------------------------------------- torch.optim API
    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        if self._early:
            torch.cuda.current_stream(self._side.device).wait_stream(self._side)   # join: early updates are part of this step
        for group in self.param_groups:
            fused = [p for p in self._fused_done if any(q is p for q in group["params"])]
            self._apply(group, [p for p in group["params"] if id(p) not in self._early], bump_only=fused)
        self._early.clear()
        self._fused_done = []
        return loss

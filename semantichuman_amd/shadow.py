"""bf16 working copies ("shadows") of fp32 master parameters for the bf16 compute path.

The large dense layers read their weight as bf16 (28 MB instead of 56.6 MB per pass); the fp32 master stays the
`nn.Parameter` (same `state_dict`, same optimizer state).  A shadow is valid while the master has not changed:

  * in-place changes made through torch (load_state_dict, torch.optim, .copy_) bump the tensor's `_version` - the next
    `get()` re-converts with one streaming kernel (sh_cast_f32_to_bf16) INTO THE SAME bf16 tensor: the copy keeps its
    address for as long as the parameter keeps its shape and device (a captured hipGraph may hold that address);
  * `semantichuman_amd.optim.Adam` updates the master through the library's kernel (raw pointer: no version bump) and
    rewrites the registered shadow IN THE SAME KERNEL (sh_adam_step_bf16) - it looks the shadow up here on every step,
    so a shadow that exists is always kept current by that optimizer.
"""
from __future__ import annotations

import weakref

import torch

from . import ops

_REG = {}      # id(param) -> [weak reference to the parameter, bf16 copy, (version, data_ptr) it was made from]


def _entry(param):
    ent = _REG.get(id(param))
    if ent is not None and ent[0]() is not param:      # a recycled id: the old parameter is gone
        _REG.pop(id(param), None)
        ent = None
    return ent


def get(param: torch.Tensor) -> torch.Tensor:
    """The bf16 copy of `param`, converting if it is missing or stale."""
    ent = _entry(param)
    key = (param._version, param.data_ptr())
    if ent is not None and ent[1].device == param.device and ent[1].shape == param.shape:
        if ent[2] != key:
            # stale (load_state_dict, an in-place torch update): refresh IN PLACE - a captured hipGraph has this copy's address
            # baked into the latent-FC kernels' and sh_adam_step_bf16's arguments, so the copy must never move while it lives
            ops.cast_bf16(param.detach().contiguous(), out=ent[1])
            ent[2] = key
        return ent[1]
    sh = ops.cast_bf16(param.detach().contiguous())             # first use, or the parameter changed device / shape
    pid = id(param)
    _REG[pid] = [weakref.ref(param, lambda _r, pid=pid: _REG.pop(pid, None)), sh, key]
    return sh


def lookup(param: torch.Tensor):
    """The registered shadow of `param` if it has one AND it is current (None otherwise) - for an optimizer that is about to
    update the master in place through a raw pointer and keeps the shadow in step itself."""
    ent = _entry(param)
    if ent is None or ent[1].device != param.device or ent[2] != (param._version, param.data_ptr()):
        return None
    return ent[1]


def refresh(param: torch.Tensor) -> bool:
    """Re-convert the registered copy of `param` IN PLACE, whatever its bookkeeping says - for a writer that changes the master
    through a view torch's version counter does not see (parallel.GradientAllReducer.gather_weights: the all-gather lands in
    `param.data`).  Returns False when the parameter has no copy (nothing reads it as bf16 yet)."""
    ent = _entry(param)
    if ent is None or ent[1].device != param.device or ent[1].shape != param.shape:
        return False
    ops.cast_bf16(param.detach().contiguous(), out=ent[1])
    ent[2] = (param._version, param.data_ptr())
    return True


def drop(param: torch.Tensor):
    _REG.pop(id(param), None)

"""ctypes binding of libsh_kernels.so (the C ABI declared in include/sh_kernels.h).

This is the binding a maintainer of the reference would add.  There is NO
fallback: if the shared library is missing or a call fails, a RuntimeError is raised - the
product path never silently runs on a CPU/PyTorch substitute.
"""
from __future__ import annotations

import ctypes
import os
from ctypes import c_char_p, c_float, c_int, c_int64, c_size_t, c_void_p

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("SH_KERNEL_LIB") or os.path.join(_HERE, "lib", "libsh_kernels.so")   # override: diagnostic builds

ACT_IDS = {"identity": 0, "relu": 1, "elu": 2, "leaky_relu": 3, "sigmoid": 4, "tanh": 5}

# name -> (restype, argtypes); must list every symbol of include/sh_kernels.h
_P, _I, _L = c_void_p, c_int, c_int64
SIGNATURES = {
    "sh_version": (c_int, []),
    "sh_last_error": (c_char_p, []),
    "sh_build_id": (c_char_p, []),
    "sh_clock_probe": (c_int, [_P, _I, _I, _P]),
    "sh_profile_enable": (c_int, [_I]),
    "sh_profile_count": (c_int, []),
    "sh_profile_get": (c_int, [_I, c_char_p, _I, ctypes.POINTER(c_float)]),
    "sh_spiral_conv_fwd": (c_int, [_P, _L, _L, _P, _P, _P, _P, _L, _L, _I, _I, _I, _I, _I, _I, _I, _I, _P]),
    "sh_spiral_conv_bwd_data": (c_int, [_P, _L, _L, _P, _P, _P, _L, _L, _P, _L, _L, _I, _I, _I, _I, _I, _I, _I, _I, _P]),
    "sh_spiral_conv_bwd_data_z": (c_int, [_P, _L, _L, _I, _P, _P, _P, _L, _L, _P, _L, _L, _I, _I, _I, _I, _I, _I, _I, _I, _P]),
    "sh_weight_transpose": (c_int, [_P, _P, _I, _I, _I, _P]),
    "sh_spiral_conv_bwd_wgt_workspace": (c_size_t, [_I, _I, _I, _I, _I]),
    "sh_spiral_conv_bwd_wgt": (c_int, [_P, _L, _L, _P, _L, _L, _P, _P, _P, _P, c_size_t, _I, _I, _I, _I, _I, _I, _P]),
    "sh_spiral_conv_bwd_wgt_presum": (c_int, [_P, _L, _L, _P, _L, _L, _P, _P, _P, _P, c_size_t, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _P]),
    "sh_spiral_conv_bwd_wgt_reduce_multi": (c_int, [_I, _P, _P, _P, _P, _P, _P, _P, _P, _P]),
    "sh_weight_transpose_multi": (c_int, [_I, _P, _P, _P, _P, _P, _P]),
    "sh_act_backward": (c_int, [_P, _L, _L, _P, _L, _L, _P, _L, _L, _I, _I, _I, _I, _I, _P]),
    "sh_act_backward_tr": (c_int, [_P, _L, _L, _P, _L, _L, _P, _L, _L, _I, _I, _I, _I, _I, _I, _P, _P, _P, _P, _P, _P]),
    "sh_spmm": (c_int, [_P, _P, _P, _P, _L, _L, _P, _L, _L, _P, _L, _L, _I, _I, _I, _I, _I, _P]),
    "sh_stack_forward": (c_int, [_I, _P, _P, _I, _I, _I, _I, _P, _P, _P, _I, _I, _P, _P, _I, _P]),
    "sh_stack_backward": (c_int, [_I, _P, _P, _I, _I, _I, _I, _P, _P, _I, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _P, _P, _P, _P, _I, _P]),
    "sh_linear_workspace": (c_size_t, [_I, _I, _I]),
    "sh_linear_fwd": (c_int, [_P, _P, _P, _P, _I, _I, _I, _P, c_size_t, _I, _P]),
    "sh_linear_bwd_data": (c_int, [_P, _P, _P, _I, _I, _I, _P, c_size_t, _I, _P]),
    "sh_linear_bwd_wgt": (c_int, [_P, _P, _P, _P, _I, _I, _I, _P, c_size_t, _I, _P]),
    "sh_linear_bwd_wgt_adam_ok": (c_int, [_I, _I, _I]),
    "sh_linear_bwd_wgt_adam": (c_int, [_P, _I, _P, _I] + [_P] * 6 + [ctypes.c_double] * 4 + [_P, _I, _I, _I, _I, _P]),
    "sh_adam_bump": (c_int, [_I, _P, _P]),
    "sh_grouped_linear_fwd": (c_int, [_I, _P, _L, _P, _P, _P, _P, _L, _P, _I, _P, _P, _P]),
    "sh_grouped_linear_bwd_data": (c_int, [_I, _P, _L, _P, _P, _P, _L, _P, _I, _P, _P, _P]),
    "sh_grouped_linear_bwd_wgt": (c_int, [_I, _P, _L, _P, _P, _L, _P, _P, _P, _I, _P, _P, _P]),
    "sh_reduce_workspace": (c_size_t, []),
    "sh_l1_loss_fwd": (c_int, [_P, _P, _L, _P, _P, _P]),
    "sh_l1_loss_bwd": (c_int, [_P, _P, _L, _P, _P, _P]),
    "sh_vertex_l2": (c_int, [_P, _P, _I, _I, _I, c_float, _P, _P, _P]),
    "sh_edge_ratio_loss_fwd": (c_int, [_P, _P, _P, _I, _I, _I, _P, _P, _P]),
    "sh_edge_ratio_loss_bwd": (c_int, [_P, _P, _P, _P, _P, _I, _I, _I, _P, _P, _P]),
    "sh_spiral_conv_bwd_wgt_thin_ok": (c_int, [_I, _I, _I, _I, _I, _I]),
    "sh_spiral_conv_bwd_wgt_thin": (c_int, [_P, _L, _L, _P, _I, _L, _L, _P, _P, c_size_t, _P, _P, _L, _L, _P, _I, _I, _I, _I, _I, _I, _I, _I, _I, _P]),
    "sh_recon_loss_workspace": (c_size_t, []),
    "sh_recon_loss_fwd": (c_int, [_P, _P, _P, _I, _I, _I, c_float, _P, _P, _P, _P]),
    "sh_recon_loss_bwd": (c_int, [_P, _P, _P, _P, _I, _I, _I, c_float, _P, _P, _P]),
    "sh_part_pairdist_tile_rows": (c_int, []),
    "sh_part_pairdist_loss_fwd": (c_int, [_P] * 9 + [_I] * 6 + [c_float, _I, _P, _P, _P, _P, c_size_t, _P]),
    "sh_part_pairdist_loss_bwd": (c_int, [_P] * 9 + [_I] * 6 + [c_float, _I, _P, _P, _P, _P]),
    "sh_part_pairdist_loss_fwd_grad": (c_int, [_P] * 9 + [_I] * 6 + [c_float, _I, _P, _P, _P, _P, _P, c_size_t, _P]),
    "sh_part_pairdist_loss_bwd_scale": (c_int, [_P] * 6 + [_I] * 4 + [_P, _P]),
    "sh_measure_girth": (c_int, [_P, _L, _P, _P, _P, _P, _I, _I, _P, _P]),
    "sh_bone_length": (c_int, [_P, _P, _I, _I, _I, _P, _P]),
    "sh_dataset_normalize": (c_int, [_P, _P, _I, _I, _I, ctypes.c_uint, _P, _P, _P, _P, _P, _P]),
    "sh_gather_meshes": (c_int, [_P, _L, _P, _I, _P, _P]),
    "sh_adam_step": (c_int, [_I, _P, _P, _P, _P, _P, _P, _P] + [ctypes.c_double] * 4 + [_P]),
    "sh_joint_regress": (c_int, [_P, _L, _P, _I, _I, _I, _P, _P]),
    "sh_joint_l1_loss_fwd": (c_int, [_P, _L, _P, _P, _P, _I, _I, _I, _I, _P, _P, _P]),
    "sh_joint_l1_loss_bwd": (c_int, [_P, _P, _P, _P, _I, _I, _I, _I, _I, _P, _P, _P]),
    "sh_part_volume_loss_fwd": (c_int, [_P, _P, _L, _P, _P, _P, _I, _I, _P, _P, _P]),
    "sh_part_volume_loss_bwd": (c_int, [_P, _L, _P, _P, _P, _P, _P, _I, _I, _I, _P, _P, _P]),
    "sh_zpart_reg": (c_int, [_P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _P, _P, _P, _P]),
    "sh_kps2skl": (c_int, [_P, _I, _I, _P, _P, _P, _I, _I, _P, _P]),
    "sh_skl2kps": (c_int, [_P, _I, _I, _I, _P, _P, _I, _P, _I, _P, _P]),
    "sh_weighted_sum": (c_int, [_I, _P, _P, _P, _P, _P, _P]),
    # bf16 compute path
    "sh_conv_wfrag_bytes": (c_size_t, [_I, _I, _I]),
    "sh_conv_wfrag_prep_multi": (c_int, [_I, _P, _P, _P, _P, _P, _P, _P]),
    "sh_spiral_conv_fwd_bf16": (c_int, [_P, _I, _L, _L, _P, _P, _P, _P, _I, _L, _L, _I, _I, _I, _I, _I, _I, _I, _P]),
    "sh_spiral_conv_bwd_data_bf16": (c_int, [_P, _I, _L, _L, _P, _P, _P, _I, _L, _L, _P, _L, _L, _I, _I, _I, _I, _I, _I, _I, _P]),
    "sh_spiral_conv_bwd_wgt_workspace_bf16": (c_size_t, [_I, _I, _I, _I, _I]),
    "sh_spiral_conv_bwd_wgt_bf16": (c_int, [_P, _I, _L, _L, _P, _I, _L, _L, _P, _P, c_size_t, _I, _I, _I, _I, _I, _P]),
    "sh_spiral_conv_bwd_wgt_reduce_multi_bf16": (c_int, [_I, _P, _P, _P, _P, _P, _P, _P, _P, _P]),
    "sh_linear_workspace_bf16": (c_size_t, [_I, _I, _I]),
    "sh_cast_f32_to_bf16": (c_int, [_P, _P, _L, _P]),
    "sh_linear_fwd_bf16": (c_int, [_P, _I, _P, _P, _P, _I, _I, _I, _I, _P, c_size_t, _P]),
    "sh_linear_bwd_data_bf16": (c_int, [_P, _I, _P, _P, _I, _I, _I, _I, _P, c_size_t, _P]),
    "sh_linear_bwd_wgt_bf16": (c_int, [_P, _I, _P, _I, _P, _P, _I, _I, _I, _P]),
    "sh_spmm_bf16": (c_int, [_P, _P, _P, _P, _L, _L, _P, _L, _L, _P, _L, _L, _I, _I, _I, _I, _I, _P]),
    "sh_act_backward_bf16": (c_int, [_P, _L, _L, _P, _L, _L, _P, _L, _L, _I, _I, _I, _I, _I, _P]),
    "sh_adam_step_bf16": (c_int, [_I, _P, _P, _P, _P, _P, _P, _P, _P] + [ctypes.c_double] * 4 + [_P]),
    "sh_stack_forward_bf16": (c_int, [_I, _P, _P, _I, _I, _I, _I, _I, _P, _P, _P, _I, _P, _I, _I, _P]),
    "sh_stack_backward_bf16": (c_int, [_I, _P, _P, _I, _I, _I, _I, _I, _P, _P, _I, _I, _P, _P, _I, _P, _P, _I, _P, _P, _P, _P, _I, _P]),
    # three-plane form of the fp32 path
    "sh_p3_bytes": (c_size_t, [_I, _I, _I]),
    "sh_to_p3": (c_int, [_P, _L, _L, _P, _I, _I, _I, _P]),
    "sh_conv_wfrag3_bytes": (c_size_t, [_I, _I, _I]),
    "sh_conv_wfrag3_prep_multi": (c_int, [_I, _P, _P, _P, _P, _P, _P, _P]),
    "sh_spiral_conv_p3_ok": (c_int, [_I, _I, _I, _I]),
    "sh_spiral_conv_p3_kind": (c_int, [_I, _I, _I, _I]),
    "sh_p3_launch_count": (c_int64, []),
    "sh_spiral_conv_fwd_img": (c_int, [_P, _L, _L, _P, _P, _P, _P, _L, _L, _P, _I, _I, _I, _I, _I, _I, _I, _I, _P]),
    "sh_act_backward_tr_img": (c_int, [_P, _L, _L, _P, _L, _L, _P, _L, _L, _P, _I, _I, _I, _I, _I, _I, _P, _P, _P, _P, _P, _P]),
    "sh_spmm_p3": (c_int, [_P, _P, _P, _P, _L, _L, _P, _L, _L, _P, _P, _L, _L, _I, _I, _I, _I, _I, _P]),
    "sh_spiral_conv_fwd_p3": (c_int, [_P, _P, _P, _P, _P, _L, _L, _P, _I, _I, _I, _I, _I, _I, _I, _P]),
    "sh_spiral_conv_bwd_data_p3": (c_int, [_P, _I, _P, _L, _L, _I, _P, _P, _P, _L, _L, _P, _P, _L, _L, _P, _I, _I, _I, _I, _I, _I, _I, _P]),
    "sh_spiral_conv_p3_rag_ok": (c_int, [_I, _I, _I, _I, _I]),
    "sh_spiral_conv_bf16_rag_ok": (c_int, [_I, _I, _I, _I, _I]),
    "sh_spiral_conv_p3_grp_ok": (c_int, [_I, _I, _I, _I, _I]),
    "sh_spiral_conv_p3_grp_members": (c_int, [_I, _I, _I, _I]),
    "sh_spiral_conv_p3_grp_pays": (c_int, [_I, _I]),
    "sh_spiral_conv_p3_grp": (c_int, [_P, _P, _P, _P, _I, _I, _P, _P, _P, _L, _L, _P, _P, _L, _L, _P, _I, _I, _I, _I, _I, _I, _I, _I, _P]),
    "sh_spiral_conv_bwd_data_bf16_rag": (c_int, [_P, _L, _L, _P, _P, _I, _P, _P, _L, _L, _P, _L, _L, _I, _I, _I, _I, _I, _I, _I, _P]),
    "sh_spiral_conv_bwd_data_p3_rag": (c_int, [_P, _P, _P, _I, _P, _P, _L, _L, _P, _P, _L, _L, _P, _I, _I, _I, _I, _I, _I, _I, _P]),
    "sh_spiral_conv_bwd_wgt_p3_ok": (c_int, [_I, _I, _I, _I, _I]),
    "sh_spiral_conv_bwd_wgt_p3_workspace": (c_size_t, [_I, _I, _I, _I, _I]),
    "sh_spiral_conv_bwd_wgt_p3": (c_int, [_P, _I, _P, _P, _P, c_size_t, _I, _I, _I, _I, _I, _P]),
    "sh_spiral_conv_bwd_wgt_p3_presum": (c_int, [_P, _I, _P, _P, _P, c_size_t, _P, _L, _L, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _P]),
    "sh_spiral_conv_bwd_wgt_reduce_multi_kinds": (c_int, [_I, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P]),
}
DTYPE_IDS = {"float32": 0, "bfloat16": 1}



class CsrRef(ctypes.Structure):                        # sh_csr_ref
    _fields_ = [("rowptr", c_void_p), ("col", c_void_p), ("val", c_void_p)]


class StackStep(ctypes.Structure):                     # sh_stack_step (include/sh_kernels.h) - field order is the ABI
    _fields_ = [("kind", c_int), ("param", c_int), ("table", c_void_p), ("table_t", c_void_p),
                ("R", c_int), ("S", c_int), ("n_in", c_int), ("cin", c_int), ("cout", c_int), ("act", c_int), ("zero_row", c_int),
                ("n1", c_int), ("n2", c_int), ("sum1", CsrRef), ("sum2", CsrRef), ("m", CsrRef), ("mt", CsrRef),
                ("m_rows", c_int), ("m_cols", c_int), ("extend", c_int),
                ("rag_rows", c_void_p), ("rag_pos", c_void_p), ("rag_L", c_int),
                ("fg_rows", c_void_p), ("fg_pos", c_void_p), ("fg_out", c_void_p), ("fg_n", c_int), ("fg_L", c_int),
                ("bg_rows", c_void_p), ("bg_pos", c_void_p), ("bg_out", c_void_p), ("bg_n", c_int), ("bg_L", c_int)]


_lib = None


class KernelLibraryError(RuntimeError):
    pass


def load(path: str | None = None):
    """Load (once) and return the ctypes library.  Raises KernelLibraryError if it is absent."""
    global _lib
    if _lib is not None and path is None:
        return _lib
    p = path or LIB_PATH
    if not os.path.exists(p):
        raise KernelLibraryError(
            "semantichuman_amd: HIP kernel library not found at %s. Build it with "
            "`python -c 'import __graft_entry__ as g; g.build()'` (or `make -C semantichuman_amd/csrc`). "
            "There is no CPU fallback." % p)
    try:
        lib = ctypes.CDLL(p)
    except OSError as e:
        raise KernelLibraryError("semantichuman_amd: cannot load %s: %s" % (p, e)) from e
    for name, (res, args) in SIGNATURES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as e:
            raise KernelLibraryError("semantichuman_amd: %s does not export %s" % (p, name)) from e
        fn.restype, fn.argtypes = res, args
    if path is None:
        _lib = lib
    return lib


def check(rc: int, what: str):
    if rc != 0:
        msg = load().sh_last_error()
        raise RuntimeError("%s failed (status %d): %s" % (what, rc, msg.decode() if msg else ""))


def stream_ptr():
    """hipStream_t of torch's current stream (kernels are only ever enqueued there, so they
    are ordered with torch's own work and can be captured into a hipGraph)."""
    import torch
    return c_void_p(torch.cuda.current_stream().cuda_stream)


def profile_enable(on: bool):
    check(load().sh_profile_enable(1 if on else 0), "sh_profile_enable")


def profile_records():
    """[(kernel name, milliseconds)] recorded since profile_enable(True); synchronises."""
    lib = load()
    out = []
    buf = ctypes.create_string_buffer(128)
    ms = c_float(0)
    for i in range(lib.sh_profile_count()):
        check(lib.sh_profile_get(i, buf, 128, ctypes.byref(ms)), "sh_profile_get")
        out.append((buf.value.decode(), ms.value))
    return out


def profile_records_by_kernel():
    """[(kernel name without the '|shape' tag, shape tag, milliseconds)]."""
    return [(n.split("|")[0], n.split("|")[1] if "|" in n else "", ms) for n, ms in profile_records()]


def build_id() -> str:
    """Hash of the kernel sources the LOADED library was compiled from (sh_build_id)."""
    return load().sh_build_id().decode()


def env_overrides() -> dict:
    """The SH_* tuning / ablation switches set in this process's environment (they select kernels, never results)."""
    return {k: v for k, v in sorted(os.environ.items()) if k.startswith("SH_") and not k.startswith(("SH_BENCH_", "SH_KERNEL_LIB"))}


MMA_MODES = {"exact": 0, "split3": 1, "planes3": 2}


def _mma_default() -> str:
    v = os.environ.get("SH_F32_MMA", "exact").strip().lower()
    if v in MMA_MODES:
        return v
    return {"0": "exact", "1": "split3", "2": "planes3"}.get(v, "exact")


_mma_mode = _mma_default()


def set_f32_mma_mode(mode: str):
    """Arithmetic form of the fp32 path's matrix products for the calls that follow (include/sh_kernels.h enum sh_mma_mode):
    "exact" (fp32 MFMA), "split3" (exact bf16x3 operand split by every consumer, six bf16 MFMAs per product, fp32
    accumulation) or "planes3" (the same arithmetic, the split written once by the producer as three bf16 planes).  A
    Python-side default only: the library has no mode of its own, every call names its form, and an autograd node keeps the
    form its forward pass ran in for its backward pass."""
    global _mma_mode
    if mode not in MMA_MODES:
        raise ValueError("f32 mma mode must be one of %s" % sorted(MMA_MODES))
    _mma_mode = mode


def get_f32_mma_mode() -> str:
    return _mma_mode


def mma_id(mode: str | None = None) -> int:
    return MMA_MODES[_mma_mode if mode is None else mode]


def ptr(t):
    """Device pointer of a tensor (None -> NULL)."""
    return c_void_p(0 if t is None else t.data_ptr())

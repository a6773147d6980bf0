"""The C-ABI library must load and export exactly the symbols include/sh_kernels.h declares.
(No kernel is launched here - this runs without a GPU.)"""
import ctypes
import os
import re

import pytest

from semantichuman_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    src = open(os.path.join(ROOT, "include", "sh_kernels.h")).read()
    return sorted(set(re.findall(r"SH_API\s+[\w\s\*]+?\b(sh_\w+)\s*\(", src)))


def test_header_and_binding_agree():
    assert declared_symbols() == sorted(_lib.SIGNATURES.keys())


def test_library_exports_every_symbol():
    if not os.path.exists(_lib.LIB_PATH):
        import __graft_entry__
        __graft_entry__.build()
    lib = _lib.load()
    for name in declared_symbols():
        assert hasattr(lib, name), name
    assert lib.sh_version() >= 100
    assert lib.sh_reduce_workspace() >= 1024
    assert lib.sh_spiral_conv_bwd_wgt_workspace(64, 863, 8, 128, 64) > 0
    assert lib.sh_spiral_conv_bwd_wgt_workspace(0, 1, 1, 1, 1) == 0


def test_argument_validation_without_gpu():
    """Entry points validate their arguments before touching the device."""
    lib = _lib.load()
    null = ctypes.c_void_p(0)
    rc = lib.sh_spiral_conv_fwd(null, 0, 0, null, null, null, null, 0, 0, 1, 1, 1, 1, 1, 2, -1, 0, null)
    assert rc == -1 and b"null pointer" in lib.sh_last_error()
    rc = lib.sh_l1_loss_fwd(null, null, 0, null, null, null)
    assert rc == -1


def test_missing_library_fails_loudly(tmp_path):
    with pytest.raises(_lib.KernelLibraryError):
        _lib.load(str(tmp_path / "nope.so"))

import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    import torch
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="no HIP device in this container")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


# The fp32 path has three arithmetic forms of its matrix products (include/sh_kernels.h, enum sh_mma_mode): the exact fp32
# MFMA (bench.py's default headline), the bf16x3 split done by every consumer, and the same arithmetic over three bf16 planes
# written once by the producer.  Every GPU parity test of the fp32 path runs in ALL of them, with the SAME tolerances - the
# condition under which a split form may be quoted as fp32 at all.
F32_PARITY_MODULES = {"test_gpu_parity", "test_configs", "test_train_loop", "test_semantic", "test_editing", "test_wgrad_thin"}


def pytest_generate_tests(metafunc):
    mod = metafunc.module.__name__.split(".")[-1]
    if mod in F32_PARITY_MODULES and metafunc.definition.get_closest_marker("gpu") is not None and "f32_mma" not in metafunc.fixturenames:
        metafunc.fixturenames.append("f32_mma")
        metafunc.parametrize("f32_mma", ["exact", "split3", "planes3"], indirect=True)


@pytest.fixture
def f32_mma(request):
    from semantichuman_amd import _lib
    was = _lib.get_f32_mma_mode()
    _lib.set_f32_mma_mode(request.param)
    yield request.param
    _lib.set_f32_mma_mode(was)

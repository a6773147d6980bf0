"""AddressSanitizer / UBSan run of the host-side stack sequencers (csrc/stack_exec.hip: pointer chaining over step tables,
the extra rows behind gradient buffers, the folded up-sampling's append offset) against stub kernels that touch the first
and last byte of every operand range - CPU only, no GPU (tests/host_asan/)."""
import os
import shutil
import subprocess

import pytest

HERE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "host_asan")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")


@pytest.fixture(scope="module")
def driver(tmp_path_factory):
    if not os.path.exists(HIPCC):
        pytest.skip("hipcc not found (the sanitizer build compiles the host pass of a .hip file)")
    out = str(tmp_path_factory.mktemp("sh_host_asan"))
    r = subprocess.run(["make", "-C", HERE, "OUT=" + out, out + "/driver"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    yield os.path.join(out, "driver")
    shutil.rmtree(out, ignore_errors=True)


def test_sequencers_are_clean_under_asan_and_ubsan(driver):
    r = subprocess.run([driver], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-3000:]
    calls = r.stdout.strip().splitlines()
    assert calls[-1].startswith("SEQUENCERS OK")
    fp32 = calls[:calls.index("wfrag_prep n=3")]
    # forward chain, then per conv (last to first): weight gradient, list pre-sums, backward-data; U^T in between; one reduction
    assert fp32[:4] == ["conv_fwd R=6 Cin=8 Cout=16", "spmm rows=4 C=16", "conv_fwd R=9 Cin=16 Cout=8", "conv_fwd R=9 Cin=8 Cout=3"]
    assert fp32[4] == "act_backward R=9 C=3 transposes=3"             # the transposes ride in the launch that opens the pass
    assert fp32[5:8] == ["bwd_wgt R=9 Cin=8 Cout=3", "spmm rows=2 C=3", "bwd_data n_in=9 Cin=8 Cout=3"]
    assert fp32[-1] == "reduce n=3" and fp32.count("spmm rows=6 C=16") == 1          # the folded up-sampling's transpose: plain spmm


def test_the_harness_sees_an_undersized_buffer(driver):
    """Negative control: without room for the folded up-sampling's appended rows the sequencer's append pointer leaves the
    buffer - AddressSanitizer must report it (otherwise the clean run above proves nothing)."""
    r = subprocess.run([driver], capture_output=True, text=True, timeout=120, env=dict(os.environ, SH_ASAN_NEGATIVE="1"))
    assert r.returncode != 0 and "AddressSanitizer" in r.stderr

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


def test_three_plane_pass_of_the_sequencers(driver):
    """The third pass of the driver: SH_MMA_PLANES3 at batch 16 with image buffers sized by sh_p3_bytes() behind the 16-channel
    tensors - the forward conv that gathers the folded buffer runs the plane kernel on an image its producers wrote (conv 0's
    rows, then the appended up-sampling rows at their offset), the backward-data pass of conv 0 gathers the image of its
    pre-activation gradient (real rows from the U^T launch; the pre-summed rows are handed over as fp32)."""
    r = subprocess.run([driver], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-3000:]
    calls = r.stdout.strip().splitlines()
    p3 = calls[len(calls) - calls[::-1].index("reduce_bf16 n=3"):-1]              # everything after the bf16 pass
    assert p3[:5] == ["conv_fwd R=6 Cin=8 Cout=16", "to_p3 rows=6 C=16", "spmm+image rows=4 C=16", "conv_fwd_p3 R=9 Cin=16 Cout=8",
                      "conv_fwd R=9 Cin=8 Cout=3"]
    assert "bwd_data_p3 n_in=7 Cin=8 Cout=16" in p3 and "spmm+image rows=6 C=16" in p3          # U^T writes conv 0's dpre image
    # the pre-summed rows (3 against 6 real ones: at least half) stay fp32 - the backward-data kernel splits them itself
    assert p3.count("spmm rows=1 C=16") == 1 and p3.count("spmm rows=2 C=16") == 1 and "spmm+image rows=1 C=16" not in p3
    assert "bwd_data n_in=7 Cin=8 Cout=16" not in p3


def test_the_harness_sees_an_undersized_buffer(driver):
    """Negative control: without room for the folded up-sampling's appended rows the sequencer's append pointer leaves the
    buffer - AddressSanitizer must report it (otherwise the clean run above proves nothing)."""
    r = subprocess.run([driver], capture_output=True, text=True, timeout=120, env=dict(os.environ, SH_ASAN_NEGATIVE="1"))
    assert r.returncode != 0 and "AddressSanitizer" in r.stderr

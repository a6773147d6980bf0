#!/usr/bin/env python3
"""One training step of the plain autoencoder in the three-plane form, arenas poisoned with NaN (stack.DEBUG_POISON), and a
SHA-256 of every parameter gradient, the loss and the reconstruction - the check behind keep_fp32 == 2 (sh_stack_forward /
sh_stack_backward, round 6): with SH_P3_DROP_FP32=1 (default) the fp32 rows neither pass reads are not written, and nothing may
change, bit for bit, against SH_P3_DROP_FP32=0; a kernel that read an unwritten row would read NaN.

    python tools/drop_fp32_check.py [template.npz] [batch]      (tests/test_p3.py runs it twice as a child process)

Also prints IMAGE_ONLY n: the number of launches of the step that wrote no fp32 rows (`f32=0` in the library's launch records)."""
import hashlib
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
FE = [[3, 16, 32, 64, 128], [[], [], [], [], []]]
FD = [[128, 64, 32, 32, 16], [[], [], [], [], 3]]


def main():
    import semantichuman_amd as sh
    from semantichuman_amd import _lib, stack, synthetic
    from semantichuman_amd.hierarchy import load_hierarchy
    tpl = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "tests", "golden", "template6890.npz")
    B = int(sys.argv[2]) if len(sys.argv) > 2 else 64
    stack.DEBUG_POISON = True
    dev = torch.device("cuda:0")
    h = load_hierarchy(tpl)
    _lib.set_f32_mma_mode("planes3")
    torch.manual_seed(5)
    m = sh.SpiralAutoencoder(FE, FD, 256, h.sizes, h.spiral_sizes, h.spirals, h.D, h.U, dev)
    x = torch.from_numpy(synthetic.synth_batch(h.verts, B, seed=3)).to(dev)
    ft = sh.FaceTables(h.faces, h.sizes[0] + 1, dev)
    digests = {}
    for rep in range(2):                                   # twice: the second pass runs on arenas the first one left behind
        m.zero_grad(set_to_none=True)
        xh, z = m(x)
        loss, _ = sh.recon_loss(xh, x, ft, 1e-2)
        loss.backward()
        torch.cuda.synchronize()
        d = {"loss": hashlib.sha256(loss.detach().cpu().numpy().tobytes()).hexdigest()[:16],
             "x_hat": hashlib.sha256(xh.detach().cpu().numpy().tobytes()).hexdigest()[:16]}
        for n, p in m.named_parameters():
            g = p.grad
            assert g is not None and bool(torch.isfinite(g).all()), "non-finite gradient of %s" % n
            d[n] = hashlib.sha256(g.detach().cpu().numpy().tobytes()).hexdigest()[:16]
        digests[rep] = d
    assert digests[0] == digests[1], "the step is not reproducible from pass to pass"
    _lib.profile_enable(True)
    m.zero_grad(set_to_none=True)
    sh.recon_loss(m(x)[0], x, ft, 1e-2)[0].backward()
    torch.cuda.synchronize()
    recs = _lib.profile_records_by_kernel()
    _lib.profile_enable(False)
    print("IMAGE_ONLY %d" % sum(1 for _n, tag, _ms in recs if " f32=0" in tag))
    print("DIGEST " + json.dumps(digests[0], sort_keys=True))


if __name__ == "__main__":
    main()

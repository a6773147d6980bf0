"""Inputs of the semantic-loop fixture (tests/golden/semantic_loop.npz), built from closed forms so that the generator
(oracle/gen_golden.py:gen_semantic_loop, which runs the REFERENCE's loop on them) and the GPU test
(tests/test_semantic.py) construct exactly the same tensors.  Test infrastructure."""
import numpy as np
import torch

from semantichuman_amd import constants as C
from semantichuman_amd import synthetic

SEM_LOOP = dict(seed=5, n_epochs=4, batch=2, n_interp=3, init_scale=1.5)


def semantic_loop_inputs(v, sizes):
    """17 Voronoi parts on the template, a random 17-way split of the coarsest level, a positive row-normalised joint
    regressor, measurements, train / interp / val batches.  -> (part_coarse, part_fine, J, train, interp, val)"""
    rs = np.random.RandomState(3)
    coarse = [np.sort(c) for c in np.array_split(rs.permutation(sizes[-1]), 17)]
    vi = v / np.asarray((0.25, 0.15, 0.9))
    seeds = [0]
    dmin = np.linalg.norm(vi - vi[0], axis=1)
    for _ in range(16):
        seeds.append(int(np.argmax(dmin)))
        dmin = np.minimum(dmin, np.linalg.norm(vi - vi[seeds[-1]], axis=1))
    owner = np.argmin(((vi[:, None, :] - vi[None, seeds, :]) ** 2).sum(2), axis=1)
    fine = [np.nonzero(owner == k)[0] for k in range(17)]
    J = np.abs(synthetic.closed_form_fill((35, sizes[0]), 1.0, 0.618, 0.3)) ** 8
    J = (J / J.sum(1, keepdims=True)).astype(np.float32)
    B = SEM_LOOP["batch"]

    def batch(seed, k):
        x = synthetic.synth_batch(v, B, seed=seed)
        m = (1.0 + 0.3 * np.abs(synthetic.closed_form_fill((B, 16), 1.0, 0.77, 0.2 + k))).astype(np.float32)
        return {"verts": torch.from_numpy(x), "idx": torch.arange(B) + 10 * k, "measure": torch.from_numpy(m)}
    train = [batch(31, 0)]
    interp = [batch(41 + k, 1 + k) for k in range(SEM_LOOP["n_interp"])]
    val = [batch(51, 9)]
    return dict(zip(C.PART_LIST, coarse)), dict(zip(C.PART_LIST, fine)), J, train, interp, val


def fill_params(model, scale=1.0):
    """Closed-form deterministic weights (oracle/gen_golden.fill_params): w = a sin(b i + c), a = scale / sqrt(fan_in)."""
    import math
    with torch.no_grad():
        for j, (name, p) in enumerate(model.named_parameters()):
            fan_in = p.shape[1] if p.dim() == 2 else p.shape[0]
            a = scale / math.sqrt(fan_in)
            p.copy_(torch.from_numpy(synthetic.closed_form_fill(tuple(p.shape), a, 0.37 + 0.011 * j, 0.1 * j)).to(p.device))

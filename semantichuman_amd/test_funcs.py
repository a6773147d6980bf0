"""Evaluation loop - reference test_funcs.py:17-57 (name, arguments and return tuple kept).

L1 = sum over batches of mean|x_hat - x| * (b / len(dataset)), L2 = the same weighting of the mean
per-vertex Euclidean error in millimetres (`mm_constant` = 1000); the dummy row is dropped when the
dataset carries one (`dataset.dummy_node`, reference :39-44).  The two reductions are HIP kernels
(sh_l1_loss_fwd, sh_vertex_l2); partial results stay on the device until the end.
"""
from __future__ import annotations

import torch

from . import losses


def test_autoencoder_dataloader(device, model, dataloader_test, shapedata=None, J_regressor=None, mm_constant=1000,
                                unnormal_flag=False, keep_outputs=True):
    model.eval()
    n_total = float(len(dataloader_test.dataset))
    dummy = bool(getattr(dataloader_test.dataset, "dummy_node", True))
    l1 = torch.zeros((), device=device)
    l2 = torch.zeros((), device=device)
    preds, zs, txs = [], [], []
    with torch.no_grad():
        for sample_dict in dataloader_test:
            tx = sample_dict["verts"].to(device)
            prediction, z = model(tx)
            if keep_outputs:
                preds.append(prediction)
                zs.append(z)
                txs.append(tx)
            w = tx.shape[0] / n_total
            l1 += losses.eval_l1(prediction, tx, dummy_node=dummy) * w
            l2 += losses.vertex_l2_mm(prediction, tx, dummy_node=dummy, mm_constant=float(mm_constant)) * w
    predictions = torch.cat(preds, 0).cpu().numpy() if preds else None
    z_s = torch.cat(zs, 0).cpu().numpy() if zs else None
    tx_s = torch.cat(txs, 0).cpu().numpy() if txs else None
    return predictions, z_s, tx_s, l1.item(), l2.item()


test_autoencoder_dataloader.__test__ = False      # not a pytest test despite its (reference) name


def test_autoencoder_dataloader_nonormal(device, model, dataloader_test, shapedata, J_regressor, mm_constant=1000,
                                         unnormal_flag=False, kpskeep_flag=True):
    """reference test_funcs.py:58-110: evaluation of the semantic model (SpiralAutoencoder_multiz_partkps).
    Joints are regressed on the device from the input meshes (`J_regressor @ verts`, :70), the rows 3/13/14 are
    dropped when `kpskeep_flag` (cfg.TRAIN.kpskeep_flag, :60-62).  Returns
    (predictions, z_s, z_kps_s, tx_s, l1, l2) as numpy arrays / floats like the reference."""
    from . import constants
    import numpy as np
    keep = constants.kps_keep() if kpskeep_flag else list(range(len(constants.NEWSKL_LIST) + 4))
    model.eval()
    n_total = float(len(dataloader_test.dataset))
    dummy = bool(getattr(dataloader_test.dataset, "dummy_node", True))
    J = torch.from_numpy(np.asarray(J_regressor, dtype=np.float32)).to(device)
    l1 = torch.zeros((), device=device)
    l2 = torch.zeros((), device=device)
    preds, zs, zks, txs = [], [], [], []
    with torch.no_grad():
        for sample_dict in dataloader_test:
            tx = sample_dict["verts"].to(device)
            kps_gt = torch.matmul(J, tx[:, :-1, :] if dummy else tx).float()
            prediction, z, z_kps = model(tx, kps_gt[:, keep])
            preds.append(prediction); zs.append(z); zks.append(z_kps); txs.append(tx)
            w = tx.shape[0] / n_total
            l1 += losses.eval_l1(prediction, tx, dummy_node=dummy) * w
            l2 += losses.vertex_l2_mm(prediction, tx, dummy_node=dummy, mm_constant=float(mm_constant)) * w
    cat = lambda ts: torch.cat(ts, 0).cpu().numpy()          # noqa: E731
    return cat(preds), cat(zs), cat(zks), cat(txs), l1.item(), l2.item()


test_autoencoder_dataloader_nonormal.__test__ = False

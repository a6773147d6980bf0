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

"""Training loop of the plain spiral autoencoder - the outer loop stays PyTorch (north_star), the
model and the losses underneath are HIP kernels.

`train_autoencoder_dataloader` keeps the reference's name, positional signature, per-epoch
logging tags and checkpoint layout (reference train_funcs.py:474-583), so `main.py` can call it
unchanged.  Differences, all invisible in the numbers:

  * the edge regulariser is ONE batched kernel instead of a per-sample Python loop with a
    `.cpu().numpy()` synchronisation per sample (reference :503-508);
  * the per-iteration `loss.item()` host syncs (:513) are replaced by on-device accumulation; the
    host reads the running sums once per epoch (and at the `eval_freq` logging points);
  * the global yacs `cfg` is replaced by keyword options (`edgereg_epoch`, `edgereg_w`,
    `ck_frequency`; defaults = configure/traincfg.yaml:40-41,52);
  * optional data-parallel training: pass `reducer=parallel.GradientAllReducer(model)` and give
    every rank its own shard of the data (`dataset.ResidentLoader(rank=, world_size=)`: equal
    batch counts on every rank).  Epoch losses are then sums over all ranks divided by the number
    of samples all ranks processed; checkpoints, sample dumps and logging happen on rank 0 only,
    followed by a barrier;
  * `save_recons=True`: every 50 epochs the last validation and training batch are written through
    `shapedata.save_meshes` exactly as the reference does (:571-582; one device-to-host copy each).

Checkpoints: `{'epoch','autoencoder_state_dict','optimizer_state_dict','scheduler_state_dict'}`
with CPU tensors, at `<metadata_dir>/<checkpoint_path><epoch>.pth.tar` (reference :562-567).
"""
from __future__ import annotations

import os

import numpy as np
import torch
import torch.nn.functional as F

from . import losses


def _as_loss(loss_fn):
    """The reference passes torch's F.l1_loss (main.py:296,311); route it to the HIP kernel."""
    return losses.l1_loss if loss_fn is F.l1_loss or loss_fn is None else loss_fn


def save_checkpoint(path, epoch, model, optim, scheduler):
    """Reference layout and CPU tensors (train_funcs.py:554-569) without moving the live model."""
    module = model.module if hasattr(model, "module") else model
    state = {k: v.detach().cpu() for k, v in module.state_dict().items()}
    torch.save({"epoch": epoch, "autoencoder_state_dict": state, "optimizer_state_dict": optim.state_dict(),
                "scheduler_state_dict": scheduler.state_dict() if scheduler else None}, path)


def _numpy_scalars_allowed():
    """Context that lets torch.load(weights_only=True) rebuild numpy scalars and dtypes (plain data)."""
    import contextlib
    sg = getattr(torch.serialization, "safe_globals", None)
    if sg is None:
        return contextlib.nullcontext()
    allowed = [np.dtype]
    try:
        from numpy._core.multiarray import scalar as _np_scalar
    except ImportError:                                        # numpy 1.x
        from numpy.core.multiarray import scalar as _np_scalar
    allowed.append(_np_scalar)
    allowed += [type(np.dtype(t)) for t in (np.float32, np.float64, np.int32, np.int64, np.bool_)]
    return sg(allowed)


def load_checkpoint(path, model, optim=None, scheduler=None, finetune=False, map_location="cpu", trust_pickle=False):
    """main.py:277-292: returns the epoch to start from."""
    import pickle
    try:                      # the saved layout is tensors + plain Python containers: no pickled code needed
        with _numpy_scalars_allowed():     # reference-written optimizer / scheduler state may hold numpy scalars (data, no code)
            ck = torch.load(path, map_location=map_location, weights_only=True)
    except (pickle.UnpicklingError, RuntimeError) as e:
        # only the weights_only REFUSAL is handled here (torch names it in the message: "Weights only load failed" / "Unsupported
        # global"); a missing / unreadable / corrupt file (also a pickle.UnpicklingError: "invalid load key") or an out-of-memory
        # error propagates unchanged - a corrupt file must never steer the user towards trust_pickle=True
        msg = str(e)
        # torch wraps EVERY failure of its restricted unpickler in "Weights only load failed ..." - also garbage bytes
        # ("Unsupported operand 0", "invalid load key"); a refusal names the global / class it would not construct
        refusal = ("Unsupported global" in msg or "not an allowed global" in msg or "Unsupported class" in msg
                   or (isinstance(e, RuntimeError) and "weights_only" in msg and "Unsupported operand" not in msg))
        if not refusal:
            raise
        if not trust_pickle:
            raise RuntimeError("load_checkpoint: %s does not load with weights_only=True (%s); pass trust_pickle=True only "
                               "for a checkpoint you wrote yourself - unpickling executes code" % (path, e)) from e
        ck = torch.load(path, map_location=map_location, weights_only=False)
    model.load_state_dict(ck["autoencoder_state_dict"])
    if finetune:
        return 1
    if optim is not None:
        optim.load_state_dict(ck["optimizer_state_dict"])
    if scheduler is not None and ck.get("scheduler_state_dict") is not None:
        scheduler.load_state_dict(ck["scheduler_state_dict"])
    return ck["epoch"] + 1


def _save_recons(shapedata, samples_dir, epoch, tx_idx, tx_hat, tx_hat_val):
    """Reference train_funcs.py:571-582: the first mesh of the epoch's last validation batch and of its last training
    batch, written by `shapedata.save_meshes` under the file stems and with the sample index the reference uses (the
    index of the last VALIDATION batch for both, :529,578,582)."""
    ind = [int(tx_idx[0])]
    for t, stem in ((tx_hat_val, "epoch_val{0}"), (tx_hat, "epoch_train{0}")):
        if t is not None:
            shapedata.save_meshes(os.path.join(samples_dir, stem.format(epoch)), t[0:1, 0:-1, :].detach().cpu().numpy(), ind)


def train_autoencoder_dataloader(dataloader_train, dataloader_val, device, model, optim, loss_fn,
                                 start_epoch, n_epochs, eval_freq, dataloader_interp, scheduler,
                                 writer, shapedata, metadata_dir, samples_dir, checkpoint_path,
                                 J_regressor=None, vert_part_index_dict=None, partname_list=None, save_recons=False,
                                 *, edgereg_epoch=0, edgereg_w=1e-2, ck_frequency=50, reducer=None, verbose=True):
    loss_fn = _as_loss(loss_fn)
    import torch.distributed as dist
    world = dist.get_world_size() if (reducer is not None and dist.is_initialized()) else 1
    rank = dist.get_rank() if world > 1 else 0
    if rank != 0:
        writer, verbose = None, False
    f_np = np.asarray(shapedata.reference_mesh.f).astype(np.int32)
    n_rows = None
    face_tables = None
    total_steps = (start_epoch - 1) * len(dataloader_train)
    eval_freq = len(dataloader_train)                                    # reference :482
    history = []

    for epoch in range(start_epoch, n_epochs + 1):
        model.train()
        tloss = torch.zeros((), device=device)
        rec_loss = torch.zeros((), device=device)
        edgereg_loss = torch.zeros((), device=device)
        n_seen, n_val = 0, 0
        tx_hat = tx_hat_val = None
        tx_idx = [0]
        for b, sample_dict in enumerate(dataloader_train):
            optim.zero_grad()
            tx = sample_dict["verts"].to(device)
            cur_bsize = tx.shape[0]
            tx_hat = model(tx)[0]
            if epoch > edgereg_epoch and edgereg_w > 0:
                if face_tables is None or n_rows != tx.shape[1]:
                    n_rows = tx.shape[1]
                    face_tables = losses.FaceTables(f_np, n_rows, device)
                if loss_fn is losses.l1_loss:                 # the reference's loss: both terms in one fused op
                    loss, parts = losses.recon_loss(tx_hat, tx, face_tables, edgereg_w)
                    rec_loss, edgereg_loss = parts[0], parts[1]
                else:
                    rec_loss = loss_fn(tx, tx_hat)
                    edgereg_loss = losses.edge_ratio_loss(tx_hat, tx, face_tables)
                    loss = rec_loss + edgereg_w * edgereg_loss
            else:
                rec_loss = loss_fn(tx, tx_hat)
                loss = rec_loss
            if reducer is not None:
                reducer.prepare()
            loss.backward()
            if reducer is not None:
                reducer.finish()
            optim.step()
            n_seen += cur_bsize
            tloss += cur_bsize * loss.detach()
            if writer and total_steps % eval_freq == 0:
                writer.add_scalar("loss/loss/data_loss", loss.item(), total_steps)
                writer.add_scalar("loss/loss/rec_loss", rec_loss.item(), total_steps)
                writer.add_scalar("loss/loss/edgereg_loss", float(edgereg_loss.detach()), total_steps)
            total_steps += 1

        model.eval()
        vloss = torch.zeros((), device=device)
        with torch.no_grad():
            for b, sample_dict in enumerate(dataloader_val):
                tx = sample_dict["verts"].to(device)
                tx_hat_val = model(tx)[0]
                if "idx" in sample_dict:
                    tx_idx = sample_dict["idx"]
                n_val += tx.shape[0]
                vloss += tx.shape[0] * loss_fn(tx[:, :-1, :], tx_hat_val[:, :-1, :])   # dummy row dropped (:535)

        if scheduler:
            scheduler.step()

        if world > 1:
            # every rank saw its own shard: sums of losses and of sample counts over all ranks (one tiny collective per epoch)
            t = torch.stack([tloss.double(), vloss.double(), torch.tensor(float(n_seen), device=device, dtype=torch.float64),
                             torch.tensor(float(n_val), device=device, dtype=torch.float64)])
            dist.all_reduce(t)
            epoch_tloss = float(t[0]) / max(1.0, float(t[2]))
            tot_val = float(t[3])
            vloss = t[1]
        else:
            epoch_tloss = float(tloss) / float(len(dataloader_train.dataset))
            tot_val = float(len(dataloader_val.dataset))
        if writer:
            writer.add_scalar("avg_epoch_train_loss", epoch_tloss, epoch)
        epoch_vloss = None
        if tot_val > 0:
            epoch_vloss = float(vloss) / tot_val
            if writer:
                writer.add_scalar("avg_epoch_valid_loss", epoch_vloss, epoch)
            if verbose:
                print("epoch {0} | tr {1} | val {2}".format(epoch, epoch_tloss, epoch_vloss))
        elif verbose:
            print("epoch {0} | tr {1} ".format(epoch, epoch_tloss))
        history.append((epoch, epoch_tloss, epoch_vloss))

        if epoch % ck_frequency == 0 and metadata_dir is not None:
            if rank == 0:
                save_checkpoint(os.path.join(metadata_dir, checkpoint_path + "%s.pth.tar" % epoch), epoch, model, optim, scheduler)
            if world > 1:
                dist.barrier()            # nobody runs ahead of (or reads) a checkpoint that is still being written
        if save_recons and epoch % 50 == 0 and rank == 0 and samples_dir is not None:
            _save_recons(shapedata, samples_dir, epoch, tx_idx, tx_hat, tx_hat_val)
    if verbose:
        print("~FIN~")
    return history

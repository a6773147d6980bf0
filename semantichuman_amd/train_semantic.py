"""Semantic training loop - reference train_funcs.train_autoencoder_dataloader_nonormal (:73-472).

The reference loop cannot run on torch >= 2 (`dataloader_interp_iter.next()`, :155,:158,:288,:291), so
it is restated here: same order of operations, same loss terms and weights, same logging tags and
checkpoint layout; the global yacs `cfg` becomes `SemanticTrainOptions` (defaults =
configure/traincfg.yaml).  Every loss is either a HIP kernel (L1, edge ratio, part pairwise distance)
or a handful of small torch ops (joint regression, latent-norm regulariser, part volumes).

One iteration = three encode+decode passes (reference :134, :224-227, :319-321):
  rec     x -> x_hat                                   L1 (+ edge ratio, + latent-norm vs girth)
  interp  scale part latents by `a`, decode            joint L1 + part pair-distance vs scaled GT
  exc     swap bone orientations or lengths in batch   joint L1 + part pair-distance (+ part volume)
"""
from __future__ import annotations

import os
import random
from dataclasses import dataclass, field

import numpy as np
import torch

from . import constants as C
from . import losses, part_losses
from .train_funcs import _as_loss, save_checkpoint


@dataclass
class SemanticTrainOptions:
    """configure/traincfg.yaml:16-52 + cfgs.py defaults."""
    edit_mode: str = "equal"          # equal | rand | exc
    rand_mode: str = "rand"
    exc_mode: str = "ori_or_m"        # ori_m | ori_or_m | ori
    editskl_flag: bool = False
    noleaf_flag: bool = True
    kpskeep_flag: bool = True
    sklkeep_flag: bool = True
    leafkeep_flag: bool = True
    relat_flag: bool = True
    w_mode: str = "threshold"
    w_threshold: float = 0.8
    w_part_mode: str = "1/K"
    factor: tuple = (0.4, 0.8)
    edgereg_epoch: int = 0; edgereg_w: float = 1e-2
    zpartreg_epoch: int = 0; zpartreg_w: float = 1e-2
    vol_epoch: int = 0; vol_w: float = 1e-2
    interp_epoch: int = 0; interp_kps_w: float = 1.0; interp_euc_w: float = 1e-2
    exc_epoch: int = 0; exc_kps_w: float = 1.0; exc_euc_w: float = 1e-2
    ck_frequency: int = 100
    part_list: list = field(default_factory=lambda: list(C.PART_LIST))
    noleaf_part_list: list = field(default_factory=lambda: list(C.NOLEAF_PART_LIST))
    measure_part_list: list = field(default_factory=lambda: list(C.MEASURE_PART_LIST))
    newskl_list: list = field(default_factory=lambda: list(C.NEWSKL_LIST))
    skl_list: list = field(default_factory=lambda: list(C.SKL_LIST))


class SemanticContext:
    """Per-run constants the reference recomputes at the top of its loop (:79-113)."""

    def __init__(self, opts, shapedata, J_regressor, vert_part_index_dict, partname_list, device):
        self.opts, self.device = opts, device
        f_np = np.asarray(shapedata.reference_mesh.f).astype(np.int32)
        self.f_np = f_np
        self.faces = torch.from_numpy(f_np.astype(np.int64)).to(device)
        self.J = torch.from_numpy(np.asarray(J_regressor, dtype=np.float32)).to(device)
        n_verts = self.J.shape[1]
        self.partname_list = list(partname_list)
        self.part_dict = vert_part_index_dict
        self.fpi = torch.from_numpy(part_losses.face_part_index(f_np, vert_part_index_dict, n_verts)).to(device)
        n_kps = len(opts.newskl_list) + 4
        self.kps_keep = [i for i in range(n_kps) if not (opts.kpskeep_flag and i in C.KPS_DROPPED)]
        self.skl_keep = [0, 1, 2, 3, 4, 6, 7, 8, 13, 14, 15, 16, 17] if opts.sklkeep_flag else list(range(len(opts.newskl_list)))
        self.newskl_keep = [i for i in range(len(opts.newskl_list)) if i not in (5, 9, 10)]
        self.leaf_list = [0, 7, 10, 13, 16] if opts.leafkeep_flag else []
        self.part_index_in_allpart = [opts.part_list.index(p) for p in opts.noleaf_part_list]
        self.part_index_in_measure = [opts.measure_part_list.index(p) for p in opts.noleaf_part_list]
        K = len(self.partname_list)
        if opts.w_part_mode == "n/N":
            wp = [len(vert_part_index_dict[p]) / n_verts for p in self.partname_list]
        else:                                   # '1/K' (and '1/rand_num' in the exc branch, :361-362)
            wp = [1.0 / K] * K
        self.tables = part_losses.PartTables({p: vert_part_index_dict[p] for p in self.partname_list}, device,
                                             leaf_parts=self.leaf_list, w_part=wp)
        self.face_tables = None
        # device-resident copies of the index lists used on device tensors every iteration
        it = part_losses.index_tensor
        self.kps_keep_t, self.skl_keep_t, self.newskl_keep_t = it(self.kps_keep, device), it(self.skl_keep, device), it(self.newskl_keep, device)

        self.kps_keep_i32 = torch.tensor(self.kps_keep, dtype=torch.int32, device=device)
        self.part_faces = None                                      # PartFaceTables, built on first use (needs the row count)

    def joints(self, x):
        """J @ x[:, :-1] (reference :131,:161,:296) - inputs only, no gradient: one kernel."""
        if x.is_cuda and not x.requires_grad:
            return part_losses.joint_regress(x, self.J)
        return torch.matmul(self.J, x[:, :-1, :]).float()


def _edit_scales(ctx, opts, B, epoch, measure=None, draw=None):
    """reference :163-221 -> (part indices being edited, a [B, len(parts)])."""
    dev = ctx.device
    lo, hi = opts.factor
    if opts.edit_mode == "rand":
        if opts.rand_mode == "warm_up" and epoch < 100:
            part_num = 1 if epoch < 20 else 2 if epoch < 50 else 4 if epoch < 75 else 8
        else:
            part_num = random.randint(1, len(ctx.partname_list))
        part_index = random.sample(list(range(len(opts.part_list))), part_num)
        if opts.noleaf_flag:
            for leaf in (0, 7, 10, 13, 16):                         # the reference removes at most one (elif chain)
                if leaf in part_index:
                    part_index.remove(leaf)
                    break
        a = torch.rand(len(part_index)).to(dev) * lo + hi            # drawn on the host generator, like the reference (:207)
        return part_index, a[None].repeat(B, 1)
    if opts.edit_mode == "equal":
        if draw is None:
            # `torch.rand(1).to(device)` (:220): the HOST generator, so a seeded run draws the reference's own stream
            return ctx.part_index_in_allpart, torch.ones((B, len(opts.noleaf_part_list)), device=dev) * (torch.rand(1).to(dev) * lo + hi)
        return ctx.part_index_in_allpart, torch.full((B, len(opts.noleaf_part_list)), float(draw), device=dev)
    if opts.edit_mode == "exc":
        return ctx.part_index_in_allpart, torch.flip(measure, dims=[0]) / measure
    raise NotImplementedError(opts.edit_mode)


def _full_scale(ctx, part_index, a, B):
    s = torch.ones((B, len(ctx.partname_list)), device=ctx.device)
    s[:, part_losses.index_tensor(part_index, ctx.device)] = a
    return s


class _WeightedSum(torch.autograd.Function):
    """total = t0 (* w0) + w1 * t1 + ... in sequence - the loop's `loss = loss + w * term` chain - as ONE launch forward and
    one backward (sh_weighted_sum) instead of a multiply, an add and a MulBackward per term.  Same arithmetic, same bits."""

    @staticmethod
    def forward(ctx, weights, *terms):
        import ctypes
        from . import _lib
        n = len(terms)
        arr = (ctypes.c_void_p * n)(*[t.data_ptr() for t in terms])
        w = (ctypes.c_float * n)(*weights)
        out = torch.empty((), dtype=torch.float32, device=terms[0].device)
        _lib.check(_lib.load().sh_weighted_sum(n, arr, w, _lib.ptr(out), None, None, _lib.stream_ptr()), "sh_weighted_sum")
        ctx.weights = tuple(weights)
        return out

    @staticmethod
    def backward(ctx, g):
        import ctypes
        from . import _lib
        n = len(ctx.weights)
        w = (ctypes.c_float * n)(*ctx.weights)
        grads = torch.empty((n,), dtype=torch.float32, device=g.device)
        _lib.check(_lib.load().sh_weighted_sum(n, None, w, None, _lib.ptr(g.contiguous()), _lib.ptr(grads), _lib.stream_ptr()), "sh_weighted_sum")
        return (None,) + tuple(grads[i] for i in range(n))


def weighted_sum(pairs):
    """pairs: [(weight, scalar loss tensor)], the first weight 1 - summed in list order."""
    ts = [t for _, t in pairs]
    if len(pairs) > 1 and len(pairs) <= 16 and all(t.is_cuda and t.dtype == torch.float32 and t.dim() == 0 for t in ts):
        return _WeightedSum.apply(tuple(float(w) for w, _ in pairs), *ts)
    loss = ts[0] if pairs[0][0] == 1 else pairs[0][0] * ts[0]
    for w, t in pairs[1:]:
        loss = loss + w * t
    return loss


class _SplitRows(torch.autograd.Function):
    """x[:n0], x[n0:n0+n1], ... as views; backward = one concatenation of the pieces' gradients (instead of a zero-filled
    full-size tensor, a copy and an add per piece)."""

    @staticmethod
    def forward(ctx, x, *sizes):
        ctx.sizes, ctx.meta = sizes, (x.shape, x.dtype, x.device)
        outs, o = [], 0
        for n in sizes:
            outs.append(x.narrow(0, o, n))
            o += n
        return tuple(outs)

    @staticmethod
    def backward(ctx, *gs):
        shape, dtype, dev = ctx.meta
        parts = [g if g is not None else torch.zeros((n,) + tuple(shape[1:]), dtype=dtype, device=dev) for g, n in zip(gs, ctx.sizes)]
        rest = shape[0] - sum(ctx.sizes)
        if rest:
            parts.append(torch.zeros((rest,) + tuple(shape[1:]), dtype=dtype, device=dev))
        return (torch.cat(parts, dim=0),) + (None,) * len(ctx.sizes)


def semantic_losses(model, ctx, tx, tx_interp, tx_exc, epoch, measure=None, interp_measure=None, loss_fn=None,
                    draw_factor=None, exc_choice=None):
    """All loss terms of one iteration (reference :129-389).  Returns (total, dict of terms).

    The reference runs the network three times per iteration (reconstruction :131-133, interpolation :221-227,
    exchange :320-321).  The three passes are the same encode -> (scale some part latents) -> decode with different
    inputs, so they run here as ONE pass over the concatenated batch: a third of the launches, three times the rows per
    launch, and every parameter gradient is produced once instead of accumulated three times."""
    o = ctx.opts
    loss_fn = _as_loss(loss_fn)
    terms = {}
    do_interp = epoch > o.interp_epoch and tx_interp is not None
    do_exc = epoch > o.exc_epoch and tx_exc is not None
    B0 = tx.shape[0]
    # one pass of the network over [reconstruction | interpolation | exchange] (below) - and ONE joint regression over the
    # same concatenation instead of one per input (the regression is per batch entry: the same bits)
    xs = [tx] + ([tx_interp] if do_interp else []) + ([tx_exc] if do_exc else [])
    X = torch.cat(xs, dim=0) if len(xs) > 1 else tx
    kps_all = ctx.joints(X)
    kps_GT = kps_all[:B0]
    ks = [kps_GT[:, ctx.kps_keep_t]]

    if do_interp:
        Bi = tx_interp.shape[0]
        kps_i = kps_all[B0:B0 + Bi]
        if o.editskl_flag:
            n = len(ctx.skl_keep) if o.edit_mode == "rand" else 1
            f = torch.rand(n).to(ctx.device) * o.factor[0] + o.factor[1]
            skl = part_losses.kps2skl(kps_i, "ori_m", o.newskl_list)
            skl[:, ctx.skl_keep_t, 3] = skl[:, ctx.skl_keep_t, 3] * (f[None] if n > 1 else f)
            new_kps_i = part_losses.skl2kps(skl, "ori_m", o.newskl_list)
        else:
            new_kps_i = kps_i[:, ctx.kps_keep_t]
        part_index, a = _edit_scales(ctx, o, Bi, epoch, interp_measure, draw_factor)
        scale = _full_scale(ctx, part_index, a, Bi)
        ks.append(new_kps_i)

    if do_exc:
        Be = tx_exc.shape[0]
        kps_e = kps_all[X.shape[0] - Be:]
        mode = o.exc_mode
        if mode == "ori_or_m":
            pick = (np.random.rand(1) > 0.5) if exc_choice is None else (exc_choice == "ori")
            mode = "ori" if pick else "m"
        if mode == "ori_m":
            new_kps_e = torch.flip(kps_e, dims=[0])[:, ctx.kps_keep_t]
            exc_kind = "ori_m"
        else:
            skl = part_losses.kps2skl(kps_e, "ori_m", o.newskl_list)
            if mode == "ori":
                skl[:, ctx.newskl_keep_t, :3] = torch.flip(skl[:, ctx.newskl_keep_t, :3], dims=[0])
            else:
                skl[:, ctx.skl_keep_t, 3] = torch.flip(skl[:, ctx.skl_keep_t, 3], dims=[0])
            new_kps_e = part_losses.skl2kps(skl, "ori_m", o.newskl_list)
            exc_kind = mode
        ks.append(new_kps_e)

    K = torch.cat(ks, dim=0) if len(ks) > 1 else ks[0]
    latent, latent_kps, dummy = model.encode(X, K)
    lat_in = latent
    if do_interp:                                              # scale the edited parts' latents of the interpolation rows
        full = torch.ones((X.shape[0], latent.shape[1]), device=ctx.device, dtype=latent.dtype)
        full[B0:B0 + Bi] = scale
        lat_in = latent * full[:, :, None]
    rec = model.decode(lat_in, latent_kps, dummy)
    tx_zpart = latent[:B0]
    pieces = _SplitRows.apply(rec, *([B0] + ([Bi] if do_interp else []) + ([Be] if do_exc else [])))
    tx_hat = pieces[0]
    ctx.last_tx_hat = tx_hat.detach()                          # reconstruction of the training batch (save_recons, :459-470)
    if do_interp:
        rec_interp = pieces[1]
    if do_exc:
        rec_exc = pieces[-1]

    terms["rec_loss"] = loss_fn(tx, tx_hat)
    weighted = [(1.0, terms["rec_loss"])]                      # `loss = loss + w * term`, summed in this order at the end
    if epoch > o.edgereg_epoch and o.edgereg_w > 0:
        if ctx.face_tables is None:
            ctx.face_tables = losses.FaceTables(ctx.f_np, tx.shape[1], ctx.device)
        terms["edgereg_loss"] = losses.edge_ratio_loss(tx_hat, tx, ctx.face_tables)
        weighted.append((o.edgereg_w, terms["edgereg_loss"]))
    if epoch > o.zpartreg_epoch and o.zpartreg_w > 0 and measure is not None:
        terms["zpartreg_loss"] = part_losses.zpart_regulariser_fused(tx_zpart, measure, ctx.part_index_in_allpart,
                                                                     ctx.part_index_in_measure, o.relat_flag)
        weighted.append((o.zpartreg_w, terms["zpartreg_loss"]))

    if do_interp:
        if o.interp_kps_w > 0:
            terms["interp_kps_loss"] = part_losses.joint_l1_loss(rec_interp, new_kps_i, ctx.J, ctx.kps_keep_i32)
            weighted.append((o.interp_kps_w, terms["interp_kps_loss"]))
        if o.interp_euc_w > 0:
            terms["interp_euc_loss"] = part_losses.part_pairdist_loss(rec_interp, tx_interp, kps_i, ctx.tables, scale=scale,
                                                                      w_mode=o.w_mode, w_threshold=o.w_threshold,
                                                                      relat=o.relat_flag, skl_list=o.skl_list)
            weighted.append((o.interp_euc_w, terms["interp_euc_loss"]))

    if do_exc:
        if epoch > o.vol_epoch and o.vol_w > 0 and exc_kind == "ori":
            if ctx.part_faces is None or ctx.part_faces.n_rows != rec_exc.shape[1]:
                ctx.part_faces = part_losses.PartFaceTables(ctx.f_np, ctx.fpi.cpu().numpy(), ctx.part_index_in_allpart, rec_exc.shape[1],
                                                            ctx.device)
            terms["vol_loss"] = part_losses.part_volume_loss_fused(rec_exc, tx_exc, ctx.part_faces)
            weighted.append((o.vol_w, terms["vol_loss"]))
        if o.exc_kps_w > 0:
            terms["exc_kps_loss"] = part_losses.joint_l1_loss(rec_exc, new_kps_e, ctx.J, ctx.kps_keep_i32)
            weighted.append((o.exc_kps_w, terms["exc_kps_loss"]))
        if o.exc_euc_w > 0:
            terms["exc_euc_loss"] = part_losses.part_pairdist_loss(rec_exc, tx_exc, kps_e, ctx.tables, scale=None, w_mode=o.w_mode,
                                                                   w_threshold=o.w_threshold, relat=o.relat_flag,
                                                                   skl_list=o.skl_list)
            weighted.append((o.exc_euc_w, terms["exc_euc_loss"]))
    return weighted_sum(weighted), terms


class _Cycler:
    """`dataloader_interp_iter.next()` with the reference's restart behaviour (:154-158): when the
    loader is exhausted it is restarted and the FIRST batch of the new pass is skipped."""

    def __init__(self, loader):
        self.loader, self.it = loader, iter(loader)

    def next(self):
        try:
            return next(self.it)
        except StopIteration:
            self.it = iter(self.loader)
            next(self.it)
            return next(self.it)


TAGS = ["rec_loss", "edgereg_loss", "zpartreg_loss", "vol_loss", "interp_kps_loss", "interp_euc_loss", "exc_kps_loss",
        "exc_euc_loss"]


def train_autoencoder_dataloader_nonormal(dataloader_train, dataloader_val, device, model, optim, loss_fn,
                                          start_epoch, n_epochs, eval_freq, dataloader_interp, scheduler,
                                          writer, shapedata, metadata_dir, samples_dir, checkpoint_path,
                                          J_regressor, vert_part_index_dict, partname_list, save_recons=False,
                                          *, options: SemanticTrainOptions = None, reducer=None, verbose=True):
    opts = options or SemanticTrainOptions()
    ctx = SemanticContext(opts, shapedata, J_regressor, vert_part_index_dict, partname_list, device)
    loss_fn = _as_loss(loss_fn)
    import torch.distributed as dist
    # data-parallel (the reference is single-process): every rank iterates its own shard of each loader, the gradient
    # all-reduce is the reducer's; epoch losses are summed over ranks, rank 0 logs / writes checkpoints and samples
    world = dist.get_world_size() if (reducer is not None and dist.is_initialized()) else 1
    rank = dist.get_rank() if world > 1 else 0
    if rank != 0:
        writer, verbose = None, False
    total_steps = (start_epoch - 1) * len(dataloader_train)
    eval_freq = len(dataloader_train)
    cyc = None
    history = []
    for epoch in range(start_epoch, n_epochs + 1):
        model.train()
        if cyc is None:
            cyc = _Cycler(dataloader_interp)
        tloss = torch.zeros((), device=device)
        n_seen, n_val = 0, 0
        tx = tx_hat = None
        tx_idx = [0]
        for b, sample in enumerate(dataloader_train):
            optim.zero_grad()
            tx = sample["verts"].to(device)
            measure = sample["measure"].to(device) if "measure" in sample else None
            inp_i = cyc.next() if epoch > opts.interp_epoch else None
            inp_e = cyc.next() if epoch > opts.exc_epoch else None
            loss, terms = semantic_losses(model, ctx, tx,
                                          None if inp_i is None else inp_i["verts"].to(device),
                                          None if inp_e is None else inp_e["verts"].to(device), epoch, measure,
                                          None if inp_i is None or "measure" not in inp_i else inp_i["measure"].to(device), loss_fn)
            tx_hat = getattr(ctx, "last_tx_hat", None)
            if reducer is not None:
                reducer.prepare()
            loss.backward()
            if reducer is not None:
                reducer.finish()
            optim.step()
            n_seen += tx.shape[0]
            tloss += tx.shape[0] * loss.detach()
            if writer and total_steps % eval_freq == 0:
                writer.add_scalar("loss/loss/data_loss", loss.item(), total_steps)
                for t in TAGS:
                    writer.add_scalar("loss/loss/" + t, float(terms[t]) if t in terms else 0.0, total_steps)
            total_steps += 1

        model.eval()
        vloss = torch.zeros((), device=device)
        with torch.no_grad():
            for sample in dataloader_val:
                txv = sample["verts"].to(device)
                if "idx" in sample:
                    tx_idx = sample["idx"]
                kps = ctx.joints(txv)
                tx_hat_val = model(txv, kps[:, ctx.kps_keep_t])[0]
                n_val += txv.shape[0]
                vloss += txv.shape[0] * loss_fn(txv[:, :-1, :], tx_hat_val[:, :-1, :])
        if scheduler:
            scheduler.step()
        if world > 1:
            # sums of losses and of sample counts over all ranks: one tiny collective per epoch
            t = torch.stack([tloss.double(), vloss.double(), torch.tensor(float(n_seen), device=device, dtype=torch.float64),
                             torch.tensor(float(n_val), device=device, dtype=torch.float64)])
            dist.all_reduce(t)
            epoch_tloss = float(t[0]) / max(1.0, float(t[2]))
            tot_val = float(t[3])
            vloss = t[1]
        else:
            epoch_tloss = float(tloss) / float(len(dataloader_train.dataset))
            tot_val = float(len(dataloader_val.dataset))
        epoch_vloss = float(vloss) / tot_val if tot_val > 0 else None
        if writer:
            writer.add_scalar("avg_epoch_train_loss", epoch_tloss, epoch)
            if epoch_vloss is not None:
                writer.add_scalar("avg_epoch_valid_loss", epoch_vloss, epoch)
        if verbose:
            print("epoch {0} | tr {1} | val {2}".format(epoch, epoch_tloss, epoch_vloss))
        history.append((epoch, epoch_tloss, epoch_vloss))
        if epoch % opts.ck_frequency == 0 and metadata_dir is not None:
            if rank == 0:
                save_checkpoint(os.path.join(metadata_dir, checkpoint_path + "%s.pth.tar" % epoch), epoch, model, optim, scheduler)
            if world > 1:
                dist.barrier()            # nobody runs ahead of (or reads) a checkpoint that is still being written
        if save_recons and epoch % 50 == 0 and rank == 0 and samples_dir is not None and tx is not None and tx_hat is not None:
            # reference :459-470: first mesh of the epoch's LAST training batch, ground truth and reconstruction, both under
            # the sample index of the last validation batch's first mesh
            ind = [int(tx_idx[0])]
            shapedata.save_meshes(os.path.join(samples_dir, "epoch{0}_GT".format(epoch)), tx[0:1, 0:-1, :].detach().cpu().numpy(), ind)
            shapedata.save_meshes(os.path.join(samples_dir, "epoch{0}_rec".format(epoch)), tx_hat[0:1, 0:-1, :].detach().cpu().numpy(), ind)
    if verbose:
        print("~FIN~")
    return history

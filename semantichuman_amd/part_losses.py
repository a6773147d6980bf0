"""Losses and geometry helpers of the semantic training loop.

  part_pairdist_loss      train_funcs.py:243-284 / :353-389   ONE HIP kernel pair (fwd + bwd) instead of
                          17 x ([B,n,n,3] direction tensors + [B,n,n] distance/weight matrices)
  part_volume_loss        train_funcs.py:56-71 (+ per-sample loop :323-329), batched torch ops (small)
  kps2skl / skl2kps       utils_SH.py:26-84 (joints <-> unit bone vector + length), torch ops (tiny)
  bone_directions         the per-part bone vector utils_SH.angle_skl builds from cfg.CONSTANTS.skl_list
  zpart_regulariser       train_funcs.py:145-152
"""
from __future__ import annotations

import numpy as np
import torch

from . import _lib, constants
from ._lib import check, ptr, stream_ptr

W_MODES = {"all_one": 0, "linear": 1, "sin": 2, "threshold": 3}


class PartTables:
    """Device tables for the part losses: parts as CSR over level-0 vertices (must be disjoint),
    per-part weights / flags, row tiles of the pair kernel."""

    def __init__(self, vert_part_index_dict, device, leaf_parts=(0, 7, 10, 13, 16), w_part=None):
        parts = [np.asarray(v, dtype=np.int32) for v in vert_part_index_dict.values()]
        allv = np.concatenate(parts)
        if np.unique(allv).size != allv.size:
            raise ValueError("part vertex lists overlap; the pair-loss gradient assumes disjoint parts")
        self.P = len(parts)
        self.sizes = [len(p) for p in parts]
        self.max_part = max(self.sizes)
        rows = _lib.load().sh_part_pairdist_tile_rows()
        ptr_ = np.zeros(self.P + 1, np.int32)
        np.cumsum(self.sizes, out=ptr_[1:])
        tile = np.zeros(self.P + 1, np.int32)
        np.cumsum([-(-n // rows) for n in self.sizes], out=tile[1:])
        self.T = int(tile[-1])
        flags = np.zeros(self.P, np.int32)
        flags[list(leaf_parts)] = 1
        wp = np.full(self.P, 1.0 / self.P, np.float32) if w_part is None else np.asarray(w_part, np.float32)   # '1/K'
        t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(device)
        self.part_ptr, self.part_vert, self.tile_ptr = t(ptr_), t(allv), t(tile)
        self.flags, self.w_part = t(flags), t(wp)
        self.device = torch.empty(0, device=device).device


class _PairDist(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x_rec, x_gt, bone, scale, tb: PartTables, w_mode, thr, relat):
        x_rec, x_gt, bone = x_rec.contiguous(), x_gt.contiguous(), bone.contiguous()
        scale = None if scale is None else scale.contiguous()
        B, N1, _ = x_rec.shape
        dev = x_rec.device
        loss = torch.empty((), dtype=torch.float32, device=dev)
        psum = torch.empty(tb.P, dtype=torch.float32, device=dev)
        pcnt = torch.empty(tb.P, dtype=torch.float32, device=dev)
        ws = torch.empty(B * tb.T * 2, dtype=torch.float32, device=dev)
        check(_lib.load().sh_part_pairdist_loss_fwd(ptr(x_rec), ptr(x_gt), ptr(bone), ptr(scale), ptr(tb.part_ptr), ptr(tb.part_vert),
                                                    ptr(tb.tile_ptr), ptr(tb.flags), ptr(tb.w_part), B, N1, tb.P, tb.T, tb.max_part,
                                                    w_mode, thr, int(relat), ptr(loss), ptr(psum), ptr(pcnt), ptr(ws), ws.numel() * 4,
                                                    stream_ptr()), "sh_part_pairdist_loss_fwd")
        ctx.save_for_backward(x_rec, x_gt, bone, pcnt) if scale is None else ctx.save_for_backward(x_rec, x_gt, bone, pcnt, scale)
        ctx.args = (tb, w_mode, thr, relat, scale is not None)
        return loss

    @staticmethod
    def backward(ctx, g):
        tb, w_mode, thr, relat, has_scale = ctx.args
        saved = ctx.saved_tensors
        x_rec, x_gt, bone, pcnt = saved[:4]
        scale = saved[4] if has_scale else None
        B, N1, _ = x_rec.shape
        grad = torch.empty_like(x_rec)
        check(_lib.load().sh_part_pairdist_loss_bwd(ptr(x_rec), ptr(x_gt), ptr(bone), ptr(scale), ptr(tb.part_ptr), ptr(tb.part_vert),
                                                    ptr(tb.tile_ptr), ptr(tb.flags), ptr(tb.w_part), B, N1, tb.P, tb.T, tb.max_part,
                                                    w_mode, thr, int(relat), ptr(pcnt), ptr(g.contiguous()), ptr(grad), stream_ptr()),
              "sh_part_pairdist_loss_bwd")
        return grad, None, None, None, None, None, None, None


_INDEX_CACHE = {}


def index_tensor(values, device):
    """Device-resident int64 index tensor of a Python index list, cached: indexing a device tensor with a list uploads
    the list on every call (a synchronous host-to-device copy, and illegal inside a hipGraph capture)."""
    if torch.is_tensor(values):
        return values
    key = (tuple(int(v) for v in values), str(device))
    t = _INDEX_CACHE.get(key)
    if t is None:
        t = torch.tensor(key[0], dtype=torch.int64, device=device)
        _INDEX_CACHE[key] = t
    return t


_BONE_INDEX_CACHE = {}


def _bone_index(skl_list, device):
    """Index tensors (head, tail, second tail or tail again) of a bone list - one gather instead of a Python loop."""
    key = (id(skl_list), len(skl_list), str(device))
    hit = _BONE_INDEX_CACHE.get(key)
    if hit is None or hit[0] is not skl_list:
        idx = [torch.tensor([b[0] for b in skl_list], device=device), torch.tensor([b[1] for b in skl_list], device=device),
               torch.tensor([b[2] if len(b) == 3 else b[1] for b in skl_list], device=device)]
        hit = (skl_list, idx)
        _BONE_INDEX_CACHE[key] = hit
    return hit[1]


def bone_directions(kps, skl_list=None):
    """[B, P, 3] bone vector of each part from the FULL joint set (utils_SH.py:449-452):
    joint a - joint b, or joint a - mean(joint b, joint c)."""
    skl_list = constants.SKL_LIST if skl_list is None else skl_list
    i0, i1, i2 = _bone_index(skl_list, kps.device)
    return kps[:, i0, :] - (kps[:, i1, :] + kps[:, i2, :]) / 2          # two-joint bones have i2 == i1: (a + a) / 2 == a exactly


def part_pairdist_loss(x_rec, x_gt, kps_gt, tables: PartTables, scale=None, w_mode="threshold", w_threshold=0.8, relat=True,
                       skl_list=None):
    """x_rec, x_gt: [B, N(+1), 3]; kps_gt: full joint set [B, J, 3] of the ground truth; scale: [B, P]
    factor applied to the ground-truth distances of edited parts (1 elsewhere) or None."""
    if w_mode not in W_MODES:
        raise NotImplementedError(w_mode)
    bone = bone_directions(kps_gt.detach(), skl_list)
    return _PairDist.apply(x_rec, x_gt.detach(), bone, None if scale is None else scale.detach(), tables, W_MODES[w_mode],
                           float(w_threshold), bool(relat))


# ----------------------------------------------------------------------------------------------
def face_part_index(faces, vert_part_index_dict, n_verts):
    """train_funcs.py:80-89: a face belongs to part k if all three corners do, else to no part (100)."""
    vpi = np.ones(n_verts, np.int64)
    for k, v in enumerate(vert_part_index_dict.values()):
        vpi[np.asarray(v)] = k
    f = np.asarray(faces)
    same = (vpi[f[:, 0]] == vpi[f[:, 1]]) & (vpi[f[:, 0]] == vpi[f[:, 2]])
    return np.where(same, vpi[f[:, 0]], 100)


def part_volume_loss(x_rec, x_gt, faces, fpi, parts):
    """train_funcs.py:56-71 averaged over the batch (:323-329): mean over `parts` of
    | |vol_rec / vol_gt| - 1 |, vol = sum over the part's faces of (a x b) . c.
    x_*: [B, N, 3] (dummy row already dropped); faces long [F,3]; fpi long [F] from face_part_index."""
    def signed(x):
        a, b, c = x[:, faces[:, 0]], x[:, faces[:, 1]], x[:, faces[:, 2]]
        return (torch.cross(a, b, dim=2) * c).sum(2)                       # [B, F]
    vr, vg = signed(x_rec), signed(x_gt)
    # per-part sums of the face volumes as one product with the [F, parts] membership matrix
    member = (fpi[:, None] == index_tensor(parts, fpi.device)[None, :]).to(vr.dtype)
    rk, gk = vr @ member, vg @ member                                       # [B, parts]
    return (torch.abs(rk / gk) - torch.abs(gk / gk)).abs().mean(0).sum() / len(parts)


def zpart_regulariser(z_part, measure, part_idx, measure_idx, relat=True):
    """train_funcs.py:145-152: L1 between the norm of each part latent and the part's girth."""
    zm = torch.sqrt(torch.sum(z_part ** 2, dim=2))
    part_idx, measure_idx = index_tensor(part_idx, zm.device), index_tensor(measure_idx, zm.device)
    if relat:
        return (zm[:, part_idx] / measure[:, measure_idx] - 1).abs().mean()
    return (zm[:, part_idx] - measure[:, measure_idx]).abs().mean()


def kps2skl(kps_tmp, skl_mode="ori_m", newskl_list=None):
    """utils_SH.py:26-69."""
    skl_list = constants.NEWSKL_LIST if newskl_list is None else newskl_list
    if kps_tmp.shape[1] == len(skl_list) + 4:
        kps = kps_tmp.clone()
    else:
        kps = torch.zeros((kps_tmp.shape[0], len(skl_list) + 4, 3), device=kps_tmp.device)
        kps[:, index_tensor(constants.kps_keep(skl_list), kps_tmp.device), :] = kps_tmp
    i0, i1, i2 = _bone_index(skl_list, kps.device)
    vec = kps[:, i0, :] - (kps[:, i1, :] + kps[:, i2, :]) / 2            # [B, n_bones, 3]; i2 == i1 for two-joint bones
    n = torch.sqrt(torch.sum(vec ** 2, dim=2, keepdim=True))
    if skl_mode in ("ori_m", "kps_ori_m"):
        return torch.cat([vec / n, n], dim=2)
    if skl_mode == "vec_m":
        return torch.cat([vec, n], dim=2)
    if skl_mode == "vec":
        return vec
    if skl_mode == "m":
        return n
    raise NotImplementedError(skl_mode)


_SKL_LEVELS_CACHE = {}


def _skl_levels(skl_list, device):
    """Bones grouped by depth in the kinematic tree, in the order the reference's loop resolves them
    (utils_SH.py:77-83: joint b[1] = joint b[0] - bone, joints not yet assigned count as the origin)."""
    key = (id(skl_list), len(skl_list), str(device))
    hit = _SKL_LEVELS_CACHE.get(key)
    if hit is None or hit[0] is not skl_list:
        n_j = len(skl_list) + 4
        depth_of_joint, levels = {}, {}
        for k, b in enumerate(skl_list):
            d = depth_of_joint[b[0]] + 1 if b[0] in depth_of_joint else 0
            depth_of_joint[b[1]] = d
            levels.setdefault(d, []).append((b[1], b[0] if b[0] in depth_of_joint and depth_of_joint[b[0]] < d else n_j, k))
        out = []
        for d in sorted(levels):
            c, p, k = zip(*levels[d])
            out.append(tuple(torch.tensor(t, device=device) for t in (c, p, k)))
        hit = (skl_list, out)
        _SKL_LEVELS_CACHE[key] = hit
    return hit[1]


def skl2kps(skl, skl_mode="ori_m", newskl_list=None):
    """utils_SH.py:71-84: rebuild joints from the root outwards (joint b[1] = joint b[0] - bone), one tree level at a
    time - the same subtraction per joint as the reference's bone-by-bone loop."""
    skl_list = constants.NEWSKL_LIST if newskl_list is None else newskl_list
    if skl_mode == "vec":
        bone = skl
    elif skl_mode == "vec_m":
        bone = skl[:, :, :3]
    elif skl_mode in ("ori_m", "kps_ori_m"):
        bone = skl[:, :, :3] * skl[:, :, 3:]
    else:
        raise NotImplementedError(skl_mode)
    n_j = len(skl_list) + 4
    kps = torch.zeros((skl.shape[0], n_j + 1, 3), device=skl.device, dtype=skl.dtype)      # slot n_j: the origin
    for child, parent, k in _skl_levels(skl_list, skl.device):
        kps = kps.index_copy(1, child, kps[:, parent, :] - bone[:, k, :])
    return kps[:, index_tensor(constants.kps_keep(skl_list), kps.device), :]

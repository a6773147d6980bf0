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
    def forward(ctx, x_rec, x_gt, bone, scale, tb: PartTables, w_mode, thr, relat, want_grad=True):
        x_rec, x_gt, bone = x_rec.contiguous(), x_gt.contiguous(), bone.contiguous()
        scale = None if scale is None else scale.contiguous()
        B, N1, _ = x_rec.shape
        dev = x_rec.device
        loss = torch.empty((), dtype=torch.float32, device=dev)
        psum = torch.empty(tb.P, dtype=torch.float32, device=dev)
        pcnt = torch.empty(tb.P, dtype=torch.float32, device=dev)
        ws = torch.empty(B * tb.T * 2, dtype=torch.float32, device=dev)
        # when the reconstruction carries a gradient the same sweep leaves the backward pass's row sums (everything but the
        # factor 2 g w_p / count_p): backward is then one scaling launch, not a second sweep over the pairs - the same bits
        # (want_grad: the caller's grad mode - inside forward() it is always off, and needs_input_grad stays true under no_grad)
        graw = torch.empty_like(x_rec) if (ctx.needs_input_grad[0] and want_grad) else None
        check(_lib.load().sh_part_pairdist_loss_fwd_grad(ptr(x_rec), ptr(x_gt), ptr(bone), ptr(scale), ptr(tb.part_ptr), ptr(tb.part_vert),
                                                         ptr(tb.tile_ptr), ptr(tb.flags), ptr(tb.w_part), B, N1, tb.P, tb.T, tb.max_part,
                                                         w_mode, thr, int(relat), ptr(loss), ptr(psum), ptr(pcnt), ptr(graw), ptr(ws),
                                                         ws.numel() * 4, stream_ptr()), "sh_part_pairdist_loss_fwd_grad")
        if graw is not None:
            ctx.save_for_backward(graw, pcnt)
            ctx.args = (tb, None)
            return loss
        ctx.save_for_backward(x_rec, x_gt, bone, pcnt) if scale is None else ctx.save_for_backward(x_rec, x_gt, bone, pcnt, scale)
        ctx.args = (tb, w_mode, thr, relat, scale is not None)
        return loss

    @staticmethod
    def backward(ctx, g):
        if ctx.args[1] is None:                                     # the forward pass left the row sums
            tb = ctx.args[0]
            graw, pcnt = ctx.saved_tensors
            B, N1, _ = graw.shape
            grad = torch.empty_like(graw)
            check(_lib.load().sh_part_pairdist_loss_bwd_scale(ptr(graw), ptr(tb.part_ptr), ptr(tb.part_vert), ptr(tb.w_part), ptr(pcnt),
                                                              ptr(g.contiguous()), B, N1, tb.P, int(tb.part_vert.numel()), ptr(grad),
                                                              stream_ptr()), "sh_part_pairdist_loss_bwd_scale")
            return grad, None, None, None, None, None, None, None, None
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
        return grad, None, None, None, None, None, None, None, None


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


_BONE_I32_CACHE = {}
_SKL_MODES = {"ori_m": 0, "kps_ori_m": 0, "vec_m": 1, "vec": 2, "m": 3}


def _bone_i32(skl_list, device):
    """(head, tail, second tail or tail again) of a bone list as int32 device tensors for the kernels."""
    key = (id(skl_list), len(skl_list), str(device))
    hit = _BONE_I32_CACHE.get(key)
    if hit is None or hit[0] is not skl_list:
        idx = [torch.tensor(v, dtype=torch.int32, device=device) for v in
               ([b[0] for b in skl_list], [b[1] for b in skl_list], [b[2] if len(b) == 3 else b[1] for b in skl_list])]
        hit = (skl_list, idx)
        _BONE_I32_CACHE[key] = hit
    return hit[1]


def _kernel_ok(t):
    """The skeleton kernels serve what the loops hand them: contiguous fp32 HIP tensors that carry no gradient."""
    return t.is_cuda and t.dtype == torch.float32 and t.is_contiguous() and not t.requires_grad


def _kps2skl_kernel(kps, skl_list, mode):
    B, J = kps.shape[0], kps.shape[1]
    i0, i1, i2 = _bone_i32(skl_list, kps.device)
    out = torch.empty((B, len(skl_list), (4, 4, 3, 1)[mode]), dtype=torch.float32, device=kps.device)
    check(_lib.load().sh_kps2skl(ptr(kps), B, J, ptr(i0), ptr(i1), ptr(i2), len(skl_list), mode, ptr(out), stream_ptr()), "sh_kps2skl")
    return out


def bone_directions(kps, skl_list=None):
    """[B, P, 3] bone vector of each part from the FULL joint set (utils_SH.py:449-452):
    joint a - joint b, or joint a - mean(joint b, joint c)."""
    skl_list = constants.SKL_LIST if skl_list is None else skl_list
    if _kernel_ok(kps):
        return _kps2skl_kernel(kps, skl_list, 2)                        # one launch instead of five
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
                           float(w_threshold), bool(relat), torch.is_grad_enabled())


# ----------------------------------------------------------------------------------------------
def face_part_index(faces, vert_part_index_dict, n_verts):
    """train_funcs.py:80-89: a face belongs to part k if all three corners do, else to no part (100)."""
    vpi = np.ones(n_verts, np.int64)
    for k, v in enumerate(vert_part_index_dict.values()):
        vpi[np.asarray(v)] = k
    f = np.asarray(faces)
    same = (vpi[f[:, 0]] == vpi[f[:, 1]]) & (vpi[f[:, 0]] == vpi[f[:, 2]])
    return np.where(same, vpi[f[:, 0]], 100)


# ---------------------------------------------------------------------------------------------- a13 as kernels
def _i32(values, device):
    key = ("i32", tuple(int(v) for v in values), str(device))
    t = _INDEX_CACHE.get(key)
    if t is None:
        t = torch.tensor(key[1], dtype=torch.int32, device=device)
        _INDEX_CACHE[key] = t
    return t


def _mesh3(x, what):
    if not (x.is_cuda and x.dtype == torch.float32 and x.dim() == 3 and x.shape[2] == 3 and x.is_contiguous()):
        raise RuntimeError("semantichuman_amd.%s needs contiguous fp32 HIP meshes [B, rows, 3] (got %s %s %s); there is no CPU path"
                           % (what, x.device, x.dtype, tuple(x.shape)))
    return x


def joint_regress(x, J):
    """kps = J @ x[:, :N] for x [B, rows >= N, 3], J [K, N] (train_funcs.py:131): one kernel, no gradient."""
    from . import _lib
    x = _mesh3(x.detach().contiguous(), "joint_regress")
    K, N = J.shape
    out = torch.empty((x.shape[0], K, 3), dtype=torch.float32, device=x.device)
    _lib.check(_lib.load().sh_joint_regress(_lib.ptr(x), x.shape[1] * 3, _lib.ptr(J), x.shape[0], N, K, _lib.ptr(out), _lib.stream_ptr()),
               "sh_joint_regress")
    return out


class _JointL1(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, target, J, keep):
        from . import _lib
        x = _mesh3(x.contiguous(), "joint_l1_loss")
        target = target.detach().contiguous().float()
        K, N = J.shape
        B, Kk = x.shape[0], keep.numel()
        kps = torch.empty((B, K, 3), dtype=torch.float32, device=x.device)
        loss = torch.empty((), dtype=torch.float32, device=x.device)
        _lib.check(_lib.load().sh_joint_l1_loss_fwd(_lib.ptr(x), x.shape[1] * 3, _lib.ptr(J), _lib.ptr(keep), _lib.ptr(target), B, N, K, Kk,
                                                    _lib.ptr(kps), _lib.ptr(loss), _lib.stream_ptr()), "sh_joint_l1_loss_fwd")
        ctx.save_for_backward(kps, target, J, keep)
        ctx.rows = x.shape[1]
        return loss

    @staticmethod
    def backward(ctx, g):
        from . import _lib
        kps, target, J, keep = ctx.saved_tensors
        K, N = J.shape
        B, Kk = kps.shape[0], keep.numel()
        grad = torch.empty((B, ctx.rows, 3), dtype=torch.float32, device=kps.device)
        g = g.contiguous().float()
        _lib.check(_lib.load().sh_joint_l1_loss_bwd(_lib.ptr(kps), _lib.ptr(keep), _lib.ptr(target), _lib.ptr(J), B, ctx.rows, N, K, Kk,
                                                    _lib.ptr(g), _lib.ptr(grad), _lib.stream_ptr()), "sh_joint_l1_loss_bwd")
        return grad, None, None, None


def joint_l1_loss(x_rec, target, J, keep):
    """mean | (J @ x_rec[:, :N])[:, keep] - target | (train_funcs.py:229-232, :335-342) as one forward launch pair and one
    backward launch; `keep`: int32 device tensor or index list."""
    if not torch.is_tensor(keep):
        keep = _i32(keep, x_rec.device)
    return _JointL1.apply(x_rec, target, J, keep)


class PartFaceTables:
    """Faces grouped by body part for the volume loss: CSR over the parts in `parts` (train_funcs.py:58-61: the faces whose
    three corners lie in one part), the face -> slot map, and the vertex -> corner lists of the gradient."""

    def __init__(self, faces, fpi, parts, n_rows, device):
        f = np.ascontiguousarray(np.asarray(faces, dtype=np.int32))
        fpi = np.asarray(fpi)
        slot = np.full(f.shape[0], -1, dtype=np.int32)
        ptr, lst = [0], []
        for k, p in enumerate(parts):
            idx = np.nonzero(fpi == p)[0].astype(np.int32)
            slot[idx] = k
            lst.append(idx)
            ptr.append(ptr[-1] + len(idx))
        order = np.argsort(f.ravel(), kind="stable")
        vptr = np.zeros(n_rows + 1, dtype=np.int32)
        np.cumsum(np.bincount(f.ravel(), minlength=n_rows), out=vptr[1:])
        dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(device)          # noqa: E731
        self.P, self.n_rows = len(parts), n_rows
        self.faces, self.slot = dev(f), dev(slot)
        self.pf_ptr, self.pf = dev(np.asarray(ptr, dtype=np.int32)), dev(np.concatenate(lst) if lst else np.zeros(0, np.int32))
        self.vptr, self.vcorner = dev(vptr), dev(np.arange(f.size, dtype=np.int32)[order])


class _PartVolume(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x_rec, x_gt, pt: PartFaceTables):
        from . import _lib
        x_rec, x_gt = _mesh3(x_rec.contiguous(), "part_volume_loss"), _mesh3(x_gt.detach().contiguous(), "part_volume_loss")
        B = x_rec.shape[0]
        vol = torch.empty((2, B, pt.P), dtype=torch.float32, device=x_rec.device)
        loss = torch.empty((), dtype=torch.float32, device=x_rec.device)
        _lib.check(_lib.load().sh_part_volume_loss_fwd(_lib.ptr(x_rec), _lib.ptr(x_gt), x_rec.shape[1] * 3, _lib.ptr(pt.faces),
                                                       _lib.ptr(pt.pf_ptr), _lib.ptr(pt.pf), B, pt.P, _lib.ptr(vol), _lib.ptr(loss),
                                                       _lib.stream_ptr()), "sh_part_volume_loss_fwd")
        ctx.save_for_backward(x_rec, vol)
        ctx.pt = pt
        return loss

    @staticmethod
    def backward(ctx, g):
        from . import _lib
        x_rec, vol = ctx.saved_tensors
        pt = ctx.pt
        if x_rec.shape[1] != pt.n_rows:
            raise RuntimeError("part_volume_loss: tables were built for %d rows per mesh, got %d" % (pt.n_rows, x_rec.shape[1]))
        grad = torch.empty_like(x_rec)
        g = g.contiguous().float()
        _lib.check(_lib.load().sh_part_volume_loss_bwd(_lib.ptr(x_rec), x_rec.shape[1] * 3, _lib.ptr(pt.faces), _lib.ptr(pt.slot),
                                                       _lib.ptr(pt.vptr), _lib.ptr(pt.vcorner), _lib.ptr(vol), x_rec.shape[0], pt.P,
                                                       x_rec.shape[1], _lib.ptr(g), _lib.ptr(grad), _lib.stream_ptr()),
                   "sh_part_volume_loss_bwd")
        return grad, None, None


def part_volume_loss_fused(x_rec, x_gt, pt: PartFaceTables):
    """cal_volloss averaged over the batch (train_funcs.py:56-71, :323-330) on full meshes [B, rows, 3] (dummy row allowed)."""
    return _PartVolume.apply(x_rec, x_gt, pt)


class _ZPart(torch.autograd.Function):
    @staticmethod
    def forward(ctx, z, measure, pi, mi, relat):
        from . import _lib
        z, measure = z.contiguous().float(), measure.detach().contiguous().float()
        if not z.is_cuda:
            raise RuntimeError("semantichuman_amd.zpart_regulariser_fused needs HIP tensors; there is no CPU path")
        B, P, L = z.shape
        loss = torch.empty((), dtype=torch.float32, device=z.device)
        _lib.check(_lib.load().sh_zpart_reg(_lib.ptr(z), _lib.ptr(measure), _lib.ptr(pi), _lib.ptr(mi), B, P, L, measure.shape[1], pi.numel(),
                                            1 if relat else 0, _lib.ptr(loss), None, None, _lib.stream_ptr()), "sh_zpart_reg")
        ctx.save_for_backward(z, measure, pi, mi)
        ctx.relat = relat
        return loss

    @staticmethod
    def backward(ctx, g):
        from . import _lib
        z, measure, pi, mi = ctx.saved_tensors
        B, P, L = z.shape
        dz = torch.empty_like(z)
        g = g.contiguous().float()
        _lib.check(_lib.load().sh_zpart_reg(_lib.ptr(z), _lib.ptr(measure), _lib.ptr(pi), _lib.ptr(mi), B, P, L, measure.shape[1], pi.numel(),
                                            1 if ctx.relat else 0, None, _lib.ptr(dz), _lib.ptr(g), _lib.stream_ptr()), "sh_zpart_reg")
        return dz, None, None, None, None


def zpart_regulariser_fused(z_part, measure, part_idx, measure_idx, relat=True):
    """zpart_regulariser as one kernel each way (train_funcs.py:145-152)."""
    return _ZPart.apply(z_part, measure, _i32(part_idx, z_part.device), _i32(measure_idx, z_part.device), relat)


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
    if skl_mode not in _SKL_MODES:
        raise NotImplementedError(skl_mode)
    if kps_tmp.shape[1] == len(skl_list) + 4:
        if _kernel_ok(kps_tmp):
            return _kps2skl_kernel(kps_tmp, skl_list, _SKL_MODES[skl_mode])     # one launch instead of twelve, the same bits
        kps = kps_tmp.clone()
    else:
        kps = torch.zeros((kps_tmp.shape[0], len(skl_list) + 4, 3), device=kps_tmp.device)
        kps[:, index_tensor(constants.kps_keep(skl_list), kps_tmp.device), :] = kps_tmp
    if _kernel_ok(kps):
        return _kps2skl_kernel(kps, skl_list, _SKL_MODES[skl_mode])
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
    if skl_mode in ("ori_m", "kps_ori_m", "vec_m", "vec") and _kernel_ok(skl) and len(skl_list) + 4 <= 64:
        # the reference's bone-by-bone loop as ONE launch (a thread per batch entry walks the list) instead of five per tree level
        head, tail, _ = _bone_i32(skl_list, skl.device)
        keep = _i32(constants.kps_keep(skl_list), skl.device)
        out = torch.empty((skl.shape[0], keep.shape[0], 3), dtype=torch.float32, device=skl.device)
        check(_lib.load().sh_skl2kps(ptr(skl), skl.shape[0], len(skl_list), {"ori_m": 0, "kps_ori_m": 0, "vec_m": 1, "vec": 2}[skl_mode],
                                     ptr(head), ptr(tail), len(skl_list) + 4, ptr(keep), keep.shape[0], ptr(out), stream_ptr()), "sh_skl2kps")
        return out
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

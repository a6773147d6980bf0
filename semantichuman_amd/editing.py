"""Inference / editing front-end: the latent and skeleton edits of reference demo.py:64-113
and utils_SH.edit_skl (:412-440) as functions, plus the OBJ writer the demo uses (utils_SH.save_obj :163-195).

All edits are small tensor manipulations on latents / joints; the heavy part is `model.decode`, which runs the
HIP decoder stack.  `decode_edits` reproduces the demo's seven decodes for a (shape, skeleton, style) triple.
"""
from __future__ import annotations

import numpy as np
import torch

from . import constants
from .part_losses import kps2skl, skl2kps

# utils_SH.py:21-24 (SMPL-style kinematic tree of the 24 body joints)
PARENT = {1: 0, 2: 0, 3: 0, 4: 1, 5: 2, 6: 3, 7: 4, 8: 5, 9: 6, 10: 7, 11: 8, 12: 9, 13: 9, 14: 9, 15: 12, 16: 13, 17: 14,
          18: 16, 19: 17, 20: 18, 21: 19, 22: 20, 23: 21}
CHILDREN = {0: [1, 2, 3], 1: [4], 2: [5], 3: [6], 4: [7], 5: [8], 6: [9], 7: [10], 8: [11], 9: [12, 13, 14], 12: [15], 13: [16],
            14: [17], 16: [18], 17: [19], 18: [20], 19: [21], 20: [22], 21: [23]}


def edit_skl(kps, kps_index, edit_length):
    """utils_SH.py:412-440: scale the bone parent(kps_index) -> kps_index by `edit_length` [N] and move the whole
    sub-tree below the joint along with it.  kps [N, K, 3] -> new kps."""
    parent = kps[:, PARENT[kps_index], :]
    direction = kps[:, kps_index, :] - parent
    shift = direction * (torch.as_tensor(edit_length, dtype=kps.dtype, device=kps.device) - 1)[:, None]
    subtree, stack = [], [kps_index]
    while stack:
        i = stack.pop()
        subtree.append(i)
        stack.extend(CHILDREN.get(i, []))
    new = kps.clone()
    new[:, subtree, :] = new[:, subtree, :] + shift[:, None, :]
    return new


def edit_bone_orientation(skl, target_skl, bone_indices):
    """demo.py:79-81: take the unit directions of the chosen bones from another skeleton ('ori_m' layout [N, n_bones, 4])."""
    out = skl.clone()
    out[:, bone_indices, :3] = target_skl[:, bone_indices, :3]
    return out


def edit_bone_length(skl, bone_indices, factor):
    """demo.py:83-86: scale the lengths (4th component of the 'ori_m' layout) of the chosen bones."""
    out = skl.clone()
    out[:, bone_indices, 3] = out[:, bone_indices, 3] * factor
    return out


def edit_part_size(z, part_indices, factor):
    """demo.py:88: the norm of a part latent encodes the part's girth - scale it."""
    out = z.clone()
    out[:, part_indices, :] = out[:, part_indices, :] * factor
    return out


def edit_part_style(z, target_z, part_indices):
    """demo.py:90-95: keep each chosen part latent's norm, take its direction from `target_z`."""
    out = z.clone()
    for k in part_indices:
        norm = torch.sqrt(torch.sum(out[:, k, :] ** 2, dim=1, keepdim=True))
        tdir = target_z[:, k, :] / torch.sqrt(torch.sum(target_z[:, k, :] ** 2, dim=1, keepdim=True))
        out[:, k, :] = norm * tdir
    return out


def decode_edits(model, z, z_kps, tx, J_regressor, shape_idx, skl_idx, style_idx, bone_pairs, length_bones, parts,
                 length_factor=1.2, size_factor=1.2):
    """The edits of demo.py:64-103 for one (shape, skeleton-donor, style-donor) triple.
    z [n, 17, d], z_kps [n, 17, d] latents of the evaluated set, tx [n, N+1, 3] its meshes; `bone_pairs` are entries of
    NEWSKL_LIST whose orientation is transferred, `length_bones` indices whose length is scaled, `parts` part indices
    whose size / style is edited.  Returns a dict of decoded meshes [1, N+1, 3]."""
    dev = z.device
    J = torch.as_tensor(np.asarray(J_regressor, dtype=np.float32), device=dev)
    kps = torch.matmul(J, tx[:, :-1, :])
    skl = kps2skl(kps, "ori_m")
    sl = slice(shape_idx, shape_idx + 1)
    bone_idx = [constants.NEWSKL_LIST.index(list(b)) for b in bone_pairs]
    dummy = torch.zeros((1, 1, model.filters_enc[0][-1] if hasattr(model, "filters_enc") else 128), device=dev)
    with torch.no_grad():
        ori = skl2kps(edit_bone_orientation(skl[sl], skl[skl_idx:skl_idx + 1], bone_idx), "ori_m")
        length = skl2kps(edit_bone_length(skl[sl], length_bones, length_factor), "ori_m")
        out = {
            "rec_editpose": model.decode(z[sl], model.kps_encode(ori), dummy),
            "rec_editlength": model.decode(z[sl], model.kps_encode(length), dummy),
            "rec_editgirth": model.decode(edit_part_size(z[sl], parts, size_factor), z_kps[sl], dummy),
            "rec_editstyle": model.decode(edit_part_style(z[sl], z[style_idx:style_idx + 1], parts), z_kps[sl], dummy),
            "rec_shape": model.decode(z[sl], z_kps[sl], dummy),
            "rec_skl": model.decode(z[skl_idx:skl_idx + 1], z_kps[skl_idx:skl_idx + 1], dummy),
            "rec_style": model.decode(z[style_idx:style_idx + 1], z_kps[style_idx:style_idx + 1], dummy),
        }
    return out


def save_obj(obj_path, v, f, partcolor_list=None, vert_part_index=None):
    """utils_SH.py:163-195 without the skeleton overlay: 'v x y z r g b' lines (grey, or the part colour) + 1-based faces."""
    v = v.detach().cpu().numpy() if torch.is_tensor(v) else np.asarray(v)
    f = f.detach().cpu().numpy() if torch.is_tensor(f) else np.asarray(f)
    with open(obj_path, "w") as fp:
        for i, p in enumerate(v):
            c = (192, 192, 192) if partcolor_list is None or vert_part_index is None else partcolor_list[int(vert_part_index[i])]
            fp.write("v %f %f %f %d %d %d\n" % (p[0], p[1], p[2], c[0], c[1], c[2]))
        for t in f + 1:
            fp.write("f %d %d %d\n" % (t[0], t[1], t[2]))

"""Synthetic pose-window batches with the reference's data contract and the
forward-kinematics helper that produces ``target_pose`` on the GPU.

Reference skeleton: configs/mouse_skeleton.yaml:86-113 (18 joints, 6 chains).  BASELINE's
synthetic 23-joint skeleton appends one joint to chains 0,2,3,4,5 (build-defined: the
reference ships no 23-joint tree).
"""
from __future__ import annotations

import torch

from .. import ops
from .._lib import make_tree

MOUSE_KINEMATIC_TREE = [[0, 1, 2, 3, 4], [0, 5], [1, 6, 7, 8], [1, 9, 10, 11], [5, 12, 13, 14], [5, 15, 16, 17]]
MOUSE_OFFSET = [[0, 0, 0], [1, 0, 0], [1, 0, 0], [1, 0, 0], [1, 0, 0], [-1, 0, 0],
                [0, 1, 0], [0, 1, 0], [0, 1, 0], [0, -1, 0], [0, -1, 0], [0, -1, 0],
                [0, 1, 0], [0, 1, 0], [0, 1, 0], [0, -1, 0], [0, -1, 0], [0, -1, 0]]


def skeleton(n_keypts):
    """(kinematic_tree, unit offsets) for 18 (reference) or 23 (BASELINE synthetic) joints."""
    tree = [list(c) for c in MOUSE_KINEMATIC_TREE]
    offs = [list(o) for o in MOUSE_OFFSET]
    if n_keypts == 18:
        return tree, offs
    if n_keypts == 23:
        for chain, j, o in zip((0, 2, 3, 4, 5), range(18, 23), ([1, 0, 0], [0, 1, 0], [0, -1, 0], [0, 1, 0], [0, -1, 0])):
            tree[chain].append(j)
            offs.append(o)
        return tree, offs
    raise ValueError("synthetic skeletons exist for 18 and 23 joints")


def fwd_kin_cont6d(x6d, kinematic_tree, offsets):
    """fwd_kin_cont6d_torch(x6d, tree, offsets, root_pos=0, eps=1e-8) (dataset.py:83-116) on the
    HIP tail kernel.  x6d [..., J, 6], offsets [..., J, 3] -> pose [..., J, 3]."""
    J = x6d.shape[-2]
    rows = x6d.numel() // (6 * J)
    dev = x6d.device
    if (6 * J) % 4 == 0 and x6d.is_contiguous() and x6d.dtype == torch.float32 and x6d.data_ptr() % 16 == 0:
        ld, xin = 6 * J, x6d.reshape(rows, 6 * J)  # the kernel only needs float4-aligned rows: no padded copy
    else:
        ld = ops.pad16(6 * J)
        xin = torch.zeros(rows, ld, device=dev)
        xin[:, : 6 * J] = x6d.reshape(rows, 6 * J)
    offs = offsets.reshape(rows, J, 3).float().contiguous()
    pose = torch.empty(rows, J, 3, device=dev)
    scratch6 = torch.empty(rows, 6 * J, device=dev)
    lp = torch.empty(ops.tail_blocks(rows), 2, device=dev)
    ops.pose_tail(xin, ld, offs, pose, None, None, make_tree(J, kinematic_tree), 0.0, 0.0, None, None, scratch6, None, lp, None,
                  rows, pre_tanh=False, pose_out=pose)
    return pose.reshape(x6d.shape[:-1] + (3,))


def make_batch(n_keypts, window, batch, seed=0, device="cuda"):
    """x6d ~ N(0,1), root ~ U(-1,1), offsets = unit skeleton offsets * per-joint segment length,
    target_pose = FK(x6d, offsets), plus the conditional features of the full SC-VAE config."""
    g = torch.Generator(device="cpu").manual_seed(1000 + seed)
    tree, offs = skeleton(n_keypts)
    x6d = torch.randn(batch, window, n_keypts, 6, generator=g).to(device)
    root = (torch.rand(batch, window, 3, generator=g) * 2 - 1).to(device)
    seg = 0.5 + torch.rand(n_keypts, generator=g)
    offsets = (torch.tensor(offs, dtype=torch.float32) * seg[:, None])[None, None].expand(batch, window, n_keypts, 3).contiguous().to(device)
    data = {"x6d": x6d, "root": root, "offsets": offsets,
            "target_pose": fwd_kin_cont6d(x6d, tree, offsets),
            "avg_speed_3d": torch.randn(batch, 3, generator=g).to(device),
            "heading": torch.nn.functional.normalize(torch.randn(batch, 2, generator=g), dim=-1).to(device),
            "ids": torch.randint(0, 4, (batch, 1), generator=g).to(torch.int16).to(device)}
    return data, tree

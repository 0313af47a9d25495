"""Eval-mode generation checks on the HIP path.

`generative_restrictiveness` mirrors src/scrubvae/eval/eval.py:22-120: re-decode the latent means
with a re-drawn conditional variable and measure that variable on the generated pose.  The decode
and the forward kinematics run on the HIP kernels (ResVAE.decode, the pose-tail kernel's pose
output); the few per-window feature reductions that follow are [B,W,J,3] torch ops at the API edge.
"""
from __future__ import annotations

import math

import torch

from ..data import synthetic

# constants of the reference (eval/eval.py:43-57,106-113)
_SPD_STD = (0.4038, 0.3586, 0.4169)
_SPD_MEAN = (0.4993, 0.7112, 0.6663)
_SPD_MIN = (-1.2323, -1.9734, -1.5858)
_SPD_MAX = (4.6167, 4.6437, 4.2551)
_PARTS = ([0, 1, 2, 3, 4, 5],          # spine and head
          [1, 6, 7, 8, 9, 10, 11],     # arms from front spine
          [5, 12, 13, 14, 15, 16, 17])  # legs from back spine


def generative_restrictiveness(model, z, data, key, kinematic_tree):
    """Returns (pred, data[key]).  Like the reference, `data[key]` is REPLACED in place by the re-drawn
    variable (eval.py:29-57) and that new value is what comes back as the target (eval.py:120)."""
    n_keypts = data["x6d"].shape[-2]
    window = data["x6d"].shape[1]
    batch_size = data["x6d"].shape[0]
    dev = z.device
    var_true = data[key]
    if key == "heading":
        rand_yaw = (torch.rand(batch_size, dtype=torch.float32, device=dev) * 2 - 1)[:, None] * math.pi
        data["heading"] = torch.cat([torch.sin(rand_yaw), torch.cos(rand_yaw)], dim=-1)
    elif key == "avg_speed_3d":
        spd_std = torch.tensor(_SPD_STD, dtype=torch.float32, device=dev)
        rand_jitter = torch.randn((batch_size, 1), dtype=torch.float32, device=dev) * spd_std * 1.5 + 0.5
        data["avg_speed_3d"] = torch.clamp(var_true.to(dev) + rand_jitter,
                                           min=torch.tensor(_SPD_MIN, dtype=torch.float32, device=dev),
                                           max=torch.tensor(_SPD_MAX, dtype=torch.float32, device=dev))
    data_o = model.decode(z, data)
    # fwd_kin_cont6d_torch(..., root_pos=root_hat, do_root_R=True, eps=1e-8): every joint position is the
    # root-at-origin position plus root_pos (dataset.py:96-115), so the tail kernel's pose output + root_hat
    pose_batch = synthetic.fwd_kin_cont6d(data_o["x6d"].reshape(-1, n_keypts, 6), kinematic_tree,
                                          data["offsets"].to(dev).reshape(-1, n_keypts, 3))
    pose_batch = (pose_batch + data_o["root"].reshape(-1, 1, 3)).reshape(-1, model.window, n_keypts, 3)

    if key == "heading":
        forward = pose_batch[:, window // 2, 1, :] - pose_batch[:, window // 2, 0, :]
        forward = forward / torch.linalg.norm(forward, dim=-1)[..., None]
        yaw = -torch.arctan2(forward[:, 1], forward[:, 0])[:, None]
        pred = torch.cat([torch.sin(yaw), torch.cos(yaw)], dim=-1)
    elif key == "avg_speed_3d":
        root_spd = torch.sqrt((torch.diff(pose_batch[:, :, 0, :], n=1, dim=-2) ** 2).sum(dim=-1)).mean(dim=-1)
        dxyz = torch.zeros((len(root_spd), 3), dtype=torch.float32, device=dev)
        for i, part in enumerate(_PARTS):
            pose_part = pose_batch - pose_batch[:, window // 2, part[0], :][:, None, None, :]
            relative_dxyz = (torch.diff(pose_part[..., part[1:], :], n=1, dim=-3) ** 2).sum(dim=-1)
            dxyz[:, i] = torch.sqrt(relative_dxyz).mean(dim=(-1, -2))
        pred = torch.cat([root_spd[:, None], dxyz[:, 0:1], dxyz[:, 1:].mean(dim=-1, keepdim=True)], dim=-1)
        pred = (pred - torch.tensor(_SPD_MEAN, dtype=torch.float32, device=dev)) / torch.tensor(_SPD_STD, dtype=torch.float32, device=dev)
    else:
        raise NotImplementedError(f"generative_restrictiveness: key {key!r} (the reference handles heading and avg_speed_3d)")
    return pred, data[key]

"""Pose preprocessing on the device: raw keypoints [frames, J, 3] -> the tensors
the training path consumes.  Mirrors `preprocess_save_data` (src/scrubvae/data/dataset.py:313-446) from the
point where the pose array has been read; the h5 reader (`neuroposelib.read.pose_h5`) stays outside.

Arithmetic runs in libscrubvae_hip.so (csrc/preprocess.hip: inv_kin + re-orientation + 6-D conversion +
segment lengths + heading in one per-frame kernel, the speed features in one per-window kernel, target poses
through the pose-tail kernel); window bookkeeping is index arithmetic on the host, the window gather a torch
index_select.
"""
from __future__ import annotations

import ctypes as C

import numpy as np
import torch

from .. import _lib, ops
from .._lib import check
from . import synthetic

SPEED_PARTS = ([0, 1, 2, 3, 4, 5],          # spine and head       (dataset.py:366-370)
               [1, 6, 7, 8, 9, 10, 11],     # arms from front spine
               [5, 12, 13, 14, 15, 16, 17])  # legs from back spine


def get_window_indices(ids, stride, window):
    """dataset.py:199-231: start..start+window index rows of every stride-th window that stays inside one id run."""
    ids = np.asarray(ids)
    frame_idx = np.arange(len(ids), dtype=np.int64)
    id_diff = np.diff(ids, prepend=ids[0])
    change = np.concatenate([[0], np.where(id_diff != 0)[0], [len(ids)]])
    out = []
    for i in range(len(change) - 1):
        if change[i + 1] - change[i] >= window:
            seg = frame_idx[change[i]: change[i + 1]]
            out.append(np.lib.stride_tricks.sliding_window_view(seg, window_shape=window, axis=0)[::stride])
        else:
            print("ID {} length smaller than window size - skipping ...".format(ids[change[i]]))
    return torch.from_numpy(np.ascontiguousarray(np.concatenate(out, axis=0)))


_stream = ops._stream


def inv_kin_windows(pose, kinematic_tree, offset, direction_process="midfwd", want_offsets=True, want_root=True):
    """pose [N,W,J,3] float32 cuda -> x6d [N,W,J,6], offsets [N,W,J,3], root [N,W,3], heading [N,2]."""
    N, W, J, _ = pose.shape
    pose = pose.contiguous().float()
    off = np.array(offset)
    truncate = int(np.issubdtype(off.dtype, np.integer))  # dataset.py:289-294: an integer OFFSET array truncates the lengths
    uo = (C.c_float * (3 * J))(*[float(v) for v in off.reshape(-1)])
    x6d = torch.empty(N, W, J, 6, device=pose.device)
    offsets = torch.empty(N, W, J, 3, device=pose.device) if want_offsets else None
    root = torch.empty(N, W, 3, device=pose.device) if want_root else None
    heading = torch.empty(N, 2, device=pose.device)
    p = lambda t: None if t is None else t.data_ptr()
    check(_lib.lib().svae_inv_kin(p(pose), uo, C.byref(_lib.make_tree(J, kinematic_tree)), W, int(direction_process == "midfwd"),
                                  int(direction_process in ("midfwd", "x360")), truncate, p(x6d), p(offsets), p(root), p(heading),
                                  N * W, _stream()), "inv_kin")
    return x6d, offsets, root, heading


def get_speed_parts(pose, parts=SPEED_PARTS):
    """dataset.py:133-163 + :373-375 -> avg_speed_3d [N,3] = [root, spine+head, mean(limbs)]."""
    N, W, J, _ = pose.shape
    pose = pose.contiguous().float()
    flat = [j for part in parts for j in part]
    out = torch.empty(N, 3, device=pose.device)
    check(_lib.lib().svae_speed_parts(pose.data_ptr(), (C.c_int * len(flat))(*flat), (C.c_int * len(parts))(*[len(q) for q in parts]),
                                      len(parts), W, J, out.data_ptr(), N, _stream()), "speed_parts")
    return out


def get_speed_outliers(pose, threshold=2.25):
    """dataset.py:298-309: windows whose mean keypoint speed exceeds `threshold`."""
    spd = torch.sqrt((torch.diff(pose, n=1, dim=-3) ** 2).sum(dim=-1)).mean(dim=(-1, -2))
    out = torch.where(spd > threshold)[0]
    print("Outlier frames above {}: {}".format(threshold, len(out)))
    return out


def preprocess_pose(pose, ids, skeleton_config, window, stride=2, data_keys=("x6d", "root", "offsets"), speed_threshold=2.25,
                    direction_process="midfwd", device="cuda"):
    """`preprocess_save_data` (dataset.py:313-446) after the h5 read: pose [frames,J,3], ids [frames] -> dict of tensors
    on `device` with the keys of `data_keys` (+ "raw_pose")."""
    tree, offset = skeleton_config["KINEMATIC_TREE"], skeleton_config["OFFSET"]
    window_inds = get_window_indices(ids, stride, window)
    pose = torch.as_tensor(np.asarray(pose), dtype=torch.float32).to(device)
    pose = pose[window_inds.to(device)]
    ids = torch.as_tensor(np.asarray(ids))[window_inds][:, window // 2]
    if speed_threshold is not None:
        bad = get_speed_outliers(pose, speed_threshold)
        keep = torch.ones(len(pose), dtype=torch.bool, device=pose.device)
        keep[bad] = False
        pose, ids = pose[keep], ids[keep.cpu()]
    data = {"raw_pose": pose}
    if "avg_speed_3d" in data_keys:
        data["avg_speed_3d"] = get_speed_parts(pose)
    need_ik = any(k in data_keys for k in ("x6d", "root", "offsets", "heading", "target_pose"))
    if need_ik:
        x6d, offsets, root, heading = inv_kin_windows(pose, tree, offset, direction_process)
        if "heading" in data_keys:
            data["heading"] = heading
        if "x6d" in data_keys:
            data["x6d"] = x6d
        if "offsets" in data_keys:
            data["offsets"] = offsets
        if "root" in data_keys:
            data["root"] = root
        if "target_pose" in data_keys:  # target pose root does not move (dataset.py:429-441)
            data["target_pose"] = synthetic.fwd_kin_cont6d(x6d, tree, offsets)
    if "ids" in data_keys:
        data["ids"] = ids.to(torch.int16).to(device)
    return data

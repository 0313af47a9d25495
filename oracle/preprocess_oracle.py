"""CPU restatement of the reference's pose preprocessing -- TEST INFRASTRUCTURE ONLY.

Follows src/scrubvae/data/dataset.py:340-446 (`preprocess_save_data` from the point where the
pose array has been windowed) and the helpers it calls, with the reference's precision at every
step: numpy float64 for the pose differences / normalisations, torch float32 inside every
quaternion helper (the `*_np` wrappers of data/quaternion.py cast with `.float()`).
Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
"""
import numpy as np
import torch

SPEED_PARTS = ([0, 1, 2, 3, 4, 5], [1, 6, 7, 8, 9, 10, 11], [5, 12, 13, 14, 15, 16, 17])  # dataset.py:366-370


def _f32(a):
    return torch.from_numpy(np.ascontiguousarray(a)).float()


def qmul(q, r):
    """quaternion.py:34-52 (Hamilton product q*r, fp32)."""
    w = q[..., 0] * r[..., 0] - q[..., 1] * r[..., 1] - q[..., 2] * r[..., 2] - q[..., 3] * r[..., 3]
    x = q[..., 1] * r[..., 0] + q[..., 0] * r[..., 1] - q[..., 3] * r[..., 2] + q[..., 2] * r[..., 3]
    y = q[..., 2] * r[..., 0] + q[..., 3] * r[..., 1] + q[..., 0] * r[..., 2] - q[..., 1] * r[..., 3]
    z = q[..., 3] * r[..., 0] - q[..., 2] * r[..., 1] + q[..., 1] * r[..., 2] + q[..., 0] * r[..., 3]
    return torch.stack((w, x, y, z), dim=-1)


def qinv(q):
    """quaternion.py:17-21."""
    return q * torch.tensor([1.0, -1.0, -1.0, -1.0])


def qbetween(v0, v1):
    """quaternion.py:409-420."""
    v = torch.cross(v0, v1, dim=-1)
    w = torch.sqrt((v0 ** 2).sum(dim=-1, keepdim=True) * (v1 ** 2).sum(dim=-1, keepdim=True)) + (v0 * v1).sum(dim=-1, keepdim=True)
    q = torch.cat([w, v], dim=-1)
    return q / torch.norm(q, dim=-1, keepdim=True)


def qrot(q, v):
    """quaternion.py:55-74."""
    qvec = q[..., 1:]
    uv = torch.cross(qvec, v, dim=-1)
    uuv = torch.cross(qvec, uv, dim=-1)
    return v + 2 * (q[..., :1] * uv + uuv)


def quaternion_to_cont6d(q):
    """quaternion.py:291-334: first two COLUMNS of the rotation matrix."""
    r, i, j, k = torch.unbind(q, -1)
    two_s = 2.0 / (q * q).sum(-1)
    col0 = torch.stack((1 - two_s * (j * j + k * k), two_s * (i * j + k * r), two_s * (i * k - j * r)), -1)
    col1 = torch.stack((two_s * (i * j - k * r), 1 - two_s * (i * i + k * k), two_s * (j * k + i * r)), -1)
    return torch.cat([col0, col1], dim=-1)


def inv_kin(pose, tree, offset, forward_indices=(0, 1)):
    """dataset.py:11-46.  pose [F,J,3] float64 -> local quaternions [F,J,4] (float64 array holding fp32 values).
    Quirk kept: the root quaternion of frame 0 of the flattened array is forced to identity (:31)."""
    forward = pose[:, forward_indices[1], :] - pose[:, forward_indices[0], :]
    forward = forward / np.linalg.norm(forward, axis=-1)[..., None]
    target = np.array([[1, 0, 0]]).repeat(len(forward), axis=0)
    root_quat = qbetween(_f32(forward), _f32(target)).numpy()
    local = np.zeros(pose.shape[:-1] + (4,))
    root_quat[0] = np.array([1.0, 0.0, 0.0, 0.0])
    local[:, 0] = root_quat
    for chain in tree:
        R = root_quat
        for i in range(len(chain) - 1):
            u = offset[chain[i + 1]][None, ...].repeat(len(pose), axis=0)
            v = pose[:, chain[i + 1]] - pose[:, chain[i]]
            v = v / np.linalg.norm(v, axis=-1)[..., None]
            rot_u_v = qbetween(_f32(u), _f32(v)).numpy()
            R_loc = qmul(qinv(_f32(R)), _f32(rot_u_v)).numpy()
            local[:, chain[i + 1], :] = R_loc
            R = qmul(_f32(R), _f32(R_loc)).numpy()
    return local


def get_segment_len(pose, tree, offset):
    """dataset.py:279-296."""
    parents = [0] * len(offset)
    parents[0] = -1
    for chain in tree:
        for j in range(1, len(chain)):
            parents[chain[j]] = chain[j - 1]
    # Quirk kept (dataset.py:289-294): `offsets` inherits the dtype of np.array(OFFSET).  The shipped skeleton
    # configs list integer unit vectors, so the array is int64 and the float segment lengths are TRUNCATED
    # toward zero on assignment; a float OFFSET list keeps them.
    offsets = np.tile(offset[None], (pose.shape[0], 1, 1))
    for i in range(1, offset.shape[0]):
        offsets[:, i] = np.linalg.norm(pose[:, i, :] - pose[:, parents[i], :], axis=1)[..., None] * offsets[:, i]
    return offsets


def get_speed_parts(pose, parts=SPEED_PARTS):
    """dataset.py:133-163."""
    root_spd = np.sqrt((np.diff(pose[..., 0, :], n=1, axis=-2) ** 2).sum(-1)).mean(-1)
    dxyz = np.zeros((len(root_spd), len(parts) + 1))
    dxyz[:, 0] = root_spd
    centered = pose - pose[..., 0:1, :]
    for i, part in enumerate(parts):
        pp = centered if part[0] == 0 else centered - centered[:, part[0]: part[0] + 1, :]
        # NB the reference subtracts centered[:, part[0]] along the WINDOW axis (frame index part[0]), :147
        rel = (np.diff(pp[..., part[1:], :], n=1, axis=-3) ** 2).sum(-1)
        dxyz[:, i + 1] = np.sqrt(rel).mean(axis=(-1, -2))
    return dxyz


def get_frame_yaw(pose, root_i=0, front_i=1):
    """dataset.py:234-241."""
    forward = pose[:, front_i, :] - pose[:, root_i, :]
    forward = forward / np.linalg.norm(forward, axis=-1)[..., None]
    return -np.arctan2(forward[:, 1], forward[:, 0])


def preprocess_windows(pose, tree, offset, data_keys, direction_process="midfwd", fwd_kin=None):
    """dataset.py:359-446: pose [N,W,J,3] float64 (already windowed / filtered) -> dict of float32 tensors."""
    N, W = pose.shape[:2]
    offset = np.array(offset)  # dtype as in the reference (int for the shipped configs, see get_segment_len)
    data = {"raw_pose": pose}
    if "avg_speed_3d" in data_keys:
        speed = get_speed_parts(pose)
        data["avg_speed_3d"] = np.concatenate([speed[:, :2], speed[:, 2:].mean(axis=-1, keepdims=True)], axis=-1)
    yaw = get_frame_yaw(pose[:, W // 2, ...], 0, 1)[..., None]
    if "heading" in data_keys:
        data["heading"] = np.concatenate([np.sin(yaw), np.cos(yaw)], axis=-1)
    root = pose[..., 0, :].copy()
    if direction_process in ("midfwd", "x360"):
        centre = np.zeros(root.shape)
        centre[..., [0, 1]] = root[:, W // 2, [0, 1]][:, None, :]
        root -= centre
    if "x6d" in data_keys:
        local = inv_kin(pose.reshape((-1,) + pose.shape[-2:]), tree, offset, forward_indices=[1, 0]).reshape(pose.shape[:-1] + (-1,))
        if direction_process == "midfwd":
            fwd = np.zeros((len(yaw), 4))
            fwd[:, [-1, 0]] = np.concatenate([np.sin(yaw / 2), np.cos(yaw / 2)], axis=-1)
            fwd = np.repeat(fwd[:, None, :], W, axis=1)
            local[..., 0, :] = qmul(_f32(fwd), _f32(local[..., 0, :])).numpy()
            if "root" in data_keys:
                root = qrot(_f32(fwd), _f32(root)).numpy()
        data["x6d"] = quaternion_to_cont6d(_f32(local)).numpy()
    if "offsets" in data_keys:
        data["offsets"] = get_segment_len(pose.reshape((-1,) + pose.shape[-2:]), tree, offset).reshape(pose.shape)
    if "root" in data_keys:
        data["root"] = root
    data = {k: torch.tensor(v, dtype=torch.float32) for k, v in data.items()}
    if "target_pose" in data_keys:
        x = data["x6d"].reshape((-1,) + data["x6d"].shape[-2:])
        offs = data["offsets"].reshape(x.shape[:2] + (-1,))
        data["target_pose"] = fwd_kin(x, tree, offs, torch.zeros(x.shape[0], 3), eps=1e-8).reshape(data["x6d"].shape[:-1] + (3,))
    return data


def get_window_indices(ids, stride, window):
    """dataset.py:199-231 (without the progress printing)."""
    frame_idx = np.arange(len(ids), dtype=int)
    id_diff = np.diff(ids, prepend=ids[0])
    change = np.concatenate([[0], np.where(id_diff != 0)[0], [len(ids)]])
    out = []
    for i in range(len(change) - 1):
        if change[i + 1] - change[i] >= window:
            seg = frame_idx[change[i]: change[i + 1]]
            out.append(np.lib.stride_tricks.sliding_window_view(seg, window_shape=window, axis=0)[::stride])
    return np.concatenate(out, axis=0)


def synthetic_raw_pose(n_frames_per_id=(150, 130), seed=0):
    """A plausible 18-joint mouse trajectory (random smooth rotations through the oracle FK) for fixtures."""
    from oracle import scvae_oracle as O
    g = torch.Generator().manual_seed(seed)
    tree, offs = O.skeleton_tree(18), torch.tensor(O.skeleton_offsets(18), dtype=torch.float64)
    poses, ids = [], []
    for a, n in enumerate(n_frames_per_id):
        t = torch.linspace(0, 1, n, dtype=torch.float64)[:, None, None]
        c0, c1 = torch.randn(1, 18, 6, generator=g, dtype=torch.float64), torch.randn(1, 18, 6, generator=g, dtype=torch.float64)
        x6d = c0 * (1 - t) + c1 * t + 0.1 * torch.randn(n, 18, 6, generator=g, dtype=torch.float64)
        seg = 0.5 + torch.rand(18, generator=g, dtype=torch.float64)
        root = torch.cumsum(0.05 * torch.randn(n, 3, generator=g, dtype=torch.float64), dim=0)
        poses.append(O.fwd_kin(x6d, tree, (offs * seg[:, None])[None].expand(n, -1, -1), root, eps=1e-8).numpy())
        ids.append(np.full(n, a))
    return np.concatenate(poses, axis=0), np.concatenate(ids, axis=0)

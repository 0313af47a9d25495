"""Preprocessing row: oracle vs the real reference's `preprocess_save_data` outputs (CPU), HIP kernels
vs those outputs (GPU).  Fixtures: tests/golden/make_preprocess_fixture.py."""
import os

import numpy as np
import pytest
import torch

from oracle import preprocess_oracle as P
from oracle import scvae_oracle as O

KEYS = ["x6d", "root", "offsets", "target_pose", "avg_speed_3d", "heading", "ids"]


def load(golden_dir, name):
    fx = np.load(os.path.join(golden_dir, name + ".npz"))
    is_float = bool(fx["offset_is_float"])
    offset = [[float(c) if is_float else int(c) for c in row] for row in O.skeleton_offsets(18)]
    return fx, {"KINEMATIC_TREE": O.skeleton_tree(18), "OFFSET": offset}


@pytest.mark.parametrize("name", ["preprocess_tiny", "preprocess_tiny_float"])
def test_oracle_preprocess_matches_reference(golden_dir, name):
    fx, skel = load(golden_dir, name)
    win = P.get_window_indices(fx["raw_ids"], int(fx["stride"]), int(fx["window"]))
    out = P.preprocess_windows(fx["raw_pose"][win], skel["KINEMATIC_TREE"], skel["OFFSET"], KEYS, "midfwd", fwd_kin=O.fwd_kin)
    # bit-identical on the machine that wrote the fixture; another CPU's vectorised fp32 paths move the last bits
    for k in KEYS:
        if k != "ids":
            assert float((out[k].double() - torch.from_numpy(fx["out/" + k]).double()).abs().max()) <= 1e-5, k


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["preprocess_tiny", "preprocess_tiny_float"])
def test_hip_preprocess_matches_reference(golden_dir, name):
    """Tolerance: the reference normalises the pose differences in float64 before its float32 quaternion maths; the
    kernel is float32 throughout.  x6d entries are O(1): 2e-5 absolute; lengths / speeds 1e-5 relative."""
    from scrubvae_amd.data import preprocess as PP
    fx, skel = load(golden_dir, name)
    out = PP.preprocess_pose(fx["raw_pose"], fx["raw_ids"], skel, int(fx["window"]), int(fx["stride"]), data_keys=KEYS,
                             speed_threshold=None, direction_process="midfwd")
    tol = {"x6d": 2e-5, "root": 1e-5, "offsets": 1e-5, "target_pose": 5e-5, "avg_speed_3d": 1e-5, "heading": 1e-5}
    for k, t in tol.items():
        want = torch.from_numpy(fx["out/" + k]).double()
        got = out[k].cpu().double()
        assert got.shape == want.shape, k
        assert float((got - want).abs().max()) <= t * max(1.0, float(want.abs().max())), (k, float((got - want).abs().max()))
    assert torch.equal(out["ids"].cpu(), torch.from_numpy(fx["out/ids"]))
    assert out["ids"].dtype == torch.int16


@pytest.mark.gpu
def test_hip_preprocess_feeds_the_training_step(golden_dir):
    """x6d / offsets produced on the device reproduce the raw keypoints through forward kinematics (the round trip the
    reference relies on): FK(x6d, offsets) + root trajectory == pose up to the window's re-orientation."""
    from scrubvae_amd.data import preprocess as PP
    fx, skel = load(golden_dir, "preprocess_tiny_float")
    out = PP.preprocess_pose(fx["raw_pose"], fx["raw_ids"], skel, int(fx["window"]), int(fx["stride"]),
                             data_keys=["x6d", "offsets", "root", "target_pose"], speed_threshold=None, direction_process="x360")
    raw = out["raw_pose"].double().cpu()
    rel = raw - raw[..., 0:1, :]                      # keypoints relative to the root joint
    assert float((out["target_pose"].cpu().double() - rel).abs().max()) < 5e-5 * float(rel.abs().max() + 1)

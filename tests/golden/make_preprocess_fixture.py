"""Golden fixture for the preprocessing row: runs the REAL reference's
`preprocess_save_data` (src/scrubvae/data/dataset.py:313-446) on a synthetic raw pose array.

    python -B tests/golden/make_preprocess_fixture.py        (build container only)

`neuroposelib.read.pose_h5` (an absent third-party loader, no arithmetic) is stubbed to hand the
synthetic array over; everything downstream is the reference's own code.  Saves inputs + the
reference's outputs, and prints the oracle's deviation from them.
"""
import os
import sys

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)

import numpy as np
import torch

import make_fixtures as MF
from oracle import preprocess_oracle as P
from oracle import scvae_oracle as O

KEYS = ["x6d", "root", "offsets", "target_pose", "avg_speed_3d", "heading", "ids"]


def main():
    MF._install_stubs()
    pose, ids = P.synthetic_raw_pose()
    sys.modules["neuroposelib"].read.pose_h5 = lambda path: (pose.copy(), ids.copy())
    sys.path.insert(0, MF.REF)
    import scrubvae  # noqa: F401
    from scrubvae.data.dataset import preprocess_save_data
    window, stride = 64, 8
    # "int": OFFSET as the shipped configs list it (integers -> the reference truncates the segment lengths);
    # "float": the same unit vectors as floats (lengths kept)
    for kind in ("int", "float"):
        skel = {"KINEMATIC_TREE": O.skeleton_tree(18),
                "OFFSET": [[float(c) if kind == "float" else int(c) for c in row] for row in O.skeleton_offsets(18)]}
        run(preprocess_save_data, skel, pose, ids, window, stride, "preprocess_tiny" + ("_float" if kind == "float" else ""))


def run(preprocess_save_data, skel, pose, ids, window, stride, name):
    ref = preprocess_save_data("unused/", skel, "synthetic", window, stride, data_keys=KEYS, speed_threshold=None,
                               direction_process="midfwd")
    fx = {"raw_pose": pose, "raw_ids": ids, "window": np.int64(window), "stride": np.int64(stride)}
    for k, v in ref.items():
        fx["out/" + k] = v.numpy()
    win = P.get_window_indices(ids, stride, window)
    mine = P.preprocess_windows(pose[win], skel["KINEMATIC_TREE"], skel["OFFSET"], KEYS, "midfwd", fwd_kin=O.fwd_kin)
    for k in KEYS:
        if k == "ids":
            continue
        a, b = mine[k].double(), ref[k].double()
        print(f"  {k:14s} shape {tuple(b.shape)}  oracle-vs-reference max abs dev {float((a - b).abs().max()):.2e}")
    fx["offset_is_float"] = np.int64(isinstance(skel["OFFSET"][1][0], float))
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **fx)
    print(f"saved {path} {os.path.getsize(path)/1e6:.2f} MB, {len(win)} windows")


if __name__ == "__main__":
    main()

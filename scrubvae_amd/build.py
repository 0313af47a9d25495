"""Build libscrubvae_hip.so in-tree with hipcc for gfx950 (cross-compiles without a GPU)."""
import os
import subprocess
import sys

CSRC = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc")
SOURCES = ["capi.hip", "gemm_f32.hip", "gemm_bf16s.hip", "elementwise.hip", "pose_tail.hip", "latent.hip", "preprocess.hip"]
OUT = os.path.join(CSRC, "libscrubvae_hip.so")


def needs_build():
    if not os.path.exists(OUT):
        return True
    t = os.path.getmtime(OUT)
    deps = [os.path.join(CSRC, s) for s in SOURCES + ["svae_internal.h", "gemm_common.h"]]
    deps.append(os.path.join(CSRC, "..", "..", "include", "scrubvae_hip.h"))
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False, ablation=False):
    """ablation=True additionally compiles the timing-experiment kernel variants used by tools/bench_split_dbg.py
    (parts of the work removed, wrong results) -- never part of the shipped library."""
    if not force and not needs_build():
        return OUT
    cmd = ["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-shared", "-fPIC", "-Wno-unused-result"]
    if ablation:
        cmd.append("-DSVAE_ABLATION_KERNELS")
    cmd += [os.path.join(CSRC, s) for s in SOURCES] + ["-o", OUT]
    if verbose:
        print(" ".join(cmd))
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        sys.stderr.write(r.stdout + r.stderr)
        raise RuntimeError("hipcc failed building libscrubvae_hip.so")
    return OUT


if __name__ == "__main__":
    print(build(force="--force" in sys.argv or "--ablation" in sys.argv, verbose=True, ablation="--ablation" in sys.argv))

"""Build libscrubvae_hip.so in-tree with hipcc for gfx950 (cross-compiles without a GPU)."""
import os
import subprocess
import sys

CSRC = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc")
SOURCES = ["capi.hip", "gemm_f32.hip", "gemm_bf16s.hip", "halo_ws_bf16s.hip", "wgrad_bf16s.hip", "elementwise.hip", "pose_tail.hip", "latent.hip", "preprocess.hip", "ensemble.hip"]
OUT = os.path.join(CSRC, "libscrubvae_hip.so")


def needs_build():
    if not os.path.exists(OUT):
        return True
    t = os.path.getmtime(OUT)
    deps = [os.path.join(CSRC, s) for s in SOURCES + ["svae_internal.h", "gemm_common.h", "split_common.h", "split_gather.h"]]
    deps.append(os.path.join(CSRC, "..", "..", "include", "scrubvae_hip.h"))
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False, ablation=False):
    """Each translation unit is compiled to its own object (in parallel, only when it or a header changed) and the objects
    are linked into the shared library.  ablation=True additionally compiles the timing-experiment kernel variants used by
    tools/bench_split_dbg.py (parts of the work removed, wrong results) -- never part of the shipped library."""
    if not force and not needs_build():
        return OUT
    from concurrent.futures import ThreadPoolExecutor
    objdir = os.path.join(CSRC, "build")
    os.makedirs(objdir, exist_ok=True)
    headers = [os.path.join(CSRC, "svae_internal.h"), os.path.join(CSRC, "gemm_common.h"), os.path.join(CSRC, "split_common.h"), os.path.join(CSRC, "split_gather.h"),
               os.path.join(CSRC, "..", "..", "include", "scrubvae_hip.h")]
    hdr_t = max(os.path.getmtime(h) for h in headers)
    flags = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-result"] + (["-DSVAE_ABLATION_KERNELS"] if ablation else [])
    tag = "_abl" if ablation else ""

    def compile_one(src):
        obj = os.path.join(objdir, os.path.splitext(src)[0] + tag + ".o")
        path = os.path.join(CSRC, src)
        if not force and os.path.exists(obj) and os.path.getmtime(obj) > max(os.path.getmtime(path), hdr_t):
            return obj, None
        cmd = ["hipcc"] + flags + ["-c", path, "-o", obj]
        if verbose:
            print(" ".join(cmd))
        r = subprocess.run(cmd, capture_output=True, text=True)
        return obj, (r.stdout + r.stderr if r.returncode != 0 else None)

    with ThreadPoolExecutor(max_workers=min(len(SOURCES), os.cpu_count() or 4)) as ex:
        results = list(ex.map(compile_one, SOURCES))
    errs = [e for _, e in results if e]
    if errs:
        sys.stderr.write("\n".join(errs))
        raise RuntimeError("hipcc failed building libscrubvae_hip.so")
    cmd = ["hipcc", "--offload-arch=gfx950", "-shared", "-fPIC"] + [o for o, _ in results] + ["-o", OUT]
    if verbose:
        print(" ".join(cmd))
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        sys.stderr.write(r.stdout + r.stderr)
        raise RuntimeError("hipcc failed linking libscrubvae_hip.so")
    return OUT


if __name__ == "__main__":
    print(build(force="--force" in sys.argv or "--ablation" in sys.argv, verbose=True, ablation="--ablation" in sys.argv))

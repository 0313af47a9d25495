"""ctypes binding of libscrubvae_hip.so (include/scrubvae_hip.h).

The product path has NO fallback: if the HIP library is missing this module raises at
import of the first op, and every op raises RuntimeError on a non-zero status.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("SVAE_LIB_PATH") or os.path.join(_HERE, "csrc", "libscrubvae_hip.so")  # env: kernel experiments only

MAX_TAPS, MAX_JOINTS, MAX_CHAINS, MAX_CHAIN_LEN = 128, 32, 8, 8


class ConvDesc(C.Structure):
    _fields_ = [(n, C.c_int) for n in (
        "batch", "l_in", "l_out", "c_in", "c_out", "ld_in", "ld_out",
        "kernel", "stride", "padding", "dilation", "transposed")] + [("tile", C.c_int * 3), ("up2", C.c_int)]


class ColsumTask(C.Structure):
    _fields_ = [("x", C.c_void_p), ("out", C.c_void_p), ("rows", C.c_longlong), ("C", C.c_int), ("ld", C.c_int)]


MAX_COLSUM_TASKS = 48


class ColsumPartTask(C.Structure):
    _fields_ = [("part", C.c_void_p), ("out", C.c_void_p), ("rows", C.c_int), ("C", C.c_int)]


class SplitTask(C.Structure):
    _fields_ = [("w", C.c_void_p), ("wsplit", C.c_void_p), ("kernel", C.c_int), ("c_in", C.c_int), ("c_out", C.c_int)]


MAX_SPLIT_TASKS = 48


ENS_MEMBERS, ENS_MAX_LAYERS = 4, 3


class EnsLayer(C.Structure):
    _fields_ = [("w", C.c_void_p), ("b", C.c_void_p), ("dw", C.c_void_p), ("db", C.c_void_p), ("K", C.c_int), ("N", C.c_int)]


class EnsMember(C.Structure):
    _fields_ = [("layer", EnsLayer * ENS_MAX_LAYERS), ("n_layers", C.c_int), ("out", C.c_void_p), ("d_out", C.c_void_p)]


class EnsDesc(C.Structure):
    _fields_ = [("member", EnsMember * ENS_MEMBERS), ("n_members", C.c_int),
                ("src0", C.c_void_p), ("ld0", C.c_int), ("n0", C.c_int),
                ("src1", C.c_void_p), ("ld1", C.c_int), ("n1", C.c_int),
                ("perm", C.c_void_p), ("shuf_col", C.c_int), ("batch", C.c_int), ("halves", C.c_int),
                ("shuf_vals", C.c_void_p)]


class BnBwdFuse(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("x", "scale", "shift", "mean", "rstd", "alpha", "part", "dalpha_part")]


class Tree(C.Structure):
    _fields_ = [("n_joints", C.c_int), ("n_chains", C.c_int),
                ("chain_len", C.c_int * MAX_CHAINS),
                ("chain", (C.c_int * MAX_CHAIN_LEN) * MAX_CHAINS)]


def make_tree(n_joints, kinematic_tree):
    t = Tree()
    t.n_joints = n_joints
    if len(kinematic_tree) > MAX_CHAINS:
        raise ValueError(f"kinematic tree has {len(kinematic_tree)} chains > {MAX_CHAINS}")
    t.n_chains = len(kinematic_tree)
    for c, chain in enumerate(kinematic_tree):
        if len(chain) > MAX_CHAIN_LEN:
            raise ValueError(f"chain {c} longer than {MAX_CHAIN_LEN}")
        t.chain_len[c] = len(chain)
        for i, j in enumerate(chain):
            t.chain[c][i] = int(j)
    return t


P, I, LL, F, D, SZ = C.c_void_p, C.c_int, C.c_longlong, C.c_float, C.c_double, C.c_size_t
DP = C.POINTER(ConvDesc)

# name -> (restype, argtypes); every symbol include/scrubvae_hip.h declares
SIGNATURES = {
    "svae_version": (I, []),
    "svae_last_error": (None, [C.c_char_p, SZ]),
    "svae_conv_fwd": (I, [DP, P, P, P, P, I, P]),
    "svae_conv_dgrad": (I, [DP, P, P, P, I, P]),
    "svae_conv_wgrad_workspace": (SZ, [DP]),
    "svae_conv_splitk_workspace": (SZ, [DP, I]),
    "svae_conv_fwd_ws": (I, [DP, P, P, P, P, I, P, SZ, P]),
    "svae_conv_dgrad_ws": (I, [DP, P, P, P, I, P, SZ, P]),
    "svae_conv_wgrad": (I, [DP, P, P, P, P, P, SZ, I, P]),
    "svae_conv_tile": (I, [DP, I, C.POINTER(I), C.POINTER(I)]),
    "svae_conv_split_bytes": (SZ, [DP]),
    "svae_conv_split_weights": (I, [DP, P, P, P]),
    "svae_conv_split_weights_batched": (I, [C.POINTER(SplitTask), I, P]),
    "svae_conv_fwd_split": (I, [DP, P, P, P, P, I, I, P]),
    "svae_conv_fwd_stats_tiles": (I, [DP]),
    "svae_conv_fwd_split_stats": (I, [DP, P, P, P, P, I, I, P, P]),
    "svae_conv_fwd_split_up2": (I, [DP, P, P, P, P, I, I, P, P, P]),
    "svae_conv_dgrad_split": (I, [DP, P, P, P, I, I, P]),
    "svae_conv_dgrad_stats_tiles": (I, [DP, C.POINTER(I)]),
    "svae_conv_dgrad_split_bn": (I, [DP, P, P, P, I, I, C.POINTER(BnBwdFuse), P]),
    "svae_conv_wgrad_split": (I, [DP, P, P, P, P, P, SZ, I, I, P]),
    "svae_conv_split_tile": (I, [DP, I, C.POINTER(I), C.POINTER(I), C.POINTER(I), C.POINTER(I)]),
    "svae_pack_input": (I, [P, P, C.POINTER(F), P, LL, I, I, P]),
    "svae_bn_chunks": (I, [LL]),
    "svae_bn_stats_partial": (I, [P, LL, I, I, P, P]),
    "svae_bn_reduce_partials": (I, [P, I, I, P, P]),
    "svae_bn_finalize": (I, [P, D, I, P, P, F, F, P, P, P, P, P, P, P]),
    "svae_bn_stats_finalize": (I, [P, I, D, I, P, P, F, F, P, P, P, P, P, P, P, P]),
    "svae_bn_bwd_reduce": (I, [P, I, I, P, P, P, P, P, I, I, P]),
    "svae_colsum_batched_workspace": (SZ, [C.POINTER(ColsumTask), I]),
    "svae_colsum_batched": (I, [C.POINTER(ColsumTask), I, P, SZ, I, P]),
    "svae_bn_eval_coeffs": (I, [I, P, P, F, P, P, P, P, P]),
    "svae_affine_prelu_fwd": (I, [P, P, P, P, P, LL, I, I, P]),
    "svae_affine_prelu_bwd_partial": (I, [P, P, P, P, P, P, P, LL, I, I, P, P, P]),
    "svae_affine_prelu_bwd_apply": (I, [P, P, P, P, P, P, P, P, P, D, P, LL, I, I, P, P, P, P, I, I, P]),
    "svae_affine_prelu_bwd_apply_colsum": (I, [P, P, P, P, P, P, P, P, P, D, P, LL, I, I, P, P, P, P, I, I, P, P]),
    "svae_affine_prelu_colsum_rows": (I, [LL, I]),
    "svae_colsum_from_partials": (I, [C.POINTER(ColsumPartTask), I, I, P]),
    "svae_upsample2_fwd": (I, [P, P, I, I, I, I, P]),
    "svae_upsample2_bwd": (I, [P, P, I, I, I, I, I, P]),
    "svae_heads_diag_fwd": (I, [P, I, P, P, P, P, I, P, I, I, I, I, P]),
    "svae_heads_blocks": (I, [I, I]),
    "svae_heads_diag_bwd": (I, [P, I, P, P, P, I, P, P, F, P, I, I, I, I, P]),
    "svae_heads_tril_fwd": (I, [P, I, P, P, I, P, P, I, P, I, I, I, P]),
    "svae_heads_tril_bwd": (I, [P, I, P, P, P, I, P, I, F, P, P, I, I, I, P]),
    "svae_tc_logvar": (I, [P, I, P, P, I, I, P]),
    "svae_tc_fwd": (I, [P, I, P, I, P, I, I, P, P, P, P]),
    "svae_tc_bwd": (I, [P, I, P, I, P, I, I, P, P, F, P, I, P, I, P, I, P]),
    "svae_tail_blocks": (I, [LL]),
    "svae_pose_tail": (I, [P, I, P, P, P, C.POINTER(F), C.POINTER(Tree), F, F, P, P, P, P, P, P, P, LL, I, P]),
    "svae_rot_loss": (I, [P, P, F, P, P, LL, P]),
    "svae_inv_kin": (I, [P, C.POINTER(F), C.POINTER(Tree), I, I, I, I, P, P, P, P, LL, P]),
    "svae_speed_parts": (I, [P, C.POINTER(I), C.POINTER(I), I, I, I, P, LL, P]),
    "svae_heads_beta_fwd": (I, [P, I, P, P, P, I, P, I, I, I, P]),
    "svae_heads_beta_bwd": (I, [P, I, P, P, P, I, P, I, P, F, P, I, I, I, P]),
    "svae_rot_blocks": (I, [LL]),
    "svae_adam_step": (I, [P, P, P, P, LL, F, F, F, F, F, I, I, F, P]),
    "svae_adam_step_dev": (I, [P, P, P, P, LL, P, F, F, F, F, I, F, P]),
    "svae_adam_advance": (I, [P, F, F, P]),
    "svae_clip_grads": (I, [P, LL, P, F, P, P]),
    "svae_small_solve": (I, [P, LL, P, P, LL, P, LL, I, I, I, P]),
    "svae_gauss_ll": (I, [P, I, P, P, P, P, I, I, I, P]),
    "svae_kde_mi": (I, [P, I, P, I, P, P, P, I, P, F, F, P, P, I, I, I, I, P]),
    "svae_sumsq_blocks": (I, [LL]),
    "svae_sumsq_partial": (I, [P, LL, P, P]),
    "svae_reduce_rows": (I, [P, I, I, F, P, I, P]),
    "svae_reduce_rows_scaled": (I, [P, I, I, C.POINTER(F), P, P]),
    "svae_loss_total": (I, [P, C.POINTER(F), I, P, P]),
    "svae_relu_fwd": (I, [P, P, LL, P]),
    "svae_relu_bwd": (I, [P, P, P, LL, P]),
    "svae_axpy": (I, [F, P, P, LL, P]),
    "svae_fill": (I, [P, F, LL, P]),
    "svae_mse_sum": (I, [P, I, P, I, I, I, F, P, P, P]),
    "svae_rowloss_blocks": (I, [I]),
    "svae_ce_sum": (I, [P, I, P, I, I, F, P, P, P]),
    "svae_double_softmax_ce_sum": (I, [P, I, I, F, P, P, P]),
    "svae_ens_fwd": (I, [C.POINTER(EnsDesc), P]),
    "svae_ens_bwd_workspace": (SZ, [C.POINTER(EnsDesc)]),
    "svae_ens_bwd": (I, [C.POINTER(EnsDesc), P, I, F, P, P, SZ, I, P]),
    "svae_ens_loss": (I, [I, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.POINTER(F), C.POINTER(F), I, P, I, P, I, I, I, P, P]),
}

_lib = None


def lib():
    """Load the HIP library (once).  Raises if it was not built: there is no CPU path."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(hipcc --offload-arch=gfx950).  scrubvae_amd has no CPU fallback.")
        # PyTorch-ROCm ships its own libamdhip64: load it FIRST so that this library binds to
        # the same HIP runtime instance (two runtimes in one process cannot share pointers)
        import torch  # noqa: F401
        l = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(l, name)  # AttributeError if the library does not export it
            fn.restype = res
            fn.argtypes = args
        _lib = l
    return _lib


def last_error():
    buf = C.create_string_buffer(512)
    lib().svae_last_error(buf, 512)
    return buf.value.decode()


MAX_LOSS_TERMS = 48  # include/scrubvae_hip.h SVAE_MAX_LOSS_TERMS
ERR_SHAPE, ERR_ALIGN, ERR_WORKSPACE, ERR_LAUNCH, ERR_ARG = -1, -2, -3, -4, -5  # include/scrubvae_hip.h svae_status


class SvaeError(RuntimeError):
    """A non-zero svae_status; `.status` holds the code (ERR_* above)."""

    def __init__(self, status, what, msg):
        super().__init__(f"libscrubvae_hip {what} failed (status {status}): {msg}")
        self.status = status


def check(status, what=""):
    if status != 0:
        raise SvaeError(status, what, last_error())

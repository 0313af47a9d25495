"""Thin tensor-level wrappers over the C ABI (pointers + sizes only cross the boundary).

PyTorch is plumbing here: it owns device memory and the HIP stream; all arithmetic on the
hot path happens inside libscrubvae_hip.so.
"""
from __future__ import annotations

import ctypes as C
import json
import os
import threading

import torch

from . import _lib
from ._lib import ConvDesc, check


def pad16(c):
    return (int(c) + 15) // 16 * 16


def _p(t):
    """device address as a plain int (None = NULL): every entry point has its argtypes declared (_lib.SIGNATURES), so ctypes
    converts in C -- constructing a c_void_p object per argument cost ~0.4 ms per optimizer step"""
    return None if t is None else t.data_ptr()


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)


def _stream():
    """hipStream_t of torch's current stream.  torch.cuda.current_stream() costs ~8 us per call (about 1 ms per
    optimizer step over ~120 calls); the raw-handle query is ~20x cheaper."""
    o = _TLS.stream_override
    if o is not None:
        return o
    if _raw_stream is not None:
        return _raw_stream(_device_index())
    return torch.cuda.current_stream().cuda_stream


# _TLS.stream_override: hipStream_t every launch of this module goes to instead of torch's current stream (None: follow torch).  Set by
# ResVAE._fork around a side-stream body that consists of C-ABI launches only: entering / leaving a `torch.cuda.stream`
# context costs ~15 us of Python per fork (~0.6 ms per optimizer step over ~40 forks).
class _ThreadState(threading.local):
    stream_override = None


_TLS = _ThreadState()  # per thread: two models driven from different threads do not see each other's side streams


_get_device = getattr(torch._C, "_cuda_getDevice", None)


def _device_index():
    """Index of torch's current device (the raw query is ~0.2 us: not cached, so a process that drives several devices
    always launches on the stream of the device it has made current; ResVAE checks at the start of every pass that this
    is the device its buffers live on)."""
    return _get_device() if _get_device is not None else torch.cuda.current_device()


def check_current_device(device):
    """Raise if `device` (where a model's buffers live) is not torch's current device: the C ABI launches on the current
    device's stream, so a mismatch would run kernels on the wrong GPU's stream."""
    if device.type == "cuda" and device.index is not None and device.index != _device_index():
        raise RuntimeError(f"model lives on cuda:{device.index} but the current device is cuda:{_device_index()}: wrap the call in "
                           f"torch.cuda.device({device.index})")


def _f32c(t, name="tensor"):
    if t is None:
        return
    if not (t.is_cuda and t.dtype == torch.float32 and t.is_contiguous()):
        raise ValueError(f"{name}: expected a contiguous float32 CUDA tensor, got {t.dtype} {t.device} contiguous={t.is_contiguous()}")


class LaunchTimer:
    """Optional per-launch HIP-event timing of the GEMM kernels (bench.py's roofline leg).
    Events are recorded on torch's current stream, which is the stream the kernels are
    launched on.  kinds: subset of {"fwd", "dgrad", "wgrad"}; only launches whose kernel
    template matches `bn` (64/128, None = any) are bracketed."""

    def __init__(self, kinds=("fwd",), only=None):
        self.kinds = set(kinds)
        self.only = only  # restrict to one kernel template name (keeps the event overhead small)
        self.records = []  # (kernel template name, flops, start_event, end_event)

    def summary(self):
        """kernel name -> launches, algorithmic flops, total ms."""
        torch.cuda.synchronize()
        out = {}
        for kind, flops, s, e in self.records:
            d = out.setdefault(kind, dict(launches=0, flops=0.0, ms=0.0))
            d["launches"] += 1
            d["flops"] += flops
            d["ms"] += s.elapsed_time(e)
        return out


TIMER = None  # set to a LaunchTimer to enable

# First-use autotuning of the GEMM launches (fwd/dgrad/wgrad): each (geometry, kind) times every
# candidate kernel variant / workgroup tile once on scratch outputs and keeps the fastest.  Within a
# kernel family the result does not depend on the tile (same per-output summation order).
AUTOTUNE = True
AUTOTUNE_MIN_FLOPS = 2e8
AUTOTUNE_REPS = 4
# Tile choices measured once on an MI355X (tools/tune_tiles.py) for the benchmark geometries;
# geometries not in the table are tuned on first use.
_TABLE_PATH = os.environ.get("SVAE_TILE_TABLE") or os.path.join(os.path.dirname(os.path.abspath(__file__)), "tuned_tiles.json")  # env: tuning experiments
try:
    with open(_TABLE_PATH) as _f:
        TILE_TABLE = json.load(_f)
except (OSError, ValueError) as _e:  # still correct (geometries are tuned on first use) but not the benchmarked kernel mix: say so
    import warnings
    warnings.warn(f"scrubvae_amd: tile table {_TABLE_PATH} unreadable ({_e}); every geometry will be tuned on first use")
    TILE_TABLE = {}
TUNED_LOG = {}  # geometry key -> code chosen in this process (dumped by tools/tune_tiles.py)
_TILES = (128128, 128064, 64128, 64064)
_GATHER_CODES = _TILES + tuple(1000000 + c for c in _TILES)  # 1BBBNNN: LDS-DMA staging variant
_WGRAD_CODES = _TILES + (1,)  # 1 = tap-fused small-weight kernel


_KIND_ID = {"fwd": 0, "dgrad": 1, "wgrad": 2}

# Arithmetic of the large contractions.  "f32": v_mfma_f32_32x32x2_f32.  "bf16x6" / "bf16x3" /
# "bf16": every fp32 operand is split into 3 / 2 / 1 bf16 pieces and the 6 / 3 / 1 leading cross
# products run on the bf16 matrix cores with fp32 accumulation (csrc/gemm_bf16s.hip); bf16x6
# is as accurate as f32.  Read when a Conv is constructed; convs below SPLIT_MIN_FLOPS stay f32.
# The library default is "f32" (or $SVAE_PRECISION); bench.py selects "bf16x6b3".
PRECISION = os.environ.get("SVAE_PRECISION", "f32")
F16X2 = 22  # svae pieces code: two fp16 pieces per operand, three products (forward only; include/scrubvae_hip.h SVAE_PIECES_F16X2)
_PIECES = {"f32": 0, "bf16x6": 3, "bf16x6w3": 3, "bf16x6b3": 3, "bf16x3": 2, "bf16": 1, "f16x3b3": F16X2}
# "bf16x6w3": forward / data-gradient contractions with 3 pieces (6 products), WEIGHT-gradient contractions with 2
# pieces (3 products).  A weight gradient sums >= thousands of rows; its own 2^-16-per-product rounding is invisible
# next to the error the forward / data-gradient arithmetic already leaves on the same tensor
# (tests/studies/precision_bf16_split.py: worst gradient error vs fp64 identical to bf16x6, 3-7x below fp32's own).
# "f16x3b3": forward with TWO FP16 pieces / 3 products (22 of 24 significand bits per operand: ~2^-22 per product, fp32-class;
# half the matrix-core work of the 6-product bf16 split), backward as bf16x6b3 (2 bf16 pieces: gradients need bf16's exponent range).
_WGRAD_PIECES = {"bf16x6w3": 2, "bf16x6b3": 2, "f16x3b3": 2}
# "bf16x6b3": the whole BACKWARD pass (data- and weight-gradient contractions) with 2 pieces / 3 products, the forward
# (outputs, every loss term, the ELBO) with 3 pieces / 6 products.  Same study, row `x6+b:x3`: worst gradient error vs
# fp64 1.2e-3 / 4.8e-4 / 3.4e-4 / 2.2e-2 on the four fixtures against 8.3e-4 / 3.1e-3 / 3.4e-4 / 2.1e-2 for six products
# everywhere and 5.6e-3 / 4.4e-3 / 9.2e-4 / 6.7e-2 for the reference's own fp32 arithmetic.  The data-gradient kernels read
# the two leading planes of the 3-piece weight split (second piece truncated instead of rounded: a 2^-17 relative bias).
_DGRAD_PIECES = {"bf16x6b3": 2, "f16x3b3": 2}
SPLIT_MIN_FLOPS = float(os.environ.get("SVAE_SPLIT_MIN_FLOPS", 2e9))
# split gather kernels, code VBBBNNN: V = 0: 4 waves, double-buffered LDS; 1: 4 waves, one LDS buffer; 2 / 3: 8 waves
# (BM = 128), one / two buffers; 4 / 5: wave-specialised (4 producer + 8 / 4 consumer waves), 2 tiles in flight;
# 6 / 7: the same with 3 tiles in flight; 8: halo-image kernel (BM = 128)
_SPLIT_GATHER_CODES = (_TILES + tuple(1000000 + c for c in _TILES) + (2128128, 2128064, 3128128, 3128064)
                       + (4128128, 4128064, 4064128) + tuple(5000000 + c for c in _TILES)
                       + (6128128, 6128064, 6064128) + tuple(7000000 + c for c in _TILES) + (8128128, 8128064)
                       + (9128128, 9128064)  # 9: halo kernel on 256-row tiles (the row field of the code stays 128)
                       + (10128128, 10128064, 11128128, 11128064)  # 10 / 11: wave-specialised halo kernel, 128- / 256-row tiles
                       + (12128128, 12128064, 13128128, 13128064)  # 12 / 13: the same with three weight-tile buffers
                       + (16128128, 16128064, 17128128, 17128064, 18128128, 18128064)
                       + (19128128, 29128128))  # 19: halo kernel on 256 x 160 tiles, 8 x 1 waves (the fields of the code are placeholders): few-column outputs (conv_out: 144)  # 18: as 16 on the 16x16x32 MFMA; 16 / 17: 8 consumer + 4 DMA-only loader waves, 256-row tiles, consumers staggered / not
# (codes 14128128 / 15128128 -- four consumer waves with 128 x 64 wave tiles, compiler-scheduled / pinned pipeline -- exist and are
#  parity-tested but measured 0-15 % slower than 11 / 13 on every benchmark layer: not tuning candidates)
_SPLIT_VARIANT = {0: "2, 2, 2, 1", 1: "2, 2, 1, 2", 2: "4, 2, 1, 2", 3: "4, 2, 2, 1"}
# 1BBBNNN: single LDS buffer; 256-edge tiles: 8 waves, half the operand bytes per FLOP through the vector-memory path
_WGRAD_BIG = (256256, 256128, 128256)
_SPLIT_WGRAD_CODES = _TILES + tuple(1000000 + c for c in _TILES) + _WGRAD_BIG + tuple(1000000 + c for c in _WGRAD_BIG)
# 2BBBNNN / 3BBBNNN: as 0 / 1 with the XCD-aware workgroup order (the workgroups of one reduction-row split share an XCD's L2)
_SPLIT_WGRAD_CODES = _SPLIT_WGRAD_CODES + tuple(2000000 + c for c in _SPLIT_WGRAD_CODES)
# 4BBBNNN / 6BBBNNN: all taps of a tile in one workgroup (wgrad_taps_bf16s_kernel), launch order / XCD-aware order
_SPLIT_WGRAD_CODES = _SPLIT_WGRAD_CODES + (4128128, 4064128, 4128064, 6128128, 6064128, 6128064)
# 16BBBNNN (+1 single buffer, +2 XCD order): the taps folded into the dY columns (transposed convs) / the X channel rows (convs): one tile padding for all taps
_SPLIT_WGRAD_CODES = _SPLIT_WGRAD_CODES + (16064128, 16128128, 16128064, 17064128, 18064128, 18128128, 18128064)
# 12BBBNNN / 14BBBNNN: the all-taps kernel on the 16x16x32 MFMA shape
_SPLIT_WGRAD_CODES = _SPLIT_WGRAD_CODES + (12128128, 12064128, 12128064, 14128128, 14064128, 14128064)
# BatchNorm batch statistics (forward) and first-pass backward sums from the conv GEMMs' epilogues where their kernels support it
# (SVAE_FUSE_BN=0: the separate passes, for A/B measurements)
FUSE_BN_STATS = os.environ.get("SVAE_FUSE_BN", "1") != "0"
MIX_F32 = True  # a bf16x6 conv may keep the fp32 MFMA kernel for a pass where that is faster (same accuracy)
WEIGHT_EPOCH = 0  # bumped whenever master weights may have changed (start of every model pass)


def up2_supported(kernel, stride=1, dilation=1, transposed=False):
    """Geometries whose forward and weight-gradient kernels can take the x2 linear upsample of their input on the fly."""
    return kernel == 6 and stride == 1 and dilation == 1 and not transposed


def set_precision(name):
    global PRECISION
    if name not in _PIECES:
        raise ValueError(f"precision {name!r} not in {sorted(_PIECES)}")
    PRECISION = name


def split_weights_batched(users):
    """users: iterable of (Conv, weight).  Splits the weights of every conv that runs a split-bf16 forward or
    data-gradient in one launch per MAX_SPLIT_TASKS and marks their copies current for this weight epoch."""
    todo = [(cv, w) for cv, w in users
            if cv.pieces and (cv._kind_pieces("fwd") or cv._kind_pieces("dgrad"))
            and (cv._split_epoch != WEIGHT_EPOCH or cv._split_src != w.data_ptr())]
    for i in range(0, len(todo), _lib.MAX_SPLIT_TASKS):
        chunk = todo[i: i + _lib.MAX_SPLIT_TASKS]
        arr = (_lib.SplitTask * len(chunk))()
        for t, (cv, w) in zip(arr, chunk):
            if cv._wsplit is None:
                n = int(_lib.lib().svae_conv_split_bytes(C.byref(cv.desc)))
                cv._wsplit = torch.empty(n + 64, dtype=torch.uint8, device=w.device)
            t.w, t.wsplit, t.kernel, t.c_in, t.c_out = w.data_ptr(), cv._wsplit.data_ptr(), cv.kernel, cv.c_in_p, cv.c_out_p
        check(_lib.lib().svae_conv_split_weights_batched(arr, len(chunk), _stream()), "conv_split_weights_batched")
        for cv, w in chunk:
            cv._split_epoch, cv._split_src = WEIGHT_EPOCH, w.data_ptr()


def bump_weight_epoch():
    """Invalidate the split-bf16 weight copies: the next use of each Conv re-splits its weights."""
    global WEIGHT_EPOCH
    WEIGHT_EPOCH += 1


def _timed(kind, conv, n_padded, fn):
    t = TIMER
    if t is None or kind not in t.kinds or (t.only is not None and conv.kernel_name(kind) != t.only):
        return fn()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    # The stream is drained first: a HIP event's timestamp is taken when the command processor reaches its marker, which for a marker
    # queued behind a long kernel can be BEFORE that kernel has finished -- brackets of back-to-back GEMMs then overlap (measured: the
    # brackets of one serialised step summed to 22 ms of a 15.3 ms step; a data-gradient launch read 950 us where rocprofv3 reports 510).
    torch.cuda.current_stream().synchronize()
    s.record()
    r = fn()
    e.record()
    t.records.append((conv.kernel_name(kind), conv.flops, s, e))
    return r


class Conv:
    """Geometry of one nn.Conv1d / nn.ConvTranspose1d / nn.Linear call (padded channels)."""

    def __init__(self, batch, l_in, c_in, c_out, kernel, stride=1, padding=0, dilation=1, transposed=False,
                 ld_in=None, ld_out=None, pieces=None, up2=False):
        """up2: the conv's input is the x2 linear upsample of a [batch * l_in / 2, c_in] tensor (l_in: the upsampled length): `fwd`
        takes that half-length tensor, its kernel blends the rows while staging them (svae_conv_desc.up2) and can leave the upsampled
        tensor behind as a by-product (`up_out`) for `wgrad`, which reads it like any conv (or, with `up2_wgrad`, blends the half-length
        tensor too); `dgrad` returns the gradient with respect to the upsampled input.  Split-precision convs only (`up2_supported`)."""
        self.c_in, self.c_out = c_in, c_out
        self.c_in_p, self.c_out_p = pad16(c_in), pad16(c_out)
        if transposed:
            l_out = (l_in - 1) * stride - 2 * padding + dilation * (kernel - 1) + 1
        else:
            l_out = (l_in + 2 * padding - dilation * (kernel - 1) - 1) // stride + 1
        self.l_in, self.l_out, self.batch = l_in, l_out, batch
        self.kernel = kernel
        self.desc = ConvDesc(batch, l_in, l_out, self.c_in_p, self.c_out_p,
                             ld_in or self.c_in_p, ld_out or self.c_out_p,
                             kernel, stride, padding, dilation, 1 if transposed else 0)
        self.up2 = bool(up2)
        # up2_wgrad: the weight gradient blends the half-length input too (all-taps kernels; measured 30 % slower than the plain
        # kernels on the upsampled tensor -- twice the operand requests -- so the default reads the forward's `up_out` by-product)
        self.up2_wgrad = False
        self._ws_bytes = None
        # algorithmic FLOPs of one pass (forward = dgrad = wgrad): 2*B*L*k*Cin*Cout with the
        # unpadded channel counts, L = output length (conv) / input length (transposed conv)
        self.flops = 2.0 * batch * (l_in if transposed else l_out) * kernel * c_in * c_out
        # bf16 pieces per operand (0 = fp32 MFMA kernels)
        self.pieces = (_PIECES[PRECISION] if self.flops >= SPLIT_MIN_FLOPS else 0) if pieces is None else int(pieces)
        # pieces of the weight-gradient contraction (differs from self.pieces only in the "bf16x6w3" precision)
        self.wgrad_pieces = (_WGRAD_PIECES.get(PRECISION, self.pieces) if self.pieces else 0) if pieces is None else int(pieces)
        self.dgrad_pieces = (_DGRAD_PIECES.get(PRECISION, self.pieces) if self.pieces else 0) if pieces is None else int(pieces)
        self._wsplit, self._split_epoch, self._split_src = None, -1, None
        if up2 and not (self.pieces and up2_supported(kernel, stride, dilation, transposed)):
            raise ValueError("Conv(up2=True): the fused upsample exists for split-precision 6-tap stride-1 nn.Conv1d geometries")

    def _dref(self, kind):
        """The descriptor as pass `kind` sees it: svae_conv_desc.up2 is set for the forward of an up2 conv (and for its weight
        gradient with up2_wgrad); the other passes run the plain geometry on full-length tensors."""
        self.desc.up2 = 1 if (self.up2 and (kind == "fwd" or (kind == "wgrad" and self.up2_wgrad))) else 0
        return C.byref(self.desc)

    def split_weights(self, w):
        """bf16 piece planes of `w` for the split kernels; re-split once per weight epoch."""
        if self._split_epoch != WEIGHT_EPOCH or self._split_src != w.data_ptr():
            if self._wsplit is None:
                n = int(_lib.lib().svae_conv_split_bytes(C.byref(self.desc)))
                self._wsplit = torch.empty(n + 64, dtype=torch.uint8, device=w.device)
            check(_lib.lib().svae_conv_split_weights(C.byref(self.desc), _p(w), _p(self._wsplit), _stream()), "conv_split_weights")
            self._split_epoch, self._split_src = WEIGHT_EPOCH, w.data_ptr()
        return self._wsplit

    _F32_FLAG = 100000000  # table / log encoding: "this kind of this split-precision conv runs the fp32 kernel <code % flag>"

    def _base_pieces(self, kind):
        return self.wgrad_pieces if kind == "wgrad" else (self.dgrad_pieces if kind == "dgrad" else self.pieces)

    def _kind_pieces(self, kind):
        return self.__dict__.get("_kp", {}).get(kind, self._base_pieces(kind))

    def _set_choice(self, kind, pieces, code):
        self.__dict__.setdefault("_kp", {})[kind] = pieces
        self.desc.tile[_KIND_ID[kind]] = code
        self._ws_bytes = None
        self.__dict__.pop("_names", None)
        self.__dict__.pop("_stats_tiles", None)
        self.__dict__.pop("_dstats_tiles", None)

    def tile_key(self, kind):
        """Key of this geometry's pass `kind` in the tile table (tuned_tiles.json / TUNED_LOG)."""
        d = self.desc
        base = self._base_pieces(kind)
        key = f"{kind}{'@' + str(base) if base else ''}:{d.batch}:{d.l_in}:{d.c_in}:{d.c_out}:{d.ld_in}:{d.ld_out}:{d.kernel}:{d.stride}:{d.padding}:{d.transposed}"
        if self.up2 and (kind == "fwd" or (kind == "wgrad" and self.up2_wgrad)):  # the other passes share the plain geometry's entries
            key += ":up2"
        return key

    def _tune(self, kind, run):
        """run(): launches this conv once with scratch outputs.  Picks the kernel family (for a
        bf16x6 conv: split-bf16 or fp32 MFMA -- both are fp32-accurate) and desc.tile[kind]."""
        tuned = self.__dict__.setdefault("_tuned", set())
        if kind in tuned:
            return
        tuned.add(kind)
        if _TLS.stream_override is not None:  # the timing events below live on torch's current stream: tune there
            keep, _TLS.stream_override = _TLS.stream_override, None
            try:
                tuned.discard(kind)
                return self._tune(kind, run)
            finally:
                _TLS.stream_override = keep
        base = self._base_pieces(kind)
        key = self.tile_key(kind)
        if key in TILE_TABLE:
            v = int(TILE_TABLE[key])
            self._set_choice(kind, 0 if v >= self._F32_FLAG else base, v % self._F32_FLAG)
            return
        if not AUTOTUNE or self.flops < AUTOTUNE_MIN_FLOPS or torch.cuda.is_current_stream_capturing():
            return
        cands = []
        if base:
            cands += [(base, c) for c in (_SPLIT_WGRAD_CODES if kind == "wgrad" else _SPLIT_GATHER_CODES)]
        if not base or (MIX_F32 and (base in (3, F16X2) or (kind in ("wgrad", "dgrad") and self.pieces in (3, F16X2)))):
            cands += [(0, c) for c in (_WGRAD_CODES if kind == "wgrad" else _GATHER_CODES)]
        best, best_t = (base, 0), float("inf")
        for pieces, code in cands:
            self._set_choice(kind, pieces, code)
            try:
                run()  # warm-up (also validates the workspace size for this tile)
            except _lib.SvaeError as e:
                if e.status in (_lib.ERR_SHAPE, _lib.ERR_WORKSPACE):  # this tile does not exist for this geometry: not a candidate
                    continue
                raise  # a launch failure or a bad argument is a bug, not a tuning outcome
            t = float("inf")
            for _ in range(AUTOTUNE_REPS):  # min of several single-launch timings: robust against one-off stalls
                s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                s.record()
                run()
                e.record()
                e.synchronize()
                t = min(t, s.elapsed_time(e))
            if t < best_t * 0.98:  # candidates are ordered large -> small: ties keep the larger tile
                best, best_t = (pieces, code), t
        self._set_choice(kind, *best)
        TUNED_LOG[key] = best[1] + (self._F32_FLAG if (base and not best[0]) else 0)

    def kernel_name(self, kind):
        """Name of the kernel template instance this call dispatches to (as rocprofv3 prints it)."""
        names = self.__dict__.setdefault("_names", {})
        if kind not in names:
            bm, bn = C.c_int(), C.c_int()
            check(_lib.lib().svae_conv_tile(self._dref(kind), _KIND_ID[kind], C.byref(bm), C.byref(bn)), "conv_tile")
            kp = self._kind_pieces(kind)
            P, H = (2, "true") if kp == F16X2 else (kp, "false")
            wcode = self.desc.tile[2] or (4064128 if (self.up2 and self.up2_wgrad) else 0)  # the library's default for up2 convs: the all-taps kernel
            if kp and kind == "wgrad" and (wcode // 1000000) & 4:
                names[kind] = (f"wgrad_taps{'16' if (wcode // 1000000) & 8 else ''}_bf16s_kernel<{bm.value}, {bn.value}, {self.kernel}, "
                               f"{self.desc.stride}, {'true' if self.desc.transposed else 'false'}, {'true' if (self.up2 and self.up2_wgrad) else 'false'}>")
            elif kp and kind == "wgrad":
                waves = {256256: "2, 4", 256128: "4, 2", 128256: "2, 4"}.get(self.desc.tile[2] % 1000000, "2, 2")
                names[kind] = f"wgrad_gemm_bf16s_kernel<{bm.value}, {bn.value}, {kp}, {1 if (self.desc.tile[2] // 1000000) & 1 else 2}, {waves}>"
            elif kp:
                v, rm = C.c_int(), C.c_int()
                check(_lib.lib().svae_conv_split_tile(self._dref(kind), _KIND_ID[kind], C.byref(bm), C.byref(bn), C.byref(v),
                                                      C.byref(rm)), "conv_split_tile")
                v = v.value
                if v in (8, 9):
                    up = "true" if (self.up2 and kind == "fwd") else "false"
                    names[kind] = f"gather_halo_bf16s_kernel<{bm.value}, {bn.value}, {P}, 4, 2, {320 if (up == 'true' and v == 9) else rm.value}, {H}, {up}>"
                elif v == 19:
                    names[kind] = f"gather_halo_bf16s_kernel<{bm.value}, {bn.value}, {P}, 8, 1, {rm.value}, {H}, false>"
                elif v == 29:
                    names[kind] = f"gather_halo_bf16s_kernel<256, 256, {P}, 4, 2, 320, {H}, {'true' if (self.up2 and kind == 'fwd') else 'false'}>"
                elif v in (10, 11, 12, 13):
                    names[kind] = f"gather_halo_ws_bf16s_kernel<{bm.value}, {bn.value}, {P}, 4, 2, {rm.value}, 0, false, {3 if v >= 12 else 2}, {H}>"
                elif v in (16, 17):
                    names[kind] = (f"gather_halo_ws4_bf16s_kernel<{bm.value}, {bn.value}, {rm.value}, {88 if rm.value == 264 else 64}, {H}, "
                                   f"{'true' if v == 16 else 'false'}, 0>")
                elif v == 18:
                    names[kind] = f"gather_halo_ws4m_bf16s_kernel<{bm.value}, {bn.value}, {rm.value}, {88 if rm.value == 264 else 64}, {H}>"
                elif v in (14, 15):
                    names[kind] = f"gather_halo_ws_bf16s_kernel<{bm.value}, {bn.value}, {P}, 2, 2, {rm.value}, 0, {'true' if v == 15 else 'false'}, 3, {H}>"
                elif v >= 4:
                    cw = "2, 2" if v in (5, 7) else ("4, 2" if bm.value == 128 else "2, 4")
                    names[kind] = f"gather_gemm_bf16s_ws_kernel<{bm.value}, {bn.value}, {P}, {cw}, {2 if v < 6 else 3}, 0, {H}>"
                else:
                    names[kind] = f"gather_gemm_bf16s_kernel<{bm.value}, {bn.value}, {P}, {_SPLIT_VARIANT[v]}, {H}>"
            elif kind == "wgrad":
                names[kind] = (f"wgrad_fused_kernel<{-bm.value}, {'false' if self.desc.transposed else 'true'}>" if bm.value < 0
                               else f"wgrad_gemm_kernel<{bm.value}, {bn.value}>")
            else:
                kn = "gather_gemm_dma_kernel" if bm.value > 1000 else "gather_gemm_kernel"
                names[kind] = f"{kn}<{bm.value % 1000}, {bn.value}, {'true' if kind == 'dgrad' else 'false'}>"
        return names[kind]

    @property
    def weight_shape(self):
        return (self.kernel, self.c_in_p, self.c_out_p)

    def wgrad_workspace_bytes(self):
        """Workspace for the current tile; before tuning: the maximum over all candidate tiles."""
        if "wgrad" not in self.__dict__.get("_tuned", ()) and AUTOTUNE and self.flops >= AUTOTUNE_MIN_FLOPS:
            keep, need = self.desc.tile[2], 0
            for code in (_SPLIT_WGRAD_CODES if self._base_pieces("wgrad") else _WGRAD_CODES):
                self.desc.tile[2] = code
                need = max(need, int(_lib.lib().svae_conv_wgrad_workspace(self._dref("wgrad"))))
            self.desc.tile[2] = keep
            return need
        if self._ws_bytes is None:
            self._ws_bytes = int(_lib.lib().svae_conv_wgrad_workspace(self._dref("wgrad")))
        return self._ws_bytes

    def _splitk_ws(self, kind_id, device):
        """Slab workspace of the fp32 split-K path (skinny problems: the Linear heads), or None when the library
        would not split this geometry.  Owned by the Conv (launches of one Conv are stream-ordered)."""
        cache = self.__dict__.setdefault("_skws", {})
        if kind_id not in cache:
            n = int(_lib.lib().svae_conv_splitk_workspace(C.byref(self.desc), kind_id))
            cache[kind_id] = torch.empty(n // 4 + 16, device=device) if n > 0 else None
        return cache[kind_id]

    def stats_tiles(self):
        """Row tiles of the forward launch when its kernel can emit the BatchNorm statistics of its output from the epilogue
        (the split-bf16 gather kernels), else 0 (fp32 kernels: the caller runs bn_stats_partial).  Valid once the tile
        choice is fixed (tune_fwd)."""
        if not FUSE_BN_STATS or not self._kind_pieces("fwd") or "fwd" not in self.__dict__.get("_tuned", ()):
            return 0
        n = self.__dict__.get("_stats_tiles")
        if n is None:
            n = self.__dict__["_stats_tiles"] = int(_lib.lib().svae_conv_fwd_stats_tiles(self._dref("fwd")))
        return n

    def _launch_fwd(self, x, w, bias, y, acc, stats=None, up_out=None):
        kp = self._kind_pieces("fwd")
        if kp and self.up2:
            return check(_lib.lib().svae_conv_fwd_split_up2(self._dref("fwd"), _p(x), _p(self.split_weights(w)), _p(bias), _p(y),
                                                             acc, kp, _p(stats), _p(up_out), _stream()), "conv_fwd_split_up2")
        if up_out is not None:
            raise RuntimeError("conv_fwd: up_out is the by-product of an up2 conv's split-precision forward")
        if kp:
            return check(_lib.lib().svae_conv_fwd_split_stats(self._dref("fwd"), _p(x), _p(self.split_weights(w)), _p(bias), _p(y),
                                                               acc, kp, _p(stats), _stream()), "conv_fwd_split")
        if stats is not None:
            raise RuntimeError("conv_fwd: fused BatchNorm statistics need a split-bf16 forward kernel (check stats_tiles())")
        ws = self._splitk_ws(0, x.device)
        if ws is not None:
            return check(_lib.lib().svae_conv_fwd_ws(self._dref("fwd"), _p(x), _p(w), _p(bias), _p(y), acc, _p(ws), ws.numel() * 4,
                                                     _stream()), "conv_fwd_ws")
        return check(_lib.lib().svae_conv_fwd(self._dref("fwd"), _p(x), _p(w), _p(bias), _p(y), acc, _stream()), "conv_fwd")

    def dgrad_stats_tiles(self):
        """(row tiles, column blocks) of the data-gradient launch when its kernel can emit the sums of a BatchNorm + activation
        backward from its epilogue (split-bf16 gather kernels), else (0, 0).  Valid once the tile choice is fixed (tune_dgrad)."""
        if not FUSE_BN_STATS or not self._kind_pieces("dgrad") or "dgrad" not in self.__dict__.get("_tuned", ()):
            return 0, 0
        r = self.__dict__.get("_dstats_tiles")
        if r is None:
            cb = C.c_int()
            n = int(_lib.lib().svae_conv_dgrad_stats_tiles(C.byref(self.desc), C.byref(cb)))
            r = self.__dict__["_dstats_tiles"] = (n, cb.value)
        return r

    def _launch_dgrad(self, dy, w, dx, acc, fuse=None):
        kp = self._kind_pieces("dgrad")
        if kp:
            return check(_lib.lib().svae_conv_dgrad_split_bn(C.byref(self.desc), _p(dy), _p(self.split_weights(w)), _p(dx),
                                                              acc, kp, None if fuse is None else C.byref(fuse), _stream()), "conv_dgrad_split")
        if fuse is not None:
            raise RuntimeError("conv_dgrad: the fused BatchNorm backward sums need a split-bf16 kernel (check dgrad_stats_tiles())")
        ws = self._splitk_ws(1, dy.device)
        if ws is not None:
            return check(_lib.lib().svae_conv_dgrad_ws(C.byref(self.desc), _p(dy), _p(w), _p(dx), acc, _p(ws), ws.numel() * 4,
                                                       _stream()), "conv_dgrad_ws")
        return check(_lib.lib().svae_conv_dgrad(C.byref(self.desc), _p(dy), _p(w), _p(dx), acc, _stream()), "conv_dgrad")

    def _launch_wgrad(self, x, dy, dw, db, ws, acc):
        nbytes = ws.numel() * ws.element_size()
        kp = self._kind_pieces("wgrad")
        if kp:
            return check(_lib.lib().svae_conv_wgrad_split(self._dref("wgrad"), _p(x), _p(dy), _p(dw), _p(db), _p(ws), nbytes,
                                                           acc, kp, _stream()), "conv_wgrad_split")
        return check(_lib.lib().svae_conv_wgrad(self._dref("wgrad"), _p(x), _p(dy), _p(dw), _p(db), _p(ws), nbytes, acc,
                                                 _stream()), "conv_wgrad")

    def tune_fwd(self, x, w, bias):
        """Fix the forward kernel / tile of this geometry (table lookup or first-use timing on scratch outputs)."""
        if "fwd" not in self.__dict__.get("_tuned", ()):
            scratch = torch.empty(self.batch * self.l_out * self.desc.ld_out + 16, device=x.device)
            self._tune("fwd", lambda: self._launch_fwd(x, w, bias, scratch, 0))

    def fwd(self, x, w, bias, y, accumulate=False, stats=None, up_out=None):
        """stats: [stats_tiles()][2][c_out_p] buffer for the fused BatchNorm statistics of y (None: off);
        up_out (up2 convs): [batch * l_in, c_in_p] buffer that receives the upsampled input rows as a by-product (None: off)"""
        self.tune_fwd(x, w, bias)
        _timed("fwd", self, self.c_out_p, lambda: self._launch_fwd(x, w, bias, y, int(accumulate), stats, up_out))
        return y

    def tune_dgrad(self, dy, w):
        if "dgrad" not in self.__dict__.get("_tuned", ()):
            scratch = torch.empty(self.batch * self.l_in * self.desc.ld_in + 16, device=dy.device)
            self._tune("dgrad", lambda: self._launch_dgrad(dy, w, scratch, 0))

    def dgrad(self, dy, w, dx, accumulate=False, fuse=None):
        """fuse: a _lib.BnBwdFuse -- dx is the gradient wrt the output of a BatchNorm + activation stage; its backward's first
        pass comes out of this launch's epilogue (see svae_conv_dgrad_split_bn)"""
        self.tune_dgrad(dy, w)
        _timed("dgrad", self, self.c_in_p, lambda: self._launch_dgrad(dy, w, dx, int(accumulate), fuse))
        return dx

    def wgrad(self, x, dy, dw, db, ws, accumulate=False):
        if "wgrad" not in self.__dict__.get("_tuned", ()):
            sdw = torch.empty(self.kernel * self.c_in_p * self.c_out_p + 16, device=x.device)
            sdb = torch.empty(self.c_out_p, device=x.device)
            self.desc.tile[2] = 0
            need = 0
            for code in (_SPLIT_WGRAD_CODES if self._base_pieces("wgrad") else _WGRAD_CODES):  # scratch workspace large enough for every candidate
                self.desc.tile[2] = code
                need = max(need, int(_lib.lib().svae_conv_wgrad_workspace(self._dref("wgrad"))))
            self.desc.tile[2] = 0
            sws = torch.empty(need // 4 + 16, device=x.device)
            self._tune("wgrad", lambda: self._launch_wgrad(x, dy, sdw, sdb, sws, 0))
            if ws.numel() * ws.element_size() < self.wgrad_workspace_bytes():
                raise RuntimeError("conv_wgrad: workspace smaller than the tuned tile needs; size it with "
                                   "Conv.wgrad_workspace_bytes() AFTER the first call or use ops.max_wgrad_workspace()")
        _timed("wgrad", self, self.c_out_p, lambda: self._launch_wgrad(x, dy, dw, db, ws, int(accumulate)))


# ----------------------------------------------------------------- weight layout (TIO)
def conv_weight_to_tio(w, transposed=False):
    """PyTorch Conv1d [Cout,Cin,k] / ConvTranspose1d [Cin,Cout,k] -> padded [k][Cin_p][Cout_p]."""
    if transposed:
        cin, cout, k = w.shape
        t = w.permute(2, 0, 1)
    else:
        cout, cin, k = w.shape
        t = w.permute(2, 1, 0)
    out = torch.zeros(k, pad16(cin), pad16(cout), dtype=w.dtype, device=w.device)
    out[:, :cin, :cout] = t
    return out


def conv_weight_from_tio(t, cin, cout, transposed=False):
    t = t[:, :cin, :cout]
    return (t.permute(1, 2, 0) if transposed else t.permute(2, 1, 0)).contiguous()


def linear_weight_to_tio(w, in_perm=None, out_perm=None, in_pad=None, out_pad=None):
    """Linear [out,in] -> [1][in_p][out_p]; in_perm/out_perm: index tensors mapping the
    library's (channels-last) feature order to PyTorch's flatten order."""
    out_f, in_f = w.shape
    wt = w.t()
    if in_perm is not None:
        wt = wt[in_perm]
    if out_perm is not None:
        wt = wt[:, out_perm]
    o = torch.zeros(1, in_pad or pad16(in_f), out_pad or pad16(out_f), dtype=w.dtype, device=w.device)
    o[0, :in_f, :out_f] = wt
    return o


def linear_weight_from_tio(t, in_f, out_f, in_perm=None, out_perm=None):
    wt = t[0, :in_f, :out_f]
    if in_perm is not None:
        inv = torch.empty_like(in_perm)
        inv[in_perm] = torch.arange(in_f, device=in_perm.device)
        wt = wt[inv]
    if out_perm is not None:
        inv = torch.empty_like(out_perm)
        inv[out_perm] = torch.arange(out_f, device=out_perm.device)
        wt = wt[:, inv]
    return wt.t().contiguous()


# ------------------------------------------------------------------------- elementwise
def _arena_ptr(arena_host):
    if arena_host is None:
        return None
    return (C.c_float * 6)(*[float(v) for v in arena_host])


def pack_input(x6d, root, arena_host, out, n_joints):
    rows = x6d.numel() // (6 * n_joints)
    check(_lib.lib().svae_pack_input(_p(x6d), _p(root), _arena_ptr(arena_host), _p(out), rows, n_joints, out.shape[-1], _stream()), "pack_input")
    return out


def bn_chunks(rows):
    return int(_lib.lib().svae_bn_chunks(rows))


def bn_stats_partial(x, rows, Cp, ld, part):
    check(_lib.lib().svae_bn_stats_partial(_p(x), rows, Cp, ld, _p(part), _stream()), "bn_stats_partial")


def bn_reduce_partials(part, n_chunks, Cp, sums):
    check(_lib.lib().svae_bn_reduce_partials(_p(part), n_chunks, Cp, _p(sums), _stream()), "bn_reduce_partials")


def bn_finalize(sums, count, Cp, gamma, beta, eps, momentum, rmean, rvar, mean, rstd, scale, shift):
    check(_lib.lib().svae_bn_finalize(_p(sums), float(count), Cp, _p(gamma), _p(beta), eps, momentum, _p(rmean), _p(rvar),
                                      _p(mean), _p(rstd), _p(scale), _p(shift), _stream()), "bn_finalize")


def bn_stats_finalize(part, n_chunks, count, Cp, gamma, beta, eps, momentum, rmean, rvar, nbt, mean, rstd, scale, shift):
    check(_lib.lib().svae_bn_stats_finalize(_p(part), n_chunks, float(count), Cp, _p(gamma), _p(beta), eps, momentum, _p(rmean),
                                            _p(rvar), _p(nbt), _p(mean), _p(rstd), _p(scale), _p(shift), _stream()), "bn_stats_finalize")


def bn_bwd_reduce(part, n_chunks, Cp, sums, dgamma, dbeta, dalpha, dalpha_part, n_parts, accumulate):
    check(_lib.lib().svae_bn_bwd_reduce(_p(part), n_chunks, Cp, _p(sums), _p(dgamma), _p(dbeta), _p(dalpha), _p(dalpha_part),
                                        n_parts, int(accumulate), _stream()), "bn_bwd_reduce")


class ColsumBatch:
    """Collects (dY, bias.grad) pairs during the backward schedule and reduces them all in two
    launches (svae_colsum_batched)."""

    def __init__(self):
        self.tasks = []
        self.part_tasks = []

    def add(self, x, rows, Cp, ld, out):
        self.tasks.append((x, rows, Cp, ld, out))

    def add_partials(self, part, rows, Cp, out):
        """out = column sums of part [rows, Cp] (partials another kernel left behind: ops.affine_prelu_bwd_apply)"""
        self.part_tasks.append((part, rows, Cp, out))

    def flush(self, get_ws, accumulate=False):
        for i in range(0, len(self.part_tasks), _lib.MAX_COLSUM_TASKS):
            chunk = self.part_tasks[i: i + _lib.MAX_COLSUM_TASKS]
            arr = (_lib.ColsumPartTask * len(chunk))()
            for t, (part, rows, Cp, out) in zip(arr, chunk):
                t.part, t.out, t.rows, t.C = part.data_ptr(), out.data_ptr(), rows, Cp
            check(_lib.lib().svae_colsum_from_partials(arr, len(chunk), int(accumulate), _stream()), "colsum_from_partials")
        self.part_tasks = []
        for i in range(0, len(self.tasks), _lib.MAX_COLSUM_TASKS):
            chunk = self.tasks[i: i + _lib.MAX_COLSUM_TASKS]
            arr = (_lib.ColsumTask * len(chunk))()
            for t, (x, rows, Cp, ld, out) in zip(arr, chunk):
                t.x, t.out, t.rows, t.C, t.ld = x.data_ptr(), out.data_ptr(), rows, Cp, ld
            need = int(_lib.lib().svae_colsum_batched_workspace(arr, len(chunk)))
            ws = get_ws(need)
            check(_lib.lib().svae_colsum_batched(arr, len(chunk), _p(ws), ws.numel() * ws.element_size(), int(accumulate),
                                                 _stream()), "colsum_batched")
        self.tasks = []


def bn_eval_coeffs(Cp, gamma, beta, eps, rmean, rvar, scale, shift):
    check(_lib.lib().svae_bn_eval_coeffs(Cp, _p(gamma), _p(beta), eps, _p(rmean), _p(rvar), _p(scale), _p(shift), _stream()), "bn_eval_coeffs")


def affine_prelu_fwd(x, scale, shift, alpha, y, rows, Cp, ld):
    check(_lib.lib().svae_affine_prelu_fwd(_p(x), _p(scale), _p(shift), _p(alpha), _p(y), rows, Cp, ld, _stream()), "affine_prelu_fwd")


def affine_prelu_bwd_partial(dy, x, scale, shift, mean, rstd, alpha, rows, Cp, ld, part, dalpha_part):
    check(_lib.lib().svae_affine_prelu_bwd_partial(_p(dy), _p(x), _p(scale), _p(shift), _p(mean), _p(rstd), _p(alpha), rows, Cp, ld,
                                                   _p(part), _p(dalpha_part), _stream()), "affine_prelu_bwd_partial")


def affine_prelu_bwd_apply(dy, x, scale, shift, mean, rstd, gamma, alpha, sums, count, dx, rows, Cp, ld,
                           dgamma, dbeta, dalpha, dalpha_part, n_parts, accumulate, colsum_part=None):
    """colsum_part [affine_prelu_colsum_rows(rows, Cp), Cp]: the pass also leaves the column sums of dx (the bias gradient of the
    conv(s) in front of the stage) as per-workgroup partials for ColsumBatch.add_partials."""
    check(_lib.lib().svae_affine_prelu_bwd_apply_colsum(_p(dy), _p(x), _p(scale), _p(shift), _p(mean), _p(rstd), _p(gamma), _p(alpha),
                                                        _p(sums), float(count), _p(dx), rows, Cp, ld, _p(dgamma), _p(dbeta), _p(dalpha),
                                                        _p(dalpha_part), n_parts, int(accumulate), _p(colsum_part), _stream()),
          "affine_prelu_bwd_apply")


def affine_prelu_colsum_rows(rows, Cp):
    """Rows of the colsum_part array of affine_prelu_bwd_apply; 0 where the pass cannot produce the sums (channel count)."""
    c4 = Cp // 4
    if Cp % 4 or c4 & (c4 - 1) or c4 > 256:
        return 0
    return int(_lib.lib().svae_affine_prelu_colsum_rows(rows, Cp))


def upsample2_fwd(x, y, batch, l_in, Cp, ld):
    check(_lib.lib().svae_upsample2_fwd(_p(x), _p(y), batch, l_in, Cp, ld, _stream()), "upsample2_fwd")


def upsample2_bwd(dy, dx, batch, l_in, Cp, ld, accumulate=False):
    check(_lib.lib().svae_upsample2_bwd(_p(dy), _p(dx), batch, l_in, Cp, ld, int(accumulate), _stream()), "upsample2_bwd")


def heads_blocks(batch, z):
    return int(_lib.lib().svae_heads_blocks(batch, z))


def heads_diag_fwd(h, ld, eps, mu, sigma, z, ldz, kl_part, batch, zdim, raw_off=None, ldm=None):
    check(_lib.lib().svae_heads_diag_fwd(_p(h), ld, _p(eps), _p(mu), _p(sigma), _p(z), ldz, _p(kl_part), batch, zdim,
                                         zdim if raw_off is None else raw_off, ldm or zdim, _stream()), "heads_diag_fwd")


def heads_diag_bwd(h, ld, eps, sigma, dz, lddz, dmu, dsigma, kl_scale, dh, batch, zdim, raw_off=None, ldm=None):
    check(_lib.lib().svae_heads_diag_bwd(_p(h), ld, _p(eps), _p(sigma), _p(dz), lddz, _p(dmu), _p(dsigma), float(kl_scale), _p(dh),
                                         batch, zdim, zdim if raw_off is None else raw_off, ldm or zdim, _stream()), "heads_diag_bwd")


def heads_tril_fwd(h, ld, eps, mu, ldm, L, z, ldz, kl_part, batch, zdim, raw_off):
    check(_lib.lib().svae_heads_tril_fwd(_p(h), ld, _p(eps), _p(mu), ldm, _p(L), _p(z), ldz, _p(kl_part), batch, zdim, raw_off,
                                         _stream()), "heads_tril_fwd")


def heads_tril_bwd(h, ld, eps, L, dz, lddz, dmu, ldm, kl_scale, dlv, dh, batch, zdim, raw_off):
    check(_lib.lib().svae_heads_tril_bwd(_p(h), ld, _p(eps), _p(L), _p(dz), lddz, _p(dmu), ldm, float(kl_scale), _p(dlv), _p(dh),
                                         batch, zdim, raw_off, _stream()), "heads_tril_bwd")


def heads_beta_fwd(h, ld, alpha, beta, mu, ldm, kl_part, batch, zdim, raw_off):
    check(_lib.lib().svae_heads_beta_fwd(_p(h), ld, _p(alpha), _p(beta), _p(mu), ldm, _p(kl_part), batch, zdim, raw_off, _stream()),
          "heads_beta_fwd")


def heads_beta_bwd(h, ld, x, alpha, beta, ldm, dz, lddz, dmu, kl_scale, dh, batch, zdim, raw_off):
    check(_lib.lib().svae_heads_beta_bwd(_p(h), ld, _p(x), _p(alpha), _p(beta), ldm, _p(dz), lddz, _p(dmu), float(kl_scale), _p(dh),
                                         batch, zdim, raw_off, _stream()), "heads_beta_bwd")


def tc_logvar(sigma, lds, L, lv, batch, zdim):
    check(_lib.lib().svae_tc_logvar(_p(sigma), lds, _p(L), _p(lv), batch, zdim, _stream()), "tc_logvar")


def tc_fwd(z, ldz, mu, ldm, lv, batch, zdim, lse_l, lse_a, loss):
    check(_lib.lib().svae_tc_fwd(_p(z), ldz, _p(mu), ldm, _p(lv), batch, zdim, _p(lse_l), _p(lse_a), _p(loss), _stream()), "tc_fwd")


def tc_bwd(z, ldz, mu, ldm, lv, batch, zdim, lse_l, lse_a, weight, d_mu, ldd, d_lv, ldv, sigma=None, lds=0):
    check(_lib.lib().svae_tc_bwd(_p(z), ldz, _p(mu), ldm, _p(lv), batch, zdim, _p(lse_l), _p(lse_a), float(weight), _p(d_mu), ldd,
                                 _p(d_lv), ldv, _p(sigma), lds, _stream()), "tc_bwd")


def tail_blocks(rows):
    return int(_lib.lib().svae_tail_blocks(rows))


def pose_tail(y, ld, offsets, target, root, arena_host, tree, jpe_scale, root_scale, ext_dx6d, ext_droot,
              x6d_hat, root_hat, loss_part, dy, rows, pre_tanh=True, pose_out=None):
    check(_lib.lib().svae_pose_tail(_p(y), ld, _p(offsets), _p(target), _p(root), _arena_ptr(arena_host), C.byref(tree),
                                    float(jpe_scale), float(root_scale), _p(ext_dx6d), _p(ext_droot), _p(x6d_hat), _p(root_hat),
                                    _p(loss_part), _p(dy), _p(pose_out), rows, int(pre_tanh), _stream()), "pose_tail")


def rot_blocks(n):
    return int(_lib.lib().svae_rot_blocks(n))


def rot_loss(x6d, x6d_hat, scale, part, dx6d_hat, n):
    check(_lib.lib().svae_rot_loss(_p(x6d), _p(x6d_hat), float(scale), _p(part), _p(dx6d_hat), n, _stream()), "rot_loss")


def adam_step(p, g, m, v, lr, beta1, beta2, eps, weight_decay, step_t, decoupled, grad_scale=1.0):
    check(_lib.lib().svae_adam_step(_p(p), _p(g), _p(m), _p(v), p.numel(), lr, beta1, beta2, eps, weight_decay, step_t,
                                    int(decoupled), grad_scale, _stream()), "adam_step")


def adam_step_dev(p, g, m, v, hyper, beta1, beta2, eps, weight_decay, decoupled, grad_scale=1.0):
    check(_lib.lib().svae_adam_step_dev(_p(p), _p(g), _p(m), _p(v), p.numel(), _p(hyper), beta1, beta2, eps, weight_decay,
                                        int(decoupled), grad_scale, _stream()), "adam_step_dev")


def adam_advance(hyper, beta1, beta2):
    check(_lib.lib().svae_adam_advance(_p(hyper), beta1, beta2, _stream()), "adam_advance")


def clip_grads(g, sumsq, max_norm, norm_out=None):
    check(_lib.lib().svae_clip_grads(_p(g), g.numel(), _p(sumsq), float(max_norm), _p(norm_out), _stream()), "clip_grads")


def small_solve(A, B, diag=None):
    """X with (A[s] + diag(diag)) X[s] = B[s]: A [S, n, n] (or [n, n]), B [S, n, nrhs] (or [n, nrhs]) fp32 on the device,
    n, nrhs <= 64 -- one launch of the batched LU kernel (csrc/latent.hip) instead of a solver-library call per system."""
    single = A.dim() == 2
    A3 = (A[None] if single else A).contiguous().float()
    B3 = (B[None] if single else B).contiguous().float()
    S, n, nrhs = A3.shape[0], A3.shape[1], B3.shape[2]
    X = torch.empty_like(B3)
    d = None if diag is None else diag.contiguous().float()
    check(_lib.lib().svae_small_solve(_p(A3), n * n, _p(d), _p(B3), n * nrhs, _p(X), n * nrhs, n, nrhs, S, _stream()), "small_solve")
    return X[0] if single else X


class _SmallInverse(torch.autograd.Function):
    """A^-1 of one small matrix on the batched LU kernel (right-hand side = identity), differentiable:
    d A = -A^-T (d A^-1) A^-T."""

    @staticmethod
    def forward(ctx, A):
        inv = small_solve(A.detach(), torch.eye(A.shape[0], device=A.device, dtype=torch.float32))
        ctx.save_for_backward(inv)
        return inv

    @staticmethod
    def backward(ctx, dinv):
        (inv,) = ctx.saved_tensors
        return -(inv.T @ dinv @ inv.T)


def small_inverse_autograd(A):
    return _SmallInverse.apply(A)


def gauss_ll(x, mean, S, want_grad=True):
    """ll [P, B] = -0.5 (logdet S_p + r^T S_p^-1 r), r = x - mean_p, and (want_grad) d ll / d x [P, B, n], for P (mean, covariance)
    pairs in one launch (csrc/latent.hip gauss_ll_kernel; reference QuadraticDiscriminantFilter.cgll, disentangle.py:129-134)."""
    x = x.float()
    if x.stride(-1) != 1:
        x = x.contiguous()
    mean = mean.reshape(-1, x.shape[1]).contiguous().float()
    S3 = S.reshape(-1, x.shape[1], x.shape[1]).contiguous().float()
    P_, B, n = mean.shape[0], x.shape[0], x.shape[1]
    ll = torch.empty(P_, B, device=x.device)
    g = torch.empty(P_, B, n, device=x.device) if want_grad else None
    check(_lib.lib().svae_gauss_ll(_p(x), x.stride(0), _p(mean), _p(S3), _p(ll), _p(g), B, n, P_, _stream()), "gauss_ll")
    return ll, g


class _GaussLL(torch.autograd.Function):
    """Differentiable in x (the latent means); the streaming means / covariances are buffers."""

    @staticmethod
    def forward(ctx, x, mean, S):
        ll, g = gauss_ll(x.detach(), mean, S, want_grad=x.requires_grad)
        ctx.g = g
        return ll

    @staticmethod
    def backward(ctx, dll):
        return torch.einsum("pb,pbn->bn", dll, ctx.g), None, None


def gauss_ll_autograd(x, mean, S):
    return _GaussLL.apply(x, mean, S)


def kde_mi(x, y, xs, ys, var, logAx, logAy, gamma, want_grad=True):
    """val [B] (per-sample log p(x,y) - log p(x) - log p(y) under the kernel-density mixture) and d val / d x [B, zx] in one launch
    (csrc/latent.hip kde_mi_kernel; reference MutInfoEstimator.forward, disentangle.py:278-317)."""
    x, y = x.float(), y.float()
    if x.stride(-1) != 1:
        x = x.contiguous()
    if y.stride(-1) != 1:
        y = y.contiguous()
    xs, ys, var, logAx = xs.contiguous().float(), ys.contiguous().float(), var.contiguous().float(), logAx.contiguous().float()
    B, zx, dy, S = x.shape[0], x.shape[1], y.shape[1], xs.shape[0]
    per_centre = int(var.numel() != 1)
    assert var.numel() == (S * zx if per_centre else 1) and logAx.numel() == (S if per_centre else 1)
    val = torch.empty(B, device=x.device)
    g = torch.empty(B, zx, device=x.device) if want_grad else None
    check(_lib.lib().svae_kde_mi(_p(x), x.stride(0), _p(y), y.stride(0), _p(xs), _p(ys), _p(var), per_centre, _p(logAx), float(logAy),
                                 float(gamma), _p(val), _p(g), B, S, zx, dy, _stream()), "kde_mi")
    return val, g


class _KdeMI(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, y, xs, ys, var, logAx, logAy, gamma):
        val, g = kde_mi(x.detach(), y, xs, ys, var, logAx, logAy, gamma, want_grad=x.requires_grad)
        ctx.g = g
        return val

    @staticmethod
    def backward(ctx, dval):
        return (dval[:, None] * ctx.g,) + (None,) * 7


def kde_mi_autograd(x, y, xs, ys, var, logAx, logAy, gamma):
    return _KdeMI.apply(x, y, xs, ys, var, logAx, logAy, gamma)


def sumsq_blocks(n):
    return int(_lib.lib().svae_sumsq_blocks(n))


def sumsq_partial(x, part):
    check(_lib.lib().svae_sumsq_partial(_p(x), x.numel(), _p(part), _stream()), "sumsq_partial")


def reduce_rows(part, rows, k, scale, out, accumulate=False):
    check(_lib.lib().svae_reduce_rows(_p(part), rows, k, float(scale), _p(out), int(accumulate), _stream()), "reduce_rows")


def reduce_rows_scaled(part, rows, scales, out):
    """out[c] = scales[c] * sum over rows of part[rows][len(scales)] (one launch; len(scales) <= 8)"""
    k = len(scales)
    check(_lib.lib().svae_reduce_rows_scaled(_p(part), rows, k, (C.c_float * k)(*[float(v) for v in scales]), _p(out), _stream()),
          "reduce_rows_scaled")


def loss_total(terms, weights, out):
    """out[0] = sum_i weights[i] * terms[i] (the reference's running `total`, losses.py:320-322) in one launch"""
    n = len(weights)
    check(_lib.lib().svae_loss_total(_p(terms), (C.c_float * n)(*[float(v) for v in weights]), n, _p(out), _stream()), "loss_total")


def relu_fwd(x, y):
    check(_lib.lib().svae_relu_fwd(_p(x), _p(y), x.numel(), _stream()), "relu_fwd")


def relu_bwd(dy, y, dx):
    check(_lib.lib().svae_relu_bwd(_p(dy), _p(y), _p(dx), y.numel(), _stream()), "relu_bwd")


def axpy(a, x, y):
    check(_lib.lib().svae_axpy(float(a), _p(x), _p(y), x.numel(), _stream()), "axpy")


def fill(x, v):
    check(_lib.lib().svae_fill(_p(x), float(v), x.numel(), _stream()), "fill")


def rowloss_blocks(rows):
    return int(_lib.lib().svae_rowloss_blocks(rows))


def mse_sum(pred, ld, target, ld_t, rows, Cn, scale, part, dpred):
    check(_lib.lib().svae_mse_sum(_p(pred), ld, _p(target), ld_t, rows, Cn, float(scale), _p(part), _p(dpred), _stream()), "mse_sum")


def ce_sum(logits, ld, labels, rows, Cn, scale, part, dlogits):
    check(_lib.lib().svae_ce_sum(_p(logits), ld, _p(labels), rows, Cn, float(scale), _p(part), _p(dlogits), _stream()), "ce_sum")


def double_softmax_ce_sum(logits, ld, rows, scale, part, dlogits):
    check(_lib.lib().svae_double_softmax_ce_sum(_p(logits), ld, rows, float(scale), _p(part), _p(dlogits), _stream()), "double_softmax_ce_sum")


# ------------------------------------------------------------------- fused MLP ensembles
FUSED_ENSEMBLE = True  # False: every ensemble Linear by Linear on the GEMM kernels (the path wide ensembles take anyway)


def ens_fwd(desc):
    check(_lib.lib().svae_ens_fwd(C.byref(desc), _stream()), "ens_fwd")


def ens_bwd_workspace(desc):
    return int(_lib.lib().svae_ens_bwd_workspace(C.byref(desc)))


def ens_bwd(desc, d_src0, ld_d, coef, gx_raw, ws, accumulate):
    check(_lib.lib().svae_ens_bwd(C.byref(desc), _p(d_src0), ld_d, float(coef), _p(gx_raw), _p(ws), ws.numel() * ws.element_size(),
                                  int(accumulate), _stream()), "ens_bwd")


def ens_loss(kind, outs, dpreds, loss_w, grad_s, target, ld_t, labels, rows, Cn, ld, part):
    n = len(outs)
    o = (C.c_void_p * n)(*[t.data_ptr() for t in outs])
    d = (C.c_void_p * n)(*[(None if t is None else t.data_ptr()) for t in dpreds])
    lw = (C.c_float * n)(*[float(v) for v in loss_w])
    gs = (C.c_float * n)(*[float(v) for v in grad_s])
    check(_lib.lib().svae_ens_loss(int(kind), o, d, lw, gs, n, _p(target), int(ld_t), _p(labels), rows, Cn, ld, _p(part), _stream()),
          "ens_loss")

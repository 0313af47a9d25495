"""MI355X-native ResVAE: the reference's module tree and Python API
(src/scrubvae/model/residual.py:183-491) over the HIP kernels of libscrubvae_hip.so.

``ResVAE.forward(data) -> data_o`` / ``encode`` / ``decode`` keep the reference's dict
contract.  Underneath, the trunk is an explicit kernel schedule:

  forward : pack -> conv_in -> [conv|conv -> BN-stats -> BN+PReLU -> conv(+=) -> BN+PReLU] x4
            -> fc_mu, fc_sigma -> heads(softplus, z = mu + sigma*eps, KL partials)
            -> fc_in -> [convT -> BN+PReLU -> convT ; upsample -> conv(+=) -> BN+PReLU] x4
            -> conv_out -> fused tanh/unpack/FK/JPE/root tail
  backward: the hand-written reverse schedule (``backward_from_seeds``), gradients land in
            one flat buffer that every ``param.grad`` is a view of.

Activations are channels-last [B*L, Cp] so the reference's moveaxis copies vanish and
every conv is an implicit GEMM over contiguous rows.  There is no autograd graph through
the trunk: ``get_batch_loss`` (train/losses.py) returns a ``total`` whose ``.backward()``
runs the reverse schedule.
"""
from __future__ import annotations

import os

import torch
import torch.nn as nn
import torch.nn.functional as F

from .. import ops
from ..ops import pad16
from .._lib import make_tree
from .layers import ConvP, LinearP, BatchNormP, PReLUP, Marker, Leaf, make_activation, slope_grad
from .disentangle import GRScrubber, AdvNetScrubber, MLPEnsemble, EnsembleRunner, FusedEnsembleRunner


def find_latent_dim(window_size, kernel, num_layers, dilation=None):
    """residual.py:6-20 -- float division per layer, one int() at the end (quirk kept)."""
    dil = [1] * num_layers if dilation is None else [int(d) for d in dilation]
    stride = 1 if any(d > 1 for d in dil) else 2
    l_out = window_size
    for i in range(num_layers):
        l_out = (l_out + 2 * (kernel // 2) - dil[i] * (kernel - 1) - 1) / stride + 1
    return int(l_out)


def find_out_dim(latent_dim, kernel, num_layers, dilation=None):
    """residual.py:23-36 (indexes dilation[-i] with i from 0: quirk kept)."""
    dil = [1] * num_layers if dilation is None else [int(d) for d in dilation]
    stride = 1 if any(d > 1 for d in dil) else 2
    l_out = latent_dim
    for i in range(num_layers):
        l_out = (l_out - 1) * stride - 2 * (kernel // 2) + dil[-i] * (kernel - 1) + 1
    return int(l_out)


_APPLY_COLSUM = os.environ.get("SVAE_APPLY_COLSUM", "1") != "0"  # env: experiments (0 = bias gradients by a separate pass over dY)
# env: A/B measurements ("0": upsample2_fwd always writes the `up` tensor the skip convs read; "force": fused wherever the kernels
# exist, tuned on first use; default: fused where the tile table says so)
FUSE_UPSAMPLE = os.environ.get("SVAE_FUSE_UPSAMPLE", "1")
_SIDE_STREAMS = {}  # device index -> side streams shared by every model of the process (see ResVAE._side_stream)


class ResidualBlock(nn.Module):
    """residual.py:71-119: skip Conv1d(s=2) || [Conv1d(s=2) -> BN -> PReLU -> Conv1d] ; add ; BN ; PReLU."""

    def __init__(self, in_channels, out_channels, kernel=3, activation="prelu", dilation=1):
        super().__init__()
        if dilation != 1:
            raise NotImplementedError("init_dilation is not supported by the HIP trunk yet")
        self.residual = nn.Sequential(
            ConvP(in_channels, out_channels // 2, kernel, 2, kernel // 2),
            BatchNormP(out_channels // 2), make_activation(activation),
            ConvP(out_channels // 2, out_channels, kernel, 1, kernel // 2))
        self.skip = ConvP(in_channels, out_channels, kernel, 2, kernel // 2)
        self.add = nn.Sequential(BatchNormP(out_channels), make_activation(activation))


class ResidualBlockTranspose(nn.Module):
    """residual.py:122-180: skip [Upsample x2 -> Conv1d(k+1)] || [ConvT(s=1) -> BN -> PReLU -> ConvT(s=2)]."""

    def __init__(self, in_channels, out_channels, kernel=3, scale_factor=2, activation="prelu", dilation=1):
        super().__init__()
        if dilation != 1 or scale_factor != 2:
            raise NotImplementedError("only dilation 1 / scale 2")
        self.residual = nn.Sequential(
            ConvP(in_channels, in_channels // 2, kernel, 1, kernel // 2, transposed=True),
            BatchNormP(in_channels // 2), make_activation(activation),
            ConvP(in_channels // 2, out_channels, kernel, 2, kernel // 2, transposed=True))
        self.skip = nn.Sequential(Marker("Upsample(scale_factor=2, mode=linear)"),
                                  ConvP(in_channels, out_channels, kernel + 1, 1, kernel // 2))
        self.add = nn.Sequential(BatchNormP(out_channels), make_activation(activation))


def _flat_index(C, L):
    """library feature l*Cp+c  ->  reference flatten index c*L+l (-1 on padded channels)."""
    Cp = pad16(C)
    idx = torch.full((L * Cp,), -1, dtype=torch.long)
    for l in range(L):
        idx[l * Cp: l * Cp + C] = torch.arange(C) * L + l
    return idx


class ResidualEncoder(nn.Module):
    """residual.py:183-240."""

    def __init__(self, in_channels, ch, kernel, z_dim, window, activation="prelu", is_diag=False, prior="gaussian",
                 init_dilation=None):
        super().__init__()
        if prior not in ("gaussian", "beta"):
            raise ValueError(f"prior {prior!r} not in ('gaussian', 'beta')")
        if init_dilation is not None:
            raise NotImplementedError("init_dilation is not supported by the HIP trunk yet")
        self.conv_in = ConvP(in_channels, ch[0], 7, 1, 3)
        self.activation = make_activation(activation)
        self.res_layers = nn.Sequential(*[ResidualBlock(ch[i], ch[i + 1], kernel, activation) for i in range(len(ch) - 1)])
        self.latent_len = find_latent_dim(window, kernel, len(ch) - 1)
        flatten_dim = self.latent_len * ch[-1]
        sig_dim = z_dim if is_diag else z_dim * (z_dim + 1) // 2
        idx = _flat_index(ch[-1], self.latent_len)
        if prior == "beta":  # residual.py:223-225: Beta posterior, alpha = softplus(fc_alpha) + 1, beta likewise
            self.fc_alpha = LinearP(flatten_dim, z_dim, in_index=idx)
            self.fc_beta = LinearP(flatten_dim, z_dim, in_index=idx)
        else:
            self.fc_mu = LinearP(flatten_dim, z_dim, in_index=idx)
            self.fc_sigma = nn.Sequential(LinearP(flatten_dim, sig_dim, in_index=idx), Marker("CholeskyL"))


class ResidualDecoder(nn.Module):
    """residual.py:243-292."""

    def __init__(self, out_channels, ch, kernel, z_dim, window, activation="prelu", conditional_dim=0, init_dilation=None):
        super().__init__()
        self.conditional_dim = conditional_dim
        n = len(ch) - 1
        self.latent_len = find_latent_dim(window, kernel, n)
        flatten_dim = self.latent_len * ch[-1]
        self.fc_in = LinearP(z_dim + conditional_dim, flatten_dim, out_index=_flat_index(ch[-1], self.latent_len))
        self.res_layers = nn.Sequential(*[ResidualBlockTranspose(ch[-i], ch[-i - 1], kernel, activation=activation)
                                          for i in range(1, len(ch))])
        l_out = find_out_dim(self.latent_len, kernel, n)
        self.final_kernel = window - l_out + 7
        self.conv_out = ConvP(ch[0], out_channels, self.final_kernel, 1, 3, transposed=True)


class _JointLinear:
    """fc_mu || fc_sigma[0] as one Linear: the joint weight / bias tensors (views of the flat parameter buffer, `.grad` = the same
    regions of the flat gradient buffer) with the attributes ResVAE._lin / _wgrad read from a LinearP."""

    def __init__(self, weight, bias, in_lib, out_lib, in_f, out_f):
        self.weight, self.bias = weight, bias
        self.in_lib, self.out_lib, self.in_f, self.out_f = in_lib, out_lib, in_f, out_f


class _LazyList:
    """A list evaluated on first use (len / index / iteration): data_o entries nobody may read in a training step."""

    def __init__(self, make):
        self._make, self._val = make, None

    def _get(self):
        if self._val is None:
            self._val, self._make = self._make(), None
        return self._val

    def __len__(self):
        return len(self._get())

    def __getitem__(self, i):
        return self._get()[i]

    def __iter__(self):
        return iter(self._get())


class _TotalLoss(torch.autograd.Function):
    """``batch_loss["total"]``: its backward runs the HIP reverse schedule and fills the
    parameters' ``.grad`` (reference: ``batch_loss["total"].backward()``, trainer.py:163)."""

    @staticmethod
    def forward(ctx, anchor, total, model):
        ctx.model = model
        return total.detach().clone()

    @staticmethod
    def backward(ctx, grad_out):
        ctx.model.backward_from_seeds()
        return None, None, None


class ResVAE(nn.Module):
    """Drop-in for scrubvae.model.residual.ResVAE (residual.py:365-491) on MI355X."""

    def __init__(self, in_channels, ch=[64, 128, 256, 512, 1024], kernel=5, z_dim=128, window=200, activation="prelu",
                 is_diag=False, conditional_dim=0, init_dilation=None, disentangle=None, kinematic_tree=None,
                 arena_size=None, disentangle_keys=None, conditional_keys=None, discrete_classes=None,
                 prior="gaussian", device="cuda"):
        super().__init__()
        self.prior = prior
        self.dist_params = ["alpha", "beta"] if prior == "beta" else ["mu", "L"]  # residual.py:299-302
        self.in_channels, self.ch, self.window = in_channels, list(ch), window
        self.kernel, self.z_dim = kernel, z_dim
        self.is_diag = is_diag
        self.conditional_dim = conditional_dim
        self.kinematic_tree = kinematic_tree
        self.register_buffer("arena_size", None if arena_size is None else torch.as_tensor(arena_size, dtype=torch.float32))
        self.disentangle_keys = disentangle_keys
        self.conditional_keys = conditional_keys
        self.discrete_classes = discrete_classes
        self.n_keypts = (in_channels - (3 if arena_size is not None else 0)) // 6
        self.sig_dim = z_dim if (is_diag or prior == "beta") else z_dim * (z_dim + 1) // 2  # beta: the second head is fc_beta [z]
        self._hw = pad16(z_dim) + pad16(self.sig_dim)  # row width of the [mu | raw] head buffer
        self.encoder = ResidualEncoder(in_channels, ch, kernel, z_dim, window, activation, is_diag, prior, init_dilation)
        self.decoder = ResidualDecoder(in_channels, ch, kernel, z_dim, window, activation, conditional_dim, init_dilation)
        if self.encoder.latent_len < 1:
            raise ValueError("window too short for the number of residual blocks")
        self.disentangle = nn.ModuleDict()
        if disentangle is not None:
            for k, v in disentangle.items():
                self.disentangle[k] = nn.ModuleDict(v)
        self.mi_estimator = None
        # engine state
        self.world_size, self.rank, self.process_group = 1, 0, None
        self.bucket_min_bytes = 4 << 20  # smallest encoder gradient bucket worth its own all-reduce (xGMI: few, large)
        self.sync_bn = True
        self.shuffle_seed, self._shuffle_draws = 0, 0  # shared-seed global permutation of the adversarial shuffle (N > 1)
        # bench.py: HIP events (on the main stream) around every point where the main chain waits for a collective -- the
        # sync-BatchNorm statistics and the tail of the gradient exchange; what they bracket is the EXPOSED communication time
        self.time_comm, self._comm_events = False, []
        # training fast path: skip the forward-time tail launch; data_o["x6d"/"root"] are then
        # only valid after get_batch_loss (which runs the fused tail once).  Off by default.
        # Contract of the fast path: get_batch_loss may already write (or, with accumulate_grads, add) the scrubber heads'
        # parameter gradients into flat_grads, beside the tail; between get_batch_loss and total.backward() the caller may
        # re-bind gradients (`param.grad = None`, the reference loop) but must not zero them in place
        # (`optimizer.zero_grad(set_to_none=False)`), and every get_batch_loss must be followed by exactly one backward.
        self.defer_tail = False
        self._tail_done = True
        # weight-gradient GEMMs / skip branches on side streams, concurrent with the main chain.  None = decide per pass from the
        # batch: with fewer than `overlap_min_rows` (batch x window) rows the kernels are shorter than the cross-stream hand-overs
        # cost (measured on configs[1]: B=32 8.3k -> 14k, B=256 51k -> 77k windows/s WITHOUT the side streams; B=1024 194k -> 212k with)
        self.overlap_wgrad = None
        self.overlap_min_rows = 32768
        self._ov = 2
        self._sides, self._side_dirty, self._events, self._event_i = [], set(), [], 0
        self._ws = {}
        self._convs = {}
        self._split_users = {}  # conv key -> (Conv, parameter module) of every conv on the split-bf16 path
        self._runners = {}
        self._pending = None
        self._db_batch = ops.ColsumBatch()
        self._dx_colsum = {}  # data_ptr of a dX produced by a backward apply pass -> (partials, their rows, rows, Cp)
        self._tree = make_tree(self.n_keypts, kinematic_tree) if kinematic_tree is not None else None
        self._arena_host = None if arena_size is None else [float(v) for v in torch.as_tensor(arena_size).flatten()]
        self._materialise(torch.device(device))
        self.reset_parameters()

    # ------------------------------------------------------------------ parameters
    def _materialise(self, device):
        leaves = [(n, m) for n, m in self.named_modules() if isinstance(m, Leaf)]
        # Joint parameters: fc_mu and fc_sigma[0] (residual.py:219-222) read the same flattened activations, so their weights live
        # side by side in ONE [1][in][z_p + sig_p] tensor (biases likewise) and each module's parameter is a column slice of it:
        # the two Linear layers are one GEMM forward, one data-gradient and one weight-gradient launch (module tree, parameter
        # names, shapes and state_dict unchanged; the slices are ordinary -- non-contiguous -- Parameters).
        fm, fs = ((self.encoder.fc_alpha, self.encoder.fc_beta) if self.prior == "beta" else
                  (self.encoder.fc_mu, self.encoder.fc_sigma[0]))
        joint = {}  # (id(leaf), pname) -> (joint key, joint shape, axis, start, length)
        n_mu, n_sig = fm.out_lib, fs.out_lib
        for leaf, start, length in ((fm, 0, n_mu), (fs, n_mu, n_sig)):
            joint[(id(leaf), "weight")] = ("heads.w", (1, fm.in_lib, n_mu + n_sig), 2, start, length)
            joint[(id(leaf), "bias")] = ("heads.b", (n_mu + n_sig,), 0, start, length)
        total = 0
        slots = []
        regions = {}
        for n, m in leaves:
            for pname, shape in m.specs.items():
                j = joint.get((id(m), pname))
                if j is not None:
                    key, jshape, axis, start, length = j
                    if key not in regions:
                        jn = 1
                        for d in jshape:
                            jn *= d
                        regions[key] = (total, jn)
                        total += (jn + 3) // 4 * 4
                    off, numel = regions[key]
                    slots.append((m, pname, shape, off, numel, (jshape, axis, start, length)))
                    continue
                numel = 1
                for d in shape:
                    numel *= d
                slots.append((m, pname, shape, total, numel, None))
                total += (numel + 3) // 4 * 4
        self.flat_params = torch.zeros(total, device=device)
        self.flat_grads = torch.zeros(total, device=device)
        self._slots = []
        frozen = set()
        for n, m in self.named_modules():
            if isinstance(m, AdvNetScrubber):
                frozen.update(id(x) for x in m.modules())
        for m, pname, shape, off, numel, jn in slots:
            if jn is None:
                view, eoff = self.flat_params[off: off + numel].view(shape), 0
            else:
                jshape, axis, start, length = jn
                view = self.flat_params[off: off + numel].view(jshape).narrow(axis, start, length)
                eoff = start  # the slices run along the last (unit-stride) axis: element offset of the view in its region
            p = nn.Parameter(view, requires_grad=id(m) not in frozen)
            m.register_parameter(pname, p)
            self._slots.append((p, off, numel, shape, jn, eoff))
        # the joint head tensors themselves (plain views of the flat buffers, not Parameters: the optimizer sees the slices)
        wo, wn = regions["heads.w"]
        bo, bn_ = regions["heads.b"]
        jw = self.flat_params[wo: wo + wn].view(1, fm.in_lib, n_mu + n_sig)
        jb = self.flat_params[bo: bo + bn_]
        jw.grad = self.flat_grads[wo: wo + wn].view(1, fm.in_lib, n_mu + n_sig)
        jb.grad = self.flat_grads[bo: bo + bn_]
        self._fc_heads = _JointLinear(jw, jb, fm.in_lib, n_mu + n_sig, fm.in_f, fm.out_f + fs.out_f)
        slots = [t[:5] for t in slots]
        # contiguous span of the decoder's parameters in the flat buffers (first all-reduce bucket)
        dec_ids = {id(m) for m in self.decoder.modules()}
        dec = [(off, off + (numel + 3) // 4 * 4) for m, pname, shape, off, numel in slots if id(m) in dec_ids]
        self._dec_span = (min(a for a, _ in dec), max(b for _, b in dec))
        # start offsets of the encoder blocks (module order == flat order): the reverse schedule finishes block i last of
        # everything at or above _enc_cuts[i], so [cut_i, previous cut) can be all-reduced while blocks < i still run
        enc_ids = {id(m) for m in self.encoder.modules()}
        enc_hi = max(off + numel for m, pname, shape, off, numel in slots if id(m) in enc_ids)
        assert enc_hi <= self._dec_span[0], "gradient buckets assume the encoder's parameters precede the decoder's in the flat buffer"
        self._enc_cuts, self._enc_mid_cuts = [], []
        for blk in self.encoder.res_layers:
            ids = {id(m) for m in blk.modules()}
            self._enc_cuts.append(min(off for m, pname, shape, off, numel in slots if id(m) in ids))
            # inside a block the order is residual.0-3, skip, add.0-1: everything from residual.3 on is final as soon as the
            # second BatchNorm's backward and the two weight gradients that read its output gradient are queued
            self._enc_mid_cuts.append(min(off for m, pname, shape, off, numel in slots if m is blk.residual[3]))
        # contiguous [lo, hi) element ranges of the flat buffers that hold TRAINABLE parameters: the optimizer, like torch's,
        # only touches those (the AdvNetScrubber ensemble is frozen, disentangle.py:670-671 -- AdamW must not decay it)
        spans = []
        seen = set()
        for m, pname, shape, off, numel in slots:
            if id(m) in frozen or off in seen:  # (the slices of a joint tensor share one region)
                continue
            seen.add(off)
            end = off + (numel + 3) // 4 * 4
            if spans and spans[-1][1] == off:
                spans[-1][1] = end
            else:
                spans.append([off, end])
        self.trainable_spans = [tuple(s) for s in spans]
        self.to(device)  # buffers
        self._assign_grad_views()

    def _assign_grad_views(self):
        """Make every trainable parameter's .grad a view of the flat gradient buffer (the
        reference loop sets param.grad = None before backward, trainer.py:160-161)."""
        base = self.flat_grads.data_ptr()
        for p, off, numel, shape, jn, eoff in self._slots:
            if not p.requires_grad:
                continue
            if p.grad is None or p.grad.data_ptr() != base + 4 * (off + eoff):
                if jn is None:
                    p.grad = self.flat_grads[off: off + numel].view(shape)
                else:  # column slice of a joint tensor's gradient
                    p.grad = self.flat_grads[off: off + numel].view(jn[0]).narrow(jn[1], jn[2], jn[3])

    def _apply(self, fn, recurse=True):
        # parameters are views of flat_params: moving/casting them individually would break
        # the aliasing, so only buffers follow .to()/.cuda(); the flat storage is fixed at
        # construction (device=...).
        for m in self.modules():
            for k, b in m._buffers.items():
                if b is not None:
                    m._buffers[k] = fn(b)
        return self

    def reset_parameters(self):
        for m in self.modules():
            if isinstance(m, Leaf):
                m.reset_parameters()

    def grads_state_dict(self):
        """Gradients under the reference's parameter names and layouts (for parity checks)."""
        out = {}
        for name, m in self.named_modules():
            if isinstance(m, Leaf):
                for pn, p in m._parameters.items():
                    if p is not None and p.grad is not None:
                        out[f"{name}.{pn}"] = m.export_tensor(pn, p.grad)
        return out

    @property
    def device(self):
        return self.flat_params.device

    # ------------------------------------------------------------------ workspace helpers
    def _buf(self, name, shape, zero=False):
        key = (name, tuple(shape))
        t = self._ws.get(key)
        if t is None:
            t = (torch.zeros if zero else torch.empty)(shape, device=self.device, dtype=torch.float32)
            self._ws[key] = t
        return t

    def _conv(self, name, p: ConvP, batch, l_in, ld_in=None, ld_out=None, up2=False):
        key = (name, batch, l_in, ld_in, ld_out)
        c = self._convs.get(key)
        if c is None:
            c = ops.Conv(batch, l_in, p.c_in, p.c_out, p.kernel, p.stride, p.padding, p.dilation, p.transposed, ld_in, ld_out)
            if up2 and FUSE_UPSAMPLE != "0" and c.pieces and ops.up2_supported(p.kernel, p.stride, p.dilation, p.transposed):
                # the conv behind the decoder's Upsample(x2): its forward can blend the half-length input on the fly.  It does where
                # that was MEASURED to pay -- the geometry has an ":up2" entry in the tile table (large batches: the fused kernel
                # fetches half the operand rows; at small ones the upsample pass is cheap and the kernel choice narrower)
                cu = ops.Conv(batch, l_in, p.c_in, p.c_out, p.kernel, p.stride, p.padding, p.dilation, p.transposed, ld_in, ld_out, up2=True)
                if FUSE_UPSAMPLE == "force" or cu.tile_key("fwd") in ops.TILE_TABLE:
                    c = cu
            self._convs[key] = c
            if c.pieces:
                self._split_users[key] = (c, p)
        return c

    def _lin(self, name, p: LinearP, batch, ld_in=None, ld_out=None, pieces=None):
        key = (name, batch, ld_in, ld_out)
        c = self._convs.get(key)
        if c is None:
            c = ops.Conv(batch, 1, p.in_lib, p.out_lib, 1, ld_in=ld_in, ld_out=ld_out, pieces=pieces)
            c.flops = 2.0 * batch * p.in_f * p.out_f  # algorithmic: unpadded features
            self._convs[key] = c
            if c.pieces:
                self._split_users[key] = (c, p)
        return c

    def _heads_pieces(self):
        """Arithmetic of the joint fc_mu || fc_sigma GEMM: with a diagonal factor it is skinny (2 z columns over a deep reduction) --
        the fp32 split-K kernels whatever the batch; a full Cholesky factor (z (z + 3) / 2 columns) follows the model's precision."""
        return 0 if (self.is_diag or self.prior == "beta") else None

    def _new_pass(self):
        """Start of forward / encode / decode: the master weights may have changed since the last pass, so
        the split-bf16 weight copies of every conv seen so far are refreshed in one launch (convs met for
        the first time in this pass split lazily at their first use)."""
        self.__dict__["_main"] = None
        if self.time_comm and self.training:
            self._comm_events.append([])
        ops.check_current_device(self.device)
        ops.bump_weight_epoch()
        if self._split_users:
            ops.split_weights_batched([(c, p.weight) for c, p in self._split_users.values()])

    def _wgrad_ws(self, conv):
        n = conv.wgrad_workspace_bytes() // 4 + 16
        t = self._ws.get("wgrad_ws")
        if t is None or t.numel() < n:
            t = torch.empty(max(n, 1 << 20), device=self.device)
            self._ws["wgrad_ws"] = t
        return t

    def _wgrad(self, cv, x, dy, p, acc):
        """Weight gradient; the bias gradient (column sums of dy) is queued and reduced with all
        the others in two launches at the end of the reverse schedule.

        Weight gradients are off the critical path of the reverse schedule (nothing downstream
        reads them), so they go to a second HIP stream: they fill the CUs that the
        data-gradient / BatchNorm chain leaves idle (single-wave grids, kernel tails).  The side
        stream is forked from the main stream after `dy` is produced and joined before anything
        consumes the gradients (_join_side)."""
        self._fork(lambda: cv.wgrad(x, dy, p.weight.grad, None, self._wgrad_ws(cv), accumulate=acc), k=0)
        ent = self._dx_colsum.get(dy.data_ptr())
        if ent is not None and ent[2:] == (cv.batch * cv.l_out, cv.c_out_p) and cv.desc.ld_out == cv.c_out_p:
            # dy came out of a BatchNorm / activation backward pass that left its column sums behind
            self._db_batch.add_partials(ent[0], ent[1], cv.c_out_p, p.bias.grad)
        else:
            self._db_batch.add(dy, cv.batch * cv.l_out, cv.c_out_p, cv.desc.ld_out, p.bias.grad)

    def _dx_colsum_part(self, tag, dx, rows, Cp):
        """Buffer for the column-sum partials of `dx` written by the backward apply pass of stage `tag` (None where the kernel cannot
        produce them); remembered by the address of dx so that _wgrad finds them when dx is a conv's dY."""
        n = ops.affine_prelu_colsum_rows(rows, Cp) if _APPLY_COLSUM else 0
        if n == 0:
            return None
        part = self._buf(f"{tag}.dbpart", (n, Cp))
        self._dx_colsum[dx.data_ptr()] = (part, n, rows, Cp)
        return part

    def _side_stream(self, k=0):
        """Side stream k of this model's device, from a PROCESS-WIDE pool: the runtime multiplexes streams onto a few hardware
        queues, and a second model with streams of its own (an evaluation model beside the training model; bench.py's second
        workload) found two of its three streams on one queue -- the batch-1024 step ran at its serial 5.3 ms instead of 4.9."""
        pool = _SIDE_STREAMS.setdefault(torch.device(self.device).index or 0, [])
        while len(pool) <= k:
            pool.append(torch.cuda.Stream(device=self.device))
        while len(self._sides) <= k:
            self._sides.append(pool[len(self._sides)])
        return self._sides[k]

    def _event(self):
        """Round-robin pool of events (an event may be re-recorded once its waiters were enqueued)."""
        if len(self._events) < 64:
            self._events.append(torch.cuda.Event())
            return self._events[-1]
        self._event_i = (self._event_i + 1) % 64
        return self._events[self._event_i]

    def _overlap_level(self, B):
        """Stream schedule of a pass over B windows: 2 = weight gradients, skip branches and the fused tail on side streams
        (from `overlap_min_rows` batch x window rows), 1 = only the weight gradients leave the main stream (from a quarter of
        that: B = 128-511 at window 64 measured 3-5 % over the serial schedule, the full one loses there), 0 = one stream
        (below: every cross-stream hand-over costs more than it overlaps).  `overlap_wgrad` = True / False / 0..2 overrides."""
        ow = self.overlap_wgrad
        if ow is not None:
            return 2 if ow is True else int(ow)
        rows = B * self.window
        return 2 if rows >= self.overlap_min_rows else (1 if rows * 4 >= self.overlap_min_rows else 0)

    def _fork(self, fn, k=1):
        """Run fn() on side stream k, ordered after everything queued on the main stream so far.
        Stream 0 carries the weight gradients, stream 1 the skip branches."""
        if not self._ov or (k != 0 and self._ov == 1):
            return fn()
        main = self._main_stream()
        side = self._side_stream(k)
        ev = self._event()
        ev.record(main)
        side.wait_event(ev)
        if ops.TIMER is None:
            # the bodies forked here are C-ABI launches only: point them at the side stream directly
            ops._TLS.stream_override = side.cuda_stream
            try:
                fn()
            finally:
                ops._TLS.stream_override = None
        else:  # the launch timer records torch events on torch's current stream
            with torch.cuda.stream(side):
                fn()
        self._side_dirty.add(k)

    def _join_side(self, k=None):
        """Make the main stream wait for side stream k (None: all of them)."""
        for kk in (list(self._side_dirty) if k is None else [k]):
            if kk in self._side_dirty:
                self._main_stream().wait_stream(self._sides[kk])
                self._side_dirty.discard(kk)

    def _main_stream(self):
        """torch's current stream, looked up once per pass (torch.cuda.current_stream() is ~8 us a call and the reverse schedule
        forks / joins ~60 times); _new_pass() and backward_from_seeds() drop the cached handle."""
        m = self.__dict__.get("_main")
        if m is None:
            m = self.__dict__["_main"] = torch.cuda.current_stream()
        return m

    def _colsum_ws(self, nbytes):
        n = nbytes // 4 + 16
        t = self._ws.get("colsum_ws")
        if t is None or t.numel() < n:
            t = torch.empty(n, device=self.device)
            self._ws["colsum_ws"] = t
        return t

    def _allreduce(self, t, async_op=False):
        """SUM all-reduce over the data-parallel group.  async_op: returns a work handle (RCCL
        runs it on its own stream after the kernels already queued on the current stream)."""
        if self.world_size > 1:
            import torch.distributed as dist
            if t.is_cuda and dist.get_backend(self.process_group) == "gloo":
                # test configuration (several ranks sharing one GPU): stage through the host
                h = t.detach().cpu()
                dist.all_reduce(h, group=self.process_group)
                t.copy_(h)
                return None
            return dist.all_reduce(t, group=self.process_group, async_op=async_op)
        return None

    def _comm_bracket(self, kind):
        """Context manager: with `time_comm` set, records a pair of timing events on the main stream around the enclosed
        collective(s) -- the time the main chain spends waiting for the exchange (kind: "bn" | "grads")."""
        import contextlib
        if not self.time_comm:
            return contextlib.nullcontext()

        @contextlib.contextmanager
        def bracket():
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(self._main_stream())
            yield
            e1.record(self._main_stream())
            if not self._comm_events:
                self._comm_events.append([])
            self._comm_events[-1].append((kind, e0, e1))
        return bracket()

    def pop_comm_times(self):
        """Per optimizer step since the last call: {"grads": ms, "bn": ms} the main stream waited for collectives (needs
        `time_comm`; synchronises the device)."""
        torch.cuda.synchronize(self.device)
        out = []
        for step in self._comm_events:
            d = {"grads": 0.0, "bn": 0.0}
            for kind, e0, e1 in step:
                d[kind] += e0.elapsed_time(e1)
            if step:
                out.append(d)
        self._comm_events = []
        return out

    def _allgather(self, t):
        """Concatenation of the ranks' 1-D tensors `t` in rank order (same length on every rank)."""
        import torch.distributed as dist
        if self.world_size == 1 and not dist.is_initialized():
            return t
        n = dist.get_world_size(self.process_group)  # (the group's own size: a test may fake model.world_size)
        if t.is_cuda and dist.get_backend(self.process_group) == "gloo":  # test configuration: stage through the host
            h = t.detach().cpu()
            out = torch.empty(n * h.numel(), dtype=h.dtype)
            dist.all_gather_into_tensor(out, h, group=self.process_group)
            return out.to(t.device)
        out = torch.empty(n * t.numel(), dtype=t.dtype, device=t.device)
        dist.all_gather_into_tensor(out, t, group=self.process_group)
        return out

    def global_permutation(self, n):
        """The adversarial shuffle's permutation of the GLOBAL batch (AdvNetScrubber.shuffle, disentangle.py:678-684) under data
        parallelism: drawn on the host from a generator every rank seeds identically (shuffle_seed, advanced per draw), so all
        ranks hold the same permutation without a broadcast."""
        g = torch.Generator().manual_seed((int(self.shuffle_seed) * 1000003 + self._shuffle_draws) & 0x7FFFFFFFFFFFFFFF)
        self._shuffle_draws += 1
        return torch.randperm(n, generator=g)

    def _bucket_allreduce(self, lo, hi, acc):
        """Data-parallel gradient bucket: flat_grads[lo:hi] is final once everything queued so far on the main and side
        streams has run.  The all-reduce is fed from its own stream, which waits for those streams -- the main chain
        itself neither waits for the weight-gradient stream nor for the collective, so the transfer overlaps the rest of
        the reverse schedule.  Returns the work handle (None in the gloo test configuration, which reduces in place)."""
        import torch.distributed as dist
        self._db_batch.flush(self._colsum_ws, accumulate=acc)  # bias gradients queued so far (main stream)
        if hi <= lo:
            return None
        t = self.flat_grads[lo:hi]
        if dist.get_backend(self.process_group) == "gloo" or not self._ov:
            self._join_side()
            return self._allreduce(t, async_op=True)
        comm = self._side_stream(2)
        comm.wait_stream(self._main_stream())
        for kk in self._side_dirty:
            comm.wait_stream(self._sides[kk])
        with torch.cuda.stream(comm):
            return dist.all_reduce(t, group=self.process_group, async_op=True)

    # ------------------------------------------------------------------ BN + PReLU stage
    def _conv_fwd_bn(self, cv, x, p, y, accumulate=False):
        """Conv forward whose output feeds a train-mode BatchNorm: where the conv runs a split-bf16 kernel the per-tile
        (sum, sum of squares) of the output come out of the GEMM epilogue.  Returns (part [tiles, 2, Cp], tiles) for _bn_act,
        or None (eval mode / fp32 kernel: _bn_act makes its own statistics pass)."""
        if self.training:
            cv.tune_fwd(x, p.weight, p.bias)
            n = cv.stats_tiles()
            if n > 0:
                part = self._buf(f"bn.tpart.{n}.{cv.c_out_p}", (n, 2, cv.c_out_p))
                cv.fwd(x, p.weight, p.bias, y, accumulate=accumulate, stats=part)
                return part, n
        cv.fwd(x, p.weight, p.bias, y, accumulate=accumulate)
        return None

    def _bn_act(self, tag, x, bn: BatchNormP, act: PReLUP, rows, out, stats=None):
        Cp = pad16(bn.c)
        scale, shift = self._buf(tag + ".scale", (Cp,)), self._buf(tag + ".shift", (Cp,))
        if self.training:
            sums = self._buf(tag + ".sums", (2, Cp))
            mean, rstd = self._buf(tag + ".mean", (Cp,)), self._buf(tag + ".rstd", (Cp,))
            if stats is not None:  # statistics came out of the producing GEMM's epilogue
                part, nch = stats
            else:
                nch = ops.bn_chunks(rows)
                part = self._buf(f"bn.part.{nch}.{Cp}", (nch, 2, Cp))
                ops.bn_stats_partial(x, rows, Cp, Cp, part)
            if self.world_size > 1 and self.sync_bn:
                ops.bn_reduce_partials(part, nch, Cp, sums)
                with self._comm_bracket("bn"):
                    self._allreduce(sums)
                ops.bn_finalize(sums, rows * self.world_size, Cp, bn.weight, bn.bias, bn.eps, bn.momentum, bn.running_mean,
                                bn.running_var, mean, rstd, scale, shift)
                bn.num_batches_tracked.add_(1)
            else:  # one launch: chunk reduction + finalize + running stats + num_batches_tracked
                ops.bn_stats_finalize(part, nch, rows, Cp, bn.weight, bn.bias, bn.eps, bn.momentum, bn.running_mean,
                                      bn.running_var, bn.num_batches_tracked, mean, rstd, scale, shift)
        else:
            ops.bn_eval_coeffs(Cp, bn.weight, bn.bias, bn.eps, bn.running_mean, bn.running_var, scale, shift)
        ops.affine_prelu_fwd(x, scale, shift, act.weight, out, rows, Cp, Cp)
        return out

    def _dgrad_into_bn(self, cv, dy, w, dx, accumulate, tag, x, act, bare=False):
        """Data-gradient launch whose output `dx` is the gradient with respect to the OUTPUT of the BatchNorm + activation stage
        `tag` (saved input `x`): where the launch runs a split-bf16 kernel, the first pass of that stage's backward (sum du,
        sum du * xhat, slope partial) comes out of its epilogue.  Returns what _bn_act_bwd needs to skip its own pass, or None."""
        from .._lib import BnBwdFuse
        cv.tune_dgrad(dy, w)
        n, cb = cv.dgrad_stats_tiles()
        if n == 0 or x.shape[-1] != cv.desc.ld_in:
            cv.dgrad(dy, w, dx, accumulate=accumulate)
            return None
        Cp = cv.c_in_p
        part = self._buf(f"bn.tpart.{n}.{Cp}", (n, 2, Cp))
        dap = self._buf(f"bn.tdap.{n * cb}", (2 * n * cb,))  # (hi, lo) pair per tile
        f = BnBwdFuse()
        f.x, f.part = x.data_ptr(), part.data_ptr()
        if not bare:
            f.scale, f.shift = self._buf(tag + ".scale", (Cp,)).data_ptr(), self._buf(tag + ".shift", (Cp,)).data_ptr()
            f.mean, f.rstd = self._buf(tag + ".mean", (Cp,)).data_ptr(), self._buf(tag + ".rstd", (Cp,)).data_ptr()
        f.alpha = None if act.weight is None else act.weight.data_ptr()
        f.dalpha_part = dap.data_ptr()
        cv.dgrad(dy, w, dx, accumulate=accumulate, fuse=f)
        return part, n, dap

    def _bn_act_bwd(self, tag, dy, x, bn: BatchNormP, act: PReLUP, rows, dx, acc, fused=None):
        Cp = pad16(bn.c)
        scale, shift = self._buf(tag + ".scale", (Cp,)), self._buf(tag + ".shift", (Cp,))
        mean, rstd = self._buf(tag + ".mean", (Cp,)), self._buf(tag + ".rstd", (Cp,))
        sums = self._buf(tag + ".dsums", (2, Cp))
        if fused is not None:  # the launch that produced dy already summed (du, du * xhat) per tile
            part, nch, dap = fused
        else:
            nch = ops.bn_chunks(rows)
            part = self._buf(f"bn.part.{nch}.{Cp}", (nch, 2, Cp))
            dap = self._buf(f"bn.dap.{nch}.{Cp}", (2 * nch * ((Cp + 63) // 64),))  # (hi, lo) pair per block
            ops.affine_prelu_bwd_partial(dy, x, scale, shift, mean, rstd, act.weight, rows, Cp, Cp, part, dap)
        count = rows
        if self.world_size > 1 and self.sync_bn:
            ops.bn_reduce_partials(part, nch, Cp, sums)
            # parameter grads come from the LOCAL sums (the gradient all-reduce sums them
            # later); the input gradient needs the global ones
            gl = self._buf(tag + ".dsums_g", (2, Cp))
            gl.copy_(sums)
            with self._comm_bracket("bn"):
                self._allreduce(gl)
            ops.affine_prelu_bwd_apply(dy, x, scale, shift, mean, rstd, bn.weight, act.weight, gl, rows * self.world_size,
                                       dx, rows, Cp, Cp, None, None, slope_grad(act), dap, dap.numel(), acc,
                                       colsum_part=self._dx_colsum_part(tag, dx, rows, Cp))
            if acc:
                ops.axpy(1.0, sums[0], bn.bias.grad)
                ops.axpy(1.0, sums[1], bn.weight.grad)
            else:
                bn.bias.grad.copy_(sums[0])
                bn.weight.grad.copy_(sums[1])
        else:
            ops.bn_bwd_reduce(part, nch, Cp, sums, bn.weight.grad, bn.bias.grad, slope_grad(act), dap, dap.numel(), acc)
            ops.affine_prelu_bwd_apply(dy, x, scale, shift, mean, rstd, bn.weight, act.weight, sums, count, dx, rows, Cp, Cp,
                                       None, None, None, dap, dap.numel(), acc, colsum_part=self._dx_colsum_part(tag, dx, rows, Cp))
        return dx

    # ------------------------------------------------------------------ forward pieces
    def normalize_root(self, root):
        a = self.arena_size
        return 2 * (root - a[0]) / (a[1] - a[0]) - 1

    def inv_normalize_root(self, norm_root):
        a = self.arena_size
        return 0.5 * (norm_root + 1) * (a[1] - a[0]) + a[0]

    def _prep(self, t):
        t = t.to(self.device)
        if t.dtype != torch.float32:
            t = t.float()
        return t.contiguous()

    def _encode_trunk(self, data):
        x6d = self._prep(data["x6d"])
        B, W = x6d.shape[0], x6d.shape[1]
        if W != self.window:
            raise ValueError(f"window {W} != model.window {self.window}")
        root = self._prep(data["root"]) if self.arena_size is not None else None
        enc = self.encoder
        Cin_p = pad16(self.in_channels)
        rows = B * W
        x_in = self._buf("x_in", (rows, Cin_p))
        ops.pack_input(x6d, root, self._arena_host, x_in, self.n_keypts)
        ch = self.ch
        c0 = self._buf("enc.c_in", (rows, pad16(ch[0])))
        self._conv("enc.conv_in", enc.conv_in, B, W).fwd(x_in, enc.conv_in.weight, enc.conv_in.bias, c0)
        a = self._buf("enc.a0", (rows, pad16(ch[0])))
        ops.affine_prelu_fwd(c0, None, None, enc.activation.weight, a, rows, pad16(ch[0]), pad16(ch[0]))
        L = W
        for i, blk in enumerate(enc.res_layers):
            t = f"enc.{i}"
            conv0, bn1, act1, conv3 = blk.residual
            cv0 = self._conv(t + ".c0", conv0, B, L)
            Lo = cv0.l_out
            cv3 = self._conv(t + ".c3", conv3, B, Lo)
            s = self._buf(t + ".s", (B * Lo, cv3.c_out_p))
            # the skip branch only needs `a`: it runs on the side stream, concurrently with the
            # residual branch's conv -> BN-stats -> BN+PReLU chain; conv3 then accumulates onto it
            cvs = self._conv(t + ".sk", blk.skip, B, L)
            self._fork(lambda: cvs.fwd(a, blk.skip.weight, blk.skip.bias, s))
            r0 = self._buf(t + ".r0", (B * Lo, cv0.c_out_p))
            st1 = self._conv_fwd_bn(cv0, a, conv0, r0)
            r0a = self._buf(t + ".r0a", (B * Lo, cv0.c_out_p))
            self._bn_act(t + ".bn1", r0, bn1, act1, B * Lo, r0a, st1)
            self._join_side(1)
            st2 = self._conv_fwd_bn(cv3, r0a, conv3, s, accumulate=True)  # statistics of skip + residual: the sum is what it writes
            a2 = self._buf(t + ".a", (B * Lo, cv3.c_out_p))
            self._bn_act(t + ".bn2", s, blk.add[0], blk.add[1], B * Lo, a2, st2)
            a, L = a2, Lo
        if L != enc.latent_len:
            raise ValueError(f"encoder output length {L} != find_latent_dim {enc.latent_len} (reference would fail too)")
        # heads: h = [mu | raw]
        zp = pad16(self.z_dim)
        flat = a.view(B, L * pad16(ch[-1]))
        h = self._buf("enc.h", (B, self._hw), zero=True)
        hd = self._fc_heads  # fc_mu || fc_sigma[0]: one GEMM writes [mu | raw]
        self._lin("fc_heads", hd, B, pieces=self._heads_pieces()).fwd(flat, hd.weight, hd.bias, h)
        return B, flat, h

    def _heads_beta(self, B, h, draw):
        """prior = "beta": alpha, beta, mu (the rescaled mode) and the KL partials from one kernel; the draw x ~ Beta(alpha, beta) is
        `draw` (data["eps"]: injected) or torch's sampler (RNG plumbing, like randn for the Gaussian heads); z = 2 x - 1 in train
        AND eval mode (residual.py:328-331)."""
        zp = pad16(self.z_dim)
        zcp = pad16(self.z_dim + self.conditional_dim)
        mu = self._buf("mu", (B, zp), zero=True)
        alpha, beta = self._buf("alpha", (B, zp), zero=True), self._buf("beta", (B, zp), zero=True)
        zc = self._buf("dec.zc", (B, zcp), zero=True)
        klp = self._buf("kl_part", (ops.heads_blocks(B, self.z_dim),))
        ops.heads_beta_fwd(h, self._hw, alpha, beta, mu, zp, klp, B, self.z_dim, zp)
        a, b = alpha[:, : self.z_dim], beta[:, : self.z_dim]
        if draw is None:
            draw = torch._sample_dirichlet(torch.stack([a, b], -1))[..., 0]
        x = self._buf("beta.x", (B, self.z_dim))
        x.copy_(draw)
        zc[:, : self.z_dim] = x * 2 - 1
        self._L = None
        return mu, (alpha, beta, x), zc, klp

    def _heads(self, B, h, eps):
        if self.prior == "beta":
            return self._heads_beta(B, h, eps)
        zp = pad16(self.z_dim)
        zcp = pad16(self.z_dim + self.conditional_dim)
        mu = self._buf("mu", (B, zp), zero=True)
        sigma = self._buf("sigma", (B, zp), zero=True)
        zc = self._buf("dec.zc", (B, zcp), zero=True)
        klp = self._buf("kl_part", (ops.heads_blocks(B, self.z_dim),))
        if self.is_diag:
            ops.heads_diag_fwd(h, self._hw, eps, mu, sigma, zc, zcp, klp, B, self.z_dim, raw_off=zp, ldm=zp)
            self._L = None
        else:  # full Cholesky factor, materialised densely (it is part of the API: data_o["L"])
            self._L = self._buf("L", (B, self.z_dim, self.z_dim))
            ops.heads_tril_fwd(h, self._hw, eps, mu, zp, self._L, zc, zcp, klp, B, self.z_dim, zp)
        return mu, sigma, zc, klp

    def _L_out(self, sigma):
        return torch.diag_embed(sigma[:, : self.z_dim]) if self.is_diag else self._L

    def _conditional_var(self, data, B):
        parts = []
        for k in self.conditional_keys:
            v = data[k].to(self.device)
            if self.discrete_classes is not None and k in self.discrete_classes.keys():
                parts.append(F.one_hot(v.ravel().long(), len(self.discrete_classes[k])).float())
            else:
                parts.append(v.float())
        return torch.cat(parts, dim=-1)

    def _decode_trunk(self, B, zc):
        dec = self.decoder
        ch = self.ch
        L = dec.latent_len
        Ctop = pad16(ch[-1])
        f = self._buf("dec.f", (B, L * Ctop))
        self._lin("fc_in", dec.fc_in, B).fwd(zc, dec.fc_in.weight, dec.fc_in.bias, f)
        d = f.view(B * L, Ctop)
        for j, blk in enumerate(dec.res_layers):
            t = f"dec.{j}"
            ct1, bn1, act1, ct2 = blk.residual
            cv1 = self._conv(t + ".t1", ct1, B, L)
            cv2 = self._conv(t + ".t2", ct2, B, L)
            Lo = cv2.l_out
            s = self._buf(t + ".s", (B * Lo, cv2.c_out_p))
            sk = blk.skip[1]
            cvs = self._conv(t + ".sk", sk, B, 2 * L, up2=True)
            if cvs.l_out != Lo:
                raise ValueError("skip / residual length mismatch")
            keep_up = self.training and not (cvs.up2 and cvs.up2_wgrad)  # the weight gradient's operand
            up = self._buf(t + ".up", (B * 2 * L, cv1.c_in_p)) if (keep_up or not cvs.up2) else None

            def skip_branch(d=d, up=up, cvs=cvs, sk=sk, s=s, L=L, cin=cv1.c_in_p):
                if cvs.up2:  # Upsample(x2, linear) folded into the conv's operand staging; `up` is a by-product (training only)
                    cvs.fwd(d, sk.weight, sk.bias, s, up_out=up)
                else:
                    ops.upsample2_fwd(d, up, B, L, cin, cin)
                    cvs.fwd(up, sk.weight, sk.bias, s)

            self._fork(skip_branch)  # upsample + skip conv on the side stream
            t0 = self._buf(t + ".t0", (B * L, cv1.c_out_p))
            st1 = self._conv_fwd_bn(cv1, d, ct1, t0)
            t0a = self._buf(t + ".t0a", (B * L, cv1.c_out_p))
            self._bn_act(t + ".bn1", t0, bn1, act1, B * L, t0a, st1)
            self._join_side(1)
            st2 = self._conv_fwd_bn(cv2, t0a, ct2, s, accumulate=True)
            d2 = self._buf(t + ".a", (B * Lo, cv2.c_out_p))
            self._bn_act(t + ".bn2", s, blk.add[0], blk.add[1], B * Lo, d2, st2)
            d, L = d2, Lo
        cvo = self._conv("dec.out", dec.conv_out, B, L)
        if cvo.l_out != self.window:
            raise ValueError(f"decoder output length {cvo.l_out} != window {self.window}")
        y = self._buf("dec.y", (B * self.window, cvo.c_out_p))
        cvo.fwd(d, dec.conv_out.weight, dec.conv_out.bias, y)
        return y

    def _tail_prealloc(self, B, data, with_grad):
        """Materialises, on the CURRENT stream, every workspace _run_tail may allocate: a first-use torch.zeros / torch.empty
        inside a forked tail would be queued on torch's current stream while the kernel that reads it runs on the side stream
        (only the C-ABI launches follow the fork)."""
        rows, J = B * self.window, self.n_keypts
        self._buf("out.x6d", (B, self.window, J, 6))
        if self.arena_size is not None:
            self._buf("out.root", (B, self.window, 3))
        self._buf("tail.part", (ops.tail_blocks(rows), 2))
        if "offsets" not in data or "target_pose" not in data:
            self._buf("zero.j3", (rows, J, 3), zero=True)
        if with_grad:
            self._buf("dec.dy", (rows, pad16(self.in_channels)))

    def _run_tail(self, B, data, jpe_scale, root_scale, ext_dx6d, with_grad):
        """Fused tanh/unpack/FK/loss tail.  Without offsets/target in `data` (pure forward)
        zero stand-ins are used and the loss partials are ignored."""
        rows = B * self.window
        J = self.n_keypts
        y = self._buf("dec.y", (rows, pad16(self.in_channels)))
        x6d_hat = self._buf("out.x6d", (B, self.window, J, 6))
        root_hat = self._buf("out.root", (B, self.window, 3)) if self.arena_size is not None else None
        lp = self._buf("tail.part", (ops.tail_blocks(rows), 2))
        offsets = self._prep(data["offsets"]) if "offsets" in data else self._buf("zero.j3", (rows, J, 3), zero=True)
        target = self._prep(data["target_pose"]) if "target_pose" in data else self._buf("zero.j3", (rows, J, 3), zero=True)
        root = self._prep(data["root"]) if self.arena_size is not None else None
        dy = self._buf("dec.dy", (rows, pad16(self.in_channels))) if with_grad else None
        ops.pose_tail(y, y.shape[1], offsets, target, root, self._arena_host, self._tree, jpe_scale, root_scale,
                      ext_dx6d, None, x6d_hat, root_hat, lp, dy, rows, pre_tanh=True)
        return x6d_hat, root_hat, lp, dy

    # ------------------------------------------------------------------ reference API
    def encode(self, data):
        """ResVAE.encode (residual.py:438-459): returns {"mu": [B,z], "L": [B,z,z]}."""
        self._new_pass()
        self._ov = self._overlap_level(data["x6d"].shape[0])
        B, flat, h = self._encode_trunk(data)
        if self.prior == "beta":  # residual.py:449-456: alpha, beta and the rescaled mode (encode draws nothing)
            zp = pad16(self.z_dim)
            mu = self._buf("mu", (B, zp), zero=True)
            alpha, beta = self._buf("alpha", (B, zp), zero=True), self._buf("beta", (B, zp), zero=True)
            ops.heads_beta_fwd(h, self._hw, alpha, beta, mu, zp, self._buf("kl_part", (ops.heads_blocks(B, self.z_dim),)), B, self.z_dim, zp)
            self._state = dict(B=B, flat=flat, h=h, eps=None)
            return {"alpha": alpha[:, : self.z_dim], "beta": beta[:, : self.z_dim], "mu": mu[:, : self.z_dim]}
        mu, sigma, zc, klp = self._heads(B, h, None)
        self._state = dict(B=B, flat=flat, h=h, eps=None)
        return {"mu": mu[:, : self.z_dim], "L": self._L_out(sigma)}

    def decode(self, z, data):
        """ResVAE.decode (residual.py:461-491)."""
        self._new_pass()
        B = z.shape[0]
        self._ov = self._overlap_level(B)
        zcp = pad16(self.z_dim + self.conditional_dim)
        zc = self._buf("dec.zc", (B, zcp), zero=True)
        zc[:, : self.z_dim] = z.to(self.device)
        data_o = {}
        if self.conditional_dim > 0:
            data_o["var"] = self._conditional_var(data, B)
            zc[:, self.z_dim: self.z_dim + self.conditional_dim] = data_o["var"]
        self._decode_trunk(B, zc)
        x6d_hat, root_hat, _, _ = self._run_tail(B, data, 0.0, 0.0, None, False)
        if root_hat is not None:
            data_o["root"] = root_hat
        data_o["x6d"] = x6d_hat
        return data_o

    def sampling_noise(self, B):
        """eps ~ N(0, I) for the reparameterisation (residual.py:315).  Override or pass
        data["eps"] to inject noise (parity tests)."""
        return torch.randn(B, self.z_dim, device=self.device)

    def forward(self, data):
        """VAE.forward (residual.py:318-362).  Returns data_o with mu, L, z, x6d, root, var,
        disentangle[method][feature]."""
        self._new_pass()
        self._ov = self._overlap_level(data["x6d"].shape[0])
        B, flat, h = self._encode_trunk(data)
        eps = None
        if self.prior == "beta":  # data["eps"] = an injected draw x in (0, 1); else torch's sampler inside _heads_beta
            eps = self._prep(data["eps"]) if "eps" in data else None
        elif self.training:
            eps = self._prep(data["eps"]) if "eps" in data else self.sampling_noise(B)
        mu, sigma, zc, klp = self._heads(B, h, eps)
        if self.prior == "beta":
            alpha, beta, eps = sigma  # (eps = the draw: what the backward differentiates implicitly)
            data_o = {"alpha": alpha[:, : self.z_dim], "beta": beta[:, : self.z_dim], "mu": mu[:, : self.z_dim]}
            data_o["beta_dist"] = torch.distributions.Beta(data_o["alpha"], data_o["beta"])
        else:
            data_o = {"mu": mu[:, : self.z_dim], "L": self._L_out(sigma)}
        data_o["z"] = zc[:, : self.z_dim]
        if self.conditional_dim > 0:
            data_o["var"] = self._conditional_var(data, B)
            zc[:, self.z_dim: self.z_dim + self.conditional_dim] = data_o["var"]
        self._decode_trunk(B, zc)
        if self.training and self.defer_tail:
            # fast path: get_batch_loss runs the fused tail once (outputs + losses + seed
            # gradients); data_o["x6d"/"root"] alias the buffers it fills
            x6d_hat = self._buf("out.x6d", (B, self.window, self.n_keypts, 6))
            root_hat = self._buf("out.root", (B, self.window, 3)) if self.arena_size is not None else None
            self._tail_done = False
        else:
            x6d_hat, root_hat, _, _ = self._run_tail(B, data, 0.0, 0.0, None, False)
            self._tail_done = True
        data_o["x6d"] = x6d_hat
        if root_hat is not None:
            data_o["root"] = root_hat
        # scrubber heads (residual.py:337-360): they take mu, or -- when the `linear` method is configured -- the feature's
        # null-space projection z_null of mu
        data_o["disentangle"] = {}
        lin = None
        if "linear" in self.disentangle:
            need = self.training and torch.is_grad_enabled()
            mu_leaf = mu[:, : self.z_dim].detach().clone().requires_grad_(need)
            with torch.set_grad_enabled(need):
                outs = {k: m(mu_leaf) for k, m in self.disentangle["linear"].items()}
            data_o["disentangle"]["linear"] = outs
            lin = dict(mu=mu_leaf, out=outs, leaves={k: [mu_leaf] + m.leaves() for k, m in self.disentangle["linear"].items()})

        def latent(k):
            """[B, zp] buffer the HIP heads of feature k read"""
            if lin is None:
                return mu
            x = self._buf(f"lin.{k}.zn", (B, pad16(self.z_dim)), zero=True)
            x[:, : self.z_dim] = lin["out"][k]["z_null"].detach()  # KeyError for a feature without projection, as in the reference
            return x

        for method, module_dict in self.disentangle.items():
            if method == "linear":
                continue
            data_o["disentangle"][method] = {}
            for k, m in module_dict.items():
                if method == "grad_reversal":
                    r = self._runner(method, k, m.ensemble, B)
                    outs = r.forward(latent(k), self.z_dim) if r.fused else r.forward(latent(k))
                    data_o["disentangle"][method][k] = [o[:, : m.ensemble.out_dim] for o in outs]
                elif method == "adversarial_net":
                    # the un-shuffled evaluation on (mu, var) that VAE.forward stores (residual.py:357-358); the loss ignores it and
                    # recomputes the ensemble with the shuffle.  On the training fast path (defer_tail: data_o aliases per-pass buffers
                    # anyway) it is evaluated on first access, so a step that never reads it does not launch it
                    def unshuffled(k=k, m=m, var=data_o["var"]):
                        r = self._runner("adversarial_net.fwd", k, m.ensemble, B)
                        if r.fused:
                            outs = r.forward(latent(k), self.z_dim, src1=self._var32(var))
                        else:
                            x = self._buf(f"an.{k}.x0", (B, pad16(m.ensemble.in_dim)), zero=True)
                            x[:, : self.z_dim] = latent(k)[:, : self.z_dim]
                            x[:, self.z_dim: self.z_dim + self.conditional_dim] = var
                            outs = r.forward(x)
                        return [torch.softmax(o[:, :2], -1) for o in outs]
                    data_o["disentangle"][method][k] = _LazyList(unshuffled) if (self.training and self.defer_tail) else unshuffled()
                elif method == "moving_avg_lsq":
                    data_o["disentangle"][method][k] = m(latent(k)[:, : self.z_dim])
                elif method in ("moving_avg", "qda"):  # stateful filters: forward() is a no-op (disentangle.py:31-32,127-128)
                    data_o["disentangle"][method][k] = m(latent(k)[:, : self.z_dim])
                else:
                    raise NotImplementedError(f"scrubber '{method}' is outside this build's scope (SURVEY 8a row A2)")
        self._state = dict(B=B, flat=flat, h=h, eps=eps, mu=mu, sigma=sigma, zc=zc, klp=klp, data=data, lin=lin)
        return data_o

    def _var32(self, var):
        """conditional variables as the contiguous fp32 [B, D] array the ensemble kernels read"""
        return var if (var.dtype == torch.float32 and var.is_contiguous()) else var.float().contiguous()

    def _runner(self, method, key, ens: MLPEnsemble, batch, halves=1):
        """Runner of one ensemble for `batch` samples (halves = 2: the adversarial net's doubled batch).  The fused kernels
        (csrc/ensemble.hip) take every ensemble whose activations fit a workgroup's LDS; wider ones (z_dim >~ 64) run Linear
        by Linear on the GEMM kernels."""
        k = (method, key, batch, halves)
        r = self._runners.get(k)
        if r is None:
            r = None
            if ops.FUSED_ENSEMBLE:
                cand = FusedEnsembleRunner(ens, batch, self.device, halves)
                if cand.fits():
                    r = cand
            if r is None:
                r = EnsembleRunner(ens, batch * halves, self.device)
            self._runners[k] = r
        return r

    # ------------------------------------------------------------------ backward schedule
    def make_total(self, total):
        """Wrap the device scalar `total` so that ``total.backward()`` runs the HIP backward."""
        anchor = self.__dict__.get("_anchor")
        if anchor is None:  # a leaf that makes the returned scalar require grad; the same one every step (no per-step fill launch)
            anchor = self.__dict__["_anchor"] = self.flat_params.new_zeros((), requires_grad=True)
        return _TotalLoss.apply(anchor, total, self)

    def _scrub_backward(self, pend):
        """Scrubber heads: MLP backward, seeds into d_mu.  Reads only what the loss section produced (nothing of the decoder's
        backward), so train.losses runs it early -- beside the fused tail on its side stream -- on the fast path (defer_tail);
        otherwise it is the first thing backward_from_seeds does."""
        B = self._state["B"]
        acc = pend.get("accumulate", False)
        zp = pad16(self.z_dim)
        d_mu = pend["d_mu"]
        # `linear` projections: gradients of their decoders come from the small torch graphs (train.losses collected the
        # loss-side part; the gradient-reversal heads add theirs below)
        if "linear" in self.disentangle and not acc:
            for m in self.disentangle["linear"].values():
                for p in m.leaves():
                    p.grad.zero_()
        for p, g in pend.get("lin_pgrads", {}).values():
            p.grad.add_(g)
        for item in pend["scrub"]:
            runner, d_outs, kind = item["runner"], item["d_outs"], item["kind"]
            if runner.fused:  # one launch + a fixed-order reduction; the latent seed is updated by the kernel
                if kind == "gr" and item.get("lin") is not None:  # head input was z_null(mu, W): chain through the projection
                    g_in = runner.backward(None, 0.0, param_grads=True, accumulate=acc, want_raw=True)
                    z_null, leaves = item["lin"]
                    gs = torch.autograd.grad(z_null, leaves, grad_outputs=-item["alpha"] * g_in[:, : self.z_dim], retain_graph=True)
                    d_mu[:, : self.z_dim] += gs[0]
                    for p, g in zip(leaves[1:], gs[1:]):
                        p.grad.add_(g)
                elif kind == "gr":
                    runner.backward(d_mu, -item["alpha"], param_grads=True, accumulate=acc)  # gradient reversal: -alpha * grad
                else:  # adversarial net on cat([mu;mu],[v;v_shuffle]): both halves feed mu; its parameters are frozen
                    runner.backward(d_mu, 1.0, param_grads=False)
                continue
            g_in = runner.backward(d_outs, param_grads=(kind == "gr"), accumulate=acc)
            if kind == "gr" and item.get("lin") is not None:  # head input was z_null(mu, W): chain through the projection
                z_null, leaves = item["lin"]
                gs = torch.autograd.grad(z_null, leaves, grad_outputs=-item["alpha"] * g_in[:, : self.z_dim], retain_graph=True)
                d_mu[:, : self.z_dim] += gs[0]
                for p, g in zip(leaves[1:], gs[1:]):
                    p.grad.add_(g)
            elif kind == "gr":
                ops.axpy(-item["alpha"], g_in, d_mu)  # gradient reversal: -alpha * grad
            else:  # adversarial net on cat([mu;mu],[v;v_shuffle]): both halves feed mu
                tmp = self._buf("an.gmu", (B, zp), zero=True)
                tmp[:, : self.z_dim] = g_in[:B, : self.z_dim] + g_in[B:, : self.z_dim]
                ops.axpy(1.0, tmp, d_mu)
        pend["scrub_done"] = True

    def backward_from_seeds(self):
        """Reverse schedule.  Needs the seeds prepared by train.losses.get_batch_loss:
        pending = {dy [rows,Cp], kl_scale, d_mu [B,zp] or None, scrub: [...]}"""
        pend = self._pending
        if pend is None:
            raise RuntimeError("backward called without a preceding get_batch_loss on this model")
        st = self._state
        B = st["B"]
        self.__dict__["_main"] = None  # the autograd engine thread has its own notion of the current stream
        self._assign_grad_views()
        self._dx_colsum = {}
        acc = pend.get("accumulate", False)
        enc, dec, ch = self.encoder, self.decoder, self.ch
        zp = pad16(self.z_dim)
        d_mu = pend["d_mu"]  # [B, zp] zero-initialised seed for scrubber grads
        if not pend.get("scrub_done"):
            self._scrub_backward(pend)
        # ---- decoder
        W = self.window
        rows = B * W
        dy = pend["dy"]
        lens = [dec.latent_len]
        for _ in dec.res_layers:
            lens.append((lens[-1] - 1) * 2 - 2 * (self.kernel // 2) + (self.kernel - 1) + 1)
        d_in_last = self._buf(f"dec.{len(dec.res_layers) - 1}.a", (B * lens[-1], pad16(ch[0])))
        cvo = self._conv("dec.out", dec.conv_out, B, lens[-1])
        self._wgrad(cvo, d_in_last, dy, dec.conv_out, acc)
        g = self._buf("g.dec.top", (B * lens[-1], pad16(ch[0])))
        nd = len(dec.res_layers)
        # fz[tag]: first-pass sums of BatchNorm stage `tag`'s backward that came out of the epilogue of the data-gradient launch
        # which produced its incoming gradient (absent: _bn_act_bwd runs its own pass over dy and x)
        fz = {}
        fz[f"dec.{nd - 1}.bn2"] = self._dgrad_into_bn(cvo, dy, dec.conv_out.weight, g, False, f"dec.{nd - 1}.bn2",
                                                       self._buf(f"dec.{nd - 1}.s", (B * lens[-1], pad16(ch[0]))), dec.res_layers[nd - 1].add[1])
        for j in range(len(dec.res_layers) - 1, -1, -1):
            blk = dec.res_layers[j]
            t = f"dec.{j}"
            L, Lo = lens[j], lens[j + 1]
            ct1, bn1, act1, ct2 = blk.residual
            sk = blk.skip[1]
            cv1 = self._conv(t + ".t1", ct1, B, L)
            cv2 = self._conv(t + ".t2", ct2, B, L)
            cvs = self._conv(t + ".sk", sk, B, 2 * L, up2=True)
            d_in = self._buf(f"dec.{j - 1}.a", (B * L, cv1.c_in_p)) if j > 0 else self._buf("dec.f", (B, L * cv1.c_in_p)).view(B * L, cv1.c_in_p)
            s = self._buf(t + ".s", (B * Lo, cv2.c_out_p))
            g_s = self._buf("g." + t + ".s", (B * Lo, cv2.c_out_p))
            self._bn_act_bwd(t + ".bn2", g, s, blk.add[0], blk.add[1], B * Lo, g_s, acc, fz.get(t + ".bn2"))
            t0a = self._buf(t + ".t0a", (B * L, cv1.c_out_p))
            self._wgrad(cvs, d_in if (cvs.up2 and cvs.up2_wgrad) else self._buf(t + ".up", (B * 2 * L, cv1.c_in_p)), g_s, sk, acc)
            self._wgrad(cv2, t0a, g_s, ct2, acc)
            g_up = self._buf("g." + t + ".up", (B * 2 * L, cv1.c_in_p))
            g_d = self._buf("g." + t + ".in", (B * L, cv1.c_in_p))

            def skip_bwd(cvs=cvs, sk=sk, g_s=g_s, g_up=g_up, g_d=g_d, L=L, cin=cv1.c_in_p):
                cvs.dgrad(g_s, sk.weight, g_up)
                ops.upsample2_bwd(g_up, g_d, B, L, cin, cin)

            self._fork(skip_bwd, k=1)  # skip branch concurrently with the residual branch below
            g_t0a = self._buf("g." + t + ".t0a", (B * L, cv1.c_out_p))
            t0 = self._buf(t + ".t0", (B * L, cv1.c_out_p))
            f1 = self._dgrad_into_bn(cv2, g_s, ct2.weight, g_t0a, False, t + ".bn1", t0, act1)
            g_t0 = self._buf("g." + t + ".t0", (B * L, cv1.c_out_p))
            self._bn_act_bwd(t + ".bn1", g_t0a, t0, bn1, act1, B * L, g_t0, acc, f1)
            self._wgrad(cv1, d_in, g_t0, ct1, acc)
            self._join_side(1)
            if j > 0:  # g_d is the gradient behind the previous block's closing BatchNorm + activation
                pt = f"dec.{j - 1}"
                fz[pt + ".bn2"] = self._dgrad_into_bn(cv1, g_t0, ct1.weight, g_d, True, pt + ".bn2", self._buf(pt + ".s", (B * L, cv1.c_in_p)),
                                                      dec.res_layers[j - 1].add[1])
            else:
                cv1.dgrad(g_t0, ct1.weight, g_d, accumulate=True)
            g = g_d
        # ---- fc_in
        zc = st["zc"]
        zcp = zc.shape[1]
        lin = self._lin("fc_in", dec.fc_in, B)
        g_f = g.view(B, -1)
        self._wgrad(lin, zc, g_f, dec.fc_in, acc)
        g_zc = self._buf("g.zc", (B, zcp))
        lin.dgrad(g_f, dec.fc_in.weight, g_zc)
        # ---- data-parallel bucket 1: the decoder's gradients are final here; their all-reduce
        # overlaps the encoder's reverse schedule (xGMI: few large transfers, not many small ones)
        works = []
        if self.world_size > 1:
            lo, hi = self._dec_span
            works.append(self._bucket_allreduce(lo, hi, acc))
        # ---- heads: dh = [dmu | draw]
        h = st["h"]
        dh = self._buf("g.h", (B, self._hw), zero=True)
        if self.prior == "beta":
            alpha, beta, _ = st["sigma"]
            ops.heads_beta_bwd(h, self._hw, st["eps"], alpha, beta, zp, g_zc, zcp, d_mu, pend["kl_scale"], dh, B, self.z_dim, zp)
        elif self.is_diag:
            ops.heads_diag_bwd(h, self._hw, st["eps"], st["sigma"], g_zc, zcp, d_mu, pend.get("dsigma"), pend["kl_scale"], dh, B,
                               self.z_dim, raw_off=zp, ldm=zp)
        else:
            ops.heads_tril_bwd(h, self._hw, st["eps"], self._L, g_zc, zcp, d_mu, zp, pend["kl_scale"], pend.get("dlv"), dh, B,
                               self.z_dim, zp)
        flat = st["flat"]
        hd = self._fc_heads  # fc_mu || fc_sigma[0] as one Linear: one weight-gradient and one data-gradient launch
        lh = self._lin("fc_heads", hd, B, pieces=self._heads_pieces())
        self._wgrad(lh, flat, dh, hd, acc)
        g_flat = self._buf("g.flat", tuple(flat.shape))
        lh.dgrad(dh, hd.weight, g_flat)
        # ---- encoder blocks
        elens = [W]
        for blk in enc.res_layers:
            elens.append((elens[-1] + 2 * (self.kernel // 2) - (self.kernel - 1) - 1) // 2 + 1)
        g = g_flat.view(B * elens[-1], pad16(ch[-1]))
        enc_cut = self._dec_span[0]  # encoder parameters (blocks, then the heads) end where the decoder's start
        for i in range(len(enc.res_layers) - 1, -1, -1):
            blk = enc.res_layers[i]
            t = f"enc.{i}"
            L, Lo = elens[i], elens[i + 1]
            conv0, bn1, act1, conv3 = blk.residual
            cv0 = self._conv(t + ".c0", conv0, B, L)
            cv3 = self._conv(t + ".c3", conv3, B, Lo)
            cvs = self._conv(t + ".sk", blk.skip, B, L)
            a_in = self._buf(f"enc.{i - 1}.a", (B * L, cv0.c_in_p)) if i > 0 else self._buf("enc.a0", (B * L, cv0.c_in_p))
            s = self._buf(t + ".s", (B * Lo, cv3.c_out_p))
            g_s = self._buf("g." + t + ".s", (B * Lo, cv3.c_out_p))
            self._bn_act_bwd(t + ".bn2", g, s, blk.add[0], blk.add[1], B * Lo, g_s, acc, fz.get(t + ".bn2"))
            r0a = self._buf(t + ".r0a", (B * Lo, cv0.c_out_p))
            self._wgrad(cvs, a_in, g_s, blk.skip, acc)
            self._wgrad(cv3, r0a, g_s, conv3, acc)
            if self.world_size > 1 and (enc_cut - self._enc_mid_cuts[i]) * 4 >= self.bucket_min_bytes:
                # data-parallel: second conv, skip conv and the closing BatchNorm / PReLU of this block are final
                works.append(self._bucket_allreduce(self._enc_mid_cuts[i], enc_cut, acc))
                enc_cut = self._enc_mid_cuts[i]
            g_a = self._buf("g." + t + ".in", (B * L, cv0.c_in_p))
            self._fork(lambda cvs=cvs, g_s=g_s, g_a=g_a, w=blk.skip.weight: cvs.dgrad(g_s, w, g_a), k=1)
            g_r0a = self._buf("g." + t + ".r0a", (B * Lo, cv0.c_out_p))
            r0 = self._buf(t + ".r0", (B * Lo, cv0.c_out_p))
            f1 = self._dgrad_into_bn(cv3, g_s, conv3.weight, g_r0a, False, t + ".bn1", r0, act1)
            g_r0 = self._buf("g." + t + ".r0", (B * Lo, cv0.c_out_p))
            self._bn_act_bwd(t + ".bn1", g_r0a, r0, bn1, act1, B * Lo, g_r0, acc, f1)
            self._wgrad(cv0, a_in, g_r0, conv0, acc)
            self._join_side(1)
            if i > 0:  # g_a is the gradient behind the previous block's closing BatchNorm + activation
                pt = f"enc.{i - 1}"
                fz[pt + ".bn2"] = self._dgrad_into_bn(cv0, g_r0, conv0.weight, g_a, True, pt + ".bn2", self._buf(pt + ".s", (B * L, cv0.c_in_p)),
                                                      enc.res_layers[i - 1].add[1])
            else:      # ... behind the bare activation after conv_in
                fz["enc.in"] = self._dgrad_into_bn(cv0, g_r0, conv0.weight, g_a, True, "enc.in", self._buf("enc.c_in", (B * L, cv0.c_in_p)),
                                                   enc.activation, bare=True)
            g = g_a
            if self.world_size > 1 and (enc_cut - self._enc_cuts[i]) * 4 >= self.bucket_min_bytes:
                # data-parallel: this block (and the heads / blocks above it) is final -> next bucket
                works.append(self._bucket_allreduce(self._enc_cuts[i], enc_cut, acc))
                enc_cut = self._enc_cuts[i]
        # ---- conv_in (bare PReLU in front)
        C0 = pad16(ch[0])
        c0 = self._buf("enc.c_in", (rows, C0))
        if fz.get("enc.in") is not None:  # the slope's partials came out of the last data-gradient launch
            dap = fz["enc.in"][2]
        else:
            nch = ops.bn_chunks(rows)
            part = self._buf(f"bn.part.{nch}.{C0}", (nch, 2, C0))
            dap = self._buf(f"bn.dap.{nch}.{C0}", (2 * nch * ((C0 + 63) // 64),))
            ops.affine_prelu_bwd_partial(g, c0, None, None, None, None, enc.activation.weight, rows, C0, C0, part, dap)
        g_c0 = self._buf("g.enc.c_in", (rows, C0))
        ops.affine_prelu_bwd_apply(g, c0, None, None, None, None, None, enc.activation.weight, None, 1.0, g_c0, rows, C0, C0,
                                   None, None, slope_grad(enc.activation), dap, dap.numel(), acc,
                                   colsum_part=self._dx_colsum_part("enc.in", g_c0, rows, C0))
        x_in = self._buf("x_in", (rows, pad16(self.in_channels)))
        cvi = self._conv("enc.conv_in", enc.conv_in, B, W)
        self._wgrad(cvi, x_in, g_c0, enc.conv_in, acc)
        self._join_side()
        self._db_batch.flush(self._colsum_ws, accumulate=acc)
        # ---- data-parallel: sum gradients over ranks (losses are normalised by the GLOBAL batch)
        if self.world_size > 1:  # what is left: conv_in + the blocks below the last bucket, and everything after the decoder
            hi = self._dec_span[1]
            with self._comm_bracket("grads"):
                if enc_cut > 0:
                    self._allreduce(self.flat_grads[:enc_cut])
                if hi < self.flat_grads.numel():
                    self._allreduce(self.flat_grads[hi:])
                for w in works:
                    if w is not None:
                        w.wait()
        self._pending = None

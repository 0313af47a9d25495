"""Parameter-holding leaf modules of the MI355X SC-VAE.

Same module tree and ``state_dict`` key names as the reference
(src/scrubvae/model/residual.py:71-292, disentangle.py:583-684) so that reference
checkpoints load and checkpoints written here load in the reference.
Internally every conv / linear weight lives in the HIP library's compute layout
("TIO": [tap][c_in_pad][c_out_pad], channels padded to multiples of 16, pads zero):
``state_dict()`` / ``load_state_dict()`` convert at the boundary, nothing is repacked per
step.  All parameters of a model are views into ONE flat fp32 buffer (and their ``.grad``s
into one flat gradient buffer): the fused Adam kernel and the RCCL gradient all-reduce then
work on single contiguous arrays.
"""
from __future__ import annotations

import math

import torch
import torch.nn as nn

from ..ops import pad16, conv_weight_to_tio, conv_weight_from_tio


class Leaf(nn.Module):
    """A module whose parameters are declared first (``specs``) and materialised later as
    views of the model's flat parameter buffer (``ResVAE._materialise``)."""

    def __init__(self):
        super().__init__()
        self.specs = {}  # name -> shape (compute layout)
        self._register_state_dict_hook(Leaf._export_hook)
        self._register_load_state_dict_pre_hook(self._import_hook)

    def declare(self, name, shape):
        self.specs[name] = tuple(int(s) for s in shape)

    # -- layout conversion, overridden by subclasses
    def export_tensor(self, name, t):
        return t

    def import_tensor(self, name, t):
        return t

    @staticmethod
    def _export_hook(module, state_dict, prefix, local_metadata):
        for name in list(module._parameters) + list(module._buffers):
            key = prefix + name
            if key in state_dict:
                state_dict[key] = module.export_tensor(name, state_dict[key])

    def _import_hook(self, state_dict, prefix, local_metadata, strict, missing_keys, unexpected_keys, error_msgs):
        for name in list(self._parameters) + list(self._buffers):
            key = prefix + name
            if key in state_dict:
                cur = self._parameters.get(name, None)
                if cur is None:
                    cur = self._buffers.get(name)
                t = state_dict[key]
                # state dicts are ALWAYS in the reference layout (state_dict() exports to it)
                if cur is not None:
                    state_dict[key] = self.import_tensor(name, t.to(cur.device))

    def reset_parameters(self):
        pass


class ConvP(Leaf):
    """nn.Conv1d / nn.ConvTranspose1d parameters (reference layouts [Cout,Cin,k] / [Cin,Cout,k])."""

    def __init__(self, c_in, c_out, kernel, stride=1, padding=0, dilation=1, transposed=False):
        super().__init__()
        self.c_in, self.c_out, self.kernel = c_in, c_out, kernel
        self.stride, self.padding, self.dilation, self.transposed = stride, padding, dilation, transposed
        self.declare("weight", (kernel, pad16(c_in), pad16(c_out)))
        self.declare("bias", (pad16(c_out),))

    def export_tensor(self, name, t):
        if name == "weight":
            return conv_weight_from_tio(t, self.c_in, self.c_out, self.transposed)
        return t[: self.c_out].clone()

    def import_tensor(self, name, t):
        if name == "weight":
            return conv_weight_to_tio(t.float(), self.transposed)
        out = torch.zeros(pad16(self.c_out), dtype=torch.float32, device=t.device)
        out[: self.c_out] = t
        return out

    def reset_parameters(self):
        # nn.Conv1d/ConvTranspose1d default init (kaiming_uniform(a=sqrt(5)), bias U(+-1/sqrt(fan_in)))
        shape = (self.c_in, self.c_out, self.kernel) if self.transposed else (self.c_out, self.c_in, self.kernel)
        w = torch.empty(shape)
        nn.init.kaiming_uniform_(w, a=math.sqrt(5))
        fan_in = shape[1] * self.kernel
        b = torch.empty(self.c_out).uniform_(-1 / math.sqrt(fan_in), 1 / math.sqrt(fan_in))
        with torch.no_grad():
            self.weight.copy_(self.import_tensor("weight", w).to(self.weight.device))
            self.bias.copy_(self.import_tensor("bias", b).to(self.bias.device))

    def extra_repr(self):
        return f"{self.c_in}, {self.c_out}, kernel={self.kernel}, stride={self.stride}, padding={self.padding}, transposed={self.transposed}"


class DenseW(Leaf):
    """nn.Linear parameters kept in the reference layout ([out, in], optional bias): used by heads that run on torch
    device ops (LinearProjection) but still live in the flat parameter / gradient buffers."""

    def __init__(self, in_f, out_f, bias=False):
        super().__init__()
        self.in_f, self.out_f, self.has_bias = in_f, out_f, bias
        self.declare("weight", (out_f, in_f))
        if bias:
            self.declare("bias", (out_f,))

    def reset_parameters(self):
        w = torch.empty(self.out_f, self.in_f)
        nn.init.kaiming_uniform_(w, a=math.sqrt(5))
        with torch.no_grad():
            self.weight.copy_(w.to(self.weight.device))
            if self.has_bias:
                self.bias.copy_(torch.empty(self.out_f).uniform_(-1 / math.sqrt(self.in_f), 1 / math.sqrt(self.in_f)).to(self.bias.device))


def _index_map(n_ref, index):
    """lib->ref index map padded to a multiple of 16 with -1."""
    if index is None:
        index = torch.arange(n_ref)
    n_lib = pad16(len(index))
    out = torch.full((n_lib,), -1, dtype=torch.long)
    out[: len(index)] = index
    return out


class LinearP(Leaf):
    """nn.Linear parameters.  in_index / out_index map each library feature (channels-last
    order, possibly with padded channels) to the reference feature it corresponds to
    (-1 = structural pad): the reference flattens [B,C,L] as c*L+l (residual.py:213,265),
    the library keeps [B,L,Cp] as l*Cp+c."""

    def __init__(self, in_f, out_f, in_index=None, out_index=None):
        super().__init__()
        self.in_f, self.out_f = in_f, out_f
        self.in_index = _index_map(in_f, in_index)
        self.out_index = _index_map(out_f, out_index)
        self.in_lib, self.out_lib = len(self.in_index), len(self.out_index)
        self.declare("weight", (1, self.in_lib, self.out_lib))
        self.declare("bias", (self.out_lib,))

    def export_tensor(self, name, t):
        dev = t.device
        ii, oi = self.in_index.to(dev), self.out_index.to(dev)
        vi, vo = (ii >= 0).nonzero().squeeze(1), (oi >= 0).nonzero().squeeze(1)
        if name == "weight":
            w = torch.zeros(self.out_f, self.in_f, dtype=t.dtype, device=dev)
            w[oi[vo][:, None], ii[vi][None, :]] = t[0][vi][:, vo].t()
            return w
        b = torch.zeros(self.out_f, dtype=t.dtype, device=dev)
        b[oi[vo]] = t[vo]
        return b

    def import_tensor(self, name, t):
        dev = t.device
        ii, oi = self.in_index.to(dev), self.out_index.to(dev)
        vi, vo = (ii >= 0).nonzero().squeeze(1), (oi >= 0).nonzero().squeeze(1)
        t = t.float()
        if name == "weight":
            w = torch.zeros(1, self.in_lib, self.out_lib, dtype=torch.float32, device=dev)
            w[0, vi[:, None], vo[None, :]] = t[oi[vo]][:, ii[vi]].t()
            return w
        b = torch.zeros(self.out_lib, dtype=torch.float32, device=dev)
        b[vo] = t[oi[vo]]
        return b

    def reset_parameters(self):
        w = torch.empty(self.out_f, self.in_f)
        nn.init.kaiming_uniform_(w, a=math.sqrt(5))
        b = torch.empty(self.out_f).uniform_(-1 / math.sqrt(self.in_f), 1 / math.sqrt(self.in_f))
        with torch.no_grad():
            self.weight.copy_(self.import_tensor("weight", w).to(self.weight.device))
            self.bias.copy_(self.import_tensor("bias", b).to(self.bias.device))

    def extra_repr(self):
        return f"in_features={self.in_f}, out_features={self.out_f}"


class BatchNormP(Leaf):
    """nn.BatchNorm1d(C, eps=1e-4) parameters and running statistics (residual.py:88,112,146,173)."""

    def __init__(self, c, eps=1e-4, momentum=0.1):
        super().__init__()
        self.c, self.eps, self.momentum = c, eps, momentum
        self.declare("weight", (pad16(c),))
        self.declare("bias", (pad16(c),))
        self.register_buffer("running_mean", torch.zeros(pad16(c)))
        self.register_buffer("running_var", torch.ones(pad16(c)))
        self.register_buffer("num_batches_tracked", torch.zeros((), dtype=torch.long))

    def export_tensor(self, name, t):
        return t if name == "num_batches_tracked" else t[: self.c].clone()

    def import_tensor(self, name, t):
        if name == "num_batches_tracked":
            return t
        out = torch.zeros(pad16(self.c), dtype=torch.float32, device=t.device)
        if name == "running_var":
            out.fill_(1.0)
        out[: self.c] = t
        return out

    def reset_parameters(self):
        with torch.no_grad():
            self.weight.zero_()
            self.weight[: self.c] = 1.0
            self.bias.zero_()
            self.running_mean.zero_()
            self.running_var.fill_(1.0)
            self.num_batches_tracked.zero_()


class TanhP(nn.Module):
    """nn.Tanh() where the default model has a PReLU (model.activation == "tanh", residual.py:89,113,147,174,199): no
    parameters; the activation kernels get a NULL slope pointer."""

    weight = None

    def extra_repr(self):
        return "tanh"


def make_activation(activation):
    return TanhP() if activation == "tanh" else PReLUP()  # the reference: nn.Tanh() if activation == "tanh" else nn.PReLU()


def slope_grad(act):
    return None if act.weight is None else act.weight.grad


class PReLUP(Leaf):
    """nn.PReLU() (one shared slope, init 0.25); stored in a 16-byte slot."""

    def __init__(self):
        super().__init__()
        self.declare("weight", (4,))

    def export_tensor(self, name, t):
        return t[:1].clone()

    def import_tensor(self, name, t):
        out = torch.zeros(4, dtype=torch.float32, device=t.device)
        out[:1] = t
        return out

    def reset_parameters(self):
        with torch.no_grad():
            self.weight.zero_()
            self.weight[0] = 0.25


class Marker(nn.Module):
    """Parameter-free placeholder keeping the reference's nn.Sequential indices
    (Upsample at skip.0, CholeskyL at fc_sigma.1, ReLU in the MLPs, GradientReversalLayer)."""

    def __init__(self, what=""):
        super().__init__()
        self.what = what

    def extra_repr(self):
        return self.what

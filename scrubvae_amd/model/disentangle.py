"""Adversarial scrubbing heads (reference: src/scrubvae/model/disentangle.py:541-714).

Module tree and state_dict names follow the reference (``reversal.1.mlp1.0.weight`` ...);
the arithmetic runs in libscrubvae_hip.so through ``EnsembleRunner``: every Linear is the
MFMA implicit-GEMM kernel (a 1x1 "conv"), ReLU / losses are the HIP row kernels.
Streaming closed-form scrubbers (MovingAvgLeastSquares, QDA, ...) are
outside this round's scope.
"""
from __future__ import annotations

import torch
import torch.nn as nn

from .. import ops
from ..ops import pad16
from .layers import LinearP, Marker


def _mlp(dims):
    mods = []
    for i in range(len(dims) - 1):
        mods.append(LinearP(dims[i], dims[i + 1]))
        if i < len(dims) - 2:
            mods.append(Marker("ReLU"))
    return nn.Sequential(*mods)


class MLPEnsemble(nn.Module):
    """Four MLPs on the same input (disentangle.py:583-632):
    (in,in,in,out), (in,in,out), (in,in,in//2,out), (in,2in,2in,out), ReLU between."""

    def __init__(self, in_dim, out_dim, bound=False):
        super().__init__()
        self.in_dim, self.out_dim = in_dim, out_dim
        self.mlp1 = _mlp([in_dim, in_dim, in_dim, out_dim])
        self.mlp2 = _mlp([in_dim, in_dim, out_dim])
        self.mlp3 = _mlp([in_dim, in_dim, in_dim // 2, out_dim])
        self.mlp4 = _mlp([in_dim, in_dim * 2, in_dim * 2, out_dim])

    def members(self):
        return [self.mlp1, self.mlp2, self.mlp3, self.mlp4]


class GRScrubber(nn.Module):
    """GradientReversalLayer(alpha) -> MLPEnsemble (disentangle.py:635-660)."""

    def __init__(self, in_dim, out_dim, alpha=1.0, bound=False):
        super().__init__()
        self.alpha = float(alpha)
        self.reversal = nn.Sequential(Marker(f"GradientReversal(alpha={alpha})"), MLPEnsemble(in_dim, out_dim, bound))

    @property
    def ensemble(self):
        return self.reversal[1]

    def reset_parameters(self):
        """Re-initialise every Linear of the four heads (called each epoch, trainer.py:368-370)."""
        for mlp in self.ensemble.members():
            for m in mlp:
                if isinstance(m, LinearP):
                    m.reset_parameters()


class AdvNetScrubber(nn.Module):
    """Frozen MLP-ensemble discriminator on cat(z, v) with softmax outputs
    (disentangle.py:663-684).  As in the reference its parameters do not require grad and,
    because the reference's fit branch is dead (trainer.py:133 compares mode to "Train"), it
    stays at its initialisation; gradients still flow through it to the encoder."""

    def __init__(self, in_dim):
        super().__init__()
        self.ensemble = MLPEnsemble(in_dim, 2, False)


class EnsembleRunner:
    """Forward/backward of one MLPEnsemble for a fixed number of rows on the HIP kernels."""

    def __init__(self, ens: MLPEnsemble, rows: int, device):
        self.ens, self.rows = ens, rows
        self.members = []
        for mlp in ens.members():
            lins = [m for m in mlp if isinstance(m, LinearP)]
            layers = []
            for lin in lins:
                cv = ops.Conv(rows, 1, lin.in_f, lin.out_f, 1)
                layers.append(dict(lin=lin, cv=cv,
                                   pre=torch.zeros(rows, cv.c_out_p, device=device),
                                   act=torch.zeros(rows, cv.c_out_p, device=device),
                                   g=torch.zeros(rows, cv.c_out_p, device=device)))
            self.members.append(layers)
        self.in_p = pad16(ens.in_dim)
        self.g_in = torch.zeros(rows, self.in_p, device=device)
        ws = max(l["cv"].wgrad_workspace_bytes() for mem in self.members for l in mem)
        self.ws = torch.empty(ws // 4 + 16, device=device)

    def forward(self, x):
        """x [rows, in_p] -> list of 4 pre-activation outputs [rows, out_p] (views)."""
        outs = []
        for layers in self.members:
            h = x
            for i, l in enumerate(layers):
                l["cv"].fwd(h, l["lin"].weight, l["lin"].bias, l["pre"])
                if i < len(layers) - 1:
                    ops.relu_fwd(l["pre"], l["act"])
                    h = l["act"]
            outs.append(layers[-1]["pre"])
        self.x = x
        return outs

    def backward(self, d_outs, param_grads=True, accumulate=False):
        """d_outs: list of 4 grads w.r.t. the outputs.  Returns grad w.r.t. x ([rows,in_p]).
        Parameter grads go to lin.weight.grad / lin.bias.grad (views of the flat grad buffer)."""
        first = True
        for layers, d_out in zip(self.members, d_outs):
            g = d_out
            for i in range(len(layers) - 1, -1, -1):
                l = layers[i]
                xin = self.x if i == 0 else layers[i - 1]["act"]
                if param_grads:
                    l["cv"].wgrad(xin, g, l["lin"].weight.grad, l["lin"].bias.grad, self.ws, accumulate=accumulate)
                if i == 0:
                    l["cv"].dgrad(g, l["lin"].weight, self.g_in, accumulate=not first)
                    first = False
                else:
                    prev = layers[i - 1]
                    l["cv"].dgrad(g, l["lin"].weight, prev["g"], accumulate=False)
                    ops.relu_bwd(prev["g"], prev["act"], prev["g"])
                    g = prev["g"]
        return self.g_in

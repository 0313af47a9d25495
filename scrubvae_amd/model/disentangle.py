"""Adversarial scrubbing heads (reference: src/scrubvae/model/disentangle.py:541-714).

Module tree and state_dict names follow the reference (``reversal.1.mlp1.0.weight`` ...);
the arithmetic runs in libscrubvae_hip.so through ``EnsembleRunner``: every Linear is the
MFMA implicit-GEMM kernel (a 1x1 "conv"), ReLU / losses are the HIP row kernels.
The streaming closed-form scrubbers of SURVEY 8a row A2 (MovingAvgLeastSquares, MovingAverageFilter,
QuadraticDiscriminantFilter, LinearProjection, MutInfoEstimator) follow below, behind the reference's
API: their solves, Gaussian log-likelihoods and the kernel-density estimator run on csrc/latent.hip kernels
(ops.small_solve, ops.gauss_ll_autograd, ops.kde_mi_autograd), the [B x z] Gram products are library GEMMs,
and every loss seeds the HIP backward.
"""
from __future__ import annotations

import torch
import torch.nn as nn

from .. import ops
from ..ops import pad16
from .layers import DenseW, LinearP, Marker


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
    """Forward/backward of one MLPEnsemble for a fixed number of rows, Linear by Linear on the GEMM kernels (ensembles too
    wide for the fused kernels of FusedEnsembleRunner)."""
    fused = False

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


class FusedEnsembleRunner:
    """One MLPEnsemble for a fixed batch on the fused kernels of csrc/ensemble.hip: ONE launch forward (all four members, the
    input assembled in the kernel) and one backward launch plus a fixed-order reduction (weight / bias gradients, input
    gradient summed over the members and added to the latent seed with the gradient-reversal coefficient).

    halves = 1: rows = batch, input = src0[:, :n0] (GRScrubber: mu, or the `linear` method's z_null).
    halves = 2: rows = 2 * batch, input = cat(src0[:, :n0], src1) with column `shuf_col` of src1 taken from row perm[b] in the
    second copy (AdvNetScrubber on cat([mu;mu],[v;v_shuffle]), disentangle.py:678-684)."""

    fused = True

    def __init__(self, ens: MLPEnsemble, batch: int, device, halves=1):
        from .. import _lib
        self.ens, self.batch, self.halves, self.rows = ens, batch, halves, batch * halves
        self.in_p = pad16(ens.in_dim)
        self.out_p = pad16(ens.out_dim)
        self.lins = [[m for m in mlp if isinstance(m, LinearP)] for mlp in ens.members()]
        if len(self.lins) > _lib.ENS_MEMBERS or any(len(l) > _lib.ENS_MAX_LAYERS for l in self.lins):
            raise ValueError("ensemble shape outside the fused kernel's limits")
        self.outs = [torch.zeros(self.rows, self.out_p, device=device) for _ in self.lins]
        self.d_outs = [torch.zeros(self.rows, self.out_p, device=device) for _ in self.lins]
        self.gx_raw = None
        self.ws = None
        self.desc = _lib.EnsDesc()
        self._fill_static()

    def _fill_static(self):
        d = self.desc
        d.n_members = len(self.lins)
        d.batch, d.halves = self.batch, self.halves
        for mi, lins in enumerate(self.lins):
            m = d.member[mi]
            m.n_layers = len(lins)
            m.out, m.d_out = self.outs[mi].data_ptr(), self.d_outs[mi].data_ptr()
            for li, lin in enumerate(lins):
                L = m.layer[li]
                L.w, L.b = lin.weight.data_ptr(), lin.bias.data_ptr()
                L.K, L.N = lin.in_lib, lin.out_lib

    def fits(self):
        """whether the fused kernels accept this ensemble (its activations fit one workgroup's LDS)"""
        d = self.desc
        d.src0, d.ld0, d.n0 = self.outs[0].data_ptr(), self.in_p, min(self.ens.in_dim, self.in_p)  # placeholders for the check
        d.src1, d.ld1, d.n1, d.perm, d.shuf_col = None, 0, 0, None, 0
        keep = d.halves
        d.halves = 1
        ok = ops.ens_bwd_workspace(d) > 0
        d.halves = keep
        return ok

    def _bind(self, src0, n0, src1=None, perm=None, shuf_col=0, param_grads=False, shuf_vals=None):
        d = self.desc
        d.shuf_vals = None if shuf_vals is None else shuf_vals.data_ptr()
        d.src0, d.ld0, d.n0 = src0.data_ptr(), src0.shape[1], n0
        if src1 is not None:
            d.src1, d.ld1, d.n1 = src1.data_ptr(), src1.shape[1], src1.shape[1]
        else:
            d.src1, d.ld1, d.n1 = None, 0, 0
        d.perm, d.shuf_col = (None if perm is None else perm.data_ptr()), int(shuf_col)
        for mi, lins in enumerate(self.lins):
            for li, lin in enumerate(lins):
                L = d.member[mi].layer[li]
                L.w, L.b = lin.weight.data_ptr(), lin.bias.data_ptr()
                g = param_grads and lin.weight.grad is not None
                L.dw = lin.weight.grad.data_ptr() if g else None
                L.db = lin.bias.grad.data_ptr() if g else None

    def forward(self, src0, n0, src1=None, perm=None, shuf_col=0, shuf_vals=None):
        """-> list of 4 pre-activation outputs [rows, out_p].  halves = 2: give `perm` (int64 [batch], local rows of src1) or
        `shuf_vals` (fp32 [batch], the shuffled column's values themselves)."""
        for t in (src0, src1, shuf_vals):
            ops._f32c(t, "ensemble input")
        if perm is not None and not (perm.is_cuda and perm.dtype == torch.int64 and perm.is_contiguous()):
            raise ValueError("perm: contiguous int64 CUDA tensor expected")
        self._inputs = (src0, n0, src1, perm, shuf_col, shuf_vals)
        self._bind(src0, n0, src1, perm, shuf_col, shuf_vals=shuf_vals)
        ops.ens_fwd(self.desc)
        return self.outs

    def backward(self, d_src0, coef, param_grads=True, accumulate=False, want_raw=False):
        """Gradients of sum_m <d_outs[m], out_m>: parameter gradients into the Linear layers' .grad (unless frozen), and
        d_src0[:, :n0] += coef * (input gradient summed over members and halves).  want_raw: also return the un-scaled input
        gradient [rows, in_p] (the `linear` method chains it through its projection)."""
        src0, n0, src1, perm, shuf_col, shuf_vals = self._inputs
        self._bind(src0, n0, src1, perm, shuf_col, param_grads=param_grads, shuf_vals=shuf_vals)
        if self.ws is None:
            self.ws = torch.empty(ops.ens_bwd_workspace(self.desc) // 4 + 64, device=src0.device)
        need = ops.ens_bwd_workspace(self.desc)
        if need == 0:
            raise RuntimeError("ens_bwd: descriptor rejected: " + __import__("scrubvae_amd")._lib.last_error())
        if self.ws.numel() * 4 < need:
            self.ws = torch.empty(need // 4 + 64, device=src0.device)
        raw = None
        if want_raw:
            if self.gx_raw is None:
                self.gx_raw = torch.zeros(self.rows, self.in_p, device=src0.device)
            raw = self.gx_raw
        ops.ens_bwd(self.desc, d_src0, 0 if d_src0 is None else d_src0.shape[1], coef, raw, self.ws, accumulate)
        return raw


class MovingAvgLeastSquares(nn.Module):
    """Streaming closed-form linear scrubber (reference: disentangle.py:393-538).

    Two exponentially-forgetting covariance pairs (Sxx, Sxy) with forgetting factors lam0 < lam1 = lam0 + lamdiff;
    forward solves both normal equations and predicts y from the latent means, `evaluate_loss` returns the mean of
    the two squared errors and nudges the forgetting factors toward the better decoder, `update` folds a batch into
    the covariances.  Both [z x z] solves run in one launch of the batched LU kernel (ops.small_solve), the [B x z] products
    are library GEMMs; the gradient into the encoder is seeded analytically by
    train.losses (the decoders W are constants of the step).  `polynomial_order` p > 1 appends, for every degree
    d = 2..p, the products of all size-d multisets of latent dimensions scaled by nx / (number of such products)
    (disentangle.py:440-464); the seed then also goes through the expansion's Jacobian."""

    def __init__(self, nx, ny, lamdiff=1e-1, delta=1e-4, bias=False, polynomial_order=1, l2_reg=0):
        super().__init__()
        import math
        self.bias = bool(bias)
        self.polynomial_order = int(polynomial_order)
        self.nx_in = int(nx)
        self.nx_poly = sum(math.comb(int(nx) + d - 1, d) for d in range(1, self.polynomial_order + 1))
        nx = self.nx_poly + int(self.bias)
        self.l2_reg = 0 if l2_reg is None else l2_reg
        print("Moving Avg Least Squares Bias: {}".format(self.bias))
        self.register_buffer("Sxx0", torch.eye(nx))
        self.register_buffer("Sxy0", torch.zeros(nx, ny))
        self.register_buffer("Sxx1", torch.eye(nx))
        self.register_buffer("Sxy1", torch.zeros(nx, ny))
        self.register_buffer("lam0", torch.tensor([0.9]))
        self.register_buffer("lam1", self.lam0 + lamdiff)
        self.delta = delta
        self.lamdiff = lamdiff
        self.process_group = None  # set by parallel.attach: batch statistics are summed over the ranks
        self._W = None

    def polynomial_expansion(self, x):
        cols = [x]
        idx = torch.arange(x.shape[1], dtype=torch.long, device=x.device)
        for d in range(2, self.polynomial_order + 1):
            c = torch.combinations(idx, d, with_replacement=True)
            cols.append(x[:, c].prod(dim=-1) / len(c) * x.shape[-1])
        return torch.column_stack(cols) if len(cols) > 1 else x

    def _design(self, x):
        x = self.polynomial_expansion(x[:, : self.nx_in])
        if self.bias:
            x = torch.column_stack((x, torch.ones(x.shape[0], 1, device=x.device)))
        return x

    def forward(self, x):
        x = self._design(x)
        l2 = torch.ones(x.shape[1], device=x.device) * self.l2_reg
        if self.bias:
            l2[-1] = 0
        if x.is_cuda and x.shape[1] <= 64 and self.Sxy0.shape[1] <= 64:  # both normal equations in one launch of the batched LU kernel
            W = ops.small_solve(torch.stack((self.Sxx0, self.Sxx1)), torch.stack((self.Sxy0, self.Sxy1)), l2)
            W0, W1 = W[0], W[1]
        else:  # host tensors (CPU unit tests of the scrubber logic) / designs wider than the kernel's 64 columns
            W0 = torch.linalg.solve(self.Sxx0.diagonal_scatter(self.Sxx0.diagonal() + l2), self.Sxy0)
            W1 = torch.linalg.solve(self.Sxx1.diagonal_scatter(self.Sxx1.diagonal() + l2), self.Sxy1)
        self._W = (W0, W1)
        return [x @ W0, x @ W1]

    def latent_seed(self, yhat0, yhat1, y, x, scale):
        """scale * d[(l0 + l1) / 2] / d x for the predictions of the last forward(x), the decoders W held constant:
        ((yhat0 - y) W0^T + (yhat1 - y) W1^T) on the design columns, chained through the polynomial expansion's Jacobian
        when there is one."""
        W0, W1 = self._W
        n = self.nx_poly
        seed = scale * ((yhat0 - y) @ W0[:n].T + (yhat1 - y) @ W1[:n].T)
        if self.polynomial_order > 1:
            lat = x[:, : self.nx_in].detach().clone().requires_grad_(True)
            with torch.enable_grad():
                seed = torch.autograd.grad(self.polynomial_expansion(lat), lat, grad_outputs=seed)[0]
        return seed

    def _allreduce(self, t):
        _rank_sum(t, self.process_group)
        return t

    def update(self, x, y):
        x = self._design(x)
        xx = self._allreduce((x.T @ x).detach())
        xy = self._allreduce((x.T @ y.to(x.dtype)).detach())
        self.Sxx0 = self.lam0 * self.Sxx0 + xx
        self.Sxy0 = self.lam0 * self.Sxy0 + xy
        self.Sxx1 = self.lam1 * self.Sxx1 + xx
        self.Sxy1 = self.lam1 * self.Sxy1 + xy
        return self

    def evaluate_loss(self, yhat0, yhat1, y):
        """(l0 + l1) / 2 with l = summed squared error (disentangle.py:505-538); the forgetting factors move by
        `delta` toward the better decoder (on device, no host sync; the sums are global under data parallelism)."""
        y = y.to(yhat0.dtype)
        l = torch.stack([((y - yhat0) ** 2).sum(), ((y - yhat1) ** 2).sum()])
        lg = self._allreduce(l.detach().clone())
        down = lg[0] < lg[1]
        lam0_dn = torch.clamp(self.lam0 - self.delta, 0.0, 1.0)
        lam1_up = torch.clamp(self.lam1 + self.delta, 0.0, 1.0)
        self.lam0 = torch.where(down, lam0_dn, lam1_up - self.lamdiff)
        self.lam1 = torch.where(down, lam0_dn + self.lamdiff, lam1_up)
        return (l[0] + l[1]) * 0.5


def _rank_sum(t, group):
    """In-place SUM over the data-parallel ranks; False when there is a single rank."""
    import torch.distributed as dist
    if dist.is_initialized() and dist.get_world_size(group) > 1:
        if t.is_cuda and dist.get_backend(group) == "gloo":  # test configuration (ranks sharing one GPU): stage through the host
            h = t.detach().cpu()
            dist.all_reduce(h, op=dist.ReduceOp.SUM, group=group)
            t.copy_(h)
        else:
            dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
        return True
    return False


class LinearProjection(nn.Module):
    """Linear decoder of one feature plus the projection of the latent onto the decoder's null space (reference:
    disentangle.py:717-734; method `linear`, loss key `<feat>_lin`, losses.py:258-265): v = z Wt (+ b),
    z_null = z - (W Wt)^-1 v ... W, i.e. z minus its component in the row space of W.  When the method is configured every
    other scrubber of that feature reads z_null instead of mu (residual.py:351-355, losses.py:232-235).  The [out x out]
    inverse runs on `small_solve_kernel` (csrc/latent.hip) behind an autograd function, the [B x z] products are library GEMMs; the model keeps the small autograd graph and
    train.losses / the HIP backward differentiate through it to mu and to `decoder.weight`."""

    def __init__(self, in_dim, out_dim, bias=False):
        super().__init__()
        self.decoder = DenseW(in_dim, out_dim, bias=bias)

    def leaves(self):
        return [p for p in (self.decoder.weight, getattr(self.decoder, "bias", None)) if p is not None]

    def forward(self, z):
        w = self.decoder.weight
        x = z @ w.T
        if self.decoder.has_bias:
            x = x + self.decoder.bias
        nrm = w @ w.T
        if nrm.is_cuda and nrm.shape[0] <= 64:  # [out x out] inverse on the batched LU kernel; the [B x out] products are library GEMMs
            from .. import ops
            z_null = z - (x @ ops.small_inverse_autograd(nrm).T) @ w
        else:
            z_null = z - torch.linalg.solve(nrm, x.T).T @ w
        return {"v": x, "z_null": z_null}


def _rank_cat(t, group):
    """Concatenation of `t` over the data-parallel ranks along dim 0 (rank order); `t` itself on a single rank."""
    import torch.distributed as dist
    if not (dist.is_initialized() and dist.get_world_size(group) > 1):
        return t
    n = dist.get_world_size(group)
    if t.is_cuda and dist.get_backend(group) == "gloo":  # test configuration: stage through the host
        h = t.detach().cpu().contiguous()
        parts = [torch.empty_like(h) for _ in range(n)]
        dist.all_gather(parts, h, group=group)
        return torch.cat(parts, 0).to(t.device)
    t = t.detach().contiguous()
    parts = [torch.empty_like(t) for _ in range(n)]
    dist.all_gather(parts, t, group=group)
    return torch.cat(parts, 0)


class MutInfoEstimator(nn.Module):
    """Kernel-density estimate of the mutual information between the latent means and the conditioning variables
    (reference: disentangle.py:234-317; loss key `mcmi`, losses.py:221-225).  The mixture centres are the means /
    conditions of the previous batch (trainer.py:184-199): p(x,y) = mean_s N(x; x_s, var_s) N(y; y_s, gamma), and the
    value is mean_b[log p(x_b,y_b) - log p(x_b) - log p(y_b)] with the 1/num_s factors dropped exactly as the reference
    drops them (its log-sum-exps are not normalised).  `var_mode="sphere"`: var_s = bandwidth; `"diagonal"`: var_s =
    diag(L_s)^2 + bandwidth per centre.  O(B * num_s * (z + D)); on the device one launch of `kde_mi_kernel` (csrc/latent.hip:
    values and d / d mu per sample, centres staged through LDS tiles -- the [B, num_s, z] difference tensor of the reference never
    exists); CPU tensors (the CPU test suite) take the chunked torch evaluation below.  train.losses seeds the HIP backward with
    d mcmi / d mu.  The reference's `device` argument only places its
    constants; here they follow the centres."""

    CHUNK = 256

    def __init__(self, x_s, y_s, bandwidth, var_mode="sphere", model_var=None, device=None):
        super().__init__()
        import math
        self.register_buffer("x_s", x_s)
        self.register_buffer("y_s", y_s.to(x_s.dtype))
        self.num_s, self.x_dim, self.y_dim = x_s.shape[0], x_s.shape[1], y_s.shape[1]
        assert y_s.shape[0] == self.num_s
        self.var_mode = var_mode
        log2pi = math.log(2 * math.pi)
        if var_mode == "sphere":
            self.register_buffer("var_s", torch.tensor([bandwidth], device=x_s.device, dtype=x_s.dtype))
            logA_x = self.x_dim * (log2pi + torch.log(self.var_s))                       # [1]
        elif var_mode == "diagonal":
            self.register_buffer("var_s", model_var.diagonal(dim1=-2, dim2=-1) ** 2 + bandwidth)   # [num_s, x_dim]
            logA_x = (self.x_dim * log2pi + torch.sum(torch.log(self.var_s), dim=-1))[None, :]     # [1, num_s]
        else:
            raise ValueError(f"var_mode {var_mode!r} (the reference defines 'sphere' and 'diagonal')")
        self.gamma = bandwidth
        self.register_buffer("logA_x", logA_x)
        self._logA_y = self.y_dim * (log2pi + math.log(bandwidth))  # host copy of the constant: no device read-back per step
        self.register_buffer("logA_y", torch.tensor([self._logA_y], device=x_s.device, dtype=x_s.dtype))

    def forward(self, x, y):
        y = y.to(x.dtype)
        if x.is_cuda and self.x_dim <= 64 and self.y_dim <= 64:  # one launch: the three log-sum-exps and d / d x per sample
            from .. import ops
            val = ops.kde_mi_autograd(x, y, self.x_s, self.y_s, self.var_s, self.logA_x, float(self._logA_y), self.gamma)
            return val.mean()
        lse = [[], [], []]
        for s0 in range(0, self.num_s, self.CHUNK):
            s1 = min(s0 + self.CHUNK, self.num_s)
            dx = x[:, None, :] - self.x_s[None, s0:s1, :]
            dy = y[:, None, :] - self.y_s[None, s0:s1, :]
            var = self.var_s if self.var_mode == "sphere" else self.var_s[None, s0:s1, :]
            sdx = ((dx / var) * dx).sum(dim=-1)
            sdy = ((dy / self.gamma) * dy).sum(dim=-1)
            la = self.logA_x if self.var_mode == "sphere" else self.logA_x[:, s0:s1]
            lse[0].append(torch.logsumexp(-0.5 * (la + self.logA_y + sdx + sdy), dim=-1))
            lse[1].append(torch.logsumexp(-0.5 * (la + sdx), dim=-1))
            lse[2].append(torch.logsumexp(-0.5 * (self.logA_y + sdy), dim=-1))
        pxy, px, py = (torch.logsumexp(torch.stack(l, 0), dim=0) for l in lse)
        return (pxy - px - py).mean()


class MovingAverageFilter(nn.Module):
    """Streaming class-mean scrubber for a discrete variable (reference: disentangle.py:9-87; loss key `<feat>_ma`,
    losses.py:286-289).  Two exponentially-forgetting estimates of every class mean of the latent; the loss is the
    Frobenius norm of all pairwise differences of the class-mean estimates after folding in the current batch, and the
    per-class forgetting factors move toward the estimate that was closer to the batch mean.  Stock torch device ops
    (SURVEY 8a row A2); train.losses differentiates the [B, z] graph with torch.autograd and seeds the HIP backward.
    Under data parallelism the class sums and counts are summed over the ranks (the gradient stays local)."""

    def __init__(self, nx, classes, lamdiff=1e-2, delta=1e-3):
        super().__init__()
        self.classes = classes
        k = len(classes)
        self.register_buffer("m1", torch.zeros(k, nx))
        self.register_buffer("m2", torch.zeros(k, nx))
        self.register_buffer("lam1", torch.ones(k) * 0.5)
        self.register_buffer("lam2", self.lam1 + lamdiff)
        self.delta = delta
        self.lamdiff = lamdiff
        self.process_group = None

    def forward(self, *args, **kwargs):
        return None

    def _class_means(self, x, y):
        labels = torch.as_tensor(self.classes, device=x.device).reshape(1, -1)
        onehot = (y.reshape(-1, 1).to(labels.dtype) == labels).to(x.dtype)  # [B, K]
        sums, cnt = onehot.T @ x, onehot.sum(0)
        g_sums, g_cnt = sums.detach().clone(), cnt.clone()
        if _rank_sum(g_sums, self.process_group):
            _rank_sum(g_cnt, self.process_group)
            sums, cnt = sums + (g_sums - sums.detach()), g_cnt
        return sums / cnt[:, None]

    def evaluate_loss(self, x, y):
        xbar = self._class_means(x, y)
        with torch.no_grad():
            down = torch.linalg.norm(xbar - self.m1, dim=1) < torch.linalg.norm(xbar - self.m2, dim=1)
            lam1_dn = torch.clamp(self.lam1 - self.delta, 0.0, 1.0)
            lam2_up = torch.clamp(self.lam2 + self.delta, 0.0, 1.0)
            self.lam1 = torch.where(down, lam1_dn, lam2_up - self.lamdiff)
            self.lam2 = torch.where(down, lam1_dn + self.lamdiff, lam2_up)
        m1 = (1 - self.lam1)[:, None] * xbar + self.lam1[:, None] * self.m1
        m2 = (1 - self.lam2)[:, None] * xbar + self.lam2[:, None] * self.m2
        est = 0.5 * (m1 + m2)
        d = torch.triu(est.T[..., None] - est.T[..., None, :], diagonal=1)
        return torch.linalg.norm(d)

    def update(self, x, y):
        with torch.no_grad():
            xbar = self._class_means(x, y)
            self.m1 = (1 - self.lam1)[:, None] * xbar + self.lam1[:, None] * self.m1
            self.m2 = (1 - self.lam2)[:, None] * xbar + self.lam2[:, None] * self.m2
        return self


class QuadraticDiscriminantFilter(nn.Module):
    """Two streaming one-vs-rest quadratic discriminants per class with automatically tuned forgetting factors
    (reference: disentangle.py:90-232; loss key `<feat>_qda`, losses.py:248-252).  On the device the 4 x classes Gaussian
    log-likelihoods and their gradient fields come from one launch of `gauss_ll_kernel` (csrc/latent.hip: in-LDS inversion with
    partial pivoting per (mean, covariance) pair, one sample per thread); the moment updates are small torch reductions.  Single-rank statistics use torch.mean / torch.cov exactly as the reference;
    under data parallelism the member / non-member moments are summed over the ranks."""

    def __init__(self, nx, classes, lamdiff=1e-2, delta=1e-3):
        super().__init__()
        self.classes = classes
        k = len(classes)
        for name in ("0a", "1a", "0b", "1b"):
            self.register_buffer("m" + name, torch.zeros(k, nx))
            self.register_buffer("S" + name, torch.eye(nx)[None, :].repeat(k, 1, 1))
        self.register_buffer("lama", torch.ones(k) * 0.2)
        self.register_buffer("lamb", self.lama + lamdiff)
        self.delta = delta
        self.lamdiff = lamdiff
        self.process_group = None

    def forward(self, *args, **kwargs):
        return None

    @staticmethod
    def cgll(x, m, S):
        r = x - m
        resids = torch.sum(r * torch.linalg.solve(S, r.T).T, dim=1)
        return -0.5 * (torch.logdet(S) + resids)

    def _moments(self, x, mask):
        import torch.distributed as dist
        if dist.is_initialized() and dist.get_world_size(self.process_group) > 1:
            w = mask.to(x.dtype)[:, None]
            stats = torch.cat([(w * x).sum(0), ((w * x).T @ x).reshape(-1), w.sum().reshape(1)])
            _rank_sum(stats, self.process_group)
            nx, n = x.shape[1], stats[-1]
            mean = stats[:nx] / n
            return mean[None], stats[nx:-1].reshape(nx, nx) / n - mean[:, None] * mean[None, :]
        xs = x[mask]
        return torch.mean(xs, dim=0, keepdim=True), torch.cov(xs.T, correction=0)

    def update(self, x, y):
        with torch.no_grad():
            for i, label in enumerate(self.classes):
                i1 = (y == label).ravel()
                x0m, x0S = self._moments(x, ~i1)
                x1m, x1S = self._moments(x, i1)
                for tag, lam in (("a", self.lama[i]), ("b", self.lamb[i])):
                    for cls, (xm, xS) in (("0", (x0m, x0S)), ("1", (x1m, x1S))):
                        getattr(self, "m" + cls + tag)[i] = (1 - lam) * getattr(self, "m" + cls + tag)[i] + lam * xm
                        getattr(self, "S" + cls + tag)[i] = (1 - lam) * getattr(self, "S" + cls + tag)[i] + lam * xS
        return self

    def evaluate_loss(self, x, y, update=True):
        ll_loss = 0
        ll_all = None
        if x.is_cuda and x.shape[1] <= 64:  # all 4 x classes log-likelihoods (and their gradient fields) in ONE launch
            from .. import ops
            k, nx = len(self.classes), x.shape[1]
            means = torch.stack([self.m0a, self.m1a, self.m0b, self.m1b])  # [4, k, nx]
            covs = torch.stack([self.S0a, self.S1a, self.S0b, self.S1b])   # [4, k, nx, nx]
            ll_all = ops.gauss_ll_autograd(x, means.reshape(4 * k, nx), covs.reshape(4 * k, nx, nx)).view(4, k, -1)
        for i, label in enumerate(self.classes):
            i1 = (y == label).ravel()
            i0 = ~i1
            if ll_all is not None:
                lla0, lla1, llb0, llb1 = ll_all[0, i], ll_all[1, i], ll_all[2, i], ll_all[3, i]
            else:
                lla0, lla1 = self.cgll(x, self.m0a[i: i + 1], self.S0a[i]), self.cgll(x, self.m1a[i: i + 1], self.S1a[i])
                llb0, llb1 = self.cgll(x, self.m0b[i: i + 1], self.S0b[i]), self.cgll(x, self.m1b[i: i + 1], self.S1b[i])
            if update:
                with torch.no_grad():
                    ll = torch.stack([torch.sum(i0 * lla0 + i1 * lla1), torch.sum(i0 * llb0 + i1 * llb1)])
                    _rank_sum(ll, self.process_group)
                    a_better = ll[0] > ll[1]
                    lama_dn = torch.clamp(self.lama[i] - self.delta, 0.0, 1.0)
                    lamb_up = torch.clamp(self.lamb[i] + self.delta, 0.0, 1.0)
                    self.lama[i] = torch.where(a_better, lama_dn, lamb_up - self.lamdiff)
                    self.lamb[i] = torch.where(a_better, lama_dn + self.lamdiff, lamb_up)
            batch_y = (i1 * 2 - 1).to(x.dtype)
            ll_loss = ll_loss + (batch_y @ (lla1 - lla0) + batch_y @ (llb1 - llb0)) * 0.5
        return ll_loss / len(self.classes)

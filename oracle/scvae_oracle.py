"""CPU oracle for the SC-VAE training step.  TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this
module; the product path (scrubvae_amd/) never does.

This is a functional restatement of the reference hot path on
stock PyTorch-CPU ops.  The reference's arithmetic lives entirely in PyTorch
(pinned upstream: pytorch=1.13.1, environment.yml:234; here 2.10 CPU), so the restatement
drives the same library calls the reference's call sites make, arranged functionally over
a reference-named ``state_dict`` instead of nn.Modules.  Works in fp32 and fp64.

Parity pin: tests/test_oracle_golden.py checks every function here against fixtures
captured from the real reference (tests/golden/make_fixtures.py, run in the build
container where /root/reference is importable).

Each function cites the reference file:line it follows (paths relative to the reference
root).
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Sequence

import torch
import torch.nn.functional as F

LN2PI = math.log(2 * math.pi)

MOUSE_KINEMATIC_TREE = [
    [0, 1, 2, 3, 4],
    [0, 5],
    [1, 6, 7, 8],
    [1, 9, 10, 11],
    [5, 12, 13, 14],
    [5, 15, 16, 17],
]  # configs/mouse_skeleton.yaml:86-92

MOUSE_OFFSET = [
    [0, 0, 0], [1, 0, 0], [1, 0, 0], [1, 0, 0], [1, 0, 0], [-1, 0, 0],
    [0, 1, 0], [0, 1, 0], [0, 1, 0], [0, -1, 0], [0, -1, 0], [0, -1, 0],
    [0, 1, 0], [0, 1, 0], [0, 1, 0], [0, -1, 0], [0, -1, 0], [0, -1, 0],
]  # configs/mouse_skeleton.yaml:95-113

FEAT_DIMS = {  # get/model.py:19-27
    "avg_speed": 1, "part_speed": 4, "avg_speed_3d": 3, "heading": 2,
    "heading_change": 1, "fluorescence": 1,
}


@dataclass
class OracleConfig:
    n_keypts: int = 18
    window: int = 64
    z_dim: int = 32
    kernel: int = 5
    channel: Sequence[int] = (64, 128, 256, 512, 1024)
    diag: bool = True
    arena_size: Optional[torch.Tensor] = None  # [2,3]
    kinematic_tree: List[List[int]] = field(default_factory=lambda: MOUSE_KINEMATIC_TREE)
    # disentangle method dict as in config["disentangle"]["method"]
    method: Dict[str, List[str]] = field(default_factory=dict)
    features: Optional[List[str]] = None  # disentangle_keys
    alpha: float = 1.0
    discrete_classes: Optional[Dict[str, torch.Tensor]] = None
    bn_eps: float = 1e-4  # residual.py:88,112,146,173
    bn_momentum: float = 0.1
    activation: str = "prelu"  # model.activation: "tanh" puts nn.Tanh() wherever the default has nn.PReLU() (residual.py:89)
    prior: str = "gaussian"    # model.prior: "beta" = Beta posterior, fc_alpha / fc_beta heads (residual.py:223-239,301-302,328-331,453-456)

    @property
    def in_channels(self):  # get/model.py:33-35
        return self.n_keypts * 6 + 3

    def feat_dim(self, k):
        if k == "frame_speed":
            return self.window - 1
        if self.discrete_classes is not None and k in self.discrete_classes:
            return len(self.discrete_classes[k])
        return FEAT_DIMS[k]

    @property
    def conditional_keys(self):
        return self.method.get("conditional", None)

    @property
    def conditional_dim(self):  # get/model.py:51-56
        ck = self.conditional_keys
        return 0 if not ck else sum(self.feat_dim(k) for k in ck)


# --------------------------------------------------------------------------- shape maths
def find_latent_dim(window, kernel, num_layers):
    """residual.py:6-20 (default dilation=1 => stride 2): float division, one int()."""
    l = window
    for _ in range(num_layers):
        l = (l + 2 * (kernel // 2) - (kernel - 1) - 1) / 2 + 1
    return int(l)


def find_out_dim(latent_dim, kernel, num_layers):
    """residual.py:23-36 (default dilation)."""
    l = latent_dim
    for _ in range(num_layers):
        l = (l - 1) * 2 - 2 * (kernel // 2) + (kernel - 1) + 1
    return int(l)


def final_kernel(cfg: OracleConfig):
    """residual.py:280-286."""
    n = len(cfg.channel) - 1
    l_out = find_out_dim(find_latent_dim(cfg.window, cfg.kernel, n), cfg.kernel, n)
    return cfg.window - l_out + 7


# --------------------------------------------------------------------------- init
def init_state_dict(cfg: OracleConfig, seed=0, dtype=torch.float32, scale=1.0):
    """Deterministic synthetic weights with the reference's state_dict names/shapes
    (SURVEY 8b 'Checkpoint compatibility').  Values: N(0, fan_in^-1/2) style so the
    activations stay O(1) -- NOT the reference's init (parity tests install the same
    state_dict on both sides, so the init law is irrelevant)."""
    g = torch.Generator().manual_seed(seed)
    sd = {}
    ch = list(cfg.channel)
    k = cfg.kernel
    C = cfg.in_channels

    def rnd(*shape, fan):
        return (torch.randn(*shape, generator=g, dtype=torch.float64) * scale / math.sqrt(fan)).to(dtype)

    def conv(name, cout, cin, kk):
        sd[name + ".weight"] = rnd(cout, cin, kk, fan=cin * kk)
        sd[name + ".bias"] = rnd(cout, fan=4.0)

    def convT(name, cin, cout, kk):
        sd[name + ".weight"] = rnd(cin, cout, kk, fan=cin * kk)
        sd[name + ".bias"] = rnd(cout, fan=4.0)

    def bn(name, c):
        sd[name + ".weight"] = (1.0 + 0.1 * torch.randn(c, generator=g, dtype=torch.float64)).to(dtype)
        sd[name + ".bias"] = (0.1 * torch.randn(c, generator=g, dtype=torch.float64)).to(dtype)
        sd[name + ".running_mean"] = torch.zeros(c, dtype=dtype)
        sd[name + ".running_var"] = torch.ones(c, dtype=dtype)
        sd[name + ".num_batches_tracked"] = torch.zeros((), dtype=torch.long)

    def prelu(name):
        if cfg.activation != "tanh":  # nn.Tanh() has no parameters
            sd[name + ".weight"] = torch.full((1,), 0.25, dtype=dtype)

    def lin(name, out, inp):
        sd[name + ".weight"] = rnd(out, inp, fan=inp)
        sd[name + ".bias"] = rnd(out, fan=4.0)

    if cfg.arena_size is not None:
        sd["arena_size"] = cfg.arena_size.to(dtype)
    conv("encoder.conv_in", ch[0], C, 7)
    prelu("encoder.activation")
    for i in range(len(ch) - 1):
        p = f"encoder.res_layers.{i}"
        conv(p + ".residual.0", ch[i + 1] // 2, ch[i], k)
        bn(p + ".residual.1", ch[i + 1] // 2)
        prelu(p + ".residual.2")
        conv(p + ".residual.3", ch[i + 1], ch[i + 1] // 2, k)
        conv(p + ".skip", ch[i + 1], ch[i], k)
        bn(p + ".add.0", ch[i + 1])
        prelu(p + ".add.1")
    flat = find_latent_dim(cfg.window, k, len(ch) - 1) * ch[-1]
    if cfg.prior == "beta":  # residual.py:223-225
        lin("encoder.fc_alpha", cfg.z_dim, flat)
        lin("encoder.fc_beta", cfg.z_dim, flat)
    else:
        lin("encoder.fc_mu", cfg.z_dim, flat)
        sig = cfg.z_dim if cfg.diag else cfg.z_dim * (cfg.z_dim + 1) // 2
        lin("encoder.fc_sigma.0", sig, flat)
    lin("decoder.fc_in", flat, cfg.z_dim + cfg.conditional_dim)
    for j, i in enumerate(range(1, len(ch))):
        cin, cout = ch[-i], ch[-i - 1]
        p = f"decoder.res_layers.{j}"
        convT(p + ".residual.0", cin, cin // 2, k)
        bn(p + ".residual.1", cin // 2)
        prelu(p + ".residual.2")
        convT(p + ".residual.3", cin // 2, cout, k)
        conv(p + ".skip.1", cout, cin, k + 1)
        bn(p + ".add.0", cout)
        prelu(p + ".add.1")
    convT("decoder.conv_out", ch[0], C, final_kernel(cfg))

    # scrubbers (get/model.py:58-113; disentangle.py:583-684)
    def ensemble(prefix, ind, outd):
        lin(prefix + ".mlp1.0", ind, ind); lin(prefix + ".mlp1.2", ind, ind); lin(prefix + ".mlp1.4", outd, ind)
        lin(prefix + ".mlp2.0", ind, ind); lin(prefix + ".mlp2.2", outd, ind)
        lin(prefix + ".mlp3.0", ind, ind); lin(prefix + ".mlp3.2", ind // 2, ind); lin(prefix + ".mlp3.4", outd, ind // 2)
        lin(prefix + ".mlp4.0", 2 * ind, ind); lin(prefix + ".mlp4.2", 2 * ind, 2 * ind); lin(prefix + ".mlp4.4", outd, 2 * ind)

    for feat in cfg.method.get("grad_reversal", []):
        ensemble(f"disentangle.grad_reversal.{feat}.reversal.1", cfg.z_dim, cfg.feat_dim(feat))
    for feat in cfg.method.get("adversarial_net", []):
        ensemble(f"disentangle.adversarial_net.{feat}.ensemble", cfg.z_dim + cfg.conditional_dim, 2)
    for feat in cfg.method.get("linear", []):  # LinearProjection(bias=False), get/model.py:40-49
        sd[f"disentangle.linear.{feat}.decoder.weight"] = rnd(cfg.feat_dim(feat), cfg.z_dim, fan=cfg.z_dim)
    return sd


TRAINABLE_EXCLUDE = ("running_mean", "running_var", "num_batches_tracked", "arena_size")


def trainable_names(sd):
    """Parameters the reference's main optimizer sees: everything but buffers and the
    adversarial_net ensemble (requires_grad=False, disentangle.py:670-671)."""
    out = []
    for n in sd:
        if n.endswith(TRAINABLE_EXCLUDE) or n == "arena_size":
            continue
        if n.startswith("disentangle.adversarial_net."):
            continue
        out.append(n)
    return out


# --------------------------------------------------------------------------- layers
def _prelu(x, w):
    """the block activation: PReLU with the given slope, or tanh when the model has none (w is None)"""
    return torch.tanh(x) if w is None else F.prelu(x, w)


def _bn(x, sd, name, cfg, train, new_stats):
    """nn.BatchNorm1d(eps=1e-4) train/eval semantics.  In train mode also records the
    updated running stats (momentum 0.1, unbiased var) into new_stats."""
    w, b = sd[name + ".weight"], sd[name + ".bias"]
    if train:
        mean = x.mean(dim=(0, 2))
        var = x.var(dim=(0, 2), unbiased=False)
        n = x.shape[0] * x.shape[2]
        if new_stats is not None:
            with torch.no_grad():
                m = cfg.bn_momentum
                new_stats[name + ".running_mean"] = (1 - m) * sd[name + ".running_mean"] + m * mean
                new_stats[name + ".running_var"] = (1 - m) * sd[name + ".running_var"] + m * var * n / max(n - 1, 1)
                new_stats[name + ".num_batches_tracked"] = sd[name + ".num_batches_tracked"] + 1
    else:
        mean, var = sd[name + ".running_mean"], sd[name + ".running_var"]
    xh = (x - mean[None, :, None]) / torch.sqrt(var[None, :, None] + cfg.bn_eps)
    return xh * w[None, :, None] + b[None, :, None]


def res_block(x, sd, p, cfg, train, new_stats):
    """ResidualBlock.forward, residual.py:71-119 (stride 2, dilation 1)."""
    k = cfg.kernel
    skip = F.conv1d(x, sd[p + ".skip.weight"], sd[p + ".skip.bias"], stride=2, padding=k // 2)
    h = F.conv1d(x, sd[p + ".residual.0.weight"], sd[p + ".residual.0.bias"], stride=2, padding=k // 2)
    h = _prelu(_bn(h, sd, p + ".residual.1", cfg, train, new_stats), sd.get(p + ".residual.2.weight"))
    h = F.conv1d(h, sd[p + ".residual.3.weight"], sd[p + ".residual.3.bias"], stride=1, padding=k // 2)
    return _prelu(_bn(h + skip, sd, p + ".add.0", cfg, train, new_stats), sd.get(p + ".add.1.weight"))


def res_block_T(x, sd, p, cfg, train, new_stats):
    """ResidualBlockTranspose.forward, residual.py:122-180."""
    k = cfg.kernel
    up = F.interpolate(x, scale_factor=2, mode="linear", align_corners=False)
    skip = F.conv1d(up, sd[p + ".skip.1.weight"], sd[p + ".skip.1.bias"], stride=1, padding=k // 2)
    h = F.conv_transpose1d(x, sd[p + ".residual.0.weight"], sd[p + ".residual.0.bias"], stride=1, padding=k // 2)
    h = _prelu(_bn(h, sd, p + ".residual.1", cfg, train, new_stats), sd.get(p + ".residual.2.weight"))
    h = F.conv_transpose1d(h, sd[p + ".residual.3.weight"], sd[p + ".residual.3.bias"], stride=2, padding=k // 2)
    return _prelu(_bn(h + skip, sd, p + ".add.0", cfg, train, new_stats), sd.get(p + ".add.1.weight"))


def normalize_root(root, arena):  # residual.py:428-431
    return 2 * (root - arena[0]) / (arena[1] - arena[0]) - 1


def inv_normalize_root(nr, arena):  # residual.py:433-436
    return 0.5 * (nr + 1) * (arena[1] - arena[0]) + arena[0]


def cholesky_L(raw, z_dim, diag):
    """CholeskyL.forward, residual.py:60-68."""
    B = raw.shape[0]
    L = torch.zeros(B, z_dim, z_dim, dtype=raw.dtype)
    if diag:
        i0 = i1 = torch.arange(z_dim)
    else:
        idx = torch.tril_indices(z_dim, z_dim)
        i0, i1 = idx[0], idx[1]
    L[:, i0, i1] = raw
    d = F.softplus(L.diagonal(dim1=-2, dim2=-1))
    return L.diagonal_scatter(d, dim1=-2, dim2=-1)


def encode(sd, cfg, data, train, new_stats=None):
    """ResVAE.encode, residual.py:438-459 + ResidualEncoder.forward :227-240."""
    x6d, root = data["x6d"], data["root"]
    B, W = x6d.shape[:2]
    if cfg.arena_size is not None:
        x_in = torch.cat((x6d.reshape(B, W, -1), normalize_root(root, sd["arena_size"])), dim=-1)
    else:
        x_in = x6d.reshape(B, W, -1)
    x = x_in.moveaxis(1, -1)
    x = _prelu(F.conv1d(x, sd["encoder.conv_in.weight"], sd["encoder.conv_in.bias"], padding=3),
               sd.get("encoder.activation.weight"))
    for i in range(len(cfg.channel) - 1):
        x = res_block(x, sd, f"encoder.res_layers.{i}", cfg, train, new_stats)
    flat = x.flatten(1)
    if cfg.prior == "beta":
        # ResidualEncoder.forward :235-239 (softplus + 1: one mode) and ResVAE.encode :453-456 (mu = the mode, rescaled to (-1, 1))
        alpha = F.softplus(F.linear(flat, sd["encoder.fc_alpha.weight"], sd["encoder.fc_alpha.bias"])) + 1
        beta = F.softplus(F.linear(flat, sd["encoder.fc_beta.weight"], sd["encoder.fc_beta.bias"])) + 1
        mu = (alpha - 1 + 1e-8) / (alpha + beta - 2 + 2e-8) * 2 - 1
        return {"alpha": alpha, "beta": beta, "mu": mu}
    mu = F.linear(flat, sd["encoder.fc_mu.weight"], sd["encoder.fc_mu.bias"])
    raw = F.linear(flat, sd["encoder.fc_sigma.0.weight"], sd["encoder.fc_sigma.0.bias"])
    return {"mu": mu, "L": cholesky_L(raw, cfg.z_dim, cfg.diag)}


class _BetaRSample(torch.autograd.Function):
    """Beta(alpha, beta).rsample() with the draw INJECTED (x in (0, 1), e.g. from torch's sampler): torch.distributions.Beta
    draws through Dirichlet([alpha, beta]) and differentiates the draw implicitly (torch/distributions/dirichlet.py
    _Dirichlet_backward: grad = _dirichlet_grad(x, conc, total) * (g - sum(x g))); with g on component 0 only:
    d alpha = D(x, alpha, total) (1 - x) g,  d beta = -D(1 - x, beta, total) x g."""

    @staticmethod
    def forward(ctx, alpha, beta, x):
        ctx.save_for_backward(alpha, beta, x)
        return x.clone()

    @staticmethod
    def backward(ctx, g):
        alpha, beta, x = ctx.saved_tensors
        conc = torch.stack([alpha, beta], -1)
        xs = torch.stack([x, 1 - x], -1)
        d = torch._dirichlet_grad(xs, conc, conc.sum(-1, True).expand_as(conc))
        return d[..., 0] * (1 - x) * g, -d[..., 1] * x * g, None


def conditional_var(cfg, data):
    """ResVAE.decode's concat, residual.py:463-473."""
    parts = []
    for k in cfg.conditional_keys:
        if cfg.discrete_classes is not None and k in cfg.discrete_classes:
            parts.append(F.one_hot(data[k].ravel().long(), len(cfg.discrete_classes[k])))
        else:
            parts.append(data[k])
    return torch.cat(parts, dim=-1)


def decode(sd, cfg, z, data, train, new_stats=None):
    """ResVAE.decode, residual.py:461-491 + ResidualDecoder.forward :288-292."""
    out = {}
    if cfg.conditional_dim > 0:
        out["var"] = conditional_var(cfg, data)
        z = torch.cat([z, out["var"].to(z.dtype)], dim=-1)
    B = z.shape[0]
    x = F.linear(z, sd["decoder.fc_in.weight"], sd["decoder.fc_in.bias"]).reshape(B, cfg.channel[-1], -1)
    for j in range(len(cfg.channel) - 1):
        x = res_block_T(x, sd, f"decoder.res_layers.{j}", cfg, train, new_stats)
    x = torch.tanh(F.conv_transpose1d(x, sd["decoder.conv_out.weight"], sd["decoder.conv_out.bias"], padding=3))
    x_hat = x.moveaxis(-1, 1)
    if cfg.arena_size is None:
        x6d = x_hat
    else:
        x6d = x_hat[..., :-3]
        out["root"] = inv_normalize_root(x_hat[..., -3:], sd["arena_size"]).reshape(B, cfg.window, 3)
    out["x6d"] = x6d.reshape(B, cfg.window, -1, 6)
    return out


def mlp_ensemble(sd, prefix, x):
    """MLPEnsemble.forward, disentangle.py:583-632."""
    def L(n, h):
        return F.linear(h, sd[f"{prefix}.{n}.weight"], sd[f"{prefix}.{n}.bias"])
    a = L("mlp1.4", F.relu(L("mlp1.2", F.relu(L("mlp1.0", x)))))
    b = L("mlp2.2", F.relu(L("mlp2.0", x)))
    c = L("mlp3.4", F.relu(L("mlp3.2", F.relu(L("mlp3.0", x)))))
    d = L("mlp4.4", F.relu(L("mlp4.2", F.relu(L("mlp4.0", x)))))
    return [a, b, c, d]


class _GradReverse(torch.autograd.Function):
    """GradientReversal, disentangle.py:541-556."""

    @staticmethod
    def forward(ctx, x, alpha):
        ctx.alpha = alpha
        return x.view_as(x)

    @staticmethod
    def backward(ctx, g):
        return -ctx.alpha * g, None


def adv_shuffle(mu, var, v_ind, perm):
    """AdvNetScrubber.shuffle, disentangle.py:678-684 with the permutation injected."""
    v_shuffle = var.clone()
    v_shuffle[:, v_ind] = var[perm, v_ind]
    return mu.repeat(2, 1), torch.cat([var, v_shuffle], dim=0)


def adv_forward(sd, feat, z, v):
    """AdvNetScrubber.forward, disentangle.py:673-676."""
    x = torch.cat([z, v.to(z.dtype)], dim=-1)
    return [torch.softmax(y, -1) for y in mlp_ensemble(sd, f"disentangle.adversarial_net.{feat}.ensemble", x)]


def forward(sd, cfg, data, train, eps=None, new_stats=None):
    """VAE.forward, residual.py:318-362.  eps [B,z] is the injected reparameterisation
    noise (reference draws randn_like(mu), :315)."""
    out = encode(sd, cfg, data, train, new_stats)
    if cfg.prior == "beta":
        # residual.py:328-331: z = Beta(alpha, beta).rsample() * 2 - 1, in train AND eval mode; eps = the injected draw in (0, 1)
        z = _BetaRSample.apply(out["alpha"], out["beta"], eps.to(out["alpha"].dtype)) * 2 - 1
    elif train:
        # sampling, residual.py:305-316: (L @ eps[...,None]).squeeze() + mu
        z = torch.matmul(out["L"], eps[..., None]).squeeze().add(out["mu"])
    else:
        z = out["mu"]
    out["z"] = z
    out.update(decode(sd, cfg, z, data, train, new_stats))
    out["disentangle"] = {}
    if "linear" in cfg.method:  # residual.py:339-344
        out["disentangle"]["linear"] = {k: linear_projection(out["mu"], sd[f"disentangle.linear.{k}.decoder.weight"])
                                        for k in cfg.method["linear"]}
    for method, feats in cfg.method.items():
        if method in ("conditional", "linear"):
            continue
        out["disentangle"][method] = {}
        for k in feats:
            # residual.py:351-355: with `linear` configured the heads read the feature's null-space projection
            latent = out["disentangle"]["linear"][k]["z_null"] if "linear" in cfg.method else out["mu"]
            if method == "grad_reversal":
                out["disentangle"][method][k] = mlp_ensemble(
                    sd, f"disentangle.grad_reversal.{k}.reversal.1", _GradReverse.apply(latent, cfg.alpha))
            elif method == "adversarial_net":
                out["disentangle"][method][k] = adv_forward(sd, k, latent, out["var"])
    return out


def linear_projection(z, w):
    """LinearProjection.forward (disentangle.py:727-734) with the textbook projector: v = z W^T and
    z_null = z (I - W^T (W W^T)^-1 W), the component of z orthogonal to the rows of W [out, z]."""
    v = z @ w.T
    proj = torch.eye(w.shape[1], dtype=w.dtype) - w.T @ torch.linalg.inv(w @ w.T) @ w
    return {"v": v, "z_null": z @ proj}


# --------------------------------------------------------------------------- pose maths
def cont6d_to_matrix(c6, eps=0.0):
    """K1: quaternion.py:337-353.  Columns [x y z]."""
    xr, yr = c6[..., 0:3], c6[..., 3:6]
    x = xr / (torch.norm(xr, dim=-1, keepdim=True) + eps)
    z = torch.cross(x, yr, dim=-1)
    z = z / (torch.norm(z, dim=-1, keepdim=True) + eps)
    y = torch.cross(z, x, dim=-1)
    return torch.stack([x, y, z], dim=-1)


def rotation_6d_to_matrix(d6):
    """K2: rotation_conversion.py:469-488.  Rows [b1;b2;b3], F.normalize clamp 1e-12."""
    a1, a2 = d6[..., :3], d6[..., 3:]
    b1 = F.normalize(a1, dim=-1)
    b2 = a2 - (b1 * a2).sum(-1, keepdim=True) * b1
    b2 = F.normalize(b2, dim=-1)
    b3 = torch.cross(b1, b2, dim=-1)
    return torch.stack((b1, b2, b3), dim=-2)


def fwd_kin(c6, tree, offsets, root_pos, eps=0.0):
    """K3: dataset.py:83-116 (do_root_R=True).  c6 [N,J,6], offsets [N,J,3] (or [J,3]),
    root_pos [N,3].  Per chain R restarts from R(joint 0); joints are written in place."""
    if offsets.dim() == 2:
        offsets = offsets.expand(c6.shape[0], -1, -1)
    J = c6.shape[1]
    pose = [None] * J
    pose[0] = root_pos
    for chain in tree:
        R = cont6d_to_matrix(c6[:, 0], eps)
        for i in range(1, len(chain)):
            R = torch.matmul(R, cont6d_to_matrix(c6[:, chain[i]], eps))
            pose[chain[i]] = torch.matmul(R, offsets[:, chain[i]].unsqueeze(-1)).squeeze(-1) + pose[chain[i - 1]]
    zero = torch.zeros_like(root_pos)
    return torch.stack([p if p is not None else zero for p in pose], dim=1)


# --------------------------------------------------------------------------- losses
def prior_loss(mu, L):
    """L3: losses.py:138-146."""
    var = torch.matmul(L, L.transpose(-2, -1))
    kl = -0.5 * torch.sum(1 + 2 * torch.log(L.diagonal(dim1=-1, dim2=-2)) - mu.pow(2) - var.diagonal(dim1=-1, dim2=-2))
    return kl / mu.shape[0]


def mpjpe_loss(pose, x6d_hat, tree, offsets):
    """L1: losses.py:148-171 (root_hat=None -> zeros)."""
    root_hat = torch.zeros_like(pose[..., 0, :])
    ph = fwd_kin(x6d_hat.reshape((-1,) + x6d_hat.shape[-2:]), tree,
                 offsets.reshape((-1,) + offsets.shape[-2:]), root_hat.reshape(-1, 3), eps=1e-8).reshape(pose.shape)
    return torch.sum((pose - ph) ** 2) / (pose.shape[0] * pose.shape[-1] * pose.shape[-2])


def stable_rotation_loss(x, x_hat, eps=1e-7):
    """L4: losses.py:123-136."""
    m1 = rotation_6d_to_matrix(x).reshape(-1, 3, 3)
    m2 = rotation_6d_to_matrix(x_hat).reshape(-1, 3, 3)
    s = torch.linalg.matrix_norm(m2 - m1) / (2 ** 1.5)
    return 2 * torch.asin(torch.clamp(s, -1 + eps, 1 - eps)).sum()


def total_correlation(z, mu, L):
    """L5: losses.py:41-101."""
    logvar = torch.log(torch.matmul(L, L.transpose(-2, -1)).diagonal(dim1=-1, dim2=-2))
    zz, mm, lv = z[:, None].detach(), mu[None, :], logvar[None, :]
    lq = -0.5 * (torch.exp(-lv) * (zz - mm) ** 2 + lv + LN2PI)
    log_qz_product = torch.logsumexp(lq, dim=1).sum(dim=1)
    log_qz = torch.logsumexp(lq.sum(dim=2), dim=1)
    return torch.mean(log_qz - log_qz_product)


def batch_loss(sd, cfg, data, out, loss_scale, adv_perm=None):
    """get_batch_loss, losses.py:182-324.  adv_perm: dict feat -> LongTensor permutation
    injected in place of torch.randperm (disentangle.py:680)."""
    B = data["x6d"].shape[0]
    bl = {}
    if "rotation" in loss_scale:
        bl["rotation"] = stable_rotation_loss(data["x6d"], out["x6d"])
    if "prior" in loss_scale:
        if "L" in out:
            bl["prior"] = prior_loss(out["mu"], out["L"])
        else:  # losses.py:198-206: KL(Beta(alpha, beta) || Beta(1, 1)) summed, / batch
            q = torch.distributions.Beta(out["alpha"], out["beta"])
            p = torch.distributions.Beta(torch.ones_like(out["alpha"]), torch.ones_like(out["beta"]))
            bl["prior"] = torch.distributions.kl_divergence(q, p).sum(-1).sum() / B
    if "jpe" in loss_scale:
        bl["jpe"] = mpjpe_loss(data["target_pose"], out["x6d"], cfg.kinematic_tree, data["offsets"])
    if "root" in loss_scale:
        bl["root"] = torch.sum((out["root"] - data["root"]) ** 2) / B
    for method, keys in cfg.method.items():
        nk = len(keys)
        for key in keys:
            if method == "grad_reversal":
                acc = 0
                ens = out["disentangle"][method][key]
                for e in ens:
                    if key == "ids":
                        acc = acc + F.cross_entropy(e, data[key].ravel().long(), reduction="sum")
                    else:
                        acc = acc + torch.sum((e - data[key]) ** 2)
                    acc = acc / len(ens) / nk / B  # normalisation inside the loop (losses.py:279-284)
                bl[key + "_gr"] = acc
            if method == "adversarial_net":
                v_ind = cfg.features.index(key)
                z_aug, v_aug = adv_shuffle(out["mu"], out["var"].to(out["mu"].dtype), v_ind, adv_perm[key])
                y_pred = adv_forward(sd, key, z_aug, v_aug)
                y = torch.tensor([0, 1])[:, None].repeat(1, B).ravel()
                y = F.one_hot(y, 2).to(out["mu"].dtype)
                acc = 0
                for ye in y_pred:
                    acc = acc + F.cross_entropy(ye, y, reduction="sum")  # CE on softmax output (quirk)
                bl[key + "_an"] = acc / (-(len(y_pred) * B))
            if method == "linear":  # losses.py:258-265
                bl[key + "_lin"] = torch.sum((out["disentangle"]["linear"][key]["v"] - data[key]) ** 2) / nk / B
    if "total_correlation" in loss_scale:
        bl["total_correlation"] = total_correlation(out["z"], out["mu"], out["L"])
    bl["total"] = sum(loss_scale[k] * bl[k] for k in list(bl.keys()) if loss_scale[k] != 0)
    return bl


# --------------------------------------------------------------------------- optimizer
def adamw_step(params, grads, state, lr, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.01, decoupled=True):
    """torch.optim.AdamW / Adam defaults (trainer.py:60-65).  state: dict name -> (m, v),
    plus state['step']."""
    state["step"] = state.get("step", 0) + 1
    t = state["step"]
    b1, b2 = betas
    out = {}
    for n, p in params.items():
        g = grads[n]
        m, v = state.get(n, (torch.zeros_like(p), torch.zeros_like(p)))
        if decoupled:
            p = p * (1 - lr * weight_decay)
        elif weight_decay != 0:
            g = g + weight_decay * p
        m = b1 * m + (1 - b1) * g
        v = b2 * v + (1 - b2) * g * g
        bc1, bc2 = 1 - b1 ** t, 1 - b2 ** t
        denom = v.sqrt() / math.sqrt(bc2) + eps
        out[n] = p - (lr / bc1) * m / denom
        state[n] = (m, v)
    return out


def train_step(sd, cfg, data, loss_scale, eps, adv_perm=None, lr=1e-4, opt_state=None,
               optimizer="adamw"):
    """One trainer.py:126-167 iteration: forward, losses, backward, (no-op clip), step.
    Returns (losses, grads, new_sd, out)."""
    names = trainable_names(sd)
    leaf = {n: sd[n].detach().clone().requires_grad_(True) for n in names}
    work = dict(sd)
    work.update(leaf)
    new_stats = {}
    out = forward(work, cfg, data, True, eps=eps, new_stats=new_stats)
    bl = batch_loss(work, cfg, data, out, loss_scale, adv_perm)
    bl["total"].backward()
    grads = {n: (leaf[n].grad if leaf[n].grad is not None else torch.zeros_like(leaf[n])) for n in names}
    opt_state = {} if opt_state is None else opt_state
    with torch.no_grad():
        newp = adamw_step({n: leaf[n].detach() for n in names}, grads, opt_state, lr,
                          weight_decay=0.01 if optimizer == "adamw" else 0.0,
                          decoupled=(optimizer == "adamw"))
    new_sd = dict(sd)
    new_sd.update(newp)
    new_sd.update(new_stats)
    return {k: v.detach() for k, v in bl.items()}, grads, new_sd, out


# --------------------------------------------------------------------------- synthetic data
def synth_batch(cfg: OracleConfig, B, seed=0, dtype=torch.float32, offsets_table=None):
    """SURVEY 8d synthetic inputs.  target_pose = FK(x6d, offsets, root=0, eps=1e-8)."""
    g = torch.Generator().manual_seed(1000 + seed)
    J, W = cfg.n_keypts, cfg.window
    x6d = torch.randn(B, W, J, 6, generator=g, dtype=torch.float64)
    root = torch.rand(B, W, 3, generator=g, dtype=torch.float64) * 2 - 1
    if offsets_table is None:
        offsets_table = skeleton_offsets(J)
    offs = torch.tensor(offsets_table, dtype=torch.float64)
    seg = 0.5 + torch.rand(J, generator=g, dtype=torch.float64)
    offsets = (offs * seg[:, None])[None, None].expand(B, W, J, 3).contiguous()
    tp = fwd_kin(x6d.reshape(-1, J, 6), cfg.kinematic_tree, offsets.reshape(-1, J, 3),
                 torch.zeros(B * W, 3, dtype=torch.float64), eps=1e-8).reshape(B, W, J, 3)
    data = {"x6d": x6d, "root": root, "offsets": offsets, "target_pose": tp,
            "avg_speed_3d": torch.randn(B, 3, generator=g, dtype=torch.float64),
            "heading": F.normalize(torch.randn(B, 2, generator=g, dtype=torch.float64), dim=-1)}
    data = {k: v.to(dtype) for k, v in data.items()}
    data["ids"] = torch.randint(0, 4, (B, 1), generator=g).to(torch.int16)
    return data


def skeleton_offsets(J):
    if J == 18:
        return MOUSE_OFFSET
    if J == 23:
        return MOUSE_OFFSET + [[1, 0, 0], [0, 1, 0], [0, -1, 0], [0, 1, 0], [0, -1, 0]]
    return [[0, 0, 0]] + [[1, 0, 0]] * (J - 1)


def skeleton_tree(J):
    """J=18: reference tree.  J=23 (BASELINE's synthetic 23-joint skeleton; build-defined,
    SURVEY 8d): the five extra joints 18..22 appended to chains 0,2,3,4,5."""
    if J == 18:
        return [list(c) for c in MOUSE_KINEMATIC_TREE]
    if J == 23:
        t = [list(c) for c in MOUSE_KINEMATIC_TREE]
        t[0].append(18); t[2].append(19); t[3].append(20); t[4].append(21); t[5].append(22)
        return t
    return [list(range(J))]


# ------------------------------------------------------------------ eval forward
GEN_SPD_STD = (0.4038, 0.3586, 0.4169)   # constants of eval/eval.py:43-57,106-113
GEN_SPD_MEAN = (0.4993, 0.7112, 0.6663)
GEN_SPD_MIN = (-1.2323, -1.9734, -1.5858)
GEN_SPD_MAX = (4.6167, 4.6437, 4.2551)
GEN_PARTS = ([0, 1, 2, 3, 4, 5], [1, 6, 7, 8, 9, 10, 11], [5, 12, 13, 14, 15, 16, 17])


def generative_restrictiveness(sd, cfg, z, data, key, rand):
    """eval/eval.py:22-120: re-decode z with a re-drawn conditional variable `key` and measure the
    variable on the generated pose.  rand = the uniform [B] (heading, torch.rand at :29) or normal
    [B,1] (avg_speed_3d, torch.randn at :46) draw.  Returns (pred, target, data_mod) where target is the
    REPLACED data[key] (the reference mutates `data` and returns the new value, :120)."""
    import math
    data = dict(data)
    dt = z.dtype
    B, W, J = data["x6d"].shape[0], data["x6d"].shape[1], data["x6d"].shape[-2]
    if key == "heading":
        yaw = (rand.to(dt) * 2 - 1)[:, None] * math.pi
        data["heading"] = torch.cat([torch.sin(yaw), torch.cos(yaw)], dim=-1)
    elif key == "avg_speed_3d":
        jitter = rand.to(dt) * torch.tensor(GEN_SPD_STD, dtype=dt) * 1.5 + 0.5
        data["avg_speed_3d"] = torch.clamp(data[key] + jitter, min=torch.tensor(GEN_SPD_MIN, dtype=dt),
                                           max=torch.tensor(GEN_SPD_MAX, dtype=dt))
    out = decode(sd, cfg, z, data, False)
    pose = fwd_kin(out["x6d"].reshape(-1, J, 6), cfg.kinematic_tree, data["offsets"].reshape(-1, J, 3).to(dt),
                   out["root"].reshape(-1, 3), eps=1e-8).reshape(B, W, J, 3)
    if key == "heading":
        fwd = pose[:, W // 2, 1, :] - pose[:, W // 2, 0, :]
        fwd = fwd / torch.linalg.norm(fwd, dim=-1)[..., None]
        yaw = -torch.arctan2(fwd[:, 1], fwd[:, 0])[:, None]
        pred = torch.cat([torch.sin(yaw), torch.cos(yaw)], dim=-1)
    else:
        root_spd = torch.sqrt((torch.diff(pose[:, :, 0, :], n=1, dim=-2) ** 2).sum(dim=-1)).mean(dim=-1)
        dxyz = []
        for part in GEN_PARTS:
            rel = pose - pose[:, W // 2, part[0], :][:, None, None, :]
            d = (torch.diff(rel[..., part[1:], :], n=1, dim=-3) ** 2).sum(dim=-1)
            dxyz.append(torch.sqrt(d).mean(dim=(-1, -2)))
        pred = torch.stack([root_spd, dxyz[0], (dxyz[1] + dxyz[2]) / 2], dim=-1)
        pred = (pred - torch.tensor(GEN_SPD_MEAN, dtype=dt)) / torch.tensor(GEN_SPD_STD, dtype=dt)
    return pred, data[key], data


# ------------------------------------------------------- streaming scrubber
def mals_init(nx, ny, bias=False, lamdiff=1e-1, dtype=torch.float32, polynomial_order=1):
    """MovingAvgLeastSquares.__init__, disentangle.py:393-438: the design has C(nx+d-1, d) columns per degree d."""
    n = sum(math.comb(nx + d - 1, d) for d in range(1, polynomial_order + 1)) + int(bias)
    return {"Sxx0": torch.eye(n, dtype=dtype), "Sxy0": torch.zeros(n, ny, dtype=dtype), "Sxx1": torch.eye(n, dtype=dtype),
            "Sxy1": torch.zeros(n, ny, dtype=dtype), "lam0": torch.tensor([0.9], dtype=dtype),
            "lam1": torch.tensor([0.9], dtype=dtype) + lamdiff, "bias": bias, "lamdiff": lamdiff, "poly": polynomial_order}


def _mals_design(st, x):
    """polynomial_expansion (disentangle.py:440-464) + bias column: degree-d block = products over every multiset of d
    latent dimensions (itertools order == torch.combinations(with_replacement=True)), times nx / (block width)."""
    import itertools
    nx = x.shape[1]
    cols = [x]
    for d in range(2, st.get("poly", 1) + 1):
        combos = list(itertools.combinations_with_replacement(range(nx), d))
        block = torch.stack([torch.stack([x[:, i] for i in c], 0).prod(0) for c in combos], 1)
        cols.append(block / len(combos) * nx)
    x = torch.column_stack(cols)
    return torch.column_stack((x, torch.ones(x.shape[0], 1, dtype=x.dtype))) if st["bias"] else x


def mals_forward(st, x, l2_reg=0.0):
    """disentangle.py:467-492."""
    x = _mals_design(st, x)
    l2 = torch.ones(x.shape[1], dtype=x.dtype) * l2_reg
    if st["bias"]:
        l2[-1] = 0
    W0 = torch.linalg.solve(st["Sxx0"].diagonal_scatter(st["Sxx0"].diagonal() + l2), st["Sxy0"])
    W1 = torch.linalg.solve(st["Sxx1"].diagonal_scatter(st["Sxx1"].diagonal() + l2), st["Sxy1"])
    return [x @ W0, x @ W1]


def mals_update(st, x, y):
    """disentangle.py:494-506 (returns the new state)."""
    x = _mals_design(st, x)
    xx, xy = (x.T @ x).detach(), (x.T @ y).detach()
    st = dict(st)
    st["Sxx0"], st["Sxy0"] = st["lam0"] * st["Sxx0"] + xx, st["lam0"] * st["Sxy0"] + xy
    st["Sxx1"], st["Sxy1"] = st["lam1"] * st["Sxx1"] + xx, st["lam1"] * st["Sxy1"] + xy
    return st


def mals_loss(st, yhat0, yhat1, y, delta=1e-4):
    """disentangle.py:508-538: returns ((l0 + l1) / 2, new state with the forgetting factors moved)."""
    l0, l1 = ((y - yhat0) ** 2).sum(), ((y - yhat1) ** 2).sum()
    st = dict(st)
    if l0 < l1:
        st["lam0"] = torch.clamp(st["lam0"] - delta, 0.0, 1.0)
        st["lam1"] = st["lam0"] + st["lamdiff"]
    else:
        st["lam1"] = torch.clamp(st["lam1"] + delta, 0.0, 1.0)
        st["lam0"] = st["lam1"] - st["lamdiff"]
    return (l0 + l1) * 0.5, st


def maf_init(nx, n_classes, lamdiff=1e-2, dtype=torch.float32):
    """MovingAverageFilter.__init__, disentangle.py:15-29."""
    lam1 = torch.ones(n_classes, dtype=dtype) * 0.5
    return {"m1": torch.zeros(n_classes, nx, dtype=dtype), "m2": torch.zeros(n_classes, nx, dtype=dtype), "lam1": lam1,
            "lam2": lam1 + lamdiff, "lamdiff": lamdiff}


def maf_loss(st, x, y, classes, delta=1e-3):
    """disentangle.py:34-75: returns (loss, state with the forgetting factors moved); m1 / m2 themselves only change in
    maf_update."""
    st = {k: (v.clone() if torch.is_tensor(v) else v) for k, v in st.items()}
    m1, m2 = torch.zeros_like(st["m1"]), torch.zeros_like(st["m2"])
    for i, label in enumerate(classes):
        xbar = torch.mean(x[(y == label).ravel(), :], dim=0)
        d1, d2 = torch.linalg.norm(xbar - st["m1"][i]), torch.linalg.norm(xbar - st["m2"][i])
        if d1 < d2:
            st["lam1"][i] = torch.clamp(st["lam1"][i] - delta, 0.0, 1.0)
            st["lam2"][i] = st["lam1"][i] + st["lamdiff"]
        else:
            st["lam2"][i] = torch.clamp(st["lam2"][i] + delta, 0.0, 1.0)
            st["lam1"][i] = st["lam2"][i] - st["lamdiff"]
        m1[i] = (1 - st["lam1"][i]) * xbar + st["lam1"][i] * st["m1"][i]
        m2[i] = (1 - st["lam2"][i]) * xbar + st["lam2"][i] * st["m2"][i]
    est = 0.5 * (m1 + m2)
    d = torch.triu(est.T[..., None] - est.T[..., None, :], diagonal=1)
    return torch.linalg.norm(d), st


def maf_update(st, x, y, classes):
    """disentangle.py:77-87."""
    st = {k: (v.clone() if torch.is_tensor(v) else v) for k, v in st.items()}
    for i, label in enumerate(classes):
        xbar = torch.mean(x[(y == label).ravel(), :], dim=0)
        st["m1"][i] = (1 - st["lam1"][i]) * xbar + st["lam1"][i] * st["m1"][i]
        st["m2"][i] = (1 - st["lam2"][i]) * xbar + st["lam2"][i] * st["m2"][i]
    return st


def qda_init(nx, n_classes, lamdiff=1e-2, dtype=torch.float32):
    """QuadraticDiscriminantFilter.__init__, disentangle.py:97-125."""
    st = {"lamdiff": lamdiff}
    for name in ("0a", "1a", "0b", "1b"):
        st["m" + name] = torch.zeros(n_classes, nx, dtype=dtype)
        st["S" + name] = torch.eye(nx, dtype=dtype)[None].repeat(n_classes, 1, 1)
    st["lama"] = torch.ones(n_classes, dtype=dtype) * 0.2
    st["lamb"] = st["lama"] + lamdiff
    return st


def _cgll(x, m, S):
    """disentangle.py:130-135."""
    resids = torch.sum((x - m) * torch.linalg.solve(S, (x - m).T).T, dim=1)
    return -0.5 * (torch.logdet(S) + resids)


def qda_update(st, x, y, classes):
    """disentangle.py:137-165."""
    st = {k: (v.clone() if torch.is_tensor(v) else v) for k, v in st.items()}
    for i, label in enumerate(classes):
        i0, i1 = (y != label).ravel(), (y == label).ravel()
        x0m, x1m = torch.mean(x[i0], dim=0, keepdim=True), torch.mean(x[i1], dim=0, keepdim=True)
        x0S, x1S = torch.cov(x[i0].T, correction=0), torch.cov(x[i1].T, correction=0)
        for tag, lam in (("a", st["lama"][i]), ("b", st["lamb"][i])):
            st["m0" + tag][i] = (1 - lam) * st["m0" + tag][i] + lam * x0m
            st["m1" + tag][i] = (1 - lam) * st["m1" + tag][i] + lam * x1m
            st["S0" + tag][i] = (1 - lam) * st["S0" + tag][i] + lam * x0S
            st["S1" + tag][i] = (1 - lam) * st["S1" + tag][i] + lam * x1S
    return st


def qda_loss(st, x, y, classes, delta=1e-3, update=True):
    """disentangle.py:167-232: returns (loss, state with the forgetting factors moved)."""
    st = {k: (v.clone() if torch.is_tensor(v) else v) for k, v in st.items()}
    ll_loss = 0
    for i, label in enumerate(classes):
        i0, i1 = (y != label).ravel(), (y == label).ravel()
        lla0, lla1 = _cgll(x, st["m0a"][i: i + 1], st["S0a"][i]), _cgll(x, st["m1a"][i: i + 1], st["S1a"][i])
        llb0, llb1 = _cgll(x, st["m0b"][i: i + 1], st["S0b"][i]), _cgll(x, st["m1b"][i: i + 1], st["S1b"][i])
        lla, llb = torch.sum(i0 * lla0 + i1 * lla1), torch.sum(i0 * llb0 + i1 * llb1)
        if update and (lla > llb):
            st["lama"][i] = torch.clamp(st["lama"][i] - delta, 0.0, 1.0)
            st["lamb"][i] = st["lama"][i] + st["lamdiff"]
        elif update:
            st["lamb"][i] = torch.clamp(st["lamb"][i] + delta, 0.0, 1.0)
            st["lama"][i] = st["lamb"][i] - st["lamdiff"]
        batch_y = (i1 * 2 - 1).to(x.dtype)
        ll_loss = ll_loss + (batch_y @ (lla1 - lla0) + batch_y @ (llb1 - llb0)) * 0.5
    return ll_loss / len(classes), st


def direct_lsq_loss(zmat, y, bias=False):
    """losses.py:173-179."""
    if bias:
        zmat = torch.column_stack((zmat, torch.ones(zmat.shape[0], 1, dtype=zmat.dtype)))
    yhat = zmat @ torch.linalg.solve(zmat.T @ zmat, zmat.T @ y)
    return ((yhat - y) ** 2).sum()


def mcmi_value(x_s, y_s, var_s, bandwidth, x, y):
    """MutInfoEstimator (disentangle.py:234-317) as one numpy float64 expression.  x_s [S, dx], y_s [S, dy]: mixture
    centres; var_s: [1] ("sphere", = bandwidth) or [S, dx] ("diagonal", = diag(L_s)^2 + bandwidth); x [B, dx], y [B, dy].
    Returns mean_b[lse_s log N(x_b,y_b | s) - lse_s log N(x_b | s) - lse_s log N(y_b | s)] with the reference's
    normalisation: log A_x = dx*(log 2pi + log var) (sphere) or dx*log 2pi + sum log var_s (diagonal), log A_y =
    dy*(log 2pi + log gamma), each density = -0.5*(log A + squared distance); log-sum-exp, not log-mean-exp
    (disentangle.py:297-317)."""
    import numpy as np
    from scipy.special import logsumexp
    x_s, y_s, var_s, x, y = (np.asarray(a, dtype=np.float64) for a in (x_s, y_s, var_s, x, y))
    dx, dy = x_s.shape[1], y_s.shape[1]
    log2pi = np.log(2 * np.pi)
    if var_s.ndim == 1:
        logA_x = np.full((1, 1), dx * (log2pi + np.log(var_s[0])))
        var = var_s[0]
    else:
        logA_x = (dx * log2pi + np.log(var_s).sum(-1))[None, :]
        var = var_s[None, :, :]
    logA_y = dy * (log2pi + np.log(bandwidth))
    ex = x[:, None, :] - x_s[None, :, :]
    ey = y[:, None, :] - y_s[None, :, :]
    sdx = (ex * ex / var).sum(-1)
    sdy = (ey * ey / bandwidth).sum(-1)
    pxy = logsumexp(-0.5 * (logA_x + logA_y + sdx + sdy), axis=-1)
    px = logsumexp(-0.5 * (logA_x + sdx), axis=-1)
    py = logsumexp(-0.5 * (logA_y + sdy), axis=-1)
    return float((pxy - px - py).mean())

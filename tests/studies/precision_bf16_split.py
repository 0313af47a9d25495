"""Precision study (CPU): what the SC-VAE step loses when every conv / linear contraction is
computed as a split-bf16 product (operands rounded to NT bf16 pieces, NP cross products kept,
fp32 accumulation) instead of fp32 -- the arithmetic a bf16-MFMA version of the GEMM kernels
would perform.  Drives the oracle on the committed fixtures; not collected by pytest.

    python tests/studies/precision_bf16_split.py
"""
import os
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from oracle import scvae_oracle as O  # noqa: E402
from tests.test_oracle_golden import load_fixture, rel  # noqa: E402

# wgrad / dgrad: optional (pieces, products) for the weight-gradient / data-gradient contractions only
MODE = {"pieces": 2, "products": 3, "wgrad": None, "dgrad": None, "fwd_b_pieces": None}  # fwd_b_pieces: weight pieces of the forward only


def split(x, n):
    out, r = [], x
    for _ in range(n):
        p = r.to(torch.bfloat16).to(torch.float32)
        out.append(p)
        r = r - p
    return out


def pairs():
    n, p = MODE["pieces"], MODE["products"]
    order = sorted(((i, j) for i in range(n) for j in range(n)), key=lambda t: (t[0] + t[1], t))
    return order[:p]


def contract(fn, a, b, wgrad=False, dgrad=False):
    pieces, products = MODE["pieces"], MODE["products"]
    if wgrad and MODE["wgrad"] is not None:
        pieces, products = MODE["wgrad"]
    if dgrad and MODE["dgrad"] is not None:
        pieces, products = MODE["dgrad"]
    if pieces == 0:
        return fn(a, b)
    pb = pieces
    if not wgrad and not dgrad and MODE.get("fwd_b_pieces"):
        pb = MODE["fwd_b_pieces"]  # forward only: fewer weight pieces (a0..a2 x b0..b1, terms with i + j <= 2)
    A, B = split(a, pieces), split(b, pb)
    order = sorted(((i, j) for i in range(pieces) for j in range(pb)), key=lambda t: (t[0] + t[1], t))
    order = [t for t in order if t[0] + t[1] <= pieces - 1] if pb != pieces else order[:products]
    acc = None
    for i, j in reversed(order):  # small terms first
        t = fn(A[i], B[j])
        acc = t if acc is None else acc + t
    return acc


class Conv(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w, kind, stride, padding):
        ctx.save_for_backward(x, w)
        ctx.meta = (kind, stride, padding)
        f = F.conv1d if kind == "conv" else F.conv_transpose1d
        return contract(lambda a, b: f(a, b, None, stride=stride, padding=padding), x, w)

    @staticmethod
    def backward(ctx, g):
        x, w = ctx.saved_tensors
        kind, stride, padding = ctx.meta
        f = F.conv1d if kind == "conv" else F.conv_transpose1d

        def dx(gg, ww):
            xx = torch.zeros_like(x, requires_grad=True)
            with torch.enable_grad():
                y = f(xx, ww, None, stride=stride, padding=padding)
            return torch.autograd.grad(y, xx, gg)[0]

        def dw(gg, xx):
            ww = torch.zeros_like(w, requires_grad=True)
            with torch.enable_grad():
                y = f(xx, ww, None, stride=stride, padding=padding)
            return torch.autograd.grad(y, ww, gg)[0]

        return contract(dx, g, w, dgrad=True), contract(dw, g, x, wgrad=True), None, None, None


class Lin(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w):
        ctx.save_for_backward(x, w)
        return contract(lambda a, b: a @ b.t(), x, w)

    @staticmethod
    def backward(ctx, g):
        x, w = ctx.saved_tensors
        return contract(lambda a, b: a @ b, g, w, dgrad=True), contract(lambda a, b: a.t() @ b, g, x, wgrad=True)


_c1, _ct, _li = F.conv1d, F.conv_transpose1d, F.linear


def conv1d(x, w, b=None, stride=1, padding=0):
    y = Conv.apply(x, w, "conv", stride, padding)
    return y if b is None else y + b[None, :, None]


def conv_transpose1d(x, w, b=None, stride=1, padding=0):
    y = Conv.apply(x, w, "convT", stride, padding)
    return y if b is None else y + b[None, :, None]


def linear(x, w, b=None):
    y = Lin.apply(x, w)
    return y if b is None else y + b


class patched:
    def __enter__(self):
        O.F = type("Fp", (), {})()
        for k in dir(F):
            if not k.startswith("__"):
                setattr(O.F, k, getattr(F, k))
        O.F.conv1d, O.F.conv_transpose1d, O.F.linear = conv1d, conv_transpose1d, linear

    def __exit__(self, *a):
        O.F = F


def run(name, golden):
    fx, cfg, loss_scale, opt, sd, data = load_fixture(golden, name)
    eps, perm = torch.from_numpy(fx["eps/0"]), torch.from_numpy(fx["perm/0"])
    advp = {k: perm for k in cfg.method.get("adversarial_net", [])}
    base = O.train_step(sd, cfg, data, loss_scale, eps, adv_perm=advp, lr=1e-4, optimizer=opt)
    sd64 = {k: (v.double() if v.is_floating_point() else v) for k, v in sd.items()}
    d64 = {k: (v.double() if v.is_floating_point() else v) for k, v in data.items()}
    cfg64 = cfg
    try:
        import dataclasses
        cfg64 = dataclasses.replace(cfg, arena_size=cfg.arena_size.double())
    except Exception:
        pass
    ref = O.train_step(sd64, cfg64, d64, loss_scale, eps.double(), adv_perm=advp, lr=1e-4, optimizer=opt)
    rows = []
    for label, mode in (("fp32", (0, 0)), ("bf16x1", (1, 1)), ("bf16x3", (2, 3)), ("bf16x4", (2, 4)), ("bf16x6", (3, 6)),
                        ("x6+w:x3", (3, 6, (2, 3))), ("x6+w:x1", (3, 6, (1, 1))), ("x6+b:x3", (3, 6, (2, 3), (2, 3))),
                        ("x6+b:x4", (3, 6, (2, 3), (2, 4))), ("f5+b:x3", (3, 6, (2, 3), (2, 3), 2))):
        MODE["pieces"], MODE["products"] = mode[:2]
        MODE["wgrad"] = mode[2] if len(mode) > 2 else None
        MODE["dgrad"] = mode[3] if len(mode) > 3 else None
        MODE["fwd_b_pieces"] = mode[4] if len(mode) > 4 else None
        if mode[0] == 0:
            got = base
        else:
            with patched():
                got = O.train_step(sd, cfg, data, loss_scale, eps, adv_perm=advp, lr=1e-4, optimizer=opt)
        bl, grads, new_sd, out = got
        rbl, rgrads, rsd, rout = ref
        lt = abs(float(bl["total"]) - float(rbl["total"])) / abs(float(rbl["total"]))
        lmax = max(abs(float(bl[k]) - float(rbl[k])) / (abs(float(rbl[k])) + 1e-30) for k in rbl)
        mu = rel(out["mu"].detach(), rout["mu"].detach())
        xh = rel(out["x6d"].detach(), rout["x6d"].detach())
        gmax = max(float(g.abs().max()) for g in rgrads.values())
        gerr = max(float((grads[k].double() - rgrads[k]).abs().max() / max(float(rgrads[k].abs().max()), 1e-3 * gmax))
                   for k in rgrads)
        rows.append((label, lt, lmax, mu, xh, gerr))
    print(f"== {name}")
    print(f"{'mode':8s} {'total':>10s} {'worst term':>10s} {'mu':>10s} {'x6d_hat':>10s} {'grads':>10s}   (max-norm relative to the fp64 oracle)")
    for r in rows:
        print(f"{r[0]:8s} " + " ".join(f"{v:10.2e}" for v in r[1:]))


if __name__ == "__main__":
    golden = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "golden")
    torch.manual_seed(0)
    for name in sys.argv[1:] or ["vanilla_tiny", "full_tiny", "vanilla_default_B4"]:
        run(name, golden)

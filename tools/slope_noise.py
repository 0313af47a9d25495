"""PReLU-slope gradients of a reference fixture on the HIP path (every layer forced onto the split kernels, tiles tuned on the fly) against the
fp64 oracle, next to the fp32 oracle's own error and the reference's stored value: python tools/slope_noise.py [fixture] [precision] [repeats]"""
import dataclasses, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from oracle import scvae_oracle as O
from tests.test_oracle_golden import load_fixture, ARENA
from tests.test_gpu_model import build_model, to_dev
from scrubvae_amd import ops
from scrubvae_amd.train.losses import get_batch_loss

name = sys.argv[1] if len(sys.argv) > 1 else "vanilla_default_j23_B4"
prec = sys.argv[2] if len(sys.argv) > 2 else "f16x3b3"
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 2
fx, cfg, loss_scale, opt, sd, data = load_fixture(os.path.join(ROOT, "tests", "golden"), name)
eps, perm = torch.from_numpy(fx["eps/0"]), torch.from_numpy(fx["perm/0"])
ap = {k: perm for k in cfg.method.get("adversarial_net", [])}
_, g32, _, _ = O.train_step(sd, cfg, data, loss_scale, eps, adv_perm=ap)
c64 = dataclasses.replace(cfg, arena_size=ARENA.double())
sd64 = {k: (v.double() if v.dtype.is_floating_point else v) for k, v in sd.items()}
d64 = {k: (v.double() if v.dtype.is_floating_point else v) for k, v in data.items()}
_, g64, _, _ = O.train_step(sd64, c64, d64, loss_scale, eps.double(), adv_perm=ap)
slopes = [n for n, t in g64.items() if t.numel() == 1]
ops.set_precision(prec)
ops.SPLIT_MIN_FLOPS = 0.0
runs = []
_st, _dt = ops.Conv.stats_tiles, ops.Conv.dgrad_stats_tiles
MODES = {"F": (True, True), "S": (False, False), "f": (True, False), "b": (False, True), "i": (True, False)}  # (forward statistics, backward sums) from the GEMM epilogues
from scrubvae_amd.model.residual import ResVAE
_cfb = ResVAE._conv_fwd_bn
def _ignore(self, *a, **k):  # mode "i": the fused-statistics launch runs, but BatchNorm takes its statistics from the separate pass
    _cfb(self, *a, **k)
    return None
MODES["E"] = (False, False)
_bna = ResVAE._bn_act
def _exact(self, tag, x, bn, act, rows, out, stats=None):  # mode "E": coefficients from the fp64 statistics of the stored tensor, rounded once
    r = _bna(self, tag, x, bn, act, rows, out, stats)
    if self.training:
        from scrubvae_amd.ops import pad16
        Cp = pad16(bn.c)
        xx = x[:rows].double()
        mean, var = xx.mean(0), xx.var(0, unbiased=False)
        rstd = 1.0 / torch.sqrt(var + bn.eps)
        g, b = bn.weight.double(), bn.bias.double()
        self._buf(tag + ".mean", (Cp,)).copy_(mean.float())
        self._buf(tag + ".rstd", (Cp,)).copy_(rstd.float())
        self._buf(tag + ".scale", (Cp,)).copy_((g * rstd).float())
        self._buf(tag + ".shift", (Cp,)).copy_((b - mean * g * rstd).float())
        ops.affine_prelu_fwd(x, self._buf(tag + ".scale", (Cp,)), self._buf(tag + ".shift", (Cp,)), act.weight, out, rows, Cp, Cp)
    return r
PERT = {"p": lambda C: torch.full((C,), 3e-7, dtype=torch.float64), "m": lambda C: torch.full((C,), -3e-7, dtype=torch.float64),
        "r": lambda C: (torch.randint(0, 2, (C,), generator=torch.Generator().manual_seed(C)).double() * 2 - 1) * 3e-7}
def _perturbed(kind):  # exact statistics, then every channel's mean moved by 3e-7 standard deviations (coherently + / - or with random signs)
    def f(self, tag, x, bn, act, rows, out, stats=None):
        r = _bna(self, tag, x, bn, act, rows, out, stats)
        if self.training:
            from scrubvae_amd.ops import pad16
            Cp = pad16(bn.c)
            xx = x[:rows].double()
            mean, var = xx.mean(0), xx.var(0, unbiased=False)
            rstd = 1.0 / torch.sqrt(var + bn.eps)
            mean = mean + PERT[kind](Cp).to(mean.device) / rstd
            g, b = bn.weight.double(), bn.bias.double()
            self._buf(tag + ".mean", (Cp,)).copy_(mean.float())
            self._buf(tag + ".rstd", (Cp,)).copy_(rstd.float())
            self._buf(tag + ".scale", (Cp,)).copy_((g * rstd).float())
            self._buf(tag + ".shift", (Cp,)).copy_((b - mean * g * rstd).float())
            ops.affine_prelu_fwd(x, self._buf(tag + ".scale", (Cp,)), self._buf(tag + ".shift", (Cp,)), act.weight, out, rows, Cp, Cp)
        return r
    return f
for k in PERT:
    MODES[k] = (False, False)
for fuse in ["F"] * reps + ["S", "f", "E", "p", "m", "r"]:
    fw, bw = MODES[fuse]
    ResVAE._conv_fwd_bn = _ignore if fuse == "i" else _cfb
    ResVAE._bn_act = _exact if fuse == "E" else (_perturbed(fuse) if fuse in PERT else _bna)
    ops.Conv.stats_tiles = _st if fw else (lambda self: 0)
    ops.Conv.dgrad_stats_tiles = _dt if bw else (lambda self: (0, 0))
    model, dis = build_model(cfg, sd)
    model.train()
    d = to_dev(data)
    d["eps"] = eps.cuda()
    bl = get_batch_loss(model, d, model(d), loss_scale, dis, adv_perm=ap)
    bl["total"].backward()
    torch.cuda.synchronize()
    runs.append(({k: v.cpu() for k, v in model.grads_state_dict().items()}, fuse))
    runs[-1][0]["__par"] = {n: p.detach().double().cpu().clone() for n, p in model.named_parameters() if "decoder.res_layers.3.add" in n}
    runs[-1][0]["__all"] = {k[0]: t.double().cpu().clone() for k, t in model._ws.items() if isinstance(k, tuple) and isinstance(k[0], str) and t.numel() < 4e6}
    runs[-1][0]["__bn"] = {k[0]: t.double().cpu().clone() for k, t in model._ws.items()
                           if isinstance(k, tuple) and isinstance(k[0], str) and k[0].endswith((".mean", ".rstd", ".scale", ".shift"))}
    if len(runs) == 1:  # every later build takes the SAME kernels: only the fusion switch differs between the columns
        ops.TILE_TABLE.update(ops.TUNED_LOG)
        for key, cv in model._convs.items():
            if "dgrad" in cv.__dict__.get("_tuned", ()):
                print("dgrad", key[0], cv.kernel_name("dgrad"), "fused-epilogue-capable" if cv.dgrad_stats_tiles()[0] else "")
    del model
print(f"{name} {prec}: relative deviation of each PReLU-slope gradient from the fp64 oracle")
print(f"{'tensor':44s} {'g64':>11s} {'cpu32':>9s} {'ref fx':>9s} " + " ".join(f"hip{f}{i:<5d}" for i, (_, f) in enumerate(runs)))
for n in slopes:
    t = float(g64[n])
    fxv = float(fx["s0/gradnorm/" + n]) if ("s0/gradnorm/" + n) in fx.files else (float(abs(torch.from_numpy(fx["s0/grad/" + n]).item())) if ("s0/grad/" + n) in fx.files else float("nan"))
    print(f"{n:44s} {t:+11.4e} {abs(float(g32[n]) - t) / abs(t):9.2e} {abs(fxv - abs(t)) / abs(t):9.2e} " +
          " ".join(f"{abs(float(g[n]) - t) / abs(t):9.2e}" for g, _ in runs))

bn_f = next(g["__bn"] for g, f in runs if f == "f")
bn_s = next(g["__bn"] for g, f in runs if f == "S")
print("BatchNorm coefficients, epilogue statistics (f) vs separate pass (S): max over channels")
for nm in sorted(bn_f):
    a, b = bn_f[nm], bn_s[nm]
    print(f"  {nm:20s} max |f - S| {float((a - b).abs().max()):.3e}   max |S| {float(b.abs().max()):.3e}")

af = next(g["__all"] for g, f in runs if f == "f")
a_s = next(g["__all"] for g, f in runs if f == "S")
print("every workspace buffer, f vs S: max |f - S| / max |S| (only > 2e-6)")
for nm in af:
    if nm in a_s and af[nm].shape == a_s[nm].shape and a_s[nm].numel() > 0:
        den = float(a_s[nm].abs().max())
        if den > 0:
            r = float((af[nm] - a_s[nm]).abs().max()) / den
            if r > 2e-6 and not nm.startswith(("bn.part", "bn.tpart", "bn.dap", "bn.tdap")):
                print(f"  {nm:24s} {r:.3e}")

print("consistency of the coefficient buffers with the STORED pre-BatchNorm tensor after the step (max over channels of |mean - mean64| rstd64, |rstd / rstd64 - 1|)")
for mode in ("f", "S"):
    a = next(g["__all"] for g, f in runs if f == mode)
    for nm in sorted(a):
        if nm.endswith((".r0", ".t0")) or (nm.endswith(".s") and not nm.startswith("g.")):
            tag = nm[:-3] + ".bn1" if nm.endswith((".r0", ".t0")) else nm[:-2] + ".bn2"
            x = a[nm]
            if tag + ".mean" not in a or x.dim() != 2:
                continue
            m64, v64 = x.mean(0), x.var(0, unbiased=False)
            r64 = 1.0 / torch.sqrt(v64 + 1e-4)
            em = ((a[tag + ".mean"] - m64) * r64).abs().max()
            er = (a[tag + ".rstd"] / r64 - 1).abs().max()
            print(f"  {mode} {tag:12s} dmean {float(em):.2e} drstd {float(er):.2e}")

print("key buffers, max |mode - S| / max |S|")
keys = ["enc.0.a", "enc.3.a", "mu", "dec.0.a", "dec.2.a", "dec.3.t0a", "dec.3.a", "dec.y", "out.x6d", "dec.dy", "g.dec.top", "g.dec.3.s", "g.dec.2.s", "g.enc.c_in"]
for mode in [f for _, f in runs if f != "S"]:
    a = next(g["__all"] for g, f in runs if f == mode)
    print("  " + mode + ": " + "  ".join(f"{k} {float((a[k] - a_s[k]).abs().max()) / float(a_s[k].abs().max()):.1e}" for k in keys if k in a and k in a_s))

print("dec.3.bn2 backward recomputed in fp64 from the dumped buffers (g.dec.top, dec.3.s, coefficients) vs the dumped g.dec.3.s")
for mode in ("f", "S", "E"):
    a = next(g["__all"] for g, f in runs if f == mode)
    par = next(g["__par"] for g, f in runs if f == mode)
    gam = par["decoder.res_layers.3.add.0.weight"]; alpha = float(par["decoder.res_layers.3.add.1.weight"][0])
    dy, x = a["g.dec.top"], a["dec.3.s"]
    sc, sh, mu, rs = a["dec.3.bn2.scale"], a["dec.3.bn2.shift"], a["dec.3.bn2.mean"], a["dec.3.bn2.rstd"]
    u = x * sc + sh
    du = torch.where(u > 0, dy, alpha * dy)
    xh = (x - mu) * rs
    n = x.shape[0]
    dx = gam * rs * (du - du.sum(0) / n - xh * (du * xh).sum(0) / n)
    got = a["g.dec.3.s"]
    print(f"  {mode}: rows {n}; max |kernel - fp64 formula| / max = {float((got - dx).abs().max() / dx.abs().max()):.2e};  dsums vs formula: "
          f"{float((a['dec.3.bn2.dsums'][0] - du.sum(0)).abs().max() / du.sum(0).abs().max()):.2e} {float((a['dec.3.bn2.dsums'][1] - (du * xh).sum(0)).abs().max() / (du * xh).sum(0).abs().max()):.2e}")

print("per stage: is the difference of the pre-BatchNorm tensor x and of the stage output a between two modes per-channel affine?")
stages = [(f"enc.{i}.r0", f"enc.{i}.r0a") for i in range(4)] + [(f"enc.{i}.s", f"enc.{i}.a") for i in range(4)] + [(f"dec.{i}.t0", f"dec.{i}.t0a") for i in range(4)] + [(f"dec.{i}.s", f"dec.{i}.a") for i in range(4)]
for mode in ("f", "E"):
    a = next(g["__all"] for g, f in runs if f == mode)
    for xn, an in sorted(stages):
        if xn not in a or an not in a:
            continue
        dx = a[xn] - a_s[xn]
        da = a[an] - a_s[an]
        print(f"  {mode}-S {xn:10s} max|dx|/max|x| {float(dx.abs().max() / a_s[xn].abs().max()):.1e}   {an:10s} max|da|/max|a| {float(da.abs().max() / a_s[an].abs().max()):.1e}")

print("dec.3.bn2 backward (fp64 formula): which input's difference between mode and S moves the result? max |dx(swap) - dx(S)| / max |dx(S)|")
def bnb(a, par, dyk, xk, ck):
    gam = par["decoder.res_layers.3.add.0.weight"]; alpha = float(par["decoder.res_layers.3.add.1.weight"][0])
    dy, x = dyk["g.dec.top"], xk["dec.3.s"]
    sc, sh, mu, rs = ck["dec.3.bn2.scale"], ck["dec.3.bn2.shift"], ck["dec.3.bn2.mean"], ck["dec.3.bn2.rstd"]
    u = x * sc + sh
    du = torch.where(u > 0, dy, alpha * dy)
    xh = (x - mu) * rs
    n = x.shape[0]
    return gam * rs * (du - du.sum(0) / n - xh * (du * xh).sum(0) / n), du
par = next(g["__par"] for g, f in runs if f == "S")
ref, du_s = bnb(a_s, par, a_s, a_s, a_s)
print(f"  max|dx(S)| = {float(ref.abs().max()):.3e}, max|du| * gamma * rstd ~ {float((du_s.abs().max(0)[0] * par['decoder.res_layers.3.add.0.weight'].abs() * a_s['dec.3.bn2.rstd']).max()):.3e}, max|dy| = {float(a_s['g.dec.top'].abs().max()):.3e}")
for mode in ("f", "E"):
    a = next(g["__all"] for g, f in runs if f == mode)
    for what, args in (("dy", (a, a_s, a_s)), ("x", (a_s, a, a_s)), ("coefficients", (a_s, a_s, a)), ("all", (a, a, a))):
        r, _ = bnb(a, par, *args)
        print(f"  {mode}: {what:13s} {float((r - ref).abs().max() / ref.abs().max()):.2e}")
    u_m = a["dec.3.s"] * a["dec.3.bn2.scale"] + a["dec.3.bn2.shift"]
    u_s = a_s["dec.3.s"] * a_s["dec.3.bn2.scale"] + a_s["dec.3.bn2.shift"]
    print(f"     elements whose PReLU side differs from S: {int(((u_m > 0) != (u_s > 0)).sum())} of {u_s.numel()}")

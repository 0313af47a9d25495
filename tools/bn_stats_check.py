"""BatchNorm batch statistics of one forward pass: from the GEMM epilogues (svae_conv_fwd_split_stats) vs from the separate pass over
the stored conv output (svae_bn_stats_partial), same kernels, against the fp64 statistics of the stored tensor.
python tools/bn_stats_check.py [fixture] [precision]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from tests.test_oracle_golden import load_fixture
from tests.test_gpu_model import build_model, to_dev
from scrubvae_amd import ops

name = sys.argv[1] if len(sys.argv) > 1 else "vanilla_default_j23_B4"
prec = sys.argv[2] if len(sys.argv) > 2 else "f16x3b3"
fx, cfg, loss_scale, opt, sd, data = load_fixture(os.path.join(ROOT, "tests", "golden"), name)
eps = torch.from_numpy(fx["eps/0"])
ops.set_precision(prec)
ops.SPLIT_MIN_FLOPS = 0.0
_st = ops.Conv.stats_tiles
res = {}
for mode in ("fused", "separate"):
    ops.Conv.stats_tiles = _st if mode == "fused" else (lambda self: 0)
    model, dis = build_model(cfg, sd)
    model.train()
    d = to_dev(data)
    d["eps"] = eps.cuda()
    model(d)
    torch.cuda.synchronize()
    if mode == "fused":
        ops.TILE_TABLE.update(ops.TUNED_LOG)
    out = {}
    for (nm, shape), t in model._ws.items():
        if isinstance(nm, str) and nm.endswith((".mean", ".rstd")):
            out[nm] = t.double().cpu().clone()
    # the stored pre-BatchNorm tensors: enc.i.r0 / enc.i.s / dec.j.t0 / dec.j.s
    truth = {}
    for (nm, shape), t in model._ws.items():
        if isinstance(nm, str) and (nm.endswith((".r0", ".t0")) or (nm.endswith(".s") and not nm.startswith("g."))) and len(shape) == 2:
            x = t.double().cpu()
            tag = nm[:-3] + ".bn1" if nm.endswith((".r0", ".t0")) else nm[:-2] + ".bn2"
            truth[tag] = (x.mean(0), x.var(0, unbiased=False))
    res[mode] = (out, truth)
    del model
print(f"{name} {prec}: max over channels of |mean - mean64| * rstd64 and |rstd / rstd64 - 1| (statistics of the STORED tensor in fp64 as truth)")
for tag in sorted(res["fused"][1]):
    line = f"{tag:12s}"
    for mode in ("fused", "separate"):
        out, truth = res[mode]
        m64, v64 = truth[tag]
        r64 = 1.0 / torch.sqrt(v64 + 1e-4)
        C = out[tag + ".mean"].numel()
        m64, r64 = torch.nn.functional.pad(m64, (0, C - m64.numel())), torch.nn.functional.pad(r64, (0, C - r64.numel()), value=1.0)
        n = truth[tag][0].numel()
        em = ((out[tag + ".mean"] - m64) * r64)[:n]
        er = (out[tag + ".rstd"] / r64 - 1)[:n]
        line += f"   {mode}: dmean max {float(em.abs().max()):.1e} signed avg {float(em.mean()):+.1e} | drstd max {float(er.abs().max()):.1e} signed avg {float(er.mean()):+.1e}"
    print(line)

"""Ablation timings of the wave-specialised halo kernel (library built with --ablation): which part of a stage costs what."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from scrubvae_amd import ops
B = int(os.environ.get("B", 4096))
LAYERS = [("enc3.c3", 4, 512, 1024, 5, 1, 2, False), ("dec0.t1", 4, 1024, 512, 5, 1, 2, True), ("enc2.c3", 8, 256, 512, 5, 1, 2, False)]
def timeit(fn):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(20): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / 20 * 1e-3
NAMES = {0: "full", 1: "no global", 2: "no LDS fetch", 3: "no global+no fetch", 4: "no MFMA", 5: "no global+no MFMA", 6: "no fetch+no MFMA", 7: "barriers only"}
for name, l_in, cin, cout, k, s, p, tr in LAYERS:
    for kind, pieces in (("fwd", 3), ("dgrad", 2)):
        row = []
        for dbg in range(8):
            code = (20 + dbg) * 1000000 + 128128
            cv = ops.Conv(B, l_in, cin, cout, k, s, p, 1, tr, pieces=3)
            cv.__dict__["_tuned"] = {"fwd", "dgrad", "wgrad"}
            cv._set_choice(kind, pieces, code)
            x = torch.randn(B * l_in, cv.c_in_p, device="cuda")
            w = torch.randn(*cv.weight_shape, device="cuda") * 0.05
            y = torch.randn(B * cv.l_out, cv.c_out_p, device="cuda")
            ops.bump_weight_epoch()
            t = timeit((lambda: cv.fwd(x, w, None, y)) if kind == "fwd" else (lambda: cv.dgrad(y, w, x)))
            row.append(f"{NAMES[dbg]}: {t*1e6:6.1f}")
        print(f"{name:8s} {kind:5s} " + " | ".join(row), flush=True)

"""Timing experiments on the wave-specialised split kernel with parts of its work removed (results are wrong).
Needs the ablation variants: `python scrubvae_amd/build.py --ablation` first (and a plain rebuild afterwards)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from scrubvae_amd import ops
B = 1024
l_in, cin, cout, k, s, p, tr = (4, 512, 1024, 5, 1, 2, False)
def timeit(fn, reps=20):
    fn(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / reps * 1e3
names = {0: "full", 1: "no global loads", 2: "no split", 3: "no loads, no split", 4: "no MFMA", 8: "no LDS reads", 12: "no MFMA, no LDS reads",
         5: "no loads, no MFMA", 13: "no loads/MFMA/LDS reads", 15: "nothing but LDS writes+barriers", 7: "no loads/split/MFMA"}
for dbg, nm in names.items():
    cv = ops.Conv(B, l_in, cin, cout, k, s, p, 1, tr, pieces=3)
    cv.__dict__["_tuned"] = {"fwd", "dgrad", "wgrad"}
    cv.desc.tile[0] = (10 + dbg) * 1000000 + 128128
    x = torch.randn(B * l_in, cv.c_in_p, device="cuda")
    w = torch.randn(*cv.weight_shape, device="cuda") * 0.05
    y = torch.empty(B * cv.l_out, cv.c_out_p, device="cuda")
    ops.bump_weight_epoch()
    print(f"dbg={dbg:2d} {nm:32s} {timeit(lambda: cv.fwd(x, w, None, y)):7.1f} us")

"""Time every split-bf16 gather kernel variant (tile code) on a few layer geometries, for 1/2/3 pieces."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from scrubvae_amd import ops

B = 1024
LAYERS = [("enc3.c3", 4, 512, 1024, 5, 1, 2, False), ("enc1.sk", 32, 128, 256, 5, 2, 2, False), ("dec2.sk", 25, 256, 128, 6, 1, 2, False),
          ("enc0.c3", 32, 64, 128, 5, 1, 2, False)]
CODES = [c for c in ops._SPLIT_GATHER_CODES]

def timeit(fn, reps=20):
    fn(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / reps * 1e-3

for name, l_in, cin, cout, k, s, p, tr in LAYERS:
    print(f"== {name}: L={l_in} {cin}->{cout} k={k} s={s}")
    for pieces in (3, 2, 1):
        row = []
        for code in CODES:
            cv = ops.Conv(B, l_in, cin, cout, k, s, p, 1, tr, pieces=pieces)
            cv.__dict__["_tuned"] = {"fwd", "dgrad", "wgrad"}
            cv.desc.tile[0] = code
            x = torch.randn(B * l_in, cv.c_in_p, device="cuda")
            w = torch.randn(*cv.weight_shape, device="cuda") * 0.05
            y = torch.empty(B * cv.l_out, cv.c_out_p, device="cuda")
            ops.bump_weight_epoch()
            try:
                t = timeit(lambda: cv.fwd(x, w, None, y))
                row.append(f"{code}:{t*1e6:.0f}us/{cv.flops/t/1e12:.0f}TF")
            except RuntimeError:
                row.append(f"{code}:n/a")
        print(f"  P={pieces}: " + "  ".join(row))

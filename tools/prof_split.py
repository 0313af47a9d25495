"""Run one layer's forward with a few split-bf16 kernel variants (for rocprofv3 --pmc passes)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from scrubvae_amd import ops
B = 1024
name, l_in, cin, cout, k, s, p, tr = ("enc3.c3", 4, 512, 1024, 5, 1, 2, False)
codes = [int(c) for c in sys.argv[1:]] or [3128128, 4128128, 128128]
for code in codes:
    for pieces in (3, 1):
        cv = ops.Conv(B, l_in, cin, cout, k, s, p, 1, tr, pieces=pieces)
        cv.__dict__["_tuned"] = {"fwd", "dgrad", "wgrad"}
        cv.desc.tile[0] = code
        x = torch.randn(B * l_in, cv.c_in_p, device="cuda")
        w = torch.randn(*cv.weight_shape, device="cuda") * 0.05
        y = torch.empty(B * cv.l_out, cv.c_out_p, device="cuda")
        ops.bump_weight_epoch()
        for _ in range(4):
            cv.fwd(x, w, None, y)
torch.cuda.synchronize()

"""Time the split-bf16 weight gradient of a few layers for given tile codes (kernel experiments)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from scrubvae_amd import ops
B = 1024
LAYERS = [("dec0.sk", 8, 1024, 512, 6, 1, 2, False), ("enc3.c3", 4, 512, 1024, 5, 1, 2, False), ("enc2.c3", 8, 256, 512, 5, 1, 2, False),
          ("enc1.c3", 16, 128, 256, 5, 1, 2, False), ("dec3.sk", 50, 128, 64, 6, 1, 2, False)]
codes = [int(c) for c in sys.argv[1:]] or [128128, 2128128]
for name, l_in, cin, cout, k, s, p, tr in LAYERS:
    row = []
    for code in codes:
        cv = ops.Conv(B, l_in, cin, cout, k, s, p, 1, tr, pieces=3)
        cv.__dict__["_tuned"] = {"fwd", "dgrad", "wgrad"}
        cv._set_choice("wgrad", 2, code)
        x = torch.randn(B * l_in, cv.c_in_p, device="cuda")
        dy = torch.randn(B * cv.l_out, cv.c_out_p, device="cuda")
        dw = torch.empty(*cv.weight_shape, device="cuda")
        ws = torch.empty(cv.wgrad_workspace_bytes() // 4 + 16, device="cuda")
        for _ in range(3):
            cv.wgrad(x, dy, dw, None, ws)
        torch.cuda.synchronize()
        s0, e0 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s0.record()
        for _ in range(20):
            cv.wgrad(x, dy, dw, None, ws)
        e0.record(); torch.cuda.synchronize()
        t = s0.elapsed_time(e0) / 20 * 1e-3
        row.append(f"{code}: {t*1e6:6.1f} us {cv.flops/t/1e12:6.1f} TF")
    print(f"{name:8s} " + " | ".join(row))

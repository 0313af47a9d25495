"""Run the weight gradient of two large layers with a few split-bf16 kernel variants (for rocprofv3 --pmc passes)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from scrubvae_amd import ops
B = 1024
LAYERS = [("dec0.sk", 8, 1024, 512, 6, 1, 2, False), ("enc3.c3", 4, 512, 1024, 5, 1, 2, False)]
codes = [int(c) for c in sys.argv[1:]] or [128128, 2128128]
for name, l_in, cin, cout, k, s, p, tr in LAYERS:
    for code in codes:
        cv = ops.Conv(B, l_in, cin, cout, k, s, p, 1, tr, pieces=3)
        cv.__dict__["_tuned"] = {"fwd", "dgrad", "wgrad"}
        cv._set_choice("wgrad", 2, code)
        x = torch.randn(B * l_in, cv.c_in_p, device="cuda")
        dy = torch.randn(B * cv.l_out, cv.c_out_p, device="cuda")
        dw = torch.empty(*cv.weight_shape, device="cuda")
        ws = torch.empty(cv.wgrad_workspace_bytes() // 4 + 16, device="cuda")
        for _ in range(4):
            cv.wgrad(x, dy, dw, None, ws)
        print(name, code, cv.kernel_name("wgrad"))
torch.cuda.synchronize()

"""Run a few GEMM layers a handful of times (for rocprofv3 --pmc passes)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from scrubvae_amd import ops
ops.AUTOTUNE = len(sys.argv) > 1 and sys.argv[1] == "tune"
B = 1024
cases = [("enc3.sk", 8, 512, 1024, 5, 2, 2, False), ("dec0.sk", 8, 1024, 512, 6, 1, 2, False), ("enc1.sk", 32, 128, 256, 5, 2, 2, False)]
for name, l_in, cin, cout, k, s, p, tr in cases:
    cv = ops.Conv(B, l_in, cin, cout, k, s, p, 1, tr)
    x = torch.randn(B * l_in, cv.c_in_p, device="cuda")
    w = torch.randn(*cv.weight_shape, device="cuda") * 0.05
    b = torch.zeros(cv.c_out_p, device="cuda")
    y = torch.empty(B * cv.l_out, cv.c_out_p, device="cuda")
    dy = torch.randn(B * cv.l_out, cv.c_out_p, device="cuda")
    dx = torch.empty_like(x); dw = torch.empty_like(w); db = torch.empty_like(b)
    ws = torch.empty(cv.wgrad_workspace_bytes() // 4 + 16, device="cuda")
    for _ in range(3):
        cv.fwd(x, w, b, y); cv.dgrad(dy, w, dx); cv.wgrad(x, dy, dw, db, ws)
torch.cuda.synchronize()

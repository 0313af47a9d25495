"""Time every weight-gradient kernel variant (tile codes + tap-fused) on the small-weight layers."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from scrubvae_amd import ops
B = 1024
cases = [("enc0.c0", 64, 64, 64, 5, 2, 2, False), ("enc0.sk", 64, 64, 128, 5, 2, 2, False), ("enc1.c0", 32, 128, 128, 5, 2, 2, False),
         ("conv_in", 64, 141, 64, 7, 1, 3, False), ("dec3.t2", 25, 64, 64, 5, 2, 2, True), ("dec3.t1", 25, 128, 64, 5, 1, 2, True),
         ("dec3.sk", 50, 128, 64, 6, 1, 2, False), ("dec.out", 49, 64, 141, 22, 1, 3, True), ("dec2.t2", 13, 128, 128, 5, 2, 2, True)]
def timeit(fn, reps=10):
    fn(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / reps * 1e3
for name, l_in, cin, cout, k, s, p, tr in cases:
    res = []
    for code in (128128, 128064, 64128, 64064, 1):
        cv = ops.Conv(B, l_in, cin, cout, k, s, p, 1, tr)
        cv.__dict__["_tuned"] = {"fwd", "dgrad", "wgrad"}
        cv.desc.tile[2] = code; cv._ws_bytes = None
        x = torch.randn(B * l_in, cv.c_in_p, device="cuda"); dy = torch.randn(B * cv.l_out, cv.c_out_p, device="cuda")
        dw = torch.empty(*cv.weight_shape, device="cuda")
        ws = torch.empty(cv.wgrad_workspace_bytes() // 4 + 16, device="cuda")
        t = timeit(lambda: cv.wgrad(x, dy, dw, None, ws))
        res.append(f"{code}:{t:6.1f}us({cv.flops/t/1e6:5.1f}TF)")
    print(f"{name:9s} " + "  ".join(res))

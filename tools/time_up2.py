"""The decoder's skip path (Upsample x2 -> Conv1d(k + 1)) with and without the upsample folded into the conv kernels' operand
staging (svae_conv_desc.up2): forward and weight gradient of the four skip convs at batch B, every candidate tile code, cold
operands (COLD=1, see tools/time_gather.py).  "plain" rows exclude the upsample2_fwd launch that writes the `up` tensor (printed per
layer); "fused+up": the fused forward that also leaves `up` behind for a plain weight gradient.

    B=4096 COLD=1 python tools/time_up2.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from scrubvae_amd import ops

B = int(os.environ.get("B", 4096))
COLD = os.environ.get("COLD", "1") != "0"
FWD_PIECES = int(os.environ.get("FWD_PIECES", 22))
LAYERS = [("dec0.sk", 4, 1024, 512), ("dec1.sk", 7, 512, 256), ("dec2.sk", 13, 256, 128), ("dec3.sk", 25, 128, 64)]
FWD_PLAIN = [8128128, 9128128, 16128128, 17128128, 29128128]
FWD_UP = [8128128, 8128064, 9128128, 29128128]
WG = [4064128, 6064128, 4128064, 6128064, 12064128, 14064128, 12128064, 14128064]
_flush = None


def timeit(fn):
    global _flush
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    if _flush is None:
        _flush = torch.empty(1 << 28, dtype=torch.float32, device="cuda")
    tot, n = 0.0, 8
    for _ in range(n):
        if COLD:
            _flush.add_(1.0)
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record(); fn(); e.record(); torch.cuda.synchronize()
        tot += s.elapsed_time(e)
    return tot / n * 1e3


def conv(L, cin, cout, up2, kind, pieces, code):
    cv = ops.Conv(B, 2 * L, cin, cout, 6, 1, 2, pieces=3, up2=up2)
    cv.wgrad_pieces = cv.dgrad_pieces = 2
    cv.__dict__["_tuned"] = {"fwd", "dgrad", "wgrad"}
    cv._set_choice(kind, pieces, code)
    return cv


for name, L, cin, cout in LAYERS:
    x = torch.randn(B * L, cin, device="cuda")
    up = torch.empty(B * 2 * L, cin, device="cuda")
    w = torch.randn(6, cin, cout, device="cuda") * 0.05
    y = torch.empty(B * (2 * L - 1), cout, device="cuda")
    dy = torch.randn_like(y)
    dw = torch.empty_like(w)
    ops.bump_weight_epoch()
    t_up = timeit(lambda: ops.upsample2_fwd(x, up, B, L, cin, cin))
    print(f"{name}: upsample2_fwd {t_up:6.1f} us")
    for label, up2, codes in (("fwd  plain", False, FWD_PLAIN), ("fwd  fused", True, FWD_UP), ("fwd  fused+up", 2, FWD_UP)):
        row = []
        for code in codes:
            cv = conv(L, cin, cout, bool(up2), "fwd", FWD_PIECES, code)
            try:
                t = timeit((lambda: cv.fwd(x, w, None, y, up_out=up if up2 == 2 else None)) if up2 else (lambda: cv.fwd(up, w, None, y)))
                row.append(f"{code}: {t:6.1f}")
            except RuntimeError:
                row.append(f"{code}:   n/a")
        print(f"  {label} " + " | ".join(row))
    for label, up2 in (("wgrad plain", False), ("wgrad fused", True)):
        row = []
        for code in WG:
            cv = conv(L, cin, cout, up2, "wgrad", 2, code)
            cv.up2_wgrad = True
            cv._ws_bytes = None
            try:
                ws = torch.empty(cv.wgrad_workspace_bytes() // 4 + 16, device="cuda")
                t = timeit((lambda: cv.wgrad(x, dy, dw, None, ws)) if up2 else (lambda: cv.wgrad(up, dy, dw, None, ws)))
                row.append(f"{code}: {t:6.1f}")
            except RuntimeError:
                row.append(f"{code}:   n/a")
        print(f"  {label} " + " | ".join(row))

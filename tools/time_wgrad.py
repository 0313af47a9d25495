"""Time the weight-gradient kernels of a few layers for given tile codes (kernel experiments)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from scrubvae_amd import ops
B = int(os.environ.get("B", 4096))
LAYERS = [("enc3.c3", 4, 512, 1024, 5, 1, 2, False), ("dec0.t1", 4, 1024, 512, 5, 1, 2, True), ("enc2.c3", 8, 256, 512, 5, 1, 2, False),
          ("enc1.c3", 16, 128, 256, 5, 1, 2, False), ("enc0.c3", 32, 64, 128, 5, 1, 2, False), ("enc2.c1", 16, 256, 512, 5, 2, 2, False),
          ("dec2.t2", 16, 256, 128, 5, 2, 2, True)]
if os.environ.get("LAYERS") == "shallow":  # many rows, few channels: conv_in, the 64 / 128-channel blocks, the decoder's last skip conv
    LAYERS = [("conv_in", 64, 141, 64, 7, 1, 3, False), ("dec3.sk", 50, 128, 64, 6, 1, 2, False), ("dec2.sk", 26, 256, 128, 6, 1, 2, False),
              ("enc0.c3", 32, 64, 128, 5, 1, 2, False), ("dec3.t1", 25, 128, 64, 5, 1, 2, True), ("dec.out", 49, 64, 141, 22, 1, 3, True)]
if os.environ.get("LAYERS") == "skinny":  # tiny outputs, 65 k - 200 k reduction rows: what the all-taps kernels serve at 20-30 % busy
    LAYERS = [("enc0.c0", 64, 64, 64, 5, 2, 2, False), ("enc0.sk", 64, 64, 128, 5, 2, 2, False), ("enc0.c3", 32, 64, 128, 5, 1, 2, False),
              ("enc1.c0", 32, 128, 128, 5, 2, 2, False), ("enc1.sk", 32, 128, 256, 5, 2, 2, False), ("dec2.t2", 13, 128, 128, 5, 2, 2, True),
              ("dec3.t1", 25, 128, 64, 5, 1, 2, True), ("dec3.t2", 25, 64, 64, 5, 2, 2, True), ("dec3.sk", 50, 128, 64, 6, 1, 2, False)]
if os.environ.get("LAYERS") == "conv_in":
    LAYERS = [("conv_in23", 64, 141, 64, 7, 1, 3, False), ("conv_in18", 64, 111, 64, 7, 1, 3, False)]
codes = [int(c) for c in sys.argv[1:]] or [256256, 2256256]
COLD = os.environ.get("COLD", "0") != "0"
_flush = None
def timeit(fn):
    """Mean launch time.  COLD=1: every timed launch follows a 1 GiB write that evicts the operands from L2 and the Infinity
    Cache -- the state a layer's operands are in when its kernel runs inside a training step (repeated launches on the same
    tensors otherwise read them from the 256 MiB Infinity Cache and flatter kernels with a short prefetch distance)."""
    global _flush
    for _ in range(3): fn()
    torch.cuda.synchronize()
    if not COLD:
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(20): fn()
        e.record(); torch.cuda.synchronize()
        return s.elapsed_time(e) / 20 * 1e-3
    if _flush is None:
        _flush = torch.empty(1 << 28, dtype=torch.float32, device="cuda")
    tot = 0.0
    for _ in range(8):
        _flush.add_(1.0)
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record(); fn(); e.record(); torch.cuda.synchronize()
        tot += s.elapsed_time(e)
    return tot / 8 * 1e-3
for name, l_in, cin, cout, k, s, p, tr in LAYERS:
    row = []
    for code in codes:
        cv = ops.Conv(B, l_in, cin, cout, k, s, p, 1, tr, pieces=2)
        cv.__dict__["_tuned"] = {"fwd", "dgrad", "wgrad"}
        cv._set_choice("wgrad", 2, code)
        x = torch.randn(B * l_in, cv.c_in_p, device="cuda")
        dy = torch.randn(B * cv.l_out, cv.c_out_p, device="cuda")
        dw = torch.empty(cv.weight_shape, device="cuda")
        db = torch.empty(cv.c_out_p, device="cuda")
        try:
            ws = torch.empty(cv.wgrad_workspace_bytes() // 4 + 4, device="cuda")
            t = timeit(lambda: cv.wgrad(x, dy, dw, None, ws))
            row.append(f"{code}: {t*1e6:6.1f} us {cv.flops/t/1e12:6.1f} TF")
        except RuntimeError as e:
            row.append(f"{code}: n/a")
    print(f"{name:8s} wgrad " + " | ".join(row), flush=True)

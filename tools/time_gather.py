"""Time forward / data-gradient of a few layers for given split-gather tile codes (kernel experiments)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from scrubvae_amd import ops
B = int(os.environ.get("B", 1024))
MID = [("dec1.sk", 14, 512, 256, 6, 1, 2, False), ("dec0.t2", 4, 512, 512, 5, 2, 2, True), ("enc3.c0", 8, 512, 512, 5, 2, 2, False),
       ("dec2.sk", 26, 256, 128, 6, 1, 2, False), ("dec1.t1", 7, 512, 256, 5, 1, 2, True), ("enc1.sk", 32, 128, 256, 5, 2, 2, False),
       ("dec3.sk", 50, 128, 64, 6, 1, 2, False), ("dec2.t1", 13, 256, 128, 5, 1, 2, True)]  # LAYERS=mid: the mid-size data-gradients
LAYERS = [("enc3.c3", 4, 512, 1024, 5, 1, 2, False), ("dec0.sk", 8, 1024, 512, 6, 1, 2, False), ("dec0.t1", 4, 1024, 512, 5, 1, 2, True),
          ("enc2.c3", 8, 256, 512, 5, 1, 2, False), ("dec1.sk", 14, 512, 256, 6, 1, 2, False), ("enc1.c3", 16, 128, 256, 5, 1, 2, False)]
S2 = [("enc3.sk", 8, 512, 1024, 5, 2, 2, False), ("enc3.c0", 8, 512, 512, 5, 2, 2, False), ("enc2.sk", 16, 256, 512, 5, 2, 2, False),
      ("enc2.c0", 16, 256, 256, 5, 2, 2, False), ("enc1.sk", 32, 128, 256, 5, 2, 2, False), ("dec0.t2", 4, 512, 512, 5, 2, 2, True),
      ("dec1.t2", 7, 256, 256, 5, 2, 2, True)]  # LAYERS=s2: the stride-2 layers (parity phases in the data-gradient / transposed forward)
if os.environ.get("LAYERS") == "mid":
    LAYERS = MID
if os.environ.get("LAYERS") == "s2":
    LAYERS = S2
codes = [int(c) for c in sys.argv[1:]] or [8128128, 9128128]
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
    for kind, pieces in (("fwd", int(os.environ.get("FWD_PIECES", 3))), ("dgrad", 2)):
        row = []
        for code in codes:
            cv = ops.Conv(B, l_in, cin, cout, k, s, p, 1, tr, pieces=3)
            cv.__dict__["_tuned"] = {"fwd", "dgrad", "wgrad"}
            cv._set_choice(kind, pieces, code)
            x = torch.randn(B * l_in, cv.c_in_p, device="cuda")
            w = torch.randn(*cv.weight_shape, device="cuda") * 0.05
            y = torch.empty(B * cv.l_out, cv.c_out_p, device="cuda")
            ops.bump_weight_epoch()
            try:
                t = timeit((lambda: cv.fwd(x, w, None, y)) if kind == "fwd" else (lambda: cv.dgrad(y, w, x)))
                row.append(f"{code}: {t*1e6:6.1f} us {cv.flops/t/1e12:6.1f} TF")
            except RuntimeError as e:
                row.append(f"{code}: n/a")
        print(f"{name:8s} {kind:5s} " + " | ".join(row))

"""In-kernel s_memtime stamps of the 12-wave halo kernel's stage loop (library built with --ablation; tile codes 37 / 38).

The stamped build logs the low 32 bits of six stamps per wave and stage (stages 8..23) to LDS and dumps the log after the loop;
consumer waves 0-3: top | fetch k0 + raw read | mma k1 | raw write + fetch k1 | mma k0 | barrier;  waves 4-7 (rotated):
top | mma k0 | fetch k0 + raw read | mma k1 | raw write + fetch k1 | barrier;  loader waves 8-11: top | requests issued | counted
wait done | barrier.  Also prints the in-kernel clock (s_memtime vs the 100 MHz s_memrealtime) and the cycles per stage.
Stamps are diagnostic only: the build is never shipped and no output depends on them."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from scrubvae_amd import ops, _lib

B = int(os.environ.get("B", 4096))
LAYERS = {"enc3.c3": (4, 512, 1024, 5, 1, 2, False), "enc2.c3": (8, 256, 512, 5, 1, 2, False), "dec0.t1": (4, 1024, 512, 5, 1, 2, True)}
NW, NS = 12, 16


def run(name, kind, pieces, variant):
    l_in, cin, cout, k, s, p, tr = LAYERS[name]
    code = variant * 1000000 + 128128
    cv = ops.Conv(B, l_in, cin, cout, k, s, p, 1, tr, pieces=3)
    cv.__dict__["_tuned"] = {"fwd", "dgrad", "wgrad"}
    cv._set_choice(kind, pieces, code)
    x = torch.randn(B * l_in, cv.c_in_p, device="cuda")
    w = torch.randn(*cv.weight_shape, device="cuda") * 0.05
    y = torch.randn(B * cv.l_out, cv.c_out_p, device="cuda")
    ops.bump_weight_epoch()
    buf = torch.zeros(8 + NW * NS * 6, dtype=torch.int64, device="cuda")
    l = _lib.lib()
    l.svae_debug_stamp_buffer.argtypes = [C.c_void_p]
    l.svae_debug_stamp_buffer.restype = C.c_int
    fn = (lambda: cv.fwd(x, w, None, y)) if kind == "fwd" else (lambda: cv.dgrad(y, w, x))
    assert l.svae_debug_stamp_buffer(None) == 0
    for _ in range(20):  # warm clocks / caches
        fn()
    torch.cuda.synchronize()
    assert l.svae_debug_stamp_buffer(C.c_void_p(buf.data_ptr())) == 0
    fn()
    torch.cuda.synchronize()
    assert l.svae_debug_stamp_buffer(None) == 0
    raw = buf.cpu().numpy().astype(np.int64)
    cyc, ref, nst = raw[0], raw[1], raw[2]
    print(f"--- variant {variant} {name} {kind} pieces={pieces}: in-kernel clock {cyc / max(ref, 1) * 0.1:.3f} GHz, "
          f"{cyc / max(nst, 1):.0f} cycles per stage ({nst} stages)")
    t = raw[8:].reshape(NW, NS, 6)
    base = t[0, 2, 0]
    t = (t - base) & 0xffffffff
    t = np.where(t > 1 << 31, t - (1 << 32), t)
    for st in range(2, 8):
        for wv in (0, 4, 8, 9, 10, 11):
            n = 6 if wv < 8 else 4
            print(f"   stage {8 + st:2d} wave {wv:2d}: " + " ".join(f"{int(v):6d}" for v in t[wv, st, :n]))
    d = np.diff(t, axis=2)
    for wv in range(NW):
        n = 5 if wv < 8 else 3
        whole = np.median(t[wv, 1:, 0] - t[wv, :-1, 0])
        print(f"wave {wv:2d} median parts {np.median(d[wv, :, :n], axis=0).astype(int).tolist()} stage {int(whole)}")


if __name__ == "__main__":
    name = os.environ.get("LAYER", "enc3.c3")
    for variant in (37, 38):  # 37: staggered consumers, 38: not
        run(name, "fwd", 22, variant)
    run(name, "dgrad", 2, 37)

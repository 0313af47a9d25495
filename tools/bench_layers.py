"""Per-layer GEMM throughput on the GPU: every conv/linear of the default SC-VAE at a given batch,
forward / dgrad / wgrad, TFLOP/s of algorithmic FLOPs (fp32 MFMA peak 157.3)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from scrubvae_amd import ops

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
if len(sys.argv) > 2:
    ops.set_precision(sys.argv[2])  # f32 | bf16x6 | bf16x3 | bf16
    ops.TILE_TABLE = {k: v for k, v in ops.TILE_TABLE.items() if "@" in k.split(":")[0]}
for kind in filter(None, os.environ.get("RETUNE", "").split(",")):  # RETUNE=wgrad: tune that pass afresh (new kernel variants)
    ops.TILE_TABLE = {k: v for k, v in ops.TILE_TABLE.items() if not k.startswith(kind)}
J = 23
C = 6 * J + 3
ch = [64, 128, 256, 512, 1024]
layers = [("enc.conv_in", 64, C, ch[0], 7, 1, 3, False)]
L = 64
for i in range(4):
    layers += [(f"enc{i}.c0", L, ch[i], ch[i + 1] // 2, 5, 2, 2, False), (f"enc{i}.sk", L, ch[i], ch[i + 1], 5, 2, 2, False),
               (f"enc{i}.c3", L // 2, ch[i + 1] // 2, ch[i + 1], 5, 1, 2, False)]
    L //= 2
layers += [("fc_mu", 1, 4096, 32, 1, 1, 0, False), ("fc_in", 1, 32, 4096, 1, 1, 0, False)]
L = 4
for j, i in enumerate(range(1, 5)):
    cin, cout = ch[-i], ch[-i - 1]
    layers += [(f"dec{j}.t1", L, cin, cin // 2, 5, 1, 2, True), (f"dec{j}.t2", L, cin // 2, cout, 5, 2, 2, True),
               (f"dec{j}.sk", 2 * L, cin, cout, 6, 1, 2, False)]
    L = 2 * L - 1
layers += [("dec.out", L, ch[0], C, 22, 1, 3, True)]

def timeit(fn, reps=10):
    fn(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / reps * 1e-3

tot = {"fwd": [0, 0], "dgrad": [0, 0], "wgrad": [0, 0]}
print(f"B={B}  {'layer':12s} {'M':>7s} {'K':>6s} {'N':>5s} | fwd TF  us | dgrad TF  us | wgrad TF  us")
for name, l_in, cin, cout, k, s, p, tr in layers:
    cv = ops.Conv(B, l_in, cin, cout, k, s, p, 1, tr)
    x = torch.randn(B * l_in, cv.c_in_p, device="cuda")
    w = torch.randn(*cv.weight_shape, device="cuda") * 0.05
    b = torch.zeros(cv.c_out_p, device="cuda")
    y = torch.empty(B * cv.l_out, cv.c_out_p, device="cuda")
    dy = torch.randn(B * cv.l_out, cv.c_out_p, device="cuda")
    dx = torch.empty_like(x)
    dw, db = torch.empty_like(w), torch.empty_like(b)
    ws = torch.empty(cv.wgrad_workspace_bytes() // 4 + 16, device="cuda")
    t = [timeit(lambda: cv.fwd(x, w, b, y)), timeit(lambda: cv.dgrad(dy, w, dx)), timeit(lambda: cv.wgrad(x, dy, dw, db, ws))]
    for kk, tt in zip(("fwd", "dgrad", "wgrad"), t):
        tot[kk][0] += cv.flops; tot[kk][1] += tt
    M = B * (l_in if tr else cv.l_out)
    codes = "/".join(str(cv.desc.tile[i]) for i in range(3))
    print(f"      {name:12s} {M:7d} {k*cin:6d} {cout:5d} | " + " | ".join(f"{cv.flops/tt/1e12:6.1f} {tt*1e6:6.0f}" for tt in t) + f"   {codes}")
for kk, (f, tt) in tot.items():
    print(f"TOTAL {kk}: {f/1e9:.1f} GFLOP in {tt*1e3:.2f} ms = {f/tt/1e12:.1f} TFLOP/s ({f/tt/1e12/157.3*100:.0f}% of fp32 MFMA peak)")

"""cProfile of the host side of one optimizer step (launch path only; the GPU runs behind).
    python tools/host_profile.py [batch] [config1|config2]"""
import argparse, cProfile, pstats, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from scrubvae_amd import ops
from scrubvae_amd.data import synthetic
from scrubvae_amd.train.losses import get_batch_loss
from scrubvae_amd.train.trainer import FusedAdam, clip_grad_norm_

B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
wl = sys.argv[2] if len(sys.argv) > 2 else "config1"
full = bench.WORKLOADS[wl]["full"]
args = argparse.Namespace(window=64, joints=23, channel_list=bench.CHANNELS, sync_bn=False)
ops.set_precision("f16x3b3")
method, feats, loss = bench.make_cfg(full)
data, tree = synthetic.make_batch(23, 64, B, seed=100, device="cuda")
model, dis = bench.build_model(args, full, method, feats, tree)
model.defer_tail = True
opt = FusedAdam(model, lr=1e-4, weight_decay=0.01, decoupled=True)
model.train()


def step():
    bl = get_batch_loss(model, data, model(data), loss, dis)
    bl["total"].backward(); clip_grad_norm_(model, 1e6); opt.step()


for _ in range(8): step()
torch.cuda.synchronize()
import time
t0 = time.perf_counter()
for _ in range(20): step()
t_host = (time.perf_counter() - t0) / 20
torch.cuda.synchronize()
print(f"host enqueue {t_host * 1e3:.3f} ms/step without the profiler (B={B}, {wl})")
pr = cProfile.Profile(); pr.enable()
for _ in range(20): step()
pr.disable(); torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(32)
st.sort_stats("cumulative").print_stats(24)

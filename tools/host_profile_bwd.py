"""cProfile of the host side of the reverse schedule (ResVAE.backward_from_seeds called directly, outside the autograd thread).
    python tools/host_profile_bwd.py [batch] [config1|config2]"""
import cProfile, pstats, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
import argparse
from scrubvae_amd import ops
from scrubvae_amd.data import synthetic
from scrubvae_amd.train.losses import get_batch_loss
from scrubvae_amd.train.trainer import FusedAdam, clip_grad_norm_
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
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
for _ in range(5):
    bl = get_batch_loss(model, data, model(data), loss, dis); bl["total"].backward(); opt.step()
torch.cuda.synchronize()
pr = cProfile.Profile()
for _ in range(20):
    bl = get_batch_loss(model, data, model(data), loss, dis)
    pr.enable(); model.backward_from_seeds(); pr.disable()
    clip_grad_norm_(model, 1e6); opt.step()
torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("tottime").print_stats(22)

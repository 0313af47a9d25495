"""Training-quality check of the contraction precisions: the default-size SC-VAE trained for N steps from the same
weights, on the same stream of synthetic batches and the same reparameterisation noise, in each precision; prints the
loss averaged over windows of steps.  (Adam amplifies rounding-level gradient differences, so individual trajectories
decorrelate; what must agree is the level the loss reaches.)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from scrubvae_amd import ops
from scrubvae_amd.data import synthetic
from scrubvae_amd.get import model as get_model
from scrubvae_amd.train.losses import get_batch_loss
from scrubvae_amd.train.trainer import FusedAdam, clip_grad_norm_

STEPS, B, NB = int(sys.argv[1]) if len(sys.argv) > 1 else 400, 256, 8
PRECISIONS = sys.argv[2:] or ["f32", "bf16x6", "bf16x6w3", "bf16x6b3"]
ARENA = torch.tensor([[-1.0, -1, -1], [1, 1, 1]])
batches = [synthetic.make_batch(23, 64, B, seed=500 + i, device="cuda") for i in range(NB)]  # a small fixed dataset: it can be fitted
tree = batches[0][1]
g = torch.Generator(device="cuda").manual_seed(7)
eps = [torch.randn(B, 32, device="cuda", generator=g) for _ in range(STEPS)]
loss = {"jpe": 1.0, "root": 1.0, "prior": 0.01}
mc = dict(type="rcnn", kernel=5, z_dim=32, window=64, activation="prelu", diag=True, init_dilation=None, prior="gaussian",
          channel=[64, 128, 256, 512, 1024])
dis = dict(method={}, alpha=1.0, features=[])
sd0 = None
for prec in PRECISIONS:
    ops.set_precision(prec)
    torch.manual_seed(0)
    m = get_model(mc, None, None, dis, 23, "midfwd", arena_size=ARENA, kinematic_tree=tree, device="cuda", verbose=0)
    if sd0 is None:
        sd0 = {k: v.clone() for k, v in m.state_dict().items()}
    m.load_state_dict(sd0)
    opt = FusedAdam(m, lr=3e-4, weight_decay=0.01, decoupled=True)
    m.train()
    hist = []
    for s in range(STEPS):
        d = dict(batches[s % NB][0]); d["eps"] = eps[s]
        bl = get_batch_loss(m, d, m(d), loss, dis)
        bl["total"].backward(); clip_grad_norm_(m, 1e6); opt.step()
        hist.append(bl["total"].detach())
    h = torch.stack(hist).cpu()
    w = max(STEPS // 8, 1)
    print(f"{prec:9s} " + " ".join(f"{float(h[i:i + w].mean()):9.3f}" for i in range(0, STEPS, w)), flush=True)
    del m, opt
    torch.cuda.empty_cache()

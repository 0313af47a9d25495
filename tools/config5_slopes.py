"""PReLU-slope gradients of BASELINE configs[4]'s geometry at B = 4 (tests/test_gpu_model.py::test_oracle_parity_config5_wide_w256):
error against the fp64 oracle of the HIP path per precision, next to the fp32 oracle's.
    python tools/config5_slopes.py [precision ...]      (SVAE_FUSE_UPSAMPLE etc. from the environment)"""
import dataclasses, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from oracle import scvae_oracle as O
from tests.test_oracle_golden import ARENA
from tests.test_gpu_model import build_model, to_dev
from scrubvae_amd import ops
from scrubvae_amd.train.losses import get_batch_loss

B = int(os.environ.get("B", 4))
cfg = O.OracleConfig(n_keypts=23, window=256, z_dim=32, kernel=5, diag=True, arena_size=ARENA,
                     channel=(64, 128, 256, 512, 1024, 2048, 4096), kinematic_tree=O.skeleton_tree(23))
sd = O.init_state_dict(cfg, seed=13)
data = O.synth_batch(cfg, B, seed=4)
eps = torch.randn(B, 32, generator=torch.Generator().manual_seed(6))
ls = {"jpe": 1.0, "root": 1.0, "prior": 1.0}
_, g_o, _, _ = O.train_step(sd, cfg, data, ls, eps)
c64 = dataclasses.replace(cfg, arena_size=ARENA.double())
f64 = lambda d_: {k: (v.double() if v.dtype.is_floating_point else v) for k, v in d_.items()}
_, g64, _, _ = O.train_step(f64(sd), c64, f64(data), ls, eps.double())
gmax = max(float(g.abs().max()) for g in g64.values())
err = lambda a, n: float((a.double() - g64[n]).abs().max()) / (float(g64[n].abs().max()) + 1e-3 * gmax)
slopes = [n for n in g64 if g64[n].numel() == 1]
rows = {n: [float(g64[n]), err(g_o[n], n)] for n in slopes}
precs = sys.argv[1:] or ["f32", "bf16x6", "bf16x6w3", "bf16x6b3", "f16x3b3"]
for prec in precs:
    ops.set_precision(prec)
    model, dis = build_model(cfg, sd)
    model.train()
    d = to_dev(data)
    d["eps"] = eps.cuda()
    bl = get_batch_loss(model, d, model(d), ls, dis)
    bl["total"].backward()
    torch.cuda.synchronize()
    grads = {k: v.cpu() for k, v in model.grads_state_dict().items()}
    for n in slopes:
        rows[n].append(err(grads[n], n))
    vec = lambda gs: float(torch.sqrt(sum(((gs[n].double() - g64[n]) ** 2).sum() for n in g64)) / torch.sqrt(sum((g64[n] ** 2).sum() for n in g64)))
    print(f"{prec}: whole vector {vec(grads):.2e} (fp32 oracle {vec(g_o):.2e})")
    del model
print(f"{'slope':44s} {'fp64 value':>11s} {'fp32 orc':>9s} " + " ".join(f"{p:>9s}" for p in precs))
for n in slopes:
    print(f"{n:44s} {rows[n][0]:11.4f} {rows[n][1]:9.1e} " + " ".join(f"{e:9.1e}" for e in rows[n][2:]))

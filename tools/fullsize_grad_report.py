"""Per-tensor gradient errors of one full-size training step: HIP vs the fp32 CPU oracle, HIP vs the fp64 oracle (truth), and the
fp32 oracle vs fp64 (the reference arithmetic's own noise).  python tools/fullsize_grad_report.py [B] [full] [precision]"""
import dataclasses
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import scvae_oracle as O  # noqa: E402
from tests.conftest import _usable_cpus  # noqa: E402
from tests.test_oracle_golden import ARENA, FULL_METHODS  # noqa: E402
from tests.test_gpu_model import build_model, to_dev  # noqa: E402

torch.set_num_threads(_usable_cpus())
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
full = len(sys.argv) > 2 and sys.argv[2] == "full"
precision = sys.argv[3] if len(sys.argv) > 3 else "bf16x6b3"
from scrubvae_amd import ops  # noqa: E402
from scrubvae_amd.train.losses import get_batch_loss  # noqa: E402

feats = ["avg_speed_3d", "heading"]
cfg = O.OracleConfig(n_keypts=23, window=64, z_dim=32, kernel=5, diag=True, arena_size=ARENA, kinematic_tree=O.skeleton_tree(23),
                     method=dict(FULL_METHODS) if full else {}, features=feats if full else None,
                     discrete_classes={"ids": torch.arange(4)} if full else None)
ls = {"jpe": 1.0, "root": 1.0, "prior": 1.0}
if full:
    ls.update({"avg_speed_3d_gr": 1.0, "heading_gr": 1.0, "heading_an": 1.0})
seed = 41 if not full else 51
sd = O.init_state_dict(cfg, seed=seed)
data = O.synth_batch(cfg, B, seed=seed + 1)
g = torch.Generator().manual_seed(seed + 2)
eps = torch.randn(B, cfg.z_dim, generator=g)
perm = {k: torch.randperm(B, generator=g) for k in cfg.method.get("adversarial_net", [])}
bl32, g32, _, _ = O.train_step(sd, cfg, data, ls, eps, adv_perm=perm)
c64 = dataclasses.replace(cfg, arena_size=ARENA.double())
sd64 = {k: (v.double() if v.dtype.is_floating_point else v) for k, v in sd.items()}
d64 = {k: (v.double() if v.dtype.is_floating_point else v) for k, v in data.items()}
bl64, g64, _, _ = O.train_step(sd64, c64, d64, ls, eps.double(), adv_perm=perm)
ops.set_precision(precision)
model, dis = build_model(cfg, sd)
model.train()
model.defer_tail = True
d = to_dev(data)
d["eps"] = eps.cuda()
bl = get_batch_loss(model, d, model(d), ls, dis, adv_perm=perm or None)
bl["total"].backward()
torch.cuda.synchronize()
gh = {k: v.cpu() for k, v in model.grads_state_dict().items()}
gmax = max(float(x.abs().max()) for x in g64.values())
print(f"B={B} full={full} {precision}; losses rel dev vs fp64: " + ", ".join(f"{k} hip {abs(float(bl[k])-float(bl64[k]))/abs(float(bl64[k])):.1e} cpu32 {abs(float(bl32[k])-float(bl64[k]))/abs(float(bl64[k])):.1e}" for k in bl64))
rows = []
for n, t in g64.items():
    den = float(t.abs().max()) + 1e-3 * gmax
    rows.append((float((gh[n].double() - t).abs().max()) / den, float((g32[n].double() - t).abs().max()) / den,
                 float((gh[n] - g32[n]).abs().max()) / (float(g32[n].abs().max()) + 1e-3 * gmax), n))
rows.sort(reverse=True)
print("   hip-vs-64   cpu32-vs-64   hip-vs-cpu32   tensor")
for r in rows[:25]:
    print(f"   {r[0]:.2e}    {r[1]:.2e}      {r[2]:.2e}     {r[3]}")
nrm = lambda a: torch.sqrt(sum((x.double() ** 2).sum() for x in a))
tot64 = nrm(g64.values())
print(f"|g| fp64 {float(tot64):.6g}; |hip - 64|/|64| = {float(nrm([gh[n].double() - g64[n] for n in g64]) / tot64):.2e}; "
      f"|cpu32 - 64|/|64| = {float(nrm([g32[n].double() - g64[n] for n in g64]) / tot64):.2e}; "
      f"norm rel dev hip {abs(float(nrm([gh[n] for n in g64])) - float(tot64)) / float(tot64):.2e} cpu32 {abs(float(nrm(g32.values())) - float(tot64)) / float(tot64):.2e}")

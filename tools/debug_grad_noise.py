"""Per-tensor gradient error of the HIP path and of the fp32 CPU oracle against the fp64 oracle."""
import dataclasses, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from oracle import scvae_oracle as O
from tests.test_oracle_golden import load_fixture, ARENA
from tests.test_gpu_model import build_model, to_dev
from scrubvae_amd.train.losses import get_batch_loss

name = sys.argv[1] if len(sys.argv) > 1 else "vanilla_tiny"
fx, cfg, loss_scale, opt, sd, data = load_fixture("tests/golden", name)
eps, perm = torch.from_numpy(fx["eps/0"]), torch.from_numpy(fx["perm/0"])
ap = {k: perm for k in cfg.method.get("adversarial_net", [])}
bl32, g32, _, _ = O.train_step(sd, cfg, data, loss_scale, eps, adv_perm=ap)
c64 = dataclasses.replace(cfg, arena_size=ARENA.double())
sd64 = {k: (v.double() if v.dtype.is_floating_point else v) for k, v in sd.items()}
d64 = {k: (v.double() if v.dtype.is_floating_point else v) for k, v in data.items()}
bl64, g64, _, _ = O.train_step(sd64, c64, d64, loss_scale, eps.double(), adv_perm=ap)
model, dis = build_model(cfg, sd)
model.train()
d = to_dev(data); d["eps"] = eps.cuda()
data_o = model(d)
bl = get_batch_loss(model, d, data_o, loss_scale, dis, adv_perm=ap)
bl["total"].backward()
grads = {k: v.cpu() for k, v in model.grads_state_dict().items()}
gmax = max(float(g.abs().max()) for g in g64.values())
rows = []
for n, g in g64.items():
    den = float(g.abs().max()) + 1e-3 * gmax
    rows.append((float((grads[n].double() - g).abs().max()) / den, float((g32[n].double() - g).abs().max()) / den, n, float(g.abs().max())))
rows.sort(reverse=True)
for r in rows[:25]:
    print("hip %.2e cpu32 %.2e  %-50s gmax %.3g" % r)
out32 = O.forward(sd, cfg, data, True, eps=eps)
out64 = O.forward(sd64, c64, d64, True, eps=eps.double())
def r(a, b):
    return float((a.double() - b.double()).abs().max() / b.double().abs().max())
for k in ("mu", "z", "x6d", "root"):
    print(k, "hip %.2e cpu32 %.2e" % (r(data_o[k].cpu(), out64[k]), r(out32[k], out64[k])))
for k in bl64:
    print("loss", k, "hip %.2e cpu32 %.2e" % (r(bl[k].detach().cpu(), bl64[k]), r(bl32[k], bl64[k])))

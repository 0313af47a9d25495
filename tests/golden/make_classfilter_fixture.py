"""Golden fixtures for the class-filter scrubbers: three training steps of the REAL reference with
`disentangle.method = {moving_avg: [ids]}` and `{qda: [ids]}` on CPU.

    python -B tests/golden/make_classfilter_fixture.py        (build container only)
"""
import os
import sys

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)

import numpy as np
import torch

import make_fixtures as MF
from oracle import scvae_oracle as O

BUFS = {"moving_avg": ("m1", "m2", "lam1", "lam2"),
        "qda": ("m0a", "m1a", "m0b", "m1b", "S0a", "S1a", "S0b", "S1b", "lama", "lamb")}
LOSS = {"moving_avg": {"jpe": 1.0, "root": 1.0, "prior": 0.5, "ids_ma": 0.8},
        "qda": {"jpe": 1.0, "root": 1.0, "prior": 0.5, "ids_qda": 0.05}}


def run(method, refs):
    get_model, get_batch_loss, predict_batch = refs
    arena = torch.tensor([[-1.0, -1.0, -1.0], [1.0, 1.0, 1.0]])
    classes = {"ids": torch.arange(4)}
    cfg = O.OracleConfig(diag=True, method={method: ["ids"]}, features=["ids"], discrete_classes=classes, n_keypts=18, window=64,
                         z_dim=8, kernel=5, channel=(8, 8, 16, 16, 32), arena_size=arena)
    B, n_steps, lr = 16, 3, 1e-4
    name = {"moving_avg": "maf_tiny", "qda": "qda_tiny"}[method]
    sd = {k: v for k, v in O.init_state_dict(cfg, seed=len(name)).items() if not k.startswith("disentangle.")}
    data = O.synth_batch(cfg, B, seed=len(name))
    data["ids"] = (torch.arange(B) % 4).reshape(B, 1).to(torch.int16)  # every class present in the batch
    g = torch.Generator().manual_seed(7)
    eps_all = [torch.randn(B, cfg.z_dim, generator=g) for _ in range(n_steps)]
    model_config = dict(type="rcnn", kernel=cfg.kernel, z_dim=cfg.z_dim, window=cfg.window, activation="prelu", diag=True,
                        init_dilation=None, prior="gaussian", channel=list(cfg.channel))
    dis_config = dict(method=cfg.method, alpha=1.0, features=["ids"])
    loss = LOSS[method]
    model = get_model(model_config, None, None, dis_config, cfg.n_keypts, "midfwd", loss_config=loss, arena_size=arena,
                      kinematic_tree=cfg.kinematic_tree, bound=False, discrete_classes=classes, device="cpu", verbose=0)
    missing, unexpected = model.load_state_dict(sd, strict=False)
    assert not unexpected and all(m.startswith("disentangle.") for m in missing), (missing, unexpected)
    fx = {}
    for k, v in data.items():
        fx["in/" + k] = v.numpy()
    for k, v in sd.items():
        fx["sd/" + k] = v.numpy()
    for i, e in enumerate(eps_all):
        fx[f"eps/{i}"] = e.numpy()
    model.train()
    opt = torch.optim.AdamW(model.parameters(), lr=lr)
    for step in range(n_steps):
        with MF.Patch(eps_all[step], torch.arange(B)):
            data_o = predict_batch(model, data, model.disentangle_keys)
            bl = get_batch_loss(model, data, data_o, loss, dis_config)
        for p in model.parameters():
            p.grad = None
        bl["total"].backward()
        torch.nn.utils.clip_grad_norm_(model.parameters(), max_norm=1e6)
        if step == 0:
            for n, p in model.named_parameters():
                if p.grad is not None and n.startswith("encoder.fc_mu"):
                    fx["s0/grad/" + n] = p.grad.numpy().copy()
        opt.step()
        model.disentangle[method]["ids"].update(data_o["mu"].detach().clone(), data["ids"].detach().clone())  # trainer.py:170-180
        for k, v in bl.items():
            fx[f"s{step}/loss/{k}"] = v.detach().numpy()
        fx[f"s{step}/mu"] = data_o["mu"].detach().numpy()
        for b in BUFS[method]:
            fx[f"s{step}/{b}"] = getattr(model.disentangle[method]["ids"], b).detach().numpy().copy()
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **fx)
    print(f"saved {path} {os.path.getsize(path)/1e6:.2f} MB", {k: round(float(v), 5) for k, v in bl.items()})


def run_lsq(refs):
    """direct_lsq (stateless least-squares decoder loss, losses.py:173-179): one step, losses + encoder-head gradient."""
    get_model, get_batch_loss, predict_batch = refs
    arena = torch.tensor([[-1.0, -1.0, -1.0], [1.0, 1.0, 1.0]])
    feats = ["avg_speed_3d", "heading"]
    cfg = O.OracleConfig(diag=True, method={"direct_lsq": feats}, features=feats, n_keypts=18, window=64, z_dim=8, kernel=5,
                         channel=(8, 8, 16, 16, 32), arena_size=arena)
    B, name = 16, "lsq_tiny"
    loss = {"jpe": 1.0, "root": 1.0, "prior": 0.5, "avg_speed_3d_lsq": 0.4, "heading_lsq": 0.25}
    sd = O.init_state_dict(cfg, seed=len(name))
    data = O.synth_batch(cfg, B, seed=len(name))
    eps = torch.randn(B, cfg.z_dim, generator=torch.Generator().manual_seed(7))
    model_config = dict(type="rcnn", kernel=cfg.kernel, z_dim=cfg.z_dim, window=cfg.window, activation="prelu", diag=True,
                        init_dilation=None, prior="gaussian", channel=list(cfg.channel))
    dis_config = dict(method=cfg.method, alpha=1.0, features=feats)
    model = get_model(model_config, None, None, dis_config, cfg.n_keypts, "midfwd", loss_config=loss, arena_size=arena,
                      kinematic_tree=cfg.kinematic_tree, bound=False, discrete_classes=None, device="cpu", verbose=0)
    missing, unexpected = model.load_state_dict(sd, strict=False)
    assert not missing and not unexpected, (missing, unexpected)
    model.train()
    with MF.Patch(eps, torch.arange(B)):
        data_o = predict_batch(model, data, model.disentangle_keys)
        bl = get_batch_loss(model, data, data_o, loss, dis_config)
    bl["total"].backward()
    fx = {"eps/0": eps.numpy(), "s0/mu": data_o["mu"].detach().numpy()}
    for k, v in data.items():
        fx["in/" + k] = v.numpy()
    for k, v in sd.items():
        fx["sd/" + k] = v.numpy()
    for k, v in bl.items():
        fx["s0/loss/" + k] = v.detach().numpy()
    for n, p in model.named_parameters():
        if p.grad is not None and n.startswith("encoder.fc_mu"):
            fx["s0/grad/" + n] = p.grad.numpy().copy()
    for k in feats:
        print("  oracle direct_lsq", k, float(O.direct_lsq_loss(data_o["mu"].detach(), data[k])), float(bl[k + "_lsq"]))
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **fx)
    print(f"saved {path} {os.path.getsize(path)/1e6:.2f} MB", {k: round(float(v), 5) for k, v in bl.items()})


if __name__ == "__main__":
    refs = MF.import_reference()
    if "--lsq-only" not in sys.argv:
        for method in ("moving_avg", "qda"):
            run(method, refs)
    run_lsq(refs)

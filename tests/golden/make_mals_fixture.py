"""Golden fixture for the streaming least-squares scrubber: three training steps
of the REAL reference with `disentangle.method = {moving_avg_lsq: [...]}` on CPU.

    python -B tests/golden/make_mals_fixture.py        (build container only)

One feature uses a negative loss scale (the scrubbing sign; it switches the reference's `bias` option on,
get/model.py:81) and one a positive scale.  The reference hard-codes device="cuda" for the bias column in
`update` (disentangle.py:494-497); with no GPU in this container torch.ones is given a CPU device for that call
only -- a stand-in for absent hardware, not for reference code.
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

FEATS = ["avg_speed_3d", "heading"]
LOSS = {"jpe": 1.0, "root": 1.0, "prior": 0.5, "avg_speed_3d_mals": -0.7, "heading_mals": 0.3}


def main(refs, name="mals_tiny", polynomial=1, l2_reg=0):
    get_model, get_batch_loss, predict_batch = refs
    arena = torch.tensor([[-1.0, -1.0, -1.0], [1.0, 1.0, 1.0]])
    cfg = O.OracleConfig(diag=True, method={"moving_avg_lsq": FEATS}, features=FEATS, n_keypts=18, window=64, z_dim=8,
                         kernel=5, channel=(8, 8, 16, 16, 32), arena_size=arena)
    B, n_steps, lr = 8, 3, 1e-4
    sd = {k: v for k, v in O.init_state_dict(cfg, seed=9).items() if not k.startswith("disentangle.")}
    data = O.synth_batch(cfg, B, seed=9)
    g = torch.Generator().manual_seed(7)
    eps_all = [torch.randn(B, cfg.z_dim, generator=g) for _ in range(n_steps)]
    model_config = dict(type="rcnn", kernel=cfg.kernel, z_dim=cfg.z_dim, window=cfg.window, activation="prelu", diag=True,
                        init_dilation=None, prior="gaussian", channel=list(cfg.channel))
    dis_config = dict(method=cfg.method, alpha=1.0, features=FEATS, polynomial=polynomial, l2_reg=l2_reg)
    model = get_model(model_config, None, None, dis_config, cfg.n_keypts, "midfwd", loss_config=LOSS, arena_size=arena,
                      kinematic_tree=cfg.kinematic_tree, bound=False, discrete_classes=None, device="cpu", verbose=0)
    missing, unexpected = model.load_state_dict(sd, strict=False)
    assert not unexpected and all(m.startswith("disentangle.") for m in missing), (missing, unexpected)
    fx = {"sd_seed": np.int64(9), "polynomial": np.int64(polynomial), "l2_reg": np.float64(l2_reg)}
    for k, v in data.items():
        fx["in/" + k] = v.numpy()
    for k, v in sd.items():
        fx["sd/" + k] = v.numpy()
    for i, e in enumerate(eps_all):
        fx[f"eps/{i}"] = e.numpy()
    model.train()
    opt = torch.optim.AdamW(model.parameters(), lr=lr)
    real_ones = torch.ones

    def cpu_ones(*a, **k):
        if k.get("device") == "cuda":
            k["device"] = "cpu"
        return real_ones(*a, **k)

    for step in range(n_steps):
        with MF.Patch(eps_all[step], torch.arange(B)):
            data_o = predict_batch(model, data, model.disentangle_keys)
            bl = get_batch_loss(model, data, data_o, LOSS, dis_config)
        for p in model.parameters():
            p.grad = None
        bl["total"].backward()
        torch.nn.utils.clip_grad_norm_(model.parameters(), max_norm=1e6)
        for n, p in model.named_parameters():  # from step 1 on (non-zero decoders) these carry d mals / d mu
            if p.grad is not None and n.startswith("encoder.fc_mu"):
                fx[f"s{step}/grad/" + n] = p.grad.numpy().copy()
        opt.step()
        torch.ones = cpu_ones
        try:
            for k in FEATS:  # trainer.py:170-180
                model.disentangle["moving_avg_lsq"][k].update(data_o["mu"].detach().clone(), data[k].detach().clone())
        finally:
            torch.ones = real_ones
        for k, v in bl.items():
            fx[f"s{step}/loss/{k}"] = v.detach().numpy()
        fx[f"s{step}/mu"] = data_o["mu"].detach().numpy()
        for k in FEATS:
            m = model.disentangle["moving_avg_lsq"][k]
            for i in range(2):
                fx[f"s{step}/yhat/{k}/{i}"] = data_o["disentangle"]["moving_avg_lsq"][k][i].detach().numpy()
            for b in ("Sxx0", "Sxy0", "Sxx1", "Sxy1", "lam0", "lam1"):
                fx[f"s{step}/{k}/{b}"] = getattr(m, b).detach().numpy()
    for n, v in model.state_dict().items():
        if v.dtype.is_floating_point and (n.startswith("encoder.fc_mu") or n.startswith("disentangle.")):
            fx["final_sd/" + n] = v.detach().numpy()
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **fx)
    print(f"saved {path} {os.path.getsize(path)/1e6:.2f} MB")
    print({k: float(v) for k, v in bl.items()})


if __name__ == "__main__":
    refs = MF.import_reference()
    main(refs)
    main(refs, "mals_poly2_tiny", polynomial=2, l2_reg=0.05)

"""Golden fixture for the `mcmi` loss (kernel-density mutual information): three training steps of the
REAL reference on CPU with `loss.mcmi`, a conditional decoder, and the estimator refreshed after every step exactly as
trainer.py:184-199 does (re-encode the batch with the updated weights, centres = those means / the batch's `var`), for
both `var_mode`s.

    python -B tests/golden/make_mcmi_fixture.py        (build container only)
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
LOSS = {"jpe": 1.0, "root": 1.0, "prior": 0.5, "mcmi": 0.7}
BANDWIDTH = 0.5


def main():
    get_model, get_batch_loss, predict_batch = MF.import_reference()
    from scrubvae.model.disentangle import MutInfoEstimator
    arena = torch.tensor([[-1.0, -1.0, -1.0], [1.0, 1.0, 1.0]])
    cfg = O.OracleConfig(diag=True, method={"conditional": FEATS}, features=FEATS, n_keypts=18, window=64, z_dim=8, kernel=5,
                         channel=(8, 8, 16, 16, 32), arena_size=arena, discrete_classes={"ids": torch.arange(4)})
    B, n_steps, lr = 16, 3, 1e-4
    sd = O.init_state_dict(cfg, seed=9)
    batches = [O.synth_batch(cfg, B, seed=9), O.synth_batch(cfg, B, seed=10)]  # step s trains on batches[s % 2]
    g = torch.Generator().manual_seed(7)
    eps_all = [torch.randn(B, cfg.z_dim, generator=g) for _ in range(n_steps)]
    model_config = dict(type="rcnn", kernel=cfg.kernel, z_dim=cfg.z_dim, window=cfg.window, activation="prelu", diag=True,
                        init_dilation=None, prior="gaussian", channel=list(cfg.channel))
    fx = {"bandwidth": np.float64(BANDWIDTH)}
    for i, data in enumerate(batches):
        for k, v in data.items():
            fx[f"in{i}/" + k] = v.numpy()
    for k, v in sd.items():
        fx["sd/" + k] = v.numpy()
    for i, e in enumerate(eps_all):
        fx[f"eps/{i}"] = e.numpy()
    for var_mode in ("sphere", "diagonal"):
        dis_config = dict(method=cfg.method, alpha=1.0, features=FEATS, bandwidth=BANDWIDTH, var_mode=var_mode)
        model = get_model(model_config, None, None, dis_config, cfg.n_keypts, "midfwd", loss_config=LOSS, arena_size=arena,
                          kinematic_tree=cfg.kinematic_tree, bound=False, discrete_classes={"ids": torch.arange(4)}, device="cpu", verbose=0)
        missing, unexpected = model.load_state_dict(sd, strict=False)
        assert not missing and not unexpected, (missing, unexpected)
        model.train()
        model.mi_estimator = None  # trainer.py:124
        opt = torch.optim.AdamW(model.parameters(), lr=lr)
        for step in range(n_steps):
            data = batches[step % 2]
            with MF.Patch(eps_all[step], torch.arange(B)):
                data_o = predict_batch(model, data, model.disentangle_keys)
                bl = get_batch_loss(model, data, data_o, LOSS, dis_config)
            for p in model.parameters():
                p.grad = None
            bl["total"].backward()
            torch.nn.utils.clip_grad_norm_(model.parameters(), max_norm=1e6)
            pre = f"{var_mode}/s{step}/"
            for n, p in model.named_parameters():
                if p.grad is not None and n.startswith("encoder.fc_mu"):
                    fx[pre + "grad/" + n] = p.grad.numpy().copy()
            opt.step()
            # trainer.py:184-199
            updated = model.encode(data)
            model.mi_estimator = MutInfoEstimator(
                x_s=updated["mu"].detach().clone(), y_s=data_o["var"].clone(), bandwidth=BANDWIDTH, var_mode=var_mode,
                model_var=updated["L"].detach().clone() if "L" in updated.keys() else None, device="cpu")
            for k, v in bl.items():
                fx[pre + "loss/" + k] = v.detach().numpy()
            fx[pre + "mu"] = data_o["mu"].detach().numpy()
            fx[pre + "var"] = data_o["var"].detach().numpy()
            fx[pre + "x_s"] = model.mi_estimator.x_s.numpy().copy()
            fx[pre + "var_s"] = model.mi_estimator.var_s.numpy().copy()
            if step:
                prev = f"{var_mode}/s{step - 1}/"
                o_val = O.mcmi_value(fx[prev + "x_s"], fx[prev + "var"], fx[prev + "var_s"], BANDWIDTH, fx[pre + "mu"], fx[pre + "var"])
                print(var_mode, step, "mcmi oracle / reference", o_val, float(bl["mcmi"]))
    path = os.path.join(HERE, "mcmi_tiny.npz")
    np.savez_compressed(path, **fx)
    print(f"saved {path} {os.path.getsize(path)/1e6:.2f} MB")


if __name__ == "__main__":
    main()

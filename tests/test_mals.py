"""Streaming least-squares scrubber against three training steps of the real reference
(tests/golden/mals_tiny.npz, make_mals_fixture.py)."""
import os

import numpy as np
import pytest
import torch

from oracle import scvae_oracle as O
from tests.test_oracle_golden import ARENA, rel

FEATS = ["avg_speed_3d", "heading"]
LOSS = {"jpe": 1.0, "root": 1.0, "prior": 0.5, "avg_speed_3d_mals": -0.7, "heading_mals": 0.3}
CFG = O.OracleConfig(diag=True, method={"moving_avg_lsq": FEATS}, features=FEATS, n_keypts=18, window=64, z_dim=8, kernel=5,
                     channel=(8, 8, 16, 16, 32), arena_size=ARENA)


NAMES = ["mals_tiny", "mals_poly2_tiny"]  # polynomial 1 / l2 0, and polynomial 2 / l2 0.05


def load(golden_dir, name="mals_tiny"):
    fx = np.load(os.path.join(golden_dir, name + ".npz"))
    sd = {k[3:]: torch.from_numpy(fx[k]) for k in fx.files if k.startswith("sd/")}
    data = {k[3:]: torch.from_numpy(fx[k]) for k in fx.files if k.startswith("in/")}
    return fx, sd, data


@pytest.mark.parametrize("name", NAMES)
def test_oracle_mals_matches_reference(golden_dir, name):
    """The oracle's MALS restatement driven with the reference's latent means reproduces its predictions, losses,
    forgetting factors and covariance buffers over three steps."""
    fx, sd, data = load(golden_dir, name)
    B = data["x6d"].shape[0]
    poly, l2 = int(fx["polynomial"]), float(fx["l2_reg"])
    st = {k: O.mals_init(CFG.z_dim, data[k].shape[-1], bias=LOSS[k + "_mals"] < 0, polynomial_order=poly) for k in FEATS}
    for step in range(3):
        mu = torch.from_numpy(fx[f"s{step}/mu"])
        for k in FEATS:
            y0, y1 = O.mals_forward(st[k], mu, l2_reg=l2)
            assert rel(y0, fx[f"s{step}/yhat/{k}/0"]) < 1e-4 and rel(y1, fx[f"s{step}/yhat/{k}/1"]) < 1e-4, (step, k)
            loss, st[k] = O.mals_loss(st[k], y0, y1, data[k])
            assert rel(loss / B, fx[f"s{step}/loss/{k}_mals"]) < 1e-4, (step, k)
            st[k] = O.mals_update(st[k], mu, data[k])
            for b in ("Sxx0", "Sxy0", "Sxx1", "Sxy1", "lam0", "lam1"):
                assert rel(st[k][b], fx[f"s{step}/{k}/{b}"]) < 1e-5, (step, k, b)


@pytest.mark.parametrize("poly,bias", [(1, False), (2, True), (3, False)])
def test_latent_seed_is_the_gradient_of_the_loss(poly, bias):
    """The analytic seed the HIP backward receives == autograd of the scrubber's loss through forward() (CPU tensors)."""
    from scrubvae_amd.model.disentangle import MovingAvgLeastSquares
    g = torch.Generator().manual_seed(poly)
    m = MovingAvgLeastSquares(5, 3, bias=bias, polynomial_order=poly, l2_reg=0.01)
    assert m.Sxx0.shape[0] == {1: 5, 2: 20, 3: 55}[poly] + int(bias)
    m.update(torch.randn(64, 5, generator=g), torch.randn(64, 3, generator=g))
    x = torch.randn(16, 5, generator=g, dtype=torch.float32).requires_grad_(True)
    y = torch.randn(16, 3, generator=g)
    y0, y1 = m(x)
    loss = 0.5 * (((y - y0) ** 2).sum() + ((y - y1) ** 2).sum())
    want = torch.autograd.grad(loss, x)[0]
    got = m.latent_seed(y0.detach(), y1.detach(), y, x.detach(), 1.0)
    assert rel(got, want) < 1e-5


@pytest.mark.gpu
@pytest.mark.parametrize("name", NAMES)
def test_hip_model_with_mals_matches_reference(golden_dir, name):
    """scrubvae_amd model with the moving_avg_lsq scrubbers: losses, predictions, forgetting factors, covariance
    buffers and the encoder-head gradient/updates over three reference-style training steps."""
    from scrubvae_amd.get import model as get_model
    from scrubvae_amd.train.losses import get_batch_loss
    from scrubvae_amd.train.trainer import clip_grad_norm_
    fx, sd, data = load(golden_dir, name)
    model_config = dict(type="rcnn", kernel=CFG.kernel, z_dim=CFG.z_dim, window=CFG.window, activation="prelu", diag=True,
                        init_dilation=None, prior="gaussian", channel=list(CFG.channel))
    dis = dict(method=CFG.method, alpha=1.0, features=FEATS, polynomial=int(fx["polynomial"]), l2_reg=float(fx["l2_reg"]))
    m = get_model(model_config, None, None, dis, CFG.n_keypts, "midfwd", loss_config=LOSS, arena_size=ARENA,
                  kinematic_tree=CFG.kinematic_tree, device="cuda", verbose=0)
    missing, unexpected = m.load_state_dict(sd, strict=False)
    assert not unexpected and all(k.startswith("disentangle.") for k in missing)
    m.train()
    opt = torch.optim.AdamW(m.parameters(), lr=1e-4)
    dev = {k: v.cuda() for k, v in data.items()}
    for step in range(3):
        batch = dict(dev, eps=torch.from_numpy(fx[f"eps/{step}"]).cuda())
        data_o = m(batch)
        bl = get_batch_loss(m, batch, data_o, LOSS, dis)
        for p in m.parameters():
            p.grad = None
        bl["total"].backward()
        clip_grad_norm_(m, max_norm=1e6)
        g = m.grads_state_dict()
        for n in ("encoder.fc_mu.weight", "encoder.fc_mu.bias"):  # from step 1 on they carry the scrubbers' analytic seed on mu
            want = torch.from_numpy(fx[f"s{step}/grad/" + n])
            assert rel(g[n].cpu(), want) < (2e-3 if step == 0 else 5e-2), (step, n)  # later steps: weights after Adam steps on noisy grads
        opt.step()
        for k in FEATS:
            m.disentangle["moving_avg_lsq"][k].update(data_o["mu"].detach().clone(), batch[k].detach().clone())
        tol = 1e-4 if step == 0 else 2e-3  # later steps see weights after Adam steps on noisy gradients (cf. test_gpu_model)
        for k in fx.files:
            if k.startswith(f"s{step}/loss/"):
                assert rel(bl[k.split("/")[-1]].detach().cpu(), fx[k]) < tol, (step, k)
        assert rel(data_o["mu"].detach().cpu(), fx[f"s{step}/mu"]) < (5e-5 if step == 0 else 5e-3)
        for k in FEATS:
            s = m.disentangle["moving_avg_lsq"][k]
            for i in range(2):
                assert rel(data_o["disentangle"]["moving_avg_lsq"][k][i].cpu(), fx[f"s{step}/yhat/{k}/{i}"]) < (1e-4 if step == 0 else 2e-2)
            for b in ("lam0", "lam1"):
                assert rel(getattr(s, b).cpu(), fx[f"s{step}/{k}/{b}"]) < 1e-6, (step, k, b)
            for b in ("Sxx0", "Sxy0", "Sxx1", "Sxy1"):
                assert rel(getattr(s, b).cpu(), fx[f"s{step}/{k}/{b}"]) < (1e-4 if step == 0 else 5e-3), (step, k, b)
    fin = m.state_dict()
    for n in ("encoder.fc_mu.weight", "encoder.fc_mu.bias"):
        assert float((fin[n].cpu() - torch.from_numpy(fx["final_sd/" + n])).abs().max()) < 3 * 1e-4 + 1e-6, n

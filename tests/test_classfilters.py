"""Class-filter scrubbers (MovingAverageFilter, QuadraticDiscriminantFilter) against three training
steps of the real reference (tests/golden/{maf,qda}_tiny.npz, make_classfilter_fixture.py)."""
import os

import numpy as np
import pytest
import torch

from oracle import scvae_oracle as O
from tests.test_oracle_golden import ARENA, rel

CLASSES = torch.arange(4)
CASES = {"moving_avg": ("maf_tiny", "ids_ma", {"jpe": 1.0, "root": 1.0, "prior": 0.5, "ids_ma": 0.8}, ("m1", "m2", "lam1", "lam2")),
         "qda": ("qda_tiny", "ids_qda", {"jpe": 1.0, "root": 1.0, "prior": 0.5, "ids_qda": 0.05},
                 ("m0a", "m1a", "m0b", "m1b", "S0a", "S1a", "S0b", "S1b", "lama", "lamb"))}


def cfg_for(method):
    return O.OracleConfig(diag=True, method={method: ["ids"]}, features=["ids"], discrete_classes={"ids": CLASSES}, n_keypts=18,
                          window=64, z_dim=8, kernel=5, channel=(8, 8, 16, 16, 32), arena_size=ARENA)


def load(golden_dir, name):
    fx = np.load(os.path.join(golden_dir, name + ".npz"))
    sd = {k[3:]: torch.from_numpy(fx[k]) for k in fx.files if k.startswith("sd/")}
    data = {k[3:]: torch.from_numpy(fx[k]) for k in fx.files if k.startswith("in/")}
    return fx, sd, data


@pytest.mark.parametrize("method", ["moving_avg", "qda"])
def test_oracle_classfilter_matches_reference(golden_dir, method):
    name, lk, loss, bufs = CASES[method]
    fx, sd, data = load(golden_dir, name)
    B = data["x6d"].shape[0]
    st = O.maf_init(8, 4) if method == "moving_avg" else O.qda_init(8, 4)
    for step in range(3):
        mu = torch.from_numpy(fx[f"s{step}/mu"])
        if method == "moving_avg":
            val, st = O.maf_loss(st, mu, data["ids"], CLASSES)
            st = O.maf_update(st, mu, data["ids"], CLASSES)
        else:
            val, st = O.qda_loss(st, mu, data["ids"], CLASSES)
            val = val / B
            st = O.qda_update(st, mu, data["ids"], CLASSES)
        assert rel(val, fx[f"s{step}/loss/{lk}"]) < 1e-4, step
        for b in bufs:
            assert rel(st[b], fx[f"s{step}/{b}"]) < 1e-5, (step, b)


@pytest.mark.gpu
@pytest.mark.parametrize("method", ["moving_avg", "qda"])
def test_hip_model_with_classfilter_matches_reference(golden_dir, method):
    from scrubvae_amd.get import model as get_model
    from scrubvae_amd.train.losses import get_batch_loss
    from scrubvae_amd.train.trainer import clip_grad_norm_
    name, lk, loss, bufs = CASES[method]
    fx, sd, data = load(golden_dir, name)
    cfg = cfg_for(method)
    mc = dict(type="rcnn", kernel=cfg.kernel, z_dim=cfg.z_dim, window=cfg.window, activation="prelu", diag=True, init_dilation=None,
              prior="gaussian", channel=list(cfg.channel))
    dis = dict(method=cfg.method, alpha=1.0, features=["ids"])
    m = get_model(mc, None, None, dis, cfg.n_keypts, "midfwd", loss_config=loss, arena_size=ARENA, kinematic_tree=cfg.kinematic_tree,
                  discrete_classes={"ids": CLASSES}, device="cuda", verbose=0)
    missing, unexpected = m.load_state_dict(sd, strict=False)
    assert not unexpected and all(k.startswith("disentangle.") for k in missing)
    m.train()
    opt = torch.optim.AdamW(m.parameters(), lr=1e-4)
    dev = {k: v.cuda() for k, v in data.items()}
    for step in range(3):
        batch = dict(dev, eps=torch.from_numpy(fx[f"eps/{step}"]).cuda())
        data_o = m(batch)
        bl = get_batch_loss(m, batch, data_o, loss, dis)
        for p in m.parameters():
            p.grad = None
        bl["total"].backward()
        clip_grad_norm_(m, max_norm=1e6)
        if step == 0:
            g = m.grads_state_dict()
            for n in ("encoder.fc_mu.weight", "encoder.fc_mu.bias"):
                assert rel(g[n].cpu(), torch.from_numpy(fx["s0/grad/" + n])) < 2e-3, n  # carries the filter's autograd seed on mu
        opt.step()
        m.disentangle[method]["ids"].update(data_o["mu"].detach().clone(), batch["ids"].detach().clone())
        tol = 1e-4 if step == 0 else 5e-3
        for k in fx.files:
            if k.startswith(f"s{step}/loss/"):
                assert rel(bl[k.split("/")[-1]].detach().cpu(), fx[k]) < tol, (step, k)
        s = m.disentangle[method]["ids"]
        for b in bufs:
            assert rel(getattr(s, b).cpu(), fx[f"s{step}/{b}"]) < (1e-4 if step == 0 else 1e-2), (step, b)


# ------------------------------------------------------------------ direct_lsq (stateless least-squares decoder loss)
LSQ_FEATS = ["avg_speed_3d", "heading"]
LSQ_LOSS = {"jpe": 1.0, "root": 1.0, "prior": 0.5, "avg_speed_3d_lsq": 0.4, "heading_lsq": 0.25}


def test_oracle_direct_lsq_matches_reference(golden_dir):
    fx, sd, data = load(golden_dir, "lsq_tiny")
    mu = torch.from_numpy(fx["s0/mu"])
    for k in LSQ_FEATS:
        assert rel(O.direct_lsq_loss(mu, data[k]), fx[f"s0/loss/{k}_lsq"]) < 1e-5, k


@pytest.mark.gpu
def test_hip_model_with_direct_lsq_matches_reference(golden_dir):
    """losses and the encoder-head gradient (which carries the 2 * res * W^T seed) vs one step of the real reference."""
    from scrubvae_amd.get import model as get_model
    from scrubvae_amd.train.losses import get_batch_loss
    fx, sd, data = load(golden_dir, "lsq_tiny")
    cfg = O.OracleConfig(diag=True, method={"direct_lsq": LSQ_FEATS}, features=LSQ_FEATS, n_keypts=18, window=64, z_dim=8, kernel=5,
                         channel=(8, 8, 16, 16, 32), arena_size=ARENA)
    mc = dict(type="rcnn", kernel=cfg.kernel, z_dim=cfg.z_dim, window=cfg.window, activation="prelu", diag=True, init_dilation=None,
              prior="gaussian", channel=list(cfg.channel))
    dis = dict(method=cfg.method, alpha=1.0, features=LSQ_FEATS)
    m = get_model(mc, None, None, dis, cfg.n_keypts, "midfwd", loss_config=LSQ_LOSS, arena_size=ARENA, kinematic_tree=cfg.kinematic_tree,
                  device="cuda", verbose=0)
    missing, unexpected = m.load_state_dict(sd, strict=False)
    assert not missing and not unexpected
    m.train()
    batch = {k: v.cuda() for k, v in data.items()}
    batch["eps"] = torch.from_numpy(fx["eps/0"]).cuda()
    bl = get_batch_loss(m, batch, m(batch), LSQ_LOSS, dis)
    for k in fx.files:
        if k.startswith("s0/loss/"):
            assert rel(bl[k.split("/")[-1]].detach().cpu(), fx[k]) < 1e-4, k
    bl["total"].backward()
    g = m.grads_state_dict()
    for n in ("encoder.fc_mu.weight", "encoder.fc_mu.bias"):
        assert rel(g[n].cpu(), torch.from_numpy(fx["s0/grad/" + n])) < 2e-3, n

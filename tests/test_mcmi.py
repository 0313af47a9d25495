"""`mcmi` loss (MutInfoEstimator) against three training steps of the real reference with the estimator
refreshed after every step (tests/golden/mcmi_tiny.npz, make_mcmi_fixture.py), both var_modes; step s trains on batch s % 2
so that the estimator's centres are not the evaluated batch."""
import os

import numpy as np
import pytest
import torch

from oracle import scvae_oracle as O
from tests.test_oracle_golden import ARENA, rel

FEATS = ["avg_speed_3d", "heading"]
LOSS = {"jpe": 1.0, "root": 1.0, "prior": 0.5, "mcmi": 0.7}
CFG = O.OracleConfig(diag=True, method={"conditional": FEATS}, features=FEATS, n_keypts=18, window=64, z_dim=8, kernel=5,
                     channel=(8, 8, 16, 16, 32), arena_size=ARENA, discrete_classes={"ids": torch.arange(4)})


def load(golden_dir):
    fx = np.load(os.path.join(golden_dir, "mcmi_tiny.npz"))
    sd = {k[3:]: torch.from_numpy(fx[k]) for k in fx.files if k.startswith("sd/")}
    batches = [{k[4:]: torch.from_numpy(fx[k]) for k in fx.files if k.startswith(f"in{i}/")} for i in range(2)]
    return fx, sd, batches


@pytest.mark.parametrize("var_mode", ["sphere", "diagonal"])
def test_oracle_mcmi_matches_reference(golden_dir, var_mode):
    fx, _, _ = load(golden_dir)
    bw = float(fx["bandwidth"])
    assert float(fx[f"{var_mode}/s0/loss/mcmi"]) == 0.0  # no estimator before the first refresh (losses.py:224-225)
    for step in (1, 2):
        prev, cur = f"{var_mode}/s{step - 1}/", f"{var_mode}/s{step}/"
        val = O.mcmi_value(fx[prev + "x_s"], fx[prev + "var"], fx[prev + "var_s"], bw, fx[cur + "mu"], fx[cur + "var"])
        assert rel(val, fx[cur + "loss/mcmi"]) < 1e-5, step


def test_build_estimator_matches_oracle_on_cpu_tensors():
    """The build's chunked torch evaluation (the product code, also usable on CPU tensors) against the oracle expression,
    with more centres than one chunk and both var_modes."""
    from scrubvae_amd.model.disentangle import MutInfoEstimator
    g = torch.Generator().manual_seed(0)
    S, B, dx, dy = 300, 37, 8, 5
    x_s, y_s, x, y = torch.randn(S, dx, generator=g), torch.randn(S, dy, generator=g), torch.randn(B, dx, generator=g), torch.randn(B, dy, generator=g)
    L = torch.diag_embed(torch.rand(S, dx, generator=g) + 0.2)
    for mode in ("sphere", "diagonal"):
        est = MutInfoEstimator(x_s, y_s, 0.8, var_mode=mode, model_var=L)
        ref = O.mcmi_value(x_s, y_s, est.var_s, 0.8, x, y)
        assert abs(float(est(x, y)) - ref) < 1e-5 * abs(ref), mode


@pytest.mark.gpu
@pytest.mark.parametrize("var_mode", ["sphere", "diagonal"])
def test_hip_model_with_mcmi_matches_reference(golden_dir, var_mode):
    from scrubvae_amd.get import model as get_model
    from scrubvae_amd.train.losses import get_batch_loss
    from scrubvae_amd.train.trainer import clip_grad_norm_, make_mi_estimator
    fx, sd, batches = load(golden_dir)
    mc = dict(type="rcnn", kernel=CFG.kernel, z_dim=CFG.z_dim, window=CFG.window, activation="prelu", diag=True, init_dilation=None,
              prior="gaussian", channel=list(CFG.channel))
    dis = dict(method=CFG.method, alpha=1.0, features=FEATS, bandwidth=float(fx["bandwidth"]), var_mode=var_mode)
    config = {"disentangle": dis, "loss": LOSS}
    m = get_model(mc, None, None, dis, CFG.n_keypts, "midfwd", loss_config=LOSS, arena_size=ARENA, kinematic_tree=CFG.kinematic_tree,
                  discrete_classes={"ids": torch.arange(4)}, device="cuda", verbose=0)
    missing, unexpected = m.load_state_dict(sd, strict=False)
    assert not missing and not unexpected
    m.train()
    m.mi_estimator = None
    opt = torch.optim.AdamW(m.parameters(), lr=1e-4)
    for step in range(3):
        batch = {k: v.cuda() for k, v in batches[step % 2].items()}
        batch["eps"] = torch.from_numpy(fx[f"eps/{step}"]).cuda()
        data_o = m(batch)
        bl = get_batch_loss(m, batch, data_o, LOSS, dis)
        for p in m.parameters():
            p.grad = None
        bl["total"].backward()
        clip_grad_norm_(m, max_norm=1e6)
        pre = f"{var_mode}/s{step}/"
        g = m.grads_state_dict()
        for n in ("encoder.fc_mu.weight", "encoder.fc_mu.bias"):  # from step 1 on these carry the d mcmi / d mu seed
            assert rel(g[n].cpu(), torch.from_numpy(fx[pre + "grad/" + n])) < (2e-3 if step == 0 else 1e-2), (step, n)
        opt.step()
        upd = m.encode(batch)
        m.mi_estimator = make_mi_estimator(m, config, upd["mu"].detach().clone(), data_o["var"].clone(), upd["L"].detach().clone())
        tol = 1e-4 if step == 0 else 5e-3
        for k in fx.files:
            if k.startswith(pre + "loss/"):
                assert rel(bl[k.split("/")[-1]].detach().cpu(), fx[k]) < tol, (step, k)
        assert rel(m.mi_estimator.x_s.cpu(), fx[pre + "x_s"]) < (1e-4 if step == 0 else 5e-3), step
        assert rel(m.mi_estimator.var_s.cpu(), fx[pre + "var_s"]) < (1e-4 if step == 0 else 5e-3), step


@pytest.mark.gpu
def test_mcmi_seed_is_the_gradient_of_the_loss(golden_dir):
    """Without the mcmi term the fc_mu gradient differs: the term's seed really reaches the HIP backward, and scales
    linearly with loss_scale['mcmi']."""
    from scrubvae_amd.get import model as get_model
    from scrubvae_amd.model.disentangle import MutInfoEstimator
    from scrubvae_amd.train.losses import get_batch_loss
    fx, sd, batches = load(golden_dir)
    mc = dict(type="rcnn", kernel=CFG.kernel, z_dim=CFG.z_dim, window=CFG.window, activation="prelu", diag=True, init_dilation=None,
              prior="gaussian", channel=list(CFG.channel))
    dis = dict(method=CFG.method, alpha=1.0, features=FEATS, bandwidth=0.5, var_mode="sphere")
    m = get_model(mc, None, None, dis, CFG.n_keypts, "midfwd", arena_size=ARENA, kinematic_tree=CFG.kinematic_tree,
                  discrete_classes={"ids": torch.arange(4)}, device="cuda", verbose=0)
    m.load_state_dict(sd, strict=False)
    m.train()
    batch = {k: v.cuda() for k, v in batches[0].items()}
    batch["eps"] = torch.from_numpy(fx["eps/0"]).cuda()
    m.mi_estimator = MutInfoEstimator(torch.from_numpy(fx["sphere/s1/x_s"]).cuda(), torch.from_numpy(fx["sphere/s1/var"]).cuda(), 0.5)
    grads = {}
    for sc in (0.0, 1.0, 3.0):
        loss = dict(LOSS, mcmi=sc)
        bl = get_batch_loss(m, batch, m(batch), loss, dis)
        for p in m.parameters():
            p.grad = None
        bl["total"].backward()
        grads[sc] = m.grads_state_dict()["encoder.fc_mu.weight"].clone()
    d1, d3 = grads[1.0] - grads[0.0], grads[3.0] - grads[0.0]
    assert float(d1.abs().max()) > 1e-4
    assert rel(d3, 3.0 * d1) < 1e-3

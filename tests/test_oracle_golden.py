"""Pin the CPU oracle (oracle/scvae_oracle.py) against fixtures captured from the real
reference (tests/golden/make_fixtures.py).  CPU only.

Tolerances: forward outputs and every loss term of step 0 are fp32-roundoff tight
(<= 2e-5 max-norm relative; measured 1e-7..1e-6).  Gradients are compared in max-norm
relative to max(tensor scale, 1e-3 * global grad scale): the reference's own fp32 grads
carry ~1e-4..1e-2 noise on this path (1/(|a|+eps) normalisations in cont6d_to_matrix on
random tanh outputs, cancellation in PReLU-slope sums; 1-vs-8-thread reference runs differ
by the same amount), so the gate is 2e-2 there and the fp64 twin test below shows the
oracle itself is no noisier than the reference.
"""
import dataclasses
import os

import numpy as np
import pytest
import torch

from oracle import scvae_oracle as O

ARENA = torch.tensor([[-1.0, -1.0, -1.0], [1.0, 1.0, 1.0]])
TINY = dict(n_keypts=18, window=64, z_dim=8, kernel=5, channel=(8, 8, 16, 16, 32), arena_size=ARENA)
FULL_METHODS = {"conditional": ["avg_speed_3d", "heading"], "grad_reversal": ["avg_speed_3d", "heading"],
                "adversarial_net": ["heading"]}

SCENARIOS = {
    "vanilla_tiny": (O.OracleConfig(diag=True, **TINY), {"jpe": 1.0, "root": 1.0, "prior": 0.5}, "adamw"),
    "full_tiny": (O.OracleConfig(diag=True, method=FULL_METHODS, features=["avg_speed_3d", "heading"],
                                 discrete_classes={"ids": torch.arange(4)}, **TINY),
                  {"jpe": 1.0, "root": 1.0, "prior": 0.5,
                   "avg_speed_3d_gr": 1.0, "heading_gr": 2.0, "heading_an": 0.5}, "adamw"),
    "tc_tiny": (O.OracleConfig(diag=True, **TINY),
                {"jpe": 1.0, "root": 1.0, "prior": 0.5, "total_correlation": 0.1}, "adamw"),
    "rotation_tiny": (O.OracleConfig(diag=True, **TINY),
                      {"jpe": 1.0, "root": 1.0, "prior": 0.5, "rotation": 0.01}, "adamw"),
    "fullL_ids_tiny": (O.OracleConfig(diag=False, method={"conditional": ["ids"], "grad_reversal": ["ids"]},
                                      features=["ids"], discrete_classes={"ids": torch.arange(4)}, **TINY),
                       {"jpe": 1.0, "root": 1.0, "prior": 0.5, "ids_gr": 1.0}, "adam"),
    "vanilla_default_B4": (O.OracleConfig(n_keypts=18, window=64, z_dim=32, kernel=5, diag=True, arena_size=ARENA),
                           {"jpe": 1.0, "root": 1.0, "prior": 1.0}, "adamw"),
    # BASELINE's 23-joint synthetic skeleton run through the REAL reference (make_fixtures.py --j23-only): default widths and
    # the full SC-VAE head set at small widths
    "vanilla_default_j23_B4": (O.OracleConfig(n_keypts=23, window=64, z_dim=32, kernel=5, diag=True, arena_size=ARENA,
                                              kinematic_tree=O.skeleton_tree(23)),
                               {"jpe": 1.0, "root": 1.0, "prior": 1.0}, "adamw"),
    "full_j23_tiny": (O.OracleConfig(n_keypts=23, window=64, z_dim=8, kernel=5, channel=(8, 8, 16, 16, 32), diag=True, arena_size=ARENA,
                                     kinematic_tree=O.skeleton_tree(23), method=FULL_METHODS, features=["avg_speed_3d", "heading"],
                                     discrete_classes={"ids": torch.arange(4)}),
                      {"jpe": 1.0, "root": 1.0, "prior": 0.5, "avg_speed_3d_gr": 1.0, "heading_gr": 2.0, "heading_an": 0.5}, "adamw"),
    "linear_gr_tiny": (O.OracleConfig(diag=True, method={"linear": ["avg_speed_3d", "heading"], "grad_reversal": ["avg_speed_3d", "heading"]},
                                      features=["avg_speed_3d", "heading"], **TINY),
                       {"jpe": 1.0, "root": 1.0, "prior": 0.5, "avg_speed_3d_lin": 0.6, "heading_lin": 1.5,
                        "avg_speed_3d_gr": 1.0, "heading_gr": 2.0}, "adamw"),
    "tanh_tiny": (O.OracleConfig(diag=True, activation="tanh", **TINY), {"jpe": 1.0, "root": 1.0, "prior": 0.5}, "adamw"),
    # model.prior = "beta" (make_fixtures.py --beta-only: the Beta draw injected through torch._sample_dirichlet, the implicit
    # reparameterisation gradient is torch's own _dirichlet_grad on both sides)
    "beta_tiny": (O.OracleConfig(diag=True, prior="beta", **TINY), {"jpe": 1.0, "root": 1.0, "prior": 0.5}, "adamw"),
    "beta_full_tiny": (O.OracleConfig(diag=True, prior="beta", method=FULL_METHODS, features=["avg_speed_3d", "heading"],
                                      discrete_classes={"ids": torch.arange(4)}, **TINY),
                       {"jpe": 1.0, "root": 1.0, "prior": 0.5, "avg_speed_3d_gr": 1.0, "heading_gr": 2.0, "heading_an": 0.5}, "adamw"),
    # BASELINE config 5's shape: window 256; four blocks = the unmodified reference, six blocks = the
    # reference with only its default dilation list lengthened (it cannot build >4 blocks otherwise,
    # see make_fixtures.py)
    "w256_tiny": (O.OracleConfig(diag=True, n_keypts=18, window=256, z_dim=8, kernel=5, channel=(8, 8, 16, 16, 32),
                                 arena_size=ARENA), {"jpe": 1.0, "root": 1.0, "prior": 0.5}, "adamw"),
    "w256_6blocks_tiny": (O.OracleConfig(diag=True, n_keypts=18, window=256, z_dim=8, kernel=5,
                                         channel=(8, 8, 16, 16, 32, 32, 64), arena_size=ARENA),
                          {"jpe": 1.0, "root": 1.0, "prior": 0.5}, "adamw"),
}


def load_fixture(golden_dir, name):
    fx = np.load(os.path.join(golden_dir, name + ".npz"))
    cfg, loss_scale, opt = SCENARIOS[name]
    if "sd_checksum" in fx.files:
        sd = O.init_state_dict(cfg, seed=int(fx["sd_seed"]))
        chk = sum(float(v.double().abs().sum()) for v in sd.values())
        assert abs(chk - float(fx["sd_checksum"])) <= 1e-9 * abs(chk), "seeded weights not reproducible"
    else:
        sd = {k[3:]: torch.from_numpy(fx[k]) for k in fx.files if k.startswith("sd/")}
    data = {k[3:]: torch.from_numpy(fx[k]) for k in fx.files if k.startswith("in/")}
    return fx, cfg, loss_scale, opt, sd, data


def out_keys(cfg):
    """data_o entries every fixture stores for step 0 (VAE.dist_params, residual.py:299-302)"""
    return ("mu", "alpha", "beta", "z", "x6d", "root") if cfg.prior == "beta" else ("mu", "L", "z", "x6d", "root")


def rel(a, b):
    a, b = torch.as_tensor(a).double(), torch.as_tensor(b).double()
    return float((a - b).abs().max() / (b.abs().max() + 1e-30))


@pytest.mark.parametrize("name", list(SCENARIOS))
def test_oracle_step0_matches_reference(golden_dir, name):
    fx, cfg, loss_scale, opt, sd, data = load_fixture(golden_dir, name)
    eps, perm = torch.from_numpy(fx["eps/0"]), torch.from_numpy(fx["perm/0"])
    bl, grads, new_sd, out = O.train_step(
        sd, cfg, data, loss_scale, eps, adv_perm={k: perm for k in cfg.method.get("adversarial_net", [])},
        lr=1e-4, optimizer=opt)
    for k in out_keys(cfg):
        assert rel(out[k].detach(), fx["s0/out/" + k]) < 2e-5, k
    for k in fx.files:
        if k.startswith("s0/loss/"):
            assert rel(bl[k[8:]], fx[k]) < 2e-5, k
        if k.startswith("s0/out/disentangle/"):
            _, _, _, method, feat, i = k.split("/")
            got = out["disentangle"][method][feat]
            assert rel((got[i] if method == "linear" else got[int(i)]).detach(), fx[k]) < 2e-5, k
    gmax = float(fx["s0/grad_absmax"])
    tot = 0.0
    for k in fx.files:
        if k.startswith("s0/grad/"):
            r = torch.from_numpy(fx[k])
            d = float((grads[k[8:]] - r).abs().max()) / (float(r.abs().max()) + 1e-3 * gmax)
            # window 256: the PReLU-slope sums run over 4x more cancelling terms; against an fp64 twin
            # the reference is off by 5e-3 and this restatement by 3.5e-2 on the worst slope
            assert d < (6e-2 if cfg.window > 64 else 2e-2), (k, d)
        if k.startswith("s0/gradnorm/"):
            assert abs(float(grads[k[12:]].norm()) - float(fx[k])) <= 2e-2 * (float(fx[k]) + 1e-3 * gmax), k
    gn = torch.sqrt(sum((g.double() ** 2).sum() for g in grads.values()))
    # rotation loss: d asin(s)/ds ~ 2e3 near the clamp => ill-conditioned grads, looser gate
    assert rel(gn, fx["s0/grad_norm"]) < (1e-2 if "rotation" in loss_scale else 1e-3)


@pytest.mark.parametrize("name", ["vanilla_tiny", "full_tiny", "linear_gr_tiny", "full_j23_tiny", "beta_tiny"])
def test_oracle_multistep_and_eval(golden_dir, name):
    """3 optimizer steps then an eval-mode forward.  Adam amplifies fp32 noise (sign-like
    first step), the reference run at 1 vs 8 threads diverges ~1e-5/step: gate 2e-3."""
    fx, cfg, loss_scale, opt, sd, data = load_fixture(golden_dir, name)
    state = {}
    for s in range(3):
        eps, perm = torch.from_numpy(fx[f"eps/{s}"]), torch.from_numpy(fx[f"perm/{s}"])
        bl, grads, sd, out = O.train_step(
            sd, cfg, data, loss_scale, eps, adv_perm={k: perm for k in cfg.method.get("adversarial_net", [])},
            lr=1e-4, opt_state=state, optimizer=opt)
        for k in fx.files:
            if k.startswith(f"s{s}/loss/"):
                assert rel(bl[k[8:]], fx[k]) < 2e-3, k
    for k in fx.files:
        if k.startswith("final_sd/") and "running" in k:
            assert rel(sd[k[9:]], fx[k]) < 5e-3, k
    out = O.forward(sd, cfg, data, False, eps=torch.from_numpy(fx["eps/0"]) if cfg.prior == "beta" else None)  # (beta draws in eval mode too)
    bl = O.batch_loss(sd, cfg, data, out, loss_scale,
                      {k: torch.from_numpy(fx["perm/0"]) for k in cfg.method.get("adversarial_net", [])})
    for k in ("mu", "x6d", "root"):
        assert rel(out[k], fx["eval/out/" + k]) < 1e-2, k
    for k in fx.files:
        if k.startswith("eval/loss/"):
            assert rel(bl[k[10:]], fx[k]) < 5e-3, k


def test_fp64_twin_sets_noise_floor(golden_dir):
    """fp32 oracle vs fp64 oracle on the fixture batch: every loss term within 1e-5."""
    fx, cfg, loss_scale, opt, sd, data = load_fixture(golden_dir, "vanilla_tiny")
    eps = torch.from_numpy(fx["eps/0"])
    bl32, g32, _, _ = O.train_step(sd, cfg, data, loss_scale, eps)
    c64 = dataclasses.replace(cfg, arena_size=ARENA.double())
    sd64 = {k: (v.double() if v.dtype.is_floating_point else v) for k, v in sd.items()}
    d64 = {k: (v.double() if v.dtype.is_floating_point else v) for k, v in data.items()}
    bl64, g64, _, _ = O.train_step(sd64, c64, d64, loss_scale, eps.double())
    for k in bl32:
        assert rel(bl32[k], bl64[k]) < 1e-5, k


def test_shape_math_quirks():
    assert O.find_latent_dim(64, 5, 4) == 4 and O.find_out_dim(4, 5, 4) == 49
    assert O.final_kernel(O.OracleConfig()) == 22
    assert O.find_latent_dim(256, 5, 6) == 4 and O.find_out_dim(4, 5, 6) == 193
    assert O.final_kernel(O.OracleConfig(window=256, channel=(8, 8, 16, 16, 32, 32, 64))) == 70


EVAL_CFG = O.OracleConfig(diag=True, method={"conditional": ["avg_speed_3d", "heading"]}, features=["avg_speed_3d", "heading"],
                          discrete_classes={"ids": torch.arange(4)}, **TINY)


def load_eval_fixture(golden_dir):
    fx = np.load(os.path.join(golden_dir, "eval_full_tiny.npz"))
    sd = {k[3:]: torch.from_numpy(fx[k]) for k in fx.files if k.startswith("sd/")}
    data = {k[3:]: torch.from_numpy(fx[k]) for k in fx.files if k.startswith("in/")}
    return fx, sd, data


def test_oracle_eval_forward_matches_reference(golden_dir):
    """eval-mode encode + eval/eval.py generative_restrictiveness of the real reference
    (tests/golden/make_fixtures.py --eval-only) vs the oracle restatement, with the re-draw injected."""
    fx, sd, data = load_eval_fixture(golden_dir)
    enc = O.encode(sd, EVAL_CFG, data, False)
    assert rel(enc["mu"], fx["enc/mu"]) < 2e-5
    assert rel(enc["L"], fx["enc/L"]) < 2e-5
    for key in ("heading", "avg_speed_3d"):
        pred, target, _ = O.generative_restrictiveness(sd, EVAL_CFG, torch.from_numpy(fx["enc/mu"]), data, key,
                                                       torch.from_numpy(fx[f"gen/{key}/draw"]))
        assert rel(target, fx[f"gen/{key}/target"]) < 1e-6, key
        assert rel(pred, fx[f"gen/{key}/pred"]) < 2e-5, key

"""End-to-end parity of the HIP SC-VAE step on a real MI355X, through the reference's own
API surface (get.model -> model(data) -> get_batch_loss -> total.backward() -> optimizer):

  * against the golden fixtures captured from the real reference (tests/golden/*.npz);
  * against the CPU oracle on the same seeded inputs (fp64 twin as the truth).

Tolerances (fp32, SURVEY 8c noise floor): forward outputs 2e-5 max-norm relative, every loss
term 1e-4 relative (north_star: "ELBO within 1e-4 relative"), gradients 2e-2 in the
scale-aware max-norm of test_oracle_golden (the reference's own fp32 grads carry that noise),
and additionally the HIP gradients are gated against the fp64 truth relative to the fp32 CPU
oracle's own error (test_grads_vs_fp64_truth).
"""
import dataclasses
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import scvae_oracle as O
from tests.test_oracle_golden import SCENARIOS, load_fixture, rel, ARENA, out_keys


def build_model(cfg, sd):
    from scrubvae_amd.get import model as get_model
    model_config = dict(type="rcnn", kernel=cfg.kernel, z_dim=cfg.z_dim, window=cfg.window, activation=cfg.activation,
                        diag=cfg.diag, init_dilation=None, prior=cfg.prior, channel=list(cfg.channel))
    dis = dict(method=cfg.method, alpha=cfg.alpha, features=cfg.features or [])
    m = get_model(model_config, None, None, dis, cfg.n_keypts, "midfwd", arena_size=cfg.arena_size,
                  kinematic_tree=cfg.kinematic_tree, discrete_classes=cfg.discrete_classes, device="cuda", verbose=0)
    missing, unexpected = m.load_state_dict(sd, strict=False)
    assert not missing and not unexpected, (missing, unexpected)
    return m, dis


def to_dev(data):
    return {k: v.cuda() for k, v in data.items()}


@pytest.mark.parametrize("name", ["vanilla_tiny", "full_tiny", "rotation_tiny", "fullL_ids_tiny", "tc_tiny", "vanilla_default_B4",
                                  "w256_tiny", "w256_6blocks_tiny", "linear_gr_tiny", "tanh_tiny", "vanilla_default_j23_B4",
                                  "full_j23_tiny", "beta_tiny", "beta_full_tiny"])
def test_step0_matches_reference_fixture(golden_dir, name):
    from scrubvae_amd.train.losses import get_batch_loss
    fx, cfg, loss_scale, opt, sd, data = load_fixture(golden_dir, name)
    model, dis = build_model(cfg, sd)
    # state_dict round trip is exact
    back = model.state_dict()
    for k, v in sd.items():
        assert torch.equal(back[k].cpu(), v), k
    model.train()
    d = to_dev(data)
    d["eps"] = torch.from_numpy(fx["eps/0"]).cuda()
    data_o = model(d)
    perm = torch.from_numpy(fx["perm/0"])
    bl = get_batch_loss(model, d, data_o, loss_scale, dis, adv_perm={k: perm for k in cfg.method.get("adversarial_net", [])})
    for k in out_keys(cfg):  # (prior="beta": alpha, beta in place of L; data["eps"] is then the injected Beta draw)
        assert rel(data_o[k].cpu(), fx["s0/out/" + k]) < 2e-5, k
    for k in fx.files:
        if k.startswith("s0/out/disentangle/"):
            _, _, _, method, feat, i = k.split("/")
            got = data_o["disentangle"][method][feat]
            assert rel((got[i] if method == "linear" else got[int(i)]).detach().cpu(), fx[k]) < 5e-5, k
        if k.startswith("s0/loss/"):
            assert rel(bl[k[8:]].detach().cpu(), fx[k]) < 1e-4, k
    for p in model.parameters():
        p.grad = None
    bl["total"].backward()
    torch.cuda.synchronize()
    grads = {k: v.cpu() for k, v in model.grads_state_dict().items()}
    gmax = float(fx["s0/grad_absmax"])
    for k in fx.files:
        if k.startswith("s0/grad/"):
            r = torch.from_numpy(fx[k])
            dd = float((grads[k[8:]] - r).abs().max()) / (float(r.abs().max()) + 1e-3 * gmax)
            # PReLU-slope / BN grads are cancellation-prone sums: the reference's own fp32 value
            # is off by up to ~4e-2 from the fp64 truth there (see test_grads_vs_fp64_truth for
            # the tight gate); the single-element slopes of the window-256 fixtures sum 4x more cancelling terms
            # (test_oracle_golden holds the oracle itself to 6e-2 there; the fp16-piece forward measured 8.7e-2 on one of them)
            assert dd < (1e-1 if (r.numel() == 1 and cfg.window > 64) else 5e-2), (k, dd)
        if k.startswith("s0/gradnorm/"):
            # Single PReLU slopes get 1e-1: the loss gradient is DISCONTINUOUS across a PReLU kink, and with B = 4 one element whose
            # pre-activation lies within the forward tolerance of zero (|u| ~ 1e-6: ~0.1 expected per step over the ~1e5 activations of
            # this model) can sit on the other side than in the reference's fp32 run.  Measured (tools/slope_noise.py on this fixture,
            # f16x3b3): 1 of 12,544 elements of the last decoder stage flipped, which moved every upstream gradient tensor by <= 1 %
            # in max-norm and the smallest slope gradient (|g| = 0.5 of a 25 gradient scale) by 6.8 %; with the same kernels and the
            # BatchNorm statistics rounded differently (10 variants) all seventeen slopes sit 1e-5..5e-4 from the fp64 oracle.
            tol = 1e-1 if grads[k[12:]].numel() == 1 else 2e-2
            assert abs(float(grads[k[12:]].norm()) - float(fx[k])) <= tol * (float(fx[k]) + 1e-3 * gmax), k
    gn = torch.sqrt(sum((g.double() ** 2).sum() for g in grads.values()))
    assert rel(gn, fx["s0/grad_norm"]) < (1e-2 if "rotation" in loss_scale else 1e-3)


@pytest.mark.parametrize("name", ["vanilla_tiny", "full_tiny", "linear_gr_tiny", "beta_tiny"])
def test_grads_vs_fp64_truth(golden_dir, name):
    """HIP fp32 gradients against the fp64 oracle.  Forward outputs/losses are as accurate as the
    fp32 CPU path (measured 1.3x / 1.0x its error); gradients are cancellation-prone sums and
    the MFMA kernels accumulate each output over K in one fp32 chain, where oneDNN uses
    blocked partial sums: measured 5-9x the CPU fp32 path's error (2.5e-4 typical, 3.5e-3 on
    PReLU slopes, in max-norm relative to the fp64 truth).  Gate: 12x the CPU noise + 1e-3."""
    from scrubvae_amd.train.losses import get_batch_loss
    fx, cfg, loss_scale, opt, sd, data = load_fixture(golden_dir, name)
    eps, perm = torch.from_numpy(fx["eps/0"]), torch.from_numpy(fx["perm/0"])
    ap = {k: perm for k in cfg.method.get("adversarial_net", [])}
    bl32, g32, _, _ = O.train_step(sd, cfg, data, loss_scale, eps, adv_perm=ap)
    c64 = dataclasses.replace(cfg, arena_size=ARENA.double())
    sd64 = {k: (v.double() if v.dtype.is_floating_point else v) for k, v in sd.items()}
    d64 = {k: (v.double() if v.dtype.is_floating_point else v) for k, v in data.items()}
    bl64, g64, _, _ = O.train_step(sd64, c64, d64, loss_scale, eps.double(), adv_perm=ap)
    model, dis = build_model(cfg, sd)
    model.train()
    d = to_dev(data)
    d["eps"] = eps.cuda()
    data_o = model(d)
    bl = get_batch_loss(model, d, data_o, loss_scale, dis, adv_perm=ap)
    bl["total"].backward()
    grads = {k: v.cpu() for k, v in model.grads_state_dict().items()}
    for k in bl64:
        assert rel(bl[k].detach().cpu(), bl64[k]) < 1e-5 + 4 * rel(bl32[k], bl64[k]), k
    gmax = max(float(g.abs().max()) for g in g64.values())
    worst_hip = worst_cpu = 0.0
    for n, g in g64.items():
        den = float(g.abs().max()) + 1e-3 * gmax
        worst_hip = max(worst_hip, float((grads[n].double() - g).abs().max()) / den)
        worst_cpu = max(worst_cpu, float((g32[n].double() - g).abs().max()) / den)
    assert worst_hip < 12 * worst_cpu + 1e-3, (worst_hip, worst_cpu)


@pytest.mark.parametrize("name", ["vanilla_tiny", "full_tiny", "linear_gr_tiny", "full_j23_tiny", "beta_tiny"])
def test_three_steps_and_eval(golden_dir, name):
    """trainer-style loop (fused AdamW) for 3 steps, then eval-mode forward; same gates as the
    oracle's own multi-step test (Adam amplifies fp32 noise)."""
    from scrubvae_amd.train.losses import get_batch_loss
    from scrubvae_amd.train.trainer import FusedAdam, clip_grad_norm_
    fx, cfg, loss_scale, opt, sd, data = load_fixture(golden_dir, name)
    model, dis = build_model(cfg, sd)
    optim = FusedAdam(model, lr=1e-4, weight_decay=0.01, decoupled=True)
    model.train()
    d = to_dev(data)
    for s in range(3):
        d["eps"] = torch.from_numpy(fx[f"eps/{s}"]).cuda()
        perm = torch.from_numpy(fx[f"perm/{s}"])
        data_o = model(d)
        bl = get_batch_loss(model, d, data_o, loss_scale, dis, adv_perm={k: perm for k in cfg.method.get("adversarial_net", [])})
        for p in model.parameters():
            p.grad = None
        bl["total"].backward()
        gn = clip_grad_norm_(model, 1e6)
        if s == 0:
            assert rel(gn.cpu(), fx["s0/grad_norm"]) < 1e-3
        optim.step()
        for k in fx.files:
            if k.startswith(f"s{s}/loss/"):
                assert rel(bl[k[8:]].detach().cpu(), fx[k]) < 2e-3, k
    new_sd = model.state_dict()
    for k in fx.files:
        if k.startswith("final_sd/") and "running" in k:
            assert rel(new_sd[k[9:]].cpu(), fx[k]) < 5e-3, k
        if k.startswith("final_sd/") and k.endswith("num_batches_tracked"):
            assert int(new_sd[k[9:]]) == int(fx[k])
        if k.startswith("final_sd/disentangle.adversarial_net."):
            # frozen discriminator (requires_grad=False, disentangle.py:670-671): torch's AdamW skips it, so no weight decay
            # either -- bit-unchanged after the three steps, in the reference and here
            assert torch.equal(new_sd[k[9:]].cpu(), torch.from_numpy(fx[k])) and torch.equal(sd[k[9:]], torch.from_numpy(fx[k])), k
    model.eval()
    d["eps"] = torch.from_numpy(fx["eps/0"]).cuda()  # (only prior="beta" draws in eval mode: the fixture's eval pass used draw 0)
    with torch.no_grad():
        data_o = model(d)
        bl = get_batch_loss(model, d, data_o, loss_scale, dis,
                            adv_perm={k: torch.from_numpy(fx["perm/0"]) for k in cfg.method.get("adversarial_net", [])})
    for k in ("mu", "x6d", "root"):
        assert rel(data_o[k].cpu(), fx["eval/out/" + k]) < 1e-2, k
    for k in fx.files:
        if k.startswith("eval/loss/"):
            assert rel(bl[k[10:]].cpu(), fx[k]) < 5e-3, k


def test_oracle_parity_seeded_j23():
    """BASELINE's synthetic 23-joint skeleton, default channels, B=16: HIP vs CPU oracle on the
    same seeded inputs (the reference's own outputs on this skeleton are the `vanilla_default_j23_B4` / `full_j23_tiny` fixtures,
    generated by passing the 23-joint tree through the real reference: test_step0_matches_reference_fixture)."""
    from scrubvae_amd.train.losses import get_batch_loss
    cfg = O.OracleConfig(n_keypts=23, window=64, z_dim=32, kernel=5, diag=True, arena_size=ARENA,
                         kinematic_tree=O.skeleton_tree(23))
    sd = O.init_state_dict(cfg, seed=11)
    data = O.synth_batch(cfg, 16, seed=3)
    eps = torch.randn(16, 32, generator=torch.Generator().manual_seed(5))
    ls = {"jpe": 1.0, "root": 1.0, "prior": 1.0}
    bl_o, g_o, _, out_o = O.train_step(sd, cfg, data, ls, eps)
    model, dis = build_model(cfg, sd)
    model.train()
    d = to_dev(data)
    d["eps"] = eps.cuda()
    data_o = model(d)
    bl = get_batch_loss(model, d, data_o, ls, dis)
    bl["total"].backward()
    for k in ("mu", "z", "x6d", "root"):
        assert rel(data_o[k].cpu(), out_o[k].detach()) < 2e-5, k
    for k in bl_o:
        assert rel(bl[k].detach().cpu(), bl_o[k]) < 1e-4, k
    grads = {k: v.cpu() for k, v in model.grads_state_dict().items()}
    gmax = max(float(g.abs().max()) for g in g_o.values())
    for n, g in g_o.items():
        dd = float((grads[n] - g).abs().max()) / (float(g.abs().max()) + 1e-3 * gmax)
        assert dd < 2e-2, (n, dd)


@pytest.mark.parametrize("precision", ["f32", "bf16x6w3", "f16x3b3", "bf16x6b3"])
def test_oracle_parity_config5_wide_w256(precision):
    """BASELINE configs[4] at its full widths (window 256, six residual blocks, channels 64..4096, 23 joints; 293 M
    parameters), B=4 so that the CPU oracle finishes in seconds: HIP vs oracle on seeded inputs.  The reference cannot
    construct six blocks (see tests/golden/make_fixtures.py); its arithmetic at this depth is pinned by the
    `w256_6blocks_tiny` fixture above."""
    from scrubvae_amd import ops
    from scrubvae_amd.train.losses import get_batch_loss
    cfg = O.OracleConfig(n_keypts=23, window=256, z_dim=32, kernel=5, diag=True, arena_size=ARENA,
                         channel=(64, 128, 256, 512, 1024, 2048, 4096), kinematic_tree=O.skeleton_tree(23))
    sd = O.init_state_dict(cfg, seed=13)
    data = O.synth_batch(cfg, 4, seed=4)
    eps = torch.randn(4, 32, generator=torch.Generator().manual_seed(6))
    ls = {"jpe": 1.0, "root": 1.0, "prior": 1.0}
    bl_o, g_o, _, out_o = O.train_step(sd, cfg, data, ls, eps)
    prev = ops.PRECISION
    ops.set_precision(precision)
    try:
        model, dis = build_model(cfg, sd)
        model.train()
        d = to_dev(data)
        d["eps"] = eps.cuda()
        data_o = model(d)
        bl = get_batch_loss(model, d, data_o, ls, dis)
        bl["total"].backward()
        torch.cuda.synchronize()
    finally:
        ops.set_precision(prev)
    for k in ("mu", "z", "x6d", "root"):
        assert rel(data_o[k].cpu(), out_o[k].detach()) < 5e-5, k
    for k in bl_o:
        assert rel(bl[k].detach().cpu(), bl_o[k]) < 1e-4, k
    grads = {k: v.cpu() for k, v in model.grads_state_dict().items()}
    gmax = max(float(g.abs().max()) for g in g_o.values())
    for n, g in g_o.items():
        dd = float((grads[n] - g).abs().max()) / (float(g.abs().max()) + 1e-3 * gmax)
        assert dd < 5e-2, (n, dd)


def test_reference_style_loop_with_torch_optimizer():
    """The reference's loop verbatim (param.grad=None; backward; clip; torch.optim.AdamW.step)
    runs on the HIP model: torch optimizers see ordinary Parameters with .grad."""
    from scrubvae_amd.train.trainer import train_test_epoch
    cfg = O.OracleConfig(n_keypts=18, window=64, z_dim=8, kernel=5, channel=(16, 16, 16, 32, 32), diag=True, arena_size=ARENA)
    sd = O.init_state_dict(cfg, seed=1)
    model, dis = build_model(cfg, sd)
    data = O.synth_batch(cfg, 8, seed=1)
    loader = [data, data]
    config = {"loss": {"jpe": 1.0, "root": 1.0, "prior": 0.1}, "disentangle": dis}
    opt = torch.optim.AdamW(model.parameters(), lr=1e-3)
    m0 = train_test_epoch(config, model, loader, "cuda", 1, opt, None, "train")
    for _ in range(5):
        m1 = train_test_epoch(config, model, loader, "cuda", 2, opt, None, "train")
    assert m1["total"] < m0["total"]  # it learns
    mt = train_test_epoch(config, model, loader, "cuda", 2, None, None, "test")
    assert np.isfinite(mt["total"])


def test_oracle_parity_full_cholesky_total_correlation():
    """model.diag=False + beta-TCVAE total correlation (losses.py:41-101) vs the CPU oracle."""
    from scrubvae_amd.train.losses import get_batch_loss
    cfg = O.OracleConfig(n_keypts=18, window=64, z_dim=8, kernel=5, channel=(8, 8, 16, 16, 32), diag=False, arena_size=ARENA)
    sd = O.init_state_dict(cfg, seed=21)
    data = O.synth_batch(cfg, 12, seed=21)
    eps = torch.randn(12, 8, generator=torch.Generator().manual_seed(2))
    ls = {"jpe": 1.0, "root": 1.0, "prior": 0.5, "total_correlation": 0.7}
    bl_o, g_o, _, out_o = O.train_step(sd, cfg, data, ls, eps)
    model, dis = build_model(cfg, sd)
    model.train()
    d = to_dev(data)
    d["eps"] = eps.cuda()
    data_o = model(d)
    bl = get_batch_loss(model, d, data_o, ls, dis)
    bl["total"].backward()
    for k in ("mu", "L", "z", "x6d", "root"):
        assert rel(data_o[k].cpu(), out_o[k].detach()) < 2e-5, k
    for k in bl_o:
        assert rel(bl[k].detach().cpu(), bl_o[k]) < 1e-4, k
    grads = {k: v.cpu() for k, v in model.grads_state_dict().items()}
    gmax = max(float(g.abs().max()) for g in g_o.values())
    for n, g in g_o.items():
        dd = float((grads[n] - g).abs().max()) / (float(g.abs().max()) + 1e-3 * gmax)
        assert dd < 2e-2, (n, dd)


def test_graphed_step_matches_eager():
    """The hipGraph replay of a whole optimizer step reproduces the same schedule launched eagerly
    bit for bit, including the step-dependent Adam scalars fed from device memory, and tracks the
    reference-style loop (autograd entry + host-scalar Adam) to fp32 noise."""
    from scrubvae_amd.train.losses import get_batch_loss
    from scrubvae_amd.train.trainer import FusedAdam, GraphedStep, clip_grad_norm_
    cfg = O.OracleConfig(n_keypts=18, window=64, z_dim=8, kernel=5, channel=(16, 16, 16, 32, 32), diag=True, arena_size=ARENA)
    sd = O.init_state_dict(cfg, seed=5)
    data = to_dev(O.synth_batch(cfg, 8, seed=5))
    data["eps"] = torch.randn(8, 8, generator=torch.Generator().manual_seed(3)).cuda()
    ls = {"jpe": 1.0, "root": 1.0, "prior": 0.3}
    runs = {}
    for capture in (False, True):
        m, dis = build_model(cfg, sd)
        o = FusedAdam(m, lr=1e-3, weight_decay=0.01, decoupled=True)
        step = GraphedStep(m, o, ls, dis, data, warmup=3, capture=capture)
        losses = [float(step()["total"]) for _ in range(3)]
        assert o.step_count == 6
        runs[capture] = (losses, m.flat_params.clone())
        # replays queued WITHOUT a host sync in between (the host runs far ahead of the GPU), with a learning-rate change
        # on the way: the step counter and bias corrections live on the device, the lr write is stream-ordered
        for i in range(12):
            if i == 6:
                o.param_groups[0]["lr"] = 3e-4
            step()
        assert o.step_count == 18
        torch.cuda.synchronize()
        runs[capture] += (m.flat_params.clone(), float(o.hyper[3]))
    assert runs[True][0] == runs[False][0]
    assert torch.equal(runs[True][1], runs[False][1])
    assert torch.equal(runs[True][2], runs[False][2]) and runs[True][3] == runs[False][3] == 18.0
    # reference-style loop
    m1, dis = build_model(cfg, sd)
    o1 = FusedAdam(m1, lr=1e-3, weight_decay=0.01, decoupled=True)
    m1.train()
    for _ in range(6):
        bl = get_batch_loss(m1, data, m1(data), ls, dis)
        bl["total"].backward()
        clip_grad_norm_(m1, 1e6)
        o1.step()
    # not bitwise: the captured step feeds Adam's bias-correction scalars from device floats, the loop computes them on
    # the host in double; six steps at lr 1e-3 amplify that last-bit difference to ~1e-4 relative in the loss
    assert abs(float(bl["total"].detach()) - runs[True][0][-1]) <= 1e-3 * abs(runs[True][0][-1])


def test_clip_grad_norm_scales_like_torch():
    """clip_grad_norm_ (trainer.py:164) with a max_norm that bites: gradients scaled by max_norm / (norm + 1e-6) as torch does;
    with the reference's 1e6 they are left bit-unchanged."""
    from scrubvae_amd.train.losses import get_batch_loss
    from scrubvae_amd.train.trainer import clip_grad_norm_
    cfg = O.OracleConfig(n_keypts=18, window=64, z_dim=8, kernel=5, channel=(16, 16, 16, 32, 32), diag=True, arena_size=ARENA)
    sd = O.init_state_dict(cfg, seed=5)
    data = to_dev(O.synth_batch(cfg, 8, seed=5))
    m, dis = build_model(cfg, sd)
    m.train()
    bl = get_batch_loss(m, data, m(data), {"jpe": 1.0, "root": 1.0, "prior": 0.3}, dis)
    bl["total"].backward()
    g0 = m.flat_grads.clone()
    n1 = clip_grad_norm_(m, 1e6)
    assert torch.equal(m.flat_grads, g0)
    p = torch.nn.Parameter(torch.zeros_like(g0))
    p.grad = g0.clone()
    n_t = torch.nn.utils.clip_grad_norm_([p], 0.5 * float(n1))
    n2 = clip_grad_norm_(m, 0.5 * float(n1))
    assert abs(float(n2) - float(n_t)) <= 1e-5 * float(n_t)
    assert rel(m.flat_grads.cpu(), p.grad.cpu()) < 1e-6


# ------------------------------------------------------------------ eval forward
@pytest.mark.gpu
def test_eval_forward_matches_reference_fixture(golden_dir, monkeypatch):
    """ResVAE.encode (eval) and eval.generative_restrictiveness on the HIP path vs the real reference's outputs
    (tests/golden/eval_full_tiny.npz), the random re-draw injected through torch.rand / torch.randn."""
    from tests.test_oracle_golden import EVAL_CFG, load_eval_fixture
    from scrubvae_amd.eval import generative_restrictiveness
    fx, sd, data = load_eval_fixture(golden_dir)
    m, _ = build_model(EVAL_CFG, sd)
    m.eval()
    with torch.no_grad():
        enc = m.encode({k: data[k].cuda() for k in ("x6d", "root")})
        assert rel(enc["mu"].cpu(), fx["enc/mu"]) < 2e-5
        assert rel(enc["L"].cpu(), fx["enc/L"]) < 2e-5
        for key in ("heading", "avg_speed_3d"):
            draw = torch.from_numpy(fx[f"gen/{key}/draw"]).cuda()
            monkeypatch.setattr(torch, "rand", lambda *a, **k: draw.clone())
            monkeypatch.setattr(torch, "randn", lambda *a, **k: draw.clone())
            d = to_dev(data)
            pred, target = generative_restrictiveness(m, enc["mu"], d, key, EVAL_CFG.kinematic_tree)
            monkeypatch.undo()
            assert d[key] is target                                    # the reference mutates data[key]
            assert rel(target.cpu(), fx[f"gen/{key}/target"]) < 1e-6, key
            assert rel(pred.cpu(), fx[f"gen/{key}/pred"]) < 5e-5, key


@pytest.mark.gpu
def test_get_latents_embeds_caches_and_reloads(golden_dir, tmp_path):
    """get.latents (get/eval.py:8-70): eval-mode encode over a loader == the real reference's `encode` means
    (eval_full_tiny), written to <out_path>/latents/<split>_<epoch>.npy and re-read from there on the next call."""
    from tests.test_oracle_golden import EVAL_CFG, load_eval_fixture
    from scrubvae_amd.get import latents
    fx, sd, data = load_eval_fixture(golden_dir)
    m, _ = build_model(EVAL_CFG, sd)
    m.train()

    class DS(torch.utils.data.Dataset):
        def __len__(self):
            return data["x6d"].shape[0]

        def __getitem__(self, i):
            return {k: v[i] for k, v in data.items()}

    loader = torch.utils.data.DataLoader(DS(), batch_size=3, shuffle=False)  # ragged last batch
    config = {"out_path": str(tmp_path)}
    z = latents(config, m, epoch=7, loader=loader, device="cuda", train_val_test="val")
    assert not m.training and z.device.type == "cpu"
    assert rel(z, fx["enc/mu"]) < 2e-5
    path = tmp_path / "latents" / "val_7.npy"
    assert path.exists()
    np.save(path, np.load(path) + 1.0)  # the second call must come from the file
    z2 = latents(config, m, epoch=7, loader=loader, device="cuda", train_val_test="val")
    assert torch.equal(z2, z + 1.0)
    z3 = latents(config, m, epoch=7, loader=loader, device="cuda", train_val_test="val", overwrite=True)
    assert torch.equal(z3, z)


@pytest.mark.gpu
def test_test_epoch_runs_and_matches_oracle_losses(golden_dir):
    """trainer.test_epoch: eval-mode losses averaged over the loader == the oracle's eval losses; returns mu on the
    CPU and an r2_gen_restrict_<key> metric per conditional feature."""
    from scrubvae_amd.train.trainer import test_epoch as run_test_epoch
    fx, cfg, loss_scale, opt, sd, data = load_fixture(golden_dir, "full_tiny")
    m, dis = build_model(cfg, sd)

    class DS(torch.utils.data.Dataset):
        kinematic_tree = cfg.kinematic_tree

        def __len__(self):
            return data["x6d"].shape[0]

        def __getitem__(self, i):
            return {k: v[i] for k, v in data.items()}

    loader = torch.utils.data.DataLoader(DS(), batch_size=4, shuffle=False)
    config = {"loss": dict(loss_scale), "disentangle": dis}
    torch.manual_seed(0)
    metrics, z = run_test_epoch(config, m, loader, device="cuda", epoch=0)
    assert z.shape == (8, cfg.z_dim) and z.device.type == "cpu"
    # the adversarial term shuffles with torch.randperm (not injectable through test_epoch): compare the others
    want = {k: 0.0 for k in config["loss"] if not k.endswith("_an")}
    for lo in (0, 4):
        part = {k: v[lo:lo + 4] for k, v in data.items()}
        out = O.forward(sd, cfg, part, False)
        bl = O.batch_loss(sd, cfg, part, out, config["loss"], {k: torch.arange(4) for k in cfg.method.get("adversarial_net", [])})
        for k in want:
            want[k] += float(bl[k]) / 2
    assert set(metrics) >= {"total"} | set(config["loss"])
    for k, v in want.items():
        assert abs(metrics[k] - v) <= 1e-4 * abs(v) + 1e-6, (k, metrics[k], v)
    assert rel(z, O.encode(sd, cfg, data, False)["mu"]) < 5e-5
    for key in ("avg_speed_3d", "heading"):
        assert np.isfinite(metrics[f"r2_gen_restrict_{key}"])


# ------------------------------------------------------------------ split-bf16 precision on the whole model
@pytest.fixture(params=["bf16x6", "bf16x6w3", "bf16x6b3", "f16x3b3"])
def bf16x6_everywhere(request):
    """Every conv / linear of the model on the split-bf16 kernels, whatever its size: 3 pieces / 6 products
    ("bf16x6"), and the same with 2 pieces / 3 products for the weight-gradient contractions ("bf16x6w3",
    bench.py's default)."""
    from scrubvae_amd import ops
    keep = (ops.PRECISION, ops.SPLIT_MIN_FLOPS)
    ops.set_precision(request.param)
    ops.SPLIT_MIN_FLOPS = 0.0
    yield
    ops.set_precision(keep[0])
    ops.SPLIT_MIN_FLOPS = keep[1]


@pytest.mark.parametrize("name", ["vanilla_tiny", "full_tiny", "fullL_ids_tiny", "vanilla_default_B4", "w256_6blocks_tiny", "tanh_tiny",
                                  "vanilla_default_j23_B4", "full_j23_tiny", "beta_full_tiny"])
def test_step0_matches_reference_fixture_bf16x6(golden_dir, name, bf16x6_everywhere):
    """The reference fixtures at the SAME fp32 tolerances with the contractions on the bf16 matrix cores
    (bench.py's default precision): the 3-piece split is fp32-accurate, not a reduced-precision mode."""
    test_step0_matches_reference_fixture(golden_dir, name)


@pytest.mark.parametrize("name", ["vanilla_tiny", "full_tiny"])
def test_three_steps_and_grads_bf16x6(golden_dir, name, bf16x6_everywhere):
    """Gradients vs the fp64 truth and three optimizer steps + eval forward, same gates as the fp32 kernels."""
    test_grads_vs_fp64_truth(golden_dir, name)
    test_three_steps_and_eval(golden_dir, name)


@pytest.fixture(params=["f16x3b3", "bf16x6b3"])
def fused_upsample_everywhere(request, monkeypatch):
    """Every layer on the split kernels AND every decoder skip conv on the fused-upsample forward (halo kernels blending the
    half-length input, the upsampled tensor left behind for the weight gradient), whatever the tile table says about the geometry."""
    from scrubvae_amd import ops
    from scrubvae_amd.model import residual
    keep = (ops.PRECISION, ops.SPLIT_MIN_FLOPS)
    ops.set_precision(request.param)
    ops.SPLIT_MIN_FLOPS = 0.0
    monkeypatch.setattr(residual, "FUSE_UPSAMPLE", "force")
    yield
    ops.set_precision(keep[0])
    ops.SPLIT_MIN_FLOPS = keep[1]


@pytest.mark.parametrize("name", ["vanilla_tiny", "full_tiny", "vanilla_default_j23_B4", "w256_6blocks_tiny"])
def test_reference_fixtures_with_the_upsample_fused_into_the_skip_convs(golden_dir, name, fused_upsample_everywhere):
    """Step 0 of the reference fixtures (outputs, every loss term, gradient norms: reference residual.py:153-170 and its autograd
    through the fused forward + the by-product `up` tensor of the weight gradient), the gradients against the fp64 truth and three
    optimizer steps, at the unfused path's tolerances."""
    from scrubvae_amd.model import residual
    test_step0_matches_reference_fixture(golden_dir, name)
    if name in ("vanilla_tiny", "full_tiny"):
        test_grads_vs_fp64_truth(golden_dir, name)
        test_three_steps_and_eval(golden_dir, name)
    # the fused path really ran: a model built now holds up2 convs for its decoder skip path
    fx, cfg, loss_scale, opt, sd, data = load_fixture(golden_dir, name)
    model, _ = build_model(cfg, sd)
    model.train()
    model(to_dev(data))
    sk = [cv for key, cv in model._convs.items() if key[0].endswith(".sk")]
    assert sk and all(cv.up2 for cv in sk if cv.kernel == 6), [(k, cv.up2) for k, cv in model._convs.items() if k[0].endswith(".sk")]


# ------------------------------------------------------------------ full benchmark size: size-independent properties
def _bench_size_model(precision, seed=0):
    from scrubvae_amd import ops
    from scrubvae_amd.get import model as get_model
    keep = ops.PRECISION
    ops.set_precision(precision)
    try:
        cfg = O.OracleConfig(n_keypts=23, window=64, z_dim=32, kernel=5, diag=True, arena_size=ARENA, kinematic_tree=O.skeleton_tree(23))
        sd = O.init_state_dict(cfg, seed=seed)
        m, dis = build_model(cfg, sd)
    finally:
        ops.set_precision(keep)
    return m, dis


def test_full_size_properties_b1024():
    """BASELINE configs[1] size (B=1024, W=64, J=23, default channels), where the CPU oracle is too slow to be the
    checker: (a) the step is deterministic (two evaluations are bitwise equal), (b) eval-mode outputs are per-sample
    (a batch permutation permutes them), (c) the three precisions of the HIP path agree on every loss term far
    inside the 1e-4 ELBO bound, (d) the first half of the batch evaluated alone gives the same outputs."""
    from scrubvae_amd.data import synthetic
    from scrubvae_amd.train.losses import get_batch_loss
    data, tree = synthetic.make_batch(23, 64, 1024, seed=3, device="cuda")
    data = {k: v for k, v in data.items() if k in ("x6d", "root", "offsets", "target_pose")}
    data["eps"] = torch.randn(1024, 32, generator=torch.Generator().manual_seed(1)).cuda()
    ls = {"jpe": 1.0, "root": 1.0, "prior": 1.0}
    losses = {}
    for precision in ("f32", "bf16x6", "bf16x6w3", "bf16x6b3", "f16x3b3"):
        m, dis = _bench_size_model(precision)
        m.train()
        with torch.no_grad():
            a = get_batch_loss(m, data, m(data), ls, dis)
            b = get_batch_loss(m, data, m(data), ls, dis)
        losses[precision] = {k: float(v) for k, v in a.items()}
        assert all(float(a[k]) == float(b[k]) for k in a), precision                      # (a)
        if precision == "bf16x6w3":
            m.eval()
            with torch.no_grad():
                o = m(data)
                x6d, mu = o["x6d"].clone(), o["mu"].clone()
                perm = torch.randperm(1024, generator=torch.Generator().manual_seed(2)).cuda()
                op = m({k: v[perm] for k, v in data.items()})
                assert rel(op["x6d"].cpu(), x6d[perm].cpu()) < 2e-5 and rel(op["mu"].cpu(), mu[perm].cpu()) < 2e-5   # (b)
                oh = m({k: v[:512] for k, v in data.items()})
                assert rel(oh["x6d"].cpu(), x6d[:512].cpu()) < 2e-5                       # (d)
        del m
        torch.cuda.empty_cache()
    for precision in ("bf16x6", "bf16x6w3", "bf16x6b3", "f16x3b3"):                       # (c)
        for k, v in losses["f32"].items():
            assert abs(losses[precision][k] - v) <= 1e-5 * abs(v), (precision, k, losses[precision][k], v)


def test_train_loop_checkpoints_and_metrics_log(golden_dir, tmp_path):
    """train() (trainer.py:322-516): epochs over a loader, weights every 5 epochs in the reference's state_dict format
    (loadable with weights_only=True and accepted by a fresh model), the test epoch from epoch 50 on, and the JSON-lines
    metrics log."""
    import json
    from scrubvae_amd.train.trainer import train
    fx, cfg, loss_scale, opt, sd, data = load_fixture(golden_dir, "vanilla_tiny")
    m, dis = build_model(cfg, sd)

    class DS(torch.utils.data.Dataset):
        kinematic_tree = cfg.kinematic_tree

        def __len__(self):
            return data["x6d"].shape[0]

        def __getitem__(self, i):
            return {k: v[i] for k, v in data.items()}

    loader = torch.utils.data.DataLoader(DS(), batch_size=8, shuffle=False)
    out = str(tmp_path) + "/"
    config = {"train": {"optimizer": "adamw", "lr": 1e-4, "lr_schedule": "cawr", "num_epochs": 51, "beta_anneal": None},
              "model": {"load_model": None, "start_epoch": 46}, "loss": dict(loss_scale), "disentangle": dis, "out_path": out,
              "data": {"batch_size": 8}}
    train(config, m, {"train": loader, "val": loader})
    lines = [json.loads(l) for l in open(out + "metrics.jsonl")]
    assert [l["epoch"] for l in lines] == [47, 48, 49, 50, 51]
    assert all("total_train" in l and "time" in l for l in lines)
    assert "total_test" in lines[3] and "total_test" not in lines[2]          # test epoch only at epoch 50 (>= 50 and % 5 == 0)
    ck = torch.load(out + "weights/epoch_50.pth", map_location="cpu", weights_only=True)
    assert set(ck) == set(sd)
    m2, _ = build_model(cfg, ck)                                               # reference-format checkpoint loads strictly
    back = m2.state_dict()
    assert all(torch.equal(back[k].cpu(), v) for k, v in ck.items())


@pytest.mark.gpu
def test_device_prefetcher_yields_the_loader_batches_in_order():
    """DevicePrefetcher == `{k: v.to(device)}` per batch (order, ragged last batch, dtypes), one batch ahead on a copy stream;
    consumed immediately by compute on the current stream without a host sync."""
    from scrubvae_amd.train.trainer import DevicePrefetcher
    g = torch.Generator().manual_seed(0)
    ds = [{"x6d": torch.randn(3, 64, 18, 6, generator=g), "ids": torch.full((3, 1), i, dtype=torch.int16)} for i in range(5)]
    ds.append({"x6d": torch.randn(1, 64, 18, 6, generator=g), "ids": torch.full((1, 1), 5, dtype=torch.int16)})
    sums = []
    for i, batch in enumerate(DevicePrefetcher(ds, "cuda")):
        assert batch["x6d"].is_cuda and batch["ids"].dtype == torch.int16
        assert int(batch["ids"][0]) == i
        sums.append(batch["x6d"].double().sum())           # queued on the compute stream right away
    assert len(sums) == 6
    for i, s in enumerate(sums):
        assert abs(float(s) - float(ds[i]["x6d"].double().sum())) < 1e-9
    assert [b["ids"].device.type for b in DevicePrefetcher(ds[:2], "cpu")] == ["cpu", "cpu"]


@pytest.mark.gpu
def test_training_trajectory_is_precision_independent(golden_dir):
    """25 AdamW steps from the same weights / batches / noise in the fp32 kernels and in the split-bf16 precisions that use
    3 products for (part of) the backward pass: the loss trajectories stay together (the reduced backward products do not
    change what is learned; Adam turns gradient noise of ANY origin into O(lr) parameter noise, so the bound is the one
    two fp32 runs with different reduction orders would also need)."""
    from scrubvae_amd import ops
    from scrubvae_amd.train.losses import get_batch_loss
    from scrubvae_amd.train.trainer import FusedAdam, clip_grad_norm_
    fx, cfg, loss_scale, opt, sd, data = load_fixture(golden_dir, "full_tiny")
    g = torch.Generator().manual_seed(3)
    eps_all = [torch.randn(8, cfg.z_dim, generator=g) for _ in range(25)]
    perms = [torch.randperm(8, generator=g) for _ in range(25)]
    keep = (ops.PRECISION, ops.SPLIT_MIN_FLOPS)
    traj = {}
    try:
        for precision in ("f32", "bf16x6", "bf16x6w3", "bf16x6b3", "f16x3b3"):
            ops.set_precision(precision)
            ops.SPLIT_MIN_FLOPS = 0.0
            model, dis = build_model(cfg, sd)
            optim = FusedAdam(model, lr=1e-3, weight_decay=0.01, decoupled=True)
            model.train()
            d = to_dev(data)
            out = []
            for s in range(25):
                d["eps"] = eps_all[s].cuda()
                bl = get_batch_loss(model, d, model(d), loss_scale, dis, adv_perm={k: perms[s] for k in cfg.method.get("adversarial_net", [])})
                bl["total"].backward()
                clip_grad_norm_(model, 1e6)
                optim.step()
                out.append(float(bl["total"].detach()))
            traj[precision] = out
    finally:
        ops.set_precision(keep[0])
        ops.SPLIT_MIN_FLOPS = keep[1]
    ref = traj["f32"]
    assert ref[-1] < 0.9 * ref[0]  # it does train
    dev = {p: max(abs(a - b) / abs(b) for a, b in zip(traj[p], ref)) for p in ("bf16x6", "bf16x6w3", "bf16x6b3", "f16x3b3")}
    print("trajectory deviation from the fp32 kernels:", dev, "final losses", {p: t[-1] for p, t in traj.items()})
    # the yardstick is bf16x6 itself: fp32-accurate in every contraction, it still drifts from the fp32 kernels because Adam
    # amplifies rounding-level gradient differences; the 3-product backward modes must not drift more than that
    for precision in ("bf16x6w3", "bf16x6b3", "f16x3b3"):
        assert dev[precision] < 3.0 * dev["bf16x6"] + 5e-3, dev  # measured 1.2x / 2.1x
    assert dev["bf16x6"] < 3e-2, dev


# ------------------------------------------------------------------ shapes off the beaten path
_ODD = {
    # name: (B, window, joints, z, channels, kernel, diag, tree)
    "B1_squeeze":      (1, 64, 18, 8, (8, 8, 16, 16, 32), 5, True, None),          # reference `.squeeze()` quirk at B=1 (residual.py:316)
    "B7_w48_z5":       (7, 48, 18, 5, (8, 8, 16, 16, 32), 5, True, None),          # nothing a multiple of anything
    "w100_odd_ch":     (3, 100, 18, 12, (12, 20, 36, 52, 68), 5, True, None),      # channel counts that are not multiples of 16
    "w128_k3":         (4, 128, 18, 8, (8, 16, 16, 32, 32), 3, False, None),       # kernel 3, full Cholesky
    "w64_k7_3blocks":  (5, 64, 18, 6, (16, 16, 32, 32), 7, True, None),            # kernel 7, three blocks
    "one_chain_J5":    (6, 32, 5, 4, (8, 8, 16, 16), 5, True, [[0, 1, 2, 3, 4]]),  # a single chain, short window
    "two_chains_J7":   (2, 64, 7, 8, (8, 8, 16, 16, 32), 5, True, [[0, 1, 2, 3], [0, 4, 5, 6]]),
}


@pytest.mark.gpu
@pytest.mark.parametrize("name", list(_ODD))
@pytest.mark.parametrize("precision", ["f32", "bf16x6b3", "f16x3b3"])
def test_odd_shapes_vs_oracle(name, precision):
    """Shapes none of the reference fixtures cover (batch 1, ragged sizes, channel counts that need padding, other kernel
    widths / depths / trees): one training step of the HIP model against the CPU oracle on seeded inputs."""
    from scrubvae_amd import ops
    from scrubvae_amd.train.losses import get_batch_loss
    B, W, J, z, ch, k, diag, tree = _ODD[name]
    cfg = O.OracleConfig(n_keypts=J, window=W, z_dim=z, kernel=k, channel=ch, diag=diag, arena_size=ARENA,
                         kinematic_tree=tree or O.skeleton_tree(J))
    sd = O.init_state_dict(cfg, seed=21)
    data = O.synth_batch(cfg, B, seed=22)
    eps = torch.randn(B, z, generator=torch.Generator().manual_seed(23))
    ls = {"jpe": 1.0, "root": 1.0, "prior": 0.7}
    # (B=1: VAE.sampling's .squeeze(), residual.py:316, drops the batch dimension of z; the comparison below flattens)
    bl_o, g_o, _, out_o = O.train_step(sd, cfg, data, ls, eps)
    keep = (ops.PRECISION, ops.SPLIT_MIN_FLOPS)
    ops.set_precision(precision)
    ops.SPLIT_MIN_FLOPS = 0.0
    try:
        model, dis = build_model(cfg, sd)
        model.train()
        d = to_dev(data)
        d["eps"] = eps.cuda()
        data_o = model(d)
        bl = get_batch_loss(model, d, data_o, ls, dis)
        bl["total"].backward()
        torch.cuda.synchronize()
    finally:
        ops.set_precision(keep[0])
        ops.SPLIT_MIN_FLOPS = keep[1]
    for kk in ("mu", "x6d", "root"):
        assert rel(data_o[kk].cpu().reshape(-1), out_o[kk].detach().reshape(-1)) < 5e-5, kk
    for kk in bl_o:
        assert rel(bl[kk].detach().cpu(), bl_o[kk]) < 1e-4, kk
    grads = {kk: v.cpu() for kk, v in model.grads_state_dict().items()}
    gmax = max(float(g.abs().max()) for g in g_o.values())
    for n, g in g_o.items():
        dd = float((grads[n] - g).abs().max()) / (float(g.abs().max()) + 1e-3 * gmax)
        assert dd < 5e-2, (n, dd)


@pytest.mark.gpu
@pytest.mark.parametrize("fused", [True, False])
@pytest.mark.parametrize("B", [1, 5, 70])
def test_full_scvae_odd_batch_vs_oracle(B, fused, monkeypatch):
    """configs[2]'s heads (conditional + gradient reversal incl. the ids classifier + adversarial net) at batch sizes the
    fixtures do not have (one sample, a ragged tile, more than one 64-row tile), against the oracle with the shuffle
    permutation injected -- on the fused ensemble kernels and on the Linear-by-Linear GEMM path wide ensembles take."""
    from scrubvae_amd import ops
    from scrubvae_amd.train.losses import get_batch_loss
    monkeypatch.setattr(ops, "FUSED_ENSEMBLE", fused)
    feats = ["avg_speed_3d", "heading", "ids"]
    cfg = O.OracleConfig(n_keypts=18, window=64, z_dim=8, kernel=5, channel=(8, 8, 16, 16, 32), diag=True, arena_size=ARENA,
                         method={"conditional": feats, "grad_reversal": feats, "adversarial_net": ["heading"]}, features=feats,
                         discrete_classes={"ids": torch.arange(4)})
    sd = O.init_state_dict(cfg, seed=31)
    data = O.synth_batch(cfg, B, seed=32)
    g = torch.Generator().manual_seed(33)
    eps, perm = torch.randn(B, 8, generator=g), torch.randperm(B, generator=g)
    ls = {"jpe": 1.0, "root": 1.0, "prior": 0.5, "avg_speed_3d_gr": 1.0, "heading_gr": 2.0, "ids_gr": 0.5, "heading_an": 0.5}
    bl_o, g_o, _, out_o = O.train_step(sd, cfg, data, ls, eps, adv_perm={"heading": perm})
    model, dis = build_model(cfg, sd)
    model.train()
    d = to_dev(data)
    d["eps"] = eps.cuda()
    data_o = model(d)
    assert all(r.fused == fused for r in model._runners.values())
    bl = get_batch_loss(model, d, data_o, ls, dis, adv_perm={"heading": perm})
    bl["total"].backward()
    for kk in bl_o:
        assert rel(bl[kk].detach().cpu(), bl_o[kk]) < 1e-4, kk
    grads = {kk: v.cpu() for kk, v in model.grads_state_dict().items()}
    gmax = max(float(x.abs().max()) for x in g_o.values())
    for n, x in g_o.items():
        dd = float((grads[n] - x).abs().max()) / (float(x.abs().max()) + 1e-3 * gmax)
        assert dd < 5e-2, (n, dd)


@pytest.mark.gpu
def test_overlapped_fast_path_equals_serial_schedule():
    """The schedule of the benchmark / train_epoch fast path -- side streams on, the fused tail forked beside the scrubbing losses,
    the heads' backward run early beside it (model.defer_tail, train/losses.py) -- against the plain serial schedule on the same
    model, batch, noise and shuffle: every loss term and every gradient (the orders of a few additions differ, nothing else), and a
    second model of the process takes its side streams from the same pool."""
    from scrubvae_amd.train.losses import get_batch_loss
    from scrubvae_amd.model import residual
    feats = ["avg_speed_3d", "heading", "ids"]
    cfg = O.OracleConfig(n_keypts=18, window=64, z_dim=8, kernel=5, channel=(8, 8, 16, 16, 32), diag=True, arena_size=ARENA,
                         method={"conditional": feats, "grad_reversal": feats, "adversarial_net": ["heading"]}, features=feats,
                         discrete_classes={"ids": torch.arange(4)})
    sd = O.init_state_dict(cfg, seed=41)
    B = 70
    data = O.synth_batch(cfg, B, seed=42)
    g = torch.Generator().manual_seed(43)
    eps, perm = torch.randn(B, 8, generator=g), torch.randperm(B, generator=g)
    ls = {"jpe": 1.0, "root": 1.0, "prior": 0.5, "avg_speed_3d_gr": 1.0, "heading_gr": 2.0, "ids_gr": 0.5, "heading_an": 0.5}
    res, models = [], []
    for fast in (False, True):
        model, dis = build_model(cfg, sd)
        model.train()
        model.overlap_wgrad = fast
        model.defer_tail = fast
        d = to_dev(data)
        d["eps"] = eps.cuda()
        bl = get_batch_loss(model, d, model(d), ls, dis, adv_perm={"heading": perm})
        if fast:
            assert model._pending.get("scrub_done")  # the heads' backward already ran, beside the tail
        bl["total"].backward()
        torch.cuda.synchronize()
        res.append(({k: float(v.detach()) for k, v in bl.items()}, {k: v.clone() for k, v in model.grads_state_dict().items()}))
        models.append(model)
    (l0, g0), (l1, g1) = res
    for k in l0:
        assert abs(l0[k] - l1[k]) <= 1e-6 * max(1.0, abs(l0[k])), k
    gmax = max(float(x.abs().max()) for x in g0.values())
    for n in g0:
        assert float((g0[n] - g1[n]).abs().max()) <= 1e-5 * (float(g0[n].abs().max()) + 1e-3 * gmax), n
    assert models[1]._sides and all(a is b for a, b in zip(models[1]._sides, residual._SIDE_STREAMS[0]))


@pytest.mark.gpu
@pytest.mark.parametrize("fast", [False, True])
def test_accumulate_grads_adds_a_second_backward(fast):
    """model.accumulate_grads = True makes total.backward() ADD to the gradients (gradient accumulation over micro-batches):
    the same batch twice gives twice the gradients of one pass -- on the serial schedule and on the overlapped fast path (tail and
    heads' backward beside each other, bias-gradient sums sharing a dY)."""
    from scrubvae_amd.train.losses import get_batch_loss
    feats = ["avg_speed_3d", "heading", "ids"]
    cfg = O.OracleConfig(n_keypts=18, window=64, z_dim=8, kernel=5, channel=(8, 8, 16, 16, 32), diag=True, arena_size=ARENA,
                         method={"conditional": feats, "grad_reversal": feats, "adversarial_net": ["heading"]}, features=feats,
                         discrete_classes={"ids": torch.arange(4)})
    sd = O.init_state_dict(cfg, seed=51)
    B = 37
    data = O.synth_batch(cfg, B, seed=52)
    g = torch.Generator().manual_seed(53)
    eps, perm = torch.randn(B, 8, generator=g), torch.randperm(B, generator=g)
    ls = {"jpe": 1.0, "root": 1.0, "prior": 0.5, "avg_speed_3d_gr": 1.0, "heading_gr": 2.0, "ids_gr": 0.5, "heading_an": 0.5}
    model, dis = build_model(cfg, sd)
    model.train()
    model.overlap_wgrad = fast
    model.defer_tail = fast
    d = to_dev(data)
    d["eps"] = eps.cuda()

    def one_pass():
        bl = get_batch_loss(model, d, model(d), ls, dis, adv_perm={"heading": perm})
        bl["total"].backward()
        torch.cuda.synchronize()
        return {k: v.clone() for k, v in model.grads_state_dict().items()}

    g1 = one_pass()
    model.accumulate_grads = True
    g2 = one_pass()
    gmax = max(float(x.abs().max()) for x in g1.values())
    for n in g1:
        if n.endswith(("running_mean", "running_var", "num_batches_tracked")):
            continue
        assert float((g2[n] - 2 * g1[n]).abs().max()) <= 2e-5 * (float(g1[n].abs().max()) + 1e-3 * gmax), n

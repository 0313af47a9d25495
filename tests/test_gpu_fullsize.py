"""One whole training step at BASELINE's full sizes -- forward, every loss term, the BACKWARD pass and the AdamW update --
on exactly the kernel mix `bench.py` times, against the CPU oracle's `train_step` on the same seeded inputs.

  * configs[1]: B=1024, W=64, J=23, default channels, recon + KL, precisions `f16x3b3` (bench default) and `bf16x6b3`, the shipped `tuned_tiles.json`,
    the library's default `SPLIT_MIN_FLOPS` (so the large layers run the split-bf16 halo / 12-wave halo / all-taps
    weight-gradient templates and the small ones the fp32 kernels, as in the timed region);
  * configs[2]: B=4096, the full SC-VAE head set (conditional + 2 gradient-reversal ensembles + adversarial net), the
    shuffle permutation injected on both sides.

Reference lines: trainer.py:126-167 (the step), losses.py:182-324, get/model.py:4-18.

Gates (fp32 tolerances of DESIGN.md 2): outputs 2e-5 max-norm relative and every loss term 1e-4 relative against the fp32
oracle.  Gradients are judged against the oracle run in fp64 (the truth: the reference's own fp32 gradients sit 1e-3 -- whole
vector -- to 2e-2 -- PReLU slopes, cancellation-prone sums over millions of terms -- away from it at these sizes): each tensor
within 5e-2 of the truth in the scale-aware max-norm of test_oracle_golden (denominator max(|g|) + 1e-3 of the global gradient
scale; for the single PReLU slopes + 1e-2 of the largest slope gradient, see the comment in `_run`) -- or within 4x the fp32 oracle's own error where that is larger (single PReLU slopes included: their partial sums are formed
in fp64) -- and within that bound + the fp32 oracle's own error of the fp32
oracle; the whole gradient vector no further from the truth than
2x the fp32 oracle is (measured with the shipped tile table: 1.7e-3 vs 1.1e-3 at B=1024, 1.3e-3 vs 1.5e-3 at B=4096; which layers'
data-gradients run the 3-product split kernels rather than the fp32 ones is the tuner's choice and moves the first number between
0.9e-3 and 1.7e-3); the total gradient norm within 1e-3 of
both; the parameters after the fused AdamW step within one sign-flip of Adam's first update (2.5 lr) of the oracle's.  The
oracles cost ~1 + 3 s (B=1024) and ~8 + 20 s (B=4096) of host time.
"""
import dataclasses
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import scvae_oracle as O
from tests.test_oracle_golden import ARENA, FULL_METHODS, rel
from tests.test_gpu_model import build_model, to_dev

LR = 1e-4


WIDE6 = (64, 128, 256, 512, 1024, 2048, 4096)  # BASELINE configs[4]: six residual blocks up to 4096 channels, window 256


def _run(B, full, precision, seed, window=64, channel=None, expect=("gather_halo_bf16s_kernel", "gather_halo_ws4", "wgrad_taps"),
         out_tol=2e-5, report=None):
    """report: a list -- per-tensor (name, numel, error vs fp64, fp32 oracle's error vs fp64) instead of the per-tensor gates
    (tools/fullsize_slopes.py)"""
    from scrubvae_amd import ops
    from scrubvae_amd.train.losses import get_batch_loss
    from scrubvae_amd.train.trainer import FusedAdam, clip_grad_norm_
    feats = ["avg_speed_3d", "heading"]
    cfg = O.OracleConfig(n_keypts=23, window=window, z_dim=32, kernel=5, diag=True, arena_size=ARENA, kinematic_tree=O.skeleton_tree(23),
                         method=dict(FULL_METHODS) if full else {}, features=feats if full else None,
                         discrete_classes={"ids": torch.arange(4)} if full else None,
                         **({} if channel is None else {"channel": tuple(channel)}))
    ls = {"jpe": 1.0, "root": 1.0, "prior": 1.0}
    if full:
        ls.update({"avg_speed_3d_gr": 1.0, "heading_gr": 1.0, "heading_an": 1.0})
    sd = O.init_state_dict(cfg, seed=seed)
    data = O.synth_batch(cfg, B, seed=seed + 1)
    g = torch.Generator().manual_seed(seed + 2)
    eps = torch.randn(B, cfg.z_dim, generator=g)
    perm = {k: torch.randperm(B, generator=g) for k in cfg.method.get("adversarial_net", [])}
    bl_o, g_o, sd_o, out_o = O.train_step(sd, cfg, data, ls, eps, adv_perm=perm, lr=LR)
    c64 = dataclasses.replace(cfg, arena_size=ARENA.double())
    sd64 = {k: (v.double() if v.dtype.is_floating_point else v) for k, v in sd.items()}
    d64 = {k: (v.double() if v.dtype.is_floating_point else v) for k, v in data.items()}
    _, g64, _, _ = O.train_step(sd64, c64, d64, ls, eps.double(), adv_perm=perm, lr=LR)
    del sd64, d64
    keep = ops.PRECISION
    ops.set_precision(precision)
    tuned_before = set(ops.TUNED_LOG)
    try:
        model, dis = build_model(cfg, sd)
        model.train()
        model.defer_tail = True  # bench.py's schedule: one fused tail launch per step
        opt = FusedAdam(model, lr=LR, weight_decay=0.01, decoupled=True)
        d = to_dev(data)
        d["eps"] = eps.cuda()
        data_o = model(d)
        bl = get_batch_loss(model, d, data_o, ls, dis, adv_perm=perm or None)
        for p in model.parameters():
            p.grad = None
        bl["total"].backward()
        gn = clip_grad_norm_(model, 1e6)
        torch.cuda.synchronize()
        grads = {k: v.cpu().clone() for k, v in model.grads_state_dict().items()}
        outs = {k: data_o[k].detach().cpu().clone() for k in ("mu", "z", "x6d", "root")}
        names = {cv.kernel_name(kind) for cv in model._convs.values() for kind in ("fwd", "dgrad", "wgrad")
                 if kind in cv.__dict__.get("_tuned", ())}
        opt.step()
        torch.cuda.synchronize()
        new_sd = {k: v.cpu() for k, v in model.state_dict().items()}
    finally:
        ops.set_precision(keep)
    # the kernel mix is the benchmark's: every layer found its entry in the shipped tile table (nothing was tuned here)
    assert set(ops.TUNED_LOG) == tuned_before, sorted(set(ops.TUNED_LOG) - tuned_before)
    assert all(cv.desc.tile[0] != 0 for cv in model._convs.values() if cv.flops >= ops.AUTOTUNE_MIN_FLOPS)
    for e in expect:  # the kernel families the benchmark's tile table takes for this workload really ran
        assert any(e in n for n in names), (e, sorted(names))
    for k in outs:
        assert rel(outs[k].reshape(-1), out_o[k].detach().reshape(-1)) < out_tol, k
    for k in bl_o:
        assert rel(bl[k].detach().cpu(), bl_o[k]) < 1e-4, (k, float(bl[k]), float(bl_o[k]))
    gmax = max(float(x.abs().max()) for x in g64.values())
    # single scalars (PReLU slopes) are sums over a whole activation map: their error scales with the magnitudes of the terms, not with
    # the (possibly cancelled) value of the sum, and the other slopes of the model give that scale.  configs[1], B = 1024: sixteen
    # slopes of 0.04 .. 8.8 carry absolute errors of 3e-4 .. 6e-3 (HIP) / 4e-6 .. 4e-3 (fp32 oracle); the seventeenth is 1.2e-4 by
    # cancellation and carries 2.3e-3 / 2.2e-4 like the rest (tools/fullsize_slopes.py) -- measured against its own value + 1e-3 of
    # the global scale that reads 16 % one run and 4 % the next.  Their denominator floor is therefore 1e-2 of the largest slope gradient.
    s_cls = max((float(t.abs().max()) for t in g64.values() if t.numel() == 1), default=0.0)
    worst = ("", 0.0, 0.0)
    for n, t in g64.items():
        den = float(t.abs().max()) + (max(1e-3 * gmax, 1e-2 * s_cls) if t.numel() == 1 else 1e-3 * gmax)
        e_hip = float((grads[n].double() - t).abs().max()) / den
        e_cpu = float((g_o[n].double() - t).abs().max()) / den
        e_pair = float((grads[n] - g_o[n]).abs().max()) / (float(g_o[n].abs().max()) + den - float(t.abs().max()))
        if e_hip > worst[1]:
            worst = (n, e_hip, e_cpu)
        # (single PReLU slopes are sums of ~1e6 cancelling terms: the fp32 oracle itself is up to 2.2e-2 off there.  Their partials
        #  are formed in fp64 -- products and sums, csrc tile_epilogue / affine_prelu_bwd_partial_kernel -- so what is left is the
        #  noise of the fp32-class inputs, as in the oracle: 4.4e-2 (f16x3b3) / 5.9e-2 (bf16x6b3, and the same with six products in
        #  the backward pass) at B=1024, 8e-3 at B=4096; no special case for them)
        gate = max(5e-2, 4 * e_cpu)
        if report is not None:
            report.append((n, t.numel(), e_hip, e_cpu, float(t.abs().max()), gmax))
            continue
        assert e_hip < gate, (n, e_hip, e_cpu)
        assert e_pair < gate + e_cpu, (n, e_pair, e_cpu)
    nrm = lambda ts: torch.sqrt(sum((x.double() ** 2).sum() for x in ts))
    gn64 = nrm(g64.values())
    gn_o, gn_h = nrm(g_o.values()), nrm([grads[n] for n in g64])
    assert rel(gn_h, gn64) < 1e-3 and rel(gn_h, gn_o) < 1e-3, (float(gn_h), float(gn_o), float(gn64))
    assert rel(gn.cpu().double(), gn64) < 1e-3  # clip_grad_norm_'s device-side norm is the same number
    v_hip = float(nrm([grads[n].double() - g64[n] for n in g64]) / gn64)
    v_cpu = float(nrm([g_o[n].double() - g64[n] for n in g64]) / gn64)
    print(f"\n[B={B} full={full} {precision}] |g_hip - g_64| / |g_64| = {v_hip:.2e} (fp32 CPU oracle: {v_cpu:.2e}); worst tensor {worst[0]} "
          f"{worst[1]:.2e} (oracle {worst[2]:.2e}); grad norm {float(gn_h):.6g} vs fp64 {float(gn64):.6g}")
    assert v_hip < 2.0 * v_cpu + 2e-4, (v_hip, v_cpu)
    for n in O.trainable_names(sd):
        dv = float((new_sd[n] - sd_o[n]).abs().max())
        assert dv <= 2.5 * LR + 1e-5 * float(sd_o[n].abs().max()), (n, dv)
    for n in sd:  # frozen discriminator + buffers: the optimizer must not touch them
        if n.startswith("disentangle.adversarial_net."):
            assert torch.equal(new_sd[n], sd[n]), n
        if "running_" in n:
            assert rel(new_sd[n], sd_o[n]) < 1e-4, n


@pytest.mark.parametrize("precision", ["bf16x6b3", "f16x3b3"])
def test_config1_b1024_whole_step_vs_oracle(precision):
    _run(1024, False, precision, seed=41)


@pytest.mark.parametrize("precision", ["bf16x6b3", "f16x3b3"])
def test_config2_b4096_full_heads_whole_step_vs_oracle(precision):
    _run(4096, True, precision, seed=51)


@pytest.mark.parametrize("precision", ["f16x3b3", "bf16x6b3"])
def test_config4_wide_w256_b32_whole_step_vs_oracle(precision):
    """BASELINE configs[4] at its full widths (window 256, six residual blocks 64..4096 channels, 293 M parameters, 70-tap output
    conv; reference shape: residual.py:183-292 with window=256) at a batch where the split-bf16 kernels engage (B = 32: the deep
    layers run the wave-specialised / all-taps templates of the shipped tile table, nothing is tuned on the fly), in the bench's
    default precision and in `bf16x6b3`, under the same gates as configs[1] / configs[2] above."""
    # outputs: 5e-5 as in test_oracle_parity_config5_wide_w256 (six blocks and a 70-tap output conv deep: measured 1.5e-5 / 2.4e-5)
    _run(32, False, precision, seed=61, window=256, channel=WIDE6, expect=("gather_gemm_bf16s_ws_kernel", "wgrad_taps"), out_tol=5e-5)

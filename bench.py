#!/usr/bin/env python
"""Headline benchmark: pose-windows/sec through one full SC-VAE optimizer step on MI355X.

    python bench.py --gpus N --steps K --warmup W
    (N>1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...)

Workload (BASELINE.json configs[1]): synthetic 64-frame, 23-joint mouse skeletons, default
residual-CNN channels [64,128,256,512,1024], z=32, batch 1024 per GPU, recon + KL
(loss = {jpe, root, prior}), AdamW, fp32 compute (exact-fp32 MFMA).  A step = forward + all
configured losses + backward + grad-norm + optimizer step, inputs already resident in HBM.
Weak scaling: the per-GPU batch is fixed as N grows; every loss is normalised by the global
batch, gradients are summed over ranks with RCCL and BatchNorm statistics are synchronised,
so N ranks compute the 1-rank result at the global batch.

Prints ONE JSON line (rank 0) with the contract fields plus
  roofline     -- the dominant kernel (fp32 MFMA implicit-GEMM), algorithmic FLOPs / launch
                  over its mean launch time, measured live with HIP events in the timed region;
  cpu_baseline -- the CPU oracle (oracle/scvae_oracle.py, stock PyTorch-CPU ops = the
                  reference's own arithmetic) timed on this box's host cores on a bounded
                  sample of the same workload.
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_F32_MFMA_TFLOPS = 157.3  # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32 dense peak
PEAK_BF16_MFMA_TFLOPS = 2500.0  # MI355X_MICROARCH.md: dense bf16 MFMA peak (v_mfma_f32_32x32x16_bf16)
PRODUCTS = {"f32": 1, "bf16x6": 6, "bf16x6w3": 6, "bf16x6b3": 6, "bf16x3": 3, "bf16": 1}  # matrix-core products per algorithmic multiply (fwd / dgrad)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=1024, help="windows per GPU")
    ap.add_argument("--joints", type=int, default=23)
    ap.add_argument("--window", type=int, default=64)
    ap.add_argument("--channels", default=",".join(map(str, CHANNELS)),
                    help="model.channel, comma separated; 'wide6' = configs[4]'s six blocks 64..4096 (use with --window 256)")
    ap.add_argument("--full", action="store_true", help="configs[2]: conditional + grad-reversal + adversarial heads")
    ap.add_argument("--sync-bn", action="store_true",
                    help="N>1: all-reduce the BatchNorm batch statistics (32 small collectives per step) so that N ranks reproduce "
                         "the 1-rank step at the global batch exactly; default = per-rank statistics, the semantics of "
                         "torch DistributedDataParallel around the reference model")
    ap.add_argument("--local-bn", action="store_true", help="(default; kept for older command lines)")
    ap.add_argument("--graph", action="store_true", help="replay the whole step as one hipGraph (single GPU)")
    ap.add_argument("--precision", default=os.environ.get("SVAE_PRECISION", "bf16x6b3"), choices=list(PRODUCTS),
                    help="arithmetic of the large contractions: f32 = fp32 MFMA; bf16x6 = fp32-accurate 3-piece split on the bf16 "
                         "matrix cores (6 products); bf16x6w3 = the same with 2 pieces / 3 products for the weight-gradient "
                         "contractions (gradient error vs fp64 unchanged, DESIGN.md 4); bf16x6b3 = 3 products for the whole "
                         "backward pass, 6 for the forward; bf16x3 / bf16 = 2 / 1 pieces everywhere "
                         "(reduced accuracy, study only)")
    ap.add_argument("--serial-streams", action="store_true",
                    help="timed region without the concurrent side streams (what the roofline region always uses)")
    ap.add_argument("--h2d", action="store_true",
                    help="copy the batch from pinned host memory inside every timed step (the PCIe-inclusive rate DESIGN.md quotes; "
                         "never the headline value: the metric is defined with inputs resident in HBM)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--timer-kinds", default="fwd,dgrad", help="GEMM kinds bracketed with HIP events (fwd,dgrad,wgrad)")
    args = ap.parse_args()
    args.channel_list = WIDE6 if args.channels == "wide6" else [int(c) for c in args.channels.split(",")]
    return args


CHANNELS = [64, 128, 256, 512, 1024]
WIDE6 = [64, 128, 256, 512, 1024, 2048, 4096]
ARENA = [[-1.0, -1.0, -1.0], [1.0, 1.0, 1.0]]


def make_cfg(args):
    method, feats, loss = {}, [], {"jpe": 1.0, "root": 1.0, "prior": 1.0}
    if args.full:
        method = {"conditional": ["avg_speed_3d", "heading"], "grad_reversal": ["avg_speed_3d", "heading"],
                  "adversarial_net": ["heading"]}
        feats = ["avg_speed_3d", "heading"]
        loss.update({"avg_speed_3d_gr": 1.0, "heading_gr": 1.0, "heading_an": 1.0})
    return method, feats, loss


def build_model(args, method, feats, tree):
    from scrubvae_amd.get import model as get_model
    mc = dict(type="rcnn", kernel=5, z_dim=32, window=args.window, activation="prelu", diag=True,
              init_dilation=None, prior="gaussian", channel=args.channel_list)
    dis = dict(method=method, alpha=1.0, features=feats)
    torch.manual_seed(0)
    m = get_model(mc, None, None, dis, args.joints, "midfwd", arena_size=torch.tensor(ARENA), kinematic_tree=tree,
                  discrete_classes={"ids": torch.arange(4)} if args.full else None, device="cuda", verbose=0)
    return m, dis


def cpu_baseline(args, method, feats, loss, sample_b=128, steps=50):
    """cpu_baseline leg: the CPU oracle's train_step (the only place bench.py touches oracle/)
    on a bounded sample (sample_b windows) of the same workload."""
    from oracle import scvae_oracle as O
    # keep the sample at ~10-30 s of CPU work whatever the model size (default model: 0.76 GFLOP/window/step)
    def cost(ch, w):
        return sum(a * b * (w >> (i + 1)) for i, (a, b) in enumerate(zip(ch, ch[1:])))
    work = cost(args.channel_list, args.window) / cost(CHANNELS, 64)
    if work > 1.5:
        sample_b, steps = max(4, int(128 / work) // 4 * 4), max(2, int(50 / work))
    cfg = O.OracleConfig(n_keypts=args.joints, window=args.window, z_dim=32, kernel=5, diag=True, channel=tuple(args.channel_list),
                         arena_size=torch.tensor(ARENA), kinematic_tree=O.skeleton_tree(args.joints), method=method,
                         features=feats, discrete_classes={"ids": torch.arange(4)} if args.full else None)
    torch.set_num_threads(max(1, min(len(os.sched_getaffinity(0)), 16)))  # the box's CPU share
    sd = O.init_state_dict(cfg, seed=0)
    data = O.synth_batch(cfg, sample_b, seed=0)
    eps = torch.randn(sample_b, cfg.z_dim)
    perm = {k: torch.randperm(sample_b) for k in cfg.method.get("adversarial_net", [])}
    state = {}
    bl0, _, _, _ = O.train_step(sd, cfg, data, loss, eps, adv_perm=perm, opt_state=state)  # warm-up; also the ELBO reference
    elbo = elbo_check(args, cfg, sd, data, eps, perm, loss, method, feats, bl0)
    t0 = time.perf_counter()
    for _ in range(steps):
        _, _, sd, _ = O.train_step(sd, cfg, data, loss, eps, adv_perm=perm, opt_state=state)
    dt = (time.perf_counter() - t0) / steps
    return {"value": round(sample_b / dt, 2), "unit": "windows/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"{steps} optimizer steps of the CPU oracle at batch {sample_b} (same model/loss config), {dt*1e3:.0f} ms/step",
            "elbo_match": elbo}


def elbo_check(args, cfg, sd, data, eps, perm, loss, method, feats, oracle_losses):
    """The "ELBO-match" condition of the metric, checked on the benchmark's own model size and precision: the oracle's
    weights, batch and noise on the HIP path; relative deviation of every loss term (bound: 1e-4)."""
    from scrubvae_amd.get import model as get_model
    from scrubvae_amd.train.losses import get_batch_loss
    mc = dict(type="rcnn", kernel=5, z_dim=32, window=args.window, activation="prelu", diag=True, init_dilation=None,
              prior="gaussian", channel=args.channel_list)
    dis = dict(method=method, alpha=1.0, features=feats)
    m = get_model(mc, None, None, dis, args.joints, "midfwd", arena_size=torch.tensor(ARENA), kinematic_tree=cfg.kinematic_tree,
                  discrete_classes=cfg.discrete_classes, device="cuda", verbose=0)
    m.load_state_dict(sd, strict=False)
    m.train()
    d = {k: v.cuda() for k, v in data.items()}
    d["eps"] = eps.cuda()
    bl = get_batch_loss(m, d, m(d), loss, dis, adv_perm=perm or None)
    rel = {k: abs(float(bl[k].detach()) - float(v.detach())) / (abs(float(v.detach())) + 1e-30) for k, v in oracle_losses.items()}
    worst = max(rel.values())
    del m
    torch.cuda.empty_cache()
    return {"batch": int(eps.shape[0]), "precision": args.precision, "max_rel_dev_of_loss_terms_vs_cpu_oracle": float(f"{worst:.3g}"),
            "total_rel_dev": float(f"{rel['total']:.3g}"), "bound": 1e-4, "ok": bool(worst <= 1e-4)}


def pmc_traffic(kernel):
    """HBM bytes per launch of `kernel` from the committed rocprofv3 PMC passes (separate
    --pmc FETCH_SIZE / WRITE_SIZE runs of this same command, gfx950 corrections applied by
    tools/summarize_pmc.py); None when no summary for this kernel is committed."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_traffic.json")))
    if not files:
        return None
    try:
        k = json.load(open(files[-1]))["kernels"].get(kernel)
        return None if not k or not k.get("hbm_bytes_per_launch") else {
            "hbm_bytes_per_launch": round(k["hbm_bytes_per_launch"]), "source": os.path.basename(files[-1])}
    except Exception:
        return None


def main():
    args = parse()
    # stdout carries exactly ONE line (the JSON): whatever libraries print while the job runs (RCCL's version banner,
    # gloo's connection messages, ...) is sent to stderr instead -- file descriptor 1 is pointed at stderr until the end
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)
    from scrubvae_amd import parallel, ops
    from scrubvae_amd.train.losses import get_batch_loss
    from scrubvae_amd.train.trainer import FusedAdam, clip_grad_norm_
    rank, local, world = parallel.init_distributed()
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}")
    torch.cuda.set_device(local % max(torch.cuda.device_count(), 1))
    ops.set_precision(args.precision)
    from scrubvae_amd.data import synthetic
    method, feats, loss = make_cfg(args)
    B = args.batch
    data, tree = synthetic.make_batch(args.joints, args.window, B, seed=100 + rank, device="cuda")
    data0 = data
    model, dis = build_model(args, method, feats, tree)
    parallel.attach(model, sync_bn=args.sync_bn)
    model.defer_tail = True  # one fused tail launch per step (outputs + losses + seed gradients)
    opt = FusedAdam(model, lr=1e-4, weight_decay=0.01, decoupled=True)
    model.train()

    graphed = None
    if args.graph:
        from scrubvae_amd.train.trainer import GraphedStep
        graphed = GraphedStep(model, opt, loss, dis, data)
        args.no_roofline = True  # per-launch events cannot be recorded inside a replay

    feed = None
    if args.h2d:  # every step consumes a batch that DevicePrefetcher copied from pinned host memory one step ahead
        import itertools
        from scrubvae_amd.train.trainer import DevicePrefetcher
        host = {k: v.cpu().pin_memory() for k, v in data.items()}
        feed = iter(DevicePrefetcher(itertools.repeat(host), "cuda"))

    def step():
        if graphed is not None:
            return graphed()
        data = next(feed) if feed is not None else data0
        data_o = model(data)
        bl = get_batch_loss(model, data, data_o, loss, dis)
        bl["total"].backward()
        clip_grad_norm_(model, 1e6)
        opt.step()
        return bl

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    # like train_test_epoch: the steps run on the high-priority compute stream (trainer.on_compute_stream)
    from scrubvae_amd.train.trainer import on_compute_stream
    torch.cuda.synchronize()
    hp = on_compute_stream("cuda")
    hp.__enter__()
    for _ in range(args.warmup):
        step()
    if args.serial_streams:
        model.overlap_wgrad = False
    # ---- headline timed region: exactly K steps, barrier + synchronize on both sides, no
    # per-launch instrumentation
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        bl = step()
    t_host = time.perf_counter() - t0  # all launches of the K steps are queued (the GPU is still working)
    barrier()
    dt = time.perf_counter() - t0
    # ---- roofline region: the same K steps again with the side streams serialised and HIP
    # events around every launch of the dominant GEMM template.  In the headline region three
    # HIP streams run kernels concurrently on shared CUs, so a launch's start-to-end time is
    # not the kernel's own time there; serialised, it is.
    timer = probe = None
    if not args.no_roofline and graphed is None:
        keep = model.overlap_wgrad
        model.overlap_wgrad = False
        probe = ops.LaunchTimer(kinds=tuple(args.timer_kinds.split(",")))
        ops.TIMER = probe
        step()
        ops.TIMER = None
        dominant = max(probe.summary().items(), key=lambda kv: kv[1]["ms"])[0]
        timer = ops.LaunchTimer(kinds=tuple(args.timer_kinds.split(",")), only=dominant)
        ops.TIMER = timer
        barrier()
        t1 = time.perf_counter()
        for _ in range(args.steps):
            step()
        barrier()
        dt_serial = time.perf_counter() - t1
        ops.TIMER = None
        model.overlap_wgrad = keep
    hp.__exit__(None, None, None)
    if world > 1:  # MAX over ranks
        tt = torch.tensor([dt], dtype=torch.float64, device="cuda" if torch.distributed.get_backend() == "nccl" else "cpu")
        torch.distributed.all_reduce(tt, op=torch.distributed.ReduceOp.MAX)
        dt = float(tt)
    total_loss = float(bl["total"].detach())

    if rank == 0:
        ms = dt / args.steps * 1e3
        out = {
            "metric": f"pose-windows/sec (ELBO-match) on synthetic {args.window}-frame mouse skeletons",
            "value": round(B * world * args.steps / dt, 1), "unit": "windows/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(ms, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32" if args.precision == "f32" else f"f32 ({args.precision} split on the bf16 matrix cores)",
            "data": "synthetic",
            "config": {"workload": ("configs[2] full SC-VAE (conditional + grad_reversal x2 + adversarial_net)" if args.full else
                                    "configs[4] wide six-block rcnn, recon+KL (jpe+root+prior), AdamW" if args.channel_list == WIDE6 else
                                    "configs[1] mouse-skeleton rcnn, recon+KL (jpe+root+prior), AdamW") +
                                   f", batch {B}/GPU, window {args.window}, {args.joints} joints, z=32, channels [{','.join(map(str, args.channel_list))}]",
                       "batch_per_gpu": B, "global_batch": B * world, "window": args.window, "joints": args.joints,
                       "launch": "hipGraph replay" if args.graph else "eager launches",
                       "inputs": ("copied from pinned host memory for every step, one batch ahead on a copy stream (PCIe-inclusive)"
                                  if args.h2d else "resident in HBM"),
                       "streams": "serialised" if args.serial_streams else "3 HIP streams (weight gradients / skip branches overlap the main chain)",
                       "precision": ("fp32 MFMA (v_mfma_f32_32x32x2_f32)" if args.precision == "f32" else
                                     f"{args.precision}: fp32 storage and accumulation; every large contraction splits its fp32 operands into "
                                     f"bf16 pieces and runs {PRODUCTS[args.precision]} cross product(s) on v_mfma_f32_32x32x16_bf16"
                                     + (" (fp32-accurate, DESIGN.md 4)" if args.precision == "bf16x6" else
                                        " forward and data-gradient (fp32-accurate), 3 (2 pieces) for the weight-gradient contractions "
                                        "(gradient error vs fp64 unchanged, DESIGN.md 4)" if args.precision == "bf16x6w3" else
                                        " forward (outputs, losses, ELBO: fp32-accurate), 3 (2 pieces) for the data- and weight-gradient "
                                        "contractions (gradient error vs fp64 within the fp32 path's own, DESIGN.md 4)"
                                        if args.precision == "bf16x6b3" else " (reduced accuracy)")),
                       "host_enqueue_ms_per_step": round(t_host / args.steps * 1e3, 3),
                       "parallelism": f"dp{world}" + ("" if world == 1 else ("+syncbn" if args.sync_bn else "+localbn")),
                       "final_total_loss": total_loss},
        }
        if timer is not None:
            summ = timer.summary()
            if summ:
                # dominant kernel = the GEMM template instance with the largest total time
                kname, s = max(summ.items(), key=lambda kv: kv[1]["ms"])
                tf = s["flops"] / (s["ms"] * 1e-3) / 1e12
                split_kernel = "bf16s" in kname
                # products per algorithmic multiply of THIS template: its piece count is the third template argument
                import re
                pm = re.search(r"bf16s\w*<\d+, \d+, (\d)", kname)
                k_products = {3: 6, 2: 3, 1: 1}[int(pm.group(1))] if pm else PRODUCTS[args.precision]
                peak = PEAK_BF16_MFMA_TFLOPS / k_products if split_kernel else PEAK_F32_MFMA_TFLOPS
                out["roofline"] = {"measured": f"second timed region of {args.steps} steps with the side HIP streams serialised "
                                               f"({dt_serial / args.steps * 1e3:.3f} ms/step; the headline region overlaps kernels "
                                               "on 3 streams, where a launch's duration is not the kernel's own time); "
                                               "profiles/*serial* is the rocprofv3 summary of `bench.py --serial-streams`",
                                   "bound": "mfma", "achieved": round(tf, 2), "peak": round(peak, 1), "unit": "TFLOP/s",
                                   "frac": round(tf / peak, 4),
                                   # the committed PMC passes are of the default command (configs[1], batch 1024)
                                   "traffic": pmc_traffic("svae::" + kname) if (args.channel_list == CHANNELS and args.window == 64 and B == 1024
                                                                                and not args.full) else None,
                                   "peak_note": (f"dense bf16 MFMA peak {PEAK_BF16_MFMA_TFLOPS:.0f} TFLOP/s / {k_products} "
                                                 "matrix-core products per algorithmic multiply (achieved counts algorithmic FLOPs)"
                                                 if split_kernel else "dense fp32 MFMA peak (v_mfma_f32_32x32x2_f32)"),
                                   "kernel": "svae::" + kname,
                                   "launches_per_step": s["launches"] // args.steps,
                                   "avg_launch_us": round(s["ms"] * 1e3 / s["launches"], 2),
                                   "avg_launch_gflop": round(s["flops"] / s["launches"] / 1e9, 3),
                                   "probe_step_all_gemm_templates": {
                                       k: {"TFLOP/s": round(v["flops"] / (v["ms"] * 1e-3) / 1e12, 2),
                                           "avg_us": round(v["ms"] * 1e3 / v["launches"], 2), "launches_per_step": v["launches"]}
                                       for k, v in sorted(probe.summary().items())}}
        if not args.no_cpu_baseline and world == 1:  # reported at N=1 only
            out["cpu_baseline"] = cpu_baseline(args, method, feats, loss)
        sys.stdout.flush()
        os.write(real_stdout, (json.dumps(out) + "\n").encode())
    if world > 1:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()

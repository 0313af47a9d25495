#!/usr/bin/env python
"""Headline benchmark: pose-windows/sec through one full SC-VAE optimizer step on MI355X.

    python bench.py --gpus N --steps K --warmup W
    (N>1: either under `python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...`, one rank per
     process, or as typed above: outside a torchrun environment the process becomes a launcher that starts exactly that command
     as a child without ever touching the GPU itself -- launch_ranks())

Headline workload = BASELINE.json configs[2] (the N=1 point of the configs[3] data-parallel series): the FULL SC-VAE --
conditional decoder + two gradient-reversal ensembles + the adversarial net -- on synthetic 64-frame, 23-joint mouse
skeletons, default residual-CNN channels [64,128,256,512,1024], z=32, 4096 windows per GPU, AdamW.  configs[1] (1024
windows per GPU, recon + KL only) is measured in the same run at N=1 and reported as the `secondary` object of the same
JSON line (`--workload config1` makes it the headline instead).

A step = forward + all configured losses + backward + grad-norm / clip + optimizer step, inputs already resident in HBM.
Arithmetic: fp32 storage and accumulation everywhere; the large contractions run on the matrix cores with every fp32 operand
split into 16-bit pieces (default `--precision f16x3b3`: forward = two FP16 pieces per operand, i.e. 22 of its 24 significand
bits, and their 3 cross products -- outputs, losses and the ELBO match the fp32 CPU oracle to ~1e-7, checked in every run by
`elbo_match` --; data- and weight-gradient contractions = two BF16 pieces / 3 products; `--precision bf16x6b3` runs the forward
with three bf16 pieces / 6 products instead).
Weak scaling: the per-GPU batch is fixed as N grows; every loss is normalised by the global batch and gradients are summed
over ranks with RCCL.  BatchNorm batch statistics are per rank by default (the semantics of torch DistributedDataParallel
around the reference model); `--sync-bn` all-reduces them so that N ranks reproduce the 1-rank step at the global batch.

Prints ONE JSON line (rank 0) with the contract fields plus
  roofline     -- the dominant GEMM kernel template: algorithmic FLOPs per launch over its mean launch time, measured live
                  with HIP events in a second timed region (side streams serialised), against the dense bf16 MFMA peak
                  divided by the matrix-core products per algorithmic multiply; `traffic` = HBM bytes per launch from the
                  committed rocprofv3 PMC passes of this same command (profiles/);
  cpu_baseline -- the CPU oracle (oracle/scvae_oracle.py, stock PyTorch-CPU ops = the reference's own arithmetic) timed on
                  ALL host cores this process may use, on a bounded sample of the same workload at the SAME batch; its
                  first step is also the reference of the metric's ELBO-match condition (`elbo_match`: the HIP path with
                  the oracle's weights, batch, noise and shuffle, every loss term within 1e-4 relative);
  secondary    -- configs[1] measured the same way (N=1 only).
"""
import argparse
import json
import os
import re
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_F32_MFMA_TFLOPS = 157.3  # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32 dense peak
PEAK_BF16_MFMA_TFLOPS = 2500.0  # MI355X_MICROARCH.md: dense bf16 MFMA peak (v_mfma_f32_32x32x16_bf16)
PRODUCTS = {"f32": 1, "bf16x6": 6, "bf16x6w3": 6, "bf16x6b3": 6, "bf16x3": 3, "bf16": 1, "f16x3b3": 3}  # matrix-core products per algorithmic multiply (fwd / dgrad)
CHANNELS = [64, 128, 256, 512, 1024]
WIDE6 = [64, 128, 256, 512, 1024, 2048, 4096]
ARENA = [[-1.0, -1.0, -1.0], [1.0, 1.0, 1.0]]
WORKLOADS = {"config2": dict(full=True, batch=4096), "config1": dict(full=False, batch=1024)}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", default="config2", choices=list(WORKLOADS),
                    help="headline workload: config2 = BASELINE configs[2], full SC-VAE heads, 4096 windows/GPU (default); "
                         "config1 = configs[1], recon+KL, 1024 windows/GPU")
    ap.add_argument("--batch", type=int, default=None, help="windows per GPU (default: the workload's)")
    ap.add_argument("--full", action="store_true", help="force the full head set (conditional + grad-reversal + adversarial) on")
    ap.add_argument("--no-heads", action="store_true", help="force recon + KL only")
    ap.add_argument("--joints", type=int, default=23)
    ap.add_argument("--window", type=int, default=64)
    ap.add_argument("--channels", default=",".join(map(str, CHANNELS)),
                    help="model.channel, comma separated; 'wide6' = configs[4]'s six blocks 64..4096 (use with --window 256)")
    ap.add_argument("--sync-bn", action="store_true",
                    help="N>1: all-reduce the BatchNorm batch statistics (one fused buffer per BatchNorm and direction) so that N "
                         "ranks reproduce the 1-rank step at the global batch exactly; default = per-rank statistics")
    ap.add_argument("--graph", action="store_true", help="replay the whole step as one hipGraph (single GPU)")
    ap.add_argument("--precision", default=os.environ.get("SVAE_PRECISION", "f16x3b3"), choices=list(PRODUCTS),
                    help="arithmetic of the large contractions: f32 = fp32 MFMA; bf16x6 = fp32-accurate 3-piece split on the bf16 "
                         "matrix cores (6 products); bf16x6w3 = the same with 2 pieces / 3 products for the weight-gradient "
                         "contractions; bf16x6b3 = 3 products for the whole backward pass, 6 for the forward; f16x3b3 = forward with two "
                         "FP16 pieces / 3 products (22-bit operands, ~2^-22 per product), backward as bf16x6b3; bf16x3 / bf16 = "
                         "2 / 1 pieces everywhere (reduced accuracy, study only)")
    ap.add_argument("--overlap-streams", action="store_true", help="side streams on at any batch (default: from 32768 batch x window rows)")
    ap.add_argument("--serial-streams", action="store_true",
                    help="timed region without the concurrent side streams (what the roofline region always uses)")
    ap.add_argument("--h2d", action="store_true",
                    help="copy the batch from pinned host memory inside every timed step (the PCIe-inclusive rate DESIGN.md quotes; "
                         "never the headline value: the metric is defined with inputs resident in HBM)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-secondary", action="store_true", help="skip the configs[1] measurement reported as `secondary`")
    ap.add_argument("--timer-kinds", default="fwd,dgrad,wgrad", help="GEMM kinds bracketed with HIP events (fwd,dgrad,wgrad)")
    args = ap.parse_args()
    args.channel_list = WIDE6 if args.channels == "wide6" else [int(c) for c in args.channels.split(",")]
    wl = WORKLOADS[args.workload]
    if args.batch is None:
        args.batch = wl["batch"]
    args.full = (wl["full"] or args.full) and not args.no_heads
    return args


def make_cfg(full):
    method, feats, loss = {}, [], {"jpe": 1.0, "root": 1.0, "prior": 1.0}
    if full:
        method = {"conditional": ["avg_speed_3d", "heading"], "grad_reversal": ["avg_speed_3d", "heading"],
                  "adversarial_net": ["heading"]}
        feats = ["avg_speed_3d", "heading"]
        loss.update({"avg_speed_3d_gr": 1.0, "heading_gr": 1.0, "heading_an": 1.0})
    return method, feats, loss


def build_model(args, full, method, feats, tree):
    from scrubvae_amd.get import model as get_model
    mc = dict(type="rcnn", kernel=5, z_dim=32, window=args.window, activation="prelu", diag=True,
              init_dilation=None, prior="gaussian", channel=args.channel_list)
    dis = dict(method=method, alpha=1.0, features=feats)
    torch.manual_seed(0)
    m = get_model(mc, None, None, dis, args.joints, "midfwd", arena_size=torch.tensor(ARENA), kinematic_tree=tree,
                  discrete_classes={"ids": torch.arange(4)} if full else None, device="cuda", verbose=0)
    return m, dis


def workload_name(args, full, B):
    wide = args.channel_list == WIDE6
    base = ("configs[4] wide six-block rcnn (channels to 4096, long window), full SC-VAE heads (conditional + grad_reversal x2 + "
            "adversarial_net), AdamW" if (wide and full) else
            "configs[2] full SC-VAE (conditional + grad_reversal x2 + adversarial_net), AdamW" if full else
            "configs[4] wide six-block rcnn, recon+KL (jpe+root+prior), AdamW" if wide else
            "configs[1] mouse-skeleton rcnn, recon+KL (jpe+root+prior), AdamW")
    return base + f", batch {B}/GPU, window {args.window}, {args.joints} joints, z=32, channels [{','.join(map(str, args.channel_list))}]"


def profile_tag(args, full, B):
    """Name of the committed PMC summaries of this workload (profiles/r*_pmc_traffic_<tag>.json), None for ad-hoc shapes."""
    if args.channel_list == WIDE6 and args.window == 256 and args.joints == 23 and full and B == 1024:
        return "config4_w256_wide6_b1024"
    if args.channel_list != CHANNELS or args.window != 64 or args.joints != 23:
        return None
    if full and B == 4096:
        return "config2_b4096"
    if not full and B == 1024:
        return "config1_b1024"
    return None


def usable_cpus():
    """CPUs this process can actually run on at once: its affinity mask, capped by the cgroup CPU quota (the GPU boxes expose
    all 256 host cores in the mask but give the job a 16-CPU quota: more runnable threads than that only throttle)."""
    n = len(os.sched_getaffinity(0))
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = min(n, max(1, -(-int(quota) // int(period))))
    except (OSError, ValueError):
        pass
    return max(1, n)


def cpu_baseline(args, full, B, max_seconds=24.0):
    """cpu_baseline leg: the CPU oracle's train_step (the only place bench.py touches oracle/) at the workload's own batch,
    on every host core this process may use.  The first (warm-up) step is the ELBO reference for `elbo_match`."""
    from oracle import scvae_oracle as O
    method, feats, loss = make_cfg(full)

    def cost(ch, w):
        return sum(a * b * (w >> (i + 1)) for i, (a, b) in enumerate(zip(ch, ch[1:])))
    work = cost(args.channel_list, args.window) / cost(CHANNELS, 64)
    sample_b = B if work <= 1.5 else max(4, int(min(B, 128 / work)) // 4 * 4)  # wide / long models: a smaller sample
    cfg = O.OracleConfig(n_keypts=args.joints, window=args.window, z_dim=32, kernel=5, diag=True, channel=tuple(args.channel_list),
                         arena_size=torch.tensor(ARENA), kinematic_tree=O.skeleton_tree(args.joints), method=method,
                         features=feats, discrete_classes={"ids": torch.arange(4)} if full else None)
    threads = usable_cpus()  # every core the box gives this process
    torch.set_num_threads(threads)
    sd = O.init_state_dict(cfg, seed=0)
    data = O.synth_batch(cfg, sample_b, seed=0)
    eps = torch.randn(sample_b, cfg.z_dim)
    perm = {k: torch.randperm(sample_b) for k in cfg.method.get("adversarial_net", [])}
    state = {}
    t0 = time.perf_counter()
    bl0, _, _, _ = O.train_step(sd, cfg, data, loss, eps, adv_perm=perm, opt_state=state)  # warm-up; also the ELBO reference
    t_first = time.perf_counter() - t0
    elbo = elbo_check(args, full, cfg, sd, data, eps, perm, loss, method, feats, bl0)
    steps = int(max(3, min(50, max_seconds / max(t_first, 1e-3))))
    t0 = time.perf_counter()
    for _ in range(steps):
        _, _, sd, _ = O.train_step(sd, cfg, data, loss, eps, adv_perm=perm, opt_state=state)
    dt = (time.perf_counter() - t0) / steps
    return {"value": round(sample_b / dt, 2), "unit": "windows/s", "cores": torch.get_num_threads(), "kind": "port",
            "host_cores": os.cpu_count(), "cpu_quota": usable_cpus(), "threads": torch.get_num_threads(),
            "sample": f"{steps} optimizer steps of the CPU oracle at batch {sample_b} (same model / loss config as the timed GPU "
                      f"workload), {dt*1e3:.0f} ms/step; torch intra-op threads = every CPU this job may use (affinity mask capped by "
                      f"its cgroup CPU quota; the host has {os.cpu_count()} cores)",
            "elbo_match": elbo}


def elbo_check(args, full, cfg, sd, data, eps, perm, loss, method, feats, oracle_losses):
    """The "ELBO-match" condition of the metric on the benchmark's own model, batch, precision and tile table: the oracle's
    weights, batch, noise and shuffle on the HIP path; relative deviation of every loss term (bound: 1e-4)."""
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


def pmc_traffic(kernel, tag):
    """HBM bytes per launch of `kernel` from the committed rocprofv3 PMC passes of this workload (separate --pmc FETCH_SIZE /
    WRITE_SIZE runs of this same command, gfx950 corrections applied by tools/summarize_pmc.py); None when no summary
    for this workload / kernel is committed."""
    import glob
    if tag is None:
        return None
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", f"r*_{tag}_pmc_traffic.json")))
    if not files:
        return None
    try:
        k = json.load(open(files[-1]))["kernels"].get(kernel)
        return None if not k or not k.get("hbm_bytes_per_launch") else {
            "hbm_bytes_per_launch": round(k["hbm_bytes_per_launch"]), "source": os.path.basename(files[-1])}
    except Exception:
        return None


def run_workload(args, full, B, rank, world, roofline=True):
    """Times `args.steps` optimizer steps of one workload (after `args.warmup` untimed ones); returns the result fields."""
    import gc
    gc.collect()  # the previous workload's model (20 GB of buffers at batch 4096) goes back to the allocator before this one builds
    torch.cuda.empty_cache()
    from scrubvae_amd import parallel, ops
    from scrubvae_amd.data import synthetic
    from scrubvae_amd.train.losses import get_batch_loss
    from scrubvae_amd.train.trainer import FusedAdam, clip_grad_norm_, on_compute_stream
    method, feats, loss = make_cfg(full)
    data, tree = synthetic.make_batch(args.joints, args.window, B, seed=100 + rank, device="cuda")
    data0 = data
    model, dis = build_model(args, full, method, feats, tree)
    parallel.attach(model, sync_bn=args.sync_bn)
    model.defer_tail = True  # one fused tail launch per step (outputs + losses + seed gradients)
    opt = FusedAdam(model, lr=1e-4, weight_decay=0.01, decoupled=True)
    model.train()

    graphed = None
    if args.graph:
        from scrubvae_amd.train.trainer import GraphedStep
        graphed = GraphedStep(model, opt, loss, dis, data)
        roofline = False  # per-launch events cannot be recorded inside a replay

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
    torch.cuda.synchronize()
    hp = on_compute_stream("cuda")
    hp.__enter__()
    for _ in range(args.warmup):
        step()
    if world > 1:
        model.time_comm = True  # HIP events around the exposed tail of the gradient exchange (ResVAE.backward_from_seeds)
        model.pop_comm_times()
    if args.serial_streams:
        model.overlap_wgrad = False
    elif args.overlap_streams:
        model.overlap_wgrad = True
    # ---- headline timed region: exactly K steps, barrier + synchronize on both sides, no per-launch instrumentation
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        bl = step()
    t_host = time.perf_counter() - t0  # all launches of the K steps are queued (the GPU is still working)
    barrier()
    dt = time.perf_counter() - t0
    ov_timed = 0 if args.serial_streams else int(model._ov)  # the stream schedule the K timed steps actually ran (the regions below change it)
    comm_ms = model.pop_comm_times() if world > 1 else []
    # ---- roofline region: the same K steps again with the side streams serialised and HIP events around every launch of
    # the dominant GEMM template.  In the headline region three HIP streams run kernels concurrently on shared CUs, so a
    # launch's start-to-end time is not the kernel's own time there; serialised, it is.
    timer = probe = None
    dt_serial = None
    if roofline and graphed is None:
        keep = model.overlap_wgrad
        model.overlap_wgrad = False
        model.time_comm = False
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
    res = {"value": round(B * world * args.steps / dt, 1), "ms_per_step": round(dt / args.steps * 1e3, 3),
           "host_enqueue_ms_per_step": round(t_host / args.steps * 1e3, 3), "final_total_loss": float(bl["total"].detach()),
           "side_streams": ov_timed,
           "workload": workload_name(args, full, B), "batch_per_gpu": B}
    if comm_ms:
        res["exposed_comm_ms_per_step"] = {k: round(sum(c[k] for c in comm_ms) / len(comm_ms), 3) for k in ("grads", "bn")}
    elif world > 1:
        res["exposed_comm_ms_per_step"] = None
    if timer is not None:
        summ = timer.summary()
        if summ:
            kname, s = max(summ.items(), key=lambda kv: kv[1]["ms"])  # the GEMM template instance with the largest total time
            tf = s["flops"] / (s["ms"] * 1e-3) / 1e12
            split_kernel = "bf16s" in kname
            k_products = template_products(kname, args.precision)  # matrix-core products per multiply of THIS template
            peak = PEAK_BF16_MFMA_TFLOPS / k_products if split_kernel else PEAK_F32_MFMA_TFLOPS
            res["roofline"] = {
                "measured": f"second timed region of {args.steps} steps with the side HIP streams serialised "
                            f"({dt_serial / args.steps * 1e3:.3f} ms/step; the headline region overlaps kernels on 3 streams, "
                            "where a launch's duration is not the kernel's own time); profiles/ holds the rocprofv3 summary of "
                            "`bench.py --serial-streams` for the same workload",
                "bound": "mfma", "achieved": round(tf, 2), "peak": round(peak, 1), "unit": "TFLOP/s", "frac": round(tf / peak, 4),
                "traffic": pmc_traffic("svae::" + kname, profile_tag(args, full, B)),
                "peak_note": (f"dense bf16 MFMA peak {PEAK_BF16_MFMA_TFLOPS:.0f} TFLOP/s / {k_products} matrix-core products per "
                              "algorithmic multiply (achieved counts algorithmic FLOPs)"
                              if split_kernel else "dense fp32 MFMA peak (v_mfma_f32_32x32x2_f32)"),
                "kernel": "svae::" + kname, "launches_per_step": s["launches"] // args.steps,
                "avg_launch_us": round(s["ms"] * 1e3 / s["launches"], 2),
                "avg_launch_gflop": round(s["flops"] / s["launches"] / 1e9, 3),
                "probe_step_all_gemm_templates": {
                    k: {"TFLOP/s": round(v["flops"] / (v["ms"] * 1e-3) / 1e12, 2),
                        "avg_us": round(v["ms"] * 1e3 / v["launches"], 2), "launches_per_step": v["launches"]}
                    for k, v in sorted(probe.summary().items())}}
    del model, opt, graphed
    torch.cuda.empty_cache()
    return res


def template_products(kname, precision):
    """Matrix-core products per algorithmic multiply of a split-kernel template instance: 6 / 3 / 1 for 3 / 2 / 1 pieces per operand.
    The piece count is the third template argument of the gather / per-tap weight-gradient kernels; the 12-wave halo kernels and the
    all-taps weight-gradient kernels are built for two pieces only (their third argument is something else)."""
    if not ("bf16s" in kname):
        return 1
    if kname.startswith(("wgrad_taps", "gather_halo_ws4")):
        return 3
    pm = re.match(r"\w+<\d+, \d+, (\d),", kname)
    return {3: 6, 2: 3, 1: 1}.get(int(pm.group(1)), PRODUCTS[precision]) if pm else PRODUCTS[precision]


def precision_text(p):
    if p == "f32":
        return "fp32 MFMA (v_mfma_f32_32x32x2_f32)"
    if p == "f16x3b3":
        return ("f16x3b3: fp32 storage and accumulation; forward contractions split every fp32 operand into TWO FP16 pieces (22 of its 24 "
                "significand bits) and run 3 cross products on v_mfma_f32_32x32x16_f16 (~2^-22 per product: outputs, losses and the ELBO "
                "match the fp32 CPU oracle to ~1e-6, the check is `elbo_match`); data- and weight-gradient contractions: 2 bf16 pieces / 3 "
                "products on v_mfma_f32_32x32x16_bf16 (whole-step gradients checked against the CPU oracle at the benchmark's size, "
                "tests/test_gpu_fullsize.py)")
    tail = {"bf16x6": " (fp32-accurate, DESIGN.md 4)",
            "bf16x6w3": " forward and data-gradient (fp32-accurate), 3 (2 pieces) for the weight-gradient contractions "
                        "(gradient error vs fp64 unchanged, DESIGN.md 4)",
            "bf16x6b3": " forward (outputs, losses, ELBO: fp32-accurate), 3 (2 pieces) for the data- and weight-gradient "
                        "contractions (whole-step gradients checked against the CPU oracle at the benchmark's size, "
                        "tests/test_gpu_fullsize.py)"}.get(p, " (reduced accuracy)")
    return (f"{p}: fp32 storage and accumulation; every large contraction splits its fp32 operands into bf16 pieces and runs "
            f"{PRODUCTS[p]} cross product(s) on v_mfma_f32_32x32x16_bf16" + tail)


def launch_ranks(args):
    """`python bench.py --gpus N` (N > 1) started WITHOUT a torchrun environment: this process only launches -- it starts
    `python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py <same arguments>` as a child (one rank process per
    GPU, RCCL), relays rank 0's JSON line to stdout and exits with the child's status.  It makes no GPU call itself (importing
    torch does not initialise the device) and never replaces its own process image."""
    import socket
    import subprocess
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC: what RCCL needs on this image
    env.setdefault("OMP_NUM_THREADS", str(max(1, usable_cpus() // args.gpus)))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    p = subprocess.Popen(cmd, stdout=subprocess.PIPE, env=env, text=True)
    line = None
    for ln in p.stdout:  # rank 0 prints exactly one JSON line; anything else a library wrote to stdout goes to stderr
        t = ln.strip()
        if t.startswith("{") and '"metric"' in t:
            line = t
        elif t:
            sys.stderr.write(ln)
    rc = p.wait()
    if line is not None:
        sys.stdout.write(line + "\n")
        sys.stdout.flush()
    elif rc == 0:
        rc = 1
    raise SystemExit(rc)


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        launch_ranks(args)
    # stdout carries exactly ONE line (the JSON): whatever libraries print while the job runs (RCCL's version banner,
    # gloo's connection messages, ...) is sent to stderr instead -- file descriptor 1 is pointed at stderr until the end
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)
    from scrubvae_amd import parallel, ops
    rank, local, world = parallel.init_distributed()  # reads the torchrun environment; no GPU call before this point
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}")
    torch.cuda.set_device(local % max(torch.cuda.device_count(), 1))
    ops.set_precision(args.precision)
    B = args.batch
    backend, ranks_seen = "none", 1
    if world > 1:
        backend = torch.distributed.get_backend()
        ones = torch.ones(1, device="cuda" if backend == "nccl" else "cpu")
        torch.distributed.all_reduce(ones)  # every rank contributes 1: the number of ranks the collective really spans
        ranks_seen = int(ones.item())
    head = run_workload(args, args.full, B, rank, world, roofline=not args.no_roofline)

    if rank == 0:
        out = {
            "metric": f"pose-windows/sec (ELBO-match) on synthetic {args.window}-frame mouse skeletons",
            "value": head["value"], "unit": "windows/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": head["ms_per_step"], "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32" if args.precision == "f32" else f"f32 ({args.precision} split on the {'fp16 / bf16' if 'f16x' in args.precision else 'bf16'} matrix cores)",
            "data": "synthetic",
            "config": {"workload": head["workload"], "batch_per_gpu": B, "global_batch": B * world, "window": args.window,
                       "joints": args.joints, "launch": "hipGraph replay" if args.graph else "eager launches",
                       "inputs": ("copied from pinned host memory for every step, one batch ahead on a copy stream (PCIe-inclusive)"
                                  if args.h2d else "resident in HBM"),
                       "streams": {2: "3 HIP streams (weight gradients / skip branches / the tail overlap the main chain)",
                                   1: "2 HIP streams (weight gradients beside the main chain: chosen from 8192 batch x window rows)",
                                   0: "one stream (serialised: chosen below 8192 batch x window rows, or --serial-streams)"}[int(head["side_streams"])],
                       "precision": precision_text(args.precision),
                       "side_streams": int(head["side_streams"]),
                       "host_enqueue_ms_per_step": head["host_enqueue_ms_per_step"],
                       "parallelism": f"dp{world}" + ("" if world == 1 else ("+syncbn" if args.sync_bn else "+localbn")),
                       "backend": {"nccl": "nccl (RCCL)"}.get(backend, backend), "ranks_seen": ranks_seen,
                       "bn": "single rank" if world == 1 else ("sync" if args.sync_bn else "local"),
                       "final_total_loss": head["final_total_loss"]},
        }
        if world > 1:
            # time the main stream of rank 0 waited for collectives, per step (HIP events; the rest of the exchange ran under compute)
            out["config"]["exposed_comm_ms_per_step"] = head["exposed_comm_ms_per_step"]
        if "roofline" in head:
            out["roofline"] = head["roofline"]
        # every GPU measurement runs BEFORE any CPU baseline: the oracle's 16 intra-op threads keep spinning for a while after their
        # last parallel region, and the batch-1024 workload -- whose host enqueue time is close to its GPU time -- measured 25 %
        # slower right behind them
        sec = None
        if world == 1 and not args.no_secondary and args.workload == "config2" and args.batch == WORKLOADS["config2"]["batch"] \
                and args.channel_list == CHANNELS and args.window == 64 and not (args.graph or args.h2d or args.serial_streams):
            w1 = WORKLOADS["config1"]
            sec = run_workload(args, w1["full"], w1["batch"], rank, world, roofline=not args.no_roofline)
            sec = {"metric": out["metric"], "unit": "windows/s", **sec}
        if not args.no_cpu_baseline and world == 1:  # reported at N=1 only
            out["cpu_baseline"] = cpu_baseline(args, args.full, B)
        if sec is not None:
            if not args.no_cpu_baseline:
                sec["cpu_baseline"] = cpu_baseline(args, w1["full"], w1["batch"], max_seconds=8.0)
            out["secondary"] = sec
        sys.stdout.flush()
        os.write(real_stdout, (json.dumps(out) + "\n").encode())
    if world > 1:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()

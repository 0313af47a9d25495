"""Data-parallel parity with the real HIP kernels: 2 ranks (gloo, both on the one GPU of the
test box) at B/2 windows each must reproduce the 1-rank step at B windows: losses, sync-BN
running statistics and the summed gradients."""
import os
import socket

import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu

from oracle import scvae_oracle as O


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


ARENA = torch.tensor([[-1.0, -1.0, -1.0], [1.0, 1.0, 1.0]])
METHODS = {"conditional": ["avg_speed_3d", "heading"], "grad_reversal": ["avg_speed_3d", "heading"], "adversarial_net": ["heading"]}
LS = {"jpe": 1.0, "root": 1.0, "prior": 0.5, "avg_speed_3d_gr": 1.0, "heading_gr": 2.0, "heading_an": 0.5}
PERM = torch.randperm(16, generator=torch.Generator().manual_seed(9))  # the adversarial shuffle: ONE permutation of the global batch


def _setup(adv=True):
    methods = dict(METHODS) if adv else {k: v for k, v in METHODS.items() if k != "adversarial_net"}
    cfg = O.OracleConfig(n_keypts=18, window=64, z_dim=8, kernel=5, channel=(16, 16, 16, 32, 32), diag=True, arena_size=ARENA,
                         method=methods, features=["avg_speed_3d", "heading"], discrete_classes={"ids": torch.arange(4)})
    sd = O.init_state_dict(cfg, seed=4)
    data = O.synth_batch(cfg, 16, seed=4)
    eps = torch.randn(16, 8, generator=torch.Generator().manual_seed(1))
    return cfg, sd, data, eps


def _run(model, dis, data, eps, ls=LS):
    from scrubvae_amd.train.losses import get_batch_loss
    model.train()
    d = {k: v.cuda() for k, v in data.items()}
    d["eps"] = eps.cuda()
    data_o = model(d)
    bl = get_batch_loss(model, d, data_o, ls, dis, adv_perm={"heading": PERM} if "heading_an" in ls else None)
    bl["total"].backward()
    torch.cuda.synchronize()
    return {k: float(v.detach()) for k, v in bl.items()}, model.flat_grads.detach().cpu().clone(), \
        {k: v.cpu() for k, v in model.state_dict().items() if "running" in k}


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0")
    import torch.distributed as dist
    from scrubvae_amd import parallel
    from tests.test_gpu_model import build_model
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        cfg, sd, data, eps = _setup()
        model, dis = build_model(cfg, sd)
        parallel.attach(model, sync_bn=True, broadcast=False)
        model.bucket_min_bytes = 0  # every encoder block gets its own gradient bucket (the default merges small ones)
        lo, hi = parallel.shard_range(16, rank, world)
        shard = {k: v[lo:hi] for k, v in data.items()}
        losses, grads, stats = _run(model, dis, shard, eps[lo:hi])
        # per-rank losses are normalised by the global batch: their sum is the global loss
        lt = torch.tensor([losses[k] for k in sorted(losses)], dtype=torch.float64)
        dist.all_reduce(lt)
        if rank == 0:
            # numpy (pickled by value): torch tensors would be shared through fds of a process that exits
            q.put((dict(zip(sorted(losses), lt.tolist())), grads.numpy(), {k: v.numpy() for k, v in stats.items()}))
        dist.barrier()
    finally:
        dist.destroy_process_group()


def test_two_ranks_match_one_rank():
    from tests.test_gpu_model import build_model
    cfg, sd, data, eps = _setup()
    model, dis = build_model(cfg, sd)
    ref_losses, ref_grads, ref_stats = _run(model, dis, data, eps)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    losses, grads, stats = q.get(timeout=300)
    grads = torch.from_numpy(grads)
    stats = {k: torch.from_numpy(v) for k, v in stats.items()}
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for k, v in ref_losses.items():
        assert abs(losses[k] - v) <= 1e-5 * abs(v) + 1e-9, (k, losses[k], v)
    for k, v in ref_stats.items():
        assert torch.allclose(stats[k], v, rtol=1e-4, atol=1e-6), k
    gmax = float(ref_grads.abs().max())
    # same data, same kernels, different split of the reductions: fp32 noise only
    assert float((grads - ref_grads).abs().max()) < 2e-3 * gmax
    rel = float((grads - ref_grads).norm() / ref_grads.norm())
    assert rel < 1e-3, rel


def _rccl_worker(port, q):
    """One rank on the RCCL backend: the collectives are identities, but every call of the data-parallel schedule --
    sync-BN all-reduces, the gradient buckets fed from the communication stream, the final waits -- goes through
    torch.distributed's RCCL process group exactly as on N GPUs."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
    import torch.distributed as dist
    from scrubvae_amd import parallel
    from tests.test_gpu_model import build_model
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1)
    try:
        cfg, sd, data, eps = _setup(adv=False)  # (the faked world size has no second rank to gather the shuffled column from)
        out = {}
        for mode in ("plain", "ddp"):
            model, dis = build_model(cfg, sd)
            if mode == "ddp":
                parallel.attach(model, sync_bn=False, broadcast=True)
                model.world_size = 2      # take the N>1 code path: every loss is then normalised by 2B
                model.bucket_min_bytes = 0
            losses, grads, stats = _run(model, dis, data, eps, ls={"jpe": 1.0, "root": 1.0, "prior": 0.5, "avg_speed_3d_gr": 0.0, "heading_gr": 0.0})
            out[mode] = (losses, grads.numpy())
        # the all-gather of the adversarial shuffle's column on the RCCL backend (rank order, here one rank)
        src = torch.arange(7, dtype=torch.float32, device="cuda") * 0.5
        out["gather"] = model._allgather(src).cpu().numpy()
        q.put(out)
    except Exception as e:  # noqa: BLE001 -- reported to the parent instead of a queue timeout
        import traceback
        q.put({"error": traceback.format_exc()})
        raise
    finally:
        dist.destroy_process_group()


def test_rccl_schedule_single_rank():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_rccl_worker, args=(_free_port(), q))
    p.start()
    out = q.get(timeout=240)
    p.join(timeout=60)
    assert "error" not in out, out.get("error")
    assert p.exitcode == 0
    import numpy as np
    assert np.array_equal(out["gather"], np.arange(7, dtype=np.float32) * 0.5)
    (l0, g0), (l1, g1) = out["plain"], out["ddp"]
    g0, g1 = torch.from_numpy(g0), torch.from_numpy(g1)
    # the faked world size halves every loss term and hence (local BatchNorm statistics) every gradient
    for k, v in l0.items():
        if not k.endswith(("_gr", "_an")):  # recursive normalisation (losses.py:279-284): not linear in 1/B; scale 0 here
            assert abs(2 * l1[k] - v) <= 1e-5 * abs(v), (k, l1[k], v)
    assert float((2 * g1 - g0).abs().max()) <= 1e-4 * float(g0.abs().max())

"""Data-parallel plumbing: one process per GPU, torch.distributed over RCCL (backend "nccl" on
ROCm) on the xGMI mesh; gloo on CPU for tests.  The reference has no distributed code at all
 -- this is new, and parity is defined as "N ranks at global batch B give
the 1-rank result at batch B".

Protocol (what ``ResVAE`` does when ``world_size > 1``):
  * rank r holds windows [r*B/N, (r+1)*B/N) of the global batch (``shard_range``);
  * every loss term is normalised by the GLOBAL batch, so per-rank gradients simply SUM;
  * train-mode BatchNorm: per-layer [2,C] (sum, sum-of-squares) partials are all-reduced
    before the finalize kernel (forward) and the [2,C] (sum du, sum du*xhat) partials before
    the apply kernel (backward); parameter gradients use the LOCAL sums so that the gradient
    all-reduce does not double count;
  * one all-reduce(sum) of the flat gradient buffer per step, issued in buckets as soon as
    the reverse schedule has finished a segment (decoder first), overlapping the encoder
    backward; xGMI is point-to-point, so a few large buckets beat many small ones.
"""
from __future__ import annotations

import os

import torch
import torch.distributed as dist


def shard_range(n_items, rank, world_size):
    """Contiguous shard [lo, hi) of n_items for `rank` (sizes differ by at most one)."""
    base, rem = divmod(n_items, world_size)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def init_distributed(backend=None):
    """Initialise from the torchrun environment (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_*).
    Returns (rank, local_rank, world_size)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        if backend is None:  # SVAE_DIST_BACKEND=gloo: rehearse N ranks on one GPU (tests only)
            backend = os.environ.get("SVAE_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        if backend == "nccl":
            torch.cuda.set_device(local)
        elif torch.cuda.is_available():
            torch.cuda.set_device(local % max(torch.cuda.device_count(), 1))
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, local, world


def attach(model, process_group=None, sync_bn=True, broadcast=True):
    """Make `model` data-parallel over the default (or given) process group."""
    if not dist.is_initialized():
        model.world_size, model.rank = 1, 0
        return model
    model.world_size = dist.get_world_size(process_group)
    model.rank = dist.get_rank(process_group)
    model.process_group = process_group
    model.sync_bn = sync_bn
    for mods in getattr(model, "disentangle", {}).values():  # streaming scrubbers sum their batch statistics over the ranks
        for m in mods.values():
            if hasattr(m, "process_group"):
                m.process_group = process_group
    if model.world_size > 1:
        # seed of the shared adversarial-shuffle permutation stream (ResVAE.global_permutation): rank 0's torch seed, so that a
        # run's shuffles follow the user's torch.manual_seed like the single-rank torch.randperm does
        seed = torch.tensor([torch.initial_seed() & 0x7FFFFFFFFFFF], dtype=torch.int64)
        if dist.get_backend(process_group) == "nccl":
            seed = seed.to(model.device)
        dist.broadcast(seed, src=0, group=process_group)
        model.shuffle_seed = int(seed.item())
    if broadcast and model.world_size > 1:
        dist.broadcast(model.flat_params, src=0, group=process_group)
        for b in model.buffers():
            if b is not None and b.is_floating_point():
                dist.broadcast(b, src=0, group=process_group)
    return model


def allreduce_sum_(t, group=None):
    if dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
    return t


def bn_sync_stats(local_sums, local_count, group=None):
    """Sync-BN forward statistics: all-reduce the [2,C] (sum, sumsq) partials; returns
    (global_sums, global_count).  Host-side mirror of what ResVAE._bn_act does."""
    sums = allreduce_sum_(local_sums.clone(), group)
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    return sums, local_count * world

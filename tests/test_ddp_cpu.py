"""N>1 path on CPU: world_size-2 gloo runs of the data-parallel protocol
(scrubvae_amd/parallel.py) with the CPU oracle as the compute stand-in.

Checks (a) shard_range partitions exactly, (b) the sync-BN statistic exchange reproduces the
global batch statistics, (c) "every loss normalised by the GLOBAL batch + SUM all-reduce of
gradients" reproduces the single-process gradients of the global batch.
"""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import scvae_oracle as O
from scrubvae_amd import parallel


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_shard_range_partitions():
    for n in (1, 7, 8, 1024, 1025):
        for w in (1, 2, 3, 8):
            spans = [parallel.shard_range(n, r, w) for r in range(w)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(spans[i][1] == spans[i + 1][0] for i in range(w - 1))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    torch.set_num_threads(2)
    parallel.init_distributed(backend="gloo")
    try:
        # (b) sync-BN statistics
        g = torch.Generator().manual_seed(0)
        x = torch.randn(64, 10, generator=g, dtype=torch.float64) * 3 + 1
        lo, hi = parallel.shard_range(64, rank, world)
        xs = x[lo:hi]
        sums, count = parallel.bn_sync_stats(torch.stack([xs.sum(0), (xs * xs).sum(0)]), hi - lo)
        mean = sums[0] / count
        var = sums[1] / count - mean ** 2
        ok_bn = torch.allclose(mean, x.mean(0)) and torch.allclose(var, x.var(0, unbiased=False))
        # (c) gradient protocol with the oracle (eval-mode BN: no cross-sample coupling left
        # except through the normalisation by the global batch)
        arena = torch.tensor([[-1.0, -1.0, -1.0], [1.0, 1.0, 1.0]], dtype=torch.float64)
        cfg = O.OracleConfig(n_keypts=18, window=64, z_dim=8, kernel=5, channel=(8, 8, 16, 16, 32), diag=True, arena_size=arena)
        sd = O.init_state_dict(cfg, seed=2, dtype=torch.float64)
        data = O.synth_batch(cfg, 8, seed=2, dtype=torch.float64)
        ls = {"jpe": 1.0, "root": 1.0, "prior": 0.5}
        names = O.trainable_names(sd)

        def grads_of(batch, global_b):
            leaf = {n: sd[n].clone().requires_grad_(True) for n in names}
            work = dict(sd); work.update(leaf)
            out = O.forward(work, cfg, batch, False)
            bl = O.batch_loss(work, cfg, batch, out, ls)
            local_b = batch["x6d"].shape[0]
            (bl["total"] * local_b / global_b).backward()  # normalise by the GLOBAL batch
            return torch.cat([leaf[n].grad.flatten() if leaf[n].grad is not None else torch.zeros_like(leaf[n]).flatten() for n in names])

        shard = {k: v[slice(*parallel.shard_range(8, rank, world))] for k, v in data.items()}
        gflat = parallel.allreduce_sum_(grads_of(shard, 8))
        ok_grad = True
        if rank == 0:
            ref = grads_of(data, 8)
            ok_grad = bool(torch.allclose(gflat, ref, rtol=1e-9, atol=1e-12))
        if rank == 0:
            q.put((ok_bn, ok_grad))
        dist.barrier()
    finally:
        dist.destroy_process_group()


def test_gloo_world2_protocol():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    ok_bn, ok_grad = q.get(timeout=240)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert ok_bn and ok_grad


def test_capi_exports_every_declared_symbol():
    """The C-ABI library loads on a GPU-less host and exports every symbol include/*.h declares."""
    import re
    from scrubvae_amd import _lib
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    header = open(os.path.join(root, "include", "scrubvae_hip.h")).read()
    declared = set(re.findall(r"\b(svae_[a-z0-9_]+)\s*\(", header))
    declared -= {"svae_status"}
    lib = _lib.lib()
    for name in sorted(declared):
        assert hasattr(lib, name), f"{name} declared in the header but not exported"
        assert name in _lib.SIGNATURES, f"{name} has no ctypes signature in scrubvae_amd/_lib.py"
    assert lib.svae_version() >= 100


def test_product_never_imports_oracle():
    """oracle/ is test infrastructure: nothing under scrubvae_amd/ may import it."""
    root = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "scrubvae_amd")
    for dp, _, files in os.walk(root):
        for f in files:
            if f.endswith(".py"):
                src = open(os.path.join(dp, f)).read()
                assert "oracle" not in src.replace("scvae_oracle", "oracle").split("import")[-1] or "from oracle" not in src, f
                assert "from oracle" not in src and "import oracle" not in src, f


def _mals_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    torch.set_num_threads(2)
    parallel.init_distributed(backend="gloo")
    try:
        from scrubvae_amd.model.disentangle import MovingAvgLeastSquares
        g = torch.Generator().manual_seed(1)
        x, y = torch.randn(16, 8, generator=g), torch.randn(16, 3, generator=g)
        lo, hi = parallel.shard_range(16, rank, world)
        m = MovingAvgLeastSquares(8, 3, bias=True)
        for _ in range(3):
            y0, y1 = m(x[lo:hi])
            m.evaluate_loss(y0, y1, y[lo:hi])
            m.update(x[lo:hi], y[lo:hi])
        q.put((rank, {k: v.numpy() for k, v in m.state_dict().items()}))
    finally:
        dist.destroy_process_group()


def test_gloo_world2_streaming_scrubber_matches_one_rank():
    """MovingAvgLeastSquares sums its batch statistics (x^T x, x^T y and the two squared errors that steer the
    forgetting factors) over the ranks: 2 ranks with half batches end with the buffers of 1 rank with the batch."""
    from scrubvae_amd.model.disentangle import MovingAvgLeastSquares
    g = torch.Generator().manual_seed(1)
    x, y = torch.randn(16, 8, generator=g), torch.randn(16, 3, generator=g)
    ref = MovingAvgLeastSquares(8, 3, bias=True)
    for _ in range(3):
        y0, y1 = ref(x)
        ref.evaluate_loss(y0, y1, y)
        ref.update(x, y)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_mals_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for r in range(2):
        for k, v in ref.state_dict().items():
            assert torch.allclose(torch.from_numpy(res[r][k]), v, rtol=1e-5, atol=1e-6), (r, k)


def _filter_worker(rank, world, port, q, which):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    torch.set_num_threads(2)
    parallel.init_distributed(backend="gloo")
    try:
        from scrubvae_amd.model import disentangle as D
        g = torch.Generator().manual_seed(2)
        x = torch.randn(32, 6, generator=g)
        y = (torch.arange(32) % 4).reshape(-1, 1)
        lo, hi = parallel.shard_range(32, rank, world)
        m = getattr(D, which)(6, torch.arange(4))
        tot = 0.0
        for _ in range(3):
            xl = x[lo:hi].clone().requires_grad_(True)
            val = m.evaluate_loss(xl, y[lo:hi])
            tot = float(val.detach())
            m.update(x[lo:hi], y[lo:hi])
        q.put((rank, tot, {k: v.numpy() for k, v in m.state_dict().items()}))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("which", ["MovingAverageFilter", "QuadraticDiscriminantFilter"])
def test_gloo_world2_class_filters_match_one_rank(which):
    """Class sums / counts (moving average) and member / non-member moments (QDA) are summed over the ranks: 2 ranks with
    half batches end with the buffers of 1 rank with the batch; the QDA loss is a sum over samples, so the rank values add up."""
    from scrubvae_amd.model import disentangle as D
    g = torch.Generator().manual_seed(2)
    x = torch.randn(32, 6, generator=g)
    y = (torch.arange(32) % 4).reshape(-1, 1)
    ref = getattr(D, which)(6, torch.arange(4))
    for _ in range(3):
        want = float(ref.evaluate_loss(x.clone().requires_grad_(True), y).detach())
        ref.update(x, y)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_filter_worker, args=(r, 2, port, q, which)) for r in range(2)]
    for p in procs:
        p.start()
    res = {r: (t, sd) for r, t, sd in (q.get(timeout=120) for _ in range(2))}
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for r in range(2):
        for k, v in ref.state_dict().items():
            assert torch.allclose(torch.from_numpy(res[r][1][k]), v, rtol=2e-4, atol=1e-5), (r, k)
    got = res[0][0] if which == "MovingAverageFilter" else res[0][0] + res[1][0]
    assert abs(got - want) <= 2e-4 * abs(want) + 1e-5


def _mi_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    torch.set_num_threads(2)
    parallel.init_distributed(backend="gloo")
    try:
        from types import SimpleNamespace
        from scrubvae_amd.train.trainer import make_mi_estimator
        g = torch.Generator().manual_seed(4)
        mu, var, x, y = (torch.randn(32, 6, generator=g), torch.randn(32, 3, generator=g), torch.randn(32, 6, generator=g),
                         torch.randn(32, 3, generator=g))
        L = torch.diag_embed(torch.rand(32, 6, generator=g) + 0.3)
        lo, hi = parallel.shard_range(32, rank, world)
        cfg = {"disentangle": {"bandwidth": 0.6, "var_mode": "diagonal"}}
        est = make_mi_estimator(SimpleNamespace(process_group=None), cfg, mu[lo:hi], var[lo:hi], L[lo:hi])
        # losses.get_batch_loss divides the rank-local mean by the world size; the rank values then add up
        q.put((rank, float(est(x[lo:hi], y[lo:hi])) / world, est.num_s))
    finally:
        dist.destroy_process_group()


def test_gloo_world2_mcmi_estimator_matches_one_rank():
    """The KDE centres of every rank are gathered, so 2 ranks with half batches evaluate the estimator of 1 rank with the
    batch; rank-local means / world add up to the global mean."""
    from scrubvae_amd.model.disentangle import MutInfoEstimator
    g = torch.Generator().manual_seed(4)
    mu, var, x, y = (torch.randn(32, 6, generator=g), torch.randn(32, 3, generator=g), torch.randn(32, 6, generator=g),
                     torch.randn(32, 3, generator=g))
    L = torch.diag_embed(torch.rand(32, 6, generator=g) + 0.3)
    want = float(MutInfoEstimator(mu, var, 0.6, var_mode="diagonal", model_var=L)(x, y))
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_mi_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(2)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(r[2] == 32 for r in res)
    assert abs(sum(r[1] for r in res) - want) < 1e-5 * abs(want)

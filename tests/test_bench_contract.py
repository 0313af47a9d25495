"""bench.py's output contract (one JSON line on stdout with the fields the driver reads), on a small batch."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REQUIRED = {"metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
            "dtype", "data", "config", "roofline", "cpu_baseline"}


def test_bench_defaults_match_baseline_config():
    sys.path.insert(0, ROOT)
    import bench
    old = sys.argv
    sys.argv = ["bench.py"]
    try:
        a = bench.parse()
    finally:
        sys.argv = old
    assert (a.gpus, a.batch, a.window, a.joints) == (1, 4096, 64, 23)      # BASELINE configs[2]: full SC-VAE, 4096 windows / GPU
    assert a.channel_list == [64, 128, 256, 512, 1024] and a.full and not a.h2d and not a.graph and not a.no_secondary
    sys.argv = ["bench.py", "--workload", "config1"]
    try:
        a = bench.parse()
    finally:
        sys.argv = old
    assert (a.batch, a.full) == (1024, False)                              # BASELINE configs[1]
    assert a.precision in bench.PRODUCTS


@pytest.mark.gpu
def test_bench_prints_exactly_one_json_line_with_the_contract_fields():
    env = dict(os.environ, PYTHONPATH=ROOT)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "3", "--warmup", "2", "--batch", "64"],
                       capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert REQUIRED <= set(d), REQUIRED - set(d)
    assert d["n_gpus"] == 1 and d["steps"] == 3 and d["warmup"] == 2 and d["higher_is_better"] is True and d["scaling"] == "weak"
    assert d["unit"] == "windows/s" and d["vs_baseline"] is None and d["data"] == "synthetic"
    assert abs(d["value"] - 64 * 3 / (d["ms_per_step"] * 3e-3)) <= 0.01 * d["value"]
    assert "workload" in d["config"] and "model" not in d["config"]
    rf = d["roofline"]
    assert rf["bound"] in ("hbm", "mfma") and rf["unit"] in ("GB/s", "TFLOP/s") and abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-3
    cb = d["cpu_baseline"]
    assert cb["kind"] in ("port", "reference") and cb["cores"] >= 1 and cb["value"] > 0 and "sample" in cb
    assert cb["elbo_match"]["ok"] is True and cb["elbo_match"]["batch"] == 64   # checked at the timed batch
    assert cb["threads"] == cb["cores"] and cb["host_cores"] >= cb["threads"]
    assert "configs[2]" in d["config"]["workload"] and "secondary" not in d       # secondary only accompanies the default batch
    c = d["config"]
    assert c["ranks_seen"] == 1 and c["backend"] == "none" and c["bn"] == "single rank"
    assert c["side_streams"] == 0 and c["streams"].startswith("one stream")       # 64 x 64 rows: the serial schedule was timed


@pytest.mark.gpu
def test_bench_two_ranks_full_config_sync_bn():
    """bench.py's own N > 1 path -- the launch line the driver uses (torch.distributed.run, one process per rank), the FULL
    configs[2] head set, --sync-bn, the shared-seed global adversarial shuffle, bucketed gradient all-reduce -- rehearsed with 2
    gloo ranks on the one GPU of the test box (SVAE_DIST_BACKEND=gloo; on an 8-GPU node the same code runs over RCCL)."""
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ, PYTHONPATH=ROOT, SVAE_DIST_BACKEND="gloo")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
                        "--batch", "64", "--sync-bn", "--no-roofline"],
                       capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.strip().startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["config"]["global_batch"] == 128 and d["config"]["parallelism"] == "dp2+syncbn"
    assert d["config"]["ranks_seen"] == 2 and d["config"]["bn"] == "sync" and d["config"]["exposed_comm_ms_per_step"]["bn"] > 0
    assert "configs[2]" in d["config"]["workload"] and "cpu_baseline" not in d and "secondary" not in d
    assert abs(d["value"] - 128 * 2 / (d["ms_per_step"] * 2e-3)) <= 0.01 * d["value"]
    import math
    assert math.isfinite(d["config"]["final_total_loss"])


def test_bench_plain_command_becomes_a_launcher_without_touching_the_gpu(monkeypatch, capsys):
    """`python bench.py --gpus N` outside a torchrun environment starts N rank processes through torch.distributed.run as a CHILD
    (never exec), relays rank 0's JSON line and returns the child's status -- checked with the child process faked."""
    sys.path.insert(0, ROOT)
    import subprocess as sp

    import bench
    seen = {}

    class FakePopen:
        def __init__(self, cmd, stdout=None, env=None, text=None):
            seen["cmd"], seen["env"] = cmd, env
            self.stdout = iter(["NCCL version banner\n", '{"metric": "m", "value": 1.0, "n_gpus": 4}\n'])

        def wait(self):
            return 0

    monkeypatch.setattr(sp, "Popen", FakePopen)
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4", "--steps", "2", "--warmup", "1"])
    import torch
    monkeypatch.setattr(torch.cuda, "is_available", lambda: (_ for _ in ()).throw(AssertionError("the launcher made a GPU call")))
    with pytest.raises(SystemExit) as e:
        bench.main()
    assert e.value.code == 0
    cmd = seen["cmd"]
    assert cmd[1:4] == ["-m", "torch.distributed.run", "--nnodes=1"] and cmd[cmd.index("--nproc-per-node") + 1] == "4"
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and cmd[-6:] == ["--gpus", "4", "--steps", "2", "--warmup", "1"]
    assert seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"
    out = capsys.readouterr()
    assert out.out.strip() == '{"metric": "m", "value": 1.0, "n_gpus": 4}' and "NCCL version banner" in out.err


@pytest.mark.gpu
def test_bench_plain_command_line_two_ranks():
    """Exactly what the driver types -- `python bench.py --gpus 2 ...`, no torchrun -- on the one GPU of the test box (two gloo
    ranks): the bench launches its own ranks and the line records what the collective really spanned."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(PYTHONPATH=ROOT, SVAE_DIST_BACKEND="gloo")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--batch", "64", "--steps", "2", "--warmup", "1"],
                       capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    c = d["config"]
    assert d["n_gpus"] == 2 and c["ranks_seen"] == 2 and c["backend"] == "gloo" and c["bn"] == "local"
    assert c["global_batch"] == 128 and c["parallelism"] == "dp2+localbn"
    ex = c["exposed_comm_ms_per_step"]
    assert ex["grads"] > 0 and ex["bn"] == 0      # per-rank BatchNorm statistics: no BatchNorm collective
    assert "roofline" in d and "cpu_baseline" not in d
    assert c["streams"].startswith("one stream") == (c["side_streams"] == 0)


def test_bench_products_per_template():
    """roofline.peak divides the dense matrix-core peak by the products per multiply of the dominant TEMPLATE: every kernel family's
    name must parse (the all-taps weight-gradient and 12-wave halo templates carry no piece count in their arguments)."""
    sys.path.insert(0, ROOT)
    import bench
    f = bench.template_products
    assert f("gather_halo_bf16s_kernel<128, 128, 2, 4, 2, 160, false>", "f16x3b3") == 3
    assert f("gather_halo_bf16s_kernel<256, 128, 3, 4, 2, 320, false>", "bf16x6") == 6
    assert f("gather_gemm_bf16s_ws_kernel<128, 128, 2, 2, 2, 2, 0, false>", "f16x3b3") == 3
    assert f("gather_halo_ws_bf16s_kernel<256, 128, 2, 4, 2, 264, 0, false, 3, false>", "f16x3b3") == 3
    assert f("gather_halo_ws4_bf16s_kernel<256, 128, 264, 88, true, false, 0>", "f16x3b3") == 3
    assert f("gather_halo_ws4m_bf16s_kernel<256, 128, 320, 64, true>", "f16x3b3") == 3
    assert f("wgrad_taps16_bf16s_kernel<128, 128, 6, 1, false>", "f16x3b3") == 3
    assert f("wgrad_taps_bf16s_kernel<64, 128, 5, 2, false>", "bf16x6b3") == 3
    assert f("wgrad_gemm_bf16s_kernel<256, 256, 2, 1, 2, 4>", "f16x3b3") == 3
    assert f("wgrad_gemm_bf16s_kernel<128, 128, 3, 2, 2, 2>", "bf16x6") == 6
    assert f("gather_gemm_kernel<64, 64, false>", "f32") == 1

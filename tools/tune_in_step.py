"""In-step refinement of the GEMM tile table: coordinate descent on the WHOLE training step's time.

tools/tune_tiles.py times every candidate kernel alone, on warm caches and idle-boost clocks.  The step is power-limited
(DESIGN.md 4b): a kernel that is 20 % faster alone can be no faster inside the step, and the ranking of candidates changes.
This tool starts from the shipped table, and for every (layer, pass) of the benchmark workload tries a short list of
alternative tile codes, timing STEPS of the real training loop (serial streams, so that the order of launches is fixed)
and keeps a change only if it wins twice by more than the noise margin.

    python tools/tune_in_step.py [--workload config2|config1] [--steps 8] [--margin 0.004]
writes gpurun_out/tuned_tiles_in_step.json (the full table with the refined entries) and prints every accepted change."""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from scrubvae_amd import ops, _lib

ap = argparse.ArgumentParser()
ap.add_argument("--workload", default="config2")
ap.add_argument("--steps", type=int, default=8)
ap.add_argument("--margin", type=float, default=0.004)
ap.add_argument("--kinds", default="fwd,dgrad,wgrad")
ap.add_argument("--min-gflop", type=float, default=15.0, help="layers below this many GFLOP per pass keep their entry")
a = ap.parse_args()

full, B = bench.WORKLOADS[a.workload]["full"], bench.WORKLOADS[a.workload]["batch"]
args = argparse.Namespace(window=64, joints=23, channel_list=bench.CHANNELS, sync_bn=False)
from scrubvae_amd.data import synthetic
from scrubvae_amd.train.losses import get_batch_loss
from scrubvae_amd.train.trainer import FusedAdam, clip_grad_norm_

ops.set_precision("f16x3b3")
method, feats, loss = bench.make_cfg(full)
data, tree = synthetic.make_batch(23, 64, B, seed=0, device="cuda")
model, dis = bench.build_model(args, full, method, feats, tree)
model.defer_tail = True
model.overlap_wgrad = False
opt = FusedAdam(model, lr=1e-4, weight_decay=0.01, decoupled=True)
model.train()


def step():
    data_o = model(data)
    bl = get_batch_loss(model, data, data_o, loss, dis)
    bl["total"].backward()
    clip_grad_norm_(model, 1e6)
    opt.step()


def measure(n=None):
    n = n or a.steps
    step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        step()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


for _ in range(6):
    step()
base = min(measure(), measure())
print(f"start: {base:.3f} ms/step ({a.workload}, B={B}, serial streams)", flush=True)

GATHER = [8128128, 8128064, 9128128, 9128064, 11128128, 13128128, 16128128, 16128064, 17128128, 17128064, 29128128, 3128128, 2128128, 4128128, 6128128]
WGRAD = [256256, 1256256, 2256256, 3256256, 256128, 128256, 128128, 2128128, 3128128, 4128128, 6128128, 12128128, 14128128,
         4064128, 6064128, 4128064, 6128064, 64128, 2064128, 16064128, 18064128, 18128128]
table = dict(ops.TILE_TABLE)
changes = []
convs = sorted(((k, cv) for k, cv in model._convs.items() if cv.flops >= a.min_gflop * 1e9), key=lambda kv: -kv[1].flops)
for key, cv in convs:
    for kind in a.kinds.split(","):
        base_p = cv._base_pieces(kind)
        if not base_p:
            continue
        d = cv.desc
        tkey = cv.tile_key(kind)
        cur_code, cur_p = int(cv.desc.tile[ops._KIND_ID[kind]]), cv._kind_pieces(kind)
        if not cur_p:  # this pass runs the fp32 kernel by the tuner's choice: leave it
            continue
        best_code, best_t = cur_code, min(measure(), measure())
        tried = []
        for code in (WGRAD if kind == "wgrad" else GATHER):
            if code == cur_code:
                continue
            cv._set_choice(kind, cur_p, code)
            try:
                t = measure()
            except _lib.SvaeError as e:
                if e.status in (_lib.ERR_SHAPE, _lib.ERR_WORKSPACE):
                    continue
                raise
            except RuntimeError as e:  # workspace smaller than this tile needs etc.
                continue
            tried.append((code, t))
            if t < best_t * (1 - a.margin):
                t2 = measure()  # confirm
                if t2 < best_t * (1 - a.margin):
                    best_code, best_t = code, min(t, t2)
        cv._set_choice(kind, cur_p, best_code)
        if best_code != cur_code:
            table[tkey] = best_code
            changes.append((tkey, cur_code, best_code, best_t))
            print(f"  {tkey}: {cur_code} -> {best_code}  ({best_t:.3f} ms/step)", flush=True)
        else:
            print(f"  {tkey}: keeps {cur_code} ({best_t:.3f}; best alternative {min(tried, key=lambda x: x[1]) if tried else None})", flush=True)

final = min(measure(12), measure(12))
print(f"end: {final:.3f} ms/step (start {base:.3f}); {len(changes)} entries changed", flush=True)
out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out", "tuned_tiles_in_step.json")
os.makedirs(os.path.dirname(out), exist_ok=True)
json.dump(table, open(out, "w"), indent=0, sort_keys=True)
print("wrote", out)

"""Measure the GEMM tile choices for the benchmark geometries on this GPU and write
scrubvae_amd/tuned_tiles.json (careful timing: 12 repetitions per candidate)."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from scrubvae_amd import ops
from scrubvae_amd.data import synthetic
from scrubvae_amd.get import model as get_model
from scrubvae_amd.train.losses import get_batch_loss

KEEP = "--keep" in sys.argv
table = dict(ops.TILE_TABLE) if KEEP else {}
# --retune fwd@3,dgrad@2: with --keep, drop the entries of these (kind@pieces) prefixes so that they are measured again (new kernel
# variants joined the candidate list)
for a in list(sys.argv):
    if a.startswith("--retune="):
        pre = tuple(x + ":" for x in a.split("=", 1)[1].split(","))
        table = {k: v for k, v in table.items() if not k.startswith(pre)}
        sys.argv.remove(a)
CONFIG4 = "--config4" in sys.argv  # BASELINE configs[4]: window 256, six blocks 64..4096 (bench.py --window 256 --channels wide6)
for a in list(sys.argv):  # --retune-match=fwd,:64:144:64:144: -- with --keep, drop the entries of that kind whose key contains the substring
    if a.startswith("--retune-match="):
        kind, sub = a.split("=", 1)[1].split(",", 1)
        table = {k: v for k, v in table.items() if not (k.startswith(kind) and sub in k)}
        sys.argv.remove(a)
sys.argv = [a for a in sys.argv if a not in ("--keep", "--config4")]
WIDE6 = [64, 128, 256, 512, 1024, 2048, 4096]
DEFAULT = [64, 128, 256, 512, 1024]
ops.TILE_TABLE = dict(table) if KEEP else {}  # --keep: only what is missing (or dropped by --retune) is measured
ops.AUTOTUNE_REPS = 12
PRECISIONS = sys.argv[1:] or ["f32", "bf16x6", "bf16x6w3", "bf16x6b3"]
CASES = (((23, 1024, True, 256, WIDE6), (23, 32, False, 256, WIDE6), (23, 32, True, 256, WIDE6), (23, 1024, False, 256, WIDE6)) if CONFIG4 else
         tuple(c + (64, DEFAULT) for c in ((23, 4096, True), (23, 1024, False), (23, 1024, True), (23, 4096, False), (18, 1024, False), (23, 256, False))))
for prec, joints, batch, full, window, channels in [(p, *c) for p in PRECISIONS for c in CASES]:
    ops.set_precision(prec)
    data, tree = synthetic.make_batch(joints, window, batch, seed=0, device="cuda")
    method = {"conditional": ["avg_speed_3d", "heading"], "grad_reversal": ["avg_speed_3d", "heading"], "adversarial_net": ["heading"]} if full else {}
    feats = ["avg_speed_3d", "heading"] if full else []
    loss = {"jpe": 1.0, "root": 1.0, "prior": 1.0}
    if full:
        loss.update({"avg_speed_3d_gr": 1.0, "heading_gr": 1.0, "heading_an": 1.0})
    mc = dict(type="rcnn", kernel=5, z_dim=32, window=window, activation="prelu", diag=True, init_dilation=None, prior="gaussian",
              channel=channels)
    dis = dict(method=method, alpha=1.0, features=feats)
    m = get_model(mc, None, None, dis, joints, "midfwd", arena_size=torch.tensor([[-1.0, -1, -1], [1, 1, 1]]), kinematic_tree=tree,
                  discrete_classes={"ids": torch.arange(4)} if full else None, device="cuda", verbose=0)
    m.train()
    m.overlap_wgrad = False
    for _ in range(2):
        bl = get_batch_loss(m, data, m(data), loss, dis)
        bl["total"].backward()
    torch.cuda.synchronize()
    table.update(ops.TUNED_LOG)
    ops.TILE_TABLE = dict(table)  # geometries tuned by an earlier precision in this run are not re-timed
    print(f"{prec} J={joints} B={batch} full={full} W={window} top={channels[-1]}: {len(ops.TUNED_LOG)} geometries tuned so far", flush=True)
    json.dump(table, open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out", "tuned_tiles_partial.json"), "w"),
              indent=0, sort_keys=True)  # a long tuning run that hits its time limit still leaves what it measured
    del m
    torch.cuda.empty_cache()
out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out", "tuned_tiles.json")
json.dump(table, open(out, "w"), indent=0, sort_keys=True)
print("wrote", out, len(table))

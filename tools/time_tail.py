"""Time the fused pose tail (forward + backward) at the benchmark's shape; SVAE_TAIL_ROWS selects the frames per workgroup."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from scrubvae_amd import ops
from scrubvae_amd.data import synthetic
from scrubvae_amd._lib import make_tree
B, W, J = int(os.environ.get("B", 4096)), 64, 23
data, tree = synthetic.make_batch(J, W, B, seed=0, device="cuda")
rows, ld = B * W, ops.pad16(6 * J + 3)
y = torch.randn(rows, ld, device="cuda")
x6d = torch.empty(B, W, J, 6, device="cuda"); root_hat = torch.empty(B, W, 3, device="cuda")
lp = torch.empty(ops.tail_blocks(rows), 2, device="cuda"); dy = torch.empty(rows, ld, device="cuda")
t = make_tree(J, tree)
arena = [-1.0, -1, -1, 1, 1, 1]
def run():
    ops.pose_tail(y, ld, data["offsets"], data["target_pose"], data["root"], arena, t, 1e-3, 1e-3, None, None, x6d, root_hat, lp, dy, rows, pre_tanh=True)
for _ in range(3): run()
torch.cuda.synchronize()
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
s.record()
for _ in range(20): run()
e.record(); torch.cuda.synchronize()
us = s.elapsed_time(e) / 20 * 1e3
byts = 2 * rows * (6 * J + 3 * J + 3 * J + 3 + 3) * 4
print(f"SVAE_TAIL_KERNEL={os.environ.get('SVAE_TAIL_KERNEL', 'lane')} SVAE_TAIL_ROWS={os.environ.get('SVAE_TAIL_ROWS', 'default')} B={B}: {us:.1f} us, {byts / us / 1e6:.2f} TB/s algorithmic ({byts / us / 1e6 / 8:.3f} of 8 TB/s)")

"""Preprocessing row on the GPU: windows/s and achieved HBM bandwidth of the per-frame kernel, next to
the CPU oracle (the reference's numpy/torch arithmetic) on a bounded sample."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from scrubvae_amd.data import preprocess as PP
from scrubvae_amd.data import synthetic
from oracle import preprocess_oracle as P
from oracle import scvae_oracle as O

N, W, J = int(sys.argv[1]) if len(sys.argv) > 1 else 4096, 64, 18
tree = O.skeleton_tree(J)
offset = [[float(c) for c in r] for r in O.skeleton_offsets(J)]
g = torch.Generator().manual_seed(0)
x6d = torch.randn(N * W, J, 6, generator=g).cuda()
seg = (0.5 + torch.rand(J, generator=g))[:, None] * torch.tensor(offset)
pose = synthetic.fwd_kin_cont6d(x6d, tree, seg[None].expand(N * W, J, 3).contiguous().cuda()).reshape(N, W, J, 3)
pose = pose + torch.cumsum(0.05 * torch.randn(N, W, 1, 3, generator=g).cuda(), dim=1)

def timeit(fn, reps=20):
    fn(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / reps * 1e-3

t_ik = timeit(lambda: PP.inv_kin_windows(pose, tree, offset, "midfwd"))
t_sp = timeit(lambda: PP.get_speed_parts(pose))
x6, offs, root, head = PP.inv_kin_windows(pose, tree, offset, "midfwd")
t_fk = timeit(lambda: synthetic.fwd_kin_cont6d(x6, tree, offs))
frames = N * W
bytes_ik = frames * (3 * J + 6 * J + 3 * J + 3) * 4 + N * 8
print(f"N={N} windows x {W} frames x {J} joints")
print(f"inv_kin kernel   {t_ik*1e6:8.1f} us  {frames/t_ik/1e6:8.1f} Mframes/s  {bytes_ik/t_ik/1e9:7.1f} GB/s algorithmic ({bytes_ik/t_ik/8e12*100:.1f}% of 8 TB/s)")
print(f"speed features   {t_sp*1e6:8.1f} us  ({frames*3*J*4/t_sp/1e9:.1f} GB/s)")
print(f"target-pose FK   {t_fk*1e6:8.1f} us  (pose-tail kernel, incl. its padded-input staging)")
tot = t_ik + t_sp + t_fk
print(f"total            {tot*1e6:8.1f} us  = {N/tot/1e3:.1f} k windows/s")
n_cpu = 64
pc = pose[:n_cpu].double().cpu().numpy()
t0 = time.perf_counter()
P.preprocess_windows(pc, tree, offset, ["x6d", "root", "offsets", "target_pose", "avg_speed_3d", "heading"], "midfwd", fwd_kin=O.fwd_kin)
dt = time.perf_counter() - t0
print(f"CPU oracle       {n_cpu} windows in {dt:.2f} s = {n_cpu/dt:.0f} windows/s ({torch.get_num_threads()} threads)")

"""Per-tensor gradient errors (vs the fp64 oracle) of the whole-step tests' cases, PReLU slopes first:
python tools/fullsize_slopes.py config4|config1|config2 [precision]   (SVAE_FUSE_UPSAMPLE=0/1 etc. from the environment)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests import test_gpu_fullsize as T

case = sys.argv[1] if len(sys.argv) > 1 else "config4"
prec = sys.argv[2] if len(sys.argv) > 2 else "f16x3b3"
rep = []
if case == "config4":
    T._run(32, False, prec, seed=61, window=256, channel=T.WIDE6, expect=(), out_tol=5e-5, report=rep)
elif case == "config1":
    T._run(1024, False, prec, seed=41, expect=(), report=rep)
else:
    T._run(4096, True, prec, seed=51, expect=(), report=rep)
rep.sort(key=lambda r: -r[2])
print("largest errors:")
for n, numel, e_hip, e_cpu, gm, gmax in rep[:10]:
    print(f"{n:48s} numel {numel:9d}  hip {e_hip:.2e}  fp32 oracle {e_cpu:.2e}  max|g| {gm:.3e}  (global {gmax:.3e})")
print("single scalars (PReLU slopes):")
for n, numel, e_hip, e_cpu, gm, gmax in sorted((r for r in rep if r[1] == 1), key=lambda r: r[0]):
    print(f"{n:48s} g {gm:11.4e}  abs err hip {e_hip * (gm + 1e-3 * gmax):.2e}  oracle {e_cpu * (gm + 1e-3 * gmax):.2e}  rel hip {e_hip:.2e}  oracle {e_cpu:.2e}")

"""Board power and shader clock while (a) idle, (b) one dense GEMM kernel is launched back to back for seconds, (c) the training
step of the benchmark workload runs, (d) a memory-bound kernel runs -- the evidence behind DESIGN.md 4b "the step is power-limited".

Sampling: the amdgpu hwmon files (power1_average / power1_input in microwatts, freq1_input in Hz) every 20 ms from a thread; where they
are not readable, `rocm-smi --showpower --showclocks` every ~0.5 s.  For (b) the per-launch time is recorded over the whole run, so
that the first launches (what a 20-launch micro-benchmark sees) can be compared with the steady state.

    python tools/power_trace.py [--seconds 4] > gpurun_out/power_trace.txt"""
import argparse
import glob
import os
import re
import subprocess
import sys
import threading
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from scrubvae_amd import ops

ap = argparse.ArgumentParser()
ap.add_argument("--seconds", type=float, default=4.0)
ap.add_argument("--workload", default="config2")
a = ap.parse_args()


def _hwmon():
    """(power file, clock file) of every card whose hwmon files are readable: the box shows all GPUs of its host, the process
    owns one of them -- the trace keeps the card whose power moves the most."""
    cards = []
    for d in sorted(glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*")):
        p = [f for f in ("power1_average", "power1_input") if os.access(os.path.join(d, f), os.R_OK)]
        if p:
            f = os.path.join(d, "freq1_input")
            cards.append((os.path.join(d, p[0]), f if os.access(f, os.R_OK) else None))
    return cards


class Sampler(threading.Thread):
    def __init__(self):
        super().__init__(daemon=True)
        self.cards = _hwmon()
        self.pw = bool(self.cards)
        self.all, self.rows, self.stop = [], [], False
        self.source = f"hwmon ({len(self.cards)} cards visible)" if self.pw else "rocm-smi"

    def read(self):
        if self.pw:
            vals = []
            for pw, fq in self.cards:
                vals.append((int(open(pw).read()) / 1e6, int(open(fq).read()) / 1e6 if fq else float("nan")))
            self.all.append((time.perf_counter(), vals))
            return vals[0]
        out = subprocess.run(["rocm-smi", "--showpower", "--showclocks"], capture_output=True, text=True).stdout
        w = re.search(r"Power \(W\):\s*([0-9.]+)", out)
        f = re.search(r"sclk clock level:\s*\d+:?\s*\(?([0-9.]+)Mhz", out, re.I)
        return (float(w.group(1)) if w else float("nan")), (float(f.group(1)) if f else float("nan"))

    def run(self):
        while not self.stop:
            try:
                self.rows.append((time.perf_counter(),) + self.read())
            except Exception as e:  # a sample that cannot be read is skipped, the phase table shows the count
                pass
            time.sleep(0.02 if self.pw else 0.05)


PHASES = []


def phase_stats(s, t0, t1):
    r = [x for x in s.rows if t0 + 0.3 * (t1 - t0) <= x[0] <= t1]  # the last 70 % of the phase: past the controller's transient
    if not r:
        return "no samples"
    w = [x[1] for x in r]
    f = [x[2] for x in r]
    return f"{sum(w) / len(w):7.0f} W (max {max(w):5.0f})   sclk {sum(f) / len(f):6.0f} MHz (min {min(f):5.0f})   {len(r)} samples"


def main():
    s = Sampler()
    s.start()
    print(f"sampling via {s.source}: {s.pw or 'rocm-smi'}")
    torch.cuda.synchronize()
    t0 = time.perf_counter(); time.sleep(1.5); t1 = time.perf_counter()
    PHASES.append(("idle", t0, t1, ""))

    # (b) one dense kernel, back to back
    ops.set_precision("f16x3b3")
    B = 4096
    cv = ops.Conv(B, 4, 512, 1024, 5, 1, 2, 1, False, pieces=3)
    cv.__dict__["_tuned"] = {"fwd", "dgrad", "wgrad"}
    cv._set_choice("fwd", 22, 17128128)
    x = torch.randn(B * 4, cv.c_in_p, device="cuda")
    w = torch.randn(*cv.weight_shape, device="cuda") * 0.05
    y = torch.empty(B * cv.l_out, cv.c_out_p, device="cuda")
    ops.bump_weight_epoch()
    cv.fwd(x, w, None, y)
    torch.cuda.synchronize()
    time.sleep(1.0)  # back to idle clocks
    per = []
    t0 = time.perf_counter()
    group = 20
    while time.perf_counter() - t0 < a.seconds:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(group):
            cv.fwd(x, w, None, y)
        e1.record()
        e1.synchronize()
        per.append((time.perf_counter() - t0, e0.elapsed_time(e1) / group * 1e3))
    t1 = time.perf_counter()
    tf = lambda us: cv.flops / us / 1e6
    PHASES.append(("dense GEMM alone (enc3.c3 fwd, ws4)", t0, t1,
                   f"\n      per-launch time, groups of {group}: first group {per[0][1]:.1f} us ({tf(per[0][1]):.0f} TFLOP/s), "
          f"groups at 0.1 s {min(per, key=lambda p: abs(p[0] - 0.1))[1]:.1f} us, 0.5 s {min(per, key=lambda p: abs(p[0] - 0.5))[1]:.1f} us, "
          f"last {per[-1][1]:.1f} us ({tf(per[-1][1]):.0f} TFLOP/s); min {min(p[1] for p in per):.1f} max {max(p[1] for p in per):.1f}"))

    # (d) a memory-bound kernel alone
    n = 1 << 28
    u = torch.empty(n, device="cuda")
    time.sleep(1.0)
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < a.seconds / 2:
        for _ in range(20):
            u.add_(1.0)
        torch.cuda.synchronize()
    t1 = time.perf_counter()
    PHASES.append(("memory-bound kernel alone (2 GiB r+w)", t0, t1, ""))
    del u

    # (c) the training step
    import argparse as _ap
    from scrubvae_amd.data import synthetic
    from scrubvae_amd.train.losses import get_batch_loss
    from scrubvae_amd.train.trainer import FusedAdam, clip_grad_norm_
    full, Bw = bench.WORKLOADS[a.workload]["full"], bench.WORKLOADS[a.workload]["batch"]
    args = _ap.Namespace(window=64, joints=23, channel_list=bench.CHANNELS, sync_bn=False)
    method, feats, loss = bench.make_cfg(full)
    data, tree = synthetic.make_batch(23, 64, Bw, seed=0, device="cuda")
    model, dis = bench.build_model(args, full, method, feats, tree)
    model.defer_tail = True
    opt = FusedAdam(model, lr=1e-4, weight_decay=0.01, decoupled=True)
    model.train()

    def step():
        data_o = model(data)
        bl = get_batch_loss(model, data, data_o, loss, dis)
        bl["total"].backward()
        clip_grad_norm_(model, 1e6)
        opt.step()

    for _ in range(5):
        step()
    torch.cuda.synchronize()
    time.sleep(1.0)
    t0 = time.perf_counter()
    n_steps = 0
    while time.perf_counter() - t0 < a.seconds * 1.5:
        step(); n_steps += 1
        if n_steps % 8 == 0:
            torch.cuda.synchronize()
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    PHASES.append((f"training step ({a.workload}, B={Bw})", t0, t1, f"   {(t1 - t0) / n_steps * 1e3:.2f} ms/step"))
    s.stop = True
    s.join()
    own = 0
    if s.pw and len(s.cards) > 1:  # keep the card whose power moved the most: the one this process drives
        rng = [max(v[1][i][0] for v in s.all) - min(v[1][i][0] for v in s.all) for i in range(len(s.cards))]
        own = rng.index(max(rng))
        s.rows = [(t, v[own][0], v[own][1]) for t, v in s.all]
        print(f"card driven by this process: {s.cards[own][0]} (power range per card: {[round(r) for r in rng]} W)")
    else:
        s.rows = [(t, v[0][0], v[0][1]) for t, v in s.all] if s.pw else s.rows
    if s.pw:
        capf = os.path.join(os.path.dirname(s.cards[own if len(s.cards) > 1 else 0][0]), "power1_cap")
        if os.access(capf, os.R_OK):
            print(f"board power cap (power1_cap): {int(open(capf).read()) / 1e6:.0f} W")
    for label, t0, t1, extra in PHASES:
        print(f"{label:40s} {phase_stats(s, t0, t1)}{extra}")
    # the raw trace, decimated, for the record
    print("trace (s, W, MHz), every 10th sample:")
    base = s.rows[0][0]
    for r in s.rows[::10]:
        print(f"  {r[0] - base:7.2f} {r[1]:7.0f} {r[2]:6.0f}")


if __name__ == "__main__":
    main()

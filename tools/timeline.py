"""Reduce a rocprofv3 kernel trace (per-dispatch start / end timestamps) of bench.py to the shape of one optimizer step: how long
the GPU runs at least one kernel, how long nothing runs (launch bubbles), how much time is covered by two or more kernels, and the
largest idle gaps with the kernels around them.  python tools/timeline.py <kernel_trace.csv>"""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
ks = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "?")) for r in rows), key=lambda t: t[0])
# steps are delimited by the Adam kernel
ends = [i for i, k in enumerate(ks) if "adam_kernel" in k[2]]
if len(ends) < 8:
    print("too few steps in the trace"); sys.exit(0)
lo, hi = ends[3] + 1, ends[-2] + 1          # whole steps in the timed region
sel = ks[lo:hi]
nsteps = len([i for i in ends if lo <= i < hi])
t0, t1 = sel[0][0], max(k[1] for k in sel)
span = (t1 - t0) / nsteps
ev = []
for s, e, n, q in sel:
    ev.append((s, 1)); ev.append((e, -1))
ev.sort()
busy = over = 0; depth = 0; last = ev[0][0]
for t, d in ev:
    if depth >= 1: busy += t - last
    if depth >= 2: over += t - last
    depth += d; last = t
print(f"{nsteps} steps, {len(sel) / nsteps:.0f} launches / step, {span / 1e6:.3f} ms / step between first start and last end")
print(f"  at least one kernel running {busy / nsteps / 1e6:.3f} ms, nothing running {(t1 - t0 - busy) / nsteps / 1e6:.3f} ms, two or more kernels {over / nsteps / 1e6:.3f} ms")
print(f"  sum of kernel durations {sum(e - s for s, e, _, _ in sel) / nsteps / 1e6:.3f} ms")
# idle gaps
gaps = []
cur_end = sel[0][1]; prev = sel[0]
for k in sel[1:]:
    if k[0] > cur_end:
        gaps.append((k[0] - cur_end, prev[2][:60], k[2][:60]))
    if k[1] > cur_end:
        cur_end = k[1]; prev = k
hist = collections.Counter()
for g, a, b in gaps:
    hist["<2us" if g < 2000 else "2-5us" if g < 5000 else "5-10us" if g < 10000 else "10-20us" if g < 20000 else ">20us"] += g
print("  idle time by gap length (ms / step): " + ", ".join(f"{k} {v / nsteps / 1e6:.3f}" for k, v in sorted(hist.items())))
agg = collections.defaultdict(lambda: [0, 0])
for g, a, b in gaps:
    agg[(a, b)][0] += g; agg[(a, b)][1] += 1
print("  largest idle gaps (total us / step, count / step, kernel before -> kernel after):")
for (a, b), (g, c) in sorted(agg.items(), key=lambda kv: -kv[1][0])[:25]:
    print(f"    {g / nsteps / 1e3:7.1f} us  x{c / nsteps:4.1f}  {a.replace('svae::', '')[:50]:50s} -> {b.replace('svae::', '')[:50]}")

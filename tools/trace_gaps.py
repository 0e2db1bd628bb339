"""GPU occupancy of a training step from a kernel timeline (tools/trace_dump.py CSV, optionally .gz): inside the window of the LAST `steps`
repetitions of a marker kernel (default: the optimizer's first kernel) it prints
  * wall time per step, the union of kernel intervals (GPU busy with at least one kernel), the idle remainder and the idle-gap histogram,
  * time with exactly 1 / 2 / 3+ kernels in flight (how much the stream lanes overlap),
  * per queue: busy time and launches,
  * the kernels that ran ALONE for the longest total time (the serial part of the step) and the largest idle gaps with their neighbours.

    python tools/trace_gaps.py fp_trace.csv.gz [marker substring] [steps]"""
import collections
import csv
import gzip
import sys

path = sys.argv[1]
marker = sys.argv[2] if len(sys.argv) > 2 else "lamb_pass1"
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
op = gzip.open if path.endswith(".gz") else open
rows = []
with op(path, "rt") as f:
    for r in csv.DictReader(f):
        rows.append((r["name"], int(r["start_ns"]), int(r["end_ns"]), r["queue"]))
rows.sort(key=lambda r: r[1])
marks = [i for i, r in enumerate(rows) if marker in r[0]]
# one marker per step: collapse runs of markers closer than 1 ms
starts = []
for i in marks:
    if not starts or rows[i][1] - rows[starts[-1]][1] > 1_000_000:
        starts.append(i)
if len(starts) < steps + 1:
    print("only", len(starts), "markers of", marker); sys.exit(1)
i0, i1 = starts[-steps - 1], starts[-1]
t0, t1 = rows[i0][1], rows[i1][1]
win = [r for r in rows if r[1] >= t0 and r[1] < t1]
wall = (t1 - t0) / steps / 1e3
print("window: %d steps, %d launches / step, wall %.1f us / step" % (steps, len(win) / steps, wall))
# sweep
ev = []
for k, (n, s, e, q) in enumerate(win):
    ev.append((s, 1, k)); ev.append((min(e, t1), -1, k))
ev.sort()
depth = 0
live = set()
last = t0
by_depth = collections.Counter()
alone = collections.Counter()
gaps = []
prev_end_name = None
for t, d, k in ev:
    dt = t - last
    if dt > 0:
        by_depth[min(depth, 3)] += dt
        if depth == 1:
            alone[win[next(iter(live))][0]] += dt
        if depth == 0 and dt > 0:
            gaps.append((dt, prev_end_name, win[k][0] if d == 1 else None))
    if d == 1:
        depth += 1; live.add(k)
    else:
        depth -= 1; live.discard(k); prev_end_name = win[k][0]
    last = t
tot = sum(by_depth.values())
print("per step: idle %.1f us (%.1f %%), 1 kernel %.1f us, 2 kernels %.1f us, 3+ %.1f us" % tuple(
    [by_depth[0] / steps / 1e3, 100.0 * by_depth[0] / tot] + [by_depth[i] / steps / 1e3 for i in (1, 2, 3)]))
h = collections.Counter()
for g, a, b in gaps:
    h["<2us" if g < 2000 else "<5us" if g < 5000 else "<10us" if g < 10000 else "<20us" if g < 20000 else ">=20us"] += g
print("idle by gap size (us / step):", {k: round(v / steps / 1e3, 1) for k, v in h.items()}, "gaps / step:", len(gaps) / steps)
qs = collections.defaultdict(lambda: [0, 0])
for n, s, e, q in win:
    qs[q][0] += e - s; qs[q][1] += 1
print("queues:", {q: "%.0f us, %d launches / step" % (v[0] / steps / 1e3, v[1] / steps) for q, v in qs.items()})
print("kernels running ALONE (us / step):")
for n, v in alone.most_common(25):
    print("  %8.1f  %s" % (v / steps / 1e3, n[:110]))
print("largest idle gaps (us): after -> before")
for g, a, b in sorted(gaps, reverse=True)[:15]:
    print("  %7.1f  %s -> %s" % (g / 1e3, (a or "?")[:60], (b or "?")[:60]))
ksum = collections.Counter(); kcnt = collections.Counter()
for n, s, e, q in win:
    ksum[n] += e - s; kcnt[n] += 1
print("kernel time sum %.1f us / step; top:" % (sum(ksum.values()) / steps / 1e3))
for n, v in ksum.most_common(14):
    print("  %8.1f us  %5.1f x %7.1f us  %s" % (v / steps / 1e3, kcnt[n] / steps, v / kcnt[n] / 1e3, n[:100]))

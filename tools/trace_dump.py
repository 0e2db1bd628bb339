"""Dump the kernel dispatch timeline of a rocprofv3 rocpd .db (rocprofv3 --kernel-trace) to CSV: name, start_ns, end_ns, queue, stream —
the input of tools/trace_gaps.py (GPU busy union, idle gaps, per-queue occupancy of a training step with the stream lanes on).

    python tools/trace_dump.py <file.db> <out.csv>"""
import csv
import sqlite3
import sys

db, out = sys.argv[1], sys.argv[2]
con = sqlite3.connect(db)
cur = con.cursor()
objs = list(cur.execute("select type, name from sqlite_master where type in ('table', 'view')"))
cands = []
for typ, name in objs:
    try:
        cols = [r[1] for r in cur.execute("pragma table_info('%s')" % name)]
    except sqlite3.Error:
        continue
    low = [c.lower() for c in cols]
    if "start" in low and "end" in low and any("kernel" in c or c == "name" for c in low):
        cands.append((name, cols))
if not cands:
    print("no kernel timeline object found; objects:", objs)
    sys.exit(1)
# prefer the `kernels` view of the rocpd schema
cands.sort(key=lambda nc: (0 if nc[0] == "kernels" else 1 if "kernel" in nc[0] else 2, len(nc[0])))
name, cols = cands[0]
print("using", name, cols)
low = {c.lower(): c for c in cols}
ncol = low.get("name") or low.get("kernel_name") or [c for c in cols if "name" in c.lower()][0]
qcol = low.get("queue_id") or low.get("queue") or "0"
scol = low.get("stream_id") or low.get("stream") or "0"
rows = list(cur.execute("select %s, %s, %s, %s, %s from %s order by %s" % (ncol, low["start"], low["end"], qcol, scol, name, low["start"])))
with open(out, "w", newline="") as f:
    w = csv.writer(f)
    w.writerow(["name", "start_ns", "end_ns", "queue", "stream"])
    for r in rows:
        n = str(r[0])
        p = n.find("(")
        w.writerow([n[:p] if p > 0 else n, r[1], r[2], r[3], r[4]])
print("wrote", out, len(rows), "dispatches")
